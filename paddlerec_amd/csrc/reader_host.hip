// Input pipeline, HOST side: text -> (label, ids, dense) batches in caller-owned (pinned) buffers.
//
// Replaces the per-line Python of the reference's readers ("next" rank 2):
//   rec_parse_slot_text   <- models/rank/deepfm/criteo_reader.py:61-103 (same code dcn_v2/reader.py:41-89,
//                            which additionally applies log(x+1) to dense values, reader.py:63-64)
//                            line = "click:L dense_feature:v x13 1:id ... 26:id"; a missing sparse slot is
//                            padded with id 0, a missing dense slot with zeros (criteo_reader.py:80-91)
//   rec_parse_criteo_tsv  <- models/rank/dnn/benchmark_reader.py:39-54: raw Criteo "label \t 13 ints \t 26
//                            strings"; dense = (x - min)/diff, sparse = xxh32(str(idx)+feat) % hash_dim
// Both split the buffer at line boundaries over host threads (the reference forks reader subprocesses,
// tools/utils/static_ps/reader_helper.py:283-308); outputs are written in line order, so results do not
// depend on the thread count.  No device code here: the caller copies the batch with hipMemcpyAsync.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "rec_common.h"

namespace rec {

struct LineSpan { const char* b; const char* e; };

// Splits [buf, buf+len) into T byte ranges that end on line boundaries; range t holds lines
// [first_line[t], first_line[t+1]).  Counting and (later) parsing both run one thread per range, so the
// whole pipeline scales with the host cores (no serial line index).
struct Chunks {
  std::vector<const char*> begin;     // T+1 pointers
  std::vector<int64_t> first_line;    // T+1 line numbers
};

template <class F>
static void run_threads(int T, F&& f) {
  if (T <= 1) { f(0); return; }
  std::vector<std::thread> pool;
  for (int t = 0; t < T; ++t) pool.emplace_back([=, &f]() { f(t); });
  for (auto& th : pool) th.join();
}

static Chunks make_chunks(const char* buf, size_t len, int T) {
  Chunks c;
  const char* end = buf + len;
  if (len < (size_t)T * 4096) T = 1;
  c.begin.resize(T + 1);
  c.begin[0] = buf;
  for (int t = 1; t < T; ++t) {
    const char* p = buf + len / T * t;
    if (p < c.begin[t - 1]) p = c.begin[t - 1];
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    c.begin[t] = nl ? nl + 1 : end;
  }
  c.begin[T] = end;
  std::vector<int64_t> cnt((size_t)T, 0);
  run_threads(T, [&](int t) {
    int64_t n = 0;
    const char* p = c.begin[t];
    const char* e = c.begin[t + 1];
    while (p < e) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
      ++n;                              // a trailing fragment without '\n' is a line too
      p = nl ? nl + 1 : e;
    }
    cnt[t] = n;
  });
  c.first_line.resize(T + 1);
  c.first_line[0] = 0;
  for (int t = 0; t < T; ++t) c.first_line[t + 1] = c.first_line[t] + cnt[t];
  return c;
}

// calls f(line_index, LineSpan) for every line (raw: [p, newline) without trimming), in parallel
template <class F>
static int64_t for_each_line(const char* buf, size_t len, int64_t max_lines, int threads, F&& f) {
  const Chunks c = make_chunks(buf, len, threads);
  const int T = (int)c.begin.size() - 1;
  run_threads(T, [&](int t) {
    int64_t i = c.first_line[t];
    const char* p = c.begin[t];
    const char* e = c.begin[t + 1];
    while (p < e && i < max_lines) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
      const char* le = nl ? nl : e;
      f(i, LineSpan{p, le});
      ++i;
      p = nl ? nl + 1 : e;
    }
  });
  const int64_t total = c.first_line[T];
  return total < max_lines ? total : max_lines;
}

static int host_threads(int requested) {
  if (requested > 0) return requested;
  unsigned hc = std::thread::hardware_concurrency();
  int t = hc ? (int)hc : 4;
  return t > 64 ? 64 : t;
}

// ---- number fields: a bounded fast path for the forms the datasets hold, the C library for everything else.
// Both see exactly [p, e): the fallbacks run on a NUL-terminated copy (the input may be an mmap that ends without one).
static double strtod_span(const char* p, const char* e) {
  char tmp[96];
  const size_t n = (size_t)(e - p);
  if (n < sizeof(tmp)) {
    memcpy(tmp, p, n);
    tmp[n] = 0;
    return strtod(tmp, nullptr);
  }
  return strtod(std::string(p, n).c_str(), nullptr);
}
static long long strtoll_span(const char* p, const char* e) {
  char tmp[96];
  const size_t n = (size_t)(e - p);
  if (n < sizeof(tmp)) {
    memcpy(tmp, p, n);
    tmp[n] = 0;
    return strtoll(tmp, nullptr, 10);
  }
  return strtoll(std::string(p, n).c_str(), nullptr, 10);
}
// "[-]digits[.digits]" with at most 15 significant digits: the integer of all digits and the power of ten are both
// exact doubles, so ONE division gives the correctly rounded value — the same double strtod returns (Clinger's fast
// path).  Anything else (exponents, inf/nan, 16+ digits, trailing characters) goes to strtod.
static inline double parse_double(const char* p, const char* e) {
  static const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                    1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  const char* q = p;
  bool neg = false;
  if (q < e && *q == '-') { neg = true; ++q; }
  uint64_t w = 0;
  int sig = 0, frac = 0, digits = 0;
  for (; q < e; ++q) {
    const unsigned d = (unsigned)(*q - '0');
    if (d > 9) break;
    w = w * 10 + d;
    sig += (sig > 0 || d != 0);
    ++digits;
  }
  if (q < e && *q == '.') {
    for (++q; q < e; ++q) {
      const unsigned d = (unsigned)(*q - '0');
      if (d > 9) break;
      w = w * 10 + d;
      sig += (sig > 0 || d != 0);
      ++digits;
      ++frac;
    }
  }
  if (q != e || digits == 0 || sig > 15 || frac > 22 || digits > 19) return strtod_span(p, e);
  const double v = (double)w / kPow10[frac];
  return neg ? -v : v;
}
// "[-]digits", at most 18 of them
static inline long long parse_int(const char* p, const char* e) {
  const char* q = p;
  bool neg = false;
  if (q < e && *q == '-') { neg = true; ++q; }
  const size_t n = (size_t)(e - q);
  if (n == 0 || n > 18) return strtoll_span(p, e);
  long long v = 0;
  for (; q < e; ++q) {
    const unsigned d = (unsigned)(*q - '0');
    if (d > 9) return strtoll_span(p, e);
    v = v * 10 + d;
  }
  return neg ? -v : v;
}

static void parse_slot_line(LineSpan ln, int S, int Dn, bool log1p_dense, int64_t* label, int64_t* ids,
                            float* dense) {
  bool have_label = false;
  int n_dense = 0;
  for (int s = 0; s < S; ++s) ids[s] = 0;       // padding for slots that never show up
  uint64_t seen_bits = 0;                        // slots 1..64; beyond that (no such config) a heap bitmap
  std::vector<char> seen_more;
  if (S > 64) seen_more.assign((size_t)S, 0);
  *label = 0;
  const char* p = ln.b;
  while (p < ln.e) {
    // one pass over the token: its end and its first ':'
    const char* te = p;
    const char* colon = nullptr;
    for (; te < ln.e && *te != ' '; ++te)
      if (*te == ':' && !colon) colon = te;
    if (colon) {
      const size_t nl = (size_t)(colon - p);
      if (nl == 5 && memcmp(p, "click", 5) == 0) {
        if (!have_label) { *label = parse_int(colon + 1, te); have_label = true; }
      } else if (nl == 13 && memcmp(p, "dense_feature", 13) == 0) {
        if (n_dense < Dn) {
          double v = parse_double(colon + 1, te);
          if (log1p_dense) v = log(v + 1.0);          // dcn_v2/reader.py:63-64  np.log(feasign + 1)
          dense[n_dense++] = (float)v;
        }
      } else if (nl > 0 && nl <= 9) {
        bool digits = p[0] != '0';      // slot names are "1".."26" verbatim (criteo_reader.py:49-50)
        int slot = 0;
        for (size_t i = 0; i < nl; ++i) {
          if (p[i] < '0' || p[i] > '9') { digits = false; break; }
          slot = slot * 10 + (p[i] - '0');
        }
        if (digits && slot >= 1 && slot <= S) {
          const bool seen = S > 64 ? seen_more[(size_t)slot - 1] != 0 : ((seen_bits >> (slot - 1)) & 1) != 0;
          if (!seen) {                                                   // first value of the slot
            ids[slot - 1] = parse_int(colon + 1, te);
            if (S > 64) seen_more[(size_t)slot - 1] = 1;
            else seen_bits |= 1ull << (slot - 1);
          }
        }
      }
    }
    p = te < ln.e ? te + 1 : ln.e;
  }
  for (int j = n_dense; j < Dn; ++j) dense[j] = 0.f;   // missing dense slot -> zeros
}

}  // namespace rec

using namespace rec;

extern "C" int rec_count_lines(const char* buf, size_t len, int32_t threads, int64_t* n_lines) {
  REC_REQUIRE(n_lines && (len == 0 || buf), REC_EINVAL, "bad arguments");
  const Chunks c = make_chunks(buf, len, host_threads(threads));
  *n_lines = c.first_line.back();
  return REC_OK;
}

// One batch cut out of whole-file parses (rec_parse_feasign_slots outputs): pieces = consecutive line ranges
// [l0, l1) of parsed files; the batch's slot s holds its ids of piece 0, then of piece 1, ... — exactly the
// slot-major CSR a parse of those lines on their own would give.  out_lod [num_slots, n_lines + 1],
// out_base [num_slots + 1], out_values [sum of the ranges] (the caller sizes it from the lod differences).
extern "C" int rec_csr_cut(int32_t num_slots, int32_t n_pieces, const int64_t* const* values,
                           const int64_t* const* lod, const int64_t* lod_stride, const int64_t* const* base,
                           const int64_t* l0, const int64_t* l1, int32_t threads, int64_t* out_values,
                           int64_t max_values, int64_t* out_lod, int64_t* out_base) {
  REC_REQUIRE(num_slots > 0 && n_pieces > 0 && values && lod && lod_stride && base && l0 && l1 && out_values &&
                  out_lod && out_base, REC_EINVAL, "bad arguments");
  const int S = num_slots, P = n_pieces;
  int64_t nl = 0;
  for (int p = 0; p < P; ++p) {
    REC_REQUIRE(l1[p] >= l0[p] && l0[p] >= 0, REC_EINVAL, "bad line range of piece %d", p);
    nl += l1[p] - l0[p];
  }
  // slot bases of the batch
  int64_t tot = 0;
  for (int s = 0; s < S; ++s) {
    out_base[s] = tot;
    for (int p = 0; p < P; ++p) {
      const int64_t* l = lod[p] + (size_t)s * lod_stride[p];
      tot += l[l1[p]] - l[l0[p]];
    }
  }
  out_base[S] = tot;
  REC_REQUIRE(tot <= max_values, REC_EWORKSPACE, "values buffer holds %lld ids, %lld needed", (long long)max_values,
              (long long)tot);
  const int T = tot < 200000 ? 1 : host_threads(threads);     // a small batch is copied faster than threads start
  run_threads(T < S ? T : S, [&](int t) {
    const int TT = T < S ? T : S;
    for (int s = t; s < S; s += TT) {
      int64_t* ol = out_lod + (size_t)s * (nl + 1);
      int64_t* ov = out_values + out_base[s];
      int64_t line = 0, at = 0;
      ol[0] = 0;
      for (int p = 0; p < P; ++p) {
        const int64_t* l = lod[p] + (size_t)s * lod_stride[p];
        const int64_t b0 = l[l0[p]], b1 = l[l1[p]];
        memcpy(ov + at, values[p] + base[p][s] + b0, (size_t)(b1 - b0) * sizeof(int64_t));
        for (int64_t i = l0[p]; i < l1[p]; ++i) ol[++line] = at + (l[i + 1] - b0);
        at += b1 - b0;
      }
    }
  });
  return REC_OK;
}

// occurrences of one byte value (the ':' of every `feasign:slot` token bounds the value count of a multi-value parse)
extern "C" int rec_count_byte(const char* buf, size_t len, int32_t byte, int32_t threads, int64_t* n) {
  REC_REQUIRE(n && (len == 0 || buf) && byte >= 0 && byte < 256, REC_EINVAL, "bad arguments");
  const int T = host_threads(threads);
  std::vector<int64_t> cnt((size_t)T, 0);
  const size_t per = (len + (size_t)T - 1) / (size_t)T;
  run_threads(T, [&](int t) {
    const size_t a = (size_t)t * per, z = a + per < len ? a + per : len;
    int64_t c = 0;
    const char* p = a < len ? buf + a : nullptr;
    const char* e = buf + z;
    while (p && p < e) {
      p = (const char*)memchr(p, byte, (size_t)(e - p));
      if (!p) break;
      ++c;
      ++p;
    }
    cnt[(size_t)t] = c;
  });
  int64_t tot = 0;
  for (int64_t c : cnt) tot += c;
  *n = tot;
  return REC_OK;
}

// indices (ascending) of the lines that hold nothing but blanks: the reference's readers iterate `for line in f` and
// the loaders built on the whole-file parsers skip such lines instead of treating them as samples
extern "C" int rec_blank_lines(const char* buf, size_t len, int32_t threads, int64_t max_out, int64_t* idx,
                               int64_t* n_blank) {
  REC_REQUIRE(n_blank && max_out >= 0 && (len == 0 || buf) && (max_out == 0 || idx), REC_EINVAL, "bad arguments");
  const int T = host_threads(threads);
  std::vector<std::vector<int64_t>> found((size_t)T + 1);
  const Chunks c = make_chunks(buf, len, T);
  const int nt = (int)c.begin.size() - 1;
  run_threads(nt, [&](int t) {
    int64_t i = c.first_line[t];
    const char* p = c.begin[t];
    const char* e = c.begin[t + 1];
    while (p < e) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
      const char* le = nl ? nl : e;
      bool blank = true;
      for (const char* q = p; q < le; ++q)
        if (*q != ' ' && *q != '\t' && *q != '\r') { blank = false; break; }
      if (blank) found[(size_t)t].push_back(i);
      ++i;
      p = nl ? nl + 1 : e;
    }
  });
  int64_t n = 0;
  for (int t = 0; t < nt; ++t)
    for (int64_t v : found[(size_t)t]) {
      if (n < max_out) idx[n] = v;
      ++n;
    }
  *n_blank = n;
  return REC_OK;
}

extern "C" int rec_parse_slot_text(const char* buf, size_t len, int32_t n_sparse, int32_t n_dense,
                                   int32_t log1p_dense, int64_t max_lines, int32_t threads,
                                   int64_t* label, int64_t* ids, float* dense, int64_t* n_lines) {
  REC_REQUIRE(n_sparse > 0 && n_dense >= 0 && max_lines >= 0 && n_lines, REC_EINVAL, "bad arguments");
  REC_REQUIRE(len == 0 || buf, REC_EINVAL, "buf is NULL");
  REC_REQUIRE(max_lines == 0 || (label && ids && (n_dense == 0 || dense)), REC_EINVAL,
              "null output pointer");
  *n_lines = for_each_line(buf, len, max_lines, host_threads(threads), [&](int64_t i, LineSpan ln) {
    while (ln.e > ln.b && (ln.e[-1] == '\r' || ln.e[-1] == ' ' || ln.e[-1] == '\t')) --ln.e;   // l.strip()
    while (ln.b < ln.e && (*ln.b == ' ' || *ln.b == '\t')) ++ln.b;
    parse_slot_line(ln, n_sparse, n_dense, log1p_dense != 0, label + i, ids + i * n_sparse,
                    dense + i * n_dense);
  });
  return REC_OK;
}

extern "C" int rec_parse_criteo_tsv(const char* buf, size_t len, int32_t n_dense, int32_t n_sparse,
                                    const float* cont_min, const float* cont_diff, uint32_t hash_dim,
                                    int64_t max_lines, int32_t threads, int64_t* label, int64_t* ids,
                                    float* dense, int64_t* n_lines) {
  REC_REQUIRE(n_sparse > 0 && n_dense >= 0 && hash_dim > 0 && max_lines >= 0 && n_lines, REC_EINVAL,
              "bad arguments");
  REC_REQUIRE(len == 0 || buf, REC_EINVAL, "buf is NULL");
  REC_REQUIRE(n_dense == 0 || (cont_min && cont_diff), REC_EINVAL, "cont_min/cont_diff missing");
  REC_REQUIRE(max_lines == 0 || (label && ids && (n_dense == 0 || dense)), REC_EINVAL,
              "null output pointer");
  // lines are split on '\n' only: fields may be empty and are tab separated (line.rstrip('\n').split('\t'))
  // str(idx) of every field index, once per call
  const int n_fields = 1 + n_dense + n_sparse;
  std::vector<char> prefix((size_t)n_fields * 12);
  std::vector<int> prefix_len((size_t)n_fields);
  for (int f = 0; f < n_fields; ++f)
    prefix_len[(size_t)f] = snprintf(&prefix[(size_t)f * 12], 12, "%d", f);
  *n_lines = for_each_line(buf, len, max_lines, host_threads(threads), [&](int64_t i, LineSpan ln) {
    const char* p = ln.b;
    const char* e = ln.e;
    for (int f = 0; f < n_fields; ++f) {
      const char* te = p;                    // fields are a few bytes long: a byte loop beats a memchr call
      if (p <= e) while (te < e && *te != '\t') ++te;
      const size_t fl = p <= e ? (size_t)(te - p) : 0;
      if (f == 0) {
        label[i] = fl ? strtoll_span(p, te) : 0;       // int(features[0]): leading blanks and '+' allowed
      } else if (f <= n_dense) {
        const int j = f - 1;     // benchmark_reader.py:44-49: "" -> 0.0 else (float(x) - min) / diff
        double v = 0.0;
        if (fl) v = (parse_double(p, te) - (double)cont_min[j]) / (double)cont_diff[j];
        dense[i * n_dense + j] = (float)v;
      } else {
        const int s = f - 1 - n_dense;   // benchmark_reader.py:50-53: xxh32(str(idx) + features[idx]) % hash_dim
        char key[128];
        const int pl = prefix_len[(size_t)f];
        uint32_t h;
        if (fl + (size_t)pl <= sizeof(key)) {
          memcpy(key, &prefix[(size_t)f * 12], (size_t)pl);
          memcpy(key + pl, p, fl);
          h = rec_xxh32(key, (size_t)pl + fl, 0);
        } else {
          std::string k(&prefix[(size_t)f * 12], (size_t)pl);
          k.append(p, fl);
          h = rec_xxh32(k.data(), k.size(), 0);
        }
        ids[i * n_sparse + s] = (int64_t)(h % hash_dim);
      }
      p = te + 1;
    }
  });
  return REC_OK;
}

// ------------------------------------------------------------------------------------------------------
// Multi-value slot lines — models/rank/slot_dnn/queuedataset_reader.py:56-82 (Reader.line_process): a line is
// "feasign:slot feasign:slot ..." with uint64 feasigns and any number of values per slot; slots outside
// [first_slot, first_slot + num_slots) are dropped, a slot that never shows up gets the single padding id 0.
// Output = what the sum-pool lookup (rec_emb_gather_sumpool, one launch per slot) consumes: slot-major CSR
//   values[slot_base[s] + lod[s][b] + j]   (j-th value of slot s in line b),   lod[s][0..n_lines]
// in two parallel passes over the lines (count, then fill) around one prefix sum per slot.
namespace rec {

// calls f(slot_index, feasign) for every "feasign:slot" token of the line whose slot is in range
template <class F>
static void scan_feasign_line(LineSpan ln, int first_slot, int num_slots, F&& f) {
  const char* p = ln.b;
  const char* const e = ln.e;
  const long lo = first_slot, hi = (long)first_slot + num_slots;
  while (p < e) {
    if (*p == ' ') { ++p; continue; }
    // one pass over the token; the only accepted form is digits ':' digits up to the next blank
    const char* q = p;
    uint64_t fs = 0;
    for (; q < e; ++q) {
      const unsigned d = (unsigned)(*q - '0');
      if (d > 9) break;
      fs = fs * 10u + d;
    }
    bool ok = q > p && q < e && *q == ':';
    long slot = 0;
    if (ok) {
      const char* const r = ++q;
      for (; q < e; ++q) {
        const unsigned d = (unsigned)(*q - '0');
        if (d > 9) break;
        slot = slot * 10 + (long)d;
        if (slot > (1l << 30)) { ok = false; break; }
      }
      ok = ok && q > r && (q == e || *q == ' ');
    }
    while (q < e && *q != ' ') ++q;        // the rest of a malformed token
    if (ok && slot >= lo && slot < hi) f((int)(slot - lo), fs);
    p = q;
  }
}

static inline void strip_line(LineSpan& ln) {
  while (ln.e > ln.b && (ln.e[-1] == '\r' || ln.e[-1] == ' ' || ln.e[-1] == '\t')) --ln.e;
  while (ln.b < ln.e && (*ln.b == ' ' || *ln.b == '\t')) ++ln.b;
}

}  // namespace rec

extern "C" int rec_parse_feasign_slots(const char* buf, size_t len, int32_t first_slot, int32_t num_slots,
                                       uint64_t hash_rows, int64_t max_lines, int64_t max_values,
                                       int32_t threads, int64_t* values, int64_t* lod, int64_t* slot_base,
                                       int64_t* n_lines, int64_t* n_values) {
  REC_REQUIRE(num_slots > 0 && first_slot >= 0 && max_lines >= 0 && max_values >= 0 && n_lines && n_values,
              REC_EINVAL, "bad arguments");
  REC_REQUIRE(hash_rows == 0 || hash_rows >= 2, REC_EINVAL, "hash_rows must be 0 (raw feasigns) or >= 2");
  REC_REQUIRE(len == 0 || buf, REC_EINVAL, "buf is NULL");
  REC_REQUIRE(max_lines == 0 || (lod && slot_base), REC_EINVAL, "null output pointer");
  const int T = host_threads(threads);
  const int S = num_slots;
  // pass 1: values per (line, slot); a slot without a token holds the one padding id.  Unless the text is huge, the
  // (slot, feasign) pairs of the accepted tokens are kept per chunk, in line order, so that pass 2 does not tokenize the
  // text a second time (tokenizing is ~8.8 us of a 34 us line: profiles/r02f_reader_host.txt); ~12 B per token.
  std::vector<int32_t> cnt((size_t)max_lines * S, 0);
  const Chunks ck = make_chunks(buf, len, T);
  const int nt = (int)ck.begin.size() - 1;
  const char* cache_env = getenv("REC_FEASIGN_CACHE");
  const bool cache = !(cache_env && *cache_env == '0') && len <= ((size_t)1 << 30);
  struct ChunkTokens {
    std::vector<uint64_t> fs;
    std::vector<int32_t> slot, per_line;
  };
  std::vector<ChunkTokens> toks((size_t)(cache ? nt : 0));
  run_threads(nt, [&](int t) {
    int64_t i = ck.first_line[t];
    const char* p = ck.begin[t];
    const char* e = ck.begin[t + 1];
    if (cache) {
      toks[(size_t)t].fs.reserve((size_t)(e - p) / 16);
      toks[(size_t)t].slot.reserve((size_t)(e - p) / 16);
    }
    while (p < e && i < max_lines) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(e - p));
      LineSpan ln{p, nl ? nl : e};
      strip_line(ln);
      int32_t* c = cnt.data() + (size_t)i * S;
      if (cache) {
        ChunkTokens& tk = toks[(size_t)t];
        const size_t before = tk.fs.size();
        scan_feasign_line(ln, first_slot, S, [&](int s, uint64_t fs) {
          ++c[s];
          tk.fs.push_back(fs);
          tk.slot.push_back(s);
        });
        tk.per_line.push_back((int32_t)(tk.fs.size() - before));
      } else {
        scan_feasign_line(ln, first_slot, S, [&](int s, uint64_t) { ++c[s]; });
      }
      for (int s = 0; s < S; ++s)
        if (c[s] == 0) c[s] = -1;            // -1 = padded (one value, id 0)
      ++i;
      p = nl ? nl + 1 : e;
    }
  });
  const int64_t n = ck.first_line[nt] < max_lines ? ck.first_line[nt] : max_lines;
  *n_lines = n;
  // per-slot offsets over the lines (parallel over slots), then the slot bases
  std::vector<int64_t> tot((size_t)S, 0);
  run_threads(T < S ? T : S, [&](int t) {
    const int TT = T < S ? T : S;
    for (int s = t; s < S; s += TT) {
      int64_t acc = 0;
      int64_t* l = lod + (size_t)s * (max_lines + 1);
      l[0] = 0;
      for (int64_t b = 0; b < n; ++b) {
        const int32_t c = cnt[(size_t)b * S + s];
        acc += c < 0 ? 1 : c;
        l[b + 1] = acc;
      }
      tot[s] = acc;
    }
  });
  int64_t total = 0;
  for (int s = 0; s < S; ++s) { slot_base[s] = total; total += tot[s]; }
  slot_base[S] = total;
  *n_values = total;
  REC_REQUIRE(total <= max_values, REC_EWORKSPACE, "values buffer holds %lld ids, %lld needed",
              (long long)max_values, (long long)total);
  if (total == 0) return REC_OK;
  REC_REQUIRE(values, REC_EINVAL, "values is NULL");
  // pass 2: fill
  auto table_row = [&](uint64_t fs) -> int64_t {
    // hashed table: row 0 stays the padding row, every other feasign lands in [1, hash_rows)
    return (int64_t)(hash_rows == 0 ? fs : (fs == 0 ? 0 : 1 + fs % (hash_rows - 1)));
  };
  if (cache) {
    run_threads(nt, [&](int t) {
      const ChunkTokens& tk = toks[(size_t)t];
      std::vector<int32_t> cur((size_t)S);      // values of the line already placed, per slot
      size_t k = 0;
      int64_t i = ck.first_line[t];
      for (size_t li = 0; li < tk.per_line.size(); ++li, ++i) {
        std::fill(cur.begin(), cur.end(), 0);
        for (int32_t q = 0; q < tk.per_line[li]; ++q, ++k) {
          const int s = tk.slot[k];
          values[slot_base[s] + lod[(size_t)s * (max_lines + 1) + i] + cur[(size_t)s]++] = table_row(tk.fs[k]);
        }
        const int32_t* c = cnt.data() + (size_t)i * S;
        for (int s = 0; s < S; ++s)
          if (c[s] < 0) values[slot_base[s] + lod[(size_t)s * (max_lines + 1) + i]] = 0;
      }
    });
    return REC_OK;
  }
  for_each_line(buf, len, n, T, [&](int64_t i, LineSpan ln) {
    strip_line(ln);
    thread_local std::vector<int32_t> cur;        // values of this line already placed, per slot
    cur.assign((size_t)S, 0);
    scan_feasign_line(ln, first_slot, S, [&](int s, uint64_t fs) {
      const int64_t at = slot_base[s] + lod[(size_t)s * (max_lines + 1) + i] + cur[s]++;
      values[at] = table_row(fs);            // raw mode: the uint64 bit pattern, as the reference feeds int64
    });
    const int32_t* c = cnt.data() + (size_t)i * S;
    for (int s = 0; s < S; ++s)
      if (c[s] < 0) values[slot_base[s] + lod[(size_t)s * (max_lines + 1) + i]] = 0;
  });
  return REC_OK;
}
