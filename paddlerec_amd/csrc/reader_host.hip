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
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "rec_common.h"

namespace rec {

struct LineSpan { const char* b; const char* e; };

static void split_lines(const char* buf, size_t len, int64_t max_lines, std::vector<LineSpan>* out) {
  const char* p = buf;
  const char* end = buf + len;
  while (p < end && (int64_t)out->size() < max_lines) {
    const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* e = nl ? nl : end;
    const char* te = e;
    while (te > p && (te[-1] == '\r' || te[-1] == ' ' || te[-1] == '\t')) --te;   // l.strip() (right side)
    const char* tb = p;
    while (tb < te && (*tb == ' ' || *tb == '\t')) ++tb;                           // l.strip() (left side)
    if (te > tb || nl) out->push_back({tb, te});   // python iterates every line, blank ones too
    p = nl ? nl + 1 : end;
  }
}

template <class F>
static void parallel_lines(int64_t n, int threads, F&& f) {
  if (threads <= 1 || n < 256) { f(0, n); return; }
  std::vector<std::thread> pool;
  const int64_t per = (n + threads - 1) / threads;
  for (int t = 0; t < threads; ++t) {
    const int64_t lo = t * per, hi = lo + per < n ? lo + per : n;
    if (lo >= hi) break;
    pool.emplace_back([=, &f]() { f(lo, hi); });
  }
  for (auto& th : pool) th.join();
}

static int host_threads(int requested) {
  if (requested > 0) return requested;
  unsigned hc = std::thread::hardware_concurrency();
  int t = hc ? (int)hc : 4;
  return t > 32 ? 32 : t;
}

static void parse_slot_line(LineSpan ln, int S, int Dn, bool log1p_dense, int64_t* label, int64_t* ids,
                            float* dense) {
  bool have_label = false;
  int n_dense = 0;
  for (int s = 0; s < S; ++s) ids[s] = 0;       // padding for slots that never show up
  std::vector<char> seen((size_t)S, 0);
  *label = 0;
  const char* p = ln.b;
  while (p < ln.e) {
    const char* te = (const char*)memchr(p, ' ', (size_t)(ln.e - p));
    if (!te) te = ln.e;
    const char* colon = (const char*)memchr(p, ':', (size_t)(te - p));
    if (colon) {
      const size_t nl = (size_t)(colon - p);
      if (nl == 5 && memcmp(p, "click", 5) == 0) {
        if (!have_label) { *label = strtoll(colon + 1, nullptr, 10); have_label = true; }
      } else if (nl == 13 && memcmp(p, "dense_feature", 13) == 0) {
        if (n_dense < Dn) {
          double v = strtod(colon + 1, nullptr);
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
        if (digits && slot >= 1 && slot <= S && !seen[slot - 1]) {   // first value of the slot
          ids[slot - 1] = strtoll(colon + 1, nullptr, 10);
          seen[slot - 1] = 1;
        }
      }
    }
    p = te < ln.e ? te + 1 : ln.e;
  }
  for (int j = n_dense; j < Dn; ++j) dense[j] = 0.f;   // missing dense slot -> zeros
}

}  // namespace rec

using namespace rec;

extern "C" int rec_parse_slot_text(const char* buf, size_t len, int32_t n_sparse, int32_t n_dense,
                                   int32_t log1p_dense, int64_t max_lines, int32_t threads,
                                   int64_t* label, int64_t* ids, float* dense, int64_t* n_lines) {
  REC_REQUIRE(n_sparse > 0 && n_dense >= 0 && max_lines >= 0 && n_lines, REC_EINVAL, "bad arguments");
  REC_REQUIRE(len == 0 || buf, REC_EINVAL, "buf is NULL");
  REC_REQUIRE(max_lines == 0 || (label && ids && (n_dense == 0 || dense)), REC_EINVAL,
              "null output pointer");
  std::vector<LineSpan> lines;
  split_lines(buf, len, max_lines, &lines);
  const int64_t n = (int64_t)lines.size();
  parallel_lines(n, host_threads(threads), [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i)
      parse_slot_line(lines[i], n_sparse, n_dense, log1p_dense != 0, label + i, ids + i * n_sparse,
                      dense + i * n_dense);
  });
  *n_lines = n;
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
  std::vector<LineSpan> lines;
  {
    const char* p = buf;
    const char* end = buf + len;
    while (p < end && (int64_t)lines.size() < max_lines) {
      const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
      const char* e = nl ? nl : end;
      lines.push_back({p, e});
      p = nl ? nl + 1 : end;
    }
  }
  const int64_t n = (int64_t)lines.size();
  parallel_lines(n, host_threads(threads), [&](int64_t lo, int64_t hi) {
    std::string key;
    for (int64_t i = lo; i < hi; ++i) {
      const char* p = lines[i].b;
      const char* e = lines[i].e;
      for (int f = 0; f < 1 + n_dense + n_sparse; ++f) {
        const char* te = (p <= e) ? (const char*)memchr(p, '\t', (size_t)(e - p)) : nullptr;
        if (!te) te = e;
        const size_t fl = p <= e ? (size_t)(te - p) : 0;
        if (f == 0) {
          label[i] = fl ? strtoll(p, nullptr, 10) : 0;
        } else if (f <= n_dense) {
          const int j = f - 1;     // benchmark_reader.py:44-49: "" -> 0.0 else (float(x) - min) / diff
          double v = 0.0;
          if (fl) v = (strtod(std::string(p, fl).c_str(), nullptr) - (double)cont_min[j]) / (double)cont_diff[j];
          dense[i * n_dense + j] = (float)v;
        } else {
          const int s = f - 1 - n_dense;   // benchmark_reader.py:50-53: xxh32(str(idx) + features[idx]) % hash_dim
          key = std::to_string(f);
          key.append(p, fl);
          ids[i * n_sparse + s] = (int64_t)(rec_xxh32(key.data(), key.size(), 0) % hash_dim);
        }
        p = te + 1;
      }
    }
  });
  *n_lines = n;
  return REC_OK;
}
