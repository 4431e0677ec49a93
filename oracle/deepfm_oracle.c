/* CPU restatement in C of the DeepFM embedding + FM forward/backward and the lazy sparse Adam
 * (oracle — TEST INFRASTRUCTURE and bench.py's cpu_baseline leg ONLY; never linked into the
 * product library).
 *
 * Follows, line by line:
 *   fm_fwd   : /root/reference/models/rank/deepfm/net.py:105-139   (FM.forward)
 *   fm_bwd   : autograd of the same lines (tools/trainer.py:151 loss.backward())
 *   adam_rows: paddle.optimizer.Adam(lazy_mode=True) [EXT], deepfm/static_model.py:83-84,
 *              formula SURVEY.md Appendix B-3
 * Parity: unpinned at the Paddle-kernel boundary (see oracle/__init__.py); this file is checked
 * against oracle/deepfm_ref.py (NumPy) and tests/golden/ in tests/test_oracle.py.
 *
 * Build: make -C oracle   ->  oracle/_build/liboracle.so   (gcc -O3 -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n);
#else
  (void)n;
#endif
}

/* row id of (b,s) or -1 for a padding hit.  slot_off may be NULL. */
static inline int64_t eff_row(const int64_t* ids, int64_t b, int s, int S, int64_t pad,
                              const int64_t* slot_off) {
  int64_t id = ids[b * S + s];
  if (pad >= 0 && id == pad) return -1;
  return slot_off ? id + slot_off[s] : id;
}

/* net.py:105-139.  feat [B,F,D], sum_emb [B,D] (saved for backward), y1,y2 [B]. */
void oracle_fm_fwd(int64_t B, int S, int Dn, int D, const int64_t* ids, const float* dense,
                   const float* W, const float* W1, const float* dense_w /*[Dn,D]*/,
                   const float* dense_w_one /*[Dn]*/, int64_t pad, const int64_t* slot_off,
                   float* y1, float* y2, float* feat, float* sum_emb) {
  const int F = S + Dn;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; ++b) {
    float s[256], q[256];
    for (int d = 0; d < D; ++d) s[d] = q[d] = 0.f;
    float first = 0.f;
    float* fb = feat + b * (int64_t)F * D;
    for (int f = 0; f < S; ++f) {                       /* net.py:108,117 lookups */
      int64_t r = eff_row(ids, b, f, S, pad, slot_off);
      const float* row = r < 0 ? NULL : W + r * D;
      if (r >= 0) first += W1[r];                       /* net.py:113 paddle.sum(sparse_emb_one,1) */
      for (int d = 0; d < D; ++d) {
        float e = row ? row[d] : 0.f;
        fb[f * D + d] = e;
        s[d] += e;                                      /* net.py:124 */
        q[d] += e * e;                                  /* net.py:130-133 */
      }
    }
    float dfirst = 0.f;
    for (int j = 0; j < Dn; ++j) {                      /* net.py:110-111,118-119 */
      float x = dense[b * Dn + j];
      dfirst += x * dense_w_one[j];
      for (int d = 0; d < D; ++d) {
        float e = x * dense_w[j * D + d];
        fb[(S + j) * D + d] = e;
        s[d] += e;
        q[d] += e * e;
      }
    }
    float acc = 0.f;
    for (int d = 0; d < D; ++d) {
      acc += s[d] * s[d] - q[d];                        /* net.py:135-136 */
      if (sum_emb) sum_emb[b * D + d] = s[d];
    }
    y1[b] = first + dfirst;                             /* net.py:113-114 */
    y2[b] = 0.5f * acc;                                 /* net.py:135 */
  }
}

/* backward of the block above.
 * d_feat_dnn [B,F,D], dy1, dy2 [B]  ->  row_grad [B*S,D], row_grad1 [B*S],
 * d_dense_w [Dn,D], d_dense_w_one [Dn] (batch sums, accumulated per thread then combined in
 * thread order -> deterministic for a fixed thread count). */
void oracle_fm_bwd(int64_t B, int S, int Dn, int D, const float* dense, const float* feat,
                   const float* sum_emb, const float* d_feat_dnn, const float* dy1,
                   const float* dy2, float* row_grad, float* row_grad1, float* d_dense_w,
                   float* d_dense_w_one) {
  const int F = S + Dn;
  int nt = oracle_num_threads();
  size_t per = (size_t)Dn * D + Dn;
  float* part = (float*)calloc((size_t)nt * per, sizeof(float));
#pragma omp parallel
  {
#ifdef _OPENMP
    int tid = omp_get_thread_num();
#else
    int tid = 0;
#endif
    float* pw = part + (size_t)tid * per;
    float* pw1 = pw + (size_t)Dn * D;
#pragma omp for schedule(static)
    for (int64_t b = 0; b < B; ++b) {
      const float* fb = feat + b * (int64_t)F * D;
      const float* gb = d_feat_dnn + b * (int64_t)F * D;
      const float* sb = sum_emb + b * D;
      float g2 = dy2[b], g1 = dy1[b];
      for (int f = 0; f < S; ++f) {
        for (int d = 0; d < D; ++d)
          row_grad[(b * S + f) * D + d] = gb[f * D + d] + g2 * (sb[d] - fb[f * D + d]);
        row_grad1[b * S + f] = g1;
      }
      for (int j = 0; j < Dn; ++j) {
        float x = dense[b * Dn + j];
        pw1[j] += g1 * x;
        for (int d = 0; d < D; ++d) {
          float de = gb[(S + j) * D + d] + g2 * (sb[d] - fb[(S + j) * D + d]);
          pw[j * D + d] += x * de;
        }
      }
    }
  }
  for (size_t k = 0; k < (size_t)Dn * D; ++k) d_dense_w[k] = 0.f;
  for (int j = 0; j < Dn; ++j) d_dense_w_one[j] = 0.f;
  for (int t = 0; t < nt; ++t) {
    const float* pw = part + (size_t)t * per;
    for (size_t k = 0; k < (size_t)Dn * D; ++k) d_dense_w[k] += pw[k];
    for (int j = 0; j < Dn; ++j) d_dense_w_one[j] += pw[(size_t)Dn * D + j];
  }
  free(part);
}

/* SelectedRows MergeAdd + lazy Adam on the merged rows (Appendix B-1, B-3).
 * spos [n] positions (b*S+s) grouped by row, seg_off [U+1], uniq [U].
 * Tables P/M/V are [N,D]; grad values [B*S,D] are summed in ascending-position order. */
void oracle_adam_rows(int64_t U, int D, const int64_t* uniq, const int64_t* seg_off,
                      const int64_t* spos, const float* row_grad, float* P, float* M, float* V,
                      float lr, float beta1, float beta2, float eps, int64_t step) {
  float b1p = powf(beta1, (float)step), b2p = powf(beta2, (float)step);
  float lr_t = lr * sqrtf(1.f - b2p) / (1.f - b1p);
  float eps_t = eps * sqrtf(1.f - b2p);
#pragma omp parallel for schedule(static)
  for (int64_t u = 0; u < U; ++u) {
    float g[256];
    for (int d = 0; d < D; ++d) g[d] = 0.f;
    for (int64_t k = seg_off[u]; k < seg_off[u + 1]; ++k) {
      const float* src = row_grad + spos[k] * D;
      for (int d = 0; d < D; ++d) g[d] += src[d];
    }
    int64_t r = uniq[u];
    for (int d = 0; d < D; ++d) {
      float m = beta1 * M[r * D + d] + (1.f - beta1) * g[d];
      float v = beta2 * V[r * D + d] + (1.f - beta2) * g[d] * g[d];
      M[r * D + d] = m;
      V[r * D + d] = v;
      P[r * D + d] -= lr_t * (m / (sqrtf(v) + eps_t));
    }
  }
}

/* XXH32 (seed 0) of a byte string — models/rank/dnn/benchmark_reader.py:52; published spec. */
static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }
uint32_t oracle_xxh32(const uint8_t* p, size_t n, uint32_t seed) {
  const uint32_t P1 = 2654435761u, P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u,
                 P5 = 374761393u;
  const uint8_t* end = p + n;
  uint32_t h;
  if (n >= 16) {
    uint32_t v1 = seed + P1 + P2, v2 = seed + P2, v3 = seed, v4 = seed - P1;
    while (p + 16 <= end) {
      uint32_t w[4];
      memcpy(w, p, 16);
      v1 = rotl32(v1 + w[0] * P2, 13) * P1;
      v2 = rotl32(v2 + w[1] * P2, 13) * P1;
      v3 = rotl32(v3 + w[2] * P2, 13) * P1;
      v4 = rotl32(v4 + w[3] * P2, 13) * P1;
      p += 16;
    }
    h = rotl32(v1, 1) + rotl32(v2, 7) + rotl32(v3, 12) + rotl32(v4, 18);
  } else {
    h = seed + P5;
  }
  h += (uint32_t)n;
  while (p + 4 <= end) {
    uint32_t w;
    memcpy(&w, p, 4);
    h = rotl32(h + w * P3, 17) * P4;
    p += 4;
  }
  while (p < end) {
    h = rotl32(h + (*p) * P5, 11) * P1;
    ++p;
  }
  h ^= h >> 15;
  h *= P2;
  h ^= h >> 13;
  h *= P3;
  h ^= h >> 16;
  return h;
}
