// Shared pieces of the f32 GEMM kernels (gemm_f32.hip, gemm_glds.h, gemm_panel.h): the MFMA vector type, the k-step, the
// register-staged tile loader and the epilogues.
#pragma once

#include "rec_common.h"

namespace rec {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kBK = 16;

// ---------------------------------------------------------------------------------- tile loader
// Logical tile T[R][C] of a matrix; MEMT: memory is contiguous along R (element (r,c) at p[c*ld + r]).
// load(): the thread's float4s of the tile (zeros outside [rmax, cmax)); vec_ok = 16-B aligned rows.
// store<TRANSPOSE>(): LDS image in memory order, [OUTER][INNER+4] — or transposed, [INNER][OUTER+4].
template <int R, int C, bool MEMT, int NTHR>
struct TileLoader {
  static constexpr int INNER = MEMT ? R : C, OUTER = MEMT ? C : R;
  static constexpr int kVecs = R * C / 4;
  static constexpr int kPerThread = (kVecs + NTHR - 1) / NTHR;
  float4 stage[kPerThread];

  // MODE 0: the tile is known to be inside the matrix (interior block, full K tile) — straight float4 loads.
  // MODE 1: branch-free edge handling — the address of a float4 outside [rmax, cmax) is clamped to a valid one
  //         and the value zeroed by a select.  Needs 16-B aligned rows and contiguous extents that are multiples
  //         of 4 (a float4 is then either inside or outside).
  // MODE 2: element-wise bounds checks (the K tail tile, unaligned operands).
  // Modes 0/1 keep a tile's loads back to back and in flight under the MFMAs; a branchy loader in the steady-state
  // loop makes the compiler wait vmcnt(0) before the first MFMA (measured: 78 vs 110 TF).
  template <int MODE>
  __device__ __forceinline__ void load(const float* __restrict__ p, int64_t ld, int64_t r0, int64_t c0,
                                       int64_t rmax, int64_t cmax, bool vec_ok, int tid) {
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      // threads beyond the tile's last float4 re-load that one (unconditional loads: no exec-mask branch, no
      // vmcnt(0) in front of the MFMAs); only their LDS store is skipped
      const int v0 = tid + it * NTHR;
      const int v = (kVecs % NTHR == 0 || v0 < kVecs) ? v0 : kVecs - 1;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      {
        const int o = v / (INNER / 4), i4 = (v % (INNER / 4)) * 4;
        const int64_t go = (MEMT ? c0 : r0) + o, gi = (MEMT ? r0 : c0) + i4;
        const int64_t omax = MEMT ? cmax : rmax, imax = MEMT ? rmax : cmax;
        if (MODE == 0) {
          x = *reinterpret_cast<const float4*>(p + go * ld + gi);
        } else if (MODE == 1) {
          const bool ok = go < omax && gi + 3 < imax;
          const int64_t go_c = go < omax ? go : omax - 1;
          const int64_t gi_c = gi + 3 < imax ? gi : 0;
          const float4 t = *reinterpret_cast<const float4*>(p + go_c * ld + gi_c);
          x.x = ok ? t.x : 0.f; x.y = ok ? t.y : 0.f; x.z = ok ? t.z : 0.f; x.w = ok ? t.w : 0.f;
        } else if (go < omax) {
          const float* q = p + go * ld + gi;
          if (vec_ok && gi + 3 < imax) {
            x = *reinterpret_cast<const float4*>(q);
          } else {
            if (gi + 0 < imax) x.x = q[0];
            if (gi + 1 < imax) x.y = q[1];
            if (gi + 2 < imax) x.z = q[2];
            if (gi + 3 < imax) x.w = q[3];
          }
        }
      }
      stage[it] = x;
    }
  }

  template <bool TRANSPOSE>
  __device__ __forceinline__ void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int it = 0; it < kPerThread; ++it) {
      const int v = tid + it * NTHR;
      if (kVecs % NTHR == 0 || v < kVecs) {
        const int o = v / (INNER / 4), i4 = (v % (INNER / 4)) * 4;
        if (!TRANSPOSE) {
          *reinterpret_cast<float4*>(lds + o * (INNER + 4) + i4) = stage[it];
        } else {
          lds[(i4 + 0) * (OUTER + 4) + o] = stage[it].x;
          lds[(i4 + 1) * (OUTER + 4) + o] = stage[it].y;
          lds[(i4 + 2) * (OUTER + 4) + o] = stage[it].z;
          lds[(i4 + 3) * (OUTER + 4) + o] = stage[it].w;
        }
      }
    }
  }
};

// ------------------------------------------------------------------------------------ epilogues
struct EpiArgs {
  const float* bias;       // [N] or null
  const float* aux0;       // [M,ld0]: RELU_MASK source / CROSS, MOE X_0 / second ADD operand
  const float* aux1;       // [M,ld1]: CROSS, MOE X_l / ADD operand
  const float* row_scale;  // [M] (stride rs_stride): MOE gate probability of this expert
  float* out2;             // [M,ldc] or null: CROSS also stores u = acc + bias (saved for backward)
  int ld0, ld1, rs_stride, ld2;
  int nt_store;            // 1: the whole-tile kernel writes C with non-temporal stores (gemm_f32.hip: the dX + ReLU' form)
  unsigned long long* relu_bits;   // bf16 x 3 forward / dX kernel only (gemm_bf16x3.h): BIAS_RELU writes, RELU_MASK reads the mask bits
};

// which per-element operands an epilogue reads besides the accumulator
template <int EPI>
struct EpiUses {
  static constexpr bool aux0 = EPI == REC_EPI_RELU_MASK || EPI == REC_EPI_CROSS || EPI == REC_EPI_DTANH ||
                               EPI == REC_EPI_DSIGMOID || EPI == REC_EPI_MOE || EPI == REC_EPI_ADD;
  static constexpr bool aux1 = EPI == REC_EPI_CROSS || EPI == REC_EPI_MOE || EPI == REC_EPI_ADD;
};
template <int EPI>
__device__ __forceinline__ float load_aux0(int64_t i, int j, const EpiArgs& e) {
  if (EPI == REC_EPI_ADD) return e.aux0 ? e.aux0[i * e.ld0 + j] : 0.f;
  return EpiUses<EPI>::aux0 ? e.aux0[i * e.ld0 + j] : 0.f;
}
template <int EPI>
__device__ __forceinline__ float load_aux1(int64_t i, int j, const EpiArgs& e) {
  return EpiUses<EPI>::aux1 ? e.aux1[i * e.ld1 + j] : 0.f;
}

// the epilogue on operands already in registers (x0 = aux0[i,j], x1 = aux1[i,j]): the tile store below first issues
// every aux load of the tile and only then starts storing — C may alias nothing the compiler can prove, so loads
// interleaved with the stores would each wait for the store in front of them (measured: 44 us of a 221 us dX GEMM)
// bias[j] (0 where the epilogue has none / the pointer is null): like the aux operands, loaded ahead of the stores
template <int EPI>
__device__ __forceinline__ float load_bias(int j, const EpiArgs& e) {
  if (EPI == REC_EPI_BIAS || EPI == REC_EPI_BIAS_RELU || EPI == REC_EPI_CROSS || EPI == REC_EPI_MOE ||
      EPI == REC_EPI_BIAS_SIGMOID)
    return e.bias[j];
  if (EPI == REC_EPI_BIAS_TANH || EPI == REC_EPI_ADD) return e.bias ? e.bias[j] : 0.f;
  return 0.f;
}
template <int EPI>
__device__ __forceinline__ float apply_epi(float acc, float x0, float x1, float bj, int64_t i, const EpiArgs& e) {
  if (EPI == REC_EPI_NONE) return acc;
  if (EPI == REC_EPI_BIAS) return acc + bj;
  if (EPI == REC_EPI_BIAS_RELU) return fmaxf(acc + bj, 0.f);
  if (EPI == REC_EPI_RELU_MASK) return x0 > 0.f ? acc : 0.f;
  if (EPI == REC_EPI_CROSS) return x1 + x0 * (acc + bj);
  if (EPI == REC_EPI_DTANH) return acc * (1.f - x0 * x0);
  if (EPI == REC_EPI_DSIGMOID) return acc * x0 * (1.f - x0);
  if (EPI == REC_EPI_MOE) return x1 + x0 * (e.row_scale[i * e.rs_stride] * (acc + bj));
  if (EPI == REC_EPI_BIAS_SIGMOID) return 1.f / (1.f + expf(-(acc + bj)));
  if (EPI == REC_EPI_BIAS_TANH) return tanhf(acc + bj);
  if (EPI == REC_EPI_ADD) return acc + bj + x1 + x0;
  return acc;
}
template <int EPI>
__device__ __forceinline__ float apply_epi(float acc, int64_t i, int j, const EpiArgs& e) {
  return apply_epi<EPI>(acc, load_aux0<EPI>(i, j, e), load_aux1<EPI>(i, j, e), load_bias<EPI>(j, e), i, e);
}

}  // namespace rec
