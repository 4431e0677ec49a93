// xDeepFM's Compressed Interaction Network around the GEMM (gfx950).
//
// Reference: /root/reference/models/rank/xdeepfm/net.py:155-202 — per layer
//   Z[b,d,f,s] = X0[b,f,d] * Xk[b,s,d]          (matmul of a [F,1] by a [1,S] per (b,d))   net.py:163-173
//   X_{k+1}[b,c,d] = sum_{f,s} Wc[c, f*S+s] Z[b,d,f,s]   (1x1 Conv2D over F*S channels)    net.py:176-190
//   pooled[b, c] = sum_d X_{k+1}[b,c,d]                                                     net.py:195-198
// Here every X_k (k >= 1) is kept "d-major", XT[(b,d), c], so that the compression is ONE row-major GEMM
//   XT_{k+1}[(b,d), :] = Z[(b,d), :] @ Wc^T          M = B*D, K = F*S, N = C           (rec_gemm_f32)
// and these kernels are the HBM-bound passes either side of it: the outer-product rows Z (written once, read by
// the GEMM; batch-chunked by the caller), its backward (dZ read once -> dX0, dXk), the sum over d and its broadcast.
// No reshapes / transposes of [B,F,D] tensors exist anywhere: X0 = feat_embeddings is read through a strided view.
#include <stdlib.h>

#include "rec_common.h"

namespace rec {
namespace {

struct View { int64_t sb; int sj, sd; };
__device__ __forceinline__ int64_t at(const View& v, int64_t b, int j, int d) { return b * v.sb + (int64_t)j * v.sj + (int64_t)d * v.sd; }

// One block walks rows (b,d); thread (fr, sc): columns [sc*VEC, sc*VEC+VEC) of Z's f-rows fr, fr+FR, ...
// Its Xk values are loaded once per row; X0[f] is a broadcast load per f (L1-resident: F floats per row).
template <int VEC>
__global__ __launch_bounds__(kBlock) void cin_outer_fwd_kernel(int64_t rows, int D, int F, int S, const float* __restrict__ X0,
                                                               View v0, const float* __restrict__ Xk, View vk,
                                                               float* __restrict__ Z, int64_t ldz, int tpr) {
  const int sc = threadIdx.x % tpr, fr = threadIdx.x / tpr, FR = kBlock / tpr;
  const int s0 = sc * VEC;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int64_t b = r / D;
    const int d = (int)(r - b * D);
    if (s0 >= S) continue;
    float kv[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) kv[i] = Xk[at(vk, b, s0 + i, d)];
    float* zrow = Z + r * ldz + s0;
    for (int f = fr; f < F; f += FR) {
      const float x = X0[at(v0, b, f, d)];
      float o[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) o[i] = x * kv[i];
      vstore<VEC>(zrow + (int64_t)f * S, o);
    }
  }
}

// One block per row (b,d): the dZ row (F x S) staged in LDS (pitch S+1), then
//   dXk[s] = sum_f dZ[f,s] X0[f]  (+ dpool[b,s])      dX0[f] = sum_s dZ[f,s] Xk[s]
// element e of either output is always produced by thread e % kBlock, so the two may alias (layer 1: Xk == X0).
__global__ __launch_bounds__(kBlock) void cin_outer_bwd_kernel(int64_t rows, int D, int F, int S, const float* __restrict__ dZ,
                                                               int64_t ldz, const float* __restrict__ X0, View v0,
                                                               const float* __restrict__ Xk, View vk, float* dX0,
                                                               View dv0, int acc0, float* dXk, View dvk, int acck,
                                                               const float* __restrict__ dpool, int64_t ldp) {
  extern __shared__ float smem[];
  const int SP = S + 1;
  float* dz = smem;             // [F][SP]
  float* x0s = dz + F * SP;     // [F]
  float* xks = x0s + F;         // [S]
  const int tid = threadIdx.x;
  for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
    const int64_t b = r / D;
    const int d = (int)(r - b * D);
    const float* zrow = dZ + r * ldz;
    for (int i = tid; i < F * S; i += kBlock) {
      const int f = i / S, s = i - f * S;
      dz[f * SP + s] = zrow[i];
    }
    for (int f = tid; f < F; f += kBlock) x0s[f] = X0[at(v0, b, f, d)];
    for (int s = tid; s < S; s += kBlock) xks[s] = Xk[at(vk, b, s, d)];
    __syncthreads();
    for (int f = tid; f < F; f += kBlock) {
      float a = 0.f;
      for (int s = 0; s < S; ++s) a += dz[f * SP + s] * xks[s];
      float* o = dX0 + at(dv0, b, f, d);
      *o = (acc0 ? *o : 0.f) + a;
    }
    for (int s = tid; s < S; s += kBlock) {
      float a = 0.f;
      for (int f = 0; f < F; ++f) a += dz[f * SP + s] * x0s[f];
      if (dpool) a += dpool[b * ldp + s];
      float* o = dXk + at(dvk, b, s, d);
      *o = (acck ? *o : 0.f) + a;
    }
    __syncthreads();
  }
}

// ---- narrow layers (F, S <= 64: the first CIN layer, 39 x 39): ONE WAVE per row (b,d), no block barriers.
// The block-per-row kernels above keep 39 of 256 threads busy in their reductions and store Z with scalar stores in
// F separate runs: 4.2 ms (0.85 TB/s) and 1.3 ms (2.8 TB/s) per call at B 65536 (profiles/r03_xdeepfm.txt).
//   forward : lane i, i + 64, ... of the row's F*S outputs, f = i / S by a multiply-shift: fully coalesced stores
//   backward: the dZ row staged in the wave's LDS with coalesced loads; lanes over s sum over f (x0[f] by readlane),
//             lanes over f sum over s (row reads at stride S: S odd -> conflict-free) — ascending order, as above.
__global__ __launch_bounds__(kBlock) void cin_outer_fwd_wave_kernel(int64_t rows, int D, int F, int S, unsigned magic,
                                                                    const float* __restrict__ X0, View v0,
                                                                    const float* __restrict__ Xk, View vk,
                                                                    float* __restrict__ Z, int64_t ldz) {
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int FS = F * S;
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + wave; r < rows; r += (int64_t)gridDim.x * (kBlock / kWave)) {
    const int64_t b = r / D;
    const int d = (int)(r - b * D);
    const float x0v = lane < F ? X0[at(v0, b, lane, d)] : 0.f;
    const float xkv = lane < S ? Xk[at(vk, b, lane, d)] : 0.f;
    float* zrow = Z + r * ldz;
    for (int i0 = 0; i0 < FS; i0 += kWave) {      // wave-uniform trip count: every lane takes part in the shuffles
      const int i = min(i0 + lane, FS - 1);
      const int f = (int)(((uint64_t)(unsigned)i * magic) >> 32);      // i / S for i < 2^16 (magic = ceil(2^32 / S))
      const int sc = i - f * S;
      const float z = __shfl(x0v, f, kWave) * __shfl(xkv, sc, kWave);
      if (i0 + lane < FS) zrow[i] = z;
    }
  }
}

__global__ __launch_bounds__(kBlock) void cin_outer_bwd_wave_kernel(int64_t rows, int D, int F, int S, int rowf,
                                                                    const float* __restrict__ dZ, int64_t ldz,
                                                                    const float* __restrict__ X0, View v0,
                                                                    const float* __restrict__ Xk, View vk, float* dX0,
                                                                    View dv0, int acc0, float* dXk, View dvk, int acck,
                                                                    const float* __restrict__ dpool, int64_t ldp) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  float* dz = smem + (size_t)wave * rowf;      // [F][S]
  const int FS = F * S;
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + wave; r < rows; r += (int64_t)gridDim.x * (kBlock / kWave)) {
    const int64_t b = r / D;
    const int d = (int)(r - b * D);
    const float* zrow = dZ + r * ldz;
    for (int i = lane; i < FS; i += kWave) dz[i] = zrow[i];
    const float x0v = lane < F ? X0[at(v0, b, lane, d)] : 0.f;
    const float xkv = lane < S ? Xk[at(vk, b, lane, d)] : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float a0 = 0.f, ak = 0.f;
    // (v_readlane: a wave-uniform source lane, read whether or not that lane is active here)
    const int fl = min(lane, F - 1), sl = min(lane, S - 1);
    for (int sc = 0; sc < S; ++sc)        // dX0[f] = sum_s dZ[f,s] Xk[s]
      a0 += dz[fl * S + sc] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xkv), sc));
    for (int f = 0; f < F; ++f)           // dXk[s] = sum_f dZ[f,s] X0[f]
      ak += dz[f * S + sl] * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x0v), f));
    // lane j writes element j of both outputs, dX0 first: the two may alias (layer 1: Xk == X0)
    if (lane < F) {
      float* o = dX0 + at(dv0, b, lane, d);
      *o = (acc0 ? *o : 0.f) + a0;
    }
    if (lane < S) {
      if (dpool) ak += dpool[b * ldp + lane];
      float* o = dXk + at(dvk, b, lane, d);
      *o = (acck ? *o : 0.f) + ak;
    }
    __builtin_amdgcn_wave_barrier();      // the stage is rewritten by the next row
  }
}

__global__ __launch_bounds__(kBlock) void cin_sumpool_kernel(int64_t B, int D, int C, const float* __restrict__ XT, int64_t ldx,
                                                             float* __restrict__ out, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B * C) return;
  const int64_t b = i / C;
  const int c = (int)(i - b * C);
  float a = 0.f;
  for (int d = 0; d < D; ++d) a += XT[(b * D + d) * ldx + c];     // fixed order over d
  out[b * ldo + c] = a;
}

__global__ __launch_bounds__(kBlock) void cin_sumpool_bwd_kernel(int64_t B, int D, int C, const float* __restrict__ dpool,
                                                                 int64_t ldp, float* __restrict__ dXT, int64_t ldx) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= B * D * C) return;
  const int64_t r = i / C;
  const int c = (int)(i - r * C);
  dXT[r * ldx + c] = dpool[(r / D) * ldp + c];
}

// ---- the other association of a CIN layer:  X_{k+1}[m,c] = sum_f X0[m,f] * Y[m, c*F + f],   Y = Xk @ W'^T  with
// W' = the conv weight viewed [C*F, S].  Y is B*D*C*F floats where Z is B*D*F*S: the smaller one goes through HBM
// (layer 2 of the reference config: C 32 < S 128 -> Y is a quarter of Z, and its GEMM is K 128 x N 1248 instead of
// K 4992 x N 32).  One wave per row (b,d): the Y row staged in LDS, lane c walks its F consecutive values.
__global__ __launch_bounds__(kBlock) void cin_contract_fwd_kernel(int64_t rows, int D, int F, int C, const float* __restrict__ Y,
                                                                  int64_t ldy, const float* __restrict__ X0, View v0,
                                                                  float* __restrict__ XT, int64_t ldx) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int CF = C * F;
  float* ys = smem + (size_t)wave * (CF + F);
  float* xs = ys + CF;
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + wave; r < rows; r += (int64_t)gridDim.x * (kBlock / kWave)) {
    const int64_t b = r / D;
    const int d = (int)(r - b * D);
    const float* yrow = Y + r * ldy;
    for (int i = lane; i < CF; i += kWave) ys[i] = yrow[i];
    for (int f = lane; f < F; f += kWave) xs[f] = X0[at(v0, b, f, d)];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < C; c += kWave) {
      float a = 0.f;
      for (int f = 0; f < F; ++f) a += xs[f] * ys[c * F + f];          // fixed order over f
      XT[r * ldx + c] = a;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// dY[m, c*F+f] = dXT[m,c] * X0[m,f];   dX0[m,f] (+)= sum_c dXT[m,c] * Y[m, c*F+f]
__global__ __launch_bounds__(kBlock) void cin_contract_bwd_kernel(int64_t rows, int D, int F, int C, const float* __restrict__ Y,
                                                                  int64_t ldy, const float* __restrict__ dXT, int64_t ldx,
                                                                  const float* __restrict__ X0, View v0,
                                                                  float* __restrict__ dY, int64_t lddy, float* dX0,
                                                                  View dv0, int acc0) {
  extern __shared__ float smem[];
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int CF = C * F;
  float* ys = smem + (size_t)wave * (CF + F + C);
  float* xs = ys + CF;
  float* gs = xs + F;
  for (int64_t r = (int64_t)blockIdx.x * (kBlock / kWave) + wave; r < rows; r += (int64_t)gridDim.x * (kBlock / kWave)) {
    const int64_t b = r / D;
    const int d = (int)(r - b * D);
    const float* yrow = Y + r * ldy;
    for (int i = lane; i < CF; i += kWave) ys[i] = yrow[i];
    for (int f = lane; f < F; f += kWave) xs[f] = X0[at(v0, b, f, d)];
    for (int c = lane; c < C; c += kWave) gs[c] = dXT[r * ldx + c];
    __builtin_amdgcn_s_waitcnt(0);
    __builtin_amdgcn_wave_barrier();
    float* dyrow = dY + r * lddy;
    for (int c = 0; c < C; ++c) {
      const float g = gs[c];
      for (int f = lane; f < F; f += kWave) dyrow[c * F + f] = g * xs[f];
    }
    for (int f = lane; f < F; f += kWave) {
      float a = 0.f;
      for (int c = 0; c < C; ++c) a += gs[c] * ys[c * F + f];          // fixed order over c
      float* o = dX0 + at(dv0, b, f, d);
      *o = (acc0 ? *o : 0.f) + a;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

View view_of(const rec_cin_view* v) { return View{v->stride_b, v->stride_j, v->stride_d}; }

}  // namespace
}  // namespace rec

using namespace rec;

extern "C" int rec_cin_outer_fwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t S, const float* X0,
                                 const rec_cin_view* v0, const float* Xk, const rec_cin_view* vk, float* Z,
                                 int64_t ldz, void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0 && F > 0 && S > 0 && ldz >= (int64_t)F * S, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(X0 && Xk && Z && v0 && vk, REC_EINVAL, "null pointer argument");
  const int64_t rows = batch * emb_dim;
  static const bool wave_on = [] { const char* v = getenv("REC_CIN_WAVE"); return !(v && *v == '0'); }();
  if (wave_on && F <= kWave && S <= kWave && S % 4 != 0) {      // narrow layer, rows that float4 stores cannot take
    const int64_t g = (rows + kBlock / kWave - 1) / (kBlock / kWave);
    hipLaunchKernelGGL(cin_outer_fwd_wave_kernel, dim3((unsigned)(g < 16384 ? g : 16384)), dim3(kBlock), 0,
                       (hipStream_t)stream, rows, emb_dim, F, S, (unsigned)((0x100000000ull + S - 1) / S), X0, view_of(v0),
                       Xk, view_of(vk), Z, ldz);
    return check_launch("rec_cin_outer_fwd (wave)");
  }
  const bool vec = S % 4 == 0 && ldz % 4 == 0 && ((uintptr_t)Z % 16) == 0;
  const int cols = vec ? S / 4 : S;
  REC_REQUIRE(cols <= kBlock, REC_ESHAPE, "previous CIN layer too wide (%d)", S);
  const int tpr = pow2_ceil(cols);
  const int64_t grid = rows < 64 * 1024 ? rows : 64 * 1024;
  if (vec)
    hipLaunchKernelGGL(cin_outer_fwd_kernel<4>, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, rows,
                       emb_dim, F, S, X0, view_of(v0), Xk, view_of(vk), Z, ldz, tpr);
  else
    hipLaunchKernelGGL(cin_outer_fwd_kernel<1>, dim3((unsigned)grid), dim3(kBlock), 0, (hipStream_t)stream, rows,
                       emb_dim, F, S, X0, view_of(v0), Xk, view_of(vk), Z, ldz, tpr);
  return check_launch("rec_cin_outer_fwd");
}

extern "C" int rec_cin_outer_bwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t S, const float* dZ, int64_t ldz,
                                 const float* X0, const rec_cin_view* v0, const float* Xk, const rec_cin_view* vk,
                                 float* dX0, const rec_cin_view* dv0, int32_t accumulate_dx0, float* dXk,
                                 const rec_cin_view* dvk, int32_t accumulate_dxk, const float* dpool,
                                 int64_t ld_dpool, void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0 && F > 0 && S > 0 && ldz >= (int64_t)F * S, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(dZ && X0 && Xk && dX0 && dXk && v0 && vk && dv0 && dvk, REC_EINVAL, "null pointer argument");
  const int64_t rows = batch * emb_dim;
  static const bool wave_on = [] { const char* v = getenv("REC_CIN_WAVE"); return !(v && *v == '0'); }();
  if (wave_on && F <= kWave && S <= kWave) {
    const int rowf = (F * S + 3) & ~3;
    const size_t sh = sizeof(float) * (size_t)rowf * (kBlock / kWave);
    if (sh <= 64 * 1024) {
      int64_t g = resident_blocks(cin_outer_bwd_wave_kernel, kBlock, sh);
      const int64_t need = (rows + kBlock / kWave - 1) / (kBlock / kWave);
      if (g > need) g = need;
      hipLaunchKernelGGL(cin_outer_bwd_wave_kernel, dim3((unsigned)g), dim3(kBlock), sh, (hipStream_t)stream, rows, emb_dim,
                         F, S, rowf, dZ, ldz, X0, view_of(v0), Xk, view_of(vk), dX0, view_of(dv0), accumulate_dx0, dXk,
                         view_of(dvk), accumulate_dxk, dpool, ld_dpool);
      return check_launch("rec_cin_outer_bwd (wave)");
    }
  }
  const size_t shmem = sizeof(float) * ((size_t)F * (S + 1) + F + S);
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "CIN row of %d x %d does not fit the LDS stage", F, S);
  int64_t grid = resident_blocks(cin_outer_bwd_kernel, kBlock, shmem);
  if (grid > rows) grid = rows;
  hipLaunchKernelGGL(cin_outer_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem, (hipStream_t)stream, rows,
                     emb_dim, F, S, dZ, ldz, X0, view_of(v0), Xk, view_of(vk), dX0, view_of(dv0), accumulate_dx0, dXk,
                     view_of(dvk), accumulate_dxk, dpool, ld_dpool);
  return check_launch("rec_cin_outer_bwd");
}

extern "C" int rec_cin_sumpool(int64_t batch, int32_t emb_dim, int32_t C, const float* XT, int64_t ldx, float* out,
                               int64_t ldo, void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0 && C > 0 && ldx >= C && ldo >= C, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(XT && out, REC_EINVAL, "null pointer argument");
  const int64_t n = batch * C;
  hipLaunchKernelGGL(cin_sumpool_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, batch, emb_dim, C, XT, ldx, out, ldo);
  return check_launch("rec_cin_sumpool");
}

extern "C" int rec_cin_sumpool_bwd(int64_t batch, int32_t emb_dim, int32_t C, const float* dpool, int64_t ldp,
                                   float* dXT, int64_t ldx, void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0 && C > 0 && ldx >= C && ldp >= C, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(dpool && dXT, REC_EINVAL, "null pointer argument");
  const int64_t n = batch * emb_dim * C;
  hipLaunchKernelGGL(cin_sumpool_bwd_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                     (hipStream_t)stream, batch, emb_dim, C, dpool, ldp, dXT, ldx);
  return check_launch("rec_cin_sumpool_bwd");
}

extern "C" int rec_cin_contract_fwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t C, const float* Y, int64_t ldy,
                                    const float* X0, const rec_cin_view* v0, float* XT, int64_t ldx, void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0 && F > 0 && C > 0 && ldy >= (int64_t)C * F && ldx >= C, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(Y && X0 && XT && v0, REC_EINVAL, "null pointer argument");
  const size_t shmem = sizeof(float) * (kBlock / kWave) * ((size_t)C * F + F);
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "CIN row of %d x %d does not fit the LDS stage", C, F);
  const int64_t rows = batch * emb_dim;
  int64_t grid = (rows + kBlock / kWave - 1) / (kBlock / kWave);
  if (grid > 32 * kNumCU) grid = 32 * kNumCU;
  hipLaunchKernelGGL(cin_contract_fwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem, (hipStream_t)stream, rows,
                     emb_dim, F, C, Y, ldy, X0, view_of(v0), XT, ldx);
  return check_launch("rec_cin_contract_fwd");
}

extern "C" int rec_cin_contract_bwd(int64_t batch, int32_t emb_dim, int32_t F, int32_t C, const float* Y, int64_t ldy,
                                    const float* dXT, int64_t ldx, const float* X0, const rec_cin_view* v0, float* dY,
                                    int64_t lddy, float* dX0, const rec_cin_view* dv0, int32_t accumulate_dx0,
                                    void* stream) {
  REC_REQUIRE(batch >= 0 && emb_dim > 0 && F > 0 && C > 0 && ldy >= (int64_t)C * F && lddy >= (int64_t)C * F &&
                  ldx >= C, REC_EINVAL, "bad sizes");
  if (batch == 0) return REC_OK;
  REC_REQUIRE(Y && dXT && X0 && dY && dX0 && v0 && dv0, REC_EINVAL, "null pointer argument");
  const size_t shmem = sizeof(float) * (kBlock / kWave) * ((size_t)C * F + F + C);
  REC_REQUIRE(shmem <= 64 * 1024, REC_ESHAPE, "CIN row of %d x %d does not fit the LDS stage", C, F);
  const int64_t rows = batch * emb_dim;
  int64_t grid = (rows + kBlock / kWave - 1) / (kBlock / kWave);
  if (grid > 32 * kNumCU) grid = 32 * kNumCU;
  hipLaunchKernelGGL(cin_contract_bwd_kernel, dim3((unsigned)grid), dim3(kBlock), shmem, (hipStream_t)stream, rows,
                     emb_dim, F, C, Y, ldy, dXT, ldx, X0, view_of(v0), dY, lddy, dX0, view_of(dv0), accumulate_dx0);
  return check_launch("rec_cin_contract_bwd");
}
