// k_norm.h - GroupNorm(32) statistics / apply(+SiLU) and LayerNorm for NHWC tensors (HBM-bound).
//
// Reference semantics: F.group_norm(x, 32, gamma, beta, eps) [+ F.silu] inside diffusers
// ResnetBlock2D / Transformer2DModel / VAE Attention / conv_norm_out (SURVEY.md Appendix A.3,A.5-A.7;
// eps 1e-6 VAE + Transformer2D, 1e-5 U-Net ResBlocks), and F.layer_norm(eps 1e-5) x3 per
// BasicTransformerBlock.  Statistics are per (image, group) over (C/32 channels x H*W pixels), biased
// variance, computed in fp32 per thread, fp32 LDS atomics per block and fp64 global atomics across
// blocks; the affine transform is applied in fp32 and rounded once to fp16 (the MFMA operand type).
//
// Work decomposition (both kernels): a thread owns ONE fixed 8-channel vector (16 B fp16 / 32 B fp32)
// and strides over pixels, so gamma/beta/scale/shift live in registers and all global accesses are
// 16-B vectors, consecutive threads -> consecutive channel vectors -> fully coalesced rows.
#pragma once
#include "sdm_common.h"
#include "k_gemm.h"      // p3_store8: the P3 operand planes of the plane-fed GEMM

struct GnSrc {
  const void* in0; const void* in1;  // channel concat (in1 may be null)
  int C0, C1; int in_f32;
  int HW;                            // pixels per image
};

// sums[n][g][2] (double) += {sum x, sum x^2}
__global__ void __launch_bounds__(512) gn_stats_kernel(GnSrc s, double* __restrict__ sums, int groups, int pix_per_block) {
  SDM_DYN_SMEM(smem);
  const int C = s.C0 + s.C1;
  float* lsum = (float*)smem;      // [C]
  float* lsq = lsum + C;           // [C]
  const int CV = C / 8;
  const int slots = blockDim.x / CV;
  const int tid = threadIdx.x;
  for (int i = tid; i < 2 * C; i += blockDim.x) lsum[i] = 0.0f;
  __syncthreads();
  const int n = blockIdx.y;
  const int cv = tid % CV, slot = tid / CV;
  if (slot < slots) {
    const int c = cv * 8;
    const void* src = s.in0; int Cs = s.C0, cc = c;
    if (c >= s.C0) { src = s.in1; Cs = s.C1; cc = c - s.C0; }
    float a[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = 0.0f; q[e] = 0.0f; }
    const int p0 = blockIdx.x * pix_per_block;
    const int p1 = min(p0 + pix_per_block, s.HW);
    for (int p = p0 + slot; p < p1; p += slots) {
      float v[8];
      sdm_load8_as_f32(src, ((size_t)n * s.HW + p) * Cs + cc, s.in_f32, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) { a[e] += v[e]; q[e] += v[e] * v[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { atomicAdd(&lsum[c + e], a[e]); atomicAdd(&lsq[c + e], q[e]); }
  }
  __syncthreads();
  const int cpg = C / groups;
  if (tid < groups) {
    double ss = 0.0, qq = 0.0;
    for (int j = 0; j < cpg; ++j) { ss += (double)lsum[tid * cpg + j]; qq += (double)lsq[tid * cpg + j]; }
    atomicAdd(&sums[((size_t)n * groups + tid) * 2 + 0], ss);
    atomicAdd(&sums[((size_t)n * groups + tid) * 2 + 1], qq);
  }
}

// scale[n][c] = rstd*gamma, shift[n][c] = beta - mean*rstd*gamma
__global__ void gn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                   int N, int C, int groups, long count, float eps) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * C) return;
  const int n = i / C, c = i % C, g = c / (C / groups);
  const double mean = sums[((size_t)n * groups + g) * 2] / (double)count;
  double var = sums[((size_t)n * groups + g) * 2 + 1] / (double)count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float a = rstd * gamma[c];
  scale[i] = a;
  shift[i] = beta[c] - (float)mean * a;
}

// One launch per norm: per-(tile, wave-row) partial {sum, sumsq} rows written by the producing conv epilogues (k_conv.h `stats`;
// st0 / st1 = [N][rows0|rows1][C0|C1][2] fp32 for the two concat sources) -> scale / shift of every channel.  One block of
// 1024 threads per (group, image); 16-byte loads (two channels' {sum, sumsq}), four independent fp64 accumulator pairs per
// thread; no memset, no atomics, no separate finalize pass (was three launches per norm, 113 norms per forward).  A group may
// straddle the concat boundary.
#define GN_PSS_THREADS 1024
// VEC = 2: two channels' {sum, sumsq} per 16-byte load (even channel ranges); VEC = 1: one channel per 8-byte load
template <int VEC>
SDM_DEV_INLINE void gn_pss_accumulate(const float* __restrict__ st, size_t img_off, int rows, int Csrc, int a, int nch, int tid, double (&acc)[4][2]) {
  const int nv = nch / VEC;
  const long items = (long)rows * nv;
  auto load = [&](long it, float (&v)[4]) {
    const long r = it / nv;
    const int c = a + VEC * (int)(it - r * nv);
    const float* pp = st + (img_off + (size_t)r * Csrc + c) * 2;
    if (VEC == 2) { const f32x4 t = *(const f32x4*)pp; v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3]; }
    else { const f32x2 t = *(const f32x2*)pp; v[0] = t[0]; v[1] = t[1]; v[2] = 0.0f; v[3] = 0.0f; }
  };
  long i = tid;
  for (; i + 3L * GN_PSS_THREADS < items; i += 4L * GN_PSS_THREADS) {
    float v[4][4];
#pragma unroll
    for (int u = 0; u < 4; ++u) load(i + (long)u * GN_PSS_THREADS, v[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc[u][0] += (double)v[u][0] + (double)v[u][2]; acc[u][1] += (double)v[u][1] + (double)v[u][3]; }
  }
  for (; i < items; i += GN_PSS_THREADS) {
    float v[4];
    load(i, v);
    acc[0][0] += (double)v[0] + (double)v[2]; acc[0][1] += (double)v[1] + (double)v[3];
  }
}
__global__ void __launch_bounds__(GN_PSS_THREADS) gn_partials_scale_shift_kernel(const float* __restrict__ st0, int rows0, const float* __restrict__ st1,
                                                                                 int rows1, int C0, int C1, const float* __restrict__ gamma,
                                                                                 const float* __restrict__ beta, float* __restrict__ scale,
                                                                                 float* __restrict__ shift, int groups, long hw, float eps) {
  SDM_SHARED double red[GN_PSS_THREADS][2];
  const int C = C0 + C1, cpg = C / groups;
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int c_lo = g * cpg, c_hi = c_lo + cpg;
  const bool vec2 = !(cpg & 1) && !(C0 & 1);      // group and concat boundaries on even channels (every layer of the real model)
  double acc[4][2];
#pragma unroll
  for (int u = 0; u < 4; ++u) { acc[u][0] = 0.0; acc[u][1] = 0.0; }
  {   // source 0: concat channels [c_lo, min(c_hi, C0))
    const int a = c_lo < C0 ? c_lo : C0, b = c_hi < C0 ? c_hi : C0;
    if (b > a) {
      if (vec2) gn_pss_accumulate<2>(st0, (size_t)n * rows0 * C0, rows0, C0, a, b - a, tid, acc);
      else gn_pss_accumulate<1>(st0, (size_t)n * rows0 * C0, rows0, C0, a, b - a, tid, acc);
    }
  }
  if (C1 > 0) {   // source 1: concat channels [max(c_lo, C0), c_hi) -> local channels - C0
    const int a = (c_lo > C0 ? c_lo : C0) - C0, b = (c_hi > C0 ? c_hi : C0) - C0;
    if (b > a) {
      if (vec2) gn_pss_accumulate<2>(st1, (size_t)n * rows1 * C1, rows1, C1, a, b - a, tid, acc);
      else gn_pss_accumulate<1>(st1, (size_t)n * rows1 * C1, rows1, C1, a, b - a, tid, acc);
    }
  }
  red[tid][0] = (acc[0][0] + acc[1][0]) + (acc[2][0] + acc[3][0]);
  red[tid][1] = (acc[0][1] + acc[1][1]) + (acc[2][1] + acc[3][1]);
  __syncthreads();
  for (int st = GN_PSS_THREADS / 2; st > 0; st >>= 1) {
    if (tid < st) { red[tid][0] += red[tid + st][0]; red[tid][1] += red[tid + st][1]; }
    __syncthreads();
  }
  const double cnt = (double)hw * cpg;
  const double mean = red[0][0] / cnt;
  double var = red[0][1] / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int c = c_lo + tid; c < c_hi; c += GN_PSS_THREADS) {
    const float a = rstd * gamma[c];
    scale[(size_t)n * C + c] = a;
    shift[(size_t)n * C + c] = beta[c] - (float)mean * a;
  }
}

// y = act(x*scale + shift) -> fp16 NHWC with C channels (concat materialised)
__global__ void __launch_bounds__(512) gn_apply_kernel(GnSrc s, const float* __restrict__ scale, const float* __restrict__ shift,
                                                       void* __restrict__ out, int out_f32, int silu, int pix_per_block) {
  const int C = s.C0 + s.C1;
  const int CV = C / 8;
  const int slots = blockDim.x / CV;
  const int tid = threadIdx.x;
  const int n = blockIdx.y;
  const int cv = tid % CV, slot = tid / CV;
  if (slot >= slots) return;
  const int c = cv * 8;
  const void* src = s.in0; int Cs = s.C0, cc = c;
  if (c >= s.C0) { src = s.in1; Cs = s.C1; cc = c - s.C0; }
  float a[8], b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { a[e] = scale[(size_t)n * C + c + e]; b[e] = shift[(size_t)n * C + c + e]; }
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, s.HW);
  for (int p = p0 + slot; p < p1; p += slots) {
    float v[8];
    sdm_load8_as_f32(src, ((size_t)n * s.HW + p) * Cs + cc, s.in_f32, v);
    float y[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      y[e] = v[e] * a[e] + b[e];
      if (silu) y[e] = sdm_silu(y[e]);
    }
    const size_t oi = ((size_t)n * s.HW + p) * C + c;
    if (out_f32) {                                 // precise mode: the consumer splits the fp32 value into an fp16 pair itself
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) { o0[e] = y[e]; o1[e] = y[4 + e]; }
      *(f32x4*)((float*)out + oi) = o0;
      *(f32x4*)((float*)out + oi + 4) = o1;
    } else {
      f16x8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)y[e];
      *(f16x8*)((half_t*)out + oi) = o;
    }
  }
}

// The same with the result as the P3 operand planes of the plane-fed GEMM (k_gemm.h; Transformer2DModel.norm -> proj_in, VAE Attention.group_norm -> q | k | v).
// A block = 16 consecutive rows (one row block of the HI plane); thread (r = tid & 15, u = tid >> 4) walks the 8-channel runs v = u, u + 16, ...: a wave
// reads 16 rows x 128 bytes (whole lines) and writes one whole 1 KB block of the HI plane plus two 256-byte runs of the XL plane per step.
__global__ void __launch_bounds__(256) gn_apply_p3_kernel(GnSrc s, const float* __restrict__ scale, const float* __restrict__ shift, unsigned char* __restrict__ out,
                                                          int silu, long rows_total) {
  const int C = s.C0 + s.C1, nv = C / 8;
  const int tid = threadIdx.x;
  const long row = (long)blockIdx.x * 16 + (tid & 15);
  if (row >= rows_total) return;
  const long n = row / s.HW;
  unsigned char* xl = out + p3_rows_pad((size_t)rows_total) * (size_t)C * 2;
  for (int v = tid >> 4; v < nv; v += 16) {
    const int c = v * 8;
    const void* src = s.in0; int Cs = s.C0, cc = c;
    if (c >= s.C0) { src = s.in1; Cs = s.C1; cc = c - s.C0; }
    float x8[8], y[8];
    sdm_load8_as_f32(src, (size_t)row * Cs + cc, s.in_f32, x8);
    const f32x4 a0 = *(const f32x4*)(scale + (size_t)n * C + c), a1 = *(const f32x4*)(scale + (size_t)n * C + c + 4);
    const f32x4 b0 = *(const f32x4*)(shift + (size_t)n * C + c), b1 = *(const f32x4*)(shift + (size_t)n * C + c + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { y[e] = x8[e] * a0[e] + b0[e]; y[4 + e] = x8[4 + e] * a1[e] + b1[e]; }
    if (silu) {
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = sdm_silu(y[e]);
    }
    p3_store8(y, out, xl, (size_t)row, C, c);
  }
}

// LayerNorm over the last dim C (multiple of 64, <= 64*SDM_LN_MAXV), one wave per row, fp32/fp16 in -> fp16 (or fp32) out.
// Every lane owns 4-channel vectors (16-byte loads of the fp32 stream, 8/16-byte stores): vector v of the row belongs to lane
// v % 64, so a row of C channels is C/256 (rounded up) fully coalesced wave loads.  Two-pass statistics in registers
// (mean, then centred sum of squares: the same arithmetic as F.layer_norm).
#define SDM_LN_MAXV 20
#define SDM_LN_MAXQ (SDM_LN_MAXV / 4)
__global__ void __launch_bounds__(256) layernorm_kernel(const void* __restrict__ x, int in_f32, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, void* __restrict__ out, int out_f32, long rows, int C,
                                                        float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + wave;
  const bool active = row < rows;
  const long rr = active ? row : rows - 1;     // keep every lane in the shuffles
  const int nq = C / 4;                        // 4-channel vectors per row
  f32x4 v[SDM_LN_MAXQ];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < SDM_LN_MAXQ; ++i) {
    const int q = i * 64 + lane;
    v[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (q < nq) {
      const size_t idx = (size_t)rr * C + (size_t)q * 4;
      if (in_f32) {
        v[i] = *(const f32x4*)((const float*)x + idx);
      } else {
        const f16x4 h = *(const f16x4*)((const half_t*)x + idx);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i][e] = (float)h[e];
      }
      s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
  }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
  const float mean = s / (float)C;
  float qs = 0.0f;
#pragma unroll
  for (int i = 0; i < SDM_LN_MAXQ; ++i)
    if (i * 64 + lane < nq) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { const float d = v[i][e] - mean; qs += d * d; }
    }
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) qs += __shfl_xor(qs, m);
  const float rstd = 1.0f / sqrtf(qs / (float)C + eps);
  if (!active) return;
#pragma unroll
  for (int i = 0; i < SDM_LN_MAXQ; ++i) {
    const int q = i * 64 + lane;
    if (q < nq) {
      const f32x4 g = *(const f32x4*)(gamma + q * 4), b = *(const f32x4*)(beta + q * 4);
      f32x4 y;
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = (v[i][e] - mean) * rstd * g[e] + b[e];
      const size_t idx = (size_t)row * C + (size_t)q * 4;
      if (out_f32) {
        *(f32x4*)((float*)out + idx) = y;
      } else {
        f16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)y[e];
        *(f16x4*)((half_t*)out + idx) = h;
      }
    }
  }
}
