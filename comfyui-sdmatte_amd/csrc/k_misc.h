// k_misc.h - pre/post-processing and glue kernels of the SDMatte path (all HBM-bound, tiny).
//
//  * resize_aa_sample(): torchvision Resize on tensors = F.interpolate(bilinear, align_corners=False,
//    antialias=True) (sdmatte_nodes.py:204-214,362).  Separable triangle filter with
//    support = max(scale,1), normalised weights - the ATen `_upsample_bilinear2d_aa` recipe.
//  * prep_image_kernel / prep_trimap_kernel: sdmatte_nodes.py:339-353 (BHWC -> resize -> (x-0.5)/0.5;
//    trimap -> resize -> *2-1, `repeat(1,3,1,1)` of meta_arch.py:141) into the engine's NHWC16 fp16 layout.
//  * mask_bias_kernel: meta_arch.py:200-204 + replace.py:401-403 + replace.py:56-63: level-k additive key
//    bias = (1 - m[2^k*8 i, 2^k*8 j]) * -10000 (times log2e for the 2^x softmax).
//  * alpha_out_kernel: meta_arch.py:258-260 (mean over the 3 decoded channels, clip(-1,1), (x+1)/2).
//  * resize_planes_kernel: sdmatte_nodes.py:362-363 (resize to (H,W), clamp(0,1)).
#pragma once
#include "sdm_common.h"

// Weighted antialiased bilinear sample of a single-channel strided plane.
template <typename F>
SDM_DEV_INLINE float resize_aa_sample(F fetch, int in_h, int in_w, int out_h, int out_w, int oy, int ox) {
  const float sy = (float)in_h / (float)out_h, sx = (float)in_w / (float)out_w;
  const float supy = sy >= 1.0f ? sy : 1.0f, supx = sx >= 1.0f ? sx : 1.0f;
  const float invy = sy >= 1.0f ? 1.0f / sy : 1.0f, invx = sx >= 1.0f ? 1.0f / sx : 1.0f;
  const float cy = sy * ((float)oy + 0.5f), cx = sx * ((float)ox + 0.5f);
  int ymin = (int)(cy - supy + 0.5f); if (ymin < 0) ymin = 0;
  int ymax = (int)(cy + supy + 0.5f); if (ymax > in_h) ymax = in_h;
  int xmin = (int)(cx - supx + 0.5f); if (xmin < 0) xmin = 0;
  int xmax = (int)(cx + supx + 0.5f); if (xmax > in_w) xmax = in_w;
  float wys = 0.0f, wxs = 0.0f;
  for (int y = ymin; y < ymax; ++y) { float w = 1.0f - fabsf(((float)y - cy + 0.5f) * invy); wys += w > 0.0f ? w : 0.0f; }
  for (int x = xmin; x < xmax; ++x) { float w = 1.0f - fabsf(((float)x - cx + 0.5f) * invx); wxs += w > 0.0f ? w : 0.0f; }
  float acc = 0.0f;
  for (int y = ymin; y < ymax; ++y) {
    float wy = 1.0f - fabsf(((float)y - cy + 0.5f) * invy); wy = wy > 0.0f ? wy / wys : 0.0f;
    float row = 0.0f;
    for (int x = xmin; x < xmax; ++x) {
      float wx = 1.0f - fabsf(((float)x - cx + 0.5f) * invx); wx = wx > 0.0f ? wx / wxs : 0.0f;
      row += wx * fetch(y, x);
    }
    acc += wy * row;
  }
  return acc;
}

// one NHWC16 pixel: channels 0..2 = (c0, c1, c2), 3..15 = 0; fp16 (fast graph) or fp32 (precise graph: the first conv splits it)
SDM_DEV_INLINE void prep_store16(void* out, size_t pix, float c0, float c1, float c2, int out_f32) {
  if (out_f32) {
    f32x4 a = {c0, c1, c2, 0.0f}, z = {0.0f, 0.0f, 0.0f, 0.0f};
    f32x4* o = (f32x4*)((float*)out + pix * 16);
    o[0] = a; o[1] = z; o[2] = z; o[3] = z;
  } else {
    f16x8 a, z;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = (half_t)0.0f; z[e] = (half_t)0.0f; }
    a[0] = (half_t)c0; a[1] = (half_t)c1; a[2] = (half_t)c2;
    *(f16x8*)((half_t*)out + pix * 16) = a;
    *(f16x8*)((half_t*)out + pix * 16 + 8) = z;
  }
}

// image fp32 [B,H,W,3] in [0,1] -> NHWC16 [B,S,S,16]: ch0..2 = (resize(x)-0.5)/0.5, ch3..15 = 0
__global__ void prep_image_kernel(const float* __restrict__ img, void* __restrict__ out, int out_f32, int B, int H, int W, int S) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * S * S) return;
  const int ox = i % S, oy = (i / S) % S, b = i / ((long)S * S);
  float v[3];
  for (int c = 0; c < 3; ++c) {
    const float* pl = img + (size_t)b * H * W * 3 + c;
    float x;
    if (H == S && W == S) x = pl[((size_t)oy * W + ox) * 3];
    else x = resize_aa_sample([&](int y, int xx) { return pl[((size_t)y * W + xx) * 3]; }, H, W, S, S, oy, ox);
    v[c] = (x - 0.5f) / 0.5f;
  }
  prep_store16(out, (size_t)i, v[0], v[1], v[2], out_f32);
}

// trimap fp32 [B,H,W] in [0,1] -> t = resize(x)*2-1: NHWC16 fp16 (ch0..2 = t) and fp32 plane [B,S,S]
__global__ void prep_trimap_kernel(const float* __restrict__ tri, void* __restrict__ out, int out_f32, float* __restrict__ plane, int B, int H,
                                   int W, int S) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * S * S) return;
  const int ox = i % S, oy = (i / S) % S, b = i / ((long)S * S);
  const float* pl = tri + (size_t)b * H * W;
  float x;
  if (H == S && W == S) x = pl[(size_t)oy * W + ox];
  else x = resize_aa_sample([&](int y, int xx) { return pl[(size_t)y * W + xx]; }, H, W, S, S, oy, ox);
  const float t = x * 2.0f - 1.0f;
  plane[i] = t;
  prep_store16(out, (size_t)i, t, t, t, out_f32);
}

// already pre-processed core-API inputs: image fp32 NCHW [B,3,S,S], trimap fp32 [B,1,S,S] (in [-1,1])
__global__ void prep_nchw_kernel(const float* __restrict__ img, const float* __restrict__ tri, void* __restrict__ out_img,
                                 void* __restrict__ out_tri, int out_f32, float* __restrict__ plane, int B, int SH, int SW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long hw = (long)SH * SW;
  if (i >= B * hw) return;
  const long b = i / hw, pq = i % hw;
  const float tv = tri[b * hw + pq];
  plane[i] = tv;
  prep_store16(out_img, (size_t)i, img[(b * 3 + 0) * hw + pq], img[(b * 3 + 1) * hw + pq], img[(b * 3 + 2) * hw + pq], out_f32);
  prep_store16(out_tri, (size_t)i, tv, tv, tv, out_f32);
}

// bias[level][b][i*wk + j] = (1 - (t[b][8*s*i][8*s*j] + 1)/2) * mask_value * log2e, s = 2^level, (hk, wk) = (SH/8, SW/8) >> level.
// The reference only admits square latents (replace.py:57-60 asserts perfect squares); the stride-2^k pick is the same rule
// on a rectangle.
__global__ void mask_bias_kernel(const float* __restrict__ plane, float* __restrict__ bias, int B, int SH, int SW, int level, float mask_value,
                                 float mult) {
  const int hk = (SH / 8) >> level, wk = (SW / 8) >> level, s = 1 << level;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * hk * wk) return;
  const int j = i % wk, ii = (i / wk) % hk, b = i / ((long)hk * wk);
  const float t = plane[((size_t)b * SH + (size_t)8 * s * ii) * SW + 8 * s * j];
  const float m = (t + 1.0f) / 2.0f;
  bias[i] = ((1.0f - m) * mask_value) * mult;
}

// decoded fp32 [B,S,S,4] (3 real channels) -> alpha fp32 [B,S,S] = (clip(mean3,-1,1)+1)/2
__global__ void alpha_out_kernel(const float* __restrict__ dec, float* __restrict__ alpha, long npix) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  const f32x4 d = *(const f32x4*)(dec + (size_t)i * 4);
  float m = (d[0] + d[1] + d[2]) / 3.0f;
  m = fminf(fmaxf(m, -1.0f), 1.0f);
  alpha[i] = (m + 1.0f) / 2.0f;
}

// fp32 planes [P,Hin,Win] -> antialiased bilinear resize to [P,Hout,Wout], optional clamp(0,1)
// (post-processing of sdmatte_nodes.py:362-363 uses P = B, clamp = 1)
__global__ void resize_planes_kernel(const float* __restrict__ in, float* __restrict__ out, int P, int Hin, int Win, int Hout,
                                     int Wout, int clamp01) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)P * Hout * Wout) return;
  const int ox = i % Wout, oy = (i / Wout) % Hout, b = i / ((long)Hout * Wout);
  const float* pl = in + (size_t)b * Hin * Win;
  float x;
  if (Hin == Hout && Win == Wout) x = pl[(size_t)oy * Win + ox];
  else x = resize_aa_sample([&](int y, int xx) { return pl[(size_t)y * Win + xx]; }, Hin, Win, Hout, Wout, oy, ox);
  if (clamp01) x = fminf(fmaxf(x, 0.0f), 1.0f);
  out[i] = x;
}

// Node tail at the ORIGINAL resolution (sdmatte_nodes.py:365-397): mask_refine with the input trimap, then the output image.
//   refine: fg = tri > c; bg = tri < 1-c; unknown = !(fg|bg); a[bg] = 0; a[fg] = clamp(1.2*a, 0, 1); a[(a < 0.3) & unknown] = 0
//   mode 0 alpha_only : matted = zeros_like(image) [B,H,W,3];  mode 1 matted_rgba : cat(image, a) [B,H,W,4];
//   mode 2 matted_rgb : image * ((tri > 0.2) & (a > 0.1)) [B,H,W,3]
// Every operation is a single fp32 compare / multiply / select, so the result is bit-identical to the reference's CPU tensor ops.
__global__ void refine_compose_kernel(const float* __restrict__ img, const float* __restrict__ tri, float* __restrict__ alpha,
                                      float* __restrict__ matted, long npix, int mode, int refine, float c, float one_minus_c) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npix) return;
  float a = alpha[i];
  // the trimap may be smaller than the image when nothing below reads it (mask_refine off, alpha_only / matted_rgba: the reference
  // indexes the alpha with the trimap only in mask_refine and matted_rgb, sdmatte_nodes.py:365-394) - then it is not loaded at all
  const float t = (refine || mode == 2) ? tri[i] : 0.0f;
  if (refine) {
    const bool fg = t > c, bg = t < one_minus_c, unk = !(fg || bg);
    if (bg) a = 0.0f;
    if (fg) a = fminf(fmaxf(a * 1.2f, 0.0f), 1.0f);
    if (a < 0.3f && unk) a = 0.0f;
    alpha[i] = a;
  }
  const float r = img[i * 3 + 0], g = img[i * 3 + 1], b = img[i * 3 + 2];
  if (mode == 1) {
    f32x4 o = {r, g, b, a};
    *(f32x4*)(matted + i * 4) = o;
  } else if (mode == 2) {
    const float gate = (t > 0.2f && a > 0.1f) ? 1.0f : 0.0f;
    matted[i * 3 + 0] = r * gate; matted[i * 3 + 1] = g * gate; matted[i * 3 + 2] = b * gate;
  } else {
    matted[i * 3 + 0] = 0.0f; matted[i * 3 + 1] = 0.0f; matted[i * 3 + 2] = 0.0f;
  }
}

// fp32 [n] -> the operand planes of the split-precision attention (what the producing GEMM's epilogue writes in the engine,
// ConvParams::out_f32): hi = fp16(x * mult) and, behind it, either lo = fp16(x * mult - hi) (mode 2) or, per 4 values,
// [e5m2(y) x 4 | e5m2((y - hi) * 2^11) x 4] (mode 3).  Test hook (sdm_op_attention_split).
__global__ void split_planes_kernel(const float* __restrict__ x, half_t* __restrict__ hi, half_t* __restrict__ lo, long n, float mult, int mode) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  f16x4 oh, ol;
  float xx[4], xl[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    float y = (i4 + e < n) ? x[i4 + e] * mult : 0.0f;
    if (mode == 3) y = fminf(fmaxf(y, -57344.0f), 57344.0f);
    oh[e] = (half_t)y;
    ol[e] = (half_t)(y - (float)oh[e]);
    xx[e] = y; xl[e] = (y - (float)oh[e]) * 2048.0f;
  }
  *(f16x4*)(hi + i4) = oh;
  if (mode == 3) {
    int a = SDM_CVT_PK_BF8(xx[0], xx[1], 0, false), b = SDM_CVT_PK_BF8(xl[0], xl[1], 0, false);
    a = SDM_CVT_PK_BF8(xx[2], xx[3], a, true); b = SDM_CVT_PK_BF8(xl[2], xl[3], b, true);
    u32x2 pr;
    pr[0] = (unsigned int)a; pr[1] = (unsigned int)b;
    *(u32x2*)(lo + i4) = pr;
  } else {
    *(f16x4*)(lo + i4) = ol;
  }
}

__global__ void scale_copy_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float mult) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] * mult;
}

// bench helpers only: pseudo-random fill (realistic operand toggling; constant data lets the chip clock ~25 % higher
// than it does on real activations, which makes micro-benchmarks lie)
__global__ void fill_random_f16_kernel(half_t* __restrict__ p, long n, unsigned int seed, float amp) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned int h = (unsigned int)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = (half_t)(((float)(h & 0xFFFF) * (1.0f / 32768.0f) - 1.0f) * amp);
  }
}
__global__ void fill_random_f32_kernel(float* __restrict__ p, long n, unsigned int seed, float amp) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned int h = (unsigned int)i * 2654435761u + seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = ((float)(h & 0xFFFF) * (1.0f / 32768.0f) - 1.0f) * amp;
  }
}

// ---- piecewise-constant inputs: the trimap half of the VAE encoder's batch.  A trimap is constant over large regions (background, sure
//      foreground, the inside of the unknown band), and a convolution of a constant region is a constant: every output pixel whose 3x3 window
//      lies inside one region has the same value, channel by channel - GroupNorm / SiLU / residual adds keep that (per-image, per-channel
//      maps).  The engine tracks, per activation tensor of that batch, a byte plane [N][H][W]: 0 = nothing known, c = 1..4 = "this pixel is
//      the image of a region of trimap value class c" (classes = the distinct values of the image's trimap, up to 4; rgb images are all 0).
//      A conv layer erodes the plane by its window; output TILES whose pixels are all of one class are not multiplied: ONE such tile per
//      (image, class) is computed by the conv kernel as usual, and const_tile_fill_kernel copies one of its pixels into the others - the
//      value every one of those pixels would have received (each output pixel is the same chain of operations on the same operands,
//      independent of its position), and the statistics they would have contributed (the representative tile's own partial sums).  Exact: nothing
//      is approximated; the engine option trimap_skip = 0 computes every tile. ----
#define SDM_CMASK_EMPTY 0xFFFFFFFFu

// class table of one trimap image: up to 4 distinct values among 64 x 64 sample points of the plane (a value that none of the samples hits gets
// no class - its pixels are simply multiplied).  One block per image; the table is built in LDS and written once: no global atomics.
__global__ __launch_bounds__(256) void cmask_table_kernel(const float* __restrict__ plane, unsigned int* __restrict__ table, int n0, int H, int W) {
  SDM_SHARED unsigned int tb[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 4) tb[tid] = SDM_CMASK_EMPTY;
  __syncthreads();
  const float* pl = plane + (size_t)b * H * W;
  for (int i = tid; i < 64 * 64; i += 256) {
    const int y = (int)(((long)(i >> 6) * H) >> 6), x = (int)(((long)(i & 63) * W) >> 6);
    const unsigned int bits = __builtin_bit_cast(unsigned int, pl[(size_t)y * W + x]);
    for (int k = 0; k < 4; ++k) {
      unsigned int cur = tb[k];
      if (cur == SDM_CMASK_EMPTY) cur = atomicCAS(&tb[k], SDM_CMASK_EMPTY, bits);      // returns the old value: EMPTY = this thread inserted
      if (cur == SDM_CMASK_EMPTY || cur == bits) break;
    }
  }
  __syncthreads();
  if (tid < 4) table[(size_t)(n0 + b) * 4 + tid] = tb[tid];
}

// class plane of the trimap images at the network's input: mask[(n0 + b)][y][x] = 1 + index of the pixel's value in the image's class table, 0 if absent
__global__ void cmask_init_kernel(const float* __restrict__ plane, unsigned char* __restrict__ mask, const unsigned int* __restrict__ table, int n0, int B, long HW) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW);
  const unsigned int bits = __builtin_bit_cast(unsigned int, plane[i]);
  const unsigned int* tb = table + (size_t)(n0 + b) * 4;
  int cls = 0;
#pragma unroll
  for (int k = 3; k >= 0; --k)
    if (tb[k] == bits && bits != SDM_CMASK_EMPTY) cls = k + 1;
  mask[(size_t)n0 * HW + i] = (unsigned char)cls;
}

// class plane through a 3x3 conv (stride 1 / 2, the conv's own padding: anything outside the image is padding, i.e. NOT the region's value):
// mout = c where all nine window pixels are c.  One block = one 8 x 32 output tile (256 threads); tile_flag (optional, [N][tiles]) = c when
// every pixel of the tile is c; tile_rep[n * 8 + c] = the smallest such tile index (atomicMin; preset to INT_MAX).  grid (tiles, N - n0).
__global__ __launch_bounds__(256) void cmask_conv_kernel(const unsigned char* __restrict__ min_, int Hin, int Win, unsigned char* __restrict__ mout, int Hout, int Wout,
                                                         int stride, int pad_t, int pad_l, unsigned char* __restrict__ tile_flag, int* __restrict__ tile_rep, int n0) {
  SDM_SHARED int red[8];
  const int tid = threadIdx.x, n = n0 + (int)blockIdx.y;      // images below n0 carry no classes: their planes are never read, their tile flags are preset to 0
  const int npx = (Wout + 31) / 32;
  const int mt = blockIdx.x, oy = (mt / npx) * 8 + (tid >> 5), ox = (mt % npx) * 32 + (tid & 31);
  int c = -1;                                    // -1: pixel outside the image (does not vote)
  if (oy < Hout && ox < Wout) {
    const unsigned char* mi = min_ + (size_t)n * Hin * Win;
    const int iy0 = oy * stride - pad_t, ix0 = ox * stride - pad_l;
    c = 0;
    if (iy0 >= 0 && ix0 >= 0 && iy0 + 2 < Hin && ix0 + 2 < Win) {
      c = mi[(size_t)iy0 * Win + ix0];
      for (int dy = 0; dy < 3; ++dy)
        for (int dx = 0; dx < 3; ++dx)
          if (mi[(size_t)(iy0 + dy) * Win + ix0 + dx] != c) c = 0;
    }
    mout[((size_t)n * Hout + oy) * Wout + ox] = (unsigned char)c;
  }
  if (!tile_flag) return;
  // all voting pixels equal and non-zero?  (min == max over the votes; non-voting pixels are neutral)
  int lo = c < 0 ? 255 : c, hi = c < 0 ? 0 : c;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) { lo = min(lo, __shfl_xor(lo, s)); hi = max(hi, __shfl_xor(hi, s)); }
  if ((tid & 63) == 0) { red[tid >> 6] = lo; red[4 + (tid >> 6)] = hi; }
  __syncthreads();
  if (tid == 0) {
    lo = min(min(red[0], red[1]), min(red[2], red[3]));
    hi = max(max(red[4], red[5]), max(red[6], red[7]));
    const int f = (lo == hi && lo > 0 && lo < 8) ? lo : 0;
    tile_flag[(size_t)n * gridDim.x + mt] = (unsigned char)f;
    if (f) atomicMin(&tile_rep[n * 8 + f], mt);
  }
}

// the tiles the F8 conv kernel left out (ConvParams::tile_flag): every pixel = pixel (0, 0) of the class's representative tile (fp32 NHWC, full
// 8 x 32 tiles), and the two partial statistics rows of the tile (one per 128-pixel wave row, the conv kernel's layout) = the representative's own rows.
// grid (tiles per image, N - n0), 256 threads.
__global__ __launch_bounds__(256) void const_tile_fill_kernel(float* __restrict__ out, int C, int Hout, int Wout, const unsigned char* __restrict__ tile_flag,
                                                              const int* __restrict__ tile_rep, float* __restrict__ stats, int wm_rows, int n0) {
  const int tid = threadIdx.x, n = n0 + (int)blockIdx.y, mt = blockIdx.x, tiles = gridDim.x;
  const int f = tile_flag[(size_t)n * tiles + mt];
  if (!f) return;
  const int rep = tile_rep[n * 8 + f];
  if (rep == mt) return;
  const int npx = Wout / 32, C4 = C / 4;
  const f32x4* src = (const f32x4*)(out + (((size_t)n * Hout + (size_t)(rep / npx) * 8) * Wout + (size_t)(rep % npx) * 32) * C);
  const int oy0 = (mt / npx) * 8, ox0 = (mt % npx) * 32;
  for (int r = 0; r < 8; ++r) {
    f32x4* dst = (f32x4*)(out + (((size_t)n * Hout + oy0 + r) * Wout + ox0) * C);
    for (int i = tid; i < 32 * C4; i += 256) dst[i] = src[i % C4];
  }
  if (stats) {
    // the partial statistics rows of the representative tile, as the conv kernel summed them: the left-out tile holds the same values in the same
    // positions, so the kernel would have produced exactly these sums for it (count x v would differ from the running fp32 sum in the last bits)
    for (int i = tid; i < wm_rows * C; i += 256) {
      const int w = i / C, ch = i - w * C;
      const f32x2 r2 = *(const f32x2*)(stats + ((((size_t)n * tiles + rep) * wm_rows + w) * C + ch) * 2);
      *(f32x2*)(stats + ((((size_t)n * tiles + mt) * wm_rows + w) * C + ch) * 2) = r2;
    }
  }
}
