// k_gemm.h - plane-fed split-precision GEMM for gfx950: every nn.Linear of the SDMatte transformer blocks
// (reference call sites: replace.py:232-362 -> diffusers Transformer2DModel / BasicTransformerBlock: proj_in, attn1/attn2
// to_q / to_k / to_v / to_out, GEGLU proj, ff.net.2, proj_out; SURVEY.md 2.2 "Linear / GEMM").
//
// Why a second GEMM kernel.  The register-staged 1x1 form of k_conv.h converts its fp32 activations to the MFMA operand
// pairs INSIDE the GEMM - once per output-channel tile (20 times for a GEGLU projection), between four barriers per 32-channel
// chunk - and spends three fp16 MFMAs per product.  Here the operands arrive PRE-SPLIT from the kernels that produce them
// (LayerNorm, GroupNorm apply, the attention and GEGLU epilogues), the GEMM moves them global -> LDS by LDS-DMA only (no VALU
// conversion, no staging registers), and the two residual terms of the split product run as ONE K = 64 fp8 MFMA:
//      x . w  ~=  x_hi . w_hi  (2 x v_mfma_f32_32x32x16_f16 per 32 channels)
//              +  [x_lo8 | x8] . [w8 | w_lo8]  (1 x v_mfma_scale_f32_32x32x64_f8f6f4, e5m2 activations x e4m3 weights)
// i.e. 2 MFMA-times per product instead of 3, the arithmetic of the F8 conv kernels (k_conv.h).
//
// "P3" activation tensor [R rows][C channels], C % 32 == 0, 3 bytes per element, both planes BLOCKED so that every LDS-DMA instruction of the
// GEMM reads 1 KB of contiguous memory (whole cache lines: row-major planes cost 2x the bytes between L2 and the CU - a 32-channel chunk is half /
// a quarter of a 128-byte line per row - and the kernel is L2-bandwidth-bound then, profiles/r06_gemm_p3_lab.txt); rows padded to a multiple of 32:
//   HI plane  fp16, blocks of 16 rows x 32 channels: [R/16][C/32][run 4][row 16][8 x fp16]      hi = fp16(clamp(x, +-57344)); run = 8 consecutive channels
//   XL plane  u8,   blocks of 32 rows x 32 channels: [R/32][C/32][half 2][row 32][16 B]          e5m2((x - hi) * 2^11); half h holds the runs h and 2 + h
//                                   (channels h*8 .. +8 | 16 + h*8 .. +8): exactly the channels lane half h of a 32x32x16 MFMA holds over the two
//                                   K16 steps of the group - a block is the fp8 operand fragment of 32 rows in register order
//   e5m2(x), the other fp8 operand, is NOT stored: it is the top byte of hi (truncation instead of rounding: the term it enters,
//   x . w_lo, is 2^-11 of the product) and is cut out of the fp16 fragments with v_perm_b32 - 4 instructions per 32 rows x 32 channels.
//
// "W3" weight layout (derive_gemm_w3_kernel, from the canonical K16 tensors): per 32-channel chunk N*128 bytes =
//   WH [N/16][granule 4][col 16][8 x fp16]   unscaled fp16 high parts; granule g = channels (g>>1)*16 + (g&1)*8 .. +8
//   W8 [N/32][part 2][half 2][col 32][16 B]  part 0 = e4m3(w * s8), part 1 = e4m3((w - hi) * 2^11 * s8); the 16 bytes of lane half h
//                                            are the channels h*8 .. +8 and 16 + h*8 .. +8 (the order of the XL plane)
//   A tile's share of a chunk is two contiguous runs (BN * 64 bytes each): every LDS-DMA instruction copies 1 KB verbatim.
//
// Kernel: 256 x 128 (or 128 x 128 / 64 x 128) output tile per 4-wave block, TWO blocks per CU (80 KB of LDS each): while one block
// stores its tile (the CU accepts one 16-byte store instruction per ~65 cycles: 8 k cycles per 128 KB tile) the other multiplies.
// Two LDS stages; per chunk ONE raw barrier: wait own DMAs of chunk c -> barrier -> issue the DMAs of chunk c + 1 into the
// stage read in chunk c - 1 -> fragment reads + 12 MFMAs per 32 rows of chunk c.  LDS images: activations in 16-row blocks of
// [granule 4][row 16][16 B] (a DMA instruction = 16 rows x 64 contiguous bytes = whole 64-byte sectors; a fragment read = four
// 256-byte runs, conflict-free for ds_read_b128's lane groups), everything else is read back in the order the DMA wrote it.
// Accumulators are [channel][pixel] (weights are the MFMA's A operand): a register quad is four consecutive output channels of one
// row; v_permlane32_swap makes it eight, so every epilogue stores 16 bytes per lane:
//   EPI 0  fp32 [rows][ldo] (+bias, +fp32 residual); EPI 4 = 0 + the GroupNorm statistics of the consumer (partial rows, k_conv.h ConvParams::stats)
//   EPI 1  GEGLU u * gelu(g) -> P3          EPI 3  linear (+bias, +residual) -> P3
//   EPI 2  q | k | v operand planes of the split-precision attention (fp16 hi + e5m2 pair plane, ConvParams::out_f32 == 3)
#pragma once
#include "sdm_common.h"

// e5m2(x) is taken by TRUNCATION (the top byte of the fp16 high part): on average 9 % too small in magnitude (mean of delta / m for delta uniform in
// [0, 1/4), mantissa m log-uniform in [1, 2)).  The w_lo8 operand it multiplies carries the inverse: the term x . w_lo is then unbiased and its rms error
// equals that of round-to-nearest operands - at no cost in the kernel (measured on the emulator: max error 1.2e-4 -> see tests/test_emu_ops.py)
#ifndef P3_X8_TRUNC_GAIN
#define P3_X8_TRUNC_GAIN 1.097f
#endif

// byte position of channel c (0..31) of a 32-channel group inside the group's 32 XL bytes (half = pos >> 4)
SDM_DEV_INLINE int p3_xl_pos(int c) { const int j = c >> 3; return ((j & 1) << 4) | ((j >> 1) << 3) | (c & 7); }
SDM_HD_INLINE size_t p3_rows_pad(size_t rows) { return (rows + 31) & ~(size_t)31; }
// byte offset, inside the HI plane of a P3 tensor with C channels, of the 16-byte run of channels c .. c + 8 (c % 8 == 0) of a row
SDM_DEV_INLINE size_t p3_hi_off(size_t row, int C, int c) {
  return ((((row >> 4) * (size_t)(C >> 5)) + (size_t)(c >> 5)) << 10) + (size_t)((((c & 31) >> 3) << 8) + ((int)(row & 15) << 4));
}
// ... and, inside the XL plane, of the 8 residual bytes of that run
SDM_DEV_INLINE size_t p3_xl_off(size_t row, int C, int c) {
  const int pos = p3_xl_pos(c & 31);
  return ((((row >> 5) * (size_t)(C >> 5)) + (size_t)(c >> 5)) << 10) + (size_t)(((pos >> 4) << 9) + ((int)(row & 31) << 4) + (pos & 15));
}

// GELU with the exact-erf semantics of F.gelu (diffusers GEGLU) on a rational erf (Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 absolute): ~14 VALU
// instructions with two transcendentals where erff() costs several times that - the GEGLU epilogue evaluates 64 of them per lane and tile
SDM_DEV_INLINE float p3_gelu(float g) {
  const float z = fabsf(g) * 0.70710678118654752f;
  const float t = sdm_rcp(1.0f + 0.3275911f * z);
  const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
  const float er = 1.0f - poly * sdm_exp2(-z * z * SDM_LOG2E);
  return 0.5f * g * (1.0f + copysignf(er, g));
}

struct GemmP3Params {
  const half_t* a_hi; const unsigned char* a_xl;      // P3 activation [M][K]: HI plane, XL plane (behind rows_pad(M) * K * 2 bytes)
  long M; int K;                                      // K % 32 == 0
  int rows_per_img;                                   // != 0: row tiles are aligned to images of this many rows (statistics)
  const unsigned char* w; int N;                      // W3 weights, GEMM N = Cout_pad (multiple of 32)
  const float* bias;                                  // [N] or null
  void* out; int ldo;                                 // output base and its row stride in channels
  int n_valid;                                        // post-epilogue channels actually stored (multiple of 32)
  size_t out_lo_off;                                  // EPI 2: element offset of the pair plane; EPI 1 / 3: BYTE offset of the XL plane behind `out` (rows_pad * ldo * 2)
  int lo_cols;                                        // EPI 2: only channels < lo_cols get the pair plane
  const float* res; int ldr;                          // optional fp32 residual
  float* stats;                                       // EPI 4: [imgs][tiles_per_img * 2][ldo][2] partial {sum, sumsq} rows (k_conv.h ConvParams::stats)
  int sa, sb;                                         // E8M0 exponents of the fp8 operand scales (activations 2^-11, weights 2^-e8)
  int tiles_m, tiles_n, xcd_chunk, tiles_per_img;     // xcd_chunk > 0: XCD-aware order (block b -> XCD b % 8 owns M tiles [x*chunk, (x+1)*chunk))
  int ablate;                                         // bench only (sdm_bench_gemm_p3; results are garbage): 1 no MFMAs, 2 no DMAs behind the prologue, 4 no epilogue
};

// 16 fp32 values of a 32-row x 32-channel accumulator block (register r = channel (r&3) + 8*(r>>2) + 4*(lane>>5) of row lane&31)
// -> P3: returns this lane's two 16-byte hi vectors (channels 8*(2*q + h) .. +8 for q = 0 / 1) and its 16 XL bytes (the lane half's 16 bytes of the group)
SDM_DEV_INLINE void p3_pack_block(const float (&v)[16], u32x4 (&hi)[2], u32x4& xl) {
  unsigned int H[4][2], X[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float y[4], l[4];
    half_t hh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      y[e] = SDM_MED3(v[4 * g + e], -57344.0f, 57344.0f);
      hh[e] = (half_t)y[e];
      l[e] = (y[e] - (float)hh[e]) * 2048.0f;
    }
    f16x2 p0, p1;
    p0[0] = hh[0]; p0[1] = hh[1]; p1[0] = hh[2]; p1[1] = hh[3];
    H[g][0] = __builtin_bit_cast(unsigned int, p0); H[g][1] = __builtin_bit_cast(unsigned int, p1);
    int b = SDM_CVT_PK_BF8(l[0], l[1], 0, false);
    b = SDM_CVT_PK_BF8(l[2], l[3], b, true);
    X[g] = (unsigned int)b;
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    sdm_permlane32_swap(H[2 * q][0], H[2 * q + 1][0]);
    sdm_permlane32_swap(H[2 * q][1], H[2 * q + 1][1]);
    sdm_permlane32_swap(X[2 * q], X[2 * q + 1]);
    hi[q][0] = H[2 * q][0]; hi[q][1] = H[2 * q][1]; hi[q][2] = H[2 * q + 1][0]; hi[q][3] = H[2 * q + 1][1];
  }
  xl[0] = X[0]; xl[1] = X[1]; xl[2] = X[2]; xl[3] = X[3];
}

// NS = LDS stages (ring): the DMAs of chunk c + NS - 1 are issued when chunk c starts - NS - 1 chunks of landing time.  2 stages of the 256 x 128 tile
// leave room for two blocks per CU; deeper rings trade the second block for landing time (one block per CU from 3 stages of the 256-row tile).
template <int MT, int NT, int EPI, int NS = 2>
__global__ void __launch_bounds__(256, (NS * (2 * MT * 32 * 96 + 2 * NT * 32 * 128) <= 80 * 1024) ? 2 : 1) gemm_p3_kernel(GemmP3Params p) {
  constexpr int BM = 2 * MT * 32, BN = 2 * NT * 32;
  constexpr int A_HI = BM * 64, A_XL = BM * 32, B_HI = BN * 64, B_F8 = BN * 64;
  constexpr int OFF_AXL = A_HI, OFF_BHI = A_HI + A_XL, OFF_BF8 = OFF_BHI + B_HI, STAGE = OFF_BF8 + B_F8;
  // DMA instructions per wave and chunk (every wave issues the same number: the counted vmcnt below is an immediate)
  constexpr int PER = (BM / 16 + 3) / 4 + (BM / 32 + 3) / 4 + 2 * (BN / 16 / 4);
  static_assert(EPI != 1 || NT == 2, "GEGLU: a wave's 64 columns are one [u32 | g32] group");
  static_assert(NS >= 2 && NS * STAGE <= 160 * 1024 && (NS - 2) * PER <= 63, "LDS ring / vmcnt range");
  SDM_DYN_SMEM(smem);
  const int tx = (int)threadIdx.x, lane = tx & 63, wave = SDM_UNIFORM_I(tx >> 6);
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int nch = p.K >> 5;
  const unsigned int K = (unsigned int)p.K, N = (unsigned int)p.N;
  const long rows_pad = (long)p3_rows_pad((size_t)p.M);
  // Tiles of this block: the virtual ids blockIdx.x, + gridDim.x, ... (a persistent grid; gridDim.x % 8 == 0 keeps a block's tiles on its XCD's M range).
  // The block treats its (tile, chunk) pairs as ONE stream: the DMAs of a tile's first chunk are issued while the previous tile's last chunk is
  // multiplied, so they land underneath that tile's epilogue instead of in front of the first MFMA.
  const int vgrid = (p.xcd_chunk > 0) ? 8 * p.xcd_chunk * p.tiles_n : p.tiles_m * p.tiles_n;
  auto next_tile = [&](int v, int& mt, int& nt) -> int {      // first valid id >= v of this block's sequence, or -1 (padding ids of the XCD rounding are skipped)
    for (; v < vgrid; v += (int)gridDim.x) {
      if (p.xcd_chunk > 0) {
        const int j = v >> 3, ml = j / p.tiles_n;
        nt = j - ml * p.tiles_n;
        mt = (v & 7) * p.xcd_chunk + ml;
        if (mt < p.tiles_m) return v;
      } else {
        mt = v / p.tiles_n;
        nt = v - mt * p.tiles_n;
        return v;
      }
    }
    return -1;
  };
  auto tile_rows = [&](int mt, int& img, int& mti, long& m_end) -> long {      // first row of M tile mt (image-aligned tiles: rows_per_img != 0)
    img = 0; mti = mt; m_end = p.M;
    if (!p.rows_per_img) return (long)mt * BM;
    img = mt / p.tiles_per_img; mti = mt - img * p.tiles_per_img;
    m_end = (long)(img + 1) * p.rows_per_img;
    return (long)img * p.rows_per_img + (long)mti * BM;
  };
  const sdm_rsrc rs_w = sdm_make_rsrc(p.w, (unsigned int)nch * N * 128u);
  const unsigned int vo_w = (unsigned int)lane * 16u;
  // chunk c of the tile at row m0 (m0 % 32 == 0), channel n0: 16-row block i of HI at (i * K/32 + c) KB behind the tile's first block, 32-row block i of XL likewise
  auto issue = [&](long m0, int n0, int c, unsigned char* st) {
    const unsigned int rows_avail = (unsigned int)(rows_pad - m0 < (long)BM ? rows_pad - m0 : (long)BM);
    const sdm_rsrc rs_ahi = sdm_make_rsrc((const unsigned char*)p.a_hi + (size_t)m0 * K * 2, rows_avail * K * 2u);
    const sdm_rsrc rs_axl = sdm_make_rsrc(p.a_xl + (size_t)m0 * K, rows_avail * K);
    // (a region of fewer than 4 pieces is copied redundantly by the surplus waves - same source, same destination - so that every wave's count is PER)
#pragma unroll
    for (int i0 = 0; i0 < BM / 16; i0 += 4) {
      const int i = (i0 + wave) % (BM / 16);
      sdm_glds16_buf(rs_ahi, vo_w, ((unsigned int)i * (K >> 5) + (unsigned int)c) << 10, st + i * 1024);
    }
#pragma unroll
    for (int i0 = 0; i0 < BM / 32; i0 += 4) {
      const int i = (i0 + wave) % (BM / 32);
      sdm_glds16_buf(rs_axl, vo_w, ((unsigned int)i * (K >> 5) + (unsigned int)c) << 10, st + OFF_AXL + i * 1024);
    }
    const unsigned int wb = (unsigned int)c * N * 128u + (unsigned int)n0 * 64u;
#pragma unroll
    for (int i0 = 0; i0 < BN / 16; i0 += 4) {
      const int i = i0 + wave;
      sdm_glds16_buf(rs_w, vo_w, wb + (unsigned int)i * 1024u, st + OFF_BHI + i * 1024);
    }
#pragma unroll
    for (int i0 = 0; i0 < BN / 16; i0 += 4) {
      const int i = i0 + wave;
      sdm_glds16_buf(rs_w, vo_w, wb + N * 64u + (unsigned int)i * 1024u, st + OFF_BF8 + i * 1024);
    }
  };

  f32x16 acc[MT][NT];

  // fragment addresses inside a stage (constant for the tile)
  int wh_off[NT], w8_off[NT], ah_off[MT], ax_off[MT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int nb = wn * NT + j;
    wh_off[j] = OFF_BHI + (((nb * 2 + (l31 >> 4)) * 4 + h) * 256) + (lane & 15) * 16;
    w8_off[j] = OFF_BF8 + ((nb * 4 + h) * 512) + l31 * 16;
  }
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int rb = wm * MT + i;
    ah_off[i] = (((rb * 2 + (l31 >> 4)) * 4 + h) * 256) + (lane & 15) * 16;
    ax_off[i] = OFF_AXL + ((rb * 2 + h) * 512) + l31 * 16;
  }
  const int sa8 = p.sa, sb8 = p.sb;

  const bool ab_mm = (p.ablate & 1) != 0, ab_dma = (p.ablate & 2) != 0;
  // the issue side of the stream: tile v_iss (row m0_i, channel n0_i), next chunk ic
  int mt_c, nt_c, v_cur = next_tile((int)blockIdx.x, mt_c, nt_c);
  if (v_cur < 0) return;
  int v_iss = v_cur, ic = 0, st_cur = 0, st_nxt = 0, ahead = 0, n0_i = nt_c * BN;
  long m0_i;
  { int im, ti; long me; m0_i = tile_rows(mt_c, im, ti, me); }
  auto stream_issue = [&]() {
    if (v_iss < 0) return;
    issue(m0_i, n0_i, ic, smem + st_nxt * STAGE);
    st_nxt = (st_nxt + 1 == NS) ? 0 : st_nxt + 1;
    ++ahead;
    if (++ic == nch) {
      int mt, nt;
      ic = 0;
      v_iss = next_tile(v_iss + (int)gridDim.x, mt, nt);
      if (v_iss >= 0) { int im, ti; long me; m0_i = tile_rows(mt, im, ti, me); n0_i = nt * BN; }
    }
  };
#pragma unroll
  for (int s = 0; s < NS - 1; ++s) stream_issue();
  bool first = true;
  while (v_cur >= 0) {
  int img, mti;
  long m_end;
  const long m0 = tile_rows(mt_c, img, mti, m_end);
  const int n0 = nt_c * BN;
  const unsigned int rows_left = (unsigned int)((m_end - m0) < (long)BM ? (m_end - m0) : (long)BM);
  // the accumulators start from the fp32 residual (register quad = four consecutive channels of a row = one 16-byte load; rows / channels beyond the end
  // read 0): the loads fly while the tile's first chunk lands, and no epilogue waits for memory
  if ((EPI == 0 || EPI == 3 || EPI == 4) && p.res) {
    const sdm_rsrc rs_res = sdm_make_rsrc(p.res + (size_t)m0 * p.ldr, rows_left * (unsigned int)p.ldr * 4u);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = n0 + (wn * NT + j) * 32 + 8 * g + 4 * h;
          const unsigned int row = (unsigned int)((wm * MT + i) * 32 + l31);
          const f32x4 r4 = __builtin_bit_cast(f32x4, sdm_buffer_load16(rs_res, ch < p.n_valid ? row * (unsigned int)p.ldr * 4u + (unsigned int)ch * 4u : SDM_BUF_INVALID, 0));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = r4[e];
        }
#if !defined(SDM_EMU) && defined(__HIP_DEVICE_COMPILE__)
    // the loads are waited for HERE: hipcc does not count LDS-DMAs, so a wait it placed in front of the first MFMA - behind the DMAs of the next chunk -
    // would drain those as well
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(acc[i][j]));
#endif
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  }
  for (int c = 0; c < nch; ++c) {
    // this wave's share of the oldest chunk in flight has landed; the `ahead - 1` chunks behind it (PER instructions each) may stay in flight
#ifndef SDM_EMU
    if (NS > 2 && ahead - 1 >= NS - 2) { asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NS - 2) * PER) : "memory"); }
    else
#endif
      SDM_WAIT_VMCNT0();
    SDM_RAW_BARRIER();           // ... everybody's has, and every wave is past its reads of the previous chunk, whose slot the next DMAs overwrite
    --ahead;
    if (!(ab_dma && !first)) stream_issue();
    first = false;
    const unsigned char* st = smem + st_cur * STAGE;
    st_cur = (st_cur + 1 == NS) ? 0 : st_cur + 1;
    if (ab_mm && c > 0) continue;
    f16x8 wh[2][NT];
    i32x8 w8[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      wh[0][j] = *(const f16x8*)(st + wh_off[j]);
      wh[1][j] = *(const f16x8*)(st + wh_off[j] + 512);
      const i32x4 q0 = *(const i32x4*)(st + w8_off[j]), q1 = *(const i32x4*)(st + w8_off[j] + 1024);
      w8[j] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    }
    // row blocks in pairs: the four accumulators of a pair take the three operand sweeps in turn, so an accumulator is touched again only four MFMAs
    // (128+ cycles) later - back to back on one accumulator the matrix pipe waits for its own result
    constexpr int RP = (MT >= 2) ? 2 : 1;
#pragma unroll
    for (int i0 = 0; i0 < MT; i0 += RP) {
      f16x8 a0[RP], a1[RP];
      i32x8 a8[RP];
#pragma unroll
      for (int r = 0; r < RP; ++r) {
        a0[r] = *(const f16x8*)(st + ah_off[i0 + r]); a1[r] = *(const f16x8*)(st + ah_off[i0 + r] + 512);
        const i32x4 xl = *(const i32x4*)(st + ax_off[i0 + r]);
        // e5m2(x) = the top bytes of the fp16 high parts: 8 + 8 values of this lane half, in the order of the XL bytes
        const u32x4 u0 = __builtin_bit_cast(u32x4, a0[r]), u1 = __builtin_bit_cast(u32x4, a1[r]);
        a8[r][0] = xl[0]; a8[r][1] = xl[1]; a8[r][2] = xl[2]; a8[r][3] = xl[3];
        a8[r][4] = (int)sdm_perm_b32(u0[1], u0[0], 0x07050301u); a8[r][5] = (int)sdm_perm_b32(u0[3], u0[2], 0x07050301u);
        a8[r][6] = (int)sdm_perm_b32(u1[1], u1[0], 0x07050301u); a8[r][7] = (int)sdm_perm_b32(u1[3], u1[2], 0x07050301u);
      }
#pragma unroll
      for (int r = 0; r < RP; ++r)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i0 + r][j] = SDM_MFMA_32x32x16_F16(wh[0][j], a0[r], acc[i0 + r][j]);
#pragma unroll
      for (int r = 0; r < RP; ++r)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i0 + r][j] = SDM_MFMA_32x32x16_F16(wh[1][j], a1[r], acc[i0 + r][j]);
#pragma unroll
      for (int r = 0; r < RP; ++r)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i0 + r][j] = SDM_MFMA_32x32x64_F8A_BF8B(w8[j], a8[r], acc[i0 + r][j], sb8, sa8);
    }
  }

  // ---------------- epilogue: straight from the accumulators, 16 bytes per lane and store ----------------
  if (p.ablate & 4) { if (acc[0][0][0] == 123.456f) ((float*)p.out)[0] = 1.0f; }
  else {
  const float* bias = p.bias;
  f32x4 bq[NT][4];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int ch = n0 + (wn * NT + j) * 32 + 8 * g + 4 * h;
      bq[j][g] = (bias && ch < p.N) ? *(const f32x4*)(bias + ch) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
  if (EPI == 0 || EPI == 4) {
    const sdm_rsrc rs_out = sdm_make_rsrc((float*)p.out + (size_t)m0 * p.ldo, rows_left * (unsigned int)p.ldo * 4u);
    constexpr bool do_stats = (EPI == 4);
    // channel quad by channel quad, the MT row blocks of a quad back to back: a quad's statistics (per channel over this wave's MT * 32 rows) are complete
    // after its MT stores and need 8 registers, not 64 (the residual is already in the accumulators)
    const size_t prow = (size_t)img * (p.tiles_per_img * 2) + (size_t)mti * 2 + wm;
#pragma unroll
    for (int jg = 0; jg < NT * 4; ++jg) {
      const int j = jg >> 2, g = jg & 3;
      const int ch = n0 + (wn * NT + j) * 32 + 8 * g + 4 * h;
      float t1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, t2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const unsigned int row = (unsigned int)((wm * MT + i) * 32 + l31);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bq[j][g][e];
        sdm_buffer_store16(__builtin_bit_cast(u32x4, v), rs_out, ch < p.n_valid ? row * (unsigned int)p.ldo * 4u + (unsigned int)ch * 4u : SDM_BUF_INVALID, 0);
        if (do_stats && row < rows_left) {
#pragma unroll
          for (int e = 0; e < 4; ++e) { t1[e] += v[e]; t2[e] += v[e] * v[e]; }
        }
        SDM_PIN_STORE_DATA(v);
      }
      if (do_stats) {
        // over the 32 lanes of the lane half; lane 0 / 32 writes its 4 channels of the partial row (image, M tile, wave row): [sum, sumsq] pairs = 32 contiguous bytes
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          t1[e] = sdm_sum_row16(t1[e]); t2[e] = sdm_sum_row16(t2[e]);
          t1[e] += __shfl_xor(t1[e], 16); t2[e] += __shfl_xor(t2[e], 16);
        }
        if (l31 == 0 && ch < p.n_valid) {
          float* dst = p.stats + (prow * p.ldo + ch) * 2;
          *(f32x4*)dst = f32x4{t1[0], t2[0], t1[1], t2[1]};
          *(f32x4*)(dst + 4) = f32x4{t1[2], t2[2], t1[3], t2[3]};
        }
      }
    }
  } else if (EPI == 1 || EPI == 3) {
    constexpr int NB = (EPI == 1) ? 1 : NT;                         // 32-channel output blocks per wave and row block
    // output planes (blocked, k_gemm.h header): descriptors over the tile's row blocks; a store instruction writes four 256-byte runs (hi) / one KB (xl)
    unsigned char* ohi = (unsigned char*)p.out;
    const unsigned int LC = (unsigned int)(p.ldo >> 5);                  // 32-channel chunks per output row
    const unsigned int rows_pad_left = (unsigned int)((long)p3_rows_pad((size_t)p.M) - m0 < (long)BM ? (long)p3_rows_pad((size_t)p.M) - m0 : (long)BM);
    const sdm_rsrc rs_hi = sdm_make_rsrc(ohi + (size_t)m0 * p.ldo * 2, rows_pad_left * (unsigned int)p.ldo * 2u);
    const sdm_rsrc rs_xl = sdm_make_rsrc(ohi + p.out_lo_off + (size_t)m0 * p.ldo, rows_pad_left * (unsigned int)p.ldo);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const unsigned int row = (unsigned int)((wm * MT + i) * 32 + l31);
#pragma unroll
      for (int jb = 0; jb < NB; ++jb) {
        const int cb = (EPI == 1) ? (n0 + wn * 64) / 2 : n0 + (wn * NT + jb) * 32;       // first output channel of the block
        float v[16];
        if (EPI == 1) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float u = acc[i][0][r] + bq[0][r >> 2][r & 3], g = acc[i][1][r] + bq[1][r >> 2][r & 3];
            v[r] = u * p3_gelu(g);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = acc[i][jb][r] + bq[jb][r >> 2][r & 3];
        }
        u32x4 hi2[2], xl;
        p3_pack_block(v, hi2, xl);
        const bool ok = cb < p.n_valid && row < rows_left;
        const unsigned int hb = (((row >> 4) * LC + (unsigned int)(cb >> 5)) << 10) + ((row & 15u) << 4);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          sdm_buffer_store16(hi2[q], rs_hi, ok ? hb + (unsigned int)((2 * q + h) << 8) : SDM_BUF_INVALID, 0);
          SDM_PIN_STORE_DATA(hi2[q]);
        }
        sdm_buffer_store16(xl, rs_xl, ok ? (((row >> 5) * LC + (unsigned int)(cb >> 5)) << 10) + (unsigned int)(h << 9) + ((row & 31u) << 4) : SDM_BUF_INVALID, 0);
        SDM_PIN_STORE_DATA(xl);
      }
    }
  } else {   // EPI == 2: fp16 hi plane + e5m2 pair plane ([e5m2(x) x 4 | e5m2((x - hi) * 2^11) x 4] per 4 channels), the q | k | v operands of k_attn.h
    half_t* oh = (half_t*)p.out;
    const sdm_rsrc rs_hi = sdm_make_rsrc(oh + (size_t)m0 * p.ldo, rows_left * (unsigned int)p.ldo * 2u);
    const sdm_rsrc rs_pr = sdm_make_rsrc(oh + p.out_lo_off + (size_t)m0 * p.ldo, rows_left * (unsigned int)p.ldo * 2u);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const unsigned int row = (unsigned int)((wm * MT + i) * 32 + l31);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int cb = n0 + (wn * NT + j) * 32;
        unsigned int H[4][2], P[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float y[4], l[4];
          half_t hh[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            y[e] = SDM_MED3(acc[i][j][4 * g + e] + bq[j][g][e], -57344.0f, 57344.0f);
            hh[e] = (half_t)y[e];
            l[e] = (y[e] - (float)hh[e]) * 2048.0f;
          }
          f16x2 p0, p1;
          p0[0] = hh[0]; p0[1] = hh[1]; p1[0] = hh[2]; p1[1] = hh[3];
          H[g][0] = __builtin_bit_cast(unsigned int, p0); H[g][1] = __builtin_bit_cast(unsigned int, p1);
          int a = SDM_CVT_PK_BF8(y[0], y[1], 0, false), b = SDM_CVT_PK_BF8(l[0], l[1], 0, false);
          a = SDM_CVT_PK_BF8(y[2], y[3], a, true); b = SDM_CVT_PK_BF8(l[2], l[3], b, true);
          P[g][0] = (unsigned int)a; P[g][1] = (unsigned int)b;
        }
        const bool ok = cb < p.n_valid;
        const unsigned int ro = row * (unsigned int)p.ldo;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          sdm_permlane32_swap(H[2 * q][0], H[2 * q + 1][0]); sdm_permlane32_swap(H[2 * q][1], H[2 * q + 1][1]);
          sdm_permlane32_swap(P[2 * q][0], P[2 * q + 1][0]); sdm_permlane32_swap(P[2 * q][1], P[2 * q + 1][1]);
          const int ch = cb + 8 * (2 * q + h);
          const u32x4 hv = {H[2 * q][0], H[2 * q][1], H[2 * q + 1][0], H[2 * q + 1][1]};
          const u32x4 pv = {P[2 * q][0], P[2 * q][1], P[2 * q + 1][0], P[2 * q + 1][1]};
          sdm_buffer_store16(hv, rs_hi, ok ? (ro + (unsigned int)ch) * 2u : SDM_BUF_INVALID, 0);
          SDM_PIN_STORE_DATA(hv);
          sdm_buffer_store16(pv, rs_pr, (ok && ch < p.lo_cols) ? (ro + (unsigned int)ch) * 2u : SDM_BUF_INVALID, 0);
          SDM_PIN_STORE_DATA(pv);
        }
      }
    }
  }
  }   // epilogue
  v_cur = next_tile(v_cur + (int)gridDim.x, mt_c, nt_c);
  }   // tiles of this block
}

// ---- W3 layout of a Linear / 1x1 weight from its canonical K16 tensors (k_hi / k_lo: [Cin_pad/16][Cout_pad][16], w * 2^w_exp = hi + lo).
//      One thread per 16-byte granule.  Value semantics as derive_conv_weight_f8_kernel (k_conv.h): fp16 high parts UNSCALED (inv_s = 2^-w_exp),
//      fp8 parts scaled by the layer's own s8 = 2^e8 (largest power of two with max|w| * s8 <= 448). ----
__global__ void derive_gemm_w3_kernel(const half_t* __restrict__ k_hi, const half_t* __restrict__ k_lo, unsigned char* __restrict__ wd, int Cin_pad, int Cout_pad,
                                      float inv_s, float s8) {
  const size_t per_chunk = (size_t)Cout_pad * 8;
  const size_t total = (size_t)(Cin_pad / 32) * per_chunk;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int chunk = (int)(idx / per_chunk);
    const int rem = (int)(idx - (size_t)chunk * per_chunk);
    auto src = [&](int c, int co) { return ((size_t)(c / 16) * Cout_pad + co) * 16 + (c % 16); };
    unsigned char* dst = wd + idx * 16;
    if (rem < Cout_pad * 4) {
      const int cb16 = rem >> 6, r2 = rem & 63, g = r2 >> 4, co = cb16 * 16 + (r2 & 15);
      const int c0 = chunk * 32 + (g >> 1) * 16 + (g & 1) * 8;
      f16x8 hv;
#pragma unroll
      for (int e = 0; e < 8; ++e) hv[e] = (half_t)((float)k_hi[src(c0 + e, co)] * inv_s);
      *(f16x8*)dst = hv;
    } else {
      const int rem2 = rem - Cout_pad * 4;
      const int nb = rem2 >> 7, r3 = rem2 & 127, part = r3 >> 6, hh = (r3 >> 5) & 1, co = nb * 32 + (r3 & 31);
      float v[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const int c = chunk * 32 + (b < 8 ? hh * 8 + b : 16 + hh * 8 + (b - 8));
        const size_t si = src(c, co);
        const float hi = (float)k_hi[si], w = (hi + (float)k_lo[si]) * inv_s;
        const float hp = (float)(half_t)(hi * inv_s);
        const float r = ((part == 0) ? w : (w - hp) * (2048.0f * P3_X8_TRUNC_GAIN)) * s8;
        v[b] = fminf(fmaxf(r, -448.0f), 448.0f);
      }
      u32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        int pk = SDM_CVT_PK_FP8(v[q * 4], v[q * 4 + 1], 0, false);
        pk = SDM_CVT_PK_FP8(v[q * 4 + 2], v[q * 4 + 3], pk, true);
        o[q] = (unsigned int)pk;
      }
      *(u32x4*)dst = o;
    }
  }
}

// ---- P3 producers.  8 consecutive channels per thread: 16 bytes of fp16 high parts + 8 XL bytes, each at its place in the blocked planes ----
SDM_DEV_INLINE void p3_store8(const float (&y)[8], unsigned char* hi_base, unsigned char* xl_base, size_t row, int C, int c) {
  f16x8 hv;
  float l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float t = SDM_MED3(y[e], -57344.0f, 57344.0f);
    hv[e] = (half_t)t;
    l[e] = (t - (float)hv[e]) * 2048.0f;
  }
  *(f16x8*)(hi_base + p3_hi_off(row, C, c)) = hv;
  int b0 = SDM_CVT_PK_BF8(l[0], l[1], 0, false), b1 = SDM_CVT_PK_BF8(l[4], l[5], 0, false);
  b0 = SDM_CVT_PK_BF8(l[2], l[3], b0, true); b1 = SDM_CVT_PK_BF8(l[6], l[7], b1, true);
  u32x2 o;
  o[0] = (unsigned int)b0; o[1] = (unsigned int)b1;
  *(u32x2*)(xl_base + p3_xl_off(row, C, c)) = o;
}

// fp32 [rows][C] -> P3 (sources whose producer does not emit planes itself).  One wave per 32 rows x 32 channels: lane (row l & 31, half l >> 5) reads its two
// 8-channel runs (whole 128-byte lines per row over the wave's loads) and writes two hi granules - per store instruction four 256-byte runs - and its 16
// XL bytes - one contiguous KB per instruction.  Rows beyond `rows` (padding of the last block) are written as zeros.
__global__ void __launch_bounds__(256) to_p3_kernel(const float* __restrict__ x, unsigned char* __restrict__ out, long rows, int C) {
  const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
  const long rb = (long)((rows + 31) >> 5), units = rb * (C >> 5);
  unsigned char* xlp = out + p3_rows_pad((size_t)rows) * (size_t)C * 2;
  for (long u = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < units; u += (long)gridDim.x * (blockDim.x >> 6)) {
    const long b = u / (C >> 5);
    const int ck = (int)(u - b * (C >> 5));
    const long row = b * 32 + r;
    float y[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f}, bq = a;
      if (row < rows) {
        const float* src = x + (size_t)row * C + ck * 32 + k * 16 + h * 8;
        a = *(const f32x4*)src; bq = *(const f32x4*)(src + 4);
      }
      y[k][0] = a[0]; y[k][1] = a[1]; y[k][2] = a[2]; y[k][3] = a[3]; y[k][4] = bq[0]; y[k][5] = bq[1]; y[k][6] = bq[2]; y[k][7] = bq[3];
    }
    u32x4 xo;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      f16x8 hv;
      float l[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float t = SDM_MED3(y[k][e], -57344.0f, 57344.0f);
        hv[e] = (half_t)t;
        l[e] = (t - (float)hv[e]) * 2048.0f;
      }
      *(f16x8*)(out + p3_hi_off((size_t)row, C, ck * 32 + k * 16 + h * 8)) = hv;
      int b0 = SDM_CVT_PK_BF8(l[0], l[1], 0, false), b1 = SDM_CVT_PK_BF8(l[4], l[5], 0, false);
      b0 = SDM_CVT_PK_BF8(l[2], l[3], b0, true); b1 = SDM_CVT_PK_BF8(l[6], l[7], b1, true);
      xo[2 * k] = (unsigned int)b0; xo[2 * k + 1] = (unsigned int)b1;
    }
    *(u32x4*)(xlp + (((size_t)b * (C >> 5) + ck) << 10) + (h << 9) + (r << 4)) = xo;
  }
}

// P3 -> fp32 (tests: what the GEMM's split operands represent, hi + xl * 2^-11)
__global__ void __launch_bounds__(256) from_p3_kernel(const unsigned char* __restrict__ in, float* __restrict__ x, long rows, int C) {
  const long n = rows * C;
  const unsigned char* xl = in + p3_rows_pad((size_t)rows) * (size_t)C * 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long row = i / C;
    const int c = (int)(i - row * C);
    const float hi = (float)*(const half_t*)(in + p3_hi_off((size_t)row, C, c & ~7) + (c & 7) * 2);
    const unsigned int b = xl[p3_xl_off((size_t)row, C, c & ~7) + (c & 7)];
    const f16x2 t = __builtin_bit_cast(f16x2, b << 8);      // an e5m2 byte is the top byte of an fp16
    x[i] = hi + (float)t[0] * (1.0f / 2048.0f);
  }
}

// LayerNorm over the last dim (F.layer_norm, eps inside the root; BasicTransformerBlock norm1 / norm2 / norm3) with P3 output.  FOUR rows per wave -
// lane (r = lane & 3, u = lane >> 2) owns the 8-channel runs v = 16 i + u of row 4 w + r - so that a store instruction covers four consecutive rows of
// sixteen runs: in the blocked planes the four rows of a run are 64 contiguous bytes (whole sectors; one row per wave wrote 64 separate 16-byte
// pieces per instruction and cost LayerNorm +35 %).  Loads stay whole lines (512 contiguous bytes per row and instruction).  Two-pass statistics in
// registers as layernorm_kernel (k_norm.h); the row sums cross the 16 lanes of a row with four xor-shuffles.  A block = 16 rows = one row block.
#define SDM_LNP_MAXI 10      /* C <= 1280 */
__global__ void __launch_bounds__(256) layernorm_p3_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           unsigned char* __restrict__ out, long rows, int C, float eps) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 3, u = lane >> 2;
  const long row = ((long)blockIdx.x * 4 + wave) * 4 + r;
  const bool active = row < rows;
  const long rr = active ? row : rows - 1;     // keep every lane in the shuffles
  const int nv = C / 8;
  f32x4 v[SDM_LNP_MAXI][2];
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < SDM_LNP_MAXI; ++i) {
    const int q = i * 16 + u;
    v[i][0] = v[i][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    if (q < nv) {
      const float* src = x + (size_t)rr * C + (size_t)q * 8;
      v[i][0] = *(const f32x4*)src; v[i][1] = *(const f32x4*)(src + 4);
      s += ((v[i][0][0] + v[i][0][1]) + (v[i][0][2] + v[i][0][3])) + ((v[i][1][0] + v[i][1][1]) + (v[i][1][2] + v[i][1][3]));
    }
  }
#pragma unroll
  for (int m = 32; m >= 4; m >>= 1) s += __shfl_xor(s, m);
  const float mean = s / (float)C;
  float qs = 0.0f;
#pragma unroll
  for (int i = 0; i < SDM_LNP_MAXI; ++i)
    if (i * 16 + u < nv) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[i][k][e] - mean; qs += d * d; }
    }
#pragma unroll
  for (int m = 32; m >= 4; m >>= 1) qs += __shfl_xor(qs, m);
  const float rstd = 1.0f / sqrtf(qs / (float)C + eps);
  if (!active) return;
  unsigned char* xl = out + p3_rows_pad((size_t)rows) * (size_t)C * 2;
#pragma unroll
  for (int i = 0; i < SDM_LNP_MAXI; ++i) {
    const int q = i * 16 + u;
    if (q < nv) {
      float y[8];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const f32x4 g = *(const f32x4*)(gamma + q * 8 + 4 * k), b = *(const f32x4*)(beta + q * 8 + 4 * k);
#pragma unroll
        for (int e = 0; e < 4; ++e) y[4 * k + e] = (v[i][k][e] - mean) * rstd * g[e] + b[e];
      }
      p3_store8(y, out, xl, (size_t)row, C, q * 8);
    }
  }
}
