// k_attn.h - flash-style attention for gfx950 (MFMA f16, fp32 online softmax, no L x L buffer).
//
// Reference semantics (fp32): probs = softmax(baddbmm(bias, q, k^T, beta=1, alpha=d^-1/2)) ; out = probs @ v
//   /root/reference/src/utils/replace.py:75-122 (custom_get_attention_scores) driven by diffusers'
//   AttnProcessor / SlicedAttnProcessor (sdmatte_nodes.py:331-337) for the 32 U-Net attentions, where
//   `bias` is the per-key additive trimap bias (1-m)*-10000 broadcast over queries and heads
//   (replace.py:401-403, 56-70), absent for cross-attention; and SDPA for the three single-head d=512
//   VAE mid-block attentions.
//
// Formulation ("swapped" QK^T, cdna_hip_programming.md T12): S^T = K.Q^T so that after the MFMA every
// lane owns ONE query column (q = lane&31) and 16 of the 32 keys of a key tile; softmax max/sum are
// in-lane reductions plus one exchange with lane^32; P^T is already in B-operand layout for
// O^T = V^T.P^T (the k-slot <-> key permutation only has to agree between A and B).  V is consumed
// through a pre-transposed copy V^T[d][key] so that its A fragments are two 8-byte LDS reads.
// Logits are computed in fp32 as s*scale*log2e + bias*log2e and exponentiated with v_exp_f32 (2^x).
#pragma once
#include "sdm_common.h"
#include "k_gemm.h"      // p3_pack_block: the attention output as the operand planes of the next GEMM

struct AttnParams {
  const half_t* q; long q_bs; int ldq;                     // q [b][row][head*D + d]
  const half_t* k; long k_bs; int ldk;                     // k [b][key][head*D + d]
  const half_t* vt; long vt_bs; long vt_hs; int ldvt;      // vt[b][head][d][key]  (key padded to ldvt, zero filled)
  const float* bias; long bias_bs;                         // bias[b][key] * log2e, or null
  half_t* o; long o_bs; int ldo;                           // o [b][row][head*D + d]
  int o_f32;                                               // 1: o is fp32 (same indexing), 0: fp16
  // o_p3 != 0 (with o_f32 == 1, Lq % 32 == 0, images contiguous): the result is written as the P3 operand planes of the GEMM that consumes it (k_gemm.h:
  // `o` = HI plane of a [batch * Lq][ldo] tensor, XL plane o_xl_off bytes behind it) from the wave's staged result tile instead of as fp32 rows; with a key split the partial sums stay fp32 and attn_combine_kernel writes the planes.
  // d = 64 kernels only (the d = 512 kernel sits at its register limit: its fp32 result takes one to_p3_kernel pass instead)
  int o_p3; long o_xl_off;
  // precise (split-fp16) variant, d = 64: q / k / vt are the HIGH parts; the low parts live in a second plane at these element
  // offsets (q = q_hi + q_lo etc., written by the producing GEMM's `out_f32 == 2` epilogue and by transpose_v on both planes)
  long q_lo, k_lo, vt_lo;
  int Lq, Lk;
  float scale_log2e;
  // Active key tiles (d = 64 only; null = all tiles).  tiles[b*tiles_bs] = n, tiles[b*tiles_bs + 1 + i] = index of the i-th 64-key tile of
  // image b that contains at least one key whose bias is within SDM_ATTN_SKIP_MARGIN of the image's largest bias.  Every other key has
  // bias <= max - margin, so its probability 2^(x - rowmax) underflows to EXACTLY 0 in fp32 - in the reference too (trimap keys carry
  // (1-m)*-10000: replace.py:401-403) - and its tile is not loaded at all.
  const int* tiles; int tiles_bs;
  // Key split (d = 64, fp32 output; grid.y = nsplit > 1): block row y walks the 64-key tiles [y, y + 1) * ceil(ntiles / nsplit) only and leaves
  // UNNORMALISED partial results - O^T sums in `o` (which then is a workspace: fp32 [nsplit][b][row][head*64 + d], split stride part_stride elements),
  // running maximum and denominator in part_ml[((y * batch + b) * heads + head) * Lq + row][2] - for attn_combine_kernel.  Launches whose blocks do not
  // fill the chip's block slots a whole number of times (one image: 640 four-wave blocks on 512 slots) run as twice or four times as many
  // short blocks.  The ranges are ranges of KEY TILES, so a walk along the active-tile list and the dense walk split at the same keys.
  int nsplit; long part_stride; float* part_ml;
  int batch, heads, nq_blocks, q_chunks;   // XCD-aware 1-D grid (attn_d64): see attn_block_coords
  int pp_flags; // attn_d64_pp_kernel: bit 0 = s_setprio 1 for the younger wave half
  int ablate;   // bench only (sdm_bench_attn): 1 skip softmax VALU, 2 skip PV MFMAs, 4 skip QK^T MFMAs, 8 skip K/V global prefetch; 0 in the engine
};

#define SDM_NEG_BIG (-1.0e30f)
// log2-domain margin below the largest key bias at which a key is dropped: 2^(x_k - rowmax) == 0 in fp32 needs x_k - rowmax < -149,
// i.e. the margin minus the spread of the raw logits q.k*scale*log2e (which would have to exceed 1850 to matter; the reference's own
// fp16 path overflows long before).  The trimap biases are 0 / -7213 / -14427 in this domain.
#define SDM_ATTN_SKIP_MARGIN 2000.0f

// the wave's staged fp32 result tile ([32 queries][64 channels] at pitch `ps` bytes in LDS) -> the P3 planes of the GEMM that consumes it (k_gemm.h).
// Lane (row = lane & 15, run = lane >> 4) x 2 row halves x 2 channel chunks: a store instruction writes one whole 1 KB block of the HI plane; the XL bytes go
// out as 256-byte runs.  (From the staged tile, not from the O^T accumulators: packing the planes in registers cost the pipelined kernels their schedule.)
SDM_DEV_INLINE void attn_store_p3(const AttnParams& p, const unsigned char* stf, int ps, int b, int q0, int head, int lane) {
  unsigned char* base = (unsigned char*)p.o;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int row = (pass >> 1) * 16 + (lane & 15), cch = (pass & 1) * 32 + (lane >> 4) * 8;
    const f32x4 a = *(const f32x4*)(stf + row * ps + cch * 4), c = *(const f32x4*)(stf + row * ps + cch * 4 + 16);
    const float y[8] = {a[0], a[1], a[2], a[3], c[0], c[1], c[2], c[3]};
    if (q0 + row < p.Lq) p3_store8(y, base, base + p.o_xl_off, (size_t)b * p.Lq + q0 + row, p.ldo, head * 64 + cch);
  }
}

// ------------------------------------------------------------------------------------------------
// d = 64, any number of heads (grid.y).  4 waves per block, QT x 32 queries per wave (QT = 2 halves the K / V^T fragment
// reads per MFMA), 64-key tiles, K / V^T / bias tiles DOUBLE-buffered in LDS: the global loads of tile t+1 are issued
// before the MFMAs of tile t, written to the other buffer afterwards, ONE barrier per tile.
// ------------------------------------------------------------------------------------------------
#define ATTN64_PK 144   /* K rows: conflict-free ds_read_b128 */
#define ATTN64_PV 136   /* V^T rows: 34 dwords -> the 32 lanes of a ds_read_b64 half-wave hit 64 distinct banks */
#define ATTN64_BUF (64 * ATTN64_PK + 64 * ATTN64_PV + 256)
#define ATTN64_SMEM (2 * ATTN64_BUF)
#define ATTN64P_BUF (2 * 64 * ATTN64_PK + 2 * 64 * ATTN64_PV + 256)   /* precise: K_hi | K_lo | V^T_hi | V^T_lo | bias */
#define ATTN64P_SMEM (2 * ATTN64P_BUF)

// XCD-aware work mapping.  The dispatcher places block id on XCD id % 8 (MI355X_MICROARCH.md, speed only - never needed for
// correctness).  All query blocks of one (image, head) should share an XCD so that its K / V^T (4 MB at L = 16384) stay in that
// XCD's private L2 instead of being re-fetched from Infinity Cache by all 8.  Work unit = (bh, query chunk) with q_chunks
// chunks per bh (units % 8 == 0); XCD x owns units [x*U/8, (x+1)*U/8) and walks them in order.
SDM_DEV_INLINE bool attn_block_coords(const AttnParams& p, int bid, int& b, int& head, int& qblk) {
  const int units = p.batch * p.heads * p.q_chunks;
  const int upx = units / 8;                                    // host guarantees units % 8 == 0
  const int qb = (p.nq_blocks + p.q_chunks - 1) / p.q_chunks;   // query blocks per unit
  const int xcd = bid & 7, s = bid >> 3;
  const int unit = xcd * upx + s / qb;
  const int bh = unit / p.q_chunks, chunk = unit % p.q_chunks;
  qblk = chunk * qb + s % qb;
  b = bh / p.heads; head = bh % p.heads;
  return qblk < p.nq_blocks;
}

// key split: the part of the tile walk that belongs to block row `sp` -> first walk index, number of tiles (block-uniform; may be 0)
SDM_DEV_INLINE void attn_split_range(const AttnParams& p, const int* tl, int ntiles, int nwalk_all, int sp, int& i0, int& nwalk) {
  i0 = 0; nwalk = nwalk_all;
  if (p.nsplit <= 1) return;
  const int per = (ntiles + p.nsplit - 1) / p.nsplit, t0 = sp * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  if (!tl) { i0 = t0 < ntiles ? t0 : ntiles; nwalk = t1 > i0 ? t1 - i0 : 0; return; }
  // sorted list of active tiles: lower bounds of t0 and t1 (binary search over wave-uniform scalar loads)
  auto lower = [&](int key) {
    int lo = 0, hi = nwalk_all;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (tl[1 + mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
  };
  i0 = lower(t0);
  nwalk = lower(t1) - i0;
}
// partial results of a block row that has no tile to walk: nothing summed, maximum at the floor (weight 0 in the combine)
SDM_DEV_INLINE void attn_write_empty_part(const AttnParams& p, int sp, int b, int head, int q0, int lane) {
  float* po = (float*)p.o + (size_t)sp * p.part_stride + (size_t)b * p.o_bs;
  for (int i = lane; i < 32 * 16; i += 64) {
    const int row = i >> 4, part = i & 15, qg = q0 + row;
    if (qg < p.Lq) *(f32x4*)(po + (size_t)qg * p.ldo + head * 64 + part * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  if (lane < 32 && q0 + lane < p.Lq) {
    float* ml = p.part_ml + ((((size_t)sp * p.batch + b) * p.heads + head) * p.Lq + q0 + lane) * 2;
    ml[0] = SDM_NEG_BIG; ml[1] = 0.0f;
  }
}

// PREC = 1 (precise mode): q, k and V^T arrive as fp16 pairs hi + lo and the probabilities are split the same way, every
// product is evaluated as hi.hi + lo.hi + hi.lo into the same fp32 accumulators (3 MFMAs instead of 1; the lo.lo term is 2^-22).
// NW = waves per block (4 or 8), 32 queries per wave: the 8-wave block shares every K / V^T tile among 256 queries - half the
// L2 / Infinity-Cache traffic and half the LDS staging work per MFMA of the 4-wave block (K | V of one head in hi | lo planes is
// 8.4 MB at 16384 keys: it does not fit the 4 MB L2 of an XCD, and 128-query blocks streamed it at ~5 TB/s) - at the same 2
// waves per SIMD; used when it still yields at least one block per CU.
// PREC = 2: only Q.K^T is split; P and V^T enter P.V as plain fp16 (V^T_lo is neither loaded nor staged).  The logits feed an
// exponential, the probabilities are averaged: on the full architecture the residual terms of P.V move alpha by 9e-6 (1.055e-4 ->
// 1.143e-4 at 512^2) and cost 16 % of the kernel (tests/tools/attn_pv_experiment.py) - PREC = 2 is what the engine uses.
// PREC = 3: PREC = 2 with the two residual terms of Q.K^T (k_lo.q + k.q_lo) on fp8 MFMAs.  The "low" plane of q and k then holds, per 4
// channels, [e5m2(x) x 4 | e5m2((x - hi) * 2^11) x 4] (written by the producing GEMM's epilogue, ConvParams::out_f32 == 3) - the same
// bytes at the same addresses as the fp16 low parts.  A dot product does not care about the order of its terms, so the 64 stored
// bytes of a key's 32 channels ARE an A operand of v_mfma_scale_f32_32x32x64_f8f6f4 (positions alternate k8 / k_lo8 in groups of
// 4), and the matching B operand is the query's 64 bytes with the two halves of every 8-byte group swapped (q_lo8 opposite k8, q8
// opposite k_lo8; built once per block): TWO K = 64 MFMAs per 32-key tile deliver both residual terms over d = 64, instead of eight
// fp16 MFMAs - 24 instead of 32 fp16-MFMA times per 64-key tile.  Every product pairs an x8 with an x_lo8 byte, so one operand scale
// of 2^-11 returns the sums to the unit of the fp16 accumulation; e5m2 has the range of fp16: nothing to calibrate or saturate.
template <int QT, int PREC = 0, int NW = 4>
__global__ void __launch_bounds__(64 * NW, 2) attn_d64_kernel(AttnParams p) {
  static_assert(!PREC || QT == 1, "the split-precision variant keeps one 32-query tile per wave");
  constexpr int PVS = (PREC == 1) ? 1 : 0;                       // P.V on split operands too
  constexpr int F8Q = (PREC == 3) ? 1 : 0;                       // residual terms of Q.K^T on fp8 (pair planes)
  constexpr int NTH = 64 * NW, VPT = 512 / NTH;                  // threads, 16-byte vectors per thread and tile (K and V^T: 512 each)
  SDM_DYN_SMEM(smem);
  constexpr int PK = ATTN64_PK, PV = ATTN64_PV;
  constexpr int BUF = PREC ? ATTN64P_BUF : ATTN64_BUF;
  constexpr int KLO = 64 * PK;                                   // PREC: byte offset of the K_lo tile behind K_hi
  constexpr int VOFF = (PREC ? 2 : 1) * 64 * PK;                 // V^T_hi tile
  constexpr int VLO = 64 * PV;                                   // PREC: V^T_lo behind V^T_hi
  constexpr int BOFF = VOFF + (PVS ? 2 : 1) * 64 * PV;           // bias row
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  int b, head, qblk;
  if (!attn_block_coords(p, blockIdx.x, b, head, qblk)) return;     // padding block of the XCD-aware grid
  const int q0 = qblk * (32 * NW * QT) + wave * (32 * QT);

  f16x8 qf[QT][4], qfl[PREC ? QT : 1][4];
  i32x8 q8p[2];                   // F8Q: B operands of the two residual MFMAs (channels 0-31 | 32-63 of this query, x8 / x_lo8 swapped)
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    int qrow = q0 + qt * 32 + l31;
    if (qrow > p.Lq - 1) qrow = p.Lq - 1;
    const half_t* qp = p.q + (size_t)b * p.q_bs + (size_t)qrow * p.ldq + head * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[qt][ks] = *(const f16x8*)(qp + ks * 16);
    if (PREC && !F8Q) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) qfl[PREC ? qt : 0][ks] = *(const f16x8*)(qp + p.q_lo + ks * 16);
    }
    if (F8Q) {
      // lane l: query l & 31, MFMA positions 32 * (l >> 5) .. +31 = the stored bytes of channels m * 32 + 16 * (l >> 5) .. +15
      const half_t* qb = p.q + (size_t)b * p.q_bs + p.q_lo + (size_t)qrow * p.ldq + head * 64 + hi * 16;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const i32x4 r0 = *(const i32x4*)(qb + m * 32), r1 = *(const i32x4*)(qb + m * 32 + 8);
        q8p[m] = i32x8{r0[1], r0[0], r0[3], r0[2], r1[1], r1[0], r1[3], r1[2]};
      }
    }
    // The logit scale d^-1/2 * log2(e) lives in Q.  The engine folds it into the to_q weights at load time (exact: one fp16
    // rounding of the GEMM result either way) and passes scale_log2e == 1; the stand-alone operator entry scales Q here.
    if (p.scale_log2e != 1.0f && !F8Q) {      // (F8Q: the pair plane cannot be rescaled here - its producer scales Q, as the engine always does)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (PREC) {                 // scale the fp32 value hi + lo, then split again
            const float qs = ((float)qf[qt][ks][j] + (float)qfl[PREC ? qt : 0][ks][j]) * p.scale_log2e;
            const half_t h = (half_t)qs;
            qf[qt][ks][j] = h;
            qfl[PREC ? qt : 0][ks][j] = (half_t)(qs - (float)h);
          } else {
            qf[qt][ks][j] = (half_t)((float)qf[qt][ks][j] * p.scale_log2e);
          }
        }
    }
  }
  f32x16 o[QT][2], ls[QT];        // ls: running softmax denominators, accumulated on the matrix pipe (ones . P^T)
  float m_i[QT];
  float lsum = 0.0f;              // PVS (P.V on split operands: three MFMAs per product, the matrix pipe is the bottleneck): the denominator is
                                  // summed on the VALU instead.  Every other variant is bound by the VALU stream of the softmax (hipcc also
                                  // SLP-packs these adds into v_pk_add_f32 + two v_mov each): the denominator rides the matrix pipe as the
                                  // sum of the SAME fp16 probabilities that enter P.V
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    m_i[qt] = SDM_NEG_BIG;
#pragma unroll
    for (int r = 0; r < 16; ++r) ls[qt][r] = 0.0f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[qt][dt][r] = 0.0f;
  }
  f16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.0f;

  const half_t* kbase = p.k + (size_t)b * p.k_bs + head * 64;
  const half_t* vbase = p.vt + (size_t)b * p.vt_bs + (size_t)head * p.vt_hs;
  const float* bbase = p.bias ? p.bias + (size_t)b * p.bias_bs : nullptr;
  const int ntiles = (p.Lk + 63) / 64;

  // one raw K / V^T / bias tile on its way from global memory to LDS.  AHEAD2 (F8Q): two of them alternate, so that the loads of tile
  // t+2 are requested at the top of iteration t and written to LDS in iteration t+1 - a load that misses the L2 has a whole iteration
  // to land instead of the Q.K^T + softmax part of one
  struct Raw {
    f16x8 k[VPT], v[VPT], kl[PREC ? VPT : 1], vl[PVS ? VPT : 1];
    float b;
    bool in;
  };
  constexpr bool AHEAD2 = F8Q && QT == 1 && NW == 8;      // (4-wave blocks stage two vectors per thread: a second set would spill)
  Raw ra, rb;
  ra.b = 0.0f; ra.in = true; rb.b = 0.0f; rb.in = true;
  // without a bias the load still happens (from the K tensor: >= Lk readable floats) and its value is discarded by a select
  const float* bsrc = bbase ? bbase : (const float*)(p.k + (size_t)b * p.k_bs);
  // all prefetch loads are UNCONDITIONAL (clamped rows): a conditional load makes hipcc wait vmcnt(0) per element.
  // Keys >= Lk are neutralised by the -1e30 bias (K rows) and by the zero padding of V^T.
  auto prefetch = [&](int t, Raw& r) {
    const int k0 = t * 64;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NTH;
      const int row = v >> 3, part = v & 7;
      int kr = k0 + row;
      if (kr > p.Lk - 1) kr = p.Lk - 1;
      r.k[i] = *(const f16x8*)(kbase + (size_t)kr * p.ldk + part * 8);
      r.v[i] = *(const f16x8*)(vbase + (size_t)row * p.ldvt + k0 + part * 8);
      if (PREC) r.kl[PREC ? i : 0] = *(const f16x8*)(kbase + p.k_lo + (size_t)kr * p.ldk + part * 8);
      if (PVS) r.vl[PVS ? i : 0] = *(const f16x8*)(vbase + p.vt_lo + (size_t)row * p.ldvt + k0 + part * 8);
    }
    // bias: UNCONDITIONAL raw load, consumed only in stage() after the MFMAs.  (`bbase ? bbase[kb] : 0` followed by a select
    // made hipcc branch around the load and wait vmcnt(0) right here - which also drains the four K / V^T prefetch loads
    // issued just above, i.e. every tile paid the full global-load latency.)
    int kb = k0 + (tid & 63);
    r.in = kb < p.Lk;
    if (!r.in) kb = p.Lk - 1;
    r.b = bsrc[kb];
  };
  auto stage = [&](int buf, const Raw& r) {
    unsigned char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NTH;
      const int row = v >> 3, part = v & 7;
      *(f16x8*)(base + row * PK + part * 16) = r.k[i];
      // V^T rows are only 8-byte aligned (pitch 136): two ds_write_b64
      f16x4 lo, hi4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = r.v[i][e]; hi4[e] = r.v[i][4 + e]; }
      *(f16x4*)(base + VOFF + row * PV + part * 16) = lo;
      *(f16x4*)(base + VOFF + row * PV + part * 16 + 8) = hi4;
      if (PREC) *(f16x8*)(base + KLO + row * PK + part * 16) = r.kl[PREC ? i : 0];
      if (PVS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] = r.vl[PVS ? i : 0][e]; hi4[e] = r.vl[PVS ? i : 0][4 + e]; }
        *(f16x4*)(base + VOFF + VLO + row * PV + part * 16) = lo;
        *(f16x4*)(base + VOFF + VLO + row * PV + part * 16 + 8) = hi4;
      }
    }
    if (tid < 64) ((float*)(base + BOFF))[tid] = r.in ? (bbase ? r.b : 0.0f) : SDM_NEG_BIG;
  };
  // tile walk: all ntiles tiles, or the active-tile list of this image (wave-uniform scalar loads)
  const int* tl = p.tiles ? p.tiles + (size_t)b * p.tiles_bs : nullptr;
  const int sp = p.nsplit > 1 ? (int)blockIdx.y : 0;
  int i0, nwalk;
  attn_split_range(p, tl, ntiles, tl ? tl[0] : ntiles, sp, i0, nwalk);
  if (nwalk <= 0) { attn_write_empty_part(p, sp, b, head, q0, lane); return; }      // (key split only; block-uniform, before the first barrier)
  auto tile_at = [&](int i) { return tl ? tl[1 + i0 + i] : i0 + i; };
  prefetch(tile_at(0), ra);
  stage(0, ra);
  __syncthreads();
  if (AHEAD2 && 1 < nwalk && !(p.ablate & 8)) prefetch(tile_at(1), ra);

  // one key tile.  cur: the raw tile t+1 (AHEAD2: requested one iteration ago; otherwise requested here), written to the other LDS buffer
  // below; nxt (AHEAD2): receives tile t+2
  auto iter = [&](const int t, Raw& cur, Raw& nxt) {
    const unsigned char* Ks = smem + (t & 1) * BUF;
    const unsigned char* Vs = Ks + VOFF;
    const float* Bs = (const float*)(Ks + BOFF);
    if (AHEAD2) { if (t + 2 < nwalk && !(p.ablate & 8)) prefetch(tile_at(t + 2), nxt); }
    else if (t + 1 < nwalk && !(p.ablate & 8)) prefetch(tile_at(t + 1), cur);

    // S^T[key][q] for 2 key tiles of 32 (x QT query tiles).  The accumulators START from the per-key additive bias (read
    // straight from LDS in accumulator layout: rows 8g+4hi..+3 = registers 4g..4g+3), so the MFMA chain delivers the final
    // logit x = q'.k + bias and no VALU pass is needed for scale/bias (this kernel is bound by the VALU pipe: exp, max, sub).
    f32x16 s[QT][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *(const f32x4*)(Bs + kt * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int e = 0; e < 4; ++e) s[qt][kt][4 * g + e] = b4[e];
      }
    // F8Q: every K fragment of the tile is read before the first MFMA, and the V^T fragments are read between the Q.K^T MFMAs and the
    // softmax (they land underneath it) instead of pairwise next to their MFMAs, where each pair waited for its own LDS round trip.
    // 96 registers more, at an unchanged occupancy (one 8-wave / two 4-wave blocks per CU).
    constexpr bool EARLY = F8Q && QT == 1;
    f16x8 ka[EARLY ? 2 : 1][EARLY ? 4 : 1];
    i32x8 ka8[EARLY ? 2 : 1][EARLY ? 2 : 1];
    if (EARLY) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ka[EARLY ? kt : 0][EARLY ? ks : 0] = *(const f16x8*)(Ks + (kt * 32 + l31) * PK + ks * 32 + hi * 16);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const unsigned char* kp = Ks + KLO + (kt * 32 + l31) * PK + m * 64 + hi * 32;
          const i32x4 a0 = *(const i32x4*)kp, a1 = *(const i32x4*)(kp + 16);
          ka8[EARLY ? kt : 0][EARLY ? m : 0] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        }
      }
      SDM_SCHED_FENCE();
    }
    if (EARLY && !(p.ablate & 4)) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) s[0][kt] = SDM_MFMA_32x32x16_F16(ka[EARLY ? kt : 0][EARLY ? ks : 0], qf[0][ks], s[0][kt]);
#pragma unroll
        for (int m = 0; m < 2; ++m) s[0][kt] = SDM_MFMA_32x32x64_BF8_BF8(ka8[EARLY ? kt : 0][EARLY ? m : 0], q8p[m], s[0][kt], 127 - 11, 127);
      }
      SDM_SCHED_FENCE();
    }
    f16x8 vfe[EARLY ? 2 : 1][EARLY ? 2 : 1][EARLY ? 2 : 1];
    if (EARLY) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int dt = 0; dt < 2; ++dt) {
            const unsigned char* vp = Vs + (dt * 32 + l31) * PV + (kt * 32 + 16 * u + 4 * hi) * 2;
            const f16x4 v0 = *(const f16x4*)vp, v1 = *(const f16x4*)(vp + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { vfe[EARLY ? kt : 0][EARLY ? u : 0][EARLY ? dt : 0][e] = v0[e]; vfe[EARLY ? kt : 0][EARLY ? u : 0][EARLY ? dt : 0][4 + e] = v1[e]; }
          }
      SDM_SCHED_FENCE();
    }
    if (!EARLY && !(p.ablate & 4)) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f16x8 a = *(const f16x8*)(Ks + (kt * 32 + l31) * PK + ks * 32 + hi * 16);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) s[qt][kt] = SDM_MFMA_32x32x16_F16(a, qf[qt][ks], s[qt][kt]);
        if (PREC && !F8Q) {
          const f16x8 al = *(const f16x8*)(Ks + KLO + (kt * 32 + l31) * PK + ks * 32 + hi * 16);
          s[0][kt] = SDM_MFMA_32x32x16_F16(al, qf[0][ks], s[0][kt]);
          s[0][kt] = SDM_MFMA_32x32x16_F16(a, qfl[0][ks], s[0][kt]);
        }
      }
    if (F8Q) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const unsigned char* kp = Ks + KLO + (kt * 32 + l31) * PK + m * 64 + hi * 32;
          const i32x4 a0 = *(const i32x4*)kp, a1 = *(const i32x4*)(kp + 16);
          const i32x8 a8 = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
          s[0][kt] = SDM_MFMA_32x32x64_BF8_BF8(a8, q8p[m], s[0][kt], 127 - 11, 127);
        }
    }
    }
    if (!(p.ablate & 1)) {
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      float mx = SDM_NEG_BIG;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[qt][kt][r]), s[qt][kt][r + 1]);      // v_max3_f32
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mnew = fmaxf(m_i[qt], mx);
      const float alpha = sdm_exp2(m_i[qt] - mnew);
      m_i[qt] = mnew;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[qt][kt][r] = sdm_exp2(s[qt][kt][r] - mnew);
      if (PVS) {
        float ts0 = 0.0f, ts1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { ts0 += s[qt][0][r]; ts1 += s[qt][1][r]; }
        lsum = lsum * alpha + (ts0 + ts1);
      }
      // the running max stops moving after the first tiles: skip the O / denominator rescale when no lane of the wave needs it
      if (__any(alpha != 1.0f)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ls[qt][r] *= alpha;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[qt][dt][r] *= alpha;
      }
    }
    }

    // next tile into the OTHER LDS buffer (nobody reads it during this iteration) - before the P.V MFMAs when early_stage is set, so
    // that the LDS writes complete under them instead of in front of the barrier
    const bool early_stage = (p.ablate & 32) == 0;
    if (early_stage && t + 1 < nwalk) stage((t + 1) & 1, cur);
    // O^T[d][q] += V^T[d][key] . P^T[key][q]
    if (!(p.ablate & 2)) {
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f16x8 pf[QT], pfl;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
          for (int j = 0; j < 8; ++j) pf[qt][j] = (half_t)s[qt][kt][8 * u + j];
        if (PVS) {
#pragma unroll
          for (int j = 0; j < 8; ++j) pfl[j] = (half_t)(s[0][kt][8 * u + j] - (float)pf[0][j]);
        }
        if (!PVS) {
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) ls[qt] = SDM_MFMA_32x32x16_F16(ones, pf[qt], ls[qt]);      // every row = sum_k P[k][q]
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned char* vp = Vs + (dt * 32 + l31) * PV + (kt * 32 + 16 * u + 4 * hi) * 2;
          f16x8 vf;
          if (EARLY) {
            vf = vfe[EARLY ? kt : 0][EARLY ? u : 0][EARLY ? dt : 0];
          } else {
            const f16x4 v0 = *(const f16x4*)vp;
            const f16x4 v1 = *(const f16x4*)(vp + 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
          }
#pragma unroll
          for (int qt = 0; qt < QT; ++qt) o[qt][dt] = SDM_MFMA_32x32x16_F16(vf, pf[qt], o[qt][dt]);
          if (PVS) {
            const f16x4 w0 = *(const f16x4*)(vp + VLO);
            const f16x4 w1 = *(const f16x4*)(vp + VLO + 16);
            f16x8 vfl;
#pragma unroll
            for (int e = 0; e < 4; ++e) { vfl[e] = w0[e]; vfl[4 + e] = w1[e]; }
            o[0][dt] = SDM_MFMA_32x32x16_F16(vfl, pf[0], o[0][dt]);
            o[0][dt] = SDM_MFMA_32x32x16_F16(vf, pfl, o[0][dt]);
          }
        }
      }
    }
    if (!early_stage && t + 1 < nwalk) stage((t + 1) & 1, cur);
    __syncthreads();
  };
  if (AHEAD2) {
    for (int t = 0; t < nwalk; t += 2) {
      iter(t, ra, rb);
      if (t + 1 < nwalk) iter(t + 1, rb, ra);
    }
  } else {
    for (int t = 0; t < nwalk; ++t) iter(t, ra, ra);
  }

  // epilogue: per-wave staging [32 q][64 d] fp16 -> coalesced 16-B row stores (all waves passed the last barrier)
  unsigned char* stg = smem + wave * (32 * PK);
  if (p.o_f32) {                    // fp32 output (precise-mode graphs): [32 q][64 d] fp32 staging at pitch 272 B
    constexpr int PS = 272;
    unsigned char* stf = smem + wave * (32 * PS);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
      const float lsum_q = PVS ? (lsum + __shfl_xor(lsum, 32)) : ls[qt][0];      // PVS: lanes l and l^32 hold the two key halves of query l&31
      const float inv = p.nsplit > 1 ? 1.0f : 1.0f / lsum_q;                      // key split: unnormalised partial sums (attn_combine_kernel divides)
      float* obase = (float*)p.o + (p.nsplit > 1 ? (size_t)sp * p.part_stride : (size_t)0) + (size_t)b * p.o_bs;
      if (p.nsplit > 1 && hi == 0 && q0 + qt * 32 + l31 < p.Lq) {
        float* ml = p.part_ml + ((((size_t)sp * p.batch + b) * p.heads + head) * p.Lq + q0 + qt * 32 + l31) * 2;
        ml[0] = m_i[qt]; ml[1] = lsum_q;
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = o[qt][dt][4 * g + e] * inv;
          *(f32x4*)(stf + l31 * PS + (dt * 32 + 8 * g + 4 * hi) * 4) = h;
        }
      SDM_WAVE_SYNC();
      if (p.o_p3 && p.nsplit <= 1) {
        attn_store_p3(p, stf, PS, b, q0 + qt * 32, head, lane);
      } else {
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 4 + (lane >> 4), part = lane & 15;
        const int qg = q0 + qt * 32 + row;
        if (qg < p.Lq)
          *(f32x4*)(obase + (size_t)qg * p.ldo + head * 64 + part * 4) = *(const f32x4*)(stf + row * PS + part * 16);
      }
      }
      SDM_WAVE_SYNC();
    }
    return;
  }
#pragma unroll
  for (int qt = 0; qt < QT; ++qt) {
    const float l = PVS ? (lsum + __shfl_xor(lsum, 32)) : ls[qt][0];      // all 32 rows of the ones.P^T tile hold the same sum over every key (both lane halves included)
    const float inv = 1.0f / l;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[qt][dt][4 * g + e] * inv);
        *(f16x4*)(stg + l31 * PK + (dt * 32 + 8 * g + 4 * hi) * 2) = h;
      }
    SDM_WAVE_SYNC();
#pragma unroll
    for (int pass = 0; pass < 4; ++pass) {
      const int row = pass * 8 + (lane >> 3), part = lane & 7;
      const int qg = q0 + qt * 32 + row;
      if (qg < p.Lq)
        *(f16x8*)(p.o + (size_t)b * p.o_bs + (size_t)qg * p.ldo + head * 64 + part * 8) = *(const f16x8*)(stg + row * PK + part * 16);
    }
    SDM_WAVE_SYNC();
  }
}

// ------------------------------------------------------------------------------------------------
// The 8-wave fp8-residual kernel above as a TWO-TILE software pipeline (default for the 8-wave launches with fp32 output; engine option attn_pipe = 0
// selects attn_d64_kernel<1,3,8>; measured -9 % on those launches, bit-identical results on hardware: profiles/r03_attn_pipe_ab.txt): the Q.K^T MFMAs
// of key tile t+1 and the softmax of key tile t are independent and sit in ONE basic block, so the scheduler can issue the VALU stream
// (max / sub / exp / pack: ~1000 cycles per tile) underneath the matrix stream instead of after it; both waves of a SIMD are in the
// same phase of attn_d64_kernel (one barrier per tile), and SQ counters put 42 % of a wave's cycles into issue stalls there.
// Three LDS buffers (108 KB, one 8-wave block per CU): iteration t reads V^T of tile t and K / bias of tile t+1 and writes tile t+2;
// the barrier is the raw s_barrier behind an LDS-only wait, so the global loads of tile t+3 stay in flight across it.
// Same arithmetic, same order per query row as attn_d64_kernel<1,3,8>: results are bit-identical (tests/test_emu_ops.py).
// ------------------------------------------------------------------------------------------------
// NW = 8 (measured, default): three whole tile buffers.  NW = 4 (engine option attn_pipe4, on since round 4; bit-identical to attn_d64_kernel<1,3,4> on the
// emulator and on hardware): K | pair plane | bias of a tile are read one iteration before its V^T, so they rotate through TWO
// buffers and only V^T through three - 63.5 KB per block, two blocks per CU as for attn_d64_kernel<1,3,4>.
#define ATTN64PIPE_SMEM (3 * ATTN64P_BUF)
#define ATTN64PIPE4_KB (2 * 64 * ATTN64_PK + 256)
#define ATTN64PIPE4_VB (64 * ATTN64_PV)
#define ATTN64PIPE4_SMEM (2 * ATTN64PIPE4_KB + 3 * ATTN64PIPE4_VB)
template <int NW>
__global__ void __launch_bounds__(64 * NW, 2) attn_d64_pipe_kernel(AttnParams p) {
  constexpr int NTH = 64 * NW, VPT = 512 / NTH;
  constexpr bool SPLITBUF = NW == 4;
  SDM_DYN_SMEM(smem);
  constexpr int PK = ATTN64_PK, PV = ATTN64_PV, BUF = ATTN64P_BUF;
  constexpr int KLO = 64 * PK, VOFF = 2 * 64 * PK, BOFF = VOFF + 64 * PV;
  // LDS addresses of tile k's parts; b3 = k % 3, b2 = k & 1
  auto k_of = [&](int b3, int b2) { return SPLITBUF ? smem + b2 * ATTN64PIPE4_KB : smem + b3 * BUF; };
  auto v_of = [&](int b3) { return SPLITBUF ? smem + 2 * ATTN64PIPE4_KB + b3 * ATTN64PIPE4_VB : smem + b3 * BUF + VOFF; };
  auto bias_of = [&](int b3, int b2) { return (float*)(SPLITBUF ? smem + b2 * ATTN64PIPE4_KB + 2 * 64 * PK : smem + b3 * BUF + BOFF); };
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  int b, head, qblk;
  if (!attn_block_coords(p, blockIdx.x, b, head, qblk)) return;
  const int q0 = qblk * (32 * NW) + wave * 32;

  f16x8 qf[4];
  i32x8 q8p[2];
  {
    int qrow = q0 + l31;
    if (qrow > p.Lq - 1) qrow = p.Lq - 1;
    const half_t* qp = p.q + (size_t)b * p.q_bs + (size_t)qrow * p.ldq + head * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qp + ks * 16);
    const half_t* qb = p.q + (size_t)b * p.q_bs + p.q_lo + (size_t)qrow * p.ldq + head * 64 + hi * 16;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const i32x4 r0 = *(const i32x4*)(qb + m * 32), r1 = *(const i32x4*)(qb + m * 32 + 8);
      q8p[m] = i32x8{r0[1], r0[0], r0[3], r0[2], r1[1], r1[0], r1[3], r1[2]};
    }
  }
  f32x16 o[2], ls;
  float m_i = SDM_NEG_BIG;
#pragma unroll
  for (int r = 0; r < 16; ++r) { ls[r] = 0.0f; o[0][r] = 0.0f; o[1][r] = 0.0f; }
  f16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.0f;

  const half_t* kbase = p.k + (size_t)b * p.k_bs + head * 64;
  const half_t* vbase = p.vt + (size_t)b * p.vt_bs + (size_t)head * p.vt_hs;
  const float* bbase = p.bias ? p.bias + (size_t)b * p.bias_bs : nullptr;
  const float* bsrc = bbase ? bbase : (const float*)(p.k + (size_t)b * p.k_bs);
  const int ntiles = (p.Lk + 63) / 64;
  const int* tl = p.tiles ? p.tiles + (size_t)b * p.tiles_bs : nullptr;
  const int sp = p.nsplit > 1 ? (int)blockIdx.y : 0;
  int i0, nwalk;
  attn_split_range(p, tl, ntiles, tl ? tl[0] : ntiles, sp, i0, nwalk);
  if (nwalk <= 0) { attn_write_empty_part(p, sp, b, head, q0, lane); return; }      // (key split only; block-uniform, before the first barrier)
  auto tile_at = [&](int i) { return tl ? tl[1 + i0 + i] : i0 + i; };

  // one raw tile between global memory and LDS (VPT 16-byte vectors of K_hi, K pair plane and V^T per thread, one bias value per lane)
  f16x8 rk[VPT], rkl[VPT], rv[VPT];
  float rb = 0.0f;
  bool rin = true;
  auto prefetch = [&](int t) {
    const int k0 = t * 64;
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NTH, srow = v >> 3, spart = v & 7;
      int kr = k0 + srow;
      if (kr > p.Lk - 1) kr = p.Lk - 1;
      rk[i] = *(const f16x8*)(kbase + (size_t)kr * p.ldk + spart * 8);
      rv[i] = *(const f16x8*)(vbase + (size_t)srow * p.ldvt + k0 + spart * 8);
      rkl[i] = *(const f16x8*)(kbase + p.k_lo + (size_t)kr * p.ldk + spart * 8);
    }
    int kb = k0 + (tid & 63);
    rin = kb < p.Lk;
    if (!rin) kb = p.Lk - 1;
    rb = bsrc[kb];
  };
  auto stage = [&](int b3, int b2) {
    unsigned char* kb_ = k_of(b3, b2);
    unsigned char* vb_ = v_of(b3);
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
      const int v = tid + i * NTH, srow = v >> 3, spart = v & 7;
      *(f16x8*)(kb_ + srow * PK + spart * 16) = rk[i];
      f16x4 lo, hi4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { lo[e] = rv[i][e]; hi4[e] = rv[i][4 + e]; }
      *(f16x4*)(vb_ + srow * PV + spart * 16) = lo;
      *(f16x4*)(vb_ + srow * PV + spart * 16 + 8) = hi4;
      *(f16x8*)(kb_ + KLO + srow * PK + spart * 16) = rkl[i];
    }
    if (tid < 64) bias_of(b3, b2)[tid] = rin ? (bbase ? rb : 0.0f) : SDM_NEG_BIG;
  };
  // logits of one key tile: S^T[key][q], accumulators started from the per-key bias
  auto qk = [&](int b3, int b2, f32x16 (&s)[2]) {
    const unsigned char* Ks = k_of(b3, b2);
    const float* Bs = bias_of(b3, b2);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *(const f32x4*)(Bs + kt * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[kt][4 * g + e] = b4[e];
      }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const f16x8 a = *(const f16x8*)(Ks + (kt * 32 + l31) * PK + ks * 32 + hi * 16);
        s[kt] = SDM_MFMA_32x32x16_F16(a, qf[ks], s[kt]);
      }
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const unsigned char* kp = Ks + KLO + (kt * 32 + l31) * PK + m * 64 + hi * 32;
        const i32x4 a0 = *(const i32x4*)kp, a1 = *(const i32x4*)(kp + 16);
        const i32x8 a8 = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        s[kt] = SDM_MFMA_32x32x64_BF8_BF8(a8, q8p[m], s[kt], 127 - 11, 127);
      }
    }
  };
  auto lds_barrier = [&]() { SDM_WAIT_LGKMCNT0(); SDM_RAW_BARRIER(); };

  // ---- prologue: tiles 0 and 1 in LDS, tile 2 on its way, logits of tile 0 in registers ----
  prefetch(tile_at(0));
  stage(0, 0);
  if (1 < nwalk) prefetch(tile_at(1));
  lds_barrier();
  if (1 < nwalk) stage(1, 1);
  if (2 < nwalk) prefetch(tile_at(2));
  f32x16 sa[2], sb[2];
  qk(0, 0, sa);
  lds_barrier();

  int bt = 0;                         // buffer of tile t (t % 3 without a division)
  // sc: logits of tile t (consumed here), sn: receives the logits of tile t+1; the two register sets swap roles every tile
  auto iter = [&](const int t, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
    const int b1 = bt == 2 ? 0 : bt + 1, b2 = b1 == 2 ? 0 : b1 + 1;      // three-buffer slots of tiles t+1, t+2 (two-buffer slots: parities)
    // (1) logits of tile t+1 (when there is none: the same instructions on a buffer nobody consumes - no branch inside this block)
    //     and the softmax of tile t: independent streams in one basic block
    qk(b1, (t + 1) & 1, sn);
    float mx = SDM_NEG_BIG;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, sc[kt][r]), sc[kt][r + 1]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(m_i, mx);
    const float alpha = sdm_exp2(m_i - mnew);
    m_i = mnew;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[kt][r] = sdm_exp2(sc[kt][r] - mnew);
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ls[r] *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    // (2) tile t+2 into the buffer tile t-1 has left (its last readers passed the previous barrier), tile t+3 requested
    if (t + 2 < nwalk) stage(b2, t & 1);
    if (t + 3 < nwalk) prefetch(tile_at(t + 3));
    // (3) O^T[d][q] += V^T[d][key] . P^T[key][q], denominator on the same probabilities
    const unsigned char* Vs = v_of(bt);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        f16x8 pf;
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[j] = (half_t)sc[kt][8 * u + j];
        ls = SDM_MFMA_32x32x16_F16(ones, pf, ls);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          const unsigned char* vp = Vs + (dt * 32 + l31) * PV + (kt * 32 + 16 * u + 4 * hi) * 2;
          const f16x4 v0 = *(const f16x4*)vp, v1 = *(const f16x4*)(vp + 16);
          f16x8 vf;
#pragma unroll
          for (int e = 0; e < 4; ++e) { vf[e] = v0[e]; vf[4 + e] = v1[e]; }
          o[dt] = SDM_MFMA_32x32x16_F16(vf, pf, o[dt]);
        }
      }
    lds_barrier();
    bt = b1;
  };
  for (int t = 0; t < nwalk; t += 2) {
    iter(t, sa, sb);
    if (t + 1 < nwalk) iter(t + 1, sb, sa);
  }

  // epilogue (fp32 output): per-wave staging [32 q][64 d] at pitch 272 B -> coalesced 16-byte row stores
  constexpr int PS = 272;
  unsigned char* stf = smem + wave * (32 * PS);
  const float inv = p.nsplit > 1 ? 1.0f : 1.0f / ls[0];                 // key split: unnormalised partial sums (attn_combine_kernel divides)
  float* obase = (float*)p.o + (p.nsplit > 1 ? (size_t)sp * p.part_stride : (size_t)0) + (size_t)b * p.o_bs;
  if (p.nsplit > 1 && hi == 0 && q0 + l31 < p.Lq) {
    float* ml = p.part_ml + ((((size_t)sp * p.batch + b) * p.heads + head) * p.Lq + q0 + l31) * 2;
    ml[0] = m_i; ml[1] = ls[0];
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = o[dt][4 * g + e] * inv;
      *(f32x4*)(stf + l31 * PS + (dt * 32 + 8 * g + 4 * hi) * 4) = h;
    }
  SDM_WAVE_SYNC();
  if (p.o_p3 && p.nsplit <= 1) { attn_store_p3(p, stf, PS, b, q0, head, lane); return; }
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 4 + (lane >> 4), part = lane & 15;
    const int qg = q0 + row;
    if (qg < p.Lq)
      *(f32x4*)(obase + (size_t)qg * p.ldo + head * 64 + part * 4) = *(const f32x4*)(stf + row * PS + part * 16);
  }
}

// ------------------------------------------------------------------------------------------------
// PING-PONG of the block's two wave halves (round 6; engine option attn_pp, default for key counts that are a multiple of 64): waves 0-3 ("A", one per
// SIMD) and waves 4-7 ("B", their SIMD partners) run the SAME instruction sequence - [softmax of tile t] [P.V of tile t, Q.K^T of tile t+1] - half a tile
// period apart, so that every SIMD holds one wave in a matrix segment (24 back-to-back MFMAs = 896 pipe cycles) beside one in a softmax segment (max / sub /
// 33 v_exp_f32 / pack).  In attn_d64_pipe_kernel both waves of a SIMD are in the same phase (one barrier per tile): the SQ counters put the matrix pipe at 0.53
// busy with VALU issue active 0.48 of the time - the pipes ran in SUM, not in MAX (profiles/r06_attn_sq_counters.txt).  MI355X_MICROARCH.md "Two waves per
// SIMD": complementary segments (matrix beside VALU / memory) are what nets.
//     A (0-3):  QK(0) softmax(0) | PV(0) QK(1)   softmax(1) | PV(1) QK(2)   softmax(2) | ...          "|" = the ONE barrier a wave passes per tile (OB = 1):
//     B (4-7):        QK(0)      | softmax(0)    PV(0) QK(1) | softmax(1)   PV(1) QK(2) | ...          A behind its softmax segment, B behind its matrix segment
// How it got here (every step measured; profiles/r06_attn_pp_*.txt, NOTES.md "Round 6"):
// * compile-time ablations of the first build (register-staged tiles, two barriers per tile): MFMAs + softmax + barriers alone took 1 837 cycles per tile pair -
//   the pipe floor is 1 792 - but the whole kernel 3 550: global loads into registers 750, LDS staging writes 370, fragment reads 520.  Hence tiles travel by
//   LDS-DMA - no staging registers, no ds_write, no compiler-placed vmcnt in the loop (hipcc does not count LDS-DMAs: tile-list entries come through scalar loads)
//   - into a ring of SIX unpadded slots: the DMAs of tile t+4 go out at the head of the softmax segment of tile t (slot of tile t-2: its last reader, B's matrix
//   segment, lies two barriers back for either half); before its barrier every wave leaves only its two youngest DMA batches in flight, so whatever a half reads
//   behind a barrier (K of tile t+1 in its matrix segment, V^T of tile t+1 in its next softmax segment) has landed on both halves' side;
// * images are [64 rows][8 chunks of 16 B], chunk c of row r at position c ^ ((r >> 1) & 7): the 16 lanes of a ds_read_b128 group read 16 different
//   (row parity, position) pairs - conflict-free without padding (a DMA writes 1 KB runs: rows cannot be padded); measured 0.7 % conflict cycles;
// * K row rho of a 32-key half holds key pi(rho) (bits 2 and 3 swapped, as in attn_d512_kernel): the 8 probabilities a lane owns per 16-key step are then 8
//   CONSECUTIVE keys and a V^T fragment is ONE ds_read_b128 (was two ds_read_b64 at a 136-byte pitch).  This permutes the k-slots of the P.V MFMAs: results
//   equal the pipelines' up to the order of the fp32 sums inside an MFMA (not bit-identical; dense and tile-list walks of THIS kernel stay bit-identical);
// * BIAS = 0 (no key bias: every cross-attention): accumulators start from 0, no bias row at all (8 of 40 fragment reads).  LIST = 0: dense walk;
// * a segment trace (s_memtime stamps, ABL = 64) showed the SOFTMAX segment to be the long pole (~1 000-1 100 cycles of VALU issue + ~120 cycles per DMA
//   instruction + an exposed LDS round trip for the V^T fragments, against ~1 100 for the matrix segment) with the matrix wave waiting 500-700 cycles at each of
//   the two barriers per tile: the V^T reads now go out behind the max (they land under the exponentials) and each wave passes ONE barrier per tile (OB = 1) - the
//   halves meet once per tile period instead of every phase being as long as its longer segment.  This form NEEDS the static s_setprio 1 of waves 4-7
//   (attention d=64 per step: 24.2 ms with it, 29.0 ms without or with per-segment flips; the round-5 pipelines: 28.8).
// Template parameters besides BIAS / LIST: KE / DS / OB = -1..2 / 0..3 / 0 keep the measured alternatives compilable (K fragment reads in the softmax segment;
// DMAs split between the head of the softmax and the end of the matrix segment; two barriers per tile): all within noise of each other or slower
// (profiles/r06_attn_pp_placements_ab.txt); KE = -1 is the first DMA build (four slots, everything inside the matrix segment).
// Needs Lk % 64 == 0 (no masked tail rows); other launches keep the pipelines.  DMAs and tile indices are unconditional with clamped indices (a rewritten slot
// nobody reads any more); Q.K^T of the tile behind the last one runs on stale bytes whose logits nobody consumes - a matrix segment is ONE basic block.
// pp_flags bit 0: s_setprio 1 for waves 4-7; bit 1: per-segment priority flips (A/B).
// ABL (bench only, sdm_bench_attn; 0 in the engine): 1 no softmax VALU, 2 no P.V MFMAs, 4 no Q.K^T MFMAs, 8 no DMAs, 32 no fragment reads, 64 segment stamps
// ------------------------------------------------------------------------------------------------
#define ATTN64PP_SLOT (3 * 8192 + 256)
#define ATTN64PP_SMEM (6 * ATTN64PP_SLOT)
template <int ABL = 0, int BIAS = 1, int LIST = 1, int KE = 0, int DS = 1, int OB = 1>
__global__ void __launch_bounds__(512, 2) attn_d64_pp_kernel(AttnParams p) {
  SDM_DYN_SMEM(smem);
  constexpr int SLOT = ATTN64PP_SLOT, KLO = 8192, VOFF = 16384, BOFF = 24576;
  constexpr int R = KE < 0 ? 4 : 6, D = KE < 0 ? 3 : 4;             // ring slots; a segment of tile t requests tile t + D
  constexpr int NDMA = BIAS ? 4 : 3;                                  // DMAs a wave issues per tile
  const int tid = threadIdx.x, lane = tid & 63, wave = SDM_UNIFORM_I(tid >> 6);
  const int grp = wave >> 2;
  const int hi = lane >> 5, l31 = lane & 31;
  int b, head, qblk;
  if (!attn_block_coords(p, blockIdx.x, b, head, qblk)) return;
  const int q0 = qblk * 256 + wave * 32;

  f16x8 qf[4];
  i32x8 q8p[2];
  {
    int qrow = q0 + l31;
    if (qrow > p.Lq - 1) qrow = p.Lq - 1;
    const half_t* qp = p.q + (size_t)b * p.q_bs + (size_t)qrow * p.ldq + head * 64 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const f16x8*)(qp + ks * 16);
    const half_t* qb = p.q + (size_t)b * p.q_bs + p.q_lo + (size_t)qrow * p.ldq + head * 64 + hi * 16;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const i32x4 r0 = *(const i32x4*)(qb + m * 32), r1 = *(const i32x4*)(qb + m * 32 + 8);
      q8p[m] = i32x8{r0[1], r0[0], r0[3], r0[2], r1[1], r1[0], r1[3], r1[2]};
    }
  }
  f32x16 o[2], ls;
  float m_i = SDM_NEG_BIG;
#pragma unroll
  for (int r = 0; r < 16; ++r) { ls[r] = 0.0f; o[0][r] = 0.0f; o[1][r] = 0.0f; }
  f16x8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.0f;

  const half_t* kbase = p.k + (size_t)b * p.k_bs + head * 64;
  const half_t* vbase = p.vt + (size_t)b * p.vt_bs + (size_t)head * p.vt_hs;
  const int ntiles = p.Lk / 64;                                        // (Lk % 64 == 0: launcher's condition)
  const int* tl = (LIST && p.tiles) ? p.tiles + (size_t)b * p.tiles_bs : nullptr;
  const int sp = p.nsplit > 1 ? (int)blockIdx.y : 0;
  int i0, nwalk;
  attn_split_range(p, tl, ntiles, tl ? tl[0] : ntiles, sp, i0, nwalk);
  if (nwalk <= 0) { attn_write_empty_part(p, sp, b, head, q0, lane); return; }      // (key split only; block-uniform, before the first barrier)
  const int* tlp = tl ? tl + 1 + i0 : nullptr;
  auto tile_at = [&](int i) { if (i > nwalk - 1) i = nwalk - 1; return LIST ? tlp[i] : i0 + i; };      // clamped: see the header

  // ---- DMA side.  Wave w fills rows 8w .. 8w+7 of each image: lane L -> row rho = 8w + (L >> 3), position L & 7 = source chunk (L & 7) ^ ((rho >> 1) & 7);
  //      K rows: key = 32 * (rho >> 5) + pi(rho & 31).  Per-lane byte offsets are loop invariants; the tile advances through the uniform offset.
  const sdm_rsrc rsK = sdm_make_rsrc(kbase, (unsigned int)(((size_t)(p.Lk - 1) * p.ldk + 64) * 2));
  const sdm_rsrc rsK8 = sdm_make_rsrc(kbase + p.k_lo, (unsigned int)(((size_t)(p.Lk - 1) * p.ldk + 64) * 2));
  const sdm_rsrc rsV = sdm_make_rsrc(vbase, (unsigned int)(((size_t)63 * p.ldvt + p.Lk) * 2));
  const sdm_rsrc rsB = sdm_make_rsrc(BIAS ? p.bias + (size_t)b * p.bias_bs : (const float*)kbase, (unsigned int)p.Lk * 4u);
  const int rho = 8 * wave + (lane >> 3), cpos = lane & 7;
  const int csrc = cpos ^ ((rho >> 1) & 7);
  const int rr = rho & 31;
  const int pir = (rr & 0x13) | (((rr >> 2) & 1) << 3) | (((rr >> 3) & 1) << 2);
  const unsigned int voffK = (unsigned int)(((rho & 32) + pir) * p.ldk * 2 + csrc * 16);
  const unsigned int voffV = (unsigned int)(rho * p.ldvt * 2 + csrc * 16);
  const int lr = lane & 31;
  const unsigned int voffB = (unsigned int)(((lane & 32) + ((lr & 0x13) | (((lr >> 2) & 1) << 3) | (((lr >> 3) & 1) << 2))) * 4);
  // the DMAs of one tile in issue order: V^T, [bias,] K pair plane, K_hi; `from` .. `to` selects a part of them (DS of them go out at the head of the softmax
  // segment, the rest at the END of the matrix segment - where the wave would otherwise sit at the barrier: a DMA instruction costs the issuing wave ~100-130 cycles)
  auto dma = [&](int tile, int slot, int from = 0, int to = 4) {
    if (ABL & 8) return;
    unsigned char* sb = smem + slot * SLOT + wave * 1024;
    const unsigned int k0 = (unsigned int)tile * 64u;
    int i = 0;
    if (i >= from && i < to) sdm_glds16_buf(rsV, voffV, k0 * 2u, sb + VOFF);
    ++i;
    if (BIAS) { if (i >= from && i < to) sdm_glds4_buf(rsB, voffB, k0 * 4u, smem + slot * SLOT + BOFF); ++i; }      // (every wave writes the same 256 bytes: one DMA count for all waves)
    if (i >= from && i < to) sdm_glds16_buf(rsK8, voffK, k0 * (unsigned int)(p.ldk * 2), sb + KLO);
    ++i;
    if (i >= from && i < to) sdm_glds16_buf(rsK, voffK, k0 * (unsigned int)(p.ldk * 2), sb);
  };
  // ---- fragment side: row l31 of a 32-row half, chunk (2 ks + hi) etc. at position chunk ^ ((l31 >> 1) & 7) - the XOR folds into the lane's base offset
  const int kxor = (l31 >> 1) & 7;
  const int kx = l31 * 128 + ((hi ^ kxor) << 4);                       // K_hi: ^ (ks * 32), + kt * 4096
  const int k8x = l31 * 128 + (((2 * hi) ^ kxor) << 4);                // pair plane: ^ (m * 64), second half ^ 16
  const int vx = l31 * 128 + ((hi ^ kxor) << 4);                       // V^T: ^ ((4 kt + 2 u) * 16), + dt * 4096
  struct KFrag { f16x8 h[4]; i32x8 f8[2]; };
  auto load_bias = [&](int slot, f32x16 (&s)[2]) {
    if (!BIAS) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.0f;
      return;
    }
    if (ABL & 32) return;
    const float* Bs = (const float*)(smem + slot * SLOT + BOFF);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 b4 = *(const f32x4*)(Bs + kt * 32 + 8 * g + 4 * hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[kt][4 * g + e] = b4[e];
      }
  };
  auto load_k = [&](int slot, int kt, KFrag& f) {
    if (ABL & 32) { SDM_PIN_HERE_V4(f.h[0], f.h[1], f.h[2], f.h[3]); SDM_PIN_HERE_V4(f.f8[0], f.f8[1], f.h[0], f.h[1]); return; }
    const unsigned char* Ks = smem + slot * SLOT + kt * 4096;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.h[ks] = *(const f16x8*)(Ks + (kx ^ (ks * 32)));
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      const i32x4 a0 = *(const i32x4*)(Ks + KLO + (k8x ^ (m * 64))), a1 = *(const i32x4*)(Ks + KLO + ((k8x ^ (m * 64)) ^ 16));
      f.f8[m] = i32x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    }
  };
  auto mma_k = [&](const KFrag& f, f32x16& sk) {
    if (ABL & 4) return;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) sk = SDM_MFMA_32x32x16_F16(f.h[ks], qf[ks], sk);
#pragma unroll
    for (int m = 0; m < 2; ++m) sk = SDM_MFMA_32x32x64_BF8_BF8(f.f8[m], q8p[m], sk, 127 - 11, 127);
  };
  struct VFrag { f16x8 v[2][2]; };      // [u][dt]
  auto load_v = [&](int slot, int kt, VFrag& f) {
    if (ABL & 32) { SDM_PIN_HERE_V4(f.v[0][0], f.v[0][1], f.v[1][0], f.v[1][1]); return; }
    const unsigned char* Vs = smem + slot * SLOT + VOFF;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) f.v[u][dt] = *(const f16x8*)(Vs + dt * 4096 + (vx ^ ((4 * kt + 2 * u) * 16)));
  };
  auto mma_v = [&](const VFrag& f, const f16x8 (&pk)[2]) {      // O^T[d][q] += V^T[d][key] . P^T[key][q], denominators on the same probabilities
    if (ABL & 2) return;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      ls = SDM_MFMA_32x32x16_F16(ones, pk[u], ls);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) o[dt] = SDM_MFMA_32x32x16_F16(f.v[u][dt], pk[u], o[dt]);
    }
  };
  // logits of one key tile alone (prologue)
  auto qk = [&](int slot, f32x16 (&s)[2]) {
    KFrag k0, k1;
    load_bias(slot, s);
    load_k(slot, 0, k0);
    load_k(slot, 1, k1);
    mma_k(k0, s[0]);
    mma_k(k1, s[1]);
  };
  // softmax of one tile: logits -> fp16 probabilities in B-operand layout; running maximum, rescale of O^T / the denominators when it moved.  Everything is
  // pinned inside the segment (the probabilities are only consumed by the NEXT segment's MFMAs: left alone, the sub / exp / pack stream sinks behind the barrier)
  auto softmax = [&](f32x16 (&s)[2], f16x8 (&pf)[2][2], auto&& mid0, auto&& mid1, auto&& after_max) {
    if (ABL & 1) {
      mid0(); mid1();
      after_max();
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int j = 0; j < 8; j += 2) { pf[kt][u][j] = (half_t)s[kt][8 * u + j]; pf[kt][u][j + 1] = pf[kt][u][j]; }
      SDM_PIN_HERE_V4(pf[0][0], pf[0][1], pf[1][0], pf[1][1]);
      return;
    }
    // the segment's remaining DMA instructions are spread over the VALU stream (mid0 behind the max, mid1 between the two halves of the exponentials): a DMA keeps the address path busy for ~120 cycles and the NEXT
    // vector-memory instruction of the wave waits for it - VALU instructions in between do not
    float mx = SDM_NEG_BIG;
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[0][r]), s[0][r + 1]);
#pragma unroll
    for (int r = 0; r < 16; r += 2) mx = fmaxf(fmaxf(mx, s[1][r]), s[1][r + 1]);
    {      // the other key half of this query sits in lane ^ 32: one v_permlane32_swap instead of a trip through the LDS crossbar
      unsigned int xa = __builtin_bit_cast(unsigned int, mx), xb = xa;
      sdm_permlane32_swap(xa, xb);
      mx = fmaxf(__builtin_bit_cast(float, xa), __builtin_bit_cast(float, xb));
    }
    const float mnew = fmaxf(m_i, mx);
    const float alpha = sdm_exp2(m_i - mnew);
    m_i = mnew;
    SDM_SCHED_FENCE();
    mid0();
    SDM_SCHED_FENCE();
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ls[r] *= alpha;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    after_max();      // the fragment reads of the coming matrix segment go out HERE: they land under the 33 exponentials instead of in front of the barrier (220 cycles)
    // (the 32 subtractions as 16 v_pk_add_f32 were measured: +7-9 % kernel time - packed fp32 shares the wide datapath with the partner wave's MFMAs,
    //  profiles/r06_attn_pp_packed_sub_ab.txt)
#pragma unroll
    for (int r = 0; r < 16; ++r) s[0][r] = sdm_exp2(s[0][r] - mnew);
    SDM_SCHED_FENCE();
    mid1();
    SDM_SCHED_FENCE();
#pragma unroll
    for (int r = 0; r < 16; ++r) s[1][r] = sdm_exp2(s[1][r] - mnew);
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) pf[kt][u][j] = (half_t)s[kt][8 * u + j];
    SDM_PIN_HERE_V4(pf[0][0], pf[0][1], pf[1][0], pf[1][1]);
  };
  auto lds_barrier = [&]() { SDM_SCHED_FENCE(); SDM_WAIT_LGKMCNT0(); SDM_RAW_BARRIER(); SDM_SCHED_FENCE(); };

  // ---- prologue: tiles 0 .. D-1 in LDS; A computes the logits of tile 0 while B waits one interval ----
#pragma unroll
  for (int i = 0; i < D; ++i) dma(tile_at(i), i);
  int tq = tile_at(D);                                                 // index of the tile the first loop segment requests
  SDM_WAIT_VMCNT0();
  lds_barrier();
  // OB = 1 (ONE barrier per tile and wave): half A passes its barrier behind the softmax segment, half B behind the matrix segment - the two meet once per tile
  // period, A then runs [matrix(t), softmax(t+1)] while B runs [softmax(t), matrix(t)].  The segment trace (profiles/r06_attn_pp_segment_trace.txt) showed what two
  // barriers per tile cost: ~150 cycles of barrier latency each, and every phase as long as its LONGER segment (softmax ~1 400 incl. DMA issue, matrix ~1 100).
  // DMAs (all at the head of the softmax segment) and waits: the DMAs of tile t+4 go out in front of softmax(t) (slot of tile t-2: its last reader, B's matrix
  // segment, lies two barriers back for either half); before its barrier every wave leaves only its two youngest batches in flight, so whatever a half reads behind
  // a barrier (K of tile t+1 in matrix(t), V^T of tile t+1 in softmax(t+1)) has landed on both halves' side.
  if (!OB) {
    if (grp) {
      if (p.pp_flags & 1) SDM_SETPRIO(1);
      lds_barrier();
    }
  } else if (grp && (p.pp_flags & 1)) SDM_SETPRIO(1);
  f32x16 s[2];
  f16x8 pf[2][2];
  qk(0, s);
  if (!OB || grp) lds_barrier();                                       // OB: B falls one half period behind here (pairs with A's barrier behind softmax(0))
  // ABL & 64 (bench only): segment stamps of waves 0 and 4 of block 0 - [softmax done | barrier passed | matrix done | barrier passed] per tile, s_memtime at points
  // where lgkmcnt is drained anyway - in the LDS above the ring, copied to the output buffer at the end (sdm_bench_attn prints the averages)
  unsigned long long* trc = (unsigned long long*)(smem + R * SLOT) + (wave >> 2) * (8 * 28);
  const bool tracing = (ABL & 64) && blockIdx.x == 0 && (wave & 3) == 0;
  auto stamp = [&](int t, int k) { if ((ABL & 64) && tracing && t < 28 && lane == 0) trc[t * 8 + k] = __builtin_readcyclecounter(); };
  int bt = 0, b1 = 1, bD = D;                                          // slots of tiles t, t+1, t+D
  for (int t = 0; t < nwalk; ++t) {
    // ---- softmax segment of tile t.  KE >= 0: it opens with the DMAs of tile t+D.  The V^T fragments of tile t (and, KE >= 1, K fragments / biases of tile
    //      t+1) are read at its end, so that the matrix segment opens with MFMAs instead of an LDS round trip ----
    if (p.pp_flags & 2) SDM_SETPRIO(0);                                // (A/B: per-segment priority flips - matrix segments at priority 2)
    constexpr int NE = OB ? NDMA : DS;                                 // DMAs of this segment: the first NE - 2 at its head, the last two inside the max chain
    if (KE >= 0) dma(tq, bD, 0, NE >= 2 ? NE - 2 : NE);
    int tqn;                                                           // index of the tile the NEXT request (KE >= 0) / this iteration's matrix segment (KE < 0) asks for
    {
      int in_ = t + D + (KE >= 0 ? 1 : 0); if (in_ > nwalk - 1) in_ = nwalk - 1;
      if (LIST) SDM_SLOAD_I32(tqn, tlp + in_);                         // list walks: a scalar load that lands under the softmax (never a vector load: header)
      else tqn = i0 + in_;
    }
    if (ABL & 64) stamp(t, 4);                                         // (DMAs issued)
    VFrag v0, v1;
    KFrag k0, k1;
    softmax(s, pf, [&]() { if (KE >= 0 && NE >= 2) dma(tq, bD, NE - 2, NE - 1); }, [&]() { if (KE >= 0 && NE >= 2) dma(tq, bD, NE - 1, NE); },
            [&]() { load_v(bt, 0, v0); load_v(bt, 1, v1); });
    if (ABL & 64) stamp(t, 5);                                         // (softmax VALU done: the probabilities are pinned in front of this point)
    if (KE >= 1) { load_bias(b1, s); load_k(b1, 0, k0); }
    if (KE >= 2) load_k(b1, 1, k1);
    if (LIST) SDM_SLOAD_WAIT(tqn);
    if (ABL & 64) { SDM_WAIT_LGKMCNT0(); stamp(t, 0); }
    if (!OB) lds_barrier();
    else if (!grp) { if (!(ABL & 8)) { if (BIAS) SDM_WAIT_VMCNT(8); else SDM_WAIT_VMCNT(6); } lds_barrier(); }
    if (ABL & 64) stamp(t, 1);
    // ---- matrix segment: P.V of tile t (operands in registers), then Q.K^T of tile t+1.  KE < 2: the remaining K fragments arrive under the P.V MFMAs;
    //      KE < 0: tile t+D is requested here ----
    if (p.pp_flags & 2) SDM_SETPRIO(2);
    if (KE < 0) dma(tqn, bD);
    mma_v(v0, pf[0]);
    mma_v(v1, pf[1]);
    if (KE < 1) { load_bias(b1, s); load_k(b1, 0, k0); }
    if (KE < 2) load_k(b1, 1, k1);
    if (KE < 1) {
#pragma unroll
      for (int i = 0; i < 12; ++i) { SDM_SCHED_GROUP(0x008, 1, 0); SDM_SCHED_GROUP(0x100, 2, 0); SDM_SCHED_GROUP(0x002, 1, 0); }
    } else if (KE < 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { SDM_SCHED_GROUP(0x008, 1, 0); SDM_SCHED_GROUP(0x100, 1, 0); }
    }
    SDM_SCHED_FENCE();
    mma_k(k0, s[0]);
    mma_k(k1, s[1]);
    SDM_SCHED_FENCE();
    // tile t+2 (requested two tiles ago) must have landed; the whole batch of tile t+3 and the early part of tile t+4 may stay in flight; then the late part of
    // tile t+4 goes out (KE >= 0) - in the time this wave would wait for its partner's softmax segment anyway
    if (!(ABL & 8) && !OB) {
      if (KE < 0) { if (BIAS) SDM_WAIT_VMCNT(4); else SDM_WAIT_VMCNT(3); }
      else if (NDMA + DS == 3) SDM_WAIT_VMCNT(3); else if (NDMA + DS == 4) SDM_WAIT_VMCNT(4); else if (NDMA + DS == 5) SDM_WAIT_VMCNT(5);
      else if (NDMA + DS == 6) SDM_WAIT_VMCNT(6); else if (NDMA + DS == 7) SDM_WAIT_VMCNT(7); else SDM_WAIT_VMCNT(8);
    }
    if (KE >= 0 && !OB) dma(tq, bD, DS, NDMA);
    if (ABL & 64) { asm volatile("s_nop 0" :: "v"(s[0][0]), "v"(s[1][15])); SDM_WAIT_LGKMCNT0(); stamp(t, 2); }      // (the stamp waits for the last MFMA's result)
    if (!OB) lds_barrier();
    else if (grp) { if (!(ABL & 8)) { if (BIAS) SDM_WAIT_VMCNT(8); else SDM_WAIT_VMCNT(6); } lds_barrier(); }
    if (ABL & 64) stamp(t, 3);
    tq = tqn;
    bt = b1; b1 = b1 == R - 1 ? 0 : b1 + 1; bD = bD == R - 1 ? 0 : bD + 1;
  }
  if (!grp) lds_barrier();                                             // B's last matrix segment still reads V^T: the epilogue reuses the buffers
  SDM_WAIT_VMCNT0();                                                   // (the clamped DMAs of the last segments still write their slots)
  lds_barrier();
  if ((ABL & 64) && blockIdx.x == 0) {                                 // trace -> 3584 bytes behind the output buffer (the bench reads them back)
    const unsigned long long* tr0 = (const unsigned long long*)(smem + R * SLOT);
    for (int i = tid; i < 2 * 8 * 28; i += 512) ((unsigned long long*)p.part_ml)[i] = tr0[i];      // (bench: part_ml points behind the output)
    __syncthreads();
  }

  // epilogue (fp32 output): per-wave staging [32 q][64 d] at pitch 272 B -> coalesced 16-byte row stores
  constexpr int PS = 272;
  unsigned char* stf = smem + wave * (32 * PS);
  const float inv = p.nsplit > 1 ? 1.0f : 1.0f / ls[0];                 // key split: unnormalised partial sums (attn_combine_kernel divides)
  float* obase = (float*)p.o + (p.nsplit > 1 ? (size_t)sp * p.part_stride : (size_t)0) + (size_t)b * p.o_bs;
  if (p.nsplit > 1 && hi == 0 && q0 + l31 < p.Lq) {
    float* ml = p.part_ml + ((((size_t)sp * p.batch + b) * p.heads + head) * p.Lq + q0 + l31) * 2;
    ml[0] = m_i; ml[1] = ls[0];
  }
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = o[dt][4 * g + e] * inv;
      *(f32x4*)(stf + l31 * PS + (dt * 32 + 8 * g + 4 * hi) * 4) = h;
    }
  SDM_WAVE_SYNC();
  if (p.o_p3 && p.nsplit <= 1) { attn_store_p3(p, stf, PS, b, q0, head, lane); return; }
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    const int row = pass * 4 + (lane >> 4), part = lane & 15;
    const int qg = q0 + row;
    if (qg < p.Lq)
      *(f32x4*)(obase + (size_t)qg * p.ldo + head * 64 + part * 4) = *(const f32x4*)(stf + row * PS + part * 16);
  }
}

// ------------------------------------------------------------------------------------------------
// d = 512, single head (VAE mid-block).  8 waves: wave w handles queries 32*(w>>1).. and the d-half
// (w&1): partial S^T over its 256 d, exchanged with the partner wave through LDS; both then run the
// same softmax and each accumulates O^T for its own 256 d.  32-key tiles.
// ------------------------------------------------------------------------------------------------
#define ATTN512_X_BYTES (8 * 16 * 64 * 4)

// ------------------------------------------------------------------------------------------------
// d = 512, pipelined (8 waves = 4 query groups x 2 d-halves as described above): the K and
// V^T tiles arrive by LDS-DMA into DOUBLE-BUFFERED, unpadded, XOR-swizzled LDS images, so tile t+1 streams in while
// tile t is multiplied (a synchronous load -> LDS write -> barrier version spent ~4/5 of its time waiting;
// there are no registers left to prefetch through).  LDS: 2 x 32 KB K + 2 x 32 KB V^T + 32 KB exchange = all 160 KB.
//   K image  [32 rows][64 chunks of 16 B]: row rho holds key k0 + pi(rho), chunk position c' holds source chunk
//            c' ^ (rho & 15)  -> conflict-free ds_read_b128 down a column of 16 rows.
//   V^T image [512 d][4 chunks]: position c' of row r holds source chunk c' ^ ((r >> 2) & 3).
//   pi swaps bits 2 and 3 of the MFMA row index, so that the 8 S^T values a lane owns per 16-key step are 8
//   CONSECUTIVE keys and the matching V^T fragment is ONE aligned ds_read_b128 (was two ds_read_b64).
// ------------------------------------------------------------------------------------------------
#define ATTN512P_K_BYTES (32 * 1024)
#define ATTN512P_V_BYTES (512 * 64)
#define ATTN512P_X_OFF (2 * ATTN512P_K_BYTES + 2 * ATTN512P_V_BYTES)
#define ATTN512P_SMEM (ATTN512P_X_OFF + ATTN512_X_BYTES)

// ABL (bench only, sdm_bench_attn; 0 in the engine): 1 no exchange / softmax, 2 no P.V MFMAs, 4 no Q.K^T MFMAs, 8 no DMAs, 32 no fragment reads
template <int ABL = 0>
__global__ void __launch_bounds__(512) attn_d512_kernel(AttnParams p) {
  SDM_DYN_SMEM(smem);
  float* Xs = (float*)(smem + ATTN512P_X_OFF);
  const int tid = threadIdx.x, lane = tid & 63, wave = SDM_UNIFORM_I(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int qgp = wave >> 1, dh = wave & 1;
  int b, head_unused, qblk;
  if (!attn_block_coords(p, blockIdx.x, b, head_unused, qblk)) return;
  const int q0 = qblk * 128 + qgp * 32;

  f16x8 qf[16];
  {
    int qrow = q0 + l31;
    if (qrow > p.Lq - 1) qrow = p.Lq - 1;
    const half_t* qp = p.q + (size_t)b * p.q_bs + (size_t)qrow * p.ldq + dh * 256 + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) qf[ks] = *(const f16x8*)(qp + ks * 16);
  }
  f32x16 o[8];
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.0f;
  float m_i = SDM_NEG_BIG, l_i = 0.0f;

  const half_t* kbase = p.k + (size_t)b * p.k_bs;
  const half_t* vbase = p.vt + (size_t)b * p.vt_bs;
  const int ntiles = (p.Lk + 31) / 32;

  // DMA descriptors of this wave: 4 K rows (rho = 4*wave + i) and 4 V^T row groups (16 d-rows each, g = 4*wave + i).
  // Buffer form: SGPR descriptors + one 32-bit per-lane offset; row/tile offsets are wave-uniform (SGPR).
  const sdm_rsrc rsK = sdm_make_rsrc(kbase, (unsigned int)((size_t)p.Lk * p.ldk * 2));
  const sdm_rsrc rsV = sdm_make_rsrc(vbase, (unsigned int)((size_t)512 * p.ldvt * 2));
  const unsigned int v_voff = (unsigned int)((lane >> 2) * p.ldvt * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));   // ((16g + (lane>>2)) >> 2) & 3 == (lane >> 4) & 3
  auto issue_tile = [&](int t, int buf) {
    if (ABL & 8) return;
    const int k0 = t * 32;
    unsigned char* Kd = smem + buf * ATTN512P_K_BYTES;
    unsigned char* Vd = smem + 2 * ATTN512P_K_BYTES + buf * ATTN512P_V_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rho = 4 * wave + i;
      const int pi = (rho & 16) | (((rho >> 2) & 1) << 3) | (((rho >> 3) & 1) << 2) | (rho & 3);
      int kr = k0 + pi;
      if (kr > p.Lk - 1) kr = p.Lk - 1;                              // keys >= Lk are masked after QK^T
      sdm_glds16_buf(rsK, (unsigned int)((lane ^ (rho & 15)) * 16), (unsigned int)kr * (unsigned int)(p.ldk * 2), Kd + rho * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = 4 * wave + i;
      sdm_glds16_buf(rsV, v_voff, (unsigned int)((16 * g * p.ldvt + k0) * 2), Vd + g * 1024);
    }
  };
  // read-side swizzles (the same involutions as on the source side)
  int kx = (l31 * 1024 + dh * 512) ^ ((hi ^ (l31 & 15)) * 16);      // K: byte offset of chunk (2*ks + hi) ^ (rho & 15) = kx ^ (ks * 32)
  int vx = (dh * 256 + l31) * 64 + ((hi ^ ((l31 >> 2) & 3)) * 16);   // V^T: chunk (2*u + hi) ^ ((r >> 2) & 3) -> vx ^ (u * 32)

  issue_tile(0, 0);
  SDM_WAIT_VMCNT0();
  SDM_RAW_BARRIER();
  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * 32, cur = t & 1;
    const unsigned char* Ks = smem + cur * ATTN512P_K_BYTES;
    const unsigned char* Vs = smem + 2 * ATTN512P_K_BYTES + cur * ATTN512P_V_BYTES;
    if (t + 1 < ntiles) issue_tile(t + 1, cur ^ 1);                  // in flight during everything below
    SDM_OPAQUE_I(kx);
    SDM_OPAQUE_I(vx);
    // Fragment reads run THREE MFMAs ahead of their use through four rotating registers (round 6): the compiler's own schedule issued each read one MFMA
    // ahead and waited lgkmcnt(0) in front of every MFMA - an LDS round trip (~100+ cycles) per 32-cycle MFMA, covered only two-fold by the SIMD's other wave:
    // matrix pipe 0.40 busy (profiles/r06_attn_sq_counters.txt).  The scheduling groups pin one read per MFMA.
    f16x8 a[4];
    auto rd_k = [&](int ks) -> f16x8 { if (ABL & 32) return qf[(ks + 1) & 15]; return *(const f16x8*)(Ks + (kx ^ (ks * 32))); };
    auto rd_v = [&](int i) -> f16x8 { if (ABL & 32) return qf[i & 15]; return *(const f16x8*)(Vs + (vx ^ ((i >> 3) * 32)) + (i & 7) * 2048); };      // i = 8 u + dt
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
    for (int ks = 0; ks < 3; ++ks) a[ks] = rd_k(ks);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 3 < 16) a[(ks + 3) & 3] = rd_k(ks + 3);
      if (!(ABL & 4)) s = SDM_MFMA_32x32x16_F16(a[ks & 3], qf[ks], s);
    }
    if (!(ABL & 36)) {
      SDM_SCHED_GROUP(0x100, 3, 0);
#pragma unroll
      for (int i = 0; i < 13; ++i) { SDM_SCHED_GROUP(0x100, 1, 0); SDM_SCHED_GROUP(0x008, 1, 0); }
      SDM_SCHED_GROUP(0x008, 3, 0);
    }
    SDM_SCHED_FENCE();
    float* xme = Xs + wave * (16 * 64);
    const float* xpt = Xs + (wave ^ 1) * (16 * 64);
    if (!(ABL & 1)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) xme[r * 64 + lane] = s[r];
    }
    SDM_WAIT_LGKMCNT0();
    SDM_RAW_BARRIER();
    // the first V^T fragments are requested before the softmax: they land underneath it
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = rd_v(i);
    float mx = SDM_NEG_BIG;
    if (!(ABL & 1)) {
    if (k0 + 32 > p.Lk) {                                              // keys >= Lk: only the last tile can hold any (wave-uniform)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = k0 + 16 * (r >> 3) + 8 * hi + (r & 7);        // pi-permuted row -> actual key
        float x = (s[r] + xpt[r * 64 + lane]) * p.scale_log2e;
        if (key >= p.Lk) x = SDM_NEG_BIG;
        s[r] = x;
        mx = fmaxf(mx, x);
      }
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float x = (s[r] + xpt[r * 64 + lane]) * p.scale_log2e;
        s[r] = x;
        mx = fmaxf(mx, x);
      }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mnew = fmaxf(m_i, mx);
    const float alpha = sdm_exp2(m_i - mnew);
    m_i = mnew;
    float rs = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = sdm_exp2(s[r] - mnew);
      s[r] = pv;
      rs += pv;
    }
    l_i = l_i * alpha + rs;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int dt = 0; dt < 8; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
    }
    } else l_i = 1.0f;
    f16x8 pf[2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 8; ++j) pf[u][j] = (half_t)s[8 * u + j];
    SDM_SCHED_FENCE();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (i + 3 < 16) a[(i + 3) & 3] = rd_v(i + 3);
      if (!(ABL & 2)) o[i & 7] = SDM_MFMA_32x32x16_F16(a[i & 3], pf[i >> 3], o[i & 7]);
    }
    if (!(ABL & 34)) {
#pragma unroll
      for (int i = 0; i < 13; ++i) { SDM_SCHED_GROUP(0x100, 1, 1); SDM_SCHED_GROUP(0x008, 1, 1); }
      SDM_SCHED_GROUP(0x008, 3, 1);
    }
    SDM_SCHED_FENCE();
    SDM_WAIT_VMCNT0();          // this wave's pieces of tile t+1 have landed ...
    SDM_RAW_BARRIER();          // ... and so have everyone else's; every wave is done with tile t and the exchange buffer
  }

  l_i += __shfl_xor(l_i, 32);
  const float inv = 1.0f / l_i;
  if (p.o_f32) {                   // fp32 output (precise-mode graphs): [32 q][128 d] fp32 staging per wave, pitch 528 B
    constexpr int PF = 528;
    unsigned char* stf = smem + wave * (32 * PF);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      __syncthreads();
#pragma unroll
      for (int d4 = 0; d4 < 4; ++d4) {
        const int dt = half * 4 + d4;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = o[dt][4 * g + e] * inv;
          *(f32x4*)(stf + l31 * PF + (d4 * 32 + 8 * g + 4 * hi) * 4) = h;
        }
      }
      __syncthreads();
#pragma unroll
      for (int pass = 0; pass < 16; ++pass) {
        const int row = pass * 2 + (lane >> 5), part = lane & 31;
        const int qg = q0 + row;
        if (qg < p.Lq)
          *(f32x4*)((float*)p.o + (size_t)b * p.o_bs + (size_t)qg * p.ldo + dh * 256 + half * 128 + part * 4) =
              *(const f32x4*)(stf + row * PF + part * 16);
      }
    }
    return;
  }
  constexpr int PS = 272;
  unsigned char* stg = smem + wave * (32 * PS);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    __syncthreads();
#pragma unroll
    for (int d4 = 0; d4 < 4; ++d4) {
      const int dt = half * 4 + d4;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f16x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)(o[dt][4 * g + e] * inv);
        *(f16x4*)(stg + l31 * PS + (d4 * 32 + 8 * g + 4 * hi) * 2) = h;
      }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
      const int row = pass * 4 + (lane >> 4), part = lane & 15;
      const int qg = q0 + row;
      if (qg < p.Lq)
        *(f16x8*)(p.o + (size_t)b * p.o_bs + (size_t)qg * p.ldo + dh * 256 + half * 128 + part * 8) =
            *(const f16x8*)(stg + row * PS + part * 16);
    }
  }
}

// Active key tiles of every image (see AttnParams::tiles): one block per image.  bias is the log2-domain key bias [B][Lk].
// Key split, second pass: out[b][row][head*64 + d] = sum_s o_s * 2^(m_s - M) / sum_s l_s * 2^(m_s - M), M = max_s m_s (log2 domain, as the kernels keep it).
// A block row that walked no contributing key has l_s = 0 or m_s far below M: weight exactly 0 - the same zero its keys have in the unsplit walk.
// 16 threads per (row, head), 4 channels each; grid = ceil(batch * Lq * heads * 16 / 256).
__global__ void __launch_bounds__(256) attn_combine_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, float* __restrict__ out,
                                                           int nsplit, long part_stride, int batch, int heads, int Lq, long o_bs, int ldo) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int part = (int)(t & 15);
  const long rh = t >> 4;                       // (b * Lq + row) * heads + head
  if (rh >= (long)batch * Lq * heads) return;
  const int head = (int)(rh % heads);
  const long br = rh / heads;
  const int row = (int)(br % Lq), b = (int)(br / Lq);
  float M = SDM_NEG_BIG;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[((((size_t)s * batch + b) * heads + head) * Lq + row) * 2]);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float L = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float* ml = part_ml + ((((size_t)s * batch + b) * heads + head) * Lq + row) * 2;
    const float w = sdm_exp2(ml[0] - M);
    L += ml[1] * w;
    const f32x4 v = *(const f32x4*)(part_o + (size_t)s * part_stride + (size_t)b * o_bs + (size_t)row * ldo + head * 64 + part * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] += v[e] * w;
  }
  const float inv = 1.0f / L;
#pragma unroll
  for (int e = 0; e < 4; ++e) acc[e] *= inv;
  *(f32x4*)(out + (size_t)b * o_bs + (size_t)row * ldo + head * 64 + part * 4) = acc;
}

// the same with the result as P3 planes (AttnParams::o_p3): 8 threads per (row, head), 8 channels each
__global__ void __launch_bounds__(256) attn_combine_p3_kernel(const float* __restrict__ part_o, const float* __restrict__ part_ml, unsigned char* __restrict__ out,
                                                              long xl_off, int nsplit, long part_stride, int batch, int heads, int Lq, long o_bs, int ldo) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int part = (int)(t & 7);
  const long rh = t >> 3;
  if (rh >= (long)batch * Lq * heads) return;
  const int head = (int)(rh % heads);
  const long br = rh / heads;
  const int row = (int)(br % Lq), b = (int)(br / Lq);
  float M = SDM_NEG_BIG;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, part_ml[((((size_t)s * batch + b) * heads + head) * Lq + row) * 2]);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  float L = 0.0f;
  for (int s = 0; s < nsplit; ++s) {
    const float* ml = part_ml + ((((size_t)s * batch + b) * heads + head) * Lq + row) * 2;
    const float w = sdm_exp2(ml[0] - M);
    L += ml[1] * w;
    const float* src = part_o + (size_t)s * part_stride + (size_t)b * o_bs + (size_t)row * ldo + head * 64 + part * 8;
    const f32x4 v0 = *(const f32x4*)src, v1 = *(const f32x4*)(src + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[e] += v0[e] * w; acc[4 + e] += v1[e] * w; }
  }
  const float inv = 1.0f / L;
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] *= inv;
  p3_store8(acc, out, out + xl_off, (size_t)b * Lq + row, ldo, head * 64 + part * 8);
}

__global__ void __launch_bounds__(256) attn_active_tiles_kernel(const float* __restrict__ bias, int Lk, int ntiles, int* __restrict__ out,
                                                                int out_bs, float margin) {
  SDM_SHARED float red[256];
  SDM_SHARED unsigned char flag[4096];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* bb = bias + (size_t)b * Lk;
  float mx = SDM_NEG_BIG;
  for (int k = tid; k < Lk; k += 256) mx = fmaxf(mx, bb[k]);
  red[tid] = mx;
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if (tid < st) red[tid] = fmaxf(red[tid], red[tid + st]);
    __syncthreads();
  }
  const float lim = red[0] - margin;
  int* o = out + (size_t)b * out_bs;
  int n = 0;                                   // meaningful in thread 0 only
  for (int t0 = 0; t0 < ntiles; t0 += 4096) {  // chunks of 4096 tiles (one pass for every size the engine uses)
    const int nt = (ntiles - t0) < 4096 ? (ntiles - t0) : 4096;
    for (int t = tid; t < nt; t += 256) {
      float tm = SDM_NEG_BIG;
      const int k0 = (t0 + t) * 64, k1 = (k0 + 64 < Lk) ? k0 + 64 : Lk;
      for (int k = k0; k < k1; ++k) tm = fmaxf(tm, bb[k]);
      flag[t] = tm >= lim ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0)
      for (int t = 0; t < nt; ++t)
        if (flag[t]) o[1 + n++] = t0 + t;
    __syncthreads();
  }
  if (tid == 0) o[0] = n;
}

// V [b][key][head*D + d]  ->  V^T [b][head][d][key] (key padded to ldvt with zeros).  64 keys x 64 d per block: 16-byte
// global loads along d, 16-byte global stores along the keys; the transposition goes through a 64 x 65-dword LDS tile (one
// value per dword, so that both the row-wise writes and the column-wise reads are 4-byte accesses with at most 2-way conflicts).
__global__ void __launch_bounds__(256) transpose_v_kernel(const half_t* __restrict__ v, long v_bs, int ldv, half_t* __restrict__ vt,
                                                          long vt_bs, long vt_hs, int ldvt, int Lk, int D) {
  SDM_SHARED float tile[64][65];
  const int b = blockIdx.z;
  const int dblocks = D / 64;
  const int head = blockIdx.y / dblocks, d0 = (blockIdx.y % dblocks) * 64;
  const int k0 = blockIdx.x * 64;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {                      // 512 vectors of 8 d: vector = (key, d-octet)
    const int vec = tid + i * 256;
    const int key = vec >> 3, oct = vec & 7;
    f16x8 x;
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = (half_t)0.0f;
    if (k0 + key < Lk) x = *(const f16x8*)(v + (size_t)b * v_bs + (size_t)(k0 + key) * ldv + head * D + d0 + oct * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[key][oct * 8 + e] = (float)x[e];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {                      // 512 vectors of 8 keys: vector = (d, key-octet)
    const int vec = tid + i * 256;
    const int d = vec >> 3, oct = vec & 7;
    f16x8 y;
#pragma unroll
    for (int e = 0; e < 8; ++e) y[e] = (half_t)tile[oct * 8 + e][d];
    if (k0 + oct * 8 < ldvt)                         // ldvt is a multiple of 64: whole octets
      *(f16x8*)(vt + (size_t)b * vt_bs + (size_t)head * vt_hs + (size_t)(d0 + d) * ldvt + k0 + oct * 8) = y;
  }
}
