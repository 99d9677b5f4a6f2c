// k_conv_f8s.h - the fp8-residual 3x3 convolution (k_conv.h, F8) with ROLE-SWAPPING wave groups.
//
// Same arithmetic, same operand layouts in LDS, same weight stream as conv_mfma_kernel<9, 1, 8, 32, 128, 32, 2, 2, 1, 0, GN, 1, 1, 1, 1>
// (reference call sites: the ResnetBlock2D / Up- / Downsample2D convolutions of the VAE and the U-Net, meta_arch.py:142,209,256 and
// replace.py:462-544) - what changes is who does what, and when.
//
// What the step tracer of round 4 measured in that kernel (profiles/r04_conv_f8_step_trace.txt): between two tiles the matrix pipes of
// all four SIMDs idle for 13-14 k cycles (25 k with a residual) of a 65 k-cycle 128-channel tile.  The four consumer waves hold the
// 128 KB output tile in their accumulators; the CU accepts one 1 KB store per ~65 cycles (8 k cycles per tile) whoever issues it; the
// statistics cost another 4 k; the next tile's residual queues behind the stores (8-12 k more); and the producer waves, whose 1 KB
// per lane of registers is the only storage on the CU that is not full, wait at a barrier all the while.
//
// Here the block still has two groups of four waves (one wave of each group on every SIMD), but a group is CONSUMER of one tile and
// PRODUCER of the next, in turn.  The moment a tile's last MFMA has issued, the other group - which has just staged the next tile's
// first chunk and weights - starts multiplying it with its own, free accumulators; the group that holds the finished tile becomes
// the producer and drains its accumulators in slices (bias, residual, statistics, four 1 KB stores per wave and step) between its
// producer duties of the next tile's first ten steps, paced so that the CU's memory pipeline takes the stores, the weight DMAs and
// the activation loads of a step within that step.  The residual is added there (no residual-init loads in front of the MFMAs), and
// the step barriers form one uninterrupted sequence over all tiles of a block.
//
// Producer duties per 32-channel chunk (six steps): step 0 loads the raw activations of the NEXT chunk of the sequence (of this tile or
// chunk 0 of the next one); steps 2-5 transform them (GroupNorm, SiLU, fp16 high part + e5m2 residual pair) into the other A buffer;
// every step issues the weight DMAs two steps ahead.  Restrictions (the launcher falls back to the one-role-per-wave kernel otherwise):
// full 8 x 32 tiles, fp32 in / out, linear epilogue with out_scale 1, every output channel of a tile valid, an even number >= 4 of
// chunks.
#pragma once
#include "k_conv.h"

#ifdef SDM_EMU
#define SDM_ALWAYS_INLINE
#else
#define SDM_ALWAYS_INLINE __attribute__((always_inline))
#endif

struct ConvF8S {
  static constexpr int TH = 8, TW = 32, BN = 128, KC = 32, NT = 256, MT = 4, NTL = 2, WTM = 128, WTN = 64;
  static constexpr int HPW = 34, HPH = 10, HP = 340, KV = 4, A_PER = 6, A_VEC = HP * KV;
  static constexpr int A_BYTES = HP * 64;             // fp16 high parts of one chunk: four planes of 16-byte rows; the fp8 region behind is as large
  static constexpr int A_BUF = 2 * A_BYTES;
  static constexpr int SLOT = 96 * BN, STEP = 2 * SLOT, PLANE = 3 * BN * 16;
  static constexpr int TILE_BYTES = 2 * A_BUF + 3 * STEP;
  static constexpr int SMEM = TILE_BYTES + 2 * BN * 4;        // + two bias tables
  static constexpr int A_HALF = HP * 16;
};

template <int GN>
__global__ void __launch_bounds__(512, 1) conv3x3_f8_swap_kernel(ConvParams p) {
  using C = ConvF8S;
  constexpr int TH = C::TH, TW = C::TW, BN = C::BN, NT = C::NT, MT = C::MT, NTL = C::NTL, HPW = C::HPW, KV = C::KV, A_PER = C::A_PER;
  constexpr int A_BYTES = C::A_BYTES, A_BUF = C::A_BUF, A_HALF = C::A_HALF, SLOT = C::SLOT, STEP = C::STEP, PLANE = C::PLANE, WTM = C::WTM, WTN = C::WTN, NR = MT + 2;
  constexpr float F8_LS = 2048.0f, F8_AMAX = 57344.0f;
  SDM_DYN_SMEM(smem);
  unsigned char* Aring = smem;
  unsigned char* Bring = smem + 2 * A_BUF;
  float* bias_tabs = (float*)(smem + C::TILE_BYTES);

  const int grp = SDM_UNIFORM_I((int)threadIdx.x / NT);              // wave group 0 / 1
  const int Cin = p.C0 + p.C1, nch = Cin / 32, nsteps = nch * 6;
  const unsigned int es = 4u;
  const int ntiles_blk = [&]() SDM_ALWAYS_INLINE {      // valid tiles of this block: vbid = blockIdx.x + k * gridDim.x (the tail of an XCD's range is padding)
    int n = 0;
    for (int k = 0;; ++k) {
      const int bid = (int)blockIdx.x + k * (int)gridDim.x;
      if (bid >= p.vgrid) break;
      const int mlin = (bid & 7) * p.xcd_chunk + (bid >> 3) / p.tiles_n;
      if (mlin >= p.tiles_m * p.N) break;
      ++n;
    }
    return n;
  }();
  if (ntiles_blk == 0) return;
#if defined(SDM_CONV_TRACE) && !defined(SDM_EMU)
  int tr_n = 0;
#endif

  struct TileXY { int img, mt, oy0, ox0, n0; };
  auto tile_of = [&](int k) SDM_ALWAYS_INLINE {
    const int bid = (int)blockIdx.x + k * (int)gridDim.x, j = bid >> 3, ml = j / p.tiles_n, mlin = (bid & 7) * p.xcd_chunk + ml;
    TileXY t;
    t.img = mlin / p.tiles_m; t.mt = mlin - t.img * p.tiles_m;
    const int npx = p.Wout / TW;
    t.oy0 = (t.mt / npx) * TH; t.ox0 = (t.mt % npx) * TW; t.n0 = (j - ml * p.tiles_n) * BN;
    return t;
  };

  // accumulators [channel][pixel] (k_conv.h, F8): register r of lane l = channel (r & 3) + 8 (r >> 2) + 4 (l >> 5) of pixel l & 31 of a 32 x 32
  // sub-tile.  The ONLY per-lane state that lives across a role change; everything else a role needs is derived from an opaque copy of
  // threadIdx.x inside that role, so that it is not carried (i.e. spilled) through the other role's code.
  f32x16 acc[MT][NTL];
  auto zero_acc = [&]() SDM_ALWAYS_INLINE {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  };

  // ==================================================================================================================
  // consumer of one tile
  // ==================================================================================================================
  auto consume_tile = [&]() SDM_ALWAYS_INLINE {
    int tx = (int)threadIdx.x;
    SDM_OPAQUE_I(tx);
    const int tid = tx & (NT - 1), lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int abase0 = ((wm * (WTM / TW)) * HPW + (lane & 31)) * 16 + (lane >> 5) * A_HALF;      // halo row of this wave's first output row, pixel lane & 31
    const int a8base = abase0 - (lane >> 5) * A_HALF + (lane >> 5) * 2 * A_HALF;
    int bq[NTL], bq8[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      bq[j] = (lane >> 5) * PLANE + (wn * WTN + j * 32 + (lane & 31)) * 16;
      bq8[j] = (lane >> 5) * SLOT + (wn * WTN + j * 32 + (lane & 31)) * 16;
    }
    const int sa8 = p.f8_sa, sb8 = p.f8_sb;
    f16x8 fbh[2][3][NTL];
    i32x8 fb8[3][NTL];
    auto ld_bh = [&](int ks, int dy, const unsigned char* Bp) SDM_ALWAYS_INLINE {
#pragma unroll
      for (int j = 0; j < NTL; ++j) fbh[ks][dy][j] = *(const f16x8*)(Bp + ks * SLOT + bq[j] + dy * (BN * 16));
    };
    auto ld_b8 = [&](int dy, const unsigned char* Bp) SDM_ALWAYS_INLINE {
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const i32x4 q0 = *(const i32x4*)(Bp + bq8[j] + dy * (BN * 16)), q1 = *(const i32x4*)(Bp + bq8[j] + PLANE + dy * (BN * 16));
        fb8[dy][j] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
      }
    };
    auto ld_ah = [&](const unsigned char* Ap, int ks, int r, int dx) SDM_ALWAYS_INLINE { return *(const f16x8*)(Ap + abase0 + ks * 2 * A_HALF + (r * HPW + dx) * 16); };
    auto ld_a8 = [&](const unsigned char* Ap, int r, int dx) SDM_ALWAYS_INLINE {
      const unsigned char* q = Ap + A_BYTES + a8base + (r * HPW + dx) * 16;
      const i32x4 q0 = *(const i32x4*)q, q1 = *(const i32x4*)(q + A_HALF);
      return i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
    };
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) ld_bh(0, dy, Bring);      // operands of the first step (landed two barriers ago)
    for (int c = 0; c < nch; ++c) {
      const bool more = c + 1 < nch;
      const unsigned char* Ab = Aring + (c & 1) * A_BUF;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const unsigned char* B1 = Bring + ((dx * 2) % 3) * STEP;
        const unsigned char* B2 = Bring + ((dx * 2 + 1) % 3) * STEP;
        f16x8 fa[2];
        i32x8 f8a[2];
        fa[0] = ld_ah(Ab, 0, 0, dx);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          fa[(r + 1) & 1] = (r + 1 < NR) ? ld_ah(Ab, 0, r + 1, dx) : ld_ah(Ab, 1, 0, dx);
          if (r < 3) ld_bh(1, r, B1);
          SDM_SCHED_FENCE();
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int i = r - dy;
            if (i >= 0 && i < MT) {
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(fbh[0][dy][j], fa[r & 1], acc[i][j]);
            }
          }
          SDM_SCHED_FENCE();
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (r + 1 < NR) fa[(NR + r + 1) & 1] = ld_ah(Ab, 1, r + 1, dx);
          else f8a[0] = ld_a8(Ab, 0, dx);
          if (r < 3) ld_b8(r, B2);
          SDM_SCHED_FENCE();
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int i = r - dy;
            if (i >= 0 && i < MT) {
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(fbh[1][dy][j], fa[(NR + r) & 1], acc[i][j]);
            }
          }
          SDM_SCHED_FENCE();
        }
        SDM_RAW_BARRIER();
        const bool last = (dx == 2) && !more;
        const unsigned char* Bn = Bring + ((dx * 2 + 2) % 3) * STEP;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
          if (r + 1 < NR) f8a[(r + 1) & 1] = ld_a8(Ab, r + 1, dx);
          if (r < 3 && !last) ld_bh(0, r, Bn);
          SDM_SCHED_FENCE();
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int i = r - dy;
            if (i >= 0 && i < MT) {
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x64_F8A_BF8B(fb8[dy][j], f8a[r & 1], acc[i][j], sb8, sa8);
            }
          }
          SDM_SCHED_FENCE();
        }
        SDM_RAW_BARRIER();
      }
    }
  };

  // ==================================================================================================================
  // the drain of a finished tile's accumulators, in 8 slices of 4 register quads: slice u = (j, g) = (u / 4, u % 4), i.e. the four consecutive
  // channels ch_lane + 32 j + 8 g .. + 3 of this lane's pixel in each of the wave's four 32-pixel rows.  bias, residual, 16-byte stores, and
  // the GroupNorm statistics of the consumer: sums over the wave's 128 pixels (in-lane over the rows, DPP over the 16 lanes of a row, one
  // cross-row exchange), written by two lanes per slice as one partial row per (tile, wave row) - no atomics, deterministic.
  // `et`: the tile being drained (uniform); lane-derived values come in through DrainLane (computed inside the role that drains).
  // ==================================================================================================================
  TileXY et = {0, 0, 0, 0, 0};
  int et_par = 0;
  struct DrainLane { int hi, l31, wn, wm, wmu; };
  auto drain_lane = [&]() SDM_ALWAYS_INLINE {
    int tx = (int)threadIdx.x;
    SDM_OPAQUE_I(tx);
    const int tid = tx & (NT - 1), lane = tid & 63, wave = tid >> 6;
    DrainLane d;
    d.hi = lane >> 5; d.l31 = lane & 31; d.wn = wave & 1; d.wm = wave >> 1; d.wmu = SDM_UNIFORM_I(wave >> 1);
    return d;
  };
  auto et_px0 = [&]() SDM_ALWAYS_INLINE { return ((size_t)et.img * p.Hout + et.oy0) * p.Wout; };
  // residual of slice u -> rq (issued BEHIND a step's weight DMAs: nothing that waits for it can then wait for them)
  auto drain_prefetch = [&](const DrainLane& d, int u, f32x4 (&rq)[MT]) SDM_ALWAYS_INLINE {
    const unsigned int rs4 = (unsigned int)p.res_C * 4u;
    const sdm_rsrc rsr = sdm_make_rsrc((const unsigned char*)p.res + et_px0() * rs4, (unsigned int)((size_t)TH * p.Wout * rs4));
    const unsigned int vr = (unsigned int)d.l31 * rs4 + (unsigned int)(et.n0 + d.wn * WTN + 4 * d.hi + (u / 4) * 32 + 8 * (u % 4)) * 4u;
#pragma unroll
    for (int i = 0; i < MT; ++i) rq[i] = __builtin_bit_cast(f32x4, sdm_buffer_load16(rsr, vr, (unsigned int)((d.wmu * (WTM / TW) + i) * p.Wout + et.ox0) * rs4));
  };
  // arithmetic of slice u, in place: acc = acc + bias (+ residual); statistics of the stored values
  auto drain_valu = [&](const DrainLane& d, int u, const f32x4 (&rq)[MT]) SDM_ALWAYS_INLINE {
    const f32x4 b4 = *(const f32x4*)(bias_tabs + et_par * BN + d.wn * WTN + (u / 4) * 32 + 8 * (u % 4) + 4 * d.hi);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (j * 4 + g == u) {        // (every register index a literal of these loops: the accumulators stay in registers)
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float x = acc[i][j][4 * g + e] + b4[e];
              if (p.res) x += rq[i][e];
              acc[i][j][4 * g + e] = x;
              s1[e] += x; s2[e] += x * x;
            }
        }
      }
    if (p.stats) {
      f32x4 o0, o1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t1 = sdm_sum_row16(s1[e]), t2 = sdm_sum_row16(s2[e]);
        t1 += __shfl_xor(t1, 16); t2 += __shfl_xor(t2, 16);
        if (e < 2) { o0[2 * e] = t1; o0[2 * e + 1] = t2; } else { o1[2 * (e - 2)] = t1; o1[2 * (e - 2) + 1] = t2; }
      }
      if (d.l31 == 0) {
        const size_t prow = (size_t)et.img * (p.tiles_m * 2) + (size_t)et.mt * 2 + d.wm;
        float* st = p.stats + (prow * p.Cout_store + p.out_ch_off + et.n0 + d.wn * WTN + 4 * d.hi + (u / 4) * 32 + 8 * (u % 4)) * 2;
        *(f32x4*)st = o0;
        *(f32x4*)(st + 4) = o1;
      }
    }
  };
  // the four 16-byte stores of slice u, straight from the accumulator registers
  auto drain_store = [&](const DrainLane& d, int u) SDM_ALWAYS_INLINE {
    const unsigned int cs4 = (unsigned int)p.Cout_store * 4u;
    const sdm_rsrc rso = sdm_make_rsrc((unsigned char*)p.out + et_px0() * cs4, (unsigned int)((size_t)TH * p.Wout * cs4));
    const unsigned int vo = (unsigned int)d.l31 * cs4 + (unsigned int)(p.out_ch_off + et.n0 + d.wn * WTN + 4 * d.hi + (u / 4) * 32 + 8 * (u % 4)) * 4u;
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (j * 4 + g == u) {
#pragma unroll
          for (int i = 0; i < MT; ++i) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e];
            sdm_buffer_store16(__builtin_bit_cast(u32x4, v), rso, vo, (unsigned int)((d.wmu * (WTM / TW) + i) * p.Wout + et.ox0) * cs4);
            SDM_PIN_STORE_DATA(v);
          }
        }
      }
  };

  // ==================================================================================================================
  // producer of tile k (consumed by the other group).  first: the block's first tile - nothing is staged yet; drain: this group still
  // holds the accumulators of tile k - 1
  // ==================================================================================================================
  auto produce_tile = [&](int k, bool first, bool drain) SDM_ALWAYS_INLINE {
    int tx = (int)threadIdx.x;
    SDM_OPAQUE_I(tx);
    const int tid = tx & (NT - 1), lane = tid & 63;
    const int wv = SDM_UNIFORM_I(tid >> 6);
    const int Hl = p.Hin << p.up, Wl = p.Win << p.up;
    const int a_part = (tid % KV) * 8, a_hp0 = tid / KV;
    const TileXY cur = tile_of(k);
    const bool has_next = k + 1 < ntiles_blk;
    TileXY nxt = cur;
    if (has_next) nxt = tile_of(k + 1);
    int a_pix[A_PER];
    sdm_rsrc rs0, rs1;
    int st_img = 0;
    auto stage_setup = [&](const TileXY& t) SDM_ALWAYS_INLINE {
      const int band0 = (t.oy0 - p.pad_t) > 0 ? ((t.oy0 - p.pad_t) >> p.up) : 0;
      const int band_rows = (p.Hin - band0) < (C::HPH + 1) ? (p.Hin - band0) : (C::HPH + 1);
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const int hp = a_hp0 + i * (NT / KV);
        a_pix[i] = -1;
        if (tid + i * NT < C::A_VEC) {
          const int hy = hp / HPW, hx = hp % HPW;
          const int iy = t.oy0 + hy - p.pad_t, ix = t.ox0 + hx - p.pad_l;
          if (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) a_pix[i] = ((iy >> p.up) - band0) * p.Win + (ix >> p.up);
        }
      }
      const size_t base_px = ((size_t)t.img * p.Hin + band0) * p.Win, npxs = (size_t)band_rows * p.Win;
      rs0 = sdm_make_rsrc((const unsigned char*)p.in0 + base_px * p.C0 * es, (unsigned int)(npxs * p.C0 * es));
      rs1 = sdm_make_rsrc(p.in1 ? (const unsigned char*)p.in1 + base_px * p.C1 * es : (const unsigned char*)p.in0, p.in1 ? (unsigned int)(npxs * p.C1 * es) : 0u);
      st_img = t.img;
    };
    u32x4 a_nx[A_PER][2];
    f32x4 gq[4];
    auto issue_loads = [&](int c0) SDM_ALWAYS_INLINE {
      const bool second = c0 >= p.C0;
      const sdm_rsrc rs = second ? rs1 : rs0;
      const unsigned int Cs = (unsigned int)(second ? p.C1 : p.C0) * es;
      const unsigned int cc = (unsigned int)((second ? c0 - p.C0 : c0) + a_part) * es;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        const unsigned int off = a_pix[i] >= 0 ? (unsigned int)a_pix[i] * Cs + cc : SDM_BUF_INVALID;
        a_nx[i][0] = sdm_buffer_load16(rs, off, 0);
        a_nx[i][1] = sdm_buffer_load16(rs, off, 16);
      }
      if (GN) {
        const float* ts = p.gn_scale + (size_t)st_img * Cin + c0 + a_part;
        const float* th = p.gn_shift + (size_t)st_img * Cin + c0 + a_part;
        gq[0] = *(const f32x4*)ts; gq[1] = *(const f32x4*)(ts + 4); gq[2] = *(const f32x4*)th; gq[3] = *(const f32x4*)(th + 4);
      }
    };
    // raw vector i -> GroupNorm / SiLU -> fp16 high parts + e5m2 images of the low part and of the value itself (layout: k_conv.h, F8)
    auto transform_vecs = [&](unsigned char* Ad, int i0, int i1) SDM_ALWAYS_INLINE {
      const int g = a_part >> 3;
#pragma unroll
      for (int i = 0; i < A_PER; ++i) {
        if (i >= i0 && i < i1 && tid + i * NT < C::A_VEC) {
          const f32x4 v0 = __builtin_bit_cast(f32x4, a_nx[i][0]), v1 = __builtin_bit_cast(f32x4, a_nx[i][1]);
          const bool inside = a_pix[i] >= 0;
          f16x8 vh;
          float xl[8], xx[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float y = e < 4 ? v0[e & 3] : v1[e & 3];
            if (GN) {
              y = y * (e < 4 ? gq[0][e & 3] : gq[1][e & 3]) + (e < 4 ? gq[2][e & 3] : gq[3][e & 3]);
              if (p.gn_silu) y = y * sdm_rcp(1.0f + sdm_exp2(-y * SDM_LOG2E));
              if (!inside) y = 0.0f;
            }
            y = fminf(fmaxf(y, -F8_AMAX), F8_AMAX);
            const half_t h = (half_t)y;
            vh[e] = h;
            xl[e] = (y - (float)h) * F8_LS;
            xx[e] = y;
          }
          const int hp = a_hp0 + i * (NT / KV);
          *(f16x8*)(Ad + g * A_HALF + hp * 16) = vh;
          int l0 = SDM_CVT_PK_BF8(xl[0], xl[1], 0, false), l1 = SDM_CVT_PK_BF8(xl[4], xl[5], 0, false);
          l0 = SDM_CVT_PK_BF8(xl[2], xl[3], l0, true); l1 = SDM_CVT_PK_BF8(xl[6], xl[7], l1, true);
          int x0 = SDM_CVT_PK_BF8(xx[0], xx[1], 0, false), x1 = SDM_CVT_PK_BF8(xx[4], xx[5], 0, false);
          x0 = SDM_CVT_PK_BF8(xx[2], xx[3], x0, true); x1 = SDM_CVT_PK_BF8(xx[6], xx[7], x1, true);
          unsigned char* a8 = Ad + A_BYTES + (g >> 1) * A_HALF + hp * 16 + (g & 1) * 8;
          u32x2 wl, wx;
          wl[0] = (unsigned int)l0; wl[1] = (unsigned int)l1; wx[0] = (unsigned int)x0; wx[1] = (unsigned int)x1;
          *(u32x2*)a8 = wl;
          *(u32x2*)(a8 + 2 * A_HALF) = wx;
        }
      }
    };
    const sdm_rsrc rs8 = sdm_make_rsrc(p.w_dma, (unsigned int)((size_t)Cin * 9 * p.Cout_pad * 4));
    const unsigned int stage_rows = (unsigned int)p.Cout_pad * 16u;
    // weight step t of a tile with first output channel n0 -> ring slot t % 3: 24 pieces of 1 KB, six per producer wave
    auto dma_step = [&](int t, int n0) SDM_ALWAYS_INLINE {
      unsigned char* dst = Bring + (t % 3) * STEP;
      const unsigned int voff = (unsigned int)((n0 + lane) * 16);
#pragma unroll
      for (int q6 = 0; q6 < 6; ++q6) {
        const int q = wv * 6 + q6, ul = q / 12, qq = q % 12, pl = qq / 6, dy = (qq >> 1) % 3, ch = qq & 1;
        const unsigned int row = (unsigned int)(((t * 2 + ul) * 2 + pl) * 3 + dy);
        sdm_glds16_buf(rs8, voff + (unsigned int)(ch * 1024), row * stage_rows, dst + ul * SLOT + pl * PLANE + dy * (BN * 16) + ch * 1024);
      }
    };
    auto write_bias_tab = [&](int par, const TileXY& t) SDM_ALWAYS_INLINE {
      const float* bsrc = p.bias;
      if (bsrc && p.bias_sel) bsrc += (size_t)p.bias_sel[t.img] * p.Cout_pad;
      if (tid < BN) bias_tabs[par * BN + tid] = (bsrc && t.n0 + tid < p.Cout_pad) ? bsrc[t.n0 + tid] : 0.0f;
    };

    stage_setup(cur);
    if (first) {      // prologue of the block's first tile: its first weight steps and chunk 0
      write_bias_tab(0, cur);
      dma_step(0, cur.n0);
      dma_step(1, cur.n0);
      SDM_SCHED_FENCE();
      issue_loads(0);
      transform_vecs(Aring, 0, A_PER);
      SDM_WAIT_VMCNT0();
      SDM_WAIT_LGKMCNT0();
      SDM_RAW_BARRIER();
    }
    const DrainLane dl = drain_lane();
    // bench-only step stamps (tools/conv_trace.py, -DSDM_CONV_TRACE): wave 0 of the producing group parks (shader clock << 2 | code) in the 2 KB
    // of LDS behind the bias tables; copied to ConvParams::trace once, at kernel end
#if defined(SDM_CONV_TRACE) && !defined(SDM_EMU)
    auto stamp = [&](int code) SDM_ALWAYS_INLINE {
      if (p.trace && tr_n < 255) {
        const unsigned int tm = (unsigned int)__builtin_amdgcn_s_memtime();
        if (tid == 0) ((unsigned int*)(smem + C::SMEM))[grp * 256 + tr_n] = (tm << 2) | (unsigned int)code;
        ++tr_n;
      }
    };
#else
    auto stamp = [&](int code) SDM_ALWAYS_INLINE { (void)code; };
#endif
    f32x4 rq[MT];                        // residual of the NEXT slice's four quads, loaded one slice ahead (one slice = one step)
    // one chunk = six steps.  DR (a literal at every call site): 0 = no drain code at all - the accumulators are DEAD in this copy of the
    // steps, which is what lets the compiler give their 128 registers to the staging code -, 1 = chunk 0 of a draining producer (slices
    // 0-4 at steps 1-5), 2 = its chunk 1 (slices 5-7 at steps 1-3)
    auto chunk_steps = [&](const int DR, int c) SDM_ALWAYS_INLINE {
      const bool more = c + 1 < nch, staging = more || has_next;      // the chunk staged during this one: c + 1, or chunk 0 of the next tile
      unsigned char* Adst = Aring + ((c + 1) & 1) * A_BUF;
#pragma unroll
      for (int k6 = 0; k6 < 6; ++k6) {
        const int t = c * 6 + k6;
        const int u = (DR == 1) ? k6 - 1 : (DR == 2 && k6 <= 3 ? 4 + k6 : -1);                 // slice of this step
        const int un = (DR == 1 && k6 < 5) ? k6 : ((DR == 2 && k6 <= 2) ? 5 + k6 : -1);      // slice of the NEXT step
        const bool slice = DR != 0 && k6 >= 1 && u >= 0;
        if (slice) drain_valu(dl, u, rq);               // (its residual was loaded a step ago: no wait that could cover this step's DMAs)
        SDM_SCHED_FENCE();
        if (t + 2 < nsteps) dma_step(t + 2, cur.n0);
        else if (has_next) dma_step(t + 2 - nsteps, nxt.n0);
        SDM_SCHED_FENCE();
        const bool pre = DR != 0 && un >= 0 && p.res != nullptr;
        if (pre) drain_prefetch(dl, un, rq);
        if (slice) drain_store(dl, u);
        SDM_SCHED_FENCE();
        bool loads = false;
        if (k6 == 0 && staging) {
          if (!more) { stage_setup(nxt); write_bias_tab((k + 1) & 1, nxt); }
          issue_loads(more ? (c + 1) * 32 : 0);
          loads = true;
        }
        SDM_SCHED_FENCE();
        if (staging) {        // (2, 1, 2, 1 vectors in steps 2 .. 5: the loads of step 0 have landed by the end of step 1)
          if (k6 == 2) transform_vecs(Adst, 0, 2);
          if (k6 == 3) transform_vecs(Adst, 2, 3);
          if (k6 == 4) transform_vecs(Adst, 3, 5);
          if (k6 == 5) transform_vecs(Adst, 5, 6);
        }
        // everything older than what this step issued behind its DMAs has landed - the DMAs (two steps ahead of their use) in particular
        {
          const int fly = (loads ? (GN ? 16 : 12) : 0) + (pre ? 4 : 0) + (slice ? 4 : 0);
          switch (fly) {
            case 0: SDM_WAIT_VMCNT0(); break;
            case 4: SDM_WAIT_VMCNT(4); break;
            case 8: SDM_WAIT_VMCNT(8); break;
            case 12: SDM_WAIT_VMCNT(12); break;
            case 16: SDM_WAIT_VMCNT(16); break;
            case 20: SDM_WAIT_VMCNT(20); break;
            default: SDM_WAIT_VMCNT0(); break;
          }
        }
        SDM_WAIT_LGKMCNT0();
        stamp(DR);                // (traced builds) arrival at the step's barrier: code = which copy of the steps
        SDM_RAW_BARRIER();
        stamp(3);                 // released
      }
    };
    int c = 0;
    if (drain) {
      chunk_steps(1, 0);
      chunk_steps(2, 1);
      c = 2;
    }
    for (; c < nch; ++c) chunk_steps(0, c);
    zero_acc();      // this group multiplies the next tile (if any): its accumulators start at zero - and are dead from the last slice to here
  };

  // ==================================================================================================================
  // the block's tile sequence: group k & 1 consumes tile k, the other group produces it
  // ==================================================================================================================
  if (grp == 0) {
    zero_acc();
    SDM_RAW_BARRIER();      // (the other group's prologue)
  }
  for (int k = 0; k < ntiles_blk; ++k) {
    if (grp == (k & 1)) {
      consume_tile();
      et = tile_of(k); et_par = k & 1;
      if (k + 1 == ntiles_blk) {      // the last tile: nothing left to overlap the drain with
        const DrainLane dl = drain_lane();
        f32x4 rq[MT];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (p.res) drain_prefetch(dl, u, rq);
          drain_valu(dl, u, rq);
          drain_store(dl, u);
        }
      }
    } else {
      produce_tile(k, k == 0, k >= 1);
    }
  }
#if defined(SDM_CONV_TRACE) && !defined(SDM_EMU)
  if (p.trace && (int)blockIdx.x >= p.trace_b0 && (int)blockIdx.x < p.trace_b0 + 16 && ((int)threadIdx.x & (NT - 1)) == 0) {
    const unsigned int* tb = (const unsigned int*)(smem + C::SMEM) + grp * 256;
    unsigned int* dst = p.trace + ((size_t)((int)blockIdx.x - p.trace_b0) * 2 + grp) * 384;
    for (int i = 0; i < tr_n && i < 255; ++i) dst[i] = tb[i];
    dst[383] = (unsigned int)tr_n;
  }
#endif
}
