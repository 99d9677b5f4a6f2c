// k_conv.h - implicit-GEMM convolution / linear kernel for gfx950 (MFMA f16 -> f32).
//
// One kernel template covers every dense contraction of the SDMatte path:
//   * conv3x3 stride 1/2 of the VAE and U-Net (ResnetBlock2D.conv1/conv2, Down/Upsample2D.conv,
//     conv_in/conv_out ...; reference call sites meta_arch.py:142,209,216,256 and replace.py:462-544
//     reaching diffusers blocks).  A 3x3 conv is 9 shifted 1x1 GEMM taps
//     accumulated into the same MFMA accumulators from ONE LDS halo tile (no im2col in HBM).
//   * 1x1 conv_shortcut / quant_conv / post_quant_conv and every nn.Linear (proj_in/out, to_q/k/v,
//     to_out, GEGLU proj, ff.net.2): NTAPS = 1 over a flat [rows, Cin] activation.
//
// Layouts: activations NHWC (fp16, or fp32 for the residual stream); weights pre-packed by
// pack_conv_weight_kernel as fp16 [Cin_pad/16][NTAPS][Cout_pad][16] ("K16" layout, independent of the
// tile configuration chosen at run time) so that the B tile of one K-chunk is a few contiguous runs;
// fp32 accumulation.
// GEMM orientation: M = output pixels (A operand, from the LDS halo tile), N = output channels
// (B operand), K = KC input channels per tap.  v_mfma_f32_32x32x16_f16; wave tile (MT*32)x(NTL*32).
// LDS rows are padded to KC*2+16 bytes: with ds_read_b128 the 16-lane groups then hit 16 distinct
// 16-B slots of the 256-B bank row (conflict-free), see MI355X_MICROARCH.md LDS table.
//
// Fused on load : zero padding (symmetric pad 1, or the VAE's asymmetric (0,1,0,1) pad), nearest
//                 2x upsample (Upsample2D), two-pointer channel concat (U-Net skip `cat`), fp32->fp16.
// Fused on store: +bias (optionally per-image: the ResBlock time-embedding projection folded into
//                 conv1's bias), *scale (VAE scaling_factor), +residual, GEGLU u*gelu(g), fp16/fp32,
//                 channel-offset stores (writes straight into the 8-ch U-Net input tensor).
#pragma once
#include "sdm_common.h"

#ifndef SDM_CONV_VREUSE
#define SDM_CONV_VREUSE 1   /* vertical A-fragment reuse across taps (A/B switch for experiments) */
#endif

struct ConvParams {
  const void* in0; const void* in1;     // NHWC sources (channel concat: in0 then in1); in1 may be null
  int C0, C1;                           // channel counts (row strides) of the sources; (C0+C1) % KC == 0
  int in_f32;                           // sources are fp32 (1) or fp16 (0): must match the IN_F32 template argument
  int N, Hin, Win;                      // source dims (before the fused upsample)
  int up;                               // 1: nearest x2 upsample fused into the load
  int Hout, Wout;
  int pad_t, pad_l;
  long M;                               // NTAPS==1: number of rows (N*H*W)
  int rows_per_img;                     // NTAPS==1: != 0 -> row tiles are aligned to images of this many rows (needed when the
                                        // epilogue emits per-image GroupNorm statistics for a batch: one launch instead of N)
  const half_t* w;                      // packed weights, K16 layout
  const half_t* w_dma;                  // DMAB kernels: the 3x3 weights in stage order [Cin/16][dx][hi|lo][k-half][dy][Cout_pad][8] (see derive_conv_weight_dma_kernel)
  const half_t* w_lo;                   // SPLIT kernels: fp16 low parts of the (scaled) weights, same layout: w = (w_hi + w_lo) * acc_scale
  float acc_scale;                      // multiplies the accumulator in the epilogue (undoes the power-of-two weight pre-scale; 1 otherwise)
  size_t out_lo_off;                    // out_f32 == 2 / 3: element offset of the low-part plane behind the high-part plane
  int lo_cols;                          // out_f32 == 2 / 3: only output channels < lo_cols get the low-part plane (the V third of a fused q|k|v GEMM needs none)
  const float* bias;                    // [nvariants][Cout_pad]
  const int* bias_sel;                  // [N] variant per image or null
  int Cout_pad;                         // GEMM N (multiple of 32)
  void* out; int out_f32;               // 0: fp16, 1: fp32, 2: two fp16 planes hi | lo with hi + lo = the fp32 value (attention operands of the precise mode),
                                        // 3: fp16 plane hi + a plane of the same size holding, per 4 channels, [e5m2(x) x 4 | e5m2((x - hi) * 2^11) x 4]: the
                                        //    operands of the split-precision attention whose Q.K^T residual terms run on fp8 MFMAs (k_attn.h, PREC = 3)
  int Cout_store;                       // row stride (channels) of the output tensor
  int Cout_valid;                       // number of (post-epilogue) channels actually stored (mult of 4)
  int out_ch_off;                       // channel offset inside the output row (mult of 4)
  const void* res; int res_f32; int res_C;
  int epi;                              // 0: linear, 1: GEGLU (cols come in [u32|g32] groups of 64)
  float out_scale;
  const float* gn_scale;                // GN template arg: fused GroupNorm apply on load: x*scale[n][c] + shift[n][c] (+SiLU),
  const float* gn_shift;                // [N][C0+C1] fp32 each (from gn_finalize_*); the table of this image sits in LDS
  int gn_silu;
  int pc;                               // DMAB + SPLIT: 1 = producer / consumer form (8 waves, one block per CU), set by the launcher
  int f8;                               // 1: w_dma holds the fp8-residual layout (derive_conv_weight_f8_kernel) -> F8 kernel
  int f8_hint;                          // tile selection only: the layer has the fp8-residual weights (cfg 0 then beats the 256x64 tile)
  int f8_sa, f8_sb;                     // E8M0 exponents (byte 0) of the fp8 MFMA operand scales: 2^(sa-127) * 2^(sb-127) maps the residual sums to accumulator units
  unsigned int* trace;                  // bench only (sdm_bench_conv, ablate bit 256; -DSDM_CONV_TRACE builds): shader-clock stamps of consumer wave 0 and
                                        // producer wave 0 of 16 blocks of the F8 3x3 kernel, [block][role][384] (slot 383 = count); null in the engine
  int trace_skip, trace_b0;             // tiles of a block before the first traced one; first traced block
  int xtile;                            // F8 3x3 kernels: 1 = the producers run the next tile's prologue during the current tile's last chunk (cross-tile prefetch)
  int epi_mode;                         // F8 kernels (accumulators [channel][pixel]): 0 = LDS-staged epilogue; 3 = 0 + the residual enters as the accumulators'
                                        // initial value; 4 = 3 + full fp32 tiles are stored straight from the registers (16-byte stores, bias from LDS,
                                        // statistics by DPP row sums) - the default
  int ablate;                           // debug/bench only (sdm_bench_conv): 1 skip global loads, 2 skip LDS writes, 4 skip MFMA,
                                        // 8 skip the epilogue stores -- after the first K-chunk; always 0 in the engine
  int vgrid, tpb;                       // F8 (tpb tiles per block): number of virtual block ids = the grid of the one-tile-per-block forms
  int tiles_m, tiles_n, xcd_chunk;      // 1-D XCD-aware grid (set by launch_conv_t): M tiles per image (9 taps) or in total (1 tap), N tiles, M tiles per XCD
  int band, img_chunk;                  // optional M-tile order of the 3x3 kernels (band > 0): bands of `band` consecutive M tiles of every image go round-robin over the
                                        // 8 XCDs (XCD x owns img_chunk tiles of each image) instead of one contiguous 1/8 of the whole batch per XCD - used with
                                        // tile_flag, whose skipped tiles would otherwise leave the XCDs that own the trimap images idle
  const unsigned char* tile_flag;       // F8 3x3 kernels, optional [N][tiles_m]: != 0 -> every output pixel of that M tile lies in a region where the layer's input is
  const int* tile_rep;                  // constant (class id 1..7, k_misc.h cmask_conv_kernel) and the tile's result is the broadcast of one pixel of the class's
                                        // representative tile tile_rep[img * 8 + class]: such tiles are not computed here (const_tile_fill_kernel writes them)
  int ksplit; size_t ks_stride;         // split-K (register-staged kernels, launch_conv_t): ksplit > 1 -> grid.y = ksplit, block row y multiplies input channels
                                        // [y, y+1) * Cin / ksplit and stores its fp32 partial sums (acc * acc_scale: no bias / residual / statistics) at
                                        // (float*)out + y * ks_stride; splitk_reduce_kernel finishes the layer
  float* stats;                         // optional [N][tiles_m*WM][Cout_store][2]: per-(image, wave row-tile, channel) partial
                                        // sum / sum of squares of the stored values = fused GroupNorm statistics of the NEXT
                                        // layer (reduced by gn_partials_scale_shift_kernel); NTAPS==9 or one image per launch
};

template <int NTAPS, int STRIDE, int TH, int TW, int BN, int KC, int WM, int WN, int DB = 0, int SPLIT = 0, int DMAB = 0, int PC = 0, int F8 = 0>
struct ConvCfg {
  static constexpr int NTHREADS = 64 * WM * WN;                 // threads of one role (PC: consumers = producers = NTHREADS)
  static constexpr int LAUNCH_THREADS = NTHREADS * (PC ? 2 : 1);
  static constexpr int BM = TH * TW;
  static constexpr int WTM = BM / WM, WTN = BN / WN, MT = WTM / 32, NTL = WTN / 32;
  static constexpr int HPH = (NTAPS == 9) ? (TH - 1) * STRIDE + 3 : TH;
  static constexpr int HPW = (NTAPS == 9) ? (TW - 1) * STRIDE + 3 : TW;
  static constexpr int HP = HPH * HPW;
  // DB (double-buffered) tiles are stored UNPADDED with an XOR swizzle of the 16-B half (KC == 16: 2 halves per 32-B row):
  // half h of row r lives at r*32 + ((h ^ ((r >> 3) & 1)) << 4) -> conflict-free ds_read_b128 for 16-lane groups of consecutive rows
  // SWZ: the unpadded XOR-swizzled image of the double-buffered tiles.
  // PL ("half planes"): split-precision kernels with 16-channel chunks.  Their A_hi | A_lo | B tiles would not leave room for two
  // blocks per CU in the padded layout (87.9 KB vs 58.6 KB per block), so the two 16-byte k-halves of every row are stored in
  // two separate planes of 16-byte rows: plane h of row r at h * rows * 16 + r * 16.  A 16-lane ds_read_b128 group then reads 16
  // rows of one plane = 256 contiguous bytes (conflict-free, no padding) and every fragment address is base + constant.
  static constexpr int SWZ = DB ? 1 : 0;
  static constexpr int PL = ((SPLIT && KC == 16) || DMAB) ? 1 : 0;
  static constexpr int PITCH = (SWZ || PL) ? KC * 2 : KC * 2 + 16;     // bytes per row (both halves)
  static constexpr int ROWB = PL ? 16 : PITCH;                         // address stride between consecutive rows
  static constexpr int KV = KC / 8;
  static constexpr int A_BYTES = HP * PITCH;
  static constexpr int B_BYTES = NTAPS * BN * PITCH;
  static constexpr int STG_BYTES = WM * WN * 32 * WTN * 4;
  // DMAB (weights by LDS-DMA): two A tiles (fp16: double buffer; SPLIT: hi | lo) + a ring of 4 weight stages, one stage = the 3 taps
  // of one kernel column dx for BN output channels in half-plane layout (2 x 3 x BN rows of 16 B)
  // one weight unit: 2 planes x (3 taps dy | 1) x BN rows of 16 B.  F8 rings: 3x3 = 3 steps x 2 units, 1x1 = 3 steps (chunks) x 4 units
  // R4 (the F8 3x3 kernel): ONE A buffer (fp16 high planes | fp8 region) - the next chunk waits in the producers' registers and is written when the
  // region it replaces has been read for the last time - and a ring of FOUR weight steps, so that a step's DMAs have two steps to land
  static constexpr int R4 = (F8 && NTAPS == 9) ? 1 : 0;
  static constexpr int DMA_SLOT = (NTAPS == 9 ? 96 : 32) * BN, DMA_SLOTS = F8 ? (NTAPS == 9 ? 8 : 12) : (PC ? 5 : 4);      // PC: weights land TWO stages ahead of their use (fragment prefetch across the barrier)
  static constexpr int TILE_BYTES = DMAB ? (R4 ? 2 : PC ? 4 : 2) * A_BYTES + DMA_SLOTS * DMA_SLOT
                                         : (SPLIT ? 2 : 1) * A_BYTES + B_BYTES;   // one K-chunk of A halo (SPLIT: high and low parts) + B taps
  static constexpr int SMEM = ((DB ? 2 : 1) * TILE_BYTES) > STG_BYTES ? ((DB ? 2 : 1) * TILE_BYTES) : STG_BYTES;
  // F8 kernels: two bias tables (BN floats each) | [traced builds: parked stamps]
#ifdef SDM_CONV_TRACE
  static constexpr int TRACE_BYTES = 1792;      // 64 consumer + 384 producer stamps
#else
  static constexpr int TRACE_BYTES = 0;
#endif
  static constexpr int F8_EXTRA = 2 * BN * 4 + TRACE_BYTES;
  // LDS map of the F8 kernels.  1x1: A (two buffers) | ring | bias tables | stamps; the epilogue's scratch (statistics / staged stores) is the second A buffer.
  // 3x3 (R4): A | bias tables | ring of 4 steps | tail; the epilogue's scratch starts at ring step 3 - free from a tile's last barrier until the next
  // tile's first step (the prologue / cross-tile prefetch fills steps 0 - 2 only) - and runs into the tail
  static constexpr int BIAS_OFF = R4 ? 2 * A_BYTES : TILE_BYTES;
  static constexpr int RING_OFF = R4 ? 2 * A_BYTES + 2 * BN * 4 : (PC ? 4 : 2) * A_BYTES;
  static constexpr int SCR_OFF = R4 ? RING_OFF + 6 * DMA_SLOT : 2 * A_BYTES;
  static constexpr int TRACE_OFF = R4 ? SCR_OFF + STG_BYTES : TILE_BYTES + 2 * BN * 4;
  static constexpr int SMEM_F8 = R4 ? SCR_OFF + STG_BYTES + TRACE_BYTES : SMEM + F8_EXTRA;      // dynamic LDS of an F8 launch
  static_assert(!R4 || SMEM_F8 <= 160 * 1024, "LDS of the F8 3x3 kernel");
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile must be a multiple of 32x32");
  static_assert(KC % 16 == 0, "KC multiple of the MFMA K (16)");
  static_assert(!DB || KC == 16, "swizzled double-buffered tiles assume 2 halves per row");
  static_assert(!(DB && SPLIT), "the split-operand kernels use the single-buffered tile");
  static_assert(!DMAB || (((NTAPS == 9 && (KC == 16 || (F8 && KC == 32))) || (NTAPS == 1 && F8 && KC == 32)) && STRIDE == 1 && TW == 32 && BN == 128 && WM * WN == 4 && !DB),
                "DMA-weight pipeline: 256 x 128 tile only");
  static_assert(!F8 || (PC && KC == 32), "fp8-residual form: producer / consumer kernel on 32-channel chunks");
  static_assert(!PC || (DMAB && SPLIT), "producer/consumer form: split-precision DMA-weight kernel only");
};

// SPLIT = 1 ("precise" mode): every operand enters the MFMAs as a pair of fp16 values hi + lo (22 significant
// bits).  The fp32 activation (after the fused GroupNorm/SiLU, if any) is split while it is staged: hi = fp16(x),
// lo = fp16(x - hi) into two LDS halo tiles; the weights were split (and pre-scaled by a power of two so that the low parts stay
// fp16-normal) when they were packed: w * 2^k = w_hi + w_lo.  Per K-chunk the accumulators receive
//   A_hi.B_hi + A_lo.B_hi   (B tile = w_hi),   then   A_hi.B_lo   (B tile re-staged with w_lo);
// the lo.lo term (2^-22 relative) is dropped.  fp32 accumulation as before; the epilogue multiplies by acc_scale = 2^-k.
//
// DMAB = 1: the weights never pass through registers.  They are packed in "stage" order - one stage = the three taps (dy) of one
// kernel column dx for one 16-channel chunk (and, for SPLIT, one of the hi / lo parts) - and arrive by LDS-DMA (`buffer_load ... lds`)
// into a ring of 4 stages, three stages ahead of the one being multiplied; the MFMA loop sweeps one kernel column per stage
// (24 MFMAs per wave, vertical A-fragment reuse as before), ONE raw s_barrier per stage with a counted vmcnt, so DMAs stay in
// flight across barriers.  fp16-operand kernels double-buffer the activation halo tile in LDS as well (written while the previous
// chunk is still being multiplied); SPLIT kernels keep A_hi | A_lo single-buffered and re-stage them between two barriers per chunk.
//
// PC = 1 (producer / consumer waves; split-precision DMA-weight kernel): the block has 8 waves.  Waves 0-3 ("consumers", one per
// SIMD) execute nothing but LDS fragment reads and MFMAs - one such wave per SIMD keeps the matrix pipe as busy as two do
// (tools/probe/mfma_mix_probe.hip: 666 vs 688 TFLOP/s algorithmic) - and the epilogue; waves 4-7 ("producers") issue the weight
// DMAs and do the whole activation staging (global loads, GroupNorm / SiLU, hi | lo split, LDS writes) into a DOUBLE-buffered
// A_hi | A_lo tile, one chunk ahead.  Staging instructions therefore never sit in an MFMA wave's in-order instruction stream;
// the two roles meet at the one raw barrier per stage.  One block per CU (8 waves at <= 256 registers, 135 KB of LDS).
//
// F8 = 1 (PC kernel on 32-channel chunks): the two RESIDUAL terms of the split product, x_lo.w and x.w_lo, are evaluated on OCP
// fp8 (e4m3) operands by ONE v_mfma_scale_f32_32x32x64_f8f6f4 per tap and 32 channels (K = 64 = [x_lo8 | x8] . [w8 | w_lo8]; twice
// the fp16 rate and less energy per MAC - the kernel is power-limited, tools/probe/mfma_mix_probe.hip), x_hi.w_hi stays on fp16.
// A residual term is 2^-11 of the product, so a 2^-3 .. 2^-4 relative rounding leaves ~2^-14 .. 2^-15: the alpha error stays ~1e-4
// (tests/tools/exp_lowprec_residual_terms.py).  Operand formats follow what is known about the ranges: the WEIGHT side is e4m3 with
// a per-layer power-of-two scale chosen from max|w| when the layer is packed (derive_conv_weight_f8_kernel), the ACTIVATION side is
// e5m2 - the range of fp16 itself, so nothing saturates whatever a checkpoint's activations look like (x8 = e5m2(x), x_lo8 =
// e5m2(x_lo * 2^11), after one clamp of x to +-57344).  The producers write, per halo pixel and 32 channels, 4 x 16 B of fp16 high
// parts and 2 x 32 B of fp8.  The E8M0 operand scales of the MFMA bring both residual terms back to the unit of the fp16 accumulation.
template <int NTAPS, int STRIDE, int TH, int TW, int BN, int KC, int WM, int WN, int IN_F32, int DB, int GN, int SPLIT = 0, int DMAB = 0, int PC = 0, int F8 = 0>
// second argument = minimum waves per SIMD: the big 4-wave tiles keep 2 blocks/CU resident (2 waves/SIMD, <= 256 registers)
__global__ void __launch_bounds__(64 * WM * WN * (PC ? 2 : 1), (WM * WN == 4 && BN == 128) ? 2 : (WM * WN == 4 && BN == 64 && KC == 16 && TW == 32) ? 3 : 1)
conv_mfma_kernel(ConvParams p) {
  static_assert(!SPLIT || IN_F32, "split operands are produced from fp32 activations");
  using C = ConvCfg<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, DB, SPLIT, DMAB, PC, F8>;
  constexpr int NT = C::NTHREADS, MT = C::MT, NTL = C::NTL, PITCH = C::PITCH, KV = C::KV;
  constexpr int HPW = C::HPW, HP = C::HP, WTM = C::WTM, WTN = C::WTN;
  constexpr int SWZ = C::SWZ, PL = C::PL, ROWB = C::ROWB;
  constexpr int A_HALF = PL ? HP * 16 : 16, B_HALF = PL ? NTAPS * BN * 16 : 16;      // byte offset of the second k-half (channels 8..15)
  constexpr int A_VEC = HP * KV, B_VEC = NTAPS * BN * KV;
  constexpr int A_PER = (A_VEC + NT - 1) / NT, B_PER = (B_VEC + NT - 1) / NT;
  SDM_DYN_SMEM(smem);
  unsigned char* As = smem;                 // current K-chunk tile (switches between the two halves when DB)
  unsigned char* Bs = smem + (SPLIT ? 2 : 1) * C::A_BYTES;      // SPLIT: A_hi | A_lo | B

  // One tile = one call of run_tile.  F8 blocks run p.tpb tiles back to back: while the consumer waves
  // are in the epilogue of tile k, the producer waves already stage the first chunk and the first weight steps of tile k+1 (the
  // epilogue's LDS staging lives in the second A buffer, which that prologue does not touch); the ~16 us of fixed cost per tile
  // (launch, first loads, epilogue) is what separated the 4-chunk 128-channel layers from the 16-chunk ones.
  // F8 3x3, producer waves: cross-tile prefetch.  During a tile's LAST chunk the producers are idle (nothing left to stage), so they run
  // the NEXT tile's prologue there - its first weight steps into the ring slots that have just been consumed, its chunk 0 into the A
  // buffer that chunk nch-2 released, its chunk 1 loads into the registers below - and the next tile starts with its operands in
  // LDS: the consumers leave their epilogue straight into MFMAs, whose first steps cover the drain of the epilogue's stores (the
  // memory pipeline serves the CU's stores and loads in order; with the prologue's loads queued BEHIND the stores every tile paid
  // the store drain plus a full load latency).  pre_done: the tile about to start was prepared that way.
  int pre_done = 0;
#if defined(SDM_CONV_TRACE) && !defined(SDM_EMU)
  int tr_n = 0;
#endif
  // virtual block id -> (linear M tile = image * tiles_m + tile, N tile); false: padding block
  auto tile_decode = [&](int bid, int& mlin, int& nt) -> bool {
    const int j = bid >> 3, ml = j / p.tiles_n;
    nt = j - ml * p.tiles_n;
    if (NTAPS == 9 && p.band > 0) {
      const int im = ml / p.img_chunk, r = ml - im * p.img_chunk;
      const int mt = ((r / p.band) * 8 + (bid & 7)) * p.band + r % p.band;
      mlin = im * p.tiles_m + mt;
      return im < p.N && mt < p.tiles_m;
    }
    mlin = (bid & 7) * p.xcd_chunk + ml;
    return mlin < p.tiles_m * (((NTAPS == 9) || p.rows_per_img != 0) ? p.N : 1);
  };
  u32x4 a_nx[A_PER][IN_F32 ? 2 : 1];
  f32x4 gqn[4];
  u32x4 f8h[A_PER];                // F8 3x3 producers: [x_lo8 | x8] image of the chunk whose high planes were written last (crosses the tile boundary like a_nx)
  // role_c: std::integral_constant<int, R>.  R = 0 / 1: this copy of the tile body is executed by consumer / producer waves only (the F8
  // tile loop below branches on the role ONCE and instantiates the body per role, so that neither role's registers - the consumers'
  // accumulators and epilogue, the producers' staging sets and the chunk they carry across tiles - are live inside the other's code:
  // one body for both roles spilled 360-420 B / lane once a_nx had to survive the tile boundary); R = -1: role computed here.
  auto run_tile = [&](auto role_c, const int vbid, const int tile_par, const bool more_tiles) {
  constexpr int ROLE_C = decltype(role_c)::value;
  // PC: `tid` / `wave` are the index inside the role (consumers: MFMA tile position; producers: staging decomposition)
  int tx = (int)threadIdx.x;
  if (F8) SDM_OPAQUE_I(tx);          // per tile: nothing derived from the lane index is shared between the inlined tiles and kept live across an epilogue
  const int role = (ROLE_C >= 0) ? ROLE_C : (PC ? SDM_UNIFORM_I(tx / NT) : 0);                  // 0: consumer (or everything), 1: producer
  const int tid = PC ? (tx & (NT - 1)) : tx, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  // bench-only event stamps (ConvParams::trace; -DSDM_CONV_TRACE builds of tools/conv_trace.py only).  The stamps of consumer wave 0 /
  // producer wave 0 are PARKED IN LDS (behind the bias tables: 64 consumer + 384 producer stamps) and copied to global memory once, when the block's last
  // tile is done: a stamp is s_memtime + one ds_write_b32 of lane 0 - no vector-memory operation, so the counted vmcnt waits of the
  // producers and the in-order store / load queue of the CU see exactly what they see in the product build.  Bit 0 of a stamp = 1:
  // first stamp of a tile.  Per-step stamps (arrival at / release from every barrier of the chunk loop) are taken by the PRODUCER only:
  // the consumers' loop runs at the register limit (any stamp there spills), and a barrier is joint - the producer's wait at a barrier is
  // the time it was ahead of the consumers, no wait means the consumers were waiting for it.  Lab-only ablations (ConvParams::ablate) live under the same macro.
#if defined(SDM_CONV_TRACE) && !defined(SDM_EMU)
  const bool tr_on = (F8 && NTAPS == 9) && p.trace && ((vbid - (int)blockIdx.x) / (int)gridDim.x) >= p.trace_skip;
  auto stamp = [&](int first = 0) {
    if (tr_on && tr_n < (ROLE_C > 0 ? 383 : 63)) {      // wave-uniform: the count stays in an SGPR
      const unsigned int t = (unsigned int)__builtin_amdgcn_s_memtime();
      unsigned int* tr_buf = (unsigned int*)(smem + C::TRACE_OFF) + (ROLE_C > 0 ? 64 : 0);
      if (tid == 0) tr_buf[tr_n] = first ? (t | 1u) : (t & ~1u);
      ++tr_n;
    }
  };
#else
  auto stamp = [&](int first = 0) { (void)first; };
#endif
  // -DSDM_CONV_TRACE2 (with SDM_CONV_TRACE): four more stamps inside every producer step - after the LDS writes / register hand-over, after the DMA
  // issue, after the load issue, after the transform - tools/conv_trace.py --fine
#if defined(SDM_CONV_TRACE2) && defined(SDM_CONV_TRACE) && !defined(SDM_EMU)
  auto stamp2 = [&]() { stamp(); };
#else
  auto stamp2 = [&]() {};
#endif
  // -DSDM_CONV_LAB builds (tools/conv_lab.py) only - differential timing of the F8 3x3 kernel, results are garbage: ConvParams::ablate bit 0 =
  // activation loads from a 64-pixel window (cache-resident), 1 = no weight DMAs after a tile's first two steps, 2 = no MFMAs, 4 = no
  // operand transform / LDS writes by the producers, 5 = no activation loads after the prologue (3 = no epilogue stores)
#if defined(SDM_CONV_LAB) && !defined(SDM_EMU)
  const bool lab_win = (p.ablate & 1) != 0, lab_nodma = (p.ablate & 2) != 0, lab_nomm = (p.ablate & 4) != 0, lab_nowr = (p.ablate & 16) != 0, lab_nold = (p.ablate & 32) != 0;
#else
  constexpr bool lab_win = false, lab_nodma = false, lab_nomm = false, lab_nowr = false, lab_nold = false;
#endif
  (void)lab_win; (void)lab_nodma; (void)lab_nomm; (void)lab_nowr; (void)lab_nold;
  // XCD-aware 1-D grid.  The dispatcher places block id on XCD id % 8 (speed only, never needed for correctness).  XCD x owns
  // the contiguous M-tile range [x*chunk, (x+1)*chunk) and walks it in order, running the tiles_n output-channel tiles of
  // one M tile back to back: the A operand (halo tile / GEMM rows) is fetched from HBM once and re-read from that XCD's L2
  // by the other N tiles and by the spatial neighbours that share its halo (a (x, y, z) grid re-reads it tiles_n times from
  // HBM / Infinity Cache, on a different XCD each time).
  int mt, n0, img = 0, oy0 = 0, ox0 = 0;
  long m0 = 0;
  {
    int mlin, nt;
    const bool per_img = (NTAPS == 9) || p.rows_per_img != 0;
    if (!tile_decode(vbid, mlin, nt)) return;      // padding block of the last XCD range
    mt = mlin;
    if (per_img) { img = mlin / p.tiles_m; mt = mlin - img * p.tiles_m; }
    if (NTAPS == 9) {
      const int npx = (p.Wout + TW - 1) / TW;
      oy0 = (mt / npx) * TH;
      ox0 = (mt % npx) * TW;
    } else {
      m0 = (long)img * p.rows_per_img + (long)mt * C::BM;
    }
    n0 = nt * BN;
  }
  const long m_end = (NTAPS == 1 && p.rows_per_img) ? (long)(img + 1) * p.rows_per_img : p.M;      // first row beyond this block's row range
  const int Cin = p.C0 + p.C1;
  const int Hl = p.Hin << p.up, Wl = p.Win << p.up;
  const bool ksp = !DMAB && p.ksplit > 1;
  const int kb = ksp ? (int)blockIdx.y * (Cin / p.ksplit) : 0, ke = ksp ? kb + Cin / p.ksplit : Cin;      // this block's input-channel range

  f32x16 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // ---- F8 kernels: the accumulators are [channel][pixel].  The consumers pass the WEIGHT fragment as the MFMA's A operand and the
  //      activation fragment as B, so register r of lane l holds channel (r & 3) + 8 * (r >> 2) + 4 * (l >> 5) of pixel l & 31 of a
  //      32 x 32 sub-tile: every register quad is 4 consecutive output channels of one pixel = ONE 16-byte store / load per lane, no
  //      LDS transpose.  ConvParams::epi_mode 4 (full fp32 tiles): bias from a small LDS table, buffer_store_dwordx4 straight from
  //      the registers - the accumulators are free as soon as the stores have issued, and the stores drain underneath the next
  //      tile's MFMAs (tools/probe/store_overlap_probe.hip: 32 such stores per wave cost 5 us alone and nothing beside >= 576 MFMAs;
  //      the memory pipeline accepts one store instruction per ~65 cycles whatever its width, so the 128 dword stores of the
  //      natural [pixel][channel] layout were 4x slower than the LDS transpose they replaced, profiles/r03_conv_epilogue_ab.txt) -
  //      the residual enters as the accumulators' initial value (16-byte loads issued before the tile's first barrier), and the
  //      GroupNorm statistics of the consumer are in-lane sums reduced over the 32 pixel lanes with DPP adds.  Ragged tiles, fp16 /
  //      plane outputs and GEGLU go through the (swizzled) LDS staging below.  With 32-pixel-wide tiles the 32 pixels of sub-tile i
  //      are ONE image row, so every address is lane part (voffset, constant for the tile) + wave-uniform part (soffset). ----
  constexpr bool FASTEPI = (F8 != 0) && (NTAPS == 1 || (TW == 32 && WTM % TW == 0));
  const int wmu = SDM_UNIFORM_I(wm), wnu = SDM_UNIFORM_I(wn);
  (void)wnu;
  bool fast_epi = false, res_init = false;
  if (FASTEPI) {
    const bool full = (NTAPS == 9) ? (oy0 + TH <= p.Hout && ox0 + TW <= p.Wout) : (m0 + (long)C::BM <= m_end);
    const bool ok = p.out_f32 == 1 && p.epi == 0 && full && (!p.res || p.res_f32);
    res_init = ok && p.epi_mode >= 3 && p.res != nullptr && p.out_scale == 1.0f;      // mode 3: residual as accumulator init + the LDS-staged store epilogue
    // the register-direct epilogue stores acc + bias as it is: no output scale, every channel of the tile valid, the residual (if any) already in
    // the accumulators; anything else takes the LDS-staged epilogue
    fast_epi = ok && p.epi_mode == 4 && p.out_scale == 1.0f && (!p.res || res_init) && (n0 + BN <= p.Cout_valid);
  }
  // first output pixel of this wave's sub-tile i, relative to the tile's first pixel (px_tile0)
  const size_t px_tile0 = (NTAPS == 9) ? ((size_t)img * p.Hout + oy0) * p.Wout : (size_t)m0;
  auto sub_px = [&](int i) { return (NTAPS == 9) ? (unsigned int)((wmu * (WTM / TW) + i) * p.Wout + ox0) : (unsigned int)(wmu * WTM + i * 32); };
  const int ch_lane = n0 + wn * WTN + 4 * (lane >> 5);        // first output channel of this lane's register quad (j, g) = ch_lane + j * 32 + 8 * g
  // bias of the tile's BN output channels: written by the producer waves in their prologue (two tables: the consumers may still read
  // the previous tile's while the next prologue runs), read by the accumulator-layout epilogue
  float* bias_tab = (float*)(smem + C::BIAS_OFF) + tile_par * BN;
  auto acc_init_residual = [&]() {
  if (FASTEPI && res_init) {
    // the residual is the accumulators' initial value: out = (acc + bias) + res.  F8 layers carry acc_scale == 1 (their fp16 high parts
    // are packed unscaled, the fp8 operand scales absorb the rest), so the loaded quads ARE accumulator registers: any arithmetic
    // on them made hipcc stage all 128 values in a second register set (1.2 KB / lane of scratch).  Called by the CONSUMER branch only,
    // in front of the barrier that ends the producers' prologue: the accumulators must not be live across the producers' code (the
    // register allocation is the union of both roles), and the loads fly while the consumers wait for the first operands.
    const unsigned int rs4 = (unsigned int)p.res_C * 4u;
    const unsigned int vr = (unsigned int)(lane & 31) * rs4 + (unsigned int)ch_lane * 4u;
#ifdef SDM_EMU
    const sdm_rsrc rsi = sdm_make_rsrc((const unsigned char*)p.res + px_tile0 * rs4, (unsigned int)(((NTAPS == 9) ? (size_t)TH * p.Wout : (size_t)C::BM) * rs4));
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const unsigned int so = sub_px(i) * rs4;
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bool ok = (ch_lane + j * 32 + 8 * g) < p.Cout_valid;
          const f32x4 q = __builtin_bit_cast(f32x4, sdm_buffer_load16(rsi, ok ? vr + (unsigned int)((j * 32 + 8 * g) * 4) : SDM_BUF_INVALID, so));
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = q[e];
        }
    }
#else
    // The loads are issued through inline asm and waited for HERE, explicitly.  As compiler-visible loads they left "a load into
    // the accumulators may be pending" in hipcc's wait-count bookkeeping along the (impossible) path that skips the chunk loop, and it
    // guarded every quad of the epilogue with s_waitcnt vmcnt(31 - q): every consumer wave waited for the COMPLETION of its own stores
    // inside the epilogue.  Measured (profiles/r04_conv_f8_step_trace.txt, r04_conv_f8_epilogue_ab.txt): removing those waits does not
    // shorten the epilogue - a wave's buffer_store_dwordx4 issues at the CU's store rate (one per ~65 cycles while all four consumer waves
    // store: 8 k cycles per 128 KB tile) whether or not anything waits for it - but it removes a false dependence, and the wait costs
    // nothing here: the next thing a consumer does is the barrier behind which the producers already wait, then MFMAs that need the values.
    static_assert(!FASTEPI || MT * NTL * 4 == 32, "operand lists of the wait statements below");
    const sdm_rsrc_raw rqi = sdm_make_rsrc_raw((const unsigned char*)p.res + px_tile0 * rs4, (unsigned int)(((NTAPS == 9) ? (size_t)TH * p.Wout : (size_t)C::BM) * rs4));
    f32x4 rq[MT][NTL][4];
    asm volatile("s_nop 4" ::: "memory");          // the descriptor SGPRs were written by v_readfirstlane just before
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const unsigned int so = (unsigned int)__builtin_amdgcn_readfirstlane((int)(sub_px(i) * rs4));
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const bool ok = (ch_lane + j * 32 + 8 * g) < p.Cout_valid;
          const unsigned int vo = ok ? vr + (unsigned int)((j * 32 + 8 * g) * 4) : SDM_BUF_INVALID;
          asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rq[i][j][g]) : "v"(vo), "s"(rqi), "s"(so) : "memory");
        }
    }
#define SDM_RQ4(i, j) "+v"(rq[i][j][0]), "+v"(rq[i][j][1]), "+v"(rq[i][j][2]), "+v"(rq[i][j][3])
    asm volatile("s_waitcnt vmcnt(0)" : SDM_RQ4(0, 0), SDM_RQ4(0, 1), SDM_RQ4(1, 0), SDM_RQ4(1, 1) :: "memory");
    asm volatile("" : SDM_RQ4(2, 0), SDM_RQ4(2, 1), SDM_RQ4(3, 0), SDM_RQ4(3, 1) :: "memory");
#undef SDM_RQ4
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * g + e] = rq[i][j][g][e];
#endif
  }
  };

  // fused GroupNorm apply: scale[Cin] | shift[Cin] of this image, behind the A/B tile region
  float* gn_tab = (float*)(smem + (DB ? 2 : 1) * C::TILE_BYTES);
  if (GN && !F8) {               // F8: no room in LDS - the producers read the 8 scale / shift values of their channels per chunk
    for (int c = tid; c < Cin; c += NT) {
      gn_tab[c] = p.gn_scale[(size_t)img * Cin + c];
      gn_tab[Cin + c] = p.gn_shift[(size_t)img * Cin + c];
    }
    // visibility: the first write_lds happens after the __syncthreads() at the top of the K loop
  }

  // per-lane fragment bases.  Padded layout: byte offsets.  Swizzled (DB) layout: A keeps the halo ROW (the swizzle bit
  // depends on the row, which moves with the tap), B the final byte offset (its swizzle bit only depends on co).
  int abase[MT], bbase[NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = wm * WTM + i * 32 + (lane & 31);
    const int py = m / TW, px = m % TW;
    const int row = (NTAPS == 9) ? (py * STRIDE * HPW + px * STRIDE) : m;
    abase[i] = SWZ ? row : row * ROWB + (lane >> 5) * A_HALF;
  }
#pragma unroll
  for (int j = 0; j < NTL; ++j) {
    const int co = wn * WTN + j * 32 + (lane & 31);
    bbase[j] = SWZ ? co * PITCH + ((((lane >> 5) ^ (co >> 3)) & 1) << 4) : co * ROWB + (lane >> 5) * B_HALF;
  }

  // ---- K-loop invariant staging descriptors.  Vector v = tid + i*NT; everything except the A pixel index is affine in
  //      the (compile-time) unrolled index i.  All global reads are raw BUFFER loads: per-block SGPR descriptors, one
  //      32-bit byte offset per load, hardware zero-fill for anything out of range (halo, ragged rows, padded channels). ----
  static_assert(NT % KV == 0 && NT % (2 * BN) == 0, "staging decomposition: one thread keeps one (co, half) for every i");
  const unsigned int es = IN_F32 ? 4u : 2u;
  // 3x3: the descriptors below start at the first input row this tile's halo can touch and span only the rows it needs, so
  // the 32-bit byte offsets stay small whatever the size of the image (one fp32 256-channel image at 2048^2 is 4.3 GB)
  const int band0 = (NTAPS == 9) ? ((oy0 * STRIDE - p.pad_t) > 0 ? ((oy0 * STRIDE - p.pad_t) >> p.up) : 0) : 0;
  const int band_rows = (NTAPS == 9) ? ((p.Hin - band0) < (C::HPH + 1) ? (p.Hin - band0) : (C::HPH + 1)) : 0;
  const int a_part = (tid % KV) * 8;                 // channel offset inside the chunk (same for every i)
  const int a_hp0 = tid / KV;                        // halo pixel of vector i: a_hp0 + i*(NT/KV)
  int a_pix[A_PER];                                  // pixel index inside the image / row inside the tile; -1: zero fill
#pragma unroll
  for (int i = 0; i < A_PER; ++i) {
    const int hp = a_hp0 + i * (NT / KV);
    a_pix[i] = -1;
    if (tid + i * NT < A_VEC) {
      if (NTAPS == 9) {
        const int hy = hp / HPW, hx = hp % HPW;
        const int iy = oy0 * STRIDE + hy - p.pad_t, ix = ox0 * STRIDE + hx - p.pad_l;
        if (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) a_pix[i] = ((iy >> p.up) - band0) * p.Win + (ix >> p.up);
      } else {
        a_pix[i] = hp;                                // rows beyond M fall outside the descriptor -> 0
      }
    }
  }
  sdm_rsrc rs0, rs1, rsw;
  {
    size_t base_px, npx;
    if (NTAPS == 9) { base_px = ((size_t)img * p.Hin + band0) * p.Win; npx = (size_t)band_rows * p.Win; }
    else { base_px = (size_t)m0; npx = (size_t)((m_end - m0) < (long)C::BM ? (m_end - m0) : (long)C::BM); }
    rs0 = sdm_make_rsrc((const unsigned char*)p.in0 + base_px * p.C0 * es, (unsigned int)(npx * p.C0 * es));
    rs1 = sdm_make_rsrc(p.in1 ? (const unsigned char*)p.in1 + base_px * p.C1 * es : (const unsigned char*)p.in0, p.in1 ? (unsigned int)(npx * p.C1 * es) : 0u);
    rsw = sdm_make_rsrc(p.w, (unsigned int)((size_t)Cin * NTAPS * p.Cout_pad * 2));
  }
  const sdm_rsrc rsw_lo = SPLIT ? sdm_make_rsrc(p.w_lo, (unsigned int)((size_t)Cin * NTAPS * p.Cout_pad * 2)) : rsw;
  // B vector v -> (h = v&1, co = (v>>1)%BN, rest = (v>>1)/BN -> tap = rest%NTAPS, sc = rest/NTAPS) of the K16-packed weights
  const int b_h = tid & 1, b_co = (tid >> 1) % BN, b_rest0 = (tid >> 1) / BN;
  constexpr int B_RSTEP_NUM = NT / 2;                // (v>>1) advances by NT/2 per i
  const unsigned int b_voff = ((n0 + b_co) < p.Cout_pad) ? (unsigned int)(((n0 + b_co) * 16 + b_h * 8) * 2) : SDM_BUF_INVALID;
  const unsigned int b_row_bytes = (unsigned int)p.Cout_pad * 32u;          // one (chunk16, tap) row of the packed tensor

  // raw staging registers (the global loads of chunk k+1 are in flight while chunk k is multiplied)
  u32x4 a_raw[A_PER][IN_F32 ? 2 : 1];
  u32x4 b_raw[B_PER];
  auto issue_loads_a = [&](int c0) {
    // every load is unconditional and independent: they are all in flight at once (a per-element `if (ok) load` makes hipcc
    // branch around each load and drain vmcnt(0) per element - measured 4.9 us per K-chunk)
    const bool second = c0 >= p.C0;
    const sdm_rsrc rs = second ? rs1 : rs0;
    const unsigned int Cs = (unsigned int)(second ? p.C1 : p.C0) * es;
    const unsigned int cc = (unsigned int)((second ? c0 - p.C0 : c0) + a_part) * es;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const unsigned int off = a_pix[i] >= 0 ? (unsigned int)a_pix[i] * Cs + cc : SDM_BUF_INVALID;
      a_raw[i][0] = sdm_buffer_load16(rs, off, 0);
      if (IN_F32) a_raw[i][IN_F32 ? 1 : 0] = sdm_buffer_load16(rs, off, 16);
    }
  };
  // DMAB: the same loads, invisible to the compiler's vmcnt bookkeeping (it would drain the weight DMAs at every use)
  const sdm_rsrc_raw rq0 = sdm_make_rsrc_raw((const unsigned char*)p.in0 + ((NTAPS == 9) ? ((size_t)img * p.Hin + band0) * p.Win : (size_t)m0) * p.C0 * es,
                                             (unsigned int)((size_t)band_rows * p.Win * p.C0 * es));
  const sdm_rsrc_raw rq1 = sdm_make_rsrc_raw(p.in1 ? (const unsigned char*)p.in1 + ((size_t)img * p.Hin + band0) * p.Win * p.C1 * es : (const unsigned char*)p.in0,
                                             p.in1 ? (unsigned int)((size_t)band_rows * p.Win * p.C1 * es) : 0u);
  auto issue_loads_a_asm = [&](int c0) {
    const bool second = c0 >= p.C0;
    const sdm_rsrc_raw rq = second ? rq1 : rq0;
    const unsigned int Cs = (unsigned int)(second ? p.C1 : p.C0) * es;
    const unsigned int cc = (unsigned int)((second ? c0 - p.C0 : c0) + a_part) * es;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      const unsigned int off = a_pix[i] >= 0 ? (unsigned int)a_pix[i] * Cs + cc : SDM_BUF_INVALID;
      if (i == 0) SDM_ASM_BUFFER_LOAD16_FIRST(a_raw[i][0], off, rq, 0);
      else SDM_ASM_BUFFER_LOAD16(a_raw[i][0], off, rq, 0);
      if (IN_F32) SDM_ASM_BUFFER_LOAD16(a_raw[i][IN_F32 ? 1 : 0], off, rq, 16);
    }
  };
  // the loads above have landed: stage_end() of the two stages in between waited for everything older than the last two
  // weight DMAs.  The (free) counted wait names every destination register so that no consumer is scheduled above it.
  auto a_loads_landed = [&]() {
#ifndef SDM_EMU
    static_assert(!DMAB || F8 || A_PER <= 3, "register list of the wait statement");
    if (IN_F32) {
      if (A_PER == 3) asm volatile("s_waitcnt vmcnt(9)" : "+v"(a_raw[0][0]), "+v"(a_raw[0][IN_F32 ? 1 : 0]), "+v"(a_raw[A_PER > 1 ? 1 : 0][0]),
                                   "+v"(a_raw[A_PER > 1 ? 1 : 0][IN_F32 ? 1 : 0]), "+v"(a_raw[A_PER > 2 ? 2 : 0][0]), "+v"(a_raw[A_PER > 2 ? 2 : 0][IN_F32 ? 1 : 0]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(9)" : "+v"(a_raw[0][0]), "+v"(a_raw[0][IN_F32 ? 1 : 0]), "+v"(a_raw[A_PER > 1 ? 1 : 0][0]),
                        "+v"(a_raw[A_PER > 1 ? 1 : 0][IN_F32 ? 1 : 0]) :: "memory");
    } else {
      if (A_PER == 3) asm volatile("s_waitcnt vmcnt(9)" : "+v"(a_raw[0][0]), "+v"(a_raw[A_PER > 1 ? 1 : 0][0]), "+v"(a_raw[A_PER > 2 ? 2 : 0][0]) :: "memory");
      else asm volatile("s_waitcnt vmcnt(9)" : "+v"(a_raw[0][0]), "+v"(a_raw[A_PER > 1 ? 1 : 0][0]) :: "memory");
    }
#endif
  };
  auto issue_loads_b = [&](int c0, const sdm_rsrc rsb) {
    const unsigned int chunk_row0 = (unsigned int)(c0 / 16) * NTAPS;
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      const int lin = (tid >> 1) + i * B_RSTEP_NUM;          // = (v >> 1)
      const int rest = (2 * BN >= NT) ? (lin / BN) : (b_rest0 + i * (NT / (2 * BN)));
      const int tap = rest % NTAPS, sc = rest / NTAPS;
      const unsigned int voff = ((tid + i * NT) < B_VEC) ? b_voff : SDM_BUF_INVALID;
      b_raw[i] = sdm_buffer_load16(rsb, voff, (chunk_row0 + (unsigned int)(sc * NTAPS + tap)) * b_row_bytes);
    }
  };
  auto issue_loads = [&](int c0) { issue_loads_a(c0); issue_loads_b(c0, rsw); };
  auto write_lds_a = [&](unsigned char* Ad, int c0w) {
    f32x4 gs0, gs1, gh0, gh1;
    if (GN) {                      // the 8 channels of this thread are the same for every i
      const float* tb = gn_tab + c0w + a_part;
      gs0 = *(const f32x4*)tb; gs1 = *(const f32x4*)(tb + 4);
      gh0 = *(const f32x4*)(tb + Cin); gh1 = *(const f32x4*)(tb + Cin + 4);
    }
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
      if (tid + i * NT < A_VEC) {
        f16x8 val;
        if (SPLIT) {
          // fp32 value (after the fused GroupNorm / SiLU) -> hi = fp16(y), lo = fp16(y - hi): both halo tiles written here
          float x[8];
          const f32x4 lo4 = __builtin_bit_cast(f32x4, a_raw[i][0]), hi4 = __builtin_bit_cast(f32x4, a_raw[i][IN_F32 ? 1 : 0]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { x[e] = lo4[e]; x[4 + e] = hi4[e]; }
          const bool inside = a_pix[i] >= 0;
          f16x8 vlo;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float y = x[e];
            if (GN) {
              y = y * (e < 4 ? gs0[e & 3] : gs1[e & 3]) + (e < 4 ? gh0[e & 3] : gh1[e & 3]);
              if (p.gn_silu) y = y * sdm_rcp(1.0f + sdm_exp2(-y * SDM_LOG2E));
              if (!inside) y = 0.0f;                     // zero padding stays zero AFTER the normalisation
            }
            const half_t h = (half_t)y;
            val[e] = h;
            vlo[e] = (half_t)(y - (float)h);
          }
          const int hp_s = a_hp0 + i * (NT / KV);
          *(f16x8*)(Ad + C::A_BYTES + hp_s * ROWB + (PL ? (a_part >> 3) * A_HALF : (a_part * 2))) = vlo;
        } else if (GN) {
          float x[8];
          if (IN_F32) {
            const f32x4 lo = __builtin_bit_cast(f32x4, a_raw[i][0]), hi4 = __builtin_bit_cast(f32x4, a_raw[i][IN_F32 ? 1 : 0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { x[e] = lo[e]; x[4 + e] = hi4[e]; }
          } else {
            const f16x8 h8 = __builtin_bit_cast(f16x8, a_raw[i][0]);
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)h8[e];
          }
          const bool inside = a_pix[i] >= 0;           // zero padding stays zero AFTER the normalisation
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float y = x[e] * (e < 4 ? gs0[e & 3] : gs1[e & 3]) + (e < 4 ? gh0[e & 3] : gh1[e & 3]);
            if (p.gn_silu) y = y * sdm_rcp(1.0f + sdm_exp2(-y * SDM_LOG2E));
            val[e] = inside ? (half_t)y : (half_t)0.0f;
          }
        } else if (IN_F32) {
          const f32x4 lo = __builtin_bit_cast(f32x4, a_raw[i][0]), hi4 = __builtin_bit_cast(f32x4, a_raw[i][IN_F32 ? 1 : 0]);
#pragma unroll
          for (int e = 0; e < 4; ++e) { val[e] = (half_t)lo[e]; val[4 + e] = (half_t)hi4[e]; }
        } else {
          val = __builtin_bit_cast(f16x8, a_raw[i][0]);
        }
        const int hp_w = a_hp0 + i * (NT / KV);
        *(f16x8*)(Ad + hp_w * ROWB + (SWZ ? ((((a_part >> 3) ^ (hp_w >> 3)) & 1) << 4) : PL ? (a_part >> 3) * A_HALF : (a_part * 2))) = val;
      }
    }
  };
  auto write_lds_b = [&](unsigned char* Bd) {
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      if ((tid + i * NT) < B_VEC) {
        const int lin = (tid >> 1) + i * B_RSTEP_NUM;
        const int rest = (2 * BN >= NT) ? (lin / BN) : (b_rest0 + i * (NT / (2 * BN)));
        const int co = (2 * BN >= NT) ? (lin % BN) : b_co;
        const int tap = rest % NTAPS, sc = rest / NTAPS;
        *(f16x8*)(Bd + (tap * BN + co) * ROWB + (SWZ ? (((b_h ^ (co >> 3)) & 1) << 4) : PL ? b_h * B_HALF : ((sc * 2 + b_h) * 16))) = __builtin_bit_cast(f16x8, b_raw[i]);
      }
    }
  };
  auto write_lds = [&](unsigned char* Ad, unsigned char* Bd, int c0w) { write_lds_a(Ad, c0w); write_lds_b(Bd); };

  auto mfma_chunk = [&](const unsigned char* Ap) {
    // ---- MFMA over taps and K sub-steps ----
    // Software-pipelined fragment reads with a pinned schedule: B fragments of step s+1 are read at the top of step s
    // (double buffer, 2*NTL regs x4), each A fragment is re-read IN PLACE for step s+1 right after the MFMAs that consumed it.
    // Every ds_read therefore has ~3/4 of a step (6 MFMAs = 192 cycles) of cover instead of none.
    constexpr int NSTEP = NTAPS * (KC / 16);
    f16x8 fa[MT], fb[2][NTL];
    auto a_addr = [&](int step, int i) {
      const int tap = step / (KC / 16), ks = step % (KC / 16);
      if (SWZ) {
        const int row = abase[i] + ((NTAPS == 9) ? (tap / 3) * HPW + (tap % 3) : 0);
        return (const f16x8*)(Ap + row * PITCH + ((((lane >> 5) ^ (row >> 3)) & 1) << 4));
      }
      const int toff = (NTAPS == 9) ? ((tap / 3) * HPW + (tap % 3)) * ROWB : 0;
      return (const f16x8*)(Ap + abase[i] + toff + ks * 32);
    };
    auto b_addr = [&](int step, int j) {
      const int tap = step / (KC / 16), ks = step % (KC / 16);
      return (const f16x8*)(Bs + tap * BN * ROWB + bbase[j] + ks * 32);
    };
    // Vertical operand reuse (3x3, stride 1, 32-pixel-wide tiles): the A fragment of output row i at tap (dy,dx) is halo row
    // i+dy shifted by dx - the SAME LDS data for every (i,dy) with equal i+dy.  Sweeping halo rows r = 0..MT+1 per dx reads
    // 3*(MT+2) A fragments per K16 step instead of 9*MT (18 vs 36 at MT=4); the three B fragments (dy) of the current dx
    // stay in registers and are re-read in place for dx+1 right after their last use.  LDS read traffic per MFMA drops by 1/3
    // (the tile is otherwise LDS-bandwidth-bound: 54 ds_read_b128 per 72 MFMAs ~ 0.9 of the CU's LDS cycles).
    constexpr bool VREUSE = SDM_CONV_VREUSE && (NTAPS == 9) && (STRIDE == 1) && (TW == 32) && (MT >= 2) && !DB;
    if (VREUSE) {
      constexpr int NR = MT + 2, NS = 3 * NR;
#pragma unroll
      for (int ks = 0; ks < KC / 16; ++ks) {
        f16x8 fav[2], fbv[3][NTL];
        auto av = [&](int s) { return *(const f16x8*)(Ap + abase[0] + ((s % NR) * HPW + (s / NR)) * ROWB + ks * 32); };
        auto bv = [&](int dy, int dx, int j) { return *(const f16x8*)(Bs + (dy * 3 + dx) * BN * ROWB + bbase[j] + ks * 32); };
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int j = 0; j < NTL; ++j) fbv[dy][j] = bv(dy, 0, j);
        fav[0] = av(0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
          const int dx = s / NR, r = s % NR;
          if (s + 1 < NS) fav[(s + 1) & 1] = av(s + 1);
          SDM_SCHED_FENCE();
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int i = r - dy;
            if (i >= 0 && i < MT) {
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(fav[s & 1], fbv[dy][j], acc[i][j]);
            }
          }
          SDM_SCHED_FENCE();
          const int dyl = r - (MT - 1);          // the B fragments of this dy were used for the last time in this dx phase
          if (dyl >= 0 && dyl < 3 && dx + 1 < 3) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) fbv[dyl][j] = bv(dyl, dx + 1, j);
          }
          SDM_SCHED_FENCE();
        }
      }
    } else {
#pragma unroll
    for (int j = 0; j < NTL; ++j) fb[0][j] = *b_addr(0, j);
#pragma unroll
    for (int i = 0; i < MT; ++i) fa[i] = *a_addr(0, i);
#pragma unroll
    for (int step = 0; step < NSTEP; ++step) {
      if (step + 1 < NSTEP) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) fb[(step + 1) & 1][j] = *b_addr(step + 1, j);
      }
      SDM_SCHED_FENCE();
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(fa[i], fb[step & 1][j], acc[i][j]);
        SDM_SCHED_FENCE();
        if (step + 1 < NSTEP) fa[i] = *a_addr(step + 1, i);
        SDM_SCHED_FENCE();
      }
    }
    }
  };

  if (DMAB) {
    constexpr int SLOT = C::DMA_SLOT, NPR = SPLIT ? 2 : 1, PLANE = 3 * BN * 16;      // PLANE: one k-half plane of a stage (3 dy x BN rows)
    unsigned char* Aring = smem;
    unsigned char* Bring = smem + C::RING_OFF;
    const int nchunks = Cin / 16, nst = nchunks * 3 * NPR;
    const sdm_rsrc rsd = sdm_make_rsrc(p.w_dma, (unsigned int)((size_t)Cin * 9 * NPR * p.Cout_pad * 2));
    const int wv = SDM_UNIFORM_I(wave);
    const unsigned int dma_voff = (unsigned int)((n0 + lane) * 16);
    const unsigned int stage_rows = (unsigned int)p.Cout_pad * 16u;                  // bytes of one (k-half, dy) row group
    // 12 pieces of 1 KB per stage: piece q = (k-half, dy, 64-channel half); this wave issues pieces 3*wave .. 3*wave+2
    auto dma_stage_sl = [&](int s, int sl) {
      unsigned char* dst = Bring + sl * SLOT;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int q = wv * 3 + k, half = q / 6, dy = (q >> 1) % 3, ch = q & 1;
        const unsigned int row = (unsigned int)((s * 2 + half) * 3 + dy);          // s enumerates (chunk, dx, part) in memory order
        sdm_glds16_buf(rsd, dma_voff + (unsigned int)(ch * 1024), row * stage_rows, dst + half * PLANE + dy * (BN * 16) + ch * 1024);
      }
    };
    auto dma_stage = [&](int s) { dma_stage_sl(s, s & 3); };
    int bq[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) bq[j] = (lane >> 5) * PLANE + (wn * WTN + j * 32 + (lane & 31)) * 16;
    // one kernel column: halo rows r = 0..MT+1 of column dx against the three taps (dy, dx); output row i = r - dy
    auto sweep = [&](const unsigned char* Ap, const unsigned char* Bp, int dx) {
      constexpr int NR = MT + 2;
      f16x8 fav[2], fbv[3][NTL];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int j = 0; j < NTL; ++j) fbv[dy][j] = *(const f16x8*)(Bp + bq[j] + dy * (BN * 16));
      fav[0] = *(const f16x8*)(Ap + abase[0] + dx * 16);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        if (r + 1 < NR) fav[(r + 1) & 1] = *(const f16x8*)(Ap + abase[0] + ((r + 1) * HPW + dx) * 16);
        SDM_SCHED_FENCE();
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int i = r - dy;
          if (i >= 0 && i < MT) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(fav[r & 1], fbv[dy][j], acc[i][j]);
          }
        }
        SDM_SCHED_FENCE();
      }
    };
    // end of a stage: the NEXT stage's weights (this wave's pieces) have landed, LDS writes of this wave are done, then the barrier.
    // vmcnt is counted: every stage issued after stage s+1 adds 3 DMAs per wave, and only those may still be in flight
    const bool ab_ald = (p.ablate & 32) != 0, ab_awr = (p.ablate & 64) != 0;      // bench only: skip only the A loads / only the A writes
    const bool ab_a = (p.ablate & 1) != 0, ab_dma = (p.ablate & 2) != 0, ab_mm = (p.ablate & 4) != 0, ab_bar = (p.ablate & 16) != 0;   // bench only
    // a_fly: the activation loads of the next chunk were issued after the DMAs of stage s+1 and are not needed yet - they may stay
    // in flight too (3 loads per thread, 6 for fp32 activations)
    auto stage_end = [&](int s, bool a_fly) {
      if (ab_bar) return;
      if (s + 3 < nst) {
        if (a_fly) { if (IN_F32) SDM_WAIT_VMCNT(12); else SDM_WAIT_VMCNT(9); }
        else SDM_WAIT_VMCNT(6);
      }
      else if (s + 2 < nst) SDM_WAIT_VMCNT(3);
      else SDM_WAIT_VMCNT0();
      SDM_WAIT_LGKMCNT0();
      SDM_RAW_BARRIER();
    };
    if (F8) {
      // ---- fp8-residual producer / consumer form.  One STEP = two 12 KB weight units: S1 = w_hi of channels 0-15 | 16-31 (two
      // fp16 K16 sweeps, 48 MFMAs), S2 = w8 | w_lo8 (one K64 fp8 sweep, 24 MFMAs of twice the duration): equal step lengths, two
      // barriers per kernel column.  Ring of 3 steps; the DMAs of step t+2 are issued at the start of step t and have landed when
      // the barrier of step t releases, so the B fragments of step t+1 are read while step t is multiplied. ----
      constexpr int NR = MT + 2, STEP = 2 * SLOT;
      constexpr float F8_LS = 2048.0f, F8_AMAX = 57344.0f;            // x8 = e5m2(x), x_lo8 = e5m2(x_lo * 2^11); |x| is clamped to the largest finite e5m2
      const int nch = Cin / 32, nsteps = nch * 6;
      auto mod3 = [](int x) { return x >= 6 ? x - 6 : (x >= 3 ? x - 3 : x); };
      const sdm_rsrc rs8 = sdm_make_rsrc(p.w_dma, (unsigned int)((size_t)Cin * NTAPS * p.Cout_pad * 4));
      // 24 pieces of 1 KB per step: piece q = (unit, plane, dy, 64-channel half); this wave issues pieces 6*wave .. 6*wave+5
      auto dma_step_v = [&](int t, int sl, unsigned int voff) {
        unsigned char* dst = Bring + sl * STEP;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
          const int q = wv * 6 + k, ul = q / 12, qq = q % 12, pl = qq / 6, dy = (qq >> 1) % 3, ch = qq & 1;
          const unsigned int row = (unsigned int)(((t * 2 + ul) * 2 + pl) * 3 + dy);
          sdm_glds16_buf(rs8, voff + (unsigned int)(ch * 1024), row * stage_rows, dst + ul * SLOT + pl * PLANE + dy * (BN * 16) + ch * 1024);
        }
      };
      auto dma_step = [&](int t, int sl) { dma_step_v(t, sl, dma_voff); };
      f32x4 gq[4];                  // producers, GN: scale[0:4], scale[4:8], shift[0:4], shift[4:8] of this thread's channels in the next chunk
      auto issue_gn = [&](int c0w) {
        if (GN) {
          const float* ts = p.gn_scale + (size_t)img * Cin + c0w + a_part;
          const float* th = p.gn_shift + (size_t)img * Cin + c0w + a_part;
          gq[0] = *(const f32x4*)ts; gq[1] = *(const f32x4*)(ts + 4); gq[2] = *(const f32x4*)th; gq[3] = *(const f32x4*)(th + 4);
        }
      };
      // fp32 (after the fused GroupNorm / SiLU) -> fp16 high parts (4 planes of 16-B rows: channel group g of the chunk) and the
      // fp8 images of x_lo and x (region behind: sub-planes [x_lo8 ch 0-15 | x_lo8 ch 16-31 | x8 ch 0-15 | x8 ch 16-31] of 16-B rows)
      // SRC_NX (a literal at every call site): the raw values come from a_raw, or straight from a_nx
      auto write_lds_a_f8 = [&](const bool SRC_NX, unsigned char* Ad, int i0, int i1, bool nxt = false, const int* npix = nullptr) {        // vectors i0 .. i1-1 of this thread
        const int g = a_part >> 3;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          if (i >= i0 && i < i1 && tid + i * NT < A_VEC) {
            const f32x4 v0 = __builtin_bit_cast(f32x4, SRC_NX ? a_nx[i][0] : a_raw[i][0]), v1 = __builtin_bit_cast(f32x4, SRC_NX ? a_nx[i][IN_F32 ? 1 : 0] : a_raw[i][IN_F32 ? 1 : 0]);
            const bool inside = (nxt ? npix[i] : a_pix[i]) >= 0;
            f16x8 vh;
            float xl[8], xx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float y = e < 4 ? v0[e & 3] : v1[e & 3];
              if (GN) {
                y = y * (e < 4 ? gq[0][e & 3] : gq[1][e & 3]) + (e < 4 ? gq[2][e & 3] : gq[3][e & 3]);
                if (p.gn_silu) y = y * sdm_rcp(1.0f + sdm_exp2(-y * SDM_LOG2E));
                if (!inside) y = 0.0f;                     // zero padding stays zero AFTER the normalisation
              }
              // e5m2 covers the whole fp16 range: ONE clamp (v_med3) keeps hi, x8 and x_lo8 finite whatever the activation (|x_lo| * 2^11 <= |x|)
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
            unsigned char* a8 = Ad + C::A_BYTES + (g >> 1) * A_HALF + hp * 16 + (g & 1) * 8;
            u32x2 wl, wx;
            wl[0] = (unsigned int)l0; wl[1] = (unsigned int)l1; wx[0] = (unsigned int)x0; wx[1] = (unsigned int)x1;
            *(u32x2*)a8 = wl;
            *(u32x2*)(a8 + 2 * A_HALF) = wx;
          }
        }
      };
      constexpr int AFLY = A_PER * (IN_F32 ? 2 : 1) + (GN ? 4 : 0);      // register loads that may stay in flight behind a step's DMAs
      static_assert(!F8 || NTAPS == 1 || AFLY == 12 || AFLY == 16, "counted wait below");
      // activations (and GroupNorm coefficients) of the chunk AFTER the next one: loaded a whole chunk before they are transformed, so
      // that the transform of the next chunk can be spread evenly over the six steps of the current one (one vector per step)
      // (a_nx / gqn live at kernel scope: they carry the next tile's chunk 1 across the tile boundary)
      // ---- the NEXT tile of this block, as far as the producers need it (3x3 only; even chunk counts keep the A buffer parity) ----
      bool has_next = false;
      int n_img = 0, n_n0 = 0;
      int n_pix[A_PER];
      sdm_rsrc n_rs0 = rs0, n_rs1 = rs1;
      sdm_rsrc_raw n_rq0 = rq0, n_rq1 = rq1;      // (3x3: the per-step loads go through inline asm, below)
#pragma unroll
      for (int i = 0; i < A_PER; ++i) n_pix[i] = -1;
      if (NTAPS == 9 && role && more_tiles && p.xtile && (nch & 1) == 0) {
        int mlin, nnt;
        if (tile_decode(vbid + (int)gridDim.x, mlin, nnt)) {
          has_next = true;
          n_img = mlin / p.tiles_m;
          const int nmt = mlin - n_img * p.tiles_m;
          n_n0 = nnt * BN;
          const int npx = (p.Wout + TW - 1) / TW;
          const int noy0 = (nmt / npx) * TH, nox0 = (nmt % npx) * TW;
          const int nband0 = ((noy0 * STRIDE - p.pad_t) > 0 ? ((noy0 * STRIDE - p.pad_t) >> p.up) : 0);
          const int nband_rows = ((p.Hin - nband0) < (C::HPH + 1) ? (p.Hin - nband0) : (C::HPH + 1));
#pragma unroll
          for (int i = 0; i < A_PER; ++i) {
            const int hp = a_hp0 + i * (NT / KV);
            if (tid + i * NT < A_VEC) {
              const int hy = hp / HPW, hx = hp % HPW;
              const int iy = noy0 * STRIDE + hy - p.pad_t, ix = nox0 * STRIDE + hx - p.pad_l;
              if (iy >= 0 && iy < Hl && ix >= 0 && ix < Wl) n_pix[i] = ((iy >> p.up) - nband0) * p.Win + (ix >> p.up);
            }
          }
          const size_t base_px = ((size_t)n_img * p.Hin + nband0) * p.Win, npxs = (size_t)nband_rows * p.Win;
          n_rs0 = sdm_make_rsrc((const unsigned char*)p.in0 + base_px * p.C0 * es, (unsigned int)(npxs * p.C0 * es));
          n_rs1 = sdm_make_rsrc(p.in1 ? (const unsigned char*)p.in1 + base_px * p.C1 * es : (const unsigned char*)p.in0, p.in1 ? (unsigned int)(npxs * p.C1 * es) : 0u);
          n_rq0 = sdm_make_rsrc_raw((const unsigned char*)p.in0 + base_px * p.C0 * es, (unsigned int)(npxs * p.C0 * es));
          n_rq1 = sdm_make_rsrc_raw(p.in1 ? (const unsigned char*)p.in1 + base_px * p.C1 * es : (const unsigned char*)p.in0, p.in1 ? (unsigned int)(npxs * p.C1 * es) : 0u);
        }
      }
      const unsigned int n_dma_voff = (unsigned int)((n_n0 + lane) * 16);
      // the same six pieces with the address arithmetic hoisted (3x3 step loop): this wave's pieces are (unit, plane) = (wave >> 1, wave & 1) x
      // (dy = k >> 1, channel half = k & 1), i.e. consecutive 1 KB blocks in LDS and rows g, g + stage_rows, g + 2 stage_rows of the packed tensor -
      // two scalar adds per piece instead of the ten the generic index expression compiled to (a producer wave issues one instruction per 4 cycles)
      const unsigned int g_wave = (unsigned int)((wv >> 1) * 6 + (wv & 1) * 3) * stage_rows, sr12 = 12u * stage_rows;
      const unsigned int l_wave = (unsigned int)(C::RING_OFF + (wv >> 1) * SLOT + (wv & 1) * PLANE);
      const unsigned int dv0 = dma_voff, dv1 = dma_voff + 1024u, ndv0 = n_dma_voff, ndv1 = n_dma_voff + 1024u;
      auto dma_fast = [&](int t, int sl, unsigned int v0, unsigned int v1) {
        const unsigned int g = (unsigned int)t * sr12 + g_wave;
        unsigned char* d = smem + l_wave + sl * STEP;
#pragma unroll
        for (int k = 0; k < 6; ++k) sdm_glds16_buf(rs8, (k & 1) ? v1 : v0, g + (unsigned int)(k >> 1) * stage_rows, d + k * 1024);
      };
      // nxt: addresses of the next tile (cross-tile prefetch)
      auto issue_loads_nx = [&](int c0, bool nxt = false) {
        const bool second = c0 >= p.C0;
        const sdm_rsrc rs = nxt ? (second ? n_rs1 : n_rs0) : (second ? rs1 : rs0);
        const unsigned int Cs = (unsigned int)(second ? p.C1 : p.C0) * es;
        const unsigned int cc = (unsigned int)((second ? c0 - p.C0 : c0) + a_part) * es;
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          const int px = lab_win ? ((nxt ? n_pix[i] : a_pix[i]) & 63) : (nxt ? n_pix[i] : a_pix[i]);
          const unsigned int off = px >= 0 ? (unsigned int)px * Cs + cc : SDM_BUF_INVALID;
          a_nx[i][0] = sdm_buffer_load16(rs, off, 0);
          if (IN_F32) a_nx[i][IN_F32 ? 1 : 0] = sdm_buffer_load16(rs, off, 16);
        }
        if (GN) {
          const float* ts = p.gn_scale + (size_t)(nxt ? n_img : img) * Cin + c0 + a_part;
          const float* th = p.gn_shift + (size_t)(nxt ? n_img : img) * Cin + c0 + a_part;
          gqn[0] = *(const f32x4*)ts; gqn[1] = *(const f32x4*)(ts + 4); gqn[2] = *(const f32x4*)th; gqn[3] = *(const f32x4*)(th + 4);
        }
      };
      auto take_nx = [&]() {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) { a_raw[i][0] = a_nx[i][0]; if (IN_F32) a_raw[i][IN_F32 ? 1 : 0] = a_nx[i][IN_F32 ? 1 : 0]; }
        if (GN) { gq[0] = gqn[0]; gq[1] = gqn[1]; gq[2] = gqn[2]; gq[3] = gqn[3]; }
      };
      // ---- 3x3: the same loads ONE VECTOR PER STEP, through inline asm (the compiler's own wait-count bookkeeping knows nothing of them: a wait it
      // inserted for one of them would drain the DMA queue) ----
      // (one descriptor: the launcher guarantees gn_shift = gn_scale + N * Cin - both halves of one scratch tensor, sdm_engine.cpp gn_scale_shift)
      const sdm_rsrc_raw rq_gs = sdm_make_rsrc_raw(p.gn_scale, GN ? (unsigned int)((size_t)p.N * Cin * 8) : 0u);
      const unsigned int gh_delta = (unsigned int)((size_t)p.N * Cin * 4);
      auto issue_nx_vec = [&](int i0, const sdm_rsrc_raw rq, unsigned int Cs, unsigned int cc, bool nxt) {      // vector i0 of a chunk -> a_nx[i0]
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          if (i != i0) continue;
          // (both values pinned in registers first: left to itself the optimiser selects between the two ADDRESSES, which keeps both
          //  arrays in scratch memory - a scratch load and a vmcnt(0) in every step)
          int pa = a_pix[i], pn = n_pix[i];
          SDM_OPAQUE_I(pa); SDM_OPAQUE_I(pn);
          const int px0 = nxt ? pn : pa;
          const int px = lab_win ? (px0 & 63) : px0;
          const unsigned int off = px >= 0 ? SDM_UMUL24((unsigned int)px, Cs) + cc : SDM_BUF_INVALID;      // px < 11 rows x Win, Cs <= 10 KB: 24-bit operands
          SDM_ASM_BUFFER_LOAD16_FIRST(a_nx[i][0], off, rq, 0);
          if (IN_F32) SDM_ASM_BUFFER_LOAD16(a_nx[i][IN_F32 ? 1 : 0], off, rq, 16);
        }
      };
      auto issue_nx_gn = [&](int c0, bool nxt) {                // GroupNorm scale / shift of this thread's 8 channels in that chunk -> gqn
        if (GN) {
          const unsigned int off = (unsigned int)(((nxt ? n_img : img) * Cin + c0 + a_part) * 4), offh = off + gh_delta;
          u32x4 t0, t1, t2, t3;
          SDM_ASM_BUFFER_LOAD16_FIRST(t0, off, rq_gs, 0);
          SDM_ASM_BUFFER_LOAD16(t1, off, rq_gs, 16);
          SDM_ASM_BUFFER_LOAD16(t2, offh, rq_gs, 0);
          SDM_ASM_BUFFER_LOAD16(t3, offh, rq_gs, 16);
          gqn[0] = __builtin_bit_cast(f32x4, t0); gqn[1] = __builtin_bit_cast(f32x4, t1); gqn[2] = __builtin_bit_cast(f32x4, t2); gqn[3] = __builtin_bit_cast(f32x4, t3);
        }
      };
      // a_raw[i0] <- a_nx[i0] (loaded six steps ago: every step's end waits for everything older than the two youngest steps' operations).
      // The empty asm pins the copy BEHIND the waits and barriers in between: for the compiler an asm load's result exists at once
      auto take_vec = [&](int i0) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          if (i != i0) continue;
#ifndef SDM_EMU
          asm volatile("" : "+v"(a_nx[i][0]), "+v"(a_nx[i][IN_F32 ? 1 : 0]) :: "memory");
#endif
          a_raw[i][0] = a_nx[i][0];
          if (IN_F32) a_raw[i][IN_F32 ? 1 : 0] = a_nx[i][IN_F32 ? 1 : 0];
        }
      };
      auto take_gn = [&]() {
        if (GN) {
#ifndef SDM_EMU
          asm volatile("" : "+v"(gqn[0]), "+v"(gqn[1]), "+v"(gqn[2]), "+v"(gqn[3]) :: "memory");
#endif
          gq[0] = gqn[0]; gq[1] = gqn[1]; gq[2] = gqn[2]; gq[3] = gqn[3];
        }
      };
      if (NTAPS == 1) {
        // ---- 1x1 / Linear GEMM in the same form.  One step = one 32-channel chunk = four 4 KB weight units (w_hi ch 0-15 | w_hi
        // ch 16-31 | w8 | w_lo8): 16 fp16 MFMAs + 8 fp8 K64 MFMAs per wave, one barrier.  No operand reuse across taps here: the
        // producers convert a 32 KB activation tile per 1024 MFMA cycles and set the pace (the 4-wave kernel did the same conversion
        // inside its MFMA waves, between four barriers per chunk). ----
        constexpr int PLANE1 = BN * 16, STEP1 = 4 * SLOT;
        static_assert(NTAPS != 1 || !F8 || A_PER == 4, "1x1 F8: four vectors per producer thread and chunk");
        auto dma_step1 = [&](int t, int sl) {      // 16 pieces of 1 KB: (unit, plane, 64-channel half); this wave issues 4
          unsigned char* dst = Bring + sl * STEP1;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int q = wv * 4 + k, u = q >> 2, pl = (q >> 1) & 1, ch = q & 1;
            const unsigned int row = (unsigned int)((t * 4 + u) * 2 + pl);
            sdm_glds16_buf(rs8, dma_voff + (unsigned int)(ch * 1024), row * stage_rows, dst + u * SLOT + pl * PLANE1 + ch * 1024);
          }
        };
        if (role) {
          {   // bias of this tile's output channels -> LDS (read by the consumers' accumulator-layout epilogue)
            const float* bsrc = p.bias;
            if (bsrc && p.bias_sel) bsrc += (size_t)p.bias_sel[img] * p.Cout_pad;
            if (tid < BN) bias_tab[tid] = (bsrc && n0 + tid < p.Cout_pad) ? bsrc[n0 + tid] : 0.0f;
          }
          dma_step1(0, 0);
          if (1 < nch) dma_step1(1, 1);
          SDM_SCHED_FENCE();
          issue_loads_a(0);
          write_lds_a_f8(false, Aring, 0, A_PER);
          SDM_WAIT_VMCNT0();
          if (nch > 1) { issue_loads_nx(32); SDM_SCHED_FENCE(); }
        }
        if (role) {
          SDM_WAIT_LGKMCNT0();
          SDM_RAW_BARRIER();
          int sl = 2;                   // ring slot of chunk c+2
          for (int c = 0; c < nch; ++c) {
            const bool more = c + 1 < nch, fly = c + 2 < nch;
            if (more) take_nx();
            SDM_SCHED_FENCE();
            if (fly) { dma_step1(c + 2, sl); SDM_SCHED_FENCE(); issue_loads_nx((c + 2) * 32); SDM_SCHED_FENCE(); }
            if (more) write_lds_a_f8(false, Aring + ((c + 1) & 1) * 2 * C::A_BYTES, 0, A_PER);
            if (fly) SDM_WAIT_VMCNT(8); else SDM_WAIT_VMCNT0();
            SDM_WAIT_LGKMCNT0();
            SDM_RAW_BARRIER();
            sl = sl == 2 ? 0 : sl + 1;
          }
        } else {
          acc_init_residual();
          SDM_RAW_BARRIER();            // the producers' prologue (same barrier as in their branch)
          const int sa8 = p.f8_sa, sb8 = p.f8_sb;
          f16x8 fbh[2][NTL];
          i32x8 fb8[NTL];
          int bq1[NTL], a8b[MT];
#pragma unroll
          for (int j = 0; j < NTL; ++j) bq1[j] = (wn * WTN + j * 32 + (lane & 31)) * 16;
#pragma unroll
          for (int i = 0; i < MT; ++i) a8b[i] = abase[i] - (lane >> 5) * A_HALF + (lane >> 5) * 2 * A_HALF;
          auto ld_b = [&](const unsigned char* Bp) {
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
              fbh[0][j] = *(const f16x8*)(Bp + (lane >> 5) * PLANE1 + bq1[j]);
              fbh[1][j] = *(const f16x8*)(Bp + SLOT + (lane >> 5) * PLANE1 + bq1[j]);
              const unsigned char* q = Bp + (2 + (lane >> 5)) * SLOT + bq1[j];
              const i32x4 q0 = *(const i32x4*)q, q1 = *(const i32x4*)(q + PLANE1);
              fb8[j] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            }
          };
          ld_b(Bring);
          int sl = 1;                   // ring slot of chunk c+1
          for (int c = 0; c < nch; ++c) {
            const unsigned char* Ab = Aring + (c & 1) * 2 * C::A_BYTES;
            f16x8 ah[2][MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) ah[0][i] = *(const f16x8*)(Ab + abase[i]);
#pragma unroll
            for (int i = 0; i < MT; ++i) ah[1][i] = *(const f16x8*)(Ab + abase[i] + 2 * A_HALF);
            SDM_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(fbh[0][j], ah[0][i], acc[i][j]);
            i32x8 a8[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
              const unsigned char* q = Ab + C::A_BYTES + a8b[i];
              const i32x4 q0 = *(const i32x4*)q, q1 = *(const i32x4*)(q + A_HALF);
              a8[i] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
            }
            SDM_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(fbh[1][j], ah[1][i], acc[i][j]);
            SDM_SCHED_FENCE();
            // the residual pair of this chunk; the fp16 fragments of the NEXT chunk's weights (landed one barrier ago) are read underneath
            i32x8 fb8c[NTL];
#pragma unroll
            for (int j = 0; j < NTL; ++j) fb8c[j] = fb8[j];
            if (c + 1 < nch) ld_b(Bring + sl * STEP1);
            SDM_SCHED_FENCE();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x64_F8A_BF8B(fb8c[j], a8[i], acc[i][j], sb8, sa8);
            SDM_RAW_BARRIER();
            sl = sl == 2 ? 0 : sl + 1;
          }
        }
      } else {
      // ---- 3x3 (ConvCfg::R4).  ONE A buffer and a ring of FOUR weight steps.  The chunk being staged (c + 1, or chunk 0 of the block's next tile) lives in
      // the producers' registers while chunk c is multiplied: one vector per step is transformed (GroupNorm, SiLU, fp16 high part, e5m2 pair) into
      // registers (in place, below); the high planes are written during step 5 of chunk c (S2 of the last kernel column: the consumers read only the fp8 region),
      // the fp8 region during step 0 of chunk c + 1 (S1: only the high planes are read) - or, for a tile's chunk 0, in front of the tile's first
      // barrier.  The 43.5 KB this frees hold the fourth weight step: the DMAs of step t + 3 are issued at the start of step t and have to be
      // visible when the barrier of step t + 1 releases - two steps of landing time.  That matters once per chunk: the activation loads of chunk
      // c + 2 (first touch of HBM, ~4.5 k cycles under this kernel's traffic) are issued at step 0, vector-memory operations return in order, and
      // with one step of landing time the DMAs queued behind them stalled every chunk by 1.5 - 1.8 k cycles (profiles/r04_conv_f8_step_trace.txt).
      // Ring slot of step t = t & 3 = (cm + k) & 3 with cm = (6 c) & 3 = 2 (c & 1), a wave-uniform run-time value: the producers' DMA destinations
      // are scalar anyway (M0), the consumers add the slot's offset to their fragment bases once per step (a handful of VALU adds beside 48 MFMAs).
      // The staged chunk is transformed IN PLACE: a_raw[i][0] becomes the vector's fp16 high parts, a_raw[i][1] its [x_lo8 | x8] image. ----
      static_assert(NTAPS != 9 || !F8 || A_PER == 6, "one vector of the next chunk per step");
      // bias of a tile's BN output channels -> LDS (read by the consumers' accumulator-layout epilogue, a whole tile later), by LDS-DMA: no register,
      // no wait - a load + ds_write here made the compiler drain the whole vector-memory queue (every DMA and activation load in flight) once
      // per tile.  Each wave copies 64 floats (waves 2 / 3 repeat those of waves 0 / 1); channels beyond Cout_pad read 0 (descriptor bounds)
      auto dma_bias_tab = [&](float* tab, int im, int nn0) {
        const float* bsrc = p.bias;
        const int sel = (bsrc && p.bias_sel) ? SDM_UNIFORM_I(p.bias_sel[im]) : 0;
        const sdm_rsrc rb = sdm_make_rsrc(bsrc ? bsrc + (size_t)sel * p.Cout_pad : (const float*)p.w, bsrc ? (unsigned int)p.Cout_pad * 4u : 0u);
        sdm_glds4_buf(rb, (unsigned int)((nn0 + (wv & 1) * 64 + lane) * 4), 0u, (unsigned char*)tab + (wv & 1) * 256);
      };
      // vector i of this thread: raw fp32 (a_raw[i][0 | 1]) -> a_raw[i][0] = high parts, a_raw[i][1] = fp8 image.
      // A producer wave issues ONE instruction per four cycles at best, and it has to stage a vector per step beside the DMA and load issue: every
      // instruction here is 1/400 of a step.  Hence: ONE conversion per pair to fp16 (v_cvt_pk_f16_f32) whose halves are converted back for the low
      // parts, SiLU as a compile-time variant chosen by a wave-uniform branch, no per-lane validity test (vectors beyond the tile are transformed
      // like the others and dropped by the LDS writes).  NO packed fp32 (v_pk_fma / mul / add): beside a saturated matrix pipe they issue far slower
      // than two scalar instructions (measured: the step 1.65 k -> 2.1 k cycles, profiles/r05_conv_f8_producer_variants.txt).
      auto xform = [&](const bool SILU, int i0, bool nxt) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
          if (i != i0) continue;
          const f32x4 v0 = __builtin_bit_cast(f32x4, a_raw[i][0]), v1 = __builtin_bit_cast(f32x4, a_raw[i][IN_F32 ? 1 : 0]);
          float y[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = e < 4 ? v0[e & 3] : v1[e & 3];
          // zero padding stays zero AFTER the normalisation: a pixel outside the image is clamped to [0, 0] - two selects per vector instead of eight
          int pa = a_pix[i], pn = n_pix[i];      // (pinned in registers: see issue_nx_vec)
          SDM_OPAQUE_I(pa); SDM_OPAQUE_I(pn);
          const bool inside = !GN || (nxt ? pn : pa) >= 0;
          const float c_lo = inside ? -F8_AMAX : 0.0f, c_hi = inside ? F8_AMAX : 0.0f;
          if (GN) {
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = y[e] * (e < 4 ? gq[0][e & 3] : gq[1][e & 3]) + (e < 4 ? gq[2][e & 3] : gq[3][e & 3]);
            if (SILU) {
              // stage by stage over the eight values (not value by value): every transcendental has seven independent instructions behind it
              float t[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) t[e] = y[e] * (-SDM_LOG2E);
#pragma unroll
              for (int e = 0; e < 8; ++e) t[e] = sdm_exp2(t[e]);
#pragma unroll
              for (int e = 0; e < 8; ++e) t[e] = t[e] + 1.0f;
#pragma unroll
              for (int e = 0; e < 8; ++e) t[e] = sdm_rcp(t[e]);
#pragma unroll
              for (int e = 0; e < 8; ++e) y[e] = y[e] * t[e];
            }
          }
          // one clamp keeps hi, x8 and x_lo8 finite (e5m2 has the range of fp16) - and makes padding pixels exactly zero
#pragma unroll
          for (int e = 0; e < 8; ++e) y[e] = SDM_MED3(y[e], c_lo, c_hi);
          u32x4 hq, q;
          float lo[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const f16x2 h2 = __builtin_convertvector(f32x2{y[2 * j], y[2 * j + 1]}, f16x2);      // ONE v_cvt_pk_f16_f32 per pair
            hq[j] = __builtin_bit_cast(unsigned int, h2);
            lo[2 * j] = (y[2 * j] - (float)h2[0]) * F8_LS;
            lo[2 * j + 1] = (y[2 * j + 1] - (float)h2[1]) * F8_LS;
          }
          int l0 = SDM_CVT_PK_BF8(lo[0], lo[1], 0, false), l1 = SDM_CVT_PK_BF8(lo[4], lo[5], 0, false);
          l0 = SDM_CVT_PK_BF8(lo[2], lo[3], l0, true); l1 = SDM_CVT_PK_BF8(lo[6], lo[7], l1, true);
          int x0 = SDM_CVT_PK_BF8(y[0], y[1], 0, false), x1 = SDM_CVT_PK_BF8(y[4], y[5], 0, false);
          x0 = SDM_CVT_PK_BF8(y[2], y[3], x0, true); x1 = SDM_CVT_PK_BF8(y[6], y[7], x1, true);
          q[0] = (unsigned int)l0; q[1] = (unsigned int)l1; q[2] = (unsigned int)x0; q[3] = (unsigned int)x1;
          a_raw[i][0] = hq;
          a_raw[i][IN_F32 ? 1 : 0] = q;
        }
      };
      // the SiLU variant of the layer (wave-uniform branch; GroupNorm without SiLU exists in the op tests only)
      auto xform_sel = [&](int i0, bool nxt) {
        if (GN && p.gn_silu) xform(true, i0, nxt); else xform(false, i0, nxt);
      };
      auto write_hi = [&](int i0, int i1) {
        const int g = a_part >> 3;
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
          if (i >= i0 && i < i1 && tid + i * NT < A_VEC) *(u32x4*)(Aring + g * A_HALF + (a_hp0 + i * (NT / KV)) * 16) = a_raw[i][0];
      };
      auto write_f8 = [&](const bool FROM_CARRY) {      // FROM_CARRY (a literal at every call site): the image that crossed the tile boundary in f8h
        const int g = a_part >> 3;
#pragma unroll
        for (int i = 0; i < A_PER; ++i)
          if (tid + i * NT < A_VEC) {
            unsigned char* a8 = Aring + C::A_BYTES + (g >> 1) * A_HALF + (a_hp0 + i * (NT / KV)) * 16 + (g & 1) * 8;
            u32x2 wl, wx;
            const u32x4 q = FROM_CARRY ? f8h[i] : a_raw[i][IN_F32 ? 1 : 0];
            wl[0] = q[0]; wl[1] = q[1]; wx[0] = q[2]; wx[1] = q[3];
            *(u32x2*)a8 = wl;
            *(u32x2*)(a8 + 2 * A_HALF) = wx;
          }
      };
      stamp(1);                         // tile start (both roles)
      if (role) {
        dma_bias_tab(bias_tab, img, n0);      // (needed by the consumers' epilogue; the step loop's waits see it landed long before)
        if (!pre_done) {                // (a tile prepared by its predecessor has its weights in the ring, its chunk 0 in LDS (high planes) / f8h, its chunk 1 in a_nx)
          dma_step(0, 0);
          dma_step(1, 1);
          dma_step(2, 2);
          SDM_SCHED_FENCE();
          issue_loads_a(0);
          issue_gn(0);
#pragma unroll
          for (int i = 0; i < A_PER; ++i) xform_sel(i, false);
          write_hi(0, A_PER);
          write_f8(false);
          SDM_WAIT_VMCNT0();
          if (nch > 1) {
            // chunk 1 - waited for HERE, inside the block that issues the loads: a_nx / gqn are loop-carried, and a register copy hipcc may place
            // where this path joins the loop must not find them half-written (an asm load's result "exists" at once for the compiler).  A
            // block's first tile only: every later tile is staged by its predecessor
#pragma unroll
            for (int i = 0; i < A_PER; ++i) issue_nx_vec(i, (32 >= p.C0) ? rq1 : rq0, (unsigned int)((32 >= p.C0) ? p.C1 : p.C0) * es,
                                                         (unsigned int)(((32 >= p.C0) ? 32 - p.C0 : 32) + a_part) * es, false);
            issue_nx_gn(32, false);
            SDM_SCHED_FENCE();
            SDM_WAIT_VMCNT0();
#ifndef SDM_EMU
#pragma unroll
            for (int i = 0; i < A_PER; ++i) asm volatile("" : "+v"(a_nx[i][0]), "+v"(a_nx[i][IN_F32 ? 1 : 0]) :: "memory");
            if (GN) asm volatile("" : "+v"(gqn[0]), "+v"(gqn[1]), "+v"(gqn[2]), "+v"(gqn[3]) :: "memory");
#endif
          }
        } else {
          write_f8(true);               // chunk 0's fp8 image: nobody reads that region between a tile's last barrier and the next tile's first
        }
        SDM_WAIT_LGKMCNT0();
        stamp();                        // prologue done, arriving at the first barrier
        SDM_RAW_BARRIER();
        stamp();                        // past the first barrier
        int cm = 0;                     // ring slot of the chunk's first step
        for (int c = 0; c < nch; ++c) {
          const bool more = c + 1 < nch;
          const bool morex = more || has_next;             // the chunk staged during this one: c + 1, or chunk 0 of the next tile
          const bool flyc = ((c + 2 < nch) || has_next) && !lab_nold;      // this chunk's steps issue the loads of the chunk after the staged one
          // where those loads come from: chunk c + 2 of this tile, or chunk 0 / 1 of the next one (descriptor, row pitch and channel offset per chunk)
          const bool ld_nxt = !(c + 2 < nch);
          const int ld_c0 = ld_nxt ? (c + 2 - nch) * 32 : (c + 2) * 32;
          const bool ld_second = ld_c0 >= p.C0;
          const sdm_rsrc_raw ld_rq = ld_nxt ? (ld_second ? n_rq1 : n_rq0) : (ld_second ? rq1 : rq0);
          const unsigned int ld_Cs = (unsigned int)(ld_second ? p.C1 : p.C0) * es, ld_cc = (unsigned int)((ld_second ? ld_c0 - p.C0 : ld_c0) + a_part) * es;
#pragma unroll
          for (int k = 0; k < 6; ++k) {                  // step t = 6c + k: (dx = k / 2, S1 | S2)
            const int t = c * 6 + k;
            if (k == 0 && c > 0) write_f8(false);         // the current chunk's fp8 image (its high planes went out during the previous step)
            // vector k of the staged chunk: loaded six steps ago (every step's end waits for everything but the two youngest steps' operations)
            if (morex) { take_vec(k); if (k == 0) take_gn(); }
            SDM_SCHED_FENCE();
            stamp2();
            bool d3 = false;                              // this step issues the DMAs of step t + 3 (of this tile, or step 0 - 2 of the next one)
            if (!(lab_nodma && t >= 2)) {
              if (t + 3 < nsteps) { dma_fast(t + 3, (cm + k + 3) & 3, dv0, dv1); d3 = true; }
              else if (has_next) { dma_fast(t + 3 - nsteps, (cm + k + 3) & 3, ndv0, ndv1); d3 = true; }      // nsteps % 4 == 0 (even chunk counts): same slots
            }
            SDM_SCHED_FENCE();
            stamp2();
            // vector k of chunk c+2 - of this tile, or of chunk 0 / 1 of the next one - BEHIND this step's DMAs: transformed six steps from now.  ONE vector
            // (two 16-byte loads) per step: the CU accepts a burst of a whole chunk's 64 load instructions (first touches of HBM) only over ~2 k
            // cycles, and the producer waves sat in that issue queue at every chunk's first step (profiles/r05_conv_f8_step_trace.txt); with
            // the ring of four a load has three steps to land before anything queued behind it is waited for
            if (flyc) {
              issue_nx_vec(k, ld_rq, ld_Cs, ld_cc, ld_nxt);
              if (k == 0) issue_nx_gn(ld_c0, ld_nxt);
            }
            SDM_SCHED_FENCE();
            stamp2();
            if (k == 5 && morex && !lab_nowr) write_hi(0, 5);      // (five of the six writes complete under the last vector's transform)
            if (morex && !lab_nowr) xform_sel(k, !more);
            if (k == 5 && morex && !lab_nowr) write_hi(5, 6);
            stamp2();
            // the DMAs of step t + 2 (issued one step ago) are visible behind this barrier; what was issued after them may stay in flight: the previous
            // step's loads (2, + 4 GroupNorm coefficient loads at step 0), this step's DMAs (6 per wave) and this step's loads
            if (!d3) SDM_WAIT_VMCNT0();
            else if (!flyc) SDM_WAIT_VMCNT(6);
            else if (k == 0 && c == 0) { if (GN) SDM_WAIT_VMCNT(15); else SDM_WAIT_VMCNT(11); }      // + this tile's bias DMA
            else if (k < 2 && GN) SDM_WAIT_VMCNT(14);
            else SDM_WAIT_VMCNT(10);
            SDM_WAIT_LGKMCNT0();
            stamp();                    // step t: arriving at its barrier
            SDM_RAW_BARRIER();
            stamp();                    // step t: released
          }
          cm ^= 2;
        }
        if (has_next) {                 // the next tile's chunk 0: its fp8 image crosses the tile boundary in registers
#pragma unroll
          for (int i = 0; i < A_PER; ++i) f8h[i] = a_raw[i][IN_F32 ? 1 : 0];
        }
        pre_done = has_next ? 1 : 0;
      } else {
        acc_init_residual();
        stamp();                        // accumulators initialised, arriving at the first barrier
        SDM_RAW_BARRIER();              // the producers' prologue (same barrier as in their branch)
        stamp();                        // past the first barrier
        const int sa8 = p.f8_sa, sb8 = p.f8_sb;
        f16x8 fbh[2][3][NTL];         // w_hi fragments of the two 16-channel halves
        i32x8 fb8[3][NTL];            // [w8 | w_lo8] fragments
        const int a8base = abase[0] - (lane >> 5) * A_HALF + (lane >> 5) * 2 * A_HALF;      // fp8 region: lanes 0-31 read x_lo8, lanes 32-63 x8
        int bq8[NTL];
#pragma unroll
        for (int j = 0; j < NTL; ++j) bq8[j] = (lane >> 5) * SLOT + (wn * WTN + j * 32 + (lane & 31)) * 16;
        auto ld_bh = [&](int ks, int dy, const unsigned char* Bp) {
#pragma unroll
          for (int j = 0; j < NTL; ++j) fbh[ks][dy][j] = *(const f16x8*)(Bp + ks * SLOT + bq[j] + dy * (BN * 16));
        };
        auto ld_b8 = [&](int dy, const unsigned char* Bp) {
#pragma unroll
          for (int j = 0; j < NTL; ++j) {
            const i32x4 q0 = *(const i32x4*)(Bp + bq8[j] + dy * (BN * 16)), q1 = *(const i32x4*)(Bp + bq8[j] + PLANE + dy * (BN * 16));
            fb8[dy][j] = i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
          }
        };
        auto ld_ah = [&](const unsigned char* Ap, int ks, int r, int dx) { return *(const f16x8*)(Ap + abase[0] + ks * 2 * A_HALF + (r * HPW + dx) * 16); };
        auto ld_a8 = [&](const unsigned char* Ap, int r, int dx) {
          const unsigned char* q = Ap + C::A_BYTES + a8base + (r * HPW + dx) * 16;
          const i32x4 q0 = *(const i32x4*)q, q1 = *(const i32x4*)(q + A_HALF);
          return i32x8{q0[0], q0[1], q0[2], q0[3], q1[0], q1[1], q1[2], q1[3]};
        };
        {   // operands of the first step
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) ld_bh(0, dy, Bring);
        }
        int cm = 0;                     // ring slot of the chunk's first step (wave-uniform)
        for (int c = 0; c < nch; ++c) {
          const bool more = c + 1 < nch;
          const unsigned char* Ab = Aring;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const unsigned char* B1 = Bring + ((cm + dx * 2) & 3) * STEP;           // S1 of this column
            const unsigned char* B2 = Bring + ((cm + dx * 2 + 1) & 3) * STEP;       // S2 of this column
            // ---- S1: A_hi . w_hi, channels 0-15 then 16-31; the second half's and the fp8 step's B fragments are read underneath ----
            f16x8 fa[2];
            i32x8 f8a[2];
            if (!lab_nomm) {
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
              else if (dx != 0) f8a[0] = ld_a8(Ab, 0, dx);      // (dx == 0: the producers write this chunk's fp8 region during this very step)
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
            }
            SDM_RAW_BARRIER();
            // ---- S2: [x_lo8 | x8] . [w8 | w_lo8]; the w_hi fragments of the next column (or chunk) are read underneath ----
            const bool last = (dx == 2) && !more;
            const unsigned char* Bn = Bring + ((cm + dx * 2 + 2) & 3) * STEP;
            if (!lab_nomm) {
            if (dx == 0) f8a[0] = ld_a8(Ab, 0, dx);
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
            }
            SDM_RAW_BARRIER();
          }
          cm ^= 2;
        }
      }
      }   // NTAPS == 9
    } else if (PC) {
      // ---- producer / consumer form: same stages and barriers for both roles, disjoint instruction streams.
      // Invariant: when the barrier that ends stage t releases, the weights of stage t+2 have landed (ring of 5, DMAs 4 stages
      // ahead), so a consumer reads the B fragments of stage t+1 WHILE it multiplies stage t and enters every stage with its
      // operands in registers; stage t's ring slot is overwritten only after barrier t (its fragments were consumed before). ----
      constexpr int NR = MT + 2;
      auto mod5 = [](int x) { return x >= 10 ? x - 10 : (x >= 5 ? x - 5 : x); };
      if (role) {
        issue_loads_a(0);
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s < nst) dma_stage_sl(s, s);
      }
      if (GN) __syncthreads();      // gn_tab filled
      if (role) {
        write_lds_a(Aring, 0);
        SDM_WAIT_VMCNT0();
      }
      SDM_WAIT_LGKMCNT0();
      SDM_RAW_BARRIER();
      int cm = 0;                   // (6 * c) % 5: ring slot of the chunk's first stage
      if (role) {
        auto stage_end_pc = [&](int t, bool a_fly) {
          if (t + 4 < nst) { if (a_fly) SDM_WAIT_VMCNT(12); else SDM_WAIT_VMCNT(6); }      // everything issued after DMA(t+2) may stay in flight
          else if (t + 3 < nst) SDM_WAIT_VMCNT(3);
          else SDM_WAIT_VMCNT0();
          SDM_WAIT_LGKMCNT0();
          SDM_RAW_BARRIER();
        };
        for (int c = 0; c < nchunks; ++c) {
          const bool more = c + 1 < nchunks;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const int t = c * 6 + dx * 2;
            if (dx == 0 && more) issue_loads_a_asm((c + 1) * 16);
            if (t + 4 < nst) dma_stage_sl(t + 4, mod5(cm + dx * 2 + 4));
            // the other A buffer was last read in chunk c-1: chunk c+1 is transformed, split and written during stage (c, dx=1)
            if (dx == 1 && more) { a_loads_landed(); write_lds_a(Aring + ((c + 1) & 1) * 2 * C::A_BYTES, (c + 1) * 16); }
            stage_end_pc(t, dx == 0 && more);
            if (t + 5 < nst) dma_stage_sl(t + 5, mod5(cm + dx * 2 + 5));
            stage_end_pc(t + 1, dx == 0 && more);
          }
          cm = mod5(cm + 6);
        }
      } else {
        f16x8 fb[2][3][NTL];          // B fragments: [0] the w_hi stage being multiplied / prefetched, [1] the w_lo stage
        f16x8 ah[NR], al[2];          // A_hi rows of the current kernel column (kept for the w_lo stage); A_lo rows rotate
        auto ld_b = [&](int set, int dy, const unsigned char* Bp) {
#pragma unroll
          for (int j = 0; j < NTL; ++j) fb[set][dy][j] = *(const f16x8*)(Bp + bq[j] + dy * (BN * 16));
        };
        auto ld_a = [&](const unsigned char* Ap, int r, int dx) { return *(const f16x8*)(Ap + abase[0] + (r * HPW + dx) * 16); };
        auto rowmm = [&](const f16x8 a, int set, int r) {
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int i = r - dy;
            if (i >= 0 && i < MT) {
#pragma unroll
              for (int j = 0; j < NTL; ++j) acc[i][j] = SDM_MFMA_32x32x16_F16(a, fb[set][dy][j], acc[i][j]);
            }
          }
        };
        {   // operands of the first stage
          const unsigned char* B0 = Bring;
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) ld_b(0, dy, B0);
          ah[0] = ld_a(Aring, 0, 0);
          al[0] = ld_a(Aring + C::A_BYTES, 0, 0);
        }
        for (int c = 0; c < nchunks; ++c) {
          const bool more = c + 1 < nchunks;
          const unsigned char* Ahi = Aring + (c & 1) * 2 * C::A_BYTES;
          const unsigned char* Alo = Ahi + C::A_BYTES;
          const unsigned char* AhiN = Aring + ((c + 1) & 1) * 2 * C::A_BYTES;
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            // ---- stage (c, dx, w_hi): A_hi . w_hi and A_lo . w_hi, row by row; the next row's fragments and the w_lo
            //      fragments of this column are read underneath ----
            const unsigned char* Blo = Bring + mod5(cm + dx * 2 + 1) * SLOT;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              if (r + 1 < NR) { ah[r + 1] = ld_a(Ahi, r + 1, dx); al[(r + 1) & 1] = ld_a(Alo, r + 1, dx); }
              if (r < 3) ld_b(1, r, Blo);
              SDM_SCHED_FENCE();
              rowmm(ah[r], 0, r);
              rowmm(al[r & 1], 0, r);
              SDM_SCHED_FENCE();
            }
            SDM_RAW_BARRIER();
            // ---- stage (c, dx, w_lo): A_hi (registers) . w_lo; the operands of the next w_hi stage are read underneath ----
            const bool last = (dx == 2) && !more;
            const unsigned char* Bn = Bring + mod5(cm + dx * 2 + 2) * SLOT;
            const unsigned char* An = (dx == 2) ? AhiN : Ahi;
            const int dxn = (dx == 2) ? 0 : dx + 1;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
              if (r < 3 && !last) ld_b(0, r, Bn);
              SDM_SCHED_FENCE();
              rowmm(ah[r], 1, r);
              SDM_SCHED_FENCE();
              if (r == 2 && !last) { ah[0] = ld_a(An, 0, dxn); al[0] = ld_a(An + C::A_BYTES, 0, dxn); }      // rows 0..2 of ah are free again
            }
            SDM_RAW_BARRIER();
          }
          cm = mod5(cm + 6);
        }
      }
    } else {
    issue_loads_a(0);
#pragma unroll
    for (int s = 0; s < 3; ++s)
      if (s < nst) dma_stage(s);
    if (GN) __syncthreads();      // gn_tab filled (no DMA data is consumed yet; this only delays the first write)
    write_lds_a(Aring, 0);
    SDM_WAIT_VMCNT0();
    SDM_WAIT_LGKMCNT0();
    SDM_RAW_BARRIER();
    for (int c = 0; c < nchunks; ++c) {
      const bool more = c + 1 < nchunks;
      if (SPLIT) {
        const unsigned char* Ahi = Aring;
        const unsigned char* Alo = Aring + C::A_BYTES;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int s = (c * 3 + dx) * 2;
          if (dx == 0 && more && !ab_a) issue_loads_a_asm((c + 1) * 16);
          if (s + 3 < nst && !ab_dma) dma_stage(s + 3);
          if (!ab_mm) {
            sweep(Ahi, Bring + (s & 3) * SLOT, dx);           // A_hi . w_hi
            sweep(Alo, Bring + (s & 3) * SLOT, dx);           // A_lo . w_hi
          }
          stage_end(s, dx == 0 && more && !ab_a);
          if (s + 4 < nst && !ab_dma) dma_stage(s + 4);
          if (!ab_mm) sweep(Ahi, Bring + ((s + 1) & 3) * SLOT, dx);     // A_hi . w_lo
          stage_end(s + 1, dx == 0 && more && !ab_a);
        }
        if (more && !ab_a) {        // every wave is past the last read of A(c): re-stage the halo tiles for chunk c+1
          a_loads_landed();
          write_lds_a(Aring, (c + 1) * 16);
          SDM_WAIT_LGKMCNT0();
          SDM_RAW_BARRIER();
        }
      } else {
        const unsigned char* Acur = Aring + (c & 1) * C::A_BYTES;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int s = c * 3 + dx;
          if (dx == 0 && more && !ab_a && !ab_ald) issue_loads_a_asm((c + 1) * 16);
          if (s + 3 < nst && !ab_dma) dma_stage(s + 3);
          // next chunk's halo tile into the OTHER A buffer (last read in chunk c-1), BEFORE this stage's sweep: the LDS writes
          // complete under the 24 MFMAs instead of in front of the barrier (measured: the write tail cost 16 % there)
          if (dx == 2 && more && !ab_a && !ab_awr) { a_loads_landed(); write_lds_a(Aring + ((c + 1) & 1) * C::A_BYTES, (c + 1) * 16); }
          if (!ab_mm) sweep(Acur, Bring + (s & 3) * SLOT, dx);
          stage_end(s, dx < 2 && more && !ab_a && !ab_ald);
        }
      }
    }
    }   // !PC
  } else {
  issue_loads(kb);
  if (DB) {                      // double-buffered tiles: chunk 0 is staged up front, ONE barrier per K-chunk afterwards
    if (GN) __syncthreads();     // gn_tab filled
    write_lds(As, Bs, kb);
    __syncthreads();
  }
  for (int c0 = kb; c0 < ke; c0 += KC) {
    if (SPLIT) {
      const bool abl_ld = (p.ablate & 1) != 0, abl_wr = (p.ablate & 2) && c0 > kb, abl_mm = (p.ablate & 4) && c0 > kb;   // bench only
      __syncthreads();            // every wave has finished reading the previous chunk from LDS
      if (!abl_wr) {
        write_lds_a(As, c0);      // A_hi and A_lo halo tiles
        write_lds_b(Bs);          // w_hi taps of this chunk
      }
      __syncthreads();
      if (!abl_ld) issue_loads_b(c0, rsw_lo);  // w_lo taps of this chunk: in flight during the two MFMA passes below
      if (!abl_mm) {
        mfma_chunk(As);                       // A_hi . w_hi
        mfma_chunk(As + C::A_BYTES);          // A_lo . w_hi
      }
      __syncthreads();            // every wave is done with the w_hi taps
      if (!abl_wr) write_lds_b(Bs);
      __syncthreads();
      // next chunk's activations and w_hi taps: in flight during the third pass (issued only now, when the B staging
      // registers are free again: the A + B staging sets together with 128 accumulators are what fits in 256 registers)
      if (c0 + KC < ke && !abl_ld) { issue_loads_a(c0 + KC); issue_loads_b(c0 + KC, rsw); }
      if (!abl_mm) mfma_chunk(As);            // A_hi . w_lo
      continue;
    }
    if (DB) {
      const int cur = ((c0 - kb) / KC) & 1;
      As = smem + cur * C::TILE_BYTES;
      Bs = As + C::A_BYTES;
      if (c0 + KC < ke) issue_loads(c0 + KC);       // in flight during the MFMAs below
    } else {
      __syncthreads();            // every wave has finished reading the previous chunk from LDS
      if (!(p.ablate & 2) || c0 == kb) write_lds(As, Bs, c0);
      __syncthreads();
      if (c0 + KC < ke && !(p.ablate & 1)) issue_loads(c0 + KC);
      if ((p.ablate & 4) && c0 > kb) continue;
    }
    mfma_chunk(As);
    if (DB) {
      if (c0 + KC < ke) {          // the other half was last read one iteration ago, before the previous barrier
        unsigned char* An = smem + ((((c0 - kb) / KC) & 1) ^ 1) * C::TILE_BYTES;
        write_lds(An, An + C::A_BYTES, c0 + KC);
      }
      __syncthreads();
    }
  }

  }   // !DMAB

  // ---- epilogue: per-wave PRIVATE fp32 staging tile (32 x WTN) in LDS -> coalesced row stores ----
  if (!DB && !C::R4) __syncthreads();      // all waves are done with the A/B tiles (DB, F8 3x3: the loop ended with a barrier)
  stamp();                       // tile-end barrier passed
  if (PC && role) return;        // the accumulators live in the consumer waves; no block-wide barrier below this line
  if (FASTEPI && fast_epi) {
    const int l31 = lane & 31, hi = lane >> 5;
    const unsigned int cs4 = (unsigned int)p.Cout_store * 4u;
    const size_t span = (NTAPS == 9) ? (size_t)TH * p.Wout : (size_t)C::BM;
    const sdm_rsrc rso = sdm_make_rsrc((unsigned char*)p.out + px_tile0 * cs4, (unsigned int)(span * cs4));
    const unsigned int vo = (unsigned int)l31 * cs4 + (unsigned int)(p.out_ch_off + ch_lane) * 4u;
    f32x4 b4[NTL][4];
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) b4[j][g] = *(const f32x4*)(bias_tab + wn * WTN + j * 32 + 8 * g + 4 * hi);
#if defined(SDM_CONV_LAB) && !defined(SDM_EMU)
    const bool do_store = !(p.ablate & 8);
#else
    constexpr bool do_store = true;
#endif
    const bool do_stats = p.stats != nullptr;
    // per-channel sums over this wave's 128 pixels for the consumer's GroupNorm: in-lane over the 4 sub-tiles, then over the 16 lanes of each DPP row; the two
    // rows of a lane half meet in a wave-private LDS scratch, from which lane c writes channel c of the partial row - one coalesced store
    float* sc = (float*)(smem + C::SCR_OFF) + wave * 256;            // [2 rows][64 channels][sum, sumsq]
    unsigned int so[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) so[i] = sub_px(i) * cs4;
    // (fast_epi: acc_scale == 1 - F8 layers accumulate in the output's unit -, out_scale == 1, every channel of the tile valid: no predicates, no scaling,
    // no vector-memory wait.)  Channel quad by channel quad, the four sub-tiles of a quad back to back: a quad's statistics are complete after its four
    // stores and are reduced (four DPP stages over eight independent values: no wait states) while the wave would otherwise sit in the store queue - the CU
    // accepts one 16-byte store instruction per ~65 cycles.  (Reduced after the last store, value by value, the statistics cost 3.8 k cycles per tile.)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float t1[4] = {0.0f, 0.0f, 0.0f, 0.0f}, t2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          // the bias is added in place and the quad stored from the accumulator registers themselves (no temporary quad whose rewrite could run into the
          // store-data hazard below)
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) { acc[i][j][4 * g + e] += b4[j][g][e]; v[e] = acc[i][j][4 * g + e]; }
          if (do_store) sdm_buffer_store16(__builtin_bit_cast(u32x4, v), rso, vo + (unsigned int)((j * 32 + 8 * g) * 4), so[i]);
          if (do_stats) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { t1[e] += v[e]; t2[e] += v[e] * v[e]; }
          }
          // the store's data registers stay untouched until here: on gfx950 a v_pk_* that rewrites them two instructions behind a
          // buffer_store_dwordx4 changes what lanes 12-15 of each 16 store (profiles/r03_conv_epilogue_branchfree_ab.txt)
          SDM_PIN_STORE_DATA(v);
        }
        if (do_stats) {
#ifdef SDM_EMU
#pragma unroll
          for (int e = 0; e < 4; ++e) { t1[e] = sdm_sum_row16(t1[e]); t2[e] = sdm_sum_row16(t2[e]); }
#else
#define SDM_STAT_STAGE(ctrl) _Pragma("unroll") for (int e = 0; e < 4; ++e) { t1[e] = SDM_DPP_ADD(t1[e], ctrl); t2[e] = SDM_DPP_ADD(t2[e], ctrl); } SDM_SCHED_FENCE();
          SDM_STAT_STAGE(0xB1) SDM_STAT_STAGE(0x4E) SDM_STAT_STAGE(0x124) SDM_STAT_STAGE(0x128)
#undef SDM_STAT_STAGE
#endif
          if ((lane & 15) == 0) {
            float* dst = sc + (((lane >> 4) & 1) * WTN + j * 32 + 8 * g + 4 * hi) * 2;
            f32x4 o0, o1;
            o0[0] = t1[0]; o0[1] = t2[0]; o0[2] = t1[1]; o0[3] = t2[1];
            o1[0] = t1[2]; o1[1] = t2[2]; o1[2] = t1[3]; o1[3] = t2[3];
            *(f32x4*)dst = o0;
            *(f32x4*)(dst + 4) = o1;
          }
        }
        if (g & 1) stamp();        // (traced builds) a quarter of the tile issued
      }
#if !defined(SDM_EMU) && defined(__HIP_DEVICE_COMPILE__)      // (device pass only: a 512-bit "v" operand is not valid x86 inline asm)
    // the biased accumulators stay live to this point, so that hipcc really keeps every store's data in its own registers
    asm volatile("" :: "v"(acc[0][0]), "v"(acc[0][NTL - 1]), "v"(acc[MT > 1 ? 1 : 0][0]), "v"(acc[MT > 1 ? 1 : 0][NTL - 1]),
                 "v"(acc[MT > 2 ? 2 : 0][0]), "v"(acc[MT > 2 ? 2 : 0][NTL - 1]), "v"(acc[MT > 3 ? 3 : 0][0]), "v"(acc[MT > 3 ? 3 : 0][NTL - 1]));
    static_assert(!FASTEPI || (MT <= 4 && NTL <= 2), "operand list above");
#endif
    if (do_stats) {
      SDM_WAVE_SYNC();
      if (lane < WTN) {
        const f32x2 a = *(const f32x2*)(sc + lane * 2), b = *(const f32x2*)(sc + (WTN + lane) * 2);
        f32x2 o2;
        o2[0] = a[0] + b[0]; o2[1] = a[1] + b[1];
        const size_t prow = (size_t)img * (p.tiles_m * WM) + (size_t)mt * WM + wm;
        *(f32x2*)(p.stats + (prow * p.Cout_store + p.out_ch_off + n0 + wn * WTN + lane) * 2) = o2;
      }
      SDM_WAVE_SYNC();
    }
    stamp();                     // epilogue issued
    return;
  }
  float* stg = (float*)(smem + (F8 ? C::SCR_OFF : 0)) + wave * (32 * WTN);      // F8: the epilogue scratch (ConvCfg: the next tile's prologue does not touch it)
  // accumulators of sub-tile i -> the wave's [32 pixels][WTN channels] staging tile.  F8 kernels hold [channel][pixel]: a register quad
  // is 4 consecutive channels of pixel (lane & 31) - one ds_write_b128 into 16-byte block ((channel / 4) ^ (pixel & 15)) of the
  // pixel's row (the XOR keeps 8 consecutive lanes = 8 rows on distinct banks without padding the tile; readers apply it again)
  auto stage_acc = [&](int i) {
    if (F8) {
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int blk = (j * 8 + 2 * g + (lane >> 5)) ^ (lane & 15);
          f32x4 q;
#pragma unroll
          for (int e = 0; e < 4; ++e) q[e] = acc[i][j][4 * g + e];
          *(f32x4*)(stg + (lane & 31) * WTN + blk * 4) = q;
        }
    } else {
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          stg[row * WTN + j * 32 + (lane & 31)] = acc[i][j][r];
        }
    }
  };
  auto stg_at = [&](int row, int col) { return stg + row * WTN + (F8 ? (((col >> 2) ^ (row & 15)) << 2) : col); };      // col % 4 == 0
  const bool geglu = (p.epi == 1);
  constexpr int LPR = WTN / 4;                    // lanes per output row (linear epilogue: 4 channels per lane)
  const float* bias = p.bias;
  if (bias && p.bias_sel) bias += (size_t)p.bias_sel[img] * p.Cout_pad;
  const int colbase = n0 + wn * WTN;
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  const bool colok = true;
  int oc;
  // pixel of row `row` of M-tile i of this wave -> (valid, linear pixel index inside the image / row block)
  auto pix_of = [&](int i, int row, long& opix_local) {
    const int m = wm * WTM + i * 32 + row;
    if (NTAPS == 9) {
      const int oy = oy0 + m / TW, ox = ox0 + m % TW;
      opix_local = (long)oy * p.Wout + ox;
      return (oy < p.Hout) && (ox < p.Wout);
    }
    opix_local = m;
    return (m0 + m) < m_end;
  };
  const size_t opix_base = (NTAPS == 9) ? (size_t)img * p.Hout * p.Wout : (size_t)m0;   // block-uniform
  if (!geglu) {
    // ---------------- linear epilogue: +bias, *scale, +residual, store, statistics ----------------
    constexpr int RPP = 64 / LPR, NPASS = 32 / RPP;
    const int lc = (lane % LPR) * 4;
    oc = colbase + lc;
    f32x4 bu = {0.f, 0.f, 0.f, 0.f};
    if (bias && oc < p.Cout_pad) bu = *(const f32x4*)(bias + oc);
    // residual through a per-block buffer descriptor: all loads of a 32-row tile are issued up front, unconditionally
    // (invalid rows/columns get an out-of-range offset -> 0), instead of one dependent global round trip per row pass
    const unsigned int res_es = p.res_f32 ? 4u : 2u;
    const long res_row0 = (NTAPS == 9) ? (long)oy0 * p.Wout : 0L;          // 3x3: descriptor from the tile's first output row (small offsets)
    const size_t res_span = (NTAPS == 9) ? (size_t)((p.Hout - oy0) < TH ? (p.Hout - oy0) : TH) * p.Wout
                                         : (size_t)(((m_end - m0) < (long)C::BM) ? (m_end - m0) : (long)C::BM);
    const sdm_rsrc rsr = sdm_make_rsrc(p.res ? (const unsigned char*)p.res + (opix_base + (size_t)res_row0) * p.res_C * res_es : (const unsigned char*)p.out,
                                       p.res ? (unsigned int)(res_span * p.res_C * res_es) : 0u);
    const bool res_late = p.res != nullptr && !(FASTEPI && res_init);      // not already in the accumulators
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      u32x4 rr[NPASS];
      if (res_late) {
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          long lp;
          const bool valid = pix_of(i, pass * RPP + lane / LPR, lp);
          const unsigned int off = (valid && oc < p.Cout_valid) ? (unsigned int)(((size_t)(lp - res_row0) * p.res_C + oc) * res_es) : SDM_BUF_INVALID;
          if (p.res_f32) {
            rr[pass] = sdm_buffer_load16(rsr, off, 0);
          } else {                                              // fp16 residual: 4 channels = 8 bytes
            const u32x2 h2 = sdm_buffer_load8(rsr, off, 0);
            rr[pass][0] = h2[0]; rr[pass][1] = h2[1]; rr[pass][2] = 0u; rr[pass][3] = 0u;
          }
        }
      }
      stage_acc(i);
      SDM_WAVE_SYNC();
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int row = pass * RPP + lane / LPR;
        long lp;
        const bool valid = pix_of(i, row, lp);
        const f32x4 t = *(const f32x4*)stg_at(row, lc);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ((SPLIT ? t[e] * p.acc_scale : t[e]) + bu[e]) * p.out_scale;
        if (res_late) {
          if (p.res_f32) {
            const f32x4 r4 = __builtin_bit_cast(f32x4, rr[pass]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += r4[e];
          } else {
            const f16x8 r8 = __builtin_bit_cast(f16x8, rr[pass]);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += (float)r8[e];
          }
        }
        if (valid && oc < p.Cout_valid && !(p.ablate & 8)) {
          const size_t oidx = (opix_base + (size_t)lp) * p.Cout_store + p.out_ch_off + oc;
          if (p.out_f32 == 1) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = v[e];
            *(f32x4*)((float*)p.out + (ksp ? (size_t)blockIdx.y * p.ks_stride : (size_t)0) + oidx) = o;
          } else if (p.out_f32 == 2) {                  // hi | lo fp16 planes (operands of the split-precision attention)
            f16x4 oh, ol;
#pragma unroll
            for (int e = 0; e < 4; ++e) { oh[e] = (half_t)v[e]; ol[e] = (half_t)(v[e] - (float)oh[e]); }
            *(f16x4*)((half_t*)p.out + oidx) = oh;
            if (oc < p.lo_cols) *(f16x4*)((half_t*)p.out + p.out_lo_off + oidx) = ol;
          } else if (p.out_f32 == 3) {                  // hi fp16 plane | e5m2 pair plane (same bytes, same addresses as the lo plane)
            f16x4 oh;
            float xx[4], xl[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float y = fminf(fmaxf(v[e], -57344.0f), 57344.0f);      // the largest finite e5m2
              oh[e] = (half_t)y; xx[e] = y; xl[e] = (y - (float)oh[e]) * 2048.0f;
            }
            *(f16x4*)((half_t*)p.out + oidx) = oh;
            if (oc < p.lo_cols) {
              int a = SDM_CVT_PK_BF8(xx[0], xx[1], 0, false), b = SDM_CVT_PK_BF8(xl[0], xl[1], 0, false);
              a = SDM_CVT_PK_BF8(xx[2], xx[3], a, true); b = SDM_CVT_PK_BF8(xl[2], xl[3], b, true);
              u32x2 pr;
              pr[0] = (unsigned int)a; pr[1] = (unsigned int)b;
              *(u32x2*)((half_t*)p.out + p.out_lo_off + oidx) = pr;
            }
          } else {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)v[e];
            *(f16x4*)((half_t*)p.out + oidx) = o;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (float)o[e];      // statistics of what the next layer will actually read
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
        }
      }
      SDM_WAVE_SYNC();
    }
  } else {
    // ---------------- GEGLU epilogue: the wave's columns are [u32|g32] pairs -> WTN/2 outputs per row, WTN/8 lanes per
    //                  row (all 64 lanes busy), u and g read as two 16-B vectors; no residual / statistics ----------------
    constexpr int GL = LPR / 2, RPP = 64 / GL, NPASS = 32 / RPP;
    const int lc = (lane % GL) * 4;
    const int ucol = (lc >> 5) * 64 + (lc & 31);
    oc = colbase / 2 + lc;
    f32x4 bu = {0.f, 0.f, 0.f, 0.f}, bg = {0.f, 0.f, 0.f, 0.f};
    if (bias && colbase + ucol + 32 < p.Cout_pad + 4) { bu = *(const f32x4*)(bias + colbase + ucol); bg = *(const f32x4*)(bias + colbase + ucol + 32); }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      stage_acc(i);
      SDM_WAVE_SYNC();
#pragma unroll
      for (int pass = 0; pass < NPASS; ++pass) {
        const int row = pass * RPP + lane / GL;
        long lp;
        const bool valid = pix_of(i, row, lp);
        const f32x4 t = *(const f32x4*)stg_at(row, ucol);
        const f32x4 g4 = *(const f32x4*)stg_at(row, ucol + 32);
        f32x4 o4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float u = (SPLIT ? t[e] * p.acc_scale : t[e]) + bu[e], g = (SPLIT ? g4[e] * p.acc_scale : g4[e]) + bg[e];
          o4[e] = (u * sdm_gelu_erf(g)) * p.out_scale;
        }
        if (valid && oc < p.Cout_valid && !(p.ablate & 8)) {
          const size_t oidx = (opix_base + (size_t)lp) * p.Cout_store + p.out_ch_off + oc;
          if (p.out_f32) {
            *(f32x4*)((float*)p.out + oidx) = o4;
          } else {
            f16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)o4[e];
            *(f16x4*)((half_t*)p.out + oidx) = o;
          }
        }
      }
      SDM_WAVE_SYNC();
    }
  }
  // ---- fused GroupNorm statistics of the consumer: one partial row per (tile, wave-row); plain stores, no atomics ----
  if (p.stats) {
    // lanes {l, l+LPR, l+2*LPR, ...} own the same 4 channels: fold them
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int m = LPR; m < 64; m <<= 1) { ssum[e] += __shfl_xor(ssum[e], m); ssq[e] += __shfl_xor(ssq[e], m); }
    }
    if (lane < LPR && colok && oc < p.Cout_valid) {
      const size_t prow = (size_t)img * (p.tiles_m * WM) + (size_t)mt * WM + wm;
      float* st = p.stats + (prow * p.Cout_store + p.out_ch_off + oc) * 2;
      f32x4 o0, o1;
      o0[0] = ssum[0]; o0[1] = ssq[0]; o0[2] = ssum[1]; o0[3] = ssq[1];
      o1[0] = ssum[2]; o1[1] = ssq[2]; o1[2] = ssum[3]; o1[3] = ssq[3];
      *(f32x4*)st = o0;
      *(f32x4*)(st + 4) = o1;
    }
  }
  };   // run_tile
  if (F8) {
    // tiles of a constant-input region (ConvParams::tile_flag) are left to const_tile_fill_kernel.  Block-uniform, evaluated identically by both
    // roles; a tile in front of a skipped one does not prefetch across it, and the bias-table parity counts the tiles that actually ran
    // bit k of `skip`: the block's k-th tile is left out.  All flags are fetched up front (independent scalar loads: one round trip per block)
    // Padding blocks of the grid rounding (tile_decode fails) count as left out as well: a tile's bias-table slot is its rank among the tiles that
    // really run, so a valid / padding / valid sequence (band order with tiles_m % (8 * band) != 0) keeps alternating slots
    unsigned int skip = 0;
    {
      int fl[9], ml9[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) { fl[k] = 0; ml9[k] = 0; }
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        if (k >= p.tpb) continue;
        const int v = (int)blockIdx.x + k * (int)gridDim.x;
        int nt;
        if (v < p.vgrid && tile_decode(v, ml9[k], nt)) { if (NTAPS == 9 && p.tile_flag) fl[k] = p.tile_flag[ml9[k]]; }
        else skip |= 1u << k;
      }
#pragma unroll
      for (int k = 0; k < 9; ++k)
        if (fl[k]) { const int im = ml9[k] / p.tiles_m; if (p.tile_rep[im * 8 + fl[k]] != ml9[k] - im * p.tiles_m) skip |= 1u << k; }
    }
    skip = (unsigned int)SDM_UNIFORM_I((int)skip);
    if (SDM_UNIFORM_I((int)threadIdx.x / NT)) {
      int par = 0;
      for (int k = 0; k < p.tpb; ++k) {
        const int v = (int)blockIdx.x + k * (int)gridDim.x;
        if (v < p.vgrid && !((skip >> k) & 1u)) {
          run_tile(std::integral_constant<int, 1>{}, v, par, (k + 1 < p.tpb) && (v + (int)gridDim.x < p.vgrid) && !((skip >> (k + 1)) & 1u));
          par ^= 1;
        }
      }
    } else {
      int par = 0;
      for (int k = 0; k < p.tpb; ++k) {
        const int v = (int)blockIdx.x + k * (int)gridDim.x;
        if (v < p.vgrid && !((skip >> k) & 1u)) {
          run_tile(std::integral_constant<int, 0>{}, v, par, (k + 1 < p.tpb) && (v + (int)gridDim.x < p.vgrid) && !((skip >> (k + 1)) & 1u));
          par ^= 1;
        }
      }
    }
#if defined(SDM_CONV_TRACE) && !defined(SDM_EMU)
    if (NTAPS == 9 && p.trace && (int)blockIdx.x >= p.trace_b0 && (int)blockIdx.x < p.trace_b0 + 16 && (threadIdx.x & (NT - 1)) == 0) {      // the parked stamps of this role -> global memory, once
      const int r = (int)threadIdx.x / NT;
      const unsigned int* tb = (const unsigned int*)(smem + C::TRACE_OFF) + r * 64;
      unsigned int* dst = p.trace + ((size_t)((int)blockIdx.x - p.trace_b0) * 2 + r) * 384;
      for (int i = 0; i < tr_n && i < 383; ++i) dst[i] = tb[i];
      dst[383] = (unsigned int)tr_n;
    }
#endif
  } else {
    run_tile(std::integral_constant<int, -1>{}, (int)blockIdx.x, 0, false);
  }
}

// ---- split-K epilogue.  The register-staged kernels above, launched with ConvParams::ksplit > 1, leave ksplit slabs of fp32 partial sums
//      [rows][ws_C]; this kernel finishes the layer exactly as their linear epilogue does: (sum + bias) * out_scale + residual, fp32 / fp16 store,
//      and the fused GroupNorm statistics of the consumer in the same partial-row layout [N][srows][Cout_store][2] with srows = row blocks per
//      image.  The slabs are added in slab order: the result does not depend on scheduling.
//      grid (N * blocks_per_img, ceil(Cout_valid / 64)), 256 threads = 16 rows x 16 lanes of 4 channels; a block walks `rb` rows of ONE image. ----
struct SplitKReduceParams {
  const float* ws; int ksplit; size_t ks_stride; int ws_C;
  int rows_per_img, rb, blocks_per_img;
  const float* bias; const int* bias_sel; int Cout_pad;
  const void* res; int res_f32, res_C;
  float out_scale;
  void* out; int out_f32, Cout_store, out_ch_off, Cout_valid;
  float* stats;
};

__global__ __launch_bounds__(256) void splitk_reduce_kernel(SplitKReduceParams q) {
  SDM_SHARED float red[16][64][2];
  const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
  const int img = (int)blockIdx.x / q.blocks_per_img, rbi = (int)blockIdx.x % q.blocks_per_img;
  const int oc = (int)blockIdx.y * 64 + l16 * 4;
  const bool cok = oc < q.Cout_valid;
  const long row0 = (long)img * q.rows_per_img + (long)rbi * q.rb;
  long row_end = row0 + q.rb;
  if (row_end > (long)(img + 1) * q.rows_per_img) row_end = (long)(img + 1) * q.rows_per_img;
  f32x4 b = {0.f, 0.f, 0.f, 0.f};
  if (q.bias && cok) b = *(const f32x4*)(q.bias + (q.bias_sel ? (size_t)q.bias_sel[img] * q.Cout_pad : (size_t)0) + oc);
  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    for (long r = row0 + rg; r < row_end; r += 16) {
      const float* src = q.ws + (size_t)r * q.ws_C + oc;
      f32x4 a = *(const f32x4*)src;
      for (int s = 1; s < q.ksplit; ++s) {
        const f32x4 t = *(const f32x4*)(src + (size_t)s * q.ks_stride);
#pragma unroll
        for (int e = 0; e < 4; ++e) a[e] += t[e];
      }
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (a[e] + b[e]) * q.out_scale;
      if (q.res) {
        if (q.res_f32) {
          const f32x4 r4 = *(const f32x4*)((const float*)q.res + (size_t)r * q.res_C + oc);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += r4[e];
        } else {
          const f16x4 r4 = *(const f16x4*)((const half_t*)q.res + (size_t)r * q.res_C + oc);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += (float)r4[e];
        }
      }
      const size_t oidx = (size_t)r * q.Cout_store + q.out_ch_off + oc;
      if (q.out_f32) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = v[e];
        *(f32x4*)((float*)q.out + oidx) = o;
      } else {
        f16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (half_t)v[e]; v[e] = (float)o[e]; }      // statistics of what the next layer will actually read
        *(f16x4*)((half_t*)q.out + oidx) = o;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) { ssum[e] += v[e]; ssq[e] += v[e] * v[e]; }
    }
  }
  if (q.stats) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[rg][l16 * 4 + e][0] = ssum[e]; red[rg][l16 * 4 + e][1] = ssq[e]; }
    __syncthreads();
    if (tid < 128) {
      const int c = tid >> 1, w = tid & 1;
      float t = 0.0f;
#pragma unroll
      for (int g = 0; g < 16; ++g) t += red[g][c][w];
      const int ch = (int)blockIdx.y * 64 + c;
      if (ch < q.Cout_valid) q.stats[(((size_t)img * q.blocks_per_img + rbi) * q.Cout_store + q.out_ch_off + ch) * 2 + w] = t;
    }
  }
}

// Pack an OIHW (or [O][I]) fp32 weight into the fp16 K16 layout [Cin_pad/16][ntaps][Cout_pad][16].
//  * ci_off : the source input channels land at packed channels [ci_off, ci_off+I) (others stay zero)
//  * co_off : the source rows land at packed rows [co_off, co_off+O) (fused q|k|v GEMMs)
//  * geglu  : rows are re-ordered so that every 64 GEMM columns hold [u_c..u_c+31 | g_c..g_c+31]
//             (diffusers GEGLU: proj -> chunk(2) = (u, g); out = u * gelu(g)); O = 2*half rows.
// Only the rows [co_off, co_off+O) (or all rows for geglu) are written; the arena is zero-initialised.
SDM_DEV_INLINE int pack_src_row(int co, int O, int co_off, int geglu) {
  if (geglu) {
    const int half = O / 2, grp = co / 64, j = co % 64;
    if (grp * 32 + (j & 31) >= half) return -1;
    return (j < 32) ? grp * 32 + j : half + grp * 32 + (j - 32);
  }
  const int o = co - co_off;
  return (o >= 0 && o < O) ? o : -1;
}

// wp_lo != null (precise mode): the scaled weight v is stored as the fp16 pair hi = fp16(v), lo = fp16(v - hi) in two tensors of
// the same layout (`scale` then carries the power-of-two pre-scale that keeps the low parts in the fp16 normal range).
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, half_t* __restrict__ wp, int O, int I, int ntaps, int Cin_pad,
                                        int Cout_pad, int ci_off, int co_off, int geglu, float scale, half_t* __restrict__ wp_lo) {
  const size_t total = (size_t)Cin_pad * ntaps * Cout_pad;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k = idx % 16;
    size_t t = idx / 16;
    const int co = t % Cout_pad; t /= Cout_pad;
    const int tap = t % ntaps;
    const int chunk = t / ntaps;
    const int ci = chunk * 16 + k - ci_off;
    const int o = pack_src_row(co, O, co_off, geglu);
    if (o < 0) continue;
    float v = 0.0f;
    if (ci >= 0 && ci < I) v = w[((size_t)o * I + ci) * ntaps + tap];
    const float vs = v * scale;
    const half_t hi = (half_t)vs;
    wp[idx] = hi;
    if (wp_lo) wp_lo[idx] = (half_t)(vs - (float)hi);
  }
}

// ---- derived weight layouts.  The canonical form of every conv / linear weight is the K16 tensor (pack_conv_weight_kernel: fp16 hi,
//      plus lo for the split-precision layers); it is what checkpoints are packed into, what sdm_export_weight_blob hands to
//      other ranks / devices, and what the two kernels below turn into the layouts the DMA-weight kernels stream - on every
//      engine, by the same arithmetic, when the weights are finalised (bit-identical across ranks). ----

// largest |value| of a packed fp16 tensor -> *out (float bits; non-negative floats order like unsigned integers)
__global__ void absmax_f16_kernel(const half_t* __restrict__ w, size_t n, unsigned int* __restrict__ out) {
  float m = 0.0f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float v = fabsf((float)w[i]);
    if (v == v && v > m) m = v;
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
  if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned int, m));
}

// 3x3 weights in the stage order of the DMAB kernels: [Cin_pad/16][dx][part][k-half][dy][Cout_pad][8], part = hi (| lo when
// nparts == 2).  One stage = (chunk, dx, part) = 2 x 3 x Cout_pad rows of 16 bytes; the BN rows of one (k-half, dy) of an
// output-channel tile are contiguous, so a DMA instruction copies 64 of them (1 KB) straight into the LDS half-plane image.
// A pure permutation of the K16 tensors.
__global__ void derive_conv_weight_dma_kernel(const half_t* __restrict__ k_hi, const half_t* __restrict__ k_lo, half_t* __restrict__ wd, int Cin_pad,
                                              int Cout_pad, int nparts) {
  const size_t total = (size_t)Cin_pad * 9 * Cout_pad * nparts;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int k8 = idx % 8;
    size_t t = idx / 8;
    const int co = t % Cout_pad; t /= Cout_pad;
    const int dy = t % 3; t /= 3;
    const int half = t % 2; t /= 2;
    const int part = t % nparts; t /= nparts;
    const int dx = t % 3;
    const int chunk = t / 3;
    const size_t src = (((size_t)chunk * 9 + dy * 3 + dx) * Cout_pad + co) * 16 + half * 8 + k8;
    wd[idx] = part == 0 ? k_hi[src] : k_lo[src];
  }
}

// fp8-residual layout of a 3x3 weight (F8 kernels), per 32-channel chunk and kernel column dx four 12 KB-per-128-channels units of
// [plane][dy][Cout_pad][16 B]:  unit 0 / 1 = fp16 high parts of channels 0-15 / 16-31 (plane = 8-channel half, 8 halfs per row),
// unit 2 = e4m3(w * s8) (plane = channels 0-15 | 16-31, one byte per channel), unit 3 = e4m3((w - hi) * s8 * 2^11), hi = fp16(w).
// Same number of bytes as the hi | lo fp16 pair (4 per weight).  ntaps == 1 (Linear / 1x1): the same without the dx / dy
// dimensions - [chunk32][unit][plane][Cout_pad][16 B].
//   * The fp16 high parts are stored UNSCALED (K16 keeps w * 2^w_exp; inv_s = 2^-w_exp), so the F8 kernels accumulate in the
//     output's own unit (acc_scale = 1: the residual can enter as the accumulators' initial value).
//   * s8 = 2^e8 is the layer's own fp8 scale, the largest power of two with max|w| * s8 <= 448 (chosen by the host from
//     absmax_f16_kernel): no weight saturates, whatever the checkpoint; the MFMA's E8M0 operand scale undoes it.
__global__ void derive_conv_weight_f8_kernel(const half_t* __restrict__ k_hi, const half_t* __restrict__ k_lo, unsigned char* __restrict__ wd, int Cin_pad,
                                             int Cout_pad, int ntaps, float inv_s, float s8) {
  const int nd = ntaps == 9 ? 3 : 1;
  const size_t total = (size_t)(Cin_pad / 32) * nd * 4 * 2 * nd * Cout_pad;         // 16-byte rows
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    size_t t = idx;
    const int co = t % Cout_pad; t /= Cout_pad;
    const int dy = t % nd; t /= nd;
    const int pl = t % 2; t /= 2;
    const int u = t % 4; t /= 4;
    const int dx = t % nd;
    const int chunk = (int)(t / nd);
    const int tap = (ntaps == 9) ? dy * 3 + dx : 0;
    auto src = [&](int c) { return (((size_t)(c / 16) * ntaps + tap) * Cout_pad + co) * 16 + (c % 16); };
    unsigned char* dst = wd + idx * 16;
    if (u < 2) {
      f16x8 h;
#pragma unroll
      for (int e = 0; e < 8; ++e) h[e] = (half_t)((float)k_hi[src(chunk * 32 + u * 16 + pl * 8 + e)] * inv_s);
      *(f16x8*)dst = h;
    } else {
      float v[16];
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const size_t si = src(chunk * 32 + pl * 16 + b);
        const float hi = (float)k_hi[si], w = (hi + (float)k_lo[si]) * inv_s;      // the weight to 22 bits
        const float hp = (float)(half_t)(hi * inv_s);                              // the high part the kernel multiplies (unit 0 / 1)
        const float r = ((u == 2) ? w : (w - hp) * 2048.0f) * s8;
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

__global__ void pack_bias_kernel(const float* __restrict__ b, float* __restrict__ bp, int O, int Cout_pad, int co_off, int geglu) {
  const int co = blockIdx.x * blockDim.x + threadIdx.x;
  if (co >= Cout_pad) return;
  const int o = pack_src_row(co, O, co_off, geglu);
  if (o >= 0) bp[co] = b[o];
}
