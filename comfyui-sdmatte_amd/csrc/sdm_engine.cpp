// sdm_engine.cpp - MI355X-native SDMatte engine: model graph, weight registry/packing, activation arena and
// the C ABI of include/sdmatte.h.  Compiled with `hipcc -x hip --offload-arch=gfx950` (product), or with
// -DSDM_EMU against tests/emu/hip_emu.h (kernel-debug build used by tests only; never shipped).
//
// The graph executed by run_model() restates SDMatte.forward (/root/reference/src/modeling/SDMatte/
// meta_arch.py:127-261) and CustomUNet.forward (/root/reference/src/utils/replace.py:379-549) over the
// diffusers SD-2.1 blocks they instantiate, re-designed for gfx950:
//   * NHWC activations, fp16 MFMA operands, fp32 accumulation/statistics, fp32 residual stream;
//   * the rgb and trimap VAE encodes run as ONE batch of 2B images (same weights);
//   * q|k|v and cross k|v projections are single fused GEMMs; GEGLU is a GEMM epilogue;
//   * time/opacity/bbox embeddings are constants per (is_trans, coords): computed once on the host and
//     folded into every ResBlock conv1 bias;
//   * the dead CLIP text branch (meta_arch.py:220-234, never consumed: replace.py:414-416) is not built.
#include "sdm_common.h"
#include "k_conv.h"
#include "k_gemm.h"
#include "k_norm.h"
#include "k_attn.h"
#include "k_misc.h"
#include "../../include/sdmatte.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <unordered_map>
#include <vector>

// ------------------------------------------------------------------------------------------------
// device runtime shim
// ------------------------------------------------------------------------------------------------
#ifdef SDM_EMU
#include <chrono>
static int dev_malloc(void** p, size_t n) { *p = aligned_alloc(256, ((n + 255) / 256) * 256 + 256); return *p ? 0 : -1; }
static void dev_free(void* p) { free(p); }
static int dev_memcpy_h2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
static int dev_memcpy_d2h(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
static int dev_memcpy_d2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
static int dev_memset(void* d, int v, size_t n, void*) { memset(d, v, n); return 0; }
static int dev_sync(void*) { return 0; }
static const char* dev_errstr(int) { return "emu"; }
#define SDM_SET_SMEM(kernel, bytes) ((void)0)
#else
static int dev_malloc(void** p, size_t n) { return (int)hipMalloc(p, n); }
static void dev_free(void* p) { (void)hipFree(p); }
static int dev_memcpy_h2d(void* d, const void* s, size_t n, void* st) { return (int)hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st); }
static int dev_memcpy_d2h(void* d, const void* s, size_t n, void* st) { return (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st); }
static int dev_memcpy_d2d(void* d, const void* s, size_t n, void* st) { return (int)hipMemcpyAsync(d, s, n, hipMemcpyDeviceToDevice, (hipStream_t)st); }
static int dev_memset(void* d, int v, size_t n, void* st) { return (int)hipMemsetAsync(d, v, n, (hipStream_t)st); }
static int dev_sync(void* st) { return (int)hipStreamSynchronize((hipStream_t)st); }
static const char* dev_errstr(int e) { return hipGetErrorString((hipError_t)e); }
// Kernels with more than 48 KB of dynamic LDS need the attribute once PER DEVICE (one process may drive one engine per GPU,
// INTEGRATION.md 3): a bit per device id, set with relaxed atomics (setting it twice is harmless).
#define SDM_SET_SMEM(kernel, bytes)                                                                              \
  do {                                                                                                           \
    if ((bytes) > 48 * 1024) {                                                                                   \
      static std::atomic<unsigned long long> done_{0ull};                                                        \
      int dev_ = 0;                                                                                              \
      (void)hipGetDevice(&dev_);                                                                                 \
      const unsigned long long bit_ = 1ull << (dev_ & 63);                                                       \
      if (!(done_.load(std::memory_order_relaxed) & bit_)) {                                                     \
        const hipError_t r_ = hipFuncSetAttribute((const void*)(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
        if (r_ != hipSuccess) fprintf(stderr, "[sdmatte] hipFuncSetAttribute(%d bytes of LDS) failed: %s\n", (int)(bytes), hipGetErrorString(r_)); \
        else done_.fetch_or(bit_, std::memory_order_relaxed);                                                    \
      }                                                                                                          \
    }                                                                                                            \
  } while (0)
#endif

static inline int rup(int a, int b) { return ((a + b - 1) / b) * b; }
// every entry point that touches the GPU selects the engine's device first: one process may own one engine per GPU
#ifdef SDM_EMU
static inline void dev_use(int) {}
#else
static inline void dev_use(int device) {
  const hipError_t r = hipSetDevice(device);
  if (r != hipSuccess) fprintf(stderr, "[sdmatte] hipSetDevice(%d) failed: %s\n", device, hipGetErrorString(r));   // the next HIP call of the entry point reports it through its status
}
#endif
static inline size_t rupz(size_t a, size_t b) { return ((a + b - 1) / b) * b; }

// ------------------------------------------------------------------------------------------------
// kernel-selection options.  The product reads NO environment variable on any path: every choice below has ONE default, and the only
// way to change one is the C ABI (sdm_set_option; tests and the A/B tools under tools/ use it).  sdm_kernel_counts reports which
// variants actually ran, so that a test can assert the kernel it meant to check.
// ------------------------------------------------------------------------------------------------
struct OptEntry { const char* name; int value; int def; const char* what; };
static OptEntry g_opts[] = {
  {"conv_f8", 1, 1, "residual terms of the wide split-precision 3x3 convs on fp8 MFMAs (read when a model is built)"},
  {"gemm_f8", 1, 1, "the same for Linear / 1x1 layers with K >= gemm_f8_min_k (read when a model is built)"},
  {"gemm_f8_min_k", 1024, 1024, "smallest K of a GEMM that takes the 8-wave fp8-residual kernel"},
  {"gemm_p3", 1, 1, "transformer-block Linear layers on the plane-fed GEMM (k_gemm.h): pre-split operand planes from LayerNorm / GroupNorm / attention / GEGLU, LDS-DMA only (the W3 weight copies are built when a model is built)"},
  {"gemm_p3_tile", 0, 0, "row tile of the plane-fed GEMM: 0 by shape, 256 / 128 / 64 forced"},
  {"gemm_p3_attn", 1, 1, "d=64 attention cores write the operand planes of to_out themselves (0: fp32 result + one conversion pass)"},
  {"gemm_p3_persist", 1, 1, "plane-fed GEMM: persistent blocks that prefetch the next tile's first chunk underneath the epilogue (0: one block per tile, n > 1: a grid of n blocks - tests)"},
  {"gemm_p3_stages", 0, 0, "LDS stages of the plane-fed GEMM: 0 default, n = lab forms (fp32 / GEGLU epilogues)"},
  {"gemm_p3_ablate", 0, 0, "bench only: 1 no MFMAs, 2 no DMAs behind the prologue, 4 no epilogue (sdm_bench_gemm_p3)"},
  {"conv_epi", 4, 4, "F8 kernels' epilogue: 4 register-direct stores + residual as accumulator init, 3 residual init only, 0 LDS-staged"},
  {"conv_xtile", 1, 1, "F8 3x3: cross-tile prefetch by the producer waves"},
  {"conv_f8_tpb", 0, 0, "F8: tiles per block (0 = by queue depth)"},
  {"conv_dma", 1, 1, "3x3 stride-1 256x128 tile: weights by LDS-DMA"},
  {"conv_dma_all", 0, 0, "keep a stage-ordered weight copy for every wide 3x3 layer, not only the split-precision ones (read when a model is built)"},
  {"conv_pc", -1, -1, "producer / consumer form of the split-precision DMA kernel: -1 by channel count, 0 off, 1 on"},
  {"conv_pc_min_cin", 256, 256, "conv_pc = -1: smallest Cin that takes the producer / consumer form"},
  {"conv_db", 0, 0, "512x128 double-buffered tile where the queue is deep"},
  {"trimap_skip", 1, 1, "VAE encoder, trimap images: output tiles inside a constant region of the trimap are not multiplied (k_misc.h cmask_*; exact). 0 = every tile"},
  {"conv_band_rows", 0, 0, "tile rows per XCD band of the convs that leave constant tiles out (0 = by image height)"},
  {"trimap_skip_min_rows", 512, 512, "smallest output height at which constant tiles are left out (measured at 1024^2: the 1024- and 512-row levels gain 3.7 + 1.6 ms, the 256-row level loses 0.7: too few of its 32 x 128-pixel tiles lie inside one region)"},
  {"conv_splitk", -1, -1, "split-K of the register-staged conv / GEMM kernels: -1 by shape (few tiles, long K), 0 off, n >= 2 forced where the shape allows"},
  {"force_cfg0", 0, 0, "always the 256x128 tile for 3x3 stride 1 (tests: fused GroupNorm at tiny sizes)"},
  {"no_gn_fuse", 0, 0, "never fuse the GroupNorm apply into the consuming conv"},
  {"split_lds_pad", 0, 0, "extra dynamic LDS of the register-staged split kernels (forces one block per CU)"},
  {"attn_f8", 1, 1, "Q.K^T residual terms of the d=64 split-precision attention on fp8 MFMAs"},
  {"attn_dense", 0, 0, "walk every key tile of the trimap-biased self-attention"},
  {"attn_pv_split", 0, 0, "residual terms of P.V too (fully split attention; tests)"},
  {"attn_nw", 0, 0, "waves per d=64 attention block: 0 by launch size, 4, 8"},
  {"attn_pipe", 1, 1, "8-wave split-precision d=64 attention: two-tile software pipeline"},
  {"attn_pipe4", 1, 1, "the same pipeline for the 4-wave launches (two K / three V^T buffers)"},
  {"attn_pp", 1, 1, "split-precision d=64 attention with fp32 output as a ping-pong of the block's wave halves (attn_d64_pp_kernel, 256 query rows per block): 0 off, 1 on, 2 on without the static priority of the younger half, 3 on with per-segment priority flips"},
  {"attn_pp_min_blocks", 128, 128, "attn_pp: launches with fewer 256-row blocks than this keep the 4-wave pipelines (0 in tests: the ping-pong kernel at any size)"},
  {"attn_ksplit", 0, 0, "key split of the d=64 split-precision attention: 0 by launch size (blocks that do not fill the chip's block slots a whole number of times), 1 off, 2 / 4 forced"},
  {"precise_mask", -1, -1, "stages in split precision (-1 = the config's own mask; per-stage attribution experiments; read at sdm_create)"},
};
static OptEntry* opt_find(const char* name) {
  for (auto& o : g_opts) if (name && strcmp(o.name, name) == 0) return &o;
  return nullptr;
}
static int opt(const char* name) { OptEntry* o = opt_find(name); return o ? o->value : 0; }
// One process may run one engine per host thread (parallel.py MultiGpuEngine, ctypes releases the GIL).  Options are process-wide: a forward
// holds g_opt_mu shared from before its dry pass to the end of its launch pass, sdm_set_option / sdm_reset_options take it exclusively - an option
// can therefore never change between the pass that sizes the arena and the pass that launches (it waits for the forwards in flight).  The launch
// counters are a mutex-guarded map: a few hundred increments per step.
static std::shared_mutex g_opt_mu;
struct OptReadLock { std::shared_lock<std::shared_mutex> l; OptReadLock() : l(g_opt_mu) {} };
static std::mutex g_count_mu;
static std::map<std::string, long> g_kernel_counts;
static void count_kernel(const char* name) { std::lock_guard<std::mutex> g(g_count_mu); g_kernel_counts[name] += 1; }

// ------------------------------------------------------------------------------------------------
// conv tile configurations
// ------------------------------------------------------------------------------------------------
template <int NTAPS, int STRIDE, int TH, int TW, int BN, int KC, int WM, int WN, int DB = 0, int GNOK = 0>
static void launch_conv_t(const ConvParams& p_in, void* stream) {
  using C = ConvCfg<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, DB>;
  ConvParams p = p_in;
  p.tiles_m = (NTAPS == 9) ? sdm_cdiv(p.Wout, TW) * sdm_cdiv(p.Hout, TH)
                           : (int)(((p.rows_per_img ? (long)p.rows_per_img : p.M) + C::BM - 1) / C::BM);
  p.tiles_n = sdm_cdiv(p.Cout_pad, BN);
  const long total_m = (long)p.tiles_m * ((NTAPS == 9 || p.rows_per_img) ? p.N : 1);
  p.xcd_chunk = (int)((total_m + 7) / 8);
  const dim3 grid((unsigned)(8L * p.xcd_chunk * p.tiles_n), (unsigned)(p.ksplit > 1 ? p.ksplit : 1), 1);      // x: XCD-aware 1-D mapping in the kernel; y: split-K
  const size_t gn_extra = (size_t)(p.C0 + p.C1) * 8;                    // fused GroupNorm apply: the scale|shift table of the image follows the tiles in LDS
  if constexpr (!DB) {
    if (p.w_lo) {                  // precise mode: split-fp16 operands (fp32 activations only)
      using CS = ConvCfg<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, 0, 1>;
      const size_t lds_pad = (size_t)opt("split_lds_pad");   // experiment option: extra dynamic LDS (forces 1 block per CU)
      if (GNOK && p.gn_scale) {
        auto k = conv_mfma_kernel<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, 1, 0, GNOK, 1>;
        SDM_SET_SMEM(k, 160 * 1024);
        SDM_LAUNCH(k, grid, dim3(CS::NTHREADS), (size_t)CS::SMEM + gn_extra + lds_pad, stream, p);
      } else {
        auto k = conv_mfma_kernel<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, 1, 0, 0, 1>;
        SDM_SET_SMEM(k, 160 * 1024);
        SDM_LAUNCH(k, grid, dim3(CS::NTHREADS), CS::SMEM + lds_pad, stream, p);
      }
      return;
    }
  }
  if (GNOK && p.gn_scale) {
    const size_t smem = (size_t)C::SMEM + gn_extra;
    if (p.in_f32) {
      auto k = conv_mfma_kernel<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, 1, DB, GNOK>;
      SDM_SET_SMEM(k, C::SMEM + 1024 * 8);
      SDM_LAUNCH(k, grid, dim3(C::NTHREADS), smem, stream, p);
    } else {
      auto k = conv_mfma_kernel<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, 0, DB, GNOK>;
      SDM_SET_SMEM(k, C::SMEM + 1024 * 8);
      SDM_LAUNCH(k, grid, dim3(C::NTHREADS), smem, stream, p);
    }
    return;
  }
  if (p.in_f32) {
    auto k = conv_mfma_kernel<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, 1, DB, 0>;
    SDM_SET_SMEM(k, C::SMEM);
    SDM_LAUNCH(k, grid, dim3(C::NTHREADS), C::SMEM, stream, p);
  } else {
    auto k = conv_mfma_kernel<NTAPS, STRIDE, TH, TW, BN, KC, WM, WN, 0, DB, 0>;
    SDM_SET_SMEM(k, C::SMEM);
    SDM_LAUNCH(k, grid, dim3(C::NTHREADS), C::SMEM, stream, p);
  }
}

struct ConvCfgInfo { int TH, TW, BN, KC, WM; };
static const ConvCfgInfo kCfg3s1[] = {{8, 32, 128, 16, 2}, {4, 32, 64, 32, 4}, {8, 8, 64, 16, 2}, {16, 32, 128, 16, 4}, {8, 32, 32, 16, 4}, {8, 32, 64, 16, 2}};
static const ConvCfgInfo kCfg3s2[] = {{4, 32, 64, 16, 4}, {8, 8, 64, 16, 2}, {4, 32, 128, 16, 2}, {8, 32, 128, 16, 4}};
static const ConvCfgInfo kCfg1[] = {{8, 32, 128, 64, 2}, {4, 32, 64, 64, 4}, {8, 8, 64, 64, 2}, {8, 8, 64, 16, 2}, {8, 32, 128, 32, 2}};

static int conv_num_cfgs(int ntaps, int stride) { return ntaps == 9 ? (stride == 1 ? 6 : 4) : 5; }
static const ConvCfgInfo* conv_cfg_table(int ntaps, int stride) { return ntaps == 9 ? (stride == 1 ? kCfg3s1 : kCfg3s2) : kCfg1; }

static bool conv_cfg_ok(const ConvCfgInfo& c, const ConvParams& p) {
  const int Cin = p.C0 + p.C1;
  if (Cin % c.KC) return false;
  if (p.C1 > 0 && (p.C0 % c.KC)) return false;
  return true;
}

static int conv_pick_cfg(int ntaps, int stride, const ConvParams& p) {
  const ConvCfgInfo* t = conv_cfg_table(ntaps, stride);
  const int n = conv_num_cfgs(ntaps, stride);
  long best_blocks = -1;
  int best = -1;
  if (ntaps == 9 && stride == 1 && opt("force_cfg0") && conv_cfg_ok(t[0], p)) return 0;   // test option: exercise the 256x128 tile (+ fused GroupNorm) at tiny sizes
  const bool use_db = opt("conv_db") != 0;   // A/B option for the 512x128 double-buffered tile
  for (int i = 0; i < n; ++i) {
    if (!conv_cfg_ok(t[i], p)) continue;
    if (ntaps == 9 && stride == 1 && i >= 3) continue;          // cfg 3 / 4: variants of cfg 0, substituted below
    if (ntaps == 9 && stride == 2 && i >= 2) continue;          // cfg 2: forced only; cfg 3: substituted below
    if (ntaps == 1 && i >= 4) continue;                         // cfg 4: the fp32-input form of cfg 0, substituted below
    long blocks;
    if (ntaps == 9) {
      if (t[i].TW > 8 && p.Wout < 24) continue;   // 32-wide strips would be mostly padding
      blocks = (long)p.N * sdm_cdiv(p.Hout, t[i].TH) * sdm_cdiv(p.Wout, t[i].TW) * sdm_cdiv(p.Cout_pad, t[i].BN);
    } else {
      blocks = ((p.M + t[i].TH * t[i].TW - 1) / (t[i].TH * t[i].TW)) * sdm_cdiv(p.Cout_pad, t[i].BN);
    }
    if (best < 0) { best = i; best_blocks = blocks; }
    // first (largest) tile that still spreads over the chip.  Measured: a 3x3 stride-1 layer with 160 blocks of the 256x128 tile
    // beats 640 blocks of the 128x64 tile by 6-12 % (and 160 x 128x64 beats 320 x 64x64 by 22 %), so half a block per CU is
    // enough there; GEMMs and stride-2 layers keep the one-block-per-CU rule.
    const long enough = (ntaps == 9 && stride == 1) ? 128 : 256;
    if (blocks >= enough) {
      // cfg 3 (512-pixel tile, 1 block per CU) needs at least ~2 blocks per CU of its own to pay off
      if (use_db && ntaps == 9 && stride == 1 && i == 0 && (long)p.N * sdm_cdiv(p.Hout, 16) * sdm_cdiv(p.Wout, 32) * sdm_cdiv(p.Cout_pad, 128) >= 512) return 3;
      // thin outputs (conv_out layers, Cout <= 32): the same 256-pixel tile with 32 output channels instead of 128 (HBM-bound
      // layers: 128 -> 3 @1024^2 1.33 -> 0.63 ms)
      if (ntaps == 9 && stride == 1 && i == 0 && p.Cout_pad <= 32 && conv_cfg_ok(t[4], p)) return 4;
      // fewer than two rounds of the 256x128 tile (2 blocks per CU): the 256x64 tile (3 blocks per CU, twice the blocks) fills the
      // chip better - measured +5..17 % on the U-Net layers (320 ch @128^2, 640 @64^2, 1280 @32^2), -3..10 % on the large VAE layers
      // (not for layers with the fp8-residual weights: their 8-wave one-block-per-CU kernel on the 256x128 tile measured
      // 1.35-1.5x the 256x64 tile on exactly these shapes)
      if (ntaps == 9 && stride == 1 && i == 0 && blocks < 1024 && p.Cout_pad > 64 && !p.f8_hint && conv_cfg_ok(t[5], p)) return 5;
      // stride 2: 256 pixels x 128 channels on 8 waves halves the input re-reads per output channel (+25 % on the VAE
      // down-samplers) once there is a block for every CU
      if (ntaps == 9 && stride == 2 && i == 0 && conv_cfg_ok(t[3], p) &&
          (long)p.N * sdm_cdiv(p.Hout, 8) * sdm_cdiv(p.Wout, 32) * sdm_cdiv(p.Cout_pad, 128) >= 256) return 3;
      // fp32 activations into the 256x128 GEMM tile: K-chunks of 32 (the 64-channel chunk needs 64 staging registers on top of
      // the 128 accumulators and spills)
      if (ntaps == 1 && i == 0 && p.in_f32 && conv_cfg_ok(t[4], p)) return 4;
      return i;
    }
    if (blocks > best_blocks) { best = i; best_blocks = blocks; }
  }
  return best;
}

// Split-K for the register-staged kernels (k_conv.h, ConvParams::ksplit).  The 16x16 / 32x32 levels of the U-Net have a handful of
// 64- or 128-pixel tiles and K = 9 x 1280 ... 2560: one block walks up to 160 chunks with a global-load round trip in each, and at
// one image per call there are fewer blocks than CUs.  Splitting K puts s times the waves in flight; the partial sums (fp32, a few MB)
// are added by splitk_reduce_kernel, which also applies bias / residual / statistics.  Returns 1 when the layer is not split.
static int conv_pick_ksplit(int ntaps, int stride, int cfg, const ConvParams& p) {
  const int o = opt("conv_splitk");
  if (o == 0 || o == 1) return 1;
  const ConvCfgInfo& c = conv_cfg_table(ntaps, stride)[cfg];
  const int Cin = p.C0 + p.C1;
  auto valid = [&](int s) { return s >= 2 && Cin % (s * c.KC) == 0; };
  if (o >= 2) return valid(o) ? o : 1;
  const long blocks = (ntaps == 9) ? (long)p.N * sdm_cdiv(p.Hout, c.TH) * sdm_cdiv(p.Wout, c.TW) * sdm_cdiv(p.Cout_pad, c.BN)
                                   : ((p.M + c.TH * c.TW - 1) / (c.TH * c.TW)) * sdm_cdiv(p.Cout_pad, c.BN);
  // measured (tools/conv_splitk_ab.py, profiles/r04_conv_splitk_ab.txt), 1 and 4 images per call: layers with <= 320 blocks gain x1.1-3.0 from as
  // many splits as keep blocks x splits <= 1280 (3x3 at 16x16: 219 -> 85 us at one image, 253 -> 135 us at four; Linear 5120 -> 1280 at 32x32:
  // 156 -> 77 us); at 640 blocks and above splitting loses or is neutral
  const long kdepth = (long)Cin * ntaps;
  const long min_total = ntaps == 9 ? 2560 : 1280, min_part = ntaps == 9 ? 1152 : 320;
  if (blocks > 320 || kdepth < min_total) return 1;
  int best = 1;
  for (int s = 2; s <= 8; s *= 2)
    if (valid(s) && blocks * s <= 1280 && kdepth / s >= min_part) best = s;
  return best;
}

// F8 conv kernel: tiles a block runs back to back (the producer waves stage tile k+1 under the epilogue of tile k).  Only when
// every CU still gets a block: 160 tiles as 80 two-tile blocks measured 0.43 vs 0.26 ms.  The option conv_f8_tpb overrides (A/B).
// compute units of the current device (256 on an MI355X)
static int device_cus() {
#ifdef SDM_EMU
  return 256;
#else
  static std::atomic<int> cus{0};
  int c = cus.load(std::memory_order_relaxed);
  if (!c) {
    int dev = 0;
    hipDeviceProp_t pr;
    c = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) ? pr.multiProcessorCount : 256;
    cus.store(c, std::memory_order_relaxed);
  }
  return c;
#endif
}

static int conv_f8_tiles_per_block(long tiles) {
#ifdef SDM_EMU
  return tiles >= 6 ? 3 : (tiles >= 2 ? 2 : 1);
#else
  if (opt("conv_f8_tpb") >= 1 && opt("conv_f8_tpb") <= 8) return opt("conv_f8_tpb");
  const int cus = device_cus();
  // measured (profiles/r02_conv_f8_tiles_per_block.txt): 128->128 @1024^2 (64 tiles per CU) 490 / 504 / 516 TFLOP/s at 1 / 2 / 4 tiles per
  // block; 320->320 @128^2 (3 per CU) 542 / 455; 1280->1280 @32^2 (0.6 per CU) 575 / 299: only deep queues gain
  const long per_cu = tiles / cus;
  return per_cu >= 32 ? 4 : (per_cu >= 8 ? 2 : 1);
#endif
}

// 256 px x 128 co, 3x3 stride 1, weights through the LDS-DMA stage ring (k_conv.h, DMAB): fp16 / fp32 activations, optional fused
// GroupNorm, optional split precision
static void launch_conv_dma(const ConvParams& p_in, void* stream) {
  ConvParams p = p_in;
  p.tiles_m = sdm_cdiv(p.Wout, 32) * sdm_cdiv(p.Hout, 8);
  p.tiles_n = sdm_cdiv(p.Cout_pad, 128);
  const long total_m = (long)p.tiles_m * p.N;
  p.xcd_chunk = (int)((total_m + 7) / 8);
  if (p.tile_flag) {
    // some tiles of some images will be left out (ConvParams::tile_flag): bands of a few tile rows of EVERY image go round-robin over the XCDs, so
    // that no XCD owns only the images (or only the image regions) that are skipped
    const int npx = sdm_cdiv(p.Wout, 32), rows = sdm_cdiv(p.Hout, 8);
    const int br = opt("conv_band_rows") > 0 ? opt("conv_band_rows") : std::max(1, std::min(4, rows / 16));
    p.band = br * npx;
    p.img_chunk = sdm_cdiv(p.tiles_m, 8 * p.band) * p.band;
    p.xcd_chunk = p.N * p.img_chunk;
  }
  const dim3 grid((unsigned)(8L * p.xcd_chunk * p.tiles_n), 1, 1);
  const bool gn = p.gn_scale != nullptr, split = p.w_lo != nullptr;
  const size_t gn_extra = gn ? (size_t)(p.C0 + p.C1) * 8 : 0;
#define SDM_DMA_CASE(F32, GNF, SPL, PCF)                                                                     \
  do {                                                                                                       \
    using CD = ConvCfg<9, 1, 8, 32, 128, 16, 2, 2, 0, SPL, 1, PCF>;                                          \
    auto k = conv_mfma_kernel<9, 1, 8, 32, 128, 16, 2, 2, F32, 0, GNF, SPL, 1, PCF>;                         \
    SDM_SET_SMEM(k, 160 * 1024);                                                                             \
    SDM_LAUNCH(k, grid, dim3(CD::LAUNCH_THREADS), (size_t)CD::SMEM + gn_extra, stream, p);                   \
  } while (0)
  if (split && p.f8) {          // fp8-residual producer / consumer kernel (32-channel chunks; no GroupNorm table in LDS)
#define SDM_F8_CASE(GNF)                                                                                     \
  do {                                                                                                       \
    using CD = ConvCfg<9, 1, 8, 32, 128, 32, 2, 2, 0, 1, 1, 1, 1>;                                           \
    auto k = conv_mfma_kernel<9, 1, 8, 32, 128, 32, 2, 2, 1, 0, GNF, 1, 1, 1, 1>;                            \
    SDM_SET_SMEM(k, 160 * 1024);                                                                             \
    p.vgrid = (int)grid.x;                                                                                   \
    p.tpb = conv_f8_tiles_per_block((long)grid.x);                                                           \
    unsigned pg = (grid.x + p.tpb - 1) / p.tpb;                                                              \
    pg = (pg + 7) & ~7u;              /* block id % 8 = XCD: the stride between a block's tiles stays a multiple of 8 */ \
    SDM_LAUNCH(k, dim3(pg, 1, 1), dim3(CD::LAUNCH_THREADS), (size_t)CD::SMEM_F8, stream, p);                 /* LDS map: ConvCfg (k_conv.h) */ \
  } while (0)
    count_kernel(gn ? "conv3x3_f8<gn>" : "conv3x3_f8");
    if (gn) SDM_F8_CASE(1); else SDM_F8_CASE(0);
#undef SDM_F8_CASE
  }
  else if (split && p.pc) { count_kernel("conv3x3_pc"); if (gn) SDM_DMA_CASE(1, 1, 1, 1); else SDM_DMA_CASE(1, 0, 1, 1); }
  else if (split) { if (gn) SDM_DMA_CASE(1, 1, 1, 0); else SDM_DMA_CASE(1, 0, 1, 0); }
  else if (p.in_f32) { if (gn) SDM_DMA_CASE(1, 1, 0, 0); else SDM_DMA_CASE(1, 0, 0, 0); }
  else { if (gn) SDM_DMA_CASE(0, 1, 0, 0); else SDM_DMA_CASE(0, 0, 0, 0); }
#undef SDM_DMA_CASE
}

// Linear / 1x1 GEMM, 256 rows x 128 channels, fp8-residual producer / consumer kernel (k_conv.h, F8 with NTAPS = 1)
static void launch_gemm_f8(const ConvParams& p_in, void* stream) {
  using CD = ConvCfg<1, 1, 8, 32, 128, 32, 2, 2, 0, 1, 1, 1, 1>;
  ConvParams p = p_in;
  p.tiles_m = (int)(((p.rows_per_img ? (long)p.rows_per_img : p.M) + CD::BM - 1) / CD::BM);
  p.tiles_n = sdm_cdiv(p.Cout_pad, 128);
  const long total_m = (long)p.tiles_m * (p.rows_per_img ? p.N : 1);
  p.xcd_chunk = (int)((total_m + 7) / 8);
  const unsigned vg = (unsigned)(8L * p.xcd_chunk * p.tiles_n);
  p.vgrid = (int)vg;
  p.tpb = conv_f8_tiles_per_block((long)vg);
  unsigned pg = (vg + p.tpb - 1) / p.tpb;
  pg = (pg + 7) & ~7u;
  count_kernel("gemm_f8");
  auto k = conv_mfma_kernel<1, 1, 8, 32, 128, 32, 2, 2, 1, 0, 0, 1, 1, 1, 1>;
  SDM_SET_SMEM(k, 160 * 1024);
  SDM_LAUNCH(k, dim3(pg, 1, 1), dim3(CD::LAUNCH_THREADS), (size_t)CD::SMEM_F8, stream, p);      // LDS map: ConvCfg (k_conv.h)
}

// ---- plane-fed GEMM (k_gemm.h): tile = (64 * MT) rows x 128 channels, 4 waves; NS LDS stages (2 stages of the 256-row tile: two blocks per CU) ----
template <int MT, int EPI, int NS>
static void launch_gemm_p3_t(GemmP3Params p, void* stream) {
  constexpr int BM = 64 * MT, SMEM = NS * (BM * 96 + 128 * 128);
  const long rows = p.rows_per_img ? (long)p.rows_per_img : p.M;
  p.tiles_per_img = (int)((rows + BM - 1) / BM);
  p.tiles_m = p.tiles_per_img * (p.rows_per_img ? (int)(p.M / p.rows_per_img) : 1);
  p.tiles_n = sdm_cdiv(p.N, 128);
  unsigned grid;
  if (p.tiles_m >= 8) { p.xcd_chunk = (p.tiles_m + 7) / 8; grid = (unsigned)(8L * p.xcd_chunk * p.tiles_n); }
  else { p.xcd_chunk = 0; grid = (unsigned)(p.tiles_m * p.tiles_n); }
  // persistent grid: as many blocks as the chip holds at once (a multiple of 8: a block's tiles stay on its XCD's M range), each walks its tiles as one
  // DMA stream (k_gemm.h); the option gemm_p3_persist = 0 launches one block per tile
  if (const int pp = opt("gemm_p3_persist")) {
    const unsigned slots = pp > 1 ? (unsigned)pp : ((unsigned)(device_cus() * ((SMEM <= 80 * 1024) ? 2 : 1)) & ~7u);      // (pp > 1: forced grid, tests)
    if (slots >= 1 && grid > slots) grid = slots;
  }
  auto k = gemm_p3_kernel<MT, 2, EPI, NS>;
  SDM_SET_SMEM(k, SMEM);
#ifndef SDM_EMU
  if (opt("gemm_p3_ablate") & 256) {      // lab: resident blocks per CU as the runtime sees them
    int nb = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k, 256, (size_t)SMEM);
    fprintf(stderr, "[gemm_p3] tile %d stages %d epi %d: %d bytes of LDS, %d blocks per CU, grid %u\n", BM, NS, EPI, SMEM, nb, grid);
  }
#endif
  SDM_LAUNCH(k, dim3(grid, 1, 1), dim3(256), (size_t)SMEM, stream, p);
}
// LDS stages per row tile: the option gemm_p3_stages = 0 takes the default of the tile, n >= 2 asks for n (lab forms exist for the fp32 and GEGLU epilogues)
template <int EPI>
static void launch_gemm_p3_e(const GemmP3Params& p, int bm, void* stream) {
  const int ns = opt("gemm_p3_stages");
  if (EPI == 0 || EPI == 1) {
    if (bm == 256 && ns == 3) return launch_gemm_p3_t<4, EPI, 3>(p, stream);
    if (bm == 256 && ns == 4) return launch_gemm_p3_t<4, EPI, 4>(p, stream);
    if (bm == 128 && ns == 4) return launch_gemm_p3_t<2, EPI, 4>(p, stream);
    if (bm == 128 && ns == 5) return launch_gemm_p3_t<2, EPI, 5>(p, stream);
    if (bm == 64 && ns == 4) return launch_gemm_p3_t<1, EPI, 4>(p, stream);
    if (bm == 64 && ns == 7) return launch_gemm_p3_t<1, EPI, 7>(p, stream);
  }
  if (bm == 256) launch_gemm_p3_t<4, EPI, 2>(p, stream);
  else if (bm == 128) launch_gemm_p3_t<2, EPI, 2>(p, stream);
  else launch_gemm_p3_t<1, EPI, 2>(p, stream);
}
// row tile: the largest of 256 / 128 / 64 that still gives every CU a block (the option gemm_p3_tile forces one)
static int gemm_p3_pick_bm(long M, int N, int rows_per_img) {
  const int forced = opt("gemm_p3_tile");
  if (forced == 256 || forced == 128 || forced == 64) return forced;
  const long tn = sdm_cdiv(N, 128);
  const long imgs = rows_per_img ? M / rows_per_img : 1, rows = rows_per_img ? rows_per_img : M;
  const int cus = device_cus();
  for (int bm : {256, 128}) if (imgs * ((rows + bm - 1) / bm) * tn >= cus) return bm;
  return 64;
}
static void launch_gemm_p3(const GemmP3Params& p, int epi, void* stream) {
  const int bm = gemm_p3_pick_bm(p.M, p.N, p.rows_per_img);
  switch (epi) {
    case 0: launch_gemm_p3_e<0>(p, bm, stream); break;
    case 1: launch_gemm_p3_e<1>(p, bm, stream); break;
    case 2: launch_gemm_p3_e<2>(p, bm, stream); break;
    case 3: launch_gemm_p3_e<3>(p, bm, stream); break;
    default: launch_gemm_p3_e<4>(p, bm, stream); break;
  }
}

static int launch_conv(int ntaps, int stride, int cfg, const ConvParams& p, void* stream) {
  if (ntaps == 9 && stride == 1) {
    switch (cfg) {
      case 0:
        if (p.w_dma) { launch_conv_dma(p, stream); return 0; }
        launch_conv_t<9, 1, 8, 32, 128, 16, 2, 2, 0, 1>(p, stream); return 0;   // 256 px x 128 co, fused-GroupNorm variant
      case 1: launch_conv_t<9, 1, 4, 32, 64, 32, 4, 1>(p, stream); return 0;
      case 2: launch_conv_t<9, 1, 8, 8, 64, 16, 2, 1>(p, stream); return 0;
      case 3: launch_conv_t<9, 1, 16, 32, 128, 16, 4, 2, 1>(p, stream); return 0;   // 512 px x 128 co, 8 waves, swizzled double-buffered LDS tiles
      case 4: launch_conv_t<9, 1, 8, 32, 32, 16, 4, 1, 0, 1>(p, stream); return 0;   // 256 px x 32 co: thin-output convs (conv_out), fused GroupNorm
      case 5: launch_conv_t<9, 1, 8, 32, 64, 16, 2, 2, 0, 1>(p, stream); return 0;   // 256 px x 64 co: 3 blocks per CU, for layers with few tiles; fused GroupNorm
    }
  } else if (ntaps == 9 && stride == 2) {
    switch (cfg) {
      case 0: launch_conv_t<9, 2, 4, 32, 64, 16, 4, 1>(p, stream); return 0;
      case 1: launch_conv_t<9, 2, 8, 8, 64, 16, 2, 1>(p, stream); return 0;
      case 2: launch_conv_t<9, 2, 4, 32, 128, 16, 2, 2>(p, stream); return 0;
      case 3: launch_conv_t<9, 2, 8, 32, 128, 16, 4, 2>(p, stream); return 0;
    }
  } else if (ntaps == 1) {
    switch (cfg) {
      case 0: launch_conv_t<1, 1, 8, 32, 128, 64, 2, 2>(p, stream); return 0;
      case 1: launch_conv_t<1, 1, 4, 32, 64, 64, 4, 1>(p, stream); return 0;
      case 2: launch_conv_t<1, 1, 8, 8, 64, 64, 2, 1>(p, stream); return 0;
      case 3: launch_conv_t<1, 1, 8, 8, 64, 16, 2, 1>(p, stream); return 0;
      case 4:
        if (p.f8 && p.w_dma) { launch_gemm_f8(p, stream); return 0; }
        launch_conv_t<1, 1, 8, 32, 128, 32, 2, 2>(p, stream); return 0;
    }
  }
  return -1;
}

// split-K launch: the conv kernel with grid.y = ksplit writes fp32 partial sums to the workspace, splitk_reduce_kernel finishes the layer
// (bias, output scale, residual, store format, statistics with one partial row per kSplitKRows rows of an image)
static const int kSplitKRows = 64;
static int launch_conv_splitk(int ntaps, int stride, int cfg, const ConvParams& p, int ksplit, float* ws, void* stream) {
  ConvParams ps = p;
  ps.out = ws; ps.out_f32 = 1; ps.Cout_store = p.Cout_pad; ps.out_ch_off = 0; ps.Cout_valid = p.Cout_pad;
  ps.bias = nullptr; ps.bias_sel = nullptr; ps.res = nullptr; ps.out_scale = 1.0f; ps.stats = nullptr; ps.rows_per_img = 0;
  ps.ksplit = ksplit; ps.ks_stride = (size_t)p.M * p.Cout_pad;
  const int rc = launch_conv(ntaps, stride, cfg, ps, stream);
  if (rc != 0) return rc;
  const int bpi = sdm_cdiv(p.Hout * p.Wout, kSplitKRows);
  SplitKReduceParams q;
  memset(&q, 0, sizeof(q));
  q.ws = ws; q.ksplit = ksplit; q.ks_stride = ps.ks_stride; q.ws_C = p.Cout_pad;
  q.rows_per_img = p.Hout * p.Wout; q.rb = kSplitKRows; q.blocks_per_img = bpi;
  q.bias = p.bias; q.bias_sel = p.bias_sel; q.Cout_pad = p.Cout_pad;
  q.res = p.res; q.res_f32 = p.res_f32; q.res_C = p.res_C; q.out_scale = p.out_scale;
  q.out = p.out; q.out_f32 = p.out_f32; q.Cout_store = p.Cout_store; q.out_ch_off = p.out_ch_off; q.Cout_valid = p.Cout_valid;
  q.stats = p.stats;
  SDM_LAUNCH(splitk_reduce_kernel, dim3((unsigned)(p.N * bpi), (unsigned)sdm_cdiv(p.Cout_valid, 64), 1), dim3(256), 0, stream, q);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// engine data structures
// ------------------------------------------------------------------------------------------------
// Residual terms of the split-precision 3x3 convs on fp8 operands (k_conv.h, F8): default on; the option conv_f8 = 0 keeps them on fp16
// (the round-2 "fp16x3" arithmetic everywhere).  Read when a model is built: the weight copy is packed for one of the two.
static bool conv_f8_enabled() { return opt("conv_f8") != 0; }

static bool gemm_f8_enabled() { return opt("gemm_f8") != 0; }

// epilogue of the F8 kernels (ConvParams::epi_mode): 4 = register-direct 16-byte stores + residual as the accumulators' initial value
// (default), 3 = LDS-staged stores + residual as initial value, 0 = LDS-staged stores, residual added in the epilogue.  Read per launch:
// A/B hook.
static int conv_epi_mode() {
  const int v = opt("conv_epi");
  return (v == 0 || v == 3 || v == 4) ? v : 4;
}

// Residual terms of Q.K^T in the split-precision attention cores on fp8 MFMAs (k_attn.h, PREC = 3; q / k arrive as fp16 + e5m2 pair planes):
// default on; the option attn_f8 = 0 keeps them on fp16 MFMAs (PREC = 2, fp16 hi | lo planes).  Read per forward: A/B hook.
static bool attn_f8_enabled() { return opt("attn_f8") != 0; }

// F8 3x3 kernels: cross-tile prefetch by the producer waves (k_conv.h); the option conv_xtile = 0 disables.  Read per launch: A/B hook.
static bool conv_xtile_enabled() { return opt("conv_xtile") != 0; }

static int gemm_f8_min_k() { return opt("gemm_f8_min_k"); }

struct ConvL {
  std::string name;
  int ntaps = 1, I = 0, O = 0, Cin_pad = 0, Cout_pad = 0, geglu = 0;
  size_t w_off = 0, b_off = 0;
  half_t* w = nullptr;
  float* b = nullptr;
  // precise mode (sdm_config::precise_mask has the layer's stage bit): weights are packed as fp16 pairs w * 2^w_exp = hi + lo
  int stage = 0, split = 0, w_exp = 0;
  size_t wlo_off = 0;
  half_t* w_lo = nullptr;
  // 3x3 layers wide enough for the 256x128 tile also keep their weights in the stage order of the DMA-weight kernels
  size_t wdma_off = 0, wdma_bytes = 0;
  half_t* w_dma = nullptr;
  int f8 = 0;                 // w_dma holds the fp8-residual layout (F8 conv kernel) instead of the stage-ordered hi | lo pair
  int f8_exp = 8;             // f8: the layer's e4m3 weight scale 2^f8_exp, the largest power of two with max|w| * 2^f8_exp <= 448 (derive_layer)
  // Linear layers of the split-precision stages also keep the W3 layout of the plane-fed GEMM (k_gemm.h; same f8_exp)
  size_t w3_off = 0, w3_bytes = 0;
  unsigned char* w3 = nullptr;
};
static const int kSplitWeightExp = 8;      // pre-scale 2^8: typical |w| ~ 1e-2 .. 1 -> low parts ~ 1e-3 .. 1e-1 * 2^-4: fp16-normal
struct NormL {
  int C = 0;
  size_t g_off = 0, b_off = 0;
  float* g = nullptr;
  float* b = nullptr;
};
enum SlotKind { SLOT_CONV_W, SLOT_CONV_B, SLOT_NORM_G, SLOT_NORM_B, SLOT_HOST };
struct Slot {
  int kind = 0, layer = -1, co_off = 0, ci_off = 0;
  float w_scale = 1.0f;    // SLOT_CONV_W: constant folded into the weight before the fp16 rounding (attention logit scale in to_q)
  std::vector<int64_t> shape;
  size_t host_off = 0;     // SLOT_HOST: float offset in the host blob
  bool loaded = false;
};
struct ResB { int norm1 = -1, conv1 = -1, norm2 = -1, conv2 = -1, sc = -1, temb = -1, cin = 0, cout = 0; };
struct VaeAttnB { int gn = -1, qkv = -1, out = -1, C = 0; };
struct TfB { int gn, proj_in, ln1, qkv1, o1, ln2, q2, kv2, o2, ln3, ff1, ff2, proj_out, C, heads; size_t k_hoff, v_hoff; };
struct TembL { size_t w_hoff = 0, b_hoff = 0, cb_hoff = 0; int cout = 0, cout_pad = 0; float* table = nullptr; };

struct T {  // NHWC activation tensor living in the arena
  size_t off = 0, bytes = 0;
  void* p = nullptr;
  int N = 0, H = 0, W = 0, C = 0, f32 = 0;
  bool want_stats = false;      // the producing conv should emit fused GroupNorm statistics
  float* stats = nullptr;       // [N][srows][C][2] partial {sum, sumsq} rows written by the producing conv epilogue
  int srows = 0;
  size_t soff = 0, sbytes = 0;
  unsigned char* cmask = nullptr;   // optional class plane [N][H][W] of a piecewise-constant batch (k_misc.h cmask_*): set by vae_encode on the encoder input,
  size_t cm_off = 0, cm_bytes = 0;  // propagated by op_conv through 3x3 convs; owned by the tensor (freed with it)
  int cm_n0 = 0;                    // first image of the batch that carries classes (the rgb images in front of it have none)
  long rows() const { return (long)N * H * W; }
};

// activation element formats (T::f32): 0 fp16, 1 fp32, 2 two fp16 planes hi | lo (split-precision attention operands),
// 3 fp16 plane hi + e5m2 pair plane (the same, with the residual operands of Q.K^T already in fp8: ConvParams::out_f32),
// 4 "P3": fp16 plane hi + one plane of e5m2 residual bytes, 3 bytes per element - the operand format of the plane-fed GEMM (k_gemm.h)
static const int kFmtP3 = 4;
static inline size_t fmt_bytes(int f) { return f == kFmtP3 ? 3 : (f ? 4 : 2); }
static const int kMinVariantRows = 8;     // initial rows of the per-ResBlock bias tables (one row per distinct conditioning); grows on demand
// conditioning of one image: opacity class + either 4 box coordinates (kind 0: bbox_embedding) or N point coordinates
// (kind 1: point_embedding), meta_arch.py:147-197 / replace.py:446-457
struct Variant {
  int trans = 0, kind = 0;
  std::vector<float> c;
  bool operator==(const Variant& o) const { return trans == o.trans && kind == o.kind && c == o.c; }
};

struct ProfRec { std::string name, desc; double flops, bytes;
#ifndef SDM_EMU
  hipEvent_t e0, e1;
#endif
};

struct sdm_ctx {
  sdm_config cfg;
  int device = 0;
  const unsigned char* dbg_cmask = nullptr;      // test hook (sdm_debug_set_input_cmask): class plane of the next sdm_op_conv_ex input
  void* stream = nullptr;
  bool own_stream = false;
  std::string err;
  // weights
  std::vector<ConvL> convs;
  std::vector<NormL> norms;
  std::unordered_map<std::string, Slot> slots;
  std::vector<std::string> slot_order;
  unsigned char* warena = nullptr;
  size_t warena_bytes = 0, canon_bytes = 0;      // whole arena; its leading canonical part (the exported / imported blob)
  std::vector<float> hostblob;
  int64_t n_loaded = 0, n_ignored = 0;
  std::vector<std::string> missing;
  bool finalized = false;
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // asynchronous weight pipeline: a ring of (pinned host, device) staging pairs; tensor i+1 is converted
  // into pinned memory on the CPU while tensor i is copied and packed on the GPU - no host synchronisation per tensor
  struct LoadSlot {
    void* host = nullptr; void* dev = nullptr; size_t cap = 0; bool busy = false;
#ifndef SDM_EMU
    hipEvent_t done = nullptr;
#endif
  };
  static const int kLoadSlots = 3;
  LoadSlot load_ring[kLoadSlots];
  int load_next = 0;
  // model structure
  int enc_conv_in, enc_norm_out, enc_conv_out, quant, post_quant, dec_conv_in, dec_norm_out, dec_conv_out;
  std::vector<std::vector<ResB>> enc_res, dec_res;
  std::vector<int> enc_down, dec_up;
  ResB enc_mid0, enc_mid1, dec_mid0, dec_mid1;
  VaeAttnB enc_attn, dec_attn;
  int u_conv_in, u_aux, u_norm_out, u_conv_out;
  std::vector<std::vector<ResB>> u_down_res, u_up_res;
  std::vector<std::vector<TfB>> u_down_tf, u_up_tf;
  std::vector<int> u_down_ds, u_up_us;
  ResB u_mid0, u_mid1;
  TfB u_midtf;
  std::vector<TembL> tembs;
  size_t h_time1w, h_time1b, h_time2w, h_time2b, h_bbox1w, h_bbox1b, h_bbox2w, h_bbox2b, h_auxw, h_auxb;
  size_t h_point1w, h_point1b, h_point2w, h_point2b;
  std::vector<Variant> variants;
  int variant_cap = 0;         // rows allocated in every TembL::table
  int act_f32 = 0;             // precise_mask != 0: every activation that is fp16 in the fast graph is kept in fp32
  int* d_bias_sel = nullptr;   // [max batch]
  int bias_sel_cap = 0;
  // activation arena
  unsigned char* arena = nullptr;
  size_t arena_bytes = 0;
  bool dry = false;
  size_t peak = 0;
  std::map<size_t, size_t> freelist;  // off -> size
  size_t arena_top = 0;
  // io staging
  void* io_in = nullptr; size_t io_in_bytes = 0;
  void* io_out = nullptr; size_t io_out_bytes = 0;
  // timing / profiling
  float last_ms = 0.f;
  bool prof_on = false;
  std::vector<ProfRec> prof;
  struct ProfAgg { std::string name; float ms; int64_t n; double flops, bytes; };
  std::vector<ProfAgg> prof_agg;
  std::string prof_dump;   // per-launch CSV of the last profiled forward
#ifndef SDM_EMU
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;   // ordering against the caller's stream (SDM_PTR_DEVICE calls)
#endif
};

static std::string g_create_err;

#define SDM_FAIL(ctx, code, ...)                       \
  do {                                                 \
    char buf_[512];                                    \
    snprintf(buf_, sizeof(buf_), __VA_ARGS__);         \
    (ctx)->err = buf_;                                 \
    return (code);                                     \
  } while (0)

#define SDM_CHECK_DEV(ctx, expr)                                                              \
  do {                                                                                        \
    int e_ = (expr);                                                                          \
    if (e_ != 0) SDM_FAIL(ctx, SDM_ERR_HIP, "%s failed: %s (%s:%d)", #expr, dev_errstr(e_), __FILE__, __LINE__); \
  } while (0)

#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// ------------------------------------------------------------------------------------------------
// model construction (mirrors comfyui-sdmatte_amd/weights.py::weight_schema)
// ------------------------------------------------------------------------------------------------
struct Builder {
  sdm_ctx* e;
  size_t woff = 0;         // canonical region: K16 weights (hi | lo), biases, norm affine - what sdm_export_weight_blob carries
  size_t doff = 0;         // derived region (behind the canonical one): DMA-ordered / fp8-residual copies, rebuilt by sdm_finalize_weights
  int stage = 0;           // sdm_precise_stage of the layers being built
  explicit Builder(sdm_ctx* c) : e(c) {}
  int conv(const std::string& name, int ntaps, int I_pad16_src, int O, int geglu = 0) {
    ConvL L;
    L.name = name; L.ntaps = ntaps; L.I = I_pad16_src; L.O = O;
    L.Cin_pad = rup(I_pad16_src, 16);
    L.Cout_pad = rup(O, geglu ? 64 : 32);
    L.geglu = geglu;
    L.w_off = woff; woff += rupz((size_t)L.Cin_pad * ntaps * L.Cout_pad * 2, 256);
    L.b_off = woff; woff += rupz((size_t)L.Cout_pad * 4, 256);
    L.stage = stage;
    L.split = (e->cfg.precise_mask & stage) ? 1 : 0;
    if (L.split) { L.w_exp = kSplitWeightExp; L.wlo_off = woff; woff += rupz((size_t)L.Cin_pad * ntaps * L.Cout_pad * 2, 256); }
    // stage-ordered copy for the DMA-weight kernel: split-precision layers only (measured +4..6 % there; neutral with fp16 operands,
    // where the register-staged kernel stays; the option conv_dma_all = 1 builds the copy for every wide 3x3 layer)
    const bool dma_all = opt("conv_dma_all") != 0;
    if (ntaps == 9 && L.Cout_pad >= 128 && !geglu && (L.split || dma_all)) {
      L.wdma_bytes = (size_t)L.Cin_pad * 9 * L.Cout_pad * 2 * (L.split ? 2 : 1);
      L.wdma_off = doff; doff += rupz(L.wdma_bytes, 256);
      L.f8 = (L.split && L.Cin_pad % 32 == 0 && conv_f8_enabled()) ? 1 : 0;      // same bytes, fp8-residual layout
    }
    // Linear / 1x1 layers of the split-precision stages with K >= 1024: fp8-residual copy for the 8-wave GEMM kernel (k_conv.h, F8
    // with NTAPS = 1), used when the launch takes the 256 x 128 tile; SDM_GEMM_F8=0 disables.  A GEMM has no operand reuse across
    // taps, so the producer waves (one 32 KB activation tile converted per 1024 MFMA cycles) set the pace: measured against the
    // 4-wave kernel x1.2-1.5 for K = 1280 ... 5120, x0.84-1.0 for K <= 640 (profiles/r02_gemm_f8_ab.txt) - hence the threshold
    if (ntaps == 1 && L.split && L.Cout_pad >= 128 && L.Cin_pad % 32 == 0 && L.Cin_pad >= gemm_f8_min_k() && conv_f8_enabled() && gemm_f8_enabled()) {
      L.wdma_bytes = (size_t)L.Cin_pad * L.Cout_pad * 4;
      L.wdma_off = doff; doff += rupz(L.wdma_bytes, 256);
      L.f8 = 1;
    }
    // every Linear of a split-precision stage whose K splits into 32-channel chunks: W3 copy for the plane-fed GEMM (k_gemm.h)
    if (ntaps == 1 && L.split && L.Cin_pad % 32 == 0 && L.Cin_pad >= 32 && conv_f8_enabled() && opt("gemm_p3") != 0) {
      L.w3_bytes = (size_t)L.Cin_pad * L.Cout_pad * 4;
      L.w3_off = doff; doff += rupz(L.w3_bytes, 256);
    }
    e->convs.push_back(L);
    return (int)e->convs.size() - 1;
  }
  int norm(int C) {
    NormL n; n.C = C;
    n.g_off = woff; woff += rupz((size_t)C * 4, 256);
    n.b_off = woff; woff += rupz((size_t)C * 4, 256);
    e->norms.push_back(n);
    return (int)e->norms.size() - 1;
  }
  void slot(const std::string& key, int kind, int layer, std::vector<int64_t> shape, int co_off = 0, int ci_off = 0) {
    Slot s; s.kind = kind; s.layer = layer; s.co_off = co_off; s.ci_off = ci_off; s.shape = std::move(shape);
    if (kind == SLOT_HOST) {
      size_t n = 1; for (auto d : s.shape) n *= (size_t)d;
      s.host_off = e->hostblob.size();
      e->hostblob.resize(e->hostblob.size() + n, 0.0f);
    }
    e->slots[key] = s;
    e->slot_order.push_back(key);
  }
  // plain conv / linear layer "<p>.weight"/"<p>.bias"
  int conv_named(const std::string& p, int ntaps, int I, int O, bool bias = true, int ci_off = 0, int Ipad = 0) {
    int id = conv(p, ntaps, Ipad ? Ipad : I, O);
    if (ntaps == 9) slot(p + ".weight", SLOT_CONV_W, id, {O, I, 3, 3}, 0, ci_off);
    else slot(p + ".weight", SLOT_CONV_W, id, {O, I}, 0, ci_off);
    if (bias) slot(p + ".bias", SLOT_CONV_B, id, {O});
    return id;
  }
  int norm_named(const std::string& p, int C) {
    int id = norm(C);
    slot(p + ".weight", SLOT_NORM_G, id, {C});
    slot(p + ".bias", SLOT_NORM_B, id, {C});
    return id;
  }
  ResB resnet(const std::string& p, int cin, int cout, int temb_dim) {
    ResB r; r.cin = cin; r.cout = cout;
    r.norm1 = norm_named(p + ".norm1", cin);
    if (temb_dim > 0) {
      // conv1 bias is replaced by a per-variant table (bias + time_emb_proj(silu(emb))): keep host copies
      r.conv1 = conv(p + ".conv1", 9, cin, cout);
      slot(p + ".conv1.weight", SLOT_CONV_W, r.conv1, {cout, cin, 3, 3});
      TembL t; t.cout = cout; t.cout_pad = e->convs[r.conv1].Cout_pad;
      slot(p + ".conv1.bias", SLOT_HOST, -1, {cout}); t.cb_hoff = e->slots[p + ".conv1.bias"].host_off;
      slot(p + ".time_emb_proj.weight", SLOT_HOST, -1, {cout, temb_dim}); t.w_hoff = e->slots[p + ".time_emb_proj.weight"].host_off;
      slot(p + ".time_emb_proj.bias", SLOT_HOST, -1, {cout}); t.b_hoff = e->slots[p + ".time_emb_proj.bias"].host_off;
      e->tembs.push_back(t);
      r.temb = (int)e->tembs.size() - 1;
    } else {
      r.conv1 = conv_named(p + ".conv1", 9, cin, cout);
    }
    r.norm2 = norm_named(p + ".norm2", cout);
    r.conv2 = conv_named(p + ".conv2", 9, cout, cout);
    if (cin != cout) r.sc = conv_named(p + ".conv_shortcut", 1, cin, cout);
    return r;
  }
  VaeAttnB vae_attn(const std::string& p, int C) {
    const int outer = stage;
    stage = SDM_PRECISE_VAE_ATTN_LIN;
    struct Restore { int& s; int v; ~Restore() { s = v; } } restore{stage, outer};
    VaeAttnB a; a.C = C;
    a.gn = norm_named(p + ".group_norm", C);
    a.qkv = conv(p + ".qkv", 1, C, 3 * C);
    const char* nm[3] = {"to_q", "to_k", "to_v"};
    const char* legacy[3] = {"query", "key", "value"};
    for (int i = 0; i < 3; ++i) {
      slot(p + "." + nm[i] + ".weight", SLOT_CONV_W, a.qkv, {C, C}, i * C);
      slot(p + "." + nm[i] + ".bias", SLOT_CONV_B, a.qkv, {C}, i * C);
      (void)legacy;
    }
    a.out = conv_named(p + ".to_out.0", 1, C, C);
    return a;
  }
  TfB transformer(const std::string& p, int C, int heads, int ctx) {
    const int outer = stage;
    stage = SDM_PRECISE_UNET_TF;
    struct Restore { int& s; int v; ~Restore() { s = v; } } restore{stage, outer};
    TfB t; t.C = C; t.heads = heads;
    t.gn = norm_named(p + ".norm", C);
    t.proj_in = conv_named(p + ".proj_in", 1, C, C);
    const std::string b = p + ".transformer_blocks.0";
    t.ln1 = norm_named(b + ".norm1", C);
    t.ln2 = norm_named(b + ".norm2", C);
    t.ln3 = norm_named(b + ".norm3", C);
    t.qkv1 = conv(b + ".attn1.qkv", 1, C, 3 * C);
    slot(b + ".attn1.to_q.weight", SLOT_CONV_W, t.qkv1, {C, C}, 0);
    slot(b + ".attn1.to_k.weight", SLOT_CONV_W, t.qkv1, {C, C}, C);
    slot(b + ".attn1.to_v.weight", SLOT_CONV_W, t.qkv1, {C, C}, 2 * C);
    t.o1 = conv_named(b + ".attn1.to_out.0", 1, C, C);
    t.q2 = conv_named(b + ".attn2.to_q", 1, C, C, false);
    // softmax(q.k^T * d^-1/2) is evaluated as 2^(q'.k^T - max) with q' = q * d^-1/2 * log2(e): the constant goes into the
    // to_q weights (no bias in these projections), so the attention kernel needs no per-logit multiply
    e->slots[b + ".attn1.to_q.weight"].w_scale = e->slots[b + ".attn2.to_q.weight"].w_scale = 0.125f * SDM_LOG2E;
    // cross-attention K|V: K = W_k (W_aux * z + b_aux) is an affine map of the 3x3 patch of the 4-channel trimap latent z
    // (exact fold): ONE 3x3 conv 16(4 real)->2C with host-folded weights instead of aux_conv_in
    // (4->1024) followed by two 1024->C GEMMs.  to_k / to_v / aux_conv_in are kept on the host and folded in finalize.
    t.kv2 = conv(b + ".attn2.kv_folded", 9, 16, 2 * C);
    slot(b + ".attn2.to_k.weight", SLOT_HOST, -1, {C, ctx}); t.k_hoff = e->slots[b + ".attn2.to_k.weight"].host_off;
    slot(b + ".attn2.to_v.weight", SLOT_HOST, -1, {C, ctx}); t.v_hoff = e->slots[b + ".attn2.to_v.weight"].host_off;
    t.o2 = conv_named(b + ".attn2.to_out.0", 1, C, C);
    t.ff1 = conv(b + ".ff.net.0.proj", 1, C, 8 * C, 1);
    slot(b + ".ff.net.0.proj.weight", SLOT_CONV_W, t.ff1, {8 * C, C});
    slot(b + ".ff.net.0.proj.bias", SLOT_CONV_B, t.ff1, {8 * C});
    t.ff2 = conv_named(b + ".ff.net.2", 1, 4 * C, C);
    t.proj_out = conv_named(p + ".proj_out", 1, C, C);
    return t;
  }
};

static std::string S(int i) { return std::to_string(i); }

static void build_model(sdm_ctx* e) {
  Builder B(e);
  const sdm_config& c = e->cfg;
  const int* vc = c.vae_channels;
  const int lc = 4;
  // ---- VAE encoder ----
  B.stage = SDM_PRECISE_VAE_ENC;
  e->enc_conv_in = B.conv_named("vae.encoder.conv_in", 9, 3, vc[0]);
  int cprev = vc[0];
  e->enc_res.resize(4);
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < c.vae_layers_per_block; ++j)
      e->enc_res[i].push_back(B.resnet("vae.encoder.down_blocks." + S(i) + ".resnets." + S(j), j == 0 ? cprev : vc[i], vc[i], 0));
    cprev = vc[i];
    if (i < 3) e->enc_down.push_back(B.conv_named("vae.encoder.down_blocks." + S(i) + ".downsamplers.0.conv", 9, vc[i], vc[i]));
  }
  const int cm = vc[3];
  e->enc_mid0 = B.resnet("vae.encoder.mid_block.resnets.0", cm, cm, 0);
  e->enc_attn = B.vae_attn("vae.encoder.mid_block.attentions.0", cm);
  e->enc_mid1 = B.resnet("vae.encoder.mid_block.resnets.1", cm, cm, 0);
  e->enc_norm_out = B.norm_named("vae.encoder.conv_norm_out", cm);
  e->enc_conv_out = B.conv_named("vae.encoder.conv_out", 9, cm, 2 * lc);
  e->quant = B.conv_named("vae.quant_conv", 1, 2 * lc, 2 * lc);
  B.stage = SDM_PRECISE_VAE_DEC;
  e->post_quant = B.conv_named("vae.post_quant_conv", 1, lc, lc);
  // ---- VAE decoder ----
  e->dec_conv_in = B.conv_named("vae.decoder.conv_in", 9, lc, cm);
  e->dec_mid0 = B.resnet("vae.decoder.mid_block.resnets.0", cm, cm, 0);
  e->dec_attn = B.vae_attn("vae.decoder.mid_block.attentions.0", cm);
  e->dec_mid1 = B.resnet("vae.decoder.mid_block.resnets.1", cm, cm, 0);
  e->dec_res.resize(4);
  cprev = vc[3];
  for (int i = 0; i < 4; ++i) {
    const int co = vc[3 - i];
    for (int j = 0; j < c.vae_layers_per_block + 1; ++j)
      e->dec_res[i].push_back(B.resnet("vae.decoder.up_blocks." + S(i) + ".resnets." + S(j), j == 0 ? cprev : co, co, 0));
    cprev = co;
    if (i < 3) e->dec_up.push_back(B.conv_named("vae.decoder.up_blocks." + S(i) + ".upsamplers.0.conv", 9, co, co));
  }
  e->dec_norm_out = B.norm_named("vae.decoder.conv_norm_out", vc[0]);
  e->dec_conv_out = B.conv_named("vae.decoder.conv_out", 9, vc[0], 3);
  // ---- U-Net ----
  B.stage = SDM_PRECISE_UNET_RES;
  const int* uc = c.unet_channels;
  const int te = uc[0] * 4, ctx = c.cross_attention_dim;
  e->u_conv_in = B.conv_named("unet.conv_in", 9, c.unet_in_channels, uc[0]);
  // aux_conv_in (4->ctx, utils.py:33-41) only ever feeds the cross-attention K/V projections: folded into them (see transformer())
  e->u_aux = -1;
  B.slot("unet.aux_conv_in.weight", SLOT_HOST, -1, {ctx, 4, 3, 3}); e->h_auxw = e->slots["unet.aux_conv_in.weight"].host_off;
  B.slot("unet.aux_conv_in.bias", SLOT_HOST, -1, {ctx}); e->h_auxb = e->slots["unet.aux_conv_in.bias"].host_off;
  B.slot("unet.time_embedding.linear_1.weight", SLOT_HOST, -1, {te, uc[0]}); e->h_time1w = e->slots["unet.time_embedding.linear_1.weight"].host_off;
  B.slot("unet.time_embedding.linear_1.bias", SLOT_HOST, -1, {te}); e->h_time1b = e->slots["unet.time_embedding.linear_1.bias"].host_off;
  B.slot("unet.time_embedding.linear_2.weight", SLOT_HOST, -1, {te, te}); e->h_time2w = e->slots["unet.time_embedding.linear_2.weight"].host_off;
  B.slot("unet.time_embedding.linear_2.bias", SLOT_HOST, -1, {te}); e->h_time2b = e->slots["unet.time_embedding.linear_2.bias"].host_off;
  const int bd = c.bbox_embeddings_input_dim;
  B.slot("unet.bbox_embedding.linear_1.weight", SLOT_HOST, -1, {te, bd}); e->h_bbox1w = e->slots["unet.bbox_embedding.linear_1.weight"].host_off;
  B.slot("unet.bbox_embedding.linear_1.bias", SLOT_HOST, -1, {te}); e->h_bbox1b = e->slots["unet.bbox_embedding.linear_1.bias"].host_off;
  B.slot("unet.bbox_embedding.linear_2.weight", SLOT_HOST, -1, {te, te}); e->h_bbox2w = e->slots["unet.bbox_embedding.linear_2.weight"].host_off;
  B.slot("unet.bbox_embedding.linear_2.bias", SLOT_HOST, -1, {te}); e->h_bbox2b = e->slots["unet.bbox_embedding.linear_2.bias"].host_off;
  // point prompts (replace.py:198,446-450): TimestepEmbedding(point_embeddings_input_dim -> 1280), host-side like bbox_embedding
  const int pd = c.point_embeddings_input_dim;
  B.slot("unet.point_embedding.linear_1.weight", SLOT_HOST, -1, {te, pd}); e->h_point1w = e->slots["unet.point_embedding.linear_1.weight"].host_off;
  B.slot("unet.point_embedding.linear_1.bias", SLOT_HOST, -1, {te}); e->h_point1b = e->slots["unet.point_embedding.linear_1.bias"].host_off;
  B.slot("unet.point_embedding.linear_2.weight", SLOT_HOST, -1, {te, te}); e->h_point2w = e->slots["unet.point_embedding.linear_2.weight"].host_off;
  B.slot("unet.point_embedding.linear_2.bias", SLOT_HOST, -1, {te}); e->h_point2b = e->slots["unet.point_embedding.linear_2.bias"].host_off;
  e->u_down_res.resize(4); e->u_down_tf.resize(4);
  cprev = uc[0];
  for (int i = 0; i < 4; ++i) {
    for (int j = 0; j < c.unet_layers_per_block; ++j) {
      e->u_down_res[i].push_back(B.resnet("unet.down_blocks." + S(i) + ".resnets." + S(j), j == 0 ? cprev : uc[i], uc[i], te));
      if (i < 3) e->u_down_tf[i].push_back(B.transformer("unet.down_blocks." + S(i) + ".attentions." + S(j), uc[i], c.unet_heads[i], ctx));
    }
    cprev = uc[i];
    if (i < 3) e->u_down_ds.push_back(B.conv_named("unet.down_blocks." + S(i) + ".downsamplers.0.conv", 9, uc[i], uc[i]));
  }
  e->u_mid0 = B.resnet("unet.mid_block.resnets.0", uc[3], uc[3], te);
  e->u_midtf = B.transformer("unet.mid_block.attentions.0", uc[3], c.unet_heads[3], ctx);
  e->u_mid1 = B.resnet("unet.mid_block.resnets.1", uc[3], uc[3], te);
  e->u_up_res.resize(4); e->u_up_tf.resize(4);
  int output_channel = uc[3];
  const int nl = c.unet_layers_per_block + 1;
  for (int i = 0; i < 4; ++i) {
    const int prev_out = output_channel;
    output_channel = uc[3 - i];
    const int input_channel = uc[3 - std::min(i + 1, 3)];
    for (int j = 0; j < nl; ++j) {
      const int res_skip = (j == nl - 1) ? input_channel : output_channel;
      const int resnet_in = (j == 0) ? prev_out : output_channel;
      e->u_up_res[i].push_back(B.resnet("unet.up_blocks." + S(i) + ".resnets." + S(j), resnet_in + res_skip, output_channel, te));
      if (i > 0) e->u_up_tf[i].push_back(B.transformer("unet.up_blocks." + S(i) + ".attentions." + S(j), output_channel, c.unet_heads[3 - i], ctx));
    }
    if (i < 3) e->u_up_us.push_back(B.conv_named("unet.up_blocks." + S(i) + ".upsamplers.0.conv", 9, output_channel, output_channel));
  }
  e->u_norm_out = B.norm_named("unet.conv_norm_out", uc[0]);
  e->u_conv_out = B.conv_named("unet.conv_out", 9, uc[0], c.unet_out_channels);
  e->canon_bytes = B.woff;
  e->warena_bytes = B.woff + B.doff;
}

// ------------------------------------------------------------------------------------------------
// arena
// ------------------------------------------------------------------------------------------------
static T talloc(sdm_ctx* e, int N, int H, int W, int C, int f32) {
  T t; t.N = N; t.H = H; t.W = W; t.C = C; t.f32 = f32;
  t.bytes = rupz((f32 == kFmtP3 ? p3_rows_pad((size_t)N * H * W) : (size_t)N * H * W) * C * fmt_bytes(f32), 256);      // (P3 planes are blocked: rows padded to 32)
  // first fit in the free list
  for (auto it = e->freelist.begin(); it != e->freelist.end(); ++it) {
    if (it->second >= t.bytes) {
      t.off = it->first;
      const size_t rem = it->second - t.bytes;
      e->freelist.erase(it);
      if (rem) e->freelist[t.off + t.bytes] = rem;
      t.p = e->dry ? nullptr : e->arena + t.off;
      return t;
    }
  }
  t.off = e->arena_top;
  e->arena_top += t.bytes;
  e->peak = std::max(e->peak, e->arena_top);
  t.p = e->dry ? nullptr : e->arena + t.off;
  return t;
}

static void tfree_raw(sdm_ctx* e, size_t off, size_t sz);
static void tfree(sdm_ctx* e, T& t) {
  if (!t.bytes) return;
  if (t.sbytes) { tfree_raw(e, t.soff, t.sbytes); t.sbytes = 0; t.stats = nullptr; }
  if (t.cm_bytes) { tfree_raw(e, t.cm_off, t.cm_bytes); t.cm_bytes = 0; t.cmask = nullptr; }
  tfree_raw(e, t.off, t.bytes);
  t.bytes = 0; t.p = nullptr;
}
static void tfree_raw(sdm_ctx* e, size_t off, size_t sz) {
  auto nx = e->freelist.lower_bound(off);
  if (nx != e->freelist.begin()) {
    auto pv = std::prev(nx);
    if (pv->first + pv->second == off) { off = pv->first; sz += pv->second; e->freelist.erase(pv); }
  }
  nx = e->freelist.lower_bound(off + sz);
  if (nx != e->freelist.end() && nx->first == off + sz) { sz += nx->second; e->freelist.erase(nx); }
  if (off + sz == e->arena_top) e->arena_top = off;
  else e->freelist[off] = sz;
}

// mark t so that the conv producing it also emits the GroupNorm statistics of its consumer (op_conv allocates the
// partial-row buffer once the tile configuration, hence the number of rows, is known)
static int tstats(sdm_ctx* e, T& t) { (void)e; t.want_stats = true; return 0; }

static void arena_reset(sdm_ctx* e) { e->freelist.clear(); e->arena_top = 0; }

// ------------------------------------------------------------------------------------------------
// profiling helpers
// ------------------------------------------------------------------------------------------------
static void prof_begin(sdm_ctx* e, const char* name, double flops, double bytes, const std::string& desc = std::string()) {
  if (!e->prof_on || e->dry) return;
  ProfRec r; r.name = name; r.desc = desc; r.flops = flops; r.bytes = bytes;
#ifndef SDM_EMU
  (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
  (void)hipEventRecord(r.e0, (hipStream_t)e->stream);
#endif
  e->prof.push_back(r);
}
static void prof_end(sdm_ctx* e) {
  if (!e->prof_on || e->dry) return;
#ifndef SDM_EMU
  (void)hipEventRecord(e->prof.back().e1, (hipStream_t)e->stream);
#endif
}

// ------------------------------------------------------------------------------------------------
// operators
// ------------------------------------------------------------------------------------------------
struct ConvArgs {
  const T* in0 = nullptr; const T* in1 = nullptr;
  int up = 0, stride = 1, pad_mode = 0;
  T* out = nullptr;            // pre-allocated output (shape/dtype/C define the store)
  int out_ch_off = 0, cout_valid = -1;
  int lo_cols = -1;            // plane outputs (T::f32 2 / 3): output channels that need the low-part plane (-1: all)
  const T* res = nullptr;
  float out_scale = 1.0f;
  const float* bias_override = nullptr; const int* bias_sel = nullptr;
  int force_cfg = -1;
  const float* gn_scale = nullptr; const float* gn_shift = nullptr; int gn_silu = 0;   // fused GroupNorm apply (tile cfg 0 only)
};

static int op_conv(sdm_ctx* e, const ConvL& L, const ConvArgs& a) {
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.in0 = a.in0->p; p.C0 = a.in0->C; p.in_f32 = a.in0->f32;
  if (a.in1) { p.in1 = a.in1->p; p.C1 = a.in1->C; }
  p.N = a.in0->N; p.Hin = a.in0->H; p.Win = a.in0->W; p.up = a.up;
  p.Hout = a.out->H; p.Wout = a.out->W;
  p.pad_t = p.pad_l = (a.pad_mode == 0) ? 1 : 0;
  p.M = a.out->rows();
  p.w = L.w; p.bias = a.bias_override ? a.bias_override : L.b; p.bias_sel = a.bias_sel;
  p.Cout_pad = L.Cout_pad;
  p.out = a.out->p; p.out_f32 = a.out->f32; p.Cout_store = a.out->C;
  p.out_lo_off = (size_t)a.out->rows() * a.out->C;
  p.lo_cols = a.lo_cols >= 0 ? a.lo_cols : (1 << 30);
  p.acc_scale = 1.0f;
  if (L.split) {
    if (!p.in_f32) SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: split-precision layers take fp32 activations", L.name.c_str());
    p.w_lo = L.w_lo; p.acc_scale = ldexpf(1.0f, -L.w_exp);
  }
  const int nout = L.geglu ? L.Cout_pad / 2 : L.Cout_pad;
  p.Cout_valid = a.cout_valid >= 0 ? a.cout_valid : std::min(nout, a.out->C - a.out_ch_off);
  p.out_ch_off = a.out_ch_off;
  if (a.res) { p.res = a.res->p; p.res_f32 = a.res->f32; p.res_C = a.res->C; }
  p.epi = L.geglu; p.out_scale = a.out_scale;
  p.epi_mode = conv_epi_mode();
  p.xtile = conv_xtile_enabled() ? 1 : 0;
  p.gn_scale = a.gn_scale; p.gn_shift = a.gn_shift; p.gn_silu = a.gn_silu;
  // the F8 3x3 kernel addresses both tables through ONE buffer descriptor: scale | shift are the two halves of one scratch tensor (gn_scale_shift)
  if (!e->dry && a.gn_scale && a.gn_shift != a.gn_scale + (size_t)a.in0->N * (a.in0->C + (a.in1 ? a.in1->C : 0)))
    SDM_FAIL(e, SDM_ERR_STATE, "fused GroupNorm: shift table is not scale + N * C");
  if ((long)a.in0->rows() >= (1L << 31) || p.M >= (1L << 31)) SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: tensor too large for 32-bit pixel indices", L.name.c_str());
  {   // 3x3: per-tile descriptors span only the rows of the tile's halo (k_conv.h band0), so an image may exceed 4 GB; what
      // stays 32-bit is the byte offset inside that band
    const size_t es_in = p.in_f32 ? 4 : 2;
    if (L.ntaps == 9 && (size_t)40 * p.Win * (size_t)std::max(p.C0, p.C1) * es_in >= 0xFFFF0000ull)
      SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: a %d-pixel-wide row band of the operand exceeds a buffer descriptor", L.name.c_str(), p.Win);
  }
  if (p.C0 + p.C1 != L.Cin_pad) SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: input channels %d+%d != %d", L.name.c_str(), p.C0, p.C1, L.Cin_pad);
  if (a.in1 && a.in1->f32 != a.in0->f32) SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: concat sources differ in dtype", L.name.c_str());
  p.f8_hint = (L.f8 && L.w_dma && p.in_f32 && L.ntaps == 9 && a.stride == 1) ? 1 : 0;
  int cfg = a.force_cfg >= 0 ? a.force_cfg : conv_pick_cfg(L.ntaps, a.stride, p);
  if (cfg < 0 || cfg >= conv_num_cfgs(L.ntaps, a.stride) || !conv_cfg_ok(conv_cfg_table(L.ntaps, a.stride)[cfg], p))
    SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: no tile configuration for Cin=%d+%d (cfg %d)", L.name.c_str(), p.C0, p.C1, cfg);
  if (a.gn_scale && !(L.ntaps == 9 && a.stride == 1 && (cfg == 0 || cfg == 4 || cfg == 5) && p.C0 + p.C1 <= 1024))
    SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: fused GroupNorm requested for an unsupported tile configuration", L.name.c_str());
  // weights by LDS-DMA (256x128 tile, 3x3 stride 1): the layer keeps a stage-ordered copy of its weights for that kernel
  const bool dma_off = opt("conv_dma") == 0;      // A/B option
  if (!dma_off && L.ntaps == 9 && a.stride == 1 && cfg == 0 && L.w_dma && (!L.split || p.in_f32)) p.w_dma = L.w_dma;
  {   // producer / consumer form of the split-precision DMA kernel (k_conv.h, PC): the option conv_pc = 0 / 1 forces it off / on
    const int pc_opt = opt("conv_pc"), pc_min_cin = opt("conv_pc_min_cin");
    if (p.w_dma && L.split) p.pc = pc_opt >= 0 ? (pc_opt == 1) : (L.Cin_pad >= pc_min_cin);
    if (p.w_dma && L.f8) {
      // the fp8-residual kernel stages 32-channel chunks: a channel concat that does not split on a chunk boundary takes the
      // register-staged split kernel (K16 weights) instead
      if (p.C1 > 0 && (p.C0 % 32)) { p.w_dma = nullptr; p.pc = 0; }
      else p.f8 = 1;
    }
    if (L.ntaps == 1 && L.f8 && L.w_dma && cfg == 4 && p.in_f32 && gemm_f8_enabled()) { p.w_dma = L.w_dma; p.f8 = 1; }
    // F8 launches: x_lo8 * w8 = (x_lo * 2^11)(w * 2^e8), x8 * w_lo8 = x (w_lo * 2^e8 * 2^11), e8 = the layer's own e4m3 scale: both residual
    // sums are 2^(11 + e8) too large (E8M0 operand scales); the fp16 high parts are packed unscaled -> the accumulators are in the
    // output's unit (acc_scale 1; the K16 hi | lo copy of the same layer, used by the other kernels, keeps 2^-w_exp)
    if (p.f8) { p.f8_sa = 127 - 11; p.f8_sb = 127 - L.f8_exp; p.acc_scale = 1.0f; }
  }
  // split-K (register-staged kernels only): partial sums into an arena workspace, finished by splitk_reduce_kernel
  int ksplit = 1;
  if (!p.w_dma && !L.geglu && p.out_f32 <= 1 && (L.ntaps == 9 || p.M == (long)p.N * p.Hout * p.Wout)) ksplit = conv_pick_ksplit(L.ntaps, a.stride, cfg, p);
  T ws; ws.bytes = 0;
  const int sk_bpi = sdm_cdiv(p.Hout * p.Wout, kSplitKRows);
  if (ksplit > 1 && (size_t)ksplit * p.M * p.Cout_pad >= (1ull << 31)) ksplit = 1;
  if (a.out->want_stats) {
    if (L.geglu || a.out_ch_off || p.out_f32 >= 2) SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: fused statistics unsupported with this epilogue", L.name.c_str());
    const ConvCfgInfo& ci = conv_cfg_table(L.ntaps, a.stride)[cfg];
    const long tiles = (L.ntaps == 9) ? (long)sdm_cdiv(p.Hout, ci.TH) * sdm_cdiv(p.Wout, ci.TW)
                                      : ((long)p.Hout * p.Wout + ci.TH * ci.TW - 1) / (ci.TH * ci.TW);
    // split-K: the statistics come from the reduce kernel, one partial row per kSplitKRows rows of an image
    a.out->srows = ksplit > 1 ? sk_bpi : (int)(tiles * ci.WM);
    T sb = talloc(e, 1, 1, 1, (int)((size_t)p.N * a.out->srows * a.out->C * 2), 1);
    a.out->soff = sb.off; a.out->sbytes = sb.bytes; a.out->stats = (float*)sb.p;
    p.stats = a.out->stats;
  }
  if (ksplit > 1) ws = talloc(e, 1, 1, 1, (int)((size_t)ksplit * p.M * p.Cout_pad), 1);
  // class plane of a piecewise-constant batch (the trimap images of the VAE encoder, k_misc.h cmask_*): eroded through this conv's window; on the
  // F8 kernel's full-tile fp32 path the output tiles that lie inside one region are not multiplied - one representative per (image, class) is,
  // and const_tile_fill_kernel copies it (and its statistics) into the others.  (cm_bytes, not the pointer, drives every decision: the dry pass
  // has no pointers and must allocate the same way.)
  // (no plane behind the last level that can leave tiles out: resolutions only shrink from here, up-sampling convs do not propagate)
  const bool cm_prop = a.in0->cm_bytes != 0 && !a.in1 && L.ntaps == 9 && !a.up && opt("trimap_skip") != 0 && p.Hout >= opt("trimap_skip_min_rows");
  bool cm_skip = false;
  T cm_flag, cm_rep;
  const int cm_tiles = sdm_cdiv(p.Hout, 8) * sdm_cdiv(p.Wout, 32);
  if (cm_prop) {
    T mb = talloc(e, 1, 1, 1, (int)(((size_t)p.N * p.Hout * p.Wout + 3) / 4), 1);
    a.out->cmask = (unsigned char*)mb.p; a.out->cm_off = mb.off; a.out->cm_bytes = mb.bytes; a.out->cm_n0 = a.in0->cm_n0;
    cm_skip = p.f8 && p.w_dma && cfg == 0 && a.stride == 1 && p.Hout % 8 == 0 && p.Wout % 32 == 0 && p.out_f32 == 1 && !L.geglu && p.out_scale == 1.0f &&
              p.Cout_valid == p.Cout_pad && p.Cout_pad % 128 == 0 && p.Cout_store == p.Cout_pad && p.out_ch_off == 0 && (!p.res || p.res_f32) && ksplit == 1;
    if (cm_skip) {
      cm_flag = talloc(e, 1, 1, 1, (p.N * cm_tiles + 3) / 4, 1);
      cm_rep = talloc(e, 1, 1, 1, p.N * 8, 1);
      p.tile_flag = (const unsigned char*)cm_flag.p; p.tile_rep = (const int*)cm_rep.p;
    }
  }
  if (e->dry) { if (ksplit > 1) tfree(e, ws); if (cm_skip) { tfree(e, cm_rep); tfree(e, cm_flag); } return 0; }
  const double flops = 2.0 * (double)p.M * L.O * L.I * L.ntaps;
  const double bytes = (double)a.in0->rows() * L.Cin_pad * (p.in_f32 ? 4 : 2) + (double)p.M * p.Cout_valid * (p.out_f32 ? 4 : 2) +
                       (double)L.Cin_pad * L.ntaps * L.Cout_pad * 2 + (a.res ? (double)p.M * p.Cout_valid * (p.res_f32 ? 4 : 2) : 0.0);
  if (e->prof_on) {
    char d[256];
    snprintf(d, sizeof(d), "%s N=%d Hout=%d Wout=%d Cin=%d Cout=%d s=%d up=%d f32in=%d cfg=%d", L.name.c_str(), p.N, p.Hout, p.Wout, L.Cin_pad, L.O, a.stride,
             a.up, p.in_f32, cfg);
    // thin convs (<= 16 real input or output channels: conv_in/conv_out, latent convs, folded cross-attn K|V) are HBM-bound
    // (SURVEY.md 2.2 "thin convs"); everything else is MFMA-bound
    const bool thin = (L.ntaps == 9) && (L.I <= 16 || L.O <= 16);
    prof_begin(e, L.ntaps == 9 ? (thin ? "conv3x3_thin" : "conv3x3_mfma") : "gemm_mfma", flops, bytes, d);
  } else {
    prof_begin(e, "conv", flops, bytes);
  }
  int rc = 0;
  // GEMMs that emit per-image GroupNorm statistics for a batch: row tiles aligned to images inside ONE launch
  if (L.ntaps == 1 && p.stats && p.N > 1) p.rows_per_img = p.Hout * p.Wout;
  if (cm_prop) {
    const int n0 = a.in0->cm_n0;
    if (cm_skip) {
      SDM_CHECK_DEV(e, dev_memset(cm_rep.p, 0x7f, (size_t)p.N * 8 * 4, e->stream));
      if (n0 > 0) SDM_CHECK_DEV(e, dev_memset(cm_flag.p, 0, (size_t)n0 * cm_tiles, e->stream));
    }
    SDM_LAUNCH(cmask_conv_kernel, dim3((unsigned)cm_tiles, (unsigned)(p.N - n0), 1), dim3(256), 0, e->stream, (const unsigned char*)a.in0->cmask, a.in0->H, a.in0->W, a.out->cmask,
               p.Hout, p.Wout, a.stride, p.pad_t, p.pad_l, cm_skip ? (unsigned char*)cm_flag.p : (unsigned char*)nullptr, cm_skip ? (int*)cm_rep.p : (int*)nullptr, n0);
    if (cm_skip) count_kernel("conv3x3_f8_const_tiles");
  }
  if (ksplit > 1) {
    count_kernel(L.ntaps == 9 ? "conv3x3_splitk" : "gemm_splitk");
    rc = launch_conv_splitk(L.ntaps, a.stride, cfg, p, ksplit, (float*)ws.p, e->stream);
    tfree(e, ws);
  } else {
    rc = launch_conv(L.ntaps, a.stride, cfg, p, e->stream);
  }
  if (cm_skip) {
    if (rc == 0) SDM_LAUNCH(const_tile_fill_kernel, dim3((unsigned)cm_tiles, (unsigned)(p.N - a.in0->cm_n0), 1), dim3(256), 0, e->stream, (float*)p.out, p.Cout_store, p.Hout, p.Wout,
                            p.tile_flag, p.tile_rep, p.stats, 2, a.in0->cm_n0);
    tfree(e, cm_rep); tfree(e, cm_flag);
  }
  if (rc != 0) SDM_FAIL(e, SDM_ERR_INVALID, "conv %s: bad cfg", L.name.c_str());
#ifndef SDM_EMU
  { const hipError_t le = hipGetLastError(); if (le != hipSuccess) SDM_FAIL(e, SDM_ERR_HIP, "conv %s: launch failed: %s", L.name.c_str(), hipGetErrorString(le)); }
#endif
  prof_end(e);
  return 0;
}

// GroupNorm statistics -> per-(image, channel) scale = rstd*gamma and shift = beta - mean*rstd*gamma.
// scratch layout: [N][max(groups,C)][2] doubles (sums) | scale [N][C] floats | shift [N][C] floats.  The caller frees `scratch`.
static int gn_scale_shift(sdm_ctx* e, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int HW, int groups, const float* gamma,
                          const float* beta, float eps, const float* st0, int rows0, const float* st1, int rows1, bool have_stats, T* scratch,
                          float** scale_out, float** shift_out) {
  const int C = C0 + C1;
  if (C % 8 || (C / groups) * groups != C || C0 % 8) SDM_FAIL(e, SDM_ERR_INVALID, "groupnorm: bad channels %d+%d", C0, C1);
  const size_t sum_bytes = (size_t)N * std::max(groups, C) * 16;
  *scratch = talloc(e, 1, 1, 1, (int)((sum_bytes + (size_t)N * C * 8 + 3) / 4), 1);
  *scale_out = nullptr; *shift_out = nullptr;
  if (e->dry) return 0;
  double* sums = (double*)scratch->p;
  float* scale = (float*)((unsigned char*)scratch->p + sum_bytes);
  float* shift = scale + (size_t)N * C;
  *scale_out = scale; *shift_out = shift;
  if (have_stats) {
    prof_begin(e, "gn_reduce", 0, ((double)rows0 * C0 + (double)rows1 * C1) * N * 8);
    SDM_LAUNCH(gn_partials_scale_shift_kernel, dim3(groups, N), dim3(GN_PSS_THREADS), 0, e->stream, st0, rows0, st1, rows1, C0, C1, gamma, beta, scale, shift,
               groups, (long)HW, eps);
    prof_end(e);
  } else {
    GnSrc s; s.in0 = in0; s.in1 = in1; s.C0 = C0; s.C1 = C1; s.in_f32 = in_f32; s.HW = HW;
    const int CV = C / 8;
    const int slots = std::max(1, 256 / CV);
    const int threads = rup(CV * slots, 64);
    int ppb = std::max(slots * 8, sdm_cdiv(HW, 2048 / std::max(1, N)));
    ppb = rup(ppb, slots);
    const int nb = sdm_cdiv(HW, ppb);
    SDM_CHECK_DEV(e, dev_memset(sums, 0, (size_t)N * groups * 16, e->stream));
    prof_begin(e, "gn_stats", 0, (double)N * HW * C * (in_f32 ? 4 : 2));
    SDM_LAUNCH(gn_stats_kernel, dim3(nb, N), dim3(threads), (size_t)2 * C * 4, e->stream, s, sums, groups, ppb);
    prof_end(e);
    SDM_LAUNCH(gn_finalize_kernel, dim3(sdm_cdiv(N * C, 256)), dim3(256), 0, e->stream, (const double*)sums, gamma, beta, scale, shift, N, C,
               groups, (long)HW * (C / groups), eps);
  }
  return 0;
}

static int op_groupnorm_raw(sdm_ctx* e, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int HW, int groups,
                            const float* gamma, const float* beta, float eps, int silu, void* out, int out_f32, const float* st0 = nullptr,
                            int rows0 = 0, const float* st1 = nullptr, int rows1 = 0, bool have_stats = false) {
  const int C = C0 + C1;
  T scratch; float* scale; float* shift;
  TRY(gn_scale_shift(e, in0, in1, C0, C1, in_f32, N, HW, groups, gamma, beta, eps, st0, rows0, st1, rows1, have_stats, &scratch, &scale, &shift));
  if (!e->dry) {
    GnSrc s; s.in0 = in0; s.in1 = in1; s.C0 = C0; s.C1 = C1; s.in_f32 = in_f32; s.HW = HW;
    const int CV = C / 8;
    const int slots = std::max(1, 256 / CV);
    const int threads = rup(CV * slots, 64);
    int ppb = std::max(slots * 8, sdm_cdiv(HW, 2048 / std::max(1, N)));
    ppb = rup(ppb, slots);
    const int nb = sdm_cdiv(HW, ppb);
    prof_begin(e, "gn_apply", 0, (double)N * HW * C * (in_f32 ? 4 : 2) + (double)N * HW * C * (double)fmt_bytes(out_f32));
    if (out_f32 == kFmtP3)
      SDM_LAUNCH(gn_apply_p3_kernel, dim3((unsigned)(((long)N * HW + 15) / 16)), dim3(256), 0, e->stream, s, (const float*)scale, (const float*)shift, (unsigned char*)out,
                 silu, (long)N * HW);
    else
    SDM_LAUNCH(gn_apply_kernel, dim3(nb, N), dim3(threads), 0, e->stream, s, (const float*)scale, (const float*)shift, out, out_f32, silu, ppb);
    prof_end(e);
  }
  tfree(e, scratch);
  return 0;
}

static int op_gn(sdm_ctx* e, const NormL& n, const T& x, const T* x2, int silu, float eps, T* out, int out_fmt = -1) {
  const int C = x.C + (x2 ? x2->C : 0);
  if (C != n.C) SDM_FAIL(e, SDM_ERR_INVALID, "groupnorm: C %d != %d", C, n.C);
  *out = talloc(e, x.N, x.H, x.W, C, out_fmt >= 0 ? out_fmt : e->act_f32);
  const bool hs = x.sbytes && (!x2 || x2->sbytes);     // statistics already produced by the conv epilogue(s)
  return op_groupnorm_raw(e, x.p, x2 ? x2->p : nullptr, x.C, x2 ? x2->C : 0, x.f32, x.N, x.H * x.W, e->cfg.groups, n.g, n.b, eps, silu,
                          out->p, out->f32, x.stats, x.srows, x2 ? x2->stats : nullptr, x2 ? x2->srows : 0, hs);
}

static int op_ln(sdm_ctx* e, const NormL& n, const T& x, float eps, T* out, int out_fmt = -1) {
  if (x.C != n.C || x.C % 64 || x.C > 64 * SDM_LN_MAXV) SDM_FAIL(e, SDM_ERR_INVALID, "layernorm: unsupported C %d", x.C);
  *out = talloc(e, x.N, x.H, x.W, x.C, out_fmt >= 0 ? out_fmt : e->act_f32);
  if (e->dry) return 0;
  const long rows = x.rows();
  if (out->f32 == kFmtP3) {
    if (x.f32 != 1 || x.C > 128 * SDM_LNP_MAXI) SDM_FAIL(e, SDM_ERR_INVALID, "layernorm (P3): fp32 input, C <= %d expected", 128 * SDM_LNP_MAXI);
    prof_begin(e, "layernorm", 0, (double)rows * x.C * 7);
    SDM_LAUNCH(layernorm_p3_kernel, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, e->stream, (const float*)x.p, (const float*)n.g, (const float*)n.b,
               (unsigned char*)out->p, rows, x.C, eps);
    prof_end(e);
    return 0;
  }
  prof_begin(e, "layernorm", 0, (double)rows * x.C * ((x.f32 ? 4 : 2) + (out->f32 ? 4 : 2)));
  SDM_LAUNCH(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, e->stream, (const void*)x.p, x.f32, (const float*)n.g,
             (const float*)n.b, out->p, out->f32, rows, x.C, eps);
  prof_end(e);
  return 0;
}

// q/k/v are views into fp16 row-major buffers; v is transposed into an arena scratch first
// Precise variant (d = 64): q / k / v point at the HIGH planes of split fp16 pairs, the low planes follow at element offsets
// q_lo / k_lo / v_lo (written by the producing GEMM with out_f32 == 2); `out` is fp16 or fp32 (out_f32).
// prec: 0 fp16 operands; 1 q / k / v as fp16 planes hi | lo; 2 q / k as fp16 plane + e5m2 pair plane (ConvParams::out_f32 == 3), v hi only
struct AttnPrec { int prec = 0; long q_lo = 0, k_lo = 0, v_lo = 0; int out_f32 = 0; int out_p3 = 0; };      // out_p3 (with out_f32 = 1): `out` is a P3 tensor (k_gemm.h)
static int op_attention_raw(sdm_ctx* e, const half_t* q, int ldq, const half_t* k, int ldk, const half_t* v, int ldv, const float* bias_l2,
                            int B, int heads, int Lq, int Lk, int D, void* out, int ldo, bool q_prescaled = false, const int* tiles = nullptr,
                            AttnPrec ap = AttnPrec()) {
  if (!(D == 64 || (D == 512 && heads == 1))) SDM_FAIL(e, SDM_ERR_INVALID, "attention: unsupported head dim %d x %d heads", D, heads);
  if ((ldq | ldk | ldv | ldo) % 8) SDM_FAIL(e, SDM_ERR_INVALID, "attention: row strides must be multiples of 8");
  if (ap.prec && D != 64) SDM_FAIL(e, SDM_ERR_INVALID, "attention: the split-precision variant exists for head dim 64 only");
  // the fp8-pair form cannot rescale Q inside the kernel (its pair plane is produced pre-scaled, as the engine always does)
  if (ap.prec == 2 && !q_prescaled) SDM_FAIL(e, SDM_ERR_INVALID, "attention: the fp8-residual form takes pre-scaled queries");
  const int ldvt = rup(Lk, 64);
  T vt = talloc(e, (ap.prec ? 2 : 1) * B, heads, D, ldvt, 0);      // precise: V^T_hi planes of all images, then V^T_lo
  // key tiles whose bias underflows the softmax are skipped (exact, AttnParams::tiles); the engine passes one list per U-Net
  // level, the stand-alone operator entry builds it here.  The option attn_dense = 1 walks every tile (A/B hook).
  const bool dense_attn = opt("attn_dense") != 0;
  const int ntiles64 = sdm_cdiv(Lk, 64);
  T tl_own;
  const bool own_list = (D == 64) && bias_l2 && !tiles && !dense_attn;
  if (own_list) tl_own = talloc(e, B, 1, 1, ntiles64 + 1, 1);
  // Key split (k_attn.h AttnParams::nsplit; split-precision d = 64 with fp32 output only).  A launch whose blocks fill the chip's block slots 1.25 times
  // takes as long as one that fills them twice; walking half (a quarter) of the keys per block and combining the partial sums afterwards turns
  // that into 2.5 (5) rounds of half (quarter) length.  Chosen by block count alone - the same for the dense and the tile-list walk, whose ranges are key
  // ranges - and only for long walks (>= 32 tiles per part).  One image at 1024^2: 640 four-wave blocks on 512 slots at the first U-Net level.
  int nsplit = 1;
  T part_o, part_ml;
  if (D == 64 && ap.prec && ap.out_f32 && !(ap.prec == 1 && opt("attn_pv_split"))) {
    const int force_nw = opt("attn_nw");
    // ping-pong kernel: always 256-row blocks, one per CU; launches of fewer than 128 such blocks (the 16^2 level: 80) keep the 4-wave pipelines, whose 128-row
    // blocks spread over twice as many CUs (measured 0.63 vs 0.79 ms at B = 4, profiles/r06_attn_pp_lab.txt)
    const bool pp = ap.prec == 2 && opt("attn_pp") != 0 && !force_nw && Lk % 64 == 0 && (long)B * heads * sdm_cdiv(Lq, 256) >= opt("attn_pp_min_blocks");
    const bool nw8 = pp || (force_nw ? (force_nw == 8) : ((long)B * heads * sdm_cdiv(Lq, 256) >= 1024));
    const long blocks = (long)B * heads * 8 * sdm_cdiv(sdm_cdiv(Lq, nw8 ? 256 : 128), 8), slots = (long)device_cus() * (nw8 ? 1 : 2);
    const int o = opt("attn_ksplit");
    if (o >= 2) nsplit = (o == 2 || o == 4) ? o : 1;
    else if (o == 0) {
      auto rounds = [&](int s) { return (double)((blocks * s + slots - 1) / slots) / s; };
      for (int s = 2; s <= 4; ++s)      // (3: 80 one-per-CU blocks - the 16^2 level's cross-attentions - become 240 of a third of the length)
        if ((s != 3 || pp) && ntiles64 / s >= 32 && rounds(s) < 0.85 * rounds(nsplit)) nsplit = s;
    }
    if (nsplit > 1) {
      part_o = talloc(e, nsplit * B, Lq, 1, ldo, 1);
      part_ml = talloc(e, nsplit * B, heads, Lq, 2, 1);
    }
  }
  if (!e->dry) {
    if (own_list) {
      SDM_LAUNCH(attn_active_tiles_kernel, dim3(B), dim3(256), 0, e->stream, bias_l2, Lk, ntiles64, (int*)tl_own.p, ntiles64 + 1, SDM_ATTN_SKIP_MARGIN);
      tiles = (const int*)tl_own.p;
    }
    if (dense_attn || D != 64 || !bias_l2) tiles = nullptr;
    const long vt_hs = (long)D * ldvt, vt_bs = (long)heads * vt_hs;
    prof_begin(e, "transpose_v", 0, (double)B * Lk * heads * D * 4);
    SDM_LAUNCH(transpose_v_kernel, dim3(ldvt / 64, heads * (D / 64), B), dim3(256), 0, e->stream, v, (long)Lk * ldv, ldv, (half_t*)vt.p, vt_bs,
               vt_hs, ldvt, Lk, D);
    // split-precision variant: Q.K^T on split operands, P.V on plain fp16 (k_attn.h, PREC = 2) unless SDM_ATTN_PV_SPLIT=1 asks for
    // the residual terms of P.V too (PREC = 1: then V^T_lo is needed as well)
    const bool pv_split = ap.prec == 1 && opt("attn_pv_split") != 0;
    if (pv_split)
      SDM_LAUNCH(transpose_v_kernel, dim3(ldvt / 64, heads * (D / 64), B), dim3(256), 0, e->stream, v + ap.v_lo, (long)Lk * ldv, ldv,
                 (half_t*)vt.p + (size_t)B * vt_bs, vt_bs, vt_hs, ldvt, Lk, D);
    prof_end(e);
    AttnParams p;
    memset(&p, 0, sizeof(p));
    p.q = q; p.q_bs = (long)Lq * ldq; p.ldq = ldq;
    p.k = k; p.k_bs = (long)Lk * ldk; p.ldk = ldk;
    p.vt = (const half_t*)vt.p; p.vt_bs = vt_bs; p.vt_hs = vt_hs; p.ldvt = ldvt;
    p.bias = bias_l2; p.bias_bs = Lk;
    p.tiles = tiles; p.tiles_bs = ntiles64 + 1;
    p.o = (half_t*)out; p.o_bs = (long)Lq * ldo; p.ldo = ldo; p.o_f32 = ap.out_f32;
    if (ap.out_p3) {
      if (!ap.out_f32 || Lq % 32 || ldo % 32 || D != 64) SDM_FAIL(e, SDM_ERR_INVALID, "attention: plane output needs d = 64, Lq %% 32 == 0 and C %% 32 == 0");
      p.o_p3 = 1; p.o_xl_off = (long)(p3_rows_pad((size_t)B * Lq) * ldo * 2);
    }
    p.q_lo = ap.q_lo; p.k_lo = ap.k_lo; p.vt_lo = (long)B * vt_bs;
    p.Lq = Lq; p.Lk = Lk;
    p.scale_log2e = q_prescaled ? 1.0f : (1.0f / sqrtf((float)D)) * SDM_LOG2E;      // engine: folded into the to_q weights (d = 64 only)
    if (nsplit > 1) { p.nsplit = nsplit; p.o = (half_t*)part_o.p; p.part_stride = (long)B * p.o_bs; p.part_ml = (float*)part_ml.p; }
    const unsigned gy = (unsigned)(nsplit > 1 ? nsplit : 1);
    double flops = 4.0 * B * heads * (double)Lq * Lk * D;
    const double bytes = 2.0 * B * heads * D * (2.0 * Lq + 2.0 * Lk);
    std::string adesc = "B=" + std::to_string(B) + " h=" + std::to_string(heads) + " Lq=" + std::to_string(Lq) + " Lk=" + std::to_string(Lk);
    if (e->prof_on && tiles) {      // profiling only: report EXECUTED flops - needs the tile counts on the host
      std::vector<int> hl((size_t)B * (ntiles64 + 1));
      (void)dev_sync(e->stream);
      (void)dev_memcpy_d2h(hl.data(), tiles, hl.size() * sizeof(int), e->stream);
      (void)dev_sync(e->stream);
      long act = 0;
      for (int bi = 0; bi < B; ++bi) act += hl[(size_t)bi * (ntiles64 + 1)];
      const double frac = (double)act / ((double)B * ntiles64);
      flops *= frac;
      char fb[48];
      snprintf(fb, sizeof(fb), " active_key_tiles=%.3f", frac);
      adesc += fb;
    }
    if (D == 64) {
      prof_begin(e, "attn_d64", flops, bytes, adesc);
      // 8-wave blocks (256 queries share every K / V^T tile: half the L2 / Infinity-Cache traffic and LDS staging per MFMA) only
      // where they measured faster: the split-precision variant with >= 4 such blocks per CU (B=4 h=5 L=16384: 4.01 vs 4.25 ms;
      // h=10 Lq=4096: 2.39 vs 2.24 ms, i.e. slower; the fp16 variant is neutral to -10 %) - profiles/r02_ablate_attn_nw8.txt
      const int force_nw = opt("attn_nw");                          // A/B / test option: 4 or 8
      const bool pp = ap.prec == 2 && ap.out_f32 && opt("attn_pp") != 0 && !force_nw && Lk % 64 == 0 &&      // (LDS-DMA tiles: no masked tail rows)
                      (long)B * heads * sdm_cdiv(Lq, 256) >= opt("attn_pp_min_blocks");
      const bool nw8 = pp || (force_nw ? (force_nw == 8) : (ap.prec && (long)B * heads * sdm_cdiv(Lq, 256) >= 1024));
      const int qrows = nw8 ? 256 : 128;
      p.batch = B; p.heads = heads; p.nq_blocks = sdm_cdiv(Lq, qrows); p.q_chunks = 8;     // B*heads*8 units: always a multiple of 8
      const int qb = sdm_cdiv(p.nq_blocks, p.q_chunks);
      const unsigned nblk = (unsigned)(B * heads * p.q_chunks * qb);                         // 1-D grid, XCD-aware mapping in the kernel
      if (ap.prec == 2) {
        // 8-wave blocks with fp32 output (the engine's level-0 attentions): the two-tile software pipeline of the kernel (k_attn.h,
        // attn_d64_pipe_kernel: same arithmetic, bit-identical results, -9 % kernel time); option attn_pipe = 0 selects the plain form.
        const bool pipe8 = opt("attn_pipe") != 0, pipe4 = opt("attn_pipe4") != 0;
        if (pp) { count_kernel("attn_d64_pp"); p.pp_flags = opt("attn_pp") == 1 ? 1 : (opt("attn_pp") == 3 ? 2 : 0); const bool pb = p.bias != nullptr;
          if (p.tiles && pb) { auto kp = attn_d64_pp_kernel<0, 1, 1>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(512), ATTN64PP_SMEM, e->stream, p); }
          else if (pb) { auto kp = attn_d64_pp_kernel<0, 1, 0>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(512), ATTN64PP_SMEM, e->stream, p); }
          else { auto kp = attn_d64_pp_kernel<0, 0, 0>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(512), ATTN64PP_SMEM, e->stream, p); } }
        else if (nw8 && pipe8 && p.o_f32) { count_kernel("attn_d64_pipe<8>"); auto kp = attn_d64_pipe_kernel<8>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(512), ATTN64PIPE_SMEM, e->stream, p); }
        else if (!nw8 && pipe4 && p.o_f32) { count_kernel("attn_d64_pipe<4>"); auto kp = attn_d64_pipe_kernel<4>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(256), ATTN64PIPE4_SMEM, e->stream, p); }
        else if (nw8) { count_kernel("attn_d64<prec3,8>"); auto kp = attn_d64_kernel<1, 3, 8>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(512), ATTN64P_SMEM, e->stream, p); }
        else { count_kernel("attn_d64<prec3,4>"); auto kp = attn_d64_kernel<1, 3, 4>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(256), ATTN64P_SMEM, e->stream, p); }
      } else if (ap.prec && pv_split) {
        if (nw8) { count_kernel("attn_d64<prec1,8>"); auto kp = attn_d64_kernel<1, 1, 8>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(512), ATTN64P_SMEM, e->stream, p); }
        else { count_kernel("attn_d64<prec1,4>"); auto kp = attn_d64_kernel<1, 1, 4>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(256), ATTN64P_SMEM, e->stream, p); }
      } else if (ap.prec) {
        if (nw8) { count_kernel("attn_d64<prec2,8>"); auto kp = attn_d64_kernel<1, 2, 8>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(512), ATTN64P_SMEM, e->stream, p); }
        else { count_kernel("attn_d64<prec2,4>"); auto kp = attn_d64_kernel<1, 2, 4>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk, gy, 1), dim3(256), ATTN64P_SMEM, e->stream, p); }
      } else {
        if (nw8) { count_kernel("attn_d64<fp16,8>"); auto kf = attn_d64_kernel<1, 0, 8>; SDM_SET_SMEM(kf, 160 * 1024); SDM_LAUNCH(kf, dim3(nblk, gy, 1), dim3(512), ATTN64P_SMEM, e->stream, p); }
        else { count_kernel("attn_d64<fp16,4>"); SDM_LAUNCH((attn_d64_kernel<1, 0, 4>), dim3(nblk, gy, 1), dim3(256), ATTN64_SMEM, e->stream, p); }
      }
      if (nsplit > 1) {
        count_kernel("attn_combine");
        const long nthr = (long)B * Lq * heads * 16;
        if (ap.out_p3)
          SDM_LAUNCH(attn_combine_p3_kernel, dim3((unsigned)((nthr / 2 + 255) / 256)), dim3(256), 0, e->stream, (const float*)part_o.p, (const float*)part_ml.p,
                     (unsigned char*)out, (long)(p3_rows_pad((size_t)B * Lq) * ldo * 2), nsplit, (long)B * (long)Lq * ldo, B, heads, Lq, (long)Lq * ldo, ldo);
        else
        SDM_LAUNCH(attn_combine_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, e->stream, (const float*)part_o.p, (const float*)part_ml.p, (float*)out,
                   nsplit, (long)B * (long)Lq * ldo, B, heads, Lq, (long)Lq * ldo, ldo);
      }
      prof_end(e);
    } else {
      prof_begin(e, "attn_d512", flops, bytes);
      p.batch = B; p.heads = 1; p.nq_blocks = sdm_cdiv(Lq, 128); p.q_chunks = 8;
      const unsigned nblk = (unsigned)(B * p.q_chunks * sdm_cdiv(p.nq_blocks, p.q_chunks));
      SDM_SET_SMEM(attn_d512_kernel<0>, ATTN512P_SMEM);
      SDM_LAUNCH(attn_d512_kernel<0>, dim3(nblk), dim3(512), ATTN512P_SMEM, e->stream, p);
      prof_end(e);
    }
  }
  if (nsplit > 1) { tfree(e, part_ml); tfree(e, part_o); }
  if (own_list) tfree(e, tl_own);
  tfree(e, vt);
  return 0;
}


// ------------------------------------------------------------------------------------------------
// blocks
// ------------------------------------------------------------------------------------------------
static int conv_simple(sdm_ctx* e, int layer, const T& in, T* out, int Cout_store, int out_f32, int stride = 1, int pad_mode = 0, int up = 0,
                       const T* res = nullptr, float scale = 1.0f, bool want_stats = false, int lo_cols = -1) {
  const ConvL& L = e->convs[layer];
  int Ho = in.H << up, Wo = in.W << up;
  if (stride == 2) { Ho /= 2; Wo /= 2; }
  *out = talloc(e, in.N, Ho, Wo, Cout_store, out_f32);
  if (want_stats) TRY(tstats(e, *out));
  ConvArgs a; a.in0 = &in; a.out = out; a.stride = stride; a.pad_mode = pad_mode; a.up = up; a.res = res; a.out_scale = scale; a.lo_cols = lo_cols;
  return op_conv(e, L, a);
}

// Would op_conv pick tile cfg 0 (the one that has the fused-GroupNorm variant) for this 3x3 stride-1 conv?
static bool conv_can_fuse_gn(sdm_ctx* e, const ConvL& L, const T& x, const T* x2) {
  const bool off = opt("no_gn_fuse") != 0;        // A/B option
  if (off || L.ntaps != 9) return false;
  const int Cin = x.C + (x2 ? x2->C : 0);
  if (Cin > 1024 || Cin != L.Cin_pad) return false;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.C0 = x.C; p.C1 = x2 ? x2->C : 0; p.in_f32 = x.f32; p.N = x.N; p.Hin = x.H; p.Win = x.W; p.Hout = x.H; p.Wout = x.W; p.Cout_pad = L.Cout_pad;
  p.M = x.rows();
  p.f8_hint = (L.f8 && L.wdma_bytes && x.f32) ? 1 : 0;
  const int cfg = conv_pick_cfg(9, 1, p);
  return cfg == 0 || cfg == 4 || cfg == 5;        // the 256-pixel tiles that carry the fused-GroupNorm variant
  (void)e;
}

// GroupNorm(32)(+SiLU) followed by a 3x3 stride-1 conv.  When the conv runs on the 256x128 tile the normalisation is applied
// inside the conv's operand staging (no normalised copy of the activation is ever written); otherwise the stand-alone apply
// kernel produces the fp16 operand first.  `a` carries everything except the inputs.
static int gn_conv(sdm_ctx* e, const NormL& n, const ConvL& L, const T& x, const T* x2, int silu, float eps, ConvArgs a) {
  const int C = x.C + (x2 ? x2->C : 0);
  if (C != n.C) SDM_FAIL(e, SDM_ERR_INVALID, "groupnorm: C %d != %d", C, n.C);
  if (conv_can_fuse_gn(e, L, x, x2)) {
    const bool hs = x.sbytes && (!x2 || x2->sbytes);
    T scratch; float* scale; float* shift;
    TRY(gn_scale_shift(e, x.p, x2 ? x2->p : nullptr, x.C, x2 ? x2->C : 0, x.f32, x.N, x.H * x.W, e->cfg.groups, n.g, n.b, eps, x.stats, x.srows,
                       x2 ? x2->stats : nullptr, x2 ? x2->srows : 0, hs, &scratch, &scale, &shift));
    a.in0 = &x; a.in1 = x2;
    a.gn_scale = e->dry ? (const float*)16 : scale; a.gn_shift = shift; a.gn_silu = silu;
    int rc = op_conv(e, L, a);          // picks the same tile conv_can_fuse_gn saw
    tfree(e, scratch);
    return rc;
  }
  T h;
  TRY(op_gn(e, n, x, x2, silu, eps, &h));
  a.in0 = &h; a.in1 = nullptr;
  int rc = op_conv(e, L, a);
  tfree(e, h);
  return rc;
}

// ResnetBlock2D (Appendix A.3).  x (+ x2: channel concat) -> new stream tensor; frees nothing.
static int resblock(sdm_ctx* e, const ResB& r, const T& x, const T* x2, float eps, T* out) {
  const int sf = e->cfg.stream_f32;
  T h1 = talloc(e, x.N, x.H, x.W, r.cout, e->act_f32);
  TRY(tstats(e, h1));
  {
    ConvArgs a; a.out = &h1;
    if (r.temb >= 0) { a.bias_override = e->tembs[r.temb].table; a.bias_sel = e->d_bias_sel; }
    TRY(gn_conv(e, e->norms[r.norm1], e->convs[r.conv1], x, x2, 1, eps, a));
  }
  T xs; const T* resid = &x;
  if (r.sc >= 0) {
    xs = talloc(e, x.N, x.H, x.W, r.cout, sf);
    ConvArgs a; a.in0 = &x; a.in1 = x2; a.out = &xs;
    TRY(op_conv(e, e->convs[r.sc], a));
    resid = &xs;
  } else if (x2) {
    SDM_FAIL(e, SDM_ERR_INVALID, "resblock: concat input without shortcut");
  }
  *out = talloc(e, x.N, x.H, x.W, r.cout, sf);
  TRY(tstats(e, *out));
  {
    ConvArgs a; a.out = out; a.res = resid;
    TRY(gn_conv(e, e->norms[r.norm2], e->convs[r.conv2], h1, nullptr, 1, eps, a));
  }
  tfree(e, h1);
  if (r.sc >= 0) tfree(e, xs);
  return 0;
}

// ---- plane-fed GEMM path (k_gemm.h).  `in` is a P3 tensor (T::f32 == kFmtP3); the output format picks the epilogue: fp32 (+residual, +statistics),
//      P3 (GEGLU layers, or linear + residual), or the q | k | v operand planes of the attention cores (format 3) ----
static bool p3_ok(sdm_ctx* e, const ConvL& L) { return L.w3 != nullptr && e->act_f32 && opt("gemm_p3") != 0; }
static int op_gemm_p3(sdm_ctx* e, const ConvL& L, const T& in, T* out, const T* res, int lo_cols) {
  if (in.f32 != kFmtP3 || in.C != L.Cin_pad) SDM_FAIL(e, SDM_ERR_INVALID, "gemm %s: needs a P3 operand of %d channels", L.name.c_str(), L.Cin_pad);
  if (!L.w3) SDM_FAIL(e, SDM_ERR_STATE, "gemm %s: no W3 weight copy", L.name.c_str());
  if (res && res->f32 != 1) SDM_FAIL(e, SDM_ERR_INVALID, "gemm %s: fp32 residual expected", L.name.c_str());
  const int nout = L.geglu ? L.Cout_pad / 2 : L.Cout_pad;
  if (out->C % 32 || std::min(nout, out->C) % 32) SDM_FAIL(e, SDM_ERR_INVALID, "gemm %s: output channels %d", L.name.c_str(), out->C);
  int epi;
  if (L.geglu) { if (out->f32 != kFmtP3) SDM_FAIL(e, SDM_ERR_INVALID, "gemm %s: GEGLU writes P3", L.name.c_str()); epi = 1; }
  else if (out->f32 == 1) epi = out->want_stats ? 4 : 0;
  else if (out->f32 == 3) epi = 2;
  else if (out->f32 == 0) { epi = 2; lo_cols = 0; }          // plain fp16 rows = the high plane alone (q | k | v of the d = 512 attention)
  else if (out->f32 == kFmtP3) epi = 3;
  else SDM_FAIL(e, SDM_ERR_INVALID, "gemm %s: unsupported output format %d", L.name.c_str(), out->f32);
  GemmP3Params p;
  memset(&p, 0, sizeof(p));
  p.M = in.rows(); p.K = L.Cin_pad;
  p.a_hi = (const half_t*)in.p; p.a_xl = (const unsigned char*)in.p + p3_rows_pad((size_t)p.M) * p.K * 2;
  p.w = L.w3; p.N = L.Cout_pad; p.bias = L.b;
  p.out = out->p; p.ldo = out->C; p.n_valid = std::min(nout, out->C);
  p.out_lo_off = (epi == 2) ? (size_t)out->rows() * out->C : p3_rows_pad((size_t)out->rows()) * out->C * 2;
  p.lo_cols = lo_cols >= 0 ? lo_cols : (1 << 30);
  if (res) { p.res = (const float*)res->p; p.ldr = res->C; }
  p.sa = 127 - 11; p.sb = 127 - L.f8_exp;
  if (epi == 4) {
    if ((in.H * in.W) % 32) SDM_FAIL(e, SDM_ERR_INVALID, "gemm %s: fused statistics need images of a multiple of 32 rows (blocked operand planes)", L.name.c_str());
    p.rows_per_img = in.H * in.W;
    const int bm = gemm_p3_pick_bm(p.M, p.N, p.rows_per_img);
    out->srows = sdm_cdiv(p.rows_per_img, bm) * 2;
    T sb = talloc(e, 1, 1, 1, (int)((size_t)in.N * out->srows * out->C * 2), 1);
    out->soff = sb.off; out->sbytes = sb.bytes; out->stats = (float*)sb.p;
    p.stats = out->stats;
  }
  if (e->dry) return 0;
  const double flops = 2.0 * (double)p.M * L.O * L.I;
  const double bytes = (double)p.M * p.K * 3 + (double)p.M * p.n_valid * (double)fmt_bytes(out->f32) + (double)p.K * p.N * 4 + (res ? (double)p.M * p.n_valid * 4 : 0.0);
  if (e->prof_on) {
    char d[256];
    snprintf(d, sizeof(d), "%s N=%d Hout=%d Wout=%d Cin=%d Cout=%d p3 epi=%d bm=%d", L.name.c_str(), in.N, in.H, in.W, L.Cin_pad, L.O, epi, gemm_p3_pick_bm(p.M, p.N, p.rows_per_img));
    prof_begin(e, "gemm_mfma", flops, bytes, d);
  } else {
    prof_begin(e, "conv", flops, bytes);
  }
  count_kernel("gemm_p3");
  launch_gemm_p3(p, epi, e->stream);
#ifndef SDM_EMU
  { const hipError_t le = hipGetLastError(); if (le != hipSuccess) SDM_FAIL(e, SDM_ERR_HIP, "gemm %s: launch failed: %s", L.name.c_str(), hipGetErrorString(le)); }
#endif
  prof_end(e);
  return 0;
}

// fp32 [rows][C] -> P3 (operands whose producer does not emit planes itself)
static int op_to_p3(sdm_ctx* e, const T& x, T* out) {
  if (x.f32 != 1 || x.C % 32) SDM_FAIL(e, SDM_ERR_INVALID, "to_p3: fp32 input with C %% 32 == 0 expected (C = %d)", x.C);
  *out = talloc(e, x.N, x.H, x.W, x.C, kFmtP3);
  if (e->dry) return 0;
  const long units = ((x.rows() + 31) / 32) * (x.C / 32);      // one wave per 32 rows x 32 channels
  prof_begin(e, "to_p3", 0, (double)x.rows() * x.C * 7);
  SDM_LAUNCH(to_p3_kernel, dim3((unsigned)std::min<long>((units + 3) / 4, 1 << 20)), dim3(256), 0, e->stream, (const float*)x.p, (unsigned char*)out->p, x.rows(), x.C);
  prof_end(e);
  return 0;
}

static int linear(sdm_ctx* e, int layer, const T& in, T* out, int Cout, int out_f32, const T* res = nullptr, bool want_stats = false, int lo_cols = -1) {
  *out = talloc(e, in.N, in.H, in.W, Cout, out_f32);
  if (want_stats) TRY(tstats(e, *out));
  if (in.f32 == kFmtP3) return op_gemm_p3(e, e->convs[layer], in, out, res, lo_cols);
  ConvArgs a; a.in0 = &in; a.out = out; a.res = res; a.lo_cols = lo_cols;
  return op_conv(e, e->convs[layer], a);
}

// VAE mid-block Attention (Appendix A.5): GN -> q|k|v (+bias) -> softmax(qk^T/sqrt(C)) v -> to_out -> + x
static int vae_attention(sdm_ctx* e, const VaeAttnB& a, const T& x, T* out) {
  T hn, qkv, ao;
  // the single-head d=512 core always takes fp16 operands; a 64-channel VAE (test architectures) runs on the d=64 kernels and
  // follows the U-Net attention-core precision bit
  const int pa = ((e->cfg.precise_mask & SDM_PRECISE_UNET_ATTN) && a.C == 64) ? 1 : 0;
  // plane format of q | k | v: fp16 hi | lo (the logit scale is applied to Q inside the kernel here - it is not folded into these
  // weights - which the fp8 pair planes do not allow); V needs no low-part plane unless the fully split P.V form is requested
  const int pf = pa ? 2 : 0;
  const bool need_vlo = pf == 2 && opt("attn_pv_split") != 0;
  const int L = x.H * x.W;
  // plane-fed GEMMs (k_gemm.h) around the fp16-operand core: GroupNorm writes the q | k | v projection's operand planes, the projection plain fp16 rows,
  // the core fp32 rows that one pass converts to the planes of to_out - which adds the residual and emits the statistics of the next ResBlock's GroupNorm
  const bool p3 = pf == 0 && e->cfg.stream_f32 == 1 && L % 32 == 0 && a.C % 32 == 0 && p3_ok(e, e->convs[a.qkv]) && p3_ok(e, e->convs[a.out]);
  TRY(op_gn(e, e->norms[a.gn], x, nullptr, 0, e->cfg.vae_eps, &hn, p3 ? kFmtP3 : -1));
  TRY(linear(e, a.qkv, hn, &qkv, 3 * a.C, pf, nullptr, false, need_vlo ? -1 : 2 * a.C));
  tfree(e, hn);
  ao = talloc(e, x.N, x.H, x.W, a.C, e->act_f32);
  const half_t* q = (const half_t*)qkv.p;
  AttnPrec ap; ap.out_f32 = ao.f32; ap.prec = pa ? pf - 1 : 0;
  ap.q_lo = ap.k_lo = ap.v_lo = (long)qkv.rows() * qkv.C;
  TRY(op_attention_raw(e, q, 3 * a.C, q ? q + a.C : nullptr, 3 * a.C, q ? q + 2 * a.C : nullptr, 3 * a.C, nullptr, x.N, 1, L, L, a.C,
                       ao.p, a.C, false, nullptr, ap));
  tfree(e, qkv);
  if (p3) { T ap3; TRY(op_to_p3(e, ao, &ap3)); tfree(e, ao); ao = ap3; }      // (the d = 512 core keeps its fp32 epilogue: k_attn.h AttnParams::o_p3)
  TRY(linear(e, a.out, ao, out, a.C, e->cfg.stream_f32, &x, true));
  tfree(e, ao);
  return 0;
}

// Transformer2DModel + BasicTransformerBlock (Appendix A.7); bias = level key-bias [N][L] (log2 domain) or null
static int transformer(sdm_ctx* e, const TfB& t, const T& x, const T& uin, const float* bias, const int* tiles, T* out) {
  const int sf = e->cfg.stream_f32;
  const int C = t.C, L = x.H * x.W, L0 = uin.H * uin.W;
  T hn, h, n, qkv, ao, h2, q2, kv, f;
  const int pa = (e->cfg.precise_mask & SDM_PRECISE_UNET_ATTN) ? 1 : 0;      // split-precision attention cores: q|k|v as hi|lo planes
  const int pf = pa ? (attn_f8_enabled() ? 3 : 2) : 0;         // plane format of q | k | v; SDM_ATTN_PV_SPLIT=1 (fully split P.V, test hook) needs V_lo too
  const bool need_vlo = pf == 2 && opt("attn_pv_split") != 0;
  // plane-fed GEMMs (k_gemm.h): every Linear of the block takes a P3 operand written by its producer - GroupNorm apply, LayerNorm, the GEGLU and
  // ff.net.2 epilogues - or, behind the attention cores (fp32 output), by one conversion pass
  bool p3 = sf == 1 && pf == 3 && C % 32 == 0;
  for (int l : {t.proj_in, t.qkv1, t.o1, t.q2, t.o2, t.ff1, t.ff2, t.proj_out}) p3 = p3 && p3_ok(e, e->convs[l]);
  const int nf = p3 ? kFmtP3 : -1;                             // operand format of the norms' outputs (-1: the engine's activation type)
  const bool p3a = p3 && L % 32 == 0 && opt("gemm_p3_attn") != 0;                           // the attention cores write the planes themselves (O^T accumulators = the GEMM's operand layout)
  auto attn_out = [&](T& a) -> int {                           // the attention output as the next GEMM's operand
    if (!p3 || p3a) return 0;
    T ap;
    TRY(op_to_p3(e, a, &ap));
    tfree(e, a);
    a = ap;
    return 0;
  };
  TRY(op_gn(e, e->norms[t.gn], x, nullptr, 0, e->cfg.unet_tf_gn_eps, &hn, nf));
  TRY(linear(e, t.proj_in, hn, &h, C, sf));
  tfree(e, hn);
  // self-attention with the trimap key bias
  TRY(op_ln(e, e->norms[t.ln1], h, e->cfg.unet_ln_eps, &n, nf));
  TRY(linear(e, t.qkv1, n, &qkv, 3 * C, pf, nullptr, false, need_vlo ? -1 : 2 * C));
  tfree(e, n);
  ao = talloc(e, x.N, x.H, x.W, C, p3a ? kFmtP3 : e->act_f32);
  {
    const half_t* q = (const half_t*)qkv.p;
    AttnPrec ap; ap.prec = pa ? pf - 1 : 0; ap.out_f32 = p3a ? 1 : ao.f32; ap.out_p3 = p3a ? 1 : 0;
    ap.q_lo = ap.k_lo = ap.v_lo = (long)qkv.rows() * qkv.C;
    TRY(op_attention_raw(e, q, 3 * C, q ? q + C : nullptr, 3 * C, q ? q + 2 * C : nullptr, 3 * C, bias, x.N, t.heads, L, L, 64, ao.p, C, true, tiles, ap));
  }
  tfree(e, qkv);
  TRY(attn_out(ao));
  TRY(linear(e, t.o1, ao, &h2, C, sf, &h));
  tfree(e, ao); tfree(e, h);
  // cross-attention to the trimap-latent tokens
  TRY(op_ln(e, e->norms[t.ln2], h2, e->cfg.unet_ln_eps, &n, nf));
  TRY(linear(e, t.q2, n, &q2, C, pf));
  tfree(e, n);
  TRY(conv_simple(e, t.kv2, uin, &kv, 2 * C, pf, 1, 0, 0, nullptr, 1.0f, false, need_vlo ? -1 : C));      // folded aux_conv_in + to_k|to_v: tokens = latent pixels, row-major
  ao = talloc(e, x.N, x.H, x.W, C, p3a ? kFmtP3 : e->act_f32);
  {
    const half_t* kk = (const half_t*)kv.p;
    AttnPrec ap; ap.prec = pa ? pf - 1 : 0; ap.out_f32 = p3a ? 1 : ao.f32; ap.out_p3 = p3a ? 1 : 0;
    ap.q_lo = (long)q2.rows() * q2.C; ap.k_lo = ap.v_lo = (long)kv.rows() * kv.C;
    TRY(op_attention_raw(e, (const half_t*)q2.p, C, kk, 2 * C, kk ? kk + C : nullptr, 2 * C, nullptr, x.N, t.heads, L, L0, 64, ao.p, C, true, nullptr, ap));
  }
  tfree(e, q2); tfree(e, kv);
  TRY(attn_out(ao));
  TRY(linear(e, t.o2, ao, &h, C, sf, &h2));
  tfree(e, ao); tfree(e, h2);
  // GEGLU feed-forward
  TRY(op_ln(e, e->norms[t.ln3], h, e->cfg.unet_ln_eps, &n, nf));
  TRY(linear(e, t.ff1, n, &f, 4 * C, p3 ? kFmtP3 : e->act_f32));
  tfree(e, n);
  // (P3: h2 only feeds proj_out - whose fused statistics need image-aligned row tiles on whole 32-row blocks; otherwise proj_out keeps the fp32 kernel)
  TRY(linear(e, t.ff2, f, &h2, C, (p3 && L % 32 == 0) ? kFmtP3 : sf, &h));
  tfree(e, f); tfree(e, h);
  TRY(linear(e, t.proj_out, h2, out, C, sf, &x, true));
  tfree(e, h2);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// embedding constants (host): emb = time_embedding(time_proj(trans)) + bbox_embedding(sincos(coords))
//   replace.py:419-459; each ResBlock adds time_emb_proj(silu(emb)) after conv1 (Appendix A.3) -> folded
//   into that conv's bias table, one row per (trans, coords) variant.
// ------------------------------------------------------------------------------------------------
static void sincos_embed(float t, int dim, float* out) {  // get_timestep_embedding(flip_sin_to_cos=True, shift=0)
  const int half = dim / 2;
  for (int i = 0; i < half; ++i) {
    const float f = expf(-logf(10000.0f) * (float)i / (float)half);
    const float a = t * f;
    out[i] = cosf(a);
    out[half + i] = sinf(a);
  }
  if (dim & 1) out[dim - 1] = 0.0f;                       // odd dim: diffusers pads one zero column
}
// meta_arch.py:153-162: N point coordinates are zero-padded to the first i >= N that divides the point-embedding input
// dim P (1680 in the reference), each padded value is embedded with P / i channels.  Returns 0 if no such i < P exists
// (the reference's loop would fall through with unbound variables).
static int point_pad_len(int N, int P) {
  for (int i = N; i < P; ++i)
    if (i > 0 && P % i == 0) return i;
  return 0;
}
static void host_linear(const float* W, const float* b, const float* x, int O, int I, float* y) {
  for (int o = 0; o < O; ++o) {
    double acc = b ? (double)b[o] : 0.0;
    const float* w = W + (size_t)o * I;
    for (int i = 0; i < I; ++i) acc += (double)w[i] * (double)x[i];
    y[o] = (float)acc;
  }
}
static inline float host_silu(float x) { return x / (1.0f + expf(-x)); }

static int compute_variant_tables(sdm_ctx* e, int vidx) {
  const Variant& v = e->variants[vidx];
  const sdm_config& c = e->cfg;
  const int c0 = c.unet_channels[0], te = c0 * 4, bd = c.bbox_embeddings_input_dim;
  const float* H = e->hostblob.data();
  std::vector<float> tp(c0), t1(te), op(te), ce(bd), b1(te), aug(te), se(te);
  sincos_embed((float)v.trans, c0, tp.data());
  host_linear(H + e->h_time1w, H + e->h_time1b, tp.data(), te, c0, t1.data());
  for (auto& x : t1) x = host_silu(x);
  host_linear(H + e->h_time2w, H + e->h_time2b, t1.data(), te, te, op.data());
  if (v.kind == 0) {          // box prompt: 4 coordinates x (bd/4) channels -> bbox_embedding (meta_arch.py:178-187, replace.py:451-455)
    for (int k = 0; k < 4; ++k) sincos_embed(v.c[k], bd / 4, ce.data() + k * (bd / 4));
    host_linear(H + e->h_bbox1w, H + e->h_bbox1b, ce.data(), te, bd, b1.data());
    for (auto& x : b1) x = host_silu(x);
    host_linear(H + e->h_bbox2w, H + e->h_bbox2b, b1.data(), te, te, aug.data());
  } else {                    // point prompt: padded coordinates x (P/i) channels -> point_embedding (meta_arch.py:153-176, replace.py:446-450)
    const int P = c.point_embeddings_input_dim, npad = point_pad_len((int)v.c.size(), P), ch = P / npad;
    std::vector<float> pe((size_t)P, 0.0f);
    for (int k = 0; k < npad; ++k) sincos_embed(k < (int)v.c.size() ? v.c[k] : 0.0f, ch, pe.data() + (size_t)k * ch);
    host_linear(H + e->h_point1w, H + e->h_point1b, pe.data(), te, P, b1.data());
    for (auto& x : b1) x = host_silu(x);
    host_linear(H + e->h_point2w, H + e->h_point2b, b1.data(), te, te, aug.data());
  }
  for (int i = 0; i < te; ++i) se[i] = host_silu(op[i] + aug[i]);
  std::vector<float> row;
  for (auto& t : e->tembs) {
    row.assign(t.cout_pad, 0.0f);
    host_linear(H + t.w_hoff, H + t.b_hoff, se.data(), t.cout, te, row.data());
    for (int o = 0; o < t.cout; ++o) row[o] += H[t.cb_hoff + o];
    SDM_CHECK_DEV(e, dev_memcpy_h2d(t.table + (size_t)vidx * t.cout_pad, row.data(), (size_t)t.cout_pad * 4, e->stream));
    SDM_CHECK_DEV(e, dev_sync(e->stream));   // `row` is reused
  }
  return 0;
}

static int prepare_variants(sdm_ctx* e, int B, const int32_t* is_trans, const float* cond, int cond_dim, int cond_kind) {
  std::vector<int> sel(B);
  std::vector<Variant> want(B);
  if (cond_kind == 1) {
    if (!cond || cond_dim <= 0 || point_pad_len(cond_dim, e->cfg.point_embeddings_input_dim) == 0)
      SDM_FAIL(e, SDM_ERR_INVALID, "point prompt: %d coordinates cannot be padded to a divisor of %d", cond_dim, e->cfg.point_embeddings_input_dim);
  } else if (cond && cond_dim != 4) {
    SDM_FAIL(e, SDM_ERR_INVALID, "box prompt: expected 4 coordinates per image, got %d", cond_dim);
  }
  for (int b = 0; b < B; ++b) {
    want[b].trans = 1 - (is_trans ? (int)is_trans[b] : 0);        // meta_arch.py:237-238
    want[b].kind = cond_kind;
    if (cond_kind == 1) {
      want[b].c.assign(cond + (size_t)b * cond_dim, cond + (size_t)(b + 1) * cond_dim);
    } else {
      const float def[4] = {0.f, 0.f, 1.f, 1.f};                  // sdmatte_nodes.py:353 / meta_arch.py:189-190
      want[b].c.assign(4, 0.0f);
      for (int k = 0; k < 4; ++k) want[b].c[k] = cond ? cond[b * 4 + k] : def[k];
    }
  }
  auto find = [&](const Variant& v) {
    for (size_t i = 0; i < e->variants.size(); ++i)
      if (e->variants[i] == v) return (int)i;
    return -1;
  };
  {   // distinct conditionings of THIS batch; the tables hold at least that many rows (plus what is cached from earlier calls)
    std::vector<Variant> uniq;
    for (int b = 0; b < B; ++b) {
      bool seen = false;
      for (auto& u : uniq) if (u == want[b]) { seen = true; break; }
      if (!seen) uniq.push_back(want[b]);
    }
    int missing = 0;
    for (auto& u : uniq) if (find(u) < 0) ++missing;
    if ((int)e->variants.size() + missing > e->variant_cap) {
      // not enough rows: drop the cached rows (they are recomputed on demand) and, if this batch alone needs more, grow the tables
      e->variants.clear();
      if ((int)uniq.size() > e->variant_cap) {
        const int cap = std::max(rup((int)uniq.size(), 8), kMinVariantRows);
        SDM_CHECK_DEV(e, dev_sync(e->stream));
        for (auto& t : e->tembs) {
          if (t.table) dev_free(t.table);
          void* q = nullptr;
          if (dev_malloc(&q, (size_t)cap * t.cout_pad * 4) != 0) { t.table = nullptr; e->variant_cap = 0; SDM_FAIL(e, SDM_ERR_NOMEM, "cannot allocate %d conditioning rows", cap); }
          t.table = (float*)q;
        }
        e->variant_cap = cap;
      }
    }
    for (int b = 0; b < B; ++b) {
      int f = find(want[b]);
      if (f < 0) {
        e->variants.push_back(want[b]);
        f = (int)e->variants.size() - 1;
        TRY(compute_variant_tables(e, f));
      }
      sel[b] = f;
    }
  }
  if (B > e->bias_sel_cap) {
    if (e->d_bias_sel) dev_free(e->d_bias_sel);
    SDM_CHECK_DEV(e, dev_malloc((void**)&e->d_bias_sel, (size_t)B * 4));
    e->bias_sel_cap = B;
  }
  SDM_CHECK_DEV(e, dev_memcpy_h2d(e->d_bias_sel, sel.data(), (size_t)B * 4, e->stream));
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// the model: x16 [2B,SH,SW,16] fp16 (rgb images then trimaps), plane [B,SH,SW] fp32 (trimap in [-1,1]) -> alpha [B,SH,SW]
// ------------------------------------------------------------------------------------------------
// plane (optional): the trimap plane [B][H][W] behind images B .. 2B-1 of x16 - those images are piecewise constant, and the encoder's wide convs do
// not multiply the output tiles that lie inside one region (k_misc.h cmask_*, op_conv; engine option trimap_skip; split-precision mode only: the F8
// kernel is the one that knows how to leave tiles out)
static int vae_encode(sdm_ctx* e, const T& x16, T* moments, const T* plane = nullptr) {
  const float eps = e->cfg.vae_eps;
  T h, t;
  T xin = x16;
  const bool cm = plane && opt("trimap_skip") != 0 && e->act_f32 && x16.N == 2 * plane->N && plane->H == x16.H && plane->W == x16.W;
  if (cm) {
    const long hw = (long)x16.H * x16.W;
    T mb = talloc(e, 1, 1, 1, (int)(((size_t)x16.N * hw + 3) / 4), 1);
    xin.cmask = (unsigned char*)mb.p; xin.cm_off = mb.off; xin.cm_bytes = mb.bytes; xin.cm_n0 = plane->N;
    T tab = talloc(e, 1, 1, 1, x16.N * 4, 1);
    if (!e->dry) {
      SDM_LAUNCH(cmask_table_kernel, dim3((unsigned)plane->N), dim3(256), 0, e->stream, (const float*)plane->p, (unsigned int*)tab.p, plane->N, x16.H, x16.W);
      SDM_LAUNCH(cmask_init_kernel, dim3((unsigned)((plane->N * hw + 255) / 256)), dim3(256), 0, e->stream, (const float*)plane->p, xin.cmask, (const unsigned int*)tab.p,
                 plane->N, plane->N, hw);
    }
    tfree(e, tab);
  }
  TRY(conv_simple(e, e->enc_conv_in, xin, &h, e->cfg.vae_channels[0], e->cfg.stream_f32, 1, 0, 0, nullptr, 1.0f, true));
  if (cm) tfree_raw(e, xin.cm_off, xin.cm_bytes);
  for (int i = 0; i < 4; ++i) {
    for (auto& r : e->enc_res[i]) { TRY(resblock(e, r, h, nullptr, eps, &t)); tfree(e, h); h = t; }
    if (i < 3) { TRY(conv_simple(e, e->enc_down[i], h, &t, e->cfg.vae_channels[i], e->cfg.stream_f32, 2, 1, 0, nullptr, 1.0f, true)); tfree(e, h); h = t; }
  }
  TRY(resblock(e, e->enc_mid0, h, nullptr, eps, &t)); tfree(e, h); h = t;
  TRY(vae_attention(e, e->enc_attn, h, &t)); tfree(e, h); h = t;
  TRY(resblock(e, e->enc_mid1, h, nullptr, eps, &t)); tfree(e, h); h = t;
  *moments = talloc(e, h.N, h.H, h.W, 16, e->act_f32);
  { ConvArgs a; a.out = moments; TRY(gn_conv(e, e->norms[e->enc_norm_out], e->convs[e->enc_conv_out], h, nullptr, 1, eps, a)); }
  tfree(e, h);
  return 0;
}

static int vae_decode(sdm_ctx* e, const T& z, T* dec) {
  const float eps = e->cfg.vae_eps;
  T h, t;
  TRY(conv_simple(e, e->dec_conv_in, z, &h, e->cfg.vae_channels[3], e->cfg.stream_f32, 1, 0, 0, nullptr, 1.0f, true));
  TRY(resblock(e, e->dec_mid0, h, nullptr, eps, &t)); tfree(e, h); h = t;
  TRY(vae_attention(e, e->dec_attn, h, &t)); tfree(e, h); h = t;
  TRY(resblock(e, e->dec_mid1, h, nullptr, eps, &t)); tfree(e, h); h = t;
  for (int i = 0; i < 4; ++i) {
    for (auto& r : e->dec_res[i]) { TRY(resblock(e, r, h, nullptr, eps, &t)); tfree(e, h); h = t; }
    if (i < 3) { TRY(conv_simple(e, e->dec_up[i], h, &t, e->cfg.vae_channels[3 - i], e->cfg.stream_f32, 1, 0, 1, nullptr, 1.0f, true)); tfree(e, h); h = t; }
  }
  *dec = talloc(e, h.N, h.H, h.W, 4, 1);
  { ConvArgs a; a.out = dec; TRY(gn_conv(e, e->norms[e->dec_norm_out], e->convs[e->dec_conv_out], h, nullptr, 1, eps, a)); }
  tfree(e, h);
  return 0;
}

static int unet_forward(sdm_ctx* e, const T& uin, float* const* bias_lvl, int* const* tiles_lvl, T* out) {
  const sdm_config& c = e->cfg;
  const float eps = c.unet_res_eps;
  const int sf = c.stream_f32;
  std::vector<T> skips;
  T h, t;
  TRY(conv_simple(e, e->u_conv_in, uin, &h, c.unet_channels[0], sf, 1, 0, 0, nullptr, 1.0f, true));
  skips.push_back(h);
  for (int i = 0; i < 4; ++i) {
    for (size_t j = 0; j < e->u_down_res[i].size(); ++j) {
      TRY(resblock(e, e->u_down_res[i][j], h, nullptr, eps, &t));
      if (i < 3) {
        T t2;
        TRY(transformer(e, e->u_down_tf[i][j], t, uin, bias_lvl[i], tiles_lvl[i], &t2));
        tfree(e, t); t = t2;
      }
      h = t;                       // previous h stays alive as a skip
      skips.push_back(h);
    }
    if (i < 3) {
      TRY(conv_simple(e, e->u_down_ds[i], h, &t, c.unet_channels[i], sf, 2, 0, 0, nullptr, 1.0f, true));
      h = t;
      skips.push_back(h);
    }
  }
  // mid (h aliases the last skip: do not free it here)
  TRY(resblock(e, e->u_mid0, h, nullptr, eps, &t)); h = t;
  TRY(transformer(e, e->u_midtf, h, uin, bias_lvl[3], tiles_lvl[3], &t)); tfree(e, h); h = t;
  TRY(resblock(e, e->u_mid1, h, nullptr, eps, &t)); tfree(e, h); h = t;
  for (int i = 0; i < 4; ++i) {
    for (size_t j = 0; j < e->u_up_res[i].size(); ++j) {
      T s = skips.back(); skips.pop_back();
      TRY(resblock(e, e->u_up_res[i][j], h, &s, eps, &t));   // cat([h, skip], dim=1) then ResBlock (replace.py:509-536)
      tfree(e, h); tfree(e, s); h = t;
      if (i > 0) {
        TRY(transformer(e, e->u_up_tf[i][j], h, uin, bias_lvl[3 - i], tiles_lvl[3 - i], &t));
        tfree(e, h); h = t;
      }
    }
    if (i < 3) { TRY(conv_simple(e, e->u_up_us[i], h, &t, c.unet_channels[3 - i], sf, 1, 0, 1, nullptr, 1.0f, true)); tfree(e, h); h = t; }
  }
  // label_latent / scaling_factor (meta_arch.py:254) folded into the conv_out epilogue
  *out = talloc(e, h.N, h.H, h.W, 16, e->act_f32);
  { ConvArgs a; a.out = out; a.out_scale = 1.0f / c.vae_scaling_factor; TRY(gn_conv(e, e->norms[e->u_norm_out], e->convs[e->u_conv_out], h, nullptr, 1, eps, a)); }
  tfree(e, h);
  return 0;
}

static int run_model(sdm_ctx* e, const T& x16, const T& plane, int B, int SH, int SW, bool use_mask, T* alpha) {
  const sdm_config& c = e->cfg;
  const int lh = SH / 8, lw = SW / 8;
  // attention key bias at the 4 U-Net levels (meta_arch.py:200-204, replace.py:401-403,56-63)
  T biasbuf[4], tilebuf[4];
  float* bias_lvl[4];
  int* tiles_lvl[4];            // per level: the key tiles that can contribute to the softmax (AttnParams::tiles), built once per forward
  for (int k = 0; k < 4; ++k) {
    const int lk2 = (lh >> k) * (lw >> k), nt = sdm_cdiv(lk2, 64);
    biasbuf[k] = talloc(e, B, 1, 1, lk2, 1);
    tilebuf[k] = talloc(e, B, 1, 1, nt + 1, 1);
    bias_lvl[k] = (float*)biasbuf[k].p;
    tiles_lvl[k] = (int*)tilebuf[k].p;
    if (!e->dry && use_mask) {
      SDM_LAUNCH(mask_bias_kernel, dim3(sdm_cdiv(B * lk2, 256)), dim3(256), 0, e->stream, (const float*)plane.p, bias_lvl[k], B, SH, SW, k,
                 c.attn_mask_value, SDM_LOG2E);
      SDM_LAUNCH(attn_active_tiles_kernel, dim3(B), dim3(256), 0, e->stream, (const float*)bias_lvl[k], lk2, nt, tiles_lvl[k], nt + 1,
                 SDM_ATTN_SKIP_MARGIN);
    }
    // prompt types outside attn_mask_aux_input run the self-attention without a key mask (meta_arch.py:199-206)
    if (!use_mask) { bias_lvl[k] = nullptr; tiles_lvl[k] = nullptr; }
  }
  // VAE encode of rgb and trimap as one batch (meta_arch.py:139-145, 209-212)
  T moments;
  TRY(vae_encode(e, x16, &moments, &plane));
  // quant_conv -> mean half * scaling_factor, written straight into the 8(+8 pad)-channel U-Net input:
  // channels 0..3 = rgb latent, 4..7 = trimap latent (torch.cat order of meta_arch.py:244)
  T uin = talloc(e, B, lh, lw, 16, e->act_f32);
  if (!e->dry) SDM_CHECK_DEV(e, dev_memset(uin.p, 0, uin.bytes, e->stream));
  for (int half = 0; half < 2; ++half) {
    T mv = moments; mv.N = B;
    if (!e->dry) mv.p = (unsigned char*)moments.p + (size_t)half * B * lh * lw * 16 * fmt_bytes(moments.f32);
    ConvArgs a; a.in0 = &mv; a.out = &uin; a.out_ch_off = half * 4; a.cout_valid = 4; a.out_scale = c.vae_scaling_factor;
    TRY(op_conv(e, e->convs[e->quant], a));
  }
  tfree(e, moments);
  // cross-attention context (meta_arch.py:215-218: aux_conv_in(trimap latent) as [B, l*l, ctx]) is never materialised:
  // every block's K|V comes straight from the latent through the folded 3x3 conv (transformer())
  T lat;
  TRY(unet_forward(e, uin, bias_lvl, tiles_lvl, &lat));
  tfree(e, uin);
  for (int k = 3; k >= 0; --k) { tfree(e, tilebuf[k]); tfree(e, biasbuf[k]); }
  // post_quant_conv + decoder (meta_arch.py:255-256)
  T z;
  TRY(conv_simple(e, e->post_quant, lat, &z, 16, e->act_f32)); tfree(e, lat);
  T dec;
  TRY(vae_decode(e, z, &dec)); tfree(e, z);
  *alpha = talloc(e, B, SH, SW, 1, 1);
  if (!e->dry)
    SDM_LAUNCH(alpha_out_kernel, dim3((unsigned)(((long)B * SH * SW + 255) / 256)), dim3(256), 0, e->stream, (const float*)dec.p, (float*)alpha->p,
               (long)B * SH * SW);
  tfree(e, dec);
  return 0;
}

// ------------------------------------------------------------------------------------------------
// top-level forward helpers
// ------------------------------------------------------------------------------------------------
static int ensure_buf(sdm_ctx* e, void** p, size_t* cap, size_t need) {
  if (*cap >= need) return 0;
  if (*p) { SDM_CHECK_DEV(e, dev_sync(e->stream)); dev_free(*p); *p = nullptr; *cap = 0; }
  SDM_CHECK_DEV(e, dev_malloc(p, need));
  *cap = need;
  return 0;
}

// mode 0: core API (NCHW preprocessed, S x S); mode 1: node API (BHWC image + BHW trimap at H x W)
// mode 0 takes the inference size as (SH, SW) = (H, W) and S is ignored; mode 1 resizes H x W to S x S like the node.
// node tail (mode 1 only): mask_refine + output composition on the GPU, sdmatte_nodes.py:365-397
// trimap_constraint stays a double up to the two thresholds: the reference compares fp32 tensors with the Python floats c and
// 1.0 - c (evaluated in double), i.e. with float32(c) and float32(1.0 - c) - for c = 0.8 the latter is 0.2f, not 1.0f - 0.8f
struct NodeTail { int output_mode = 0, mask_refine = 0; double c = 0.8; float* matted = nullptr; int TH = 0, TW = 0; int channels() const { return output_mode == 1 ? 4 : 3; } };

static int forward_impl(sdm_ctx* e, int mode, const float* image, const float* trimap, int B, int H, int W, int S, const int32_t* is_trans,
                        const float* cond, int cond_dim, int cond_kind, bool use_mask, float* out, int ptr_kind, void* stream_arg,
                        const NodeTail* tail = nullptr) {
  if (!e->finalized) SDM_FAIL(e, SDM_ERR_STATE, "weights not finalised: call sdm_load_tensor(...) and sdm_finalize_weights first");
  OptReadLock opt_lock;          // kernel-selection options stay put for both passes of this forward
  const int SH = (mode == 0) ? H : S, SW = (mode == 0) ? W : S;
  if (B <= 0 || SH <= 0 || SW <= 0 || SH % 64 || SW % 64)
    SDM_FAIL(e, SDM_ERR_INVALID, "inference size must be a positive multiple of 64 (got %dx%d)", SH, SW);
  if (mode == 1 && (H <= 0 || W <= 0)) SDM_FAIL(e, SDM_ERR_INVALID, "bad image size %dx%d", H, W);
  // Stream contract (include/sdmatte.h): kernels run on the engine's own stream.  For DEVICE pointers the caller names the
  // stream on which it produced the inputs and will consume the outputs (NULL = the device's default stream): the engine
  // stream waits for everything queued there at call time, and that stream waits for the outputs before the call returns
  // control (no host synchronisation).  HOST pointers are copied on the engine stream, followed by a host sync below.
#ifndef SDM_EMU
  if (ptr_kind == SDM_PTR_DEVICE) {
    SDM_CHECK_DEV(e, (int)hipEventRecord(e->ev_in, (hipStream_t)stream_arg));
    SDM_CHECK_DEV(e, (int)hipStreamWaitEvent((hipStream_t)e->stream, e->ev_in, 0));
  }
#else
  (void)stream_arg;
#endif
  // node API: the trimap may have its own size (the reference resizes image and trimap independently, sdmatte_nodes.py:212-214,349)
  const int TH = (tail && tail->TH > 0) ? tail->TH : H, TW = (tail && tail->TW > 0) ? tail->TW : W;
  const size_t in_img = (size_t)B * H * W * 3 * 4;          // mode 0: [B,3,SH,SW]; mode 1: [B,H,W,3]
  const size_t in_tri = (size_t)B * TH * TW * 4;
  const size_t alpha_bytes = (size_t)B * H * W * 4;
  const size_t out_bytes = alpha_bytes * (tail ? 1 + tail->channels() : 1);      // host hand-over: alpha, then the composed image
  const float* d_img = image; const float* d_tri = trimap; float* d_out = out;
  float* d_matted = tail ? tail->matted : nullptr;
  if (ptr_kind == SDM_PTR_HOST) {
    TRY(ensure_buf(e, &e->io_in, &e->io_in_bytes, in_img + in_tri));
    TRY(ensure_buf(e, &e->io_out, &e->io_out_bytes, out_bytes));
    SDM_CHECK_DEV(e, dev_memcpy_h2d(e->io_in, image, in_img, e->stream));
    SDM_CHECK_DEV(e, dev_memcpy_h2d((unsigned char*)e->io_in + in_img, trimap, in_tri, e->stream));
    d_img = (const float*)e->io_in; d_tri = (const float*)((unsigned char*)e->io_in + in_img); d_out = (float*)e->io_out;
    if (tail) d_matted = (float*)((unsigned char*)e->io_out + alpha_bytes);
  }
  TRY(prepare_variants(e, B, is_trans, cond, cond_dim, cond_kind));
  for (int pass = 0; pass < 2; ++pass) {
    e->dry = (pass == 0);
    arena_reset(e);
    if (pass == 0) e->peak = 0;
    else if (e->peak > e->arena_bytes) {
      if (e->arena) { SDM_CHECK_DEV(e, dev_sync(e->stream)); dev_free(e->arena); e->arena = nullptr; e->arena_bytes = 0; }
      void* p = nullptr;
      if (dev_malloc(&p, e->peak) != 0) SDM_FAIL(e, SDM_ERR_NOMEM, "cannot allocate %zu bytes of activation arena", e->peak);
      e->arena = (unsigned char*)p; e->arena_bytes = e->peak;
    }
#ifndef SDM_EMU
    if (pass == 1) (void)hipEventRecord(e->ev0, (hipStream_t)e->stream);
#endif
    T x16 = talloc(e, 2 * B, SH, SW, 16, e->act_f32);
    T plane = talloc(e, B, SH, SW, 1, 1);
    if (!e->dry) {
      const unsigned nb = (unsigned)(((long)B * SH * SW + 255) / 256);
      void* img16 = x16.p;
      void* tri16 = (unsigned char*)x16.p + (size_t)B * SH * SW * 16 * fmt_bytes(x16.f32);
      if (mode == 0) {
        SDM_LAUNCH(prep_nchw_kernel, dim3(nb), dim3(256), 0, e->stream, d_img, d_tri, img16, tri16, x16.f32, (float*)plane.p, B, SH, SW);
      } else {
        SDM_LAUNCH(prep_image_kernel, dim3(nb), dim3(256), 0, e->stream, d_img, img16, x16.f32, B, H, W, S);
        SDM_LAUNCH(prep_trimap_kernel, dim3(nb), dim3(256), 0, e->stream, d_tri, tri16, x16.f32, (float*)plane.p, B, TH, TW, S);
      }
    }
    T alpha;
    int rc = run_model(e, x16, plane, B, SH, SW, use_mask, &alpha);
    if (rc) { e->dry = false; return rc; }
    if (!e->dry) {
      if (mode == 0) {
        SDM_CHECK_DEV(e, dev_memcpy_d2d(d_out, alpha.p, (size_t)B * SH * SW * 4, e->stream));
      } else {
        SDM_LAUNCH(resize_planes_kernel, dim3((unsigned)(((long)B * H * W + 255) / 256)), dim3(256), 0, e->stream, (const float*)alpha.p, d_out, B,
                   S, S, H, W, 1);
        if (tail)
          SDM_LAUNCH(refine_compose_kernel, dim3((unsigned)(((long)B * H * W + 255) / 256)), dim3(256), 0, e->stream, d_img, d_tri, d_out, d_matted,
                     (long)B * H * W, tail->output_mode, tail->mask_refine, (float)tail->c, (float)(1.0 - tail->c));
      }
    }
    tfree(e, alpha); tfree(e, plane); tfree(e, x16);
  }
  e->dry = false;
#ifndef SDM_EMU
  (void)hipEventRecord(e->ev1, (hipStream_t)e->stream);
#endif
  if (ptr_kind == SDM_PTR_HOST) {
    SDM_CHECK_DEV(e, dev_memcpy_d2h(out, e->io_out, alpha_bytes, e->stream));
    if (tail) SDM_CHECK_DEV(e, dev_memcpy_d2h(tail->matted, (unsigned char*)e->io_out + alpha_bytes, out_bytes - alpha_bytes, e->stream));
    SDM_CHECK_DEV(e, dev_sync(e->stream));
  }
#ifndef SDM_EMU
  else {
    SDM_CHECK_DEV(e, (int)hipEventRecord(e->ev_out, (hipStream_t)e->stream));
    SDM_CHECK_DEV(e, (int)hipStreamWaitEvent((hipStream_t)stream_arg, e->ev_out, 0));
  }
#endif
  return 0;
}

// runs an op outside forward(): arena sized by a dry pass of the same code
template <typename F>
static int run_two_pass(sdm_ctx* e, F body) {
  OptReadLock opt_lock;          // (never nested: forward_impl does not come through here)
  for (int pass = 0; pass < 2; ++pass) {
    e->dry = (pass == 0);
    arena_reset(e);
    if (pass == 0) e->peak = 0;
    else if (e->peak > e->arena_bytes) {
      if (e->arena) { dev_sync(e->stream); dev_free(e->arena); e->arena = nullptr; e->arena_bytes = 0; }
      void* p = nullptr;
      if (dev_malloc(&p, std::max(e->peak, (size_t)1 << 20)) != 0) { e->dry = false; SDM_FAIL(e, SDM_ERR_NOMEM, "arena alloc failed"); }
      e->arena = (unsigned char*)p; e->arena_bytes = std::max(e->peak, (size_t)1 << 20);
    }
    int rc = body();
    if (rc) { e->dry = false; return rc; }
  }
  e->dry = false;
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return 0;
}


// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static void load_ring_release(sdm_ctx* e);

extern "C" {

void sdm_default_config(sdm_config* c) {
  memset(c, 0, sizeof(*c));
  const int vc[4] = {128, 256, 512, 512}, uc[4] = {320, 640, 1280, 1280}, uh[4] = {5, 10, 20, 20};
  for (int i = 0; i < 4; ++i) { c->vae_channels[i] = vc[i]; c->unet_channels[i] = uc[i]; c->unet_heads[i] = uh[i]; }
  c->vae_layers_per_block = 2; c->unet_layers_per_block = 2;
  c->cross_attention_dim = 1024; c->unet_in_channels = 8; c->unet_out_channels = 4;
  c->bbox_embeddings_input_dim = 1280; c->point_embeddings_input_dim = 1680; c->groups = 32;
  c->vae_eps = 1e-6f; c->unet_res_eps = 1e-5f; c->unet_tf_gn_eps = 1e-6f; c->unet_ln_eps = 1e-5f;
  c->vae_scaling_factor = 0.18215f; c->attn_mask_value = -10000.0f;
  c->stream_f32 = 1;
  c->precise_mask = SDM_PRECISE_ALL;      // the precision that meets the 1e-3 parity bar (include/sdmatte.h); 0 selects the fast fp16-operand graph
}

int sdm_create(sdm_ctx** out, int device_id, const sdm_config* cfg) {
  if (!out) return SDM_ERR_INVALID;
  *out = nullptr;
#ifndef SDM_EMU
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    g_create_err = "no HIP device visible: the SDMatte engine is gfx950-only and has no CPU fallback";
    return SDM_ERR_NODEVICE;
  }
  if (device_id < 0 || device_id >= ndev) { g_create_err = "bad device id"; return SDM_ERR_INVALID; }
  if (hipSetDevice(device_id) != hipSuccess) { g_create_err = "hipSetDevice failed"; return SDM_ERR_HIP; }
#endif
  sdm_ctx* e = new sdm_ctx();
  if (cfg) e->cfg = *cfg; else sdm_default_config(&e->cfg);
  if (e->cfg.point_embeddings_input_dim <= 0) e->cfg.point_embeddings_input_dim = 1680;
  e->cfg.precise_mask &= SDM_PRECISE_ALL;
  if (opt("precise_mask") >= 0) e->cfg.precise_mask = opt("precise_mask") & SDM_PRECISE_ALL;      // experiment option (per-stage attribution)
  e->act_f32 = e->cfg.precise_mask ? 1 : 0;
  if (e->act_f32) e->cfg.stream_f32 = 1;
  e->device = device_id;
  const sdm_config& c = e->cfg;
  for (int i = 0; i < 4; ++i) {
    if (c.vae_channels[i] % 32 || c.unet_channels[i] % 64 || c.unet_heads[i] * 64 != c.unet_channels[i]) {
      g_create_err = "unsupported config: channels must be multiples of 32 (VAE) / 64 (U-Net) with head_dim 64";
      delete e; return SDM_ERR_INVALID;
    }
  }
  if (!(c.vae_channels[3] == 64 || c.vae_channels[3] == 512)) {
    g_create_err = "unsupported config: VAE mid channels must be 64 or 512 (single-head attention kernels)";
    delete e; return SDM_ERR_INVALID;
  }
  if (c.unet_in_channels > 16 || c.bbox_embeddings_input_dim % 8 || c.cross_attention_dim % 64) {
    g_create_err = "unsupported config"; delete e; return SDM_ERR_INVALID;
  }
  build_model(e);
#ifndef SDM_EMU
  hipStream_t st;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { g_create_err = "hipStreamCreate failed"; delete e; return SDM_ERR_HIP; }
  e->stream = st; e->own_stream = true;
  (void)hipEventCreate(&e->ev0); (void)hipEventCreate(&e->ev1);
  (void)hipEventCreateWithFlags(&e->ev_in, hipEventDisableTiming); (void)hipEventCreateWithFlags(&e->ev_out, hipEventDisableTiming);
#endif
  // weight arena (+ temb tables)
  void* p = nullptr;
  if (dev_malloc(&p, e->warena_bytes) != 0) { g_create_err = "cannot allocate weight arena"; sdm_destroy(e); return SDM_ERR_NOMEM; }
  e->warena = (unsigned char*)p;
  dev_memset(e->warena, 0, e->warena_bytes, e->stream);
  for (auto& L : e->convs) {
    L.w = (half_t*)(e->warena + L.w_off); L.b = (float*)(e->warena + L.b_off);
    if (L.split) L.w_lo = (half_t*)(e->warena + L.wlo_off);
    if (L.wdma_bytes) L.w_dma = (half_t*)(e->warena + e->canon_bytes + L.wdma_off);
    if (L.w3_bytes) L.w3 = e->warena + e->canon_bytes + L.w3_off;
  }
  for (auto& n : e->norms) { n.g = (float*)(e->warena + n.g_off); n.b = (float*)(e->warena + n.b_off); }
  for (auto& t : e->tembs) {
    void* q = nullptr;
    if (dev_malloc(&q, (size_t)kMinVariantRows * t.cout_pad * 4) != 0) { g_create_err = "cannot allocate temb tables"; sdm_destroy(e); return SDM_ERR_NOMEM; }
    t.table = (float*)q;
    dev_memset(q, 0, (size_t)kMinVariantRows * t.cout_pad * 4, e->stream);
  }
  e->variant_cap = kMinVariantRows;
  dev_sync(e->stream);
  *out = e;
  return SDM_OK;
}

void sdm_destroy(sdm_ctx* e) {
  if (e) dev_use(e->device);
  if (!e) return;
  dev_sync(e->stream);
  if (e->warena) dev_free(e->warena);
  if (e->arena) dev_free(e->arena);
  if (e->stage) dev_free(e->stage);
  load_ring_release(e);
  if (e->io_in) dev_free(e->io_in);
  if (e->io_out) dev_free(e->io_out);
  if (e->d_bias_sel) dev_free(e->d_bias_sel);
  for (auto& t : e->tembs) if (t.table) dev_free(t.table);
#ifndef SDM_EMU
  if (e->ev0) (void)hipEventDestroy(e->ev0);
  if (e->ev1) (void)hipEventDestroy(e->ev1);
  if (e->ev_in) (void)hipEventDestroy(e->ev_in);
  if (e->ev_out) (void)hipEventDestroy(e->ev_out);
  if (e->own_stream) (void)hipStreamDestroy((hipStream_t)e->stream);
#endif
  delete e;
}

const char* sdm_last_error(sdm_ctx* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

static float to_f32(const void* p, int dtype, size_t i) {
  if (dtype == SDM_F32) return ((const float*)p)[i];
  if (dtype == SDM_F16) return (float)((const half_t*)p)[i];
  uint32_t u = (uint32_t)((const uint16_t*)p)[i] << 16;   // bf16
  float f; memcpy(&f, &u, 4); return f;
}

int sdm_load_tensor(sdm_ctx* e, const char* name, int dtype, int ndim, const int64_t* shape, const void* host_ptr) {
  if (e) dev_use(e->device);
  if (!e || !name || !host_ptr) return SDM_ERR_INVALID;
  std::string key(name);
  // legacy VAE attention names
  static const char* legacy[4][2] = {{".query.", ".to_q."}, {".key.", ".to_k."}, {".value.", ".to_v."}, {".proj_attn.", ".to_out.0."}};
  if (key.find("mid_block.attentions.0") != std::string::npos)
    for (auto& l : legacy) { size_t pos = key.find(l[0]); if (pos != std::string::npos) key.replace(pos, strlen(l[0]), l[1]); }
  auto it = e->slots.find(key);
  if (it == e->slots.end()) { e->n_ignored++; return 0; }
  Slot& s = it->second;
  size_t n = 1, nexp = 1;
  for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
  for (auto d : s.shape) nexp *= (size_t)d;
  bool ok = (n == nexp);
  if (ok && (s.kind == SLOT_CONV_W)) {
    ok = ndim >= 2 && shape[0] == s.shape[0] && shape[1] == s.shape[1];
  }
  if (!ok) SDM_FAIL(e, SDM_ERR_INVALID, "size mismatch for %s: checkpoint has %zu elements, model expects %zu", name, n, nexp);
  e->finalized = false;
  if (s.kind == SLOT_HOST) {
    float* dst = e->hostblob.data() + s.host_off;
    for (size_t i = 0; i < n; ++i) dst[i] = to_f32(host_ptr, dtype, i);
  } else {
    // convert to fp32 into a pinned staging slot, copy + pack asynchronously on the engine stream; the slot is reused only after
    // its event has fired (the ring keeps kLoadSlots tensors in flight)
    float* dsrc = nullptr;
#ifndef SDM_EMU
    sdm_ctx::LoadSlot& sl = e->load_ring[e->load_next];
    e->load_next = (e->load_next + 1) % sdm_ctx::kLoadSlots;
    if (sl.busy) { SDM_CHECK_DEV(e, (int)hipEventSynchronize(sl.done)); sl.busy = false; }
    if (sl.cap < n * 4) {
      if (sl.host) (void)hipHostFree(sl.host);
      if (sl.dev) dev_free(sl.dev);
      sl.host = sl.dev = nullptr; sl.cap = 0;
      const size_t cap = std::max(rupz(n * 4, (size_t)1 << 20), (size_t)8 << 20);
      if (hipHostMalloc(&sl.host, cap, hipHostMallocDefault) != hipSuccess || dev_malloc(&sl.dev, cap) != 0) SDM_FAIL(e, SDM_ERR_NOMEM, "weight staging alloc failed");
      sl.cap = cap;
      if (!sl.done) SDM_CHECK_DEV(e, (int)hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    float* hs = (float*)sl.host;
    if (dtype == SDM_F32) memcpy(hs, host_ptr, n * 4);
    else for (size_t i = 0; i < n; ++i) hs[i] = to_f32(host_ptr, dtype, i);
    SDM_CHECK_DEV(e, dev_memcpy_h2d(sl.dev, hs, n * 4, e->stream));
    dsrc = (float*)sl.dev;
#else
    if (ensure_buf(e, &e->stage, &e->stage_bytes, std::max(n * 4, (size_t)1 << 20)) != 0) return SDM_ERR_NOMEM;
    std::vector<float> tmp;
    const void* src = host_ptr;
    if (dtype != SDM_F32) { tmp.resize(n); for (size_t i = 0; i < n; ++i) tmp[i] = to_f32(host_ptr, dtype, i); src = tmp.data(); }
    SDM_CHECK_DEV(e, dev_memcpy_h2d(e->stage, src, n * 4, e->stream));
    dsrc = (float*)e->stage;
#endif
    if (s.kind == SLOT_CONV_W) {
      ConvL& L = e->convs[s.layer];
      const int O = (int)s.shape[0], I = (int)s.shape[1];
      const size_t total = (size_t)L.Cin_pad * L.ntaps * L.Cout_pad;
      SDM_LAUNCH(pack_conv_weight_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, e->stream,
                 (const float*)dsrc, L.w, O, I, L.ntaps, L.Cin_pad, L.Cout_pad, s.ci_off, s.co_off, L.geglu,
                 s.w_scale * ldexpf(1.0f, L.w_exp), L.w_lo);
    } else if (s.kind == SLOT_CONV_B) {
      ConvL& L = e->convs[s.layer];
      SDM_LAUNCH(pack_bias_kernel, dim3(sdm_cdiv(L.Cout_pad, 256)), dim3(256), 0, e->stream, (const float*)dsrc, L.b, (int)s.shape[0],
                 L.Cout_pad, s.co_off, L.geglu);
    } else {
      NormL& nn = e->norms[s.layer];
      SDM_CHECK_DEV(e, dev_memcpy_d2d(s.kind == SLOT_NORM_G ? nn.g : nn.b, dsrc, n * 4, e->stream));
    }
#ifndef SDM_EMU
    SDM_CHECK_DEV(e, (int)hipEventRecord(sl.done, (hipStream_t)e->stream));
    sl.busy = true;
#else
    SDM_CHECK_DEV(e, dev_sync(e->stream));
#endif
  }
  if (!s.loaded) { s.loaded = true; e->n_loaded++; }
  return 1;
}

// Exact fold of aux_conv_in into every cross-attention K|V projection; fp64 accumulation on the host.
static int fold_cross_kv(sdm_ctx* e) {
  const sdm_config& c = e->cfg;
  const int ctx = c.cross_attention_dim;
  const float* H = e->hostblob.data();
  // Waux transposed to [36][ctx] so that the inner loop is contiguous
  std::vector<double> wa((size_t)36 * ctx), ba(ctx);
  for (int m = 0; m < ctx; ++m) {
    ba[m] = H[e->h_auxb + m];
    for (int j = 0; j < 36; ++j) wa[(size_t)j * ctx + m] = H[e->h_auxw + (size_t)m * 36 + j];
  }
  std::vector<const TfB*> blocks;
  for (auto& v : e->u_down_tf) for (auto& t : v) blocks.push_back(&t);
  blocks.push_back(&e->u_midtf);
  for (auto& v : e->u_up_tf) for (auto& t : v) blocks.push_back(&t);
  std::vector<float> wf, bf;
  for (const TfB* t : blocks) {
    const int C = t->C;
    ConvL& L = e->convs[t->kv2];
    wf.assign((size_t)2 * C * 36, 0.f); bf.assign((size_t)2 * C, 0.f);
    for (int o = 0; o < 2 * C; ++o) {
      const float* wrow = H + (o < C ? t->k_hoff + (size_t)o * ctx : t->v_hoff + (size_t)(o - C) * ctx);
      double b = 0.0;
      for (int m = 0; m < ctx; ++m) b += (double)wrow[m] * ba[m];
      bf[o] = (float)b;                                            // W_k . b_aux   (to_k / to_v have no bias of their own)
      for (int j = 0; j < 36; ++j) {                               // j = ci*9 + tap (OIHW order of aux_conv_in.weight)
        const double* wj = wa.data() + (size_t)j * ctx;
        double acc = 0.0;
        for (int m = 0; m < ctx; ++m) acc += (double)wrow[m] * wj[m];
        wf[(size_t)o * 36 + j] = (float)acc;
      }
    }
    // upload as an OIHW [2C][4][3][3] tensor; the 4 latent channels sit at channels 4..7 of the 16-channel U-Net input
    if (ensure_buf(e, &e->stage, &e->stage_bytes, std::max(wf.size() * 4, (size_t)1 << 20)) != 0) return SDM_ERR_NOMEM;
    SDM_CHECK_DEV(e, dev_memset(L.w, 0, (size_t)L.Cin_pad * 9 * L.Cout_pad * 2, e->stream));
    if (L.w_lo) SDM_CHECK_DEV(e, dev_memset(L.w_lo, 0, (size_t)L.Cin_pad * 9 * L.Cout_pad * 2, e->stream));
    SDM_CHECK_DEV(e, dev_memcpy_h2d(e->stage, wf.data(), wf.size() * 4, e->stream));
    const size_t total = (size_t)L.Cin_pad * 9 * L.Cout_pad;
    SDM_LAUNCH(pack_conv_weight_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, e->stream, (const float*)e->stage,
               L.w, 2 * C, 4, 9, L.Cin_pad, L.Cout_pad, 4, 0, 0, ldexpf(1.0f, L.w_exp), L.w_lo);
    SDM_CHECK_DEV(e, dev_sync(e->stream));
    SDM_CHECK_DEV(e, dev_memcpy_h2d(e->stage, bf.data(), bf.size() * 4, e->stream));
    SDM_LAUNCH(pack_bias_kernel, dim3(sdm_cdiv(L.Cout_pad, 256)), dim3(256), 0, e->stream, (const float*)e->stage, L.b, 2 * C, L.Cout_pad, 0, 0);
    SDM_CHECK_DEV(e, dev_sync(e->stream));
  }
  return 0;
}

// Derived weight layouts (k_conv.h "derived weight layouts") of a set of layers from their canonical K16 tensors.  Two passes so that
// ONE host round trip serves every layer: the |max| of each fp8-residual layer (device reductions into a small table), then the
// layout kernels with the per-layer e4m3 scale 2^e8 = the largest power of two that keeps max|w| * 2^e8 <= 448.
static int derive_layers(sdm_ctx* e, std::vector<ConvL*>& layers) {
  std::vector<ConvL*> f8;
  for (ConvL* L : layers) if ((L->w_dma && L->f8) || L->w3) f8.push_back(L);
  if (!f8.empty()) {
    const size_t tb = rupz(f8.size() * 4, 256);
    if (ensure_buf(e, &e->stage, &e->stage_bytes, std::max(tb, (size_t)1 << 20)) != 0) return SDM_ERR_NOMEM;
    SDM_CHECK_DEV(e, dev_memset(e->stage, 0, tb, e->stream));
    for (size_t i = 0; i < f8.size(); ++i) {
      const size_t n = (size_t)f8[i]->Cin_pad * f8[i]->ntaps * f8[i]->Cout_pad;
      SDM_LAUNCH(absmax_f16_kernel, dim3((unsigned)std::min<size_t>((n + 255) / 256, 2048)), dim3(256), 0, e->stream, (const half_t*)f8[i]->w, n,
                 (unsigned int*)e->stage + i);
    }
    std::vector<float> mx(f8.size());
    SDM_CHECK_DEV(e, dev_memcpy_d2h(mx.data(), e->stage, f8.size() * 4, e->stream));
    SDM_CHECK_DEV(e, dev_sync(e->stream));
    for (size_t i = 0; i < f8.size(); ++i) {
      const float wmax = ldexpf(mx[i], -f8[i]->w_exp);            // K16 keeps w * 2^w_exp
      int ex = 8;
      if (wmax > 0.0f && std::isfinite(wmax)) {
        ex = (int)floorf(log2f(448.0f / wmax));
        while (ldexpf(wmax, ex) > 448.0f) --ex;                   // guard the rounding of log2f
        ex = std::max(-60, std::min(60, ex));
      }
      f8[i]->f8_exp = ex;
    }
  }
  for (ConvL* L : layers) {
    if (L->w3)
      SDM_LAUNCH(derive_gemm_w3_kernel, dim3((unsigned)std::min<size_t>(((size_t)L->Cin_pad * L->Cout_pad / 4 + 255) / 256, 65535)), dim3(256), 0, e->stream,
                 (const half_t*)L->w, (const half_t*)L->w_lo, L->w3, L->Cin_pad, L->Cout_pad, ldexpf(1.0f, -L->w_exp), ldexpf(1.0f, L->f8_exp));
    if (!L->w_dma) continue;
    const size_t total = (size_t)L->Cin_pad * L->ntaps * L->Cout_pad;
    if (L->f8)
      SDM_LAUNCH(derive_conv_weight_f8_kernel, dim3((unsigned)std::min<size_t>((total / 4 + 255) / 256, 65535)), dim3(256), 0, e->stream,
                 (const half_t*)L->w, (const half_t*)L->w_lo, (unsigned char*)L->w_dma, L->Cin_pad, L->Cout_pad, L->ntaps, ldexpf(1.0f, -L->w_exp),
                 ldexpf(1.0f, L->f8_exp));
    else
      SDM_LAUNCH(derive_conv_weight_dma_kernel, dim3((unsigned)std::min<size_t>((total * (L->split ? 2 : 1) + 255) / 256, 65535)), dim3(256), 0, e->stream,
                 (const half_t*)L->w, (const half_t*)L->w_lo, L->w_dma, L->Cin_pad, L->Cout_pad, L->split ? 2 : 1);
  }
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return 0;
}

static void load_ring_release(sdm_ctx* e) {
#ifndef SDM_EMU
  for (auto& sl : e->load_ring) {
    if (sl.busy) { (void)hipEventSynchronize(sl.done); sl.busy = false; }
    if (sl.host) (void)hipHostFree(sl.host);
    if (sl.dev) dev_free(sl.dev);
    if (sl.done) (void)hipEventDestroy(sl.done);
    sl.host = sl.dev = nullptr; sl.cap = 0; sl.done = nullptr;
  }
#else
  (void)e;
#endif
}

int sdm_finalize_weights(sdm_ctx* e) {
  if (e) dev_use(e->device);
  if (!e) return SDM_ERR_INVALID;
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  load_ring_release(e);          // the checkpoint is in: give the pinned / device staging ring back
  { int rc = fold_cross_kv(e); if (rc) return rc; }
  {   // the canonical K16 tensors are complete: build every derived layout from them
    std::vector<ConvL*> all;
    for (auto& L : e->convs) all.push_back(&L);
    int rc = derive_layers(e, all);
    if (rc) return rc;
  }
  e->missing.clear();
  for (auto& k : e->slot_order) if (!e->slots[k].loaded) e->missing.push_back(k);
  e->variants.clear();
  e->finalized = true;
  return SDM_OK;
}

int sdm_weight_stats(sdm_ctx* e, int64_t* n_loaded, int64_t* n_missing, int64_t* n_ignored) {
  if (!e) return SDM_ERR_INVALID;
  if (n_loaded) *n_loaded = e->n_loaded;
  if (n_missing) *n_missing = (int64_t)e->missing.size();
  if (n_ignored) *n_ignored = e->n_ignored;
  return SDM_OK;
}

const char* sdm_missing_key(sdm_ctx* e, int64_t i) {
  if (!e || i < 0 || i >= (int64_t)e->missing.size()) return nullptr;
  return e->missing[(size_t)i].c_str();
}

int64_t sdm_weight_blob_bytes(sdm_ctx* e) { return e ? (int64_t)e->canon_bytes : 0; }
int sdm_export_weight_blob(sdm_ctx* e, void* dst) {
  if (e) dev_use(e->device);
  if (!e || !dst) return SDM_ERR_INVALID;
  SDM_CHECK_DEV(e, dev_memcpy_d2d(dst, e->warena, e->canon_bytes, e->stream));
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return SDM_OK;
}
int sdm_import_weight_blob(sdm_ctx* e, const void* src) {
  if (e) dev_use(e->device);
  if (!e || !src) return SDM_ERR_INVALID;
  SDM_CHECK_DEV(e, dev_memcpy_d2d(e->warena, src, e->canon_bytes, e->stream));
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  for (auto& kv : e->slots) if (kv.second.kind != SLOT_HOST && !kv.second.loaded) { kv.second.loaded = true; e->n_loaded++; }
  e->finalized = false;
  return SDM_OK;
}
int64_t sdm_host_blob_bytes(sdm_ctx* e) { return e ? (int64_t)(e->hostblob.size() * 4) : 0; }
int sdm_export_host_blob(sdm_ctx* e, void* dst) {
  if (!e || !dst) return SDM_ERR_INVALID;
  memcpy(dst, e->hostblob.data(), e->hostblob.size() * 4);
  return SDM_OK;
}
int sdm_import_host_blob(sdm_ctx* e, const void* src) {
  if (!e || !src) return SDM_ERR_INVALID;
  memcpy(e->hostblob.data(), src, e->hostblob.size() * 4);
  for (auto& kv : e->slots) if (kv.second.kind == SLOT_HOST && !kv.second.loaded) { kv.second.loaded = true; e->n_loaded++; }
  e->finalized = false;
  return SDM_OK;
}

int sdm_forward(sdm_ctx* e, const float* image, const float* trimap, int B, int S, const int32_t* is_trans, const float* coords, float* alpha,
                int ptr_kind, void* stream) {
  if (e) dev_use(e->device);
  if (!e || !image || !trimap || !alpha) return SDM_ERR_INVALID;
  return forward_impl(e, 0, image, trimap, B, S, S, S, is_trans, coords, 4, 0, true, alpha, ptr_kind, stream);
}

int sdm_forward_ex(sdm_ctx* e, const float* image, const float* aux, int B, int S, const int32_t* is_trans, const float* cond, int cond_dim,
                   int cond_kind, int use_attention_mask, float* alpha, int ptr_kind, void* stream) {
  if (e) dev_use(e->device);
  if (!e || !image || !aux || !alpha) return SDM_ERR_INVALID;
  if (cond_kind != SDM_COND_BOX && cond_kind != SDM_COND_POINTS) SDM_FAIL(e, SDM_ERR_INVALID, "unknown conditioning kind %d", cond_kind);
  return forward_impl(e, 0, image, aux, B, S, S, S, is_trans, cond, cond_dim, cond_kind, use_attention_mask != 0, alpha, ptr_kind, stream);
}

int sdm_forward_rect(sdm_ctx* e, const float* image, const float* aux, int B, int SH, int SW, const int32_t* is_trans, const float* cond,
                     int cond_dim, int cond_kind, int use_attention_mask, float* alpha, int ptr_kind, void* stream) {
  if (e) dev_use(e->device);
  if (!e || !image || !aux || !alpha) return SDM_ERR_INVALID;
  if (cond_kind != SDM_COND_BOX && cond_kind != SDM_COND_POINTS) SDM_FAIL(e, SDM_ERR_INVALID, "unknown conditioning kind %d", cond_kind);
  return forward_impl(e, 0, image, aux, B, SH, SW, 0, is_trans, cond, cond_dim, cond_kind, use_attention_mask != 0, alpha, ptr_kind, stream);
}

int sdm_apply_matte(sdm_ctx* e, const float* image, const float* trimap, int B, int H, int W, int S, int is_transparent, float* alpha,
                    int ptr_kind, void* stream) {
  if (e) dev_use(e->device);
  if (!e || !image || !trimap || !alpha) return SDM_ERR_INVALID;
  std::vector<int32_t> it((size_t)std::max(B, 1), is_transparent ? 1 : 0);
  return forward_impl(e, 1, image, trimap, B, H, W, S, it.data(), nullptr, 4, 0, true, alpha, ptr_kind, stream);
}

/* Give the activation arena and the I/O staging buffers back to the driver (weights stay resident).  The next forward
 * re-allocates what it needs. */
int sdm_release_memory(sdm_ctx* e) {
  if (e) dev_use(e->device);
  if (!e) return SDM_ERR_INVALID;
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  if (e->arena) { dev_free(e->arena); e->arena = nullptr; e->arena_bytes = 0; }
  if (e->io_in) { dev_free(e->io_in); e->io_in = nullptr; e->io_in_bytes = 0; }
  if (e->io_out) { dev_free(e->io_out); e->io_out = nullptr; e->io_out_bytes = 0; }
  if (e->stage) { dev_free(e->stage); e->stage = nullptr; e->stage_bytes = 0; }
  return SDM_OK;
}
int sdm_set_option(const char* name, int value) {
  OptEntry* o = opt_find(name);
  if (!o) return SDM_ERR_INVALID;
  std::unique_lock<std::shared_mutex> g(g_opt_mu);      // waits for the forwards in flight on other host threads
  o->value = value;
  return SDM_OK;
}

int sdm_get_option(const char* name, int* value) {
  OptEntry* o = opt_find(name);
  if (!o || !value) return SDM_ERR_INVALID;
  *value = o->value;
  return SDM_OK;
}

void sdm_reset_options(void) {
  std::unique_lock<std::shared_mutex> g(g_opt_mu);
  for (auto& o : g_opts) o.value = o.def;
}

const char* sdm_option_name(int i) {
  return (i >= 0 && i < (int)(sizeof(g_opts) / sizeof(g_opts[0]))) ? g_opts[i].name : nullptr;
}

const char* sdm_option_help(int i) {
  return (i >= 0 && i < (int)(sizeof(g_opts) / sizeof(g_opts[0]))) ? g_opts[i].what : nullptr;
}

int sdm_kernel_counts(char* buf, int cap) {
  std::string out;
  std::lock_guard<std::mutex> g(g_count_mu);
  for (auto& kv : g_kernel_counts) { out += kv.first; out += "="; out += std::to_string(kv.second); out += ";"; }
  if (buf && cap > 0) { const int n = (int)out.size() < cap - 1 ? (int)out.size() : cap - 1; memcpy(buf, out.data(), (size_t)n); buf[n] = 0; }
  return (int)out.size();
}

void sdm_kernel_counts_reset(void) { std::lock_guard<std::mutex> g(g_count_mu); g_kernel_counts.clear(); }

int64_t sdm_weight_bytes(sdm_ctx* e) { return e ? (int64_t)e->warena_bytes : 0; }

int64_t sdm_resident_bytes(sdm_ctx* e) {
  return e ? (int64_t)(e->warena_bytes + e->arena_bytes + e->io_in_bytes + e->io_out_bytes + e->stage_bytes) : 0;
}

int sdm_apply_matte_node(sdm_ctx* e, const float* image, const float* trimap, int B, int H, int W, int trimap_h, int trimap_w, int S,
                         int is_transparent, int output_mode, int mask_refine, double trimap_constraint, float* alpha, float* matted, int ptr_kind,
                         void* stream) {
  if (e) dev_use(e->device);
  if (!e || !image || !trimap || !alpha || !matted) return SDM_ERR_INVALID;
  if (output_mode < 0 || output_mode > 2) SDM_FAIL(e, SDM_ERR_INVALID, "unknown output mode %d", output_mode);
  if (trimap_h <= 0 || trimap_w <= 0) SDM_FAIL(e, SDM_ERR_INVALID, "bad trimap size %dx%d", trimap_h, trimap_w);
  // the reference indexes the (H, W) alpha with the trimap only for mask_refine and matted_rgb (sdmatte_nodes.py:365-380,390-394):
  // only those need equal sizes
  if ((trimap_h != H || trimap_w != W) && (mask_refine || output_mode == 2))
    SDM_FAIL(e, SDM_ERR_INVALID, "trimap %dx%d does not match the image %dx%d (needed by mask_refine / matted_rgb)", trimap_h, trimap_w, H, W);
  std::vector<int32_t> it((size_t)std::max(B, 1), is_transparent ? 1 : 0);
  NodeTail tail; tail.output_mode = output_mode; tail.mask_refine = mask_refine ? 1 : 0; tail.c = trimap_constraint; tail.matted = matted;
  tail.TH = trimap_h; tail.TW = trimap_w;
  return forward_impl(e, 1, image, trimap, B, H, W, S, it.data(), nullptr, 4, 0, true, alpha, ptr_kind, stream, &tail);
}

int sdm_synchronize(sdm_ctx* e) {
  if (e) dev_use(e->device);
  if (!e) return SDM_ERR_INVALID;
  SDM_CHECK_DEV(e, dev_sync(e->stream));
#ifndef SDM_EMU
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, e->ev0, e->ev1) == hipSuccess) e->last_ms = ms;
  if (!e->prof.empty()) {
    std::map<std::string, sdm_ctx::ProfAgg> agg;
    e->prof_dump = "kernel,ms,gflop,mbytes,desc\n";
    for (auto& r : e->prof) {
      float t = 0.f;
      (void)hipEventElapsedTime(&t, r.e0, r.e1);
      char line[512];
      snprintf(line, sizeof(line), "%s,%.4f,%.3f,%.3f,%s\n", r.name.c_str(), t, r.flops * 1e-9, r.bytes * 1e-6, r.desc.c_str());
      e->prof_dump += line;
      auto& a = agg[r.name];
      a.name = r.name; a.ms += t; a.n += 1; a.flops += r.flops; a.bytes += r.bytes;
      (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1);
    }
    e->prof.clear();
    e->prof_agg.clear();
    for (auto& kv : agg) e->prof_agg.push_back(kv.second);
  }
#endif
  return SDM_OK;
}

float sdm_last_forward_ms(sdm_ctx* e) { return e ? e->last_ms : 0.f; }

int sdm_profile_enable(sdm_ctx* e, int on) { if (!e) return SDM_ERR_INVALID; e->prof_on = on != 0; return SDM_OK; }
int sdm_profile_count(sdm_ctx* e) { return e ? (int)e->prof_agg.size() : 0; }
const char* sdm_profile_dump(sdm_ctx* e) { return e ? e->prof_dump.c_str() : ""; }
int sdm_profile_get(sdm_ctx* e, int i, const char** name, float* ms, int64_t* launches, double* flops, double* bytes) {
  if (!e || i < 0 || i >= (int)e->prof_agg.size()) return SDM_ERR_INVALID;
  auto& a = e->prof_agg[(size_t)i];
  if (name) *name = a.name.c_str();
  if (ms) *ms = a.ms;
  if (launches) *launches = a.n;
  if (flops) *flops = a.flops;
  if (bytes) *bytes = a.bytes;
  return SDM_OK;
}

// ---- single-operator entry points ----
int sdm_conv_num_cfgs(int ntaps, int stride) { return conv_num_cfgs(ntaps, stride); }

int sdm_op_conv_ex(sdm_ctx* e, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int Hin, int Win, int up, int stride,
                   int pad_mode, int ntaps, const float* w, const float* bias, int O, void* out, int out_f32, const void* res, int res_f32,
                   int geglu, float out_scale, int tile_cfg, int split, const float* gn_gamma, const float* gn_beta, float gn_eps,
                   int gn_groups, int gn_silu) {
  if (e) dev_use(e->device);
  if (!e || !in0 || !w || !out) return SDM_ERR_INVALID;
  if (C0 % 16 || C1 % 16) SDM_FAIL(e, SDM_ERR_INVALID, "sdm_op_conv: channel counts must be multiples of 16");
  if (split && !in_f32) SDM_FAIL(e, SDM_ERR_INVALID, "sdm_op_conv_ex: split precision takes fp32 activations");
  ConvL L;
  L.name = "op"; L.ntaps = ntaps; L.I = C0 + C1; L.O = O; L.Cin_pad = C0 + C1; L.Cout_pad = rup(O, geglu ? 64 : 32); L.geglu = geglu;
  L.split = split ? 1 : 0; L.w_exp = split ? kSplitWeightExp : 0;
  void* wp = nullptr; void* bp = nullptr; void* wl = nullptr;
  const size_t wbytes = (size_t)L.Cin_pad * ntaps * L.Cout_pad * 2;
  SDM_CHECK_DEV(e, dev_malloc(&wp, wbytes));
  SDM_CHECK_DEV(e, dev_malloc(&bp, (size_t)L.Cout_pad * 4));
  if (split) { SDM_CHECK_DEV(e, dev_malloc(&wl, wbytes)); dev_memset(wl, 0, wbytes, e->stream); }
  dev_memset(wp, 0, wbytes, e->stream); dev_memset(bp, 0, (size_t)L.Cout_pad * 4, e->stream);
  L.w = (half_t*)wp; L.b = (float*)bp; L.w_lo = (half_t*)wl;
  const size_t total = (size_t)L.Cin_pad * ntaps * L.Cout_pad;
  SDM_LAUNCH(pack_conv_weight_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, e->stream, w, L.w, O, L.I, ntaps,
             L.Cin_pad, L.Cout_pad, 0, 0, geglu, ldexpf(1.0f, L.w_exp), L.w_lo);
  if (bias) SDM_LAUNCH(pack_bias_kernel, dim3(sdm_cdiv(L.Cout_pad, 256)), dim3(256), 0, e->stream, bias, L.b, O, L.Cout_pad, 0, geglu);
  void* wd = nullptr;
  if (ntaps == 9 && L.Cout_pad >= 128 && !geglu) {      // stage-ordered copy for the DMA-weight kernel (tile cfg 0), as in the engine
    SDM_CHECK_DEV(e, dev_malloc(&wd, total * (split ? 2 : 1) * 2));
    L.w_dma = (half_t*)wd;
    L.f8 = (split && L.Cin_pad % 32 == 0 && conv_f8_enabled()) ? 1 : 0;
  }
  if (ntaps == 1 && split && L.Cout_pad >= 128 && L.Cin_pad % 32 == 0 && conv_f8_enabled() && gemm_f8_enabled()) {      // fp8-residual copy for the 8-wave GEMM kernel
    SDM_CHECK_DEV(e, dev_malloc(&wd, total * 4));
    dev_memset(wd, 0, total * 4, e->stream);
    L.w_dma = (half_t*)wd; L.f8 = 1;
  }
  if (L.w_dma) { std::vector<ConvL*> one{&L}; TRY(derive_layers(e, one)); }
  int Ho = Hin << up, Wo = Win << up;
  if (stride == 2) { Ho /= 2; Wo /= 2; }
  const int Cst = rup(geglu ? O / 2 : O, 4);   // rows are stored with 4-channel vectors
  T tin0, tin1, tout, tres;
  tin0.p = (void*)in0; tin0.N = N; tin0.H = Hin; tin0.W = Win; tin0.C = C0; tin0.f32 = in_f32;
  tin1 = tin0; tin1.p = (void*)in1; tin1.C = C1;
  tout.p = out; tout.N = N; tout.H = Ho; tout.W = Wo; tout.C = Cst; tout.f32 = out_f32;
  tres = tout; tres.p = (void*)res; tres.f32 = res_f32;
  if (e->dbg_cmask) { tin0.cmask = const_cast<unsigned char*>(e->dbg_cmask); tin0.cm_bytes = 1; e->dbg_cmask = nullptr; }      // (borrowed: tin0 is never tfree'd)
  ConvArgs a; a.in0 = &tin0; a.in1 = in1 ? &tin1 : nullptr; a.out = &tout; a.stride = stride; a.pad_mode = pad_mode; a.up = up;
  a.res = res ? &tres : nullptr; a.out_scale = out_scale; a.force_cfg = tile_cfg; a.cout_valid = Cst;
  int rc;
  if (gn_gamma) {
    // GroupNorm(+SiLU) of the input applied inside the conv's operand staging (the production path of every ResBlock conv):
    // statistics by the stand-alone kernel, scale/shift table, then the fused-GN instantiation of tile cfg 0 / 4 / 5
    if (ntaps != 9 || stride != 1 || up) { dev_free(wp); dev_free(bp); if (wl) dev_free(wl); SDM_FAIL(e, SDM_ERR_INVALID, "sdm_op_conv_ex: fused GroupNorm needs a 3x3 stride-1 conv"); }
    if (a.force_cfg != 0 && a.force_cfg != 4 && a.force_cfg != 5) a.force_cfg = (L.Cout_pad <= 32) ? 4 : 0;
    const int act_prev = e->act_f32;
    rc = run_two_pass(e, [&]() {
      T scratch; float* scale; float* shift;
      TRY(gn_scale_shift(e, in0, in1, C0, C1, in_f32, N, Hin * Win, gn_groups, gn_gamma, gn_beta, gn_eps, nullptr, 0, nullptr, 0, false, &scratch,
                         &scale, &shift));
      ConvArgs b = a;
      b.gn_scale = e->dry ? (const float*)16 : scale; b.gn_shift = shift; b.gn_silu = gn_silu;
      int r2 = op_conv(e, L, b);
      tfree(e, scratch);
      return r2;
    });
    e->act_f32 = act_prev;
  } else {
    rc = run_two_pass(e, [&]() { return op_conv(e, L, a); });      // (the arena holds the split-K workspace, if the layer is split)
  }
  dev_sync(e->stream);
  dev_free(wp); dev_free(bp); if (wl) dev_free(wl);
  if (wd) dev_free(wd);
  return rc;
}

/* Plane-fed GEMM (k_gemm.h) as a stand-alone operator: x fp32 [N*H*W][K] (device) is converted to P3 planes (to_p3_kernel, or LayerNorm with P3 output
 * when ln_gamma is given), w fp32 [O][K] is packed to K16 -> W3 exactly as a model layer.  mode 0: fp32 out (+bias, +fp32 residual); 1: GEGLU (O = 2 x outputs);
 * 3: linear (+residual) to P3; both P3 results are decoded to fp32 (hi + xl * 2^-11) into `out`; 2: raw q | k | v operand planes (fp16 hi [rows][O] then the
 * e5m2 pair plane, pair plane for channels < lo_cols only); 4: mode 0 + the consumer's GroupNorm statistics, [N][*srows][O][2] floats into `stats`. */
int sdm_op_gemm_p3(sdm_ctx* e, const float* x, int N, int H, int W, int K, const float* w, const float* bias, int O, int mode, const float* res,
                   const float* ln_gamma, const float* ln_beta, float ln_eps, int lo_cols, void* out, float* stats, int* srows) {
  if (e) dev_use(e->device);
  if (!e || !x || !w || !out) return SDM_ERR_INVALID;
  const int geglu = mode == 1;
  if (K % 32 || O % (geglu ? 64 : 32)) SDM_FAIL(e, SDM_ERR_INVALID, "sdm_op_gemm_p3: K %% 32 and O %% 32 (GEGLU: 64) required");
  ConvL L;
  L.name = "op_gemm_p3"; L.ntaps = 1; L.I = K; L.O = O; L.Cin_pad = K; L.Cout_pad = O; L.geglu = geglu; L.split = 1; L.w_exp = kSplitWeightExp;
  void* wp = nullptr; void* bp = nullptr; void* wl = nullptr; void* w3 = nullptr;
  const size_t total = (size_t)K * O, wbytes = total * 2;
  SDM_CHECK_DEV(e, dev_malloc(&wp, wbytes)); SDM_CHECK_DEV(e, dev_malloc(&wl, wbytes)); SDM_CHECK_DEV(e, dev_malloc(&bp, (size_t)O * 4));
  SDM_CHECK_DEV(e, dev_malloc(&w3, total * 4));
  dev_memset(wp, 0, wbytes, e->stream); dev_memset(wl, 0, wbytes, e->stream); dev_memset(bp, 0, (size_t)O * 4, e->stream);
  L.w = (half_t*)wp; L.w_lo = (half_t*)wl; L.b = (float*)bp; L.w3 = (unsigned char*)w3; L.w3_bytes = total * 4;
  SDM_LAUNCH(pack_conv_weight_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65535)), dim3(256), 0, e->stream, w, L.w, O, K, 1, K, O, 0, 0, geglu,
             ldexpf(1.0f, L.w_exp), L.w_lo);
  if (bias) SDM_LAUNCH(pack_bias_kernel, dim3(sdm_cdiv(O, 256)), dim3(256), 0, e->stream, bias, L.b, O, O, 0, geglu);
  { std::vector<ConvL*> one{&L}; TRY(derive_layers(e, one)); }
  const int Cst = geglu ? O / 2 : O;
  const int act_prev = e->act_f32;
  e->act_f32 = 1;
  int rc = run_two_pass(e, [&]() -> int {
    T tx, xp, to, tres;
    tx.p = (void*)x; tx.N = N; tx.H = H; tx.W = W; tx.C = K; tx.f32 = 1;
    if (ln_gamma) {
      NormL nl; nl.C = K; nl.g = const_cast<float*>(ln_gamma); nl.b = const_cast<float*>(ln_beta);
      TRY(op_ln(e, nl, tx, ln_eps, &xp, kFmtP3));
    } else {
      TRY(op_to_p3(e, tx, &xp));
    }
    tres.p = (void*)res; tres.N = N; tres.H = H; tres.W = W; tres.C = Cst; tres.f32 = 1;
    const bool planes = (mode == 1 || mode == 3);
    if (planes) to = talloc(e, N, H, W, Cst, kFmtP3);
    else { to.p = out; to.N = N; to.H = H; to.W = W; to.C = Cst; to.f32 = (mode == 2) ? 3 : 1; }
    to.want_stats = (mode == 4);
    TRY(op_gemm_p3(e, L, xp, &to, res ? &tres : nullptr, lo_cols));
    if (!e->dry) {
      if (planes) SDM_LAUNCH(from_p3_kernel, dim3((unsigned)std::min<long>((to.rows() * Cst + 255) / 256, 1 << 20)), dim3(256), 0, e->stream, (const unsigned char*)to.p, (float*)out, to.rows(), Cst);
      if (mode == 4 && stats) {
        SDM_CHECK_DEV(e, dev_memcpy_d2d(stats, to.stats, (size_t)N * to.srows * Cst * 2 * 4, e->stream));
        if (srows) *srows = to.srows;
      }
    }
    if (planes) tfree(e, to);
    else if (to.sbytes) { tfree_raw(e, to.soff, to.sbytes); to.sbytes = 0; }
    tfree(e, xp);
    return 0;
  });
  e->act_f32 = act_prev;
  dev_sync(e->stream);
  dev_free(wp); dev_free(wl); dev_free(bp); dev_free(w3);
  return rc;
}

/* Test hook: `mask` (device, [N][Hin][Win] bytes, class ids 0..4; k_misc.h cmask_*) is the class plane of the input of the NEXT sdm_op_conv_ex call -
 * the conv then leaves the output tiles of constant regions to const_tile_fill_kernel, as the VAE encoder does for the trimap images. */
int sdm_debug_set_input_cmask(sdm_ctx* e, const unsigned char* mask) {
  if (!e) return SDM_ERR_INVALID;
  e->dbg_cmask = mask;
  return SDM_OK;
}

int sdm_op_conv(sdm_ctx* e, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int Hin, int Win, int up, int stride,
                int pad_mode, int ntaps, const float* w, const float* bias, int O, void* out, int out_f32, const void* res, int res_f32,
                int geglu, float out_scale, int tile_cfg) {
  return sdm_op_conv_ex(e, in0, in1, C0, C1, in_f32, N, Hin, Win, up, stride, pad_mode, ntaps, w, bias, O, out, out_f32, res, res_f32, geglu,
                        out_scale, tile_cfg, 0, nullptr, nullptr, 0.0f, 32, 0);
}

/* ---- test hooks for the exact algebraic folds (SURVEY.md 8a "each needs a fold == unfold CPU test") ----------------------- */

/* Run ONE packed layer of the loaded model (by name, e.g. "unet.down_blocks.0.attentions.0.transformer_blocks.0.attn2.kv_folded",
 * "...attn1.qkv") on an fp32 NHWC input with the layer's padded input channel count; fp32 NHWC output with `Cout` channels. */
int sdm_debug_run_layer(sdm_ctx* e, const char* layer_name, const float* x, int N, int H, int W, float* out, int Cout) {
  if (e) dev_use(e->device);
  if (!e || !layer_name || !x || !out) return SDM_ERR_INVALID;
  if (!e->finalized) SDM_FAIL(e, SDM_ERR_STATE, "weights not finalised");
  const ConvL* L = nullptr;
  for (auto& c : e->convs) if (c.name == layer_name) { L = &c; break; }
  if (!L) SDM_FAIL(e, SDM_ERR_INVALID, "no packed layer named %s", layer_name);
  if (Cout % 4 || Cout > L->Cout_pad) SDM_FAIL(e, SDM_ERR_INVALID, "bad Cout %d for layer %s", Cout, layer_name);
  T tin, tout;
  tin.p = (void*)x; tin.N = N; tin.H = H; tin.W = W; tin.C = L->Cin_pad; tin.f32 = 1;
  tout.p = out; tout.N = N; tout.H = H; tout.W = W; tout.C = Cout; tout.f32 = 1;
  ConvArgs a; a.in0 = &tin; a.out = &tout; a.cout_valid = Cout;
  int rc = run_two_pass(e, [&]() { return op_conv(e, *L, a); });
  dev_sync(e->stream);
  return rc;
}

/* Folded conv1 bias row (conv1.bias + time_emb_proj(silu(emb)), emb = time_embedding(trans) + bbox_embedding(coords)) of the
 * i-th ResBlock that has a time embedding, for one (is_trans, box) conditioning; out: cout floats on the HOST. */
int sdm_debug_temb_row(sdm_ctx* e, int temb_index, int is_trans, const float* coords4, float* out_host, int cout) {
  if (e) dev_use(e->device);
  if (!e || !out_host || temb_index < 0 || temb_index >= (int)e->tembs.size()) return SDM_ERR_INVALID;
  if (!e->finalized) SDM_FAIL(e, SDM_ERR_STATE, "weights not finalised");
  int32_t it = is_trans;
  TRY(prepare_variants(e, 1, &it, coords4, 4, 0));
  Variant v; v.trans = 1 - is_trans; v.kind = 0; v.c.assign(4, 0.0f);
  const float def[4] = {0.f, 0.f, 1.f, 1.f};
  for (int k = 0; k < 4; ++k) v.c[k] = coords4 ? coords4[k] : def[k];
  int idx = -1;
  for (size_t i = 0; i < e->variants.size(); ++i) if (e->variants[i] == v) idx = (int)i;
  const TembL& t = e->tembs[(size_t)temb_index];
  if (idx < 0 || cout > t.cout) SDM_FAIL(e, SDM_ERR_INVALID, "temb row: variant not found / bad cout");
  SDM_CHECK_DEV(e, dev_memcpy_d2h(out_host, t.table + (size_t)idx * t.cout_pad, (size_t)cout * 4, e->stream));
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return SDM_OK;
}

/* Bench/ablation helper (not used by the engine): times `iters` launches of one conv with HIP events; returns ms per launch
 * (negative on error).  ablate bits: see ConvParams::ablate. */
/* bench only: ms per launch of the plane-fed GEMM (k_gemm.h) on random operands: M rows, K -> O, epilogue `epi` (0 fp32, 1 GEGLU, 2 q|k|v planes, 3 P3, 4 fp32 +
 * statistics; bit 8: + fp32 residual).  The row tile follows the option gemm_p3_tile. */
float sdm_bench_gemm_p3(sdm_ctx* e, long M, int K, int O, int epi_flags, int iters) {
  if (e) dev_use(e->device);
  if (!e || K % 32 || O % 64) return -1.f;
#ifdef SDM_EMU
  (void)M; (void)epi_flags; (void)iters;
  return -1.f;
#else
  const int epi = epi_flags & 255, resf = (epi_flags >> 8) & 1;
  void *xf = nullptr, *xp = nullptr, *w3 = nullptr, *bp = nullptr, *out = nullptr, *resb = nullptr, *st = nullptr;
  const int Cst = epi == 1 ? O / 2 : O;
  int p_sb = 127 - 8;
  const size_t w3b = (size_t)K * O * 4;
  if (dev_malloc(&xf, (size_t)M * K * 4) || dev_malloc(&xp, p3_rows_pad((size_t)M) * K * 3) || dev_malloc(&w3, w3b) || dev_malloc(&bp, (size_t)O * 4) ||
      dev_malloc(&out, p3_rows_pad((size_t)M) * Cst * 4 + 256)) return -2.f;
  if (resf && dev_malloc(&resb, (size_t)M * Cst * 4)) return -2.f;
  if (epi == 4 && dev_malloc(&st, ((size_t)(M + 63) / 64 * 2 + 8) * O * 8)) return -2.f;
  SDM_LAUNCH(fill_random_f32_kernel, dim3(4096), dim3(256), 0, e->stream, (float*)xf, (long)M * K, 5u, 1.0f);
  SDM_LAUNCH(to_p3_kernel, dim3(4096), dim3(256), 0, e->stream, (const float*)xf, (unsigned char*)xp, M, K);
  {   // weights: random fp32 [O][K] -> K16 hi | lo -> W3, as a model layer
    void *wf = nullptr, *wp = nullptr, *wl = nullptr;
    if (dev_malloc(&wf, (size_t)K * O * 4) || dev_malloc(&wp, (size_t)K * O * 2) || dev_malloc(&wl, (size_t)K * O * 2)) return -2.f;
    SDM_LAUNCH(fill_random_f32_kernel, dim3(2048), dim3(256), 0, e->stream, (float*)wf, (long)K * O, 17u, 0.05f);
    ConvL L;
    L.name = "bench"; L.ntaps = 1; L.I = K; L.O = O; L.Cin_pad = K; L.Cout_pad = O; L.split = 1; L.w_exp = kSplitWeightExp;
    L.w = (half_t*)wp; L.w_lo = (half_t*)wl; L.w3 = (unsigned char*)w3; L.w3_bytes = w3b;
    SDM_LAUNCH(pack_conv_weight_kernel, dim3((unsigned)std::min<size_t>(((size_t)K * O + 255) / 256, 65535)), dim3(256), 0, e->stream, (const float*)wf, L.w, O, K, 1, K, O,
               0, 0, 0, ldexpf(1.0f, L.w_exp), L.w_lo);
    std::vector<ConvL*> one{&L};
    if (derive_layers(e, one) != 0) return -2.f;
    dev_free(wf); dev_free(wp); dev_free(wl);
    p_sb = 127 - L.f8_exp;
  }
  if (resf) SDM_LAUNCH(fill_random_f32_kernel, dim3(4096), dim3(256), 0, e->stream, (float*)resb, (long)M * Cst, 31u, 1.0f);
  dev_memset(bp, 0, (size_t)O * 4, e->stream);
  GemmP3Params p;
  memset(&p, 0, sizeof(p));
  p.M = M; p.K = K; p.a_hi = (const half_t*)xp; p.a_xl = (const unsigned char*)xp + p3_rows_pad((size_t)M) * K * 2;
  p.w = (const unsigned char*)w3; p.N = O; p.bias = (const float*)bp; p.out = out; p.ldo = Cst; p.n_valid = Cst;
  p.out_lo_off = epi == 2 ? (size_t)M * Cst : p3_rows_pad((size_t)M) * Cst * 2; p.lo_cols = (O / 3) * 2;
  if (resf) { p.res = (const float*)resb; p.ldr = Cst; }
  if (epi == 4) { p.stats = (float*)st; p.rows_per_img = (int)M; }
  p.sa = 127 - 11; p.sb = p_sb; p.ablate = opt("gemm_p3_ablate");
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch_gemm_p3(p, epi, e->stream);
  (void)hipEventRecord(e0, (hipStream_t)e->stream);
  for (int i = 0; i < iters; ++i) launch_gemm_p3(p, epi, e->stream);
  (void)hipEventRecord(e1, (hipStream_t)e->stream);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const hipError_t le = hipGetLastError();
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  dev_free(xf); dev_free(xp); dev_free(w3); dev_free(bp); dev_free(out); if (resb) dev_free(resb); if (st) dev_free(st);
  if (le != hipSuccess) { e->err = hipGetErrorString(le); return -3.f; }
  return ms / (float)iters;
#endif
}

float sdm_bench_conv(sdm_ctx* e, int N, int H, int W, int Cin, int Cout, int ntaps, int stride, int in_f32, int tile_cfg, int ablate, int iters) {
  // in_f32: bit 0 = fp32 activations, bit 1 = split-precision kernel (implies fp32), bit 2 = fused GroupNorm+SiLU staging,
  // bit 3 = producer / consumer form of the split-precision DMA kernel
  const int split = (in_f32 >> 1) & 1, gnf = (in_f32 >> 2) & 1, pcf = (in_f32 >> 3) & 1, f8f = (in_f32 >> 4) & 1;      // bit 4: fp8-residual kernel
  // bits 5-7: what the engine's ResBlock convs do in the default precision - fp32 output, fp32 residual, GroupNorm statistics of the consumer
  const int of32 = (in_f32 >> 5) & 1, resf = (in_f32 >> 6) & 1, statf = (in_f32 >> 7) & 1;
  in_f32 = (in_f32 & 1) | split;
  if (e) dev_use(e->device);
  if (!e) return -1.f;
#ifdef SDM_EMU
  return -1.f;
#else
  ConvL L;
  L.name = "bench"; L.ntaps = ntaps; L.I = Cin; L.O = Cout; L.Cin_pad = rup(Cin, 16); L.Cout_pad = rup(Cout, 32);
  void *wp = nullptr, *bp = nullptr, *in = nullptr, *out = nullptr, *wl = nullptr, *gnt = nullptr, *resb = nullptr, *statb = nullptr;
  const int Ho = stride == 2 ? H / 2 : H, Wo = stride == 2 ? W / 2 : W;
  const size_t wbytes = (size_t)L.Cin_pad * ntaps * L.Cout_pad * 2, inb = (size_t)N * H * W * L.Cin_pad * (in_f32 ? 4 : 2),
               outb = (size_t)N * Ho * Wo * L.Cout_pad * (of32 ? 4 : 2);
  if (dev_malloc(&wp, wbytes) || dev_malloc(&bp, (size_t)L.Cout_pad * 4) || dev_malloc(&in, inb) || dev_malloc(&out, outb)) return -2.f;
  if (resf) {
    if (dev_malloc(&resb, (size_t)N * Ho * Wo * L.Cout_pad * 4)) return -2.f;
    SDM_LAUNCH(fill_random_f32_kernel, dim3(4096), dim3(256), 0, e->stream, (float*)resb, (long)N * Ho * Wo * L.Cout_pad, 31u, 1.0f);
  }
  if (statf && dev_malloc(&statb, (size_t)N * (sdm_cdiv(Ho, 4) * sdm_cdiv(Wo, 8) * 4 + 64) * L.Cout_pad * 8)) return -2.f;      // enough partial rows for every tile cfg
  if (split) { if (dev_malloc(&wl, wbytes)) return -2.f; SDM_LAUNCH(fill_random_f16_kernel, dim3(2048), dim3(256), 0, e->stream, (half_t*)wl, (long)(wbytes / 2), 19u, 0.0001f); }
  if (gnf) {      // scale = 1, shift = 0 table [N][Cin] x 2
    if (dev_malloc(&gnt, (size_t)N * L.Cin_pad * 8)) return -2.f;
    SDM_LAUNCH(fill_random_f32_kernel, dim3(64), dim3(256), 0, e->stream, (float*)gnt, (long)N * L.Cin_pad * 2, 23u, 1.0f);
  }
  dev_memset(bp, 0, (size_t)L.Cout_pad * 4, e->stream);
  SDM_LAUNCH(fill_random_f16_kernel, dim3(2048), dim3(256), 0, e->stream, (half_t*)wp, (long)(wbytes / 2), 17u, 0.05f);
  if (in_f32) SDM_LAUNCH(fill_random_f32_kernel, dim3(4096), dim3(256), 0, e->stream, (float*)in, (long)(inb / 4), 5u, 1.0f);
  else SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)in, (long)(inb / 2), 5u, 1.0f);
  L.w = (half_t*)wp; L.b = (float*)bp;
  T tin, tout;
  tin.p = in; tin.N = N; tin.H = H; tin.W = W; tin.C = L.Cin_pad; tin.f32 = in_f32;
  tout.p = out; tout.N = N; tout.H = Ho; tout.W = Wo; tout.C = L.Cout_pad; tout.f32 = 0;
  ConvParams p;
  memset(&p, 0, sizeof(p));
  p.in0 = in; p.C0 = L.Cin_pad; p.in_f32 = in_f32; p.N = N; p.Hin = H; p.Win = W; p.Hout = Ho; p.Wout = Wo; p.pad_t = p.pad_l = 1;
  p.M = (long)N * Ho * Wo; p.w = L.w; p.bias = L.b; p.Cout_pad = L.Cout_pad; p.out = out; p.Cout_store = L.Cout_pad; p.Cout_valid = L.Cout_pad;
  p.out_scale = 1.f; p.ablate = ablate & 255; p.acc_scale = split ? ldexpf(1.0f, -kSplitWeightExp) : 1.f;
  p.out_f32 = of32; p.epi_mode = conv_epi_mode(); p.xtile = conv_xtile_enabled() ? 1 : 0;
  if (resf) { p.res = resb; p.res_f32 = 1; p.res_C = L.Cout_pad; }
  if (statf) p.stats = (float*)statb;
  void* wdm = nullptr;
  const bool bench_dma_off = opt("conv_dma") == 0;
  if (!bench_dma_off && ntaps == 9 && stride == 1 && L.Cout_pad >= 128) {
    const size_t nb = wbytes * (split ? 2 : 1);
    if (dev_malloc(&wdm, nb)) return -2.f;
    SDM_LAUNCH(fill_random_f16_kernel, dim3(2048), dim3(256), 0, e->stream, (half_t*)wdm, (long)(nb / 2), 29u, 0.05f);
    p.w_dma = (const half_t*)wdm;
  }
  if (split) p.w_lo = (const half_t*)wl;
  p.pc = (split && p.w_dma && pcf) ? 1 : 0;
  if (split && p.w_dma && f8f && L.Cin_pad % 32 == 0) { p.f8 = 1; p.f8_sa = 127 - 11; p.f8_sb = 127 - kSplitWeightExp; p.acc_scale = 1.f; }
  if (split && ntaps == 1 && f8f && L.Cin_pad % 32 == 0 && L.Cout_pad >= 128) {      // 1x1 GEMM on the fp8-residual kernel (tile cfg 4)
    if (dev_malloc(&wdm, wbytes * 2)) return -2.f;
    SDM_LAUNCH(fill_random_f16_kernel, dim3(2048), dim3(256), 0, e->stream, (half_t*)wdm, (long)wbytes, 29u, 0.05f);
    p.w_dma = (const half_t*)wdm; p.f8 = 1; p.f8_sa = 127 - 11; p.f8_sb = 127 - kSplitWeightExp; p.acc_scale = 1.f;
  }
  if (gnf) { p.gn_scale = (const float*)gnt; p.gn_shift = (const float*)gnt + (size_t)N * L.Cin_pad; p.gn_silu = 1; }
  int cfg = tile_cfg >= 0 ? tile_cfg : conv_pick_cfg(ntaps, stride, p);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  // split-K as the engine would run it (option conv_splitk: -1 by shape, n forced); the partial sums + the reduce kernel are inside the timed loop
  int ksplit = 1;
  void* wsb = nullptr;
  if (!(p.w_dma && ((ntaps == 9 && stride == 1 && cfg == 0) || (ntaps == 1 && cfg == 4)))) ksplit = conv_pick_ksplit(ntaps, stride, cfg, p);      // register-staged kernels only
  if (ksplit > 1 && dev_malloc(&wsb, (size_t)ksplit * p.M * p.Cout_pad * 4)) return -2.f;
  if (ksplit > 1) fprintf(stderr, "[bench_conv] split-K %d\n", ksplit);
  auto run = [&](const ConvParams& pp) { if (ksplit > 1) launch_conv_splitk(ntaps, stride, cfg, pp, ksplit, (float*)wsb, e->stream); else launch_conv(ntaps, stride, cfg, pp, e->stream); };
  run(p);
  if (ablate & 256) {      // one traced launch of the F8 3x3 kernel (-DSDM_CONV_TRACE builds): the LDS-parked shader-clock stamps of the first 16 blocks -> stderr
    void* tr = nullptr;
    const size_t tb = (size_t)16 * 2 * 384 * 4;
    if (dev_malloc(&tr, tb) == 0) {
      dev_memset(tr, 0, tb, e->stream);
      ConvParams pt = p; pt.trace = (unsigned int*)tr; pt.ablate = ablate & 255; pt.trace_skip = (ablate >> 9) & 7; pt.trace_b0 = ((ablate >> 12) & 0xFF) * 256;
      launch_conv(ntaps, stride, cfg, pt, e->stream);
      std::vector<unsigned int> h(tb / 4);
      (void)dev_memcpy_d2h(h.data(), tr, tb, e->stream);
      (void)dev_sync(e->stream);
      for (int b = 0; b < 16; ++b)
        for (int r = 0; r < 2; ++r) {
          const unsigned int* ev = &h[((size_t)b * 2 + r) * 384];
          const int n = (int)ev[383] < 383 ? (int)ev[383] : 383;
          if (!n) continue;
          fprintf(stderr, "[trace] block %d %s n=%d:", pt.trace_b0 + b, r ? "producer" : "consumer", n);
          for (int i = 0; i < n; ++i) fprintf(stderr, " %u", ev[i]);
          fprintf(stderr, "\n");
        }
      dev_free(tr);
    }
    p.ablate = ablate & 255;
  }
  (void)hipEventRecord(e0, (hipStream_t)e->stream);
  for (int i = 0; i < iters; ++i) run(p);
  (void)hipEventRecord(e1, (hipStream_t)e->stream);
  (void)hipStreamSynchronize((hipStream_t)e->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  dev_free(wp); dev_free(bp); dev_free(in); dev_free(out);
  if (wl) dev_free(wl);
  if (gnt) dev_free(gnt);
  if (wdm) dev_free(wdm);
  if (resb) dev_free(resb);
  if (statb) dev_free(statb);
  if (wsb) dev_free(wsb);
  return ms / (float)iters;
#endif
}

/* Bench/ablation helper for the d=64 attention kernel (not used by the engine). */
float sdm_bench_attn(sdm_ctx* e, int B, int heads, int Lq, int Lk, int qt, int ablate, int iters) {
  if (e) dev_use(e->device);
  if (!e) return -1.f;
#ifdef SDM_EMU
  return -1.f;
#else
  if (qt & 64) {      // bit 64: the d = 512 single-head kernel (VAE mid-block), ablate = its compile-time ABL mask
    const int ldvt5 = rup(Lk, 64);
    void *q5 = nullptr, *k5 = nullptr, *v5 = nullptr, *o5 = nullptr;
    if (dev_malloc(&q5, (size_t)B * Lq * 512 * 2) || dev_malloc(&k5, (size_t)B * Lk * 512 * 2) || dev_malloc(&v5, (size_t)B * 512 * ldvt5 * 2) || dev_malloc(&o5, (size_t)B * Lq * 512 * 4)) return -2.f;
    SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)q5, (long)B * Lq * 512, 3u, 0.3f);
    SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)k5, (long)B * Lk * 512, 7u, 0.3f);
    SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)v5, (long)B * 512 * ldvt5, 11u, 1.0f);
    AttnParams p5;
    memset(&p5, 0, sizeof(p5));
    p5.q = (const half_t*)q5; p5.q_bs = (long)Lq * 512; p5.ldq = 512; p5.k = (const half_t*)k5; p5.k_bs = (long)Lk * 512; p5.ldk = 512;
    p5.vt = (const half_t*)v5; p5.vt_hs = (long)512 * ldvt5; p5.vt_bs = p5.vt_hs; p5.ldvt = ldvt5; p5.o = (half_t*)o5; p5.o_bs = (long)Lq * 512; p5.ldo = 512; p5.o_f32 = 1;
    p5.Lq = Lq; p5.Lk = Lk; p5.scale_log2e = 0.0441941738f * SDM_LOG2E;
    p5.batch = B; p5.heads = 1; p5.nq_blocks = sdm_cdiv(Lq, 128); p5.q_chunks = 8;
    const unsigned nb5 = (unsigned)(B * p5.q_chunks * sdm_cdiv(p5.nq_blocks, p5.q_chunks));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i <= iters; ++i) {
      if (i == 1) (void)hipEventRecord(e0, (hipStream_t)e->stream);
#define SDM_D512_ABL(A) case A: { auto kp = attn_d512_kernel<A>; SDM_SET_SMEM(kp, ATTN512P_SMEM); SDM_LAUNCH(kp, dim3(nb5), dim3(512), ATTN512P_SMEM, e->stream, p5); } break;
      switch (ablate) { SDM_D512_ABL(0) SDM_D512_ABL(1) SDM_D512_ABL(6) SDM_D512_ABL(7) SDM_D512_ABL(8) SDM_D512_ABL(32) SDM_D512_ABL(40) SDM_D512_ABL(41) default: break; }
#undef SDM_D512_ABL
    }
    (void)hipEventRecord(e1, (hipStream_t)e->stream);
    (void)hipStreamSynchronize((hipStream_t)e->stream);
    float ms5 = 0.f;
    (void)hipEventElapsedTime(&ms5, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    dev_free(q5); dev_free(k5); dev_free(v5); dev_free(o5);
    return ms5 / (float)iters;
  }
  const int C = heads * 64, ldvt = rup(Lk, 64);
  const int prec = (qt & 2) ? 1 : 0, nw8 = (qt & 4) ? 1 : 0;          // qt bits: 2 = split-precision variant (hi | lo planes, fp32 output), 4 = 8-wave blocks
  void *q = nullptr, *k = nullptr, *vt = nullptr, *o = nullptr;
  if (dev_malloc(&q, (size_t)B * Lq * C * 2 * (1 + prec)) || dev_malloc(&k, (size_t)B * Lk * C * 2 * (1 + prec)) ||
      dev_malloc(&vt, (size_t)B * heads * 64 * ldvt * 2 * (1 + prec)) || dev_malloc(&o, (size_t)B * Lq * C * (prec ? 4 : 2) + 4096)) return -2.f;
  if (prec) {
    SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)q + (size_t)B * Lq * C, (long)B * Lq * C, 13u, 0.0003f);
    SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)k + (size_t)B * Lk * C, (long)B * Lk * C, 17u, 0.0003f);
    SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)vt + (size_t)B * heads * 64 * ldvt, (long)B * heads * 64 * ldvt, 19u, 0.0003f);
  }
  if (prec && (qt & 16)) {      // pair planes: random fp16 bit patterns would hold e5m2 NaNs; zero residual pairs time the same instructions
    (void)hipMemsetAsync((half_t*)q + (size_t)B * Lq * C, 0, (size_t)B * Lq * C * 2, (hipStream_t)e->stream);
    (void)hipMemsetAsync((half_t*)k + (size_t)B * Lk * C, 0, (size_t)B * Lk * C * 2, (hipStream_t)e->stream);
  }
  SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)q, (long)B * Lq * C, 3u, 1.0f);
  SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)k, (long)B * Lk * C, 7u, 1.0f);
  SDM_LAUNCH(fill_random_f16_kernel, dim3(4096), dim3(256), 0, e->stream, (half_t*)vt, (long)B * heads * 64 * ldvt, 11u, 1.0f);
  AttnParams p;
  memset(&p, 0, sizeof(p));
  p.q = (const half_t*)q; p.q_bs = (long)Lq * C; p.ldq = C; p.k = (const half_t*)k; p.k_bs = (long)Lk * C; p.ldk = C;
  p.vt = (const half_t*)vt; p.vt_hs = (long)64 * ldvt; p.vt_bs = heads * p.vt_hs; p.ldvt = ldvt; p.o = (half_t*)o; p.o_bs = (long)Lq * C; p.ldo = C;
  p.Lq = Lq; p.Lk = Lk; p.scale_log2e = 0.125f * SDM_LOG2E; p.ablate = ablate;
  if (prec) { p.q_lo = (long)B * Lq * C; p.k_lo = (long)B * Lk * C; p.vt_lo = (long)B * p.vt_bs; p.o_f32 = 1; }
  p.batch = B; p.heads = heads; p.nq_blocks = sdm_cdiv(Lq, nw8 ? 256 : 128); p.q_chunks = 8;
  const unsigned nblk = (unsigned)(B * heads * p.q_chunks * sdm_cdiv(p.nq_blocks, p.q_chunks));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i <= iters; ++i) {
    if (i == 1) (void)hipEventRecord(e0, (hipStream_t)e->stream);
    if (qt & 16) {      // bit 16: the ping-pong kernel (pair planes as the engine's self- and cross-attentions), ablate = its compile-time ABL mask
      p.pp_flags = (qt & 32) ? 0 : 1;
      p.part_ml = (float*)((unsigned char*)o + (size_t)B * Lq * C * 4);      // (ablate 64: the segment trace lands behind the output)
      if (qt & 128) p.pp_flags |= 2;
#define SDM_PP_ABL(A) case A: { auto kp = attn_d64_pp_kernel<A, 0, 0>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk), dim3(512), ATTN64PP_SMEM + 4096, e->stream, p); } break;
#define SDM_PP_DS(A, S, V) case V: { auto kp = attn_d64_pp_kernel<A, 0, 0, 0, S, 0>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk), dim3(512), ATTN64PP_SMEM + 4096, e->stream, p); } break;
#define SDM_PP_KE(K, V) case V: { auto kp = attn_d64_pp_kernel<0, 0, 0, K, 1, 0>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk), dim3(512), ATTN64PP_SMEM, e->stream, p); } break;
      switch (ablate) { SDM_PP_ABL(0) SDM_PP_ABL(1) SDM_PP_ABL(6) SDM_PP_ABL(7) SDM_PP_ABL(8) SDM_PP_ABL(32) SDM_PP_ABL(56) SDM_PP_ABL(63)
                        SDM_PP_KE(-1, 100) SDM_PP_KE(2, 101) SDM_PP_KE(1, 102) SDM_PP_ABL(64)
                        SDM_PP_DS(0, 0, 110) SDM_PP_DS(0, 1, 111) SDM_PP_DS(0, 2, 112) SDM_PP_DS(0, 3, 113) SDM_PP_DS(64, 0, 114) SDM_PP_DS(64, 2, 116) SDM_PP_DS(64, 3, 117) default: break; }      // 100-102: the other fragment / DMA placements (KE), no ablation
#undef SDM_PP_ABL
#undef SDM_PP_KE
#undef SDM_PP_DS
    }
    else if (prec && (qt & 8) && nw8) { auto kp = attn_d64_kernel<1, 2, 8>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk), dim3(512), ATTN64P_SMEM, e->stream, p); }      // bit 8: P.V on plain fp16
    else if (prec && (qt & 8)) { auto kp = attn_d64_kernel<1, 2, 4>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk), dim3(256), ATTN64P_SMEM, e->stream, p); }
    else if (prec && nw8) { auto kp = attn_d64_kernel<1, 1, 8>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk), dim3(512), ATTN64P_SMEM, e->stream, p); }
    else if (prec) { auto kp = attn_d64_kernel<1, 1, 4>; SDM_SET_SMEM(kp, 160 * 1024); SDM_LAUNCH(kp, dim3(nblk), dim3(256), ATTN64P_SMEM, e->stream, p); }
    else if (nw8) { auto kf = attn_d64_kernel<1, 0, 8>; SDM_SET_SMEM(kf, 160 * 1024); SDM_LAUNCH(kf, dim3(nblk), dim3(512), ATTN64P_SMEM, e->stream, p); }
    else { SDM_LAUNCH((attn_d64_kernel<1, 0, 4>), dim3(nblk), dim3(256), ATTN64_SMEM, e->stream, p); }
  }
  (void)hipEventRecord(e1, (hipStream_t)e->stream);
  (void)hipStreamSynchronize((hipStream_t)e->stream);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if ((qt & 16) && (ablate == 64 || (ablate >= 114 && ablate <= 117))) {      // segment stamps of waves 0 (half A) and 4 (half B) of block 0: averages over tiles 4 .. 51 of the last launch
    std::vector<unsigned long long> tr(2 * 8 * 28);
    (void)hipMemcpy(tr.data(), (unsigned char*)o + (size_t)B * Lq * C * 4, tr.size() * 8, hipMemcpyDeviceToHost);
    for (int g = 0; g < 2; ++g) {
      double dm = 0, sm = 0, vr = 0, wa = 0, mx = 0, wb = 0; int n = 0;
      for (int t = 5; t < 27; ++t) {
        const unsigned long long* c = &tr[(size_t)g * 224 + (size_t)t * 8];
        const unsigned long long prev3 = tr[(size_t)g * 224 + (size_t)(t - 1) * 8 + 3];
        dm += (double)(c[4] - prev3); sm += (double)(c[5] - c[4]); vr += (double)(c[0] - c[5]); wa += (double)(c[1] - c[0]); mx += (double)(c[2] - c[1]); wb += (double)(c[3] - c[2]); ++n;
      }
      fprintf(stderr, "[attn_pp trace] wave %d: DMA issue %.0f | softmax VALU %.0f | V^T reads + lgkmcnt(0) %.0f | wait at barrier %.0f | matrix segment %.0f | wait at barrier %.0f  (cycles per tile, mean of %d tiles; each stamp costs an s_memtime round trip)\n",
              g * 4, dm / n, sm / n, vr / n, wa / n, mx / n, wb / n, n);
    }
  }
  dev_free(q); dev_free(k); dev_free(vt); dev_free(o);
  return ms / (float)iters;
#endif
}

int sdm_op_groupnorm(sdm_ctx* e, const void* in0, const void* in1, int C0, int C1, int in_f32, int N, int HW, int groups, const float* gamma,
                     const float* beta, float eps, int silu, void* out) {
  if (e) dev_use(e->device);
  if (!e || !in0 || !out) return SDM_ERR_INVALID;
  return run_two_pass(e, [&]() { return op_groupnorm_raw(e, in0, in1, C0, C1, in_f32, N, HW, groups, gamma, beta, eps, silu, out, 0); });
}

int sdm_op_layernorm(sdm_ctx* e, const void* x, int in_f32, long rows, int C, const float* gamma, const float* beta, float eps, void* out) {
  if (e) dev_use(e->device);
  if (!e || !x || !out) return SDM_ERR_INVALID;
  if (C % 64 || C > 64 * SDM_LN_MAXV) SDM_FAIL(e, SDM_ERR_INVALID, "layernorm: unsupported C %d", C);
  SDM_LAUNCH(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, e->stream, x, in_f32, gamma, beta, out, 0, rows, C, eps);
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return 0;
}

int sdm_op_attention(sdm_ctx* e, const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const float* bias, int B, int heads,
                     int Lq, int Lk, int D, void* out, int ldo) {
  if (e) dev_use(e->device);
  if (!e || !q || !k || !v || !out) return SDM_ERR_INVALID;
  return run_two_pass(e, [&]() {
    T b2 = talloc(e, B, 1, 1, Lk, 1);
    const float* bl2 = nullptr;
    if (bias) {
      bl2 = (const float*)b2.p;
      if (!e->dry) {
        // natural-log bias (reference domain) -> log2 domain used by the kernel
        SDM_LAUNCH(scale_copy_kernel, dim3(sdm_cdiv(B * Lk, 256)), dim3(256), 0, e->stream, bias, (float*)b2.p, (long)B * Lk, SDM_LOG2E);
      }
    }
    int rc = op_attention_raw(e, (const half_t*)q, ldq, (const half_t*)k, ldk, (const half_t*)v, ldv, bias ? bl2 : nullptr,
                              B, heads, Lq, Lk, D, (half_t*)out, ldo);
    tfree(e, b2);
    return rc;
  });
}

/* Split-precision d = 64 attention cores as the default precision runs them.  q [B,Lq,heads*64], k / v [B,Lk,heads*64]: contiguous fp32
 * DEVICE tensors.  They are first turned into the operand planes the producing GEMMs write in the engine (split_planes_kernel: fp16 hi plane
 * + fp16 lo plane, or + e5m2 pair plane when the Q.K^T residual terms run on fp8 MFMAs - the default; the option attn_f8 = 0 selects the former),
 * with the logit scale d^-1/2 * log2(e) applied to Q as the engine's to_q weights do; fp32 output [B,Lq,heads*64].  Test hook. */
int sdm_op_attention_split(sdm_ctx* e, const float* q, const float* k, const float* v, const float* bias, int B, int heads, int Lq, int Lk, float* out) {
  if (e) dev_use(e->device);
  if (!e || !q || !k || !v || !out) return SDM_ERR_INVALID;
  const int C = heads * 64;
  const int mode = (attn_f8_enabled() && !opt("attn_pv_split")) ? 3 : 2;
  return run_two_pass(e, [&]() {
    T b2 = talloc(e, B, 1, 1, Lk, 1);
    T qp = talloc(e, B, 1, Lq, C, mode), kp = talloc(e, B, 1, Lk, C, mode), vp = talloc(e, B, 1, Lk, C, mode);
    const float* bl2 = nullptr;
    const long nq = (long)B * Lq * C, nk = (long)B * Lk * C;
    if (!e->dry) {
      if (bias) {
        bl2 = (const float*)b2.p;
        SDM_LAUNCH(scale_copy_kernel, dim3(sdm_cdiv(B * Lk, 256)), dim3(256), 0, e->stream, bias, (float*)b2.p, (long)B * Lk, SDM_LOG2E);
      }
      SDM_LAUNCH(split_planes_kernel, dim3((unsigned)((nq / 4 + 255) / 256)), dim3(256), 0, e->stream, q, (half_t*)qp.p, (half_t*)qp.p + nq, nq, 0.125f * SDM_LOG2E, mode);
      SDM_LAUNCH(split_planes_kernel, dim3((unsigned)((nk / 4 + 255) / 256)), dim3(256), 0, e->stream, k, (half_t*)kp.p, (half_t*)kp.p + nk, nk, 1.0f, mode);
      SDM_LAUNCH(split_planes_kernel, dim3((unsigned)((nk / 4 + 255) / 256)), dim3(256), 0, e->stream, v, (half_t*)vp.p, (half_t*)vp.p + nk, nk, 1.0f, 2);
    }
    AttnPrec ap; ap.prec = mode - 1; ap.q_lo = nq; ap.k_lo = nk; ap.v_lo = nk; ap.out_f32 = 1;
    int rc = op_attention_raw(e, (const half_t*)qp.p, C, (const half_t*)kp.p, C, (const half_t*)vp.p, C, bias ? bl2 : nullptr, B, heads, Lq, Lk, 64,
                              out, C, true, nullptr, ap);
    tfree(e, vp); tfree(e, kp); tfree(e, qp); tfree(e, b2);
    return rc;
  });
}

int sdm_op_resize_aa(sdm_ctx* e, const float* in, int P, int Hin, int Win, float* out, int Hout, int Wout) {
  if (e) dev_use(e->device);
  if (!e || !in || !out) return SDM_ERR_INVALID;
  SDM_LAUNCH(resize_planes_kernel, dim3((unsigned)(((long)P * Hout * Wout + 255) / 256)), dim3(256), 0, e->stream, in, out, P, Hin, Win, Hout, Wout, 0);
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return 0;
}

int sdm_op_mask_bias(sdm_ctx* e, const float* plane, int B, int S, int level, float* out) {
  if (e) dev_use(e->device);
  if (!e || !plane || !out) return SDM_ERR_INVALID;
  const int lk = (S / 8) >> level;
  SDM_LAUNCH(mask_bias_kernel, dim3(sdm_cdiv(B * lk * lk, 256)), dim3(256), 0, e->stream, plane, out, B, S, S, level, e->cfg.attn_mask_value, 1.0f);
  SDM_CHECK_DEV(e, dev_sync(e->stream));
  return 0;
}

}  // extern "C"
