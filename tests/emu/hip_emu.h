// hip_emu.h - TEST INFRASTRUCTURE ONLY.  A minimal CPU emulation of the HIP execution model so that
// the *same kernel sources* (comfyui-sdmatte_amd/csrc/*.h) can be executed in the GPU-less build
// container to debug index arithmetic, LDS layouts, barrier placement and MFMA fragment usage.
//
// It is NOT a fallback: the product library (libsdmatte_hip.so) is compiled by hipcc for gfx950
// only, never links this file, and refuses to load without a GPU.  The emulated library
// (tests/emu/_build/libsdmatte_emu.so) is built and loaded exclusively by tests/test_emu_*.py.
//
// Model: one OS thread runs one workgroup at a time; each work-item is a fiber (hand-rolled x86-64
// context switch).  __syncthreads() and wave-level operations (shuffles, MFMA) are rendezvous
// points.  __shared__ is `static thread_local`.  MFMA follows the gfx950 fragment layouts of
// /opt/skills/guides/cdna_hip_programming.md section 3.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <atomic>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct emu_dim3 { unsigned x, y, z; emu_dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef emu_dim3 dim3;

using std::min; using std::max;
namespace emu {
struct Fiber { void* sp; char* stack; bool done; };
struct WaveState { int arrived; unsigned gen; uint32_t slot[64][16]; };
struct BlockState {
  std::vector<Fiber> fibers; std::vector<WaveState> waves;
  int nthreads; int cur; int arrived; unsigned gen; int live;
  void* main_sp; const std::function<void()>* body;
};
extern thread_local BlockState* g_blk;
extern thread_local emu_dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;
extern thread_local unsigned char* g_dyn_smem;
void yield();
void block_barrier();
void wave_barrier();
int lane_id();
WaveState& wave();
void launch(dim3 grid, dim3 block, size_t smem, const std::function<void()>& body);
extern std::mutex g_atomic_mu;
}  // namespace emu

#define threadIdx (emu::g_threadIdx)
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

static inline void __syncthreads() { emu::block_barrier(); }

template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) {
  static_assert(sizeof(T) == 4, "emu shfl: 4-byte types only");
  (void)width;
  auto& w = emu::wave(); int l = emu::lane_id();
  memcpy(&w.slot[l][0], &v, 4); emu::wave_barrier();
  T r; memcpy(&r, &w.slot[l ^ mask][0], 4); emu::wave_barrier();
  return r;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  static_assert(sizeof(T) == 4, "emu shfl: 4-byte types only");
  (void)width;
  auto& w = emu::wave(); int l = emu::lane_id();
  memcpy(&w.slot[l][0], &v, 4); emu::wave_barrier();
  T r; memcpy(&r, &w.slot[src & 63][0], 4); emu::wave_barrier();
  return r;
}

// wave vote: true if any lane's predicate is true
static inline int __any(int pred) {
  auto& w = emu::wave(); int l = emu::lane_id();
  w.slot[l][0] = pred ? 1u : 0u; emu::wave_barrier();
  int r = 0; for (int i = 0; i < 64; ++i) r |= (int)w.slot[i][0];
  emu::wave_barrier();
  return r;
}

// ---- atomics (global memory may be touched by several OS threads = several workgroups) ----
static inline float atomicAdd(float* p, float v) { std::lock_guard<std::mutex> g(emu::g_atomic_mu); float o = *p; *p = o + v; return o; }
static inline double atomicAdd(double* p, double v) { std::lock_guard<std::mutex> g(emu::g_atomic_mu); double o = *p; *p = o + v; return o; }
static inline unsigned int atomicMax(unsigned int* p, unsigned int v) { std::lock_guard<std::mutex> g(emu::g_atomic_mu); unsigned int o = *p; if (v > o) *p = v; return o; }
static inline int atomicMin(int* p, int v) { std::lock_guard<std::mutex> g(emu::g_atomic_mu); int o = *p; if (v < o) *p = v; return o; }
static inline unsigned int atomicCAS(unsigned int* p, unsigned int cmp, unsigned int v) { std::lock_guard<std::mutex> g(emu::g_atomic_mu); unsigned int o = *p; if (o == cmp) *p = v; return o; }
static inline int atomicAdd(int* p, int v) { std::lock_guard<std::mutex> g(emu::g_atomic_mu); int o = *p; *p = o + v; return o; }

// ---- MFMA 32x32x16 f16 (gfx950): A[i][k]: lane l holds i=l&31, k=8*(l>>5)+j; B[k][n]: n=l&31, same k;
//      D[row][col]: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5), r in [0,16). ----
typedef _Float16 emu_h8 __attribute__((ext_vector_type(8)));
typedef float emu_f16v __attribute__((ext_vector_type(16)));
static inline emu_f16v emu_mfma_f32_32x32x16_f16(emu_h8 a, emu_h8 b, emu_f16v c) {
  auto& w = emu::wave(); int l = emu::lane_id();
  memcpy(&w.slot[l][0], &a, 16); memcpy(&w.slot[l][4], &b, 16); emu::wave_barrier();
  int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    float acc = c[r];
    for (int kh = 0; kh < 2; ++kh) {
      emu_h8 av, bv;
      memcpy(&av, &w.slot[row + 32 * kh][0], 16);
      memcpy(&bv, &w.slot[col + 32 * kh][4], 16);
      for (int j = 0; j < 8; ++j) acc += (float)av[j] * (float)bv[j];
    }
    c[r] = acc;
  }
  emu::wave_barrier();
  return c;
}

// ---- OCP fp8 e4m3 (gfx950): conversion (round to nearest even, NaN beyond +-448 like v_cvt_pk_fp8_f32) and the K = 64 scaled
//      MFMA (lane l: row / column l & 31, K values 32*(l>>5) .. +31 as 32 consecutive bytes; product scaled by
//      2^(sa-127) * 2^(sb-127)); semantics pinned on hardware by tools/probe/f8_semantics_probe.hip ----
static inline float emu_e4m3_to_f32(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 0) r = ldexpf((float)m, -9);
  else if (e == 15 && m == 7) r = NAN;
  else r = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -r : r;
}
static inline unsigned char emu_f32_to_e4m3(float x) {
  const unsigned char sign = std::signbit(x) ? 0x80 : 0;
  float a = fabsf(x);
  if (!(a == a)) return sign | 0x7f;
  if (a > 464.0f) return sign | 0x7f;                      // beyond the rounding range of 448: NaN (callers clamp)
  if (a >= 448.0f) return sign | 0x7e;
  if (a < ldexpf(1.0f, -10)) return sign;                  // below half of the smallest subnormal (ties to even -> 0)
  int e; frexpf(a, &e); e -= 1;                            // a = 1.xxx * 2^e
  if (e < -6) e = -6;                                      // subnormal range: fixed step 2^-9
  const float step = ldexpf(1.0f, e - 3);
  float q = nearbyintf(a / step);                          // default rounding mode: nearest even
  float v = q * step;
  int ee; frexpf(v, &ee); ee -= 1;
  if (v < ldexpf(1.0f, -6)) return sign | (unsigned char)q;                       // subnormal: mantissa = q
  const int m = (int)nearbyintf((v / ldexpf(1.0f, ee) - 1.0f) * 8.0f);
  return sign | (unsigned char)(((ee + 7) << 3) | m);
}
static inline int emu_cvt_pk_fp8_f32(float a, float b, int old, bool hi_word) {
  const unsigned int pk = (unsigned int)emu_f32_to_e4m3(a) | ((unsigned int)emu_f32_to_e4m3(b) << 8);
  const unsigned int o = (unsigned int)old;
  return (int)(hi_word ? ((o & 0x0000ffffu) | (pk << 16)) : ((o & 0xffff0000u) | pk));
}
// OCP e5m2 ("bf8"): 1-5-2, bias 15, inf / NaN as in fp16 (it is the top byte of an fp16); round to nearest even
static inline float emu_e5m2_to_f32(unsigned char v) {
  const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
  float r;
  if (e == 0) r = ldexpf((float)m, -16);
  else if (e == 31) r = m ? NAN : INFINITY;
  else r = ldexpf(1.0f + m / 4.0f, e - 15);
  return s ? -r : r;
}
static inline unsigned char emu_f32_to_e5m2(float x) {
  const unsigned char sign = std::signbit(x) ? 0x80 : 0;
  float a = fabsf(x);
  if (!(a == a)) return sign | 0x7f;
  if (a >= 61440.0f) return sign | 0x7c;                   // rounds beyond the largest finite value 57344 (callers clamp): inf
  if (a < ldexpf(1.0f, -17)) return sign;                  // below half of the smallest subnormal 2^-16 (tie -> even = 0)
  int e; frexpf(a, &e); e -= 1;                            // a = 1.xxx * 2^e
  if (e < -14) e = -14;                                    // subnormal range: fixed step 2^-16
  const float step = ldexpf(1.0f, e - 2);
  const float v = nearbyintf(a / step) * step;             // default rounding mode: nearest even
  if (v < ldexpf(1.0f, -14)) return sign | (unsigned char)nearbyintf(v / ldexpf(1.0f, -16));
  int ee; frexpf(v, &ee); ee -= 1;
  const int m = (int)nearbyintf((v / ldexpf(1.0f, ee) - 1.0f) * 4.0f);
  return sign | (unsigned char)(((ee + 15) << 2) | m);
}
static inline int emu_cvt_pk_bf8_f32(float a, float b, int old, bool hi_word) {
  const unsigned int pk = (unsigned int)emu_f32_to_e5m2(a) | ((unsigned int)emu_f32_to_e5m2(b) << 8);
  const unsigned int o = (unsigned int)old;
  return (int)(hi_word ? ((o & 0x0000ffffu) | (pk << 16)) : ((o & 0xffff0000u) | pk));
}
typedef int emu_i8v __attribute__((ext_vector_type(8)));
static inline emu_f16v emu_mfma_scale_f32_32x32x64_fp8(emu_i8v a, emu_i8v b, emu_f16v c, int sa, int sb) {
  auto& w = emu::wave(); int l = emu::lane_id();
  memcpy(&w.slot[l][0], &a, 32); memcpy(&w.slot[l][8], &b, 32); emu::wave_barrier();
  const float scale = ldexpf(1.0f, (sa & 255) - 127 + (sb & 255) - 127);
  int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    double acc = 0.0;
    for (int kh = 0; kh < 2; ++kh) {
      const unsigned char* av = (const unsigned char*)&w.slot[row + 32 * kh][0];
      const unsigned char* bv = (const unsigned char*)&w.slot[col + 32 * kh][8];
      for (int j = 0; j < 32; ++j) acc += (double)emu_e4m3_to_f32(av[j]) * (double)emu_e4m3_to_f32(bv[j]);
    }
    c[r] = c[r] + (float)acc * scale;
  }
  emu::wave_barrier();
  return c;
}

// the same MFMA with the A operand in e5m2 (cbsz = 1) and B in e4m3
static inline emu_f16v emu_mfma_scale_f32_32x32x64_bf8_fp8(emu_i8v a, emu_i8v b, emu_f16v c, int sa, int sb) {
  auto& w = emu::wave(); int l = emu::lane_id();
  memcpy(&w.slot[l][0], &a, 32); memcpy(&w.slot[l][8], &b, 32); emu::wave_barrier();
  const float scale = ldexpf(1.0f, (sa & 255) - 127 + (sb & 255) - 127);
  int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    double acc = 0.0;
    for (int kh = 0; kh < 2; ++kh) {
      const unsigned char* av = (const unsigned char*)&w.slot[row + 32 * kh][0];
      const unsigned char* bv = (const unsigned char*)&w.slot[col + 32 * kh][8];
      for (int j = 0; j < 32; ++j) acc += (double)emu_e5m2_to_f32(av[j]) * (double)emu_e4m3_to_f32(bv[j]);
    }
    c[r] = c[r] + (float)acc * scale;
  }
  emu::wave_barrier();
  return c;
}

// ... and with both operands in e5m2 (cbsz = blgp = 1)
static inline emu_f16v emu_mfma_scale_f32_32x32x64_bf8_bf8(emu_i8v a, emu_i8v b, emu_f16v c, int sa, int sb) {
  auto& w = emu::wave(); int l = emu::lane_id();
  memcpy(&w.slot[l][0], &a, 32); memcpy(&w.slot[l][8], &b, 32); emu::wave_barrier();
  const float scale = ldexpf(1.0f, (sa & 255) - 127 + (sb & 255) - 127);
  int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    double acc = 0.0;
    for (int kh = 0; kh < 2; ++kh) {
      const unsigned char* av = (const unsigned char*)&w.slot[row + 32 * kh][0];
      const unsigned char* bv = (const unsigned char*)&w.slot[col + 32 * kh][8];
      for (int j = 0; j < 32; ++j) acc += (double)emu_e5m2_to_f32(av[j]) * (double)emu_e5m2_to_f32(bv[j]);
    }
    c[r] = c[r] + (float)acc * scale;
  }
  emu::wave_barrier();
  return c;
}

// A in e4m3, B in e5m2 (cbsz = 0, blgp = 1)
static inline emu_f16v emu_mfma_scale_f32_32x32x64_fp8_bf8(emu_i8v a, emu_i8v b, emu_f16v c, int sa, int sb) {
  auto& w = emu::wave(); int l = emu::lane_id();
  memcpy(&w.slot[l][0], &a, 32); memcpy(&w.slot[l][8], &b, 32); emu::wave_barrier();
  const float scale = ldexpf(1.0f, (sa & 255) - 127 + (sb & 255) - 127);
  int col = l & 31, hi = l >> 5;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
    double acc = 0.0;
    for (int kh = 0; kh < 2; ++kh) {
      const unsigned char* av = (const unsigned char*)&w.slot[row + 32 * kh][0];
      const unsigned char* bv = (const unsigned char*)&w.slot[col + 32 * kh][8];
      for (int j = 0; j < 32; ++j) acc += (double)emu_e4m3_to_f32(av[j]) * (double)emu_e5m2_to_f32(bv[j]);
    }
    c[r] = c[r] + (float)acc * scale;
  }
  emu::wave_barrier();
  return c;
}

// ---- host runtime stand-ins used by the engine ----
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
