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

// ---- host runtime stand-ins used by the engine ----
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
