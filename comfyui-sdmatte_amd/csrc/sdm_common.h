// sdm_common.h - shared device/host helpers for the SDMatte gfx950 engine.
//
// The product build is `hipcc --offload-arch=gfx950` only.  When SDM_EMU is defined (tests/emu
// only) the same sources compile for the host against tests/emu/hip_emu.h so that index math can
// be debugged without a GPU; that build is never shipped or loaded by the package.
#pragma once
#include <stdint.h>
#include <stddef.h>
#ifdef SDM_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#endif

typedef _Float16 half_t;
typedef half_t f16x8 __attribute__((ext_vector_type(8)));
typedef half_t f16x4 __attribute__((ext_vector_type(4)));
typedef half_t f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifdef SDM_EMU
#define SDM_DYN_SMEM(name) unsigned char* name = emu::g_dyn_smem
#define SDM_SHARED static thread_local
#define SDM_MFMA_32x32x16_F16(a, b, c) emu_mfma_f32_32x32x16_f16((a), (b), (c))
#define SDM_MFMA_32x32x64_F8(a, b, c, sa, sb) emu_mfma_scale_f32_32x32x64_fp8((a), (b), (c), (sa), (sb))
#define SDM_CVT_PK_FP8(a, b, old, hi_word) emu_cvt_pk_fp8_f32((a), (b), (old), (hi_word))
#define SDM_MFMA_32x32x64_BF8A_F8B(a, b, c, sa, sb) emu_mfma_scale_f32_32x32x64_bf8_fp8((a), (b), (c), (sa), (sb))
#define SDM_CVT_PK_BF8(a, b, old, hi_word) emu_cvt_pk_bf8_f32((a), (b), (old), (hi_word))
#define SDM_MFMA_32x32x64_BF8_BF8(a, b, c, sa, sb) emu_mfma_scale_f32_32x32x64_bf8_bf8((a), (b), (c), (sa), (sb))
#define SDM_MFMA_32x32x64_F8A_BF8B(a, b, c, sa, sb) emu_mfma_scale_f32_32x32x64_fp8_bf8((a), (b), (c), (sa), (sb))
#define SDM_LAUNCH(kernel, grid, block, smem, stream, ...) \
  emu::launch((grid), (block), (smem), [=]() { kernel(__VA_ARGS__); })
#define SDM_DEV_INLINE static inline
#define SDM_HD_INLINE static inline
#define SDM_WAVE_SYNC() emu::wave_barrier()
#define SDM_SCHED_FENCE() ((void)0)
#define SDM_PIN_STORE_DATA(v) ((void)(v))
#define SDM_SCHED_GROUP(mask, n, id) ((void)0)
#define SDM_SETPRIO(n) ((void)0)
static inline float sdm_exp2(float x) { return exp2f(x); }
static inline float sdm_rcp(float x) { return 1.0f / x; }
#define SDM_MED3(x, lo, hi) fminf(fmaxf((x), (lo)), (hi))
#define SDM_UMUL24(a, b) ((unsigned int)(a) * (unsigned int)(b))
#else
#define SDM_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) unsigned char name[]
#define SDM_SHARED __shared__
#define SDM_MFMA_32x32x16_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
// v_mfma_scale_f32_32x32x64_f8f6f4 on OCP e4m3 operands: lane l holds row / column (l & 31) and the 32 K values 32*(l>>5) .. +31
// as 32 consecutive bytes; the product is multiplied by 2^(sa-127) * 2^(sb-127) (E8M0 exponents in byte 0 of two VGPRs; a
// run-time 0 means 2^-127, NOT "unscaled").  Verified on hardware by tools/probe/f8_semantics_probe.hip.
#define SDM_MFMA_32x32x64_F8(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 0, 0, 0, (sa), 0, (sb))
// v_cvt_pk_fp8_f32: two fp32 -> two e4m3 bytes (round to nearest even; |x| > 448 gives NaN, so callers clamp) into the low or
// high 16 bits of `old`
#define SDM_CVT_PK_FP8(a, b, old, hi_word) __builtin_amdgcn_cvt_pk_fp8_f32((a), (b), (old), (hi_word))
// the same instruction with the A operand in OCP e5m2 ("bf8": 2 mantissa bits, the range of fp16) and B in e4m3 (cbsz = 1, blgp = 0):
// activations are not bounded at pack time the way weights are, so their residual operands take the wide format
#define SDM_MFMA_32x32x64_BF8A_F8B(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 1, 0, 0, (sa), 0, (sb))
// v_cvt_pk_bf8_f32: two fp32 -> two e5m2 bytes (round to nearest even); callers clamp to +-57344 (the largest finite e5m2) first
#define SDM_CVT_PK_BF8(a, b, old, hi_word) __builtin_amdgcn_cvt_pk_bf8_f32((a), (b), (old), (hi_word))
// both operands in e5m2 (cbsz = 1, blgp = 1): the residual terms of Q.K^T (k_attn.h, PREC = 3)
#define SDM_MFMA_32x32x64_BF8_BF8(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 1, 1, 0, (sa), 0, (sb))
// A in e4m3 (weights), B in e5m2 (activations): the F8 conv / GEMM kernels, whose accumulators are [channel][pixel] (k_conv.h)
#define SDM_MFMA_32x32x64_F8A_BF8B(a, b, c, sa, sb) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4((a), (b), (c), 0, 1, 0, (sa), 0, (sb))
#define SDM_LAUNCH(kernel, grid, block, smem, stream, ...) \
  hipLaunchKernelGGL(kernel, (grid), (block), (smem), (hipStream_t)(stream), __VA_ARGS__)
#define SDM_DEV_INLINE __device__ __forceinline__
#define SDM_HD_INLINE __host__ __device__ __forceinline__      // layout arithmetic shared by kernels and their launchers
// LDS operations of ONE wave execute in issue order, so lanes of a wave may exchange data through a wave-private LDS
// region without s_barrier; only the compiler must not reorder the accesses.
#define SDM_WAVE_SYNC() __builtin_amdgcn_wave_barrier()
// pin the instruction schedule at this point (used to keep hand-pipelined LDS fragment reads ahead of the MFMAs)
#define SDM_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
// keeps the registers of `v` (the data of a 16-byte buffer store issued just before) live and unmodified up to this point, plus 4 idle
// cycles: gfx950 reads the data of a buffer_store_dwordx4 over several cycles after issue, and nothing in hipcc's hazard tables
// keeps a following VALU write away from them when the store carries an SGPR offset
#define SDM_PIN_STORE_DATA(v) asm volatile("s_nop 3" ::"v"(v))
// compile-time interleave request: the next `n` instructions of class `mask` (0x8 MFMA, 0x2 VALU, 0x400 TRANS, 0x100 DS read)
#define SDM_SCHED_GROUP(mask, n, id) __builtin_amdgcn_sched_group_barrier((mask), (n), (id))
// wave priority for the SIMD's issue arbitration (0 default .. 3): static, set once in front of a main loop
#define SDM_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
__device__ __forceinline__ float sdm_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float sdm_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
#define SDM_MED3(x, lo, hi) __builtin_amdgcn_fmed3f((x), (lo), (hi))      // clamp in one instruction (lo <= hi)
#define SDM_UMUL24(a, b) __umul24((a), (b))                                // both operands below 2^24: full-rate multiply (v_mul_u32_u24 / v_mad_u32_u24)
#endif

// ---- raw buffer loads: SGPR resource descriptor + 32-bit byte offset; out-of-range offsets return 0 in hardware
//      (free zero padding for conv halos / ragged tiles) and cost no 64-bit address VGPRs.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
#define SDM_BUF_INVALID 0x7FFFFFF0u
#ifdef SDM_EMU
struct sdm_rsrc { const unsigned char* base; unsigned int bytes; };
static inline sdm_rsrc sdm_make_rsrc(const void* p, unsigned int bytes) { sdm_rsrc r; r.base = (const unsigned char*)p; r.bytes = bytes; return r; }
static inline u32x4 sdm_buffer_load16(sdm_rsrc r, unsigned int voff, unsigned int soff) {
  u32x4 v = {0u, 0u, 0u, 0u};
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 16 <= r.bytes) memcpy(&v, r.base + o, 16);
  return v;
}
static inline u32x2 sdm_buffer_load8(sdm_rsrc r, unsigned int voff, unsigned int soff) {
  u32x2 v = {0u, 0u};
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 8 <= r.bytes) memcpy(&v, r.base + o, 8);
  return v;
}
static inline void sdm_buffer_store16(u32x4 v, sdm_rsrc r, unsigned int voff, unsigned int soff) {      // out-of-range stores are dropped, as in hardware
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 16 <= r.bytes) memcpy((unsigned char*)r.base + o, &v, 16);
}
#else
typedef __amdgpu_buffer_rsrc_t sdm_rsrc;
__device__ __forceinline__ sdm_rsrc sdm_make_rsrc(const void* p, unsigned int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 sdm_buffer_load16(sdm_rsrc r, unsigned int voff, unsigned int soff) {
  return __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
}
__device__ __forceinline__ u32x2 sdm_buffer_load8(sdm_rsrc r, unsigned int voff, unsigned int soff) {
  return __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
}
// 16 bytes per lane (the register-direct epilogue of the F8 kernels: 4 consecutive channels of one pixel)
__device__ __forceinline__ void sdm_buffer_store16(u32x4 v, sdm_rsrc r, unsigned int voff, unsigned int soff) {
  __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)voff, (int)soff, 0);
}
#endif

// ---- buffer loads hidden from the compiler's s_waitcnt bookkeeping (cdna_hip_programming.md 5.7 form (ii)).  hipcc does not count
//      LDS-DMA operations in its vmcnt scoreboard, so every wait it emits for one of ITS OWN loads drains the whole DMA queue.  A
//      kernel that keeps DMAs in flight across barriers issues its register loads through these macros and waits for them with
//      SDM_ASM_LOADS_WAIT (a counted s_waitcnt that names every destination register, so no consumer can be scheduled above it).
#ifdef SDM_EMU
typedef sdm_rsrc sdm_rsrc_raw;
static inline sdm_rsrc_raw sdm_make_rsrc_raw(const void* p, unsigned int bytes) { return sdm_make_rsrc(p, bytes); }
#define SDM_ASM_BUFFER_LOAD16(dst, voff, rsrc, imm) (dst) = sdm_buffer_load16((rsrc), (voff), (imm))
#define SDM_ASM_BUFFER_LOAD16_FIRST(dst, voff, rsrc, imm) (dst) = sdm_buffer_load16((rsrc), (voff), (imm))
#else
typedef u32x4 sdm_rsrc_raw;      // the four descriptor words, held in SGPRs
__device__ __forceinline__ sdm_rsrc_raw sdm_make_rsrc_raw(const void* p, unsigned int bytes) {
  const unsigned long long a = (unsigned long long)p;
  sdm_rsrc_raw d;
  d[0] = (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)a);
  d[1] = (unsigned int)__builtin_amdgcn_readfirstlane((int)((unsigned int)(a >> 32) & 0xFFFFu));      // stride 0
  d[2] = (unsigned int)__builtin_amdgcn_readfirstlane((int)bytes);
  d[3] = 0x00020000u;
  return d;
}
#define SDM_ASM_BUFFER_LOAD16(dst, voff, rsrc, imm) \
  asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:" #imm : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory")
// first load of a batch: the descriptor SGPRs may have been written by a VALU (v_readfirstlane) just before
#define SDM_ASM_BUFFER_LOAD16_FIRST(dst, voff, rsrc, imm) \
  asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, 0 offen offset:" #imm : "=v"(dst) : "v"(voff), "s"(rsrc) : "memory")
#endif

// ---- LDS-DMA (global -> LDS without staging registers): every lane supplies its own 16-B source address, the
//      destination is the wave-uniform `lds_base` + lane*16.  Completion is counted on vmcnt: SDM_WAIT_VMCNT0() then a
//      barrier orders the data for every reader.  SDM_RAW_BARRIER() is s_barrier without the vmcnt(0) drain that
//      __syncthreads() implies, so a DMA issued earlier may stay in flight across it.
#ifdef SDM_EMU
static inline void sdm_glds16(const void* gsrc, unsigned char* lds_base) { memcpy(lds_base + (threadIdx.x & 63) * 16, gsrc, 16); }
static inline void sdm_glds16_buf(sdm_rsrc r, unsigned int voff, unsigned int soff, unsigned char* lds_base) {
  const u32x4 v = sdm_buffer_load16(r, voff, soff);
  memcpy(lds_base + (threadIdx.x & 63) * 16, &v, 16);
}
// 4 bytes per lane (a bias table: 64 floats per wave)
static inline void sdm_glds4_buf(sdm_rsrc r, unsigned int voff, unsigned int soff, unsigned char* lds_base) {
  unsigned int v = 0u;
  const unsigned long long o = (unsigned long long)voff + soff;
  if (o + 4 <= r.bytes) memcpy(&v, r.base + o, 4);
  memcpy(lds_base + (threadIdx.x & 63) * 4, &v, 4);
}
#define SDM_UNIFORM_I(x) (x)
#define SDM_OPAQUE_I(x) ((void)0)
#define SDM_PIN_HERE_V4(a, b, c, d) ((void)0)
#define SDM_WAIT_VMCNT0() ((void)0)
#define SDM_WAIT_VMCNT(n) ((void)0)
#define SDM_WAIT_LGKMCNT0() ((void)0)
#define SDM_SLOAD_I32(dst, ptr) (dst) = *(ptr)
#define SDM_SLOAD_WAIT(dst) ((void)0)
#define SDM_RAW_BARRIER() __syncthreads()
#else
__device__ __forceinline__ void sdm_glds16(const void* gsrc, unsigned char* lds_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_base, 16, 0, 0);
}
// buffer form: SGPR descriptor + 32-bit per-lane byte offset + uniform byte offset (no 64-bit address VGPRs; OOB -> 0)
__device__ __forceinline__ void sdm_glds16_buf(sdm_rsrc r, unsigned int voff, unsigned int soff, unsigned char* lds_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ void sdm_glds4_buf(sdm_rsrc r, unsigned int voff, unsigned int soff, unsigned char* lds_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_base, 4, (int)voff, (int)soff, 0, 0);
}
// value known to be wave-uniform (e.g. threadIdx.x >> 6): lets the compiler keep everything derived from it in SGPRs
#define SDM_UNIFORM_I(x) __builtin_amdgcn_readfirstlane(x)
// make a loop-invariant VGPR value opaque at this point, so that the compiler recomputes cheap address arithmetic derived
// from it inside the loop instead of hoisting N precomputed addresses into N long-lived registers
#define SDM_OPAQUE_I(x) asm volatile("" : "+v"(x))
// force four 128-bit register values to be fully computed at this point (stops the optimiser from sinking their producers
// below a later branch, out of the block whose instruction interleave is being pinned)
#define SDM_PIN_HERE_V4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
#define SDM_WAIT_VMCNT0() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
// counted wait: at most n vector-memory operations of this wave (loads, LDS-DMAs, stores) may still be outstanding afterwards
#define SDM_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define SDM_WAIT_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// one wave-uniform 32-bit word through the scalar cache, counted on lgkmcnt - NOT on vmcnt: a kernel that keeps LDS-DMAs in flight cannot afford a compiler-
// visible vector load (its wait would be a vmcnt that drains the DMA queue, see above).  The result is not valid before SDM_SLOAD_WAIT(dst), which also
// carries the value (so that no use can be scheduled above the wait); issue and wait inside one straight-line region (no loop-carried in-flight register)
#define SDM_SLOAD_I32(dst, ptr) asm volatile("s_load_dword %0, %1, 0x0" : "=s"(dst) : "s"(ptr) : "memory")
#define SDM_SLOAD_WAIT(dst) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(dst) :: "memory")
#define SDM_RAW_BARRIER() do { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#endif

// sum over the 16 lanes of this lane's DPP row (lanes 16r .. 16r+15), returned in every lane of the row: four VALU adds with DPP
// operands (quad_perm [1,0,3,2], [2,3,0,1], row_ror:4, row_ror:8) - no LDS crossbar
#ifdef SDM_EMU
static inline float sdm_sum_row16(float v) {
  v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8);
  return v;
}
#else
#define SDM_DPP_ADD(v, ctrl) ((v) + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), (ctrl), 0xF, 0xF, false)))
__device__ __forceinline__ float sdm_sum_row16(float v) {
  v = SDM_DPP_ADD(v, 0xB1);
  v = SDM_DPP_ADD(v, 0x4E);
  v = SDM_DPP_ADD(v, 0x124);
  v = SDM_DPP_ADD(v, 0x128);
  return v;
}
#endif

// v_perm_b32: byte i of the result = byte sel[i] of the 8-byte value {s0 (bytes 4-7), s1 (bytes 0-3)} (selector values 0-7 only)
// v_permlane32_swap_b32: lanes 32-63 of `a` trade places with lanes 0-31 of `b` (a register pair crosses the wave's two halves in
// one VALU instruction, no LDS crossbar): the accumulator layout of the 32x32 MFMAs splits every 8-channel run over the two halves
#ifdef SDM_EMU
static inline unsigned int sdm_perm_b32(unsigned int s0, unsigned int s1, unsigned int sel) {
  const unsigned long long v = ((unsigned long long)s0 << 32) | s1;
  unsigned int r = 0;
  for (int i = 0; i < 4; ++i) r |= (unsigned int)((v >> (8 * ((sel >> (8 * i)) & 7))) & 0xffull) << (8 * i);
  return r;
}
static inline void sdm_permlane32_swap(unsigned int& a, unsigned int& b) {
  auto& w = emu::wave(); const int l = emu::lane_id();
  w.slot[l][0] = a; w.slot[l][1] = b; emu::wave_barrier();
  if (l < 32) b = w.slot[l + 32][0]; else a = w.slot[l - 32][1];
  emu::wave_barrier();
}
#else
__device__ __forceinline__ unsigned int sdm_perm_b32(unsigned int s0, unsigned int s1, unsigned int sel) { return __builtin_amdgcn_perm(s0, s1, sel); }
__device__ __forceinline__ void sdm_permlane32_swap(unsigned int& a, unsigned int& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
#endif

#define SDM_LOG2E 1.4426950408889634f

SDM_DEV_INLINE float sdm_silu(float x) {
  // x * sigmoid(x) = x / (1 + 2^(-x*log2e))
  return x / (1.0f + sdm_exp2(-x * SDM_LOG2E));
}

SDM_DEV_INLINE float sdm_gelu_erf(float x) {
  // exact GELU (diffusers GEGLU uses F.gelu default, approximate='none')
  return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
}

// 8 consecutive channels from an fp16 or fp32 tensor -> f16x8 (the MFMA operand type)
SDM_DEV_INLINE f16x8 sdm_load8_as_f16(const void* base, size_t elem_index, int is_f32) {
  if (is_f32) {
    const f32x4* p = (const f32x4*)((const float*)base + elem_index);
    f32x4 a = p[0], b = p[1];
    f16x8 r;
    r[0] = (half_t)a[0]; r[1] = (half_t)a[1]; r[2] = (half_t)a[2]; r[3] = (half_t)a[3];
    r[4] = (half_t)b[0]; r[5] = (half_t)b[1]; r[6] = (half_t)b[2]; r[7] = (half_t)b[3];
    return r;
  }
  return *(const f16x8*)((const half_t*)base + elem_index);
}

// 8 consecutive channels -> 8 floats
SDM_DEV_INLINE void sdm_load8_as_f32(const void* base, size_t elem_index, int is_f32, float (&v)[8]) {
  if (is_f32) {
    const f32x4* p = (const f32x4*)((const float*)base + elem_index);
    f32x4 a = p[0], b = p[1];
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  } else {
    f16x8 h = *(const f16x8*)((const half_t*)base + elem_index);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)h[i];
  }
}

static inline int sdm_cdiv(int a, int b) { return (a + b - 1) / b; }
