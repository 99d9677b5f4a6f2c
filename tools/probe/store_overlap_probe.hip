// Micro-benchmark (bench helper, not part of the product): can a wave's epilogue stores drain underneath the MFMAs that follow them in
// program order?  One 256-thread block per CU (4 waves, one per SIMD - the consumer waves of the F8 conv kernel), each wave repeats
//   [NM back-to-back v_mfma_f32_32x32x16_f16 on 8 accumulator tiles]  then  [32 x buffer_store_dwordx4 of the 128 accumulator registers]
// (the store pattern of an accumulator-layout epilogue with the operands swapped: lane = pixel, 4 consecutive registers = 4 consecutive
// channels -> 16 B per lane, 32 B contiguous per pixel and instruction).  mode 0: MFMAs only, 1: stores only, 2: both.  If the stores
// of tile k overlapped the MFMAs of tile k+1, t(2) would be ~max(t(0), t(1)); if the wave stalls until the memory pipeline has accepted
// every store, t(2) ~ t(0) + t(1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256, 1) probe(float* out, int tiles, int nm, int cstride) {
  extern __shared__ unsigned char smem[];      // 120 KB: one block per CU
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    unsigned h = (unsigned)(tid * 8 + j) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    a[j] = (_Float16)(((float)(h & 0xffff) / 32768.0f - 1.0f));
    b[j] = (_Float16)(((float)(h >> 16) / 32768.0f - 1.0f) * 0.25f);
  }
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  // tile t of block blk: 256 pixels x 128 channels fp32; this wave: 128 pixels (4 rows of 32) x 64 channels
  const size_t tile_floats = (size_t)256 * cstride;
  for (int t = 0; t < tiles; ++t) {
    if (MODE != 1) {
      for (int it = 0; it < nm; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[i], 0, 0, 0);
      }
    }
    if (MODE != 0) {
      float* tb = out + ((size_t)blockIdx.x * tiles + t) % 4096 * tile_floats;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int px = (wave >> 1) * 128 + i * 32 + (lane & 31), ch = (wave & 1) * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
            f32x4 v = {acc[i * 2 + j][4 * g], acc[i * 2 + j][4 * g + 1], acc[i * 2 + j][4 * g + 2], acc[i * 2 + j][4 * g + 3]};
            *(f32x4*)(tb + (size_t)px * cstride + ch) = v;
          }
    }
  }
  if (MODE == 0) { float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0]; if (s == 123.456f) out[tid] = s; }
}

template <int MODE> static float run(float* d, int tiles, int nm, int cstride, int grid = 256) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<grid, 256, 120 * 1024>>>(d, tiles, nm, cstride);
  hipEventRecord(e0);
  probe<MODE><<<grid, 256, 120 * 1024>>>(d, tiles, nm, cstride);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const int cstride = 128, tiles = 64;
  float* d; hipMalloc(&d, (size_t)4096 * 256 * cstride * 4);      // 512 MB ring of tiles
  for (int nm : {36, 72, 144, 288}) {      // 288, 576, 1152, 2304 MFMAs per wave and tile (the 128-channel F8 tile executes ~1152 fp16-MFMA times)
    const float t0 = run<0>(d, tiles, nm, cstride), t1 = run<1>(d, tiles, nm, cstride), t2 = run<2>(d, tiles, nm, cstride);
    printf("per tile: %4d MFMAs/wave: mfma only %7.2f us | 32 dwordx4 stores/wave only %6.2f us | both %7.2f us  (sum %7.2f, max %7.2f)\n", nm * 8,
           t0 * 1e3 / tiles, t1 * 1e3 / tiles, t2 * 1e3 / tiles, (t0 + t1) * 1e3 / tiles, (t0 > t1 ? t0 : t1) * 1e3 / tiles);
  }
  // the same stores from fewer CUs: is the ~5 us per 128 KB tile the CU's store path or the chip's HBM write bandwidth?
  for (int grid : {256, 128, 64, 32, 8}) {
    const float t1 = run<1>(d, tiles, 36, cstride, grid), t2 = run<2>(d, tiles, 36, cstride, grid), t0 = run<0>(d, tiles, 36, cstride, grid);
    printf("grid %3d blocks: 32 dwordx4 stores/wave only %6.2f us per tile (%6.1f GB/s per CU, %5.2f TB/s chip) | 288 MFMAs only %6.2f | both %6.2f\n", grid,
           t1 * 1e3 / tiles, 131072.0 / (t1 * 1e3 / tiles) * 1e-3, grid * 131072.0 / (t1 * 1e3 / tiles) * 1e-6, t0 * 1e3 / tiles, t2 * 1e3 / tiles);
  }
  return 0;
}
