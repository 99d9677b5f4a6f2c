// Hardware-semantics probe (bench helper): operand layout and scale handling of v_mfma_scale_f32_32x32x64_f8f6f4 with fp8 e4m3
// operands, and rounding / saturation of v_cvt_pk_fp8_f32 on gfx950.  Prints PASS/FAIL lines; the conv kernel's fp8 residual
// path and its CPU emulation (tests/emu) are written against exactly these checks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

static float e4m3_to_f(unsigned char v) {
  const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float r;
  if (e == 0) r = ldexpf((float)m, -9);                 // subnormal: m/8 * 2^-6
  else if (e == 15 && m == 7) r = NAN;
  else r = ldexpf(1.0f + m / 8.0f, e - 7);
  return s ? -r : r;
}

__global__ void mfma_k(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb) {
  // A: [32 rows][64 k] bytes, B: [32 cols][64 k] bytes (B^T); hypothesis: lane l holds row/col (l & 31), k = 32*(l>>5) .. +31
  const int l = threadIdx.x;
  i32x8 a, b;
  const int* ap = (const int*)(A + (l & 31) * 64 + (l >> 5) * 32);
  const int* bp = (const int*)(B + (l & 31) * 64 + (l >> 5) * 32);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
  // C/D: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

static float e5m2_to_f(unsigned char v) {
  const int s = v >> 7, e = (v >> 2) & 31, m = v & 3;
  float r;
  if (e == 0) r = ldexpf((float)m, -16);
  else if (e == 31) r = m ? NAN : INFINITY;
  else r = ldexpf(1.0f + m / 4.0f, e - 15);
  return s ? -r : r;
}

// the same MFMA with A in e5m2 (cbsz = 1), B in e4m3: what the F8 conv / GEMM kernels issue since round 3
__global__ void mfma_bf8a_k(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb) {
  const int l = threadIdx.x;
  i32x8 a, b;
  const int* ap = (const int*)(A + (l & 31) * 64 + (l >> 5) * 32);
  const int* bp = (const int*)(B + (l & 31) * 64 + (l >> 5) * 32);
  for (int i = 0; i < 8; ++i) { a[i] = ap[i]; b[i] = bp[i]; }
  f32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 1, 0, 0, sa, 0, sb);
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}

__global__ void cvt_bf8_k(const float* x, unsigned int* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = (unsigned)__builtin_amdgcn_cvt_pk_bf8_f32(x[2 * i], x[2 * i + 1], 0x55555555, false);
}

__global__ void cvt_k(const float* x, unsigned int* out, int n) {
  const int i = threadIdx.x;
  if (i < n) {
    out[i * 2 + 0] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0x55555555, false);   // low word
    out[i * 2 + 1] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0x55555555, true);    // high word
  }
}

int main() {
  std::vector<unsigned char> A(32 * 64), B(32 * 64);
  srand(1);
  for (auto& v : A) { v = (unsigned char)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v ^= 1; if (((v >> 3) & 15) > 9) v &= 0xbf; }   // moderate exponents, no NaN
  for (auto& v : B) { v = (unsigned char)(rand() & 0xff); if ((v & 0x7f) == 0x7f) v ^= 1; if (((v >> 3) & 15) > 9) v &= 0xbf; }
  unsigned char *dA, *dB; float* dD;
  hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 32 * 32 * 4);
  hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
  std::vector<float> D(32 * 32), R(32 * 32);
  for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
    double s = 0; for (int k = 0; k < 64; ++k) s += (double)e4m3_to_f(A[i * 64 + k]) * (double)e4m3_to_f(B[j * 64 + k]);
    R[i * 32 + j] = (float)s;
  }
  const int cases[4][2] = {{127, 127}, {130, 127}, {127, 120}, {124, 133}};
  for (auto& cs : cases) {
    mfma_k<<<1, 64>>>(dA, dB, dD, cs[0], cs[1]);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    const double f = ldexp(1.0, cs[0] - 127 + cs[1] - 127);
    double md = 0, mr = 0;
    for (int i = 0; i < 1024; ++i) { md = fmax(md, fabs(D[i] - R[i] * f)); mr = fmax(mr, fabs(R[i] * f)); }
    printf("mfma f8 layout+scale sa=%d sb=%d: max|d|=%.3e of max|ref|=%.3e  %s\n", cs[0], cs[1], md, mr, md <= 1e-5 * mr ? "PASS" : "FAIL");
  }
  // scale = 0 -> "unscaled": treated as 2^0 or 2^-127 ?
  mfma_k<<<1, 64>>>(dA, dB, dD, 0, 0);
  hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
  printf("scale args 0,0: D[0]=%g ref=%g ratio=%g\n", D[0], R[0], D[0] / R[0]);

  // conversions
  const float xs[16] = {1.0f, -1.0f, 0.3f, 448.0f, 449.0f, 1000.0f, -1e6f, 0.0625f, 0.001f, 1.0625f, 1.1875f, 17.0f, 0.0019f, 464.0f, 3.3e-3f, -0.0f};
  float* dx; unsigned int* dout; hipMalloc(&dx, 64); hipMalloc(&dout, 64);
  hipMemcpy(dx, xs, 64, hipMemcpyHostToDevice);
  cvt_k<<<1, 64>>>(dx, dout, 8);
  unsigned int o[16]; hipMemcpy(o, dout, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 8; ++i) {
    const unsigned lo = o[2 * i], hi = o[2 * i + 1];
    printf("cvt_pk_fp8(%g, %g): word_sel=0 -> 0x%08x  (bytes %g, %g)   word_sel=1 -> 0x%08x\n", xs[2 * i], xs[2 * i + 1], lo,
           e4m3_to_f(lo & 0xff), e4m3_to_f((lo >> 8) & 0xff), hi);
  }
  // ---- e5m2 ("bf8") A operand: layout + scales, then v_cvt_pk_bf8_f32 rounding / range ----
  {
    std::vector<unsigned char> A5(32 * 64);
    for (auto& v : A5) { v = (unsigned char)(rand() & 0xff); if (((v >> 2) & 31) > 18) v &= 0xbf; if (((v >> 2) & 31) == 31) v &= 0x83; }   // moderate exponents, no inf / NaN
    hipMemcpy(dA, A5.data(), A5.size(), hipMemcpyHostToDevice);
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
      double s = 0; for (int k = 0; k < 64; ++k) s += (double)e5m2_to_f(A5[i * 64 + k]) * (double)e4m3_to_f(B[j * 64 + k]);
      R[i * 32 + j] = (float)s;
    }
    const int cs5[2][2] = {{127, 127}, {116, 121}};
    for (auto& cs : cs5) {
      mfma_bf8a_k<<<1, 64>>>(dA, dB, dD, cs[0], cs[1]);
      hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
      const double f = ldexp(1.0, cs[0] - 127 + cs[1] - 127);
      double md = 0, mr = 0;
      for (int i = 0; i < 1024; ++i) { md = fmax(md, fabs(D[i] - R[i] * f)); mr = fmax(mr, fabs(R[i] * f)); }
      printf("mfma A=e5m2 B=e4m3 layout+scale sa=%d sb=%d: max|d|=%.3e of max|ref|=%.3e  %s\n", cs[0], cs[1], md, mr, md <= 2e-4 * mr ? "PASS" : "FAIL");
    }
    const float x5[24] = {1.0f, -1.0f, 0.3f, 57344.0f, 57345.0f, 61439.0f, 61440.0f, 65504.0f, 1e6f, -1e6f, 1.125f, 1.375f, 1.625f, 1.875f,
                          1.5259e-5f, 7.6294e-6f, 7.7e-6f, 2.2888e-5f, 6.1035e-5f, 3.0e-5f, 5.0f, 7.0f, 0.1f, -0.0f};
    float* dx5; unsigned int* do5; hipMalloc(&dx5, 96); hipMalloc(&do5, 48);
    hipMemcpy(dx5, x5, 96, hipMemcpyHostToDevice);
    cvt_bf8_k<<<1, 64>>>(dx5, do5, 12);
    unsigned int o5[12]; hipMemcpy(o5, do5, 48, hipMemcpyDeviceToHost);
    for (int i = 0; i < 12; ++i)
      printf("cvt_pk_bf8(%g, %g) -> 0x%08x  (bytes 0x%02x = %g, 0x%02x = %g)\n", x5[2 * i], x5[2 * i + 1], o5[i], o5[i] & 0xff, e5m2_to_f(o5[i] & 0xff),
             (o5[i] >> 8) & 0xff, e5m2_to_f((o5[i] >> 8) & 0xff));
  }
  return 0;
}