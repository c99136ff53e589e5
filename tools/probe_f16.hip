// gfx950 probes for the fp16 engine (not part of the product):
//   1. does v_mfma_f32_16x16x32_f16 honour fp16 SUBNORMAL inputs (MI200 flushed them)?
//   2. v_cvt_pk_f16_f32: rounding (nearest even?) and subnormal results
//   3. sustained MFMA rate under the power cap: f16 vs bf16 operands, random data / zeros (register-resident loop, the NT
//      kernel's wave tile), as tools/mfma_power does for the two bf16 shapes
// Build: hipcc --offload-arch=gfx950 -O3 tools/probe_f16.hip -o tools/probe_f16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// D[i][j] = sum_k A[i][k] B[j][k]; every lane passes a = {av,0,...}, b = {bv,0,..}: lanes with fg==0 contribute k=0 only
__global__ void denorm_kernel(float* out, unsigned short abits, unsigned short bbits) {
  const int lane = threadIdx.x;
  s8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
  if ((lane >> 4) == 0) { a[0] = (short)abits; b[0] = (short)bbits; }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), acc, 0, 0, 0);
  if (lane == 0) out[0] = acc[0];
}
__global__ void cvt_kernel(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) { f2 v = {in[2 * i], in[2 * i + 1]}; out[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(v, h2)); }
}

template <bool F16>
__global__ __launch_bounds__(512, 2) void kpow(int iters, int zero, float* out) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
  s8 a[8], b[4];
  auto frag = [&](s8& r) {
    for (int i = 0; i < 8; ++i) {
      s = s * 1664525u + 1013904223u;
      unsigned short u;
      if (F16) u = (unsigned short)(((s >> 9) & 0x83FF) | 0x3800 | ((s >> 3) & 0x0400));       // sign, exponent 14/15, 10 random mantissa bits
      else u = (unsigned short)(((s >> 9) & 0x807F) | 0x3F00 | ((s >> 3) & 0x0080));
      r[i] = zero ? 0 : (short)u;
    }
  };
  for (int i = 0; i < 8; ++i) frag(a[i]);
  for (int i = 0; i < 4; ++i) frag(b[i]);
  f32x4 acc[8][4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, b[j]), __builtin_bit_cast(h8, a[i]), acc[i][j], 0, 0, 0);
        else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
      }
    s8 t = a[0];
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i] = a[i + 1];
    a[7] = t;
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][3];
  if (r == 12345.678f) out[0] = r;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out; CK(hipMalloc(&out, 64));
  float h;
  // 1. subnormal inputs: a = 2^-24 (smallest subnormal, bits 0x0001), b = 2^10 (0x6400) -> product 2^-14 if honoured, 0 if flushed
  struct { unsigned short a, b; const char* what; double expect; } cases[] = {
      {0x0001, 0x6400, "2^-24 (min subnormal) x 1024", ldexp(1.0, -14)},
      {0x03FF, 0x3C00, "max subnormal x 1", 1023 * ldexp(1.0, -24)},
      {0x0400, 0x3C00, "min normal x 1", ldexp(1.0, -14)},
      {0x0001, 0x0001, "min subnormal squared (2^-48: f32 normal)", ldexp(1.0, -48)}};
  for (auto& c : cases) {
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, out, c.a, c.b);
    CK(hipMemcpy(&h, out, 4, hipMemcpyDeviceToHost));
    printf("mfma_f16 %-45s -> %.9g (exact %.9g) %s\n", c.what, h, c.expect, h == (float)c.expect ? "HONOURED" : (h == 0.f ? "FLUSHED" : "??"));
  }
  // 2. conversions
  {
    float in[16] = {1.0f + ldexpf(1.f, -11), 1.0f + 3 * ldexpf(1.f, -11), 1.0f + ldexpf(1.f, -11) + ldexpf(1.f, -20), 65519.f, 65520.f, 1e-5f, 3e-8f, 2.9e-8f,
                    -1e-7f, 6.0e-8f, 70000.f, -70000.f, 0.1f, 1e-3f, 5.96e-8f, 0.f};
    float* din; unsigned* dout; unsigned ho[8];
    CK(hipMalloc(&din, sizeof(in))); CK(hipMalloc(&dout, 32));
    CK(hipMemcpy(din, in, sizeof(in), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, dout, 8);
    CK(hipMemcpy(ho, dout, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) printf("cvt_pk_f16_f32 %-14.9g -> 0x%04x\n", in[i], (ho[i / 2] >> ((i & 1) * 16)) & 0xFFFF);
  }
  // 3. power
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int zero = 0; zero < 2; ++zero)
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (which == 0) hipLaunchKernelGGL(kpow<false>, dim3(256), dim3(512), 0, 0, iters, zero, out);
        else hipLaunchKernelGGL(kpow<true>, dim3(256), dim3(512), 0, 0, iters, zero, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double fl = 2.0 * 128 * 64 * 32 * 8.0 * 256 * iters;
        if (rep == 2) printf("16x16x32 %s %s: %.2f ms  %.0f TFLOP/s\n", which ? "f16 " : "bf16", zero ? "zeros " : "random", ms, fl / ms / 1e9);
      }
    }
  return 0;
}
