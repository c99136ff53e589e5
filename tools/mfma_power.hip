// Which MFMA shape sustains more FLOP/s under the power cap?  Pure register-resident MFMA loops on random bf16 data
// (no LDS / memory traffic), same accumulator count and operand reuse as the NT kernel's wave tile (128x64).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_power.hip -o tools/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ bf16x8 rnd_frag(unsigned& s, int zero) {
  union { bf16x8 v; unsigned short u[8]; } r;
  for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; r.u[i] = zero ? 0 : (unsigned short)(((s >> 9) & 0x807F) | 0x3F00 | ((s >> 3) & 0x0080)); }   // ~N(0,1)-ish magnitudes
  return r.v;
}

__global__ __launch_bounds__(512, 2) void k16(int iters, int zero, float* out) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
  bf16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = rnd_frag(s, zero);
  for (int i = 0; i < 4; ++i) b[i] = rnd_frag(s, zero);
  f32x4 acc[8][4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    // rotate operands so successive iterations see different data (toggle activity like a real stream)
    bf16x8 t = a[0];
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i] = a[i + 1];
    a[7] = t;
  }
  float r = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) r += acc[i][j][0] + acc[i][j][3];
  if (r == 12345.678f) out[0] = r;
}

__global__ __launch_bounds__(512, 2) void k32(int iters, int zero, float* out) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1;
  // wave tile 128x64 = 4x2 blocks of 32x32; per K=32: two k16 halves -> a[4][2], b[2][2]
  bf16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i) a[i] = rnd_frag(s, zero);
  for (int i = 0; i < 4; ++i) b[i] = rnd_frag(s, zero);
  f32x16 acc[4][2] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j * 2 + h], a[i * 2 + h], acc[i][j], 0, 0, 0);
    bf16x8 t = a[0];
#pragma unroll
    for (int i = 0; i < 7; ++i) a[i] = a[i + 1];
    a[7] = t;
  }
  float r = 0;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) r += acc[i][j][0] + acc[i][j][15];
  if (r == 12345.678f) out[0] = r;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  float* out; CK(hipMalloc(&out, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int zero = 0; zero < 2; ++zero)
    for (int which = 0; which < 2; ++which) {
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        if (which == 0) hipLaunchKernelGGL(k16, dim3(256), dim3(512), 0, 0, iters, zero, out);
        else hipLaunchKernelGGL(k32, dim3(256), dim3(512), 0, 0, iters, zero, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // per wave per iteration: 128x64x32 MACs (both shapes) = 262144 MAC = 524288 flop; 8 waves x 256 blocks
        const double fl = 2.0 * 128 * 64 * 32 * 8.0 * 256 * iters;
        if (rep == 2) printf("%s %s: %.2f ms  %.0f TFLOP/s\n", which ? "32x32x16" : "16x16x32", zero ? "zeros " : "random", ms, fl / ms / 1e9);
      }
    }
  return 0;
}
