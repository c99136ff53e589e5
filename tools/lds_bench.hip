// LDS read throughput of one CU by read form, with the addressing of the persistent GEMM kernels: ds_read_b128 on the NT kernel's
// swizzled [row][128 B] image (gemm_nt_p3_kernel), ds_read_b64_tr_b16 pairs on the TN kernel's [k row][512 B] image
// (gemm_tn2_kernel), plain ds_read_b64 on the latter as the control; optionally with the MFMAs the fragments feed (32 per 12
// fragments = one k-step of a 128 x 64 wave tile).  8 waves per CU, 256 workgroups, bytes per clock per CU at the measured clock.
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_bench.hip -o tools/lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef short bf16x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bfx8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, bool MFMA>   // 0: ds_read_b128 (12 per step), 1: ds_read_b64_tr_b16 (24 per step), 2: ds_read_b64 (24 per step)
__global__ __launch_bounds__(512, 2) void k(int iters, float* out, long long* clk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 65536 / 4; i += 512) ((uint32_t*)smem)[i] = 0x3c003c00u + i;   // finite fp16 pairs
  __syncthreads();
  const int wr = wave >> 2, wc = wave & 3, fr = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 fa[8], fb[4];
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      // NT image: A rows (wr*128 + i*16 + fr) * 128 B, 16-B chunk (fg ^ swz); two k halves alternate by iteration
      const int swz = (fr >> 1) & 7;
      const uint32_t a = lds0 + (uint32_t)((wr * 128 + fr) * 128 + ((((it & 1) * 4 + fg) ^ swz) << 4));
      const uint32_t b = lds0 + 32768u + (uint32_t)((wc * 64 + fr) * 128 + ((((it & 1) * 4 + fg) ^ swz) << 4));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fa[i]) : "v"(a), "n"(i * 2048) : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(fb[j]) : "v"(b), "n"(j * 2048) : "memory");
    } else {
      // TN image: k row kr at kr * 512 B, 32-B chunk q at q ^ (kr & 7); fragment = rows 4 fg + (fr >> 2) and + 16
      const int r_lo = 4 * fg + (fr >> 2), x_lo = r_lo & 7, x_hi = (r_lo + 16) & 7;
      const uint32_t kb = lds0 + (uint32_t)((it & 1) * 32 * 512);
      const uint32_t row_lo = kb + (uint32_t)(r_lo * 512 + (fr & 3) * 8), row_hi = kb + (uint32_t)((r_lo + 16) * 512 + (fr & 3) * 8);
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const int ct = i < 8 ? wr * 8 + i : wc * 4 + (i - 8);
        const uint32_t base = i < 8 ? 0u : 32768u;
        const uint32_t alo = base + row_lo + (uint32_t)((ct ^ x_lo) << 5), ahi = base + row_hi + (uint32_t)((ct ^ x_hi) << 5);
        bf16x4 lo, hi;
        if (MODE == 1) {
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(lo) : "v"(alo) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=&v"(hi) : "v"(ahi) : "memory");
        } else {
          asm volatile("ds_read_b64 %0, %1" : "=&v"(lo) : "v"(alo) : "memory");
          asm volatile("ds_read_b64 %0, %1" : "=&v"(hi) : "v"(ahi) : "memory");
        }
        asm volatile("" : "+v"(lo), "+v"(hi));
        const bf16x8 f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        if (i < 8) fa[i] = f; else fb[i - 8] = f;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fa[2]), "+v"(fa[3]), "+v"(fa[4]), "+v"(fa[5]), "+v"(fa[6]), "+v"(fa[7]),
                 "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3])::"memory");
    if (MFMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, fb[j]), __builtin_bit_cast(bfx8, fa[i]), acc[i][j], 0, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i][0][0] += __uint_as_float((uint32_t)(uint16_t)fa[i][0] << 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[0][j][1] += __uint_as_float((uint32_t)(uint16_t)fb[j][0] << 16);
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  if (s == 123.456f) out[0] = s;
  if (tid == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE, bool MFMA>
void run(const char* name, int iters) {
  float* out; long long* clk;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&clk, 64));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, MFMA>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  k<MODE, MFMA><<<256, 512, 65536>>>(iters / 10, out, clk);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  k<MODE, MFMA><<<256, 512, 65536>>>(iters, out, clk);
  CK(hipEventRecord(b)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double bytes_cu = (double)iters * 8 /*waves*/ * 12 * 1024;      // 12 KB of fragments per wave and k-step
  const double tf = MFMA ? (double)iters * 256 * 8 * 32 * 16384.0 / (ms * 1e-3) / 1e12 : 0.0;
  printf("%-34s %8.3f ms  %7.1f GB/s per CU  %6.1f TB/s chip", name, ms, bytes_cu / (ms * 1e-3) / 1e9, bytes_cu * 256 / (ms * 1e-3) / 1e12);
  if (MFMA) printf("  %7.1f TFLOP/s", tf);
  printf("\n");
}

int main() {
  const int iters = 20000;
  run<0, false>("ds_read_b128 (NT image)", iters);
  run<1, false>("ds_read_b64_tr_b16 (TN image)", iters);
  run<2, false>("ds_read_b64 (TN image)", iters);
  run<0, true>("ds_read_b128 + 32 MFMA / step", iters);
  run<1, true>("ds_read_b64_tr_b16 + 32 MFMA / step", iters);
  run<2, true>("ds_read_b64 + 32 MFMA / step", iters);
  return 0;
}
