// Is the GEMM epilogue's ~13 B/clk/CU store rate a per-CU limit or the chip's write bandwidth?  Every workgroup (512 threads)
// writes 256x256 bf16 output tiles (128 KB) with the persistent NT kernel's epilogue pattern -- per wave-instruction 8 full
// 128-B lines of 8 rows, global_store_dwordx4 -- and nothing else; grids of 256 / 128 / 64 / 32 workgroups on the same total
// bytes per workgroup.  Build: hipcc --offload-arch=gfx950 -O3 tools/store_bench.hip -o tools/store_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int WIDTH>   // bytes per lane per store: 16 or 8
__global__ __launch_bounds__(512, 2) void k(char* __restrict__ C, int ldc2 /* row bytes */, int tiles_n, int tiles_per_wg, int stride_tiles) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  uint4 val = make_uint4(tid, wave, lane, 7);
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int tile = blockIdx.x + t * stride_tiles;
    const long m0 = (long)(tile / tiles_n) * 256, n0 = (long)(tile % tiles_n) * 256;
    char* base = C + (m0 + wr * 128) * ldc2 + (n0 + wc * 64) * 2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {              // 8 chunks of 16 rows, 2 stores of 8 rows x 128 B each
      char* p0 = base + (long)(i * 16 + (lane >> 3)) * ldc2 + (lane & 7) * 16;
      char* p1 = p0 + 8L * ldc2;
      if (WIDTH == 16) {
        *reinterpret_cast<uint4*>(p0) = val;
        *reinterpret_cast<uint4*>(p1) = val;
      } else {
        *reinterpret_cast<uint2*>(p0) = make_uint2(val.x, val.y); *reinterpret_cast<uint2*>(p0 + 8) = make_uint2(val.z, val.w);
        *reinterpret_cast<uint2*>(p1) = make_uint2(val.x, val.y); *reinterpret_cast<uint2*>(p1 + 8) = make_uint2(val.z, val.w);
      }
      val.x += i;
    }
  }
}

int main() {
  const long M = 262144; const int N = 2304, tiles_n = N / 256, ldc2 = N * 2;
  char* C; CK(hipMalloc(&C, (size_t)M * ldc2));
  const int total_tiles = (int)(M / 256) * tiles_n;          // 9216
  for (int wgs : {256, 128, 64, 32, 8}) {
    const int per = 36;                                       // tiles per workgroup: what one CU writes in the QKV GEMM
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<16>), dim3(wgs), dim3(512), 0, 0, C, ldc2, tiles_n, per, wgs);
    CK(hipEventRecord(a, 0));
    for (int it = 0; it < 5; ++it) hipLaunchKernelGGL((k<16>), dim3(wgs), dim3(512), 0, 0, C, ldc2, tiles_n, per, wgs);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    const double bytes = (double)wgs * per * 131072.0;
    printf("%3d workgroups x %d tiles: %7.3f ms  %6.2f TB/s chip  %6.1f GB/s per WG (%4.1f B/clk at 2.1 GHz)  %5.2f us per 128-KB tile\n",
           wgs, per, ms, bytes / ms / 1e9, bytes / ms / 1e6 / wgs, bytes / ms / 1e6 / wgs / 2.1, ms * 1e3 / per);
  }
  (void)total_tiles;
  return 0;
}
