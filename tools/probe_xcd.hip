// Which XCD does block b land on?  (s_getreg_b32 HW_REG_XCC_ID).  Build: hipcc --offload-arch=gfx950 -O2 tools/probe_xcd.hip -o tools/probe_xcd
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out, int spin) {
  extern __shared__ char smem[];
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) out[blockIdx.x] = (int)x;
  long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 1) smem[0] = 1;
}
int main() {
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int grid = cfg == 0 ? 256 : cfg == 1 ? 1024 : 3072, threads = cfg == 0 ? 512 : 512, lds = cfg == 2 ? 131072 : 163840;
    int* d; hipMalloc(&d, grid * 4);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, d, 200000);
    std::vector<int> h(grid); hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost);
    int ok = 0; for (int b = 0; b < grid; ++b) ok += ((h[b] & 15) == (b & 7));
    printf("grid %d lds %d: xcc_id(b) == b%%8 for %d/%d blocks; first 24:", grid, lds, ok, grid);
    for (int b = 0; b < 24; ++b) printf(" %d", h[b] & 15);
    printf("\n");
    hipFree(d);
  }
  return 0;
}
