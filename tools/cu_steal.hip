// What does a kernel that holds k CUs for the whole launch -- the RCCL ring of the overlapped gradient all-reduce -- do to the
// persistent GEMMs, and does a compute-CU budget (simx_set_compute_cus) fix it?  One GPU, not part of the product.
//   hog:  k workgroups (256 threads, 32 KB of LDS: a persistent GEMM workgroup needs the whole CU's 160 KB and cannot share the CU)
//         copy memory at a few hundred GB/s on a second stream until the host raises a flag -- the stand-in for the ring;
//   work: N launches each of the QKV projection (gemm_nt_p3), the FFN-out projection (gemm_nt_p5) and the FFN-in weight gradient
//         (gemm_tn5 + slab pass) on the first stream, timed with HIP events.
// Build: hipcc -O2 --offload-arch=gfx950 tools/cu_steal.hip -Iinclude -Lsimxns_amd -lsimx_hip -Wl,-rpath,$PWD/simxns_amd -o tools/cu_steal
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "simx.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define SX(x) do { int r = (x); if (r) { printf("simx error %d: %s (line %d)\n", r, simx_last_error(), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void hog_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16_per_wg, volatile int* stop,
                                                  unsigned long long* bytes) {
  extern __shared__ char lds[];
  lds[threadIdx.x] = 0;
  const uint4* s = src + (size_t)blockIdx.x * n16_per_wg;
  uint4* d = dst + (size_t)blockIdx.x * n16_per_wg;
  unsigned long long moved = 0;
  for (int it = 0; it < 20000; ++it) {          // (bounded: a host that dies before raising the flag must not leave the GPU spinning)
    for (size_t i = threadIdx.x; i < n16_per_wg; i += 256) d[i] = s[i];
    moved += n16_per_wg * 32;
    __syncthreads();
    if (*stop) break;
  }
  if (threadIdx.x == 0) atomicAdd(bytes, moved);
}

static void* dalloc(size_t bytes) {
  void* p; CK(hipMalloc(&p, bytes));
  std::vector<unsigned short> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) { _Float16 hf = (_Float16)((((int)((i * 2654435761u) >> 20 & 1023)) - 512) / 1024.0f); memcpy(&h[i], &hf, 2); }
  for (size_t off = 0; off < bytes; off += h.size() * 2) CK(hipMemcpy((char*)p + off, h.data(), std::min(h.size() * 2, bytes - off), hipMemcpyHostToDevice));
  return p;
}

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 262144, H = 768, F = 3072, reps = 8;
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  void* A = dalloc((size_t)T * F * 2); void* A2 = dalloc((size_t)T * F * 2); void* C = dalloc((size_t)T * F * 2);
  void* W = dalloc((size_t)F * H * 2 * 2);
  float* bias = (float*)dalloc(F * 4 * 2); float* G = (float*)dalloc((size_t)F * H * 4);
  CK(hipMemset(bias, 0, F * 8));
  size_t wsb = 4 * simx_gemm_tn_workspace_bytes(F, H, T);       // (a smaller compute-CU budget changes the split plan, not the bound used here)
  void* ws; CK(hipMalloc(&ws, wsb + 256));
  const size_t hog_bytes_per_wg = 4u << 20;                      // 4 MB source + 4 MB destination per workgroup
  void *hs, *hd; CK(hipMalloc(&hs, 32 * hog_bytes_per_wg)); CK(hipMalloc(&hd, 32 * hog_bytes_per_wg));
  int* stop; CK(hipHostMalloc((void**)&stop, sizeof(int), hipHostMallocMapped));
  unsigned long long* moved; CK(hipMalloc(&moved, 8));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 32768));
  hipEvent_t e0, e1, h0, h1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&h0)); CK(hipEventCreate(&h1));

  struct Work { const char* name; int id; } works[] = {{"gemm_nt_p3 qkv N=2304 K=768", 0}, {"gemm_nt_p5 ffn-out N=768 K=3072", 1}, {"gemm_tn5 w1 [3072,768] + slab pass", 2}};
  auto launch = [&](int id) {
    if (id == 0) SX(simx_gemm_nt(s1, SIMX_F16, T, 3 * H, H, A, H, W, H, C, 3 * H, bias, nullptr, 0, 0, nullptr, 0, nullptr, 0));
    else if (id == 1) SX(simx_gemm_nt(s1, SIMX_F16, T, H, F, A, F, W, F, C, H, bias, nullptr, 0, 0, nullptr, 0, nullptr, 0));
    else SX(simx_gemm_tn_bias(s1, SIMX_F16, F, H, T, A, F, A2, H, G, H, 1, ws, simx_gemm_tn_workspace_bytes(F, H, T), bias));
  };
  const int ks[] = {0, 8, 16, 32};
  printf("[\n");
  bool first = true;
  for (int k : ks) {
    for (int pol = 0; pol < (k ? 2 : 1); ++pol) {               // 0: every CU (static shares of 256); 1: budget = 256 - k
      const int budget = pol ? 256 - k : 0;
      SX(simx_set_compute_cus(budget));
      for (auto& w : works) {
        launch(w.id); launch(w.id);                             // warm-up (and the plan's one-time setup)
        CK(hipStreamSynchronize(s1));
        *stop = 0;
        CK(hipMemsetAsync(moved, 0, 8, s2));
        if (k) {
          CK(hipEventRecord(h0, s2));
          hipLaunchKernelGGL(hog_kernel, dim3(k), dim3(256), 32768, s2, (const uint4*)hs, (uint4*)hd, hog_bytes_per_wg / 16, stop, moved);
          CK(hipEventRecord(h1, s2));
          usleep(3000);                                          // the hog is resident before the first GEMM workgroup looks for a CU
        }
        CK(hipEventRecord(e0, s1));
        for (int r = 0; r < reps; ++r) launch(w.id);
        CK(hipEventRecord(e1, s1));
        CK(hipEventSynchronize(e1));
        *stop = 1;
        CK(hipStreamSynchronize(s2));
        float ms = 0, hms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        double gbs = 0;
        if (k) { CK(hipEventElapsedTime(&hms, h0, h1)); unsigned long long mv; CK(hipMemcpy(&mv, moved, 8, hipMemcpyDeviceToHost)); gbs = mv / (hms * 1e6); }
        printf("%s {\"hog_cus\": %d, \"compute_cu_budget\": %d, \"kernel\": \"%s\", \"ms_per_launch\": %.4f, \"hog_GBps\": %.1f}", first ? " " : ",\n ", k, budget ? budget : 256,
               w.name, ms / reps, gbs);
        first = false;
      }
    }
  }
  printf("\n]\n");
  SX(simx_set_compute_cus(0));
  return 0;
}
