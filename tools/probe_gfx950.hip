// Hardware-semantics probe for gfx950 (not part of the product): verifies the lane
// layouts the kernels in simxns_amd/csrc assume.  Build: hipcc --offload-arch=gfx950 -O2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <cstring>
#include <vector>

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); }
static inline float bf2f(uint16_t h) { uint32_t u = ((uint32_t)h) << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void k_tr(const int* addr_elems, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  uint32_t a = (uint32_t)(uintptr_t)(&lds[0]) + addr_elems[threadIdx.x] * 2;
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (uint16_t)v[j];
}

__global__ void k_glds(const uint16_t* src, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = 0xFFFF;
  __syncthreads();
  // lane l fetches 16 B from src + perm(l)*8 elements ; expected to land at lds + 512 B + l*16 B
  int l = threadIdx.x;
  const uint16_t* g = src + ((l * 7) % 64) * 8;
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)(&lds[256]), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = lds[i];
}

__global__ void k_mfma_bf16(const uint16_t* A, const uint16_t* B, float* D) {
  // A [16][32] row-major, B [32][16] row-major (k-major), D [16][16]
  int l = threadIdx.x, i = l & 15, g = l >> 4;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (short)A[i * 32 + g * 8 + j]; b[j] = (short)B[(g * 8 + j) * 16 + i]; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = c[r];
}

__global__ void k_mfma_f32(const float* A, const float* B, float* D) {
  // A [16][4], B [4][16]
  int l = threadIdx.x, i = l & 15, g = l >> 4;
  float a = A[i * 4 + g], b = B[g * 16 + i];
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(g * 4 + r) * 16 + i] = c[r];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s  CUs %d  gcn %s  clock %d kHz  mem %.1f GB\n", p.name, p.multiProcessorCount, p.gcnArchName, p.clockRate, p.totalGlobalMem / 1e9);
  // ---- tr read, linear addresses
  int h_addr[64]; uint16_t h_out[1024]; int* d_addr; uint16_t* d_out;
  CK(hipMalloc(&d_addr, 256)); CK(hipMalloc(&d_out, 2048));
  for (int variant = 0; variant < 2; ++variant) {
    for (int l = 0; l < 64; ++l) h_addr[l] = variant == 0 ? l * 4 : ((l & 15) * 20 + (l >> 4) * 340);   // variant 1: row stride 20 elems
    CK(hipMemcpy(d_addr, h_addr, 256, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(d_addr, d_out); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_out, d_out, 512, hipMemcpyDeviceToHost));
    // hypothesis: within 16-lane group, out[lane c][j] = elem (c&3) at address of lane (4j + (c>>2)) of the same group
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
      int c = l & 15, grp = l >> 4; int srcl = grp * 16 + 4 * j + (c >> 2);
      int expect = h_addr[srcl] + (c & 3);
      if (h_out[l * 4 + j] != expect) ++bad;
    }
    printf("tr_b16 variant %d: hypothesis mismatches = %d\n", variant, bad);
    if (bad) { for (int l = 0; l < 64; ++l) printf(" lane %2d: %4d %4d %4d %4d\n", l, h_out[l*4], h_out[l*4+1], h_out[l*4+2], h_out[l*4+3]); }
  }
  // ---- global_load_lds
  { std::vector<uint16_t> src(512); for (int i = 0; i < 512; ++i) src[i] = (uint16_t)(1000 + i);
    uint16_t* d_src; CK(hipMalloc(&d_src, 1024)); CK(hipMemcpy(d_src, src.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 64>>>(d_src, d_out); CK(hipDeviceSynchronize());
    CK(hipMemcpy(h_out, d_out, 2048, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) if (h_out[256 + l * 8 + e] != 1000 + ((l * 7) % 64) * 8 + e) ++bad;
    for (int i = 0; i < 256; ++i) if (h_out[i] != 0xFFFF) ++bad;
    for (int i = 768; i < 1024; ++i) if (h_out[i] != 0xFFFF) ++bad;
    printf("global_load_lds x16: lane-linear hypothesis mismatches = %d\n", bad);
  }
  // ---- mfma bf16
  { std::vector<uint16_t> A(512), B(512); std::vector<float> D(256), R(256, 0.f);
    for (int i = 0; i < 512; ++i) { A[i] = f2bf((float)((i * 37 % 17) - 8) / 4.f); B[i] = f2bf((float)((i * 53 % 23) - 11) / 8.f); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 32; ++k) s += bf2f(A[i*32+k]) * bf2f(B[k*16+j]); R[i*16+j] = s; }
    uint16_t *dA, *dB; float* dD; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    k_mfma_bf16<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(D[i] - R[i]));
    printf("mfma_f32_16x16x32_bf16 layout: max err %.3e\n", e);
  }
  { std::vector<float> A(64), B(64), D(256), R(256, 0.f);
    for (int i = 0; i < 64; ++i) { A[i] = (float)((i * 37 % 17) - 8) / 4.f; B[i] = (float)((i * 53 % 23) - 11) / 8.f; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { float s = 0; for (int k = 0; k < 4; ++k) s += A[i*4+k] * B[k*16+j]; R[i*16+j] = s; }
    float *dA, *dB, *dD; CK(hipMalloc(&dA, 256)); CK(hipMalloc(&dB, 256)); CK(hipMalloc(&dD, 1024));
    CK(hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice));
    k_mfma_f32<<<1, 64>>>(dA, dB, dD); CK(hipDeviceSynchronize());
    CK(hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost));
    double e = 0; for (int i = 0; i < 256; ++i) e = fmax(e, fabs(D[i] - R[i]));
    printf("mfma_f32_16x16x4f32 layout: max err %.3e\n", e);
  }
  return 0;
}
