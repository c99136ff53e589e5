// Generate job: exhaustive inner-product search over a corpus shard resident in HBM -- the role of
// faiss.IndexFlatIP + index_cpu_to_all_gpus in SimANS/co_training/co_training_generate.py:359-384, 415-421.
//
//   ip_scores_kernel   S[nq, chunk] = Q . C^T in float32, ONE fused-multiply-add chain per score in ascending h
//                      (bit-reproducible: oracle/topk_ref.c restates exactly this), 128x128x16 register-tiled
//                      VALU kernel (the f32 MFMA has the same peak as packed f32 FMA on gfx950 and no defined
//                      accumulation order).  Bound: f32 FMA, 2*nq*nc*H FLOP.
//   topk_stream_kernel per query: streaming exact top-k over [running best (k) U new candidates] with 64-bit
//                      composite keys (orderable score << 32 | ~id): candidates below the running k-th key are
//                      dropped while scanning, survivors are collected in LDS and folded in by a bitonic sort of
//                      4096 keys whenever the buffer could overflow.  Ties: lower id first.  HBM-bound: reads the
//                      score row once (4 B per candidate).
#include <math.h>
#include "common.h"
#include "prof.h"

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

#define IP_BM 128
#define IP_BN 128
#define IP_BK 16

__global__ __launch_bounds__(256) void ip_scores_kernel(int nq, int nc, int H, const float* __restrict__ Q,
                                                        const float* __restrict__ Cp, float* __restrict__ S, long ldS) {
  __shared__ __attribute__((aligned(16))) float As[IP_BK][IP_BM + 4];
  __shared__ __attribute__((aligned(16))) float Bs[IP_BK][IP_BN + 4];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * IP_BM, n0 = blockIdx.x * IP_BN;
  float acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < H; k0 += IP_BK) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int idx = tid + e * 256, row = idx >> 2, kq = (idx & 3) * 4;
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
      if (k0 + kq < H) {
        if (m0 + row < nq) a = *reinterpret_cast<const float4*>(Q + (long)(m0 + row) * H + k0 + kq);
        if (n0 + row < nc) b = *reinterpret_cast<const float4*>(Cp + (long)(n0 + row) * H + k0 + kq);
      }
      As[kq + 0][row] = a.x; As[kq + 1][row] = a.y; As[kq + 2][row] = a.z; As[kq + 3][row] = a.w;
      Bs[kq + 0][row] = b.x; Bs[kq + 1][row] = b.y; Bs[kq + 2][row] = b.z; Bs[kq + 3][row] = b.w;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < IP_BK; ++kk) {
      const float4 a0 = *reinterpret_cast<const float4*>(&As[kk][ty * 4]), a1 = *reinterpret_cast<const float4*>(&As[kk][64 + ty * 4]);
      const float4 b0 = *reinterpret_cast<const float4*>(&Bs[kk][tx * 4]), b1 = *reinterpret_cast<const float4*>(&Bs[kk][64 + tx * 4]);
      const float a[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + (i < 4 ? ty * 4 + i : 64 + ty * 4 + (i - 4));
    if (m >= nq) continue;
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      const int n = n0 + jh * 64 + tx * 4;
      float* dst = S + (long)m * ldS + n;
      if (n + 4 <= nc && (ldS & 3) == 0) {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[i][jh * 4 + 0], acc[i][jh * 4 + 1], acc[i][jh * 4 + 2], acc[i][jh * 4 + 3]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (n + j < nc) dst[j] = acc[i][jh * 4 + j];
      }
    }
  }
}

// ---- top-k ------------------------------------------------------------------------------------------------------
#define TK_N 4096            // keys sorted per flush (32 KB of LDS)
#define TK_KMAX 1024
#define TK_SEG 2048          // candidates scanned between overflow checks

__device__ __forceinline__ uint64_t tk_key(float s, long long id) {
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)id);
}
__device__ __forceinline__ void tk_unkey(uint64_t key, float* s, long long* id) {
  if (key == 0) { *s = -INFINITY; *id = -1; return; }
  uint32_t u = (uint32_t)(key >> 32);
  u = (u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u;
  *s = __uint_as_float(u);
  *id = (long long)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu));
}

// descending bitonic sort of arr[0..TK_N) (256 threads)
__device__ void tk_sort(uint64_t* arr) {
  for (int k = 2; k <= TK_N; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < TK_N / 2; t += 256) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));      // lower index of the pair
        const int p = i | j;
        const bool desc = (i & k) == 0;
        const uint64_t x = arr[i], y = arr[p];
        if ((x < y) == desc) { arr[i] = y; arr[p] = x; }
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(256) void topk_stream_kernel(int m, const float* __restrict__ scores, long ld,
                                                          const long long* __restrict__ ids, long ld_ids, long long id_base,
                                                          int k, float* __restrict__ run_s, long long* __restrict__ run_i) {
  __shared__ uint64_t arr[TK_N];
  __shared__ int cnt;
  const int q = blockIdx.x, tid = threadIdx.x;
  const float* row = scores + (long)q * ld;
  const long long* idr = ids ? ids + (long)q * ld_ids : nullptr;
  for (int t = tid; t < TK_N; t += 256) {
    uint64_t key = 0;
    if (t < k) { const long long id = run_i[(long)q * k + t]; if (id >= 0) key = tk_key(run_s[(long)q * k + t], id); }
    arr[t] = key;
  }
  if (tid == 0) cnt = 0;
  __syncthreads();
  const int cap = TK_N - k;
  for (int s0 = 0; s0 < m; s0 += TK_SEG) {
    if (cnt + TK_SEG > cap) {                      // uniform: fold the buffer into the best-k first
      __syncthreads();
      tk_sort(arr);
      for (int t = k + tid; t < TK_N; t += 256) arr[t] = 0;
      if (tid == 0) cnt = 0;
      __syncthreads();
    }
    const uint64_t tau = arr[k - 1];               // running k-th key (0 while fewer than k are known)
    __syncthreads();
    const int s1 = min(m, s0 + TK_SEG);
    for (int j = s0 + tid; j < s1; j += 256) {
      const long long id = idr ? idr[j] : id_base + j;
      if (id < 0) continue;
      const uint64_t key = tk_key(row[j], id);
      if (key > tau) arr[k + atomicAdd(&cnt, 1)] = key;
    }
    __syncthreads();
  }
  tk_sort(arr);
  for (int t = tid; t < k; t += 256) {
    float s; long long id;
    tk_unkey(arr[t], &s, &id);
    run_s[(long)q * k + t] = s;
    run_i[(long)q * k + t] = id;
  }
}

extern "C" int simx_ip_scores(simx_stream_t stream, int nq, int nc, int H, const float* q, const float* c, float* scores, long ld) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_TOPK, s, 2.0 * nq * nc * H);
  SIMX_REQUIRE(nq > 0 && nc > 0 && H > 0 && H % 4 == 0 && ld >= nc, SIMX_ERR_BAD_SHAPE, "ip_scores: bad shape nq=%d nc=%d H=%d ld=%ld", nq, nc, H, ld);
  SIMX_REQUIRE(aligned16(q) && aligned16(c) && aligned16(scores), SIMX_ERR_BAD_SHAPE, "ip_scores: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(ip_scores_kernel, dim3(cdiv(nc, IP_BN), cdiv(nq, IP_BM)), dim3(256), 0, s, nq, nc, H, q, c, scores, ld);
  SIMX_CHECK_LAUNCH("ip_scores");
  return SIMX_OK;
}

extern "C" int simx_topk_update(simx_stream_t stream, int nq, int m, const float* scores, long ld, const int64_t* ids, long ld_ids,
                                int64_t id_base, int k, float* run_scores, int64_t* run_ids) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_REQUIRE(nq > 0 && m >= 0 && k > 0 && k <= TK_KMAX, SIMX_ERR_BAD_SHAPE, "topk_update: need 0 < k=%d <= %d", k, TK_KMAX);
  SIMX_REQUIRE(scores && run_scores && run_ids && ld >= m, SIMX_ERR_BAD_SHAPE, "topk_update: NULL argument / ld < m");
  SIMX_REQUIRE(ids != nullptr || (id_base >= 0 && id_base + m < 0xFFFFFFFFll), SIMX_ERR_BAD_SHAPE, "topk_update: ids must fit 32 bits");
  hipLaunchKernelGGL(topk_stream_kernel, dim3(nq), dim3(256), 0, s, m, scores, ld, (const long long*)ids, ld_ids, (long long)id_base, k,
                     run_scores, (long long*)run_ids);
  SIMX_CHECK_LAUNCH("topk_update");
  return SIMX_OK;
}

__global__ void topk_init_kernel(long n, float* s, long long* i) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t < n) { s[t] = -INFINITY; i[t] = -1; }
}

extern "C" size_t simx_flat_ip_workspace_bytes(int nq, int chunk) { return (size_t)nq * (size_t)((chunk + 3) & ~3) * sizeof(float); }

extern "C" int simx_flat_ip_search(simx_stream_t stream, int nq, long nc, int H, const float* q, const float* corpus,
                                   int64_t id_base, int k, int chunk, void* workspace, size_t workspace_bytes,
                                   float* out_scores, int64_t* out_ids) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_REQUIRE(nq > 0 && nc >= 0 && chunk > 0, SIMX_ERR_BAD_SHAPE, "flat_ip_search: bad shape");
  SIMX_REQUIRE(workspace_bytes >= simx_flat_ip_workspace_bytes(nq, chunk), SIMX_ERR_WORKSPACE, "flat_ip_search: workspace too small");
  const long ld = (chunk + 3) & ~3;
  hipLaunchKernelGGL(topk_init_kernel, dim3(cdiv((int)((long)nq * k), 256)), dim3(256), 0, s, (long)nq * k, out_scores, (long long*)out_ids);
  SIMX_CHECK_LAUNCH("topk_init");
  for (long c0 = 0; c0 < nc; c0 += chunk) {
    const int m = (int)((nc - c0) < chunk ? (nc - c0) : chunk);
    int rc = simx_ip_scores(stream, nq, m, H, q, corpus + c0 * H, (float*)workspace, ld);
    if (rc) return rc;
    rc = simx_topk_update(stream, nq, m, (const float*)workspace, ld, nullptr, 0, id_base + c0, k, out_scores, out_ids);
    if (rc) return rc;
  }
  return SIMX_OK;
}
