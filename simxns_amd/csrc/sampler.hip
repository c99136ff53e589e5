// SimANS ambiguous-negative sampler on the GPU: one wavefront per query.
// Reference: SimANS/utils/MARCO_until_new.py:174-202 (Laplace weights, tau=3),
// util_wiki.py:609-639 / MARCO_until_Doc.py:110-148 (Gaussian a,b).  Scheme (bit-for-bit the
// oracle's scheme_draw): f64 weights -> wave prefix sum (lane l owns the contiguous chunk
// [l*K,(l+1)*K); serial local sums, Hillis-Steele scan of lane totals) -> N inverse-CDF lookups per
// round with Philox4x32-10 uniforms (bisect_right == count of cum <= x, clamped) -> dedupe ->
// zero the chosen weights -> repeat until >= N -> truncate in draw order.
#include "common.h"
#include "prof.h"

#define SAMP_MAXK 16   // candidates per lane -> C <= 1024

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double philox_uniform(uint64_t seed, uint32_t offset, uint32_t q, uint32_t rnd, uint32_t j) {
  uint32_t r[4];
  philox4x32_10(q, rnd, j >> 1, offset, (uint32_t)seed, (uint32_t)(seed >> 32), r);
  const uint32_t lo = (j & 1) ? r[2] : r[0], hi = (j & 1) ? r[3] : r[1];
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6)) * (1.0 / 9007199254740992.0);
}

__global__ __launch_bounds__(256) void simans_kernel(int nq, int C, int N, const double* __restrict__ scores,
                                                     const double* __restrict__ pos_score, int form, double a, double b,
                                                     double tau, uint64_t seed, uint32_t offset, int* __restrict__ neg_idx,
                                                     int* __restrict__ union_idx, int* __restrict__ union_cnt,
                                                     double* __restrict__ weights_out) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wv;
  if (q >= nq) return;
  const int K = (C + 63) / 64;
  const double sp = pos_score[q];
  int* negs = neg_idx + (long)q * N;
  int* uni = union_idx ? union_idx + (long)q * 2 * N : nullptr;
  if (sp == 0.0) {                                           // positive was not retrieved: last N candidates
    for (int j = lane; j < N; j += 64) { negs[j] = C - N + j; if (uni) uni[j] = C - N + j; }
    if (union_cnt && lane == 0) union_cnt[q] = N;
    if (weights_out) for (int i = lane; i < C; i += 64) weights_out[(long)q * C + i] = 0.0;
    return;
  }
  double w[SAMP_MAXK];
  uint32_t taken = 0;
#pragma unroll
  for (int k = 0; k < SAMP_MAXK; ++k) {
    const int i = lane * K + k;
    double v = 0.0;
    if (k < K && i < C) {
      const double sc = scores[(long)q * C + i];
      if (form == 0) v = exp(-fabs(sc - sp) * tau);
      else { const double d = sc - sp + b; v = exp(-(d * d) * a); }
      if (weights_out) weights_out[(long)q * C + i] = v;
    }
    w[k] = v;
  }
  int cnt = 0;
  for (int rnd = 0; rnd < 64 && cnt < N; ++rnd) {
    double inc[SAMP_MAXK];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < SAMP_MAXK; ++k) { if (k < K) acc = acc + w[k]; inc[k] = acc; }
    double scan = acc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const double o = __shfl_up(scan, off, 64);
      if (lane >= off) scan = scan + o;
    }
    double excl = __shfl_up(scan, 1, 64);
    if (lane == 0) excl = 0.0;
    const double total = __shfl(scan, 63, 64);
    if (!(total > 0.0)) break;
    for (int j = 0; j < N; ++j) {
      const double x = philox_uniform(seed, offset, (uint32_t)q, (uint32_t)rnd, (uint32_t)j) * total;
      int le = 0;
#pragma unroll
      for (int k = 0; k < SAMP_MAXK; ++k)
        if (k < K && lane * K + k < C) le += ((excl + inc[k]) <= x) ? 1 : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) le += __shfl_xor(le, o, 64);
      const int idx = le < C - 1 ? le : C - 1;
      const int owner = idx / K, kk = idx - owner * K;
      const int was = __shfl((int)((taken >> kk) & 1u), owner, 64);
      if (!was) {
        if (lane == owner) taken |= (1u << kk);
        if (lane == 0) { if (cnt < N) negs[cnt] = idx; if (uni && cnt < 2 * N) uni[cnt] = idx; }
        ++cnt;
      }
    }
#pragma unroll
    for (int k = 0; k < SAMP_MAXK; ++k) if ((taken >> k) & 1u) w[k] = 0.0;
  }
  // degenerate weights (total underflowed to 0): fill with the lowest free indices
  for (int i = 0; i < C && cnt < N; ++i) {
    const int owner = i / K, kk = i - owner * K;
    const int was = __shfl((int)((taken >> kk) & 1u), owner, 64);
    if (!was) {
      if (lane == owner) taken |= (1u << kk);
      if (lane == 0) { negs[cnt] = i; if (uni && cnt < 2 * N) uni[cnt] = i; }
      ++cnt;
    }
  }
  if (union_cnt && lane == 0) union_cnt[q] = cnt;
}

extern "C" int simx_simans_sample(simx_stream_t stream, int nq, int C, int N, const double* scores, const double* pos_score,
                                  int form, double a, double b, double tau, uint64_t seed, uint32_t offset,
                                  int32_t* neg_idx, int32_t* union_idx, int32_t* union_cnt, double* weights_out) {
  SIMX_PROF(SIMX_K_SAMPLER, stream, (double)nq * (8.0 * C + 8 + 4.0 * N));
  SIMX_REQUIRE(nq > 0 && N > 0 && C >= N, SIMX_ERR_BAD_SHAPE, "simans_sample: need nq>0 and C >= N > 0 (C=%d N=%d)", C, N);
  SIMX_REQUIRE(C <= 64 * SAMP_MAXK, SIMX_ERR_UNSUPPORTED, "simans_sample: C=%d > %d", C, 64 * SAMP_MAXK);
  SIMX_REQUIRE(form == 0 || form == 1, SIMX_ERR_UNSUPPORTED, "simans_sample: form %d", form);
  SIMX_REQUIRE(scores && pos_score && neg_idx, SIMX_ERR_BAD_SHAPE, "simans_sample: NULL pointer");
  hipLaunchKernelGGL(simans_kernel, dim3(cdiv(nq, 4)), dim3(256), 0, (hipStream_t)stream, nq, C, N, scores, pos_score, form, a, b,
                     tau, seed, offset, neg_idx, union_idx, union_cnt, weights_out);
  SIMX_CHECK_LAUNCH("simans_sample");
  return SIMX_OK;
}
