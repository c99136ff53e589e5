// f32 self-attention on the f32 matrix cores (fp32 parity mode, head size 64, sequences <= 256 tokens): BertSelfAttention
// core, LEAD/modeling_bert.py:318-374, forward and backward on the packed token layout.  v_mfma_f32_32x32x2_f32: exact f32
// products, f32 accumulation.  Replaces the one-wave-per-row kernels (attention.hip, "generic") that took 42 % of the fp32
// step; those stay for other head sizes and longer sequences.
//
// One workgroup (4 waves) per (sequence, head); the operands a phase contracts against stay in LDS in their natural
// [token][64] layout with a row pitch of 65 floats, which serves both read patterns the MFMA needs:
//   "T" reads (scores): A[m = token = lane%32][k = d = t + 32*(lane/32)]  -- lanes walk tokens, stride 65: conflict-free;
//   "N" reads (P.V, dS.K, ...): A[m = d = lane%32][k = token(e, lane/32)] -- lanes walk d: consecutive words.
// The second operand of every product comes from REGISTERS:
//   * scores are produced transposed, S^T[key][query] = K . Q^T, with the wave's 32 query rows held as 32 floats per lane
//     (row lane%32, columns 32*(lane/32) .. +31: the contraction pairs d = t with d = t + 32 in one k-step, for both
//     operands), so a query's softmax row lives in one lane column: the reduction is over registers plus one exchange with
//     lane ^ 32;
//   * the MFMA result layout (lane = column, acc[e] = row 8*(e/4) + 4*(lane/32) + e%4) is exactly the B-operand layout of a
//     k-step that pairs row(e, 0) with row(e, 1), so P / dS feed the second product straight from the accumulators.
// Keys are restricted to the sequence's own tokens (== the reference's additive finfo.min mask); dropout masks are the
// stateless hash of common.h, keyed as in attention.hip (row = head*T + query token, column = key index in the sequence).
#include "common.h"
#include "prof.h"

#define AF_PITCH 65
#define AF_LOG2E 1.4426950408889634f

__device__ __forceinline__ int af_row(int e, int half) { return 8 * (e >> 2) + 4 * half + (e & 3); }

// stage rows [0, npad) x 64 floats of one head slice into LDS (pitch 65); rows >= len are clamped copies of row len-1
__device__ __forceinline__ void af_stage(const float* __restrict__ G, long ld, int len, int npad, float* __restrict__ S, int tid) {
  for (int idx = tid; idx < npad * 16; idx += 256) {
    const int r = idx >> 4, c4 = (idx & 15) * 4;
    const int gr = r < len ? r : len - 1;
    const float4 v = *reinterpret_cast<const float4*>(G + (long)gr * ld + c4);
    float* d = S + r * AF_PITCH + c4;
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}
// dense form for the forward at 160 tokens (5 key tiles): pitch 64, the K tile XOR-swizzled (column c of row r at c ^ (r & 31):
// the 32 lanes of a "T" read walk 32 rows at one logical column and land on 32 banks; "N" reads of V walk columns and need
// nothing).  K + V are then exactly 80 KB and TWO workgroups fit a CU -- with the padded pitch it was one, and its fifth query
// tile ran on one wave while three idled: 3.6 ms per launch against 1.53 for 128 tokens.
__device__ __forceinline__ void af_stage_dense(const float* __restrict__ G, long ld, int len, int npad, float* __restrict__ S, int tid, bool swz) {
  for (int idx = tid; idx < npad * 16; idx += 256) {
    const int r = idx >> 4, c4 = (idx & 15) * 4;
    const int gr = r < len ? r : len - 1;
    const float4 v = *reinterpret_cast<const float4*>(G + (long)gr * ld + c4);
    float* d = S + r * 64;
    const int x = swz ? (r & 31) : 0;
    d[(c4 + 0) ^ x] = v.x; d[(c4 + 1) ^ x] = v.y; d[(c4 + 2) ^ x] = v.z; d[(c4 + 3) ^ x] = v.w;
  }
}
// a lane's half row (32 floats: columns 32*half .. +31 of row `row`) of a [.., ld] matrix, into registers
__device__ __forceinline__ void af_row_regs(const float* __restrict__ G, long ld, int row, int half, float (&r)[32]) {
  const float4* p = reinterpret_cast<const float4*>(G + (long)row * ld + 32 * half);
#pragma unroll
  for (int i = 0; i < 8; ++i) { const float4 v = p[i]; r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w; }
}
// store a transposed accumulator pair (O^T[d][row]: lane = row column, acc[e] = d row) as rows of a [.., ld] matrix
__device__ __forceinline__ void af_store_t(const f32x16 (&o)[2], float mul, float* __restrict__ dst, int half) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<float4*>(dst + dt * 32 + 8 * g + 4 * half) =
          make_float4(o[dt][4 * g] * mul, o[dt][4 * g + 1] * mul, o[dt][4 * g + 2] * mul, o[dt][4 * g + 3] * mul);
}

// "operand planes" (csrc/gemm_xp.hip): the same store with every value split into a 16-bit pair, hi at dst, lo at dst + ps
template <typename F>
__device__ __forceinline__ void af_store_t_planes(const f32x16 (&o)[2], float mul, bf16_t* __restrict__ dst, long ps, int half) {
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float a = f32_pin(o[dt][4 * g] * mul), b = f32_pin(o[dt][4 * g + 1] * mul), c = f32_pin(o[dt][4 * g + 2] * mul), d = f32_pin(o[dt][4 * g + 3] * mul);
      const uint32_t h0 = H16<F>::pack2(a, b), h1 = H16<F>::pack2(c, d);
      const uint32_t l0 = H16<F>::pack2(a - H16<F>::lo(h0), b - H16<F>::hi(h0)), l1 = H16<F>::pack2(c - H16<F>::lo(h1), d - H16<F>::hi(h1));
      bf16_t* q = dst + dt * 32 + 8 * g + 4 * half;
      *reinterpret_cast<uint2*>(q) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(q + ps) = make_uint2(l0, l1);
    }
}
// 4 consecutive values hi + lo of an fp16 plane pair
__device__ __forceinline__ float4 af_ld4_planes(const bf16_t* __restrict__ p, long ps) {
  const uint2 h = *reinterpret_cast<const uint2*>(p), l = *reinterpret_cast<const uint2*>(p + ps);
  return make_float4(H16<f16_t>::lo(h.x) + H16<f16_t>::lo(l.x), H16<f16_t>::hi(h.x) + H16<f16_t>::hi(l.x),
                     H16<f16_t>::lo(h.y) + H16<f16_t>::lo(l.y), H16<f16_t>::hi(h.y) + H16<f16_t>::hi(l.y));
}

// ------------------------------------------------------------------------------------------ forward
// PL: ctx leaves as the fp16 plane pair the attention-output GEMM stages (ctx = the pair, cps = its plane stride)
template <int NKT, bool DENSE = false, bool PL = false>
__global__ __launch_bounds__(256, DENSE ? 2 : 1) void mha_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, float* __restrict__ lse,
                                                          const int* __restrict__ cu, int heads, int T, float scale, DropCtx drop, long cps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int PITCH = DENSE ? 64 : AF_PITCH;
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + NKT * 32 * PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const float* Qg = qkv + (long)t0 * H3 + h * 64;
  const int nkt = (len + 31) >> 5;
  if (DENSE) {
    af_stage_dense(Qg + H, H3, len, nkt * 32, sK, tid, true);
    af_stage_dense(Qg + 2 * H, H3, len, nkt * 32, sV, tid, false);
  } else {
    af_stage(Qg + H, H3, len, nkt * 32, sK, tid);
    af_stage(Qg + 2 * H, H3, len, nkt * 32, sV, tid);
  }
  __syncthreads();
  const int col = lane & 31, half = lane >> 5;
  const float c2 = scale * AF_LOG2E;
  for (int qt = wave; qt < nkt; qt += 4) {
    const int q = qt * 32 + col;
    const int qc = q < len ? q : len - 1;
    float qr[32];
    af_row_regs(Qg, H3, qc, half, qr);
    f32x16 s[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
      if (kt < nkt) {
        const float* ak = sK + (kt * 32 + col) * PITCH + 32 * half;
#pragma unroll
        for (int t = 0; t < 32; ++t) s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[DENSE ? (t ^ col) : t], qr[t], s[kt], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * 32 + af_row(e, half);
          s[kt][e] = key < len ? s[kt][e] : -INFINITY;
          m = fmaxf(m, s[kt][e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[kt][e] = -INFINITY;
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f((s[kt][e] - m) * c2);
        s[kt][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    if (drop.thr) {                               // dropout on the probabilities (the normaliser stays unmasked)
      const uint32_t drow = (uint32_t)(h * T + t0 + q);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[kt][e] *= drop_mult(drop, drow, (uint32_t)(kt * 32 + af_row(e, half)));
    }
    f32x16 o[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; }
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float* av = sV + (kt * 32 + af_row(e, half)) * PITCH + col;
          o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], s[kt][e], o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[32], s[kt][e], o[1], 0, 0, 0);
        }
      }
    if (q < len) {
      if (PL) af_store_t_planes<f16_t>(o, 1.0f / sum, reinterpret_cast<bf16_t*>(ctx) + (long)(t0 + q) * H + h * 64, cps, half);
      else af_store_t(o, 1.0f / sum, ctx + (long)(t0 + q) * H + h * 64, half);
      if (half == 0) lse[(long)h * T + t0 + q] = m * scale + logf(sum);
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: dQ
// K, V resident; a wave owns 32 query rows (Q, dO, O half rows in registers) and walks the key tiles.
// PL: O is the forward's fp16 plane pair (ops = its plane stride) and dq / dk / dv leave as the bf16 plane pair the dgrad and
// wgrad GEMMs stage (dqkv = the pair, dps = its plane stride)
template <int NKT, bool PL = false>
__global__ __launch_bounds__(256) void mha_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ O,
                                                             const float* __restrict__ lse, const float* __restrict__ dO,
                                                             float* __restrict__ dqkv, const int* __restrict__ cu, int heads, int T,
                                                             float scale, DropCtx drop, long ops, long dps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + NKT * 32 * AF_PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const float* Qg = qkv + (long)t0 * H3 + h * 64;
  const float* Og = O + (long)t0 * H + h * 64;
  const bf16_t* Opl = reinterpret_cast<const bf16_t*>(O) + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int nkt = (len + 31) >> 5;
  af_stage(Qg + H, H3, len, nkt * 32, sK, tid);
  af_stage(Qg + 2 * H, H3, len, nkt * 32, sV, tid);
  __syncthreads();
  const int col = lane & 31, half = lane >> 5;
  const float c2 = scale * AF_LOG2E;
  for (int qt = wave; qt < nkt; qt += 4) {
    const int q = qt * 32 + col;
    const int qc = q < len ? q : len - 1;
    float qr[32], dr[32];
    af_row_regs(Qg, H3, qc, half, qr);
    af_row_regs(dOg, H, qc, half, dr);
    float delta = 0.f;
    {
      const float4* po = reinterpret_cast<const float4*>(Og + (long)qc * H + 32 * half);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 v = PL ? af_ld4_planes(Opl + (long)qc * H + 32 * half + 4 * i, ops) : po[i];
        delta += dr[4 * i] * v.x + dr[4 * i + 1] * v.y + dr[4 * i + 2] * v.z + dr[4 * i + 3] * v.w;
      }
    }
    delta += __shfl_xor(delta, 32, 64);
    const float lq = lse[(long)h * T + t0 + qc] * AF_LOG2E;
    const bool qok = q < len;
    f32x16 dq[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dq[0][e] = 0.f; dq[1][e] = 0.f; }
    for (int kt = 0; kt < nkt; ++kt) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
      const float* ak = sK + (kt * 32 + col) * AF_PITCH + 32 * half;
      const float* av = sV + (kt * 32 + col) * AF_PITCH + 32 * half;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[t], qr[t], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], dr[t], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt * 32 + af_row(e, half);
        float p = __builtin_amdgcn_exp2f(s[e] * c2 - lq);
        p = (key < len && qok) ? p : 0.f;
        const float mm = drop.thr ? drop_mult(drop, (uint32_t)(h * T + t0 + q), (uint32_t)key) : 1.f;
        s[e] = p * (dp[e] * mm - delta) * scale;            // dS^T[key][q]
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float* an = sK + (kt * 32 + af_row(e, half)) * AF_PITCH + col;
        dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[0], s[e], dq[0], 0, 0, 0);
        dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[32], s[e], dq[1], 0, 0, 0);
      }
    }
    if (qok) {
      if (PL) af_store_t_planes<bf16_t>(dq, 1.0f, reinterpret_cast<bf16_t*>(dqkv) + (long)(t0 + q) * H3 + h * 64, dps, half);
      else af_store_t(dq, 1.0f, dqkv + (long)(t0 + q) * H3 + h * 64, half);
    }
  }
}

// ------------------------------------------------------------------------------------------ backward: dK, dV
// Q, dO (and lse, delta) resident; a wave owns 32 key rows (K, V half rows in registers) and walks the query tiles.
template <int NKT, bool PL = false>
__global__ __launch_bounds__(256) void mha_bwd_dkv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ O,
                                                              const float* __restrict__ lse, const float* __restrict__ dO,
                                                              float* __restrict__ dqkv, const int* __restrict__ cu, int heads, int T,
                                                              float scale, DropCtx drop, long ops, long dps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sQ = reinterpret_cast<float*>(smem);
  float* sD = sQ + NKT * 32 * AF_PITCH;
  float* sLse = sD + NKT * 32 * AF_PITCH;
  float* sDel = sLse + NKT * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const float* Qg = qkv + (long)t0 * H3 + h * 64;
  const float* Og = O + (long)t0 * H + h * 64;
  const bf16_t* Opl = reinterpret_cast<const bf16_t*>(O) + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int nkt = (len + 31) >> 5;
  af_stage(Qg, H3, len, nkt * 32, sQ, tid);
  af_stage(dOg, H, len, nkt * 32, sD, tid);
  for (int r = tid; r < nkt * 32; r += 256) {
    float del = 0.f, l = 0.f;
    if (r < len) {
      const float4* po = reinterpret_cast<const float4*>(Og + (long)r * H);
      const float4* pd = reinterpret_cast<const float4*>(dOg + (long)r * H);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 a = PL ? af_ld4_planes(Opl + (long)r * H + 4 * i, ops) : po[i], b = pd[i];
        del += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
      }
      l = lse[(long)h * T + t0 + r] * AF_LOG2E;
    }
    sDel[r] = del;
    sLse[r] = l;
  }
  __syncthreads();
  const int col = lane & 31, half = lane >> 5;
  const float c2 = scale * AF_LOG2E;
  for (int kt = wave; kt < nkt; kt += 4) {
    const int key = kt * 32 + col;
    const int kc = key < len ? key : len - 1;
    float kr[32], vr[32];
    af_row_regs(Qg + H, H3, kc, half, kr);
    af_row_regs(Qg + 2 * H, H3, kc, half, vr);
    const bool kok = key < len;
    f32x16 dk[2], dv[2];
#pragma unroll
    for (int e = 0; e < 16; ++e) { dk[0][e] = 0.f; dk[1][e] = 0.f; dv[0][e] = 0.f; dv[1][e] = 0.f; }
    for (int qt = 0; qt < nkt; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
      const float* aq = sQ + (qt * 32 + col) * AF_PITCH + 32 * half;
      const float* ad = sD + (qt * 32 + col) * AF_PITCH + 32 * half;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t], kr[t], s, 0, 0, 0);        // S[q][key]: lane = key column, acc = query rows
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[t], vr[t], dp, 0, 0, 0);      // dP[q][key]
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qr_ = qt * 32 + af_row(e, half);
        float p = __builtin_amdgcn_exp2f(s[e] * c2 - sLse[qr_]);
        p = (qr_ < len && kok) ? p : 0.f;
        const float mm = drop.thr ? drop_mult(drop, (uint32_t)(h * T + t0 + qr_), (uint32_t)key) : 1.f;
        s[e] = p * mm;                                                              // P~[q][key]
        dp[e] = p * (dp[e] * mm - sDel[qr_]) * scale;                               // dS[q][key]
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qr_ = qt * 32 + af_row(e, half);
        const float* dn = sD + qr_ * AF_PITCH + col;
        const float* qn = sQ + qr_ * AF_PITCH + col;
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dn[0], s[e], dv[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dn[32], s[e], dv[1], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qn[0], dp[e], dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qn[32], dp[e], dk[1], 0, 0, 0);
      }
    }
    if (kok) {
      if (PL) {
        bf16_t* dst = reinterpret_cast<bf16_t*>(dqkv) + (long)(t0 + key) * H3 + H + h * 64;
        af_store_t_planes<bf16_t>(dk, 1.0f, dst, dps, half);
        af_store_t_planes<bf16_t>(dv, 1.0f, dst + H, dps, half);
      } else {
        float* dst = dqkv + (long)(t0 + key) * H3 + H + h * 64;
        af_store_t(dk, 1.0f, dst, half);
        af_store_t(dv, 1.0f, dst + H, half);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ long sequences (256 < S <= 4096)
// MS-MARCO Document (BASELINE configs[4]: 512-token documents) in the fp32 arithmetic its recipe selects.  The resident-K/V
// kernels above need the whole sequence in LDS; here the sequence is cut into 128-token chunks:
//   forward : one workgroup per (sequence, head, 128-query chunk), each wave owns ONE 32-query tile (Q half rows, the running
//             maximum / normaliser and the output accumulators stay in registers) and the workgroup walks the key chunks --
//             K and V of a chunk staged in LDS, online softmax (the accumulators are rescaled when the running maximum moves);
//   dQ      : the same grid and walk with the forward's lse, dq accumulates in registers;
//   dK, dV  : one workgroup per (sequence, head, 128-key chunk), each wave owns one 32-key tile and the workgroup walks the
//             query chunks (Q, dO, lse, rowsum(dO . O) of a chunk staged in LDS).
// No atomics, no f32 scratch in HBM, every wave takes part in every barrier (tiles past the sequence end compute on clamped
// rows and store nothing).  The one-wave-per-row kernels these replace took 5.2 of the 5.8 s of a BERT-large S = 512 step.
#define AFL_NKT 4                              // 32-token tiles per chunk
template <bool PL>
__global__ __launch_bounds__(256, 2) void mha_fwd_f32_long_kernel(const float* __restrict__ qkv, float* __restrict__ ctx, float* __restrict__ lse,
                                                                  const int* __restrict__ cu, int heads, int T, int nchunk, float scale,
                                                                  DropCtx drop, long cps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + AFL_NKT * 32 * AF_PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qc = blockIdx.x % nchunk, sh = blockIdx.x / nchunk;
  const int seq = sh / heads, h = sh % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0 || qc * 128 >= len) return;       // (uniform per workgroup)
  const int H = heads * 64;
  const long H3 = 3L * H;
  const float* Qg = qkv + (long)t0 * H3 + h * 64;
  const int col = lane & 31, half = lane >> 5;
  const float c2 = scale * AF_LOG2E;
  const int q = qc * 128 + wave * 32 + col;
  const int qcl = q < len ? q : len - 1;
  float qr[32];
  af_row_regs(Qg, H3, qcl, half, qr);
  float m_run = -INFINITY, l_run = 0.f;
  f32x16 o[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { o[0][e] = 0.f; o[1][e] = 0.f; }
  const int nkc = (len + 127) >> 7;
  for (int kc = 0; kc < nkc; ++kc) {
    const int k0 = kc * 128, klen = min(128, len - k0), nkt = (klen + 31) >> 5;
    if (kc) __syncthreads();                     // every wave is done with the previous chunk
    af_stage(Qg + H + (long)k0 * H3, H3, klen, nkt * 32, sK, tid);
    af_stage(Qg + 2 * H + (long)k0 * H3, H3, klen, nkt * 32, sV, tid);
    __syncthreads();
    f32x16 s[AFL_NKT];
    float m = m_run;
#pragma unroll
    for (int kt = 0; kt < AFL_NKT; ++kt) {
#pragma unroll
      for (int e = 0; e < 16; ++e) s[kt][e] = 0.f;
      if (kt < nkt) {
        const float* ak = sK + (kt * 32 + col) * AF_PITCH + 32 * half;
#pragma unroll
        for (int t = 0; t < 32; ++t) s[kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[t], qr[t], s[kt], 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int key = kt * 32 + af_row(e, half);
          s[kt][e] = key < klen ? s[kt][e] : -INFINITY;
          m = fmaxf(m, s[kt][e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[kt][e] = -INFINITY;
      }
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));         // (finite: every chunk holds at least one real key)
    const float alpha = __builtin_amdgcn_exp2f((m_run - m) * c2);     // exp2(-inf) = 0 on the first chunk
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < AFL_NKT; ++kt)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float p = __builtin_amdgcn_exp2f((s[kt][e] - m) * c2);
        s[kt][e] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 32, 64);
    l_run = l_run * alpha + sum;
    m_run = m;
#pragma unroll
    for (int e = 0; e < 16; ++e) { o[0][e] *= alpha; o[1][e] *= alpha; }
    if (drop.thr) {                               // dropout on the probabilities (the normaliser stays unmasked)
      const uint32_t drow = (uint32_t)(h * T + t0 + q);
#pragma unroll
      for (int kt = 0; kt < AFL_NKT; ++kt)
        if (kt < nkt)
#pragma unroll
          for (int e = 0; e < 16; ++e) s[kt][e] *= drop_mult(drop, drow, (uint32_t)(k0 + kt * 32 + af_row(e, half)));
    }
#pragma unroll
    for (int kt = 0; kt < AFL_NKT; ++kt)
      if (kt < nkt) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float* av = sV + (kt * 32 + af_row(e, half)) * AF_PITCH + col;
          o[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], s[kt][e], o[0], 0, 0, 0);
          o[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[32], s[kt][e], o[1], 0, 0, 0);
        }
      }
  }
  if (q < len) {
    if (PL) af_store_t_planes<f16_t>(o, 1.0f / l_run, reinterpret_cast<bf16_t*>(ctx) + (long)(t0 + q) * H + h * 64, cps, half);
    else af_store_t(o, 1.0f / l_run, ctx + (long)(t0 + q) * H + h * 64, half);
    if (half == 0) lse[(long)h * T + t0 + q] = m_run * scale + logf(l_run);
  }
}

template <bool PL>
__global__ __launch_bounds__(256, 2) void mha_bwd_dq_f32_long_kernel(const float* __restrict__ qkv, const float* __restrict__ O,
                                                                     const float* __restrict__ lse, const float* __restrict__ dO,
                                                                     float* __restrict__ dqkv, const int* __restrict__ cu, int heads, int T,
                                                                     int nchunk, float scale, DropCtx drop, long ops, long dps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sK = reinterpret_cast<float*>(smem);
  float* sV = sK + AFL_NKT * 32 * AF_PITCH;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qc = blockIdx.x % nchunk, sh = blockIdx.x / nchunk;
  const int seq = sh / heads, h = sh % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0 || qc * 128 >= len) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const float* Qg = qkv + (long)t0 * H3 + h * 64;
  const float* Og = O + (long)t0 * H + h * 64;
  const bf16_t* Opl = reinterpret_cast<const bf16_t*>(O) + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int col = lane & 31, half = lane >> 5;
  const float c2 = scale * AF_LOG2E;
  const int q = qc * 128 + wave * 32 + col;
  const int qcl = q < len ? q : len - 1;
  const bool qok = q < len;
  float qr[32], dr[32];
  af_row_regs(Qg, H3, qcl, half, qr);
  af_row_regs(dOg, H, qcl, half, dr);
  float delta = 0.f;
  {
    const float4* po = reinterpret_cast<const float4*>(Og + (long)qcl * H + 32 * half);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float4 v = PL ? af_ld4_planes(Opl + (long)qcl * H + 32 * half + 4 * i, ops) : po[i];
      delta += dr[4 * i] * v.x + dr[4 * i + 1] * v.y + dr[4 * i + 2] * v.z + dr[4 * i + 3] * v.w;
    }
  }
  delta += __shfl_xor(delta, 32, 64);
  const float lq = lse[(long)h * T + t0 + qcl] * AF_LOG2E;
  f32x16 dq[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { dq[0][e] = 0.f; dq[1][e] = 0.f; }
  const int nkc = (len + 127) >> 7;
  for (int kc = 0; kc < nkc; ++kc) {
    const int k0 = kc * 128, klen = min(128, len - k0), nkt = (klen + 31) >> 5;
    if (kc) __syncthreads();
    af_stage(Qg + H + (long)k0 * H3, H3, klen, nkt * 32, sK, tid);
    af_stage(Qg + 2 * H + (long)k0 * H3, H3, klen, nkt * 32, sV, tid);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
      const float* ak = sK + (kt * 32 + col) * AF_PITCH + 32 * half;
      const float* av = sV + (kt * 32 + col) * AF_PITCH + 32 * half;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(ak[t], qr[t], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], dr[t], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int key = kt * 32 + af_row(e, half);
        float p = __builtin_amdgcn_exp2f(s[e] * c2 - lq);
        p = (key < klen && qok) ? p : 0.f;
        const float mm = drop.thr ? drop_mult(drop, (uint32_t)(h * T + t0 + q), (uint32_t)(k0 + key)) : 1.f;
        s[e] = p * (dp[e] * mm - delta) * scale;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float* an = sK + (kt * 32 + af_row(e, half)) * AF_PITCH + col;
        dq[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[0], s[e], dq[0], 0, 0, 0);
        dq[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[32], s[e], dq[1], 0, 0, 0);
      }
    }
  }
  if (qok) {
    if (PL) af_store_t_planes<bf16_t>(dq, 1.0f, reinterpret_cast<bf16_t*>(dqkv) + (long)(t0 + q) * H3 + h * 64, dps, half);
    else af_store_t(dq, 1.0f, dqkv + (long)(t0 + q) * H3 + h * 64, half);
  }
}

template <bool PL>
__global__ __launch_bounds__(256, 2) void mha_bwd_dkv_f32_long_kernel(const float* __restrict__ qkv, const float* __restrict__ O,
                                                                      const float* __restrict__ lse, const float* __restrict__ dO,
                                                                      float* __restrict__ dqkv, const int* __restrict__ cu, int heads, int T,
                                                                      int nchunk, float scale, DropCtx drop, long ops, long dps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sQ = reinterpret_cast<float*>(smem);
  float* sD = sQ + AFL_NKT * 32 * AF_PITCH;
  float* sLse = sD + AFL_NKT * 32 * AF_PITCH;
  float* sDel = sLse + AFL_NKT * 32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kc = blockIdx.x % nchunk, sh = blockIdx.x / nchunk;
  const int seq = sh / heads, h = sh % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0 || kc * 128 >= len) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const float* Qg = qkv + (long)t0 * H3 + h * 64;
  const float* Og = O + (long)t0 * H + h * 64;
  const bf16_t* Opl = reinterpret_cast<const bf16_t*>(O) + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int col = lane & 31, half = lane >> 5;
  const float c2 = scale * AF_LOG2E;
  const int key = kc * 128 + wave * 32 + col;
  const int kcl = key < len ? key : len - 1;
  const bool kok = key < len;
  float kr[32], vr[32];
  af_row_regs(Qg + H, H3, kcl, half, kr);
  af_row_regs(Qg + 2 * H, H3, kcl, half, vr);
  f32x16 dk[2], dv[2];
#pragma unroll
  for (int e = 0; e < 16; ++e) { dk[0][e] = 0.f; dk[1][e] = 0.f; dv[0][e] = 0.f; dv[1][e] = 0.f; }
  const int nqc = (len + 127) >> 7;
  for (int qc = 0; qc < nqc; ++qc) {
    const int q0 = qc * 128, qlen = min(128, len - q0), nqt = (qlen + 31) >> 5;
    if (qc) __syncthreads();
    af_stage(Qg + (long)q0 * H3, H3, qlen, nqt * 32, sQ, tid);
    af_stage(dOg + (long)q0 * H, H, qlen, nqt * 32, sD, tid);
    for (int r = tid; r < nqt * 32; r += 256) {
      float del = 0.f, l = 0.f;
      if (r < qlen) {
        const float4* po = reinterpret_cast<const float4*>(Og + (long)(q0 + r) * H);
        const float4* pd = reinterpret_cast<const float4*>(dOg + (long)(q0 + r) * H);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float4 a = PL ? af_ld4_planes(Opl + (long)(q0 + r) * H + 4 * i, ops) : po[i], b = pd[i];
          del += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
        l = lse[(long)h * T + t0 + q0 + r] * AF_LOG2E;
      }
      sDel[r] = del;
      sLse[r] = l;
    }
    __syncthreads();
    for (int qt = 0; qt < nqt; ++qt) {
      f32x16 s, dp;
#pragma unroll
      for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
      const float* aq = sQ + (qt * 32 + col) * AF_PITCH + 32 * half;
      const float* ad = sD + (qt * 32 + col) * AF_PITCH + 32 * half;
#pragma unroll
      for (int t = 0; t < 32; ++t) {
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[t], kr[t], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[t], vr[t], dp, 0, 0, 0);
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qr_ = qt * 32 + af_row(e, half);
        float p = __builtin_amdgcn_exp2f(s[e] * c2 - sLse[qr_]);
        p = (qr_ < qlen && kok) ? p : 0.f;
        const float mm = drop.thr ? drop_mult(drop, (uint32_t)(h * T + t0 + q0 + qr_), (uint32_t)key) : 1.f;
        s[e] = p * mm;
        dp[e] = p * (dp[e] * mm - sDel[qr_]) * scale;
      }
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qr_ = qt * 32 + af_row(e, half);
        const float* dn = sD + qr_ * AF_PITCH + col;
        const float* qn = sQ + qr_ * AF_PITCH + col;
        dv[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(dn[0], s[e], dv[0], 0, 0, 0);
        dv[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(dn[32], s[e], dv[1], 0, 0, 0);
        dk[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qn[0], dp[e], dk[0], 0, 0, 0);
        dk[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qn[32], dp[e], dk[1], 0, 0, 0);
      }
    }
  }
  if (kok) {
    if (PL) {
      bf16_t* dst = reinterpret_cast<bf16_t*>(dqkv) + (long)(t0 + key) * H3 + H + h * 64;
      af_store_t_planes<bf16_t>(dk, 1.0f, dst, dps, half);
      af_store_t_planes<bf16_t>(dv, 1.0f, dst + H, dps, half);
    } else {
      float* dst = dqkv + (long)(t0 + key) * H3 + H + h * 64;
      af_store_t(dk, 1.0f, dst, half);
      af_store_t(dv, 1.0f, dst + H, half);
    }
  }
}

// ------------------------------------------------------------------------------------------ host (called from attention.hip)
template <typename K>
static int af_set_lds(K kernel, size_t bytes, const char* name) {
  if (bytes > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    simx_set_error("%s: cannot raise dynamic LDS to %zu", name, bytes);
    return SIMX_ERR_HIP;
  }
  return SIMX_OK;
}

bool simx_mha_f32_ok(int d, int max_len) {
  static const char* pin = getenv("SIMX_MHA_F32");            // SIMX_MHA_F32=generic pins the one-wave-per-row kernels (A/B)
  return d == 64 && max_len <= 4096 && !(pin && pin[0] == 'g');
}

// ctx_ps > 0: ctx is an fp16 plane pair (hi at ctx, lo at + ctx_ps 16-bit elements) instead of an f32 matrix
int simx_mha_fwd_f32(hipStream_t s, int nseq, int heads, const int32_t* cu, int max_len, int T, const float* qkv, float* ctx, float* lse,
                     float scale, DropCtx drop, long ctx_ps) {
  int rc = SIMX_OK;
#define LF(NKT, DENSE)                                                                                                     \
  do {                                                                                                                     \
    const size_t lds = (size_t)2 * NKT * 32 * (DENSE ? 64 : AF_PITCH) * sizeof(float);                                     \
    if (ctx_ps > 0) {                                                                                                      \
      rc = af_set_lds((mha_fwd_f32_kernel<NKT, DENSE, true>), lds, "mha_fwd_f32");                                         \
      if (rc) return rc;                                                                                                   \
      hipLaunchKernelGGL((mha_fwd_f32_kernel<NKT, DENSE, true>), dim3(nseq * heads), dim3(256), lds, s, qkv, ctx, lse, cu, heads, T, scale, drop, ctx_ps); \
    } else {                                                                                                               \
      rc = af_set_lds((mha_fwd_f32_kernel<NKT, DENSE, false>), lds, "mha_fwd_f32");                                        \
      if (rc) return rc;                                                                                                   \
      hipLaunchKernelGGL((mha_fwd_f32_kernel<NKT, DENSE, false>), dim3(nseq * heads), dim3(256), lds, s, qkv, ctx, lse, cu, heads, T, scale, drop, 0L); \
    }                                                                                                                      \
  } while (0)
  static const bool dense5 = [] { const char* e = getenv("SIMX_MHA_F32_DENSE"); return !e || e[0] != '0'; }();
  if (max_len > 256) {                             // chunked kernels: one workgroup per (sequence, head, 128-query chunk)
    const int nchunk = (max_len + 127) / 128;
    const size_t lds = (size_t)2 * AFL_NKT * 32 * AF_PITCH * sizeof(float);
    if (ctx_ps > 0) {
      rc = af_set_lds(mha_fwd_f32_long_kernel<true>, lds, "mha_fwd_f32_long");
      if (rc) return rc;
      hipLaunchKernelGGL(mha_fwd_f32_long_kernel<true>, dim3(nseq * heads * nchunk), dim3(256), lds, s, qkv, ctx, lse, cu, heads, T, nchunk, scale, drop, ctx_ps);
    } else {
      rc = af_set_lds(mha_fwd_f32_long_kernel<false>, lds, "mha_fwd_f32_long");
      if (rc) return rc;
      hipLaunchKernelGGL(mha_fwd_f32_long_kernel<false>, dim3(nseq * heads * nchunk), dim3(256), lds, s, qkv, ctx, lse, cu, heads, T, nchunk, scale, drop, 0L);
    }
    SIMX_CHECK_LAUNCH("mha_fwd_f32_long");
    return SIMX_OK;
  }
  if (max_len <= 32) LF(1, false);
  else if (max_len <= 128) LF(4, false);
  else if (max_len <= 160 && dense5) LF(5, true);   // exactly 80 KB: two workgroups per CU
  else if (max_len <= 160) LF(5, false);
  else LF(8, false);
#undef LF
  SIMX_CHECK_LAUNCH("mha_fwd_f32");
  return SIMX_OK;
}

// ctx_ps > 0 (planes form): ctx is the forward's fp16 plane pair and dqkv leaves as a bf16 plane pair (lo at + dqkv_ps elements)
int simx_mha_bwd_f32(hipStream_t s, int nseq, int heads, const int32_t* cu, int max_len, int T, const float* qkv, const float* ctx,
                     const float* lse, const float* dctx, float* dqkv, float scale, DropCtx drop, long ctx_ps, long dqkv_ps) {
  int rc = SIMX_OK;
#define LB2(NKT, PL)                                                                                                       \
  do {                                                                                                                     \
    const size_t lds = (size_t)2 * NKT * 32 * AF_PITCH * sizeof(float);                                                    \
    const size_t lds2 = lds + (size_t)2 * NKT * 32 * sizeof(float);                                                        \
    rc = af_set_lds((mha_bwd_dq_f32_kernel<NKT, PL>), lds, "mha_bwd_f32");                                                 \
    if (rc) return rc;                                                                                                     \
    rc = af_set_lds((mha_bwd_dkv_f32_kernel<NKT, PL>), lds2, "mha_bwd_f32");                                               \
    if (rc) return rc;                                                                                                     \
    hipLaunchKernelGGL((mha_bwd_dq_f32_kernel<NKT, PL>), dim3(nseq * heads), dim3(256), lds, s, qkv, ctx, lse, dctx, dqkv, cu, heads, T, scale, drop, ctx_ps, dqkv_ps); \
    hipLaunchKernelGGL((mha_bwd_dkv_f32_kernel<NKT, PL>), dim3(nseq * heads), dim3(256), lds2, s, qkv, ctx, lse, dctx, dqkv, cu, heads, T, scale, drop, ctx_ps, dqkv_ps); \
  } while (0)
#define LB(NKT) do { if (ctx_ps > 0) LB2(NKT, true); else LB2(NKT, false); } while (0)
  if (max_len > 256) {
    const int nchunk = (max_len + 127) / 128;
    const size_t lds = (size_t)2 * AFL_NKT * 32 * AF_PITCH * sizeof(float), lds2 = lds + (size_t)2 * AFL_NKT * 32 * sizeof(float);
#define LBL(PL)                                                                                                            \
  do {                                                                                                                     \
    rc = af_set_lds(mha_bwd_dq_f32_long_kernel<PL>, lds, "mha_bwd_f32_long");                                              \
    if (rc) return rc;                                                                                                     \
    rc = af_set_lds(mha_bwd_dkv_f32_long_kernel<PL>, lds2, "mha_bwd_f32_long");                                            \
    if (rc) return rc;                                                                                                     \
    hipLaunchKernelGGL(mha_bwd_dq_f32_long_kernel<PL>, dim3(nseq * heads * nchunk), dim3(256), lds, s, qkv, ctx, lse, dctx, dqkv, cu, heads, T, nchunk, scale, drop, ctx_ps, dqkv_ps); \
    hipLaunchKernelGGL(mha_bwd_dkv_f32_long_kernel<PL>, dim3(nseq * heads * nchunk), dim3(256), lds2, s, qkv, ctx, lse, dctx, dqkv, cu, heads, T, nchunk, scale, drop, ctx_ps, dqkv_ps); \
  } while (0)
    if (ctx_ps > 0) LBL(true); else LBL(false);
#undef LBL
    SIMX_CHECK_LAUNCH("mha_bwd_f32_long");
    return SIMX_OK;
  }
  if (max_len <= 32) LB(1);
  else if (max_len <= 128) LB(4);
  else if (max_len <= 160) LB(5);
  else LB(8);
#undef LB
#undef LB2
  SIMX_CHECK_LAUNCH("mha_bwd_f32");
  return SIMX_OK;
}
