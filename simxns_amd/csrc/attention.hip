// Fused multi-head self-attention (BertSelfAttention core, LEAD/modeling_bert.py:318-374) on the
// packed token layout, forward and backward, for gfx950.
//
// bf16 / head_dim 64 path (MFMA 16x16x32):
//   one workgroup (4 waves) per (sequence, head); that head's K and V (forward) or Q, K, V, dO
//   (backward) are staged ONCE into LDS by global_load_lds and stay resident (S <= 512 forward,
//   S <= 256 backward: 160 KB LDS per CU);  scores are produced TRANSPOSED (keys x queries) so that
//   the softmax row of a query lives in one lane column and the probabilities feed the second MFMA
//   straight from registers; the operand whose contraction index is the LDS row index (V in PV,
//   K in dS.K, dO / Q in the dV / dK products) is fetched with ds_read_b64_tr_b16.
//   LDS image: 128-B rows, 16-B chunk c of row r stored at chunk c ^ f((r>>1)&7), f = (0,2,4,6,5,7,1,3):
//   conflict-free for both the ds_read_b128 row-fragment pattern and the transpose-read pattern.
// generic path (f32 parity mode, other head dims, longer sequences): one wave per query / key row.
//
// Keys are restricted to the sequence's own real tokens, which equals the reference's additive
// (1-mask)*finfo.min bias (exp underflows to exactly 0).
#include "common.h"
#include "prof.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
#define LOG2E 1.4426950408889634f

__device__ __forceinline__ int att_f(int row) {
  const int i = (row >> 1) & 7;
  return (((i << 1) & 7) + (i >> 2) * 5) & 7;
}
__device__ __forceinline__ int att_off(int row, int c16) { return row * 128 + ((c16 ^ att_f(row)) << 4); }

// stage rows [0, rows_pad) x 64 bf16 of one head slice; rows >= len are clamped copies of row len-1
__device__ __forceinline__ void att_stage(const bf16_t* __restrict__ G, int ld, int len, int rows_pad, char* lds,
                                          int wave, int lane) {
  for (int i = wave; i < (rows_pad >> 3); i += 4) {
    const int r = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ att_f(r);
    const int gr = r < len ? r : len - 1;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(G + (long)gr * ld + c * 8), (lds_ptr_t)(lds + i * 1024), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 lds_row_frag(const char* tile, int row, int c16) {
  return *reinterpret_cast<const bf16x8*>(tile + att_off(row, c16));
}

// transpose-read fragment: lane (fg,fs) receives column dt*16+fs of rows {base0+4fg..+3, base1+4fg..+3}
__device__ __forceinline__ bf16x8 lds_tr_frag(uint32_t tile_addr, int base0, int base1, int dt, int fg, int fs) {
  const int r0 = base0 + 4 * fg + (fs >> 2), r1 = base1 + 4 * fg + (fs >> 2);
  const int c16 = dt * 2 + ((fs & 3) >> 1);
  const uint32_t a0 = tile_addr + att_off(r0, c16) + (fs & 1) * 8;
  const uint32_t a1 = tile_addr + att_off(r1, c16) + (fs & 1) * 8;
  bf16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(lo), "=&v"(hi)
               : "v"(a0), "v"(a1)
               : "memory");
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  bf16x8 r;
  r[0] = (short)f2bf(a[0]); r[1] = (short)f2bf(a[1]); r[2] = (short)f2bf(a[2]); r[3] = (short)f2bf(a[3]);
  r[4] = (short)f2bf(b[0]); r[5] = (short)f2bf(b[1]); r[6] = (short)f2bf(b[2]); r[7] = (short)f2bf(b[3]);
  return r;
}

// ------------------------------------------------------------------------------------------ forward
template <int NKT>
__global__ __launch_bounds__(256) void mha_fwd_bf16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                           float* __restrict__ lse, const int* __restrict__ cu,
                                                           int heads, int T, float scale, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64, H3 = 3 * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const int nkt = (len + 15) >> 4;
  const int nkt2 = (nkt + 1) & ~1;
  char* sK = smem;
  char* sV = smem + NKT * 16 * 128;
  att_stage(Kg, H3, len, nkt2 * 16, sK, wave, lane);
  att_stage(Vg, H3, len, nkt2 * 16, sV, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint32_t sV_addr = (uint32_t)(uintptr_t)sV;
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * LOG2E;

  for (int qt = wave; qt < nkt; qt += 4) {
    const int q = qt * 16 + fr;
    const int qc = q < len ? q : len - 1;
    bf16x8 qf[2];
    qf[0] = *reinterpret_cast<const bf16x8*>(Qg + (long)qc * H3 + fg * 8);
    qf[1] = *reinterpret_cast<const bf16x8*>(Qg + (long)qc * H3 + 32 + fg * 8);
    f32x4 s[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (kt < nkt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sK, kt * 16 + fr, fg), qf[0], a, 0, 0, 0);
        a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sK, kt * 16 + fr, 4 + fg), qf[1], a, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * fg + r;
          a[r] = key < len ? a[r] : -INFINITY;
          m = fmaxf(m, a[r]);
        }
        s[kt] = a;
      } else {
        s[kt] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      }
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = exp2f((s[kt][r] - m) * c2);
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (drop.thr) {                              // dropout on the probabilities (the normaliser stays unmasked)
      const uint32_t drow = (uint32_t)(h * T + t0 + q);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        if (kt < nkt) {
          float m4[4];
          drop_mult4(drop, drow, (uint32_t)(kt * 16 + 4 * fg), m4);
          s[kt][0] *= m4[0]; s[kt][1] *= m4[1]; s[kt][2] *= m4[2]; s[kt][3] *= m4[3];
        }
    }

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kp = 0; kp < NKT / 2; ++kp) {
      if (2 * kp < nkt) {
        const bf16x8 pf = pack8(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          const bf16x8 vf = lds_tr_frag(sV_addr, 2 * kp * 16, 2 * kp * 16 + 16, dt, fg, fr);
          __builtin_amdgcn_sched_barrier(0);
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[dt], 0, 0, 0);
        }
      }
    }
    if (q < len) {
      bf16_t* dst = ctx + (long)(t0 + q) * H + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float v[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
        st4(dst + dt * 16, v);
      }
      if (fg == 0) lse[(long)h * T + t0 + q] = m * scale + logf(sum);
    }
  }
}

// ------------------------------------------------------------------------------------------ backward
template <int NKT>
__global__ __launch_bounds__(256) void mha_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ O,
                                                           const float* __restrict__ lse, const bf16_t* __restrict__ dO,
                                                           bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                           int heads, int T, float scale, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64, H3 = 3 * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const bf16_t* Og = O + (long)t0 * H + h * 64;
  const bf16_t* dOg = dO + (long)t0 * H + h * 64;
  const int nkt = (len + 15) >> 4;
  const int nkt2 = (nkt + 1) & ~1;
  const int TILE = NKT * 16 * 128;
  char* sQ = smem;
  char* sK = smem + TILE;
  char* sV = smem + 2 * TILE;
  char* sD = smem + 3 * TILE;
  float* sLse = reinterpret_cast<float*>(smem + 4 * TILE);
  float* sDel = sLse + NKT * 16;
  att_stage(Qg, H3, len, nkt2 * 16, sQ, wave, lane);
  att_stage(Kg, H3, len, nkt2 * 16, sK, wave, lane);
  att_stage(Vg, H3, len, nkt2 * 16, sV, wave, lane);
  att_stage(dOg, H, len, nkt2 * 16, sD, wave, lane);
  // delta_i = dO_i . O_i ; lse_i (in log2 units)
  for (int r = tid; r < nkt2 * 16; r += 256) {
    float del = 0.f, l2 = 0.f;
    if (r < len) {
      const bf16_t* po = Og + (long)r * H;
      const bf16_t* pd = dOg + (long)r * H;
#pragma unroll
      for (int c = 0; c < 64; c += 4) {
        float a[4], b[4];
        ld4(po + c, a);
        ld4(pd + c, b);
        del += a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3];
      }
      l2 = lse[(long)h * T + t0 + r] * LOG2E;
    }
    sDel[r] = del;
    sLse[r] = l2;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint32_t sQ_addr = (uint32_t)(uintptr_t)sQ, sK_addr = (uint32_t)(uintptr_t)sK, sD_addr = (uint32_t)(uintptr_t)sD;
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * LOG2E;

  // ---------------- phase A: dQ, waves own query tiles, loop over key-tile pairs
  for (int qt = wave; qt < nkt; qt += 4) {
    const int q = qt * 16 + fr;
    const bf16x8 qf0 = lds_row_frag(sQ, q, fg), qf1 = lds_row_frag(sQ, q, 4 + fg);
    const bf16x8 df0 = lds_row_frag(sD, q, fg), df1 = lds_row_frag(sD, q, 4 + fg);
    const float lq = sLse[q], dq_ = sDel[q];
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kp = 0; kp < (nkt2 >> 1); ++kp) {
      f32x4 ds[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int kt = 2 * kp + hf;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sK, kt * 16 + fr, fg), qf0, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sK, kt * 16 + fr, 4 + fg), qf1, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sV, kt * 16 + fr, fg), df0, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sV, kt * 16 + fr, 4 + fg), df1, dp, 0, 0, 0);
        float m4[4] = {1.f, 1.f, 1.f, 1.f};
        if (drop.thr) drop_mult4(drop, (uint32_t)(h * T + t0 + q), (uint32_t)(kt * 16 + 4 * fg), m4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * fg + r;
          const float p = (key < len && q < len) ? exp2f(s[r] * c2 - lq) : 0.f;
          ds[hf][r] = p * (dp[r] * m4[r] - dq_) * scale;
        }
      }
      const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 kf = lds_tr_frag(sK_addr, 2 * kp * 16, 2 * kp * 16 + 16, dt, fg, fr);
        __builtin_amdgcn_sched_barrier(0);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, dsf, dq[dt], 0, 0, 0);
      }
    }
    if (q < len) {
      bf16_t* dst = dqkv + (long)(t0 + q) * H3 + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float v[4] = {dq[dt][0], dq[dt][1], dq[dt][2], dq[dt][3]};
        st4(dst + dt * 16, v);
      }
    }
  }

  // ---------------- phase B: dK, dV, waves own key tiles, loop over query-tile pairs
  for (int kt = wave; kt < nkt; kt += 4) {
    const int key = kt * 16 + fr;
    const bf16x8 kf0 = lds_row_frag(sK, key, fg), kf1 = lds_row_frag(sK, key, 4 + fg);
    const bf16x8 vf0 = lds_row_frag(sV, key, fg), vf1 = lds_row_frag(sV, key, 4 + fg);
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int qp = 0; qp < (nkt2 >> 1); ++qp) {
      f32x4 pp[2], ds[2];
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int qt = 2 * qp + hf;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sQ, qt * 16 + fr, fg), kf0, s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sQ, qt * 16 + fr, 4 + fg), kf1, s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sD, qt * 16 + fr, fg), vf0, dp, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lds_row_frag(sD, qt * 16 + fr, 4 + fg), vf1, dp, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = qt * 16 + 4 * fg + r;
          const float p = (q < len && key < len) ? exp2f(s[r] * c2 - sLse[q]) : 0.f;
          const float mm = drop.thr ? drop_mult(drop, (uint32_t)(h * T + t0 + q), (uint32_t)key) : 1.f;
          pp[hf][r] = p * mm;
          ds[hf][r] = p * (dp[r] * mm - sDel[q]) * scale;
        }
      }
      const bf16x8 pf = pack8(pp[0], pp[1]);
      const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 dof = lds_tr_frag(sD_addr, 2 * qp * 16, 2 * qp * 16 + 16, dt, fg, fr);
        const bf16x8 qtf = lds_tr_frag(sQ_addr, 2 * qp * 16, 2 * qp * 16 + 16, dt, fg, fr);
        __builtin_amdgcn_sched_barrier(0);
        dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof, pf, dv[dt], 0, 0, 0);
        dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf, dsf, dk[dt], 0, 0, 0);
      }
    }
    if (key < len) {
      bf16_t* dstk = dqkv + (long)(t0 + key) * H3 + H + h * 64 + 4 * fg;
      bf16_t* dstv = dstk + H;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float a[4] = {dk[dt][0], dk[dt][1], dk[dt][2], dk[dt][3]};
        float b[4] = {dv[dt][0], dv[dt][1], dv[dt][2], dv[dt][3]};
        st4(dstk + dt * 16, a);
        st4(dstv + dt * 16, b);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// generic kernels: one wave per query (fwd, dQ) / per key (dK,dV); any head_dim <= 128
// ------------------------------------------------------------------------------------------
template <typename TT>
__global__ __launch_bounds__(256) void mha_fwd_simple_kernel(const TT* __restrict__ qkv, TT* __restrict__ ctx,
                                                             float* __restrict__ lse, const int* __restrict__ cu,
                                                             int heads, int d, int T, float scale, int max_len, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sq = reinterpret_cast<float*>(smem);              // [4][128]
  float* sc = sq + 4 * 128;                                // [4][max_len]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0 || (int)blockIdx.y * 4 >= len) return;
  const int H = heads * d, H3 = 3 * H;
  const int q = blockIdx.y * 4 + w;
  const int qc = q < len ? q : len - 1;
  const TT* Qp = qkv + (long)(t0 + qc) * H3 + h * d;
  const TT* Kb = qkv + (long)t0 * H3 + H + h * d;
  const TT* Vb = Kb + H;
  for (int dd = lane; dd < d; dd += 64) sq[w * 128 + dd] = Elem<TT>::ld(Qp + dd);
  __syncthreads();
  float mx = -INFINITY;
  for (int j = lane; j < len; j += 64) {
    const TT* kp = Kb + (long)j * H3;
    float dot = 0.f;
    for (int dd = 0; dd < d; ++dd) dot = fmaf(sq[w * 128 + dd], Elem<TT>::ld(kp + dd), dot);
    const float sv = dot * scale;
    sc[w * max_len + j] = sv;
    mx = fmaxf(mx, sv);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < len; j += 64) {
    const float p = expf(sc[w * max_len + j] - mx);
    sc[w * max_len + j] = drop.thr ? p * drop_mult(drop, (uint32_t)(h * T + t0 + qc), (uint32_t)j) : p;
    sum += p;
  }
  sum = wave_sum(sum);
  __syncthreads();
  if (q < len) {
    for (int dd = lane; dd < d; dd += 64) {
      float o = 0.f;
      for (int j = 0; j < len; ++j) o = fmaf(sc[w * max_len + j], Elem<TT>::ld(Vb + (long)j * H3 + dd), o);
      Elem<TT>::st(ctx + (long)(t0 + q) * H + h * d + dd, o / sum);
    }
    if (lane == 0) lse[(long)h * T + t0 + q] = mx + logf(sum);
  }
}

// mode 0: dQ (wave per query row i, inner index j = keys); mode 1: dK,dV (wave per key row j, inner = queries)
template <typename TT, int MODE>
__global__ __launch_bounds__(256) void mha_bwd_simple_kernel(const TT* __restrict__ qkv, const TT* __restrict__ O,
                                                             const float* __restrict__ lse, const TT* __restrict__ dO,
                                                             TT* __restrict__ dqkv, const int* __restrict__ cu, int heads,
                                                             int d, int T, float scale, int max_len, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sa = reinterpret_cast<float*>(smem);              // [4][128] own row of Q (mode 0) / K (mode 1)
  float* sb = sa + 4 * 128;                                // [4][128] own row of dO (mode 0) / V (mode 1)
  float* s1 = sb + 4 * 128;                                // [4][max_len]  ds
  float* s2 = s1 + 4 * max_len;                            // [4][max_len]  p (mode 1)
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0 || (int)blockIdx.y * 4 >= len) return;
  const int H = heads * d, H3 = 3 * H;
  const int row = blockIdx.y * 4 + w;
  const int rc = row < len ? row : len - 1;
  const TT* Qb = qkv + (long)t0 * H3 + h * d;
  const TT* Kb = Qb + H;
  const TT* Vb = Kb + H;
  const TT* Ob = O + (long)t0 * H + h * d;
  const TT* dOb = dO + (long)t0 * H + h * d;
  if (MODE == 0) {
    float del = 0.f;
    for (int dd = lane; dd < d; dd += 64) {
      sa[w * 128 + dd] = Elem<TT>::ld(Qb + (long)rc * H3 + dd);
      const float g = Elem<TT>::ld(dOb + (long)rc * H + dd);
      sb[w * 128 + dd] = g;
      del += g * Elem<TT>::ld(Ob + (long)rc * H + dd);
    }
    del = wave_sum(del);
    const float li = lse[(long)h * T + t0 + rc];
    __syncthreads();
    for (int j = lane; j < len; j += 64) {
      float dot = 0.f, dp = 0.f;
      for (int dd = 0; dd < d; ++dd) {
        dot = fmaf(sa[w * 128 + dd], Elem<TT>::ld(Kb + (long)j * H3 + dd), dot);
        dp = fmaf(sb[w * 128 + dd], Elem<TT>::ld(Vb + (long)j * H3 + dd), dp);
      }
      const float p = expf(dot * scale - li);
      const float mm = drop.thr ? drop_mult(drop, (uint32_t)(h * T + t0 + rc), (uint32_t)j) : 1.f;
      s1[w * max_len + j] = p * (dp * mm - del) * scale;
    }
    __syncthreads();
    if (row < len)
      for (int dd = lane; dd < d; dd += 64) {
        float acc = 0.f;
        for (int j = 0; j < len; ++j) acc = fmaf(s1[w * max_len + j], Elem<TT>::ld(Kb + (long)j * H3 + dd), acc);
        Elem<TT>::st(dqkv + (long)(t0 + row) * H3 + h * d + dd, acc);
      }
  } else {
    for (int dd = lane; dd < d; dd += 64) {
      sa[w * 128 + dd] = Elem<TT>::ld(Kb + (long)rc * H3 + dd);
      sb[w * 128 + dd] = Elem<TT>::ld(Vb + (long)rc * H3 + dd);
    }
    __syncthreads();
    for (int i = lane; i < len; i += 64) {
      float dot = 0.f, dp = 0.f, del = 0.f;
      for (int dd = 0; dd < d; ++dd) {
        const float g = Elem<TT>::ld(dOb + (long)i * H + dd);
        dot = fmaf(Elem<TT>::ld(Qb + (long)i * H3 + dd), sa[w * 128 + dd], dot);
        dp = fmaf(g, sb[w * 128 + dd], dp);
        del = fmaf(g, Elem<TT>::ld(Ob + (long)i * H + dd), del);
      }
      const float p = expf(dot * scale - lse[(long)h * T + t0 + i]);
      const float mm = drop.thr ? drop_mult(drop, (uint32_t)(h * T + t0 + i), (uint32_t)rc) : 1.f;
      s2[w * max_len + i] = p * mm;
      s1[w * max_len + i] = p * (dp * mm - del) * scale;
    }
    __syncthreads();
    if (row < len)
      for (int dd = lane; dd < d; dd += 64) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < len; ++i) {
          ak = fmaf(s1[w * max_len + i], Elem<TT>::ld(Qb + (long)i * H3 + dd), ak);
          av = fmaf(s2[w * max_len + i], Elem<TT>::ld(dOb + (long)i * H + dd), av);
        }
        Elem<TT>::st(dqkv + (long)(t0 + row) * H3 + H + h * d + dd, ak);
        Elem<TT>::st(dqkv + (long)(t0 + row) * H3 + 2 * H + h * d + dd, av);
      }
  }
}

// ------------------------------------------------------------------------------------------ host
template <typename K>
static int set_lds(K kernel, size_t bytes, const char* name) {
  if (bytes > 65536) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) !=
        hipSuccess) {
      simx_set_error("%s: cannot raise dynamic LDS to %zu", name, bytes);
      return SIMX_ERR_HIP;
    }
  }
  return SIMX_OK;
}

static int check_common(int dtype, int nseq, int heads, int d, int max_len, int T, const char* who) {
  SIMX_REQUIRE(dtype == SIMX_F32 || dtype == SIMX_BF16, SIMX_ERR_BAD_DTYPE, "%s: dtype %d", who, dtype);
  SIMX_REQUIRE(nseq > 0 && heads > 0 && T > 0 && max_len > 0, SIMX_ERR_BAD_SHAPE, "%s: bad shape", who);
  SIMX_REQUIRE(d > 0 && d <= 128, SIMX_ERR_UNSUPPORTED, "%s: head_dim %d > 128", who, d);
  SIMX_REQUIRE(max_len <= 4096, SIMX_ERR_UNSUPPORTED, "%s: max_len %d > 4096", who, max_len);
  return SIMX_OK;
}

extern "C" int simx_mha_fwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, void* ctx, float* lse, const simx_dropout* dropd);
extern "C" int simx_mha_bwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv,
                               const simx_dropout* dropd);
extern "C" int simx_mha_fwd(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                            int T, const void* qkv, void* ctx, float* lse) {
  return simx_mha_fwd_ex(stream, dtype, nseq, heads, d, cu, max_len, T, qkv, ctx, lse, nullptr);
}
extern "C" int simx_mha_bwd(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                            int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv) {
  return simx_mha_bwd_ex(stream, dtype, nseq, heads, d, cu, max_len, T, qkv, ctx, lse, dctx, dqkv, nullptr);
}

extern "C" int simx_mha_fwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, void* ctx, float* lse, const simx_dropout* dropd) {
  hipStream_t s = (hipStream_t)stream;
  const DropCtx drop = make_drop(dropd);
  SIMX_PROF(SIMX_K_MHA_FWD, s, 4.0 * T * max_len * heads * d);
  int rc = check_common(dtype, nseq, heads, d, max_len, T, "mha_fwd");
  if (rc) return rc;
  const float scale = 1.0f / sqrtf((float)d);
  if (dtype == SIMX_BF16 && d == 64 && max_len <= 512) {
#define LF(NKT)                                                                                                      \
  do {                                                                                                               \
    const size_t lds = (size_t)2 * NKT * 16 * 128;                                                                   \
    rc = set_lds(mha_fwd_bf16_kernel<NKT>, lds, "mha_fwd");                                                          \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_fwd_bf16_kernel<NKT>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv,        \
                       (bf16_t*)ctx, lse, cu, heads, T, scale, drop);                                                \
  } while (0)
    if (max_len <= 32) LF(2);
    else if (max_len <= 128) LF(8);
    else if (max_len <= 160) LF(10);
    else if (max_len <= 256) LF(16);
    else LF(32);
#undef LF
    SIMX_CHECK_LAUNCH("mha_fwd_bf16");
    return SIMX_OK;
  }
  const size_t lds = (size_t)(4 * 128 + 4 * max_len) * sizeof(float);
  dim3 grid(nseq * heads, cdiv(max_len, 4));
  if (dtype == SIMX_F32) {
    rc = set_lds(mha_fwd_simple_kernel<float>, lds, "mha_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL((mha_fwd_simple_kernel<float>), grid, dim3(256), lds, s, (const float*)qkv, (float*)ctx, lse, cu,
                       heads, d, T, scale, max_len, drop);
  } else {
    rc = set_lds(mha_fwd_simple_kernel<bf16_t>, lds, "mha_fwd");
    if (rc) return rc;
    hipLaunchKernelGGL((mha_fwd_simple_kernel<bf16_t>), grid, dim3(256), lds, s, (const bf16_t*)qkv, (bf16_t*)ctx, lse,
                       cu, heads, d, T, scale, max_len, drop);
  }
  SIMX_CHECK_LAUNCH("mha_fwd_simple");
  return SIMX_OK;
}

extern "C" int simx_mha_bwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv,
                               const simx_dropout* dropd) {
  hipStream_t s = (hipStream_t)stream;
  const DropCtx drop = make_drop(dropd);
  SIMX_PROF(SIMX_K_MHA_BWD, s, 8.0 * T * max_len * heads * d);
  int rc = check_common(dtype, nseq, heads, d, max_len, T, "mha_bwd");
  if (rc) return rc;
  const float scale = 1.0f / sqrtf((float)d);
  if (dtype == SIMX_BF16 && d == 64 && max_len <= 256) {
#define LB(NKT)                                                                                                      \
  do {                                                                                                               \
    const size_t lds = (size_t)4 * NKT * 16 * 128 + 2 * NKT * 16 * sizeof(float);                                    \
    rc = set_lds(mha_bwd_bf16_kernel<NKT>, lds, "mha_bwd");                                                          \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_bwd_bf16_kernel<NKT>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv,        \
                       (const bf16_t*)ctx, lse, (const bf16_t*)dctx, (bf16_t*)dqkv, cu, heads, T, scale, drop);      \
  } while (0)
    if (max_len <= 32) LB(2);
    else if (max_len <= 128) LB(8);
    else if (max_len <= 160) LB(10);
    else LB(16);
#undef LB
    SIMX_CHECK_LAUNCH("mha_bwd_bf16");
    return SIMX_OK;
  }
  const size_t lds = (size_t)(8 * 128 + 8 * max_len) * sizeof(float);
  dim3 grid(nseq * heads, cdiv(max_len, 4));
#define LS(TT, MODE)                                                                                                 \
  do {                                                                                                               \
    rc = set_lds(mha_bwd_simple_kernel<TT, MODE>, lds, "mha_bwd");                                                   \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_bwd_simple_kernel<TT, MODE>), grid, dim3(256), lds, s, (const TT*)qkv, (const TT*)ctx, lse, \
                       (const TT*)dctx, (TT*)dqkv, cu, heads, d, T, scale, max_len, drop);                           \
  } while (0)
  if (dtype == SIMX_F32) { LS(float, 0); LS(float, 1); } else { LS(bf16_t, 0); LS(bf16_t, 1); }
#undef LS
  SIMX_CHECK_LAUNCH("mha_bwd_simple");
  return SIMX_OK;
}
