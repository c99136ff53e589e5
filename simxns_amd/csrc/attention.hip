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
#include <type_traits>
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

// In the MFMA kernels below `bf16_t` is the raw 16-bit storage of EITHER operand format; the template parameter
// F in {bf16_t, f16_t} (H16<F>, common.h) selects the conversions and the MFMA instruction.
template <typename F>
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  // (element by element on purpose: built from four packed words and bit-cast, hipcc kept every score tile of the forward
  // kernel live across the softmax and mha_fwd<8> went from 114 to 239 VGPRs = half the occupancy, +30 % time)
  bf16x8 r;
  r[0] = H16<F>::bits(a[0]); r[1] = H16<F>::bits(a[1]); r[2] = H16<F>::bits(a[2]); r[3] = H16<F>::bits(a[3]);
  r[4] = H16<F>::bits(b[0]); r[5] = H16<F>::bits(b[1]); r[6] = H16<F>::bits(b[2]); r[7] = H16<F>::bits(b[3]);
  return r;
}


// Two layouts of the packed q / k / v (and dq / dk / dv) tensor:
//   token-major (hm_rows == 0): [T][3H], row t = [q(heads x 64) | k | v] -- what a [T,3H] GEMM output looks like;
//   head-major  (hm_rows  = R): [3][heads][R][64] -- plane (which, head) holds that head's 64 values of every token, rows
//     128 B apart, so the 128 x 64 block one (sequence, head) workgroup reads is ONE contiguous 16 KB range instead of 128
//     pieces of 128 B that sit 3H * 2 B apart (DRAM pages, TLB reach).  The QKV GEMM's epilogue writes it (each wave of that
//     kernel owns exactly one head's 64 columns) and the dgrad / wgrad loaders read it (simx_gemm_nt_hm, simx_gemm_tn_hm).
// Either way element (which w, head h, token t) sits at  base + w * ws + t * ld  with base including the head offset.
struct QkvLay { long ws; int ld; };
__device__ __forceinline__ QkvLay qkv_lay(int heads, int hm_rows) {
  return hm_rows > 0 ? QkvLay{(long)heads * hm_rows * 64, 64} : QkvLay{(long)heads * 64, 3 * heads * 64};
}
template <typename P>
__device__ __forceinline__ P* qkv_head(P* base, int heads, int h, int t0, int hm_rows) {
  return hm_rows > 0 ? base + ((long)h * hm_rows + t0) * 64 : base + (long)t0 * (3 * heads * 64) + h * 64;
}

// ------------------------------------------------------------------------------------------ forward
// (ALLT: every one of the NKT key tiles is present -- all-max batches and the cross-encoder's 158 of 160 -- is a copy of the
// query-tile loop without uniform branches: with them hipcc emitted one key tile at a time, ds_read -> wait -> two MFMAs -> wait
// -> its share of the row maximum; as ONE basic block the sixteen QK^T MFMAs of a query tile issue back to back.  Dropout on /
// off is a template parameter for the same reason.)
template <typename F, int NKT, bool DROP>
__global__ __launch_bounds__(256, (NKT <= 8 ? 4 : NKT <= 10 ? 3 : NKT <= 16 ? 2 : 1)) void mha_fwd_h16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx,
                                                           float* __restrict__ lse, const int* __restrict__ cu,
                                                           int heads, int T, float scale, DropCtx drop, int hm_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const QkvLay lay = qkv_lay(heads, hm_rows);
  const int H3 = lay.ld;                                  // row pitch of q / k / v in either layout
  const bf16_t* Qg = qkv_head(qkv, heads, h, t0, hm_rows);
  const bf16_t* Kg = Qg + lay.ws;
  const bf16_t* Vg = Kg + lay.ws;
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
#define F2_RDTR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=&v"(DST) : "v"(ADDR) : "memory")
#define F2_CAT(LO, HI) ((bf16x8){LO[0], LO[1], LO[2], LO[3], HI[0], HI[1], HI[2], HI[3]})
  uint32_t vtr[4];                                 // transpose-fragment lane constants (see mha_bwd2_bf16_kernel)
  {
    const int rr = 4 * fg + (fr >> 2), tsw = att_f(rr), tx = (fr & 3) >> 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vtr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  }

  auto tiles = [&](auto allt_c) {
  constexpr bool ALLT = decltype(allt_c)::value;
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
      if (ALLT || kt < nkt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = H16<F>::mfma(lds_row_frag(sK, kt * 16 + fr, fg), qf[0], a);
        a = H16<F>::mfma(lds_row_frag(sK, kt * 16 + fr, 4 + fg), qf[1], a);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * fg + r;
          if (!ALLT || kt == NKT - 1) a[r] = key < len ? a[r] : -INFINITY;      // (all tiles present: only the last can be ragged)
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
        const float p = __builtin_amdgcn_exp2f((s[kt][r] - m) * c2);     // raw v_exp_f32: argument <= 0, exp2(-inf) = 0
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (DROP) {                                  // dropout on the probabilities (the normaliser stays unmasked)
      const uint32_t drow = (uint32_t)(h * T + t0 + q);
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
        if (ALLT || kt < nkt) {
          float m4[4];
          drop_mult4(drop, drow, (uint32_t)(kt * 16 + 4 * fg), m4);
          s[kt][0] *= m4[0]; s[kt][1] *= m4[1]; s[kt][2] *= m4[2]; s[kt][3] *= m4[3];
        }
    }

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // V^T fragments two key-tile pairs at a time: 16 transpose reads in flight, ONE wait (the per-fragment wait of the
    // first version left the wave parked ~40 % of its cycles); addresses = tile + pair*4096 + lane constant (+2048)
#pragma unroll
    for (int kb = 0; kb < NKT / 2; kb += 2) {
      if (ALLT || 2 * kb < nkt) {
        const uint32_t b0 = sV_addr + (uint32_t)(kb * 4096);
        bf16x4 a0l, a0h, a1l, a1h, a2l, a2h, a3l, a3h, c0l, c0h, c1l, c1h, c2l, c2h, c3l, c3h;
        F2_RDTR(a0l, b0 + vtr[0], 0); F2_RDTR(a0h, b0 + vtr[0], 2048); F2_RDTR(a1l, b0 + vtr[1], 0); F2_RDTR(a1h, b0 + vtr[1], 2048);
        F2_RDTR(a2l, b0 + vtr[2], 0); F2_RDTR(a2h, b0 + vtr[2], 2048); F2_RDTR(a3l, b0 + vtr[3], 0); F2_RDTR(a3h, b0 + vtr[3], 2048);
        if (kb + 1 < NKT / 2) {                       // compile-time: the second pair's slots exist in the LDS tile
          F2_RDTR(c0l, b0 + vtr[0], 4096); F2_RDTR(c0h, b0 + vtr[0], 6144); F2_RDTR(c1l, b0 + vtr[1], 4096); F2_RDTR(c1h, b0 + vtr[1], 6144);
          F2_RDTR(c2l, b0 + vtr[2], 4096); F2_RDTR(c2h, b0 + vtr[2], 6144); F2_RDTR(c3l, b0 + vtr[3], 4096); F2_RDTR(c3h, b0 + vtr[3], 6144);
          // (ONE wait names every register an asm read above is still filling: a register the wait does not name may be copied
          // by the compiler before it)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0l), "+v"(a0h), "+v"(a1l), "+v"(a1h), "+v"(a2l), "+v"(a2h), "+v"(a3l), "+v"(a3h),
                       "+v"(c0l), "+v"(c0h), "+v"(c1l), "+v"(c1h), "+v"(c2l), "+v"(c2h), "+v"(c3l), "+v"(c3h)::"memory");
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0l), "+v"(a0h), "+v"(a1l), "+v"(a1h), "+v"(a2l), "+v"(a2h), "+v"(a3l), "+v"(a3h)::"memory");
        }
        const bf16x8 pf = pack8<F>(s[2 * kb], s[2 * kb + 1]);
        o[0] = H16<F>::mfma(F2_CAT(a0l, a0h), pf, o[0]);
        o[1] = H16<F>::mfma(F2_CAT(a1l, a1h), pf, o[1]);
        o[2] = H16<F>::mfma(F2_CAT(a2l, a2h), pf, o[2]);
        o[3] = H16<F>::mfma(F2_CAT(a3l, a3h), pf, o[3]);
        if (kb + 1 < NKT / 2) {
          if (ALLT || 2 * (kb + 1) < nkt) {           // rows past the padded length are uninitialised LDS: skip, never multiply
            const bf16x8 pg = pack8<F>(s[2 * kb + 2], s[2 * kb + 3]);
            o[0] = H16<F>::mfma(F2_CAT(c0l, c0h), pg, o[0]);
            o[1] = H16<F>::mfma(F2_CAT(c1l, c1h), pg, o[1]);
            o[2] = H16<F>::mfma(F2_CAT(c2l, c2h), pg, o[2]);
            o[3] = H16<F>::mfma(F2_CAT(c3l, c3h), pg, o[3]);
          }
        }
      }
    }
    if (q < len) {
      bf16_t* dst = ctx + (long)(t0 + q) * H + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float v[4] = {o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
        st4h<F>(dst + dt * 16, v);
      }
      if (fg == 0) lse[(long)h * T + t0 + q] = m * scale + logf(sum);
    }
  }
  };
  if (nkt == NKT) tiles(std::true_type{}); else tiles(std::false_type{});
}

// ------------------------------------------------------------------------------------------ forward, long sequences
// 256 < S <= 4096 (S > 512 used to run the one-wave-per-row kernel; the resident kernel above holds K and V of the whole sequence).
// Here one workgroup takes a (sequence, head, 128-query chunk), a wave two 16-query tiles (Q fragments, running maximum / normaliser
// and the output accumulators in registers), and the workgroup walks the 128-key chunks: K and V of a chunk in 32 KB of LDS, online
// softmax (the accumulators are rescaled when the running maximum moves).  Either q/k/v layout.
template <typename F, bool DROP>
__global__ __launch_bounds__(256, 2) void mha_fwd_long_h16_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ ctx, float* __restrict__ lse,
                                                                 const int* __restrict__ cu, int heads, int T, int nchunk, float scale, DropCtx drop,
                                                                 int hm_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x / nchunk, qc = blockIdx.x % nchunk;
  const int seq = blk / heads, h = blk % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  const int q0 = qc * 128;
  if (len <= 0 || q0 >= len) return;
  const int H = heads * 64;
  const QkvLay lay = qkv_lay(heads, hm_rows);
  const int H3 = lay.ld;
  const bf16_t* Qg = qkv_head(qkv, heads, h, t0, hm_rows);
  const bf16_t* Kg = Qg + lay.ws;
  const bf16_t* Vg = Kg + lay.ws;
  const int nkc = (len + 127) >> 7;
  char* sK = smem;
  char* sV = smem + 16384;
  const uint32_t sV_addr = (uint32_t)(uintptr_t)sV;
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * LOG2E;
  uint32_t vtr[4];
  {
    const int rr = 4 * fg + (fr >> 2), tsw = att_f(rr), tx = (fr & 3) >> 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vtr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  }
  bf16x8 qf[2][2];
  f32x4 o[2][4];
  float mrun[2], lrun[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = q0 + (wave + 4 * j) * 16 + fr;
    const int qcl = q < len ? q : len - 1;
    qf[j][0] = *reinterpret_cast<const bf16x8*>(Qg + (long)qcl * H3 + fg * 8);
    qf[j][1] = *reinterpret_cast<const bf16x8*>(Qg + (long)qcl * H3 + 32 + fg * 8);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mrun[j] = -INFINITY;
    lrun[j] = 0.f;
  }
  for (int kc = 0; kc < nkc; ++kc) {
    const int k0 = kc * 128;
    const int nk = len - k0 < 128 ? len - k0 : 128;
    const int nkt = (nk + 15) >> 4, nkt2 = (nkt + 1) & ~1;
    __syncthreads();                                 // every wave is done with the previous chunk's tiles
    att_stage(Kg + (long)k0 * H3, H3, nk, nkt2 * 16, sK, wave, lane);
    att_stage(Vg + (long)k0 * H3, H3, nk, nkt2 * 16, sV, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto chunk = [&](auto allt_c) {
      constexpr bool ALLT = decltype(allt_c)::value;   // all 128 keys of the chunk are real: no masks, no tile guards
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (q0 + (wave + 4 * j) * 16 >= len) continue;                       // (wave-uniform)
        const int q = q0 + (wave + 4 * j) * 16 + fr;
        f32x4 s[8];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt) {
          if (ALLT || kt < nkt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = H16<F>::mfma(lds_row_frag(sK, kt * 16 + fr, fg), qf[j][0], a);
            a = H16<F>::mfma(lds_row_frag(sK, kt * 16 + fr, 4 + fg), qf[j][1], a);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              if (!ALLT) a[r] = k0 + kt * 16 + 4 * fg + r < len ? a[r] : -INFINITY;
              mx = fmaxf(mx, a[r]);
            }
            s[kt] = a;
          } else {
            s[kt] = (f32x4){-INFINITY, -INFINITY, -INFINITY, -INFINITY};
          }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mnew = fmaxf(mrun[j], mx);                               // finite: a chunk holds at least one real key
        const float alpha = __builtin_amdgcn_exp2f((mrun[j] - mnew) * c2);   // first chunk: exp2(-inf) = 0
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 8; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float p = __builtin_amdgcn_exp2f((s[kt][r] - mnew) * c2);
            s[kt][r] = p;
            sum += p;
          }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        lrun[j] = lrun[j] * alpha + sum;
        mrun[j] = mnew;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[j][dt][e] *= alpha;
        if (DROP) {
          const uint32_t drow = (uint32_t)(h * T + t0 + q);
#pragma unroll
          for (int kt = 0; kt < 8; ++kt)
            if (ALLT || kt < nkt) {
              float m4[4];
              drop_mult4(drop, drow, (uint32_t)(k0 + kt * 16 + 4 * fg), m4);
              s[kt][0] *= m4[0]; s[kt][1] *= m4[1]; s[kt][2] *= m4[2]; s[kt][3] *= m4[3];
            }
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          if (ALLT || 2 * kb < nkt) {                  // (rows past the padded length are uninitialised LDS: never multiplied)
            const uint32_t b0 = sV_addr + (uint32_t)(kb * 4096);
            bf16x4 a0l, a0h, a1l, a1h, a2l, a2h, a3l, a3h;
            F2_RDTR(a0l, b0 + vtr[0], 0); F2_RDTR(a0h, b0 + vtr[0], 2048); F2_RDTR(a1l, b0 + vtr[1], 0); F2_RDTR(a1h, b0 + vtr[1], 2048);
            F2_RDTR(a2l, b0 + vtr[2], 0); F2_RDTR(a2h, b0 + vtr[2], 2048); F2_RDTR(a3l, b0 + vtr[3], 0); F2_RDTR(a3h, b0 + vtr[3], 2048);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0l), "+v"(a0h), "+v"(a1l), "+v"(a1h), "+v"(a2l), "+v"(a2h), "+v"(a3l), "+v"(a3h)::"memory");
            const bf16x8 pf = pack8<F>(s[2 * kb], s[2 * kb + 1]);
            o[j][0] = H16<F>::mfma(F2_CAT(a0l, a0h), pf, o[j][0]);
            o[j][1] = H16<F>::mfma(F2_CAT(a1l, a1h), pf, o[j][1]);
            o[j][2] = H16<F>::mfma(F2_CAT(a2l, a2h), pf, o[j][2]);
            o[j][3] = H16<F>::mfma(F2_CAT(a3l, a3h), pf, o[j][3]);
          }
        }
      }
    };
    if (nk == 128) chunk(std::true_type{}); else chunk(std::false_type{});
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = q0 + (wave + 4 * j) * 16 + fr;
    if (q < len) {
      const float inv = 1.0f / lrun[j];
      bf16_t* dst = ctx + (long)(t0 + q) * H + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        float v[4] = {o[j][dt][0] * inv, o[j][dt][1] * inv, o[j][dt][2] * inv, o[j][dt][3] * inv};
        st4h<F>(dst + dt * 16, v);
      }
      if (fg == 0) lse[(long)h * T + t0 + q] = mrun[j] * scale + logf(lrun[j]);
    }
  }
}

// ------------------------------------------------------------------------------------------ backward
// Q, K, V, dO of the (sequence, head) staged once in LDS; phase A: dQ (waves own query tiles), phase B: dK, dV (waves
// own key tiles).  How the instruction stream is built matters more than the MFMA count here (PMC on the first version: 38 % of wave cycles issuing ~5000 non-MFMA instructions per wave, 43 % parked in s_waitcnt):
//   * every fragment address is (tile + pair*4096 + LANE CONSTANT [+2048]): the swizzle term f((row>>1)&7) does not
//     depend on the 16-row tile index, so the six lane constants are computed once instead of ~12 VALU per read;
//   * all LDS reads of one pair iteration (row fragments, transpose fragments, lse / delta vectors) are issued
//     back to back by inline asm and waited for ONCE (v1 waited after every transpose fragment);
//   * delta = dO.O uses 16-B loads, two threads per row;
//   * dQ / dK / dV leave through a 2 KB per-wave LDS patch as 16 B per lane = full 128-B (token, head) lines
//     instead of 8-B stores (32-B segments).
#define A2_RD128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=&v"(DST) : "v"(ADDR) : "memory")
#define A2_RDTR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=&v"(DST) : "v"(ADDR) : "memory")
#define A2_RD128O(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define A2_RDTRO(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=&v"(DST) : "v"(ADDR), "n"(OFF) : "memory")
#define A2_CAT(LO, HI) ((bf16x8){LO[0], LO[1], LO[2], LO[3], HI[0], HI[1], HI[2], HI[3]})

// stage a wave's 16x64 tile (MFMA layout: lane = row fr, 4 columns dt*16 + 4*fg) through its 2 KB LDS patch and store it
// as full 128-B lines: rows t_row0 + r (r < nrows valid), column block h*64 of a [.., ld] bf16 matrix
template <typename F>
__device__ __forceinline__ void a2_store_tile(const f32x4 (&acc)[4], char* patch, uint32_t patch_addr, bf16_t* __restrict__ dst,
                                              long ld, int nrows, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
  const int sw = (fr >> 1) & 7;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const uint2 o = make_uint2(H16<F>::pack2(acc[dt][0], acc[dt][1]), H16<F>::pack2(acc[dt][2], acc[dt][3]));
    const uint32_t ad = patch_addr + (uint32_t)(fr * 128 + (((dt * 2 + (fg >> 1)) ^ sw) << 4) + (fg & 1) * 8);
    asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(o) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int r = it * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    const uint4 v = *reinterpret_cast<const uint4*>(patch + r * 128 + (lane & 7) * 16);
    if (r < nrows) *reinterpret_cast<uint4*>(dst + (long)r * ld + c * 8) = v;
  }
}

// What bounds it (tools/att_bench.py on the bench shape, fp16, 2048 x 12 blocks of S = 128, p = 0.1 / no dropout): 0.916 / 0.805 ms
// as first written; 0.773 / 0.624 with every workgroup reading ONE block's operands (no HBM traffic: the instruction stream
// alone); 0.605 with the two phases' loops removed (loads and stores only); the MFMAs alone would take 0.15.  So the kernel is
// bound by its own VALU work (exp2, the softmax-gradient algebra, packing, and the dropout hash: 0.11 ms of it), not by HBM
// and not by occupancy.  The mask is hashed ONCE per block, in the prologue: thread (query, key tile) draws the 16 bits of its
// tile and files them in the LDS at [query][key tile]; phase A reads 32 bits per (query, key pair), phase B -- whose lanes hold
// four different query ROWS and paid four hashes per four elements where phase A paid two -- four 16-bit words per query
// tile.  One third fewer hashes: 0.916 -> 0.902 ms, the bit tests eat most of it; with the three items below 0.891 (no
// dropout 0.805 -> 0.761; ragged lengths 0.817 -> 0.751; S = 256: 4.66 -> 4.09).
// (Also measured: the bits drawn by phase A and handed to phase B through wave ballots, which needs a workgroup barrier
// between the phases -- 0.912; a three-tile form at THREE workgroups per CU instead of two -- 1.10 ms, and 1.13 at two; V
// fragments from global memory +0.07..0.16 ms; 512-B store patches +0.04 ms.)
// VG: the V tile is not staged.  Both phases only need V's ROW fragments (dP = dO V^T), which are exactly what a lane loads
// from global memory with one 16-B access (phase A one key pair ahead: L1 / L2 hits after the first wave's).  Slower per
// block, but S = 160 (the cross-encoder's 158 tokens) then needs 74 KB instead of 93 KB of LDS = two workgroups per CU
// instead of one: 2.41 -> 1.58 ms on 2048 x 12 blocks.
//   * the pair iteration is ONE basic block: dropout on/off is a template parameter and "no ragged tail" selects one of two
//     copies of each phase, so the two key (query) tiles of a pair are scheduled together -- with the uniform branches inside,
//     hipcc emitted tile 0's MFMAs, waited for them, did its VALU work, and only then issued tile 1's;
//   * operands that are read with the same lane constant (K and V row fragments; Q and dO row and transpose fragments) sit in
//     adjacent LDS tiles and share one address register (the second through the instruction's immediate offset): 8
//     address additions per iteration instead of 20;
//   * 1/sqrt(d) multiplies the finished dQ / dK tiles, not every dS element.
template <typename F, int NKT, bool VG, bool DROP>
__global__ __launch_bounds__(256) void mha_bwd2_h16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ O,
                                                            const float* __restrict__ lse, const bf16_t* __restrict__ dO,
                                                            bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                            int heads, int T, float scale, DropCtx drop, int hm_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const QkvLay lay = qkv_lay(heads, hm_rows);
  const int H3 = lay.ld;                                  // row pitch of q / k / v and dq / dk / dv in either layout
  const bf16_t* Qg = qkv_head(qkv, heads, h, t0, hm_rows);
  const bf16_t* Kg = Qg + lay.ws;
  const bf16_t* Vg = Kg + lay.ws;
  bf16_t* dQg = qkv_head(dqkv, heads, h, t0, hm_rows);
  const bf16_t* Og = O + (long)t0 * H + h * 64;
  const bf16_t* dOg = dO + (long)t0 * H + h * 64;
  const int nkt = (len + 15) >> 4;
  const int nkt2 = (nkt + 1) & ~1;
  constexpr int TILE = NKT * 16 * 128;
  constexpr int OFF_Q = 0, OFF_D = TILE, OFF_K = 2 * TILE, OFF_V = 3 * TILE;      // (the V tile is last: absent with VG)
  constexpr int VEC = (VG ? 3 : 4) * TILE;           // lse[NKT*16], delta[NKT*16] (f32)
  constexpr int MSK = VEC + 2 * NKT * 16 * 4;        // dropout bits [key tile][query]: 16 bits = the keys of the tile
  constexpr int PATCH = MSK + NKT * 16 * NKT * 2;
  float* sLse = reinterpret_cast<float*>(smem + VEC);
  float* sDel = sLse + NKT * 16;
  char* patch = smem + PATCH + wave * 2048;
  att_stage(Qg, H3, len, nkt2 * 16, smem + OFF_Q, wave, lane);
  att_stage(Kg, H3, len, nkt2 * 16, smem + OFF_K, wave, lane);
  if (!VG) att_stage(Vg, H3, len, nkt2 * 16, smem + OFF_V, wave, lane);
  att_stage(dOg, H, len, nkt2 * 16, smem + OFF_D, wave, lane);
  // delta_i = dO_i . O_i (two threads per row, 16-B loads) ; lse_i in log2 units
  for (int idx = tid; idx < nkt2 * 32; idx += 256) {
    const int r = idx >> 1, half = idx & 1;
    float del = 0.f;
    if (r < len) {
      const bf16_t* po = Og + (long)r * H + half * 32;
      const bf16_t* pd = dOg + (long)r * H + half * 32;
#pragma unroll
      for (int c = 0; c < 32; c += 8) {
        const uint4 a = *reinterpret_cast<const uint4*>(po + c), b = *reinterpret_cast<const uint4*>(pd + c);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          del += H16<F>::lo(aw[e]) * H16<F>::lo(bw[e]) + H16<F>::hi(aw[e]) * H16<F>::hi(bw[e]);
      }
    }
    del += __shfl_xor(del, 1, 64);
    if (half == 0) {
      sDel[r] = del;
      sLse[r] = r < len ? lse[(long)h * T + t0 + r] * LOG2E : 0.f;
    }
  }
  if (DROP) {                                        // keep-bits of (query q, keys kt*16 .. +15): wave -> key tile, lane -> query
    for (int kt = wave; kt < nkt2; kt += 4)
      for (int q = lane; q < nkt * 16; q += 64) {
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t hsh = drop_mix(drop.seed, drop.stream, (uint32_t)(h * T + t0 + q), (uint32_t)(kt * 4 + j));
          w |= drop_keep4(hsh, drop.thr) << (4 * j);
        }
        *reinterpret_cast<unsigned short*>(smem + MSK + (kt * NKT * 16 + q) * 2) = (unsigned short)w;
      }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * LOG2E;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t patch_addr = lds0 + (uint32_t)(PATCH + wave * 2048);
  // lane constants: row-fragment offsets (chunks fg and 4+fg of row fr) and transpose-fragment offsets per 16-column tile
  const int fsw = att_f(fr);
  const uint32_t rf_lo = (uint32_t)(fr * 128 + ((fg ^ fsw) << 4)), rf_hi = (uint32_t)(fr * 128 + (((4 + fg) ^ fsw) << 4));
  const int rr = 4 * fg + (fr >> 2), tsw = att_f(rr), tx = (fr & 3) >> 1;
  uint32_t tr[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) tr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  const bool full = (len == nkt2 * 16);             // no ragged tail: the copies of the phases without per-element masks
  const int npair = nkt2 >> 1;
  // V row fragments of key pair KP from global memory (VG; rows past the end: copies of the last row)
#define A2_VLOAD(KP, V00, V01, V10, V11)                                                                   \
  do {                                                                                                     \
    const int r0__ = (KP) * 32 + fr, r1__ = r0__ + 16;                                                     \
    const bf16_t* p0__ = Vg + (long)(r0__ < len ? r0__ : len - 1) * H3 + fg * 8;                           \
    const bf16_t* p1__ = Vg + (long)(r1__ < len ? r1__ : len - 1) * H3 + fg * 8;                           \
    V00 = *reinterpret_cast<const bf16x8*>(p0__); V01 = *reinterpret_cast<const bf16x8*>(p0__ + 32);      \
    V10 = *reinterpret_cast<const bf16x8*>(p1__); V11 = *reinterpret_cast<const bf16x8*>(p1__ + 32);      \
  } while (0)

  // ---------------- phase A: dQ, waves own query tiles, loop over key-tile pairs
  auto phase_a = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    for (int qt = wave; qt < nkt; qt += 4) {
      const int q = qt * 16 + fr;
      bf16x8 qf0, qf1, df0, df1;
      {
        const uint32_t aq = lds0 + (uint32_t)(OFF_Q + qt * 2048);
        A2_RD128O(qf0, aq + rf_lo, 0); A2_RD128O(qf1, aq + rf_hi, 0);
        A2_RD128O(df0, aq + rf_lo, OFF_D - OFF_Q); A2_RD128O(df1, aq + rf_hi, OFF_D - OFF_Q);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qf0), "+v"(qf1), "+v"(df0), "+v"(df1)::"memory");
      }
      const float lq = sLse[q], dq_ = sDel[q];
      const bool qok = q < len;
      f32x4 dq[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      bf16x8 vn00, vn01, vn10, vn11;
      if (VG) A2_VLOAD(0, vn00, vn01, vn10, vn11);
      for (int kp = 0; kp < npair; ++kp) {
        const uint32_t po = (uint32_t)(OFF_K + kp * 4096);
        const uint32_t a_lo = lds0 + rf_lo + po, a_hi = lds0 + rf_hi + po;
        const uint32_t a_t0 = lds0 + tr[0] + po, a_t1 = lds0 + tr[1] + po, a_t2 = lds0 + tr[2] + po, a_t3 = lds0 + tr[3] + po;
        bf16x8 k00, k01, k10, k11, v00, v01, v10, v11;
        bf16x4 t0l, t0h, t1l, t1h, t2l, t2h, t3l, t3h;
        A2_RD128O(k00, a_lo, 0); A2_RD128O(k01, a_hi, 0); A2_RD128O(k10, a_lo, 2048); A2_RD128O(k11, a_hi, 2048);
        if (VG) {
          v00 = vn00; v01 = vn01; v10 = vn10; v11 = vn11;
          A2_VLOAD(kp + 1 < npair ? kp + 1 : kp, vn00, vn01, vn10, vn11);
        } else {
          A2_RD128O(v00, a_lo, OFF_V - OFF_K); A2_RD128O(v01, a_hi, OFF_V - OFF_K);
          A2_RD128O(v10, a_lo, OFF_V - OFF_K + 2048); A2_RD128O(v11, a_hi, OFF_V - OFF_K + 2048);
        }
        A2_RDTRO(t0l, a_t0, 0); A2_RDTRO(t0h, a_t0, 2048); A2_RDTRO(t1l, a_t1, 0); A2_RDTRO(t1h, a_t1, 2048);
        A2_RDTRO(t2l, a_t2, 0); A2_RDTRO(t2h, a_t2, 2048); A2_RDTRO(t3l, a_t3, 0); A2_RDTRO(t3h, a_t3, 2048);
        uint32_t mwa0 = 0xFFFFu, mwa1 = 0xFFFFu;
        if (DROP) {
          const uint32_t ma = lds0 + (uint32_t)(MSK + (kp * 2 * NKT * 16 + q) * 2);
          asm volatile("ds_read_u16 %0, %2\n\tds_read_u16 %1, %2 offset:%3" : "=&v"(mwa0), "=&v"(mwa1) : "v"(ma), "n"(NKT * 16 * 2) : "memory");
        }
        // (ONE wait names every register an asm read above is still filling: one it does not name may be copied before it)
        if (VG) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k00), "+v"(k01), "+v"(k10), "+v"(k11), "+v"(mwa0), "+v"(mwa1), "+v"(t0l), "+v"(t0h), "+v"(t1l), "+v"(t1h), "+v"(t2l), "+v"(t2h), "+v"(t3l), "+v"(t3h)::"memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k00), "+v"(k01), "+v"(k10), "+v"(k11), "+v"(v00), "+v"(v01), "+v"(v10), "+v"(v11), "+v"(mwa0), "+v"(mwa1),
                          "+v"(t0l), "+v"(t0h), "+v"(t1l), "+v"(t1h), "+v"(t2l), "+v"(t2h), "+v"(t3l), "+v"(t3h)::"memory");
        const uint32_t mqa = ((mwa0 & 0xFFFFu) | (mwa1 << 16)) >> (4 * fg);      // bit r: key 4 fg + r of tile 2 kp ; bit 16 + r: of tile 2 kp + 1
        f32x4 ds[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int kt = 2 * kp + hf;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = H16<F>::mfma(hf ? k10 : k00, qf0, s);
          s = H16<F>::mfma(hf ? k11 : k01, qf1, s);
          dp = H16<F>::mfma(hf ? v10 : v00, df0, dp);
          dp = H16<F>::mfma(hf ? v11 : v01, df1, dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(s[r] * c2 - lq);      // raw v_exp_f32: argument <= ~0, underflow -> 0
            if (!FULL) p = (kt * 16 + 4 * fg + r < len && qok) ? p : 0.f;
            float dpm = dp[r];
            if (DROP) dpm = ((mqa >> (hf * 16 + r)) & 1u) ? dpm * drop.scale : 0.f;
            ds[hf][r] = p * (dpm - dq_);
          }
        }
        const bf16x8 dsf = pack8<F>(ds[0], ds[1]);
        dq[0] = H16<F>::mfma(A2_CAT(t0l, t0h), dsf, dq[0]);
        dq[1] = H16<F>::mfma(A2_CAT(t1l, t1h), dsf, dq[1]);
        dq[2] = H16<F>::mfma(A2_CAT(t2l, t2h), dsf, dq[2]);
        dq[3] = H16<F>::mfma(A2_CAT(t3l, t3h), dsf, dq[3]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) dq[dt][e] *= scale;
      a2_store_tile<F>(dq, patch, patch_addr, dQg + (long)(qt * 16) * H3, H3, len - qt * 16, lane);
    }
  };
  if (full) phase_a(std::true_type{}); else phase_a(std::false_type{});
#undef A2_VLOAD

  // ---------------- phase B: dK, dV, waves own key tiles, loop over query-tile pairs
  auto phase_b = [&](auto full_c) {
    constexpr bool FULL = decltype(full_c)::value;
    for (int kt = wave; kt < nkt; kt += 4) {
      const int key = kt * 16 + fr;
      bf16x8 kf0, kf1, vf0, vf1;
      {
        const uint32_t ak = lds0 + (uint32_t)(OFF_K + kt * 2048);
        A2_RD128O(kf0, ak + rf_lo, 0); A2_RD128O(kf1, ak + rf_hi, 0);
        if (VG) {
          const bf16_t* pv = Vg + (long)(key < len ? key : len - 1) * H3 + fg * 8;
          vf0 = *reinterpret_cast<const bf16x8*>(pv); vf1 = *reinterpret_cast<const bf16x8*>(pv + 32);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf0), "+v"(kf1)::"memory");
        } else {
          A2_RD128O(vf0, ak + rf_lo, OFF_V - OFF_K); A2_RD128O(vf1, ak + rf_hi, OFF_V - OFF_K);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf0), "+v"(kf1), "+v"(vf0), "+v"(vf1)::"memory");
        }
      }
      const bool kok = key < len;
      f32x4 dk[4], dv[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
      for (int qp = 0; qp < npair; ++qp) {
        const uint32_t po = (uint32_t)(OFF_Q + qp * 4096);
        const uint32_t a_lo = lds0 + rf_lo + po, a_hi = lds0 + rf_hi + po;
        const uint32_t a_t0 = lds0 + tr[0] + po, a_t1 = lds0 + tr[1] + po, a_t2 = lds0 + tr[2] + po, a_t3 = lds0 + tr[3] + po;
        const uint32_t bl = lds0 + (uint32_t)(VEC + (qp * 32 + 4 * fg) * 4);     // sLse[qp*32 + 4 fg ..], sDel = + NKT*64 B
        bf16x8 q00, q01, q10, q11, d00, d01, d10, d11;
        bf16x4 e0l, e0h, e1l, e1h, e2l, e2h, e3l, e3h, u0l, u0h, u1l, u1h, u2l, u2h, u3l, u3h;
        f32x4 ls0, ls1, de0, de1;
        uint2 mb[2] = {make_uint2(~0u, ~0u), make_uint2(~0u, ~0u)};          // keep-bits of the lane's four query rows (16 bits each), per tile of the pair
        A2_RD128O(q00, a_lo, 0); A2_RD128O(q01, a_hi, 0); A2_RD128O(q10, a_lo, 2048); A2_RD128O(q11, a_hi, 2048);
        A2_RD128O(d00, a_lo, OFF_D - OFF_Q); A2_RD128O(d01, a_hi, OFF_D - OFF_Q);
        A2_RD128O(d10, a_lo, OFF_D - OFF_Q + 2048); A2_RD128O(d11, a_hi, OFF_D - OFF_Q + 2048);
        A2_RD128O(ls0, bl, 0); A2_RD128O(ls1, bl, 64); A2_RD128O(de0, bl, NKT * 64); A2_RD128O(de1, bl, NKT * 64 + 64);
        if (DROP) {
          const uint32_t ma = lds0 + (uint32_t)(MSK + (kt * NKT * 16 + qp * 32 + 4 * fg) * 2);
          asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:32" : "=&v"(mb[0]), "=&v"(mb[1]) : "v"(ma) : "memory");
        }
        A2_RDTRO(e0l, a_t0, OFF_D - OFF_Q); A2_RDTRO(e0h, a_t0, OFF_D - OFF_Q + 2048); A2_RDTRO(e1l, a_t1, OFF_D - OFF_Q); A2_RDTRO(e1h, a_t1, OFF_D - OFF_Q + 2048);
        A2_RDTRO(e2l, a_t2, OFF_D - OFF_Q); A2_RDTRO(e2h, a_t2, OFF_D - OFF_Q + 2048); A2_RDTRO(e3l, a_t3, OFF_D - OFF_Q); A2_RDTRO(e3h, a_t3, OFF_D - OFF_Q + 2048);
        A2_RDTRO(u0l, a_t0, 0); A2_RDTRO(u0h, a_t0, 2048); A2_RDTRO(u1l, a_t1, 0); A2_RDTRO(u1h, a_t1, 2048);
        A2_RDTRO(u2l, a_t2, 0); A2_RDTRO(u2h, a_t2, 2048); A2_RDTRO(u3l, a_t3, 0); A2_RDTRO(u3h, a_t3, 2048);
        // (ONE wait names all 30 registers in flight -- the operand limit of an asm statement)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q00), "+v"(q01), "+v"(q10), "+v"(q11), "+v"(d00), "+v"(d01), "+v"(d10), "+v"(d11),
                     "+v"(ls0), "+v"(ls1), "+v"(de0), "+v"(de1), "+v"(mb[0]), "+v"(mb[1]),
                     "+v"(e0l), "+v"(e0h), "+v"(e1l), "+v"(e1h), "+v"(e2l), "+v"(e2h), "+v"(e3l), "+v"(e3h),
                     "+v"(u0l), "+v"(u0h), "+v"(u1l), "+v"(u1h), "+v"(u2l), "+v"(u2h), "+v"(u3l), "+v"(u3h)::"memory");
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int qt = 2 * qp + hf;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = H16<F>::mfma(hf ? q10 : q00, kf0, s);
          s = H16<F>::mfma(hf ? q11 : q01, kf1, s);
          dp = H16<F>::mfma(hf ? d10 : d00, vf0, dp);
          dp = H16<F>::mfma(hf ? d11 : d01, vf1, dp);
          const f32x4 lsv = hf ? ls1 : ls0, dev = hf ? de1 : de0;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(s[r] * c2 - lsv[r]);
            if (!FULL) p = (qt * 16 + 4 * fg + r < len && kok) ? p : 0.f;
            float pm = p, dpm = dp[r];
            if (DROP) {
              const uint32_t mword = r < 2 ? mb[hf].x : mb[hf].y;
              const bool keep = ((mword >> ((r & 1) * 16 + fr)) & 1u) != 0;
              pm = keep ? p * drop.scale : 0.f;
              dpm = keep ? dpm * drop.scale : 0.f;
            }
            pp[hf][r] = pm;
            ds[hf][r] = p * (dpm - dev[r]);
          }
        }
        const bf16x8 pf = pack8<F>(pp[0], pp[1]);
        const bf16x8 dsf = pack8<F>(ds[0], ds[1]);
        dv[0] = H16<F>::mfma(A2_CAT(e0l, e0h), pf, dv[0]);
        dk[0] = H16<F>::mfma(A2_CAT(u0l, u0h), dsf, dk[0]);
        dv[1] = H16<F>::mfma(A2_CAT(e1l, e1h), pf, dv[1]);
        dk[1] = H16<F>::mfma(A2_CAT(u1l, u1h), dsf, dk[1]);
        dv[2] = H16<F>::mfma(A2_CAT(e2l, e2h), pf, dv[2]);
        dk[2] = H16<F>::mfma(A2_CAT(u2l, u2h), dsf, dk[2]);
        dv[3] = H16<F>::mfma(A2_CAT(e3l, e3h), pf, dv[3]);
        dk[3] = H16<F>::mfma(A2_CAT(u3l, u3h), dsf, dk[3]);
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int e = 0; e < 4; ++e) dk[dt][e] *= scale;
      bf16_t* dstk = dQg + lay.ws + (long)(kt * 16) * H3;
      a2_store_tile<F>(dk, patch, patch_addr, dstk, H3, len - kt * 16, lane);
      a2_store_tile<F>(dv, patch, patch_addr, dstk + lay.ws, H3, len - kt * 16, lane);
    }
  };
  if (full) phase_b(std::true_type{}); else phase_b(std::false_type{});
}

// ------------------------------------------------------------------------------------------ backward, long sequences
// 256 < S <= 4096 (MS-Doc documents and BASELINE config 5 run at S = 512; the generic kernel took 95 % of a
// fwd+bwd step there).  Q, K, V, dO of a whole sequence no longer fit the LDS, so the sequence is cut into chunks of
// CH tokens (128 as launched) and the two gradients are computed by two launches of one template:
//   DKV = false: block (seq, head, query chunk) keeps Q, dO of its chunk resident, walks the key chunks (K, V
//                restaged per chunk) and accumulates dQ of its four query tiles per wave in registers;
//   DKV = true : block (seq, head, key chunk) keeps K, V resident, walks the query chunks (Q, dO, lse, delta restaged)
//                and accumulates dK, dV of its four key tiles per wave in registers (one wave per SIMD: 128 KB of LDS per
//                block anyway, so the 512-register budget is there).
// No atomics, no f32 scratch; the inner pair iteration is the one of mha_bwd2_bf16_kernel.

// (dropout on / off is a template parameter and "both chunks complete" selects one of two copies of the tile loop, so that the pair
// iteration is one basic block; dK / dV read the dropout keep-bits of the (key chunk, query chunk) pair from the LDS, hashed once
// while the query chunk is staged: see mha_bwd2_h16_kernel)
// CH: tokens per chunk.  256: one workgroup per CU (4 tiles of 32 KB), four tiles per wave; 128: two workgroups per CU, two tiles per wave.
template <typename F, bool DKV, bool DROP, int CH>
__global__ __launch_bounds__(256, (CH == 128 ? 2 : 1)) void mha_bwd_long_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ O,
                                                              const float* __restrict__ lse, const bf16_t* __restrict__ dO,
                                                              bf16_t* __restrict__ dqkv, const int* __restrict__ cu,
                                                              int heads, int T, int nchunk, float scale, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mine = blockIdx.x % nchunk;                     // the chunk this block owns (queries for dQ, keys for dK/dV)
  const int sh = blockIdx.x / nchunk;
  const int seq = sh / heads, h = sh % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  const int own0 = mine * CH;
  if (own0 >= len) return;
  const int ownlen = min(CH, len - own0);
  const int H = heads * 64, H3 = 3 * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const bf16_t* Og = O + (long)t0 * H + h * 64;
  const bf16_t* dOg = dO + (long)t0 * H + h * 64;
  char* sQ = smem;
  char* sK = smem + (CH * 128);
  char* sV = smem + 2 * (CH * 128);
  char* sD = smem + 3 * (CH * 128);
  float* sLse = reinterpret_cast<float*>(smem + 4 * (CH * 128));
  float* sDel = sLse + CH;
  char* patch = smem + 4 * (CH * 128) + 2 * CH * 4 + wave * 2048;
  constexpr int MSK = 4 * (CH * 128) + 2 * CH * 4 + 4 * 2048;   // dropout bits [key tile of the own chunk][query of the staged chunk]
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t patch_addr = lds0 + (uint32_t)(4 * (CH * 128) + 2 * CH * 4 + wave * 2048);
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * LOG2E;
  const int fsw = att_f(fr);
  const uint32_t rf_lo = (uint32_t)(fr * 128 + ((fg ^ fsw) << 4)), rf_hi = (uint32_t)(fr * 128 + (((4 + fg) ^ fsw) << 4));
  const int rr = 4 * fg + (fr >> 2), tsw = att_f(rr), tx = (fr & 3) >> 1;
  uint32_t tr[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) tr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);

  // stage rows [r0, r0+n) of the query side (Q, dO, lse, delta) / key side (K, V); rows padded to a multiple of 32
  auto stage_q = [&](int r0, int n) {
    const int npad = ((n + 31) >> 5) << 5;
    att_stage(Qg + (long)r0 * H3, H3, n, npad, sQ, wave, lane);
    att_stage(dOg + (long)r0 * H, H, n, npad, sD, wave, lane);
    for (int idx = tid; idx < npad * 2; idx += 256) {
      const int r = idx >> 1, half = idx & 1;
      float del = 0.f;
      if (r < n) {
        const bf16_t* po = Og + (long)(r0 + r) * H + half * 32;
        const bf16_t* pd = dOg + (long)(r0 + r) * H + half * 32;
#pragma unroll
        for (int c = 0; c < 32; c += 8) {
          const uint4 a = *reinterpret_cast<const uint4*>(po + c), b = *reinterpret_cast<const uint4*>(pd + c);
          const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
          for (int e = 0; e < 4; ++e)
            del += H16<F>::lo(aw[e]) * H16<F>::lo(bw[e]) + H16<F>::hi(aw[e]) * H16<F>::hi(bw[e]);
        }
      }
      del += __shfl_xor(del, 1, 64);
      if (half == 0) {
        sDel[r] = del;
        sLse[r] = r < n ? lse[(long)h * T + t0 + r0 + r] * LOG2E : 0.f;
      }
    }
    if (DKV && DROP) {                               // keep-bits of (query r0 + q, keys own0 + kt*16 .. +15): wave -> key tile, lane -> query
      for (int kt = wave; kt < CH / 16; kt += 4)
        for (int q = lane; q < npad; q += 64) {
          uint32_t w = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t hsh = drop_mix(drop.seed, drop.stream, (uint32_t)(h * T + t0 + r0 + q), (uint32_t)((own0 >> 2) + kt * 4 + j));
            w |= drop_keep4(hsh, drop.thr) << (4 * j);
          }
          *reinterpret_cast<unsigned short*>(smem + MSK + (kt * CH + q) * 2) = (unsigned short)w;
        }
    }
  };
  auto stage_k = [&](int r0, int n) {
    const int npad = ((n + 31) >> 5) << 5;
    att_stage(Kg + (long)r0 * H3, H3, n, npad, sK, wave, lane);
    att_stage(Vg + (long)r0 * H3, H3, n, npad, sV, wave, lane);
  };

  if (!DKV) {
    // ------------------------------------------------ dQ of query chunk `mine`
    stage_q(own0, ownlen);
    f32x4 dq[CH / 64][4];
#pragma unroll
    for (int t = 0; t < CH / 64; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nqt = (ownlen + 15) >> 4;
    for (int kc = 0; kc * CH < len; ++kc) {
      const int k0 = kc * CH, klen = min(CH, len - k0);
      const int nkp = (((klen + 15) >> 4) + 1) >> 1;
      __syncthreads();                                     // everyone is done with the previous K, V chunk
      stage_k(k0, klen);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool full = (ownlen == CH) && (klen == CH);
      auto tiles = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int t = 0; t < CH / 64; ++t) {
        const int qt = wave + 4 * t;
        if (qt >= nqt) continue;                           // wave-uniform
        const int q = qt * 16 + fr;
        bf16x8 qf0, qf1, df0, df1;
        {
          const uint32_t aq = lds0 + (uint32_t)(qt * 2048), ad = aq + 3 * (CH * 128);
          A2_RD128(qf0, aq + rf_lo, 0); A2_RD128(qf1, aq + rf_hi, 0);
          A2_RD128(df0, ad + rf_lo, 0); A2_RD128(df1, ad + rf_hi, 0);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qf0), "+v"(qf1), "+v"(df0), "+v"(df1)::"memory");
        }
        const float lq = sLse[q], dq_ = sDel[q];
        const bool qok = q < ownlen;
        for (int kp = 0; kp < nkp; ++kp) {
          const uint32_t bk = lds0 + (uint32_t)((CH * 128) + kp * 4096), bv = bk + (CH * 128);
          bf16x8 k00, k01, k10, k11, v00, v01, v10, v11;
          bf16x4 t0l, t0h, t1l, t1h, t2l, t2h, t3l, t3h;
          A2_RD128(k00, bk + rf_lo, 0); A2_RD128(k01, bk + rf_hi, 0); A2_RD128(k10, bk + rf_lo, 2048); A2_RD128(k11, bk + rf_hi, 2048);
          A2_RD128(v00, bv + rf_lo, 0); A2_RD128(v01, bv + rf_hi, 0); A2_RD128(v10, bv + rf_lo, 2048); A2_RD128(v11, bv + rf_hi, 2048);
          A2_RDTR(t0l, bk + tr[0], 0); A2_RDTR(t0h, bk + tr[0], 2048); A2_RDTR(t1l, bk + tr[1], 0); A2_RDTR(t1h, bk + tr[1], 2048);
          A2_RDTR(t2l, bk + tr[2], 0); A2_RDTR(t2h, bk + tr[2], 2048); A2_RDTR(t3l, bk + tr[3], 0); A2_RDTR(t3h, bk + tr[3], 2048);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k00), "+v"(k01), "+v"(k10), "+v"(k11), "+v"(v00), "+v"(v01), "+v"(v10), "+v"(v11),
                       "+v"(t0l), "+v"(t0h), "+v"(t1l), "+v"(t1h), "+v"(t2l), "+v"(t2h), "+v"(t3l), "+v"(t3h)::"memory");
          f32x4 ds[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int kt = 2 * kp + hf;
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = H16<F>::mfma(hf ? k10 : k00, qf0, s);
            s = H16<F>::mfma(hf ? k11 : k01, qf1, s);
            dp = H16<F>::mfma(hf ? v10 : v00, df0, dp);
            dp = H16<F>::mfma(hf ? v11 : v01, df1, dp);
            float m4[4] = {1.f, 1.f, 1.f, 1.f};
            if (DROP) drop_mult4(drop, (uint32_t)(h * T + t0 + own0 + q), (uint32_t)(k0 + kt * 16 + 4 * fg), m4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = __builtin_amdgcn_exp2f(s[r] * c2 - lq);
              if (!FULL) p = (kt * 16 + 4 * fg + r < klen && qok) ? p : 0.f;
              ds[hf][r] = p * (dp[r] * m4[r] - dq_) * scale;
            }
          }
          const bf16x8 dsf = pack8<F>(ds[0], ds[1]);
          dq[t][0] = H16<F>::mfma(A2_CAT(t0l, t0h), dsf, dq[t][0]);
          dq[t][1] = H16<F>::mfma(A2_CAT(t1l, t1h), dsf, dq[t][1]);
          dq[t][2] = H16<F>::mfma(A2_CAT(t2l, t2h), dsf, dq[t][2]);
          dq[t][3] = H16<F>::mfma(A2_CAT(t3l, t3h), dsf, dq[t][3]);
        }
      }
      };
      if (full) tiles(std::true_type{}); else tiles(std::false_type{});
    }
#pragma unroll
    for (int t = 0; t < CH / 64; ++t) {
      const int qt = wave + 4 * t;
      if (qt < nqt)
        a2_store_tile<F>(dq[t], patch, patch_addr, dqkv + (long)(t0 + own0 + qt * 16) * H3 + h * 64, H3, ownlen - qt * 16, lane);
    }
  } else {
    // ------------------------------------------------ dK, dV of key chunk `mine`
    stage_k(own0, ownlen);
    f32x4 dk[CH / 64][4], dv[CH / 64][4];
#pragma unroll
    for (int t = 0; t < CH / 64; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) { dk[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int nkt = (ownlen + 15) >> 4;
    for (int qc = 0; qc * CH < len; ++qc) {
      const int q0 = qc * CH, qlen = min(CH, len - q0);
      const int nqp = (((qlen + 15) >> 4) + 1) >> 1;
      __syncthreads();
      stage_q(q0, qlen);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const bool full = (ownlen == CH) && (qlen == CH);
      auto tiles = [&](auto full_c) {
      constexpr bool FULL = decltype(full_c)::value;
#pragma unroll
      for (int t = 0; t < CH / 64; ++t) {
        const int kt = wave + 4 * t;
        if (kt >= nkt) continue;
        const int key = kt * 16 + fr;
        bf16x8 kf0, kf1, vf0, vf1;
        {
          const uint32_t ak = lds0 + (uint32_t)((CH * 128) + kt * 2048), av = ak + (CH * 128);
          A2_RD128(kf0, ak + rf_lo, 0); A2_RD128(kf1, ak + rf_hi, 0);
          A2_RD128(vf0, av + rf_lo, 0); A2_RD128(vf1, av + rf_hi, 0);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf0), "+v"(kf1), "+v"(vf0), "+v"(vf1)::"memory");
        }
        const bool kok = key < ownlen;
        for (int qp = 0; qp < nqp; ++qp) {
          // (dO tile = Q tile + 3 * CH * 128 bytes: as an instruction offset where it encodes -- 16 bits, the 128-token chunks -- so
          // that Q and dO requests share their address registers)
          constexpr int DOFF = CH == 128 ? 3 * (CH * 128) : 0;
          const uint32_t bq = lds0 + (uint32_t)(qp * 4096), bd = bq + (uint32_t)(3 * (CH * 128) - DOFF);
          const uint32_t bl = lds0 + (uint32_t)(4 * (CH * 128) + (qp * 32 + 4 * fg) * 4);
          bf16x8 q00, q01, q10, q11, d00, d01, d10, d11;
          bf16x4 e0l, e0h, e1l, e1h, e2l, e2h, e3l, e3h, u0l, u0h, u1l, u1h, u2l, u2h, u3l, u3h;
          f32x4 ls0, ls1, de0, de1;
          uint2 mb[2] = {make_uint2(~0u, ~0u), make_uint2(~0u, ~0u)};          // keep-bits of the lane's four query rows (16 bits each), per tile of the pair
          A2_RD128(q00, bq + rf_lo, 0); A2_RD128(q01, bq + rf_hi, 0); A2_RD128(q10, bq + rf_lo, 2048); A2_RD128(q11, bq + rf_hi, 2048);
          A2_RD128O(d00, bd + rf_lo, DOFF); A2_RD128O(d01, bd + rf_hi, DOFF); A2_RD128O(d10, bd + rf_lo, DOFF + 2048); A2_RD128O(d11, bd + rf_hi, DOFF + 2048);
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:64"
                       : "=&v"(ls0), "=&v"(ls1), "=&v"(de0), "=&v"(de1) : "v"(bl), "v"(bl + (uint32_t)(CH * 4)) : "memory");
          if (DROP) {
            const uint32_t ma = lds0 + (uint32_t)(MSK + (kt * CH + qp * 32 + 4 * fg) * 2);
            asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:32" : "=&v"(mb[0]), "=&v"(mb[1]) : "v"(ma) : "memory");
          }
          // two LDS phases (register budget: 256 per wave at two workgroups per CU): the Q / dO fragments are consumed by the
          // S and dP MFMAs before the transposed tiles of the same pair are requested, so the two sets never live together;
          // the transposed reads are in flight under the exp2 / dS arithmetic
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q00), "+v"(q01), "+v"(q10), "+v"(q11), "+v"(d00), "+v"(d01), "+v"(d10), "+v"(d11),
                       "+v"(ls0), "+v"(ls1), "+v"(de0), "+v"(de1), "+v"(mb[0]), "+v"(mb[1])::"memory");
          f32x4 sv[2], dpv[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            s = H16<F>::mfma(hf ? q10 : q00, kf0, s);
            s = H16<F>::mfma(hf ? q11 : q01, kf1, s);
            dp = H16<F>::mfma(hf ? d10 : d00, vf0, dp);
            dp = H16<F>::mfma(hf ? d11 : d01, vf1, dp);
            sv[hf] = s; dpv[hf] = dp;
          }
          // (the empty asm orders the requests after the MFMA issue: it names the accumulators, it does not read them)
          asm volatile("" : "+v"(sv[0]), "+v"(sv[1]), "+v"(dpv[0]), "+v"(dpv[1]));
          A2_RDTRO(e0l, bd + tr[0], DOFF); A2_RDTRO(e0h, bd + tr[0], DOFF + 2048); A2_RDTRO(e1l, bd + tr[1], DOFF); A2_RDTRO(e1h, bd + tr[1], DOFF + 2048);
          A2_RDTRO(e2l, bd + tr[2], DOFF); A2_RDTRO(e2h, bd + tr[2], DOFF + 2048); A2_RDTRO(e3l, bd + tr[3], DOFF); A2_RDTRO(e3h, bd + tr[3], DOFF + 2048);
          A2_RDTR(u0l, bq + tr[0], 0); A2_RDTR(u0h, bq + tr[0], 2048); A2_RDTR(u1l, bq + tr[1], 0); A2_RDTR(u1h, bq + tr[1], 2048);
          A2_RDTR(u2l, bq + tr[2], 0); A2_RDTR(u2h, bq + tr[2], 2048); A2_RDTR(u3l, bq + tr[3], 0); A2_RDTR(u3h, bq + tr[3], 2048);
          f32x4 pp[2], ds[2];
#pragma unroll
          for (int hf = 0; hf < 2; ++hf) {
            const int qt = 2 * qp + hf;
            const f32x4 s = sv[hf], dp = dpv[hf];
            const f32x4 lsv = hf ? ls1 : ls0, dev = hf ? de1 : de0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float p = __builtin_amdgcn_exp2f(s[r] * c2 - lsv[r]);
              if (!FULL) p = (qt * 16 + 4 * fg + r < qlen && kok) ? p : 0.f;
              float mm = 1.f;
              if (DROP) {
                const uint32_t mword = r < 2 ? mb[hf].x : mb[hf].y;
                mm = ((mword >> ((r & 1) * 16 + fr)) & 1u) ? drop.scale : 0.f;
              }
              pp[hf][r] = p * mm;
              ds[hf][r] = p * (dp[r] * mm - dev[r]) * scale;
            }
          }
          bf16x8 pf = pack8<F>(pp[0], pp[1]);
          bf16x8 dsf = pack8<F>(ds[0], ds[1]);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(e0l), "+v"(e0h), "+v"(e1l), "+v"(e1h), "+v"(e2l), "+v"(e2h), "+v"(e3l), "+v"(e3h),
                       "+v"(u0l), "+v"(u0h), "+v"(u1l), "+v"(u1h), "+v"(u2l), "+v"(u2h), "+v"(u3l), "+v"(u3h), "+v"(pf), "+v"(dsf)::"memory");
          dv[t][0] = H16<F>::mfma(A2_CAT(e0l, e0h), pf, dv[t][0]);
          dk[t][0] = H16<F>::mfma(A2_CAT(u0l, u0h), dsf, dk[t][0]);
          dv[t][1] = H16<F>::mfma(A2_CAT(e1l, e1h), pf, dv[t][1]);
          dk[t][1] = H16<F>::mfma(A2_CAT(u1l, u1h), dsf, dk[t][1]);
          dv[t][2] = H16<F>::mfma(A2_CAT(e2l, e2h), pf, dv[t][2]);
          dk[t][2] = H16<F>::mfma(A2_CAT(u2l, u2h), dsf, dk[t][2]);
          dv[t][3] = H16<F>::mfma(A2_CAT(e3l, e3h), pf, dv[t][3]);
          dk[t][3] = H16<F>::mfma(A2_CAT(u3l, u3h), dsf, dk[t][3]);
        }
      }
      };
      if (full) tiles(std::true_type{}); else tiles(std::false_type{});
    }
    // (the store addresses are derived from a laundered copy of the lane id: computed here, not kept in registers across the
    // chunk loop -- the dK/dV copy with dropout has no register to spare at two workgroups per CU)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
#pragma unroll
    for (int t = 0; t < CH / 64; ++t) {
      const int kt = wave + 4 * t;
      if (kt < nkt) {
        bf16_t* dstk = dqkv + (long)(t0 + own0 + kt * 16) * H3 + H + h * 64;
        a2_store_tile<F>(dk[t], patch, patch_addr, dstk, H3, ownlen - kt * 16, lane_e);
        a2_store_tile<F>(dv[t], patch, patch_addr, dstk + H, H3, ownlen - kt * 16, lane_e);
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
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "%s: dtype %d", who, dtype);
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
extern "C" int simx_mha_fwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, void* ctx, float* lse, const simx_dropout* dropd, int hm_rows);
extern "C" int simx_mha_bwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv,
                               const simx_dropout* dropd, int hm_rows);
extern "C" int simx_mha_fwd(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                            int T, const void* qkv, void* ctx, float* lse) {
  return simx_mha_fwd_ex(stream, dtype, nseq, heads, d, cu, max_len, T, qkv, ctx, lse, nullptr);
}
extern "C" int simx_mha_bwd(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                            int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv) {
  return simx_mha_bwd_ex(stream, dtype, nseq, heads, d, cu, max_len, T, qkv, ctx, lse, dctx, dqkv, nullptr);
}

// fp32 engine with operand planes (csrc/gemm_xp.hip): f32 q/k/v in; the context leaves as the fp16 plane pair the
// attention-output GEMM stages, backward reads it (for rowsum(dO . O)) and writes dq/dk/dv as the bf16 plane pair of the QKV
// dgrad / wgrad GEMMs.  Head size 64, sequences <= 4096 (simx_mha_planes_ok).
extern "C" int simx_mha_planes_ok(int d, int max_len) { return simx_mha_f32_ok(d, max_len) ? 1 : 0; }
extern "C" int simx_mha_fwd_planes(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const float* qkv,
                                   void* ctx_planes, long ctx_plane_stride, float* lse, const simx_dropout* dropd) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_MHA_FWD, s, 4.0 * T * max_len * heads * d);
  int rc = check_common(SIMX_F32, nseq, heads, d, max_len, T, "mha_fwd_planes");
  if (rc) return rc;
  SIMX_REQUIRE(simx_mha_f32_ok(d, max_len) && ctx_plane_stride > 0 && ctx_plane_stride % 4 == 0, SIMX_ERR_UNSUPPORTED,
               "mha_fwd_planes: needs head size 64, max_len <= 4096 and a plane stride %% 4 == 0");
  return simx_mha_fwd_f32(s, nseq, heads, cu, max_len, T, qkv, (float*)ctx_planes, lse, 1.0f / sqrtf((float)d), make_drop(dropd), ctx_plane_stride);
}
extern "C" int simx_mha_bwd_planes(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const float* qkv,
                                   const void* ctx_planes, long ctx_plane_stride, const float* lse, const float* dctx, void* dqkv_planes,
                                   long dqkv_plane_stride, const simx_dropout* dropd) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_MHA_BWD, s, 8.0 * T * max_len * heads * d);
  int rc = check_common(SIMX_F32, nseq, heads, d, max_len, T, "mha_bwd_planes");
  if (rc) return rc;
  SIMX_REQUIRE(simx_mha_f32_ok(d, max_len) && ctx_plane_stride > 0 && ctx_plane_stride % 4 == 0 && dqkv_plane_stride > 0 && dqkv_plane_stride % 4 == 0,
               SIMX_ERR_UNSUPPORTED, "mha_bwd_planes: needs head size 64, max_len <= 4096 and plane strides %% 4 == 0");
  return simx_mha_bwd_f32(s, nseq, heads, cu, max_len, T, qkv, (const float*)ctx_planes, lse, dctx, (float*)dqkv_planes, 1.0f / sqrtf((float)d),
                          make_drop(dropd), ctx_plane_stride, dqkv_plane_stride);
}

extern "C" int simx_mha_fwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, void* ctx, float* lse, const simx_dropout* dropd) {
  return simx_mha_fwd_hm(stream, dtype, nseq, heads, d, cu, max_len, T, qkv, ctx, lse, dropd, 0);
}
extern "C" int simx_mha_fwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, void* ctx, float* lse, const simx_dropout* dropd, int hm_rows) {
  hipStream_t s = (hipStream_t)stream;
  // (the same limit as simx_mha_bwd_hm: a forward the library accepts must have a backward it accepts)
  SIMX_REQUIRE(hm_rows == 0 || (simx_is16(dtype) && d == 64 && max_len <= 256 && hm_rows >= T), SIMX_ERR_UNSUPPORTED,
               "mha_fwd: the head-major qkv layout needs a 16-bit dtype, head size 64, max_len <= 256 and hm_rows >= T");
  const DropCtx drop = make_drop(dropd);
  SIMX_PROF(SIMX_K_MHA_FWD, s, 4.0 * T * max_len * heads * d);
  int rc = check_common(dtype, nseq, heads, d, max_len, T, "mha_fwd");
  if (rc) return rc;
  const float scale = 1.0f / sqrtf((float)d);
  if (simx_is16(dtype) && d == 64 && max_len <= 256) {
#define LF(NKT)                                                                                                      \
  do {                                                                                                               \
    const size_t lds = (size_t)2 * NKT * 16 * 128;                                                                   \
    if (drop.thr) {                                                                                                  \
      rc = set_lds(mha_fwd_h16_kernel<FF, NKT, true>, lds, "mha_fwd");                                               \
      if (rc) return rc;                                                                                             \
      hipLaunchKernelGGL((mha_fwd_h16_kernel<FF, NKT, true>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv, \
                         (bf16_t*)ctx, lse, cu, heads, T, scale, drop, hm_rows);                                     \
    } else {                                                                                                         \
      rc = set_lds(mha_fwd_h16_kernel<FF, NKT, false>, lds, "mha_fwd");                                              \
      if (rc) return rc;                                                                                             \
      hipLaunchKernelGGL((mha_fwd_h16_kernel<FF, NKT, false>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv, \
                         (bf16_t*)ctx, lse, cu, heads, T, scale, drop, hm_rows);                                     \
    }                                                                                                                \
  } while (0)
#define LF_ALL()                                                                                                     \
  do {                                                                                                               \
    if (max_len <= 32) LF(2);                                                                                        \
    else if (max_len <= 128) LF(8);                                                                                  \
    else if (max_len <= 160) LF(10);                                                                                 \
    else LF(16);                                                                                                     \
  } while (0)
    SIMX_DISPATCH16(dtype, FF, LF_ALL());
#undef LF_ALL
#undef LF
    SIMX_CHECK_LAUNCH("mha_fwd_h16");
    return SIMX_OK;
  }
  if (simx_is16(dtype) && d == 64 && max_len <= 4096) {            // 256 < max_len: chunked (768 x 16 blocks of 512 tokens: 2.33 ms against 2.41
                                                                    // with K / V resident in 128 KB of LDS; 300 tokens: 1.12 against 1.91)
    const int nchunk = cdiv(max_len, 128);
#define LFL(DR) hipLaunchKernelGGL((mha_fwd_long_h16_kernel<FF, DR>), dim3(nseq * heads * nchunk), dim3(256), 32768, s, (const bf16_t*)qkv, \
                                   (bf16_t*)ctx, lse, cu, heads, T, nchunk, scale, drop, hm_rows)
    SIMX_DISPATCH16(dtype, FF, if (drop.thr) LFL(true); else LFL(false));
#undef LFL
    SIMX_CHECK_LAUNCH("mha_fwd_long_h16");
    return SIMX_OK;
  }
  if (dtype == SIMX_F32 && simx_mha_f32_ok(d, max_len))
    return simx_mha_fwd_f32(s, nseq, heads, cu, max_len, T, (const float*)qkv, (float*)ctx, lse, scale, drop);
  const size_t lds = (size_t)(4 * 128 + 4 * max_len) * sizeof(float);
  dim3 grid(nseq * heads, cdiv(max_len, 4));
#define LS(TT)                                                                                                       \
  do {                                                                                                               \
    rc = set_lds(mha_fwd_simple_kernel<TT>, lds, "mha_fwd");                                                         \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_fwd_simple_kernel<TT>), grid, dim3(256), lds, s, (const TT*)qkv, (TT*)ctx, lse, cu, heads, d, T, scale, \
                       max_len, drop);                                                                               \
  } while (0)
  SIMX_DISPATCH3(dtype, TT, LS(TT));
#undef LS
  SIMX_CHECK_LAUNCH("mha_fwd_simple");
  return SIMX_OK;
}

extern "C" int simx_mha_bwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv,
                               const simx_dropout* dropd) {
  return simx_mha_bwd_hm(stream, dtype, nseq, heads, d, cu, max_len, T, qkv, ctx, lse, dctx, dqkv, dropd, 0);
}
extern "C" int simx_mha_bwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                               int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv,
                               const simx_dropout* dropd, int hm_rows) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_REQUIRE(hm_rows == 0 || (simx_is16(dtype) && d == 64 && max_len <= 256 && hm_rows >= T), SIMX_ERR_UNSUPPORTED,
               "mha_bwd: the head-major qkv layout needs a 16-bit dtype, head size 64, max_len <= 256 and hm_rows >= T");
  const DropCtx drop = make_drop(dropd);
  SIMX_PROF(SIMX_K_MHA_BWD, s, 8.0 * T * max_len * heads * d);
  int rc = check_common(dtype, nseq, heads, d, max_len, T, "mha_bwd");
  if (rc) return rc;
  const float scale = 1.0f / sqrtf((float)d);
  if (simx_is16(dtype) && d == 64 && max_len <= 256) {               // (161-256 tokens on the chunked kernels instead: 4.20 vs 4.02 ms at 256, 3.60 vs 3.78 at 200 -- not taken)
#define LB(NKT, VG)                                                                                                  \
  do {                                                                                                               \
    const size_t lds = (size_t)((VG) ? 3 : 4) * NKT * 16 * 128 + 2 * NKT * 16 * sizeof(float) + (size_t)NKT * 16 * NKT * 2 + 4 * 2048; \
    if (drop.thr) {                                                                                                  \
      rc = set_lds(mha_bwd2_h16_kernel<FF, NKT, VG, true>, lds, "mha_bwd");                                          \
      if (rc) return rc;                                                                                             \
      hipLaunchKernelGGL((mha_bwd2_h16_kernel<FF, NKT, VG, true>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv, \
                         (const bf16_t*)ctx, lse, (const bf16_t*)dctx, (bf16_t*)dqkv, cu, heads, T, scale, drop, hm_rows); \
    } else {                                                                                                         \
      rc = set_lds(mha_bwd2_h16_kernel<FF, NKT, VG, false>, lds, "mha_bwd");                                         \
      if (rc) return rc;                                                                                             \
      hipLaunchKernelGGL((mha_bwd2_h16_kernel<FF, NKT, VG, false>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv, \
                         (const bf16_t*)ctx, lse, (const bf16_t*)dctx, (bf16_t*)dqkv, cu, heads, T, scale, drop, hm_rows); \
    }                                                                                                                \
  } while (0)
#define LB_ALL()                                                                                                     \
  do {                                                                                                               \
    if (max_len <= 32) LB(2, false);                                                                                 \
    else if (max_len <= 128) LB(8, false);                                                                           \
    else if (max_len <= 160) LB(10, true);         /* 74 KB: two workgroups per CU (93 KB with the V tile: one) */   \
    else LB(16, false);                                                                                              \
  } while (0)
    SIMX_DISPATCH16(dtype, FF, LB_ALL());
#undef LB_ALL
#undef LB
    SIMX_CHECK_LAUNCH("mha_bwd_h16");
    return SIMX_OK;
  }
  if (simx_is16(dtype) && d == 64) {                            // 256 < max_len <= 4096: chunked MFMA kernels
    // 128-token chunks: two workgroups per CU, two tiles per wave (256-token chunks -- one workgroup per CU, four tiles per wave -- measured
    // on the MS-Doc step, one box: attention backward 37.8 ms per step, 128-token chunks 30.5)
    constexpr int ch = 128;
    const int nchunk = cdiv(max_len, ch);
    const size_t ldsl = (size_t)4 * ch * 128 + 2 * ch * sizeof(float) + 4 * 2048 + (size_t)(ch / 16) * ch * 2;   // (+ the dropout bits of dK / dV)
#define LLD(DKV, DROP)                                                                                               \
  do {                                                                                                               \
    rc = set_lds(mha_bwd_long_kernel<FF, DKV, DROP, ch>, ldsl, "mha_bwd");                                           \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_bwd_long_kernel<FF, DKV, DROP, ch>), dim3(nseq * heads * nchunk), dim3(256), ldsl, s, (const bf16_t*)qkv, \
                       (const bf16_t*)ctx, lse, (const bf16_t*)dctx, (bf16_t*)dqkv, cu, heads, T, nchunk, scale, drop); \
  } while (0)
#define LL(DKV) do { if (drop.thr) LLD(DKV, true); else LLD(DKV, false); } while (0)
    SIMX_DISPATCH16(dtype, FF, LL(false); LL(true));
#undef LL
#undef LLD
    SIMX_CHECK_LAUNCH("mha_bwd_long");
    return SIMX_OK;
  }
  if (dtype == SIMX_F32 && simx_mha_f32_ok(d, max_len))
    return simx_mha_bwd_f32(s, nseq, heads, cu, max_len, T, (const float*)qkv, (const float*)ctx, lse, (const float*)dctx, (float*)dqkv, scale,
                            drop);
  const size_t lds = (size_t)(8 * 128 + 8 * max_len) * sizeof(float);
  dim3 grid(nseq * heads, cdiv(max_len, 4));
#define LS(TT, MODE)                                                                                                 \
  do {                                                                                                               \
    rc = set_lds(mha_bwd_simple_kernel<TT, MODE>, lds, "mha_bwd");                                                   \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_bwd_simple_kernel<TT, MODE>), grid, dim3(256), lds, s, (const TT*)qkv, (const TT*)ctx, lse, \
                       (const TT*)dctx, (TT*)dqkv, cu, heads, d, T, scale, max_len, drop);                           \
  } while (0)
  SIMX_DISPATCH3(dtype, TT, LS(TT, 0); LS(TT, 1));
#undef LS
  SIMX_CHECK_LAUNCH("mha_bwd_simple");
  return SIMX_OK;
}

// ------------------------------------------------------------------------------------------
// Single-query attention for the [CLS]-only last layer (simx_bert_cfg.cls_only_last_layer): the only query of a
// sequence that anything downstream reads is its token 0, so layer L-1 needs softmax(q0 K^T / sqrt(d)) V for that one
// row.  One wave per (sequence, head); q comes from a compact [nseq, H] tensor, K / V from the usual packed qkv rows.
// Phase 1: lane = key (own K / V row, 2d bytes contiguous); phase 2: lane = head dimension (coalesced row accesses).
// HBM-bound: reads the K, V columns of qkv once.  Dropout masks use the full kernels' key (row = h*T + t0, col = key).
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void mha_cls_fwd_kernel(int nitems, int heads, int d, int H, int Ttot, int max_len, float scale,
                                                          const int* __restrict__ cu, const T* __restrict__ qc,
                                                          const T* __restrict__ qkv, T* __restrict__ ctxc, DropCtx drop, int hm_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + w;
  if (item >= nitems) return;                      // (no block-level synchronisation below: waves are independent)
  float* sq = reinterpret_cast<float*>(smem) + (size_t)w * (d + max_len);
  float* sp = sq + d;
  const int s = item / heads, h = item % heads;
  const int t0 = cu[s], len = cu[s + 1] - t0;
  for (int c = lane; c < d; c += 64) sq[c] = Elem<T>::ld(qc + (long)s * H + h * d + c);
  __builtin_amdgcn_wave_barrier();
  // (head-major: d == 64, plane (which, head) = [hm_rows][64]; see QkvLay)
  const long ld = hm_rows > 0 ? 64 : 3 * (long)H;
  const T* Kb = hm_rows > 0 ? qkv + ((long)(heads + h) * hm_rows + t0) * 64 : qkv + (long)t0 * 3 * H + H + h * d;
  const T* Vb = hm_rows > 0 ? qkv + ((long)(2 * heads + h) * hm_rows + t0) * 64 : qkv + (long)t0 * 3 * H + 2 * H + h * d;
  float mx = -3.0e38f;
  for (int j = lane; j < len; j += 64) {
    const T* kr = Kb + (long)j * ld;
    float acc = 0.f;
    for (int c = 0; c < d; c += 4) {
      float k4[4];
      ld4(kr + c, k4);
      acc = fmaf(sq[c], k4[0], acc); acc = fmaf(sq[c + 1], k4[1], acc); acc = fmaf(sq[c + 2], k4[2], acc); acc = fmaf(sq[c + 3], k4[3], acc);
    }
    acc *= scale;
    sp[j] = acc;
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < len; j += 64) { const float e = __expf(sp[j] - mx); sp[j] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < len; j += 64) {
    float p = sp[j] * inv;
    if (drop.thr) p *= drop_mult(drop, (uint32_t)(h * Ttot + t0), (uint32_t)j);
    sp[j] = p;
  }
  __builtin_amdgcn_wave_barrier();
  for (int c = lane; c < d; c += 64) {
    float o = 0.f;
    for (int j = 0; j < len; ++j) o = fmaf(sp[j], Elem<T>::ld(Vb + (long)j * ld + c), o);
    Elem<T>::st(ctxc + (long)s * H + h * d + c, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void mha_cls_bwd_kernel(int nitems, int heads, int d, int H, int Ttot, int max_len, float scale,
                                                          const int* __restrict__ cu, const T* __restrict__ qc,
                                                          const T* __restrict__ qkv, const T* __restrict__ dctxc,
                                                          T* __restrict__ dqc, T* __restrict__ dqkv, DropCtx drop, int hm_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int item = blockIdx.x * 4 + w;
  if (item >= nitems) return;
  float* sq = reinterpret_cast<float*>(smem) + (size_t)w * (2 * d + 3 * max_len);
  float* sdo = sq + d;
  float* sp = sdo + d;                             // p (softmax)
  float* spm = sp + max_len;                       // P~ = dropout(p): what multiplies V
  float* sds = spm + max_len;                      // d(loss)/d(score) * scale
  const int s = item / heads, h = item % heads;
  const int t0 = cu[s], len = cu[s + 1] - t0;
  for (int c = lane; c < d; c += 64) {
    sq[c] = Elem<T>::ld(qc + (long)s * H + h * d + c);
    sdo[c] = Elem<T>::ld(dctxc + (long)s * H + h * d + c);
  }
  __builtin_amdgcn_wave_barrier();
  // (head-major: d == 64, plane (which, head) = [hm_rows][64]; see QkvLay)
  const long ld = hm_rows > 0 ? 64 : 3 * (long)H;
  const T* Kb = hm_rows > 0 ? qkv + ((long)(heads + h) * hm_rows + t0) * 64 : qkv + (long)t0 * 3 * H + H + h * d;
  const T* Vb = hm_rows > 0 ? qkv + ((long)(2 * heads + h) * hm_rows + t0) * 64 : qkv + (long)t0 * 3 * H + 2 * H + h * d;
  float mx = -3.0e38f;
  for (int j = lane; j < len; j += 64) {
    const T* kr = Kb + (long)j * ld;
    const T* vr = Vb + (long)j * ld;
    float acc = 0.f, dp = 0.f;
    for (int c = 0; c < d; c += 4) {
      float k4[4], v4[4];
      ld4(kr + c, k4);
      ld4(vr + c, v4);
      acc = fmaf(sq[c], k4[0], acc); acc = fmaf(sq[c + 1], k4[1], acc); acc = fmaf(sq[c + 2], k4[2], acc); acc = fmaf(sq[c + 3], k4[3], acc);
      dp = fmaf(sdo[c], v4[0], dp); dp = fmaf(sdo[c + 1], v4[1], dp); dp = fmaf(sdo[c + 2], v4[2], dp); dp = fmaf(sdo[c + 3], v4[3], dp);
    }
    acc *= scale;
    sp[j] = acc;
    sds[j] = dp;                                   // d(loss)/dP~_j
    mx = fmaxf(mx, acc);
  }
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < len; j += 64) { const float e = __expf(sp[j] - mx); sp[j] = e; sum += e; }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  float dot = 0.f;                                 // sum_i p_i * d(loss)/dp_i
  for (int j = lane; j < len; j += 64) {
    const float p = sp[j] * inv;
    const float m = drop.thr ? drop_mult(drop, (uint32_t)(h * Ttot + t0), (uint32_t)j) : 1.f;
    const float dpj = sds[j] * m;
    dot += p * dpj;
    sp[j] = p;
    spm[j] = p * m;
    sds[j] = dpj;
  }
  dot = wave_sum(dot);
  for (int j = lane; j < len; j += 64) sds[j] = sp[j] * (sds[j] - dot) * scale;
  __builtin_amdgcn_wave_barrier();
  T* dKb = hm_rows > 0 ? dqkv + ((long)(heads + h) * hm_rows + t0) * 64 : dqkv + (long)t0 * 3 * H + H + h * d;
  T* dVb = hm_rows > 0 ? dqkv + ((long)(2 * heads + h) * hm_rows + t0) * 64 : dqkv + (long)t0 * 3 * H + 2 * H + h * d;
  for (int c = lane; c < d; c += 64) {
    const float qv = sq[c], dov = sdo[c];
    float dq = 0.f;
    for (int j = 0; j < len; ++j) {
      const float ds = sds[j];
      dq = fmaf(ds, Elem<T>::ld(Kb + (long)j * ld + c), dq);
      Elem<T>::st(dKb + (long)j * ld + c, ds * qv);
      Elem<T>::st(dVb + (long)j * ld + c, spm[j] * dov);
    }
    Elem<T>::st(dqc + (long)s * H + h * d + c, dq);
  }
}

extern "C" int simx_mha_cls_fwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                                   int T, const void* q_cls, const void* qkv, void* ctx_cls, const simx_dropout* dropd, int hm_rows);
extern "C" int simx_mha_cls_fwd(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                                int T, const void* q_cls, const void* qkv, void* ctx_cls, const simx_dropout* dropd) {
  return simx_mha_cls_fwd_hm(stream, dtype, nseq, heads, d, cu, max_len, T, q_cls, qkv, ctx_cls, dropd, 0);
}
extern "C" int simx_mha_cls_fwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                                   int T, const void* q_cls, const void* qkv, void* ctx_cls, const simx_dropout* dropd, int hm_rows) {
  SIMX_REQUIRE(hm_rows == 0 || (d == 64 && hm_rows >= T), SIMX_ERR_UNSUPPORTED, "mha_cls_fwd: head-major layout needs head size 64, hm_rows >= T");
  SIMX_REQUIRE(nseq > 0 && heads > 0 && d > 0 && d % 4 == 0 && max_len > 0 && T >= nseq, SIMX_ERR_BAD_SHAPE, "mha_cls_fwd: bad shape");
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "mha_cls_fwd: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_MHA_FWD, s, 4.0 * nseq * heads * max_len * d);
  const DropCtx drop = make_drop(dropd);
  const float scale = 1.0f / sqrtf((float)d);
  const int nitems = nseq * heads, H = heads * d;
  const size_t lds = (size_t)4 * (d + max_len) * sizeof(float);
  int rc;
#define LC(TT)                                                                                                       \
  do {                                                                                                               \
    rc = set_lds(mha_cls_fwd_kernel<TT>, lds, "mha_cls_fwd");                                                        \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_cls_fwd_kernel<TT>), dim3(cdiv(nitems, 4)), dim3(256), lds, s, nitems, heads, d, H, T, max_len, scale, cu, \
                       (const TT*)q_cls, (const TT*)qkv, (TT*)ctx_cls, drop, hm_rows);                               \
  } while (0)
  SIMX_DISPATCH3(dtype, TT, LC(TT));
#undef LC
  SIMX_CHECK_LAUNCH("mha_cls_fwd");
  return SIMX_OK;
}

extern "C" int simx_mha_cls_bwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                                   int T, const void* q_cls, const void* qkv, const void* dctx_cls, void* dq_cls, void* dqkv,
                                   const simx_dropout* dropd, int hm_rows);
extern "C" int simx_mha_cls_bwd(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                                int T, const void* q_cls, const void* qkv, const void* dctx_cls, void* dq_cls, void* dqkv,
                                const simx_dropout* dropd) {
  return simx_mha_cls_bwd_hm(stream, dtype, nseq, heads, d, cu, max_len, T, q_cls, qkv, dctx_cls, dq_cls, dqkv, dropd, 0);
}
extern "C" int simx_mha_cls_bwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int d, const int32_t* cu, int max_len,
                                   int T, const void* q_cls, const void* qkv, const void* dctx_cls, void* dq_cls, void* dqkv,
                                   const simx_dropout* dropd, int hm_rows) {
  SIMX_REQUIRE(hm_rows == 0 || (d == 64 && hm_rows >= T), SIMX_ERR_UNSUPPORTED, "mha_cls_bwd: head-major layout needs head size 64, hm_rows >= T");
  SIMX_REQUIRE(nseq > 0 && heads > 0 && d > 0 && d % 4 == 0 && max_len > 0 && T >= nseq, SIMX_ERR_BAD_SHAPE, "mha_cls_bwd: bad shape");
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "mha_cls_bwd: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_MHA_BWD, s, 8.0 * nseq * heads * max_len * d);
  const DropCtx drop = make_drop(dropd);
  const float scale = 1.0f / sqrtf((float)d);
  const int nitems = nseq * heads, H = heads * d;
  const size_t lds = (size_t)4 * (2 * d + 3 * max_len) * sizeof(float);
  int rc;
#define LC(TT)                                                                                                       \
  do {                                                                                                               \
    rc = set_lds(mha_cls_bwd_kernel<TT>, lds, "mha_cls_bwd");                                                        \
    if (rc) return rc;                                                                                               \
    hipLaunchKernelGGL((mha_cls_bwd_kernel<TT>), dim3(cdiv(nitems, 4)), dim3(256), lds, s, nitems, heads, d, H, T, max_len, scale, cu, \
                       (const TT*)q_cls, (const TT*)qkv, (const TT*)dctx_cls, (TT*)dq_cls, (TT*)dqkv, drop, hm_rows); \
  } while (0)
  SIMX_DISPATCH3(dtype, TT, LC(TT));
#undef LC
  SIMX_CHECK_LAUNCH("mha_cls_bwd");
  return SIMX_OK;
}
