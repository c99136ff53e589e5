// Self-attention of the fp32 engine on the 16-bit matrix cores from fp16 plane pairs (BertSelfAttention core,
// LEAD/modeling_bert.py:318-374, in the fp32 arithmetic every train_*_AR2.sh selects).  Head size 64, sequences <= 4096 (K / V of the
// (sequence, head) resident in LDS up to 160 tokens, 128-token chunks above).
//
// attention_f32.hip runs these products on v_mfma_f32_32x32x2_f32 (157 TFLOP/s peak) and spends 100 of the fp32 step's 590 ms
// there.  Here every product is taken as hi.lo + lo.hi + hi.hi of fp16 pairs (three v_mfma_f32_16x16x32_f16 per tile pair:
// 833 TFLOP/s of f32-grade throughput at peak), exactly as the dense GEMMs of gemm_xp.hip:
//   * q / k / v arrive as the fp16 plane pair the QKV projection's epilogue writes (hi = rnd16(x), lo = rnd16(x - hi):
//     22 significand bits), staged by LDS-DMA into hi / lo tiles with the swizzle of attention.hip;
//   * probabilities are split in registers after the f32 softmax, scaled by 2^10 first (p in (0, 1024]: the split's absolute
//     floor of 2^-25 then sits 2^-35 below the largest probability; the factor cancels against the normaliser);
//   * backward: dO is scaled per (sequence, head) block by a power of two that brings its largest element to [1, 2) before it
//     is split into an fp16 pair -- every gradient product (dP = dO.V^T, dV = P^T.dO, dS, dQ = dS.K, dK = dS^T.Q) is then a
//     product of fp16 pairs, with an absolute error floor 2^-25 BELOW THE BLOCK'S LARGEST dO ELEMENT (f32 itself rounds sums at
//     2^-24 of their largest term); dq / dk / dv are multiplied back and leave as the bf16 plane pair the dgrad / wgrad GEMMs
//     stage.  No loss scale, no overflow: |dS| <= 64 max|v| after the scaling.
// Kernel structure = attention.hip's (scores transposed so that a softmax row is a lane column, P / dS feed the second product
// from registers, token-contracted operands by ds_read_b64_tr_b16), with hi / lo tiles and three MFMAs per product.
#include <type_traits>
#include "common.h"
#include "prof.h"

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
#define X3_LOG2E 1.4426950408889634f

__device__ __forceinline__ int x3a_f(int row) {
  const int i = (row >> 1) & 7;
  return (((i << 1) & 7) + (i >> 2) * 5) & 7;
}
__device__ __forceinline__ int x3a_off(int row, int c16) { return row * 128 + ((c16 ^ x3a_f(row)) << 4); }
// stage rows [0, rows_pad) x 64 16-bit values of one head slice by LDS-DMA; rows >= len are clamped copies of row len-1
__device__ __forceinline__ void x3a_stage(const bf16_t* __restrict__ G, long ld, int len, int rows_pad, char* lds, int wave, int lane, int nwaves) {
  for (int i = wave; i < (rows_pad >> 3); i += nwaves) {
    const int r = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ x3a_f(r);
    const int gr = r < len ? r : len - 1;
    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(G + (long)gr * ld + c * 8), (lds_ptr_t)(lds + i * 1024), 16, 0, 0);
  }
}
__device__ __forceinline__ bf16x8 x3a_row_frag(const char* tile, int row, int c16) {
  return *reinterpret_cast<const bf16x8*>(tile + x3a_off(row, c16));
}
__device__ __forceinline__ f32x4 x3a_mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) { return H16<f16_t>::mfma(a, b, c); }
// acc += (ah + al) . (bh + bl) without the lo.lo term, corrections first
__device__ __forceinline__ f32x4 x3a_mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x4 c) {
  c = x3a_mfma(al, bh, c);
  c = x3a_mfma(ah, bl, c);
  return x3a_mfma(ah, bh, c);
}
// fp16 pair of 8 f32 values (two accumulator quads) in fragment order
__device__ __forceinline__ void x3a_split8(const f32x4& a, const f32x4& b, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float ar = f32_pin(a[r]), br = f32_pin(b[r]);
    const f16_t ha = (f16_t)ar, hb = (f16_t)br;
    hi[r] = __builtin_bit_cast(short, ha);
    hi[4 + r] = __builtin_bit_cast(short, hb);
    lo[r] = __builtin_bit_cast(short, (f16_t)(ar - (float)ha));
    lo[4 + r] = __builtin_bit_cast(short, (f16_t)(br - (float)hb));
  }
}
#define X3A_RDTR(DST, ADDR, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #OFF : "=&v"(DST) : "v"(ADDR) : "memory")
#define X3A_CAT(LO, HI) ((bf16x8){LO[0], LO[1], LO[2], LO[3], HI[0], HI[1], HI[2], HI[3]})

// ------------------------------------------------------------------------------------------ forward
// LDS: Khi | Klo | Vhi | Vlo, each NKT * 16 rows x 128 B.  One workgroup (4 waves) per (sequence, head).
// (DROP and ALLT -- every one of the NKT key tiles present -- are compile-time: the query-tile loop then has no uniform branch
// inside and is scheduled as one block, see mha_fwd_h16_kernel)
template <int NKT, bool DROP>
__global__ __launch_bounds__(256, 2) void mha_fwd_x3_kernel(const bf16_t* __restrict__ qkv, long qps, bf16_t* __restrict__ ctx, long cps,
                                                         float* __restrict__ lse, const int* __restrict__ cu, int heads, int T, float scale,
                                                         DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const int nkt = (len + 15) >> 4;
  const int nkt2 = (nkt + 1) & ~1;
  constexpr int TILE = NKT * 16 * 128;
  char* sKh = smem;
  char* sKl = smem + TILE;
  char* sVh = smem + 2 * TILE;
  x3a_stage(Kg, H3, len, nkt2 * 16, sKh, wave, lane, 4);
  x3a_stage(Kg + qps, H3, len, nkt2 * 16, sKl, wave, lane, 4);
  x3a_stage(Vg, H3, len, nkt2 * 16, sVh, wave, lane, 4);
  x3a_stage(Vg + qps, H3, len, nkt2 * 16, sVh + TILE, wave, lane, 4);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const uint32_t sV_addr = (uint32_t)(uintptr_t)sVh;
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * X3_LOG2E;
  uint32_t vtr[4];
  {
    const int rr = 4 * fg + (fr >> 2), tsw = x3a_f(rr), tx = (fr & 3) >> 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vtr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  }
  auto tiles = [&](auto allt_c) {
  constexpr bool ALLT = decltype(allt_c)::value;
  for (int qt = wave; qt < nkt; qt += 4) {
    const int q = qt * 16 + fr;
    const int qc = q < len ? q : len - 1;
    bf16x8 qh[2], ql[2];
    qh[0] = *reinterpret_cast<const bf16x8*>(Qg + (long)qc * H3 + fg * 8);
    qh[1] = *reinterpret_cast<const bf16x8*>(Qg + (long)qc * H3 + 32 + fg * 8);
    ql[0] = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qc * H3 + fg * 8);
    ql[1] = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qc * H3 + 32 + fg * 8);
    f32x4 s[NKT];
    float m = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (ALLT || kt < nkt) {
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        a = x3a_mfma3(x3a_row_frag(sKh, kt * 16 + fr, fg), x3a_row_frag(sKl, kt * 16 + fr, fg), qh[0], ql[0], a);
        a = x3a_mfma3(x3a_row_frag(sKh, kt * 16 + fr, 4 + fg), x3a_row_frag(sKl, kt * 16 + fr, 4 + fg), qh[1], ql[1], a);
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
        const float p = __builtin_amdgcn_exp2f((s[kt][r] - m) * c2 + 10.0f);     // 2^10 p: see the file header
        s[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    if (DROP) {
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
#pragma unroll
    for (int kb = 0; kb < NKT / 2; ++kb) {
      if (ALLT || 2 * kb < nkt) {                    // (rows past the padded length are uninitialised LDS: never multiplied)
        const uint32_t b0 = sV_addr + (uint32_t)(kb * 4096);
        bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
        X3A_RDTR(h0l, b0 + vtr[0], 0); X3A_RDTR(h0h, b0 + vtr[0], 2048); X3A_RDTR(h1l, b0 + vtr[1], 0); X3A_RDTR(h1h, b0 + vtr[1], 2048);
        X3A_RDTR(h2l, b0 + vtr[2], 0); X3A_RDTR(h2h, b0 + vtr[2], 2048); X3A_RDTR(h3l, b0 + vtr[3], 0); X3A_RDTR(h3h, b0 + vtr[3], 2048);
        const uint32_t b1 = b0 + (uint32_t)TILE;
        X3A_RDTR(l0l, b1 + vtr[0], 0); X3A_RDTR(l0h, b1 + vtr[0], 2048); X3A_RDTR(l1l, b1 + vtr[1], 0); X3A_RDTR(l1h, b1 + vtr[1], 2048);
        X3A_RDTR(l2l, b1 + vtr[2], 0); X3A_RDTR(l2h, b1 + vtr[2], 2048); X3A_RDTR(l3l, b1 + vtr[3], 0); X3A_RDTR(l3h, b1 + vtr[3], 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                     "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");   // (one wait names every register in flight)
        bf16x8 ph, pl;
        x3a_split8(s[2 * kb], s[2 * kb + 1], ph, pl);
        o[0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), ph, pl, o[0]);
        o[1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), ph, pl, o[1]);
        o[2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), ph, pl, o[2]);
        o[3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), ph, pl, o[3]);
      }
    }
    if (q < len) {
      bf16_t* dst = ctx + (long)(t0 + q) * H + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float v0 = f32_pin(o[dt][0] * inv), v1 = f32_pin(o[dt][1] * inv), v2 = f32_pin(o[dt][2] * inv), v3 = f32_pin(o[dt][3] * inv);
        const uint32_t h0 = pack2h(v0, v1), h1 = pack2h(v2, v3);
        const uint32_t l0 = pack2h(v0 - H16<f16_t>::lo(h0), v1 - H16<f16_t>::hi(h0)), l1 = pack2h(v2 - H16<f16_t>::lo(h1), v3 - H16<f16_t>::hi(h1));
        *reinterpret_cast<uint2*>(dst + dt * 16) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + cps + dt * 16) = make_uint2(l0, l1);
      }
      if (fg == 0) lse[(long)h * T + t0 + q] = m * scale + (logf(sum) - 10.0f * 0.69314718055994531f);
    }
  }
  };
  if (nkt == NKT) tiles(std::true_type{}); else tiles(std::false_type{});
}

// ------------------------------------------------------------------------------------------ backward
#define X3A_RD128(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=&v"(DST) : "v"(ADDR) : "memory")
// power-of-two scale that brings g (> 0, finite) into [1, 2): returns {2^-floor(log2 g), 2^floor(log2 g)}; g == 0 / inf / nan -> {1, 1}
__device__ __forceinline__ void x3a_pow2_scale(float g, float& sc, float& inv) {
  const uint32_t e = (__float_as_uint(g) >> 23) & 255u;
  const bool ok = e >= 1u && e <= 253u;
  sc = ok ? __uint_as_float((254u - e) << 23) : 1.0f;
  inv = ok ? __uint_as_float(e << 23) : 1.0f;
}
// block maximum of |dO| over the (sequence, head) slice [len, 64] (f32, row pitch H); every thread returns it.  red: 4+ floats of LDS.
__device__ __forceinline__ float x3a_block_absmax(const float* __restrict__ dOg, int H, int len, int tid, int nthreads, float* red) {
  float g = 0.f;
  for (int idx = tid; idx < len * 16; idx += nthreads) {
    const float4 v = *reinterpret_cast<const float4*>(dOg + (long)(idx >> 4) * H + (idx & 15) * 4);
    g = fmaxf(g, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
  g = wave_max(g);
  if ((tid & 63) == 0) red[tid >> 6] = g;
  __syncthreads();
  float r = red[0];
  for (int w = 1; w < (nthreads >> 6); ++w) r = fmaxf(r, red[w]);
  __syncthreads();
  return r;
}
// 8 consecutive values hi + lo of an fp16 plane pair
__device__ __forceinline__ void x3a_ld8_planes(const bf16_t* __restrict__ p, long ps, float (&v)[8]) {
  const uint4 h = *reinterpret_cast<const uint4*>(p), l = *reinterpret_cast<const uint4*>(p + ps);
  const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    v[2 * e] = H16<f16_t>::lo(hw[e]) + H16<f16_t>::lo(lw[e]);
    v[2 * e + 1] = H16<f16_t>::hi(hw[e]) + H16<f16_t>::hi(lw[e]);
  }
}
// fp16 pair fragment of 8 f32 values
__device__ __forceinline__ void x3a_split_frag(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float ve = f32_pin(v[e]);
    const f16_t h = (f16_t)ve;
    hi[e] = __builtin_bit_cast(short, h);
    lo[e] = __builtin_bit_cast(short, (f16_t)(ve - (float)h));
  }
}
// a wave's 16 x 64 accumulator tile (lane = row fr, 4 columns dt * 16 + 4 fg) x mul as a bf16 plane pair, rows < nrows
__device__ __forceinline__ void x3a_store_planes_bf16(const f32x4 (&acc)[4], float mul, bf16_t* __restrict__ dst, long ld, long ps, int nrows, int lane) {
  const int fr = lane & 15, fg = lane >> 4;
  if (fr >= nrows) return;
  bf16_t* d = dst + (long)fr * ld + 4 * fg;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) {
    const float v0 = f32_pin(acc[dt][0] * mul), v1 = f32_pin(acc[dt][1] * mul), v2 = f32_pin(acc[dt][2] * mul), v3 = f32_pin(acc[dt][3] * mul);
    const uint32_t h0 = pack2bf(v0, v1), h1 = pack2bf(v2, v3);
    const uint32_t l0 = pack2bf(v0 - H16<bf16_t>::lo(h0), v1 - H16<bf16_t>::hi(h0)), l1 = pack2bf(v2 - H16<bf16_t>::lo(h1), v3 - H16<bf16_t>::hi(h1));
    *reinterpret_cast<uint2*>(d + dt * 16) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(d + ps + dt * 16) = make_uint2(l0, l1);
  }
}

// column sums of a wave's accumulated tiles -> colsum[64] (this head's slice of the bias gradient): 16-lane row sums by DPP, the
// four waves' partials through LDS (part: 4 x 64 floats, free after the last barrier of the main loop), 64 atomics per workgroup
__device__ __forceinline__ void x3a_colsum_flush(const f32x4 (&cs)[4], float mul, float* __restrict__ colsum, float* part, int tid) {
  const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fg = lane >> 4;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = cs[dt][e];
      t += dpp_mov<0xB1, 0xF>(t);
      t += dpp_mov<0x4E, 0xF>(t);
      t += dpp_mov<0x141, 0xF>(t);
      t += dpp_mov<0x140, 0xF>(t);
      if (fr == 0) part[wave * 64 + dt * 16 + 4 * fg + e] = t;
    }
  __syncthreads();
  if (tid < 64) atomicAdd(colsum + tid, (part[tid] + part[64 + tid] + part[128 + tid] + part[192 + tid]) * mul);
  __syncthreads();
}

// dQ: K and V pairs resident (LDS: Khi | Klo | Vhi | Vlo), waves own 16-query tiles (Q pair fragments and the scaled dO pair
// fragments in registers) and walk the key-tile pairs.
template <int NKT, bool DROP>
__global__ __launch_bounds__(256, 2) void mha_bwd_dq_x3_kernel(const bf16_t* __restrict__ qkv, long qps, const bf16_t* __restrict__ O, long ops,
                                                               const float* __restrict__ lse, const float* __restrict__ dO,
                                                               bf16_t* __restrict__ dqkv, long dps, const int* __restrict__ cu, int heads, int T,
                                                               float scale, DropCtx drop, float* __restrict__ dbias) {
  // dbias (may be NULL): [3H] += column sums of dq | dk | dv over the tokens -- the QKV projection's bias gradient
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const bf16_t* Og = O + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int nkt = (len + 15) >> 4;
  const int nkt2 = (nkt + 1) & ~1;
  constexpr int TILE = NKT * 16 * 128;
  f32x4 csq[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) csq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  x3a_stage(Kg, H3, len, nkt2 * 16, smem, wave, lane, 4);
  x3a_stage(Kg + qps, H3, len, nkt2 * 16, smem + TILE, wave, lane, 4);
  x3a_stage(Vg, H3, len, nkt2 * 16, smem + 2 * TILE, wave, lane, 4);
  x3a_stage(Vg + qps, H3, len, nkt2 * 16, smem + 3 * TILE, wave, lane, 4);
  float* red = reinterpret_cast<float*>(smem + 4 * TILE);
  float sc, isc;
  x3a_pow2_scale(x3a_block_absmax(dOg, H, len, tid, 256, red), sc, isc);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * X3_LOG2E;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int fsw = x3a_f(fr);
  const uint32_t rf_lo = (uint32_t)(fr * 128 + ((fg ^ fsw) << 4)), rf_hi = (uint32_t)(fr * 128 + (((4 + fg) ^ fsw) << 4));
  const int rr = 4 * fg + (fr >> 2), tsw = x3a_f(rr), tx = (fr & 3) >> 1;
  uint32_t tr[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) tr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  for (int qt = wave; qt < nkt; qt += 4) {
    const int q = qt * 16 + fr;
    const int qc = q < len ? q : len - 1;
    const bool qok = q < len;
    bf16x8 qh0, qh1, ql0, ql1, dh0, dh1, dl0, dl1;
    qh0 = *reinterpret_cast<const bf16x8*>(Qg + (long)qc * H3 + fg * 8);
    qh1 = *reinterpret_cast<const bf16x8*>(Qg + (long)qc * H3 + 32 + fg * 8);
    ql0 = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qc * H3 + fg * 8);
    ql1 = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qc * H3 + 32 + fg * 8);
    float delta = 0.f;
    {
      float d0[8], d1[8], o0[8], o1[8];
      const float4* pd = reinterpret_cast<const float4*>(dOg + (long)qc * H + fg * 8);
      const float4 a = pd[0], b = pd[1], c = pd[8], e = pd[9];
      d0[0] = a.x * sc; d0[1] = a.y * sc; d0[2] = a.z * sc; d0[3] = a.w * sc; d0[4] = b.x * sc; d0[5] = b.y * sc; d0[6] = b.z * sc; d0[7] = b.w * sc;
      d1[0] = c.x * sc; d1[1] = c.y * sc; d1[2] = c.z * sc; d1[3] = c.w * sc; d1[4] = e.x * sc; d1[5] = e.y * sc; d1[6] = e.z * sc; d1[7] = e.w * sc;
      x3a_ld8_planes(Og + (long)qc * H + fg * 8, ops, o0);
      x3a_ld8_planes(Og + (long)qc * H + 32 + fg * 8, ops, o1);
#pragma unroll
      for (int i = 0; i < 8; ++i) delta += d0[i] * o0[i] + d1[i] * o1[i];
      x3a_split_frag(d0, dh0, dl0);
      x3a_split_frag(d1, dh1, dl1);
    }
    delta += __shfl_xor(delta, 16, 64);
    delta += __shfl_xor(delta, 32, 64);
    const float lq = lse[(long)h * T + t0 + qc] * X3_LOG2E;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int kp = 0; kp < (nkt2 >> 1); ++kp) {
      const uint32_t bkh = lds0 + (uint32_t)(kp * 4096), bkl = bkh + TILE, bvh = bkh + 2 * TILE, bvl = bkh + 3 * TILE;
      f32x4 ds[2];
      // both key tiles' row fragments in flight, one wait (the pair iteration is one basic block: DROP is compile-time)
      bf16x8 kh[2][2], kl[2][2], vh[2][2], vl[2][2];
      X3A_RD128(kh[0][0], bkh + rf_lo, 0); X3A_RD128(kh[0][1], bkh + rf_hi, 0); X3A_RD128(kl[0][0], bkl + rf_lo, 0); X3A_RD128(kl[0][1], bkl + rf_hi, 0);
      X3A_RD128(vh[0][0], bvh + rf_lo, 0); X3A_RD128(vh[0][1], bvh + rf_hi, 0); X3A_RD128(vl[0][0], bvl + rf_lo, 0); X3A_RD128(vl[0][1], bvl + rf_hi, 0);
      X3A_RD128(kh[1][0], bkh + rf_lo, 2048); X3A_RD128(kh[1][1], bkh + rf_hi, 2048); X3A_RD128(kl[1][0], bkl + rf_lo, 2048); X3A_RD128(kl[1][1], bkl + rf_hi, 2048);
      X3A_RD128(vh[1][0], bvh + rf_lo, 2048); X3A_RD128(vh[1][1], bvh + rf_hi, 2048); X3A_RD128(vl[1][0], bvl + rf_lo, 2048); X3A_RD128(vl[1][1], bvl + rf_hi, 2048);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh[0][0]), "+v"(kh[0][1]), "+v"(kl[0][0]), "+v"(kl[0][1]), "+v"(vh[0][0]), "+v"(vh[0][1]), "+v"(vl[0][0]), "+v"(vl[0][1]),
                   "+v"(kh[1][0]), "+v"(kh[1][1]), "+v"(kl[1][0]), "+v"(kl[1][1]), "+v"(vh[1][0]), "+v"(vh[1][1]), "+v"(vl[1][0]), "+v"(vl[1][1])::"memory");
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int kt = 2 * kp + hf;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = x3a_mfma3(kh[hf][0], kl[hf][0], qh0, ql0, s);
        s = x3a_mfma3(kh[hf][1], kl[hf][1], qh1, ql1, s);
        dp = x3a_mfma3(vh[hf][0], vl[hf][0], dh0, dl0, dp);
        dp = x3a_mfma3(vh[hf][1], vl[hf][1], dh1, dl1, dp);
        float m4[4] = {1.f, 1.f, 1.f, 1.f};
        if (DROP) drop_mult4(drop, (uint32_t)(h * T + t0 + q), (uint32_t)(kt * 16 + 4 * fg), m4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = __builtin_amdgcn_exp2f(s[r] * c2 - lq);
          p = (kt * 16 + 4 * fg + r < len && qok) ? p : 0.f;
          ds[hf][r] = p * (dp[r] * m4[r] - delta) * scale;
        }
      }
      bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
      X3A_RDTR(h0l, bkh + tr[0], 0); X3A_RDTR(h0h, bkh + tr[0], 2048); X3A_RDTR(h1l, bkh + tr[1], 0); X3A_RDTR(h1h, bkh + tr[1], 2048);
      X3A_RDTR(h2l, bkh + tr[2], 0); X3A_RDTR(h2h, bkh + tr[2], 2048); X3A_RDTR(h3l, bkh + tr[3], 0); X3A_RDTR(h3h, bkh + tr[3], 2048);
      X3A_RDTR(l0l, bkl + tr[0], 0); X3A_RDTR(l0h, bkl + tr[0], 2048); X3A_RDTR(l1l, bkl + tr[1], 0); X3A_RDTR(l1h, bkl + tr[1], 2048);
      X3A_RDTR(l2l, bkl + tr[2], 0); X3A_RDTR(l2h, bkl + tr[2], 2048); X3A_RDTR(l3l, bkl + tr[3], 0); X3A_RDTR(l3h, bkl + tr[3], 2048);
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                   "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");   // (one wait names every register in flight)
      bf16x8 sh, sl;
      x3a_split8(ds[0], ds[1], sh, sl);
      dq[0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), sh, sl, dq[0]);
      dq[1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), sh, sl, dq[1]);
      dq[2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), sh, sl, dq[2]);
      dq[3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), sh, sl, dq[3]);
    }
    x3a_store_planes_bf16(dq, isc, dqkv + (long)(t0 + qt * 16) * H3 + h * 64, H3, dps, len - qt * 16, lane);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) csq[dt] += dq[dt];                    // (rows past the sequence end are exactly 0)
  }
  if (dbias != nullptr) {                                                // (uniform)
    __syncthreads();                                                     // every wave is done with the K / V tiles
    x3a_colsum_flush(csq, isc, dbias + h * 64, reinterpret_cast<float*>(smem), tid);
  }
}

// dK, dV: Q pair and the scaled dO pair resident (LDS: Qhi | Qlo | Dhi | Dlo | lse | delta), waves own 16-key tiles (K and V
// pair fragments in registers) and walk the query-tile pairs.
// (the dropout keep-bits are hashed once per block in the prologue, [query][key tile] in the LDS, as in mha_bwd2_h16_kernel:
// this kernel's lanes hold four different query ROWS of the mask = four hashes per four elements otherwise)
template <int NKT, bool DROP>
__global__ __launch_bounds__(256, 2) void mha_bwd_dkv_x3_kernel(const bf16_t* __restrict__ qkv, long qps, const bf16_t* __restrict__ O, long ops,
                                                                const float* __restrict__ lse, const float* __restrict__ dO,
                                                                bf16_t* __restrict__ dqkv, long dps, const int* __restrict__ cu, int heads, int T,
                                                                float scale, DropCtx drop, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int seq = blockIdx.x / heads, h = blockIdx.x % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  if (len <= 0) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const bf16_t* Og = O + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int nkt = (len + 15) >> 4;
  const int nkt2 = (nkt + 1) & ~1;
  constexpr int TILE = NKT * 16 * 128;
  char* sDh = smem + 2 * TILE;
  char* sDl = smem + 3 * TILE;
  f32x4 csk[4], csv[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) { csk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; csv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  float* sLse = reinterpret_cast<float*>(smem + 4 * TILE);
  float* sDel = sLse + NKT * 16;
  float* red = sDel + NKT * 16;
  constexpr int VEC = 4 * TILE;
  constexpr int MSK = VEC + 2 * NKT * 16 * 4 + 64;   // dropout bits [key tile][query]: 16 bits = the keys of the tile
  x3a_stage(Qg, H3, len, nkt2 * 16, smem, wave, lane, 4);
  x3a_stage(Qg + qps, H3, len, nkt2 * 16, smem + TILE, wave, lane, 4);
  float sc, isc;
  x3a_pow2_scale(x3a_block_absmax(dOg, H, len, tid, 256, red), sc, isc);
  // scaled dO pair into the swizzled tile image, delta_i = dO_i . O_i (8 threads per row), lse in log2 units
  for (int idx = tid; idx < nkt2 * 16 * 8; idx += 256) {
    const int r = idx >> 3, c = idx & 7;
    const int gr = r < len ? r : len - 1;
    float d[8], o[8];
    const float4* pd = reinterpret_cast<const float4*>(dOg + (long)gr * H + c * 8);
    const float4 a = pd[0], b = pd[1];
    d[0] = a.x * sc; d[1] = a.y * sc; d[2] = a.z * sc; d[3] = a.w * sc; d[4] = b.x * sc; d[5] = b.y * sc; d[6] = b.z * sc; d[7] = b.w * sc;
    x3a_ld8_planes(Og + (long)gr * H + c * 8, ops, o);
    float del = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) del += d[i] * o[i];
    del += __shfl_xor(del, 1, 64);
    del += __shfl_xor(del, 2, 64);
    del += __shfl_xor(del, 4, 64);
    bf16x8 fh, fl;
    x3a_split_frag(d, fh, fl);
    *reinterpret_cast<bf16x8*>(sDh + x3a_off(r, c)) = fh;
    *reinterpret_cast<bf16x8*>(sDl + x3a_off(r, c)) = fl;
    if (c == 0) {
      sDel[r] = r < len ? del : 0.f;
      sLse[r] = r < len ? lse[(long)h * T + t0 + r] * X3_LOG2E : 0.f;
    }
  }
  if (DROP) {                                        // keep-bits of (query q, keys kt*16 .. +15): wave -> key tile, lane -> query
    for (int kt = wave; kt < nkt2; kt += 4)
      for (int q = lane; q < nkt2 * 16; q += 64) {
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
  const float c2 = scale * X3_LOG2E;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int fsw = x3a_f(fr);
  const uint32_t rf_lo = (uint32_t)(fr * 128 + ((fg ^ fsw) << 4)), rf_hi = (uint32_t)(fr * 128 + (((4 + fg) ^ fsw) << 4));
  const int rr = 4 * fg + (fr >> 2), tsw = x3a_f(rr), tx = (fr & 3) >> 1;
  uint32_t tr[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) tr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  for (int kt = wave; kt < nkt; kt += 4) {
    const int key = kt * 16 + fr;
    const int kc = key < len ? key : len - 1;
    const bool kok = key < len;
    bf16x8 kh0, kh1, kl0, kl1, vh0, vh1, vl0, vl1;
    kh0 = *reinterpret_cast<const bf16x8*>(Kg + (long)kc * H3 + fg * 8);
    kh1 = *reinterpret_cast<const bf16x8*>(Kg + (long)kc * H3 + 32 + fg * 8);
    kl0 = *reinterpret_cast<const bf16x8*>(Kg + qps + (long)kc * H3 + fg * 8);
    kl1 = *reinterpret_cast<const bf16x8*>(Kg + qps + (long)kc * H3 + 32 + fg * 8);
    vh0 = *reinterpret_cast<const bf16x8*>(Vg + (long)kc * H3 + fg * 8);
    vh1 = *reinterpret_cast<const bf16x8*>(Vg + (long)kc * H3 + 32 + fg * 8);
    vl0 = *reinterpret_cast<const bf16x8*>(Vg + qps + (long)kc * H3 + fg * 8);
    vl1 = *reinterpret_cast<const bf16x8*>(Vg + qps + (long)kc * H3 + 32 + fg * 8);
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    for (int qp = 0; qp < (nkt2 >> 1); ++qp) {
      const uint32_t bqh = lds0 + (uint32_t)(qp * 4096), bql = bqh + TILE, bdh = bqh + 2 * TILE, bdl = bqh + 3 * TILE;
      f32x4 pp[2], ds[2];
      bf16x8 qhf[2][2], qlf[2][2], dhf[2][2], dlf[2][2];
      f32x4 lsv[2], dev[2];
      uint2 mb[2] = {make_uint2(~0u, ~0u), make_uint2(~0u, ~0u)};              // keep-bits of the lane's four query rows (16 bits each), per tile
      X3A_RD128(qhf[0][0], bqh + rf_lo, 0); X3A_RD128(qhf[0][1], bqh + rf_hi, 0); X3A_RD128(qlf[0][0], bql + rf_lo, 0); X3A_RD128(qlf[0][1], bql + rf_hi, 0);
      X3A_RD128(dhf[0][0], bdh + rf_lo, 0); X3A_RD128(dhf[0][1], bdh + rf_hi, 0); X3A_RD128(dlf[0][0], bdl + rf_lo, 0); X3A_RD128(dlf[0][1], bdl + rf_hi, 0);
      X3A_RD128(qhf[1][0], bqh + rf_lo, 2048); X3A_RD128(qhf[1][1], bqh + rf_hi, 2048); X3A_RD128(qlf[1][0], bql + rf_lo, 2048); X3A_RD128(qlf[1][1], bql + rf_hi, 2048);
      X3A_RD128(dhf[1][0], bdh + rf_lo, 2048); X3A_RD128(dhf[1][1], bdh + rf_hi, 2048); X3A_RD128(dlf[1][0], bdl + rf_lo, 2048); X3A_RD128(dlf[1][1], bdl + rf_hi, 2048);
      {
        const uint32_t bl = lds0 + (uint32_t)(VEC + (qp * 32 + 4 * fg) * 4);       // sLse[qp*32 + 4 fg ..]; sDel = + NKT * 64 B
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:64"
                     : "=&v"(lsv[0]), "=&v"(lsv[1]), "=&v"(dev[0]), "=&v"(dev[1]) : "v"(bl), "v"(bl + (uint32_t)(NKT * 64)) : "memory");
      }
      if (DROP) {
        const uint32_t ma = lds0 + (uint32_t)(MSK + (kt * NKT * 16 + qp * 32 + 4 * fg) * 2);
        asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:32" : "=&v"(mb[0]), "=&v"(mb[1]) : "v"(ma) : "memory");
      }
      // (ONE wait names every register an asm read above is still filling: one it does not name may be copied before it)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qhf[0][0]), "+v"(qhf[0][1]), "+v"(qlf[0][0]), "+v"(qlf[0][1]), "+v"(dhf[0][0]), "+v"(dhf[0][1]), "+v"(dlf[0][0]), "+v"(dlf[0][1]),
                   "+v"(qhf[1][0]), "+v"(qhf[1][1]), "+v"(qlf[1][0]), "+v"(qlf[1][1]), "+v"(dhf[1][0]), "+v"(dhf[1][1]), "+v"(dlf[1][0]), "+v"(dlf[1][1]),
                   "+v"(lsv[0]), "+v"(lsv[1]), "+v"(dev[0]), "+v"(dev[1]), "+v"(mb[0]), "+v"(mb[1])::"memory");
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int qt = 2 * qp + hf;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
        s = x3a_mfma3(qhf[hf][0], qlf[hf][0], kh0, kl0, s);
        s = x3a_mfma3(qhf[hf][1], qlf[hf][1], kh1, kl1, s);
        dp = x3a_mfma3(dhf[hf][0], dlf[hf][0], vh0, vl0, dp);
        dp = x3a_mfma3(dhf[hf][1], dlf[hf][1], vh1, vl1, dp);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qrow = qt * 16 + 4 * fg + r;
          float p = __builtin_amdgcn_exp2f(s[r] * c2 - lsv[hf][r]);
          p = (qrow < len && kok) ? p : 0.f;
          const uint32_t mword = r < 2 ? mb[hf].x : mb[hf].y;
          const float mm = DROP ? (((mword >> ((r & 1) * 16 + fr)) & 1u) ? drop.scale : 0.f) : 1.f;
          pp[hf][r] = p * mm * 1024.0f;                  // 2^10 P~: see the file header
          ds[hf][r] = p * (dp[r] * mm - dev[hf][r]) * scale;
        }
      }
      bf16x8 ph, pl, sh, sl;
      x3a_split8(pp[0], pp[1], ph, pl);
      x3a_split8(ds[0], ds[1], sh, sl);
      {
        bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
        X3A_RDTR(h0l, bdh + tr[0], 0); X3A_RDTR(h0h, bdh + tr[0], 2048); X3A_RDTR(h1l, bdh + tr[1], 0); X3A_RDTR(h1h, bdh + tr[1], 2048);
        X3A_RDTR(h2l, bdh + tr[2], 0); X3A_RDTR(h2h, bdh + tr[2], 2048); X3A_RDTR(h3l, bdh + tr[3], 0); X3A_RDTR(h3h, bdh + tr[3], 2048);
        X3A_RDTR(l0l, bdl + tr[0], 0); X3A_RDTR(l0h, bdl + tr[0], 2048); X3A_RDTR(l1l, bdl + tr[1], 0); X3A_RDTR(l1h, bdl + tr[1], 2048);
        X3A_RDTR(l2l, bdl + tr[2], 0); X3A_RDTR(l2h, bdl + tr[2], 2048); X3A_RDTR(l3l, bdl + tr[3], 0); X3A_RDTR(l3h, bdl + tr[3], 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                     "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");   // (one wait names every register in flight)
        dv[0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), ph, pl, dv[0]);
        dv[1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), ph, pl, dv[1]);
        dv[2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), ph, pl, dv[2]);
        dv[3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), ph, pl, dv[3]);
      }
      {
        bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
        X3A_RDTR(h0l, bqh + tr[0], 0); X3A_RDTR(h0h, bqh + tr[0], 2048); X3A_RDTR(h1l, bqh + tr[1], 0); X3A_RDTR(h1h, bqh + tr[1], 2048);
        X3A_RDTR(h2l, bqh + tr[2], 0); X3A_RDTR(h2h, bqh + tr[2], 2048); X3A_RDTR(h3l, bqh + tr[3], 0); X3A_RDTR(h3h, bqh + tr[3], 2048);
        X3A_RDTR(l0l, bql + tr[0], 0); X3A_RDTR(l0h, bql + tr[0], 2048); X3A_RDTR(l1l, bql + tr[1], 0); X3A_RDTR(l1h, bql + tr[1], 2048);
        X3A_RDTR(l2l, bql + tr[2], 0); X3A_RDTR(l2h, bql + tr[2], 2048); X3A_RDTR(l3l, bql + tr[3], 0); X3A_RDTR(l3h, bql + tr[3], 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                     "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");   // (one wait names every register in flight)
        dk[0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), sh, sl, dk[0]);
        dk[1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), sh, sl, dk[1]);
        dk[2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), sh, sl, dk[2]);
        dk[3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), sh, sl, dk[3]);
      }
    }
    bf16_t* dst = dqkv + (long)(t0 + kt * 16) * H3 + H + h * 64;
    x3a_store_planes_bf16(dk, isc, dst, H3, dps, len - kt * 16, lane);
    x3a_store_planes_bf16(dv, isc * (1.0f / 1024.0f), dst + H, H3, dps, len - kt * 16, lane);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { csk[dt] += dk[dt]; csv[dt] += dv[dt]; }
  }
  if (dbias != nullptr) {
    __syncthreads();
    x3a_colsum_flush(csk, isc, dbias + H + h * 64, reinterpret_cast<float*>(smem), tid);
    x3a_colsum_flush(csv, isc * (1.0f / 1024.0f), dbias + 2 * H + h * 64, reinterpret_cast<float*>(smem), tid);
  }
}

// ------------------------------------------------------------------------------------------ long sequences (256 < S <= 4096)
// MS-MARCO Document (BASELINE configs[4]: 512-token documents) in the fp32 arithmetic its recipe selects, on the 16-bit matrix
// cores: the kernels above with the sequence cut into 128-token chunks (the walk of attention_f32.hip's long kernels):
//   forward : one workgroup per (sequence, head, 128-query chunk); a wave owns two 16-query tiles (Q pair fragments, running maximum /
//             normaliser and the output accumulators in registers) and the workgroup walks the key chunks -- the K and V pairs of a
//             chunk staged in LDS, online softmax (accumulators rescaled when the running maximum moves);
//   dQ      : the same grid and walk with the forward's lse; dq accumulates in registers;
//   dK, dV  : one workgroup per (sequence, head, 128-key chunk); a wave owns two 16-key tiles (accumulators in registers, K / V pair
//             fragments reloaded per chunk) and the workgroup walks the query chunks (Q pair by LDS-DMA, the scaled dO pair split on
//             the way into LDS, lse, rowsum(dO . O) and the dropout bits of the chunk).
// The power-of-two scale of dO is the one of the whole (sequence, head) slice, as above.  No atomics (but the bias gradient's), no
// f32 scratch; every wave takes part in every barrier.
#define XL_CH 128
#define XL_TILE (XL_CH * 128)

template <bool DROP>
__global__ __launch_bounds__(256, 2) void mha_fwd_x3_long_kernel(const bf16_t* __restrict__ qkv, long qps, bf16_t* __restrict__ ctx, long cps,
                                                                float* __restrict__ lse, const int* __restrict__ cu, int heads, int T, int nchunk,
                                                                float scale, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x / nchunk, qc = blockIdx.x % nchunk;
  const int seq = blk / heads, h = blk % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  const int q0 = qc * XL_CH;
  if (len <= 0 || q0 >= len) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const int nkc = (len + XL_CH - 1) / XL_CH;
  char* sKh = smem;
  char* sKl = smem + XL_TILE;
  char* sVh = smem + 2 * XL_TILE;
  const uint32_t sV_addr = (uint32_t)(uintptr_t)sVh;
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * X3_LOG2E;
  uint32_t vtr[4];
  {
    const int rr = 4 * fg + (fr >> 2), tsw = x3a_f(rr), tx = (fr & 3) >> 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) vtr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  }
  bf16x8 qh[2][2], ql[2][2];
  f32x4 o[2][4];
  float mrun[2], lrun[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = q0 + (wave + 4 * j) * 16 + fr;
    const int qcl = q < len ? q : len - 1;
    qh[j][0] = *reinterpret_cast<const bf16x8*>(Qg + (long)qcl * H3 + fg * 8);
    qh[j][1] = *reinterpret_cast<const bf16x8*>(Qg + (long)qcl * H3 + 32 + fg * 8);
    ql[j][0] = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qcl * H3 + fg * 8);
    ql[j][1] = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qcl * H3 + 32 + fg * 8);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mrun[j] = -INFINITY;
    lrun[j] = 0.f;
  }
  for (int kc = 0; kc < nkc; ++kc) {
    const int k0 = kc * XL_CH;
    const int nk = len - k0 < XL_CH ? len - k0 : XL_CH;
    const int nkt = (nk + 15) >> 4, nkt2 = (nkt + 1) & ~1;
    __syncthreads();                                 // every wave is done with the previous chunk's tiles
    x3a_stage(Kg + (long)k0 * H3, H3, nk, nkt2 * 16, sKh, wave, lane, 4);
    x3a_stage(Kg + qps + (long)k0 * H3, H3, nk, nkt2 * 16, sKl, wave, lane, 4);
    x3a_stage(Vg + (long)k0 * H3, H3, nk, nkt2 * 16, sVh, wave, lane, 4);
    x3a_stage(Vg + qps + (long)k0 * H3, H3, nk, nkt2 * 16, sVh + XL_TILE, wave, lane, 4);
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
            a = x3a_mfma3(x3a_row_frag(sKh, kt * 16 + fr, fg), x3a_row_frag(sKl, kt * 16 + fr, fg), qh[j][0], ql[j][0], a);
            a = x3a_mfma3(x3a_row_frag(sKh, kt * 16 + fr, 4 + fg), x3a_row_frag(sKl, kt * 16 + fr, 4 + fg), qh[j][1], ql[j][1], a);
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
            const float p = __builtin_amdgcn_exp2f((s[kt][r] - mnew) * c2 + 10.0f);     // 2^10 p: see the file header
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
            bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
            X3A_RDTR(h0l, b0 + vtr[0], 0); X3A_RDTR(h0h, b0 + vtr[0], 2048); X3A_RDTR(h1l, b0 + vtr[1], 0); X3A_RDTR(h1h, b0 + vtr[1], 2048);
            X3A_RDTR(h2l, b0 + vtr[2], 0); X3A_RDTR(h2h, b0 + vtr[2], 2048); X3A_RDTR(h3l, b0 + vtr[3], 0); X3A_RDTR(h3h, b0 + vtr[3], 2048);
            const uint32_t b1 = b0 + (uint32_t)XL_TILE;
            X3A_RDTR(l0l, b1 + vtr[0], 0); X3A_RDTR(l0h, b1 + vtr[0], 2048); X3A_RDTR(l1l, b1 + vtr[1], 0); X3A_RDTR(l1h, b1 + vtr[1], 2048);
            X3A_RDTR(l2l, b1 + vtr[2], 0); X3A_RDTR(l2h, b1 + vtr[2], 2048); X3A_RDTR(l3l, b1 + vtr[3], 0); X3A_RDTR(l3h, b1 + vtr[3], 2048);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                         "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");
            bf16x8 ph, pl;
            x3a_split8(s[2 * kb], s[2 * kb + 1], ph, pl);
            o[j][0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), ph, pl, o[j][0]);
            o[j][1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), ph, pl, o[j][1]);
            o[j][2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), ph, pl, o[j][2]);
            o[j][3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), ph, pl, o[j][3]);
          }
        }
      }
    };
    if (nk == XL_CH) chunk(std::true_type{}); else chunk(std::false_type{});
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = q0 + (wave + 4 * j) * 16 + fr;
    if (q < len) {
      const float inv = 1.0f / lrun[j];
      bf16_t* dst = ctx + (long)(t0 + q) * H + h * 64 + 4 * fg;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const float v0 = f32_pin(o[j][dt][0] * inv), v1 = f32_pin(o[j][dt][1] * inv), v2 = f32_pin(o[j][dt][2] * inv), v3 = f32_pin(o[j][dt][3] * inv);
        const uint32_t h0 = pack2h(v0, v1), h1 = pack2h(v2, v3);
        const uint32_t l0 = pack2h(v0 - H16<f16_t>::lo(h0), v1 - H16<f16_t>::hi(h0)), l1 = pack2h(v2 - H16<f16_t>::lo(h1), v3 - H16<f16_t>::hi(h1));
        *reinterpret_cast<uint2*>(dst + dt * 16) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + cps + dt * 16) = make_uint2(l0, l1);
      }
      if (fg == 0) lse[(long)h * T + t0 + q] = mrun[j] * scale + (logf(lrun[j]) - 10.0f * 0.69314718055994531f);
    }
  }
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void mha_bwd_dq_x3_long_kernel(const bf16_t* __restrict__ qkv, long qps, const bf16_t* __restrict__ O, long ops,
                                                                   const float* __restrict__ lse, const float* __restrict__ dO,
                                                                   bf16_t* __restrict__ dqkv, long dps, const int* __restrict__ cu, int heads, int T,
                                                                   int nchunk, float scale, DropCtx drop, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x / nchunk, qc = blockIdx.x % nchunk;
  const int seq = blk / heads, h = blk % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  const int q0 = qc * XL_CH;
  if (len <= 0 || q0 >= len) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const bf16_t* Og = O + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int nkc = (len + XL_CH - 1) / XL_CH;
  float* red = reinterpret_cast<float*>(smem + 4 * XL_TILE);
  float sc, isc;
  x3a_pow2_scale(x3a_block_absmax(dOg, H, len, tid, 256, red), sc, isc);
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * X3_LOG2E;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int fsw = x3a_f(fr);
  const uint32_t rf_lo = (uint32_t)(fr * 128 + ((fg ^ fsw) << 4)), rf_hi = (uint32_t)(fr * 128 + (((4 + fg) ^ fsw) << 4));
  const int rr = 4 * fg + (fr >> 2), tsw = x3a_f(rr), tx = (fr & 3) >> 1;
  uint32_t tr[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) tr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  bf16x8 qh[2][2], ql[2][2], dh[2][2], dl[2][2];
  float delta[2], lq[2];
  f32x4 dq[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int q = q0 + (wave + 4 * j) * 16 + fr;
    const int qcl = q < len ? q : len - 1;
    qh[j][0] = *reinterpret_cast<const bf16x8*>(Qg + (long)qcl * H3 + fg * 8);
    qh[j][1] = *reinterpret_cast<const bf16x8*>(Qg + (long)qcl * H3 + 32 + fg * 8);
    ql[j][0] = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qcl * H3 + fg * 8);
    ql[j][1] = *reinterpret_cast<const bf16x8*>(Qg + qps + (long)qcl * H3 + 32 + fg * 8);
    float d0[8], d1[8], o0[8], o1[8];
    const float4* pd = reinterpret_cast<const float4*>(dOg + (long)qcl * H + fg * 8);
    const float4 a = pd[0], b = pd[1], c = pd[8], e = pd[9];
    d0[0] = a.x * sc; d0[1] = a.y * sc; d0[2] = a.z * sc; d0[3] = a.w * sc; d0[4] = b.x * sc; d0[5] = b.y * sc; d0[6] = b.z * sc; d0[7] = b.w * sc;
    d1[0] = c.x * sc; d1[1] = c.y * sc; d1[2] = c.z * sc; d1[3] = c.w * sc; d1[4] = e.x * sc; d1[5] = e.y * sc; d1[6] = e.z * sc; d1[7] = e.w * sc;
    x3a_ld8_planes(Og + (long)qcl * H + fg * 8, ops, o0);
    x3a_ld8_planes(Og + (long)qcl * H + 32 + fg * 8, ops, o1);
    float del = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) del += d0[i] * o0[i] + d1[i] * o1[i];
    x3a_split_frag(d0, dh[j][0], dl[j][0]);
    x3a_split_frag(d1, dh[j][1], dl[j][1]);
    del += __shfl_xor(del, 16, 64);
    del += __shfl_xor(del, 32, 64);
    delta[j] = del;
    lq[j] = lse[(long)h * T + t0 + qcl] * X3_LOG2E;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (int kc = 0; kc < nkc; ++kc) {
    const int k0 = kc * XL_CH;
    const int nk = len - k0 < XL_CH ? len - k0 : XL_CH;
    const int nkt = (nk + 15) >> 4, nkt2 = (nkt + 1) & ~1;
    __syncthreads();
    x3a_stage(Kg + (long)k0 * H3, H3, nk, nkt2 * 16, smem, wave, lane, 4);
    x3a_stage(Kg + qps + (long)k0 * H3, H3, nk, nkt2 * 16, smem + XL_TILE, wave, lane, 4);
    x3a_stage(Vg + (long)k0 * H3, H3, nk, nkt2 * 16, smem + 2 * XL_TILE, wave, lane, 4);
    x3a_stage(Vg + qps + (long)k0 * H3, H3, nk, nkt2 * 16, smem + 3 * XL_TILE, wave, lane, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (q0 + (wave + 4 * j) * 16 >= len) continue;                         // (wave-uniform)
      const int q = q0 + (wave + 4 * j) * 16 + fr;
      const bool qok = q < len;
      for (int kp = 0; kp < (nkt2 >> 1); ++kp) {
        const uint32_t bkh = lds0 + (uint32_t)(kp * 4096), bkl = bkh + XL_TILE, bvh = bkh + 2 * XL_TILE, bvl = bkh + 3 * XL_TILE;
        f32x4 ds[2];
        bf16x8 kh[2][2], kl[2][2], vh[2][2], vl[2][2];
        X3A_RD128(kh[0][0], bkh + rf_lo, 0); X3A_RD128(kh[0][1], bkh + rf_hi, 0); X3A_RD128(kl[0][0], bkl + rf_lo, 0); X3A_RD128(kl[0][1], bkl + rf_hi, 0);
        X3A_RD128(vh[0][0], bvh + rf_lo, 0); X3A_RD128(vh[0][1], bvh + rf_hi, 0); X3A_RD128(vl[0][0], bvl + rf_lo, 0); X3A_RD128(vl[0][1], bvl + rf_hi, 0);
        X3A_RD128(kh[1][0], bkh + rf_lo, 2048); X3A_RD128(kh[1][1], bkh + rf_hi, 2048); X3A_RD128(kl[1][0], bkl + rf_lo, 2048); X3A_RD128(kl[1][1], bkl + rf_hi, 2048);
        X3A_RD128(vh[1][0], bvh + rf_lo, 2048); X3A_RD128(vh[1][1], bvh + rf_hi, 2048); X3A_RD128(vl[1][0], bvl + rf_lo, 2048); X3A_RD128(vl[1][1], bvl + rf_hi, 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kh[0][0]), "+v"(kh[0][1]), "+v"(kl[0][0]), "+v"(kl[0][1]), "+v"(vh[0][0]), "+v"(vh[0][1]), "+v"(vl[0][0]), "+v"(vl[0][1]),
                     "+v"(kh[1][0]), "+v"(kh[1][1]), "+v"(kl[1][0]), "+v"(kl[1][1]), "+v"(vh[1][0]), "+v"(vh[1][1]), "+v"(vl[1][0]), "+v"(vl[1][1])::"memory");
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int kt = 2 * kp + hf;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = x3a_mfma3(kh[hf][0], kl[hf][0], qh[j][0], ql[j][0], s);
          s = x3a_mfma3(kh[hf][1], kl[hf][1], qh[j][1], ql[j][1], s);
          dp = x3a_mfma3(vh[hf][0], vl[hf][0], dh[j][0], dl[j][0], dp);
          dp = x3a_mfma3(vh[hf][1], vl[hf][1], dh[j][1], dl[j][1], dp);
          float m4[4] = {1.f, 1.f, 1.f, 1.f};
          if (DROP) drop_mult4(drop, (uint32_t)(h * T + t0 + q), (uint32_t)(k0 + kt * 16 + 4 * fg), m4);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(s[r] * c2 - lq[j]);
            p = (k0 + kt * 16 + 4 * fg + r < len && qok) ? p : 0.f;
            ds[hf][r] = p * (dp[r] * m4[r] - delta[j]) * scale;
          }
        }
        bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
        X3A_RDTR(h0l, bkh + tr[0], 0); X3A_RDTR(h0h, bkh + tr[0], 2048); X3A_RDTR(h1l, bkh + tr[1], 0); X3A_RDTR(h1h, bkh + tr[1], 2048);
        X3A_RDTR(h2l, bkh + tr[2], 0); X3A_RDTR(h2h, bkh + tr[2], 2048); X3A_RDTR(h3l, bkh + tr[3], 0); X3A_RDTR(h3h, bkh + tr[3], 2048);
        X3A_RDTR(l0l, bkl + tr[0], 0); X3A_RDTR(l0h, bkl + tr[0], 2048); X3A_RDTR(l1l, bkl + tr[1], 0); X3A_RDTR(l1h, bkl + tr[1], 2048);
        X3A_RDTR(l2l, bkl + tr[2], 0); X3A_RDTR(l2h, bkl + tr[2], 2048); X3A_RDTR(l3l, bkl + tr[3], 0); X3A_RDTR(l3h, bkl + tr[3], 2048);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                     "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");
        bf16x8 sh, sl;
        x3a_split8(ds[0], ds[1], sh, sl);
        dq[j][0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), sh, sl, dq[j][0]);
        dq[j][1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), sh, sl, dq[j][1]);
        dq[j][2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), sh, sl, dq[j][2]);
        dq[j][3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), sh, sl, dq[j][3]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r0 = q0 + (wave + 4 * j) * 16;
    x3a_store_planes_bf16(dq[j], isc, dqkv + (long)(t0 + r0) * H3 + h * 64, H3, dps, len - r0, lane);
  }
  if (dbias != nullptr) {                                                  // (uniform; rows past the sequence end are exactly 0)
    f32x4 csq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) csq[dt] = dq[0][dt] + dq[1][dt];
    __syncthreads();
    x3a_colsum_flush(csq, isc, dbias + h * 64, reinterpret_cast<float*>(smem), tid);
  }
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void mha_bwd_dkv_x3_long_kernel(const bf16_t* __restrict__ qkv, long qps, const bf16_t* __restrict__ O, long ops,
                                                                    const float* __restrict__ lse, const float* __restrict__ dO,
                                                                    bf16_t* __restrict__ dqkv, long dps, const int* __restrict__ cu, int heads, int T,
                                                                    int nchunk, float scale, DropCtx drop, float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int blk = blockIdx.x / nchunk, kc = blockIdx.x % nchunk;
  const int seq = blk / heads, h = blk % heads;
  const int t0 = cu[seq], len = cu[seq + 1] - t0;
  const int k0 = kc * XL_CH;
  if (len <= 0 || k0 >= len) return;
  const int H = heads * 64;
  const long H3 = 3L * H;
  const bf16_t* Qg = qkv + (long)t0 * H3 + h * 64;
  const bf16_t* Kg = Qg + H;
  const bf16_t* Vg = Kg + H;
  const bf16_t* Og = O + (long)t0 * H + h * 64;
  const float* dOg = dO + (long)t0 * H + h * 64;
  const int nqc = (len + XL_CH - 1) / XL_CH;
  constexpr int VEC = 4 * XL_TILE;                    // lse[128], delta[128], red[16], dropout bits [key tile][query]
  constexpr int MSK = VEC + 2 * XL_CH * 4 + 64;
  char* sDh = smem + 2 * XL_TILE;
  char* sDl = smem + 3 * XL_TILE;
  float* sLse = reinterpret_cast<float*>(smem + VEC);
  float* sDel = sLse + XL_CH;
  float* red = sDel + XL_CH;
  float sc, isc;
  x3a_pow2_scale(x3a_block_absmax(dOg, H, len, tid, 256, red), sc, isc);
  const int fr = lane & 15, fg = lane >> 4;
  const float c2 = scale * X3_LOG2E;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int fsw = x3a_f(fr);
  const uint32_t rf_lo = (uint32_t)(fr * 128 + ((fg ^ fsw) << 4)), rf_hi = (uint32_t)(fr * 128 + (((4 + fg) ^ fsw) << 4));
  const int rr = 4 * fg + (fr >> 2), tsw = x3a_f(rr), tx = (fr & 3) >> 1;
  uint32_t tr[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) tr[dt] = (uint32_t)(rr * 128 + (((dt * 2 + tx) ^ tsw) << 4) + (fr & 1) * 8);
  f32x4 dk[2][4], dv[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[j][dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  for (int qc = 0; qc < nqc; ++qc) {
    const int q0 = qc * XL_CH;
    const int nq = len - q0 < XL_CH ? len - q0 : XL_CH;
    const int nqt = (nq + 15) >> 4, nqt2 = (nqt + 1) & ~1;
    __syncthreads();                                 // every wave is done with the previous chunk's tiles
    x3a_stage(Qg + (long)q0 * H3, H3, nq, nqt2 * 16, smem, wave, lane, 4);
    x3a_stage(Qg + qps + (long)q0 * H3, H3, nq, nqt2 * 16, smem + XL_TILE, wave, lane, 4);
    // scaled dO pair into the swizzled tile image, delta_i = dO_i . O_i (8 threads per row), lse in log2 units
    for (int idx = tid; idx < nqt2 * 16 * 8; idx += 256) {
      const int r = idx >> 3, c = idx & 7;
      const int gr = q0 + (r < nq ? r : nq - 1);
      float d[8], o[8];
      const float4* pd = reinterpret_cast<const float4*>(dOg + (long)gr * H + c * 8);
      const float4 a = pd[0], b = pd[1];
      d[0] = a.x * sc; d[1] = a.y * sc; d[2] = a.z * sc; d[3] = a.w * sc; d[4] = b.x * sc; d[5] = b.y * sc; d[6] = b.z * sc; d[7] = b.w * sc;
      x3a_ld8_planes(Og + (long)gr * H + c * 8, ops, o);
      float del = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) del += d[i] * o[i];
      del += __shfl_xor(del, 1, 64);
      del += __shfl_xor(del, 2, 64);
      del += __shfl_xor(del, 4, 64);
      bf16x8 fh, fl;
      x3a_split_frag(d, fh, fl);
      *reinterpret_cast<bf16x8*>(sDh + x3a_off(r, c)) = fh;
      *reinterpret_cast<bf16x8*>(sDl + x3a_off(r, c)) = fl;
      if (c == 0) {
        sDel[r] = r < nq ? del : 0.f;
        sLse[r] = r < nq ? lse[(long)h * T + t0 + gr] * X3_LOG2E : 0.f;
      }
    }
    if (DROP) {                                      // keep-bits of (query q0 + q, keys k0 + ktl*16 .. +15): wave -> key tile, lane -> query
      for (int ktl = wave; ktl < 8; ktl += 4)
        for (int q = lane; q < nqt2 * 16; q += 64) {
          uint32_t w = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t hsh = drop_mix(drop.seed, drop.stream, (uint32_t)(h * T + t0 + q0 + q), (uint32_t)((k0 >> 2) + ktl * 4 + j));
            w |= drop_keep4(hsh, drop.thr) << (4 * j);
          }
          *reinterpret_cast<unsigned short*>(smem + MSK + (ktl * XL_CH + q) * 2) = (unsigned short)w;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ktl = wave + 4 * j;
      if (k0 + ktl * 16 >= len) continue;                                    // (wave-uniform)
      const int key = k0 + ktl * 16 + fr;
      const int kcl = key < len ? key : len - 1;
      const bool kok = key < len;
      bf16x8 kh0, kh1, kl0, kl1, vh0, vh1, vl0, vl1;
      kh0 = *reinterpret_cast<const bf16x8*>(Kg + (long)kcl * H3 + fg * 8);
      kh1 = *reinterpret_cast<const bf16x8*>(Kg + (long)kcl * H3 + 32 + fg * 8);
      kl0 = *reinterpret_cast<const bf16x8*>(Kg + qps + (long)kcl * H3 + fg * 8);
      kl1 = *reinterpret_cast<const bf16x8*>(Kg + qps + (long)kcl * H3 + 32 + fg * 8);
      vh0 = *reinterpret_cast<const bf16x8*>(Vg + (long)kcl * H3 + fg * 8);
      vh1 = *reinterpret_cast<const bf16x8*>(Vg + (long)kcl * H3 + 32 + fg * 8);
      vl0 = *reinterpret_cast<const bf16x8*>(Vg + qps + (long)kcl * H3 + fg * 8);
      vl1 = *reinterpret_cast<const bf16x8*>(Vg + qps + (long)kcl * H3 + 32 + fg * 8);
      for (int qp = 0; qp < (nqt2 >> 1); ++qp) {
        const uint32_t bqh = lds0 + (uint32_t)(qp * 4096), bql = bqh + XL_TILE, bdh = bqh + 2 * XL_TILE, bdl = bqh + 3 * XL_TILE;
        f32x4 pp[2], ds[2];
        bf16x8 qhf[2][2], qlf[2][2], dhf[2][2], dlf[2][2];
        f32x4 lsv[2], dev[2];
        uint2 mb[2] = {make_uint2(~0u, ~0u), make_uint2(~0u, ~0u)};
        X3A_RD128(qhf[0][0], bqh + rf_lo, 0); X3A_RD128(qhf[0][1], bqh + rf_hi, 0); X3A_RD128(qlf[0][0], bql + rf_lo, 0); X3A_RD128(qlf[0][1], bql + rf_hi, 0);
        X3A_RD128(dhf[0][0], bdh + rf_lo, 0); X3A_RD128(dhf[0][1], bdh + rf_hi, 0); X3A_RD128(dlf[0][0], bdl + rf_lo, 0); X3A_RD128(dlf[0][1], bdl + rf_hi, 0);
        X3A_RD128(qhf[1][0], bqh + rf_lo, 2048); X3A_RD128(qhf[1][1], bqh + rf_hi, 2048); X3A_RD128(qlf[1][0], bql + rf_lo, 2048); X3A_RD128(qlf[1][1], bql + rf_hi, 2048);
        X3A_RD128(dhf[1][0], bdh + rf_lo, 2048); X3A_RD128(dhf[1][1], bdh + rf_hi, 2048); X3A_RD128(dlf[1][0], bdl + rf_lo, 2048); X3A_RD128(dlf[1][1], bdl + rf_hi, 2048);
        {
          const uint32_t bl = lds0 + (uint32_t)(VEC + (qp * 32 + 4 * fg) * 4);       // sLse[qp*32 + 4 fg ..]; sDel = + 512 B
          asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %5\n\tds_read_b128 %3, %5 offset:64"
                       : "=&v"(lsv[0]), "=&v"(lsv[1]), "=&v"(dev[0]), "=&v"(dev[1]) : "v"(bl), "v"(bl + (uint32_t)(XL_CH * 4)) : "memory");
        }
        if (DROP) {
          const uint32_t ma = lds0 + (uint32_t)(MSK + (ktl * XL_CH + qp * 32 + 4 * fg) * 2);
          asm volatile("ds_read_b64 %0, %2\n\tds_read_b64 %1, %2 offset:32" : "=&v"(mb[0]), "=&v"(mb[1]) : "v"(ma) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(qhf[0][0]), "+v"(qhf[0][1]), "+v"(qlf[0][0]), "+v"(qlf[0][1]), "+v"(dhf[0][0]), "+v"(dhf[0][1]), "+v"(dlf[0][0]), "+v"(dlf[0][1]),
                     "+v"(qhf[1][0]), "+v"(qhf[1][1]), "+v"(qlf[1][0]), "+v"(qlf[1][1]), "+v"(dhf[1][0]), "+v"(dhf[1][1]), "+v"(dlf[1][0]), "+v"(dlf[1][1]),
                     "+v"(lsv[0]), "+v"(lsv[1]), "+v"(dev[0]), "+v"(dev[1]), "+v"(mb[0]), "+v"(mb[1])::"memory");
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int qt = 2 * qp + hf;
          f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
          s = x3a_mfma3(qhf[hf][0], qlf[hf][0], kh0, kl0, s);
          s = x3a_mfma3(qhf[hf][1], qlf[hf][1], kh1, kl1, s);
          dp = x3a_mfma3(dhf[hf][0], dlf[hf][0], vh0, vl0, dp);
          dp = x3a_mfma3(dhf[hf][1], dlf[hf][1], vh1, vl1, dp);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int qrow = qt * 16 + 4 * fg + r;
            float p = __builtin_amdgcn_exp2f(s[r] * c2 - lsv[hf][r]);
            p = (qrow < nq && kok) ? p : 0.f;
            const uint32_t mword = r < 2 ? mb[hf].x : mb[hf].y;
            const float mm = DROP ? (((mword >> ((r & 1) * 16 + fr)) & 1u) ? drop.scale : 0.f) : 1.f;
            pp[hf][r] = p * mm * 1024.0f;                  // 2^10 P~: see the file header
            ds[hf][r] = p * (dp[r] * mm - dev[hf][r]) * scale;
          }
        }
        bf16x8 ph, pl, sh, sl;
        x3a_split8(pp[0], pp[1], ph, pl);
        x3a_split8(ds[0], ds[1], sh, sl);
        {
          bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
          X3A_RDTR(h0l, bdh + tr[0], 0); X3A_RDTR(h0h, bdh + tr[0], 2048); X3A_RDTR(h1l, bdh + tr[1], 0); X3A_RDTR(h1h, bdh + tr[1], 2048);
          X3A_RDTR(h2l, bdh + tr[2], 0); X3A_RDTR(h2h, bdh + tr[2], 2048); X3A_RDTR(h3l, bdh + tr[3], 0); X3A_RDTR(h3h, bdh + tr[3], 2048);
          X3A_RDTR(l0l, bdl + tr[0], 0); X3A_RDTR(l0h, bdl + tr[0], 2048); X3A_RDTR(l1l, bdl + tr[1], 0); X3A_RDTR(l1h, bdl + tr[1], 2048);
          X3A_RDTR(l2l, bdl + tr[2], 0); X3A_RDTR(l2h, bdl + tr[2], 2048); X3A_RDTR(l3l, bdl + tr[3], 0); X3A_RDTR(l3h, bdl + tr[3], 2048);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                       "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");
          dv[j][0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), ph, pl, dv[j][0]);
          dv[j][1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), ph, pl, dv[j][1]);
          dv[j][2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), ph, pl, dv[j][2]);
          dv[j][3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), ph, pl, dv[j][3]);
        }
        {
          bf16x4 h0l, h0h, h1l, h1h, h2l, h2h, h3l, h3h, l0l, l0h, l1l, l1h, l2l, l2h, l3l, l3h;
          X3A_RDTR(h0l, bqh + tr[0], 0); X3A_RDTR(h0h, bqh + tr[0], 2048); X3A_RDTR(h1l, bqh + tr[1], 0); X3A_RDTR(h1h, bqh + tr[1], 2048);
          X3A_RDTR(h2l, bqh + tr[2], 0); X3A_RDTR(h2h, bqh + tr[2], 2048); X3A_RDTR(h3l, bqh + tr[3], 0); X3A_RDTR(h3h, bqh + tr[3], 2048);
          X3A_RDTR(l0l, bql + tr[0], 0); X3A_RDTR(l0h, bql + tr[0], 2048); X3A_RDTR(l1l, bql + tr[1], 0); X3A_RDTR(l1h, bql + tr[1], 2048);
          X3A_RDTR(l2l, bql + tr[2], 0); X3A_RDTR(l2h, bql + tr[2], 2048); X3A_RDTR(l3l, bql + tr[3], 0); X3A_RDTR(l3h, bql + tr[3], 2048);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(h0l), "+v"(h0h), "+v"(h1l), "+v"(h1h), "+v"(h2l), "+v"(h2h), "+v"(h3l), "+v"(h3h),
                       "+v"(l0l), "+v"(l0h), "+v"(l1l), "+v"(l1h), "+v"(l2l), "+v"(l2h), "+v"(l3l), "+v"(l3h)::"memory");
          dk[j][0] = x3a_mfma3(X3A_CAT(h0l, h0h), X3A_CAT(l0l, l0h), sh, sl, dk[j][0]);
          dk[j][1] = x3a_mfma3(X3A_CAT(h1l, h1h), X3A_CAT(l1l, l1h), sh, sl, dk[j][1]);
          dk[j][2] = x3a_mfma3(X3A_CAT(h2l, h2h), X3A_CAT(l2l, l2h), sh, sl, dk[j][2]);
          dk[j][3] = x3a_mfma3(X3A_CAT(h3l, h3h), X3A_CAT(l3l, l3h), sh, sl, dk[j][3]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r0 = k0 + (wave + 4 * j) * 16;
    bf16_t* dst = dqkv + (long)(t0 + r0) * H3 + H + h * 64;
    x3a_store_planes_bf16(dk[j], isc, dst, H3, dps, len - r0, lane);
    x3a_store_planes_bf16(dv[j], isc * (1.0f / 1024.0f), dst + H, H3, dps, len - r0, lane);
  }
  if (dbias != nullptr) {                              // (uniform; rows past the sequence end are exactly 0)
    f32x4 csk[4], csv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { csk[dt] = dk[0][dt] + dk[1][dt]; csv[dt] = dv[0][dt] + dv[1][dt]; }
    __syncthreads();
    x3a_colsum_flush(csk, isc, dbias + H + h * 64, reinterpret_cast<float*>(smem), tid);
    x3a_colsum_flush(csv, isc * (1.0f / 1024.0f), dbias + 2 * H + h * 64, reinterpret_cast<float*>(smem), tid);
  }
}

// ------------------------------------------------------------------------------------------ host
template <typename K>
static int x3a_set_lds(K kernel, size_t bytes, const char* name) {
  if (bytes > 65536 && hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
    simx_set_error("%s: cannot raise dynamic LDS to %zu", name, bytes);
    return SIMX_ERR_HIP;
  }
  return SIMX_OK;
}

extern "C" int simx_mha_x3_ok(int d, int max_len) {
  static const bool off = [] { const char* e = getenv("SIMX_MHA_X3"); return e && e[0] == '0'; }();     // SIMX_MHA_X3=0: the f32 MFMA kernels (A/B)
  return !off && d == 64 && max_len > 0 && max_len <= 4096 ? 1 : 0;
}

// q / k / v: fp16 plane pair [T, 3H] (lo at + qkv_plane_stride elements); ctx: fp16 plane pair [T, H]; lse: f32 [heads, T]
extern "C" int simx_mha_fwd_x3(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const void* qkv_planes,
                               long qkv_plane_stride, void* ctx_planes, long ctx_plane_stride, float* lse, const simx_dropout* dropd) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_MHA_FWD, s, 4.0 * T * max_len * heads * d);
  SIMX_REQUIRE(nseq > 0 && heads > 0 && T > 0 && qkv_planes && ctx_planes && lse && cu, SIMX_ERR_BAD_SHAPE, "mha_fwd_x3: bad arguments");
  SIMX_REQUIRE(simx_mha_x3_ok(d, max_len) && qkv_plane_stride % 8 == 0 && ctx_plane_stride % 4 == 0, SIMX_ERR_UNSUPPORTED,
               "mha_fwd_x3: needs head size 64, max_len <= 4096, plane strides %% 8 / %% 4 == 0");
  // (probabilities travel as 2^10 p / (1 - p_drop) in fp16 halves: finite only while 1 / (1 - p_drop) stays small)
  SIMX_REQUIRE(!dropd || dropd->p <= 0.9f, SIMX_ERR_UNSUPPORTED, "mha_x3: attention dropout above 0.9 is not supported by the plane-pair kernels");
  const DropCtx drop = make_drop(dropd);
  const float scale = 1.0f / sqrtf((float)d);
  int rc = SIMX_OK;
#define LF(NKT)                                                                                                            \
  do {                                                                                                                     \
    const size_t lds = (size_t)4 * NKT * 16 * 128;                                                                         \
    if (drop.thr) {                                                                                                        \
      rc = x3a_set_lds(mha_fwd_x3_kernel<NKT, true>, lds, "mha_fwd_x3");                                                   \
      if (rc) return rc;                                                                                                   \
      hipLaunchKernelGGL((mha_fwd_x3_kernel<NKT, true>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv_planes, qkv_plane_stride, \
                         (bf16_t*)ctx_planes, ctx_plane_stride, lse, cu, heads, T, scale, drop);                           \
    } else {                                                                                                               \
      rc = x3a_set_lds(mha_fwd_x3_kernel<NKT, false>, lds, "mha_fwd_x3");                                                  \
      if (rc) return rc;                                                                                                   \
      hipLaunchKernelGGL((mha_fwd_x3_kernel<NKT, false>), dim3(nseq * heads), dim3(256), lds, s, (const bf16_t*)qkv_planes, qkv_plane_stride, \
                         (bf16_t*)ctx_planes, ctx_plane_stride, lse, cu, heads, T, scale, drop);                           \
    }                                                                                                                      \
  } while (0)
  if (max_len <= 32) LF(2);
  else if (max_len <= 128) LF(8);
  else if (max_len <= 160) LF(10);
  else {                                               // chunked (161..256 tokens too: 2048 x 12 blocks of 256, 2.54 ms resident, 2.30 chunked): one workgroup per (sequence, head, 128-query chunk)
    const int nchunk = (max_len + XL_CH - 1) / XL_CH;
    const size_t lds = (size_t)4 * XL_TILE;
    if (drop.thr)
      hipLaunchKernelGGL((mha_fwd_x3_long_kernel<true>), dim3(nseq * heads * nchunk), dim3(256), lds, s, (const bf16_t*)qkv_planes, qkv_plane_stride,
                         (bf16_t*)ctx_planes, ctx_plane_stride, lse, cu, heads, T, nchunk, scale, drop);
    else
      hipLaunchKernelGGL((mha_fwd_x3_long_kernel<false>), dim3(nseq * heads * nchunk), dim3(256), lds, s, (const bf16_t*)qkv_planes, qkv_plane_stride,
                         (bf16_t*)ctx_planes, ctx_plane_stride, lse, cu, heads, T, nchunk, scale, drop);
  }
#undef LF
  SIMX_CHECK_LAUNCH("mha_fwd_x3");
  return SIMX_OK;
}

// dctx: f32 [T, H]; ctx: the forward's fp16 plane pair; dq / dk / dv leave as a bf16 plane pair [T, 3H]
extern "C" int simx_mha_bwd_x3(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const void* qkv_planes,
                               long qkv_plane_stride, const void* ctx_planes, long ctx_plane_stride, const float* lse, const float* dctx,
                               void* dqkv_planes, long dqkv_plane_stride, const simx_dropout* dropd) {
  return simx_mha_bwd_x3_bias(stream, nseq, heads, d, cu, max_len, T, qkv_planes, qkv_plane_stride, ctx_planes, ctx_plane_stride, lse, dctx,
                              dqkv_planes, dqkv_plane_stride, dropd, nullptr);
}
// same; dbias [3H] (may be NULL) += column sums of dq | dk | dv over the T tokens (the QKV projection's bias gradient; atomics)
extern "C" int simx_mha_bwd_x3_bias(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const void* qkv_planes,
                                    long qkv_plane_stride, const void* ctx_planes, long ctx_plane_stride, const float* lse, const float* dctx,
                                    void* dqkv_planes, long dqkv_plane_stride, const simx_dropout* dropd, float* dbias) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_MHA_BWD, s, 8.0 * T * max_len * heads * d);
  SIMX_REQUIRE(nseq > 0 && heads > 0 && T > 0 && qkv_planes && ctx_planes && lse && dctx && dqkv_planes && cu, SIMX_ERR_BAD_SHAPE, "mha_bwd_x3: bad arguments");
  SIMX_REQUIRE(simx_mha_x3_ok(d, max_len) && qkv_plane_stride % 8 == 0 && ctx_plane_stride % 8 == 0 && dqkv_plane_stride % 4 == 0 &&
                   (heads * 64) % 8 == 0 && (((uintptr_t)dctx) & 15) == 0, SIMX_ERR_UNSUPPORTED,
               "mha_bwd_x3: needs head size 64, max_len <= 4096, aligned plane strides");
  // (probabilities travel as 2^10 p / (1 - p_drop) in fp16 halves: finite only while 1 / (1 - p_drop) stays small)
  SIMX_REQUIRE(!dropd || dropd->p <= 0.9f, SIMX_ERR_UNSUPPORTED, "mha_x3: attention dropout above 0.9 is not supported by the plane-pair kernels");
  const DropCtx drop = make_drop(dropd);
  const float scale = 1.0f / sqrtf((float)d);
  int rc = SIMX_OK;
#define LBD(NKT, DROP)                                                                                                     \
  do {                                                                                                                     \
    const size_t lds1 = (size_t)4 * NKT * 16 * 128 + 64;                                                                   \
    const size_t lds2 = (size_t)4 * NKT * 16 * 128 + 2 * NKT * 16 * 4 + 64 + ((DROP) ? (size_t)NKT * 16 * NKT * 2 : 0);    \
    rc = x3a_set_lds(mha_bwd_dq_x3_kernel<NKT, DROP>, lds1, "mha_bwd_x3");                                                 \
    if (rc) return rc;                                                                                                     \
    rc = x3a_set_lds(mha_bwd_dkv_x3_kernel<NKT, DROP>, lds2, "mha_bwd_x3");                                                \
    if (rc) return rc;                                                                                                     \
    hipLaunchKernelGGL((mha_bwd_dq_x3_kernel<NKT, DROP>), dim3(nseq * heads), dim3(256), lds1, s, (const bf16_t*)qkv_planes, qkv_plane_stride, \
                       (const bf16_t*)ctx_planes, ctx_plane_stride, lse, dctx, (bf16_t*)dqkv_planes, dqkv_plane_stride, cu, heads, T, scale, drop, dbias); \
    hipLaunchKernelGGL((mha_bwd_dkv_x3_kernel<NKT, DROP>), dim3(nseq * heads), dim3(256), lds2, s, (const bf16_t*)qkv_planes, qkv_plane_stride, \
                       (const bf16_t*)ctx_planes, ctx_plane_stride, lse, dctx, (bf16_t*)dqkv_planes, dqkv_plane_stride, cu, heads, T, scale, drop, dbias); \
  } while (0)
#define LB(NKT) do { if (drop.thr) LBD(NKT, true); else LBD(NKT, false); } while (0)
#define LBL(DROP)                                                                                                          \
  do {                                                                                                                     \
    const int nchunk = (max_len + XL_CH - 1) / XL_CH;                                                                      \
    const size_t lds1 = (size_t)4 * XL_TILE + 64, lds2 = (size_t)4 * XL_TILE + 2 * XL_CH * 4 + 64 + ((DROP) ? (size_t)8 * XL_CH * 2 : 0); \
    rc = x3a_set_lds(mha_bwd_dq_x3_long_kernel<DROP>, lds1, "mha_bwd_x3");                                                 \
    if (rc) return rc;                                                                                                     \
    rc = x3a_set_lds(mha_bwd_dkv_x3_long_kernel<DROP>, lds2, "mha_bwd_x3");                                                \
    if (rc) return rc;                                                                                                     \
    hipLaunchKernelGGL((mha_bwd_dq_x3_long_kernel<DROP>), dim3(nseq * heads * nchunk), dim3(256), lds1, s, (const bf16_t*)qkv_planes, qkv_plane_stride, \
                       (const bf16_t*)ctx_planes, ctx_plane_stride, lse, dctx, (bf16_t*)dqkv_planes, dqkv_plane_stride, cu, heads, T, nchunk, scale, drop, dbias); \
    hipLaunchKernelGGL((mha_bwd_dkv_x3_long_kernel<DROP>), dim3(nseq * heads * nchunk), dim3(256), lds2, s, (const bf16_t*)qkv_planes, qkv_plane_stride, \
                       (const bf16_t*)ctx_planes, ctx_plane_stride, lse, dctx, (bf16_t*)dqkv_planes, dqkv_plane_stride, cu, heads, T, nchunk, scale, drop, dbias); \
  } while (0)
  if (max_len <= 32) LB(2);
  else if (max_len <= 128) LB(8);
  else if (max_len <= 160) LB(10);
  else if (drop.thr) LBL(true);          // (161..256 tokens too: 2048 x 12 blocks of 256, 9.58 ms resident -- one workgroup per CU -- 8.57 chunked)
  else LBL(false);
#undef LBL
#undef LB
#undef LBD
  SIMX_CHECK_LAUNCH("mha_bwd_x3");
  return SIMX_OK;
}
