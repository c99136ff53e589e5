// Shared device/host helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/simx.h"

typedef unsigned short bf16_t;   // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(2))) float f32x2;

#define WAVE 64
#define SIMX_MAX_DEVICES 64   // per-device launch state tables (one process per GPU is the deployment; this is the bound)

// ---- error plumbing (api.cpp) ---------------------------------------------------------
void simx_set_error(const char* fmt, ...);
int simx_compute_cus(int device_cus);      // CUs the persistent kernels may fill (simx_set_compute_cus / SIMX_COMPUTE_CUS; encoder.hip)
// SIMX_DETERMINISTIC=1 (det.hip): ordered reductions instead of f32 atomics
bool simx_det();
float* simx_det_ws(hipStream_t st, size_t bytes);                       // per-stream scratch, NULL (+ error text) on failure
int simx_det_reduce(hipStream_t st, const float* part, long stride, int nparts, int n, float* o0, float* o1, float* o2, const float* gs);
int simx_det_scatter_rows(hipStream_t st, int T, int H, const int* idx, const float* rows, float* table, int table_rows);
#define SIMX_CHECK_LAUNCH(name)                                                        \
  do {                                                                                 \
    hipError_t e__ = hipGetLastError();                                                \
    if (e__ != hipSuccess) {                                                           \
      simx_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return SIMX_ERR_HIP;                                                             \
    }                                                                                  \
  } while (0)
#define SIMX_REQUIRE(cond, code, ...)                                                  \
  do {                                                                                 \
    if (!(cond)) {                                                                     \
      simx_set_error(__VA_ARGS__);                                                     \
      return code;                                                                     \
    }                                                                                  \
  } while (0)

// ---- bf16 <-> f32 (round to nearest even; NaN kept quiet) --------------------------------
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// f32 -> bf16 through the native type: hipcc lowers it to v_cvt_pk_bf16_f32 (round to nearest even, NaN kept),
// one instruction per PAIR -- the hand-rolled integer rounding cost ~12 VALU ops and a divergent NaN branch per element.
typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));
typedef float f32v2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, static_cast<__bf16>(f)); }
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32v2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16v2_t));
}

// ---- fp16 (IEEE half): the second 16-bit operand format (apex-O1-like mode: 11-bit significands, needs a loss scale in
// backward).  gfx950 has v_cvt_pk_f16_f32 (round to nearest even, subnormal results kept) and its MFMAs honour fp16
// subnormal inputs (tools/probe_f16.hip), so small probabilities / gradients degrade gracefully instead of flushing.
typedef _Float16 f16_t;
typedef _Float16 f16v2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {
  const f32v2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16v2_t));
}

// The two 16-bit formats behind one interface: kernels are templates over T16 in {bf16_t, f16_t}; fragments travel as
// bf16x8 (8 x 16 raw bits) whatever the format.  lo / hi unpack the two halves of a 32-bit word, pack2 rounds a pair.
template <typename T> struct H16;
template <> struct H16<bf16_t> {
  static constexpr int code = SIMX_BF16;
  static __device__ __forceinline__ float lo(uint32_t w) { return __uint_as_float(w << 16); }
  static __device__ __forceinline__ float hi(uint32_t w) { return __uint_as_float(w & 0xFFFF0000u); }
  static __device__ __forceinline__ float one(short s) { return __uint_as_float(((uint32_t)(unsigned short)s) << 16); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack2bf(a, b); }
  static __device__ __forceinline__ short bits(float a) { return (short)f2bf(a); }
  static __device__ __forceinline__ f32x4 mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
};
template <> struct H16<f16_t> {
  static constexpr int code = SIMX_F16;
  static __device__ __forceinline__ float lo(uint32_t w) { return (float)__builtin_bit_cast(f16v2_t, w)[0]; }
  static __device__ __forceinline__ float hi(uint32_t w) { return (float)__builtin_bit_cast(f16v2_t, w)[1]; }
  static __device__ __forceinline__ float one(short s) { return (float)__builtin_bit_cast(f16_t, s); }
  static __device__ __forceinline__ uint32_t pack2(float a, float b) { return pack2h(a, b); }
  static __device__ __forceinline__ short bits(float a) { return __builtin_bit_cast(short, (f16_t)a); }
  static __device__ __forceinline__ f32x4 mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  }
};

// A value that is about to be split into a 16-bit pair (hi = rnd16(x), lo = rnd16(x - hi)) must exist as ONE f32 register
// first.  Handed an unrounded product x = a * b, hipcc contracts the subtraction into v_fma_mix(a, b, -hi) and selects the
// conversion (f16)x TWICE: as v_fma_mixlo(a, b, 0) -- rounded once -- for that subtraction, and as v_mul + v_cvt_pk -- rounded
// twice -- for the hi that is stored or fed to the MFMA.  In double-rounding cases the two differ by one ulp of the 16-bit
// format, and the pair is then off by 2^-11 instead of 2^-22 (found in mha_fwd_x3_kernel once its dropout multiply and the
// split shared a basic block: three rows in 10^4 wrong at 5e-5).  Every split goes through this.
__device__ __forceinline__ float f32_pin(float x) { asm("" : "+v"(x)); return x; }

// ---- stream correction in ONE byte per element (simx.h stream_lo): x = hi + (b - 128) * ulp(hi) / 256, ulp(hi) = the
// spacing of the 16-bit format at hi's exponent.  |x - hi| <= ulp / 2, so b lands in [0, 256]: clamped to [1, 255] (at most
// 1/256 ulp lost at the two ends).  The stream then carries 11 + 8 = 19 significand bits in fp16 (8 + 8 = 16 in bf16) at 3 B
// per element instead of 4.  lo8_scale_bits: the f32 bit pattern of ulp(hi) / 256 from hi's raw 16 bits.
template <typename T> struct Lo8;
template <> struct Lo8<f16_t> {
  static __device__ __forceinline__ uint32_t field(uint32_t h) { const uint32_t e = (h >> 10) & 31u; return (e > 1u ? e : 1u) + 94u; }      // 2^(e-15-10-8)
};
template <> struct Lo8<bf16_t> {
  static __device__ __forceinline__ uint32_t field(uint32_t h) { const uint32_t e = (h >> 7) & 255u; return (e > 16u ? e : 16u) - 15u; }   // 2^(e-127-7-8)
};
template <> struct Lo8<float> {                          // (f32 tensors carry no correction; never instantiated on a live path)
  static __device__ __forceinline__ uint32_t field(uint32_t) { return 127u; }
};
// the four corrections of the elements packed in (w0, w1) [two 16-bit values each] from the byte quad q
template <typename T>
__device__ __forceinline__ void lo8_decode4(uint32_t w0, uint32_t w1, uint32_t q, float (&lo)[4]) {
  const uint32_t hb[4] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float s = __uint_as_float(Lo8<T>::field(hb[e]) << 23);
    lo[e] = ((float)((q >> (8 * e)) & 0xFFu) - 128.0f) * s;
  }
}
// r[e] = what the 16-bit rounding of element e dropped (o - hi); (w0, w1) = the packed 16-bit values just stored
template <typename T>
__device__ __forceinline__ uint32_t lo8_encode4(uint32_t w0, uint32_t w1, const float (&r)[4]) {
  const uint32_t hb[4] = {w0 & 0xFFFFu, w0 >> 16, w1 & 0xFFFFu, w1 >> 16};
  uint32_t q = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float inv = __uint_as_float((254u - Lo8<T>::field(hb[e])) << 23);
    const float t = __builtin_amdgcn_fmed3f(fmaf(r[e], inv, 128.0f), 1.0f, 255.0f);
    q |= ((uint32_t)__builtin_rintf(t)) << (8 * e);
  }
  return q;
}

// 4 consecutive elements of raw 16-bit storage in format F (the MFMA kernels keep `bf16_t*` = raw 16-bit pointers for both)
template <typename F>
__device__ __forceinline__ void ld4h(const bf16_t* p, float (&v)[4]) {
  const uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = H16<F>::lo(t.x); v[1] = H16<F>::hi(t.x); v[2] = H16<F>::lo(t.y); v[3] = H16<F>::hi(t.y);
}
template <typename F>
__device__ __forceinline__ void st4h(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(H16<F>::pack2(v[0], v[1]), H16<F>::pack2(v[2], v[3]));
}

// element load/store by activation type
template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float rnd(float v) { return v; }        // the value a store of v leaves in memory
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
  static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }
};
template <> struct Elem<f16_t> {
  static __device__ __forceinline__ float ld(const f16_t* p) { return (float)*p; }
  static __device__ __forceinline__ void st(f16_t* p, float v) { *p = (f16_t)v; }
  static __device__ __forceinline__ float rnd(float v) { return (float)(f16_t)v; }
};

// 4-element vector load/store (16 B for f32, 8 B for bf16); pointers must be so aligned
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
  float4 t = *reinterpret_cast<const float4*>(p);
  v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xFFFF0000u);
  v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xFFFF0000u);
}
__device__ __forceinline__ void ld4(const f16_t* p, float (&v)[4]) {
  uint2 t = *reinterpret_cast<const uint2*>(p);
  v[0] = H16<f16_t>::lo(t.x); v[1] = H16<f16_t>::hi(t.x); v[2] = H16<f16_t>::lo(t.y); v[3] = H16<f16_t>::hi(t.y);
}
__device__ __forceinline__ void st4(f16_t* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2h(v[0], v[1]), pack2h(v[2], v[3]));
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack2bf(v[0], v[1]), pack2bf(v[2], v[3]));
}

// ---- stateless dropout mask (see simx_dropout in include/simx.h) ---------------------------------------------
// One 32-bit hash serves FOUR consecutive columns (its four bytes against an 8-bit threshold): the 32-bit integer multiplies of
// the mix are the cost of a mask (v_mul_lo_u32 issues at a quarter of the VALU rate), and until round 5 a hash served two
// columns (16-bit lanes).  The drop probability is therefore realised in steps of 1/256 -- thr = round(256 p), 26/256 =
// 0.1016 for the reference's p = 0.1 -- and kept values are scaled by 256 / (256 - thr), the reciprocal of the REALISED keep
// rate, so that E[mask] = 1 exactly.  The reference fixes p only (SimANS/model/models.py:70-72); its RNG stream cannot be
// replayed on a GPU kernel in any case (oracle/bert.py restates THIS definition).
struct DropCtx { uint32_t thr; float scale; uint32_t seed; uint32_t stream; };   // thr == 0: disabled
static inline DropCtx make_drop(const simx_dropout* d) {
  DropCtx c = {0u, 1.0f, 0u, 0u};
  // p is never clamped into the representable range: 0 < p < 1/512 runs as NO dropout (thr = 0, the nearest realisable rate) and
  // p >= 1 drops everything like torch (thr = 256: no byte passes; scale 0 instead of 256 / 0).  In between the rate is
  // round(256 p) / 256, at most 255/256.
  if (d && d->p >= 1.0f / 512.0f) {
    uint32_t t = (uint32_t)(d->p * 256.0f + 0.5f);
    t = d->p >= 1.0f ? 256u : (t > 255u ? 255u : t);
    c.thr = t;
    c.scale = t >= 256u ? 0.f : 256.0f / (float)(256u - t);
    c.seed = d->seed;
    c.stream = d->stream;
  }
  return c;
}
__device__ __forceinline__ uint32_t drop_mix(uint32_t seed, uint32_t stream, uint32_t row, uint32_t colquad) {
  uint32_t h = (row * 0x9E3779B1u) ^ ((colquad + stream * 0x632BE5ABu) * 0x85EBCA77u) ^ seed;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h;
}
// keep flags of the four columns of a hash: bit e = byte e >= thr
__device__ __forceinline__ uint32_t drop_keep4(uint32_t h, uint32_t thr) {
  return ((h & 0xFFu) >= thr ? 1u : 0u) | (((h >> 8) & 0xFFu) >= thr ? 2u : 0u) | (((h >> 16) & 0xFFu) >= thr ? 4u : 0u) |
         ((h >> 24) >= thr ? 8u : 0u);
}
// multiplier (0 or 256 / (256 - thr)) for element (row, col)
__device__ __forceinline__ float drop_mult(const DropCtx& d, uint32_t row, uint32_t col) {
  const uint32_t h = drop_mix(d.seed, d.stream, row, col >> 2);
  return (((h >> ((col & 3u) * 8u)) & 0xFFu) >= d.thr) ? d.scale : 0.f;
}
// 4 consecutive columns starting at a col that is a MULTIPLE OF 4 (one hash)
__device__ __forceinline__ void drop_mult4(const DropCtx& d, uint32_t row, uint32_t col, float (&m)[4]) {
  const uint32_t h = drop_mix(d.seed, d.stream, row, col >> 2);
  m[0] = ((h & 0xFFu) >= d.thr) ? d.scale : 0.f;
  m[1] = (((h >> 8) & 0xFFu) >= d.thr) ? d.scale : 0.f;
  m[2] = (((h >> 16) & 0xFFu) >= d.thr) ? d.scale : 0.f;
  m[3] = ((h >> 24) >= d.thr) ? d.scale : 0.f;
}

// ---- wave64 reductions -------------------------------------------------------------------
// DPP butterflies (no LDS traffic, ~1 VALU op per step): quad swaps, half-row / row mirrors give every lane its
// 16-lane row total; row_bcast15 / row_bcast31 chain the four rows into lane 63; v_readlane broadcasts it as a scalar.
// (__shfl_xor lowers to ds_bpermute_b32: an LDS round trip per step.)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_mov<0xB1, 0xF>(v);          // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E, 0xF>(v);          // quad_perm [2,3,0,1]
  v += dpp_mov<0x141, 0xF>(v);         // row_half_mirror
  v += dpp_mov<0x140, 0xF>(v);         // row_mirror      -> every lane: total of its row of 16
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));   // row_bcast15 -> rows 1,3
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));   // row_bcast31 -> rows 2,3
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_mov<0xB1, 0xF>(v));
  v = fmaxf(v, dpp_mov<0x4E, 0xF>(v));
  v = fmaxf(v, dpp_mov<0x141, 0xF>(v));
  v = fmaxf(v, dpp_mov<0x140, 0xF>(v));
  // rows -> wave: lanes 15/31/47/63 hold the row maxima
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 15));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 31));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 47));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
  return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// ---- erf-GELU (HF "gelu", LEAD/modeling_bert.py:440-452) -----------------------------------
__device__ __forceinline__ float gelu_erf(float u) { return 0.5f * u * (1.0f + erff(u * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_erf_grad(float u) {
  return 0.5f * (1.0f + erff(u * 0.70710678118654752f)) + u * __expf(-0.5f * u * u) * 0.39894228040143268f;
}

// fast erf-GELU for the bf16 epilogues.  gelu(u) = u * Phi(u) with Phi(u) ~= sigmoid(u * (c0 + c1 u^2 + c2 u^4)): the
// three coefficients are a minimax fit against the exact erf form (tools/fit_gelu.py): |gelu error| <= 2.5e-5 for every
// u, i.e. 1/100 of the bf16 rounding of an O(1) activation, and the derivative of the same expression is used in
// backward (|gelu' error| <= 1.1e-4).  8 VALU + v_exp + v_rcp per element instead of the 13 + 2 of Abramowitz-Stegun
// 7.1.26 (kept below for reference; the f32 parity kernels use erff).  Saturates correctly: u -> -inf gives
// exp -> inf, rcp -> 0; u -> +inf gives exp -> 0, rcp -> 1.
#define GELU_C0 1.5950157608f
#define GELU_C1 0.0740112985f
#define GELU_C2 (-0.000703034548f)
__device__ __forceinline__ float gelu_sigmoid(float u, float& s) {
  const float uc = __builtin_amdgcn_fmed3f(u, -7.0f, 7.0f);      // the quartic turns around near |u| = 10.7; sigmoid(u p) is
  s = uc * uc;                                                    // 1 - 2e-9 / 2e-9 at |u| = 7 already
  const float p = fmaf(fmaf(GELU_C2, s, GELU_C1), s, GELU_C0);
  const float e = __builtin_amdgcn_exp2f(uc * p * -1.4426950408889634f);
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_fast(float u) {
  float s;
  return u * gelu_sigmoid(u, s);
}
__device__ __forceinline__ float gelu_grad_fast(float u) {
  float s;
  const float r = gelu_sigmoid(u, s);
  const float up = fmaf(fmaf(5.0f * GELU_C2, s, 3.0f * GELU_C1), s, GELU_C0);       // d/du [u p(u^2)]
  return r * fmaf(u * (1.0f - r), up, 1.0f);
}
// gelu(u) and gelu'(u) from ONE sigmoid: the forward's GELU epilogue stores both (the derivative in the slot the
// pre-activation used to occupy -- backward never needs u itself), so the DGELU epilogue is a multiply
__device__ __forceinline__ void gelu_both_fast(float u, float& g, float& dg) {
  float s;
  const float r = gelu_sigmoid(u, s);
  const float up = fmaf(fmaf(5.0f * GELU_C2, s, 3.0f * GELU_C1), s, GELU_C0);
  g = u * r;
  dg = r * fmaf(u * (1.0f - r), up, 1.0f);
}
// The same on a PAIR of elements with explicitly packed arithmetic (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32: two f32 per
// lane per instruction): the GELU epilogue of the persistent GEMM is VALU-bound -- no MFMA runs beside it -- and left to the
// SLP vectoriser about half of its multiplies stayed scalar.  Coefficients carry the -log2(e) of the exponent already.
typedef float f32p_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_both_fast2(f32p_t u, f32p_t& g, f32p_t& dg) {
  const f32p_t uc = {__builtin_amdgcn_fmed3f(u.x, -7.0f, 7.0f), __builtin_amdgcn_fmed3f(u.y, -7.0f, 7.0f)};
  const f32p_t s = uc * uc;
  constexpr float L2E = -1.4426950408889634f;
  const f32p_t p = (s * (GELU_C2 * L2E) + (GELU_C1 * L2E)) * s + (GELU_C0 * L2E);
  const f32p_t t = uc * p;
  const f32p_t d = {1.0f + __builtin_amdgcn_exp2f(t.x), 1.0f + __builtin_amdgcn_exp2f(t.y)};
  const f32p_t r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  const f32p_t up = (s * (5.0f * GELU_C2) + (3.0f * GELU_C1)) * s + GELU_C0;       // d/du [u p(u^2)]
  g = u * r;
  dg = r * ((u - g) * up + 1.0f);                                                     // u (1 - r) = u - g
}
__device__ __forceinline__ f32p_t gelu_fast2(f32p_t u) {
  const f32p_t uc = {__builtin_amdgcn_fmed3f(u.x, -7.0f, 7.0f), __builtin_amdgcn_fmed3f(u.y, -7.0f, 7.0f)};
  const f32p_t s = uc * uc;
  constexpr float L2E = -1.4426950408889634f;
  const f32p_t t = uc * ((s * (GELU_C2 * L2E) + (GELU_C1 * L2E)) * s + (GELU_C0 * L2E));
  const f32p_t r = {__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t.x)), __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t.y))};
  return u * r;
}
// Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7): one v_rcp + one v_exp + 7 FMAs; the exponential is shared with the derivative.
__device__ __forceinline__ void erf_and_gauss(float u, float& erf_x, float& gauss) {
  const float x = fabsf(u) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, x, 1.0f));      // v_rcp_f32 (1 ulp), not an IEEE divide
  gauss = __expf(-x * x);                                 // = exp(-u^2/2)
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  erf_x = copysignf(1.0f - p * t * gauss, u);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline bool simx_is16(int dtype) { return dtype == SIMX_BF16 || dtype == SIMX_F16; }
static inline bool simx_dtype_ok(int dtype) { return dtype == SIMX_F32 || simx_is16(dtype); }
static inline bool simx_is_f32(int dtype) { return dtype == SIMX_F32 || dtype == SIMX_F32_SPLIT_H || dtype == SIMX_F32_SPLIT_B; }
static inline size_t simx_esz(int dtype) { return simx_is_f32(dtype) ? 4 : 2; }
// csrc/attention_f32.hip: f32 attention on the f32 matrix cores (head size 64, sequences <= 256)
bool simx_mha_f32_ok(int d, int max_len);
// (ctx_ps / dqkv_ps > 0: the "operand planes" forms -- ctx an fp16 plane pair, dqkv a bf16 plane pair; csrc/gemm_xp.hip)
int simx_mha_fwd_f32(hipStream_t s, int nseq, int heads, const int32_t* cu, int max_len, int T, const float* qkv, float* ctx, float* lse,
                     float scale, DropCtx drop, long ctx_ps = 0);
int simx_mha_bwd_f32(hipStream_t s, int nseq, int heads, const int32_t* cu, int max_len, int T, const float* qkv, const float* ctx,
                     const float* lse, const float* dctx, float* dqkv, float scale, DropCtx drop, long ctx_ps = 0, long dqkv_ps = 0);
// csrc/gemm_x3.hip: f32 GEMMs on the 16-bit matrix cores (fmt = SIMX_F16 / SIMX_BF16: the format of the split halves)
bool simx_x3_nt_ok(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* C, int ldc, const float* bias,
                   const float* res, int ldr, const float* aux, int ldaux, const float* C2, int ldc2);
int simx_x3_gemm_nt(hipStream_t s, int fmt, int epi, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    const float* bias, const float* res, int ldr, const float* aux, int ldaux, float* C2, int ldc2, DropCtx drop);
// grouped weight cast (gemm.hip): up to SIMX_CAST_GROUP_MAX matrices in one launch (2.5 KB of kernel arguments); out / outT may be NULL per job
#define SIMX_CAST_GROUP_MAX 64
struct SimxCastJob { const float* w; void* out; void* outT; int rows, cols, tile_end, pad_; };   // tile_end: cumulative 32 x 32 tiles
struct SimxCastGroup { int n; SimxCastJob job[SIMX_CAST_GROUP_MAX]; };
int simx_transpose_cast_group(hipStream_t s, int out_dtype, const SimxCastGroup* g);
// csrc/gemm_xp.hip: the fp32 engine's GEMMs on pre-split operand planes.  Weight planes in one launch: per matrix W [rows, cols]
// f32 -> fp16 plane pair of W (planes_h: hi [rows, cols], lo at + rows * cols elements), bf16 plane pair of W^T (planesT_b,
// [cols, rows]) and W^T in f32 (wT); any output may be NULL
#define SIMX_SPLIT_GROUP_MAX 48
struct SimxSplitJob { const float* w; void* planes_h; void* planesT_b; float* wT; int rows, cols, tile_end, pad_; };
struct SimxSplitGroup { int n; SimxSplitJob job[SIMX_SPLIT_GROUP_MAX]; };
int simx_split_weight_group(hipStream_t s, const SimxSplitGroup* g);
bool simx_x3_tn_ok(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* C, int ldc);
size_t simx_x3_tn_workspace_bytes(int M, int N, int K);
int simx_x3_gemm_tn(hipStream_t s, int fmt, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    int accumulate, void* ws, size_t ws_bytes, float* dbias);
// run STMT with TT bound to the element type of `dtype` (the three activation types of the engine)
#define SIMX_DISPATCH3(DTYPE, TT, ...)                                   \
  do {                                                                   \
    if ((DTYPE) == SIMX_F32) { using TT = float; __VA_ARGS__; }          \
    else if ((DTYPE) == SIMX_BF16) { using TT = bf16_t; __VA_ARGS__; }   \
    else { using TT = f16_t; __VA_ARGS__; }                              \
  } while (0)
#define SIMX_DISPATCH16(DTYPE, TT, ...)                                  \
  do {                                                                   \
    if ((DTYPE) == SIMX_BF16) { using TT = bf16_t; __VA_ARGS__; }        \
    else { using TT = f16_t; __VA_ARGS__; }                              \
  } while (0)

// ---- gradient scale of the fp16 engine (simx.h "loss scale"): gs = device pointer to {S, 1/S} or NULL.  Activation
// gradients travel multiplied by S; every kernel that ACCUMULATES INTO THE f32 PARAMETER GRADIENTS multiplies by 1/S, so
// the flat gradient buffers always hold true gradients.
__device__ __forceinline__ float gs_scale(const float* gs) { return gs ? gs[0] : 1.0f; }
__device__ __forceinline__ float gs_inv(const float* gs) { return gs ? gs[1] : 1.0f; }
