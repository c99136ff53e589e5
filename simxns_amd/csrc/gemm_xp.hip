// The fp32 engine's dense GEMMs on PRE-SPLIT operand planes.
//
// gemm_x3.hip computes an f32 product on the 16-bit matrix cores from hi + lo splits of its f32 operands taken ON THE FLY:
// f32 global loads through registers, a VALU split per element per use, 128x128 tiles, every M-tile re-reading its weight
// block -- 0.35 of the three-MFMA ceiling, bound by that load stream.  Here the split is taken ONCE, by the
// kernel that produces a tensor: every f32 tensor that feeds a GEMM also exists as a "plane pair", two 16-bit matrices
//     x  =  hi + lo          hi = rnd16(x),  lo = rnd16(x - hi)        (lo at hi + plane_stride elements, same leading dimension)
// and the GEMMs are the 16-bit persistent kernels of gemm.hip with the contraction tripled:
//     A . B^T  =  Ahi.Blo + Alo.Bhi + Ahi.Bhi                          (corrections first; lo.lo is below f32 round-off)
// i.e. LDS-DMA operand staging with no registers and no VALU in the loop, 256x256 tiles, three A / two B stage slots, the
// hand-scheduled fragment-read / MFMA interleave of gemm_nt_p3_kernel -- the stream stage s of a tile is k block s / 3 of
// term s % 3, so the hi plane a term re-reads was fetched one or two stages earlier and is an L2 hit.
//   format F = f16_t : forward GEMMs (22 significand bits per operand element; see gemm_x3.hip)
//   format F = bf16_t: backward GEMMs (16 bits at any magnitude)
// Outputs are f32 (through an LDS transposition: full 128-B lines per row) and / or plane pairs for the next GEMM:
//   SIMX_EPI_NONE  : C = acc + bias (dropout) (+ in)                      f32
//   SIMX_EPI_GELU  : C = gelu'(u) f32 (skipped when C == NULL), Cp = planes of gelu(u), u = acc + bias
//   SIMX_EPI_DGELU : Cp = planes of acc * in                              (in = the stored derivative, f32)
// Reference op: the nn.Linear's of LEAD/modeling_bert.py:285-310, 385, 450, 463 in fp32 (SimANS/train_MS_Pas_AR2.sh:8-26).
#include <mutex>
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "p3.h"

template <typename F>
__device__ __forceinline__ void xp_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  a = f32_pin(a); b = f32_pin(b);
  hi = H16<F>::pack2(a, b);
  lo = H16<F>::pack2(a - H16<F>::lo(hi), b - H16<F>::hi(hi));
}

// element offset of stream stage s inside an operand: k block s / 3, and the lo plane for the one term that takes it
// (A: term 1, B: term 0).  s < 2^15.
__device__ __forceinline__ long xp_off(int s, long plane_stride, int lo_term) {
  const int ks = (s * 43691) >> 17;
  const int term = s - 3 * ks;
  return (long)ks * 64 + (term == lo_term ? plane_stride : 0L);
}

// gelu(u), gelu'(u) at f32 grade without erff: Abramowitz-Stegun 7.1.26 (|erf error| <= 1.5e-7), one v_rcp + one v_exp,
// the exponential shared with the derivative (common.h erf_and_gauss)
__device__ __forceinline__ void xp_gelu_both(float u, float& g, float& dg) {
  float e, ga;
  erf_and_gauss(u, e, ga);
  const float phi = 0.5f * (1.0f + e);
  g = u * phi;
  dg = fmaf(u * ga, 0.39894228040143268f, phi);
}

// ------------------------------------------------------------------------------------------ NT
// The main loop is gemm_nt_p3_kernel's (csrc/gemm.hip; comments there), with stage -> (k block, term) addressing.  The
// epilogue differs: inputs (`in`) arrive by direct 16-B global loads in the accumulator layout, one 16-row chunk ahead, all
// waits counted; results pass through the wave's 4 KB of the A slot the tile's last stage freed and leave as full lines.
template <typename F, int EPI, bool HAS_IN, bool DROP, bool RING>
__global__ __launch_bounds__(512, 2) void gemm_nt_xp_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, long a_ps, const bf16_t* __restrict__ B, int ldb, long b_ps,
    float* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ in, int ldin,
    bf16_t* __restrict__ Cp, int ldcp, long cp_ps, int tiles_n, int ntiles, DropCtx drop, float* __restrict__ colsum, int m_valid) {
  // colsum (SIMX_EPI_DGELU only, may be NULL): += column sums of the output over rows < m_valid -- the bias gradient of the dense
  // layer whose pre-activation gradient this launch produces (B1 from du), so that its wgrad GEMM needs no fused bias pass
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int nst = 3 * (K / 64);                   // >= 6
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int swz = (fr >> 1) & 7;
  const uint32_t rowA = (uint32_t)((wr * 128 + fr) * 128), rowB = (uint32_t)((wc * 64 + fr) * 128);
  const uint32_t oA0 = rowA + (uint32_t)(((0 + fg) ^ swz) << 4), oA1 = rowA + (uint32_t)(((4 + fg) ^ swz) << 4);
  const uint32_t oB0 = rowB + (uint32_t)(((0 + fg) ^ swz) << 4), oB1 = rowB + (uint32_t)(((4 + fg) ^ swz) << 4);
  const uint32_t ldsB = lds0 + 98304u;            // A slots: lds0 + {0,1,2} * 32 KB ; B slots: ldsB + {0,1} * 32 KB
  const int lr = lane >> 3;
  const int ec0 = ((lane & 7) ^ (lane >> 4)) << 3, ec1 = ((lane & 7) ^ (4 + (lane >> 4))) << 3;
  const uint32_t bmask = bias ? 0xFFFFFFFFu : 0u;
  const uint32_t offA0 = (uint32_t)(lr * lda + ec0) * 2, offA1 = (uint32_t)(lr * lda + ec1) * 2;
  const uint32_t offB0 = (uint32_t)(lr * ldb + ec0) * 2, offB1 = (uint32_t)(lr * ldb + ec1) * 2;
#define XP_AK(S) xp_off((S), a_ps, 1)
#define XP_BK(S) xp_off((S), b_ps, 0)
#define P_LANE(L) int L = lane; asm volatile("" : "+v"(L))

  int v = blockIdx.x;
  int tile = xcd_remap(v, ntiles);
  int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  // RING (needs lda == ldb): FOUR loads per k block instead of six.  The three terms of a k block run as sub-stages
  //   S0 = Ahi.Blo, S1 = Ahi.Bhi, S2 = Alo.Bhi  -- S0 / S1 share the staged Ahi tile, S1 / S2 the Bhi tile --
  // from five 32 KB tile slots: Blo always in slot L, the other four a ring with roles H (Ahi of this k block), G (Bhi), M (Alo)
  // and Hn (Ahi of the next k block), rotating by one per k block.  Load order ... Ahi(k) Blo(k) Bhi(k) Alo(k) Ahi(k+1) ...; at
  // the boundary inside S0 the freed L takes Blo(k+1), inside S1 the freed H takes Bhi(k+1), inside S2 the freed G and M take
  // Alo(k+1) and Ahi(k+2): three tile loads (96 KB) stay in flight behind the two being consumed, as in the p3 ring, and the
  // counted waits are vmcnt(8), vmcnt(8), vmcnt(4).
  const int nks = K / 64;
  const uint32_t ldsL = lds0 + 131072u;
#define R_SLOT(J) (lds0 + (uint32_t)(((J) & 3) * 32768))
  int rr = 0;                                      // ring position of H
  if constexpr (RING) {
    p3_half(A, lda, m0, 0, R_SLOT(0), wave, offA0, offA1);
    p3_half(B + b_ps, ldb, n0, 0, ldsL, wave, offB0, offB1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    p3_half(B, ldb, n0, 0, R_SLOT(1), wave, offB0, offB1);
    p3_half(A + a_ps, lda, m0, 0, R_SLOT(2), wave, offA0, offA1);
    p3_half(A, lda, m0, 64, R_SLOT(3), wave, offA0, offA1);
  } else {
    p3_half(B, ldb, n0, XP_BK(0), ldsB, wave, offB0, offB1);
    p3_half(A, lda, m0, XP_AK(0), lds0, wave, offA0, offA1);
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    p3_half(B, ldb, n0, XP_BK(1), ldsB + 32768u, wave, offB0, offB1);
    p3_half(A, lda, m0, XP_AK(1), lds0 + 32768u, wave, offA0, offA1);
    p3_half(A, lda, m0, XP_AK(2), lds0 + 65536u, wave, offA0, offA1);
  }
  int a0 = 0, b0 = 0;
  f32x4 bq0, bq1, bq2, bq3;
  {
    const float* bp0 = bias ? bias + n0 + wc * 64 : reinterpret_cast<const float*>(A);
    const uint32_t boff = (uint32_t)(fg * 16);
    P_GLD4(bq0, boff, bp0, 0); P_GLD4(bq1, boff, bp0, 64); P_GLD4(bq2, boff, bp0, 128); P_GLD4(bq3, boff, bp0, 192);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq0), "+v"(bq1), "+v"(bq2), "+v"(bq3)::"memory");
  }

#define V3_MFMA_ROW(I, AF, B0, B1, B2, B3)                                                     \
  acc[I][0] = H16<F>::mfma(B0, AF, acc[I][0]);             \
  acc[I][1] = H16<F>::mfma(B1, AF, acc[I][1]);             \
  acc[I][2] = H16<F>::mfma(B2, AF, acc[I][2]);             \
  acc[I][3] = H16<F>::mfma(B3, AF, acc[I][3])
#define V3_RD1(FR, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=&v"(FR) : "v"(ADDR) : "memory")
#define V3_SB __builtin_amdgcn_sched_barrier(0)
#define P3_BOUNDARY_WAIT() asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory")
#define P3_LGKM_WAIT "s_waitcnt lgkmcnt(0)"
  const char* pa_g = nullptr; const char* pb_g = nullptr;
  uint32_t pa_slot = 0, pb_slot = 0;
  bool pa_pend = false, pb_pend = false;
#define P3_HA(J) do { if (pa_pend) { P_DMA16(((J) & 1) ? offA1 : offA0, pa_g + (long)(J) * 8 * lda * 2, pa_slot + (uint32_t)((J) * 1024)); if ((J) == 3) pa_pend = false; } } while (0)
#define P3_HB(J) do { if (pb_pend) { P_DMA16(((J) & 1) ? offB1 : offB0, pb_g + (long)(J) * 8 * ldb * 2, pb_slot + (uint32_t)((J) * 1024)); if ((J) == 3) pb_pend = false; } } while (0)
#define P3_HAX(ROW, S1) do { if ((ROW) < 4) P3_HA(ROW); } while (0)
#define P_STEP(CURA, NA, NB, BC0, BC1, BC2, BC3, BN0, BN1, BN2, BN3, BOUNDARY, S1)               \
  do {                                                                                         \
    const uint32_t aa__ = (CURA), na__ = (NA), nb__ = (NB);                                    \
    V3_SB; V3_MFMA_ROW(0, al0, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah0, aa__, 8192); V3_RD1(ah1, aa__, 10240); P3_HAX(0, S1); \
    V3_SB; V3_MFMA_ROW(1, al1, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah2, aa__, 12288); V3_RD1(ah3, aa__, 14336); P3_HAX(1, S1); \
    V3_SB; V3_MFMA_ROW(2, al2, BC0, BC1, BC2, BC3); V3_SB; P3_HAX(2, S1);                      \
    V3_SB; V3_MFMA_ROW(3, al3, BC0, BC1, BC2, BC3); V3_SB; P3_HAX(3, S1);                      \
    V3_SB;                                                                                     \
    V3_PIN4(P3_LGKM_WAIT, ah0, ah1, ah2, ah3);                                                 \
    BOUNDARY();                                                                                \
    V3_SB; V3_MFMA_ROW(4, ah0, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al0, na__, 0); V3_RD1(BN0, nb__, 0); V3_RD1(al1, na__, 2048); P3_HB(1); P3_HAX(4, S1); \
    V3_SB; V3_MFMA_ROW(5, ah1, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(BN1, nb__, 2048); V3_RD1(al2, na__, 4096); V3_RD1(BN2, nb__, 4096); P3_HB(2); P3_HAX(5, S1); \
    V3_SB; V3_MFMA_ROW(6, ah2, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al3, na__, 6144); V3_RD1(BN3, nb__, 6144); P3_HB(3); P3_HAX(6, S1); \
    V3_SB; V3_MFMA_ROW(7, ah3, BC0, BC1, BC2, BC3);                                            \
    V3_SB;                                                                                     \
    V3_PIN8(P3_LGKM_WAIT, al0, al1, al2, al3, BN0, BN1, BN2, BN3);                             \
  } while (0)
#define P_BND_NONE() do { } while (0)
#define P_BND_MID()                                                                            \
  do {                                                                                         \
    P3_BOUNDARY_WAIT();                                                                        \
    const bool cb__ = st + 2 < nst, ca__ = st + 3 < nst;                                       \
    pb_g = reinterpret_cast<const char*>(B + (long)((cb__ ? n0 : n0n) + wave * 32) * ldb + XP_BK(cb__ ? st + 2 : st + 2 - nst)); \
    pb_slot = ldsB + (uint32_t)(bc * 32768 + wave * 4096);                                     \
    pa_g = reinterpret_cast<const char*>(A + (long)((ca__ ? m0 : m0n) + wave * 32) * lda + XP_AK(ca__ ? st + 3 : st + 3 - nst)); \
    pa_slot = lds0 + (uint32_t)(ac * 32768 + wave * 4096);                                     \
    pb_pend = pa_pend = true;                                                                  \
    P_DMA16(offB0, pb_g, pb_slot);                                                             \
  } while (0)
  // last boundary of the tile: the next tile's bias and this tile's first input chunk are requested BEFORE the next tile's
  // stage 1, so the epilogue can wait for them without waiting for that stage.  Issue order: bias x4, in x4, B(1) x4.
#define P_BND_LAST()                                                                           \
  do {                                                                                         \
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");                              \
    P_LANE(lb__);                                                                              \
    const uint32_t boff__ = (uint32_t)((lb__ >> 4) * 16);                                      \
    P_GLD4(bq0, boff__, bptr, 0); P_GLD4(bq1, boff__, bptr, 64); P_GLD4(bq2, boff__, bptr, 128); P_GLD4(bq3, boff__, bptr, 192); \
    if (HAS_IN) {                                                                              \
      const uint32_t io__ = (uint32_t)((lb__ & 15) * ldin + (lb__ >> 4) * 4) * 4;              \
      P_GLD4(rin[0][0], io__, ibase, 0); P_GLD4(rin[0][1], io__, ibase, 64); P_GLD4(rin[0][2], io__, ibase, 128); P_GLD4(rin[0][3], io__, ibase, 192); \
    }                                                                                          \
    pb_g = reinterpret_cast<const char*>(B + (long)(n0n + wave * 32) * ldb + XP_BK(1));        \
    pb_slot = ldsB + (uint32_t)(bc * 32768 + wave * 4096);                                     \
    pb_pend = true;                                                                            \
    P_DMA16(offB0, pb_g, pb_slot);                                                             \
  } while (0)

  // ---- RING boundaries.  Source of k block KB of the running stream: this tile while KB < nks, else the next tile's KB - nks.
#define R_SRC(PTR, LD, ROW_CUR, ROW_NXT, KB) \
  reinterpret_cast<const char*>((PTR) + (long)(((KB) < nks ? (ROW_CUR) : (ROW_NXT)) + wave * 32) * (LD) + (long)((KB) < nks ? (KB) : (KB) - nks) * 64)
#define R_ISSUE_X(SRC, SLOT) do { pb_g = (SRC); pb_slot = (SLOT) + (uint32_t)(wave * 4096); pb_pend = true; P_DMA16(offB0, pb_g, pb_slot); } while (0)
#define R_BND0() do { asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); R_ISSUE_X(R_SRC(B + b_ps, ldb, n0, n0n, kb + 1), ldsL); } while (0)
#define R_BND1() do { asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory"); R_ISSUE_X(R_SRC(B, ldb, n0, n0n, kb + 1), sH); } while (0)
#define R_BND2()                                                                               \
  do {                                                                                         \
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");                              \
    R_ISSUE_X(R_SRC(A + a_ps, lda, m0, m0n, kb + 1), sG);                                      \
    pa_g = R_SRC(A, lda, m0, m0n, kb + 2); pa_slot = sM + (uint32_t)(wave * 4096); pa_pend = true; \
  } while (0)
#define R_BND_LAST()                                                                           \
  do {                                                                                         \
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");                              \
    P_LANE(lb__);                                                                              \
    const uint32_t boff__ = (uint32_t)((lb__ >> 4) * 16);                                      \
    P_GLD4(bq0, boff__, bptr, 0); P_GLD4(bq1, boff__, bptr, 64); P_GLD4(bq2, boff__, bptr, 128); P_GLD4(bq3, boff__, bptr, 192); \
    if (HAS_IN) {                                                                              \
      const uint32_t io__ = (uint32_t)((lb__ & 15) * ldin + (lb__ >> 4) * 4) * 4;              \
      P_GLD4(rin[0][0], io__, ibase, 0); P_GLD4(rin[0][1], io__, ibase, 64); P_GLD4(rin[0][2], io__, ibase, 128); P_GLD4(rin[0][3], io__, ibase, 192); \
    }                                                                                          \
    R_ISSUE_X(R_SRC(A + a_ps, lda, m0, m0n, kb + 1), sG);       /* (Ahi(kb + 2) goes into the slot the epilogue borrows: after it) */ \
  } while (0)
  // one k block: three sub-stages of two k-steps; BND2 = R_BND2 (inside a tile) or R_BND_LAST
#define R_KBLOCK(BND2)                                                                         \
  do {                                                                                         \
    P_STEP(sH + oA0, sH + oA1, ldsL + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, P_BND_NONE, 1); \
    P_STEP(sH + oA1, sH + oA0, sG + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, R_BND0, 0);   \
    P_STEP(sH + oA0, sH + oA1, sG + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, P_BND_NONE, 1); \
    P_STEP(sH + oA1, sM + oA0, sG + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, R_BND1, 0);   \
    P_STEP(sM + oA0, sM + oA1, sG + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, P_BND_NONE, 1); \
    P_STEP(sM + oA1, sHn + oA0, ldsL + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, BND2, 0);  \
  } while (0)

  for (;;) {
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntiles;
    int m0n = m0, n0n = n0;
    if (has_next) { const int tn_ = xcd_remap(vn, ntiles); m0n = (tn_ / tiles_n) * 256; n0n = (tn_ % tiles_n) * 256; }
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    const float* bptr = bias ? bias + n0n + wc * 64 : reinterpret_cast<const float*>(A);   // uniform
    const char* ibase = !HAS_IN ? nullptr : reinterpret_cast<const char*>(in + (long)mw * ldin + nw);   // uniform

    f32x4 acc[8][4];
    {
      f32x4 bi[4] = {bq0, bq1, bq2, bq3};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) bi[j][e] = __uint_as_float(__float_as_uint(bi[j][e]) & bmask);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = bi[j];
    }
    bf16x8 al0, al1, al2, al3, ah0, ah1, ah2, ah3, bx0, bx1, bx2, bx3, by0, by1, by2, by3;
    f32x4 rin[2][4];                              // the epilogue's input chunks (accumulator layout), ping-pong
    {
      const uint32_t aa = RING ? R_SLOT(rr) + oA0 : lds0 + (uint32_t)(a0 * 32768) + oA0;
      const uint32_t ab = RING ? ldsL + oB0 : ldsB + (uint32_t)(b0 * 32768) + oB0;
      V3_READ4(al0, al1, al2, al3, aa, 0, 2048, 4096, 6144);
      V3_READ4(bx0, bx1, bx2, bx3, ab, 0, 2048, 4096, 6144);
      V3_PIN8("s_waitcnt lgkmcnt(0)", al0, al1, al2, al3, bx0, bx1, bx2, bx3);
    }
    int ac = a0, bc = b0;
    uint32_t ereg;                                 // this wave's slice of the A slot the tile's last (sub-)stage frees
    if constexpr (RING) {
      int kb = 0;
      for (; kb < nks - 1; ++kb) {
        const uint32_t sH = R_SLOT(rr), sG = R_SLOT(rr + 1), sM = R_SLOT(rr + 2), sHn = R_SLOT(rr + 3);
        R_KBLOCK(R_BND2);
        rr = (rr + 3) & 3;
      }
      {
        const uint32_t sH = R_SLOT(rr), sG = R_SLOT(rr + 1), sM = R_SLOT(rr + 2), sHn = R_SLOT(rr + 3);
        ereg = sM + (uint32_t)(wave * 4096);
        R_KBLOCK(R_BND_LAST);
        rr = (rr + 3) & 3;
      }
    } else {
      for (int st = 0; st < nst - 1; ++st) {
        const int an = ac == 2 ? 0 : ac + 1, bn = bc ^ 1;
        const uint32_t sa = lds0 + (uint32_t)(ac * 32768), sb = ldsB + (uint32_t)(bc * 32768);
        const uint32_t na = lds0 + (uint32_t)(an * 32768), nb = ldsB + (uint32_t)(bn * 32768);
        P_STEP(sa + oA0, sa + oA1, sb + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, P_BND_NONE, 1);
        P_STEP(sa + oA1, na + oA0, nb + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, P_BND_MID, 0);
        ac = an; bc = bn;
      }
      ereg = lds0 + (uint32_t)(ac * 32768 + wave * 4096);
      {
        const int an = ac == 2 ? 0 : ac + 1, bn = bc ^ 1;
        const uint32_t sa = lds0 + (uint32_t)(ac * 32768), sb = ldsB + (uint32_t)(bc * 32768);
        const uint32_t na = lds0 + (uint32_t)(an * 32768), nb = ldsB + (uint32_t)(bn * 32768);
        P_STEP(sa + oA0, sa + oA1, sb + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, P_BND_NONE, 1);
        P_STEP(sa + oA1, na + oA0, nb + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, P_BND_LAST, 0);
      }
    }

    // ---- epilogue: 8 chunks of 16 rows (= accumulator row block i) per wave, straight-line, no barrier.
    // Outstanding VMEM per wave here, oldest first: A_next(1) x4, bias x4, [in(0) x4], B_next(1) x4.
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(bq0), "+v"(bq1), "+v"(bq2), "+v"(bq3) : "n"(HAS_IN ? 8 : 4) : "memory");
    {
      P_LANE(le);
      const int fr = le & 15, fg = le >> 4, lr = le >> 3;          // shadow the kernel-scope copies on purpose
      const int sw = (fr >> 1) & 7;
      const int pc0 = (le & 7) ^ (le >> 4), pc1 = (le & 7) ^ (4 + (le >> 4));   // logical 16-B chunk this lane re-reads, rows lr / lr + 8
      const uint32_t wr0 = ereg + (uint32_t)(fr * 128 + ((fg ^ sw) << 4));       // f32 image: chunk (j & 1) * 4 + fg of row fr
      const uint32_t wr1 = ereg + (uint32_t)(fr * 128 + (((4 + fg) ^ sw) << 4));
      const uint32_t slot = ereg + (uint32_t)(fr * 128 + (fg & 1) * 8);         // 16-bit image (gemm_nt_p3_kernel's)
      const uint32_t rd = ereg + (uint32_t)(le * 16);
      const uint32_t io = HAS_IN ? (uint32_t)(fr * ldin + fg * 4) * 4 : 0;
      // f32 stores: byte offsets of the lane's two 16-B pieces of a 16 x 32 half chunk; 16-bit plane stores: of a 16 x 64 chunk
      const uint32_t eo0 = (uint32_t)(lr * ldc + pc0 * 4) * 4, eo1 = (uint32_t)((lr + 8) * ldc + pc1 * 4) * 4;
      const uint32_t po0 = (uint32_t)(lr * ldcp + pc0 * 8) * 2, po1 = (uint32_t)((lr + 8) * ldcp + pc1 * 8) * 2;
      constexpr int NS = 4;                         // stores per chunk of the HAS_IN forms (f32: 2 x 2, planes: 2 x 2)
      const bool want_c = EPI != SIMX_EPI_GELU || C != nullptr;     // (GELU with C == NULL: the inference form)
      float cs[4][4];                                // DGELU: this lane's column partial sums over its 8 rows
      if (EPI == SIMX_EPI_DGELU) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) cs[j][e] = 0.f;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (HAS_IN) {
          if (i < 7) {
            const char* ib = ibase + (long)(i + 1) * 16 * ldin * 4;
            P_GLD4(rin[(i + 1) & 1][0], io, ib, 0); P_GLD4(rin[(i + 1) & 1][1], io, ib, 64);
            P_GLD4(rin[(i + 1) & 1][2], io, ib, 128); P_GLD4(rin[(i + 1) & 1][3], io, ib, 192);
          }
          // younger than in(i): i == 0: B(1) x4 + in(1) x4; else the previous chunk's stores + in(i+1) x4
#define XP_PIN_IN(CNT) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(rin[i & 1][0]), "+v"(rin[i & 1][1]), "+v"(rin[i & 1][2]), "+v"(rin[i & 1][3]) : "n"(CNT) : "memory")
          if (i == 0) XP_PIN_IN(8);
          else if (i < 7) XP_PIN_IN(NS + 4);
          else XP_PIN_IN(NS);
#undef XP_PIN_IN
        }
        float vv[4][4], dd[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int e = 0; e < 4; ++e) vv[j][e] = acc[i][j][e];
          if (EPI == SIMX_EPI_NONE && DROP) {
            float m4[4];
            drop_mult4(drop, (uint32_t)(mw + i * 16 + fr), (uint32_t)(nw + j * 16 + fg * 4), m4);
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[j][e] *= m4[e];
          }
          if (HAS_IN) {
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[j][e] = EPI == SIMX_EPI_NONE ? vv[j][e] + rin[i & 1][j][e] : vv[j][e] * rin[i & 1][j][e];
          }
          if (EPI == SIMX_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { float g, dg; xp_gelu_both(vv[j][e], g, dg); vv[j][e] = g; dd[j][e] = dg; }
          }
        }
        if (EPI == SIMX_EPI_NONE || (EPI == SIMX_EPI_GELU && want_c)) {
          // f32 output: C (NONE) or gelu'(u) (GELU)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = EPI == SIMX_EPI_GELU ? dd[j][e] : vv[j][e];
            const uint32_t ad = ((j & 1) ? wr1 : wr0) + (uint32_t)((j >> 1) * 2048);
            asm volatile("ds_write_b128 %0, %1" ::"v"(ad), "v"(o) : "memory");
          }
          u32x4 w0, w1, w2, w3;
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                       "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(rd) : "memory");
          const char* ob = reinterpret_cast<const char*>(C + (long)(mw + i * 16) * ldc + nw);      // uniform
          P_GST4(eo0, ob, w0);
          P_GST4(eo1, ob, w1);
          P_GST4(eo0, ob + 128, w2);
          P_GST4(eo1, ob + 128, w3);
        }
        if (EPI == SIMX_EPI_DGELU) {
          const float keep = (mw + i * 16 + fr) < m_valid ? 1.0f : 0.0f;       // rows past the real tokens hold garbage
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) cs[j][e] += keep != 0.0f ? vv[j][e] : 0.0f;
        }
        if (EPI != SIMX_EPI_NONE) {
          // plane pair output: hi in the first 2 KB of the slice, lo in the second
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            uint32_t h0, l0, h1, l1;
            xp_split2<F>(vv[j][0], vv[j][1], h0, l0);
            xp_split2<F>(vv[j][2], vv[j][3], h1, l1);
            const uint32_t ad = slot + (uint32_t)(((j * 2 + (fg >> 1)) ^ sw) << 4);
            const uint2 oh = make_uint2(h0, h1), ol = make_uint2(l0, l1);
            asm volatile("ds_write_b64 %0, %1\n\tds_write_b64 %0, %2 offset:2048" ::"v"(ad), "v"(oh), "v"(ol) : "memory");
          }
          u32x4 w0, w1, w2, w3;
          asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\t"
                       "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3) : "v"(rd) : "memory");
          const char* pb = reinterpret_cast<const char*>(Cp + (long)(mw + i * 16) * ldcp + nw);    // uniform
          P_GST4(po0, pb, w0);
          P_GST4(po1, pb, w1);
          P_GST4(po0, pb + cp_ps * 2, w2);
          P_GST4(po1, pb + cp_ps * 2, w3);
        }
      }
      if (EPI == SIMX_EPI_DGELU) {
        if (colsum != nullptr) {                   // (uniform) 16-lane row sums by DPP, one atomic per column from the fr == 0 lanes
#pragma unroll
          for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float t = cs[j][e];
              t += dpp_mov<0xB1, 0xF>(t);
              t += dpp_mov<0x4E, 0xF>(t);
              t += dpp_mov<0x141, 0xF>(t);
              t += dpp_mov<0x140, 0xF>(t);
              if (fr == 0) atomicAdd(colsum + nw + j * 16 + fg * 4 + e, t);
            }
        }
      }
    }
    if constexpr (RING) {                          // Ahi of the next tile's k block 1 -> the slot the epilogue borrowed
      if (has_next) {
        pa_g = reinterpret_cast<const char*>(A + (long)(m0n + wave * 32) * lda + 64);
        pa_slot = ereg;
        pa_pend = true;
      } else {
        p3_half(A, lda, m0n, 64, ereg - (uint32_t)(wave * 4096), wave, offA0, offA1);
      }
    } else if (has_next) {
      pa_g = reinterpret_cast<const char*>(A + (long)(m0n + wave * 32) * lda + XP_AK(2));
      pa_slot = lds0 + (uint32_t)(ac * 32768 + wave * 4096);
      pa_pend = true;
    } else {
      p3_half(A, lda, m0n, XP_AK(2), lds0 + (uint32_t)(ac * 32768), wave, offA0, offA1);
    }
    if (!has_next) break;
    v = vn; m0 = m0n; n0 = n0n; a0 = ac == 2 ? 0 : ac + 1; b0 = bc ^ 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef P_LANE
#undef P3_HA
#undef P3_HAX
#undef P3_HB
#undef XP_AK
#undef XP_BK
#undef R_KBLOCK
#undef R_BND_LAST
#undef R_BND2
#undef R_BND1
#undef R_BND0
#undef R_ISSUE_X
#undef R_SRC
#undef R_SLOT
#undef P_BND_LAST
#undef P_BND_MID
#undef P_BND_NONE
#undef P_STEP
#undef V3_RD1
#undef V3_SB
#undef V3_MFMA_ROW
}

// ------------------------------------------------------------------------------------------ TN (wgrad)
// slab[split][M][N] = sum over the split's tokens of  Ahi^T.Blo + Alo^T.Bhi + Ahi^T.Bhi,  A = dY planes [K, M], B = X planes
// [K, N] (K = tokens, the row index of both).  gemm_tn2_kernel's loop (csrc/gemm.hip) with stream stage s = token block s / 3
// of term s % 3; the fused bias gradient (column sums of A) is taken from the fragments of the terms that stage Alo (1) and
// Ahi (2), so it is the column sum of hi + lo.  A ragged last token block is three ragged stages; each is zeroed past its
// valid rows after it has landed, as tn2 does for its one.
#define XS_KS(S) (((S) * 43691) >> 17)
#define XS_TERM(S) ((S) - 3 * XS_KS(S))
template <typename F>
__global__ __launch_bounds__(512, 2) void gemm_tn_xp_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, long a_ps, const bf16_t* __restrict__ B, int ldb, long b_ps,
    float* __restrict__ out, long slab_stride, int ldo, int tiles_n, int tiles_mn, int k_per_split, int accumulate,
    float* __restrict__ dbias, int dbias_parts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vb / tiles_mn;
  const int tile = vb % tiles_mn;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int wr = wave >> 2, wc = wave & 3;
  const int kb = split * k_per_split;
  const int ke = min(K, kb + k_per_split);
  const int fs = lane & 15, fg = lane >> 4;
  const int bias_slot = (tile % tiles_n) * 4 + wc, bias_mod = tiles_n * 4;
  bool do_bias = false;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nks = (ke - kb + 63) / 64;
  const int nst = 3 * nks;
  const bool ragged = (ke - kb) % 64 != 0;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int r_lo = 4 * fg + (fs >> 2);
  const int x_lo = r_lo & 7, x_hi = (r_lo + 16) & 7;
  const uint32_t row_lo = (uint32_t)(r_lo * 512 + (fs & 3) * 8), row_hi = (uint32_t)((r_lo + 16) * 512 + (fs & 3) * 8);
#define XS_A(S) (A + (XS_TERM(S) == 1 ? a_ps : 0L))
#define XS_B(S) (B + (XS_TERM(S) == 0 ? b_ps : 0L))
#define XS_K0(S) (kb + XS_KS(S) * 64)
#define XS_FULL(S) ((XS_KS(S) + 1) * 64 <= ke - kb)

  uint32_t oa[4], ob[4];
  tn2_lane_offsets(lda, m0, M, ldb, n0, N, lane, oa, ob, 0);
  tn2_stage(XS_A(0), lda, m0, M, XS_B(0), ldb, n0, N, kb, ke, smem, wave, lane, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  tn2_stage(XS_A(1), lda, m0, M, XS_B(1), ldb, n0, N, kb, ke, smem + TN2_STAGE, wave, lane, 0);     // (nst >= 3)

  bf16x4 al_lo[4], al_hi[4], ah_lo[4], ah_hi[4], bx_lo[4], bx_hi[4], by_lo[4], by_hi[4];
#define TN2_ADDR_LO(KBASE, OP, CT) ((KBASE) + (OP) + row_lo + (uint32_t)((((CT)) ^ x_lo) << 5))
#define TN2_ADDR_HI(KBASE, OP, CT) ((KBASE) + (OP) + row_hi + (uint32_t)((((CT)) ^ x_hi) << 5))
#define TN2_FRAG(LO, HI) ((bf16x8){LO[0], LO[1], LO[2], LO[3], HI[0], HI[1], HI[2], HI[3]})
#define TN2_MFMA_ROW(I, ALO, AHI, BLO, BHI)                                                                     \
  do {                                                                                                          \
    const bf16x8 af__ = TN2_FRAG(ALO, AHI);                                                                     \
    acc[I][0] = H16<F>::mfma(TN2_FRAG(BLO[0], BHI[0]), af__, acc[I][0]);    \
    acc[I][1] = H16<F>::mfma(TN2_FRAG(BLO[1], BHI[1]), af__, acc[I][1]);    \
    acc[I][2] = H16<F>::mfma(TN2_FRAG(BLO[2], BHI[2]), af__, acc[I][2]);    \
    acc[I][3] = H16<F>::mfma(TN2_FRAG(BLO[3], BHI[3]), af__, acc[I][3]);    \
    if (do_bias) {                                                                                              \
      _Pragma("unroll") for (int e__ = 0; e__ < 4; ++e__)                                                       \
          bsum[I] += H16<F>::one(ALO[e__]) + H16<F>::one(AHI[e__]);                                           \
    }                                                                                                           \
  } while (0)
#define TN2_SB __builtin_amdgcn_sched_barrier(0)
#define TN2_ONE_A(ST, J) tn2_stage_one(XS_A((ST) + 2), lda, XS_K0((ST) + 2), lds0 + (uint32_t)(((ST) & 1) * TN2_STAGE), wave, J, oa[J])
#define TN2_ONE_B(ST, J) tn2_stage_one(XS_B((ST) + 2), ldb, XS_K0((ST) + 2), lds0 + 32768u + (uint32_t)(((ST) & 1) * TN2_STAGE), wave, J, ob[J])
#define XS_ZERO_TAIL(BUF)                                                                                       \
  do {                                                                                                          \
    const int valid__ = (ke - kb) - (nks - 1) * 64;                                                             \
    char* sp__ = smem + (BUF) * TN2_STAGE;                                                                      \
    for (int idx = tid; idx < 64 * 64; idx += 512) {                                                            \
      const int kr = idx >> 6, c16 = idx & 63;                                                                  \
      if (kr >= valid__) *reinterpret_cast<uint4*>(sp__ + kr * 512 + (c16 & 31) * 16 + (c16 >> 5) * 32768) = make_uint4(0, 0, 0, 0); \
    }                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                             \
  } while (0)

  if (nks == 1 && ragged) XS_ZERO_TAIL(0);        // stage 0 itself is ragged: zero before anything is consumed
  {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      TN2_RD(al_lo[i], al_hi[i], TN2_ADDR_LO(lds0, 0u, wr * 8 + i), TN2_ADDR_HI(lds0, 0u, wr * 8 + i));
      TN2_RD(bx_lo[i], bx_hi[i], TN2_ADDR_LO(lds0, 32768u, wc * 4 + i), TN2_ADDR_HI(lds0, 32768u, wc * 4 + i));
    }
  }

  // one k-step (32 tokens): CUR = k-step base address, NXT = next k-step base address
#define TN2_STEP(CUR, NXT, BCL, BCH, BNL, BNH, SYNC, ST)                                                        \
  do {                                                                                                          \
    const uint32_t cur__ = (CUR), nxt__ = (NXT);                                                                \
    bool spread__ = false;                                                                                      \
    const bool tail__ = !(SYNC) && pend;       /* second half of the previous boundary's stage */               \
    TN2_SB; TN2_MFMA_ROW(0, al_lo[0], al_hi[0], BCL, BCH); TN2_SB;                                              \
    TN2_RD(ah_lo[0], ah_hi[0], TN2_ADDR_LO(cur__, 0u, wr * 8 + 4), TN2_ADDR_HI(cur__, 0u, wr * 8 + 4));         \
    TN2_RD(ah_lo[1], ah_hi[1], TN2_ADDR_LO(cur__, 0u, wr * 8 + 5), TN2_ADDR_HI(cur__, 0u, wr * 8 + 5));         \
    if (tail__) TN2_ONE_A((ST) - 1, 2);                                                                         \
    TN2_SB; TN2_MFMA_ROW(1, al_lo[1], al_hi[1], BCL, BCH); TN2_SB;                                              \
    TN2_RD(ah_lo[2], ah_hi[2], TN2_ADDR_LO(cur__, 0u, wr * 8 + 6), TN2_ADDR_HI(cur__, 0u, wr * 8 + 6));         \
    TN2_RD(ah_lo[3], ah_hi[3], TN2_ADDR_LO(cur__, 0u, wr * 8 + 7), TN2_ADDR_HI(cur__, 0u, wr * 8 + 7));         \
    if (tail__) TN2_ONE_B((ST) - 1, 2);                                                                         \
    TN2_SB; TN2_MFMA_ROW(2, al_lo[2], al_hi[2], BCL, BCH);                                                      \
    if (tail__) TN2_ONE_A((ST) - 1, 3);                                                                         \
    TN2_SB; TN2_MFMA_ROW(3, al_lo[3], al_hi[3], BCL, BCH);                                                      \
    if (tail__) { TN2_ONE_B((ST) - 1, 3); pend = false; }                                                       \
    TN2_SB;                                                                                                     \
    if (SYNC) {                                                                                                 \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                             \
      const int s2__ = (ST) + 2;                                                                                \
      spread__ = s2__ < nst && XS_FULL(s2__);                                                                   \
      if (spread__) { TN2_ONE_A(ST, 0); pend = true; }                                                          \
      else if (s2__ < nst)                                                                                      \
        tn2_stage(XS_A(s2__), lda, m0, M, XS_B(s2__), ldb, n0, N, XS_K0(s2__), ke, smem + ((ST) & 1) * TN2_STAGE, wave, lane, 0); \
      if (ragged && (ST) + 1 < nst && XS_KS((ST) + 1) == nks - 1) XS_ZERO_TAIL(((ST) + 1) & 1);                 \
    }                                                                                                           \
    TN2_SB; TN2_MFMA_ROW(4, ah_lo[0], ah_hi[0], BCL, BCH); TN2_SB;                                              \
    TN2_RD(al_lo[0], al_hi[0], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 0), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 0));         \
    TN2_RD(BNL[0], BNH[0], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 0), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 0));     \
    TN2_RD(al_lo[1], al_hi[1], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 1), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 1));         \
    if (SYNC && spread__) TN2_ONE_B(ST, 0);                                                                     \
    TN2_SB; TN2_MFMA_ROW(5, ah_lo[1], ah_hi[1], BCL, BCH); TN2_SB;                                              \
    TN2_RD(BNL[1], BNH[1], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 1), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 1));     \
    TN2_RD(al_lo[2], al_hi[2], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 2), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 2));         \
    TN2_RD(BNL[2], BNH[2], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 2), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 2));     \
    if (SYNC && spread__) TN2_ONE_A(ST, 1);                                                                     \
    TN2_SB; TN2_MFMA_ROW(6, ah_lo[2], ah_hi[2], BCL, BCH); TN2_SB;                                              \
    TN2_RD(al_lo[3], al_hi[3], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 3), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 3));         \
    TN2_RD(BNL[3], BNH[3], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 3), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 3));     \
    if (SYNC && spread__) TN2_ONE_B(ST, 1);                                                                     \
    TN2_SB; TN2_MFMA_ROW(7, ah_lo[3], ah_hi[3], BCL, BCH);                                                      \
    TN2_SB;                                                                                                     \
  } while (0)

  bool pend = false;
  for (int st = 0; st < nst; ++st) {
    const uint32_t sc = lds0 + (uint32_t)((st & 1) * TN2_STAGE), sn = lds0 + (uint32_t)(((st + 1) & 1) * TN2_STAGE);
    do_bias = dbias != nullptr && XS_TERM(st) != 0 && (XS_KS(st) % bias_mod) == bias_slot;
    TN2_STEP(sc, sc + 32 * 512, bx_lo, bx_hi, by_lo, by_hi, false, st);
    TN2_STEP(sc + 32 * 512, sn, by_lo, by_hi, bx_lo, bx_hi, true, st);
  }
#undef TN2_STEP

  float* o = out + (long)split * slab_stride;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + fs;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fg * 4;
      if (n >= N) continue;
      float4* dst = reinterpret_cast<float4*>(o + (long)m * ldo + n);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (accumulate) { float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
  if (dbias != nullptr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = bsum[i];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      const int m = m0 + wr * 128 + i * 16 + fs;
      if (fg == 0 && m < M) {
        if (dbias_parts) dbias[(long)(split * bias_mod + bias_slot) * M + m] = t;
        else atomicAdd(dbias + m, t);
      }
    }
  }
#undef XS_ZERO_TAIL
#undef TN2_ONE_A
#undef TN2_ONE_B
#undef TN2_SB
#undef TN2_MFMA_ROW
#undef TN2_FRAG
#undef TN2_ADDR_LO
#undef TN2_ADDR_HI
#undef XS_A
#undef XS_B
#undef XS_K0
#undef XS_FULL
}

// ------------------------------------------------------------------------------------------ TN (wgrad), four-plane stages
// The same product with HALF-DEPTH, FOUR-PLANE stages: a 64 KB stage holds 32 tokens of Ahi | Alo | Bhi | Blo (16 KB each;
// token-major rows are 512 B whatever the depth, so the LDS-DMA still moves whole lines) and feeds three k-steps,
// Ahi.Blo, Alo.Bhi, Ahi.Bhi: 96 MFMAs per wave and barrier instead of 64, and 64 KB of operand traffic per 96 MFMAs where the
// tripled-K form above stages 96 KB (every hi plane twice).  The Bhi fragments of the second k-step serve the third.
// LDS image of a stage: A region (32 KB) rows 0-31 = Ahi, 32-63 = Alo; B region (32 KB) rows 0-31 = Bhi, 32-63 = Blo;
// row kr at kr * 512 B, 32-B chunk q at q ^ (kr & 7).  Waves 0-3 stage the hi planes, waves 4-7 the lo planes.
// No fused bias gradient here: with the per-row `do_bias` branches and their conversions in the loop the kernel needs 40 more
// VGPRs than the 256 a wave of a 512-thread workgroup has and spills into scratch inside the loop (2.3x slower, measured);
// without them it fits.  The wgrad GEMMs that also produce a bias gradient (W1, Wqkv) stay on gemm_tn_xp_kernel.
template <typename F>
__global__ __launch_bounds__(512, 2) void gemm_tn_xq_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, long a_ps, const bf16_t* __restrict__ B, int ldb, long b_ps,
    float* __restrict__ out, long slab_stride, int ldo, int tiles_n, int tiles_mn, int k_per_split, int accumulate,
    float* __restrict__ dbias, int dbias_parts) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vb / tiles_mn;
  const int tile = vb % tiles_mn;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int wr = wave >> 2, wc = wave & 3;
  const int kb = split * k_per_split;
  const int ke = min(K, kb + k_per_split);
  const int fs = lane & 15, fg = lane >> 4;
  const int bias_slot = (tile % tiles_n) * 4 + wc, bias_mod = tiles_n * 4;
  bool do_bias = false;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nst = (ke - kb + 31) / 32;             // 32-token stages
  const int last_valid = (ke - kb) - (nst - 1) * 32;               // rows of the last stage (32 = not ragged)
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int r_lo = 4 * fg + (fs >> 2);
  const int x_lo = r_lo & 7, x_hi = (r_lo + 16) & 7;
  const uint32_t row_lo = (uint32_t)(r_lo * 512 + (fs & 3) * 8), row_hi = (uint32_t)((r_lo + 16) * 512 + (fs & 3) * 8);
  // this wave's share of a stage: plane (hi: waves 0-3, lo: 4-7), token rows (wave & 3) * 8 + 2 j + (lane >> 5)
  const bf16_t* Aw = A + (wave >= 4 ? a_ps : 0L);
  const bf16_t* Bw = B + (wave >= 4 ? b_ps : 0L);
  const int trow = (wave & 3) * 8;
  uint32_t oa[4], ob[4];
  tn2_lane_offsets(lda, m0, M, ldb, n0, N, lane, oa, ob, 0);
  // piece j of stage ST's operand (full stages): SGPR base carries the token row, the lane offset the column swizzle
#define XQ_ONE_A(ST, J) P_DMA16(oa[J], reinterpret_cast<const char*>(Aw + (long)(kb + (ST) * 32 + trow + 2 * (J)) * lda), \
                                lds0 + (uint32_t)(((ST) & 1) * TN2_STAGE + (wave * 4 + (J)) * 1024))
#define XQ_ONE_B(ST, J) P_DMA16(ob[J], reinterpret_cast<const char*>(Bw + (long)(kb + (ST) * 32 + trow + 2 * (J)) * ldb), \
                                lds0 + 32768u + (uint32_t)(((ST) & 1) * TN2_STAGE + (wave * 4 + (J)) * 1024))
  // a (possibly ragged) stage at once, token rows clamped into [0, valid): rows >= valid are zeroed after landing
#define XQ_CLAMPED_ONE(ST, VALID, J)                                                                            \
  do {                                                                                                          \
    int t__ = trow + 2 * (J) + (lane >> 5);                                                                     \
    t__ = t__ < (VALID) ? t__ : (VALID) - 1;                                                                    \
    const uint32_t ca__ = oa[J] - (uint32_t)((lane >> 5) * lda) * 2, cb__ = ob[J] - (uint32_t)((lane >> 5) * ldb) * 2; \
    P_DMA16(ca__ + (uint32_t)(t__ * lda) * 2, reinterpret_cast<const char*>(Aw + (long)(kb + (ST) * 32) * lda),  \
            lds0 + (uint32_t)(((ST) & 1) * TN2_STAGE + (wave * 4 + (J)) * 1024));                               \
    P_DMA16(cb__ + (uint32_t)(t__ * ldb) * 2, reinterpret_cast<const char*>(Bw + (long)(kb + (ST) * 32) * ldb),  \
            lds0 + 32768u + (uint32_t)(((ST) & 1) * TN2_STAGE + (wave * 4 + (J)) * 1024));                      \
  } while (0)
#define stage_clamped(ST, VALID) do { XQ_CLAMPED_ONE(ST, VALID, 0); XQ_CLAMPED_ONE(ST, VALID, 1); XQ_CLAMPED_ONE(ST, VALID, 2); XQ_CLAMPED_ONE(ST, VALID, 3); } while (0)
#define XQ_ZERO_TAIL(BUF)                                                                                       \
  do {                                                                                                          \
    char* sp__ = smem + (BUF) * TN2_STAGE;                                                                      \
    int tz__ = tid;                    /* laundered: the (rare) tail's addresses are computed here, not hoisted over the k loop */ \
    asm volatile("" : "+v"(tz__));                                                                              \
    for (int idx = tz__; idx < 128 * 32; idx += 512) {          /* 128 rows (4 planes x 32) x 32 16-B pieces */ \
      const int row = idx >> 5, c16 = idx & 31;                                                                 \
      if ((row & 31) >= last_valid) *reinterpret_cast<uint4*>(sp__ + row * 512 + c16 * 16) = make_uint4(0, 0, 0, 0); \
    }                                                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                             \
  } while (0)
#define XQ_VALID(ST) ((ST) == nst - 1 ? last_valid : 32)

  stage_clamped(0, XQ_VALID(0));
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (nst > 1) stage_clamped(1, XQ_VALID(1));
  if (nst == 1 && last_valid < 32) XQ_ZERO_TAIL(0);

  bf16x4 al_lo[4], al_hi[4], ah_lo[4], ah_hi[4], bx_lo[4], bx_hi[4], by_lo[4], by_hi[4];
#define XQ_ADDR_LO(TILE, CT) ((TILE) + row_lo + (uint32_t)((((CT)) ^ x_lo) << 5))
#define XQ_ADDR_HI(TILE, CT) ((TILE) + row_hi + (uint32_t)((((CT)) ^ x_hi) << 5))
#define TN2_FRAG(LO, HI) ((bf16x8){LO[0], LO[1], LO[2], LO[3], HI[0], HI[1], HI[2], HI[3]})
#define XQ_MFMA_ROW(I, ALO, AHI, BLO, BHI, BIAS)                                                                \
  do {                                                                                                          \
    const bf16x8 af__ = TN2_FRAG(ALO, AHI);                                                                     \
    acc[I][0] = H16<F>::mfma(TN2_FRAG(BLO[0], BHI[0]), af__, acc[I][0]);    \
    acc[I][1] = H16<F>::mfma(TN2_FRAG(BLO[1], BHI[1]), af__, acc[I][1]);    \
    acc[I][2] = H16<F>::mfma(TN2_FRAG(BLO[2], BHI[2]), af__, acc[I][2]);    \
    acc[I][3] = H16<F>::mfma(TN2_FRAG(BLO[3], BHI[3]), af__, acc[I][3]);    \
    if (false && (BIAS) && do_bias) {                                                                                    \
      _Pragma("unroll") for (int e__ = 0; e__ < 4; ++e__)                                                       \
          bsum[I] += H16<F>::one(ALO[e__]) + H16<F>::one(AHI[e__]);                                           \
    }                                                                                                           \
  } while (0)
#define XQ_SB __builtin_amdgcn_sched_barrier(0)
  {   // fragments of k-step 0 of stage 0: A rows 0-3 of Ahi, B = Blo
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      TN2_RD(al_lo[i], al_hi[i], XQ_ADDR_LO(lds0, wr * 8 + i), XQ_ADDR_HI(lds0, wr * 8 + i));
      TN2_RD(bx_lo[i], bx_hi[i], XQ_ADDR_LO(lds0 + 49152u, wc * 4 + i), XQ_ADDR_HI(lds0 + 49152u, wc * 4 + i));
    }
  }
  // one k-step (32 tokens of one term).  CURA: this k-step's A tile (its row fragments 4-7 are read here), NXTA / NXTB: the
  // next k-step's A and B tiles (A row fragments 0-3 and the B fragments are read here; RDB = 0: the next k-step keeps this
  // one's B fragments).  SYNC: the stage boundary sits in the middle of this k-step (every fragment of the stage's tiles
  // that is still needed has been issued before it).  PIECES: 1 = first half of the next-next stage's DMA pieces, 2 = second.
#define XQ_STEP(CURA, NXTA, NXTB, BCL, BCH, BNL, BNH, RDB, BIAS, SYNC, ST)                                      \
  do {                                                                                                          \
    const uint32_t cur__ = (CURA), nxa__ = (NXTA), nxb__ = (NXTB);                                              \
    bool spread__ = false;                                                                                      \
    const bool tail__ = !(SYNC) && pend && (BIAS) == 0;       /* second half of the previous boundary's stage (k-step 0 only) */ \
    XQ_SB; XQ_MFMA_ROW(0, al_lo[0], al_hi[0], BCL, BCH, BIAS); XQ_SB;                                           \
    TN2_RD(ah_lo[0], ah_hi[0], XQ_ADDR_LO(cur__, wr * 8 + 4), XQ_ADDR_HI(cur__, wr * 8 + 4));                   \
    TN2_RD(ah_lo[1], ah_hi[1], XQ_ADDR_LO(cur__, wr * 8 + 5), XQ_ADDR_HI(cur__, wr * 8 + 5));                   \
    if (tail__) XQ_ONE_A((ST) + 1, 2);                                                                          \
    XQ_SB; XQ_MFMA_ROW(1, al_lo[1], al_hi[1], BCL, BCH, BIAS); XQ_SB;                                           \
    TN2_RD(ah_lo[2], ah_hi[2], XQ_ADDR_LO(cur__, wr * 8 + 6), XQ_ADDR_HI(cur__, wr * 8 + 6));                   \
    TN2_RD(ah_lo[3], ah_hi[3], XQ_ADDR_LO(cur__, wr * 8 + 7), XQ_ADDR_HI(cur__, wr * 8 + 7));                   \
    if (tail__) XQ_ONE_B((ST) + 1, 2);                                                                          \
    XQ_SB; XQ_MFMA_ROW(2, al_lo[2], al_hi[2], BCL, BCH, BIAS);                                                  \
    if (tail__) XQ_ONE_A((ST) + 1, 3);                                                                          \
    XQ_SB; XQ_MFMA_ROW(3, al_lo[3], al_hi[3], BCL, BCH, BIAS);                                                  \
    if (tail__) { XQ_ONE_B((ST) + 1, 3); pend = false; }                                                        \
    XQ_SB;                                                                                                      \
    if (SYNC) {                                                                                                 \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                             \
      const int s2__ = (ST) + 2;                                                                                \
      spread__ = s2__ < nst && XQ_VALID(s2__) == 32;                                                            \
      if (spread__) { XQ_ONE_A(s2__, 0); pend = true; }                                                         \
      else if (s2__ < nst) stage_clamped(s2__, XQ_VALID(s2__));                                                 \
      if ((ST) + 1 == nst - 1 && last_valid < 32) XQ_ZERO_TAIL(((ST) + 1) & 1);                                 \
    }                                                                                                           \
    XQ_SB; XQ_MFMA_ROW(4, ah_lo[0], ah_hi[0], BCL, BCH, BIAS); XQ_SB;                                           \
    TN2_RD(al_lo[0], al_hi[0], XQ_ADDR_LO(nxa__, wr * 8 + 0), XQ_ADDR_HI(nxa__, wr * 8 + 0));                   \
    if (RDB) TN2_RD(BNL[0], BNH[0], XQ_ADDR_LO(nxb__, wc * 4 + 0), XQ_ADDR_HI(nxb__, wc * 4 + 0));              \
    TN2_RD(al_lo[1], al_hi[1], XQ_ADDR_LO(nxa__, wr * 8 + 1), XQ_ADDR_HI(nxa__, wr * 8 + 1));                   \
    if (SYNC && spread__) XQ_ONE_B((ST) + 2, 0);                                                                \
    XQ_SB; XQ_MFMA_ROW(5, ah_lo[1], ah_hi[1], BCL, BCH, BIAS); XQ_SB;                                           \
    if (RDB) TN2_RD(BNL[1], BNH[1], XQ_ADDR_LO(nxb__, wc * 4 + 1), XQ_ADDR_HI(nxb__, wc * 4 + 1));              \
    TN2_RD(al_lo[2], al_hi[2], XQ_ADDR_LO(nxa__, wr * 8 + 2), XQ_ADDR_HI(nxa__, wr * 8 + 2));                   \
    if (RDB) TN2_RD(BNL[2], BNH[2], XQ_ADDR_LO(nxb__, wc * 4 + 2), XQ_ADDR_HI(nxb__, wc * 4 + 2));              \
    if (SYNC && spread__) XQ_ONE_A((ST) + 2, 1);                                                                \
    XQ_SB; XQ_MFMA_ROW(6, ah_lo[2], ah_hi[2], BCL, BCH, BIAS); XQ_SB;                                           \
    TN2_RD(al_lo[3], al_hi[3], XQ_ADDR_LO(nxa__, wr * 8 + 3), XQ_ADDR_HI(nxa__, wr * 8 + 3));                   \
    if (RDB) TN2_RD(BNL[3], BNH[3], XQ_ADDR_LO(nxb__, wc * 4 + 3), XQ_ADDR_HI(nxb__, wc * 4 + 3));              \
    if (SYNC && spread__) XQ_ONE_B((ST) + 2, 1);                                                                \
    XQ_SB; XQ_MFMA_ROW(7, ah_lo[3], ah_hi[3], BCL, BCH, BIAS);                                                  \
    XQ_SB;                                                                                                      \
  } while (0)

  bool pend = false;
  for (int st = 0; st < nst; ++st) {
    const uint32_t sc = lds0 + (uint32_t)((st & 1) * TN2_STAGE), sn = lds0 + (uint32_t)(((st + 1) & 1) * TN2_STAGE);
    do_bias = dbias != nullptr && (st % bias_mod) == bias_slot;
    // Ahi.Blo (bx) ; Alo.Bhi (by) ; Ahi.Bhi (by, kept) -- the last k-step reads the next stage's Ahi rows 0-3 and Blo
    XQ_STEP(sc, sc + 16384u, sc + 32768u, bx_lo, bx_hi, by_lo, by_hi, 1, 0, false, st);
    XQ_STEP(sc + 16384u, sc, sc + 32768u, by_lo, by_hi, by_lo, by_hi, 0, 1, false, st);
    XQ_STEP(sc, sn, sn + 49152u, by_lo, by_hi, bx_lo, bx_hi, 1, 2, true, st);
  }
#undef XQ_STEP

  float* o = out + (long)split * slab_stride;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + fs;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fg * 4;
      if (n >= N) continue;
      float4* dst = reinterpret_cast<float4*>(o + (long)m * ldo + n);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (accumulate) { float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
  if (false && dbias != nullptr) {              // (this kernel carries no fused bias gradient: see the header comment)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = bsum[i];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      const int m = m0 + wr * 128 + i * 16 + fs;
      if (fg == 0 && m < M) {
        if (dbias_parts) dbias[(long)(split * bias_mod + bias_slot) * M + m] = t;
        else atomicAdd(dbias + m, t);
      }
    }
  }
#undef XQ_ZERO_TAIL
#undef stage_clamped
#undef XQ_CLAMPED_ONE
#undef XQ_VALID
#undef XQ_ONE_A
#undef XQ_ONE_B
#undef XQ_SB
#undef XQ_MFMA_ROW
#undef TN2_FRAG
#undef XQ_ADDR_LO
#undef XQ_ADDR_HI
}

__global__ __launch_bounds__(256) void xp_slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int splits, int M, int N,
                                                             float* __restrict__ C, int ldc, int accumulate) {
  const long total4 = (long)M * N / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    const int m = (int)(e / N), n = (int)(e % N);
    float4 s = make_float4(0, 0, 0, 0);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (long)k * slab_stride + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4* dst = reinterpret_cast<float4*>(C + (long)m * ldc + n);
    if (accumulate) { const float4 c = *dst; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
    *dst = s;
  }
}

// ------------------------------------------------------------------------------------------ plane producers of last resort
// f32 [rows, cols] -> plane pair (SRC = float), or plane pair in format SRC -> plane pair in format DST (the wgrad GEMM needs
// the bf16 planes of an activation the forward wrote as fp16 planes).  8 elements per thread: whole 16-B accesses on every
// stream; one wave covers 2 KB of a row (f32) -- full lines.
template <typename SRC, typename DST>
__global__ __launch_bounds__(256) void planes_kernel(int rows, int cols8, const void* __restrict__ src, int lds_, long src_ps,
                                                     bf16_t* __restrict__ dst, int ldd, long dst_ps) {
  const long total = (long)rows * cols8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / cols8), c = (int)(i % cols8) * 8;
    float v[8];
    if constexpr (std::is_same<SRC, float>::value) {
      const float4* p = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + (long)r * lds_ + c);
      const float4 a = p[0], b = p[1];
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
      const bf16_t* ph = reinterpret_cast<const bf16_t*>(src) + (long)r * lds_ + c;
      const uint4 h = *reinterpret_cast<const uint4*>(ph), l = *reinterpret_cast<const uint4*>(ph + src_ps);
      const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[2 * e] = H16<SRC>::lo(hw[e]) + H16<SRC>::lo(lw[e]);
        v[2 * e + 1] = H16<SRC>::hi(hw[e]) + H16<SRC>::hi(lw[e]);
      }
    }
    uint32_t ho[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xp_split2<DST>(v[2 * e], v[2 * e + 1], ho[e], lo[e]);
    bf16_t* pd = dst + (long)r * ldd + c;
    *reinterpret_cast<uint4*>(pd) = make_uint4(ho[0], ho[1], ho[2], ho[3]);
    *reinterpret_cast<uint4*>(pd + dst_ps) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
  }
}

// plane pair -> f32 (tests, and consumers that have no plane form)
template <typename SRC>
__global__ __launch_bounds__(256) void planes_join_kernel(int rows, int cols8, const bf16_t* __restrict__ src, int lds_, long src_ps,
                                                          float* __restrict__ dst, int ldd) {
  const long total = (long)rows * cols8;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int r = (int)(i / cols8), c = (int)(i % cols8) * 8;
    const bf16_t* ph = src + (long)r * lds_ + c;
    const uint4 h = *reinterpret_cast<const uint4*>(ph), l = *reinterpret_cast<const uint4*>(ph + src_ps);
    const uint32_t hw[4] = {h.x, h.y, h.z, h.w}, lw[4] = {l.x, l.y, l.z, l.w};
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[2 * e] = H16<SRC>::lo(hw[e]) + H16<SRC>::lo(lw[e]);
      v[2 * e + 1] = H16<SRC>::hi(hw[e]) + H16<SRC>::hi(lw[e]);
    }
    float4* pd = reinterpret_cast<float4*>(dst + (long)r * ldd + c);
    pd[0] = make_float4(v[0], v[1], v[2], v[3]);
    pd[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
}

// Weight planes, once per optimiser step, all dense weights of an encoder in one launch (blockIdx.x walks the jobs' 32 x 32
// tiles): W [rows, cols] f32 -> fp16 planes of W (forward operand), bf16 planes of W^T (dgrad operand), and W^T in f32 for
// the shapes that stay on gemm_x3.hip.
template <int DUMMY>
__global__ __launch_bounds__(256) void split_weight_group_kernel(SimxSplitGroup g) {
  int ji = 0;
  while (ji + 1 < g.n && (int)blockIdx.x >= g.job[ji].tile_end) ++ji;
  const SimxSplitJob j = g.job[ji];
  const int t = (int)blockIdx.x - (ji ? g.job[ji - 1].tile_end : 0), tc = (j.cols + 31) >> 5;
  __shared__ float tile[32][33];
  const int c0 = (t % tc) * 32, r0 = (t / tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const long ps = (long)j.rows * j.cols;
  bf16_t* oh = reinterpret_cast<bf16_t*>(j.planes_h);
  bf16_t* ot = reinterpret_cast<bf16_t*>(j.planesT_b);
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < j.rows && c < j.cols) {
      v = j.w[(long)r * j.cols + c];
      if (oh) {
        const f16_t h = (f16_t)v;
        const f16_t l = (f16_t)(v - (float)h);
        oh[(long)r * j.cols + c] = __builtin_bit_cast(bf16_t, h);
        oh[ps + (long)r * j.cols + c] = __builtin_bit_cast(bf16_t, l);
      }
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (r < j.rows && c < j.cols) {
      const float v = tile[tx][i];
      if (j.wT) j.wT[(long)c * j.rows + r] = v;
      if (ot) {
        const bf16_t h = f2bf(v);
        ot[(long)c * j.rows + r] = h;
        ot[ps + (long)c * j.rows + r] = f2bf(v - bf2f(h));
      }
    }
  }
}

// The same job list on 64 x 64 tiles with 16-byte accesses on every stream (rows % 64 == 0 and cols % 64 == 0: every dense
// weight of the BERT geometries).  The 32 x 32 kernel above issues 2-byte stores to four arrays and a 4-byte store to a fifth:
// 6.2 ms for the 1.36 GB of a BERT-base tower (0.22 TB/s).  Here a thread owns 8 consecutive columns of a row in the first
// phase (two float4 loads -> one uint4 store per fp16 plane) and 8 consecutive rows of a column in the second (eight LDS reads
// at stride 65 words, conflict-free: lanes of a wave cover 8 columns x 8 row groups -> 64 distinct banks; two float4 stores of
// W^T, one uint4 store per bf16 plane).
__global__ __launch_bounds__(256) void split_weight_group64_kernel(SimxSplitGroup g) {
  int ji = 0;
  while (ji + 1 < g.n && (int)blockIdx.x >= g.job[ji].tile_end) ++ji;
  const SimxSplitJob j = g.job[ji];
  const int t = (int)blockIdx.x - (ji ? g.job[ji - 1].tile_end : 0), tc = j.cols >> 6;
  __shared__ float tile[64][65];
  const int c0 = (t % tc) * 64, r0 = (t / tc) * 64;
  const long ps = (long)j.rows * j.cols;
  bf16_t* oh = reinterpret_cast<bf16_t*>(j.planes_h);
  bf16_t* ot = reinterpret_cast<bf16_t*>(j.planesT_b);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = (int)threadIdx.x + k * 256;            // 64 rows x 8 column groups
    const int r = idx >> 3, cg = (idx & 7) * 8;
    const long off = (long)(r0 + r) * j.cols + c0 + cg;
    const float4 a = *reinterpret_cast<const float4*>(j.w + off), b = *reinterpret_cast<const float4*>(j.w + off + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r][cg + e] = v[e];
    if (oh) {
      uint32_t hw[4], lw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) xp_split2<f16_t>(v[2 * e], v[2 * e + 1], hw[e], lw[e]);
      *reinterpret_cast<uint4*>(oh + off) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *reinterpret_cast<uint4*>(oh + ps + off) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = (int)threadIdx.x + k * 256;            // 64 columns x 8 row groups
    const int c = idx >> 3, rg = (idx & 7) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[rg + e][c];
    const long off = (long)(c0 + c) * j.rows + r0 + rg;
    if (j.wT) {
      *reinterpret_cast<float4*>(j.wT + off) = make_float4(v[0], v[1], v[2], v[3]);
      *reinterpret_cast<float4*>(j.wT + off + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
    if (ot) {
      uint32_t hw[4], lw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) xp_split2<bf16_t>(v[2 * e], v[2 * e + 1], hw[e], lw[e]);
      *reinterpret_cast<uint4*>(ot + off) = make_uint4(hw[0], hw[1], hw[2], hw[3]);
      *reinterpret_cast<uint4*>(ot + ps + off) = make_uint4(lw[0], lw[1], lw[2], lw[3]);
    }
  }
}

// ------------------------------------------------------------------------------------------ host
struct XpDevice { std::once_flag once; int ncu = 0; bool ok = false; };
static XpDevice g_xp_dev[SIMX_MAX_DEVICES];
static const XpDevice* xp_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIMX_MAX_DEVICES) return nullptr;
  XpDevice& g = g_xp_dev[dev];
  std::call_once(g.once, [&g, dev] {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
    g.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g.ncu -= g.ncu % 8;
    bool ok = true;
#define XP_ATTR(KERNEL, BYTES) \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)) == hipSuccess
    XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE, false, false, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE, false, false, true>), P_LDS);
    XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE, true, false, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE, true, false, true>), P_LDS);
    XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE, true, true, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE, true, true, true>), P_LDS);
    XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE_PLANES, false, false, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_NONE_PLANES, false, false, true>), P_LDS);
    XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_GELU, false, false, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<f16_t, SIMX_EPI_GELU, false, false, true>), P_LDS);
    XP_ATTR((gemm_nt_xp_kernel<bf16_t, SIMX_EPI_NONE, false, false, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<bf16_t, SIMX_EPI_NONE, false, false, true>), P_LDS);
    XP_ATTR((gemm_nt_xp_kernel<bf16_t, SIMX_EPI_NONE, true, false, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<bf16_t, SIMX_EPI_NONE, true, false, true>), P_LDS);
    XP_ATTR((gemm_nt_xp_kernel<bf16_t, SIMX_EPI_DGELU, true, false, false>), P_LDS); XP_ATTR((gemm_nt_xp_kernel<bf16_t, SIMX_EPI_DGELU, true, false, true>), P_LDS);
    XP_ATTR(gemm_tn_xp_kernel<bf16_t>, TN2_LDS);
    XP_ATTR(gemm_tn_xq_kernel<bf16_t>, TN2_LDS);
#undef XP_ATTR
    g.ok = ok;
  });
  return g.ok ? &g : nullptr;
}
static bool xal16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int simx_gemm_nt_planes_ok(int M, int N, int K) {
  return M > 0 && N > 0 && M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && K >= 128 && K <= 8192 ? 1 : 0;
}

extern "C" int simx_gemm_nt_planes(simx_stream_t stream, int fmt, int epilogue, int M, int N, int K, const void* A, int lda, long a_ps,
                                   const void* B, int ldb, long b_ps, float* C, int ldc, const float* bias, const float* in, int ldin,
                                   void* Cp, int ldcp, long cp_ps, const simx_dropout* dropd) {
  return simx_gemm_nt_planes_cs(stream, fmt, epilogue, M, N, K, A, lda, a_ps, B, ldb, b_ps, C, ldc, bias, in, ldin, Cp, ldcp, cp_ps, dropd, nullptr, M);
}
// SIMX_EPI_DGELU with colsum != NULL: colsum[N] += column sums of the output rows [0, rows_valid) (atomics; not used in the
// deterministic mode, where the wgrad GEMM's ordered bias pass stays)
extern "C" int simx_gemm_nt_planes_cs(simx_stream_t stream, int fmt, int epilogue, int M, int N, int K, const void* A, int lda, long a_ps,
                                      const void* B, int ldb, long b_ps, float* C, int ldc, const float* bias, const float* in, int ldin,
                                      void* Cp, int ldcp, long cp_ps, const simx_dropout* dropd, float* colsum, int rows_valid) {
  hipStream_t s = (hipStream_t)stream;
  const int m_valid = rows_valid;
  SIMX_REQUIRE(!colsum || epilogue == SIMX_EPI_DGELU, SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: colsum is a DGELU output");
  SIMX_PROF(SIMX_K_GEMM_NT_XP, s, 2.0 * M * N * K);
  SIMX_REQUIRE(fmt == SIMX_F16 || fmt == SIMX_BF16, SIMX_ERR_BAD_DTYPE, "gemm_nt_planes: fmt %d (the format of the operand planes)", fmt);
  SIMX_REQUIRE(simx_gemm_nt_planes_ok(M, N, K), SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: shape %d x %d x %d (needs M, N %% 256 == 0, K %% 64 == 0, K >= 128)", M, N, K);
  SIMX_REQUIRE(A && B && lda >= K && ldb >= K && lda % 8 == 0 && ldb % 8 == 0 && a_ps % 8 == 0 && b_ps % 8 == 0 && xal16(A) && xal16(B),
               SIMX_ERR_BAD_SHAPE, "gemm_nt_planes: operand planes must be 16-B aligned with leading dimensions %% 8 == 0");
  SIMX_REQUIRE(!bias || xal16(bias), SIMX_ERR_BAD_SHAPE, "gemm_nt_planes: bias not 16-B aligned");
  SIMX_REQUIRE(!in || (ldin >= N && ldin % 4 == 0 && xal16(in)), SIMX_ERR_BAD_SHAPE, "gemm_nt_planes: bad `in`");
  SIMX_REQUIRE(!dropd || (dropd->p >= 0.f && dropd->p < 1.f), SIMX_ERR_BAD_SHAPE, "gemm_nt_planes: dropout p must be in [0,1)");
  const bool c_ok = C && ldc >= N && ldc % 4 == 0 && xal16(C);
  const bool p_ok = Cp && ldcp >= N && ldcp % 8 == 0 && cp_ps % 8 == 0 && xal16(Cp);
  const XpDevice* gd = xp_device();
  SIMX_REQUIRE(gd != nullptr, SIMX_ERR_HIP, "gemm_nt_planes: cannot query / configure the current device");
  const int tn = N / 256, nt = (M / 256) * tn;
  const int ncu_c = simx_compute_cus(gd->ncu);
  const int grid = nt < ncu_c ? nt : ncu_c;
  const DropCtx drop = make_drop(epilogue == SIMX_EPI_NONE ? dropd : nullptr);
  // SIMX_NT_XP=k3 pins the tripled-K stream (six tile loads per k block; A/B measurements); the ring form needs lda == ldb
  static const bool k3 = [] { const char* e = getenv("SIMX_NT_XP"); return e && e[0] == 'k'; }();
  const bool ring = !k3 && lda == ldb;
#define LXP(FF, E, HI, DR) do { if (ring) hipLaunchKernelGGL((gemm_nt_xp_kernel<FF, E, HI, DR, true>), dim3(grid), dim3(512), P_LDS, s, M, N, K, (const bf16_t*)A, lda, a_ps, \
                                              (const bf16_t*)B, ldb, b_ps, C, ldc, bias, in, ldin, (bf16_t*)Cp, ldcp, cp_ps, tn, nt, drop, colsum, m_valid);    \
                                else hipLaunchKernelGGL((gemm_nt_xp_kernel<FF, E, HI, DR, false>), dim3(grid), dim3(512), P_LDS, s, M, N, K, (const bf16_t*)A, lda, a_ps, \
                                              (const bf16_t*)B, ldb, b_ps, C, ldc, bias, in, ldin, (bf16_t*)Cp, ldcp, cp_ps, tn, nt, drop, colsum, m_valid); } while (0)
  if (epilogue == SIMX_EPI_NONE) {
    SIMX_REQUIRE(c_ok, SIMX_ERR_BAD_SHAPE, "gemm_nt_planes: C must be a 16-B aligned f32 matrix");
    if (fmt == SIMX_F16) {
      if (drop.thr) { SIMX_REQUIRE(in, SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: dropout is built with a residual only"); LXP(f16_t, SIMX_EPI_NONE, true, true); }
      else if (in) LXP(f16_t, SIMX_EPI_NONE, true, false);
      else LXP(f16_t, SIMX_EPI_NONE, false, false);
    } else {
      SIMX_REQUIRE(!drop.thr, SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: dropout is a forward (fp16-plane) epilogue");
      if (in) LXP(bf16_t, SIMX_EPI_NONE, true, false);
      else LXP(bf16_t, SIMX_EPI_NONE, false, false);
    }
  } else if (epilogue == SIMX_EPI_GELU || epilogue == SIMX_EPI_GELU_INFER) {
    SIMX_REQUIRE(fmt == SIMX_F16 && p_ok && !in, SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: GELU is a forward epilogue (fp16 planes out, no `in`)");
    if (epilogue == SIMX_EPI_GELU_INFER) C = nullptr;
    SIMX_REQUIRE(!C || c_ok, SIMX_ERR_BAD_SHAPE, "gemm_nt_planes: C (the stored derivative) must be a 16-B aligned f32 matrix");
    if (!C) ldc = N;
    LXP(f16_t, SIMX_EPI_GELU, false, false);
  } else if (epilogue == SIMX_EPI_NONE_PLANES) {
    SIMX_REQUIRE(fmt == SIMX_F16 && p_ok && !in && !drop.thr, SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: NONE_PLANES is a forward epilogue (fp16 planes out, no `in`, no dropout)");
    LXP(f16_t, SIMX_EPI_NONE_PLANES, false, false);
  } else if (epilogue == SIMX_EPI_DGELU) {
    SIMX_REQUIRE(fmt == SIMX_BF16 && p_ok && in, SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: DGELU is a backward epilogue (bf16 planes out, `in` = the stored derivative)");
    LXP(bf16_t, SIMX_EPI_DGELU, true, false);
  } else {
    SIMX_REQUIRE(false, SIMX_ERR_UNSUPPORTED, "gemm_nt_planes: epilogue %d", epilogue);
  }
#undef LXP
  SIMX_CHECK_LAUNCH("gemm_nt_xp");
  return SIMX_OK;
}

static void xp_tn_plan(int M, int N, int K, int* splits, int* kps) {
  const int tiles = cdiv(M, 256) * cdiv(N, 256);
  const char* rounds_env = getenv("SIMX_TN_ROUNDS");       // (read per call, as tn_plan)
  const int round = simx_compute_cus(256);
  int sp = (rounds_env && rounds_env[0] == '2' ? 2 * round : round) / tiles;      // ONE whole round of the chip (one workgroup per CU), as tn_plan (csrc/gemm.hip): 32768 tokens 1.60 -> 1.48 ms
  const int max_s = cdiv(K, 512);
  if (sp > max_s) sp = max_s;
  if (sp < 1) sp = 1;
  const int k = cdiv(cdiv(K, sp), 64) * 64;
  *splits = cdiv(K, k);
  *kps = k;
}
extern "C" size_t simx_gemm_tn_planes_workspace_bytes(int M, int N, int K) {
  int sp, kps;
  xp_tn_plan(M, N, K, &sp, &kps);
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}
extern "C" int simx_gemm_tn_planes(simx_stream_t stream, int M, int N, int K, const void* A, int lda, long a_ps, const void* B, int ldb,
                                   long b_ps, float* C, int ldc, int accumulate, void* ws, size_t ws_bytes, float* dbias) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_GEMM_TN_XP, s, 2.0 * M * N * K);
  SIMX_REQUIRE(M >= 8 && N >= 8 && K > 0 && A && B && C, SIMX_ERR_BAD_SHAPE, "gemm_tn_planes: bad shape %d %d %d", M, N, K);
  SIMX_REQUIRE(M % 8 == 0 && N % 8 == 0 && lda >= M && ldb >= N && lda % 8 == 0 && ldb % 8 == 0 && a_ps % 8 == 0 && b_ps % 8 == 0 && ldc >= N &&
                   ldc % 4 == 0 && xal16(A) && xal16(B) && xal16(C),
               SIMX_ERR_BAD_SHAPE, "gemm_tn_planes: operands must be 16-B aligned with dimensions %% 8 == 0");
  const XpDevice* gd = xp_device();
  SIMX_REQUIRE(gd != nullptr, SIMX_ERR_HIP, "gemm_tn_planes: cannot query / configure the current device");
  int sp, kps;
  xp_tn_plan(M, N, K, &sp, &kps);
  const int t_n = cdiv(N, 256), t_mn = cdiv(M, 256) * t_n;
  float* dbias_out = dbias;
  const int nbp = sp * t_n * 4;
  if (dbias && simx_det()) {
    dbias = simx_det_ws(s, (size_t)nbp * M * sizeof(float));
    if (!dbias) return SIMX_ERR_WORKSPACE;
  }
  const int dparts = dbias != dbias_out;
  // SIMX_TN_XP=k3 pins the tripled-K kernel (A/B measurements); default: four-plane stages
  static const bool k3 = [] { const char* e = getenv("SIMX_TN_XP"); return e && e[0] == 'k'; }();
  auto kern = (k3 || dbias) ? gemm_tn_xp_kernel<bf16_t> : gemm_tn_xq_kernel<bf16_t>;
  if (sp == 1) {
    hipLaunchKernelGGL(kern, dim3(t_mn), dim3(512), TN2_LDS, s, M, N, K, (const bf16_t*)A, lda, a_ps, (const bf16_t*)B, ldb, b_ps,
                       C, 0L, ldc, t_n, t_mn, kps, accumulate, dbias, dparts);
    SIMX_CHECK_LAUNCH("gemm_tn_xp");
    if (dparts) return simx_det_reduce(s, dbias, (long)M, nbp, M, dbias_out, nullptr, nullptr, nullptr);
    return SIMX_OK;
  }
  const size_t need = (size_t)sp * M * N * sizeof(float);
  SIMX_REQUIRE(ws && ws_bytes >= need && xal16(ws), SIMX_ERR_WORKSPACE, "gemm_tn_planes: workspace %zu < %zu (or not 16-B aligned)", ws_bytes, need);
  hipLaunchKernelGGL(kern, dim3(t_mn * sp), dim3(512), TN2_LDS, s, M, N, K, (const bf16_t*)A, lda, a_ps, (const bf16_t*)B, ldb, b_ps,
                     (float*)ws, (long)M * N, N, t_n, t_mn, kps, 0, dbias, dparts);
  SIMX_CHECK_LAUNCH("gemm_tn_xp");
  if (dparts) { int rcd = simx_det_reduce(s, dbias, (long)M, nbp, M, dbias_out, nullptr, nullptr, nullptr); if (rcd) return rcd; }
  const long tot4 = (long)M * N / 4;
  int rb = (int)((tot4 + 255) / 256);
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(xp_slab_reduce_kernel, dim3(rb), dim3(256), 0, s, (const float*)ws, (long)M * N, sp, M, N, C, ldc, accumulate);
  SIMX_CHECK_LAUNCH("slab_reduce");
  return SIMX_OK;
}

// planes of a [rows, cols] tensor.  src_fmt: SIMX_F32 (src = f32 matrix, src_ps ignored) or the format of a source plane pair
extern "C" int simx_split_weight(simx_stream_t stream, const float* W, int rows, int cols, void* planes_f16, void* planesT_bf16, float* WT) {
  SimxSplitGroup g{};
  g.n = 1;
  g.job[0] = SimxSplitJob{W, planes_f16, planesT_bf16, WT, rows, cols, 0, 0};
  return simx_split_weight_group((hipStream_t)stream, &g);
}

extern "C" int simx_planes_from(simx_stream_t stream, int src_fmt, int dst_fmt, int rows, int cols, const void* src, int lds_, long src_ps,
                                void* dst, int ldd, long dst_ps) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_CAST, s, (double)rows * cols * 8);
  SIMX_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && src && dst && lds_ >= cols && ldd >= cols && ldd % 8 == 0 && dst_ps % 8 == 0 && xal16(dst) && xal16(src) &&
                   (src_fmt == SIMX_F32 ? lds_ % 4 == 0 : (lds_ % 8 == 0 && src_ps % 8 == 0)),
               SIMX_ERR_BAD_SHAPE, "planes_from: bad shape / alignment");
  SIMX_REQUIRE(dst_fmt == SIMX_F16 || dst_fmt == SIMX_BF16, SIMX_ERR_BAD_DTYPE, "planes_from: dst_fmt %d", dst_fmt);
  const long total = (long)rows * (cols / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
#define LPL(S_, D_) hipLaunchKernelGGL((planes_kernel<S_, D_>), dim3(blocks), dim3(256), 0, s, rows, cols / 8, src, lds_, src_ps, (bf16_t*)dst, ldd, dst_ps)
  if (src_fmt == SIMX_F32) { if (dst_fmt == SIMX_F16) LPL(float, f16_t); else LPL(float, bf16_t); }
  else if (src_fmt == SIMX_F16) { if (dst_fmt == SIMX_F16) LPL(f16_t, f16_t); else LPL(f16_t, bf16_t); }
  else if (src_fmt == SIMX_BF16) { if (dst_fmt == SIMX_F16) LPL(bf16_t, f16_t); else LPL(bf16_t, bf16_t); }
  else SIMX_REQUIRE(false, SIMX_ERR_BAD_DTYPE, "planes_from: src_fmt %d", src_fmt);
#undef LPL
  SIMX_CHECK_LAUNCH("planes_from");
  return SIMX_OK;
}
extern "C" int simx_planes_join(simx_stream_t stream, int src_fmt, int rows, int cols, const void* src, int lds_, long src_ps, float* dst, int ldd) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_CAST, s, (double)rows * cols * 8);
  SIMX_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && src && dst && lds_ >= cols && lds_ % 8 == 0 && src_ps % 8 == 0 && ldd >= cols && ldd % 4 == 0 &&
                   xal16(src) && xal16(dst), SIMX_ERR_BAD_SHAPE, "planes_join: bad shape / alignment");
  const long total = (long)rows * (cols / 8);
  int blocks = (int)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (src_fmt == SIMX_F16) hipLaunchKernelGGL(planes_join_kernel<f16_t>, dim3(blocks), dim3(256), 0, s, rows, cols / 8, (const bf16_t*)src, lds_, src_ps, dst, ldd);
  else if (src_fmt == SIMX_BF16) hipLaunchKernelGGL(planes_join_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, rows, cols / 8, (const bf16_t*)src, lds_, src_ps, dst, ldd);
  else SIMX_REQUIRE(false, SIMX_ERR_BAD_DTYPE, "planes_join: src_fmt %d", src_fmt);
  SIMX_CHECK_LAUNCH("planes_join");
  return SIMX_OK;
}

int simx_split_weight_group(hipStream_t s, const SimxSplitGroup* g) {
  SIMX_REQUIRE(g && g->n > 0 && g->n <= SIMX_SPLIT_GROUP_MAX, SIMX_ERR_BAD_SHAPE, "split_weight_group: bad job count");
  double bytes = 0;
  SimxSplitGroup gg = *g;
  int tiles = 0;
  bool wide = true;                               // every job on whole 64 x 64 tiles with 16-byte-aligned streams
  for (int i = 0; i < gg.n; ++i) {
    const SimxSplitJob& jb = gg.job[i];
    SIMX_REQUIRE(jb.rows > 0 && jb.cols > 0 && jb.w, SIMX_ERR_BAD_SHAPE, "split_weight_group: bad job %d", i);
    wide = wide && jb.rows % 64 == 0 && jb.cols % 64 == 0 && xal16(jb.w) && xal16(jb.planes_h) && xal16(jb.planesT_b) && xal16(jb.wT);
  }
  const int ts = wide ? 64 : 32;
  for (int i = 0; i < gg.n; ++i) {
    tiles += cdiv(gg.job[i].rows, ts) * cdiv(gg.job[i].cols, ts);
    gg.job[i].tile_end = tiles;
    bytes += (double)gg.job[i].rows * gg.job[i].cols * 16;
  }
  SIMX_PROF(SIMX_K_CAST, s, bytes);
  if (wide) hipLaunchKernelGGL(split_weight_group64_kernel, dim3(tiles), dim3(256), 0, s, gg);
  else hipLaunchKernelGGL(split_weight_group_kernel<0>, dim3(tiles), dim3(256), 0, s, gg);
  SIMX_CHECK_LAUNCH("split_weight_group");
  return SIMX_OK;
}
