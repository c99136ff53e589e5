// EXPERIMENT, NOT PART OF THE PRODUCT (not built into libsimx_hip.so).  Kept with its measurements because the question it
// answers -- "can the NT GEMM's epilogue be hidden by one wave per SIMD + stores deferred into the next tile's main loop?" --
// will come up again.  Outcome on MI355X (tools/kbench, M = 262144, QKV shape N = 2304, K = 768; p3 = the product kernel):
//   main loop only            : p3 0.840-0.865 ms (1243 TFLOP/s over the seven shapes), this kernel 0.855-0.865 (1265): the
//                               128x128 wave tile + compiler-scheduled MFMA stream + DMA pieces spread one per MFMA pair work;
//   with the deferred epilogue: p3 1.007 ms, this kernel 1.06-1.08 ms.  Breaking it down (build flags P4_EXP_*):
//                               16 deferred stores cost 0.07 ms (~250 cycles of wave time EACH, in the main loop or not),
//                               16 immediate stores 0.095 ms (~340 cycles each), packing the accumulators 0.05 ms.
//   A store straight from the MFMA accumulator layout covers 16 rows x 64 B (sixteen half-line requests); p3's LDS-transposed
//   stores cover 8 full 128-B lines.  The cost is per REQUEST, and the issuing wave -- alone on its SIMD -- is blocked while the
//   address/data path takes them, so deferring such stores hides nothing.  Also: hipcc's register allocator does not hold a
//   finished tile (128 VGPRs) beside 96 fragment registers; single-buffered fragments (72 VGPRs, the E-step / O-step scheme
//   below) brought the in-loop spills to one, at no loss of main-loop speed.
//   => the product keeps p3 (LDS-staged full-line stores).  What carried over: nothing yet; the E/O fragment scheme and the
//   spread DMA issue are candidates for a p3 successor.
//
// Persistent NT GEMM, ONE wave per SIMD, epilogue stores deferred into the next tile's main loop ("p4").
//
//   C[M,N] = A[M,K] . B[N,K]^T  (+bias) (+dropout) (+residual) | GELU (two outputs) | x GELU'(aux)
//   bf16 operands, f32 accumulate; full 256x256 tiles, K % 64 == 0, K >= 768 (>= 12 stages per tile).
//
// Why a second persistent kernel beside gemm_nt_bf16_p3_kernel (csrc/gemm.hip: 8 waves, 128x64 wave tiles, two waves per
// SIMD).  p3 leaves its epilogue exposed: both waves of a SIMD reach it together (~20 % of a K = 768 GEMM, 35 % with GELU)
// and at 245 VGPRs it has no registers to hold a finished tile.  Measurements that shaped this kernel (tools/kbench,
// M = 262144, main loop only unless noted; DESIGN.md section 5 has the table):
//   * p3 with ONE wave per SIMD doing half of each tile: 71 % of what the pair achieves together -- one wave can keep
//     most of a matrix pipe busy;
//   * four waves with 128x128 wave tiles (this kernel), compiler-scheduled, DMA issued as a burst after the barrier:
//     1174 TFLOP/s vs p3's 1240; with the 16 LDS-DMA instructions of a stage spread one per MFMA pair: 1265 (the burst
//     left the pipe idle for its whole issue time -- the wave is alone on its SIMD);
//   * the store rate of a lone CU is 55 B/clk (tools/store_bench); what an exposed epilogue costs is the wave's issue
//     time and the read-modify VALU work, not a bandwidth limit -- so the stores are moved to where issue slots are free.
//
// Structure
//   * 256 threads = 4 waves (2 x 2), wave tile 128 x 128 = 8 x 8 MFMA 16x16x32 blocks: 256 accumulator registers in
//     AGPRs, which leaves the architectural VGPRs for fragments (A single-buffered: a row block's registers are refilled
//     for the next k-step as soon as its MFMAs are issued; B double-buffered) and for ONE FINISHED TILE in bf16
//     (`out`: 32 units x 16 B per lane = 128 VGPRs).
//   * operand stream exactly as in p3: 64-deep stages HBM -> LDS by global_load_lds (SGPR base + per-lane offset),
//     XOR-swizzled 128-B rows, three 32 KB slots for A and two for B, running ACROSS tiles; one barrier per stage:
//       boundary of stage g (after the first half of its second k-step, every fragment of the stage in registers):
//       s_waitcnt vmcnt(8 + n)  ("all but the A(g+2) pieces and the n deferred VMEM instructions of this stage" = stage g+1
//       landed; vmcnt completes in order), s_barrier, then B(g+2), A(g+3) one piece per MFMA pair.
//   * B fragment j takes, for MFMA column slot n, the weight row 32*(j>>1) + 8*(n>>2) + (n&3) + 4*(j&1): a lane then owns the
//     8 consecutive output columns 32p + 8fg .. +7 of row 16i + fr in blocks 2p, 2p+1 -> one 16-byte store (and one
//     16-byte input load) per unit (i, p).  B has its own LDS swizzle for that access pattern (see offBq).
//   * a tile's life:  stages 0-3 (GELU: 0-7) of tile t also DRAIN tile t-1 (8 (4) units per stage, one after each
//     (second) row block of MFMAs: global_store_dwordx4 from `out`; GELU computes gelu(u) of the packed pre-activation there);
//     the last four stages PREFETCH tile t's input operand (residual / GELU input) into the units just drained;
//     FINISH (exposed, short): accumulators -> (x dropout) (+ input) (x gelu'(input)) -> bf16 -> `out`.
//     The first tile of a workgroup has nothing to drain (plain stage bodies), the last one is drained after the loop.
#include <type_traits>
#include "common.h"
#include "prof.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define P4_DMA16(VOFF, SBASE, LDSADDR) \
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSADDR) : "memory")
// (the trailing s_nop covers the gfx9 "VMEM store of > 64 bits, then a write of its data VGPRs" hazard, see P_GST4 in gemm.hip)
#define P4_GST4(VOFF, SBASE, VAL) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 5" ::"v"(VOFF), "v"(VAL), "s"(SBASE) : "memory")
#ifdef P4_EXP_NODRAIN      /* timing experiments only (wrong results) */
#define P4_GST4_DRAIN(VOFF, SBASE, VAL) asm volatile("" ::"v"(VOFF), "v"(VAL), "s"(SBASE) : "memory")
#else
#define P4_GST4_DRAIN P4_GST4
#endif
#ifdef P4_EXP_NOIMM
#define P4_GST4_IMM(VOFF, SBASE, VAL) asm volatile("" ::"v"(VOFF), "v"(VAL), "s"(SBASE) : "memory")
#else
#define P4_GST4_IMM P4_GST4
#endif
#define P4_GLD4(DST, VOFF, SBASE) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(DST) : "v"(VOFF), "s"(SBASE) : "memory")
#define P4_SB __builtin_amdgcn_sched_barrier(0)
// MFMA as inline asm with the accumulator TIED in an AGPR: with the builtin hipcc renames accumulators between the stage
// bodies (v_mfma a[64:67], .., .., a[60:63]) and moves them through VGPRs (v_accvgpr_read inside the main loop), which
// costs the ~100 registers the finished tile needs.  (Same-accumulator back-to-back MFMAs need no wait states; the
// operands' s_waitcnt are still inserted by the compiler, which tracks registers through asm operands.)
#define P4_MFMA(ACC, BF, AF) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(BF), "v"(AF))
#define P4_IC(X) std::integral_constant<int, (X)>{}
// lane id recomputed where a block needs it (volatile: neither hoisted nor kept live across the main loop -- a copy of the
// kernel-scope `lane` was being spilled and re-loaded, each reload with an s_waitcnt vmcnt(0) that drains the DMA stream)
#define P4_LANE_ID(L) asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(L))

__device__ __forceinline__ int p4_xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

struct P4Cursor { int k, st, m0, n0; };      // a position in the stage stream: tile number k of this workgroup, stage st

template <int EPI, bool HAS_IN>
__global__ __launch_bounds__(256) void gemm_nt_bf16_p4_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    bf16_t* __restrict__ C, int ldc, const float* __restrict__ bias, const bf16_t* __restrict__ in, int ldin,
    bf16_t* __restrict__ C2, int ldc2, int tiles_n, int ntiles, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int nst = K / 64;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ldsB = lds0 + 98304u;
  const int lr = lane >> 3;
  const int ec0 = ((lane & 7) ^ (lane >> 4)) << 3, ec1 = ((lane & 7) ^ (4 + (lane >> 4))) << 3;
  const uint32_t offA0 = (uint32_t)(lr * lda + ec0) * 2, offA1 = (uint32_t)(lr * lda + ec1) * 2;
  // B uses its own LDS swizzle f(r) = bit1(r) | bits3-4(r) << 1 (chunk c of row r stored at c ^ f(r)): the B fragments are
  // read with PERMUTED rows (below), and this f keeps those reads -- and the 8-row DMA writes -- bank-conflict free.
  // DMA piece i covers rows i*8 + lr: f = ((lr >> 1) & 1) | ((i & 3) << 1)  ->  four per-lane source offsets
  uint32_t offBq[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) offBq[q] = (uint32_t)(lr * ldb + (((lane & 7) ^ (((lr >> 1) & 1) | (q << 1))) << 3)) * 2;
  // fragment byte offsets inside a slot: A row (wr*128 + i*16 + fr) at row*128, 16-B chunk (ks*4 + fg) ^ ((fr >> 1) & 7)
  const int swz = (fr >> 1) & 7;
  const uint32_t rowA = (uint32_t)((wr * 128 + fr) * 128);
  const uint32_t ch0 = (uint32_t)(((0 + fg) ^ swz) << 4), ch1 = (uint32_t)(((4 + fg) ^ swz) << 4);
  const int swzB = ((fr >> 1) & 1) | ((fr >> 2) << 1);
  const uint32_t rowB = (uint32_t)((wc * 128 + 8 * (fr >> 2) + (fr & 3)) * 128);
  const uint32_t chB0 = (uint32_t)(((0 + fg) ^ swzB) << 4), chB1 = (uint32_t)(((4 + fg) ^ swzB) << 4);
#define P4_BOFF(J) ((((J) >> 1) * 32 + ((J) & 1) * 4) * 128)          /* byte offset of B fragment J from fragment 0 */

  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto tile_of = [&](int k, int& m0, int& n0) {
    const int kk = k < my_tiles ? k : my_tiles - 1;            // past the end: re-fetch the last tile (harmless)
    const int t = p4_xcd_remap((int)blockIdx.x + kk * (int)gridDim.x, ntiles);
    m0 = (t / tiles_n) * 256; n0 = (t % tiles_n) * 256;
  };
  auto advance = [&](P4Cursor& c) {
    if (++c.st == nst) { c.st = 0; ++c.k; tile_of(c.k, c.m0, c.n0); }
  };
  P4Cursor cb, ca;                 // next stage to REQUEST for B / for A
  cb.k = 0; cb.st = 0; tile_of(0, cb.m0, cb.n0);
  ca = cb;
  int gb = 0, ga = 0;              // their stream indices (slot = index % 2 / % 3)
  // one DMA instruction (8 rows) of an operand stage: piece j of this wave's 8; the piece bases step by 8 rows
  auto piece = [&](const bf16_t* __restrict__ G, int ld, int row0, int k0, uint32_t slot, uint32_t off0, uint32_t off1, int j) {
    const char* sg = reinterpret_cast<const char*>(G + (long)(row0 + wave * 64) * ld + k0) + (long)j * 16 * ld;
    P4_DMA16((j & 1) ? off1 : off0, sg, slot + (uint32_t)((wave * 8 + j) * 1024));
  };
  auto piece_b = [&](int j) { piece(B, ldb, cb.n0, cb.st * 64, ldsB + (uint32_t)((gb & 1) * 32768), offBq[j & 2], offBq[(j & 2) + 1], j); };
  auto piece_a = [&](int j) { piece(A, lda, ca.m0, ca.st * 64, lds0 + (uint32_t)((ga % 3) * 32768), offA0, offA1, j); };
  auto done_b = [&]() { ++gb; advance(cb); };
  auto done_a = [&]() { ++ga; advance(ca); };
#pragma unroll
  for (int j = 0; j < 8; ++j) piece_b(j);
  done_b();
#pragma unroll
  for (int j = 0; j < 8; ++j) piece_a(j);
  done_a();                                                     // stage 0
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int j = 0; j < 8; ++j) piece_b(j);
  done_b();
#pragma unroll
  for (int j = 0; j < 8; ++j) piece_a(j);
  done_a();
#pragma unroll
  for (int j = 0; j < 8; ++j) piece_a(j);
  done_a();                                                     // B(1) A(1) A(2)

  // Fragments, single-buffered (18 x 4 VGPRs).  A stage is two k-steps of opposite loop order:
  //   E-step (k-step 0): A RESIDENT in a[0..7], B STREAMS -- column block j uses b[j] (j = 0: the spare bx) against all 8 row
  //                      blocks, then b[j] is refilled with k-step 1's B fragment j (resident there);
  //   O-step (k-step 1): B RESIDENT in b[0..7], A STREAMS -- row block i uses a[i] (i = 0: the spare ax), then a[i] is refilled
  //                      with the NEXT stage's k-step-0 A fragment i (resident there).
  // The streaming operand's fragments 1..7 can only be read when the previous step has released their registers (that step
  // held them resident), i.e. at the step's start: fragment 0 therefore comes from a spare loaded one step ahead, and
  // fragment 1 is needed 8 MFMAs (~130 cycles, an LDS latency) later.
  bf16x8 a[8], b[8], ax, bx;
  auto a_addr = [&](int g, int ks) { return smem + (g % 3) * 32768 + rowA + (ks ? ch1 : ch0); };
  auto b_addr = [&](int g, int ks) { return smem + 98304 + (g & 1) * 32768 + rowB + (ks ? chB1 : chB0); };
  {
    const char* sa = a_addr(0, 0);
    const char* sb = b_addr(0, 0);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa + i * 2048);
    bx = *reinterpret_cast<const bf16x8*>(sb);
  }

  // the finished tile waiting to be stored: unit q = i*4 + p holds row 16i + fr, columns 32p + 8fg .. +7 (8 bf16 = 16 B)
  // P4_NDEF of a tile's 32 units are deferred; the rest (the last row blocks) are stored by FINISH itself: 128 + 96 fragment
  // registers + addressing do not fit 256 VGPRs (the allocator spills hundreds of registers at P4_NDEF = 32), and 8 stores
  // fired at the tile boundary cost ~1 % of a K = 768 tile
#ifndef P4_NDEF
#define P4_NDEF 16
#endif
  u32x4 out[P4_NDEF];
  int pm0 = 0, pn0 = 0;            // its tile coordinates
  const bool gelu_train = EPI == SIMX_EPI_GELU && ldin != 1;    // (ldin == 1 on a GELU launch: inference, no pre-activation stored)
  const bool do_epi = C != nullptr;                             // (nullptr: measurement hook SIMX_NOEPI, main loop only)

  int g = 0;
  for (int k = 0; k < my_tiles; ++k) {
    int m0, n0;
    tile_of(k, m0, n0);
    f32x4 acc[8][8];
    {
      int lb; P4_LANE_ID(lb);      // (laundered: per-lane constants of this block must not stay live in the loop)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n0 + wc * 128 + (j >> 1) * 32 + (lb >> 4) * 8 + (j & 1) * 4);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i][j] = bv;
      }
    }
    // wave-uniform corners of the tile being drained (the previous one) and of this tile's input operand
    bf16_t* const dbase = C + (long)(pm0 + wr * 128) * ldc + pn0 + wc * 128;
    bf16_t* const dbase2 = EPI == SIMX_EPI_GELU ? C2 + (long)(pm0 + wr * 128) * ldc2 + pn0 + wc * 128 : nullptr;

    // ---- deferred work of one unit (macros, not lambdas: inline-asm operands cannot name captured variables), placed
    // between MFMA row blocks.  Unit Q = i*4 + p.
#define P4_DRAIN_UNIT(Q)                                                                                                  \
    {                                                                                                                     \
      int le; P4_LANE_ID(le);                                                                         \
      if (EPI == SIMX_EPI_GELU) {                                                                                         \
        if (gelu_train) {                                                                                                 \
          const uint32_t eo = (uint32_t)((le & 15) * ldc + (le >> 4) * 8) * 2;                                            \
          P4_GST4_DRAIN(eo, dbase + (long)((Q) >> 2) * 16 * ldc + ((Q) & 3) * 32, out[Q]);                                      \
        }                                                                                                                 \
        const uint32_t eo2 = (uint32_t)((le & 15) * ldc2 + (le >> 4) * 8) * 2;                                            \
        u32x4 h;                       /* gelu of the bf16-ROUNDED pre-activation: exactly what p3 writes */               \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                                   \
          const float x0 = __uint_as_float(out[Q][e] << 16), x1 = __uint_as_float(out[Q][e] & 0xFFFF0000u);               \
          h[e] = pack2bf(gelu_fast(x0), gelu_fast(x1));                                                                   \
        }                                                                                                                 \
        P4_GST4_DRAIN(eo2, dbase2 + (long)((Q) >> 2) * 16 * ldc2 + ((Q) & 3) * 32, h);                                          \
      } else {                                                                                                            \
        const uint32_t eo = (uint32_t)((le & 15) * ldc + (le >> 4) * 8) * 2;                                              \
        P4_GST4_DRAIN(eo, dbase + (long)((Q) >> 2) * 16 * ldc + ((Q) & 3) * 32, out[Q]);                                        \
      }                                                                                                                   \
    }

    // E-step of stage g.  UNIT_STMT runs after every column block j (`i` there is the block index: the deferred work)
#define P4_KSTEP0(UNIT_STMT)                                                                                              \
    {                                                                                                                     \
      const char* sb0 = b_addr(g, 0);                                                                                     \
      const char* sa1 = a_addr(g, 1);                                                                                     \
      const char* sb1 = b_addr(g, 1);                                                                                     \
      _Pragma("unroll") for (int j = 1; j < 8; ++j) b[j] = *reinterpret_cast<const bf16x8*>(sb0 + P4_BOFF(j));            \
      ax = *reinterpret_cast<const bf16x8*>(sa1);                 /* the O-step's first streaming fragment */              \
      P4_SB;                                                                                                              \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) {             /* (i = column block here) */                             \
        _Pragma("unroll") for (int r = 0; r < 8; ++r) {                                                                   \
          if (i == 0) { P4_MFMA(acc[r][0], bx, a[r]); } else { P4_MFMA(acc[r][i], b[i], a[r]); }                         \
        }                                                                                                                 \
        P4_SB;                                                                                                            \
        b[i] = *reinterpret_cast<const bf16x8*>(sb1 + P4_BOFF(i));                                                        \
        P4_SB;                                                                                                            \
        UNIT_STMT                                                                                                         \
      }                                                                                                                   \
    }
    // O-step of stage g with the stage boundary after its second row block; NV = deferred VMEM instructions of this stage
#define P4_REST(NV)                                                                                                       \
    {                                                                                                                     \
      const char* sa1 = a_addr(g, 1);                                                                                     \
      _Pragma("unroll") for (int i = 1; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(sa1 + i * 2048);              \
      P4_SB;                                                                                                              \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) P4_MFMA(acc[0][j], b[j], ax);                                         \
      _Pragma("unroll") for (int j = 0; j < 8; ++j) P4_MFMA(acc[1][j], b[j], a[1]);                                       \
      /* every fragment of stage g is in registers (lgkmcnt(0)); all but the 8 A(g+2) pieces and this stage's NV deferred    \
         VMEM instructions have completed = stage g+1 has landed */                                                         \
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(8 + (NV)) : "memory");                             \
      const char* sa = a_addr(g + 1, 0);                        /* (past the end: reads a slot nobody needs) */             \
      const char* sb = b_addr(g + 1, 0);                                                                                  \
      a[0] = *reinterpret_cast<const bf16x8*>(sa);                                                                        \
      a[1] = *reinterpret_cast<const bf16x8*>(sa + 2048);                                                                 \
      bx = *reinterpret_cast<const bf16x8*>(sb);                                                                          \
      P4_SB;                                                                                                              \
      _Pragma("unroll") for (int i = 2; i < 8; ++i) {                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                                \
          P4_MFMA(acc[i][2 * jj], b[2 * jj], a[i]);                                                                       \
          P4_MFMA(acc[i][2 * jj + 1], b[2 * jj + 1], a[i]);                                                               \
          P4_SB;                                                                                                          \
          const int q = (i - 2) * 4 + jj;                       /* B(g+2) THEN A(g+3) (the counted wait relies on the order),  \
                                                                   one piece per MFMA pair of row blocks 2-5 */                \
          if (q < 8) piece_b(q); else if (q < 16) piece_a(q - 8);                                                         \
          P4_SB;                                                                                                          \
        }                                                                                                                 \
        a[i] = *reinterpret_cast<const bf16x8*>(sa + i * 2048);                                                           \
        P4_SB;                                                                                                            \
      }                                                                                                                   \
      done_b(); done_a();                                                                                                 \
      ++g;                                                                                                                \
    }
    // eight units of one kind, unit S*8 + i after row block i
#define P4_UNITS8(FN, S)                                                                                                  \
        P4_SB;                                                                                                            \
        if (i == 0) FN((S) * 8 + 0) if (i == 1) FN((S) * 8 + 1) if (i == 2) FN((S) * 8 + 2) if (i == 3) FN((S) * 8 + 3)   \
        if (i == 4) FN((S) * 8 + 4) if (i == 5) FN((S) * 8 + 5) if (i == 6) FN((S) * 8 + 6) if (i == 7) FN((S) * 8 + 7)   \
        P4_SB;
    // four units, unit S*4 + i/2 after every second row block (gelu of a unit is ~100 VALU instructions: spread thinner)
#define P4_UNITS4(FN, S)                                                                                                  \
        P4_SB;                                                                                                            \
        if (i == 1) FN((S) * 4 + 0) if (i == 3) FN((S) * 4 + 1) if (i == 5) FN((S) * 4 + 2) if (i == 7) FN((S) * 4 + 3)   \
        P4_SB;

    // NV of the tile's FIRST stage also counts the stores FINISH fired for the previous tile's last units
#define P4_IMM ((32 - P4_NDEF) * (EPI == SIMX_EPI_GELU ? 2 : 1))
#define P4_IMM_INFER (32 - P4_NDEF)
    int st = 0;
    if (k > 0 && do_epi) {
      if (EPI == SIMX_EPI_GELU) {
        if (gelu_train) {
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 0)) P4_REST(8 + P4_IMM) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 1)) P4_REST(8)
#if P4_NDEF >= 16
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 2)) P4_REST(8) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 3)) P4_REST(8)
#endif
#if P4_NDEF >= 24
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 4)) P4_REST(8) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 5)) P4_REST(8)
#endif
#if P4_NDEF >= 32
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 6)) P4_REST(8) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 7)) P4_REST(8)
#endif
        } else {
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 0)) P4_REST(4 + P4_IMM_INFER) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 1)) P4_REST(4)
#if P4_NDEF >= 16
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 2)) P4_REST(4) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 3)) P4_REST(4)
#endif
#if P4_NDEF >= 24
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 4)) P4_REST(4) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 5)) P4_REST(4)
#endif
#if P4_NDEF >= 32
          P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 6)) P4_REST(4) P4_KSTEP0(P4_UNITS4(P4_DRAIN_UNIT, 7)) P4_REST(4)
#endif
        }
        st = P4_NDEF / 4;
      } else {
        P4_KSTEP0(P4_UNITS8(P4_DRAIN_UNIT, 0)) P4_REST(8 + P4_IMM)
#if P4_NDEF >= 16
        P4_KSTEP0(P4_UNITS8(P4_DRAIN_UNIT, 1)) P4_REST(8)
#endif
#if P4_NDEF >= 24
        P4_KSTEP0(P4_UNITS8(P4_DRAIN_UNIT, 2)) P4_REST(8)
#endif
#if P4_NDEF >= 32
        P4_KSTEP0(P4_UNITS8(P4_DRAIN_UNIT, 3)) P4_REST(8)
#endif
        st = P4_NDEF / 8;
      }
    }
    for (; st < nst; ++st) {
      P4_KSTEP0(;) P4_REST(0)
    }

    // ---- FINISH: accumulators -> bf16 units (the only exposed part of the epilogue).  Units >= P4_NDEF (the last row
    // blocks) first: they are stored at once and fly while the others are packed.
    if (do_epi) {
      int le; P4_LANE_ID(le);
      const int frl = le & 15, fgl = le >> 4;
      const int mw = m0 + wr * 128, nw = n0 + wc * 128;
      bf16_t* const cbase = C + (long)mw * ldc + nw;
      bf16_t* const cbase2 = EPI == SIMX_EPI_GELU ? C2 + (long)mw * ldc2 + nw : nullptr;
      const uint32_t eo = (uint32_t)(frl * ldc + fgl * 8) * 2;
      const uint32_t eo2 = EPI == SIMX_EPI_GELU ? (uint32_t)(frl * ldc2 + fgl * 8) * 2 : 0u;
#define P4_FIN(WITH_DROP) \
_Pragma("unroll") \
        for (int qq = 0; qq < 32; ++qq) { \
          const int q = qq < 32 - P4_NDEF ? P4_NDEF + qq : qq - (32 - P4_NDEF); \
          const int i = q >> 2, p = q & 3; \
          P4_SB; \
          float v[8] = {acc[i][2 * p][0], acc[i][2 * p][1], acc[i][2 * p][2], acc[i][2 * p][3], \
                        acc[i][2 * p + 1][0], acc[i][2 * p + 1][1], acc[i][2 * p + 1][2], acc[i][2 * p + 1][3]}; \
          if (WITH_DROP) { \
            float m4[4]; \
            drop_mult4(drop, (uint32_t)(mw + i * 16 + frl), (uint32_t)(nw + p * 32 + fgl * 8), m4); \
            v[0] *= m4[0]; v[1] *= m4[1]; v[2] *= m4[2]; v[3] *= m4[3]; \
            drop_mult4(drop, (uint32_t)(mw + i * 16 + frl), (uint32_t)(nw + p * 32 + fgl * 8 + 4), m4); \
            v[4] *= m4[0]; v[5] *= m4[1]; v[6] *= m4[2]; v[7] *= m4[3]; \
          } \
          const u32x4 o = (u32x4){pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])}; \
          if (q < P4_NDEF) { \
            out[q < P4_NDEF ? q : 0] = o; \
          } else { \
            if (EPI != SIMX_EPI_GELU || gelu_train) P4_GST4_IMM(eo, cbase + (long)i * 16 * ldc + p * 32, o); \
            if (EPI == SIMX_EPI_GELU) { \
              u32x4 h; \
_Pragma("unroll") \
              for (int e = 0; e < 4; ++e) { \
                const float x0 = __uint_as_float(o[e] << 16), x1 = __uint_as_float(o[e] & 0xFFFF0000u); \
                h[e] = pack2bf(gelu_fast(x0), gelu_fast(x1)); \
              } \
              P4_GST4_IMM(eo2, cbase2 + (long)i * 16 * ldc2 + p * 32, h); \
            } \
          } \
        }
      if (EPI == SIMX_EPI_NONE && drop.thr) { P4_FIN(true) } else { P4_FIN(false) }
    }
    pm0 = m0; pn0 = n0;
  }

  // ---- the last tile: drained in the open
  if (do_epi) {
    bf16_t* const dbase = C + (long)(pm0 + wr * 128) * ldc + pn0 + wc * 128;
    bf16_t* const dbase2 = EPI == SIMX_EPI_GELU ? C2 + (long)(pm0 + wr * 128) * ldc2 + pn0 + wc * 128 : nullptr;
    int le; P4_LANE_ID(le);
    const uint32_t eo = (uint32_t)((le & 15) * ldc + (le >> 4) * 8) * 2;
    const uint32_t eo2 = EPI == SIMX_EPI_GELU ? (uint32_t)((le & 15) * ldc2 + (le >> 4) * 8) * 2 : 0u;
#pragma unroll
    for (int q = 0; q < P4_NDEF; ++q) {
      const int i = q >> 2, p = q & 3;
      if (EPI != SIMX_EPI_GELU || gelu_train) P4_GST4(eo, dbase + (long)i * 16 * ldc + p * 32, out[q]);
      if (EPI == SIMX_EPI_GELU) {
        u32x4 h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float x0 = __uint_as_float(out[q][e] << 16), x1 = __uint_as_float(out[q][e] & 0xFFFF0000u);
          h[e] = pack2bf(gelu_fast(x0), gelu_fast(x1));
        }
        P4_GST4(eo2, dbase2 + (long)i * 16 * ldc2 + p * 32, h);
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores out, trailing (dummy) stage loads landed before the LDS is released
}

#define P4_LDS (5 * 32768)
int simx_launch_nt_p4(hipStream_t s, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                      const float* bias, const void* in, int ldin, void* C2, int ldc2, int epilogue, int has_in, DropCtx drop, int ncu) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_p4_kernel<SIMX_EPI_NONE, false>), hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_bf16_p4_kernel<SIMX_EPI_GELU, false>), hipFuncAttributeMaxDynamicSharedMemorySize, P4_LDS);
    attr = true;
  }
  const int t_n = N / 256, ntiles = (M / 256) * t_n;
  const int grid = ntiles < ncu ? ntiles : ncu;
#define LP4(E, HI) hipLaunchKernelGGL((gemm_nt_bf16_p4_kernel<E, HI>), dim3(grid), dim3(256), P4_LDS, s, M, N, K, (const bf16_t*)A, lda, \
                                      (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, (const bf16_t*)in, ldin, (bf16_t*)C2, ldc2, t_n, ntiles, drop)
  SIMX_REQUIRE(!has_in && epilogue != SIMX_EPI_DGELU, SIMX_ERR_UNSUPPORTED, "gemm_nt p4: epilogues with an input operand stay on the p3 kernel");
  if (epilogue == SIMX_EPI_NONE) LP4(SIMX_EPI_NONE, false);
  else LP4(SIMX_EPI_GELU, false);
#undef LP4
  SIMX_CHECK_LAUNCH("gemm_nt_bf16_p4");
  return SIMX_OK;
}
