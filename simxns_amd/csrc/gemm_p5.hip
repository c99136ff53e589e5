// Persistent NT GEMM with the epilogue of tile t hidden under the main loop of tile t + 1 ("p5"), gfx950.
//
//   C[M,N] = A[M,K] . B[N,K]^T + bias     16-bit operands (fp16 / bf16), f32 accumulate, 16-bit output
//   (the dense projections of LEAD/modeling_bert.py:229-232, 440-466 behind HFBertEncoder.forward, SimANS/model/models.py:77-82)
//
// Why a second persistent kernel beside gemm_nt_p3_kernel (csrc/gemm.hip).  p3's main loop runs at the chip's power limit, but
// its epilogue is exposed: the two waves of a SIMD reach it together (tools/p3_timeline.py: 3.3k cycles of a 35k-cycle K = 768
// tile when plain, 8-11k with an input tensor, 15-20k for the GELU pair) and at 237 of 256 registers it has nowhere to park a
// finished tile.  Round 2's gemm_p4 (tools/experiments/README.md) showed that ONE wave per SIMD with 128 x 128 wave tiles and
// the accumulators in AGPRs sustains the same main loop (1265 vs 1243 TFLOP/s); it lost because it stored straight from the
// accumulator layout (sixteen half-line requests per store, 250-340 cycles of wave time each).  This kernel keeps that register
// layout and sends the parked tile through LDS instead:
//
//   * 256 threads = 4 waves (2 x 2), one per SIMD, wave tile 128 x 128 = 8 x 8 blocks of v_mfma_f32_16x16x32: 256 accumulator
//     registers in AGPRs (inline-asm MFMA with a tied "+a" operand), which leaves the 256 architectural VGPRs for 18 fragment
//     quads (p4's E-step / O-step scheme: the streaming operand's register is refilled with the NEXT k-step's resident fragment
//     as soon as its MFMAs are issued) and for the FINISHED TILE packed to 16 bits (`pk`: 8 B per lane and block; seven of the
//     eight row blocks = 112 VGPRs stay parked, row block 0 leaves in the k-step that rounds it).  229 VGPRs, no spill.
//   * operand stream exactly as p3: 64-deep stages HBM -> LDS by global_load_lds, XOR-swizzled 128-B rows, three 32 KB A slots +
//     two B slots = all 160 KB, running ACROSS tiles, one barrier per stage (in the O-step, after the second row block: every
//     fragment of the stage is in registers by then):  s_waitcnt vmcnt(8)  ("all but the eight A(g+2) pieces" = stage g+1 has
//     landed: vmcnt completes in order), s_barrier, then B(g+2) x 8 and A(g+3) x 8.  A request keeps one base pointer; a piece is
//     two instructions (its lane offset lives in one of eight VGPRs shared by A and B: lda == ldb; M0 carries the LDS address).
//   * THE ISSUE RULE.  A wave issues at most one instruction per four cycles and is alone on its SIMD; an MFMA 16x16x32 keeps the
//     pipe busy 16 cycles, so ~3 other instructions fit behind each one for free and every further one is matrix-pipe idle time.
//     Every fragment read, DMA piece, drain instruction and address computation therefore sits in a GAP behind one particular
//     MFMA (P5_ECOLG / P5_OROWG take eight gap statements), never as a block between groups (profiles/r06_experiments/01_p5.md:
//     the same instructions as blocks cost 8 %, one instruction too many on the 16 DMA gaps of a stage 2.4 %).
//   * a tile's life:
//       stages 0..6       : also DRAIN tile t-1, one 16-row unit per stage (units 1..7): right after the stage barrier the wave's
//                           own 8 KB slice of the A slot that barrier freed is dead until the wave's own share of A(g+3) overwrites
//                           it (p3's epilogue borrows the same bytes), so the unit goes pk -> 8 x ds_write_b64 (swizzled) ->
//                           4 x ds_read_b128 (16 B per lane = full 128-B lines) -> 4 x global_store_dwordx4 (nt), in gaps, and
//                           the A pieces are issued behind the stores;
//       stage nst-2       : requests the NEXT tile's bias columns (two dwords per lane, ahead of the A pieces: the vmcnt(8)
//                           rule at the next barrier covers them);
//       O-step of stage nst-1 : writes those columns as a 512-B table into the last KB of the slice; row block i is final after
//                           its eight MFMAs -> in the gaps of row block i + 1 each of its blocks is read from the AGPRs, rounded
//                           into pk and RE-INITIALISED for the next tile by one ds_read_b128 of its bias quad straight into the
//                           AGPRs (an MFMA's srcC and vdst share one AGPR / VGPR select bit on gfx90a+, so "vdst in AGPRs, srcC = a
//                           VGPR bias quad" does not encode; 4 x v_accvgpr_write per block is issue time).  Row block 0 is drained
//                           here.  These ~7 instructions per gap (instead of 3) and the last row block's rounding are the only
//                           exposed epilogue work: ~1.3k cycles per tile against p3's 3.3k.
//     The last tile's parked row blocks are drained in the open.
//   Results are bit-identical to gemm_nt_p3_kernel (same k order, bias as the initial accumulator value, one rounding).
//
// Built in: EPI_NONE with bias, no epilogue input, no dropout; row-major C or plane-blocked C (HMC: the head-major q / k / v
// output of the QKV projection); K >= 576, lda == ldb, full 256 x 256 tiles.  Measured (profiles/r06_experiments/01_p5.md): 4 % faster
// than p3 at K = 3072 (0.877 vs 0.914 ms at M = 262144, 1410 TFLOP/s), a tie at K = 768 -- the dispatcher (csrc/gemm.hip) routes
// plain, dropout-free launches with K >= 1536 here.  Why the GELU pair and the input-tensor epilogues are not here: DESIGN.md 5.
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "p3.h"

#define P5_LDS (5 * 32768)
#define P5_SB __builtin_amdgcn_sched_barrier(0)

// MFMA as inline asm with the accumulator TIED in an AGPR (with the builtin hipcc renames accumulators between the unrolled
// stage bodies and moves them through VGPRs, which costs the registers the parked tile needs).  volatile: program order among
// the MFMAs, the LDS-DMA pieces, the stores and the barriers is the schedule.
template <typename F>
__device__ __forceinline__ void p5_mfma(f32x4& acc, const bf16x8& bw, const bf16x8& ax) {
  if constexpr (std::is_same<F, f16_t>::value) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(bw), "v"(ax));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(bw), "v"(ax));
}
// lane id recomputed where a block needs it (never hoisted, never live across the main loop)
#define P5_LANE(L) int L; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(L))


template <typename F, bool HMC>
__global__ __launch_bounds__(256) void gemm_nt_p5_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    bf16_t* __restrict__ C, int ldc, const float* __restrict__ bias, int hmR, int tiles_n, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int nst = K / 64;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ldsB = lds0 + 98304u;
  // LDS-DMA lane offsets: an instruction covers 8 rows x 128 B; lane -> row lr, 16-B chunk (lane & 7) ^ swizzle(row).  One offset
  // register PER PIECE (rows wave*64 + 8j + lr): the request's base pointer stays put and a piece costs two instructions -- an MFMA
  // gap holds about three.  lda == ldb here (both operands are [rows, K] with K contiguous), so A and B share the eight registers.
  uint32_t offA[8];
  {
    const int lr = lane >> 3;
    const int ec0 = ((lane & 7) ^ (lane >> 4)) << 3, ec1 = ((lane & 7) ^ (4 + (lane >> 4))) << 3;
#pragma unroll
    for (int j = 0; j < 8; ++j) offA[j] = (uint32_t)((lr + 8 * j) * lda + ((j & 1) ? ec1 : ec0)) * 2;
  }
#define offB offA
  // fragment byte offsets inside a slot: row (w*128 + i*16 + fr) at row*128, 16-B chunk (ks*4 + fg) ^ ((fr >> 1) & 7): ONE lane
  // constant fx (k-step 0, rows of the first wave); k-step 1 is ^ 64, the wave's rows and the slot ride in a uniform offset
  const uint32_t fx = (uint32_t)((lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) << 4));
  const uint32_t uA = (uint32_t)(wr * 16384), uB = (uint32_t)(98304 + wc * 16384);

  const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto tile_of = [&](int k, int& m0, int& n0) {
    const int kk = k < my_tiles ? k : my_tiles - 1;            // past the end: re-fetch the last tile (harmless)
    const int t = xcd_remap((int)blockIdx.x + kk * (int)gridDim.x, ntiles);
    m0 = (t / tiles_n) * 256; n0 = (t % tiles_n) * 256;
  };
  // The request stream.  B(g+2) and A(g+3) are requested behind the barrier of stage g; a request = 8 pieces from one base pointer
  // (piece 0 of the stage: row wave*64 of the tile, k offset of the stage) into this wave's 8 KB slice of the slot.  M0 carries the
  // running LDS address through the pieces of a request (nothing else in this kernel uses M0).
  // One wave issues at most one instruction per four cycles, and with ONE wave per SIMD nothing else fills the slots: a piece is
  // two instructions, the pointer / slot bookkeeping of a request sits in otherwise empty gaps (P5_DONE_*), the tile change (once
  // per nst requests) is the only long path.
  const char* pB; const char* pA;          // base pointer of the next B / A request
  int stB = 0, stA = 0, kB = 0, kA = 0;    // its stage within the tile, its tile number
  uint32_t rB = ldsB + (uint32_t)(wave * 8192), rA = lds0 + (uint32_t)(wave * 8192);     // this wave's slice of the slot it goes to
  const uint32_t rBsum = 2u * rB + 32768u, oBsum = 2u * uB + 32768u;                      // (the two B slots: other = sum - this)
  {
    int m0_, n0_;
    tile_of(0, m0_, n0_);
    pB = reinterpret_cast<const char*>(B + (long)(n0_ + wave * 64) * ldb);
    pA = reinterpret_cast<const char*>(A + (long)(m0_ + wave * 64) * lda);
  }
#define P5_REQ_B() asm volatile("s_mov_b32 m0, %0" ::"s"(rB) : "memory")
#define P5_REQ_A() asm volatile("s_mov_b32 m0, %0" ::"s"(rA) : "memory")
#define P5_PIECE_B(J) asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x400" ::"v"(offB[J]), "s"(pB) : "memory", "scc")
#define P5_PIECE_A(J) asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x400" ::"v"(offA[J]), "s"(pA) : "memory", "scc")
#define P5_DONE_B()                                                                                                            \
  do {                                                                                                                         \
    pB += 128; rB = rBsum - rB;                                                                                                    \
    if (++stB == nst) { stB = 0; ++kB; int m_, n_; tile_of(kB, m_, n_); pB = reinterpret_cast<const char*>(B + (long)(n_ + wave * 64) * ldb); } \
  } while (0)
#define P5_DONE_A()                                                                                                            \
  do {                                                                                                                         \
    pA += 128; rA = rA + 32768u >= ldsB ? rA - 65536u : rA + 32768u;                                                           \
    if (++stA == nst) { stA = 0; ++kA; int m_, n_; tile_of(kA, m_, n_); pA = reinterpret_cast<const char*>(A + (long)(m_ + wave * 64) * lda); } \
  } while (0)
#define P5_ALL8(M_) do { M_(0); M_(1); M_(2); M_(3); M_(4); M_(5); M_(6); M_(7); } while (0)

  // the wave's 128 bias columns in two VGPRs (lane l: columns l and 64 + l of the wave tile); tile 0's here, tile k+1's in stage
  // nst-2 of tile k
  float vb0 = 0.f, vb1 = 0.f;
  P5_REQ_B(); P5_ALL8(P5_PIECE_B); P5_DONE_B();
  P5_REQ_A(); P5_ALL8(P5_PIECE_A); P5_DONE_A();                 // stage 0
  if (bias) {
    int m0_, n0_;
    tile_of(0, m0_, n0_);
    const float* bp = bias + n0_ + wc * 128;
    const uint32_t bo = (uint32_t)lane * 4u;
    asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:256" : "=&v"(vb0), "=&v"(vb1) : "v"(bo), "s"(bp) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" : "+v"(vb0), "+v"(vb1)::"memory");
  P5_REQ_B(); P5_ALL8(P5_PIECE_B); P5_DONE_B();
  P5_REQ_A(); P5_ALL8(P5_PIECE_A); P5_DONE_A();
  P5_REQ_A(); P5_PIECE_A(0); P5_PIECE_A(1); P5_PIECE_A(2);      // B(1) A(1) A(2): pieces 3..7 of an A request are always issued by the next E-step
                                                                // from here on: behind the barrier of stage g, B(g+2) then A(g+3)

  // Fragments, single-buffered (18 quads).  A stage is two k-steps of opposite loop order:
  //   E-step (k-step 0): A RESIDENT in a[0..7], B STREAMS -- column block j uses b[j] (j = 0: the spare bx) against all 8 row
  //                      blocks, then b[j] is refilled with k-step 1's B fragment j (resident there);
  //   O-step (k-step 1): B RESIDENT in b[0..7], A STREAMS -- row block i uses a[i] (i = 0: the spare ax), then a[i] is refilled
  //                      with the NEXT stage's k-step-0 A fragment i (resident there).
  bf16x8 a[8], b[8], ax, bx;
  uint32_t oA = uA, oB = uB;       // byte offsets of the slots of the stage being CONSUMED (A: 3 slots, B: 2), wave rows included
  uint32_t va0 = fx + oA, vq0 = fx + oB;   // k-step-0 fragment addresses of that stage (A, B): carried from the O-step that computed them
#define P5_LDF(PTR) (*reinterpret_cast<const bf16x8*>(PTR))
  {
    const char* sa = smem + va0;
    const char* sb = smem + vq0;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = P5_LDF(sa + i * 2048);
    bx = P5_LDF(sb);
  }

  // the finished tile waiting to be stored: pk[i][j] = row 16i + fr, columns 16j + 4fg .. +3 of the wave tile, 16-bit (8 B)
  uint2 pk[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) pk[i][j] = make_uint2(0u, 0u);
  uint32_t tad = 0;                // LDS address of this lane's bias quads in the table of the current O-step's slice (P5_PACK1)
  int pm0, pn0;                    // its tile coordinates (tile 0 at first: the zeroed pk is 'drained' into rows tile 0 itself rewrites later)
  tile_of(0, pm0, pn0);
  f32x4 acc[8][8];

  // ---- pieces of the schedule (macros: the inline-asm operands name kernel-scope variables)
  // ---- One wave issues one instruction per four cycles at best and is alone on its SIMD: whatever stands between two MFMAs is
  // issue time the matrix pipe idles through unless it fits the ~3 slots an MFMA (16 cycles) leaves.  Every fragment read, DMA
  // piece, drain instruction and address computation of a stage is therefore placed in a GAP behind one particular MFMA
  // (G0..G7 of a group of eight), a few instructions each, never as a block between groups.
#define P5_NOP_ (void)0
#define P5_ECOLG(J, BF, G0, G1, G2, G3, G4, G5, G6, G7)                                                                        \
  do {                                                                                                                         \
    P5_SB; p5_mfma<F>(acc[0][J], BF, a[0]); P5_SB; G0; P5_SB; p5_mfma<F>(acc[1][J], BF, a[1]); P5_SB; G1;                      \
    P5_SB; p5_mfma<F>(acc[2][J], BF, a[2]); P5_SB; G2; P5_SB; p5_mfma<F>(acc[3][J], BF, a[3]); P5_SB; G3;                      \
    P5_SB; p5_mfma<F>(acc[4][J], BF, a[4]); P5_SB; G4; P5_SB; p5_mfma<F>(acc[5][J], BF, a[5]); P5_SB; G5;                      \
    P5_SB; p5_mfma<F>(acc[6][J], BF, a[6]); P5_SB; G6; P5_SB; p5_mfma<F>(acc[7][J], BF, a[7]); P5_SB; G7; P5_SB;               \
  } while (0)
#define P5_OROWG(I, AF, G0, G1, G2, G3, G4, G5, G6, G7)                                                                        \
  do {                                                                                                                         \
    P5_SB; p5_mfma<F>(acc[I][0], b[0], AF); P5_SB; G0; P5_SB; p5_mfma<F>(acc[I][1], b[1], AF); P5_SB; G1;                      \
    P5_SB; p5_mfma<F>(acc[I][2], b[2], AF); P5_SB; G2; P5_SB; p5_mfma<F>(acc[I][3], b[3], AF); P5_SB; G3;                      \
    P5_SB; p5_mfma<F>(acc[I][4], b[4], AF); P5_SB; G4; P5_SB; p5_mfma<F>(acc[I][5], b[5], AF); P5_SB; G5;                      \
    P5_SB; p5_mfma<F>(acc[I][6], b[6], AF); P5_SB; G6; P5_SB; p5_mfma<F>(acc[I][7], b[7], AF); P5_SB; G7; P5_SB;               \
  } while (0)

  // ---- E-step of the stage.  FIRST: a tile's first k-step -- the accumulators were re-initialised to the bias by the previous
  // O-step's LDS reads (P5_INITRD), which must have landed.  Column block 0 reads in k-step 0's B fragments 1..7 and the O-step's
  // spare; block J refills b[J-1] with k-step 1's fragment and issues piece J + 2 of the pending A request.
#define P5_EW(F_, N_) P5_IF(F_, asm volatile("s_waitcnt lgkmcnt(" #N_ ")" ::: "memory"))
#define P5_ESTEP(FIRST)                                                                                                        \
  do {                                                                                                                         \
    const char* sa0__ = smem + va0;                                                                                            \
    const char* sa1__ = smem + (va0 ^ 64u);                                                                                    \
    const char* sb0__ = smem + vq0;                                                                                            \
    const char* sb1__ = smem + (vq0 ^ 64u);                                                                                    \
    P5_EW(FIRST, 0);                            /* (the accumulators' re-initialisation reads of the last O-step have landed) */ \
    a[7] = P5_LDF(sa0__ + 7 * 2048);            /* (the previous O-step's last refill: its row block 7 has just been issued) */ \
    P5_ECOLG(0, bx, b[1] = P5_LDF(sb0__ + 1 * 2048), b[2] = P5_LDF(sb0__ + 2 * 2048), b[3] = P5_LDF(sb0__ + 3 * 2048), \
             b[4] = P5_LDF(sb0__ + 4 * 2048), b[5] = P5_LDF(sb0__ + 5 * 2048), b[6] = P5_LDF(sb0__ + 6 * 2048), \
             b[7] = P5_LDF(sb0__ + 7 * 2048), ax = P5_LDF(sa1__));                     \
    P5_ECOLG(1, b[1], b[0] = P5_LDF(sb1__ + 0 * 2048), P5_NOP_, P5_NOP_, P5_PIECE_A(3), P5_NOP_, \
             P5_NOP_, P5_NOP_, P5_NOP_);                                                      \
    P5_ECOLG(2, b[2], b[1] = P5_LDF(sb1__ + 1 * 2048), P5_NOP_, P5_NOP_, P5_PIECE_A(4), P5_NOP_, \
             P5_NOP_, P5_NOP_, P5_NOP_);                                                      \
    P5_ECOLG(3, b[3], b[2] = P5_LDF(sb1__ + 2 * 2048), P5_NOP_, P5_NOP_, P5_PIECE_A(5), P5_NOP_, \
             P5_NOP_, P5_NOP_, P5_NOP_);                                                      \
    P5_ECOLG(4, b[4], b[3] = P5_LDF(sb1__ + 3 * 2048), P5_NOP_, P5_NOP_, P5_PIECE_A(6), P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_);   \
    P5_ECOLG(5, b[5], b[4] = P5_LDF(sb1__ + 4 * 2048), P5_NOP_, P5_NOP_, P5_PIECE_A(7), P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_);   \
    P5_ECOLG(6, b[6], b[5] = P5_LDF(sb1__ + 5 * 2048), P5_NOP_, P5_NOP_, P5_DONE_A(), P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_);     \
    P5_ECOLG(7, b[7], b[6] = P5_LDF(sb1__ + 6 * 2048), P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_);         \
  } while (0)

  // block (I, J) of the finished accumulators: rounded into pk, then RE-INITIALISED for the next tile -- its bias quad (columns
  // 16J + 4fg .. +3 of the wave) comes straight into the AGPRs by one ds_read_b128 from a 512-B table of the wave's 128 bias
  // columns that the O-step has put into the free half of its slice (on gfx90a+ an MFMA's srcC and vdst share one AGPR / VGPR
  // select bit, so "vdst in AGPRs, srcC = a VGPR quad" does not encode, and 4 x v_accvgpr_write per block is issue time).
#define P5_PACK1(I, J)                                                                                                         \
  do {                                                                                                                         \
    asm volatile("" : "+a"(acc[I][J]));          /* (still in its AGPRs HERE: the allocator otherwise evacuates a finished */   \
    const f32x4 v__ = acc[I][J];                 /*  row block into 32 VGPRs right behind its last MFMA)                   */   \
    pk[I][J] = make_uint2(H16<F>::pack2(v__[0], v__[1]), H16<F>::pack2(v__[2], v__[3]));                                       \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(acc[I][J]) : "v"(tad), "n"((J) * 64) : "memory");                    \
  } while (0)

  // ---- drain of unit U (row block U of a finished tile at rows DM0.., columns DN0.. of the wave) through this wave's slice
  // of the A slot the barrier just freed (BO: byte offset inside the slice, 0 or 4096), in gap-sized steps:
  //   WP: the four write addresses of the p3 epilogue's swizzle (two halves of 64 columns, 2 KB each: row fr at fr*128, 16-B
  //       chunk (2jj + (fg>>1)) ^ ((fr>>1)&7), + (fg&1)*8);  W(j): one ds_write_b64;
  //   R(H): half H back as 2 x ds_read_b128 (lane -> row it*8 + (lane>>3), chunk lane&7 = full 128-B lines);
  //   SP: the two store offsets;  S(H): 2 x global_store_dwordx4 (nt).
#define P5_DRAIN_WP(BO)                                                                                                        \
  do {                                                                                                                         \
    P5_LANE(le__);                                                                                                             \
    const int fr__ = le__ & 15, fg__ = le__ >> 4;                                                                              \
    char* bw__ = bounce__ + (BO) + fr__ * 128 + (fg__ & 1) * 8;                                                                \
    const int x__ = ((fg__ >> 1) ^ ((fr__ >> 1) & 7)) << 4;                                                                    \
    wa0__ = bw__ + x__; wa1__ = bw__ + (x__ ^ 32); wa2__ = bw__ + (x__ ^ 64); wa3__ = bw__ + (x__ ^ 96);                       \
  } while (0)
#define P5_DRAIN_W1(U, J) (*reinterpret_cast<uint2*>(((J) & 3) == 0 ? wa0__ + ((J) >> 2) * 2048 : ((J) & 3) == 1 ? wa1__ + ((J) >> 2) * 2048 : \
                                                    ((J) & 3) == 2 ? wa2__ + ((J) >> 2) * 2048 : wa3__ + ((J) >> 2) * 2048) = pk[U][J])
#define P5_DRAIN_R(H, IT, BO)                                                                                                  \
  do {                                                                                                                         \
    P5_LANE(le__);                                                                                                             \
    w0__ = *reinterpret_cast<const u32x4*>(bounce__ + (BO) + (H) * 2048 + (IT) * 1024 + le__ * 16);                            \
  } while (0)
#define P5_DRAIN_SP()                                                                                                          \
  do {                                                                                                                         \
    P5_LANE(le__);                                                                                                             \
    const int lr__ = le__ >> 3;                                                                                                \
    const int e0__ = ((le__ & 7) ^ (le__ >> 4)) << 3, e1__ = ((le__ & 7) ^ (4 + (le__ >> 4))) << 3;                            \
    const int ldo__ = HMC ? 64 : ldc;                                                                                          \
    eo0__ = (uint32_t)(lr__ * ldo__ + e0__) * 2; eo1__ = (uint32_t)((lr__ + 8) * ldo__ + e1__) * 2;                            \
  } while (0)
#define P5_DRAIN_S(U, H, IT, DM0, DN0)                                                                                         \
  do {                                                                                                                         \
    const int mw__ = (DM0) + wr * 128 + (U) * 16, nw__ = (DN0) + wc * 128 + (H) * 64;                                          \
    bf16_t* const o__ = HMC ? C + ((long)(nw__ >> 6) * hmR + mw__) * 64 : C + (long)mw__ * ldc + nw__;                         \
    P_GST4((IT) ? eo1__ : eo0__, o__, w0__);                                                                                   \
  } while (0)
#define P5_IF(C_, X) do { if (C_) { X; } } while (0)
#define P5_NEXTBIAS()                                                                                                          \
  do {                                                                                                                         \
    int nm0__, nn0__;                                                                                                          \
    tile_of(k + 1, nm0__, nn0__);                                                                                              \
    if (bias) {                                                                                                                \
      P5_LANE(lb__);                                                                                                           \
      const float* bp__ = bias + nn0__ + wc * 128;                                                                             \
      const uint32_t bo__ = (uint32_t)lb__ * 4u;                                                                               \
      asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:256" : "=&v"(vb0), "=&v"(vb1) : "v"(bo__), "s"(bp__) : "memory"); \
    }                                                                                                                          \
  } while (0)

  // ---- O-step of the stage, with the stage barrier and the requests B(g+2), A(g+3) behind it.
  //   UA >= 0: drain that unit of the PARKED tile (the previous one; on the workgroup's first tile the zero-initialised pk goes to
  //            the first tile's own rows and is overwritten by its real drain later: no branch);
  //   NEXTBIAS: request the next tile's bias columns;
  //   LAST: the tile's last k-step -- row block I is rounded into pk and re-initialised in the gaps of row block I + 1 (the bias
  //   table sits in the last KB of the slice -- the A piece that overwrites it is issued in the next E-step's column block 5 --,
  //   the lower half drains row block 0 at once: seven row blocks stay parked).
  // VMEM issue order behind the barrier: B x 8 with the stores [and the bias request] among them, then A x 3 (the other five A
  // pieces: next E-step) -- the 8 youngest at the next barrier are exactly the A request.
#define P5_LG(L_, I, J) P5_IF(L_, P5_PACK1(I, J))
#define P5_OSTEP(UA, NEXTBIAS, LAST)                                                                                           \
  do {                                                                                                                         \
    const char* sa1__ = smem + (va0 ^ 64u);                                                                                    \
    const char* sb1__ = smem + (vq0 ^ 64u);                                                                                    \
    char* bounce__ = smem + (oA - uA) + wave * 8192;                                                                           \
    const uint32_t oAn__ = oA + 32768u >= 98304u ? oA - 65536u : oA + 32768u;                                                  \
    u32x4 w0__;                                                                                                                \
    char *wa0__, *wa1__, *wa2__, *wa3__;                                                                                       \
    uint32_t eo0__, eo1__;                                                                                                     \
    constexpr bool da__ = (LAST) || (UA) >= 0;                                                                                 \
    constexpr int ua__ = (LAST) ? 0 : ((UA) < 0 ? 0 : (UA));                                                                   \
    const int dm__ = (LAST) ? m0 : pm0, dn__ = (LAST) ? n0 : pn0;                                                              \
    P5_OROWG(0, ax, b[7] = P5_LDF(sb1__ + 7 * 2048); a[1] = P5_LDF(sa1__ + 1 * 2048), a[2] = P5_LDF(sa1__ + 2 * 2048),         \
             a[3] = P5_LDF(sa1__ + 3 * 2048), a[4] = P5_LDF(sa1__ + 4 * 2048), a[5] = P5_LDF(sa1__ + 5 * 2048),                \
             a[6] = P5_LDF(sa1__ + 6 * 2048), a[7] = P5_LDF(sa1__ + 7 * 2048), P5_NOP_);                                       \
    P5_OROWG(1, a[1], P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_, P5_NOP_);                                 \
    /* every fragment of the stage is in registers (lgkmcnt(0)); all but the 8 youngest VMEM operations (the A(g+2) pieces)      \
       have completed = stage g+1 has landed (and the bias columns requested ahead of them) */                                   \
    if (LAST) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" : "+v"(vb0), "+v"(vb1)::"memory");                     \
    else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                              \
    va0 = fx + oAn__; vq0 = fx + (oBsum - oB);                                                                                 \
    const char* san__ = smem + va0;                                                                                            \
    const char* sbn__ = smem + vq0;                                                                                            \
    if (LAST) {                   /* the next tile's bias columns -> table (lane l: columns l, 64 + l), then row block 0 */     \
      P5_LANE(lt__);                                                                                                           \
      *reinterpret_cast<float*>(bounce__ + 7168 + lt__ * 4) = vb0;                                                             \
      *reinterpret_cast<float*>(bounce__ + 7168 + 256 + lt__ * 4) = vb1;                                                       \
      tad = lds0 + (oA - uA) + (uint32_t)(wave * 8192 + 7168) + (uint32_t)((lt__ >> 4) << 4);                                  \
      P5_SB;                                                                                                                   \
      P5_PACK1(0, 0); P5_PACK1(0, 1); P5_PACK1(0, 2); P5_PACK1(0, 3); P5_PACK1(0, 4); P5_PACK1(0, 5); P5_PACK1(0, 6); P5_PACK1(0, 7); \
    }                                                                                                                          \
    P5_OROWG(2, a[2], a[0] = P5_LDF(san__); P5_LG(LAST, 1, 0), a[1] = P5_LDF(san__ + 2048); P5_LG(LAST, 1, 1), bx = P5_LDF(sbn__); P5_LG(LAST, 1, 2), \
             P5_REQ_B(); P5_LG(LAST, 1, 3), P5_PIECE_B(0); P5_LG(LAST, 1, 4), P5_PIECE_B(1); P5_LG(LAST, 1, 5),                \
             P5_IF(da__, P5_DRAIN_WP(0)); P5_LG(LAST, 1, 6), P5_LG(LAST, 1, 7));                                               \
    P5_OROWG(3, a[3], a[2] = P5_LDF(san__ + 2 * 2048); P5_IF(da__, P5_DRAIN_W1(ua__, 0)); P5_LG(LAST, 2, 0), P5_IF(da__, P5_DRAIN_W1(ua__, 1)); P5_LG(LAST, 2, 1), \
             P5_IF(da__, P5_DRAIN_W1(ua__, 2)); P5_LG(LAST, 2, 2), P5_PIECE_B(2); P5_IF(da__, P5_DRAIN_W1(ua__, 3)); P5_LG(LAST, 2, 3), \
             P5_IF(da__, P5_DRAIN_W1(ua__, 4)); P5_LG(LAST, 2, 4), P5_IF(da__, P5_DRAIN_W1(ua__, 5)); P5_LG(LAST, 2, 5),       \
             P5_PIECE_B(3); P5_IF(da__, P5_DRAIN_W1(ua__, 6)); P5_LG(LAST, 2, 6), P5_IF(da__, P5_DRAIN_W1(ua__, 7)); P5_LG(LAST, 2, 7)); \
    P5_OROWG(4, a[4], a[3] = P5_LDF(san__ + 3 * 2048); P5_LG(LAST, 3, 0), P5_IF(da__, P5_DRAIN_R(0, 0, 0)); P5_LG(LAST, 3, 1), P5_IF(da__, P5_DRAIN_SP()); P5_LG(LAST, 3, 2), \
             P5_PIECE_B(4); P5_LG(LAST, 3, 3), P5_LG(LAST, 3, 4), P5_IF(da__, P5_DRAIN_S(ua__, 0, 0, dm__, dn__)); P5_IF(da__, P5_DRAIN_R(0, 1, 0)); P5_LG(LAST, 3, 5), \
             P5_PIECE_B(5); P5_LG(LAST, 3, 6), P5_LG(LAST, 3, 7));                                                             \
    P5_OROWG(5, a[5], a[4] = P5_LDF(san__ + 4 * 2048); P5_LG(LAST, 4, 0), P5_LG(LAST, 4, 1), P5_IF(da__, P5_DRAIN_S(ua__, 0, 1, dm__, dn__)); P5_IF(da__, P5_DRAIN_R(1, 0, 0)); P5_LG(LAST, 4, 2), \
             P5_PIECE_B(6); P5_LG(LAST, 4, 3), P5_LG(LAST, 4, 4), P5_IF(da__, P5_DRAIN_S(ua__, 1, 0, dm__, dn__)); P5_IF(da__, P5_DRAIN_R(1, 1, 0)); P5_LG(LAST, 4, 5), \
             P5_PIECE_B(7); P5_LG(LAST, 4, 6), P5_IF(NEXTBIAS, P5_NEXTBIAS()); P5_LG(LAST, 4, 7));                             \
    P5_OROWG(6, a[6], a[5] = P5_LDF(san__ + 5 * 2048); P5_LG(LAST, 5, 0), P5_LG(LAST, 5, 1), P5_IF(da__, P5_DRAIN_S(ua__, 1, 1, dm__, dn__)); P5_LG(LAST, 5, 2), \
             P5_LG(LAST, 5, 3), P5_DONE_B(); P5_LG(LAST, 5, 4), P5_LG(LAST, 5, 5), P5_LG(LAST, 5, 6), P5_LG(LAST, 5, 7));      \
    P5_OROWG(7, a[7], a[6] = P5_LDF(san__ + 6 * 2048); P5_LG(LAST, 6, 0), P5_LG(LAST, 6, 1), P5_REQ_A(); P5_LG(LAST, 6, 2), P5_PIECE_A(0); P5_LG(LAST, 6, 3), \
             P5_PIECE_A(1); P5_LG(LAST, 6, 4), P5_PIECE_A(2); P5_LG(LAST, 6, 5), P5_LG(LAST, 6, 6), P5_LG(LAST, 6, 7));        \
    if (LAST) {                                                                                                                \
      /* (an MFMA's result may be read by the VALU only some passes after issue; the hazard recognizer cannot see the MFMAs      \
         inside the inline asm, so the last row block waits explicitly) */                                                       \
      asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");                                                              \
      P5_PACK1(7, 0); P5_PACK1(7, 1); P5_PACK1(7, 2); P5_PACK1(7, 3); P5_PACK1(7, 4); P5_PACK1(7, 5); P5_PACK1(7, 6); P5_PACK1(7, 7); \
      P5_SB;                                                                                                                   \
    }                                                                                                                          \
    oA = oAn__; oB = oBsum - oB;                                                                                               \
  } while (0)

  // the first tile's accumulators: tile 0's bias through the same table (this wave's slice of A slot 2: its first real use is
  // behind the barrier of stage 2; the prologue's A(2) pieces 0..2 wrote bytes [0, 3K) of it, the table sits at [7K, 7.5K): piece 7 is issued in column block 5)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (A(2)'s first pieces must not be overtaken -- once per kernel)
  {
    char* tb = smem + 65536 + wave * 8192 + 7168;
    *reinterpret_cast<float*>(tb + lane * 4) = vb0;
    *reinterpret_cast<float*>(tb + 256 + lane * 4) = vb1;
    const uint32_t tad__ = lds0 + 65536u + (uint32_t)(wave * 8192 + 7168) + (uint32_t)((lane >> 4) << 4);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("ds_read_b128 %0, %1 offset:0" : "=a"(acc[i][0]) : "v"(tad__) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:64" : "=a"(acc[i][1]) : "v"(tad__) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:128" : "=a"(acc[i][2]) : "v"(tad__) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:192" : "=a"(acc[i][3]) : "v"(tad__) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:256" : "=a"(acc[i][4]) : "v"(tad__) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:320" : "=a"(acc[i][5]) : "v"(tad__) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:384" : "=a"(acc[i][6]) : "v"(tad__) : "memory");
      asm volatile("ds_read_b128 %0, %1 offset:448" : "=a"(acc[i][7]) : "v"(tad__) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // (all 64 quads are in their AGPRs HERE: without the pins the allocator gives every read the same four AGPRs and copies the
    // still in-flight result into VGPRs behind it)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
  }

  for (int k = 0; k < my_tiles; ++k) {
    int m0, n0;
    tile_of(k, m0, n0);
    // (the accumulators cross the loop edge IN their AGPRs: left alone, the allocator carries the not-yet-rounded column blocks
    // 4..7 of the finished tile -- 112 f32 values -- through the edge in VGPRs and spills)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" : "+a"(acc[i][j]));
    P5_ESTEP(true);  P5_OSTEP(1, false, false);
    P5_ESTEP(false); P5_OSTEP(2, false, false);
    P5_ESTEP(false); P5_OSTEP(3, false, false);
    P5_ESTEP(false); P5_OSTEP(4, false, false);
    P5_ESTEP(false); P5_OSTEP(5, false, false);
    P5_ESTEP(false); P5_OSTEP(6, false, false);
    P5_ESTEP(false); P5_OSTEP(7, false, false);
    for (int st = 7; st < nst - 2; ++st) {
      P5_ESTEP(false); P5_OSTEP(-1, false, false);
    }
    P5_ESTEP(false); P5_OSTEP(-1, true, false);
    P5_ESTEP(false); P5_OSTEP(-1, false, true);
    // the re-initialisation reads (inline-asm outputs hipcc cannot see in flight) land BEFORE the loop edge: a copy the allocator
    // places on the edge would otherwise copy registers whose data has not arrived (~200 cycles per tile, 0.15 % at K = 3072)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pm0 = m0; pn0 = n0;
  }

  // ---- the last tile's parked row blocks: drained in the open.  The trailing (dummy) stage loads must land before this wave's
  // slice is reused and before the LDS is released.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  {
    char* bounce__ = smem + wave * 8192;
    u32x4 w0__;
#define P5_TAIL(U)                                                                                                             \
  do {                                                                                                                         \
    P5_DRAIN_WP(0);                                                                                                            \
    P5_DRAIN_W1(U, 0); P5_DRAIN_W1(U, 1); P5_DRAIN_W1(U, 2); P5_DRAIN_W1(U, 3); P5_DRAIN_W1(U, 4); P5_DRAIN_W1(U, 5);          \
    P5_DRAIN_W1(U, 6); P5_DRAIN_W1(U, 7);                                                                                      \
    P5_DRAIN_R(0, 0, 0); P5_DRAIN_S(U, 0, 0, pm0, pn0); P5_DRAIN_R(0, 1, 0); P5_DRAIN_S(U, 0, 1, pm0, pn0);                    \
    P5_DRAIN_R(1, 0, 0); P5_DRAIN_S(U, 1, 0, pm0, pn0); P5_DRAIN_R(1, 1, 0); P5_DRAIN_S(U, 1, 1, pm0, pn0);                    \
  } while (0)
    char *wa0__, *wa1__, *wa2__, *wa3__;
    uint32_t eo0__, eo1__;
    P5_DRAIN_SP();
    P5_TAIL(1); P5_TAIL(2); P5_TAIL(3); P5_TAIL(4); P5_TAIL(5); P5_TAIL(6); P5_TAIL(7);
#undef P5_TAIL
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static bool p5_attr_done[SIMX_MAX_DEVICES];
// Eligibility + launch; returns SIMX_ERR_UNSUPPORTED (without an error text) when the shape is outside this kernel's rules, so that
// the caller falls through to gemm_nt_p3_kernel.
int simx_launch_nt_p5(hipStream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                      const float* bias, int hm_c_rows, int ncu) {
  if (!(M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && K >= 576 && lda % 8 == 0 && lda == ldb && simx_is16(dtype))) return SIMX_ERR_UNSUPPORTED;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIMX_MAX_DEVICES) return SIMX_ERR_UNSUPPORTED;
  if (!p5_attr_done[dev]) {
    bool ok = true;
#define P5_ATTR(KRN) ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(KRN), hipFuncAttributeMaxDynamicSharedMemorySize, P5_LDS) == hipSuccess
    P5_ATTR((gemm_nt_p5_kernel<f16_t, false>)); P5_ATTR((gemm_nt_p5_kernel<f16_t, true>));
    P5_ATTR((gemm_nt_p5_kernel<bf16_t, false>)); P5_ATTR((gemm_nt_p5_kernel<bf16_t, true>));
#undef P5_ATTR
    if (!ok) return SIMX_ERR_UNSUPPORTED;
    p5_attr_done[dev] = true;
  }
  const int t_n = N / 256, ntiles = (M / 256) * t_n;
  const int grid = ntiles < ncu ? ntiles : ncu;
#define LP5(FF, HM) hipLaunchKernelGGL((gemm_nt_p5_kernel<FF, HM>), dim3(grid), dim3(256), P5_LDS, s, M, N, K, (const bf16_t*)A, lda, \
                                       (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, hm_c_rows, t_n, ntiles)
  if (dtype == SIMX_F16) { if (hm_c_rows > 0) LP5(f16_t, true); else LP5(f16_t, false); }
  else { if (hm_c_rows > 0) LP5(bf16_t, true); else LP5(bf16_t, false); }
#undef LP5
  SIMX_CHECK_LAUNCH("gemm_nt_p5");
  return SIMX_OK;
}
