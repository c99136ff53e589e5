// Weight-gradient GEMM (TN) with ONE wave per SIMD and the accumulators in AGPRs ("tn5"), gfx950.
//
//   slab[split][M][N] = A[kslice, M]^T . B[kslice, N]      A = dY, B = X: 16-bit, [tokens, columns], f32 accumulate, f32 slabs
//   (loss.backward() of the dense projections: SimANS/co_training/co_training_marco_train.py:222, LEAD/modeling_bert.py:229-232,
//    440-466; the slabs are summed in slice order by slab_reduce_kernel, csrc/gemm.hip)
//
// Why a second large wgrad kernel beside gemm_tn2_kernel (csrc/gemm.hip: 8 waves, 128 x 64 wave tiles, two 64 KB stages).  tn2 runs
// at 0.39 of the MFMA peak with the matrix pipes idle half the time at 2.0 GHz -- it is not power-limited; both of its operands
// stream from HBM and its two-stage ring has ONE 64 KB stage in flight (LDS-DMA delivers ~18 B/clk/CU with 64 KB in flight, ~27
// with 96, 36 with 128: tools/vmem_bench).  Round 2's gemm_tn4 (one wave per SIMD) lost to it by 2 %; round 6's gemm_nt_p5_kernel
// (csrc/gemm_p5.hip, profiles/r06_experiments/01_p5.md) showed why such kernels lose -- a lone wave issues one instruction per
// four cycles, so whatever stands between two MFMAs beyond ~3 instructions is matrix-pipe idle time -- and that with every
// instruction placed in a gap behind one particular MFMA the one-wave-per-SIMD loop is 4 % FASTER than the two-wave one.  This
// kernel is that loop for the TN product:
//   * 256 threads = 4 waves (2 x 2), wave tile 128 x 128 = 8 x 8 blocks of v_mfma_f32_16x16x32: 256 accumulators in AGPRs (inline-
//     asm MFMA, tied "+a"); both fragment sets of a k-step (8 + 8 quads) double-buffered in VGPRs -- nothing is parked here, a
//     workgroup owns ONE output tile over its token slice.
//   * LDS: p3 / p5's ring -- 64-token stages, three 32 KB A slots + two B slots = 160 KB, 96 KB in flight; a stage slot is
//     [64 tokens][256 columns] (512 B per token row, 32-B chunk q at q ^ (row & 7)), fragments by ds_read_b64_tr_b16 (two per
//     fragment: token rows 0-15 / 16-31 of the k-step, the second through the immediate offset: (row + 16) & 7 == row & 7).
//   * one barrier per stage, at the start of its second k-step (every fragment of the stage is in registers):
//     s_waitcnt vmcnt(8)  ("all but the eight A(g+2) pieces" = stage g+1 has landed), s_barrier, then B(g+2) x 8 in the first
//     half of that k-step (it is needed one stage later and comes from HBM like A) and A(g+3) x 8 over the next k-step.
//   * fused bias gradient (column sums of A over the tokens) on 1 / (2 tiles_n) of the stages per wave, v_dot2_f32_f16 against
//     (1, 1) for fp16, in the gaps behind the row block's own MFMAs.
//   * epilogue: the f32 slab tile goes out straight from the AGPRs (global_store_dwordx4 takes AGPR data): no VALU.
// Rules: full 256 x 256 tiles, >= 2 splits (slab path) of >= 4 stages, not the deterministic mode (gemm_tn2 keeps those); a ragged
// token count (K % 64 != 0) ends the last split with one clamped, zero-padded stage behind the main loop.  Same sums as tn2 up to the order of the f32 additions inside a slice.
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "p3.h"

#define T5_LDS (5 * 32768)
#define T5_SB __builtin_amdgcn_sched_barrier(0)

template <typename F>
__device__ __forceinline__ void t5_mfma(f32x4& acc, const bf16x8& bw, const bf16x8& ax) {
  if constexpr (std::is_same<F, f16_t>::value) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(bw), "v"(ax));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(bw), "v"(ax));
}
// one fragment = two transpose reads (token rows base + 4g .. and base + 16 + 4g ..) into one 128-bit tuple
__device__ __forceinline__ bf16x8 t5_frag(uint32_t addr) {
  const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds4_t)(uintptr_t)addr);
  const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds4_t)(uintptr_t)(addr + 8192u));
  return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// c + the two values of pair E (0..3 / 4..7: elements 2E, 2E+1) of a fragment, in f32
template <typename F>
__device__ __forceinline__ float t5_sum2(const bf16x8& f, int e, float c) {
  if constexpr (std::is_same<F, f16_t>::value) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef short s2 __attribute__((ext_vector_type(2)));
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h2, (s2){f[2 * e], f[2 * e + 1]}), one, c, false);
  } else {
    return c + H16<F>::one(f[2 * e]) + H16<F>::one(f[2 * e + 1]);
  }
}
template <typename F>
__device__ __forceinline__ float t5_sum8(const bf16x8& f, float c) {
  c = t5_sum2<F>(f, 0, c); c = t5_sum2<F>(f, 1, c); c = t5_sum2<F>(f, 2, c);
  return t5_sum2<F>(f, 3, c);
}

template <typename F>
__global__ __launch_bounds__(256) void gemm_tn5_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    float* __restrict__ out, long slab_stride, int ldo, int tiles_n, int tiles_mn, int k_per_split,
    float* __restrict__ dbias, int hm_a, const float* __restrict__ gs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // XCD-aware order as gemm_tn2_kernel: the tiles of one split share their token range behind one L2
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vb / tiles_mn;
  const int tile = vb % tiles_mn;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int kb = split * k_per_split;
  const int ke = min(K, kb + k_per_split);
  const int nst = (ke - kb) / 64;                                   // whole stages; `tail` more tokens (last split of a ragged K) follow
  const int tail = (ke - kb) - nst * 64;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t ldsB = lds0 + 98304u;
  // fused bias gradient: the column sums of A over this workgroup's tokens are shared out stage by stage over the tiles_n
  // workgroups that stage the same A tile and over their two wc waves (which hold the same A fragments)
  const int bias_slot = (tile % tiles_n) * 2 + wc, bias_mod = tiles_n * 2;

  // ---- LDS-DMA lane offsets: an instruction covers 2 token rows x 512 B; lane -> row (lane >> 5), 16-B piece lane & 31 of the
  // row, which holds source chunk ((lane & 31) >> 1) ^ (row & 7).  One offset register per piece j (rows wave*16 + 2j ..) and
  // operand: the request's base pointer stays put, a piece costs two instructions.
  const int lda_e = hm_a > 0 ? 64 : lda;                            // head-major A ([M/64][hm][64], csrc/attention.hip QkvLay): row pitch 64
  uint32_t offA[8], offB[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int krl = 2 * j + (lane >> 5);                            // (wave*16 + krl) & 7 == krl & 7
    const int p16 = lane & 31;
    const int col = (((p16 >> 1) ^ (krl & 7)) << 4) + (p16 & 1) * 8;
    offA[j] = ((uint32_t)(krl * lda_e) + tn2_acol(m0 + col, hm_a)) * 2;
    offB[j] = (uint32_t)(krl * ldb + n0 + col) * 2;
  }
  const long strA = (long)lda_e * 128, strB = (long)ldb * 128;     // bytes per stage (64 tokens)
  const char* pA = reinterpret_cast<const char*>(A + (long)(kb + wave * 16) * lda_e);
  const char* pB = reinterpret_cast<const char*>(B + (long)(kb + wave * 16) * ldb);
  int stA = 0, stB = 0;                                             // stage of the next request (clamped to the last one past the end)
  uint32_t rB = ldsB + (uint32_t)(wave * 8192), rA = lds0 + (uint32_t)(wave * 8192);
  const uint32_t rBsum = 2u * rB + 32768u;
#define T5_REQ_B() asm volatile("s_mov_b32 m0, %0" ::"s"(rB) : "memory")
#define T5_REQ_A() asm volatile("s_mov_b32 m0, %0" ::"s"(rA) : "memory")
#define T5_PIECE_B(J) asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x400" ::"v"(offB[J]), "s"(pB) : "memory", "scc")
#define T5_PIECE_A(J) asm volatile("global_load_lds_dwordx4 %0, %1\n\ts_add_u32 m0, m0, 0x400" ::"v"(offA[J]), "s"(pA) : "memory", "scc")
#define T5_DONE_B() do { rB = rBsum - rB; if (++stB < nst) pB += strB; } while (0)
#define T5_DONE_A() do { rA = rA + 32768u >= ldsB ? rA - 65536u : rA + 32768u; if (++stA < nst) pA += strA; } while (0)
#define T5_ALL8(M_) do { M_(0); M_(1); M_(2); M_(3); M_(4); M_(5); M_(6); M_(7); } while (0)
  T5_REQ_B(); T5_ALL8(T5_PIECE_B); T5_DONE_B();
  T5_REQ_A(); T5_ALL8(T5_PIECE_A); T5_DONE_A();                     // stage 0
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  T5_REQ_B(); T5_ALL8(T5_PIECE_B); T5_DONE_B();
  T5_REQ_A(); T5_ALL8(T5_PIECE_A); T5_DONE_A();                     // B(1) A(1); A(2) goes out in stage 0's first k-step like every A(g+2):
                                                                    // B(g+2) is requested behind the barrier of stage g, A(g+3) in the k-step after it

  // ---- fragment addressing: lane (fg, fs) supplies token row 4 fg + (fs >> 2) of a 32-row k-step, 8 B at element column
  // ct*16 + (fs & 3)*4 of column tile ct -> 32-B chunk ct ^ (row & 7).  V = row*512 + (fs & 3)*8 + ((row & 7) << 5) is one lane
  // constant; the fragment of the wave's tile i is at (V ^ (i << 5)) + slot + k-step*16384 + wave columns*2 (bits 5..7 of the
  // uniform part are clear).
  const int r_lo = 4 * (lane >> 4) + ((lane & 15) >> 2);
  const uint32_t V = (uint32_t)(r_lo * 512 + (lane & 3) * 8 + ((r_lo & 7) << 5));
  uint32_t oA = lds0 + (uint32_t)(wr * 256), oB = ldsB + (uint32_t)(wc * 256);     // slots of the stage being consumed (+ the wave's columns)
  const uint32_t oA0 = oA, oBsum = 2u * oB + 32768u;

// a fragment's two transpose reads go out in two consecutive gaps (address + rows 0-15, then rows 16-31 through the immediate
// offset): two instructions per gap at most, so that the s_waitcnt hipcc puts in front of an MFMA still fits the gap's three slots
#define T5_RDLO(ARR, NXT, Q, BASE) do { ra__ = (V ^ (uint32_t)((Q) << 5)) + (BASE); ARR##lo[NXT][Q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds4_t)(uintptr_t)ra__); } while (0)
#define T5_RDHI(ARR, NXT, Q) ARR##hi[NXT][Q] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn_lds4_t)(uintptr_t)(ra__ + 8192u))
#define T5_FR(ARR, S, Q) ((bf16x8){ARR##lo[S][Q][0], ARR##lo[S][Q][1], ARR##lo[S][Q][2], ARR##lo[S][Q][3], ARR##hi[S][Q][0], ARR##hi[S][Q][1], ARR##hi[S][Q][2], ARR##hi[S][Q][3]})
  bf16x4 falo[2][8], fahi[2][8], fblo[2][8], fbhi[2][8];          // fragment halves (token rows 0-15 / 16-31 of the k-step)
  uint32_t ra__ = 0;
  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; asm volatile("" : "+a"(acc[i][j])); }
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) { T5_RDLO(fb, 0, i, oB); T5_RDHI(fb, 0, i); }
#pragma unroll
  for (int i = 0; i < 8; ++i) { T5_RDLO(fa, 0, i, oA); T5_RDHI(fa, 0, i); }

  // ---- one k-step: 64 MFMAs on fragment set CUR, the 16 fragments of the NEXT k-step read into set NXT behind every third
  // MFMA (B first: the next k-step's first row block needs all of B and A[0]); G(g): the gap's other work (DMA pieces, request
  // bookkeeping, bias sums).  `na`, `nb`: LDS base of the next k-step's A / B fragments.
#ifdef T5_UNSPLIT_READS        /* tools/build_variant.sh experiment: both reads of a fragment in one gap (three instructions) */
#define T5_RD(NXT, G)                                                                                          \
  do {                                                                                                         \
    if ((G) % 3 == 1 && (G) / 3 < 8) { T5_RDLO(fb, NXT, ((G) / 3) & 7, nb__); T5_RDHI(fb, NXT, ((G) / 3) & 7); }  \
    else if ((G) % 3 == 1 && (G) / 3 < 16) { T5_RDLO(fa, NXT, ((G) / 3) & 7, na__); T5_RDHI(fa, NXT, ((G) / 3) & 7); } \
  } while (0)
#else
#define T5_RD(NXT, G)                                                                                          \
  do {                                                                                                         \
    if ((G) % 3 == 1 && (G) / 3 < 8) T5_RDLO(fb, NXT, ((G) / 3) & 7, nb__);                                    \
    else if ((G) % 3 == 2 && (G) / 3 < 8) T5_RDHI(fb, NXT, ((G) / 3) & 7);                                     \
    else if ((G) % 3 == 1 && (G) / 3 < 16) T5_RDLO(fa, NXT, ((G) / 3) & 7, na__);                              \
    else if ((G) % 3 == 2 && (G) / 3 < 16) T5_RDHI(fa, NXT, ((G) / 3) & 7);                                    \
  } while (0)
#endif
// (a bias-gradient stage adds the row block's A fragment to its column sums behind the block's fifth MFMA)
#define T5_BS(BS, CUR, I, J) do { if ((BS) && (J) == 4 && do_bias) { asm volatile("" ::: "memory"); bsum[I] = t5_sum8<F>(T5_FR(fa, CUR, I), bsum[I]); } } while (0)
#define T5_M(BS, CUR, NXT, I, J, X)                                                                            \
  do {                                                                                                         \
    T5_SB; t5_mfma<F>(acc[I][J], T5_FR(fb, CUR, J), T5_FR(fa, CUR, I)); T5_SB;                                               \
    T5_RD(NXT, (I) * 8 + (J)); T5_BS(BS, CUR, I, J); X;                                                        \
  } while (0)
#define T5_NOP_ (void)0
#define T5_ROW(BS, CUR, NXT, I, X0, X1, X2, X3, X4, X5, X6, X7)                                                \
  do {                                                                                                         \
    T5_M(BS, CUR, NXT, I, 0, X0); T5_M(BS, CUR, NXT, I, 1, X1); T5_M(BS, CUR, NXT, I, 2, X2); T5_M(BS, CUR, NXT, I, 3, X3); \
    T5_M(BS, CUR, NXT, I, 4, X4); T5_M(BS, CUR, NXT, I, 5, X5); T5_M(BS, CUR, NXT, I, 6, X6); T5_M(BS, CUR, NXT, I, 7, X7); \
  } while (0)
  // k-step 0 of a stage (set 0 -> reads set 1 = the stage's own second half; the pending A request goes out)
#define T5_KS0(BS)                                                                                             \
  do {                                                                                                         \
    const uint32_t na__ = oA + 16384u, nb__ = oB + 16384u;                                                     \
    T5_ROW(BS, 0, 1, 0, T5_NOP_, T5_NOP_, T5_REQ_A(), T5_PIECE_A(0), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);          \
    T5_ROW(BS, 0, 1, 1, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_A(1), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 0, 1, 2, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_A(2), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 0, 1, 3, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_A(3), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 0, 1, 4, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_A(4), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 0, 1, 5, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_A(5), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 0, 1, 6, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_A(6), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 0, 1, 7, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_A(7), T5_NOP_, T5_DONE_A(), T5_NOP_, T5_NOP_);         \
  } while (0)
  // where the eight B pieces of a request go (tools/build_variant.sh -DT5_B_WHOLE_KSTEP: over the whole k-step like A; default: its
  // first half -- B(g+2) is needed one stage later and comes from HBM like A)
#ifdef T5_B_WHOLE_KSTEP
#define T5_KS1_ROWS(BS)                                                                                        \
  do {                                                                                                         \
    T5_ROW(BS, 1, 0, 0, T5_NOP_, T5_NOP_, T5_REQ_B(), T5_PIECE_B(0), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);          \
    T5_ROW(BS, 1, 0, 1, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(1), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 1, 0, 2, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(2), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 1, 0, 3, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(3), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 1, 0, 4, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(4), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 1, 0, 5, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(5), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 1, 0, 6, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(6), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);             \
    T5_ROW(BS, 1, 0, 7, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(7), T5_NOP_, T5_DONE_B(), T5_NOP_, T5_NOP_);         \
  } while (0)
#else
#define T5_KS1_ROWS(BS)                                                                                        \
  do {                                                                                                         \
    T5_ROW(BS, 1, 0, 0, T5_NOP_, T5_NOP_, T5_REQ_B(), T5_PIECE_B(0), T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(1));    \
    T5_ROW(BS, 1, 0, 1, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(2), T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(3));       \
    T5_ROW(BS, 1, 0, 2, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(4), T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(5));       \
    T5_ROW(BS, 1, 0, 3, T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(6), T5_NOP_, T5_NOP_, T5_NOP_, T5_PIECE_B(7));       \
    T5_ROW(BS, 1, 0, 4, T5_NOP_, T5_NOP_, T5_NOP_, T5_DONE_B(), T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);               \
    T5_ROW(BS, 1, 0, 5, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);                   \
    T5_ROW(BS, 1, 0, 6, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);                   \
    T5_ROW(BS, 1, 0, 7, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_, T5_NOP_);                   \
  } while (0)
#endif
  // k-step 1: the stage barrier first (every fragment of the stage is in registers, stage g+1 has landed), B(g+2) in the first
  // half, reads of stage g+1's first k-step into set 0
#define T5_KS1(BS)                                                                                             \
  do {                                                                                                         \
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");                                   \
    const uint32_t oAn__ = oA + 32768u >= ldsB ? oA - 65536u : oA + 32768u;                                     \
    const uint32_t na__ = oAn__, nb__ = oBsum - oB;                                                            \
    T5_KS1_ROWS(BS);                                                                                           \
    oA = oAn__; oB = oBsum - oB;                                                                               \
  } while (0)

  (void)oA0;
  for (int st = 0; st < nst; ++st) {
    // (the bias sums sit behind a REAL branch -- the empty asm keeps hipcc from turning `if (do_bias)` into selects, which made
    // every stage pay 64 v_dot2 + 16 v_cndmask in gaps that hold three instructions for free; two copies of the stage body, one
    // branch per stage, made the allocator merge the accumulators of the two paths through VGPRs: 250-850 spilled registers)
    const bool do_bias = dbias != nullptr && (st % bias_mod) == bias_slot;
    T5_KS0(true);
    T5_KS1(true);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the trailing (dummy) stage loads land before the LDS is released

  // ---- ragged end of the token range (K % 64 != 0: the last split only): the last `tail` tokens as one more stage, loaded HERE with
  // clamped rows (the caller's buffers end at row K), rows >= tail zeroed in LDS (both operands: 0 x NaN would still be NaN), then
  // two plain k-steps.  One exposed memory latency in the workgroups of the last split; nothing in the main loop.
  if (tail) {
    asm volatile("s_barrier" ::: "memory");                         // every wave is done with the ring
    {
      const char* gA = reinterpret_cast<const char*>(A + (long)(kb + nst * 64) * lda_e);
      const char* gB = reinterpret_cast<const char*>(B + (long)(kb + nst * 64) * ldb);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int kr = wave * 16 + 2 * j + (lane >> 5);             // LDS row (its swizzle term is kr & 7); source row clamped into the range
        const int rk = kr < tail ? kr : tail - 1;
        const int p16 = lane & 31;
        const int col = (((p16 >> 1) ^ (kr & 7)) << 4) + (p16 & 1) * 8;
        P_DMA16(((uint32_t)(rk * lda_e) + tn2_acol(m0 + col, hm_a)) * 2, gA, lds0 + (uint32_t)(wave * 8192 + j * 1024));
        P_DMA16((uint32_t)(rk * ldb + n0 + col) * 2, gB, ldsB + (uint32_t)(wave * 8192 + j * 1024));
      }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      const int kr = idx >> 6, c16 = idx & 63;
      if (kr >= tail) *reinterpret_cast<uint4*>(smem + (c16 >> 5) * 98304 + kr * 512 + (c16 & 31) * 16) = make_uint4(0, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const bool do_bias = dbias != nullptr && (nst % bias_mod) == bias_slot;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const uint32_t ta = lds0 + (uint32_t)(wr * 256 + ks * 16384), tb = ldsB + (uint32_t)(wc * 256 + ks * 16384);
#pragma unroll
      for (int i = 0; i < 8; ++i) { T5_RDLO(fa, 0, i, ta); T5_RDHI(fa, 0, i); T5_RDLO(fb, 0, i, tb); T5_RDHI(fb, 0, i); }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) t5_mfma<F>(acc[i][j], T5_FR(fb, 0, j), T5_FR(fa, 0, i));
        if (do_bias) bsum[i] = t5_sum8<F>(T5_FR(fa, 0, i), bsum[i]);
      }
    }
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");               // (the stores below read the AGPRs the last MFMAs write)
  }

  // ---- epilogue: the f32 tile straight from the AGPRs (block (i, j): row 16i + fs, columns 16j + 4fg .. +3 = 16 B per lane)
  {
    float* o = out + (long)split * slab_stride + (long)(m0 + wr * 128) * ldo + n0 + wc * 128;
    int le;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
    const uint32_t eo = (uint32_t)((le & 15) * ldo + (le >> 4) * 4) * 4;
#define T5_ST(I, J, OFF) asm volatile("global_store_dwordx4 %0, %1, %2 offset:" #OFF ::"v"(eo), "a"(acc[I][J]), "s"(oi__) : "memory")
#define T5_STROW(I)                                                                                            \
  do {                                                                                                         \
    const float* oi__ = o + (long)(I) * 16 * ldo;                                                              \
    T5_ST(I, 0, 0); T5_ST(I, 1, 64); T5_ST(I, 2, 128); T5_ST(I, 3, 192); T5_ST(I, 4, 256); T5_ST(I, 5, 320);   \
    T5_ST(I, 6, 384); T5_ST(I, 7, 448);                                                                        \
  } while (0)
    T5_STROW(0); T5_STROW(1); T5_STROW(2); T5_STROW(3); T5_STROW(4); T5_STROW(5); T5_STROW(6); T5_STROW(7);
  }
  if (dbias != nullptr) {
    const float inv_b = gs_inv(gs);
    const int fs = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = bsum[i];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      if (fg == 0) atomicAdd(dbias + m0 + wr * 128 + i * 16 + fs, t * inv_b);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

static bool t5_attr_done[SIMX_MAX_DEVICES];
// Eligibility + launch of the slab kernel (the caller runs slab_reduce_kernel behind it); SIMX_ERR_UNSUPPORTED (no error text)
// when the shape is outside the rules above, so that the caller falls through to gemm_tn2_kernel.
int simx_launch_tn5(hipStream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* slabs,
                    int splits, int kps, float* dbias, int a_hm_rows, const float* gs) {
  if (!(simx_is16(dtype) && M % 256 == 0 && N % 256 == 0 && kps % 64 == 0 && splits >= 2 && kps >= 256 &&
        K - (splits - 1) * (long)kps >= 256))
    return SIMX_ERR_UNSUPPORTED;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIMX_MAX_DEVICES) return SIMX_ERR_UNSUPPORTED;
  if (!t5_attr_done[dev]) {
    bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn5_kernel<f16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS) == hipSuccess;
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn5_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, T5_LDS) == hipSuccess;
    if (!ok) return SIMX_ERR_UNSUPPORTED;
    t5_attr_done[dev] = true;
  }
  const int t_n = N / 256, t_mn = (M / 256) * t_n;
  if (dtype == SIMX_F16)
    hipLaunchKernelGGL(gemm_tn5_kernel<f16_t>, dim3(t_mn * splits), dim3(256), T5_LDS, s, M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                       slabs, (long)M * N, N, t_n, t_mn, kps, dbias, a_hm_rows, gs);
  else
    hipLaunchKernelGGL(gemm_tn5_kernel<bf16_t>, dim3(t_mn * splits), dim3(256), T5_LDS, s, M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                       slabs, (long)M * N, N, t_n, t_mn, kps, dbias, a_hm_rows, gs);
  SIMX_CHECK_LAUNCH("gemm_tn5");
  return SIMX_OK;
}
