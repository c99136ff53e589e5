// EXPERIMENT -- NOT BUILT INTO libsimx_hip.so.  Outcome (round 2, same box, tools/kbench, M = 262144 tokens): bit-correct
// (tests/test_kernels_gpu.py -k gemm: 145 passed with this kernel dispatched for the full-tile shapes) but NOT faster than
// tn2: 886 / 844 / 985 / 979 TFLOP/s (QKV / O / FFN-in / FFN-out wgrad; 944 avg) against tn2's 938 / 870 / 996 / 975 (962 avg).
// Fewer transpose reads per MFMA and one wave per SIMD do not move the wgrad kernel: like the NT kernel it sits at the
// ~1.0 PFLOP/s the chip sustains with the LDS + DMA stream drawing from the MFMA power budget.
// To try it again: copy to simxns_amd/csrc/, compile WITHOUT -mllvm -amdgpu-mfma-vgpr-form, call simx_launch_tn4 from
// simx_gemm_tn_bias for M % 256 == 0 && N % 256 == 0.
//
// bf16 TN kernel "tn4" (large wgrad):  slab[split][M][N] = A[kslice, M]^T . B[kslice, N]   (A = dY [T,out], B = X [T,in])
//
// Successor of gemm_tn2_bf16_kernel (csrc/gemm.hip) for the shapes that carry the wgrad time.  tn2 (8 waves, 128x64 wave
// tiles, two waves per SIMD) is bound by its fragment traffic: the contraction index (tokens) is the ROW index of both
// operands, so every fragment is two ds_read_b64_tr_b16 (hardware transpose reads) -- 24 of them per 32 MFMAs.  Here:
//   * 256 threads = 4 waves (2 x 2), ONE wave per SIMD, wave tile 128 x 128 = 8 x 8 MFMA 16x16x32 blocks: 32 transpose reads
//     per 64 MFMAs (a third fewer per MFMA), half the waves at every barrier;
//   * the 256 accumulator registers live in AGPRs (this file is built WITHOUT -amdgpu-mfma-vgpr-form; the MFMA is inline asm
//     with the accumulator tied in an AGPR -- with the builtin hipcc renames accumulators and moves them through VGPRs), so the
//     architectural VGPRs hold the fragments: A single-buffered (a row block's registers are refilled for the next k-step
//     right after its MFMAs), B double-buffered;
//   * the MFMA stream is compiler-scheduled between sched_barrier fences; the 16 LDS-DMA instructions of a stage are issued
//     ONE PER MFMA PAIR after the stage barrier, not as a burst (alone on its SIMD the wave would leave the matrix pipe idle
//     for the whole issue time of a burst: measured +8 % on the NT sibling of this loop, tools/experiments/gemm_p4.hip).
// Everything else is tn2's: 256x256 block tile, 64-token stages (two 64 KB stages, full 512-B rows by LDS-DMA, 32-B chunk q of
// k-row kr stored at q ^ (kr & 7)), split over tokens into f32 slabs reduced by slab_reduce_kernel, XCD-aware (split, tile)
// order, optional fused bias gradient (column sums of A over tokens on the VALU, shared out over the workgroups that stage the
// same A tile and their waves).  Full tiles only (M, N multiples of 256; the token range may be ragged).
#include "common.h"
#include "prof.h"

#define TN4_STAGE 65536
#define TN4_LDS (2 * TN4_STAGE)
#define TN4_DMA16(VOFF, SBASE, LDSADDR) \
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSADDR) : "memory")
#define TN4_SB __builtin_amdgcn_sched_barrier(0)
#define TN4_MFMA(ACC, BF, AF) asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(ACC) : "v"(BF), "v"(AF))
typedef __attribute__((address_space(3))) bf16x4* tn4_lds4_t;

__device__ __forceinline__ int tn4_xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

__global__ __launch_bounds__(256) void gemm_tn4_bf16_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    float* __restrict__ out, long slab_stride, int ldo, int tiles_n, int tiles_mn, int k_per_split, int accumulate,
    float* __restrict__ dbias) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int vb = tn4_xcd_remap(blockIdx.x, gridDim.x);
  const int split = vb / tiles_mn;
  const int tile = vb % tiles_mn;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int wr = wave >> 1, wc = wave & 1;
  const int kb = split * k_per_split;
  const int ke = min(K, kb + k_per_split);
  const int fs = lane & 15, fg = lane >> 4;
  const int nst = (ke - kb + 63) / 64;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // fused bias gradient: the column sums of A over this workgroup's k-range are shared out stage by stage over the tiles_n
  // workgroups that stage the same A tile and over their two wc waves (which hold the same A fragments)
  const int bias_slot = (tile % tiles_n) * 2 + wc, bias_mod = tiles_n * 2;

  // one DMA instruction = 2 k-rows x 512 B of an operand stage; this wave's 16 pieces of A and 16 of B per stage
  // (piece i of 32: k-rows 2i, 2i+1; lane -> row 2i + (lane >> 5), 16-B unit p16 = lane & 31 of the 512-B row, stored at
  //  32-B chunk (p16 >> 1) ^ (kr & 7))
  auto piece = [&](const bf16_t* __restrict__ G, int ld, int col0, int ncols, int k0, uint32_t stage, int i) {
    const int kr = i * 2 + (lane >> 5);
    const int p16 = lane & 31;
    const int q = (p16 >> 1) ^ (kr & 7);
    int rk = kr;
    rk = k0 + rk < ke ? rk : ke - 1 - k0;                       // clamp into the k-range (ragged tail rows are zeroed later)
    const int c = col0 + q * 16 + (p16 & 1) * 8;
    const char* g = reinterpret_cast<const char*>(G + (long)k0 * ld);
    TN4_DMA16((uint32_t)(rk * ld + c) * 2, g, stage + (uint32_t)(i * 1024));
  };
  // this wave issues pieces wave*8 .. wave*8+7 of each operand (32 pieces per operand per stage)
  auto issue_piece = [&](int j, int k0, uint32_t stage) {       // j 0..7: A, 8..15: B
    if (j < 8) piece(A, lda, m0, M, k0, stage, wave * 8 + j);
    else piece(B, ldb, n0, N, k0, stage + 32768u, wave * 8 + (j - 8));
  };

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  // transpose-read addressing: lane (fg, fs) supplies row 4*fg + (fs>>2) (+16 for the high half) of a 32-row k-step,
  // 8 B at element column ct*16 + (fs&3)*4 of 16-column tile ct -> 32-B chunk ct, byte (fs&3)*8 in it.
  const int r_lo = 4 * fg + (fs >> 2);
  const int x_lo = r_lo & 7, x_hi = (r_lo + 16) & 7;
  const uint32_t row_lo = (uint32_t)(r_lo * 512 + (fs & 3) * 8), row_hi = (uint32_t)((r_lo + 16) * 512 + (fs & 3) * 8);
#define TN4_FRAG(KBASE, OP, CT)                                                                                           \
  ([&]() {                                                                                                                \
    const bf16x4 lo__ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn4_lds4_t)(uintptr_t)((KBASE) + (OP) + row_lo + (uint32_t)(((CT) ^ x_lo) << 5))); \
    const bf16x4 hi__ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((tn4_lds4_t)(uintptr_t)((KBASE) + (OP) + row_hi + (uint32_t)(((CT) ^ x_hi) << 5))); \
    return (bf16x8){lo__[0], lo__[1], lo__[2], lo__[3], hi__[0], hi__[1], hi__[2], hi__[3]};                              \
  }())

#pragma unroll
  for (int j = 0; j < 16; ++j) issue_piece(j, kb, lds0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (nst > 1) {
#pragma unroll
    for (int j = 0; j < 16; ++j) issue_piece(j, kb + 64, lds0 + TN4_STAGE);
  }
  auto zero_tail = [&](int stage_idx, int valid) {              // rows >= valid of a ragged last stage hold clamped copies
    char* sp = smem + (stage_idx & 1) * TN4_STAGE;
    for (int idx = tid; idx < 64 * 64; idx += 256) {
      const int kr = idx >> 6, c16 = idx & 63;
      if (kr >= valid) *reinterpret_cast<uint4*>(sp + kr * 512 + (c16 & 31) * 16 + (c16 >> 5) * 32768) = make_uint4(0, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  };
  if (nst == 1 && (ke - kb) % 64 != 0) zero_tail(0, ke - kb);

  bf16x8 a[8], b[2][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = TN4_FRAG(lds0, 0u, wr * 8 + i);
    b[0][i] = TN4_FRAG(lds0, 32768u, wc * 8 + i);
  }

  for (int st = 0; st < nst; ++st) {
    const uint32_t sc = lds0 + (uint32_t)((st & 1) * TN4_STAGE), sn = lds0 + (uint32_t)(((st + 1) & 1) * TN4_STAGE);
    const bool do_bias = dbias != nullptr && (st % bias_mod) == bias_slot;
    // ---- k-step 0 (tokens 0-31 of the stage; B in b[0]); k-step 1's B streams into b[1], its A row blocks replace this
    // k-step's as they retire
    {
      const uint32_t k1 = sc + 32 * 512;
#pragma unroll
      for (int j = 0; j < 8; ++j) b[1][j] = TN4_FRAG(k1, 32768u, wc * 8 + j);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) TN4_MFMA(acc[i][j], b[0][j], a[i]);
        if (do_bias) {
#pragma unroll
          for (int e = 0; e < 8; ++e) bsum[i] += bf2f((bf16_t)a[i][e]);
        }
        TN4_SB;
        a[i] = TN4_FRAG(k1, 0u, wr * 8 + i);
        TN4_SB;
      }
    }
    // ---- k-step 1, first half
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) TN4_MFMA(acc[i][j], b[1][j], a[i]);
      if (do_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i] += bf2f((bf16_t)a[i][e]);
      }
    }
    // ---- stage boundary: every fragment of stage st is in registers; stage st+1 has landed once vmcnt hits 0 (for
    // everyone after the barrier); slot st&1 is refilled with stage st+2, one DMA piece per MFMA pair
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (st + 1 == nst - 1 && (ke - kb) % 64 != 0) zero_tail(st + 1, (ke - kb) - (nst - 1) * 64);
    const bool more = st + 2 < nst;
#pragma unroll
    for (int i = 4; i < 8; ++i) {
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        TN4_MFMA(acc[i][2 * jj], b[1][2 * jj], a[i]);
        TN4_MFMA(acc[i][2 * jj + 1], b[1][2 * jj + 1], a[i]);
        TN4_SB;
        const int q = (i - 4) * 4 + jj;                         // 0..15
        if (more) issue_piece(q, kb + (st + 2) * 64, sc);
        if (q < 4) a[q] = TN4_FRAG(sn, 0u, wr * 8 + q);          // (past the last stage: reads stale data nobody uses)
        else if (q < 12) b[0][q - 4] = TN4_FRAG(sn, 32768u, wc * 8 + (q - 4));
        TN4_SB;
      }
      if (do_bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum[i] += bf2f((bf16_t)a[i][e]);
      }
      TN4_SB;
      a[i] = TN4_FRAG(sn, 0u, wr * 8 + i);
      TN4_SB;
    }
  }

  float* o = out + (long)split * slab_stride;
  int le; asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(le));
  const int fsl = le & 15, fgl = le >> 4;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + fsl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      TN4_SB;
      const int n = n0 + wc * 128 + j * 16 + fgl * 4;
      float4* dst = reinterpret_cast<float4*>(o + (long)m * ldo + n);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (accumulate) { const float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
  if (dbias != nullptr) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = bsum[i];
      t += __shfl_xor(t, 16, 64);
      t += __shfl_xor(t, 32, 64);
      const int m = m0 + wr * 128 + i * 16 + fsl;
      if (fgl == 0) atomicAdd(dbias + m, t);
    }
  }
}

int simx_launch_tn4(hipStream_t s, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* out, long slab_stride,
                    int ldo, int splits, int k_per_split, int accumulate, float* dbias) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn4_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, TN4_LDS);
    attr = true;
  }
  const int t_n = N / 256, t_mn = (M / 256) * t_n;
  hipLaunchKernelGGL(gemm_tn4_bf16_kernel, dim3(t_mn * splits), dim3(256), TN4_LDS, s, M, N, K, (const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                     out, slab_stride, ldo, t_n, t_mn, k_per_split, accumulate, dbias);
  SIMX_CHECK_LAUNCH("gemm_tn4_bf16");
  return SIMX_OK;
}
