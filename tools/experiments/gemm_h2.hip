// EXPERIMENT (not built into the product).  Question (round-2 verdict item 2): do TWO half-size workgroups per CU, out of
// phase, hide the GELU epilogue of the FFN-in GEMM under each other's main loops?
//
// Result (one box, tools/kbench, fp16, M = 262144, N = 3072, K = 768, A/B/A/B against p3, the product kernel):
//     training form (two outputs):  p3 1.463-1.473 ms (840-845 TFLOP/s)   this kernel 1.860-1.862 ms (664)
//     inference form (one output):  p3 1.254-1.269 ms (975-986)           this kernel 1.573-1.580 ms (783-786)
//     main loop alone (epilogue compiled out, all accumulators live):      this kernel 1.237-1.266 ms (977-1000); p3: ~0.95 (1300)
// Bit-correct (tests/test_kernels_gpu.py -k "gemm_nt_persistent and gelu" with the kernel dispatched: 10 passed).
// Two things fail at once.  The half-size loop is 30 % slower than p3's: 32-deep stages mean one barrier per 32 MFMAs and
// 1.5x the LDS-DMA instructions per FLOP, and this version issues them as a burst and waits for all twelve fragment reads
// before its first MFMA.  And the epilogue is NOT hidden: it still adds 0.33 ms (inference) / 0.60 ms (training) on top of
// the loop although a second workgroup is resident.  Caveat on reading that: a loop that is latency-bound cannot speed up
// when its neighbour leaves for the epilogue, so "not hidden" here is first of all a statement about THIS loop.  What bounds
// it: not LDS latency -- issuing the fragment reads in consumption order and starting the MFMAs after five of twelve changed
// nothing (1.866 / 1.565 ms), a second fragment register set spills (728 B of scratch) -- but its operand stream: at 128 x 256
// a launch pulls 14.5 GB through the L2 (p3's 256 x 256 tiles: 9.7 GB) with a ring that holds two stages in flight (24 KB each;
// p3: 96 KB in flight), and a third stage in flight does not fit two workgroups into the LDS.  Not pursued.
//
// To rebuild: copy to simxns_amd/csrc/, add gemm_h2 to build.sh, declare simx_h2_ok / simx_h2_gemm_nt in common.h and call
// them from simx_gemm_nt for SIMX_EPI_GELU / _INFER (commit "gemm_h2: half-size ..." has the wiring).
//
// Half-size persistent NT GEMM for the GELU shapes: TWO independent 4-wave workgroups per CU instead of p3's one 8-wave
// workgroup (csrc/gemm.hip), so that one workgroup's epilogue -- two 16-bit outputs and the GELU arithmetic for the FFN-in
// forward, a multiply by the stored derivative for its dgrad: the instantiations of p3 whose matrix pipes are busy 46-50 %
// of the time -- runs under the other workgroup's main loop.  (Round-2 verdict item 2; measurements in DESIGN.md 5c.)
//
//   C[M,N] = A[M,K] . B[N,K]^T   16-bit operands (bf16 / fp16), f32 accumulate
//   u = acc + bias;  C2 = gelu(u),  C = gelu'(u)   (store_pre = 0: inference form, C is not written)
// (the GELU forward forms only: they carry the largest exposed epilogue, so they decide whether the structure pays)
//
// Workgroup tile 128 x 256, four waves of 128 x 64 (p3's wave tile: 8 x 4 MFMA 16x16x32 blocks, 128 accumulator VGPRs), one wave
// per SIMD and workgroup.  Half of p3's LDS per workgroup (80 KB) holds a 128 x 256 tile only at 32-deep stages: A 8 KB + B 16 KB
// per stage, a ring of three = 72 KB; rows are 64 B, 16-B chunk c of row r sits at c ^ 2((r >> 3) & 1) (the conflict-free
// form for ds_read_b128 over 64-B rows, see gemm_x3.hip), which the LDS-DMA realises by permuting its per-lane SOURCE chunk.
// One barrier per stage (32 MFMAs per wave between barriers, p3: 64), 6 DMA instructions per wave and stage (1.5x p3's per FLOP).
// The epilogue borrows the ring: wave w stages 16-row chunks through its own 4 KB and leaves with full-line 16-B stores.
#include "common.h"
#include "prof.h"

#define H2_DMA16(VOFF, SBASE, LDSADDR) \
  asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(VOFF), "s"(SBASE), "s"(LDSADDR) : "memory")
#define H2_STAGE 24576                         /* A 128 x 64 B | B 256 x 64 B */
#define H2_LDS (3 * H2_STAGE)

__device__ __forceinline__ int h2_xcd_remap(int b, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, loc = b >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
}

template <typename F>
__global__ __launch_bounds__(256, 2) void gemm_nt_h2_kernel(int M, int N, int K, const bf16_t* __restrict__ A, int lda,
                                                            const bf16_t* __restrict__ B, int ldb, bf16_t* __restrict__ C, int ldc,
                                                            const float* __restrict__ bias, bf16_t* __restrict__ C2, int ldc2, int tiles_n,
                                                            int ntiles, int store_pre) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int nst = K / 32;
  // DMA: a piece = 16 rows x 64 B; lane l lands on (row l >> 2, chunk position l & 3) and therefore fetches logical chunk
  // (l & 3) ^ 2 (l >> 5 & 1) of that row
  const int srcc = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  const uint32_t offA = (uint32_t)((lane >> 2) * lda + srcc * 8) * 2, offB = (uint32_t)((lane >> 2) * ldb + srcc * 8) * 2;
  // fragment reads: row (block * 16 + fr), logical chunk fg
  const uint32_t frag = (uint32_t)(fr * 64 + ((fg ^ (((fr >> 3) & 1) << 1)) << 4));
  const uint32_t fragB = frag + 8192u + (uint32_t)(wave * 64 * 64);

  for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
    const int tile = h2_xcd_remap(v, ntiles);
    const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 256;
    const char* gA = reinterpret_cast<const char*>(A + (long)(m0 + wave * 32) * lda);      // this wave's 2 A pieces: rows wave*32 ..
    const char* gB = reinterpret_cast<const char*>(B + (long)(n0 + wave * 64) * ldb);      // and 4 B pieces: rows wave*64 ..
    auto issue = [&](int st) {
      const uint32_t slot = lds0 + (uint32_t)((st % 3) * H2_STAGE);
      const long k0 = (long)st * 64;                                                       // 32 elements = 64 B
      H2_DMA16(offA, gA + k0, slot + (uint32_t)(wave * 2048));
      H2_DMA16(offA, gA + k0 + (long)16 * lda * 2, slot + (uint32_t)(wave * 2048 + 1024));
#pragma unroll
      for (int q = 0; q < 4; ++q)
        H2_DMA16(offB, gB + k0 + (long)q * 16 * ldb * 2, slot + 8192u + (uint32_t)(wave * 4096 + q * 1024));
    };
    issue(0);
    issue(1);
    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < nst; ++st) {
      // this wave's pieces of stage st have landed when at most the 6 of stage st+1 are outstanding
      if (st + 1 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                              // ... and everybody's; everybody is also done with stage st-1
      if (st + 2 < nst) issue(st + 2);              // into the slot stage st-1 occupied
      const uint32_t slot = lds0 + (uint32_t)((st % 3) * H2_STAGE);
      bf16x8 af[8], bfr[4];
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1" : "=v"(af[i]) : "v"(slot + frag + (uint32_t)(i * 1024)) : "memory");
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("ds_read_b128 %0, %1" : "=v"(bfr[j]) : "v"(slot + fragB + (uint32_t)(j * 1024)) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]), "+v"(af[6]), "+v"(af[7]), "+v"(bfr[0]),
                     "+v"(bfr[1]), "+v"(bfr[2]), "+v"(bfr[3])::"memory");
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = H16<F>::mfma(bfr[j], af[i], acc[i][j]);
    }
    __syncthreads();                                // the ring is free: every wave has read the last stage

    // ---- epilogue: lane holds rows i*16 + fr, columns wave*64 + j*16 + fg*4 .. +3.  16-row chunk i goes through this wave's
    // 4 KB of the ring (two 2 KB images for the GELU pair: rows of 64 columns = 128 B, 16-B chunk c of row r at c ^ (r >> 1 & 7))
    // and leaves as 16 B per lane: lane l owns chunk (l & 7) of rows (l >> 3), (l >> 3) + 8 -- full 128-B lines.
    char* ep = smem + wave * 4096;
    const int nw = n0 + wave * 64;
    float4 bv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bv[j] = bias ? *reinterpret_cast<const float4*>(bias + nw + j * 16 + fg * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int er = lane >> 3, ech = lane & 7;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      asm volatile("" ::: "memory");
      const int sw = (fr >> 1) & 7;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        char* d = ep + fr * 128 + (((j * 2 + (fg >> 1)) ^ sw) << 4) + (fg & 1) * 8;
        float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
        vv[0] += bv[j].x; vv[1] += bv[j].y; vv[2] += bv[j].z; vv[3] += bv[j].w;
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          f32p_t gp, dp;
          gelu_both_fast2((f32p_t){vv[e], vv[e + 1]}, gp, dp);
          g[e] = gp.x; g[e + 1] = gp.y; vv[e] = dp.x; vv[e + 1] = dp.y;
        }
        *reinterpret_cast<uint2*>(d + 2048) = make_uint2(H16<F>::pack2(g[0], g[1]), H16<F>::pack2(g[2], g[3]));
        if (store_pre) *reinterpret_cast<uint2*>(d) = make_uint2(H16<F>::pack2(vv[0], vv[1]), H16<F>::pack2(vv[2], vv[3]));
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int r = it * 8 + er;
        const int m = m0 + i * 16 + r;
        const int pos = (ech ^ ((r >> 1) & 7)) << 4;
        const uint4 w2 = *reinterpret_cast<const uint4*>(ep + 2048 + r * 128 + pos);
        *reinterpret_cast<uint4*>(C2 + (long)m * ldc2 + nw + ech * 8) = w2;
        if (store_pre) {
          const uint4 w1 = *reinterpret_cast<const uint4*>(ep + r * 128 + pos);
          *reinterpret_cast<uint4*>(C + (long)m * ldc + nw + ech * 8) = w1;
        }
      }
    }
    __syncthreads();                                // the next tile's DMA may overwrite the staging regions
  }
}

bool simx_h2_ok(int M, int N, int K, int epilogue, const void* residual, int lda, int ldb, int ldc, int ldc2) {
  const bool epi = (epilogue == SIMX_EPI_GELU || epilogue == SIMX_EPI_GELU_INFER) && !residual;
  return epi && M % 128 == 0 && N % 256 == 0 && K % 32 == 0 && K >= 64 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && ldc2 % 8 == 0 &&
         (long)(M / 128) * (N / 256) >= 512;
}

int simx_h2_gemm_nt(hipStream_t s, int dtype, int epilogue, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                    const float* bias, void* C2, int ldc2, int ncu) {
  const int tiles_n = N / 256, ntiles = (M / 128) * tiles_n;
  const int grid = ntiles < 2 * ncu ? ntiles : 2 * ncu;
  const int store_pre = epilogue == SIMX_EPI_GELU ? 1 : 0;
#define LH2(FF)                                                                                                                         \
  do {                                                                                                                                  \
    static bool attr_done = false;                                                                                                      \
    if (!attr_done) {                                                                                                                   \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_h2_kernel<FF>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                              H2_LDS) != hipSuccess) { simx_set_error("gemm_nt_h2: cannot raise the dynamic LDS limit"); return SIMX_ERR_HIP; } \
      attr_done = true;                                                                                                                 \
    }                                                                                                                                   \
    hipLaunchKernelGGL((gemm_nt_h2_kernel<FF>), dim3(grid), dim3(256), H2_LDS, s, M, N, K, (const bf16_t*)A, lda,        \
                       (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, (bf16_t*)C2, ldc2, tiles_n, ntiles, store_pre);                   \
                                                                                                      \
  } while (0)
  if (dtype == SIMX_F16) LH2(f16_t); else LH2(bf16_t);
#undef LH2
  SIMX_CHECK_LAUNCH("gemm_nt_h2");
  return SIMX_OK;
}
