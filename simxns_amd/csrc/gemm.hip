// GEMMs of the BERT dense layers for gfx950.
//
//   simx_gemm_nt : C[M,N] = A[M,K] . B[N,K]^T      forward (x W^T) and dgrad (dY . (W^T)^T)
//   simx_gemm_tn : C[M,N] += A[K,M]^T . B[K,N]     wgrad (dY^T X), split over K (tokens)
//
// bf16 path: 128x128x64 block tile, 4 waves (2x2) of 64x64, v_mfma_f32_16x16x32_bf16.
//   * operands staged HBM -> LDS with global_load_lds (16 B per lane, lane-linear LDS image),
//     bank conflicts removed by an XOR swizzle applied on the per-lane SOURCE address and
//     again on the ds_read address (LDS destination stays linear);
//   * NT: K-contiguous rows, fragments by ds_read_b128;
//   * TN: the contraction index (tokens) is the row index of both operands, fragments by
//     ds_read_b64_tr_b16 (hardware transpose read); the k-slot permutation this implies is
//     applied identically to both operands so the contraction is unchanged;
//   * the MFMA is issued with swapped operands so that each lane ends up with 4 CONSECUTIVE
//     output columns of one row -> 8 B (bf16) / 16 B (f32) stores;
//   * blockIdx -> tile map is XCD-aware: the 8 XCDs (private L2s) each walk a contiguous
//     range of tiles so that the N-tiles sharing an A panel hit the same L2.
// f32 path ("parity mode") and odd shapes: a plain LDS-tiled FMA kernel, k-ordered f32 accumulate.
#include <mutex>
#include <type_traits>
#include "common.h"
#include "prof.h"
#include "p3.h"

// ------------------------------------------------------------------------------------------
// generic tiled kernel (f32 parity mode, and bf16 shapes the MFMA kernels do not take)
// ------------------------------------------------------------------------------------------
template <typename TI, typename TO, int EPI>
__global__ __launch_bounds__(256) void gemm_simple_kernel(
    int M, int N, int K, const TI* __restrict__ A, long a_rs, long a_cs,
    const TI* __restrict__ B, long b_ks, long b_ns, TO* __restrict__ C, int ldc,
    const float* __restrict__ bias, const TI* __restrict__ res, int ldr,
    const TI* __restrict__ aux, int ldaux, TO* __restrict__ C2, int ldc2, int accumulate, DropCtx drop,
    const float* __restrict__ gs = nullptr) {           // gs: the product is a parameter gradient of the fp16 backward (x 1/S)
  __shared__ float As[16][68];
  __shared__ float Bs[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  float acc[4][4] = {};
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int idx = tid + e * 256;
      int mm, kk;
      if (a_cs == 1) { kk = idx & 15; mm = idx >> 4; } else { mm = idx & 63; kk = idx >> 6; }
      int gm = m0 + mm, gk = k0 + kk;
      As[kk][mm] = (gm < M && gk < K) ? Elem<TI>::ld(A + (long)gm * a_rs + (long)gk * a_cs) : 0.f;
      int nn;
      if (b_ks == 1) { kk = idx & 15; nn = idx >> 4; } else { nn = idx & 63; kk = idx >> 6; }
      int gn = n0 + nn; gk = k0 + kk;
      Bs[kk][nn] = (gn < N && gk < K) ? Elem<TI>::ld(B + (long)gk * b_ks + (long)gn * b_ns) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; b[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int m = m0 + ty * 4 + i;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int n = n0 + tx * 4 + j;
      if (n >= N) continue;
      float v = acc[i][j] * gs_inv(gs);
      if (bias) v += bias[n];
      if (EPI == SIMX_EPI_NONE) {
        if (drop.thr) v *= drop_mult(drop, (uint32_t)m, (uint32_t)n);
        if (res) v += Elem<TI>::ld(res + (long)m * ldr + n);
        if (accumulate) v += Elem<TO>::ld(C + (long)m * ldc + n);
        Elem<TO>::st(C + (long)m * ldc + n, v);
      } else if (EPI == SIMX_EPI_GELU) {
        Elem<TO>::st(C + (long)m * ldc + n, gelu_erf_grad(v));          // C = gelu'(u): what backward multiplies by
        Elem<TO>::st(C2 + (long)m * ldc2 + n, gelu_erf(v));
      } else {
        if (res) v += Elem<TI>::ld(res + (long)m * ldr + n);
        Elem<TO>::st(C + (long)m * ldc + n, v * Elem<TI>::ld(aux + (long)m * ldaux + n));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// f32 MFMA kernel (parity mode NT / TN GEMMs and the M2 score matrix): v_mfma_f32_32x32x2_f32 -- exact f32 products,
// f32 accumulation in ascending k (the instruction adds its two k terms in order into the accumulator), 157 TFLOP/s peak,
// the same rate as the vector FMA pipe but with 1/16 of the instruction issue and LDS traffic of the FMA kernel above.
// 128x128x16 block tile, 4 waves (2x2) of 64x64 = 2x2 MFMA tiles; operands staged global -> registers -> LDS in k-major
// images As[k][m], Bs[k][n] (row pitch 132 floats) so that a fragment read (lane -> m = lane%32, k = lane/32) is one
// conflict-free ds_read_b32; the next stage's global loads are issued before the current stage's MFMAs.  Each operand is
// read with 16-B loads along whichever of its two dimensions is contiguous (element stride 1); anything else (or a
// misaligned base / leading dimension) takes the scalar path of the same kernel.
// ------------------------------------------------------------------------------------------
#define FM_PITCH 132
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_kernel(
    int M, int N, int K, const float* __restrict__ A, long a_rs, long a_cs, const float* __restrict__ B, long b_ks, long b_ns,
    float* __restrict__ C, int ldc, const float* __restrict__ bias, const float* __restrict__ res, int ldr,
    const float* __restrict__ aux, int ldaux, float* __restrict__ C2, int ldc2, int accumulate, DropCtx drop, int a_mode, int b_mode,
    int k_per_split, long slab_stride) {
  // mode 0: scalar loads; 1: 16-B loads along k (k contiguous); 2: 16-B loads along m / n (m / n contiguous)
  // split-K (k_per_split > 0, plain epilogue only): block z contracts k in [z*kps, (z+1)*kps) into slab z of C (= the
  // workspace); slab_reduce_kernel adds the slabs in index order, so the result does not depend on scheduling
  const int k_lo = k_per_split > 0 ? (int)blockIdx.z * k_per_split : 0;
  if (k_per_split > 0) { K = min(K, k_lo + k_per_split); C += (long)blockIdx.z * slab_stride; }
  __shared__ __attribute__((aligned(16))) float As[2][16 * FM_PITCH];
  __shared__ __attribute__((aligned(16))) float Bs[2][16 * FM_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = blockIdx.y * 128, n0 = blockIdx.x * 128;
  const int wr = wave >> 1, wc = wave & 1;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float ra[8], rb[8];            // this thread's share of a 128x16 operand stage (2 x 4 floats)
  // element (mm, kk) of the stage held in r[e*4 + j]:
  //   mode 1 / scalar: unit u = tid + e*256 -> mm = u >> 2, kk = (u & 3)*4 + j      (4 consecutive k of one row)
  //   mode 2         : unit u = tid + e*256 -> kk = u >> 5, mm = (u & 31)*4 + j     (4 consecutive m of one k)
  auto load_stage = [&](const float* __restrict__ G, long rs, long cs, int row0, int nrows, int k0, int mode, float* r) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int u = tid + e * 256;
      if (mode == 2) {
        const int kk = u >> 5, mm = (u & 31) * 4;
        const int gk = k0 + kk, gm = row0 + mm;
        if (gk < K && gm + 3 < nrows) {
          const float4 v = *reinterpret_cast<const float4*>(G + (long)gk * cs + gm);
          r[e * 4 + 0] = v.x; r[e * 4 + 1] = v.y; r[e * 4 + 2] = v.z; r[e * 4 + 3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) r[e * 4 + j] = (gk < K && gm + j < nrows) ? G[(long)gk * cs + (long)(gm + j) * rs] : 0.f;
        }
      } else {
        const int mm = u >> 2, kk = (u & 3) * 4;
        const int gm = row0 + mm, gk = k0 + kk;
        if (mode == 1 && gm < nrows && gk + 3 < K) {
          const float4 v = *reinterpret_cast<const float4*>(G + (long)gm * rs + gk);
          r[e * 4 + 0] = v.x; r[e * 4 + 1] = v.y; r[e * 4 + 2] = v.z; r[e * 4 + 3] = v.w;
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) r[e * 4 + j] = (gm < nrows && gk + j < K) ? G[(long)gm * rs + (long)(gk + j) * cs] : 0.f;
        }
      }
    }
  };
  auto store_stage = [&](float* __restrict__ S, int mode, const float* r) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int u = tid + e * 256;
      if (mode == 2) {
        const int kk = u >> 5, mm = (u & 31) * 4;
        *reinterpret_cast<float4*>(S + kk * FM_PITCH + mm) = make_float4(r[e * 4], r[e * 4 + 1], r[e * 4 + 2], r[e * 4 + 3]);
      } else {
        const int mm = u >> 2, kk = (u & 3) * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) S[(kk + j) * FM_PITCH + mm] = r[e * 4 + j];
      }
    }
  };
  // B is addressed as B[k*b_ks + n*b_ns]: "row" index = n with stride b_ns, k stride b_ks
  load_stage(A, a_rs, a_cs, m0, M, k_lo, a_mode, ra);
  load_stage(B, b_ns, b_ks, n0, N, k_lo, b_mode, rb);
  store_stage(As[0], a_mode, ra);
  store_stage(Bs[0], b_mode, rb);
  __syncthreads();
  const int nst = (K - k_lo + 15) / 16;
  const int fm = lane & 31, fk = lane >> 5;
  for (int st = 0; st < nst; ++st) {
    const int cur = st & 1;
    if (st + 1 < nst) {
      load_stage(A, a_rs, a_cs, m0, M, k_lo + (st + 1) * 16, a_mode, ra);
      load_stage(B, b_ns, b_ks, n0, N, k_lo + (st + 1) * 16, b_mode, rb);
    }
    const float* sa = As[cur] + wr * 64 + fm;
    const float* sb = Bs[cur] + wc * 64 + fm;
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      const float a0 = sa[(kk + fk) * FM_PITCH], a1 = sa[(kk + fk) * FM_PITCH + 32];
      const float b0 = sb[(kk + fk) * FM_PITCH], b1 = sb[(kk + fk) * FM_PITCH + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (st + 1 < nst) {
      store_stage(As[cur ^ 1], a_mode, ra);
      store_stage(Bs[cur ^ 1], b_mode, rb);
    }
    __syncthreads();
  }
  // D layout of the 32x32 tile: lane -> column n = lane%32; acc[e] -> row 8*(e/4) + 4*(lane/32) + e%4
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + wc * 64 + j * 32 + fm;
      if (n >= N) continue;
      const float bv = bias ? bias[n] : 0.f;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int m = m0 + wr * 64 + i * 32 + 8 * (e >> 2) + 4 * fk + (e & 3);
        if (m >= M) continue;
        float v = acc[i][j][e] + bv;
        if (EPI == SIMX_EPI_NONE) {
          if (drop.thr) v *= drop_mult(drop, (uint32_t)m, (uint32_t)n);
          if (res) v += res[(long)m * ldr + n];
          if (accumulate) v += C[(long)m * ldc + n];
          C[(long)m * ldc + n] = v;
        } else if (EPI == SIMX_EPI_GELU) {
          C[(long)m * ldc + n] = gelu_erf_grad(v);
          C2[(long)m * ldc2 + n] = gelu_erf(v);
        } else {
          if (res) v += res[(long)m * ldr + n];
          C[(long)m * ldc + n] = v * aux[(long)m * ldaux + n];
        }
      }
    }
}


// ------------------------------------------------------------------------------------------
// bf16 NT kernel
// ------------------------------------------------------------------------------------------
#define NT_BM 128
#define NT_BN 128
#define NT_BK 64
#define NT_STAGE_BYTES (2 * 128 * 64 * 2)   // A tile + B tile

// stage one 128x64 bf16 tile (rows K-contiguous).  LDS image: row r at r*128 B, the 16-B chunk
// c of the row stored at chunk position c ^ ((r>>1)&7).
__device__ __forceinline__ void nt_stage_tile(const bf16_t* __restrict__ G, int ld, int row0, int nrows_total,
                                              int k0, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave * 4 + j;                 // wave-instruction index: 8 rows each
    const int r = i * 8 + (lane >> 3);
    const int p = lane & 7;
    const int c = p ^ ((r >> 1) & 7);
    int gr = row0 + r;
    gr = gr < nrows_total ? gr : nrows_total - 1;
    glds16(G + (long)gr * ld + k0 + c * 8, lds_tile + i * 1024);
  }
}

template <typename F, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_h16_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    bf16_t* __restrict__ C, int ldc, const float* __restrict__ bias, const bf16_t* __restrict__ res, int ldr,
    const bf16_t* __restrict__ aux, int ldaux, bf16_t* __restrict__ C2, int ldc2, int tiles_n, int nwg, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int m0 = (tile / tiles_n) * NT_BM, n0 = (tile % tiles_n) * NT_BN;
  const int wr = wave >> 1, wc = wave & 1;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = K / NT_BK;
  nt_stage_tile(A, lda, m0, M, 0, smem, wave, lane);
  nt_stage_tile(B, ldb, n0, N, 0, smem + 16384, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fr = lane & 15, fg = lane >> 4;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    char* sa = smem + cur * NT_STAGE_BYTES;
    char* sb = sa + 16384;
    if (t + 1 < nt) {
      char* na = smem + (cur ^ 1) * NT_STAGE_BYTES;
      nt_stage_tile(A, lda, m0, M, (t + 1) * NT_BK, na, wave, lane);
      nt_stage_tile(B, ldb, n0, N, (t + 1) * NT_BK, na + 16384, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ra = wr * 64 + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8*>(sa + ra * 128 + (((ks * 4 + fg) ^ ((ra >> 1) & 7)) << 4));
        const int rb = wc * 64 + i * 16 + fr;
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + rb * 128 + (((ks * 4 + fg) ^ ((rb >> 1) & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = H16<F>::mfma(bfr[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  // epilogue: lane holds row m = ..+fr, columns n = ..+fg*4 .. +3 of each 16x16 tile
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fg * 4;
      if (n >= N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (bias) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (EPI == SIMX_EPI_NONE) {
        if (drop.thr) { float m4[4]; drop_mult4(drop, (uint32_t)m, (uint32_t)n, m4); v[0] *= m4[0]; v[1] *= m4[1]; v[2] *= m4[2]; v[3] *= m4[3]; }
        if (res) { float r4[4]; ld4h<F>(res + (long)m * ldr + n, r4); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
        st4h<F>(C + (long)m * ldc + n, v);
      } else if (EPI == SIMX_EPI_GELU) {
        // C = gelu'(u) (what backward multiplies by), C2 = gelu(u), both from the f32 pre-activation
        float g4[4], d4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) gelu_both_fast(v[e], g4[e], d4[e]);
        st4h<F>(C + (long)m * ldc + n, d4);
        st4h<F>(C2 + (long)m * ldc2 + n, g4);
      } else {
        if (res) { float r4[4]; ld4h<F>(res + (long)m * ldr + n, r4); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
        float u4[4]; ld4h<F>(aux + (long)m * ldaux + n, u4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= u4[e];
        st4h<F>(C + (long)m * ldc + n, v);
      }
    }
  }
}


// v5: 256x256x64 stages (full 128-B lines per row: the LDS-DMA path is request-bound, measured ~20 B/clk/CU with full
// lines on v2 and only ~13 with v3's half lines), TWO 64 KB stages, 128x64 wave tiles.  A stage is re-filled as soon as
// its last fragment has been read (stage boundary = the only barrier, once per 64 k), so the DMA queue never drains.
#define V5_LDS (2 * V5_STAGE)
__device__ __forceinline__ void v5_stage(const bf16_t* __restrict__ A, int lda, int m0, int M,
                                         const bf16_t* __restrict__ B, int ldb, int n0, int N, int k0,
                                         char* stage, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {                 // 32 wave-instructions of 8 rows per operand
    const int i = wave * 4 + j;
    const int r = i * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    int ga = m0 + r, gb = n0 + r;
    ga = ga < M ? ga : M - 1;
    gb = gb < N ? gb : N - 1;
    glds16(A + (long)ga * lda + k0 + c * 8, stage + i * 1024);
    glds16(B + (long)gb * ldb + k0 + c * 8, stage + 32768 + i * 1024);
  }
}

template <typename F, int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_h16_v5_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    bf16_t* __restrict__ C, int ldc, const float* __restrict__ bias, const bf16_t* __restrict__ res, int ldr,
    const bf16_t* __restrict__ aux, int ldaux, bf16_t* __restrict__ C2, int ldc2, int tiles_n, int nwg, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int wr = wave >> 2, wc = wave & 3;       // wave tile: rows wr*128.., cols wc*64..
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nst = K / 64;                         // 64-deep stages; two 32-deep k-steps ("slabs") per stage
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int swz = (fr >> 1) & 7;
  const uint32_t rowA = (uint32_t)((wr * 128 + fr) * 128), rowB = (uint32_t)(32768 + (wc * 64 + fr) * 128);
  // fragment byte offsets inside a stage for k-step 0 / 1
  const uint32_t oA0 = rowA + (uint32_t)(((0 + fg) ^ swz) << 4), oA1 = rowA + (uint32_t)(((4 + fg) ^ swz) << 4);
  const uint32_t oB0 = rowB + (uint32_t)(((0 + fg) ^ swz) << 4), oB1 = rowB + (uint32_t)(((4 + fg) ^ swz) << 4);

  v5_stage(A, lda, m0, M, B, ldb, n0, N, 0, smem, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (nst > 1) v5_stage(A, lda, m0, M, B, ldb, n0, N, 64, smem + V5_STAGE, wave, lane);

  bf16x8 al0, al1, al2, al3, ah0, ah1, ah2, ah3, bx0, bx1, bx2, bx3, by0, by1, by2, by3;
  {
    const uint32_t aa = lds0 + oA0, ab = lds0 + oB0;
    V3_READ4(al0, al1, al2, al3, aa, 0, 2048, 4096, 6144);
    V3_READ4(bx0, bx1, bx2, bx3, ab, 0, 2048, 4096, 6144);
    V3_PIN8("s_waitcnt lgkmcnt(0)", al0, al1, al2, al3, bx0, bx1, bx2, bx3);
  }

#define V3_MFMA_ROW(I, AF, B0, B1, B2, B3)                                                     \
  acc[I][0] = H16<F>::mfma(B0, AF, acc[I][0]);             \
  acc[I][1] = H16<F>::mfma(B1, AF, acc[I][1]);             \
  acc[I][2] = H16<F>::mfma(B2, AF, acc[I][2]);             \
  acc[I][3] = H16<F>::mfma(B3, AF, acc[I][3])
#define V3_RD1(F, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=&v"(F) : "v"(ADDR) : "memory")
#define V3_SB __builtin_amdgcn_sched_barrier(0)
  // one k-step: CUR = this k-step's A-high address, NA/NB = next k-step's A-low / B addresses; SYNC: stage boundary
#define V5_STEP(CURA, NA, NB, BC0, BC1, BC2, BC3, BN0, BN1, BN2, BN3, SYNC, ST)                  \
  do {                                                                                         \
    const uint32_t aa__ = (CURA), na__ = (NA), nb__ = (NB);                                    \
    V3_SB; V3_MFMA_ROW(0, al0, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah0, aa__, 8192);            \
    V3_SB; V3_MFMA_ROW(1, al1, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah1, aa__, 10240);           \
    V3_SB; V3_MFMA_ROW(2, al2, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah2, aa__, 12288);           \
    V3_SB; V3_MFMA_ROW(3, al3, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah3, aa__, 14336);           \
    V3_SB;                                                                                     \
    V3_PIN4("s_waitcnt lgkmcnt(0)", ah0, ah1, ah2, ah3);                                       \
    if (SYNC) {                                                                                \
      /* every read of stage ST is done (pinned above); the next stage has landed for this wave  \
         once vmcnt hits 0, for everyone after the barrier; stage ST is then free for ST+2 */    \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                            \
      if ((ST) + 2 < nst)                                                                      \
        v5_stage(A, lda, m0, M, B, ldb, n0, N, ((ST) + 2) * 64, smem + ((ST) & 1) * V5_STAGE, wave, lane); \
    }                                                                                          \
    V3_SB; V3_MFMA_ROW(4, ah0, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al0, na__, 0); V3_RD1(BN0, nb__, 0);       \
    V3_SB; V3_MFMA_ROW(5, ah1, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al1, na__, 2048); V3_RD1(BN1, nb__, 2048); \
    V3_SB; V3_MFMA_ROW(6, ah2, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al2, na__, 4096); V3_RD1(BN2, nb__, 4096); \
    V3_SB; V3_MFMA_ROW(7, ah3, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al3, na__, 6144); V3_RD1(BN3, nb__, 6144); \
    V3_SB;                                                                                     \
    V3_PIN8("s_waitcnt lgkmcnt(0)", al0, al1, al2, al3, BN0, BN1, BN2, BN3);                   \
  } while (0)

  for (int st = 0; st < nst; ++st) {
    const uint32_t sc = lds0 + (uint32_t)((st & 1) * V5_STAGE), sn = lds0 + (uint32_t)(((st + 1) & 1) * V5_STAGE);
    // k-step 0 of stage st (B in bx*), prefetching k-step 1 of the same stage into by*
    V5_STEP(sc + oA0, sc + oA1, sc + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, false, st);
    // k-step 1 (B in by*), stage boundary inside, prefetching k-step 0 of stage st+1 into bx* (stale past the end)
    V5_STEP(sc + oA1, sn + oA0, sn + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, true, st);
  }
#undef V5_STEP
#undef V3_RD1
#undef V3_SB
#undef V3_MFMA_ROW

  if (C == nullptr) return;                      // (main-loop-only measurement build, -DSIMX_MEASUREMENT_HOOKS)
  // ---- epilogue through LDS.  The memory path is REQUEST-bound (~1 request / 6 clk / CU, measured): the natural
  // MFMA-layout epilogue issues 32-B requests (16 rows x 32 B per store instruction, 4096 per output tile, as many
  // as a K=768 main loop).  Each wave therefore stages its 128x64 tile in its own 16 KB of the (now free) LDS ring:
  // residual / GELU-input rows arrive by full-line LDS-DMA, results leave as 16 B per lane = full 128-B lines.
  asm volatile("s_barrier" ::: "memory");        // every wave is done with the operand stages
  {
    char* reg = smem + wave * 16384;             // [128 rows][128 B], 16-B chunk c of row r at c ^ ((r>>1)&7)
    const uint32_t reg_a = lds0 + (uint32_t)(wave * 16384);
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    const bf16_t* in = (EPI == SIMX_EPI_DGELU) ? aux : res;
    const int ldin = (EPI == SIMX_EPI_DGELU) ? ldaux : ldr;
    if (in) {
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = it * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        int gm = mw + r, gn = nw + c * 8;
        gm = gm < M ? gm : M - 1;
        gn = gn + 8 <= N ? gn : N - 8;
        glds16(in + (long)gm * ldin + gn, reg + it * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // lane's 8-B slot for (i,j): row i*16+fr, cols j*16+fg*4..+3
    const uint32_t slot0 = reg_a + (uint32_t)(fr * 128 + (fg & 1) * 8);
    const int sw = (fr >> 1) & 7;
#pragma unroll
    for (int pass = 0; pass < (EPI == SIMX_EPI_GELU ? 2 : 1); ++pass) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t ad = slot0 + (uint32_t)(i * 2048 + (((j * 2 + (fg >> 1)) ^ sw) << 4));
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          if (bias) {
            const int n = nw + j * 16 + fg * 4;
            const float4 bv = *reinterpret_cast<const float4*>(bias + (n + 4 <= N ? n : 0));
            v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
          }
          if (EPI == SIMX_EPI_NONE && drop.thr) {
            float m4[4];
            drop_mult4(drop, (uint32_t)(mw + i * 16 + fr), (uint32_t)(nw + j * 16 + fg * 4), m4);
            v[0] *= m4[0]; v[1] *= m4[1]; v[2] *= m4[2]; v[3] *= m4[3];
          }
          if (EPI != SIMX_EPI_GELU && in) {
            uint2 t;
            asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(ad) : "memory");
            const float x0 = H16<F>::lo(t.x), x1 = H16<F>::hi(t.x);
            const float x2 = H16<F>::lo(t.y), x3 = H16<F>::hi(t.y);
            if (EPI == SIMX_EPI_NONE) { v[0] += x0; v[1] += x1; v[2] += x2; v[3] += x3; }
            else { v[0] *= x0; v[1] *= x1; v[2] *= x2; v[3] *= x3; }
          }
          if (EPI == SIMX_EPI_GELU) {                 // pass 0: C = gelu'(u), pass 1: C2 = gelu(u)
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) v[e2] = pass == 0 ? gelu_grad_fast(v[e2]) : gelu_fast(v[e2]);
          }
          const uint2 o = make_uint2(H16<F>::pack2(v[0], v[1]), H16<F>::pack2(v[2], v[3]));
          asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(o) : "memory");
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      bf16_t* out = pass == 0 ? C : C2;
      const int ldo = pass == 0 ? ldc : ldc2;
#pragma unroll
      for (int it = 0; it < 16; ++it) {
        const int r = it * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const uint4 val = *reinterpret_cast<const uint4*>(reg + r * 128 + (lane & 7) * 16);
        const int gm = mw + r, gn = nw + c * 8;
        if (gm < M && gn + 8 <= N) *reinterpret_cast<uint4*>(out + (long)gm * ldo + gn) = val;
      }
    }
  }
}



// HMF: which operands are PLANE-BLOCKED, i.e. a [rows, cols] tensor stored as [cols/64][R][64] (the head-major q/k/v of
// attention.hip is the 64 = head-size case; the FFN's [T, 3072] tensors use the same form).  `ldc2` carries R, the rows of
// a plane; a plane-blocked operand's own leading dimension is 64.
//   bit 0: A          -- a 64-wide K stage is one plane: a contiguous 256 x 128 B block instead of 256 rows K*2 B apart;
//   bit 1: C (and C2) -- a wave's 64 output columns are exactly one plane: its 16-row chunks are contiguous 2 KB stores
//                        instead of 16 pieces of 128 B that sit ldc*2 B apart.  For the QKV projection that is 0.95 -> 0.78 ms,
//                        but the gain is specific to its row pitch: 2304 columns = 4608 B (4096 + 512) between the rows of a
//                        chunk is a bad stride for the memory system, while 6144 B (N = 3072: 1.056 ms row-major, 1.07
//                        plane-blocked) and 1536 B (N = 768: 0.28 / 0.28, 0.93 / 0.93) are not;
//   bit 2: the epilogue input (residual / GELU' rows), addressed like C.
template <typename F, int EPI, bool HAS_IN, int HMF = 0>
__global__ __launch_bounds__(512, 2) void gemm_nt_p3_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    bf16_t* __restrict__ C, int ldc, const float* __restrict__ bias, const bf16_t* __restrict__ in, int ldin,
    bf16_t* __restrict__ C2, int ldc2, int tiles_n, int ntiles, DropCtx drop) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int nst = K / 64;                         // >= 4
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int swz = (fr >> 1) & 7;
  const uint32_t rowA = (uint32_t)((wr * 128 + fr) * 128), rowB = (uint32_t)((wc * 64 + fr) * 128);
  const uint32_t oA0 = rowA + (uint32_t)(((0 + fg) ^ swz) << 4), oA1 = rowA + (uint32_t)(((4 + fg) ^ swz) << 4);
  const uint32_t oB0 = rowB + (uint32_t)(((0 + fg) ^ swz) << 4), oB1 = rowB + (uint32_t)(((4 + fg) ^ swz) << 4);
  // epilogue geometry (store layout): lane -> row it*8 + (lane>>3), 16-B chunk (lane&7) ^ swizzle(row)
  const uint32_t ldsB = lds0 + 98304u;            // A slots: lds0 + {0,1,2} * 32 KB ; B slots: ldsB + {0,1} * 32 KB
  const int lr = lane >> 3;
  const int ec0 = ((lane & 7) ^ (lane >> 4)) << 3, ec1 = ((lane & 7) ^ (4 + (lane >> 4))) << 3;   // element column of the lane's 16-B chunk, rows lr / 8+lr
  const uint32_t bmask = bias ? 0xFFFFFFFFu : 0u;
  constexpr bool HM_A = (HMF & 1) != 0, HM_C = (HMF & 2) != 0, HM_I = (HMF & 4) != 0;
  // DEEP_IN: the epilogue's input tensor (residual / stored GELU') is fetched FOUR 16-row chunks ahead instead of one.  The wave's
  // 4 KB slice of the freed A slot holds two 2 KB chunk buffers; the other two are the wave's 4 KB slice of the B slot the tile's
  // last stage frees -- the bytes wave w's own share of the next tile's B(1) load will overwrite -- so that load is issued AFTER
  // the epilogue (weights: L2 hits, one stage of slack) instead of at the last stage boundary.  One chunk ahead (round 4) the
  // epilogue waited out a memory latency per chunk: 12-13k cycles per tile against 3.3k without an input (tools/p3_timeline.py).
#ifdef SIMX_P3_NODEEP
  constexpr bool DEEP_IN = false;
#else
  constexpr bool DEEP_IN = HAS_IN;
#endif
  // (the counted vmcnt immediates of the DEEP_IN epilogue assume two stores per chunk and one input stream: EPI NONE / DGELU)
  static_assert(!(DEEP_IN && EPI == SIMX_EPI_GELU), "DEEP_IN epilogue: the vmcnt schedule has no room for the GELU pair's second store");
  const int hmR = ldc2;                           // rows of a plane (ldc2 is free: a plane-blocked C2 has pitch 64)
  if (HM_A) lda = 64;                             // row pitch inside a plane; the stage's k offset selects the plane (P3_AK)
  if (HM_I) ldin = 64;
  const uint32_t offA0 = (uint32_t)(lr * lda + ec0) * 2, offA1 = (uint32_t)(lr * lda + ec1) * 2;
  const uint32_t offB0 = (uint32_t)(lr * ldb + ec0) * 2, offB1 = (uint32_t)(lr * ldb + ec1) * 2;
  // element offset of K stage s of the A operand: s * 64 columns, or plane s of ldc2 rows x 64
#define P3_AK(S) (HM_A ? (long)(S) * ((long)hmR * 64) : (long)(S) * 64)
  // Epilogue-only per-lane constants are recomputed per tile from a laundered copy of the lane id (P_LANE): left to
  // LICM they are hoisted out of the tile loop and stay live (~20 VGPRs) through the main loop, which then spills.
#define P_LANE(L) int L = lane; asm volatile("" : "+v"(L))

  int v = blockIdx.x;
  // Tile order.  Default: the XCD-aware row-major order -- the tiles_n N-tiles of an M-panel run at the same time on neighbouring CUs of
  // one XCD and each fetches the A panel itself (traffic 1.43x algorithmic).  -DSIMX_P3_PANEL_ORDER (tools/build_variant.sh, the
  // refetch-vs-clock A/B of DESIGN.md section 5): a workgroup walks all N-tiles of ITS M-panels one after the other, so the panel comes
  // from HBM once and from the L2 tiles_n - 1 times (needs tiles_m % gridDim.x == 0: M = 262144 or 327680 on 256 CUs).
#ifdef SIMX_P3_PANEL_ORDER
#define P3_TILE(V, M0, N0) do { const int k__ = (V) / (int)gridDim.x, w__ = (V) % (int)gridDim.x; \
                                M0 = (w__ + (k__ / tiles_n) * (int)gridDim.x) * 256; N0 = (k__ % tiles_n) * 256; } while (0)
#else
#define P3_TILE(V, M0, N0) do { const int t__ = xcd_remap(V, ntiles); M0 = (t__ / tiles_n) * 256; N0 = (t__ % tiles_n) * 256; } while (0)
#endif
  int m0, n0;
  P3_TILE(v, m0, n0);
  p3_half(B, ldb, n0, 0, ldsB, wave, offB0, offB1);
  p3_half(A, lda, m0, P3_AK(0), lds0, wave, offA0, offA1);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  p3_half(B, ldb, n0, 64, ldsB + 32768u, wave, offB0, offB1);
  p3_half(A, lda, m0, P3_AK(1), lds0 + 32768u, wave, offA0, offA1);
  p3_half(A, lda, m0, P3_AK(2), lds0 + 65536u, wave, offA0, offA1);
  int a0 = 0, b0 = 0;                             // LDS slots of the current tile's stage 0 (A: mod 3, B: mod 2)
  // the bias enters as the accumulators' initial value; bq* always hold the CURRENT tile's 4x4 bias columns at the
  // top of the loop (the next tile's are requested at the last stage boundary and carried across the epilogue)
  f32x4 bq0, bq1, bq2, bq3;
  {
    const float* b0 = bias ? bias + n0 + wc * 64 : reinterpret_cast<const float*>(A);
    const uint32_t boff = (uint32_t)(fg * 16);
    P_GLD4(bq0, boff, b0, 0); P_GLD4(bq1, boff, b0, 64); P_GLD4(bq2, boff, b0, 128); P_GLD4(bq3, boff, b0, 192);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bq0), "+v"(bq1), "+v"(bq2), "+v"(bq3)::"memory");
  }

#define V3_MFMA_ROW(I, AF, B0, B1, B2, B3)                                                     \
  acc[I][0] = H16<F>::mfma(B0, AF, acc[I][0]);             \
  acc[I][1] = H16<F>::mfma(B1, AF, acc[I][1]);             \
  acc[I][2] = H16<F>::mfma(B2, AF, acc[I][2]);             \
  acc[I][3] = H16<F>::mfma(B3, AF, acc[I][3])
#define V3_RD1(F, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:" #OFF : "=&v"(F) : "v"(ADDR) : "memory")
#define V3_SB __builtin_amdgcn_sched_barrier(0)
  // (timing experiments that priced the waits of this loop -- no barrier, no vmcnt wait, no lgkmcnt waits, s_setprio around
  // the MFMA bursts, all stores to one tile's rows -- are recorded in DESIGN.md 5 / 5b; the switches are not part of the product)
#define P3_BOUNDARY_WAIT() asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory")
#define P3_LGKM_WAIT "s_waitcnt lgkmcnt(0)"
  // the 8 LDS-DMA instructions a mid-tile boundary issues go out ONE PER MFMA ROW (B in the rest of this k-step, A in the
  // first half of the next), never as a burst right after the barrier (+5 %)
  const char* pa_g = nullptr; const char* pb_g = nullptr;
  uint32_t pa_slot = 0, pb_slot = 0;
  bool pa_pend = false, pb_pend = false;
#define P3_HA(J) do { if (pa_pend) { P_DMA16(((J) & 1) ? offA1 : offA0, pa_g + (long)(J) * 8 * lda * 2, pa_slot + (uint32_t)((J) * 1024)); if ((J) == 3) pa_pend = false; } } while (0)
#define P3_HB(J) do { if (pb_pend) { P_DMA16(((J) & 1) ? offB1 : offB0, pb_g + (long)(J) * 8 * ldb * 2, pb_slot + (uint32_t)((J) * 1024)); if ((J) == 3) pb_pend = false; } } while (0)
#define P3_HAX(ROW, S1) do { if ((ROW) < 4) P3_HA(ROW); } while (0)
#define P_STEP(CURA, NA, NB, BC0, BC1, BC2, BC3, BN0, BN1, BN2, BN3, BOUNDARY, S1)               \
  do {                                                                                         \
    /* fragment reads are front-loaded: the last read before each pin is issued two MFMA rows ahead of it */ \
    const uint32_t aa__ = (CURA), na__ = (NA), nb__ = (NB);                                    \
    V3_SB; V3_MFMA_ROW(0, al0, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah0, aa__, 8192); V3_RD1(ah1, aa__, 10240); P3_HAX(0, S1); \
    V3_SB; V3_MFMA_ROW(1, al1, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(ah2, aa__, 12288); V3_RD1(ah3, aa__, 14336); P3_HAX(1, S1); \
    V3_SB; V3_MFMA_ROW(2, al2, BC0, BC1, BC2, BC3); V3_SB; P3_HAX(2, S1);                      \
    V3_SB; V3_MFMA_ROW(3, al3, BC0, BC1, BC2, BC3); V3_SB; P3_HAX(3, S1);                      \
    V3_SB;                                                                                     \
    V3_PIN4(P3_LGKM_WAIT, ah0, ah1, ah2, ah3);                                       \
    BOUNDARY();                                                                                \
    V3_SB; V3_MFMA_ROW(4, ah0, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al0, na__, 0); V3_RD1(BN0, nb__, 0); V3_RD1(al1, na__, 2048); P3_HB(1); P3_HAX(4, S1); \
    V3_SB; V3_MFMA_ROW(5, ah1, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(BN1, nb__, 2048); V3_RD1(al2, na__, 4096); V3_RD1(BN2, nb__, 4096); P3_HB(2); P3_HAX(5, S1); \
    V3_SB; V3_MFMA_ROW(6, ah2, BC0, BC1, BC2, BC3); V3_SB; V3_RD1(al3, na__, 6144); V3_RD1(BN3, nb__, 6144); P3_HB(3); P3_HAX(6, S1); \
    V3_SB; V3_MFMA_ROW(7, ah3, BC0, BC1, BC2, BC3);                                            \
    V3_SB;                                                                                     \
    V3_PIN8(P3_LGKM_WAIT, al0, al1, al2, al3, BN0, BN1, BN2, BN3);                   \
  } while (0)
#define P_BND_NONE() do { } while (0)
  // stage boundary inside the tile: every fragment of stage st is in registers, stage st+1 has landed once vmcnt
  // hits 0 (for everyone after the barrier); slot (st+par)&1 is refilled with stream stage st+2, which is the
  // NEXT tile's stage 0 when st == nst-2
#define P_BND_MID()                                                                            \
  do {                                                                                         \
    /* issue order so far: ... [B(st+1) A(st+2)] ; stage st+1 = everything but the 4 youngest (A(st+2)) */ \
    P3_BOUNDARY_WAIT();                                                                        \
    const bool cb__ = st + 2 < nst, ca__ = st + 3 < nst;                                       \
    /* same issue ORDER as a burst would have (B x4, then A x4 -- the vmcnt(4) rule holds), one instruction per MFMA row */ \
    pb_g = reinterpret_cast<const char*>(B + (long)((cb__ ? n0 : n0n) + wave * 32) * ldb + (cb__ ? st + 2 : st + 2 - nst) * 64); \
    pb_slot = ldsB + (uint32_t)(bc * 32768 + wave * 4096);                                     \
    pa_g = reinterpret_cast<const char*>(A + (long)((ca__ ? m0 : m0n) + wave * 32) * lda + P3_AK(ca__ ? st + 3 : st + 3 - nst)); \
    pa_slot = lds0 + (uint32_t)(ac * 32768 + wave * 4096);                                     \
    pb_pend = pa_pend = true;                                                                  \
    P_DMA16(offB0, pb_g, pb_slot);                                                             \
  } while (0)
  // last boundary of the tile: bias and the first residual chunk are requested BEFORE the next tile's stage 1,
  // so the epilogue can wait for them without waiting for that stage
#define P_BND_LAST()                                                                           \
  do {                                                                                         \
    asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");                              \
    P_LANE(lb__);                                                                              \
    const uint32_t boff__ = (uint32_t)((lb__ >> 4) * 16);                                      \
    P_GLD4(bq0, boff__, bptr, 0); P_GLD4(bq1, boff__, bptr, 64); P_GLD4(bq2, boff__, bptr, 128); P_GLD4(bq3, boff__, bptr, 192); \
    if (HAS_IN) {                                                                              \
      const int lr__ = lb__ >> 3;                                                              \
      const uint32_t io0__ = (uint32_t)(lr__ * ldin + (((lb__ & 7) ^ (lb__ >> 4)) << 3)) * 2;  \
      const uint32_t io1__ = (uint32_t)((lr__ + 8) * ldin + (((lb__ & 7) ^ (4 + (lb__ >> 4))) << 3)) * 2; \
      P_DMA16(io0__, ibase, ereg); P_DMA16(io1__, ibase, ereg + 1024u);                        \
      if (DEEP_IN) {                     /* chunks 1-3: second half of the A slice, then this wave's slice of the freed B slot */ \
        const uint32_t breg__ = ldsB + (uint32_t)(bc * 32768 + wave * 4096);                   \
        P_DMA16(io0__, ibase + (long)32 * ldin, ereg + 2048u); P_DMA16(io1__, ibase + (long)32 * ldin, ereg + 3072u); \
        P_DMA16(io0__, ibase + (long)64 * ldin, breg__); P_DMA16(io1__, ibase + (long)64 * ldin, breg__ + 1024u); \
        P_DMA16(io0__, ibase + (long)96 * ldin, breg__ + 2048u); P_DMA16(io1__, ibase + (long)96 * ldin, breg__ + 3072u); \
      }                                                                                        \
    }                                                                                          \
    /* next tile's stage 1 of B; its stage 2 of A goes into the slot this tile's epilogue borrows -> issued after it */ \
    pb_g = reinterpret_cast<const char*>(B + (long)(n0n + wave * 32) * ldb + 64);              \
    pb_slot = ldsB + (uint32_t)(bc * 32768 + wave * 4096);                                     \
    if (!DEEP_IN) {                                                                            \
      pb_pend = true;                                                                          \
      P_DMA16(offB0, pb_g, pb_slot);                                                           \
    }                                                                                          \
  } while (0)

  for (;;) {
    const int vn = v + (int)gridDim.x;
    const bool has_next = vn < ntiles;
    int m0n = m0, n0n = n0;                       // past the end: re-fetch this tile (harmless, keeps the loop branch-free)
    if (has_next) P3_TILE(vn, m0n, n0n);
    const int mw = m0 + wr * 128, nw = n0 + wc * 64;
    const float* bptr = bias ? bias + n0n + wc * 64 : reinterpret_cast<const float*>(A);   // uniform
    const char* ibase = !HAS_IN ? nullptr : reinterpret_cast<const char*>(HM_I ? in + ((long)(nw >> 6) * hmR + mw) * 64 : in + (long)mw * ldin + nw);   // uniform

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
    {
      const uint32_t aa = lds0 + (uint32_t)(a0 * 32768) + oA0, ab = ldsB + (uint32_t)(b0 * 32768) + oB0;
      V3_READ4(al0, al1, al2, al3, aa, 0, 2048, 4096, 6144);
      V3_READ4(bx0, bx1, bx2, bx3, ab, 0, 2048, 4096, 6144);
      V3_PIN8("s_waitcnt lgkmcnt(0)", al0, al1, al2, al3, bx0, bx1, bx2, bx3);
    }
    int ac = a0, bc = b0;                         // slots of the stage being consumed
    for (int st = 0; st < nst - 1; ++st) {
      const int an = ac == 2 ? 0 : ac + 1, bn = bc ^ 1;
      const uint32_t sa = lds0 + (uint32_t)(ac * 32768), sb = ldsB + (uint32_t)(bc * 32768);
      const uint32_t na = lds0 + (uint32_t)(an * 32768), nb = ldsB + (uint32_t)(bn * 32768);
      P_STEP(sa + oA0, sa + oA1, sb + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, P_BND_NONE, 1);
      P_STEP(sa + oA1, na + oA0, nb + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, P_BND_MID, 0);
      ac = an; bc = bn;
    }
    const uint32_t ereg = lds0 + (uint32_t)(ac * 32768 + wave * 4096);     // this wave's slice of the last stage's A slot
    {
      const int an = ac == 2 ? 0 : ac + 1, bn = bc ^ 1;
      const uint32_t sa = lds0 + (uint32_t)(ac * 32768), sb = ldsB + (uint32_t)(bc * 32768);
      const uint32_t na = lds0 + (uint32_t)(an * 32768), nb = ldsB + (uint32_t)(bn * 32768);
      P_STEP(sa + oA0, sa + oA1, sb + oB1, bx0, bx1, bx2, bx3, by0, by1, by2, by3, P_BND_NONE, 1);
      P_STEP(sa + oA1, na + oA0, nb + oB0, by0, by1, by2, by3, bx0, bx1, bx2, bx3, P_BND_LAST, 0);
    }

    // ---- epilogue, 8 chunks of 16 rows (= accumulator row-block i), per wave, no barrier
    // (the bias registers are pinned here, BEFORE any branch: a branch between an asm load and its pin makes hipcc
    // copy the in-flight registers and the copies read garbage)
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(bq0), "+v"(bq1), "+v"(bq2), "+v"(bq3) : "n"(DEEP_IN ? 8 : HAS_IN ? 6 : 4) : "memory");
    if (C != nullptr) {                          // (nullptr: the main-loop-only measurement build, -DSIMX_MEASUREMENT_HOOKS)
      P_LANE(le);
      const int fr = le & 15, fg = le >> 4, lr = le >> 3;          // shadow the kernel-scope copies on purpose
      const int ec0 = ((le & 7) ^ (le >> 4)) << 3, ec1 = ((le & 7) ^ (4 + (le >> 4))) << 3;
      const int sw = (fr >> 1) & 7;
      const uint32_t slot = (uint32_t)(fr * 128 + (fg & 1) * 8);
      if (HM_C) ldc = 64;                                      // (per tile: `ldc` is dead in the main loop)
      const uint32_t eo0 = (uint32_t)(lr * ldc + ec0) * 2, eo1 = (uint32_t)((lr + 8) * ldc + ec1) * 2;
      const uint32_t io0 = HAS_IN ? (uint32_t)(lr * ldin + ec0) * 2 : 0, io1 = HAS_IN ? (uint32_t)((lr + 8) * ldin + ec1) * 2 : 0;
      bf16_t* const obase = HM_C ? C + ((long)(nw >> 6) * hmR + mw) * 64 : C + (long)mw * ldc + nw;          // uniform
      const bool store_pre = !(EPI == SIMX_EPI_GELU && ldin == 1);   // ldin == 1 on a GELU launch: SIMX_EPI_GELU_INFER
#ifndef SIMX_P3_NOPIPE
      // Plain epilogue (bias (+ dropout), no epilogue input), software-pipelined over the wave's two 2 KB sub-buffers: chunk
      // i + 1 is packed and written while chunk i's two 16-B reads are in flight, and chunk i's global stores are issued
      // while chunk i + 1's writes land -- one exposed LDS latency per chunk instead of two.  (LDS operations of a wave
      // complete in order: lgkmcnt(4) after [read, read, 4 writes] = the reads are back.)  The dropout decision selects one of
      // two copies of the loop: a branch between an asm read and the wait that names its registers makes hipcc copy them.
      if (!HAS_IN && EPI == SIMX_EPI_NONE) {
        auto run = [&](auto drop_c) {
          constexpr bool DROPE = decltype(drop_c)::value;
#define P3_PACK(I)                                                                                              \
          do {                                                                                                  \
            const uint32_t sb__ = ereg + (uint32_t)(((I) & 1) * 2048) + slot;                                  \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                     \
              float vv[4] = {acc[I][j][0], acc[I][j][1], acc[I][j][2], acc[I][j][3]};                           \
              if (DROPE) {                                                                                      \
                float m4[4];                                                                                    \
                drop_mult4(drop, (uint32_t)(mw + (I) * 16 + fr), (uint32_t)(nw + j * 16 + fg * 4), m4);         \
                vv[0] *= m4[0]; vv[1] *= m4[1]; vv[2] *= m4[2]; vv[3] *= m4[3];                                 \
              }                                                                                                 \
              const uint2 o = make_uint2(H16<F>::pack2(vv[0], vv[1]), H16<F>::pack2(vv[2], vv[3]));             \
              asm volatile("ds_write_b64 %0, %1" ::"v"(sb__ + (uint32_t)(((2 * j + (fg >> 1)) ^ sw) << 4)), "v"(o) : "memory"); \
            }                                                                                                   \
          } while (0)
#define P3_CHUNK(I)                                                                                             \
          do {                                                                                                  \
            const uint32_t rd__ = ereg + (uint32_t)(((I) & 1) * 2048) + (uint32_t)(le * 16);                    \
            u32x4 w0__, w1__;                                                                                   \
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"       \
                         : "=&v"(w0__), "=&v"(w1__) : "v"(rd__) : "memory");                                    \
            if ((I) < 7) {                                                                                      \
              P3_PACK((I) + 1 < 8 ? (I) + 1 : 7);                                                               \
              asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(w0__), "+v"(w1__)::"memory");                          \
            } else {                                                                                            \
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(w0__), "+v"(w1__)::"memory");                          \
            }                                                                                                   \
            P_GST4(eo0, obase + (long)(I) * 16 * ldc, w0__);                                                    \
            P_GST4(eo1, obase + (long)(I) * 16 * ldc, w1__);                                                    \
          } while (0)
          P3_PACK(0);
          P3_CHUNK(0); P3_CHUNK(1); P3_CHUNK(2); P3_CHUNK(3); P3_CHUNK(4); P3_CHUNK(5); P3_CHUNK(6); P3_CHUNK(7);
#undef P3_CHUNK
#undef P3_PACK
        };
        if (drop.thr) run(std::true_type{}); else run(std::false_type{});
      } else
#endif
      {
      // (the training / inference decision of the GELU launch -- is gelu'(u) stored? -- selects one of two copies of the loop: as a
      // runtime test it put a branch into every 16 x 16 sub-tile of the GELU arithmetic and the scheduler could not interleave
      // the transcendental chains of neighbouring sub-tiles)
      const int ld2 = HM_C ? 64 : ldc2;
      const uint32_t go0 = (uint32_t)(lr * ld2 + ec0) * 2, go1 = (uint32_t)((lr + 8) * ld2 + ec1) * 2;     // the gelu(u) output's lane offsets
      auto body = [&, go0, go1](auto pre_c) {
      constexpr bool SP = decltype(pre_c)::value;
      // DEEP_IN: the input of chunk i + 1 is read from its buffer while chunk i's output is on its way through its own
      // (tn0..3: issued behind chunk i's four writes, waited for together with chunk i's two 16-B reads) -- two LDS waits per chunk
      // instead of three.  Issue order of the wave's VMEM operations from the last stage boundary on: D0 D1 D2 D3 | S0 D4 | S1 D5 |
      // S2 D6 | S3 D7 | S4 | S5 | S6 | S7 (Dk, Sk: 2 instructions each)
      uint2 tn0 = make_uint2(0, 0), tn1 = tn0, tn2 = tn0, tn3 = tn0;
      if (DEEP_IN) {                                     // chunk 0 (younger than D0: D1 D2 D3)
        const uint32_t b0 = ereg + slot;
        asm volatile("s_waitcnt vmcnt(6)\n\tds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(tn0), "=&v"(tn1), "=&v"(tn2), "=&v"(tn3)
                     : "v"(b0 + (uint32_t)(((0 + (fg >> 1)) ^ sw) << 4)), "v"(b0 + (uint32_t)(((2 + (fg >> 1)) ^ sw) << 4)),
                       "v"(b0 + (uint32_t)(((4 + (fg >> 1)) ^ sw) << 4)), "v"(b0 + (uint32_t)(((6 + (fg >> 1)) ^ sw) << 4)) : "memory");
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // DEEP_IN: chunk i lives (input, then output, in place) in buffer i % 4: two in the A slice, two in the B-slot slice
        const uint32_t sub = DEEP_IN ? ((i & 2) ? pb_slot : ereg) + (uint32_t)((i & 1) * 2048)
                                     : ereg + (uint32_t)((EPI == SIMX_EPI_GELU ? 0 : (i & 1)) * 2048);
        if (DEEP_IN) {
        } else if (HAS_IN) {
          if (i < 7) {
            const uint32_t nx = ereg + (uint32_t)(((i + 1) & 1) * 2048);
            const char* ib = ibase + (long)(i + 1) * 32 * ldin;
            P_DMA16(io0, ib, nx);
            P_DMA16(io1, ib, nx + 1024u);
          }
          if (i == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
          else if (i < 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        }
        uint2 t0, t1, t2, t3;
        const uint32_t ad0 = sub + slot + (uint32_t)(((0 + (fg >> 1)) ^ sw) << 4), ad1 = sub + slot + (uint32_t)(((2 + (fg >> 1)) ^ sw) << 4);
        const uint32_t ad2 = sub + slot + (uint32_t)(((4 + (fg >> 1)) ^ sw) << 4), ad3 = sub + slot + (uint32_t)(((6 + (fg >> 1)) ^ sw) << 4);
        if (DEEP_IN) {
          t0 = tn0; t1 = tn1; t2 = tn2; t3 = tn3;
        } else if (HAS_IN) {
          asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(ad0), "v"(ad1), "v"(ad2), "v"(ad3) : "memory");
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t ad = j == 0 ? ad0 : j == 1 ? ad1 : j == 2 ? ad2 : ad3;
          float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          if (EPI == SIMX_EPI_NONE && drop.thr) {
            float m4[4];
            drop_mult4(drop, (uint32_t)(mw + i * 16 + fr), (uint32_t)(nw + j * 16 + fg * 4), m4);
            vv[0] *= m4[0]; vv[1] *= m4[1]; vv[2] *= m4[2]; vv[3] *= m4[3];
          }
          if (HAS_IN) {
            const uint2 t = j == 0 ? t0 : j == 1 ? t1 : j == 2 ? t2 : t3;
            const float x0 = H16<F>::lo(t.x), x1 = H16<F>::hi(t.x);
            const float x2 = H16<F>::lo(t.y), x3 = H16<F>::hi(t.y);
            if (EPI == SIMX_EPI_NONE) { vv[0] += x0; vv[1] += x1; vv[2] += x2; vv[3] += x3; }
            else { vv[0] *= x0; vv[1] *= x1; vv[2] *= x2; vv[3] *= x3; }          // aux = gelu'(u), stored by the forward
          }
          if (EPI == SIMX_EPI_GELU) {             // C2 = gelu(u); C = gelu'(u) (skipped on the inference form: no backward)
            float g[4];
            if (SP) {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                f32p_t gp, dp;
                gelu_both_fast2((f32p_t){vv[e], vv[e + 1]}, gp, dp);
                g[e] = gp.x; g[e + 1] = gp.y; vv[e] = dp.x; vv[e + 1] = dp.y;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; e += 2) {
                const f32p_t gp = gelu_fast2((f32p_t){vv[e], vv[e + 1]});
                g[e] = gp.x; g[e + 1] = gp.y;
              }
            }
            const uint2 og = make_uint2(H16<F>::pack2(g[0], g[1]), H16<F>::pack2(g[2], g[3]));
            asm volatile("ds_write_b64 %0, %1 offset:2048" ::"v"(ad), "v"(og) : "memory");
          }
          const uint2 o = make_uint2(H16<F>::pack2(vv[0], vv[1]), H16<F>::pack2(vv[2], vv[3]));
          asm volatile("ds_write_b64 %0, %1" ::"v"(ad), "v"(o) : "memory");
        }
        const uint32_t rd = sub + (uint32_t)(le * 16);
        u32x4 w0, w1;
        if (DEEP_IN && i < 7) {
          // wait for D(i+1) (younger than it here: 4, 6, 8, 8, 8, 6, 4 operations for i = 0..6), request its four 8-B pieces,
          // lgkmcnt(4) = this chunk's four writes have landed, read them back as two 16-B rows, wait for everything
          const uint32_t bn = (((i + 1) & 2) ? pb_slot : ereg) + (uint32_t)(((i + 1) & 1) * 2048) + slot;
          asm volatile("s_waitcnt vmcnt(%12)\n\tds_read_b64 %0, %6\n\tds_read_b64 %1, %7\n\tds_read_b64 %2, %8\n\tds_read_b64 %3, %9\n\t"
                       "s_waitcnt lgkmcnt(4)\n\tds_read_b128 %4, %10\n\tds_read_b128 %5, %10 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(tn0), "=&v"(tn1), "=&v"(tn2), "=&v"(tn3), "=&v"(w0), "=&v"(w1)
                       : "v"(bn + (uint32_t)(((0 + (fg >> 1)) ^ sw) << 4)), "v"(bn + (uint32_t)(((2 + (fg >> 1)) ^ sw) << 4)),
                         "v"(bn + (uint32_t)(((4 + (fg >> 1)) ^ sw) << 4)), "v"(bn + (uint32_t)(((6 + (fg >> 1)) ^ sw) << 4)), "v"(rd),
                         "n"(0), "n"(i == 0 || i == 6 ? 4 : i == 1 || i == 5 ? 6 : 8)
                       : "memory");
        } else
        asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(w0), "=&v"(w1) : "v"(rd) : "memory");
        if (SP) {                                 // (inference GELU: the pre-activation has no reader)
          P_GST4(eo0, obase + (long)i * 16 * ldc, w0);
          P_GST4(eo1, obase + (long)i * 16 * ldc, w1);
        }
        if (DEEP_IN && i < 4) {                          // the buffer is drained (w0, w1 are back): chunk i + 4 into it
          const char* ib = ibase + (long)(i + 4) * 32 * ldin;
          P_DMA16(io0, ib, sub);
          P_DMA16(io1, ib, sub + 1024u);
        }
        if (EPI == SIMX_EPI_GELU) {
          u32x4 w2, w3;
          asm volatile("ds_read_b128 %0, %2 offset:2048\n\tds_read_b128 %1, %2 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                       : "=&v"(w2), "=&v"(w3) : "v"(rd) : "memory");
          bf16_t* const gbase = HM_C ? C2 + ((long)(nw >> 6) * hmR + mw + i * 16) * 64 : C2 + (long)(mw + i * 16) * ldc2 + nw;   // uniform
          P_GST4(go0, gbase, w2);
          P_GST4(go1, gbase, w3);
        }
      }
      };
      if (store_pre) body(std::true_type{}); else body(std::false_type{});
      }
    }
    if (DEEP_IN) {       // the borrowed B-slot slice is free again: next tile's stage 1 of B (ahead of A(2): the vmcnt(4) rule)
      if (C == nullptr) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (measurement build only: the input chunks requested at the last boundary were never consumed)
      P_DMA16(offB0, pb_g, pb_slot); P_DMA16(offB1, pb_g + (long)8 * ldb * 2, pb_slot + 1024u);
      P_DMA16(offB0, pb_g + (long)16 * ldb * 2, pb_slot + 2048u); P_DMA16(offB1, pb_g + (long)24 * ldb * 2, pb_slot + 3072u);
    }
    // the borrowed A slot is free again (this wave's slice only ever holds this wave's rows): next tile's stage 2
    if (has_next) {                                // issued by the next tile's first four MFMA rows (same order: after B(1))
      pa_g = reinterpret_cast<const char*>(A + (long)(m0n + wave * 32) * lda + P3_AK(2));
      pa_slot = lds0 + (uint32_t)(ac * 32768 + wave * 4096);
      pa_pend = true;
    } else {
      p3_half(A, lda, m0n, P3_AK(2), lds0 + (uint32_t)(ac * 32768), wave, offA0, offA1);
    }
    if (!has_next) break;
    v = vn; m0 = m0n; n0 = n0n; a0 = ac == 2 ? 0 : ac + 1; b0 = bc ^ 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the trailing (dummy) stage loads must land before the LDS is released
#undef P3_TILE
#undef P_LANE
#undef P3_HA
#undef P3_HAX
#undef P3_HB
#undef P3_AK
#undef P_BND_LAST
#undef P_BND_MID
#undef P_BND_NONE
#undef P_STEP
#undef V3_RD1
#undef V3_SB
#undef V3_MFMA_ROW
}


// ------------------------------------------------------------------------------------------
// bf16 TN kernel (wgrad) : slab[split][M][N] = A[kslice,M]^T . B[kslice,N]
// LDS image of a 64(k) x 128(col) tile: row kr at kr*256 B; the 32-B chunk q of the row stored
// at chunk position q ^ (kr & 7).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void tn_stage_tile(const bf16_t* __restrict__ G, int ld, int k0, int k_end,
                                              int col0, int ncols_total, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = wave * 4 + j;                 // wave-instruction index: 4 k-rows each
    const int kr = i * 4 + (lane >> 4);
    const int p16 = lane & 15;
    const int q = (p16 >> 1) ^ (kr & 7);
    int gk = k0 + kr;
    gk = gk < k_end ? gk : k_end - 1;
    int gc = col0 + q * 16 + (p16 & 1) * 8;
    gc = gc < ncols_total ? gc : ncols_total - 8;
    glds16(G + (long)gk * ld + gc, lds_tile + i * 1024);
  }
}

__device__ __forceinline__ bf16x4 lds_tr_read(uint32_t addr) {
  bf16x4 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

template <typename F>
__global__ __launch_bounds__(256, 2) void gemm_tn_h16_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    float* __restrict__ out, long slab_stride, int ldo, int tiles_n, int tiles_mn, int k_per_split, int accumulate,
    const float* __restrict__ gs) {                     // gs != NULL: `out` is the gradient itself (no slab pass follows): x 1/S
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x / tiles_mn;
  const int tile = blockIdx.x % tiles_mn;
  const int m0 = (tile / tiles_n) * 128, n0 = (tile % tiles_n) * 128;
  const int wr = wave >> 1, wc = wave & 1;
  const int kb = split * k_per_split;
  const int ke = min(K, kb + k_per_split);

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nt = (ke - kb + 63) / 64;
  tn_stage_tile(A, lda, kb, ke, m0, M, smem, wave, lane);
  tn_stage_tile(B, ldb, kb, ke, n0, N, smem + 16384, wave, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const int fs = lane & 15, fg = lane >> 4;
  const uint32_t lds_base = (uint32_t)(uintptr_t)smem;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    char* sa = smem + cur * NT_STAGE_BYTES;
    const int kt = kb + t * 64;
    if (kt + 64 > ke) {
      // ragged tail: rows >= ke-kt hold clamped copies; zero them in both tiles
      const int valid = ke - kt;
      for (int idx = tid; idx < 2 * 64 * 16; idx += 256) {
        const int tl = idx >> 10, rem = idx & 1023, kr = rem >> 4, c16 = rem & 15;
        if (kr >= valid) *reinterpret_cast<uint4*>(sa + tl * 16384 + kr * 256 + c16 * 16) = make_uint4(0, 0, 0, 0);
      }
      __syncthreads();
    }
    if (t + 1 < nt) {
      char* na = smem + (cur ^ 1) * NT_STAGE_BYTES;
      tn_stage_tile(A, lda, kt + 64, ke, m0, M, na, wave, lane);
      tn_stage_tile(B, ldb, kt + 64, ke, n0, N, na + 16384, wave, lane);
    }
    const uint32_t a_base = lds_base + cur * NT_STAGE_BYTES;
    const uint32_t b_base = a_base + 16384;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        bf16x4 lo, hi, lo2, hi2;
        {
          const int tr0 = ks * 32 + 4 * fg + (fs >> 2), tr1 = tr0 + 16;
          const int cm = wr * 4 + i, cn = wc * 4 + i;
          lo = lds_tr_read(a_base + tr0 * 256 + ((cm ^ (tr0 & 7)) << 5) + (fs & 3) * 8);
          hi = lds_tr_read(a_base + tr1 * 256 + ((cm ^ (tr1 & 7)) << 5) + (fs & 3) * 8);
          lo2 = lds_tr_read(b_base + tr0 * 256 + ((cn ^ (tr0 & 7)) << 5) + (fs & 3) * 8);
          hi2 = lds_tr_read(b_base + tr1 * 256 + ((cn ^ (tr1 & 7)) << 5) + (fs & 3) * 8);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi), "+v"(lo2), "+v"(hi2)::"memory");
        af[i] = (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        bfr[i] = (bf16x8){lo2[0], lo2[1], lo2[2], lo2[3], hi2[0], hi2[1], hi2[2], hi2[3]};
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = H16<F>::mfma(bfr[j], af[i], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }

  float* o = out + (long)split * slab_stride;
  const float inv = gs_inv(gs);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + fs;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fg * 4;
      if (n >= N) continue;
      float4* dst = reinterpret_cast<float4*>(o + (long)m * ldo + n);
      float4 v = make_float4(acc[i][j][0] * inv, acc[i][j][1] * inv, acc[i][j][2] * inv, acc[i][j][3] * inv);
      if (accumulate) { float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// bf16 TN kernel v2 (large wgrad): 256x256 block tile, 8 waves (2x4) of 128x64, 64-token stages (two 64 KB
// stages, full 512-B rows by LDS-DMA), fragments by ds_read_b64_tr_b16 interleaved with the MFMAs, one barrier
// per stage.  With 128x128 tiles the wgrad GEMM sat exactly on the LDS-DMA request ceiling (~20 B/clk/CU x
// 64 flop/B = 690 TFLOP/s measured 640-660); this tile needs half the bytes per flop.
// Optional fused bias gradient: workgroups of the first N-tile (waves with wc == 0) also sum the A operand
// (= dY) over tokens on the VALU, in the shadow of the MFMAs, and flush with one atomic per column.
// LDS image of a 64(k) x 256(col) operand stage: row kr at kr*512 B, 32-B chunk q stored at q ^ (kr & 7).
// ------------------------------------------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(512, 2) void gemm_tn2_kernel(
    int M, int N, int K, const bf16_t* __restrict__ A, int lda, const bf16_t* __restrict__ B, int ldb,
    float* __restrict__ out, long slab_stride, int ldo, int tiles_n, int tiles_mn, int k_per_split, int accumulate,
    float* __restrict__ dbias, int hm_a, const float* __restrict__ gs, int scale_out, int dbias_parts) {
  // dbias_parts (deterministic mode): dbias is a [splits * tiles_n * 4][M] partial buffer, one row per contributor
  // gs = {S, 1/S} of the fp16 backward or NULL: the bias gradient always leaves x 1/S; the product only when `out` is the
  // gradient itself (scale_out; with split-K the slab reduction applies it)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: block b runs on XCD b%8 (private L2).  The tiles of one split read the same token range: the
  // tiles_n workgroups of an M-tile share its dY slab, the tiles_m of an N-tile its X slab.  In launch order those
  // sharers sit on 8 different XCDs and every slab is fetched from the fabric once per sharer (PMC: 22 % L2 hits,
  // ~6 TB/s of fabric reads = what bounded this kernel); giving each XCD a contiguous range of (split, tile) puts them
  // behind one L2.
  const int vb = xcd_remap(blockIdx.x, gridDim.x);
  const int split = vb / tiles_mn;
  const int tile = vb % tiles_mn;
  const int m0 = (tile / tiles_n) * 256, n0 = (tile % tiles_n) * 256;
  const int wr = wave >> 2, wc = wave & 3;
  const int kb = split * k_per_split;
  const int ke = min(K, kb + k_per_split);
  const int fs = lane & 15, fg = lane >> 4;
  // fused bias gradient: the column sums of A over this workgroup's k-range are shared out stage by stage over the
  // tiles_n workgroups that stage the same A tile and over their four wc waves (which hold the same A fragments), so
  // every wave of every workgroup carries 1/(4 tiles_n) of it.  (Summing in the first N-tile's wc == 0 waves only made
  // those workgroups ~40 % slower and the whole launch 14 % slower.)
  const int bias_slot = (tile % tiles_n) * 4 + wc, bias_mod = tiles_n * 4;
  bool do_bias = false;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

  const int nst = (ke - kb + 63) / 64;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // transpose-read addressing: lane (fg, fs) supplies row 4*fg + (fs>>2) (+16 for the high half) of a 32-row k-step,
  // 8 B at element column ct*16 + (fs&3)*4 of 16-column tile ct -> 32-B chunk ct, byte (fs&3)*8 in it.
  const int r_lo = 4 * fg + (fs >> 2);                      // row within the k-step (low half); high half = +16
  const int x_lo = r_lo & 7, x_hi = (r_lo + 16) & 7;        // XOR terms (k-step bases are multiples of 32)
  const uint32_t row_lo = (uint32_t)(r_lo * 512 + (fs & 3) * 8), row_hi = (uint32_t)((r_lo + 16) * 512 + (fs & 3) * 8);

  uint32_t oa[4], ob[4];
  tn2_lane_offsets(lda, m0, M, ldb, n0, N, lane, oa, ob, hm_a);
  tn2_stage(A, lda, m0, M, B, ldb, n0, N, kb, ke, smem, wave, lane, hm_a);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (nst > 1) tn2_stage(A, lda, m0, M, B, ldb, n0, N, kb + 64, ke, smem + TN2_STAGE, wave, lane, hm_a);

  bf16x4 al_lo[4], al_hi[4], ah_lo[4], ah_hi[4], bx_lo[4], bx_hi[4], by_lo[4], by_hi[4];
  // address of fragment (operand base, 16-col tile ct) for k-step base address `kbase` (stage + ks*32*512)
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
// (What bounds this loop was priced with timing-only variants, M = 262144 tokens, average of the four wgrad shapes: as shipped
// 992 TFLOP/s; every stage re-reading stage 0 (all L2 hits) 1156; no operand loads at all 1445, 1614 on all-zero data.  So the
// LDS-DMA stream itself costs the loop 20 % and the fabric / HBM misses another 14 %; an L2 prefetch of stage s+3 made it 5 %
// SLOWER.  The 8 DMA instructions of a stage go out ONE PER MFMA ROW after the boundary: burst 1003, one (A,B) pair per row
// 1056, one instruction per row 1080 TFLOP/s.)
#define TN2_KSEL(S) (S)
#define TN2_SPREAD 2
#define TN2_PIECE(ST, J) tn2_stage_piece(A, hm_a > 0 ? 64 : lda, B, ldb, kb + TN2_KSEL((ST) + 2) * 64, lds0 + (uint32_t)(((ST) & 1) * TN2_STAGE), wave, J, oa[J], ob[J])
#define TN2_ONE_A(ST, J) tn2_stage_one(A, hm_a > 0 ? 64 : lda, kb + TN2_KSEL((ST) + 2) * 64, lds0 + (uint32_t)(((ST) & 1) * TN2_STAGE), wave, J, oa[J])
#define TN2_ONE_B(ST, J) tn2_stage_one(B, ldb, kb + TN2_KSEL((ST) + 2) * 64, lds0 + 32768u + (uint32_t)(((ST) & 1) * TN2_STAGE), wave, J, ob[J])
#define TN2_NOLOAD 0
#define TN2_PIN8(TXT, X, Y) do { } while (0)     /* waits are hipcc's (counted lgkmcnt): the reads are builtins */

  {
    // A[0..3] and B of k-step 0: column tile index ct = (wave col offset)/16 + i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      TN2_RD(al_lo[i], al_hi[i], TN2_ADDR_LO(lds0, 0u, wr * 8 + i), TN2_ADDR_HI(lds0, 0u, wr * 8 + i));
      TN2_RD(bx_lo[i], bx_hi[i], TN2_ADDR_LO(lds0, 32768u, wc * 4 + i), TN2_ADDR_HI(lds0, 32768u, wc * 4 + i));
    }
    TN2_PIN8("s_waitcnt lgkmcnt(0)", al_lo, al_hi);
    TN2_PIN8("s_waitcnt lgkmcnt(0)", bx_lo, bx_hi);
  }

  // one k-step (32 tokens): CUR = k-step base address, NXT = next k-step base address
#define TN2_STEP(CUR, NXT, BCL, BCH, BNL, BNH, SYNC, ST)                                                        \
  do {                                                                                                          \
    const uint32_t cur__ = (CUR), nxt__ = (NXT);                                                                \
    bool spread__ = false;                                                                                      \
    const bool tail__ = TN2_SPREAD == 2 && !(SYNC) && pend;       /* second half of the previous boundary's stage */ \
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
    TN2_PIN8("s_waitcnt lgkmcnt(0)", ah_lo, ah_hi);                                                             \
    if (SYNC) {                                                                                                 \
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");                                             \
      spread__ = !TN2_NOLOAD && TN2_SPREAD && ((ST) + 3 < nst || ((ST) + 3 == nst && (ke - kb) % 64 == 0));     \
      if (TN2_NOLOAD) { }                                                                                       \
      else if (spread__) {                                                                                      \
        if (TN2_SPREAD == 2) { TN2_ONE_A(ST, 0); pend = true; } else TN2_PIECE(ST, 0);                          \
      }                                                                                                         \
      else if ((ST) + 3 < nst || ((ST) + 3 == nst && (ke - kb) % 64 == 0))                                      \
        tn2_stage_full(A, hm_a > 0 ? 64 : lda, B, ldb, kb + TN2_KSEL((ST) + 2) * 64, lds0 + (uint32_t)(((ST) & 1) * TN2_STAGE), wave, oa, ob); \
      else if ((ST) + 2 < nst)                                                                                  \
        tn2_stage(A, lda, m0, M, B, ldb, n0, N, kb + ((ST) + 2) * 64, ke, smem + ((ST) & 1) * TN2_STAGE, wave, lane, hm_a); \
      if ((ST) + 1 == nst - 1 && (ke - kb) % 64 != 0) {                                                         \
        /* ragged last stage: rows >= valid hold clamped copies -> zero them (both operands) */                \
        const int valid__ = (ke - kb) - (nst - 1) * 64;                                                         \
        char* sp__ = smem + (((ST) + 1) & 1) * TN2_STAGE;                                                       \
        for (int idx = tid; idx < 64 * 64; idx += 512) {                                                        \
          const int kr = idx >> 6, c16 = idx & 63;                                                              \
          if (kr >= valid__) *reinterpret_cast<uint4*>(sp__ + kr * 512 + (c16 & 31) * 16 + (c16 >> 5) * 32768) = make_uint4(0, 0, 0, 0); \
        }                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                                         \
      }                                                                                                         \
    }                                                                                                           \
    TN2_SB; TN2_MFMA_ROW(4, ah_lo[0], ah_hi[0], BCL, BCH); TN2_SB;                                              \
    TN2_RD(al_lo[0], al_hi[0], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 0), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 0));         \
    TN2_RD(BNL[0], BNH[0], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 0), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 0));     \
    TN2_RD(al_lo[1], al_hi[1], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 1), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 1));         \
    if (SYNC && spread__) { if (TN2_SPREAD == 2) TN2_ONE_B(ST, 0); else TN2_PIECE(ST, 1); }                     \
    TN2_SB; TN2_MFMA_ROW(5, ah_lo[1], ah_hi[1], BCL, BCH); TN2_SB;                                              \
    TN2_RD(BNL[1], BNH[1], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 1), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 1));     \
    TN2_RD(al_lo[2], al_hi[2], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 2), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 2));         \
    TN2_RD(BNL[2], BNH[2], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 2), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 2));     \
    if (SYNC && spread__) { if (TN2_SPREAD == 2) TN2_ONE_A(ST, 1); else TN2_PIECE(ST, 2); }                     \
    TN2_SB; TN2_MFMA_ROW(6, ah_lo[2], ah_hi[2], BCL, BCH); TN2_SB;                                              \
    TN2_RD(al_lo[3], al_hi[3], TN2_ADDR_LO(nxt__, 0u, wr * 8 + 3), TN2_ADDR_HI(nxt__, 0u, wr * 8 + 3));         \
    TN2_RD(BNL[3], BNH[3], TN2_ADDR_LO(nxt__, 32768u, wc * 4 + 3), TN2_ADDR_HI(nxt__, 32768u, wc * 4 + 3));     \
    if (SYNC && spread__) { if (TN2_SPREAD == 2) TN2_ONE_B(ST, 1); else TN2_PIECE(ST, 3); }                     \
    TN2_SB; TN2_MFMA_ROW(7, ah_lo[3], ah_hi[3], BCL, BCH);                                                      \
    TN2_SB;                                                                                                     \
    TN2_PIN8("s_waitcnt lgkmcnt(0)", al_lo, al_hi);                                                             \
    TN2_PIN8("s_waitcnt lgkmcnt(0)", BNL, BNH);                                                                 \
  } while (0)

  if (nst == 1 && (ke - kb) % 64 != 0) {
    // single ragged stage: zero the tail rows before anything is consumed (fragments above are re-read)
    const int valid = ke - kb;
    for (int idx = tid; idx < 64 * 64; idx += 512) {
      const int kr = idx >> 6, c16 = idx & 63;
      if (kr >= valid) *reinterpret_cast<uint4*>(smem + kr * 512 + (c16 & 31) * 16 + (c16 >> 5) * 32768) = make_uint4(0, 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      TN2_RD(al_lo[i], al_hi[i], TN2_ADDR_LO(lds0, 0u, wr * 8 + i), TN2_ADDR_HI(lds0, 0u, wr * 8 + i));
      TN2_RD(bx_lo[i], bx_hi[i], TN2_ADDR_LO(lds0, 32768u, wc * 4 + i), TN2_ADDR_HI(lds0, 32768u, wc * 4 + i));
    }
    TN2_PIN8("s_waitcnt lgkmcnt(0)", al_lo, al_hi);
    TN2_PIN8("s_waitcnt lgkmcnt(0)", bx_lo, bx_hi);
  }
  bool pend = false;                      // TN2_SPREAD == 2: half of the last boundary's stage is still to be issued
  for (int st = 0; st < nst; ++st) {
    const uint32_t sc = lds0 + (uint32_t)((st & 1) * TN2_STAGE), sn = lds0 + (uint32_t)(((st + 1) & 1) * TN2_STAGE);
    do_bias = dbias != nullptr && (st % bias_mod) == bias_slot;
    TN2_STEP(sc, sc + 32 * 512, bx_lo, bx_hi, by_lo, by_hi, false, st);
    TN2_STEP(sc + 32 * 512, sn, by_lo, by_hi, bx_lo, bx_hi, true, st);
  }
#undef TN2_STEP

  float* o = out + (long)split * slab_stride;
  const float inv_b = gs_inv(gs), inv = scale_out ? inv_b : 1.0f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + wr * 128 + i * 16 + fs;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fg * 4;
      if (n >= N) continue;
      float4* dst = reinterpret_cast<float4*>(o + (long)m * ldo + n);
      float4 v = make_float4(acc[i][j][0] * inv, acc[i][j][1] * inv, acc[i][j][2] * inv, acc[i][j][3] * inv);
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
        else atomicAdd(dbias + m, t * inv_b);
      }
    }
  }
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int splits,
                                                          int M, int N, float* __restrict__ C, int ldc, int accumulate,
                                                          const float* __restrict__ gs) {
  const long total4 = (long)M * N / 4;
  const float inv = gs_inv(gs);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    const int m = (int)(e / N), n = (int)(e % N);
    float4 s = make_float4(0, 0, 0, 0);
    for (int k = 0; k < splits; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(slabs + (long)k * slab_stride + e);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= inv; s.y *= inv; s.z *= inv; s.w *= inv;
    float4* dst = reinterpret_cast<float4*>(C + (long)m * ldc + n);
    if (accumulate) { const float4 c = *dst; s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w; }
    *dst = s;
  }
}

// ------------------------------------------------------------------------------------------
// column sums (bias gradients) and weight casts
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(int Trows, int N, const T* __restrict__ x, int ldx,
                                                     float* __restrict__ out, int rows_per_block, const float* __restrict__ gs,
                                                     float* __restrict__ det_part) {
  // block = 64 columns x 4 row-lanes; grid.x over column groups, grid.y over row chunks; atomics to out (det_part: one
  // partial row per row chunk instead, added in chunk order by det_reduce_kernel)
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_block, r1 = min(Trows, r0 + rows_per_block);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) s += Elem<T>::ld(x + (long)r * ldx + c);
  part[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N) {
    const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
    if (det_part) det_part[(long)blockIdx.y * N + c] = t;
    else atomicAdd(out + c, t * gs_inv(gs));
  }
}

template <typename TO>
__global__ __launch_bounds__(256) void cast_weight_kernel(const float* __restrict__ w, int rows, int cols,
                                                          TO* __restrict__ o, TO* __restrict__ ot) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < rows && c < cols) { v = w[(long)r * cols + c]; if (o) Elem<TO>::st(o + (long)r * cols + c, v); }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (ot)
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (r < rows && c < cols) Elem<TO>::st(ot + (long)c * rows + r, tile[tx][i]);
    }
}

// ------------------------------------------------------------------------------------------
// host entry points
// ------------------------------------------------------------------------------------------
// Per-device launch state (simx.h: "per-device caches are initialised once per device, thread-safely"): the CU count that
// sizes the persistent grids and the dynamic-LDS limits of the kernels that need more than 64 KB.  hipFuncSetAttribute is
// a per-device setting, so every device the process touches gets its own std::call_once.
struct GemmDevice { std::once_flag once; int ncu = 0; bool ok = false; };
static GemmDevice g_gemm_dev[SIMX_MAX_DEVICES];
static const GemmDevice* gemm_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SIMX_MAX_DEVICES) return nullptr;
  GemmDevice& g = g_gemm_dev[dev];
  std::call_once(g.once, [&g, dev] {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return;
    g.ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    g.ncu -= g.ncu % 8;                           // whole XCD rows: xcd_remap assumes block b -> XCD b % 8
    bool ok = true;
#define SIMX_LDS_ATTR(KERNEL, BYTES) \
    ok = ok && hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(BYTES)) == hipSuccess
#define SIMX_LDS_ATTRS16(F)                                                           \
    SIMX_LDS_ATTR((gemm_nt_h16_kernel<F, SIMX_EPI_NONE>), 2 * NT_STAGE_BYTES);           \
    SIMX_LDS_ATTR((gemm_nt_h16_kernel<F, SIMX_EPI_GELU>), 2 * NT_STAGE_BYTES);           \
    SIMX_LDS_ATTR((gemm_nt_h16_kernel<F, SIMX_EPI_DGELU>), 2 * NT_STAGE_BYTES);          \
    SIMX_LDS_ATTR(gemm_tn_h16_kernel<F>, 2 * NT_STAGE_BYTES);                            \
    SIMX_LDS_ATTR((gemm_nt_h16_v5_kernel<F, SIMX_EPI_NONE>), V5_LDS);                    \
    SIMX_LDS_ATTR((gemm_nt_h16_v5_kernel<F, SIMX_EPI_GELU>), V5_LDS);                    \
    SIMX_LDS_ATTR((gemm_nt_h16_v5_kernel<F, SIMX_EPI_DGELU>), V5_LDS);                   \
    SIMX_LDS_ATTR((gemm_nt_p3_kernel<F, SIMX_EPI_NONE, false>), P_LDS);                  \
    SIMX_LDS_ATTR((gemm_nt_p3_kernel<F, SIMX_EPI_NONE, true>), P_LDS);                   \
    SIMX_LDS_ATTR((gemm_nt_p3_kernel<F, SIMX_EPI_GELU, false>), P_LDS);                  \
    SIMX_LDS_ATTR((gemm_nt_p3_kernel<F, SIMX_EPI_DGELU, true>), P_LDS);                  \
    SIMX_LDS_ATTR((gemm_nt_p3_kernel<F, SIMX_EPI_NONE, false, 2>), P_LDS);               \
    SIMX_LDS_ATTR((gemm_nt_p3_kernel<F, SIMX_EPI_NONE, true, 1>), P_LDS);                \
    SIMX_LDS_ATTR(gemm_tn2_kernel<F>, TN2_LDS)
    SIMX_LDS_ATTRS16(bf16_t);
    SIMX_LDS_ATTRS16(f16_t);
#undef SIMX_LDS_ATTRS16
#undef SIMX_LDS_ATTR
    g.ok = ok;
  });
  return g.ok ? &g : nullptr;
}

static int operand_mode(const float* p, long stride_mn, long stride_k) {
  // 1: k contiguous, 2: m/n contiguous (16-B loads need a 16-B aligned base and leading dimension), 0: scalar
  const bool al16 = (((uintptr_t)p) & 15) == 0;
  if (stride_k == 1 && al16 && stride_mn % 4 == 0) return 1;
  if (stride_mn == 1 && al16 && stride_k % 4 == 0) return 2;
  return 0;
}
// Split-K plan of the f32 MFMA kernel for problems whose 128x128 tile grid leaves most of the chip idle while K is long
// (parity-mode wgrad: K = tokens; M2's dQ: K = the gathered passages).  Needs a workspace of splits*M*N floats.
static int f32_splits(int M, int N, int K, size_t ws_bytes) {
  const long blocks = (long)cdiv(M, 128) * cdiv(N, 128);
  if (blocks >= 256 || K < 1024 || (N % 4) != 0) return 1;
  long sp = 512 / blocks;
  if (sp > K / 256) sp = K / 256;
  const long cap = (long)(ws_bytes / ((size_t)M * N * sizeof(float)));
  if (sp > cap) sp = cap;
  return sp < 2 ? 1 : (int)sp;
}
extern "C" size_t simx_gemm_f32_workspace_bytes(int M, int N, int K) {
  const int sp = f32_splits(M, N, K, (size_t)1 << 62);
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}
static int launch_f32_mfma(hipStream_t s, int epi, int M, int N, int K, const float* A, long a_rs, long a_cs, const float* B,
                           long b_ks, long b_ns, float* C, int ldc, const float* bias, const float* res, int ldr,
                           const float* aux, int ldaux, float* C2, int ldc2, int accumulate, DropCtx drop, void* ws = nullptr,
                           size_t ws_bytes = 0) {
  dim3 grid(cdiv(N, 128), cdiv(M, 128));
  const int am = operand_mode(A, a_rs, a_cs), bm = operand_mode(B, b_ns, b_ks);
  if (epi == SIMX_EPI_NONE && !bias && !res && !drop.thr && ws && (((uintptr_t)ws) & 15) == 0 && ldc % 4 == 0 &&
      (((uintptr_t)C) & 15) == 0) {
    const int sp = f32_splits(M, N, K, ws_bytes);
    if (sp > 1) {
      const int kps = cdiv(cdiv(K, sp), 16) * 16;
      const int nsp = cdiv(K, kps);
      hipLaunchKernelGGL((gemm_f32_mfma_kernel<SIMX_EPI_NONE>), dim3(grid.x, grid.y, nsp), dim3(256), 0, s, M, N, K, A, a_rs, a_cs, B, b_ks,
                         b_ns, (float*)ws, N, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, 0, drop, am, bm, kps, (long)M * N);
      SIMX_CHECK_LAUNCH("gemm_f32_mfma(split)");
      const long tot4 = (long)M * N / 4;
      int rb = (int)((tot4 + 255) / 256);
      if (rb > 2048) rb = 2048;
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, s, (const float*)ws, (long)M * N, nsp, M, N, C, ldc, accumulate,
                         (const float*)nullptr);
      SIMX_CHECK_LAUNCH("slab_reduce");
      return SIMX_OK;
    }
  }
#define LF(E) hipLaunchKernelGGL((gemm_f32_mfma_kernel<E>), grid, dim3(256), 0, s, M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc, bias, \
                                 res, ldr, aux, ldaux, C2, ldc2, accumulate, drop, am, bm, 0, 0L)
  if (epi == SIMX_EPI_NONE) LF(SIMX_EPI_NONE);
  else if (epi == SIMX_EPI_GELU) LF(SIMX_EPI_GELU);
  else LF(SIMX_EPI_DGELU);
#undef LF
  SIMX_CHECK_LAUNCH("gemm_f32_mfma");
  return SIMX_OK;
}

template <typename TI, typename TO>
static int launch_simple(hipStream_t s, int epi, int M, int N, int K, const TI* A, long a_rs, long a_cs,
                         const TI* B, long b_ks, long b_ns, TO* C, int ldc, const float* bias, const TI* res, int ldr,
                         const TI* aux, int ldaux, TO* C2, int ldc2, int accumulate, DropCtx drop = DropCtx{0u, 1.f, 0u, 0u},
                         void* ws = nullptr, size_t ws_bytes = 0, const float* gs = nullptr) {
  if constexpr (std::is_same<TI, float>::value && std::is_same<TO, float>::value) {
    static const char* pin = getenv("SIMX_GEMM_F32");           // SIMX_GEMM_F32=fma pins the VALU kernel (A/B measurements)
    if (!(pin && pin[0] == 'f'))
      return launch_f32_mfma(s, epi, M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc, bias, res, ldr, aux, ldaux, C2, ldc2, accumulate, drop,
                             ws, ws_bytes);
  }
  dim3 grid(cdiv(N, 64), cdiv(M, 64));
#define L(E) hipLaunchKernelGGL((gemm_simple_kernel<TI, TO, E>), grid, dim3(256), 0, s, M, N, K, A, a_rs, a_cs, B, b_ks, \
                                b_ns, C, ldc, bias, res, ldr, aux, ldaux, C2, ldc2, accumulate, drop, gs)
  if (epi == SIMX_EPI_NONE) L(SIMX_EPI_NONE);
  else if (epi == SIMX_EPI_GELU) L(SIMX_EPI_GELU);
  else L(SIMX_EPI_DGELU);
#undef L
  SIMX_CHECK_LAUNCH("gemm_simple");
  return SIMX_OK;
}

static bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

extern "C" int simx_gemm_nt_ex(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda,
                               const void* B, int ldb, void* C, int ldc, const float* bias, const void* residual,
                               int ldr, int epilogue, const void* aux, int ldaux, void* C2, int ldc2,
                               const simx_dropout* dropd);
// csrc/gemm_p5.hip: the persistent kernel whose epilogue runs under the next tile's main loop (plain bias epilogue only so far).
// SIMX_P5=0 keeps every launch on gemm_nt_p3_kernel (A/B measurements, tests/test_kernels_gpu.py compares the two bit for bit).
int simx_launch_nt_p5(hipStream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                      const float* bias, int hm_c_rows, int ncu);
// Default: the shapes where it measures faster -- K >= 1536 (FFN-out geometry: 0.877 vs 0.914 ms at M = 262144, N = 768, K = 3072;
// at K = 768 the two tie, profiles/r06_experiments/01_p5.md).  SIMX_P5=1: every eligible launch, SIMX_P5=0: none.
static bool p5_enabled(int K) {
  const char* e = getenv("SIMX_P5");           // (read per call: the kernel test switches it inside one process)
  if (e && e[0] == '0') return false;
  if (e && e[0] == '1') return true;
  return K >= 1536;
}
extern "C" int simx_gemm_nt(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda,
                            const void* B, int ldb, void* C, int ldc, const float* bias, const void* residual,
                            int ldr, int epilogue, const void* aux, int ldaux, void* C2, int ldc2) {
  return simx_gemm_nt_ex(stream, dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, epilogue, aux, ldaux, C2, ldc2,
                         nullptr);
}

extern "C" int simx_gemm_nt_ex(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda,
                               const void* B, int ldb, void* C, int ldc, const float* bias, const void* residual,
                               int ldr, int epilogue, const void* aux, int ldaux, void* C2, int ldc2,
                               const simx_dropout* dropd) {
  hipStream_t s = (hipStream_t)stream;
  const DropCtx drop = make_drop(epilogue == SIMX_EPI_NONE ? dropd : nullptr);
  SIMX_REQUIRE(!dropd || (dropd->p >= 0.f && dropd->p < 1.f), SIMX_ERR_BAD_SHAPE, "gemm_nt: dropout p must be in [0,1)");
  SIMX_PROF(SIMX_K_GEMM_NT, s, 2.0 * M * N * K);
  SIMX_REQUIRE(M > 0 && N > 0 && K > 0, SIMX_ERR_BAD_SHAPE, "gemm_nt: bad shape %d %d %d", M, N, K);
  SIMX_REQUIRE(lda >= K && ldb >= K && ldc >= N, SIMX_ERR_BAD_SHAPE, "gemm_nt: leading dims too small");
  const bool gelu_infer = epilogue == SIMX_EPI_GELU_INFER;     // C is scratch: the kernel may skip writing the pre-activation
  if (gelu_infer) epilogue = SIMX_EPI_GELU;
  SIMX_REQUIRE(epilogue >= 0 && epilogue <= 2, SIMX_ERR_UNSUPPORTED, "gemm_nt: epilogue %d", epilogue);
  SIMX_REQUIRE(epilogue != SIMX_EPI_GELU || C2, SIMX_ERR_BAD_SHAPE, "gemm_nt: GELU epilogue needs C2");
  SIMX_REQUIRE(epilogue != SIMX_EPI_DGELU || aux, SIMX_ERR_BAD_SHAPE, "gemm_nt: DGELU epilogue needs aux");
  if (simx_is_f32(dtype)) {
    if (dtype != SIMX_F32 && simx_x3_nt_ok(M, N, K, (const float*)A, lda, (const float*)B, ldb, (const float*)C, ldc, bias, (const float*)residual,
                                           ldr, (const float*)aux, ldaux, (const float*)C2, ldc2))
      return simx_x3_gemm_nt(s, dtype == SIMX_F32_SPLIT_H ? SIMX_F16 : SIMX_BF16, epilogue, M, N, K, (const float*)A, lda, (const float*)B, ldb,
                             (float*)C, ldc, bias, (const float*)residual, ldr, (const float*)aux, ldaux, (float*)C2, ldc2, drop);
    return launch_simple<float, float>(s, epilogue, M, N, K, (const float*)A, lda, 1, (const float*)B, 1, ldb,
                                       (float*)C, ldc, bias, (const float*)residual, ldr, (const float*)aux, ldaux,
                                       (float*)C2, ldc2, 0, drop);
  }
  SIMX_REQUIRE(simx_is16(dtype), SIMX_ERR_BAD_DTYPE, "gemm_nt: dtype %d", dtype);
  const bool fast = (K % 64 == 0) && (N % 4 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) && (ldc % 4 == 0) &&
                    aligned16(A) && aligned16(B) && aligned16(C) && (!bias || aligned16(bias)) &&
                    (!residual || (ldr % 4 == 0 && aligned16(residual))) && (!aux || (ldaux % 4 == 0 && aligned16(aux))) &&
                    (!C2 || (ldc2 % 4 == 0 && aligned16(C2)));
  if (!fast) {
    int rcs = SIMX_OK;
    SIMX_DISPATCH16(dtype, TT, rcs = (launch_simple<TT, TT>(s, epilogue, M, N, K, (const TT*)A, lda, 1, (const TT*)B, 1, ldb, (TT*)C, ldc, bias,
                                                            (const TT*)residual, ldr, (const TT*)aux, ldaux, (TT*)C2, ldc2, 0, drop)));
    return rcs;
  }
  const int tiles_m = cdiv(M, NT_BM), tiles_n = cdiv(N, NT_BN), nwg = tiles_m * tiles_n;
  const size_t lds = 2 * NT_STAGE_BYTES;
  const GemmDevice* gd = gemm_device();
  SIMX_REQUIRE(gd != nullptr, SIMX_ERR_HIP, "gemm_nt: cannot query the current device");
  {
    // SIMX_GEMM pins a kernel for A/B measurements: v1 = the 128x128 kernel, v5 = the per-tile 256x256 kernel
    static const char* pin = getenv("SIMX_GEMM");
    const bool force_v1 = pin && pin[1] == '1', force_v5 = pin && pin[1] == '5';
    const int t3m = cdiv(M, 256), t3n = cdiv(N, 256), nwg3 = t3m * t3n;
    // full 256x256 tiles, K >= 256: the persistent kernel, one workgroup per CU
    if (!force_v1 && !force_v5 && nwg3 >= 192 && M % 256 == 0 && N % 256 == 0 && K >= 256 && ldc % 8 == 0 &&
        (!residual || ldr % 8 == 0) && (!aux || ldaux % 8 == 0) && (!C2 || ldc2 % 8 == 0) &&
        !(epilogue == SIMX_EPI_DGELU && residual) && !(epilogue == SIMX_EPI_GELU && residual)) {
      const int ncu = simx_compute_cus(gd->ncu);
      const int grid = nwg3 < ncu ? nwg3 : ncu;
      if (epilogue == SIMX_EPI_NONE && !residual && !drop.thr && p5_enabled(K) &&
          simx_launch_nt_p5(s, dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, 0, ncu) == SIMX_OK) {
        simx_prof_retag(SIMX_K_GEMM_NT_P3);
        return SIMX_OK;
      }
#ifdef SIMX_MEASUREMENT_HOOKS                   /* tools/build_variant.sh builds only: SIMX_NOEPI=1 times the main loop alone */
      static const bool noepi_p = getenv("SIMX_NOEPI") != nullptr;
      if (noepi_p) C = nullptr;
#endif
#define LP3(E, HI, INP, LDI) hipLaunchKernelGGL((gemm_nt_p3_kernel<FF, E, HI>), dim3(grid), dim3(512), P_LDS, s, M, N, K, (const bf16_t*)A, lda, \
                                 (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, (const bf16_t*)(INP), LDI, (bf16_t*)C2, ldc2, t3n, nwg3, drop)
#define LP3_ALL()                                                                                                               \
  do {                                                                                                                          \
    if (epilogue == SIMX_EPI_NONE) { if (residual) LP3(SIMX_EPI_NONE, true, residual, ldr); else LP3(SIMX_EPI_NONE, false, nullptr, 0); } \
    else if (epilogue == SIMX_EPI_GELU) LP3(SIMX_EPI_GELU, false, nullptr, gelu_infer ? 1 : 0);                                 \
    else LP3(SIMX_EPI_DGELU, true, aux, ldaux);                                                                                 \
  } while (0)
      SIMX_DISPATCH16(dtype, FF, LP3_ALL());
#undef LP3_ALL
#undef LP3
      simx_prof_retag(SIMX_K_GEMM_NT_P3);
      SIMX_CHECK_LAUNCH("gemm_nt_p3");
      return SIMX_OK;
    }
    // large ragged problems: the per-tile 256x256x64 two-stage kernel
    if (!force_v1 && nwg3 >= 192 && N % 8 == 0 && ldc % 8 == 0 && (!residual || ldr % 8 == 0) &&
        (!aux || ldaux % 8 == 0) && (!C2 || ldc2 % 8 == 0)) {
#ifdef SIMX_MEASUREMENT_HOOKS
      static const bool noepi = getenv("SIMX_NOEPI") != nullptr;
      if (noepi) C = nullptr;
#endif
#define L5(E) hipLaunchKernelGGL((gemm_nt_h16_v5_kernel<FF, E>), dim3(nwg3), dim3(512), V5_LDS, s, M, N, K, (const bf16_t*)A, lda, \
                                 (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, (const bf16_t*)residual, ldr,                  \
                                 (const bf16_t*)aux, ldaux, (bf16_t*)C2, ldc2, t3n, nwg3, drop)
#define L5_ALL() do { if (epilogue == SIMX_EPI_NONE) L5(SIMX_EPI_NONE); else if (epilogue == SIMX_EPI_GELU) L5(SIMX_EPI_GELU); else L5(SIMX_EPI_DGELU); } while (0)
      SIMX_DISPATCH16(dtype, FF, L5_ALL());
#undef L5_ALL
#undef L5
      SIMX_CHECK_LAUNCH("gemm_nt_h16_v5");
      return SIMX_OK;
    }
  }
#define L(E) hipLaunchKernelGGL((gemm_nt_h16_kernel<FF, E>), dim3(nwg), dim3(256), lds, s, M, N, K, (const bf16_t*)A, lda, \
                                (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, (const bf16_t*)residual, ldr,               \
                                (const bf16_t*)aux, ldaux, (bf16_t*)C2, ldc2, tiles_n, nwg, drop)
#define L_ALL() do { if (epilogue == SIMX_EPI_NONE) L(SIMX_EPI_NONE); else if (epilogue == SIMX_EPI_GELU) L(SIMX_EPI_GELU); else L(SIMX_EPI_DGELU); } while (0)
  SIMX_DISPATCH16(dtype, FF, L_ALL());
#undef L_ALL
#undef L
  SIMX_CHECK_LAUNCH("gemm_nt_h16");
  return SIMX_OK;
}

// Head-major q / k / v operands (QkvLay, csrc/attention.hip) on the persistent kernel; bf16, full 256x256 tiles only:
//   c_hm_rows = R > 0: C is [N/64][R][64] (the QKV projection writes it; no residual);
//   a_hm_rows = R > 0: A is [K/64][R][64] (dq/dk/dv as the dgrad operand; `residual` required: it is the other gradient branch).
// simx_gemm_hm_ok tells a caller whether both forms (and the wgrad form, simx_gemm_tn_hm) run for its shapes.
extern "C" int simx_gemm_hm_ok(int rows, int H, int tokens) {
  const GemmDevice* gd = gemm_device();
  if (!gd || rows <= 0 || H <= 0) return 0;
  const bool shapes = rows % 256 == 0 && H % 256 == 0 && H >= 256;
  const long dgrad_tiles = (long)(rows / 256) * (H / 256);            // the narrowest of the three GEMMs (N = H)
  return shapes && dgrad_tiles >= 192 && tokens >= 2048 ? 1 : 0;
}
// General plane-blocked form on the persistent kernel (bf16, full 256x256 tiles).  flags: bit 0 = A, bit 1 = C (and C2),
// bit 2 = `in` (the residual of SIMX_EPI_NONE / the stored derivative of SIMX_EPI_DGELU) are [cols/64][rows][64] tensors with
// planes of `rows` rows; the others are ordinary row-major tensors with their leading dimensions.  Built: SIMX_EPI_NONE with
// flags 1 (A, with a residual) or 2 (C, without).  The kernel template also has the GELU (flags 2) and DGELU (flags 6) forms
// for the FFN's [T, 3072] tensors; measured (tools/kbench, M = 262144) they gain 4 % and 1.5 % where the QKV projection gains
// 18 % (see bit 1 above: the row pitch, not locality as such) -- so they are not built in.  The same probes price the GELU
// epilogue: plain bias 1.056 ms, + gelu (inference form) 1.266, + gelu' as a second output 1.49.
extern "C" int simx_gemm_nt_pb(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                               void* C, int ldc, const float* bias, const void* in, int ldin, int epilogue, void* C2, int ldc2,
                               const simx_dropout* dropd, int flags, int rows) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_GEMM_NT, s, 2.0 * M * N * K);
  SIMX_REQUIRE(simx_is16(dtype), SIMX_ERR_BAD_DTYPE, "gemm_nt_pb: 16-bit dtypes only");
  SIMX_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, SIMX_ERR_BAD_SHAPE, "gemm_nt_pb: bad arguments");
  const bool infer = epilogue == SIMX_EPI_GELU_INFER;
  if (infer) epilogue = SIMX_EPI_GELU;
  const bool combo = epilogue == SIMX_EPI_NONE && ((flags == 1 && in) || (flags == 2 && !in));
  SIMX_REQUIRE(combo, SIMX_ERR_UNSUPPORTED, "gemm_nt_pb: epilogue %d with plane flags %d is not built", epilogue, flags);
  const GemmDevice* gd = gemm_device();
  SIMX_REQUIRE(gd != nullptr, SIMX_ERR_HIP, "gemm_nt_pb: cannot query the current device");
  const int t3n = N / 256, nwg3 = (M / 256) * t3n;
  const bool pa = flags & 1, pc = flags & 2, pi = flags & 4;
  const bool ok = M % 256 == 0 && N % 256 == 0 && K % 64 == 0 && K >= 256 && nwg3 >= 192 && rows >= M && ldb % 8 == 0 && ldb >= K &&
                  aligned16(A) && aligned16(B) && aligned16(C) && (!C2 || aligned16(C2)) && (!bias || aligned16(bias)) &&
                  (!in || aligned16(in)) && (pa || (lda % 8 == 0 && lda >= K)) && (pc || (ldc % 8 == 0 && ldc >= N)) &&
                  (!in || pi || (ldin % 8 == 0 && ldin >= N)) && (!C2 || pc || (ldc2 % 8 == 0 && ldc2 >= N));
  SIMX_REQUIRE(ok, SIMX_ERR_UNSUPPORTED, "gemm_nt_pb: shape %d x %d x %d (planes of %d rows) is outside the persistent kernel's rules", M, N, K, rows);
  const DropCtx drop = make_drop(epilogue == SIMX_EPI_NONE ? dropd : nullptr);
  const int ncu = simx_compute_cus(gd->ncu);
  const int grid = nwg3 < ncu ? nwg3 : ncu;
  if (flags == 2 && !drop.thr && p5_enabled(K) && simx_launch_nt_p5(s, dtype, M, N, K, A, lda, B, ldb, C, 64, bias, rows, ncu) == SIMX_OK) {
    simx_prof_retag(SIMX_K_GEMM_NT_P3);
    return SIMX_OK;
  }
#define LPB(E, HI, PF, INP, LDI) hipLaunchKernelGGL((gemm_nt_p3_kernel<FF, E, HI, PF>), dim3(grid), dim3(512), P_LDS, s, M, N, K, (const bf16_t*)A, lda, \
                                 (const bf16_t*)B, ldb, (bf16_t*)C, ldc, bias, (const bf16_t*)(INP), LDI, (bf16_t*)C2, rows, t3n, nwg3, drop)
  (void)infer;
  SIMX_DISPATCH16(dtype, FF, if (flags == 1) LPB(SIMX_EPI_NONE, true, 1, in, ldin); else LPB(SIMX_EPI_NONE, false, 2, nullptr, 0));
#undef LPB
  simx_prof_retag(SIMX_K_GEMM_NT_P3);
  SIMX_CHECK_LAUNCH("gemm_nt_p3(pb)");
  return SIMX_OK;
}
// the two q/k/v forms under their first names
extern "C" int simx_gemm_nt_hm(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                               void* C, int ldc, const float* bias, const void* residual, int ldr, const simx_dropout* dropd,
                               int a_hm_rows, int c_hm_rows) {
  SIMX_REQUIRE((a_hm_rows > 0) != (c_hm_rows > 0), SIMX_ERR_BAD_SHAPE, "gemm_nt_hm: exactly one of A / C is head-major");
  return simx_gemm_nt_pb(stream, dtype, M, N, K, A, lda, B, ldb, C, ldc, bias, residual, ldr, SIMX_EPI_NONE, nullptr, 0, dropd,
                         a_hm_rows > 0 ? 1 : 2, a_hm_rows > 0 ? a_hm_rows : c_hm_rows);
}

static bool tn_use_v2(int M, int N, int K) {
  static const char* pin = getenv("SIMX_GEMM_TN");
  if (pin && pin[1] == '1') return false;
  return M >= 256 && N >= 256 && K >= 2048;
}
static void tn_plan(int M, int N, int K, int* splits, int* k_per_split) {
  const bool v2 = tn_use_v2(M, N, K);
  const int tiles = v2 ? cdiv(M, 256) * cdiv(N, 256) : cdiv(M, 128) * cdiv(N, 128);
  // whole rounds of the chip: v2 runs one workgroup per CU (256 per round), v1 two.  Rounding the split count UP
  // (513 workgroups for wqkv/wo, 540 for w1/w2) left a third, almost empty round; rounds 2-4 filled exactly TWO rounds; ONE
  // round (half the splits: each workgroup pays its prologue, its 256 KB slab epilogue and its share of the slab pass once
  // for twice the tokens) measures better at every size -- 262144 tokens 3.54 -> 3.37 ms over the four wgrad shapes, 32768
  // tokens 0.61 -> 0.52, 16384 tokens 0.41 -> 0.32 (tools/kbench, profiles/r05_experiments/09).  SIMX_TN_ROUNDS=2: the old rule.
  const char* rounds_env = getenv("SIMX_TN_ROUNDS");       // (read per call: tests/test_fullsize_gpu.py covers both rules in one process)
  const int round = simx_compute_cus(256);      // one workgroup per CU the launch may fill (256 unless a compute-CU budget is set)
  int s = (v2 ? (rounds_env && rounds_env[0] == '2' ? 2 * round : round) : 1024) / tiles;
  const int max_s = cdiv(K, 512);          // at least 8 k-tiles per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  int kps = cdiv(cdiv(K, s), 64) * 64;
  s = cdiv(K, kps);
  *splits = s;
  *k_per_split = kps;
}

extern "C" size_t simx_gemm_tn_workspace_bytes(int M, int N, int K) {
  int s, kps;
  tn_plan(M, N, K, &s, &kps);
  const size_t bf = s > 1 ? (size_t)s * M * N * sizeof(float) : 0, f32 = simx_gemm_f32_workspace_bytes(M, N, K);
  const size_t x3 = simx_x3_tn_workspace_bytes(M, N, K);
  const size_t m2 = bf > f32 ? bf : f32;
  return m2 > x3 ? m2 : x3;                     // (the dtype is not an argument: enough for any of them)
}

int simx_launch_tn5(hipStream_t s, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* slabs,
                    int splits, int kps, float* dbias, int a_hm_rows, const float* gs);
static int gemm_tn_impl(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                        int ldc, int accumulate, void* ws, size_t ws_bytes, float* dbias, int a_hm_rows, const float* gs);
extern "C" int simx_gemm_tn(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda,
                            const void* B, int ldb, float* C, int ldc, int accumulate, void* ws, size_t ws_bytes) {
  return gemm_tn_impl(stream, dtype, M, N, K, A, lda, B, ldb, C, ldc, accumulate, ws, ws_bytes, nullptr, 0, nullptr);
}
extern "C" int simx_gemm_tn_bias(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda,
                                 const void* B, int ldb, float* C, int ldc, int accumulate, void* ws, size_t ws_bytes,
                                 float* dbias) {
  return gemm_tn_impl(stream, dtype, M, N, K, A, lda, B, ldb, C, ldc, accumulate, ws, ws_bytes, dbias, 0, nullptr);
}
// wgrad with a head-major A (dq/dk/dv planes of a_hm_rows rows, [M/64][R][64]); large 16-bit shapes only (the 256x256 kernel)
extern "C" int simx_gemm_tn_hm(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int a_hm_rows, const void* B,
                               int ldb, float* C, int ldc, int accumulate, void* ws, size_t ws_bytes, float* dbias) {
  return simx_gemm_tn_gs(stream, dtype, M, N, K, A, M, B, ldb, C, ldc, accumulate, ws, ws_bytes, dbias, a_hm_rows, nullptr);
}
// every form of the wgrad GEMM in one call, with the gradient scale of the fp16 backward (A = S x dY): C and dbias receive
// (1/S) x the products.  a_hm_rows > 0: A is head-major (lda ignored).
extern "C" int simx_gemm_tn_gs(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                               float* C, int ldc, int accumulate, void* ws, size_t ws_bytes, float* dbias, int a_hm_rows,
                               const float* gs) {
  if (a_hm_rows > 0) {
    SIMX_REQUIRE(simx_is16(dtype) && a_hm_rows >= K && M % 256 == 0 && N % 8 == 0 && K >= 2048, SIMX_ERR_UNSUPPORTED,
                 "gemm_tn_hm: needs a 16-bit dtype, M %% 256 == 0, K >= 2048 tokens and planes of >= K rows");
    lda = M;
  }
  return gemm_tn_impl(stream, dtype, M, N, K, A, lda, B, ldb, C, ldc, accumulate, ws, ws_bytes, dbias, a_hm_rows, gs);
}
static int gemm_tn_impl(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb, float* C,
                        int ldc, int accumulate, void* ws, size_t ws_bytes, float* dbias, int a_hm_rows, const float* gs) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_GEMM_TN, s, 2.0 * M * N * K);
  SIMX_REQUIRE(M > 0 && N > 0 && K > 0, SIMX_ERR_BAD_SHAPE, "gemm_tn: bad shape %d %d %d", M, N, K);
  SIMX_REQUIRE(lda >= M && ldb >= N && ldc >= N, SIMX_ERR_BAD_SHAPE, "gemm_tn: leading dims too small");
  if (simx_is_f32(dtype)) {
    SIMX_REQUIRE(gs == nullptr, SIMX_ERR_UNSUPPORTED, "gemm_tn: the f32 engine carries no gradient scale");
    if (dtype != SIMX_F32 && simx_x3_tn_ok(M, N, K, (const float*)A, lda, (const float*)B, ldb, C, ldc))   // (bias gradient fused)
      return simx_x3_gemm_tn(s, dtype == SIMX_F32_SPLIT_H ? SIMX_F16 : SIMX_BF16, M, N, K, (const float*)A, lda, (const float*)B, ldb, C, ldc,
                             accumulate, ws, ws_bytes, dbias);
    if (dbias) { int rcb = simx_colsum_gs(stream, SIMX_F32, K, M, A, lda, dbias, 1, gs); if (rcb) return rcb; }
    return launch_simple<float, float>(s, SIMX_EPI_NONE, M, N, K, (const float*)A, 1, lda, (const float*)B, ldb, 1, C,
                                       ldc, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, accumulate, DropCtx{0u, 1.f, 0u, 0u}, ws, ws_bytes);
  }
  SIMX_REQUIRE(simx_is16(dtype), SIMX_ERR_BAD_DTYPE, "gemm_tn: dtype %d", dtype);
  const bool fast = (M % 8 == 0) && (N % 8 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) && (ldc % 4 == 0) && aligned16(A) &&
                    aligned16(B) && aligned16(C);
  SIMX_REQUIRE(a_hm_rows == 0 || fast, SIMX_ERR_UNSUPPORTED, "gemm_tn_hm: operands not 16-B aligned");
  if (!fast) {
    if (dbias) { int rcb = simx_colsum_gs(stream, dtype, K, M, A, lda, dbias, 1, gs); if (rcb) return rcb; }
    // generic path: 16-bit in, f32 out
    dim3 grid(cdiv(N, 64), cdiv(M, 64));
    SIMX_DISPATCH16(dtype, TT, hipLaunchKernelGGL((gemm_simple_kernel<TT, float, SIMX_EPI_NONE>), grid, dim3(256), 0, s, M, N, K,
                                                  (const TT*)A, 1L, (long)lda, (const TT*)B, (long)ldb, 1L, C, ldc, nullptr, (const TT*)nullptr, 0,
                                                  (const TT*)nullptr, 0, (float*)nullptr, 0, accumulate, DropCtx{0u, 1.f, 0u, 0u}, gs));
    SIMX_CHECK_LAUNCH("gemm_simple(tn)");
    return SIMX_OK;
  }
  int splits, kps;
  tn_plan(M, N, K, &splits, &kps);
  SIMX_REQUIRE(a_hm_rows == 0 || tn_use_v2(M, N, K), SIMX_ERR_UNSUPPORTED, "gemm_tn_hm: shape outside the 256x256 kernel's rules");
  if (tn_use_v2(M, N, K)) {
    SIMX_REQUIRE(gemm_device() != nullptr, SIMX_ERR_HIP, "gemm_tn: cannot query the current device");
    const int t_m = cdiv(M, 256), t_n = cdiv(N, 256), t_mn = t_m * t_n;
    simx_prof_retag(SIMX_K_GEMM_TN2);              // the large-shape wgrad kernel (with its slab pass) apart from the small launches
    float* dbias_out = dbias;
    const int nbp = splits * t_n * 4;
    if (dbias && simx_det()) {                    // ordered bias gradient: one partial row per contributing wave column
      dbias = simx_det_ws(s, (size_t)nbp * M * sizeof(float));
      if (!dbias) return SIMX_ERR_WORKSPACE;
    }
    const int dparts = dbias != dbias_out;
    if (splits == 1) {
      SIMX_DISPATCH16(dtype, FF, hipLaunchKernelGGL(gemm_tn2_kernel<FF>, dim3(t_mn), dim3(512), TN2_LDS, s, M, N, K, (const bf16_t*)A, lda,
                                                    (const bf16_t*)B, ldb, C, 0L, ldc, t_n, t_mn, kps, accumulate, dbias, a_hm_rows, gs, 1, dparts));
      SIMX_CHECK_LAUNCH("gemm_tn2");
      if (dparts) return simx_det_reduce(s, dbias, (long)M, nbp, M, dbias_out, nullptr, nullptr, gs);
      return SIMX_OK;
    }
    const size_t need2 = (size_t)splits * M * N * sizeof(float);
    SIMX_REQUIRE(ws && ws_bytes >= need2, SIMX_ERR_WORKSPACE, "gemm_tn: workspace %zu < %zu", ws_bytes, need2);
    SIMX_REQUIRE(aligned16(ws), SIMX_ERR_WORKSPACE, "gemm_tn: workspace not 16-B aligned");
    // csrc/gemm_tn5.hip: one wave per SIMD, AGPR accumulators, 96 KB in flight -- full tiles, whole stages, not the ordered mode.
    // SIMX_TN5=0 keeps gemm_tn2_kernel (A/B measurements; read per call: the kernel test compares the two in one process).
    {
      const char* e5 = getenv("SIMX_TN5");
      if (!(e5 && e5[0] == '0') && !dparts &&
          simx_launch_tn5(s, dtype, M, N, K, A, lda, B, ldb, (float*)ws, splits, kps, dbias, a_hm_rows, gs) == SIMX_OK) {
        const long tot4 = (long)M * N / 4;
        int rb = (int)((tot4 + 255) / 256);
        if (rb > 2048) rb = 2048;
        hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, s, (const float*)ws, (long)M * N, splits, M, N, C, ldc, accumulate, gs);
        SIMX_CHECK_LAUNCH("slab_reduce");
        return SIMX_OK;
      }
    }
    SIMX_DISPATCH16(dtype, FF, hipLaunchKernelGGL(gemm_tn2_kernel<FF>, dim3(t_mn * splits), dim3(512), TN2_LDS, s, M, N, K, (const bf16_t*)A, lda,
                                                  (const bf16_t*)B, ldb, (float*)ws, (long)M * N, N, t_n, t_mn, kps, 0, dbias, a_hm_rows, gs, 0, dparts));
    SIMX_CHECK_LAUNCH("gemm_tn2");
    if (dparts) { int rcd = simx_det_reduce(s, dbias, (long)M, nbp, M, dbias_out, nullptr, nullptr, gs); if (rcd) return rcd; }
    const long tot4 = (long)M * N / 4;
    int rb = (int)((tot4 + 255) / 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(rb), dim3(256), 0, s, (const float*)ws, (long)M * N, splits, M, N, C, ldc, accumulate, gs);
    SIMX_CHECK_LAUNCH("slab_reduce");
    return SIMX_OK;
  }
  if (dbias) {                                   // small problems: separate column-sum pass
    int rcb = simx_colsum_gs(stream, dtype, K, M, A, lda, dbias, 1, gs);
    if (rcb) return rcb;
  }
  const int tiles_m = cdiv(M, 128), tiles_n = cdiv(N, 128), tiles_mn = tiles_m * tiles_n;
  const size_t lds = 2 * NT_STAGE_BYTES;
  if (splits == 1) {
    SIMX_DISPATCH16(dtype, FF, hipLaunchKernelGGL(gemm_tn_h16_kernel<FF>, dim3(tiles_mn), dim3(256), lds, s, M, N, K, (const bf16_t*)A, lda,
                                                  (const bf16_t*)B, ldb, C, 0L, ldc, tiles_n, tiles_mn, kps, accumulate, gs));
    SIMX_CHECK_LAUNCH("gemm_tn_h16");
    return SIMX_OK;
  }
  const size_t need = (size_t)splits * M * N * sizeof(float);
  SIMX_REQUIRE(ws && ws_bytes >= need, SIMX_ERR_WORKSPACE, "gemm_tn: workspace %zu < %zu", ws_bytes, need);
  SIMX_REQUIRE(aligned16(ws), SIMX_ERR_WORKSPACE, "gemm_tn: workspace not 16-B aligned");
  SIMX_DISPATCH16(dtype, FF, hipLaunchKernelGGL(gemm_tn_h16_kernel<FF>, dim3(tiles_mn * splits), dim3(256), lds, s, M, N, K, (const bf16_t*)A, lda,
                                                (const bf16_t*)B, ldb, (float*)ws, (long)M * N, N, tiles_n, tiles_mn, kps, 0, (const float*)nullptr));
  SIMX_CHECK_LAUNCH("gemm_tn_h16");
  const long total4 = (long)M * N / 4;
  int blocks = (int)((total4 + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, (const float*)ws, (long)M * N, splits, M, N, C,
                     ldc, accumulate, gs);
  SIMX_CHECK_LAUNCH("slab_reduce");
  return SIMX_OK;
}

extern "C" int simx_colsum(simx_stream_t stream, int dtype, int T, int N, const void* x, int ldx, float* out,
                           int accumulate) {
  return simx_colsum_gs(stream, dtype, T, N, x, ldx, out, accumulate, nullptr);
}
extern "C" int simx_colsum_gs(simx_stream_t stream, int dtype, int T, int N, const void* x, int ldx, float* out,
                              int accumulate, const float* gs) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_COLSUM, s, (double)T * N * simx_esz(dtype));
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "colsum: dtype %d", dtype);
  SIMX_REQUIRE(T > 0 && N > 0 && ldx >= N, SIMX_ERR_BAD_SHAPE, "colsum: bad shape");
  if (!accumulate) {
    if (hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s) != hipSuccess) { simx_set_error("colsum: memset failed"); return SIMX_ERR_HIP; }
  }
  int chunks = cdiv(T, 512);
  if (chunks > 512) chunks = 512;
  const int rpb = cdiv(T, chunks);
  dim3 grid(cdiv(N, 64), cdiv(T, rpb));
  float* det = nullptr;
  if (simx_det()) {
    det = simx_det_ws(s, (size_t)grid.y * N * sizeof(float));
    if (!det) return SIMX_ERR_WORKSPACE;
  }
  SIMX_DISPATCH3(dtype, TT, hipLaunchKernelGGL((colsum_kernel<TT>), grid, dim3(256), 0, s, T, N, (const TT*)x, ldx, out, rpb, gs, det));
  SIMX_CHECK_LAUNCH("colsum");
  if (det) return simx_det_reduce(s, det, (long)N, (int)grid.y, N, out, nullptr, nullptr, gs);
  return SIMX_OK;
}

int simx_transpose_cast_group(hipStream_t s, int out_dtype, const SimxCastGroup* g);
extern "C" int simx_transpose_cast(simx_stream_t stream, int out_dtype, const float* w, int rows, int cols, void* out,
                                   void* outT) {
  SIMX_REQUIRE(rows > 0 && cols > 0 && w, SIMX_ERR_BAD_SHAPE, "transpose_cast: bad shape");
  SIMX_REQUIRE(simx_dtype_ok(out_dtype), SIMX_ERR_BAD_DTYPE, "transpose_cast: dtype %d", out_dtype);
  if (simx_is16(out_dtype) && rows % 64 == 0 && cols % 64 == 0) {      // the grouped path's 64 x 64-tile kernel, one job
    SimxCastGroup g{};
    g.n = 1;
    g.job[0] = SimxCastJob{w, out, outT, rows, cols, 0, 0};
    return simx_transpose_cast_group((hipStream_t)stream, out_dtype, &g);
  }
  SIMX_PROF(SIMX_K_CAST, stream, (double)rows * cols * 8);
  dim3 grid(cdiv(cols, 32), cdiv(rows, 32));
  SIMX_DISPATCH3(out_dtype, TT, hipLaunchKernelGGL((cast_weight_kernel<TT>), grid, dim3(256), 0, (hipStream_t)stream, w, rows, cols, (TT*)out,
                                                   (TT*)outT));
  SIMX_CHECK_LAUNCH("cast_weight");
  return SIMX_OK;
}

// One launch for all dense weights of an encoder (blockIdx.z = matrix): per-matrix launches of this 32 x 32-tile kernel take
// 7.6 us each, 96 of them per step for the two trained towers = 0.8 ms of a 216 ms step for 0.7 GB of traffic.
template <typename TO>
__global__ __launch_bounds__(256) void cast_weight_group_kernel(SimxCastGroup g) {
  int ji = 0;                                      // (uniform scan over the cumulative tile counts)
  while (ji + 1 < g.n && (int)blockIdx.x >= g.job[ji].tile_end) ++ji;
  const SimxCastJob j = g.job[ji];
  const int t = (int)blockIdx.x - (ji ? g.job[ji - 1].tile_end : 0), tc = (j.cols + 31) >> 5;
  __shared__ float tile[32][33];
  const int c0 = (t % tc) * 32, r0 = (t / tc) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  TO* o = reinterpret_cast<TO*>(j.out);
  TO* ot = reinterpret_cast<TO*>(j.outT);
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    float v = 0.f;
    if (r < j.rows && c < j.cols) { v = j.w[(long)r * j.cols + c]; if (o) Elem<TO>::st(o + (long)r * j.cols + c, v); }
    tile[i][tx] = v;
  }
  __syncthreads();
  if (ot)
    for (int i = ty; i < 32; i += 8) {
      const int c = c0 + i, r = r0 + tx;
      if (r < j.rows && c < j.cols) Elem<TO>::st(ot + (long)c * j.rows + r, tile[tx][i]);
    }
}
// the same job list on 64 x 64 tiles with 16-byte accesses on all three streams (16-bit outputs, rows % 64 == 0 and cols % 64 == 0:
// every dense weight of the BERT geometries; csrc/gemm_xp.hip split_weight_group64_kernel is the fp32 engine's twin): a thread owns
// 8 consecutive columns of a row in the first phase and 8 consecutive rows of a column in the second (LDS reads at stride 65
// words: the lanes of a wave cover 8 columns x 8 row groups = 64 distinct banks)
template <typename TO>
__global__ __launch_bounds__(256) void cast_weight_group64_kernel(SimxCastGroup g) {
  int ji = 0;
  while (ji + 1 < g.n && (int)blockIdx.x >= g.job[ji].tile_end) ++ji;
  const SimxCastJob j = g.job[ji];
  const int t = (int)blockIdx.x - (ji ? g.job[ji - 1].tile_end : 0), tc = j.cols >> 6;
  __shared__ float tile[64][65];
  const int c0 = (t % tc) * 64, r0 = (t / tc) * 64;
  bf16_t* o = reinterpret_cast<bf16_t*>(j.out);
  bf16_t* ot = reinterpret_cast<bf16_t*>(j.outT);
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int idx = (int)threadIdx.x + k * 256;            // 64 rows x 8 column groups
    const int r = idx >> 3, cg = (idx & 7) * 8;
    const long off = (long)(r0 + r) * j.cols + c0 + cg;
    const float4 a = *reinterpret_cast<const float4*>(j.w + off), b = *reinterpret_cast<const float4*>(j.w + off + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[r][cg + e] = v[e];
    if (o) *reinterpret_cast<uint4*>(o + off) = make_uint4(H16<TO>::pack2(v[0], v[1]), H16<TO>::pack2(v[2], v[3]), H16<TO>::pack2(v[4], v[5]), H16<TO>::pack2(v[6], v[7]));
  }
  __syncthreads();
  if (ot) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = (int)threadIdx.x + k * 256;          // 64 columns x 8 row groups
      const int c = idx >> 3, rg = (idx & 7) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[rg + e][c];
      *reinterpret_cast<uint4*>(ot + (long)(c0 + c) * j.rows + r0 + rg) =
          make_uint4(H16<TO>::pack2(v[0], v[1]), H16<TO>::pack2(v[2], v[3]), H16<TO>::pack2(v[4], v[5]), H16<TO>::pack2(v[6], v[7]));
    }
  }
}
int simx_transpose_cast_group(hipStream_t s, int out_dtype, const SimxCastGroup* g) {
  SIMX_REQUIRE(g && g->n > 0 && g->n <= SIMX_CAST_GROUP_MAX, SIMX_ERR_BAD_SHAPE, "transpose_cast_group: bad job count");
  SIMX_REQUIRE(simx_dtype_ok(out_dtype), SIMX_ERR_BAD_DTYPE, "transpose_cast_group: dtype %d", out_dtype);
  double bytes = 0;
  SimxCastGroup gg = *g;
  int tiles = 0;
  bool wide = simx_is16(out_dtype);               // every job on whole 64 x 64 tiles with 16-byte-aligned streams
  for (int i = 0; i < gg.n; ++i) {
    const SimxCastJob& jb = gg.job[i];
    SIMX_REQUIRE(jb.rows > 0 && jb.cols > 0 && jb.w, SIMX_ERR_BAD_SHAPE, "transpose_cast_group: bad job %d", i);
    wide = wide && jb.rows % 64 == 0 && jb.cols % 64 == 0 && aligned16(jb.w) && aligned16(jb.out) && aligned16(jb.outT);
  }
  const int ts = wide ? 64 : 32;
  for (int i = 0; i < gg.n; ++i) {
    tiles += cdiv(gg.job[i].rows, ts) * cdiv(gg.job[i].cols, ts);
    gg.job[i].tile_end = tiles;
    bytes += (double)gg.job[i].rows * gg.job[i].cols * 8;
  }
  SIMX_PROF(SIMX_K_CAST, s, bytes);
  if (wide) {
    SIMX_DISPATCH16(out_dtype, TT, hipLaunchKernelGGL((cast_weight_group64_kernel<TT>), dim3(tiles), dim3(256), 0, s, gg));
    SIMX_CHECK_LAUNCH("cast_weight_group64");
    return SIMX_OK;
  }
  SIMX_DISPATCH3(out_dtype, TT, hipLaunchKernelGGL((cast_weight_group_kernel<TT>), dim3(tiles), dim3(256), 0, s, gg));
  SIMX_CHECK_LAUNCH("cast_weight_group");
  return SIMX_OK;
}

extern "C" int simx_cast_weight(simx_stream_t stream, const float* w, int rows, int cols, void* w_bf16, void* wT_bf16) {
  return simx_transpose_cast(stream, SIMX_BF16, w, rows, cols, w_bf16, wT_bf16);
}

extern "C" int simx_gemm_f32_strided_ws(simx_stream_t stream, int M, int N, int K, const float* A, long a_rs, long a_cs,
                                        const float* B, long b_ks, long b_ns, float* C, int ldc, int accumulate, void* ws,
                                        size_t ws_bytes) {
  SIMX_REQUIRE(M > 0 && N > 0 && K > 0 && A && B && C, SIMX_ERR_BAD_SHAPE, "gemm_f32_strided: bad arguments");
  return launch_simple<float, float>((hipStream_t)stream, SIMX_EPI_NONE, M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc, nullptr,
                                     nullptr, 0, nullptr, 0, nullptr, 0, accumulate, DropCtx{0u, 1.f, 0u, 0u}, ws, ws_bytes);
}
extern "C" int simx_gemm_f32_strided(simx_stream_t stream, int M, int N, int K, const float* A, long a_rs, long a_cs,
                                     const float* B, long b_ks, long b_ns, float* C, int ldc, int accumulate) {
  return simx_gemm_f32_strided_ws(stream, M, N, K, A, a_rs, a_cs, B, b_ks, b_ns, C, ldc, accumulate, nullptr, 0);
}
