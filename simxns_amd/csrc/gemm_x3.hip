// f32 GEMMs on the 16-bit matrix cores: every f32 operand element is split on the fly into a 16-bit high part and a
// 16-bit correction, x = hi + lo, and the product is taken as hi.hi + hi.lo + lo.hi in f32 MFMA accumulators
// (three v_mfma_f32_16x16x32 per tile pair instead of one; lo.lo is below the f32 round-off of the sum).
//
//   format F = f16_t : hi, lo = IEEE half.  x = hi + lo carries 22 significand bits while |lo| stays normal and an ABSOLUTE
//                      error <= 2^-25 below that (gfx950's MFMA honours fp16 subnormals, tools/probe_f16.hip): weights
//                      (|w| ~ 0.03) keep ~2^-20 relative, O(1) activations 2^-22.  Used for the FORWARD GEMMs of the
//                      fp32 parity mode (x W^T, and the teacher's): 2.5 PFLOP/s / 3 = 833 TFLOP/s of f32-grade
//                      throughput at peak against the f32 MFMA's 157.
//   format F = bf16_t: hi, lo = bf16, 16 significand bits whatever the magnitude (f32's exponent range).  Used for the
//                      BACKWARD GEMMs (dgrad, wgrad), whose gradient operand spans 1e-2 ... 1e-9 and would need a loss
//                      scale in fp16.  2^-17 relative per operand: two orders below the 1e-3 / 2e-4 gradient tolerances.
//
// The exact kernel (gemm_f32_mfma_kernel, csrc/gemm.hip: v_mfma_f32_32x32x2_f32, exact f32 products) stays for small
// shapes, for the M2 score matrix and behind SIMX_GEMM_F32=exact.
//
// NT form  C[M,N] = A[M,K] . B[N,K]^T  (both K-contiguous; forward and dgrad), 128x128x32 tile, 4 waves of 64x64:
//   global f32 (whole-line 16-B loads, next stage in flight during the MFMAs) -> registers -> split -> LDS planes
//   Ahi|Alo|Bhi|Blo, each [128 rows][32 k] 16-bit = 64 B per row, so a wave's fragment read (16 rows x 64 B) is one contiguous KB.
//   What bounds it (kbench on the seven bench shapes, variants under tools/variants): without the MFMAs the kernel is 10 %
//   faster, without the global loads 50 % -- the loop is its load stream.  Measured steps: loads that cover whole 128-B rows
//   instead of two 16-B pieces of 32 rows per instruction 231 -> 272 TFLOP/s; one LDS stage and three workgroups per CU
//   (more loads in flight) -> 300; holding two stages in registers 227 -> 207 (slower); conflict-free fragment reads +0.8 %.
// TN form  C[M,N] (+)= A[K,M]^T . B[K,N]  (wgrad: rows = tokens, M / N contiguous), same tile, planes [32 k][128 cols] with
//   the 32-B chunk swizzle of gemm_tn_h16_kernel, fragments by ds_read_b64_tr_b16 (the k-slot permutation it implies is
//   the same for both operands); split over K into f32 slabs reduced in slice order (slab_reduce_kernel): deterministic.
#include "common.h"
#include "prof.h"

template <typename F>
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  a = f32_pin(a); b = f32_pin(b);
  hi = H16<F>::pack2(a, b);
  lo = H16<F>::pack2(a - H16<F>::lo(hi), b - H16<F>::hi(hi));
}

#define X3_BM 128
#define X3_BN 128
#define X3_BK 32
#define X3_PLANE (128 * 64)                 // one 128 x 32 16-bit plane
#define X3_STAGE (4 * X3_PLANE)             // Ahi | Alo | Bhi | Blo

// slabs added in slice order (deterministic), as slab_reduce_kernel of csrc/gemm.hip (kernels do not link across translation units)
__global__ __launch_bounds__(256) void x3_slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int splits, int M, int N,
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

template <typename F, int EPI>
__global__ __launch_bounds__(256, 3) void gemm_x3_nt_kernel(
    int M, int N, int K, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ C, int ldc,
    const float* __restrict__ bias, const float* __restrict__ res, int ldr, const float* __restrict__ aux, int ldaux,
    float* __restrict__ C2, int ldc2, int tiles_n, int ntiles, DropCtx drop) {
  // ONE LDS stage per workgroup (two barriers per k step) and three workgroups per CU: the loop is bound by its global
  // loads (without them 402 instead of 265 TFLOP/s), so 12 waves' worth of loads in flight beat 8 waves with a second LDS
  // buffer: 265 -> 300 TFLOP/s on the bench shapes.  (Four per CU would need <= 128 VGPRs: spills.)
  __shared__ __attribute__((aligned(16))) char smem[X3_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // XCD-aware order: block b runs on XCD b % 8; every XCD walks a contiguous range of tiles so that the N-tiles sharing an
  // A panel (and all tiles sharing B) sit behind one L2
  const int q8 = ntiles >> 3, r8 = ntiles & 7, xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
  const int m0 = (tile / tiles_n) * X3_BM, n0 = (tile % tiles_n) * X3_BN;
  const int wr = wave >> 1, wc = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging: a wave's 16-B loads cover WHOLE rows -- lane l reads floats 4 (l & 7) .. + 3 of row (l >> 3) of an 8-row slab,
  // 8 full 128-B lines per instruction; thread t owns k chunk t & 7 of rows (t >> 3) + 32 e, e = 0..3, of both tiles.
  // (The first form gave every thread 16 consecutive floats of one row: each instruction then touched 32 lines and used 32 B
  // of each, and the kernel was bound by exactly that -- with the loads removed it ran 1.75x faster, with a third workgroup
  // per CU only 2 % faster.)
  const int srow = tid >> 3, sk = (tid & 7) * 4;
  const float *pa[4], *pb[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int ga = m0 + srow + 32 * e, gb = n0 + srow + 32 * e;
    ga = ga < M ? ga : M - 1;                     // (clamped rows are computed and never stored)
    gb = gb < N ? gb : N - 1;
    pa[e] = A + (long)ga * lda + sk;
    pb[e] = B + (long)gb * ldb + sk;
  }
  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
    const bool in = k0 + sk + 3 < K;               // ragged K tail (K % 4 == 0 is required: whole float4s)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ra[e] = in ? *reinterpret_cast<const float4*>(pa[e] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[e] = in ? *reinterpret_cast<const float4*>(pb[e] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&](char* stage) {
    // 16-bit plane row = 32 k = 64 B = four 16-B chunks; this thread's 4 k are 8 B: half (tid & 1) of chunk (tid & 7) >> 1,
    // chunks of rows 8-15 of every 16 stored XOR 2 (see the fragment reads)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int row = srow + 32 * e;
      char* d = stage + row * 64 + (((((tid & 7) >> 1) ^ (((row >> 3) & 1) << 1))) << 4) + (tid & 1) * 8;
      uint32_t h0, h1, l0, l1;
      split2<F>(ra[e].x, ra[e].y, h0, l0); split2<F>(ra[e].z, ra[e].w, h1, l1);
      *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + X3_PLANE) = make_uint2(l0, l1);
      split2<F>(rb[e].x, rb[e].y, h0, l0); split2<F>(rb[e].z, rb[e].w, h1, l1);
      *reinterpret_cast<uint2*>(d + 2 * X3_PLANE) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + 3 * X3_PLANE) = make_uint2(l0, l1);
    }
  };

  const int nst = (K + X3_BK - 1) / X3_BK;
  gload(0);
  sstore(smem);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const char* cur = smem;
    if (st + 1 < nst) gload((st + 1) * X3_BK);
    // fragments: row (tile*16 + fr), k chunk fg (8 consecutive k = 16 B) of each plane
    // 64-B rows: ds_read_b128 serves lanes in four 16-lane groups ({0-3, 12-15, 20-27}, ...), i.e. rows fr, fr + 12 at
    // chunk c and rows fr + 4, fr + 8 at chunk c + 1 land on the four 16-B slots of ONE 64-B row image (row & 3): stored
    // linearly, two of them share a slot (2-way conflict on every fragment read).  Rows 8-15 of every 16 keep their chunks
    // XOR 2: the four become distinct slots.
    const int fsw = (fg ^ (((fr >> 3) & 1) << 1)) * 16;
    const char* fa = cur + (wr * 64 + fr) * 64 + fsw;
    const char* fb = cur + 2 * X3_PLANE + (wc * 64 + fr) * 64 + fsw;
    bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = *reinterpret_cast<const bf16x8*>(fa + i * 1024);
      al[i] = *reinterpret_cast<const bf16x8*>(fa + i * 1024 + X3_PLANE);
      bh[i] = *reinterpret_cast<const bf16x8*>(fb + i * 1024);
      bl[i] = *reinterpret_cast<const bf16x8*>(fb + i * 1024 + X3_PLANE);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        // (operands swapped so that a lane ends up with 4 consecutive output columns of one row; corrections first)
        acc[i][j] = H16<F>::mfma(bl[j], ah[i], acc[i][j]);
        acc[i][j] = H16<F>::mfma(bh[j], al[i], acc[i][j]);
        acc[i][j] = H16<F>::mfma(bh[j], ah[i], acc[i][j]);
      }
    __syncthreads();                               // every wave has its fragments of this stage
    if (st + 1 < nst) sstore(smem);
    __syncthreads();
  }

  // epilogue through LDS.  In the MFMA layout a lane holds row fr, 4 columns of 16-column block j: a 16-B store per lane then
  // covers 16 rows x 64 B = sixteen half lines per instruction, and so do the residual / aux loads.  Each
  // wave therefore passes its 64 x 64 tile through its own 4.25 KB of the (now free) stage buffer, 16 rows at a time, and
  // leaves with rows as the fast index: lane l owns columns 4 (l & 15) .. + 3 of row (l >> 4) + 4 it, i.e. every 16-B load
  // and store instruction covers 4 rows x 256 B = eight full lines.  (Row pitch 68 floats: the writes of 8 rows at one
  // column offset land on 8 different bank groups.)  Only this wave touches its region, and a wave's LDS operations are
  // served in issue order: no barrier.  (+1 % only: with every accumulator kept live but nothing stored the kernel is 16 %
  // faster, and that difference is the HBM time of the f32 outputs and epilogue inputs -- 6.4 GB for the GELU pair;
  // non-temporal stores / loads +0.6 %.)
  float* ep = reinterpret_cast<float*>(smem) + wave * (16 * 68);
  const int er = lane >> 4, ec = (lane & 15) * 4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j)
      *reinterpret_cast<float4*>(ep + fr * 68 + j * 16 + fg * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int r = it * 4 + er;
      const int m = m0 + wr * 64 + i * 16 + r, n = n0 + wc * 64 + ec;
      const float4 t = *reinterpret_cast<const float4*>(ep + r * 68 + ec);
      if (m >= M || n >= N) continue;              // (N % 4 == 0 is required: a lane's 4 columns are all in or all out)
      float v[4] = {t.x, t.y, t.z, t.w};
      if (bias) {
        const float4 bv = *reinterpret_cast<const float4*>(bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if (EPI == SIMX_EPI_NONE) {
        if (drop.thr) { float m4[4]; drop_mult4(drop, (uint32_t)m, (uint32_t)n, m4); v[0] *= m4[0]; v[1] *= m4[1]; v[2] *= m4[2]; v[3] *= m4[3]; }
        if (res) { float r4[4]; ld4(res + (long)m * ldr + n, r4); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
        st4(C + (long)m * ldc + n, v);
      } else if (EPI == SIMX_EPI_GELU) {
        float g4[4], d4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { g4[e] = gelu_erf(v[e]); d4[e] = gelu_erf_grad(v[e]); }
        st4(C + (long)m * ldc + n, d4);             // C = gelu'(u): what backward multiplies by
        st4(C2 + (long)m * ldc2 + n, g4);
      } else {
        if (res) { float r4[4]; ld4(res + (long)m * ldr + n, r4); v[0] += r4[0]; v[1] += r4[1]; v[2] += r4[2]; v[3] += r4[3]; }
        float u4[4];
        ld4(aux + (long)m * ldaux + n, u4);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= u4[e];
        st4(C + (long)m * ldc + n, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ TN (wgrad)
// LDS plane of a 32(k) x 128(col) 16-bit tile: row kr at kr * 256 B, 32-B chunk q (16 columns) stored at q ^ (kr & 7).
// (two workgroups per CU with a double-buffered stage here: three single-buffered ones measured 275 against 298 TFLOP/s -- the
// token-major operand rows of this form were whole-line loads from the start)
template <typename F>
__global__ __launch_bounds__(256, 2) void gemm_x3_tn_kernel(
    int M, int N, int K, const float* __restrict__ A, int lda, const float* __restrict__ B, int ldb, float* __restrict__ out,
    long slab_stride, int ldo, int tiles_n, int tiles_mn, int k_per_split, int accumulate, float* __restrict__ dbias, int dbias_parts) {
  // dbias: the bias gradient = column sums of A over the tokens, taken from the staging registers of the workgroups of the
  // first N-tile (a separate column-sum pass re-read A: 17.7 ms per fp32 step).  dbias_parts (deterministic mode): dbias
  // is a [splits][M] partial buffer instead of the accumulation target.
  __shared__ __attribute__((aligned(16))) char smem[2 * X3_STAGE];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int split = blockIdx.x / tiles_mn, tile = blockIdx.x % tiles_mn;
  const int m0 = (tile / tiles_n) * X3_BM, n0 = (tile % tiles_n) * X3_BN;
  const int wr = wave >> 1, wc = wave & 1;
  const int kb = split * k_per_split, ke = min(K, kb + k_per_split);
  const int fs = lane & 15, fg = lane >> 4;

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // staging: thread t owns k rows (t >> 5) + 8 e, e < 4, columns (t & 31) * 4 .. +3 of both operand tiles
  const int skr = tid >> 5, sc4 = (tid & 31) * 4;
  float4 ra[4], rb[4];
  auto gload = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = k0 + skr + 8 * e;
      const bool kin = k < ke;
      ra[e] = (kin && m0 + sc4 + 3 < M) ? *reinterpret_cast<const float4*>(A + (long)k * lda + m0 + sc4) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[e] = (kin && n0 + sc4 + 3 < N) ? *reinterpret_cast<const float4*>(B + (long)k * ldb + n0 + sc4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  const bool do_bias = dbias != nullptr && n0 == 0;
  float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
  auto sstore = [&](char* stage) {
    if (do_bias) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { bs.x += ra[e].x; bs.y += ra[e].y; bs.z += ra[e].z; bs.w += ra[e].w; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int kr = skr + 8 * e;
      char* d = stage + kr * 256 + (((sc4 >> 4) ^ (kr & 7)) << 5) + (sc4 & 15) * 2;
      uint32_t h0, l0, h1, l1;
      split2<F>(ra[e].x, ra[e].y, h0, l0); split2<F>(ra[e].z, ra[e].w, h1, l1);
      *reinterpret_cast<uint2*>(d) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + X3_PLANE) = make_uint2(l0, l1);
      split2<F>(rb[e].x, rb[e].y, h0, l0); split2<F>(rb[e].z, rb[e].w, h1, l1);
      *reinterpret_cast<uint2*>(d + 2 * X3_PLANE) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(d + 3 * X3_PLANE) = make_uint2(l0, l1);
    }
  };
  typedef __attribute__((address_space(3))) bf16x4* lds4_t;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  // transpose-read addressing (as gemm_tn_h16_kernel): lane (fg, fs) supplies k row 4 fg + (fs >> 2) (+16 for the high half),
  // 8 B at columns ct*16 + (fs & 3)*4 of 16-column tile ct; after the read it holds 4 k values of column fs of that tile
  const int r_lo = 4 * fg + (fs >> 2);
  auto frag = [&](uint32_t plane, int ct) -> bf16x8 {
    const int r0 = r_lo, r1 = r_lo + 16;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(uintptr_t)(plane + r0 * 256 + ((ct ^ (r0 & 7)) << 5) + (fs & 3) * 8));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds4_t)(uintptr_t)(plane + r1 * 256 + ((ct ^ (r1 & 7)) << 5) + (fs & 3) * 8));
    return (bf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };

  const int nst = (ke - kb + X3_BK - 1) / X3_BK;
  gload(kb);
  sstore(smem);
  __syncthreads();
  for (int st = 0; st < nst; ++st) {
    const uint32_t cur = lds0 + (uint32_t)((st & 1) * X3_STAGE);
    if (st + 1 < nst) gload(kb + (st + 1) * X3_BK);
    bf16x8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ah[i] = frag(cur, wr * 4 + i);
      al[i] = frag(cur + X3_PLANE, wr * 4 + i);
      bh[i] = frag(cur + 2 * X3_PLANE, wc * 4 + i);
      bl[i] = frag(cur + 3 * X3_PLANE, wc * 4 + i);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[i][j] = H16<F>::mfma(bl[j], ah[i], acc[i][j]);
        acc[i][j] = H16<F>::mfma(bh[j], al[i], acc[i][j]);
        acc[i][j] = H16<F>::mfma(bh[j], ah[i], acc[i][j]);
      }
    if (st + 1 < nst) sstore(smem + ((st + 1) & 1) * X3_STAGE);
    __syncthreads();
  }
  if (do_bias) {                                   // (uniform per workgroup) 8 token-row lanes per column group -> one sum per column
    float* red = reinterpret_cast<float*>(smem);   // the loop's last barrier has passed: the stages are free
    *reinterpret_cast<float4*>(red + skr * 128 + sc4) = bs;
    __syncthreads();
    if (tid < 128 && m0 + tid < M) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 8; ++r) t += red[r * 128 + tid];
      if (dbias_parts) dbias[(long)split * M + m0 + tid] = t;
      else atomicAdd(dbias + m0 + tid, t);
    }
  }
  float* o = out + (long)split * slab_stride;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wr * 64 + i * 16 + fs;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wc * 64 + j * 16 + fg * 4;
      if (n >= N) continue;
      float4* dst = reinterpret_cast<float4*>(o + (long)m * ldo + n);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (accumulate) { const float4 c = *dst; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
      *dst = v;
    }
  }
}

// ------------------------------------------------------------------------------------------ host
static bool al16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// can the split kernels take this NT problem?  (whole float4s everywhere; anything else stays on the exact kernel)
bool simx_x3_nt_ok(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* C, int ldc, const float* bias,
                   const float* res, int ldr, const float* aux, int ldaux, const float* C2, int ldc2) {
  return K % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && al16(A) && al16(B) && al16(C) && (!bias || al16(bias)) &&
         (!res || (ldr % 4 == 0 && al16(res))) && (!aux || (ldaux % 4 == 0 && al16(aux))) && (!C2 || (ldc2 % 4 == 0 && al16(C2))) &&
         (long)M * N >= 64 * 1024 && K >= 64;
}

// fmt: SIMX_F16 (forward GEMMs) or SIMX_BF16 (backward GEMMs) -- the format of the 16-bit halves, see the file header
int simx_x3_gemm_nt(hipStream_t s, int fmt, int epi, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    const float* bias, const float* res, int ldr, const float* aux, int ldaux, float* C2, int ldc2, DropCtx drop) {
  const int tiles_m = cdiv(M, X3_BM), tiles_n = cdiv(N, X3_BN), nt = tiles_m * tiles_n;
#define LX(FF, E) hipLaunchKernelGGL((gemm_x3_nt_kernel<FF, E>), dim3(nt), dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, ldc, bias, res, ldr, aux, \
                                     ldaux, C2, ldc2, tiles_n, nt, drop)
#define LX_ALL(FF) do { if (epi == SIMX_EPI_NONE) LX(FF, SIMX_EPI_NONE); else if (epi == SIMX_EPI_GELU) LX(FF, SIMX_EPI_GELU); else LX(FF, SIMX_EPI_DGELU); } while (0)
  SIMX_DISPATCH16(fmt, FF, LX_ALL(FF));
#undef LX_ALL
#undef LX
  SIMX_CHECK_LAUNCH("gemm_x3_nt");
  return SIMX_OK;
}

bool simx_x3_tn_ok(int M, int N, int K, const float* A, int lda, const float* B, int ldb, const float* C, int ldc) {
  return M % 4 == 0 && N % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && al16(A) && al16(B) && al16(C) && (long)M * N >= 16 * 1024 &&
         K >= 256;
}
static void x3_tn_plan(int M, int N, int K, int* splits, int* kps) {
  const int tiles = cdiv(M, X3_BM) * cdiv(N, X3_BN);
  int sp = 1024 / tiles;                            // two workgroups per CU: ~4 per CU keeps the tail short
  const int max_s = cdiv(K, 512);
  if (sp > max_s) sp = max_s;
  if (sp < 1) sp = 1;
  int k = cdiv(cdiv(K, sp), X3_BK) * X3_BK;
  *splits = cdiv(K, k);
  *kps = k;
}
size_t simx_x3_tn_workspace_bytes(int M, int N, int K) {
  int sp, kps;
  x3_tn_plan(M, N, K, &sp, &kps);
  return sp > 1 ? (size_t)sp * M * N * sizeof(float) : 0;
}
int simx_x3_gemm_tn(hipStream_t s, int fmt, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                    int accumulate, void* ws, size_t ws_bytes, float* dbias) {
  int sp, kps;
  x3_tn_plan(M, N, K, &sp, &kps);
  const int t_n = cdiv(N, X3_BN), t_mn = cdiv(M, X3_BM) * t_n;
  if (sp > 1 && (!ws || ws_bytes < (size_t)sp * M * N * sizeof(float) || !al16(ws))) { sp = 1; kps = cdiv(K, X3_BK) * X3_BK; }
  float* dbias_out = dbias;
  if (dbias && simx_det()) {                      // ordered bias gradient: one partial row per split
    dbias = simx_det_ws(s, (size_t)sp * M * sizeof(float));
    if (!dbias) return SIMX_ERR_WORKSPACE;
  }
  const int dparts = dbias != dbias_out;
  if (sp == 1) {
    SIMX_DISPATCH16(fmt, FF, hipLaunchKernelGGL(gemm_x3_tn_kernel<FF>, dim3(t_mn), dim3(256), 0, s, M, N, K, A, lda, B, ldb, C, 0L, ldc, t_n, t_mn,
                                                kps, accumulate, dbias, dparts));
    SIMX_CHECK_LAUNCH("gemm_x3_tn");
    if (dparts) return simx_det_reduce(s, dbias, (long)M, sp, M, dbias_out, nullptr, nullptr, nullptr);
    return SIMX_OK;
  }
  SIMX_DISPATCH16(fmt, FF, hipLaunchKernelGGL(gemm_x3_tn_kernel<FF>, dim3(t_mn * sp), dim3(256), 0, s, M, N, K, A, lda, B, ldb, (float*)ws,
                                              (long)M * N, N, t_n, t_mn, kps, 0, dbias, dparts));
  SIMX_CHECK_LAUNCH("gemm_x3_tn");
  if (dparts) { int rcd = simx_det_reduce(s, dbias, (long)M, sp, M, dbias_out, nullptr, nullptr, nullptr); if (rcd) return rcd; }
  const long tot4 = (long)M * N / 4;
  int rb = (int)((tot4 + 255) / 256);
  if (rb > 2048) rb = 2048;
  hipLaunchKernelGGL(x3_slab_reduce_kernel, dim3(rb), dim3(256), 0, s, (const float*)ws, (long)M * N, sp, M, N, C, ldc, accumulate);
  SIMX_CHECK_LAUNCH("slab_reduce");
  return SIMX_OK;
}
