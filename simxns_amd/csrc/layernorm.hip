// Embedding gather + LayerNorm, residual LayerNorm, [CLS] gather/scatter -- the HBM-bound row
// kernels of the encoder (BertEmbeddings / BertSelfOutput / BertOutput, LEAD/modeling_bert.py:181-240,
// 377-388, 455-466).  One 64-lane wavefront owns one token row (H <= 1024: the row lives in registers,
// 4 elements per lane per step, 8/16-byte accesses), statistics in f32 with a centred second pass,
// wave-level butterflies only (no LDS, no barriers) in the forward kernels.  Backward kernels keep
// per-lane column partial sums (dgamma, dbeta, dbias / dtype0) across the rows a workgroup walks and
// flush them once with f32 atomics.
#include <type_traits>
#include "common.h"
#include "prof.h"

#define LN_VPL 4   // 4-element vectors per lane -> H <= 64*4*4 = 1024

// raw (still packed) 4-element vectors: lets the next row's loads stay in flight while this row is processed
template <typename T> struct Raw4;
template <> struct Raw4<float> {
  float4 r;
  __device__ __forceinline__ void load(const float* p) { r = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void unpack(float (&v)[4]) const { v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w; }
  __device__ __forceinline__ uint32_t w0() const { return 0u; }      // (raw 16-bit words: meaningless for f32)
  __device__ __forceinline__ uint32_t w1() const { return 0u; }
};
template <> struct Raw4<bf16_t> {
  uint2 r;
  __device__ __forceinline__ void load(const bf16_t* p) { r = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void unpack(float (&v)[4]) const {
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xFFFF0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xFFFF0000u);
  }
  __device__ __forceinline__ uint32_t w0() const { return r.x; }
  __device__ __forceinline__ uint32_t w1() const { return r.y; }
};

template <> struct Raw4<f16_t> {
  uint2 r;
  __device__ __forceinline__ void load(const f16_t* p) { r = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void unpack(float (&v)[4]) const {
    v[0] = H16<f16_t>::lo(r.x); v[1] = H16<f16_t>::hi(r.x); v[2] = H16<f16_t>::lo(r.y); v[3] = H16<f16_t>::hi(r.y);
  }
  __device__ __forceinline__ uint32_t w0() const { return r.x; }
  __device__ __forceinline__ uint32_t w1() const { return r.y; }
};
// store 4 values in T and return what the rounding dropped (r) plus the packed words (for lo8_encode4)
template <typename T>
__device__ __forceinline__ void st4_residue(T* p, const float (&o)[4], float (&r)[4], uint32_t& w0, uint32_t& w1) {
  if constexpr (sizeof(T) == 2) {
    w0 = H16<T>::pack2(o[0], o[1]);
    w1 = H16<T>::pack2(o[2], o[3]);
    *reinterpret_cast<uint2*>(p) = make_uint2(w0, w1);
    r[0] = o[0] - H16<T>::lo(w0); r[1] = o[1] - H16<T>::hi(w0); r[2] = o[2] - H16<T>::lo(w1); r[3] = o[3] - H16<T>::hi(w1);
  } else {
    st4(p, o);
    w0 = w1 = 0u;
    r[0] = r[1] = r[2] = r[3] = 0.f;
  }
}

// plane pair (csrc/gemm_xp.hip, simx.h "operand planes") of 4 consecutive f32 values: hi = rnd16(o), lo = rnd16(o - hi)
template <typename F>
__device__ __forceinline__ void st4_planes(bf16_t* p, long plane_stride, const float (&oo)[4]) {
  const float o[4] = {f32_pin(oo[0]), f32_pin(oo[1]), f32_pin(oo[2]), f32_pin(oo[3])};
  const uint32_t h0 = H16<F>::pack2(o[0], o[1]), h1 = H16<F>::pack2(o[2], o[3]);
  const uint32_t l0 = H16<F>::pack2(o[0] - H16<F>::lo(h0), o[1] - H16<F>::hi(h0));
  const uint32_t l1 = H16<F>::pack2(o[2] - H16<F>::lo(h1), o[3] - H16<F>::hi(h1));
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(p + plane_stride) = make_uint2(l0, l1);
}

// One wave per row, rows strided by 4 inside a block; the next row's loads are issued before this row's
// reductions, and gamma/beta are read from LDS so the row body never queues behind those loads on vmcnt.
// RES ("f32-grade residual stream" of the 16-bit engines, simx.h stream_lo): the input row is  z = d + r_hi + r_lo  -- the
// dense output (bias and dropout applied by the GEMM epilogue, no residual) plus the residual stream kept as a 16-bit value
// and ONE correction byte (lo8, common.h: x = hi + (b - 128) ulp(hi) / 256) -- summed in f32, and the result leaves as
// y_hi = round16(y) and the byte that encodes y - y_hi: the stream carries ~19 significand bits from layer to layer (apex O1
// keeps it in fp32: residual additions promote to fp32 and LayerNorm is an fp32 function there) at 3 B per element, and z
// never makes a round trip through HBM.
template <typename T, int VPL, bool RES>
__global__ __launch_bounds__(256) void ln_fwd_kernel(int rows, int H, int rows_per_block, const T* __restrict__ z,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float eps, T* __restrict__ y, const T* __restrict__ rh,
                                                     const uint8_t* __restrict__ rl, uint8_t* __restrict__ ylo,
                                                     bf16_t* __restrict__ yp = nullptr, long yps = 0) {
  // yp (f32 engine, "operand planes"): y also leaves as the fp16 plane pair the next GEMM stages
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sgam = reinterpret_cast<float*>(smem);
  float* sbet = sgam + H;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < H; c += 256) { sgam[c] = gamma[c]; sbet[c] = beta[c]; }
  __syncthreads();
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  Raw4<T> nx[VPL], nh[RES ? VPL : 1];
  uint32_t nl[RES ? VPL : 1];                     // the residual's correction bytes (lo8, common.h), 4 elements per dword
  auto issue = [&](int row) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        nx[v].load(z + (long)row * H + c);
        if (RES) { nh[v].load(rh + (long)row * H + c); if (rl) nl[v] = *reinterpret_cast<const uint32_t*>(rl + (long)row * H + c); }
      }
    }
  };
  if (r0 + w < r1) issue(r0 + w);
  for (int row = r0 + w; row < r1; row += 4) {
    float x[VPL][4];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        nx[v].unpack(x[v]);
        if (RES) {
          float a[4];
          nh[v].unpack(a);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[v][e] += a[e];
          if (rl) {
            lo8_decode4<T>(nh[v].w0(), nh[v].w1(), nl[v], a);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[v][e] += a[e];
          }
        }
        sum += x[v][0] + x[v][1] + x[v][2] + x[v][3];
      }
    }
    if (row + 4 < r1) issue(row + 4);
    const float mu = wave_sum(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H)
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[v][e] -= mu; sq += x[v][e] * x[v][e]; }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
    T* yr = y + (long)row * H;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        float g[4], b[4], o[4];
        ld4(sgam + c, g);
        ld4(sbet + c, b);
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = x[v][e] * rstd * g[e] + b[e];
        if (RES && ylo) {                       // the correction: what the 16-bit rounding of y drops, one byte per element
          float rr[4];
          uint32_t w0, w1;
          st4_residue(yr + c, o, rr, w0, w1);
          *reinterpret_cast<uint32_t*>(ylo + (long)row * H + c) = lo8_encode4<T>(w0, w1, rr);
        } else {
          st4(yr + c, o);
          if constexpr (std::is_same<T, float>::value) { if (yp) st4_planes<f16_t>(yp + (long)row * H + c, yps, o); }
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(int rows, int H, const int* __restrict__ ids,
                                                           const int* __restrict__ pos, const float* __restrict__ word,
                                                           const float* __restrict__ posw, const float* __restrict__ typew,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, T* __restrict__ y, DropCtx drop, uint8_t* __restrict__ ylo,
                                                           bf16_t* __restrict__ yp = nullptr, long yps = 0) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int row = blockIdx.x * 4 + w;
  if (row >= rows) return;
  const float* wr = word + (long)ids[row] * H;
  const float* pr = posw + (long)pos[row] * H;
  float x[LN_VPL][4];
  float sum = 0.f;
#pragma unroll
  for (int v = 0; v < LN_VPL; ++v) {
    const int c = (v * 64 + lane) * 4;
    if (c < H) {
      float a[4], b[4], t[4];
      ld4(wr + c, a);
      ld4(pr + c, b);
      ld4(typew + c, t);
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[v][e] = a[e] + b[e] + t[e]; sum += x[v][e]; }
    }
  }
  const float mu = wave_sum(sum) / (float)H;
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < LN_VPL; ++v) {
    const int c = (v * 64 + lane) * 4;
    if (c < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) { x[v][e] -= mu; sq += x[v][e] * x[v][e]; }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
  T* yr = y + (long)row * H;
#pragma unroll
  for (int v = 0; v < LN_VPL; ++v) {
    const int c = (v * 64 + lane) * 4;
    if (c < H) {
      float g[4], b[4], o[4];
      ld4(gamma + c, g);
      ld4(beta + c, b);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = x[v][e] * rstd * g[e] + b[e];
      if (drop.thr) { float m4[4]; drop_mult4(drop, (uint32_t)row, (uint32_t)c, m4); o[0] *= m4[0]; o[1] *= m4[1]; o[2] *= m4[2]; o[3] *= m4[3]; }
      if (ylo) {                                 // residual-stream correction (simx.h stream_lo)
        float rr[4];
        uint32_t w0, w1;
        st4_residue(yr + c, o, rr, w0, w1);
        *reinterpret_cast<uint32_t*>(ylo + (long)row * H + c) = lo8_encode4<T>(w0, w1, rr);
      } else {
        st4(yr + c, o);
        if constexpr (std::is_same<T, float>::value) { if (yp) st4_planes<f16_t>(yp + (long)row * H + c, yps, o); }
      }
    }
  }
}

// shared row-backward: given centred x (in/out: becomes xhat), dy -> dz ; accumulates column partials
template <int VPL>
__device__ __forceinline__ void ln_row_bwd(int H, int lane, float (&x)[VPL][4], float (&dy)[VPL][4],
                                           const float* __restrict__ gamma, float eps, float (&dz)[VPL][4],
                                           float (&pg)[VPL][4], float (&pb)[VPL][4]) {
  float sq = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v)
    if ((v * 64 + lane) * 4 < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) sq += x[v][e] * x[v][e];
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)H + eps);
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = (v * 64 + lane) * 4;
    if (c < H) {
      float g[4];
      ld4(gamma + c, g);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        x[v][e] *= rstd;                       // xhat
        pg[v][e] += dy[v][e] * x[v][e];
        pb[v][e] += dy[v][e];
        dy[v][e] *= g[e];                      // dxhat
        s1 += dy[v][e];
        s2 += dy[v][e] * x[v][e];
      }
    }
  }
  s1 = wave_sum(s1) / (float)H;
  s2 = wave_sum(s2) / (float)H;
#pragma unroll
  for (int v = 0; v < VPL; ++v)
    if ((v * 64 + lane) * 4 < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) dz[v][e] = rstd * (dy[v][e] - s1 - x[v][e] * s2);
}

template <int VPL>
__device__ __forceinline__ void flush_cols(int H, int lane, int w, float (&p)[VPL][4], float* __restrict__ out,
                                           float* sred /* [4][H] */, float mul = 1.0f, float* det = nullptr) {
  // sum the 4 waves' partials through LDS, wave 0 issues the atomics (det: this workgroup's partial row of the
  // deterministic mode -- stored unscaled, det_reduce_kernel adds the rows in workgroup order)
#pragma unroll
  for (int v = 0; v < VPL; ++v) {
    const int c = (v * 64 + lane) * 4;
    if (c < H)
#pragma unroll
      for (int e = 0; e < 4; ++e) sred[w * H + c + e] = p[v][e];
  }
  __syncthreads();
  // all four waves issue the atomics, each instruction on 64 CONSECUTIVE columns (4 cache lines; the lane-owns-4-columns
  // layout would touch 16) -- every block of the grid adds into the same H addresses, so the L2 transaction count of
  // this tail is what the kernel's last microseconds are made of
  if (det)
    for (int c = threadIdx.x; c < H; c += 256) det[c] = (sred[c] + sred[H + c]) + (sred[2 * H + c] + sred[3 * H + c]);
  else
    for (int c = threadIdx.x; c < H; c += 256) atomicAdd(out + c, (sred[c] + sred[H + c] + sred[2 * H + c] + sred[3 * H + c]) * mul);
  __syncthreads();
}

// One wave per row, rows strided by 4 inside a block.  The loads of row+4 are issued before row is processed
// (gamma comes from LDS so nothing in the row body queues behind them on vmcnt), which doubles the bytes each
// wave keeps in flight; the three column partials stay in registers and are flushed once per block.
// RES: the LayerNorm input is rebuilt as z = d + r_hi + r_lo (see ln_fwd_kernel), rl may be NULL.
template <typename T, int VPL, bool RES>
__global__ __launch_bounds__(256) void ln_bwd_kernel(int rows, int H, int rows_per_block, const T* __restrict__ z,
                                                     const float* __restrict__ gamma, float eps, const T* __restrict__ dyp,
                                                     T* __restrict__ dzp, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, float* __restrict__ dbias,
                                                     T* __restrict__ dzm, DropCtx drop, const int* __restrict__ row_keys,
                                                     const float* __restrict__ gs, const T* __restrict__ rh,
                                                     const uint8_t* __restrict__ rl, float* __restrict__ det_part,
                                                     bf16_t* __restrict__ dzpl = nullptr, long dzps = 0) {
  // dzpl (f32 engine, "operand planes"): the gradient of the (dropped) dense output leaves as the bf16 plane pair the dgrad /
  // wgrad GEMMs stage -- with or without dropout -- and the f32 masked copy dzm is not written
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sred = reinterpret_cast<float*>(smem);          // [4][H] flush scratch
  float* sgam = sred + 4 * H;                            // [H]
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < H; c += 256) sgam[c] = gamma[c];
  __syncthreads();
  float pg[VPL][4] = {}, pb[VPL][4] = {}, pz[VPL][4] = {};
  // Rows are dealt out grid-strided (block b, wave w: rows 4b + w, + 4*gridDim, ...): at any moment the whole grid works
  // on one contiguous window of ~4*gridDim rows.  Giving every block its own contiguous range instead made 2048 waves
  // walk 2048 regions 0.8 MB apart in four tensors at once, and the DRAM pages thrashed (4.1 TB/s, and MORE blocks per
  // CU made it slower).
  (void)rows_per_block;
  const int r0 = blockIdx.x * 4, r1 = rows, rstep = (int)gridDim.x * 4;
  Raw4<T> nx[VPL], nd[VPL], nh[RES ? VPL : 1];
  uint32_t nl[RES ? VPL : 1];
  auto issue = [&](int row) {
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        nx[v].load(z + (long)row * H + c);
        nd[v].load(dyp + (long)row * H + c);
        if (RES) { nh[v].load(rh + (long)row * H + c); if (rl) nl[v] = *reinterpret_cast<const uint32_t*>(rl + (long)row * H + c); }
      }
    }
  };
  if (r0 + w < r1) issue(r0 + w);
  for (int row = r0 + w; row < r1; row += rstep) {
    float x[VPL][4], dy[VPL][4], dz[VPL][4];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        nx[v].unpack(x[v]);
        nd[v].unpack(dy[v]);
        if (RES) {
          float a[4];
          nh[v].unpack(a);
#pragma unroll
          for (int e = 0; e < 4; ++e) x[v][e] += a[e];
          if (rl) {
            lo8_decode4<T>(nh[v].w0(), nh[v].w1(), nl[v], a);
#pragma unroll
            for (int e = 0; e < 4; ++e) x[v][e] += a[e];
          }
        }
        sum += x[v][0] + x[v][1] + x[v][2] + x[v][3];
      }
    }
    if (row + rstep < r1) issue(row + rstep);
    const float mu = wave_sum(sum) / (float)H;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) x[v][e] -= mu;
    ln_row_bwd(H, lane, x, dy, sgam, eps, dz, pg, pb);
    T* o = dzp + (long)row * H;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        st4(o + c, dz[v]);
        if (drop.thr) {                      // gradient of the dropped dense output (feeds dgrad / wgrad / bias grad)
          float m4[4];
          drop_mult4(drop, (uint32_t)(row_keys ? row_keys[row] : row), (uint32_t)c, m4);   // keys: rows gathered from a larger tensor
#pragma unroll
          for (int e = 0; e < 4; ++e) dz[v][e] *= m4[e];
          if (!dzpl) st4(dzm + (long)row * H + c, dz[v]);
        }
        if constexpr (std::is_same<T, float>::value) { if (dzpl) st4_planes<bf16_t>(dzpl + (long)row * H + c, dzps, dz[v]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) pz[v][e] += dz[v][e];
      }
    }
  }
  const float inv = gs_inv(gs);                          // dy / dz travel multiplied by the loss scale; parameter gradients do not
  float* dp = det_part ? det_part + (long)blockIdx.x * 3 * H : nullptr;
  flush_cols(H, lane, w, pg, dgamma, sred, inv, dp);
  flush_cols(H, lane, w, pb, dbeta, sred, inv, dp ? dp + H : nullptr);
  if (dbias) flush_cols(H, lane, w, pz, dbias, sred, inv, dp ? dp + 2 * H : nullptr);
}

template <typename T, int VPL>
__global__ __launch_bounds__(256) void embed_ln_bwd_kernel(int rows, int H, int rows_per_block, const int* __restrict__ ids,
                                                           const int* __restrict__ pos, const float* __restrict__ word,
                                                           const float* __restrict__ posw, const float* __restrict__ typew,
                                                           const float* __restrict__ gamma, float eps,
                                                           const T* __restrict__ dyp, float* __restrict__ dword,
                                                           float* __restrict__ dpos, float* __restrict__ dtype0,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, DropCtx drop,
                                                           const float* __restrict__ gs, float* __restrict__ det_part,
                                                           float* __restrict__ det_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sred = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float inv = gs_inv(gs);
  float pg[VPL][4] = {}, pb[VPL][4] = {}, pz[VPL][4] = {};
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(rows, r0 + rows_per_block);
  for (int row = r0 + w; row < r1; row += 4) {
    const long wid = ids[row], pid = pos[row];
    const float* wr = word + wid * H;
    const float* pr = posw + pid * H;
    const T* dr = dyp + (long)row * H;
    float x[VPL][4], dy[VPL][4], dz[VPL][4];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        float a[4], b[4], t[4];
        ld4(wr + c, a);
        ld4(pr + c, b);
        ld4(typew + c, t);
        ld4(dr + c, dy[v]);
        if (drop.thr) { float m4[4]; drop_mult4(drop, (uint32_t)row, (uint32_t)c, m4); dy[v][0] *= m4[0]; dy[v][1] *= m4[1]; dy[v][2] *= m4[2]; dy[v][3] *= m4[3]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[v][e] = a[e] + b[e] + t[e]; sum += x[v][e]; }
      }
    }
    const float mu = wave_sum(sum) / (float)H;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) x[v][e] -= mu;
    ln_row_bwd(H, lane, x, dy, gamma, eps, dz, pg, pb);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        if (det_rows)                              // deterministic mode: the table scatters are done by det_scatter_rows_kernel
          *reinterpret_cast<float4*>(det_rows + (long)row * H + c) = make_float4(dz[v][0] * inv, dz[v][1] * inv, dz[v][2] * inv, dz[v][3] * inv);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!det_rows) {
            atomicAdd(dword + wid * H + c + e, dz[v][e] * inv);
            atomicAdd(dpos + pid * H + c + e, dz[v][e] * inv);
          }
          pz[v][e] += dz[v][e];
        }
      }
    }
  }
  float* dp = det_part ? det_part + (long)blockIdx.x * 3 * H : nullptr;
  flush_cols(H, lane, w, pg, dgamma, sred, inv, dp);
  flush_cols(H, lane, w, pb, dbeta, sred, inv, dp ? dp + H : nullptr);
  flush_cols(H, lane, w, pz, dtype0, sred, inv, dp ? dp + 2 * H : nullptr);
}

// Position-major embedding backward: wave w of block (pg, sc) owns ONE in-sequence position p = 4 pg + w and walks the
// sequences of chunk sc (row = cu[s] + p), so the position-embedding gradient of p accumulates in registers and is
// flushed with one atomic per column per wave -- the row-major kernel sends every row's 768 values to the same 128
// position rows (2048-way contended f32 atomics, half of its 400 M atomics).  The word-embedding scatter stays atomic
// (ids are arbitrary).  Requires pos_ids to depend on the in-sequence index only (the encoder driver guarantees it).
template <typename T, int VPL>
__global__ __launch_bounds__(256) void embed_ln_bwd_seq_kernel(int nseq, int seq_per_block, int H, const int* __restrict__ cu,
                                                               const int* __restrict__ ids, const int* __restrict__ pos,
                                                               const float* __restrict__ word, const float* __restrict__ posw,
                                                               const float* __restrict__ typew, const float* __restrict__ gamma,
                                                               float eps, const T* __restrict__ dyp, float* __restrict__ dword,
                                                               float* __restrict__ dpos, float* __restrict__ dtype0,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta, DropCtx drop,
                                                               const float* __restrict__ gs, float* __restrict__ det_part,
                                                               float* __restrict__ det_rows) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* sred = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const float inv = gs_inv(gs);
  // per-wave transposition patch for the word-embedding scatter: a lane owns 4 consecutive columns per vector, so an
  // atomic instruction issued from that layout touches 64 lanes x 4 B at a 16-B stride = 16 cache lines; through the
  // patch every instruction covers 64 CONSECUTIVE floats = 4 lines, a quarter of the L2 atomic transactions
  float* patch = sred + 4 * H + w * (VPL * 256);
  const int p = blockIdx.x * 4 + w;
  const int s0 = blockIdx.y * seq_per_block, s1 = min(nseq, s0 + seq_per_block);
  float pg[VPL][4] = {}, pb[VPL][4] = {}, pz[VPL][4] = {};
  long pid = -1;
  for (int sq = s0; sq < s1; ++sq) {
    const int t0 = cu[sq], len = cu[sq + 1] - t0;
    if (p >= len) continue;                          // wave-uniform
    const int row = t0 + p;
    const long wid = ids[row];
    pid = pos[row];
    const float* wr = word + wid * H;
    const float* pr = posw + pid * H;
    const T* dr = dyp + (long)row * H;
    float x[VPL][4], dy[VPL][4], dz[VPL][4];
    float sum = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        float a[4], b[4], t[4];
        ld4(wr + c, a);
        ld4(pr + c, b);
        ld4(typew + c, t);
        ld4(dr + c, dy[v]);
        if (drop.thr) { float m4[4]; drop_mult4(drop, (uint32_t)row, (uint32_t)c, m4); dy[v][0] *= m4[0]; dy[v][1] *= m4[1]; dy[v][2] *= m4[2]; dy[v][3] *= m4[3]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[v][e] = a[e] + b[e] + t[e]; sum += x[v][e]; }
      }
    }
    const float mu = wave_sum(sum) / (float)H;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) x[v][e] -= mu;
    ln_row_bwd(H, lane, x, dy, gamma, eps, dz, pg, pb);
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H) {
        *reinterpret_cast<float4*>(patch + c) = make_float4(dz[v][0] * inv, dz[v][1] * inv, dz[v][2] * inv, dz[v][3] * inv);
#pragma unroll
        for (int e = 0; e < 4; ++e) pz[v][e] += dz[v][e];
      }
    }
    __builtin_amdgcn_wave_barrier();
    float* wrow = det_rows ? det_rows + (long)row * H : dword + wid * H;   // deterministic mode: rows out, scattered in token order later
#pragma unroll
    for (int k = 0; k < VPL * 4; ++k) {
      const int c = k * 64 + lane;
      if (c < H) { if (det_rows) wrow[c] = patch[c]; else atomicAdd(wrow + c, patch[c]); }
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (pid >= 0 && !det_rows) {                       // this wave's position row: its sum over the chunk's sequences
#pragma unroll
    for (int v = 0; v < VPL; ++v) {
      const int c = (v * 64 + lane) * 4;
      if (c < H)
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(dpos + pid * H + c + e, pz[v][e] * inv);
    }
  }
  float* dp = det_part ? det_part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * 3 * H : nullptr;
  flush_cols(H, lane, w, pg, dgamma, sred, inv, dp);
  flush_cols(H, lane, w, pb, dbeta, sred, inv, dp ? dp + H : nullptr);
  flush_cols(H, lane, w, pz, dtype0, sred, inv, dp ? dp + 2 * H : nullptr);
}

template <typename T>
__global__ __launch_bounds__(256) void cls_gather_kernel(int nseq, int H, const int* __restrict__ cu,
                                                         const T* __restrict__ x, float* __restrict__ cls) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nseq * H) return;
  const int s = (int)(i / H), c = (int)(i % H);
  cls[i] = Elem<T>::ld(x + (long)cu[s] * H + c);
}

template <typename T>
__global__ __launch_bounds__(256) void cls_scatter_kernel(int nseq, int H, const int* __restrict__ cu,
                                                          const float* __restrict__ dcls, T* __restrict__ dx,
                                                          const float* __restrict__ gs) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)nseq * H) return;
  const int s = (int)(i / H), c = (int)(i % H);
  Elem<T>::st(dx + (long)cu[s] * H + c, dcls[i] * gs_scale(gs));
}

// dst[dst_idx ? dst_idx[s] : s] = src[src_idx ? src_idx[s] : s]   (row gather / scatter / dtype conversion)
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void rows_copy_kernel(int n, int H, const int* __restrict__ src_idx, const int* __restrict__ dst_idx,
                                                        const TI* __restrict__ src, TO* __restrict__ dst, const float* __restrict__ gs) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)n * H) return;
  const int s = (int)(i / H), c = (int)(i % H);
  const long rs = src_idx ? src_idx[s] : s, rd = dst_idx ? dst_idx[s] : s;
  const float v = Elem<TI>::ld(src + rs * H + c);
  Elem<TO>::st(dst + rd * H + c, gs ? v * gs[0] : v);       // (gs: an f32 gradient entering the fp16 backward)
}

// z[s] = dropout(y[s]; mask row key_idx[s]) + res[res_idx ? res_idx[s] : s]: what the dense GEMM's epilogue does for full
// tensors, for n rows that were gathered out of a larger tensor (the mask of a row is keyed by its ORIGINAL row index)
template <typename T>
__global__ __launch_bounds__(256) void drop_residual_rows_kernel(int n, int H, const T* __restrict__ y, const T* __restrict__ res,
                                                                 const int* __restrict__ res_idx, const int* __restrict__ key_idx,
                                                                 DropCtx drop, T* __restrict__ z) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (long)n * H) return;
  const int s = (int)(i / H), c = (int)(i % H);
  float v[4], r[4] = {0.f, 0.f, 0.f, 0.f};
  ld4(y + i, v);
  if (res) ld4(res + (long)(res_idx ? res_idx[s] : s) * H + c, r);       // (NULL: dropout only -- the stream_lo path adds the residual in LayerNorm)
  if (drop.thr) {
    float m4[4];
    drop_mult4(drop, (uint32_t)(key_idx ? key_idx[s] : s), (uint32_t)c, m4);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= m4[e];
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) v[e] += r[e];
  st4(z + i, v);
}

// rows of a residual-stream tensor (16-bit values + correction bytes, simx.h stream_lo): gather rows src_idx[s] into a compact
// pair (hi_out / lo_out, either may be NULL), and / or decode them to f32 (f32_out)
template <typename T>
__global__ __launch_bounds__(256) void stream_rows_kernel(int n, int H, const int* __restrict__ src_idx, const T* __restrict__ hi,
                                                          const uint8_t* __restrict__ lo, T* __restrict__ hi_out,
                                                          uint8_t* __restrict__ lo_out, float* __restrict__ f32_out) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (long)n * H) return;
  const int s = (int)(i / H), c = (int)(i % H);
  const long src = (long)(src_idx ? src_idx[s] : s) * H + c;
  Raw4<T> h;
  h.load(hi + src);
  const uint32_t q = lo ? *reinterpret_cast<const uint32_t*>(lo + src) : 0x80808080u;
  if (hi_out) {
    float v[4];
    h.unpack(v);
    st4(hi_out + i, v);                            // (exact: a 16-bit value through f32 and back)
  }
  if (lo_out) *reinterpret_cast<uint32_t*>(lo_out + i) = q;
  if (f32_out) {
    float v[4], a[4];
    h.unpack(v);
    lo8_decode4<T>(h.w0(), h.w1(), q, a);
    *reinterpret_cast<float4*>(f32_out + i) = make_float4(v[0] + a[0], v[1] + a[1], v[2] + a[2], v[3] + a[3]);
  }
}

// masked mean over each sequence's real tokens (packed layout: rows cu[s] .. cu[s+1]) and its adjoint
template <typename T>
__global__ __launch_bounds__(256) void seq_mean_fwd_kernel(int H, const int* __restrict__ cu, const T* __restrict__ x,
                                                           float* __restrict__ out) {
  const int s = blockIdx.x, t0 = cu[s], len = cu[s + 1] - t0;
  const float inv = 1.0f / (float)len;
  for (int c = threadIdx.x; c < H; c += 256) {
    float acc = 0.f;
    for (int r = 0; r < len; ++r) acc += Elem<T>::ld(x + (long)(t0 + r) * H + c);
    out[(long)s * H + c] = acc * inv;
  }
}
template <typename T>
__global__ __launch_bounds__(256) void seq_mean_bwd_kernel(int H, const int* __restrict__ cu, const float* __restrict__ dmean,
                                                           T* __restrict__ dx, const float* __restrict__ gs) {
  const int s = blockIdx.x, t0 = cu[s], len = cu[s + 1] - t0;
  const float inv = gs_scale(gs) / (float)len;
  for (int c = threadIdx.x; c < H; c += 256) {
    const float g = dmean[(long)s * H + c] * inv;
    for (int r = 0; r < len; ++r) Elem<T>::st(dx + (long)(t0 + r) * H + c, g);
  }
}

// ------------------------------------------------------------------------------------------ host
static int ln_check(int dtype, int T, int H, const char* who) {
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "%s: dtype %d", who, dtype);
  SIMX_REQUIRE(T > 0 && H > 0, SIMX_ERR_BAD_SHAPE, "%s: bad shape T=%d H=%d", who, T, H);
  SIMX_REQUIRE(H % 4 == 0 && H <= 64 * 4 * LN_VPL, SIMX_ERR_UNSUPPORTED, "%s: H=%d must be a multiple of 4 and <= 1024", who, H);
  return SIMX_OK;
}

// rows per block for the row-looping LN kernels: ~2 blocks per CU measured best on MI355X (sweep in DESIGN.md)
static int ln_rows_per_block(int T, const char* env, int dflt) {
  const char* e = getenv(env);
  int rpb = cdiv(T, e ? atoi(e) : dflt);
  if (rpb < 16) rpb = 16;
  return cdiv(rpb, 4) * 4;
}
static int bwd_rows_per_block(int T) { return ln_rows_per_block(T, "SIMX_LN_BWD_BLOCKS", 768); }
// vectors per lane by hidden size (186 -> 140 VGPRs for H = 768)
#define LN_BY_H(TT, LAUNCH) do { if (H <= 256) LAUNCH(TT, 1); else if (H <= 768) LAUNCH(TT, 3); else LAUNCH(TT, 4); } while (0)

extern "C" int simx_ln_fwd(simx_stream_t stream, int dtype, int T, int H, const void* z, const float* gamma,
                           const float* beta, float eps, void* y) {
  return simx_ln_fwd_res(stream, dtype, T, H, z, nullptr, nullptr, gamma, beta, eps, y, nullptr);
}

extern "C" int simx_ln_fwd_res(simx_stream_t stream, int dtype, int T, int H, const void* d, const void* res_hi, const void* res_lo,
                               const float* gamma, const float* beta, float eps, void* y, void* y_lo) {
  SIMX_PROF(SIMX_K_LN_FWD, stream, (double)T * H * ((2 + (res_hi ? 1 : 0)) * simx_esz(dtype) + (res_lo ? 1 : 0) + (y_lo ? 1 : 0)));
  SIMX_REQUIRE(!(res_lo || y_lo) || simx_is16(dtype), SIMX_ERR_BAD_DTYPE, "ln_fwd_res: the stream correction exists for the 16-bit dtypes only");
  int rc = ln_check(dtype, T, H, "ln_fwd");
  if (rc) return rc;
  SIMX_REQUIRE(res_hi || (!res_lo && !y_lo), SIMX_ERR_BAD_SHAPE, "ln_fwd_res: res_lo / y_lo need res_hi");
  hipStream_t s = (hipStream_t)stream;
  const int rpb = ln_rows_per_block(T, "SIMX_LN_FWD_BLOCKS", 1 << 30);   // 16 rows (4 per wave) per block measured best
  const size_t lds = (size_t)2 * H * sizeof(float);
#define LF(TT, V) do { if (res_hi) hipLaunchKernelGGL((ln_fwd_kernel<TT, V, true>), dim3(cdiv(T, rpb)), dim3(256), lds, s, T, H, rpb, (const TT*)d, gamma, beta, \
                                    eps, (TT*)y, (const TT*)res_hi, (const uint8_t*)res_lo, (uint8_t*)y_lo);                                    \
                       else hipLaunchKernelGGL((ln_fwd_kernel<TT, V, false>), dim3(cdiv(T, rpb)), dim3(256), lds, s, T, H, rpb, (const TT*)d, gamma, beta, \
                                    eps, (TT*)y, (const TT*)nullptr, (const uint8_t*)nullptr, (uint8_t*)nullptr); } while (0)
  SIMX_DISPATCH3(dtype, TT, LN_BY_H(TT, LF));
#undef LF
  SIMX_CHECK_LAUNCH("ln_fwd");
  return SIMX_OK;
}

// f32 engine with operand planes: y (f32) and its fp16 plane pair (y_planes, lo at + plane_stride elements; leading dimension H)
extern "C" int simx_ln_fwd_planes(simx_stream_t stream, int T, int H, const float* z, const float* gamma, const float* beta, float eps,
                                  float* y, void* y_planes, long plane_stride) {
  SIMX_PROF(SIMX_K_LN_FWD, stream, (double)T * H * 12);
  int rc = ln_check(SIMX_F32, T, H, "ln_fwd_planes");
  if (rc) return rc;
  SIMX_REQUIRE(y_planes && plane_stride % 4 == 0 && (((uintptr_t)y_planes) & 7) == 0, SIMX_ERR_BAD_SHAPE, "ln_fwd_planes: bad plane buffer");
  hipStream_t s = (hipStream_t)stream;
  const int rpb = ln_rows_per_block(T, "SIMX_LN_FWD_BLOCKS", 1 << 30);
  const size_t lds = (size_t)2 * H * sizeof(float);
#define LFP(TT, V) hipLaunchKernelGGL((ln_fwd_kernel<float, V, false>), dim3(cdiv(T, rpb)), dim3(256), lds, s, T, H, rpb, z, gamma, beta, eps, y, \
                                      (const float*)nullptr, (const uint8_t*)nullptr, (uint8_t*)nullptr, (bf16_t*)y_planes, plane_stride)
  LN_BY_H(float, LFP);
#undef LFP
  SIMX_CHECK_LAUNCH("ln_fwd_planes");
  return SIMX_OK;
}

extern "C" int simx_ln_bwd(simx_stream_t stream, int dtype, int T, int H, const void* z, const float* gamma, float eps,
                           const void* dy, void* dz, float* dgamma, float* dbeta, float* dbias) {
  return simx_ln_bwd_gs(stream, dtype, T, H, z, gamma, eps, dy, dz, nullptr, dgamma, dbeta, dbias, nullptr, nullptr, nullptr);
}

extern "C" int simx_ln_bwd_ex(simx_stream_t stream, int dtype, int T, int H, const void* z, const float* gamma, float eps,
                              const void* dy, void* dz, void* dz_masked, float* dgamma, float* dbeta, float* dbias,
                              const simx_dropout* dropd) {
  return simx_ln_bwd_gs(stream, dtype, T, H, z, gamma, eps, dy, dz, dz_masked, dgamma, dbeta, dbias, dropd, nullptr, nullptr);
}

extern "C" int simx_ln_bwd_keyed(simx_stream_t stream, int dtype, int T, int H, const void* z, const float* gamma, float eps,
                                 const void* dy, void* dz, void* dz_masked, float* dgamma, float* dbeta, float* dbias,
                                 const simx_dropout* dropd, const int32_t* row_keys) {
  return simx_ln_bwd_gs(stream, dtype, T, H, z, gamma, eps, dy, dz, dz_masked, dgamma, dbeta, dbias, dropd, row_keys, nullptr);
}

extern "C" int simx_ln_bwd_gs(simx_stream_t stream, int dtype, int T, int H, const void* z, const float* gamma, float eps,
                              const void* dy, void* dz, void* dz_masked, float* dgamma, float* dbeta, float* dbias,
                              const simx_dropout* dropd, const int32_t* row_keys, const float* gs) {
  return simx_ln_bwd_res(stream, dtype, T, H, z, nullptr, nullptr, gamma, eps, dy, dz, dz_masked, dgamma, dbeta, dbias, dropd, row_keys, gs);
}

extern "C" int simx_ln_bwd_res(simx_stream_t stream, int dtype, int T, int H, const void* z, const void* res_hi, const void* res_lo,
                               const float* gamma, float eps, const void* dy, void* dz, void* dz_masked, float* dgamma, float* dbeta,
                               float* dbias, const simx_dropout* dropd, const int32_t* row_keys, const float* gs) {
  SIMX_PROF(SIMX_K_LN_BWD, stream, (double)T * H * ((3 + (res_hi ? 1 : 0) + (dz_masked ? 1 : 0)) * simx_esz(dtype) + (res_lo ? 1 : 0)));
  SIMX_REQUIRE(!res_lo || simx_is16(dtype), SIMX_ERR_BAD_DTYPE, "ln_bwd_res: the stream correction exists for the 16-bit dtypes only");
  int rc = ln_check(dtype, T, H, "ln_bwd");
  if (rc) return rc;
  SIMX_REQUIRE(res_hi || !res_lo, SIMX_ERR_BAD_SHAPE, "ln_bwd_res: res_lo needs res_hi");
  const DropCtx drop = make_drop(dropd);
  SIMX_REQUIRE(!drop.thr || dz_masked, SIMX_ERR_BAD_SHAPE, "ln_bwd: dropout needs the dz_masked output");
  hipStream_t s = (hipStream_t)stream;
  const int rpb = bwd_rows_per_block(T);
  const size_t lds = (size_t)5 * H * sizeof(float);
  const int nblk = cdiv(T, rpb);
  float* det = nullptr;
  if (simx_det()) {
    det = simx_det_ws(s, (size_t)nblk * 3 * H * sizeof(float));
    if (!det) return SIMX_ERR_WORKSPACE;
  }
#define LB(TT, V) do { if (res_hi) hipLaunchKernelGGL((ln_bwd_kernel<TT, V, true>), dim3(cdiv(T, rpb)), dim3(256), lds, s, T, H, rpb, (const TT*)z, gamma, eps, \
                                    (const TT*)dy, (TT*)dz, dgamma, dbeta, dbias, (TT*)dz_masked, drop, row_keys, gs, (const TT*)res_hi, (const uint8_t*)res_lo, det); \
                       else hipLaunchKernelGGL((ln_bwd_kernel<TT, V, false>), dim3(cdiv(T, rpb)), dim3(256), lds, s, T, H, rpb, (const TT*)z, gamma, eps, \
                                    (const TT*)dy, (TT*)dz, dgamma, dbeta, dbias, (TT*)dz_masked, drop, row_keys, gs, (const TT*)nullptr, (const uint8_t*)nullptr, det); } while (0)
  SIMX_DISPATCH3(dtype, TT, LN_BY_H(TT, LB));
#undef LB
  SIMX_CHECK_LAUNCH("ln_bwd");
  if (det) return simx_det_reduce(s, det, 3L * H, nblk, H, dgamma, dbeta, dbias, gs);
  return SIMX_OK;
}

// f32 engine with operand planes: dz (f32, the residual branch's gradient) and the bf16 plane pair of the dropped dense
// output's gradient (dz x mask when dropout is on, dz otherwise) for the dgrad / wgrad GEMMs
extern "C" int simx_ln_bwd_planes(simx_stream_t stream, int T, int H, const float* z, const float* gamma, float eps, const float* dy, float* dz,
                                  void* dzm_planes, long plane_stride, float* dgamma, float* dbeta, float* dbias, const simx_dropout* dropd) {
  SIMX_PROF(SIMX_K_LN_BWD, stream, (double)T * H * 16);
  int rc = ln_check(SIMX_F32, T, H, "ln_bwd_planes");
  if (rc) return rc;
  SIMX_REQUIRE(dzm_planes && plane_stride % 4 == 0 && (((uintptr_t)dzm_planes) & 7) == 0, SIMX_ERR_BAD_SHAPE, "ln_bwd_planes: bad plane buffer");
  const DropCtx drop = make_drop(dropd);
  hipStream_t s = (hipStream_t)stream;
  const int rpb = bwd_rows_per_block(T);
  const size_t lds = (size_t)5 * H * sizeof(float);
  const int nblk = cdiv(T, rpb);
  float* det = nullptr;
  if (simx_det()) {
    det = simx_det_ws(s, (size_t)nblk * 3 * H * sizeof(float));
    if (!det) return SIMX_ERR_WORKSPACE;
  }
#define LBP(TT, V) hipLaunchKernelGGL((ln_bwd_kernel<float, V, false>), dim3(cdiv(T, rpb)), dim3(256), lds, s, T, H, rpb, z, gamma, eps, dy, dz, dgamma, dbeta, \
                                      dbias, (float*)nullptr, drop, (const int*)nullptr, (const float*)nullptr, (const float*)nullptr,                     \
                                      (const uint8_t*)nullptr, det, (bf16_t*)dzm_planes, plane_stride)
  LN_BY_H(float, LBP);
#undef LBP
  SIMX_CHECK_LAUNCH("ln_bwd_planes");
  if (det) return simx_det_reduce(s, det, 3L * H, nblk, H, dgamma, dbeta, dbias, nullptr);
  return SIMX_OK;
}

extern "C" int simx_embed_ln_fwd_planes(simx_stream_t stream, int T, int H, const int32_t* ids, const int32_t* pos_ids, const float* word,
                                        const float* posw, const float* typew, const float* gamma, const float* beta, float eps, float* out,
                                        void* out_planes, long plane_stride, const simx_dropout* dropd) {
  const DropCtx drop = make_drop(dropd);
  SIMX_PROF(SIMX_K_EMBED_FWD, stream, (double)T * H * 12);
  int rc = ln_check(SIMX_F32, T, H, "embed_ln_fwd_planes");
  if (rc) return rc;
  SIMX_REQUIRE(out_planes && plane_stride % 4 == 0 && (((uintptr_t)out_planes) & 7) == 0, SIMX_ERR_BAD_SHAPE, "embed_ln_fwd_planes: bad plane buffer");
  hipLaunchKernelGGL((embed_ln_fwd_kernel<float>), dim3(cdiv(T, 4)), dim3(256), 0, (hipStream_t)stream, T, H, ids, pos_ids, word, posw, typew, gamma,
                     beta, eps, out, drop, (uint8_t*)nullptr, (bf16_t*)out_planes, plane_stride);
  SIMX_CHECK_LAUNCH("embed_ln_fwd_planes");
  return SIMX_OK;
}

extern "C" int simx_embed_ln_fwd(simx_stream_t stream, int dtype, int T, int H, const int32_t* ids, const int32_t* pos_ids,
                                 const float* word, const float* posw, const float* typew, const float* gamma,
                                 const float* beta, float eps, void* out) {
  return simx_embed_ln_fwd_ex(stream, dtype, T, H, ids, pos_ids, word, posw, typew, gamma, beta, eps, out, nullptr);
}

extern "C" int simx_embed_ln_fwd_ex(simx_stream_t stream, int dtype, int T, int H, const int32_t* ids, const int32_t* pos_ids,
                                    const float* word, const float* posw, const float* typew, const float* gamma,
                                    const float* beta, float eps, void* out, const simx_dropout* dropd) {
  return simx_embed_ln_fwd_lo(stream, dtype, T, H, ids, pos_ids, word, posw, typew, gamma, beta, eps, out, nullptr, dropd);
}
extern "C" int simx_embed_ln_fwd_lo(simx_stream_t stream, int dtype, int T, int H, const int32_t* ids, const int32_t* pos_ids,
                                    const float* word, const float* posw, const float* typew, const float* gamma,
                                    const float* beta, float eps, void* out, void* out_lo, const simx_dropout* dropd) {
  const DropCtx drop = make_drop(dropd);
  SIMX_PROF(SIMX_K_EMBED_FWD, stream, (double)T * H * (4 + simx_esz(dtype)));
  int rc = ln_check(dtype, T, H, "embed_ln_fwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  SIMX_DISPATCH3(dtype, TT, hipLaunchKernelGGL((embed_ln_fwd_kernel<TT>), dim3(cdiv(T, 4)), dim3(256), 0, s, T, H, ids, pos_ids, word,
                                               posw, typew, gamma, beta, eps, (TT*)out, drop, (uint8_t*)out_lo));
  SIMX_CHECK_LAUNCH("embed_ln_fwd");
  return SIMX_OK;
}

// deterministic mode: the kernels left the scaled dz rows in det_rows and one column partial per workgroup in det_part
static int embed_det_finish(hipStream_t s, int T, int H, const int32_t* ids, const int32_t* pos_ids, const float* det_part,
                            const float* det_rows, int nblk, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                            const float* gs) {
  int rc = simx_det_scatter_rows(s, T, H, ids, det_rows, dword, 4096);
  if (rc) return rc;
  rc = simx_det_scatter_rows(s, T, H, pos_ids, det_rows, dpos, 512);
  if (rc) return rc;
  return simx_det_reduce(s, det_part, 3L * H, nblk, H, dgamma, dbeta, dtype0, gs);
}

extern "C" int simx_embed_ln_bwd(simx_stream_t stream, int dtype, int T, int H, const int32_t* ids, const int32_t* pos_ids,
                                 const float* word, const float* posw, const float* typew, const float* gamma, float eps,
                                 const void* dy, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta) {
  return simx_embed_ln_bwd_ex(stream, dtype, T, H, ids, pos_ids, word, posw, typew, gamma, eps, dy, dword, dpos, dtype0, dgamma,
                              dbeta, nullptr);
}

extern "C" int simx_embed_ln_bwd_ex(simx_stream_t stream, int dtype, int T, int H, const int32_t* ids, const int32_t* pos_ids,
                                    const float* word, const float* posw, const float* typew, const float* gamma, float eps,
                                    const void* dy, float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta,
                                    const simx_dropout* dropd) {
  const DropCtx drop = make_drop(dropd);
  SIMX_PROF(SIMX_K_EMBED_BWD, stream, (double)T * H * (12 + simx_esz(dtype)));
  int rc = ln_check(dtype, T, H, "embed_ln_bwd");
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int rpb = bwd_rows_per_block(T);
  const size_t lds = (size_t)4 * H * sizeof(float);
  const float* gs = nullptr;
  const int nblk = cdiv(T, rpb);
  float *det = nullptr, *det_rows = nullptr;
  if (simx_det()) {
    det = simx_det_ws(s, ((size_t)nblk * 3 * H + (size_t)T * H) * sizeof(float));
    if (!det) return SIMX_ERR_WORKSPACE;
    det_rows = det + (size_t)nblk * 3 * H;
  }
#define EB(TT, V) hipLaunchKernelGGL((embed_ln_bwd_kernel<TT, V>), dim3(nblk), dim3(256), lds, s, T, H, rpb, ids, pos_ids, word, posw, \
                                    typew, gamma, eps, (const TT*)dy, dword, dpos, dtype0, dgamma, dbeta, drop, gs, det, det_rows)
  SIMX_DISPATCH3(dtype, TT, LN_BY_H(TT, EB));
#undef EB
  SIMX_CHECK_LAUNCH("embed_ln_bwd");
  if (det) return embed_det_finish(s, T, H, ids, pos_ids, det, det_rows, nblk, dword, dpos, dtype0, dgamma, dbeta, gs);
  return SIMX_OK;
}

extern "C" int simx_embed_ln_bwd_seq(simx_stream_t stream, int dtype, int nseq, int max_len, int T, int H, const int32_t* cu_seqlens,
                                     const int32_t* ids, const int32_t* pos_ids, const float* word, const float* posw,
                                     const float* typew, const float* gamma, float eps, const void* dy, float* dword, float* dpos,
                                     float* dtype0, float* dgamma, float* dbeta, const simx_dropout* dropd) {
  return simx_embed_ln_bwd_seq_gs(stream, dtype, nseq, max_len, T, H, cu_seqlens, ids, pos_ids, word, posw, typew, gamma, eps, dy, dword,
                                  dpos, dtype0, dgamma, dbeta, dropd, nullptr);
}

extern "C" int simx_embed_ln_bwd_seq_gs(simx_stream_t stream, int dtype, int nseq, int max_len, int T, int H, const int32_t* cu_seqlens,
                                        const int32_t* ids, const int32_t* pos_ids, const float* word, const float* posw,
                                        const float* typew, const float* gamma, float eps, const void* dy, float* dword, float* dpos,
                                        float* dtype0, float* dgamma, float* dbeta, const simx_dropout* dropd, const float* gs) {
  SIMX_PROF(SIMX_K_EMBED_BWD, stream, (double)T * H * (simx_esz(dtype) + 3 * 4));
  const DropCtx drop = make_drop(dropd);
  int rc = ln_check(dtype, T, H, "embed_ln_bwd_seq");
  if (rc) return rc;
  SIMX_REQUIRE(nseq > 0 && max_len > 0 && cu_seqlens, SIMX_ERR_BAD_SHAPE, "embed_ln_bwd_seq: bad nseq / max_len / cu_seqlens");
  hipStream_t s = (hipStream_t)stream;
  const int pgroups = cdiv(max_len, 4);
  int chunks = 1024 / pgroups;                       // ~1024 blocks; every (position, chunk) pair flushes dpos once
  if (chunks < 1) chunks = 1;
  if (chunks > nseq) chunks = nseq;
  const int spb = cdiv(nseq, chunks);
  const dim3 grid(pgroups, cdiv(nseq, spb));
  const size_t lds = (size_t)(4 * H + 4 * LN_VPL * 256) * sizeof(float);      // column-flush scratch + per-wave scatter patches
  const int nblk = (int)(grid.x * grid.y);
  float *det = nullptr, *det_rows = nullptr;
  if (simx_det()) {
    det = simx_det_ws(s, ((size_t)nblk * 3 * H + (size_t)T * H) * sizeof(float));
    if (!det) return SIMX_ERR_WORKSPACE;
    det_rows = det + (size_t)nblk * 3 * H;
  }
#define ES(TT, V) hipLaunchKernelGGL((embed_ln_bwd_seq_kernel<TT, V>), grid, dim3(256), lds, s, nseq, spb, H, cu_seqlens, ids, pos_ids, word, \
                                    posw, typew, gamma, eps, (const TT*)dy, dword, dpos, dtype0, dgamma, dbeta, drop, gs, det, det_rows)
  SIMX_DISPATCH3(dtype, TT, LN_BY_H(TT, ES));
#undef ES
  SIMX_CHECK_LAUNCH("embed_ln_bwd_seq");
  if (det) return embed_det_finish(s, T, H, ids, pos_ids, det, det_rows, nblk, dword, dpos, dtype0, dgamma, dbeta, gs);
  return SIMX_OK;
}

extern "C" int simx_cls_gather(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu, const void* x, float* cls) {
  SIMX_REQUIRE(nseq > 0 && H > 0, SIMX_ERR_BAD_SHAPE, "cls_gather: bad shape");
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "cls_gather: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)(((long)nseq * H + 255) / 256);
  SIMX_DISPATCH3(dtype, TT, hipLaunchKernelGGL((cls_gather_kernel<TT>), dim3(blocks), dim3(256), 0, s, nseq, H, cu, (const TT*)x, cls));
  SIMX_CHECK_LAUNCH("cls_gather");
  return SIMX_OK;
}

extern "C" int simx_cls_scatter(simx_stream_t stream, int dtype, int nseq, int H, int T, const int32_t* cu,
                                const float* dcls, void* dx) {
  return simx_cls_scatter_gs(stream, dtype, nseq, H, T, cu, dcls, dx, nullptr);
}
extern "C" int simx_cls_scatter_gs(simx_stream_t stream, int dtype, int nseq, int H, int T, const int32_t* cu,
                                   const float* dcls, void* dx, const float* gs) {
  SIMX_REQUIRE(nseq > 0 && H > 0 && T >= nseq, SIMX_ERR_BAD_SHAPE, "cls_scatter: bad shape");
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "cls_scatter: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(dx, 0, (size_t)T * H * simx_esz(dtype), s) != hipSuccess) { simx_set_error("cls_scatter: memset failed"); return SIMX_ERR_HIP; }
  const int blocks = (int)(((long)nseq * H + 255) / 256);
  SIMX_DISPATCH3(dtype, TT, hipLaunchKernelGGL((cls_scatter_kernel<TT>), dim3(blocks), dim3(256), 0, s, nseq, H, cu, dcls, (TT*)dx, gs));
  SIMX_CHECK_LAUNCH("cls_scatter");
  return SIMX_OK;
}

extern "C" int simx_rows_copy(simx_stream_t stream, int src_dtype, int dst_dtype, int n, int H, const int32_t* src_idx,
                              const int32_t* dst_idx, const void* src, void* dst) {
  return simx_rows_copy_gs(stream, src_dtype, dst_dtype, n, H, src_idx, dst_idx, src, dst, nullptr);
}
extern "C" int simx_rows_copy_gs(simx_stream_t stream, int src_dtype, int dst_dtype, int n, int H, const int32_t* src_idx,
                                 const int32_t* dst_idx, const void* src, void* dst, const float* gs) {
  SIMX_REQUIRE(n > 0 && H > 0 && src && dst, SIMX_ERR_BAD_SHAPE, "rows_copy: bad arguments");
  SIMX_REQUIRE(simx_dtype_ok(src_dtype) && simx_dtype_ok(dst_dtype), SIMX_ERR_BAD_DTYPE, "rows_copy: dtypes %d -> %d", src_dtype, dst_dtype);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)(((long)n * H + 255) / 256);
  SIMX_DISPATCH3(src_dtype, TI, SIMX_DISPATCH3(dst_dtype, TO, hipLaunchKernelGGL((rows_copy_kernel<TI, TO>), dim3(blocks), dim3(256), 0, s, n,
                                                                                 H, src_idx, dst_idx, (const TI*)src, (TO*)dst, gs)));
  SIMX_CHECK_LAUNCH("rows_copy");
  return SIMX_OK;
}

extern "C" int simx_drop_residual_rows(simx_stream_t stream, int dtype, int n, int H, const void* y, const void* res,
                                       const int32_t* res_idx, const int32_t* key_idx, const simx_dropout* dropd, void* z) {
  SIMX_REQUIRE(n > 0 && H > 0 && H % 4 == 0 && y && z, SIMX_ERR_BAD_SHAPE, "drop_residual_rows: bad arguments");
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "drop_residual_rows: dtype %d", dtype);
  const DropCtx drop = make_drop(dropd);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)(((long)n * H / 4 + 255) / 256);
  SIMX_DISPATCH3(dtype, TT, hipLaunchKernelGGL((drop_residual_rows_kernel<TT>), dim3(blocks), dim3(256), 0, s, n, H, (const TT*)y,
                                               (const TT*)res, res_idx, key_idx, drop, (TT*)z));
  SIMX_CHECK_LAUNCH("drop_residual_rows");
  return SIMX_OK;
}

extern "C" int simx_stream_rows(simx_stream_t stream, int dtype, int n, int H, const int32_t* src_idx, const void* hi, const void* lo,
                                void* hi_out, void* lo_out, float* f32_out) {
  SIMX_REQUIRE(n > 0 && H > 0 && H % 4 == 0 && hi && (hi_out || lo_out || f32_out), SIMX_ERR_BAD_SHAPE, "stream_rows: bad arguments");
  SIMX_REQUIRE(simx_is16(dtype), SIMX_ERR_BAD_DTYPE, "stream_rows: 16-bit dtypes only (dtype %d)", dtype);
  hipStream_t s = (hipStream_t)stream;
  const int blocks = (int)(((long)n * H / 4 + 255) / 256);
  SIMX_DISPATCH16(dtype, TT, hipLaunchKernelGGL((stream_rows_kernel<TT>), dim3(blocks), dim3(256), 0, s, n, H, src_idx, (const TT*)hi,
                                                (const uint8_t*)lo, (TT*)hi_out, (uint8_t*)lo_out, f32_out));
  SIMX_CHECK_LAUNCH("stream_rows");
  return SIMX_OK;
}

extern "C" int simx_seq_mean_fwd(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu, const void* x, float* mean) {
  SIMX_REQUIRE(nseq > 0 && H > 0 && cu && x && mean, SIMX_ERR_BAD_SHAPE, "seq_mean_fwd: bad arguments");
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "seq_mean_fwd: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  SIMX_DISPATCH3(dtype, TT, hipLaunchKernelGGL((seq_mean_fwd_kernel<TT>), dim3(nseq), dim3(256), 0, s, H, cu, (const TT*)x, mean));
  SIMX_CHECK_LAUNCH("seq_mean_fwd");
  return SIMX_OK;
}

extern "C" int simx_seq_mean_bwd(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu, const float* dmean, void* dx) {
  return simx_seq_mean_bwd_gs(stream, dtype, nseq, H, cu, dmean, dx, nullptr);
}
extern "C" int simx_seq_mean_bwd_gs(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu, const float* dmean, void* dx,
                                    const float* gs) {
  SIMX_REQUIRE(nseq > 0 && H > 0 && cu && dmean && dx, SIMX_ERR_BAD_SHAPE, "seq_mean_bwd: bad arguments");
  SIMX_REQUIRE(simx_dtype_ok(dtype), SIMX_ERR_BAD_DTYPE, "seq_mean_bwd: dtype %d", dtype);
  hipStream_t s = (hipStream_t)stream;
  SIMX_DISPATCH3(dtype, TT, hipLaunchKernelGGL((seq_mean_bwd_kernel<TT>), dim3(nseq), dim3(256), 0, s, H, cu, dmean, (TT*)dx, gs));
  SIMX_CHECK_LAUNCH("seq_mean_bwd");
  return SIMX_OK;
}
