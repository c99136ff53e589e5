// clip_grad_norm_ + transformers.AdamW + zero_grad as two HBM-bound passes over the flat f32 buffers
// (SimANS/co_training/co_training_marco_train.py:57-69, 246-254; update rule SURVEY App. C).
// Pass 1 reads g (4 B/param) for the global L2 norm; pass 2 reads p,g,m,v and writes p,m,v,(g=0):
// 28 (+4) B/param, 16-byte accesses, no host synchronisation (the clip coefficient is computed on
// device from the norm accumulator).
#include "common.h"
#include "prof.h"

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, size_t n, float* __restrict__ out) {
  __shared__ float part[4];
  float s = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// deterministic variant: pass 1 leaves one partial per block (fixed element -> thread assignment, fixed in-block order), pass 2
// (one block) adds the partials in index order.  No float atomics: every rank of a data-parallel job gets the SAME bits
// from the same all-reduced gradients, so the clip coefficient -- and with it the replicas -- cannot drift apart.
__global__ __launch_bounds__(256) void sqnorm_partial_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partials) {
  __shared__ float part[4];
  float s = 0.f;
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ __launch_bounds__(256) void sqnorm_final_kernel(const float* __restrict__ partials, int nparts, float* __restrict__ out) {
  __shared__ float part[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += partials[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) *out += (part[0] + part[1]) + (part[2] + part[3]);
}

// Dynamic loss scaler of the fp16 engine (apex.amp's dynamic loss scaling, which the reference's --fp16 mode runs under:
// SimANS/co_training/co_training_marco_train.py:97-104, 218-220; apex defaults: start 2^16, halve on overflow, double after
// 2000 clean steps, cap 2^24).  State, 8 floats on the device:
//   [0] S   [1] 1/S   [2] clean steps since S last changed   [3] 1 = skip the current optimiser step (overflow)
//   [4] optimiser steps applied   [5] steps skipped   [6] growth interval   [7] largest S
// The gradient buffers hold TRUE gradients (every accumulating kernel multiplies by 1/S), so an overflow of the scaled fp16
// activation gradients shows as inf / nan in the squared norm.  No host synchronisation: the AdamW kernel reads [3], [4].
__global__ void scaler_init_kernel(float* __restrict__ st, float init_scale, float growth_interval, float max_scale) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  st[0] = init_scale; st[1] = 1.0f / init_scale; st[2] = 0.f; st[3] = 0.f; st[4] = 0.f; st[5] = 0.f; st[6] = growth_interval; st[7] = max_scale;
}
__global__ void scaler_update_kernel(float* __restrict__ st, const float* __restrict__ sqnorm) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float sq = *sqnorm;
  const bool ok = sq == sq && sq < 3.0e38f && sq >= 0.f;
  float S = st[0];
  if (!ok) {
    S = fmaxf(S * 0.5f, 1.0f);
    st[2] = 0.f; st[3] = 1.f; st[5] += 1.f;
  } else {
    st[3] = 0.f; st[4] += 1.f;
    const float clean = st[2] + 1.f;
    if (st[6] > 0.f && clean >= st[6]) { S = fminf(S * 2.0f, st[7]); st[2] = 0.f; } else st[2] = clean;
  }
  st[0] = S;
  st[1] = 1.0f / S;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float step_size, float beta1, float beta2,
                                                    float eps, float decay, const float* __restrict__ sqnorm, float max_norm,
                                                    float grad_scale, int zero_grad, const float* __restrict__ scaler, float lr) {
  float coef = grad_scale;
  if (sqnorm && max_norm > 0.f) {
    const float total = sqrtf(*sqnorm) * grad_scale;
    coef *= fminf(1.0f, max_norm / (total + 1e-6f));
  }
  if (scaler) {
    // the step count that enters the bias correction is the number of steps APPLIED (skipped ones do not age the moments)
    const double t = (double)scaler[4];
    step_size = (float)((double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t)));
    if (scaler[3] != 0.f) {                        // overflow: parameters and moments keep their values, gradients are dropped
      if (zero_grad) {
        const size_t n4z = n / 4;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4z; i += (size_t)gridDim.x * 256)
          reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (blockIdx.x == 0) for (size_t i = n4z * 4 + threadIdx.x; i < n; i += 256) g[i] = 0.f;
      }
      return;
    }
  }
  const size_t n4 = n / 4;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i], gv = reinterpret_cast<float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = gp[e] * coef;
      mp[e] = beta1 * mp[e] + (1.0f - beta1) * gg;
      vp[e] = beta2 * vp[e] + (1.0f - beta2) * gg * gg;
      pp[e] = pp[e] - step_size * mp[e] / (sqrtf(vp[e]) + eps);
      pp[e] = pp[e] - decay * pp[e];
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    if (zero_grad) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  if (blockIdx.x == 0)
    for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
      const float gg = g[i] * coef;
      m[i] = beta1 * m[i] + (1.0f - beta1) * gg;
      v[i] = beta2 * v[i] + (1.0f - beta2) * gg * gg;
      float pn = p[i] - step_size * m[i] / (sqrtf(v[i]) + eps);
      p[i] = pn - decay * pn;
      if (zero_grad) g[i] = 0.f;
    }
}

extern "C" int simx_sqnorm_accum(simx_stream_t stream, const float* g, size_t n, float* sqnorm) {
  SIMX_PROF(SIMX_K_ADAMW, stream, 4.0 * n);
  SIMX_REQUIRE(g && sqnorm && n > 0, SIMX_ERR_BAD_SHAPE, "sqnorm_accum: bad arguments");
  SIMX_REQUIRE((((uintptr_t)g) & 15) == 0, SIMX_ERR_BAD_SHAPE, "sqnorm_accum: g not 16-B aligned");
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sqnorm_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, sqnorm);
  SIMX_CHECK_LAUNCH("sqnorm");
  return SIMX_OK;
}

extern "C" int simx_sqnorm_accum_det(simx_stream_t stream, const float* g, size_t n, float* sqnorm, float* ws) {
  SIMX_PROF(SIMX_K_ADAMW, stream, 4.0 * n);
  SIMX_REQUIRE(g && sqnorm && ws && n > 0, SIMX_ERR_BAD_SHAPE, "sqnorm_accum_det: bad arguments");
  SIMX_REQUIRE((((uintptr_t)g) & 15) == 0, SIMX_ERR_BAD_SHAPE, "sqnorm_accum_det: g not 16-B aligned");
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > SIMX_SQNORM_WS_FLOATS) blocks = SIMX_SQNORM_WS_FLOATS;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, n, ws);
  SIMX_CHECK_LAUNCH("sqnorm_partial");
  hipLaunchKernelGGL(sqnorm_final_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const float*)ws, (int)blocks, sqnorm);
  SIMX_CHECK_LAUNCH("sqnorm_final");
  return SIMX_OK;
}

extern "C" int simx_scaler_init(simx_stream_t stream, float* state, float init_scale, float growth_interval, float max_scale) {
  SIMX_REQUIRE(state && init_scale >= 1.f && max_scale >= init_scale, SIMX_ERR_BAD_SHAPE, "scaler_init: bad arguments");
  hipLaunchKernelGGL(scaler_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, init_scale, growth_interval, max_scale);
  SIMX_CHECK_LAUNCH("scaler_init");
  return SIMX_OK;
}
extern "C" int simx_scaler_update(simx_stream_t stream, float* state, const float* sqnorm) {
  SIMX_REQUIRE(state && sqnorm, SIMX_ERR_BAD_SHAPE, "scaler_update: bad arguments");
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, sqnorm);
  SIMX_CHECK_LAUNCH("scaler_update");
  return SIMX_OK;
}

extern "C" int simx_adamw_step(simx_stream_t stream, float* p, float* g, float* m, float* v, size_t n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, const float* sqnorm, float max_norm,
                               float grad_scale, int zero_grad) {
  return simx_adamw_step_sc(stream, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, sqnorm, max_norm, grad_scale, zero_grad,
                            nullptr);
}
extern "C" int simx_adamw_step_sc(simx_stream_t stream, float* p, float* g, float* m, float* v, size_t n, float lr, float beta1,
                                  float beta2, float eps, float weight_decay, int step, const float* sqnorm, float max_norm,
                                  float grad_scale, int zero_grad, const float* scaler) {
  SIMX_PROF(SIMX_K_ADAMW, stream, 32.0 * n);
  SIMX_REQUIRE(p && g && m && v && n > 0 && step >= 1, SIMX_ERR_BAD_SHAPE, "adamw_step: bad arguments");
  SIMX_REQUIRE(((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0, SIMX_ERR_BAD_SHAPE,
               "adamw_step: buffers not 16-B aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr * sqrt(bc2) / bc1);
  size_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, step_size, beta1,
                     beta2, eps, lr * weight_decay, sqnorm, max_norm, grad_scale, zero_grad, scaler, lr);
  SIMX_CHECK_LAUNCH("adamw");
  return SIMX_OK;
}
