// Similarity + loss kernels (f32): wavefront-reduction kernels, one wave per query row.
//   sim_loss_kernel : M1 einsum("bh,bdh->bd") + {L1 KL-distill, L2 NQ/TQ normal+adv, L3 CE+KD, L6 CE}
//                     forward AND closed-form backward in one launch -- replaces the
//                     5+ ATen launches and two .item() syncs of co_training_marco_train.py:199-224.
//   nll rows        : M2 log_softmax / nll_loss / argmax over the all-pairs score matrix
//                     (SimANS/model/models.py:468-505) and dS in place.
#include "common.h"
#include "prof.h"

#define LOSS_EPS 1e-7f

// the four loss scalars of one row: f32 atomics, or (SIMX_DETERMINISTIC=1) row `row` of a [rows][4] partial buffer that
// det_reduce_kernel adds in row order
__device__ __forceinline__ void loss_row_out(float* losses, float* det_part, int row, float l0, float l1, float l2, float l3) {
  if (det_part) {
    *reinterpret_cast<float4*>(det_part + 4L * row) = make_float4(l0, l1, l2, l3);
  } else {
    atomicAdd(losses + 0, l0);
    if (l1 != 0.f) atomicAdd(losses + 1, l1);
    if (l2 != 0.f) atomicAdd(losses + 2, l2);
    atomicAdd(losses + 3, l3);
  }
}
static int loss_det_begin(hipStream_t s, int rows, float** part) {
  *part = nullptr;
  if (!simx_det()) return SIMX_OK;
  *part = simx_det_ws(s, (size_t)rows * 4 * sizeof(float));
  return *part ? SIMX_OK : SIMX_ERR_WORKSPACE;
}

__global__ __launch_bounds__(256) void sim_loss_kernel(int B, int D, int H, const float* __restrict__ q,
                                                       const float* __restrict__ ctx, const float* __restrict__ teacher,
                                                       simx_loss_params lp, float* __restrict__ sim,
                                                       float* __restrict__ losses, float* __restrict__ dq,
                                                       float* __restrict__ dctx, float* __restrict__ det_part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int b = blockIdx.x * 4 + w;
  if (b >= B) return;
  const bool act = lane < D;
  // ---- similarity: lane d ends up holding s[b,d]
  float s = 0.f;
  if (q) {
    const float* qr = q + (long)b * H;
    for (int d = 0; d < D; ++d) {
      const float* cr = ctx + ((long)b * D + d) * H;
      float part = 0.f;
      for (int h = lane; h < H; h += 64) part = fmaf(qr[h], cr[h], part);
      part = wave_sum(part);
      if (lane == d) s = part;
    }
    if (act) sim[(long)b * D + lane] = s;
  } else if (act) {
    s = sim[(long)b * D + lane];
  }
  const float NEG = -INFINITY;
  const float invB = 1.0f / (float)B;
  const float z = (act && teacher) ? teacher[(long)b * D + lane] : 0.f;
  float ds = 0.f, l0 = 0.f, l1 = 0.f, l2 = 0.f, corr = 0.f;
  // softmax of the (scaled) student logits
  const float ss = s * lp.scale;
  const float ms = wave_max(act ? ss : NEG);
  const float es = act ? expf(ss - ms) : 0.f;
  const float sums = wave_sum(es);
  const float p = es / sums;
  if (lp.kind == SIMX_LOSS_KL || lp.kind == SIMX_LOSS_WIKI) {
    const float zt = z / lp.temperature;
    const float mt = wave_max(act ? zt : NEG);
    const float et = act ? expf(zt - mt) : 0.f;
    const float t = et / wave_sum(et);
    const float lp_eps = logf(p + LOSS_EPS);
    float g;
    if (lp.kind == SIMX_LOSS_KL) {
      const float term = (act && t > 0.f) ? t * (logf(t) - lp_eps) : 0.f;
      l1 = wave_sum(term) * invB;                                  // distill loss
      l0 = l1;
      g = -t * invB / (p + LOSS_EPS);
    } else {
      // reward = log(softmax([z0, zd])[0] + eps)
      const float z0 = __shfl(z, 0, 64);
      const float mx = fmaxf(z0, z);
      const float r = logf(expf(z0 - mx) / (expf(z0 - mx) + expf(z - mx)) + LOSS_EPS);
      l1 = wave_sum(act ? -t * lp_eps : 0.f) * invB;               // normal
      l2 = wave_sum(act ? r * lp_eps : 0.f);                       // adv (not / B)
      l0 = lp.adv_lambda * l2 + (1.0f - lp.adv_lambda) * l1;
      g = (lp.adv_lambda * r - (1.0f - lp.adv_lambda) * t * invB) / (p + LOSS_EPS);
    }
    const float gp = wave_sum(act ? g * p : 0.f);
    ds = act ? p * (g - gp) * lp.scale : 0.f;
  } else {
    // hard CE on target 0 (log_softmax of the raw logits; scale is 1 for these losses)
    const float lsm = ss - ms - logf(sums);
    const float hard = -__shfl(lsm, 0, 64) * invB;
    const float s0 = __shfl(ss, 0, 64);
    corr = (s0 >= ms) ? 1.f : 0.f;
    const float e0 = lane == 0 ? 1.f : 0.f;
    if (lp.kind == SIMX_LOSS_CE) {
      l0 = hard; l1 = hard;
      ds = act ? (p - e0) * invB : 0.f;
    } else {
      const float Tm = lp.temperature;
      const float sT = s / Tm, zT = z / Tm;
      const float m1 = wave_max(act ? sT : NEG), m2 = wave_max(act ? zT : NEG);
      const float e1 = act ? expf(sT - m1) : 0.f, e2 = act ? expf(zT - m2) : 0.f;
      const float su1 = wave_sum(e1), su2 = wave_sum(e2);
      const float pT = e1 / su1, u = e2 / su2;
      const float lpT = sT - m1 - logf(su1);
      const float term = (act && u > 0.f) ? u * (logf(u) - lpT) : 0.f;
      const float soft = wave_sum(term) * Tm * Tm * invB;
      l1 = hard; l2 = soft;
      l0 = lp.ce_w * hard + lp.kd_w * soft;
      ds = act ? lp.ce_w * (p - e0) * invB + lp.kd_w * (Tm * invB) * (pT - u) : 0.f;
    }
  }
  const float ga = 1.0f / lp.grad_accum;
  ds *= ga;
  if (lane == 0) loss_row_out(losses, det_part, b, l0 * ga, l1, l2, corr);
  // ---- backward of the similarity
  if (q) {
    const float* qr = q + (long)b * H;
    for (int h = lane; h < H; h += 64) {
      float acc = 0.f;
      const float qv = qr[h];
      for (int d = 0; d < D; ++d) {
        const float dsd = __shfl(ds, d, 64);
        const long o = ((long)b * D + d) * H + h;
        acc = fmaf(dsd, ctx[o], acc);
        dctx[o] = dsd * qv;
      }
      dq[(long)b * H + h] = acc;
    }
  } else if (act) {
    dctx[(long)b * D + lane] = ds;
  }
}

// M2 rows: scores [Q,C] -> per-row loss/correct, then scores <- dS = (softmax - onehot) * gscale
__global__ __launch_bounds__(256) void nll_rows_kernel(int Q, int C, float* __restrict__ scores,
                                                       const int* __restrict__ pos_idx, float gscale, float lscale,
                                                       float* __restrict__ row_stats, float* __restrict__ losses,
                                                       float* __restrict__ det_part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + w;
  if (i >= Q) return;
  float* r = scores + (long)i * C;
  float m = -INFINITY;
  int am = 0x7FFFFFFF;
  for (int c = lane; c < C; c += 64) {
    const float v = r[c];
    if (v > m) { m = v; am = c; }
  }
  // wave arg-max with first-index tie break (torch.max semantics)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += expf(r[c] - m);
  sum = wave_sum(sum);
  const float lse = m + logf(sum);
  const int pos = pos_idx[i];
  const float sp = r[pos];
  if (row_stats) { if (lane == 0) { row_stats[2 * i] = m; row_stats[2 * i + 1] = lse; } }
  if (lane == 0) loss_row_out(losses, det_part, i, (lse - sp) * lscale / (float)Q, 0.f, 0.f, am == pos ? 1.f : 0.f);
  for (int c = lane; c < C; c += 64) {
    const float pr = expf(r[c] - lse);
    r[c] = (pr - (c == pos ? 1.f : 0.f)) * gscale;
  }
}

// L4 rows (BiEncoderKDLoss, KD_softmax): student scores S [Q,C] and teacher scores Z [Q,C] ->
//   hard = NLL(log_softmax(S), pos), soft = T^2 * sum_c u (log u - log_softmax(S/T)), u = softmax(Z/T)
// per row; then S <- dS = gscale * (ce_w (softmax(S) - onehot) + kd_w T (softmax(S/T) - u)).
__global__ __launch_bounds__(256) void kd_rows_kernel(int Q, int C, float* __restrict__ scores, const float* __restrict__ tscores,
                                                      const int* __restrict__ pos_idx, float T, float ce_w, float kd_w,
                                                      float gscale, float lscale, float* __restrict__ losses,
                                                      float* __restrict__ det_part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + w;
  if (i >= Q) return;
  float* r = scores + (long)i * C;
  const float* z = tscores + (long)i * C;
  const float iT = 1.0f / T;
  float m = -INFINITY, mz = -INFINITY;
  int am = 0x7FFFFFFF;
  for (int c = lane; c < C; c += 64) {
    const float v = r[c];
    if (v > m) { m = v; am = c; }
    mz = fmaxf(mz, z[c]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float om = __shfl_xor(m, o, 64);
    const int oa = __shfl_xor(am, o, 64);
    if (om > m || (om == m && oa < am)) { m = om; am = oa; }
  }
  mz = wave_max(mz);
  float s1 = 0.f, sT = 0.f, sz = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = r[c];
    s1 += expf(v - m);
    sT += expf((v - m) * iT);
    sz += expf((z[c] - mz) * iT);
  }
  s1 = wave_sum(s1); sT = wave_sum(sT); sz = wave_sum(sz);
  const float lse = m + logf(s1), lseT = m * iT + logf(sT), lseZ = mz * iT + logf(sz);
  const int pos = pos_idx[i];
  const float sp = r[pos];
  float soft = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = r[c];
    const float lu = z[c] * iT - lseZ;                 // log u
    const float u = expf(lu);
    const float lpT = v * iT - lseT;                   // log_softmax(S/T)
    if (u > 0.f) soft += u * (lu - lpT);
    const float p1 = expf(v - lse), pT = expf(lpT);
    r[c] = gscale * (ce_w * (p1 - (c == pos ? 1.f : 0.f)) + kd_w * T * (pT - u));
  }
  soft = wave_sum(soft) * T * T;
  if (lane == 0) {
    const float hard = lse - sp;
    loss_row_out(losses, det_part, i, (ce_w * hard + kd_w * soft) * lscale / (float)Q, hard / (float)Q, soft / (float)Q, am == pos ? 1.f : 0.f);
  }
}

// strided f32 GEMM from gemm.hip
extern "C" int simx_gemm_f32_strided(simx_stream_t stream, int M, int N, int K, const float* A, long a_rs, long a_cs,
                                     const float* B, long b_ks, long b_ns, float* C, int ldc, int accumulate);
extern "C" int simx_gemm_f32_strided_ws(simx_stream_t stream, int M, int N, int K, const float* A, long a_rs, long a_cs,
                                        const float* B, long b_ks, long b_ns, float* C, int ldc, int accumulate, void* ws,
                                        size_t ws_bytes);
extern "C" size_t simx_gemm_f32_workspace_bytes(int M, int N, int K);

// split-K workspace of the two backward products (dQ_loc: [q_n,H] over K = C; dC_loc: [c_n,H] over K = Q)
extern "C" size_t simx_scores_workspace_bytes(int Q, int C, int H, int q_n, int c_n) {
  const size_t a = q_n > 0 ? simx_gemm_f32_workspace_bytes(q_n, H, C) : 0, b = c_n > 0 ? simx_gemm_f32_workspace_bytes(c_n, H, Q) : 0;
  return a > b ? a : b;
}

extern "C" int simx_sim_loss_fwd_bwd(simx_stream_t stream, int B, int D, int H, const float* q, const float* ctx,
                                     const float* teacher, const simx_loss_params* lp, float* sim, float* losses,
                                     float* dq, float* dctx) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_LOSS, s, 16.0 * B * D * (H > 0 ? H : 1));
  SIMX_REQUIRE(lp != nullptr, SIMX_ERR_BAD_SHAPE, "sim_loss: params are NULL");
  SIMX_REQUIRE(B > 0 && D > 0 && D <= 64, SIMX_ERR_UNSUPPORTED, "sim_loss: need 0 < D=%d <= 64", D);
  SIMX_REQUIRE(lp->kind >= 0 && lp->kind <= 3, SIMX_ERR_UNSUPPORTED, "sim_loss: kind %d", lp->kind);
  SIMX_REQUIRE(lp->kind == SIMX_LOSS_CE || teacher != nullptr, SIMX_ERR_BAD_SHAPE, "sim_loss: teacher logits missing");
  SIMX_REQUIRE(q == nullptr || (ctx && dq && H > 0), SIMX_ERR_BAD_SHAPE, "sim_loss: q given without ctx/dq");
  SIMX_REQUIRE(sim && losses && dctx, SIMX_ERR_BAD_SHAPE, "sim_loss: NULL output");
  SIMX_REQUIRE(lp->grad_accum > 0.f && lp->temperature != 0.f, SIMX_ERR_BAD_SHAPE, "sim_loss: bad grad_accum/temperature");
  if (hipMemsetAsync(losses, 0, 4 * sizeof(float), s) != hipSuccess) { simx_set_error("sim_loss: memset failed"); return SIMX_ERR_HIP; }
  float* det = nullptr;
  if (int rcd = loss_det_begin(s, B, &det)) return rcd;
  hipLaunchKernelGGL(sim_loss_kernel, dim3(cdiv(B, 4)), dim3(256), 0, s, B, D, H, q, ctx, teacher, *lp, sim, losses, dq, dctx, det);
  SIMX_CHECK_LAUNCH("sim_loss");
  if (det) return simx_det_reduce(s, det, 4L, B, 4, losses, nullptr, nullptr, nullptr);
  return SIMX_OK;
}

extern "C" int simx_scores_nll_fwd_bwd(simx_stream_t stream, int Q, int C, int H, const float* q, const float* ctx,
                                       const int32_t* pos_idx, float loss_scale, int q_lo, int q_n, int c_lo, int c_n,
                                       float* scores, float* row_stats, float* losses, float* dq_local, float* dctx_local,
                                       void* ws, size_t ws_bytes) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_REQUIRE(Q > 0 && C > 0 && H > 0, SIMX_ERR_BAD_SHAPE, "scores_nll: bad shape");
  SIMX_REQUIRE(q_lo >= 0 && q_n >= 0 && q_lo + q_n <= Q && c_lo >= 0 && c_n >= 0 && c_lo + c_n <= C, SIMX_ERR_BAD_SHAPE,
               "scores_nll: local slot out of range");
  if (hipMemsetAsync(losses, 0, 4 * sizeof(float), s) != hipSuccess) { simx_set_error("scores_nll: memset failed"); return SIMX_ERR_HIP; }
  int rc = simx_gemm_f32_strided(stream, Q, C, H, q, H, 1, ctx, 1, H, scores, C, 0);       // S = q ctx^T
  if (rc) return rc;
  const float ls = loss_scale == 0.f ? 1.f : loss_scale;
  float* det = nullptr;
  if (int rcd = loss_det_begin(s, Q, &det)) return rcd;
  hipLaunchKernelGGL(nll_rows_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, s, Q, C, scores, pos_idx, ls / (float)Q, ls, row_stats, losses, det);
  SIMX_CHECK_LAUNCH("nll_rows");
  if (det) { rc = simx_det_reduce(s, det, 4L, Q, 4, losses, nullptr, nullptr, nullptr); if (rc) return rc; }
  if (q_n > 0 && dq_local) {                                                                // dQ_loc = dS[rows] ctx
    rc = simx_gemm_f32_strided_ws(stream, q_n, H, C, scores + (long)q_lo * C, C, 1, ctx, H, 1, dq_local, H, 0, ws, ws_bytes);
    if (rc) return rc;
  }
  if (c_n > 0 && dctx_local) {                                                              // dC_loc = dS[:,cols]^T q
    rc = simx_gemm_f32_strided_ws(stream, c_n, H, Q, scores + c_lo, 1, C, q, H, 1, dctx_local, H, 0, ws, ws_bytes);
    if (rc) return rc;
  }
  return SIMX_OK;
}

extern "C" int simx_scores_kd_fwd_bwd(simx_stream_t stream, int Q, int C, int H, int HT, const float* q, const float* ctx,
                                      const float* tq, const float* tctx, const int32_t* pos_idx, float temperature,
                                      float ce_w, float kd_w, float loss_scale, int q_lo, int q_n, int c_lo, int c_n,
                                      float* scores, float* tscores, float* losses, float* dq_local, float* dctx_local,
                                      void* ws, size_t ws_bytes) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_REQUIRE(Q > 0 && C > 0 && H > 0 && HT > 0, SIMX_ERR_BAD_SHAPE, "scores_kd: bad shape");
  SIMX_REQUIRE(temperature > 0.f, SIMX_ERR_BAD_SHAPE, "scores_kd: temperature must be > 0");
  SIMX_REQUIRE(q_lo >= 0 && q_n >= 0 && q_lo + q_n <= Q && c_lo >= 0 && c_n >= 0 && c_lo + c_n <= C, SIMX_ERR_BAD_SHAPE,
               "scores_kd: local slot out of range");
  SIMX_REQUIRE(q && ctx && tq && tctx && pos_idx && scores && tscores && losses, SIMX_ERR_BAD_SHAPE, "scores_kd: NULL argument");
  if (hipMemsetAsync(losses, 0, 4 * sizeof(float), s) != hipSuccess) { simx_set_error("scores_kd: memset failed"); return SIMX_ERR_HIP; }
  int rc = simx_gemm_f32_strided(stream, Q, C, H, q, H, 1, ctx, 1, H, scores, C, 0);          // S = q ctx^T
  if (rc) return rc;
  rc = simx_gemm_f32_strided(stream, Q, C, HT, tq, HT, 1, tctx, 1, HT, tscores, C, 0);        // Z = tq tctx^T
  if (rc) return rc;
  const float ls = loss_scale == 0.f ? 1.f : loss_scale;
  float* det = nullptr;
  if (int rcd = loss_det_begin(s, Q, &det)) return rcd;
  hipLaunchKernelGGL(kd_rows_kernel, dim3(cdiv(Q, 4)), dim3(256), 0, s, Q, C, scores, tscores, pos_idx, temperature, ce_w, kd_w,
                     ls / (float)Q, ls, losses, det);
  SIMX_CHECK_LAUNCH("kd_rows");
  if (det) { rc = simx_det_reduce(s, det, 4L, Q, 4, losses, nullptr, nullptr, nullptr); if (rc) return rc; }
  if (q_n > 0 && dq_local) {
    rc = simx_gemm_f32_strided_ws(stream, q_n, H, C, scores + (long)q_lo * C, C, 1, ctx, H, 1, dq_local, H, 0, ws, ws_bytes);
    if (rc) return rc;
  }
  if (c_n > 0 && dctx_local) {
    rc = simx_gemm_f32_strided_ws(stream, c_n, H, Q, scores + c_lo, 1, C, q, H, 1, dctx_local, H, 0, ws, ws_bytes);
    if (rc) return rc;
  }
  return SIMX_OK;
}
