// D1: device-side batch assembly -- the collate of SimANS/utils/MARCO_until_new.py:204-258 on PRE-TOKENISED rows
// held in HBM (question table [NQ,QL], passage table [NP,PL], int32, rows = tokens + pad).  One wave per
// (query, passage) pair: gathers the passage row, builds the cross-encoder row
//   question tokens + passage tokens without the first one and without a trailing [SEP]   (:220-226)
// and the `ids != pad` masks, in the reference's int64 layout.  HBM-bound integer work: 4*(PL) B read,
// 16*(PL+CL) B written per pair.
#include "common.h"
#include "prof.h"

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

__global__ __launch_bounds__(256) void assemble_kernel(int B, int D, int QL, int PL, int CL, const int32_t* __restrict__ q_tok,
                                                       const int32_t* __restrict__ p_tok, const int32_t* __restrict__ q_rows,
                                                       const int32_t* __restrict__ p_rows, int pad, int sep,
                                                       long long* __restrict__ q_ids, long long* __restrict__ q_mask,
                                                       long long* __restrict__ c_ids, long long* __restrict__ c_mask,
                                                       long long* __restrict__ ce_ids, long long* __restrict__ ce_mask,
                                                       int32_t* __restrict__ q_len, int32_t* __restrict__ c_len,
                                                       int32_t* __restrict__ ce_len) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int r = blockIdx.x * 4 + w;
  if (r >= B * D) return;
  const int b = r / D;
  const int32_t* q = q_tok + (long)q_rows[b] * QL;
  const int32_t* p = p_tok + (long)p_rows[r] * PL;
  int ql = QL, pl = PL;                                   // first pad position = number of real tokens
  for (int c = lane; c < QL; c += 64) if (q[c] == pad) { ql = c; break; }
  for (int c = lane; c < PL; c += 64) if (p[c] == pad) { pl = c; break; }
  ql = wave_min_i(ql);
  pl = wave_min_i(pl);
  for (int c = lane; c < PL; c += 64) {
    const int t = p[c];
    c_ids[(long)r * PL + c] = t;
    c_mask[(long)r * PL + c] = t != pad;
  }
  if (r % D == 0) {
    for (int c = lane; c < QL; c += 64) {
      const int t = q[c];
      q_ids[(long)b * QL + c] = t;
      q_mask[(long)b * QL + c] = t != pad;
    }
    if (lane == 0 && q_len) q_len[b] = ql;
  }
  const int body = pl > 0 ? pl - 1 - (p[pl - 1] == sep ? 1 : 0) : 0;     // passage tokens kept: [1, 1+body)
  const int n = min(CL, ql + (body > 0 ? body : 0));
  for (int c = lane; c < CL; c += 64) {
    int t = pad;
    if (c < n) t = c < ql ? q[c] : p[1 + (c - ql)];
    ce_ids[(long)r * CL + c] = t;
    ce_mask[(long)r * CL + c] = t != pad;
  }
  if (lane == 0) {
    if (c_len) c_len[r] = pl;
    if (ce_len) ce_len[r] = n;
  }
}

extern "C" int simx_assemble_batch(simx_stream_t stream, int B, int D, int QL, int PL, int CL, const int32_t* q_tok,
                                   const int32_t* p_tok, const int32_t* q_rows, const int32_t* p_rows, int pad_id, int sep_id,
                                   int64_t* q_ids, int64_t* q_mask, int64_t* ctx_ids, int64_t* ctx_mask, int64_t* ce_ids,
                                   int64_t* ce_mask, int32_t* q_len, int32_t* ctx_len, int32_t* ce_len) {
  hipStream_t s = (hipStream_t)stream;
  SIMX_PROF(SIMX_K_COLLATE, s, (double)B * D * (4.0 * PL + 16.0 * (PL + CL)));
  SIMX_REQUIRE(B > 0 && D > 0 && QL > 0 && PL > 0 && CL > 0, SIMX_ERR_BAD_SHAPE, "assemble_batch: bad shape B=%d D=%d QL=%d PL=%d CL=%d", B, D, QL, PL, CL);
  SIMX_REQUIRE(q_tok && p_tok && q_rows && p_rows && q_ids && q_mask && ctx_ids && ctx_mask && ce_ids && ce_mask, SIMX_ERR_BAD_SHAPE,
               "assemble_batch: NULL argument");
  hipLaunchKernelGGL(assemble_kernel, dim3(cdiv(B * D, 4)), dim3(256), 0, s, B, D, QL, PL, CL, q_tok, p_tok, q_rows, p_rows, pad_id,
                     sep_id, (long long*)q_ids, (long long*)q_mask, (long long*)ctx_ids, (long long*)ctx_mask, (long long*)ce_ids,
                     (long long*)ce_mask, q_len, ctx_len, ce_len);
  SIMX_CHECK_LAUNCH("assemble_batch");
  return SIMX_OK;
}
