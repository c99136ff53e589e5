/*
 * simx.h -- C ABI of libsimx_hip.so, the MI355X (gfx950) engine behind the
 * SimANS/co_training bi-encoder hot path.
 *
 * The reference (microsoft/SimXNS) has no FFI layer: its boundary is the Python
 * module API (SimANS/model/models.py, utils/dpr_utils.py).  This header is the
 * C-level boundary a maintainer binds instead (ctypes stub: simxns_amd/_lib.py;
 * see INTEGRATION.md).  Each entry point names the reference lines it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator),
 *     unless the name ends in _host;
 *   - no allocation and no stream synchronisation inside; scratch memory is passed
 *     in (size from the matching *_bytes query);
 *   - `stream` is a hipStream_t;
 *   - return value: SIMX_OK (0) or a negative SIMX_ERR_*; simx_last_error() gives
 *     the thread-local message;
 *   - dtype selects the activation / GEMM-operand type: SIMX_F32 (parity mode, exact
 *     f32 arithmetic), SIMX_BF16 (bf16 operands, f32 accumulate, f32 statistics -- EXPERIMENTAL: the round-1/2 engine,
 *     kept for A/B measurements; 5 % logits error on the hot fixture, not a product mode) or SIMX_F16
 *     (IEEE half operands -- the operand width of the reference's optional apex-O1 mode,
 *     SimANS/co_training/co_training_marco_train.py:97-104 -- f32 accumulate and statistics;
 *     its backward carries a loss scale, see "gradient scale" below).
 *     Parameters, gradients, LayerNorm statistics, embeddings handed to the loss and
 *     all loss arithmetic are f32 in every mode.
 *   - gradient scale (SIMX_F16): `gs` arguments are device pointers to two floats {S, 1/S}
 *     or NULL (= 1).  The gradient entering the encoder backward is multiplied by S when it is
 *     rounded to fp16 and every activation gradient travels scaled (apex.amp.scale_loss,
 *     co_training_marco_train.py:218-220); each kernel that accumulates into the f32 PARAMETER
 *     gradients multiplies by 1/S, so gradient buffers always hold true gradients and an
 *     overflow shows up there as inf / nan (simx_scaler_update reacts to it).
 *   - packed ("varlen") token layout: the T real tokens of nseq sequences are
 *     stored back to back; cu_seqlens[nseq+1] holds the prefix sums (int32).
 */
#ifndef SIMX_H
#define SIMX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* simx_stream_t;

enum { SIMX_OK = 0, SIMX_ERR_BAD_SHAPE = -1, SIMX_ERR_BAD_DTYPE = -2, SIMX_ERR_WORKSPACE = -3,
       SIMX_ERR_HIP = -4, SIMX_ERR_UNSUPPORTED = -5 };
enum { SIMX_F32 = 0, SIMX_BF16 = 1, SIMX_F16 = 2,
       /* GEMM entry points only (simx_gemm_nt*, simx_gemm_tn*): f32 tensors, products on the 16-bit matrix cores from an
        * on-the-fly split x = hi + lo of every operand element (hi.hi + hi.lo + lo.hi in f32 accumulators, three MFMAs per
        * product; csrc/gemm_x3.hip).  _H: halves in IEEE half -- ~2^-20..2^-22 relative for weights / O(1) activations, an
        * absolute floor of 2^-25 per element: the forward GEMMs of the fp32 engine.  _B: halves in bf16 -- 2^-17 relative at
        * any magnitude: its backward GEMMs (gradient operands span 1e-2..1e-9).  Shapes the split kernels do not take run the
        * exact f32 kernel. */
       SIMX_F32_SPLIT_H = 3, SIMX_F32_SPLIT_B = 4 };

int simx_version(void);
/* CUs the persistent kernels (gemm_nt_p3 / p5 / xp: one workgroup per CU with a static share of the tiles; the wgrad plans: one round
 * of the chip) may fill; 0 = all.  The data-parallel step sets ncu - (CUs of the RCCL ring) while the gradient all-reduce overlaps the
 * backward (DDP of SimANS/co_training/co_training_marco_train.py:107-114): a workgroup that finds no free CU starts only when a
 * resident one has finished its whole share and the launch takes up to twice as long (profiles/r06_cu_steal.json). */
int simx_set_compute_cus(int n);
const char* simx_last_error(void);

/* ------------------------------------------------------------------ GEMMs
 * The dense layers of BertSelfAttention / BertSelfOutput / BertIntermediate /
 * BertOutput (LEAD/modeling_bert.py:285-310, 385, 450, 463) and their backward. */
enum { SIMX_EPI_NONE = 0,   /* C = acc (+bias) (+residual)                              */
       SIMX_EPI_GELU = 1,   /* u = acc + bias: C2 = gelu_erf(u), C = gelu_erf'(u) -- the factor backward multiplies by,
                               kept instead of u itself (nothing downstream reads the pre-activation)             */
       SIMX_EPI_DGELU = 2,  /* C = (acc (+residual)) * aux,  aux = the C a SIMX_EPI_GELU launch wrote           */
       SIMX_EPI_GELU_INFER = 3, /* as GELU, but C is scratch: kernels may skip computing / storing it (no backward) */
       SIMX_EPI_NONE_PLANES = 4 /* simx_gemm_nt_planes only: acc + bias leaves as a plane pair (Cp), no f32 output */ };

/* Dropout descriptor (nn.Dropout of BertEmbeddings / BertSelfAttention / BertSelfOutput / BertOutput,
 * LEAD/modeling_bert.py:239, 358, 386, 464; p = 0.1 forced in training, SimANS/model/models.py:70-72).
 * The keep-mask is a stateless hash of (seed, stream, row, column): element (row, col) is kept iff byte (col & 3) of
 * mix32(seed, stream, row, col >> 2) is >= thr = round(256 p) (one hash per four columns: the drop probability is realised
 * in steps of 1/256, 26/256 = 0.1016 for p = 0.1); kept values are scaled by 256 / (256 - thr), the reciprocal of the
 * realised keep rate, so the mask has mean 1 exactly.  0 < p < 1/512 rounds up to 1/256, p > 255/256 down to 255/256.  Nothing
 * is stored: backward kernels recompute the mask from the same descriptor.  p == 0 (or a NULL descriptor) = no dropout. */
typedef struct simx_dropout {
  float p;
  uint32_t seed;
  uint32_t stream;      /* layer * 8 + site: 0 embeddings, 1 attention output, 2 FFN output, 3 attention probabilities */
} simx_dropout;

/* C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous; nn.Linear: B = weight [out,in]).
 * bias: f32 [N] or NULL; residual/aux/C/C2: same dtype as A, row strides in elements. */
int simx_gemm_nt(simx_stream_t stream, int dtype, int M, int N, int K,
                 const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                 const float* bias, const void* residual, int ldr, int epilogue,
                 const void* aux, int ldaux, void* C2, int ldc2);

/* same; with SIMX_EPI_NONE the dense result (acc + bias) goes through dropout `drop` (rows = M index, cols = N index)
 * BEFORE the residual is added -- BertSelfOutput / BertOutput order. */
int simx_gemm_nt_ex(simx_stream_t stream, int dtype, int M, int N, int K,
                    const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                    const float* bias, const void* residual, int ldr, int epilogue,
                    const void* aux, int ldaux, void* C2, int ldc2, const simx_dropout* drop);

/* C[M,N] (f32) (+)= A[K,M]^T . B[K,N]   (weight gradient: A = dY [T,out], B = X [T,in]).
 * Split over K into f32 slabs in `ws`, reduced deterministically. */
size_t simx_gemm_tn_workspace_bytes(int M, int N, int K);
int simx_gemm_tn(simx_stream_t stream, int dtype, int M, int N, int K,
                 const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                 int accumulate, void* ws, size_t ws_bytes);

/* same, and dbias[M] (f32, may be NULL) += column sums of A over K (the dense layer's bias gradient, fused into
 * the wgrad GEMM on the large-shape path; accumulates with atomics). */
int simx_gemm_tn_bias(simx_stream_t stream, int dtype, int M, int N, int K,
                      const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                      int accumulate, void* ws, size_t ws_bytes, float* dbias);

/* every form of the wgrad GEMM in one call (a_hm_rows > 0: A is head-major, see "head-major q / k / v"; lda then ignored),
 * with the gradient scale of the SIMX_F16 backward: A = S x dY, and C / dbias receive (1/S) x the products (gs = {S, 1/S}
 * on the device; NULL = 1). */
int simx_gemm_tn_gs(simx_stream_t stream, int dtype, int M, int N, int K,
                    const void* A, int lda, const void* B, int ldb, float* C, int ldc,
                    int accumulate, void* ws, size_t ws_bytes, float* dbias, int a_hm_rows, const float* gs);

/* out[N] (f32) (+)= column sums of x[T,N]  (bias gradients). */
int simx_colsum(simx_stream_t stream, int dtype, int T, int N, const void* x, int ldx,
                float* out, int accumulate);
int simx_colsum_gs(simx_stream_t stream, int dtype, int T, int N, const void* x, int ldx,
                   float* out, int accumulate, const float* gs);      /* sums multiplied by 1/S = gs[1] */

/* f32 master weight [rows,cols] -> bf16 copy and (optional) bf16 transposed copy [cols,rows]. */
int simx_cast_weight(simx_stream_t stream, const float* w, int rows, int cols, void* w_bf16, void* wT_bf16);
/* same with a selectable output dtype (either output may be NULL). */
int simx_transpose_cast(simx_stream_t stream, int out_dtype, const float* w, int rows, int cols, void* out, void* outT);
/* plain f32 GEMM with arbitrary element strides: C[m,n] (+)= sum_k A[m*a_rs + k*a_cs] * B[k*b_ks + n*b_ns]
 * (k-ordered f32 FMA chain; the all-pairs score matrix of M2 and its two backward products). */
int simx_gemm_f32_strided(simx_stream_t stream, int M, int N, int K, const float* A, long a_rs, long a_cs,
                          const float* B, long b_ks, long b_ns, float* C, int ldc, int accumulate);
/* same with a split-K workspace (simx_gemm_f32_workspace_bytes; may be NULL): problems whose 128x128 tile grid is small
 * while K is long are contracted in K slices into f32 slabs that are then added in slice order (deterministic). */
size_t simx_gemm_f32_workspace_bytes(int M, int N, int K);
int simx_gemm_f32_strided_ws(simx_stream_t stream, int M, int N, int K, const float* A, long a_rs, long a_cs,
                             const float* B, long b_ks, long b_ns, float* C, int ldc, int accumulate, void* ws, size_t ws_bytes);

/* --------------------------------------------------------- fp32 engine: dense GEMMs on pre-split operand planes
 * (csrc/gemm_xp.hip; the nn.Linear's of LEAD/modeling_bert.py:285-310, 385, 450, 463 in the fp32 arithmetic every
 * train_*_AR2.sh selects).  A "plane pair" of an f32 tensor x [rows, cols] is two 16-bit matrices of one format (`fmt`:
 * SIMX_F16 for forward operands, SIMX_BF16 for backward operands) with a common leading dimension: hi = rnd16(x) at the
 * pointer, lo = rnd16(x - hi) at pointer + plane_stride ELEMENTS.  Products are hi.lo + lo.hi + hi.hi in f32 accumulators
 * on the 16-bit matrix cores (as SIMX_F32_SPLIT_H / _B above), but the operands are split ONCE by their producer and
 * staged by LDS-DMA into the persistent 256x256 kernels.
 *   C[M,N] = A[M,K] . B[N,K]^T, A / B plane pairs (K-contiguous).  M, N % 256 == 0, K % 64 == 0, K >= 128
 *   (simx_gemm_nt_planes_ok).  Epilogues:
 *     SIMX_EPI_NONE        C (f32) = acc + bias, dropout `drop` (SIMX_F16 only, needs `in`), + in (f32 [M,N], may be NULL)
 *     SIMX_EPI_GELU        SIMX_F16: u = acc + bias; Cp = plane pair of gelu_erf(u); C (f32) = gelu_erf'(u)
 *     SIMX_EPI_GELU_INFER  as GELU, C not written (may be NULL)
 *     SIMX_EPI_DGELU       SIMX_BF16: Cp = plane pair of acc * in  (in = the C a GELU launch wrote)
 *     SIMX_EPI_NONE_PLANES SIMX_F16: Cp = plane pair of acc + bias (the QKV projection feeding simx_mha_*_x3) */
int simx_gemm_nt_planes_ok(int M, int N, int K);
int simx_gemm_nt_planes(simx_stream_t stream, int fmt, int epilogue, int M, int N, int K, const void* A, int lda, long a_plane_stride,
                        const void* B, int ldb, long b_plane_stride, float* C, int ldc, const float* bias, const float* in, int ldin,
                        void* Cp, int ldcp, long cp_plane_stride, const simx_dropout* drop);
/* same; SIMX_EPI_DGELU only: colsum[N] (may be NULL) += column sums of the output rows [0, rows_valid) -- the bias gradient of
 * the dense layer whose pre-activation gradient the launch produces (f32 atomics) */
int simx_gemm_nt_planes_cs(simx_stream_t stream, int fmt, int epilogue, int M, int N, int K, const void* A, int lda, long a_plane_stride,
                           const void* B, int ldb, long b_plane_stride, float* C, int ldc, const float* bias, const float* in, int ldin,
                           void* Cp, int ldcp, long cp_plane_stride, const simx_dropout* drop, float* colsum, int rows_valid);
/* wgrad: C[M,N] (+)= A[K,M]^T . B[K,N], A = dY, B = X as SIMX_BF16 plane pairs (K = tokens); dbias[M] (may be NULL) +=
 * column sums of A.  Split over K into f32 slabs added in slice order; ws >= simx_gemm_tn_planes_workspace_bytes. */
size_t simx_gemm_tn_planes_workspace_bytes(int M, int N, int K);
int simx_gemm_tn_planes(simx_stream_t stream, int M, int N, int K, const void* A, int lda, long a_plane_stride, const void* B, int ldb,
                        long b_plane_stride, float* C, int ldc, int accumulate, void* ws, size_t ws_bytes, float* dbias);
/* plane pair (dst_fmt) of an f32 matrix (src_fmt = SIMX_F32; src_plane_stride ignored) or of another plane pair (src_fmt =
 * its format); and back to f32.  cols % 8 == 0, 16-B aligned rows. */
int simx_planes_from(simx_stream_t stream, int src_fmt, int dst_fmt, int rows, int cols, const void* src, int ld_src, long src_plane_stride,
                     void* dst, int ld_dst, long dst_plane_stride);
int simx_planes_join(simx_stream_t stream, int src_fmt, int rows, int cols, const void* src, int ld_src, long src_plane_stride, float* dst,
                     int ld_dst);
/* operand forms of ONE dense weight W [rows, cols] f32 (an nn.Linear.weight, LEAD/modeling_bert.py:285-310, 385, 450, 463), as
 * simx_bert_cast_weights produces them for every weight of an fp32-engine tower once per optimiser step: the SIMX_F16 plane pair
 * of W (forward operand; lo plane at + rows * cols elements), the SIMX_BF16 plane pair of W^T [cols, rows] (dgrad operand) and
 * W^T in f32.  Any output may be NULL.  rows, cols multiples of 64 run on 64 x 64 tiles with 16-byte accesses. */
int simx_split_weight(simx_stream_t stream, const float* W, int rows, int cols, void* planes_f16, void* planesT_bf16, float* WT);

/* the producers that write plane pairs directly (fp32 engine; leading dimension of every pair = the tensor's own):
 *   LayerNorm / embedding LayerNorm: y (f32) and its SIMX_F16 pair (LEAD/modeling_bert.py:230-240, 384-388, 462-466);
 *   LayerNorm backward: dz (f32) and the SIMX_BF16 pair of dz x dropout mask (the gradient of the dropped dense output);
 *   attention (head size 64, sequences <= 4096: simx_mha_planes_ok): f32 q/k/v in, context as a SIMX_F16 pair; backward reads
 *   that pair and writes dq/dk/dv as a SIMX_BF16 pair (LEAD/modeling_bert.py:318-374). */
int simx_ln_fwd_planes(simx_stream_t stream, int T, int H, const float* z, const float* gamma, const float* beta, float eps,
                       float* y, void* y_planes, long plane_stride);
int simx_ln_bwd_planes(simx_stream_t stream, int T, int H, const float* z, const float* gamma, float eps, const float* dy, float* dz,
                       void* dzm_planes, long plane_stride, float* dgamma, float* dbeta, float* dbias, const simx_dropout* drop);
int simx_embed_ln_fwd_planes(simx_stream_t stream, int T, int H, const int32_t* ids, const int32_t* pos_ids, const float* word,
                             const float* posw, const float* typew, const float* gamma, const float* beta, float eps, float* out,
                             void* out_planes, long plane_stride, const simx_dropout* drop);
int simx_mha_planes_ok(int d, int max_len);
/* the same products on the 16-bit matrix cores from fp16 plane pairs (csrc/attention_x3.hip; head size 64, sequences <= 4096 -- K / V resident in LDS up to 160 tokens, 128-token chunks with an online softmax above:
 * simx_mha_x3_ok): q / k / v = the plane pair a SIMX_EPI_NONE_PLANES QKV projection wrote, context as a SIMX_F16 pair; backward
 * takes dctx in f32 and writes dq / dk / dv as a SIMX_BF16 pair.  Attention dropout p <= 0.9 (probabilities travel as
 * 2^10 p / (1 - p_drop) in fp16 halves); above that the calls return SIMX_ERR_UNSUPPORTED. */
int simx_mha_x3_ok(int d, int max_len);
int simx_mha_fwd_x3(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const void* qkv_planes,
                    long qkv_plane_stride, void* ctx_planes, long ctx_plane_stride, float* lse, const simx_dropout* drop);
int simx_mha_bwd_x3(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const void* qkv_planes,
                    long qkv_plane_stride, const void* ctx_planes, long ctx_plane_stride, const float* lse, const float* dctx,
                    void* dqkv_planes, long dqkv_plane_stride, const simx_dropout* drop);
/* same; dbias [3H] (may be NULL) += column sums of dq | dk | dv over the tokens: the QKV projection's bias gradient (f32 atomics) */
int simx_mha_bwd_x3_bias(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const void* qkv_planes,
                         long qkv_plane_stride, const void* ctx_planes, long ctx_plane_stride, const float* lse, const float* dctx,
                         void* dqkv_planes, long dqkv_plane_stride, const simx_dropout* drop, float* dbias);
int simx_mha_fwd_planes(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const float* qkv,
                        void* ctx_planes, long ctx_plane_stride, float* lse, const simx_dropout* drop);
int simx_mha_bwd_planes(simx_stream_t stream, int nseq, int heads, int d, const int32_t* cu, int max_len, int T, const float* qkv,
                        const void* ctx_planes, long ctx_plane_stride, const float* lse, const float* dctx, void* dqkv_planes,
                        long dqkv_plane_stride, const simx_dropout* drop);

/* --------------------------------------------------------- embeddings + LayerNorm
 * BertEmbeddings (LEAD/modeling_bert.py:181-240): LN(word[ids] + pos[pos_ids] + type[0]). */
int simx_embed_ln_fwd(simx_stream_t stream, int dtype, int T, int H,
                      const int32_t* ids, const int32_t* pos_ids,
                      const float* word, const float* posw, const float* typew,
                      const float* gamma, const float* beta, float eps, void* out);
/* accumulates into dword/dpos/dtype0/dgamma/dbeta (f32, atomics). */
int simx_embed_ln_bwd(simx_stream_t stream, int dtype, int T, int H,
                      const int32_t* ids, const int32_t* pos_ids,
                      const float* word, const float* posw, const float* typew,
                      const float* gamma, float eps, const void* dy,
                      float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta);

/* dropout variants: the forward masks the LayerNorm OUTPUT, the backward masks the incoming dy first. */
int simx_embed_ln_fwd_ex(simx_stream_t stream, int dtype, int T, int H,
                         const int32_t* ids, const int32_t* pos_ids,
                         const float* word, const float* posw, const float* typew,
                         const float* gamma, const float* beta, float eps, void* out, const simx_dropout* drop);
/* same, also writing the residual-stream correction bytes out_lo (uint8 [T,H], see simx_ln_fwd_res; may be NULL) */
int simx_embed_ln_fwd_lo(simx_stream_t stream, int dtype, int T, int H,
                         const int32_t* ids, const int32_t* pos_ids,
                         const float* word, const float* posw, const float* typew,
                         const float* gamma, const float* beta, float eps, void* out, void* out_lo, const simx_dropout* drop);
int simx_embed_ln_bwd_ex(simx_stream_t stream, int dtype, int T, int H,
                         const int32_t* ids, const int32_t* pos_ids,
                         const float* word, const float* posw, const float* typew,
                         const float* gamma, float eps, const void* dy,
                         float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta, const simx_dropout* drop);

/* position-major variant for the packed layout (cu_seqlens[nseq+1], rows cu[s]..cu[s+1]): one wave per in-sequence
 * position accumulates that position's gradient in registers (no contended atomics on the 128..512 position rows).
 * pos_ids must depend on the in-sequence index only (pos_ids[cu[s]+p] identical for all s). */
int simx_embed_ln_bwd_seq(simx_stream_t stream, int dtype, int nseq, int max_len, int T, int H, const int32_t* cu_seqlens,
                          const int32_t* ids, const int32_t* pos_ids,
                          const float* word, const float* posw, const float* typew,
                          const float* gamma, float eps, const void* dy,
                          float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta, const simx_dropout* drop);
/* same; dy carries the gradient scale, the five parameter gradients receive (1/S) x their sums */
int simx_embed_ln_bwd_seq_gs(simx_stream_t stream, int dtype, int nseq, int max_len, int T, int H, const int32_t* cu_seqlens,
                             const int32_t* ids, const int32_t* pos_ids,
                             const float* word, const float* posw, const float* typew,
                             const float* gamma, float eps, const void* dy,
                             float* dword, float* dpos, float* dtype0, float* dgamma, float* dbeta, const simx_dropout* drop,
                             const float* gs);

/* y = LN(z) ; z already holds dense(x)+bias+residual (BertSelfOutput / BertOutput,
 * LEAD/modeling_bert.py:384-388, 462-466). */
int simx_ln_fwd(simx_stream_t stream, int dtype, int T, int H, const void* z,
                const float* gamma, const float* beta, float eps, void* y);
/* "f32-grade residual stream" of the 16-bit engines (simx_bert_cfg.stream_lo): y = LN(d + res_hi + res_lo) where d is the
 * dense output (bias and dropout already applied by the GEMM epilogue; no residual there) and the residual stream is kept as
 * a 16-bit value plus a ONE-BYTE correction per element: x = hi + (b - 128) * ulp(hi) / 256 with ulp(hi) the spacing of the
 * 16-bit format at hi's exponent (res_lo / y_lo: uint8 [T,H]).  y leaves the same way.  The stream carries 19 significand bits
 * in fp16 (16 in bf16) at 3 B per element.  apex O1 (the reference's --fp16 mode) keeps this stream in fp32: residual additions
 * promote to fp32 and LayerNorm is an fp32 function there.  res_lo / y_lo may be NULL (plain 16-bit residual / output);
 * res_hi == NULL is simx_ln_fwd. */
int simx_ln_fwd_res(simx_stream_t stream, int dtype, int T, int H, const void* d, const void* res_hi, const void* res_lo,
                    const float* gamma, const float* beta, float eps, void* y, void* y_lo);
/* its backward: the LayerNorm input is rebuilt as d + res_hi + res_lo; everything else as simx_ln_bwd_gs. */
int simx_ln_bwd_res(simx_stream_t stream, int dtype, int T, int H, const void* d, const void* res_hi, const void* res_lo,
                    const float* gamma, float eps, const void* dy, void* dz, void* dz_masked,
                    float* dgamma, float* dbeta, float* dbias, const simx_dropout* drop, const int32_t* row_keys,
                    const float* gs);
/* dz = LN'(z) . dy ; dgamma/dbeta/dbias (colsum of dz, may be NULL) accumulate (f32 atomics). */
int simx_ln_bwd(simx_stream_t stream, int dtype, int T, int H, const void* z,
                const float* gamma, float eps, const void* dy, void* dz,
                float* dgamma, float* dbeta, float* dbias);

/* with dropout on the dense branch that produced z (z = drop(dense) + residual): dz (gradient of the residual branch)
 * and dz_masked = dz * mask / (1-p) (gradient of the dense output; feeds dgrad / wgrad); dbias = colsum(dz_masked). */
int simx_ln_bwd_ex(simx_stream_t stream, int dtype, int T, int H, const void* z,
                   const float* gamma, float eps, const void* dy, void* dz, void* dz_masked,
                   float* dgamma, float* dbeta, float* dbias, const simx_dropout* drop);
/* same, for T rows that were gathered out of a larger tensor: the dropout mask of row r is the one of row row_keys[r]
 * of that tensor (row_keys == NULL: identity).  Used by the encoder's [CLS]-only last layer. */
int simx_ln_bwd_keyed(simx_stream_t stream, int dtype, int T, int H, const void* z,
                      const float* gamma, float eps, const void* dy, void* dz, void* dz_masked,
                      float* dgamma, float* dbeta, float* dbias, const simx_dropout* drop, const int32_t* row_keys);
/* same with the gradient scale of the SIMX_F16 backward: dy, dz, dz_masked are S x the true gradients, what is added into
 * dgamma / dbeta / dbias is multiplied by 1/S (gs = {S, 1/S} on the device, NULL = 1; "gradient scale" at the top). */
int simx_ln_bwd_gs(simx_stream_t stream, int dtype, int T, int H, const void* z,
                   const float* gamma, float eps, const void* dy, void* dz, void* dz_masked,
                   float* dgamma, float* dbeta, float* dbias, const simx_dropout* drop, const int32_t* row_keys,
                   const float* gs);

/* ------------------------------------------------------------ self-attention
 * BertSelfAttention core (LEAD/modeling_bert.py:318-374): softmax(QK^T/sqrt(d)) V per head,
 * keys restricted to the sequence's own real tokens (== the additive finfo.min mask).
 * qkv [T,3H] rows = [q | k | v]; ctx [T,H]; lse [heads,T] f32 (log-sum-exp of scaled scores). */
int simx_mha_fwd(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim,
                 const int32_t* cu_seqlens, int max_len, int T,
                 const void* qkv, void* ctx, float* lse);
int simx_mha_bwd(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim,
                 const int32_t* cu_seqlens, int max_len, int T,
                 const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv);

/* dropout on the attention probabilities (rows = head*T + query token, cols = key index inside the sequence). */
int simx_mha_fwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim,
                    const int32_t* cu_seqlens, int max_len, int T,
                    const void* qkv, void* ctx, float* lse, const simx_dropout* drop);
int simx_mha_bwd_ex(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim,
                    const int32_t* cu_seqlens, int max_len, int T,
                    const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv,
                    const simx_dropout* drop);

/* Single-query attention for the [CLS]-only last layer: softmax(q0 K^T / sqrt(d)) V for token 0 of every sequence
 * (the only query whose output the path reads, SimANS/model/models.py:81).  q_cls / ctx_cls / dctx_cls / dq_cls are
 * compact [nseq, heads*head_dim] tensors; K and V are read from (dK, dV written to) columns [H, 3H) of the packed
 * qkv / dqkv rows; the Q columns of dqkv are not touched.  Same dropout masks as simx_mha_fwd_ex / simx_mha_bwd_ex. */
int simx_mha_cls_fwd(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim,
                     const int32_t* cu_seqlens, int max_len, int T,
                     const void* q_cls, const void* qkv, void* ctx_cls, const simx_dropout* drop);
int simx_mha_cls_bwd(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim,
                     const int32_t* cu_seqlens, int max_len, int T,
                     const void* q_cls, const void* qkv, const void* dctx_cls, void* dq_cls, void* dqkv,
                     const simx_dropout* drop);

/* ---- head-major q / k / v (performance layout; bf16, head size 64).  BertSelfAttention.transpose_for_scores
 * (LEAD/modeling_bert.py:312-316) turns [T, 3H] into per-head [S, 64] views; the token-major tensor leaves each
 * (sequence, head) block as S pieces of 128 B that sit 3H*2 B apart.  In the head-major form the packed tensor is
 * [3][heads][hm_rows][64] (hm_rows >= T, the row capacity of a plane): the QKV projection writes it
 * (simx_gemm_nt_hm, c_hm_rows), attention reads q/k/v and writes dq/dk/dv in it (the *_hm calls below; hm_rows = 0 is
 * the token-major form of the plain calls), the dgrad GEMM reads dq/dk/dv as its A operand (simx_gemm_nt_hm, a_hm_rows)
 * and the wgrad GEMM as its token-contracted operand (simx_gemm_tn_hm).  For the [CLS]-only last layer the K / V planes
 * are planes [heads, 3*heads).  Values are identical to the token-major path; only addresses change.
 * simx_mha_fwd_hm / simx_mha_bwd_hm accept hm_rows != 0 for max_len <= 256 only (the chunked kernels for longer sequences
 * are token-major; both calls return SIMX_ERR_UNSUPPORTED above that, so a forward that is accepted has a backward).
 * simx_gemm_hm_ok(rows, H, tokens) != 0 when the three GEMM forms exist for a tower of `rows` padded token rows (they
 * run on the persistent full-tile kernels only: rows % 256 == 0, H % 256 == 0, >= 192 output tiles at N = H). */
int simx_gemm_hm_ok(int rows, int H, int tokens);
int simx_gemm_nt_hm(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                    void* C, int ldc, const float* bias, const void* residual, int ldr, const simx_dropout* drop,
                    int a_hm_rows, int c_hm_rows);
/* the general plane-blocked form ([cols/64][rows][64] tensors; the FFN's [T, 3072] intermediates use it too).  flags: bit 0 = A,
 * bit 1 = C and C2, bit 2 = `in` (residual of SIMX_EPI_NONE / stored derivative of SIMX_EPI_DGELU).  Built combinations:
 * SIMX_EPI_NONE with flags 1 (with `in`) or 2 (without) -- the two q/k/v forms; anything else: SIMX_ERR_UNSUPPORTED
 * (plane-blocking the FFN tensors was measured and gains 1.5-4 %: not built in). */
int simx_gemm_nt_pb(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int lda, const void* B, int ldb,
                    void* C, int ldc, const float* bias, const void* in, int ldin, int epilogue, void* C2, int ldc2,
                    const simx_dropout* drop, int flags, int rows);
int simx_gemm_tn_hm(simx_stream_t stream, int dtype, int M, int N, int K, const void* A, int a_hm_rows, const void* B, int ldb,
                    float* C, int ldc, int accumulate, void* ws, size_t ws_bytes, float* dbias);
int simx_mha_fwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim, const int32_t* cu_seqlens, int max_len,
                    int T, const void* qkv, void* ctx, float* lse, const simx_dropout* drop, int hm_rows);
int simx_mha_bwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim, const int32_t* cu_seqlens, int max_len,
                    int T, const void* qkv, const void* ctx, const float* lse, const void* dctx, void* dqkv,
                    const simx_dropout* drop, int hm_rows);
int simx_mha_cls_fwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim, const int32_t* cu_seqlens,
                        int max_len, int T, const void* q_cls, const void* qkv, void* ctx_cls, const simx_dropout* drop,
                        int hm_rows);
int simx_mha_cls_bwd_hm(simx_stream_t stream, int dtype, int nseq, int heads, int head_dim, const int32_t* cu_seqlens,
                        int max_len, int T, const void* q_cls, const void* qkv, const void* dctx_cls, void* dq_cls, void* dqkv,
                        const simx_dropout* drop, int hm_rows);

/* [CLS] slice sequence_output[:,0,:] (SimANS/model/models.py:81) -> f32 [nseq,H], and its adjoint
 * (writes dcls into the first row of each sequence of dx, zero elsewhere). */
int simx_cls_gather(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu_seqlens,
                    const void* x, float* cls);
int simx_cls_scatter(simx_stream_t stream, int dtype, int nseq, int H, int T, const int32_t* cu_seqlens,
                     const float* dcls, void* dx);
/* dx = S * dcls: where an f32 gradient enters the SIMX_F16 backward (gs = {S, 1/S}, NULL = 1) */
int simx_cls_scatter_gs(simx_stream_t stream, int dtype, int nseq, int H, int T, const int32_t* cu_seqlens,
                        const float* dcls, void* dx, const float* gs);
/* Row gather / scatter / dtype conversion: dst[dst_idx ? dst_idx[s] : s] = src[src_idx ? src_idx[s] : s], s < n, rows of
 * H elements (sequence_output[:, 0, :] and its transpose for same-dtype tensors; SimANS/model/models.py:81). */
int simx_rows_copy(simx_stream_t stream, int src_dtype, int dst_dtype, int n, int H, const int32_t* src_idx,
                   const int32_t* dst_idx, const void* src, void* dst);
/* same, values multiplied by S = gs[0] on the way (gs == NULL: plain copy) */
int simx_rows_copy_gs(simx_stream_t stream, int src_dtype, int dst_dtype, int n, int H, const int32_t* src_idx,
                      const int32_t* dst_idx, const void* src, void* dst, const float* gs);
/* Rows of a residual-stream tensor (16-bit values hi + correction bytes lo, see simx_ln_fwd_res; lo may be NULL): gather rows
 * src_idx[s] (NULL: s) into a compact pair hi_out / lo_out and / or decode them to f32_out = hi + correction (each output may
 * be NULL).  The [CLS]-only last layer of a stream_lo tower uses it for the residual rows and for the f32 embeddings. */
int simx_stream_rows(simx_stream_t stream, int dtype, int n, int H, const int32_t* src_idx, const void* hi, const void* lo,
                     void* hi_out, void* lo_out, float* f32_out);
/* z[s] = dropout(y[s]) + res[res_idx ? res_idx[s] : s] on n gathered rows (res == NULL: dropout only); the mask of row s is the one of row
 * key_idx[s] of the full tensor (BertSelfOutput / BertOutput before the LayerNorm, LEAD/modeling_bert.py:384-388,
 * 462-466, restricted to the rows the path reads). */
int simx_drop_residual_rows(simx_stream_t stream, int dtype, int n, int H, const void* y, const void* res,
                            const int32_t* res_idx, const int32_t* key_idx, const simx_dropout* drop, void* z);

/* ------------------------------------------------- whole-encoder driver (native runtime)
 * HFBertEncoder.forward (SimANS/model/models.py:77-82 -> HF BertModel.forward, spec
 * LEAD/modeling_bert.py:916-1038) and its backward, sequenced in C++ over the kernels above.
 * Parameters live in ONE f32 buffer in the canonical layout described by
 * simx_bert_param_offset(); gradients in a buffer of the same layout (accumulated). */
typedef struct simx_bert_cfg {
  int32_t dtype, layers, hidden, heads, inter, vocab, max_pos, type_vocab;
  float eps;
  float hidden_dropout, attn_dropout;   /* 0 = off (eval / parity mode) */
  uint32_t dropout_seed;                /* per forward call; the matching backward call must pass the same value */
  /* 1: the caller only reads the [CLS] embedding (models.py:81 `sequence_output[:, 0, :]`), so the LAST layer's
   * attention-output, LayerNorm and FFN blocks run on the nseq [CLS] rows only -- every other row of that layer's output
   * is dead code on this path (it feeds neither cls_out nor any gradient).  cls_out and all gradients are unchanged;
   * hidden_out must be NULL.  The matching backward call must pass the same value.  0: every row is computed. */
  int32_t cls_only_last_layer;
  /* 1: gradient checkpointing per encoder layer (`cfg.gradient_checkpointing`, SimANS/model/models.py:73-74; every
   * train_*_AR2.sh passes --gradient_checkpointing): a forward with save_for_bwd keeps only each layer's INPUT
   * ([T,hidden] per layer instead of 8*hidden + 2*inter elements per token and layer) and the backward re-runs the layer's
   * forward -- same stateless dropout masks -- before differentiating it (4/3 of the FLOPs, as in the reference).
   * Must have the same value in simx_bert_act_bytes / simx_bert_fwd / simx_bert_bwd*.  Results are identical to 0. */
  int32_t grad_checkpoint;
  /* layout of the packed q / k / v tensor ("head-major q / k / v" above): 0 = chosen per call from the shapes (head-major
   * when every kernel that touches the tensor has the form), 1 = token-major pinned.  A forward and its backward must pass
   * the same value -- they then make the same choice (it depends on nothing but cfg, T, max_len). */
  int32_t qkv_layout;
  /* SIMX_F32 only.  0: the dense GEMMs run as SIMX_F32_SPLIT_H (forward) / SIMX_F32_SPLIT_B (backward) -- the fp32 engine's
   * default, ~4x the exact kernel's rate at parity two orders inside the 1e-3 tolerance; 1: exact f32 products
   * (v_mfma_f32_32x32x2_f32) everywhere. */
  int32_t f32_gemm;
  /* 16-bit dtypes only.  1: the residual stream (embedding output and every LayerNorm output) is kept as a 16-bit value plus
   * a one-byte correction per element and the residual additions move from the GEMM epilogues into the LayerNorm kernels,
   * which sum dense + hi + lo in f32 (simx_ln_fwd_res): 19 significand bits from layer to layer (fp16) at 3 B per element --
   * the role of apex O1's fp32 stream -- without a round trip of the pre-LayerNorm sum through HBM.  Must match between simx_bert_act_bytes / fwd /
   * bwd.  0: the stream is a plain 16-bit tensor, the residual rides in the GEMM epilogue. */
  int32_t stream_lo;
  /* SIMX_F16 backward: device pointer to {S, 1/S} (the loss scale, "gradient scale" above) or NULL = 1. */
  const float* grad_scale;
} simx_bert_cfg;

enum { SIMX_P_WORD = 0, SIMX_P_POS, SIMX_P_TYPE, SIMX_P_EMB_LN_G, SIMX_P_EMB_LN_B,   /* layer = -1 */
       SIMX_P_WQKV, SIMX_P_BQKV, SIMX_P_WO, SIMX_P_BO, SIMX_P_LN1_G, SIMX_P_LN1_B,
       SIMX_P_W1, SIMX_P_B1, SIMX_P_W2, SIMX_P_B2, SIMX_P_LN2_G, SIMX_P_LN2_B,        /* layer = 0..L-1 */
       SIMX_P_POOL_W, SIMX_P_POOL_B,                                                   /* layer = L */
       SIMX_P_END };
size_t simx_bert_param_count(const simx_bert_cfg* cfg);
/* element offset of tensor `which` of `layer` inside the flat buffer; (size_t)-1 on bad input */
size_t simx_bert_param_offset(const simx_bert_cfg* cfg, int layer, int which);
size_t simx_bert_wcache_bytes(const simx_bert_cfg* cfg);               /* bf16 weight + transposed copies */
size_t simx_bert_act_bytes(const simx_bert_cfg* cfg, int T, int nseq, int save_for_bwd);
size_t simx_bert_bwd_scratch_bytes(const simx_bert_cfg* cfg, int T, int nseq);
int simx_bert_cast_weights(simx_stream_t stream, const simx_bert_cfg* cfg, const float* params, void* wcache);
/* cls_out f32 [nseq,hidden]; hidden_out (may be NULL) receives the packed last hidden state [T,hidden]
 * in the activation dtype. */
int simx_bert_fwd(simx_stream_t stream, const simx_bert_cfg* cfg, const float* params, const void* wcache,
                  const int32_t* ids, const int32_t* pos_ids, const int32_t* cu_seqlens,
                  int nseq, int T, int max_len, void* act, size_t act_bytes, int save_for_bwd,
                  float* cls_out, void* hidden_out);
int simx_bert_bwd(simx_stream_t stream, const simx_bert_cfg* cfg, const float* params, const void* wcache,
                  const int32_t* ids, const int32_t* pos_ids, const int32_t* cu_seqlens,
                  int nseq, int T, int max_len, const void* act, size_t act_bytes,
                  const float* dcls, float* grads, void* scratch, size_t scratch_bytes);
/* same with the upstream gradient given for the WHOLE last hidden state (sequence_output of HFBertEncoder.forward,
 * models.py:77-82; e.g. the masked-mean pooling of EmbeddingMixin, models.py:296-305): dhidden [T,hidden] in the
 * activation dtype.  Exactly one of dcls / dhidden is non-NULL; dhidden needs cfg->cls_only_last_layer == 0. */
/* The backward in parts, for overlapping the data-parallel gradient all-reduce (DistributedDataParallel's bucketed
 * reduction, SimANS/co_training/co_training_marco_train.py:107-114) with the rest of the backward: this call
 * differentiates encoder layers layer_hi, layer_hi-1, ..., layer_lo; layer_hi == layers-1 seeds from dcls / dhidden,
 * layer_lo == 0 also runs the embedding backward.  Parts must be called top-down with the SAME `scratch` (it carries the
 * inter-layer gradient between calls).  Because the flat gradient buffer is laid out embeddings | layer 0 | ... | layer
 * L-1 | pooler, after a part returns the slice [simx_bert_param_offset(cfg, layer_lo, SIMX_P_WQKV), end of the previous
 * part's slice) is final and can be reduced while the next part runs.  simx_bert_bwd_ex == range(layers-1, 0). */
int simx_bert_bwd_range(simx_stream_t stream, const simx_bert_cfg* cfg, const float* params, const void* wcache,
                        const int32_t* ids, const int32_t* pos_ids, const int32_t* cu_seqlens,
                        int nseq, int T, int max_len, void* act, size_t act_bytes,
                        const float* dcls, const void* dhidden, float* grads, void* scratch, size_t scratch_bytes,
                        int layer_hi, int layer_lo);
int simx_bert_bwd_ex(simx_stream_t stream, const simx_bert_cfg* cfg, const float* params, const void* wcache,
                     const int32_t* ids, const int32_t* pos_ids, const int32_t* cu_seqlens,
                     int nseq, int T, int max_len, const void* act, size_t act_bytes,
                     const float* dcls, const void* dhidden, float* grads, void* scratch, size_t scratch_bytes);
/* masked mean over the real tokens of each sequence of a packed [T,H] tensor -> f32 [nseq,H] (EmbeddingMixin.masked_mean,
 * SimANS/model/models.py:296-299), and its adjoint (dx[t] = dmean[seq(t)] / len) */
int simx_seq_mean_fwd(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu_seqlens, const void* x, float* mean);
int simx_seq_mean_bwd(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu_seqlens, const float* dmean, void* dx);
int simx_seq_mean_bwd_gs(simx_stream_t stream, int dtype, int nseq, int H, const int32_t* cu_seqlens, const float* dmean, void* dx,
                         const float* gs);   /* dx multiplied by S = gs[0]: the f32 gradient enters the SIMX_F16 backward here */

/* -------------------------------------------------- similarity + losses (f32)
 * M1 local similarity einsum("bh,bdh->bd") (co_training_marco_train.py:199-202) fused with the
 * step loss and its closed-form backward:
 *   SIMX_LOSS_KL    L1  KLDiv(batchmean)(log(softmax(s*scale)+1e-7), softmax(z/temp))  co_training_marco_train.py:203-217
 *   SIMX_LOSS_WIKI  L2  adv_lambda*adv + (1-adv_lambda)*normal                            co_training_wiki_train.py:203-228
 *   SIMX_LOSS_CEKD  L3  ce_w*NLL(target 0) + kd_w*T^2*KL(softmax(z/T)||softmax(s/T))    PROD/ProD_KD/model/models.py:668-781
 *   SIMX_LOSS_CE    L6  CrossEntropy(target 0) on s itself                                co_training_marco_train.py:228-236
 * q [B,H], ctx [B*D,H], teacher [B,D] (ignored for CE) ; outputs: sim [B,D], losses[4] =
 * {loss, aux0, aux1, correct_count}, dq [B,H], dctx [B*D,H] (gradients already divided by grad_accum).
 * If q == NULL the D logits are read from `sim` directly (no similarity, dctx receives ds [B,D]). */
enum { SIMX_LOSS_KL = 0, SIMX_LOSS_WIKI = 1, SIMX_LOSS_CEKD = 2, SIMX_LOSS_CE = 3 };
typedef struct simx_loss_params {
  int32_t kind;
  float scale;          /* 1 or 1/sqrt(H) (--scale_simmila) */
  float temperature;    /* temperature_distill / temperature_normal / TEMPERATURE */
  float adv_lambda;     /* L2 */
  float ce_w, kd_w;     /* L3 */
  float grad_accum;     /* loss and grads divided by this */
} simx_loss_params;
int simx_sim_loss_fwd_bwd(simx_stream_t stream, int B, int D, int H,
                          const float* q, const float* ctx, const float* teacher,
                          const simx_loss_params* lp_host, float* sim, float* losses,
                          float* dq, float* dctx);

/* M2: dot_product_scores + BiEncoderNllLoss.calc (SimANS/model/models.py:468-505, 564-572) with the
 * multi-GPU semantics of caculate_cont_loss (PROD/ProD_base/train_DE_model_marco.py:224-278):
 * q [Q,H], ctx [C,H] are the rank-ordered global concatenations; gradients are produced only for
 * rows [q_lo,q_lo+q_n) of q and [c_lo,c_lo+c_n) of ctx (the local slots).  scores [Q,C] f32 scratch.
 * losses[4] = {loss, 0, 0, correct_count}; loss_scale multiplies loss and grads (1 = none).
 * ws / ws_bytes: split-K workspace of the two backward products (simx_scores_workspace_bytes; NULL = unsplit: correct,
 * but dQ of a gathered batch -- q_n x H outputs contracted over all C passages -- then runs on a handful of CUs). */
size_t simx_scores_workspace_bytes(int Q, int C, int H, int q_n, int c_n);
int simx_scores_nll_fwd_bwd(simx_stream_t stream, int Q, int C, int H,
                            const float* q, const float* ctx, const int32_t* pos_idx,
                            float loss_scale, int q_lo, int q_n, int c_lo, int c_n,
                            float* scores, float* row_stats, float* losses, float* dq_local, float* dctx_local,
                            void* ws, size_t ws_bytes);

/* L4: BiEncoderKDLoss.calc, KD_softmax (PROD/ProD_KD/model/models.py:970-1038, kd_loss :772-781) on all-pairs scores
 * of the student (q [Q,H], ctx [C,H]) and of the teacher embeddings (tq [Q,HT], tctx [C,HT], constants):
 *   loss = ce_w * NLL(log_softmax(S), pos) + kd_w * T^2 * mean_q sum_c u (log u - log_softmax(S/T)),  u = softmax(Z/T).
 * Same local-slot gradient semantics as simx_scores_nll_fwd_bwd.  scores / tscores: [Q,C] f32 scratch.
 * losses[4] = {loss, hard, soft, correct_count}. */
int simx_scores_kd_fwd_bwd(simx_stream_t stream, int Q, int C, int H, int HT,
                           const float* q, const float* ctx, const float* tq, const float* tctx,
                           const int32_t* pos_idx, float temperature, float ce_w, float kd_w, float loss_scale,
                           int q_lo, int q_n, int c_lo, int c_n,
                           float* scores, float* tscores, float* losses, float* dq_local, float* dctx_local,
                           void* ws, size_t ws_bytes);

/* -------------------------------------------------------------- D1: device-side batch assembly
 * The collate of SimANS/utils/MARCO_until_new.py:204-258 (Rocketqa_v2Dataset.__getitem__ tail +
 * create_biencoder_input2) on PRE-TOKENISED rows resident in HBM: q_tok [NQ,QL], p_tok [NP,PL] int32, each row =
 * tokens (special tokens included) + pad_id.  For query b (row q_rows[b]) and its D passages (rows p_rows[b*D+d],
 * d = 0 the positive): q_ids/q_mask [B,QL], ctx_ids/ctx_mask [B*D,PL], and the cross-encoder rows
 * ce_ids/ce_mask [B*D,CL] = question tokens + passage tokens without the first one and without a trailing sep_id
 * (remove_special_token, :220-224), truncated to CL, padded with pad_id; masks are ids != pad_id.  int64 like the
 * reference's LongTensors.  Optional (may be NULL) token counts q_len [B], ctx_len [B*D], ce_len [B*D]. */
int simx_assemble_batch(simx_stream_t stream, int B, int D, int QL, int PL, int CL,
                        const int32_t* q_tok, const int32_t* p_tok, const int32_t* q_rows, const int32_t* p_rows,
                        int pad_id, int sep_id,
                        int64_t* q_ids, int64_t* q_mask, int64_t* ctx_ids, int64_t* ctx_mask,
                        int64_t* ce_ids, int64_t* ce_mask, int32_t* q_len, int32_t* ctx_len, int32_t* ce_len);

/* -------------------------------------------------------------- generate job: exhaustive inner-product search
 * Replaces faiss.IndexFlatIP + index_cpu_to_all_gpus(shard=True) + index.search(q, 200 | 1000)
 * (SimANS/co_training/co_training_generate.py:359-384, 415-421) on a corpus shard resident in HBM.
 * Scores are float32, one fused-multiply-add chain per (query, passage) in ascending h (bit-reproducible; the
 * restatement is oracle/topk_ref.c); results per query are sorted by descending score, ties by ascending id.
 *
 * simx_ip_scores       scores[nq, ld] (first nc columns) = q[nq,H] . c[nc,H]^T
 * simx_topk_update     folds m new candidates per query -- scores[nq, ld] with ids[nq, ld_ids] (int64, < 2^32, negative =
 *                      skip) or, when ids == NULL, implicit ids id_base + column -- into the running result
 *                      run_scores / run_ids [nq,k] (initialise to -inf / -1; k <= 1024).  Also the cross-shard merge.
 * simx_flat_ip_search  = index.search: walks the shard in `chunk`-passage pieces (scores chunk -> top-k update);
 *                      workspace >= simx_flat_ip_workspace_bytes(nq, chunk); ids returned are id_base + row. */
int simx_ip_scores(simx_stream_t stream, int nq, int nc, int H, const float* q, const float* c, float* scores, long ld);
int simx_topk_update(simx_stream_t stream, int nq, int m, const float* scores, long ld, const int64_t* ids, long ld_ids,
                     int64_t id_base, int k, float* run_scores, int64_t* run_ids);
size_t simx_flat_ip_workspace_bytes(int nq, int chunk);
int simx_flat_ip_search(simx_stream_t stream, int nq, long nc, int H, const float* q, const float* corpus,
                        int64_t id_base, int k, int chunk, void* workspace, size_t workspace_bytes,
                        float* out_scores, int64_t* out_ids);

/* -------------------------------------------------------------- SimANS sampler
 * S1+S2 (SimANS/utils/MARCO_until_new.py:174-202, util_wiki.py:609-639, MARCO_until_Doc.py:110-148).
 * scores [nq,C] f64 candidate scores (rank order), pos_score [nq] f64.  form 0: exp(-|s-s+|*tau);
 * form 1: exp(-(s-s+ +b)^2*a).  pos_score == 0 -> last N candidates.  One wavefront per query,
 * rounds of N with-replacement inverse-CDF draws (Philox4x32-10 keyed by seed, counter
 * (query, round, draw/2, offset)), dedupe, zero the chosen weights, repeat until >= N.
 * neg_idx [nq,N] candidate indices (draw order), union_idx [nq,2N] / union_cnt [nq] the pre-truncation
 * union (may be NULL), weights_out [nq,C] f64 (may be NULL). */
int simx_simans_sample(simx_stream_t stream, int nq, int C, int N,
                       const double* scores, const double* pos_score,
                       int form, double a, double b, double tau,
                       uint64_t seed, uint32_t offset,
                       int32_t* neg_idx, int32_t* union_idx, int32_t* union_cnt, double* weights_out);

/* ------------------------------------------------------------------ optimiser
 * clip_grad_norm_ + transformers.AdamW + zero_grad (co_training_marco_train.py:57-69, 246-254;
 * update rule SURVEY App. C).  sqnorm: device scalar accumulator (zero it first). */
int simx_sqnorm_accum(simx_stream_t stream, const float* g, size_t n, float* sqnorm);
/* the same without float atomics (two launches; ws: SIMX_SQNORM_WS_FLOATS floats of scratch): bit-reproducible, so the
 * replicas of a data-parallel job compute the same clip coefficient from the same all-reduced gradients. */
#define SIMX_SQNORM_WS_FLOATS 2048
int simx_sqnorm_accum_det(simx_stream_t stream, const float* g, size_t n, float* sqnorm, float* ws);
/* p,m,v updated in place; g is read, scaled by min(1, max_norm/(sqrt(*sqnorm)+1e-6)) * grad_scale
 * (max_norm <= 0 or sqnorm == NULL: no clipping), and zeroed afterwards when zero_grad != 0. */
int simx_adamw_step(simx_stream_t stream, float* p, float* g, float* m, float* v, size_t n,
                    float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                    const float* sqnorm, float max_norm, float grad_scale, int zero_grad);

/* Dynamic loss scaler of the SIMX_F16 engine (apex.amp dynamic loss scaling, under which the reference's --fp16 mode runs:
 * co_training_marco_train.py:97-104, 218-220).  `state`: 8 floats on the device -- [0] S, [1] 1/S (what `gs` arguments point
 * to), [2] clean steps since S changed, [3] 1 = the current optimiser step is skipped, [4] steps applied, [5] steps skipped,
 * [6] growth interval, [7] largest S.  Per optimiser step: squared norm of ALL gradient buffers -> simx_scaler_update (inf /
 * nan: skip, S /= 2; else count, S *= 2 every `growth_interval` clean steps) -> simx_adamw_step_sc for every buffer, which
 * leaves p, m, v untouched on a skipped step (gradients are still zeroed) and takes its bias-correction step count from [4].
 * No host synchronisation anywhere. */
int simx_scaler_init(simx_stream_t stream, float* state, float init_scale, float growth_interval, float max_scale);
int simx_scaler_update(simx_stream_t stream, float* state, const float* sqnorm);
int simx_adamw_step_sc(simx_stream_t stream, float* p, float* g, float* m, float* v, size_t n,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       const float* sqnorm, float max_norm, float grad_scale, int zero_grad, const float* scaler);

/* ------------------------------------------------------------------ measurement
 * Optional per-launch timing with HIP events recorded on the launch stream (bench.py's live roofline
 * figure).  simx_prof_begin(max_launches) starts recording; simx_prof_end() stops, waits for the
 * events and fills host arrays of length simx_prof_kernel_count(): launches, total milliseconds and
 * total "work" (flops for GEMM/attention classes, algorithmic bytes for the HBM-bound classes).
 * Class ids: 0 gemm_nt, 1 gemm_tn, 2 mha_fwd, 3 mha_bwd, 4 ln_fwd, 5 ln_bwd, 6 embed_fwd, 7 embed_bwd,
 * 8 colsum, 9 cast, 10 loss, 11 sampler, 12 adamw(+norm), 13 other, 14 collate, 15 top-k, 16 gemm_nt: the persistent
 * 256x256 kernel (apart from class 0 = the small-shape launches), 17 gemm_tn: the 256x256 wgrad kernel with its slab pass
 * (apart from class 1 = the small-shape launches). */
int simx_prof_begin(int max_launches);
int simx_prof_end(int32_t* counts_host, double* total_ms_host, double* total_work_host);
int simx_prof_kernel_count(void);

/* 1 when the library runs with SIMX_DETERMINISTIC=1 (read once at first use): every cross-workgroup f32 reduction of the
 * backward (LayerNorm / bias / embedding gradients, loss scalars) is then done in a fixed order instead of with atomics,
 * so two runs of the same step on the same device give bit-identical gradients.  The reference offers the same through
 * torch.use_deterministic_algorithms; its scripts only seed the RNGs (SimANS/co_training/co_training_marco_train.py:33-44). */
int simx_deterministic(void);

#ifdef __cplusplus
}
#endif
#endif /* SIMX_H */
