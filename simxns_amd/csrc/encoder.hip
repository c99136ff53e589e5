// Native runtime: the whole-encoder forward / backward driver and library plumbing.
// Sequences the gfx950 kernels for HFBertEncoder.forward (SimANS/model/models.py:77-82 -> HF
// BertModel.forward, spec LEAD/modeling_bert.py:916-1038) and its backward on the packed token
// layout.  No allocation, no synchronisation: every buffer is carved from caller-provided memory.
//
// HBM layout (all 256-B aligned):
//   params / grads : one f32 buffer each, canonical order (simx_bert_param_offset)
//   wcache         : per layer  Wqkv, Wqkv^T, Wo, Wo^T, W1, W1^T, W2, W2^T   (bf16 / fp16; f32 mode keeps only
//                    the transposes, in f32) -- rebuilt once per optimiser step
//   act            : x0 | per layer { qkv[T,3H] ctx[T,H] lse[heads,T](f32) z1[T,H] x1[T,H] u[T,F] h[T,F]
//                    z2[T,H] xout[T,H] }   -- everything backward needs is KEPT (288 GB HBM: no recompute)
//   bwd scratch    : bufA[T,H] bufB[T,H] du[T,F] dqkv[T,3H] | split-K slabs of the wgrad GEMMs
#include <stdarg.h>
#include <string.h>
#include "common.h"

// ------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void simx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* simx_last_error(void) { return g_err; }
extern "C" int simx_version(void) { return 100; }

// ---- compute-CU budget of the persistent kernels (include/simx.h simx_set_compute_cus) --------------------------------
// The persistent GEMMs launch one workgroup per CU (each needs a whole CU's 160 KB of LDS) and give every workgroup a STATIC share
// of the tile list; the wgrad plans fill exactly one round of the chip.  When another kernel holds k CUs for the whole launch -- an
// RCCL ring during the overlapped backward -- k of those workgroups only start when a resident one has finished its WHOLE share:
// the launch takes up to twice as long (tools/cu_steal_bench, profiles/r06_cu_steal.json).  With a budget of ncu - k the launches
// fit beside the communication kernel and lose k / ncu instead.  0 = every CU (default; SIMX_COMPUTE_CUS overrides at first use).
static int g_compute_cus = -1;
extern "C" int simx_set_compute_cus(int n) {
  SIMX_REQUIRE(n >= 0, SIMX_ERR_BAD_SHAPE, "simx_set_compute_cus: %d", n);
  g_compute_cus = n;
  return SIMX_OK;
}
int simx_compute_cus(int device_cus) {
  if (g_compute_cus < 0) { const char* e = getenv("SIMX_COMPUTE_CUS"); g_compute_cus = e ? atoi(e) : 0; if (g_compute_cus < 0) g_compute_cus = 0; }
  int n = g_compute_cus > 0 && g_compute_cus < device_cus ? g_compute_cus : device_cus;
  n -= n % 8;                                   // whole XCD rows: xcd_remap assumes block b -> XCD b % 8
  return n < 8 ? 8 : n;
}

extern "C" int simx_transpose_cast(simx_stream_t stream, int out_dtype, const float* w, int rows, int cols, void* out, void* outT);

// ------------------------------------------------------------------------------------------ layouts
static inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }
static inline size_t esz(int dtype) { return dtype == SIMX_F32 ? 4 : 2; }

static bool cfg_ok(const simx_bert_cfg* c) {
  return c && simx_dtype_ok(c->dtype) && c->qkv_layout >= 0 && c->qkv_layout <= 1 && c->f32_gemm >= 0 && c->f32_gemm <= 1 && c->stream_lo >= 0 && c->stream_lo <= 1 && c->layers > 0 && c->hidden > 0 && c->heads > 0 &&
         c->hidden % c->heads == 0 && c->hidden % 4 == 0 && c->inter > 0 && c->inter % 4 == 0 && c->vocab > 0 &&
         c->max_pos > 0 && c->type_vocab > 0 && c->hidden <= 1024;
}

static size_t layer_param_count(const simx_bert_cfg* c) {
  const size_t H = c->hidden, F = c->inter;
  return 3 * H * H + 3 * H + H * H + H + 2 * H + F * H + F + H * F + H + 2 * H;
}

extern "C" size_t simx_bert_param_count(const simx_bert_cfg* c) {
  if (!cfg_ok(c)) return 0;
  const size_t H = c->hidden;
  return (size_t)c->vocab * H + (size_t)c->max_pos * H + (size_t)c->type_vocab * H + 2 * H +
         (size_t)c->layers * layer_param_count(c) + H * H + H;
}

extern "C" size_t simx_bert_param_offset(const simx_bert_cfg* c, int layer, int which) {
  if (!cfg_ok(c)) return (size_t)-1;
  const size_t H = c->hidden, F = c->inter;
  const size_t emb = (size_t)c->vocab * H + (size_t)c->max_pos * H + (size_t)c->type_vocab * H + 2 * H;
  if (layer == -1) {
    switch (which) {
      case SIMX_P_WORD: return 0;
      case SIMX_P_POS: return (size_t)c->vocab * H;
      case SIMX_P_TYPE: return (size_t)c->vocab * H + (size_t)c->max_pos * H;
      case SIMX_P_EMB_LN_G: return emb - 2 * H;
      case SIMX_P_EMB_LN_B: return emb - H;
      default: return (size_t)-1;
    }
  }
  if (layer >= 0 && layer < c->layers) {
    size_t o = emb + (size_t)layer * layer_param_count(c);
    const size_t sizes[12] = {3 * H * H, 3 * H, H * H, H, H, H, F * H, F, H * F, H, H, H};
    if (which < SIMX_P_WQKV || which > SIMX_P_LN2_B) return (size_t)-1;
    for (int i = 0; i < which - SIMX_P_WQKV; ++i) o += sizes[i];
    return o;
  }
  if (layer == c->layers) {
    const size_t o = emb + (size_t)c->layers * layer_param_count(c);
    if (which == SIMX_P_POOL_W) return o;
    if (which == SIMX_P_POOL_B) return o + H * H;
  }
  return (size_t)-1;
}

// "Operand planes" (csrc/gemm_xp.hip, simx.h): the fp32 engine's large towers run their dense GEMMs on pre-split 16-bit plane
// pairs.  Three predicates, each a function of its arguments only so that sizing calls, forward and backward agree:
//   pl_weights(c)            the weight cache also holds fp16 planes of W and bf16 planes of W^T
//   pl_layout(c, Tp)         activation / scratch buffers have room for the planes of the residual stream (x0, x1, xout)
//   pl_run(c, Tp, max_len)   the tower's full layers run on planes (needs the f32 MFMA attention: head size 64, sequences <= 4096)
// SIMX_F32_PLANES=0 pins the register-split kernels of gemm_x3.hip (A/B measurements); SIMX_F32_PLANES_MIN_TILES lowers the
// size threshold (tests run small towers through the plane kernels).
static bool pl_weights(const simx_bert_cfg* c) {
  static const bool on = [] { const char* e = getenv("SIMX_F32_PLANES"); return !e || e[0] != '0'; }();
  return on && c->dtype == SIMX_F32 && c->f32_gemm == 0 && c->hidden % 256 == 0 && c->inter % 256 == 0 && c->hidden / c->heads == 64;
}
static bool pl_layout(const simx_bert_cfg* c, size_t Tp) {
  static const long min_tiles = [] { const char* e = getenv("SIMX_F32_PLANES_MIN_TILES"); return e ? atol(e) : 192L; }();
  return pl_weights(c) && (long)(Tp / 256) * (c->hidden / 256) >= min_tiles;
}
static bool pl_run(const simx_bert_cfg* c, size_t Tp, int max_len) { return pl_layout(c, Tp) && simx_mha_planes_ok(c->hidden / c->heads, max_len); }

// wqkvP .. w2P: fp16 plane pairs of W [out, in] (forward operand); wqkvTP .. w2TP: bf16 plane pairs of W^T [in, out] (dgrad)
struct WLayer { const char *wqkv, *wqkvT, *wo, *woT, *w1, *w1T, *w2, *w2T, *wqkvP, *woP, *w1P, *w2P, *wqkvTP, *woTP, *w1TP, *w2TP; };
static size_t wcache_layer_bytes(const simx_bert_cfg* c) {
  const size_t H = c->hidden, F = c->inter, e = esz(c->dtype);
  const size_t one = al(3 * H * H * e) + al(H * H * e) + 2 * al(F * H * e);
  if (pl_weights(c)) return 3 * one;              // f32 transposes | fp16 planes of W | bf16 planes of W^T (4 B per element each)
  return c->dtype == SIMX_F32 ? one : 2 * one;
}
extern "C" size_t simx_bert_wcache_bytes(const simx_bert_cfg* c) {
  return cfg_ok(c) ? (size_t)c->layers * wcache_layer_bytes(c) : 0;
}
static WLayer wlayer(const simx_bert_cfg* c, const float* params, const void* wcache, int l) {
  const size_t H = c->hidden, F = c->inter, e = esz(c->dtype);
  const char* b = (const char*)wcache + (size_t)l * wcache_layer_bytes(c);
  WLayer w = {};
  if (c->dtype == SIMX_F32) {
    w.wqkv = (const char*)(params + simx_bert_param_offset(c, l, SIMX_P_WQKV));
    w.wo = (const char*)(params + simx_bert_param_offset(c, l, SIMX_P_WO));
    w.w1 = (const char*)(params + simx_bert_param_offset(c, l, SIMX_P_W1));
    w.w2 = (const char*)(params + simx_bert_param_offset(c, l, SIMX_P_W2));
    w.wqkvT = b; b += al(3 * H * H * e);
    w.woT = b; b += al(H * H * e);
    w.w1T = b; b += al(F * H * e);
    w.w2T = b; b += al(F * H * e);
    if (pl_weights(c)) {
      w.wqkvP = b; b += al(3 * H * H * e);
      w.woP = b; b += al(H * H * e);
      w.w1P = b; b += al(F * H * e);
      w.w2P = b; b += al(F * H * e);
      w.wqkvTP = b; b += al(3 * H * H * e);
      w.woTP = b; b += al(H * H * e);
      w.w1TP = b; b += al(F * H * e);
      w.w2TP = b;
    }
  } else {
    w.wqkv = b; b += al(3 * H * H * e);
    w.wqkvT = b; b += al(3 * H * H * e);
    w.wo = b; b += al(H * H * e);
    w.woT = b; b += al(H * H * e);
    w.w1 = b; b += al(F * H * e);
    w.w1T = b; b += al(F * H * e);
    w.w2 = b; b += al(F * H * e);
    w.w2T = b;
  }
  return w;
}

// Every token-major buffer is carved with its row count rounded up to a multiple of 256, and the NT GEMMs run on the
// padded row count: the persistent full-tile kernels then also serve ragged (packed, variable-length) batches.  Rows
// [T, Tp) hold garbage that never leaves its own row (an NT GEMM row depends on that row of A only; LayerNorm,
// attention, the wgrad contraction and every reduction run over the T real rows).
static inline int rows_cap(int T) { return (T + 255) & ~255; }
// stream_lo (simx.h): the residual stream -- x0 and every LayerNorm output -- is a 16-bit tensor PLUS a one-byte correction
// per element (x1l / xoutl, NULL otherwise); z1 / z2 then hold the dense outputs without the residual, which the LayerNorm
// kernels add
static inline bool stream_lo(const simx_bert_cfg* c) { return c->stream_lo != 0 && simx_is16(c->dtype); }
// x1p / xoutp (operand planes, pl_layout): the fp16 plane pairs of x1 / xout beside their f32 forms; in a layer that RUNS on planes
// the ctx and h slots hold plane pairs instead of f32 (same bytes: two 16-bit planes)
struct ALayer { char *qkv, *ctx, *z1, *x1, *u, *h, *z2, *xout, *x1l, *xoutl, *x1p, *xoutp; float* lse; };
static size_t act_layer_bytes(const simx_bert_cfg* c, size_t T) {
  const size_t H = c->hidden, F = c->inter, e = esz(c->dtype);
  return al(T * 3 * H * e) + 5 * al(T * H * e) + (stream_lo(c) ? 2 * al(T * H) : 0) + (pl_layout(c, T) ? 2 * al(T * H * 4) : 0) + 2 * al(T * F * e) +
         al((size_t)c->heads * T * 4);
}
// a [Tp,H] tensor of the residual stream: the 16-bit values, followed by the corrections (ONE BYTE per element: common.h
// lo8) when stream_lo
static size_t xs_bytes(const simx_bert_cfg* c, size_t Tp) {
  return al(Tp * c->hidden * esz(c->dtype)) + (stream_lo(c) ? al(Tp * c->hidden) : 0) + (pl_layout(c, Tp) ? al(Tp * c->hidden * 4) : 0);
}
static const char* xs_lo(const simx_bert_cfg* c, const char* hi, size_t Tp) { return stream_lo(c) ? hi + al(Tp * c->hidden * esz(c->dtype)) : nullptr; }
// the plane pair that follows an f32 [Tp,H] tensor of the residual stream (NULL without pl_layout)
static const char* xs_pl(const simx_bert_cfg* c, const char* x, size_t Tp) { return pl_layout(c, Tp) ? x + al(Tp * c->hidden * 4) : nullptr; }
// Activation memory, three modes:
//   save = 0                      : x0 | 2 layer slots used as a ring                                  (inference)
//   save = 1, grad_checkpoint = 0 : x0 | L layer slots -- everything backward needs is kept            (default)
//   save = 1, grad_checkpoint = 1 : x0 | L layer OUTPUTS [Tp,H] | 2 layer slots (ring) -- only every layer's input is kept
//                                   (torch.utils.checkpoint per BertLayer, SimANS/model/models.py:73-74); backward re-runs
//                                   a layer's forward into ring slot 0 (same stateless dropout masks) before its backward
// followed by three [nseq,H] tensors of the [CLS]-only last layer (q, attention context, a temporary).
static inline bool ckpt_mode(const simx_bert_cfg* c, int save) { return save && c->grad_checkpoint != 0; }
static size_t act_slots_bytes(const simx_bert_cfg* c, size_t Tp, int save) {
  if (ckpt_mode(c, save)) return (size_t)c->layers * xs_bytes(c, Tp) + 2 * act_layer_bytes(c, Tp);
  return (size_t)(save ? c->layers : 2) * act_layer_bytes(c, Tp);
}
extern "C" size_t simx_bert_act_bytes(const simx_bert_cfg* c, int T, int nseq, int save_for_bwd) {
  if (!cfg_ok(c) || T <= 0) return 0;
  const size_t Tp = (size_t)rows_cap(T);
  const size_t x0 = xs_bytes(c, Tp);
  const size_t n = nseq > 0 ? (size_t)nseq : Tp;
  // (+ the [CLS]-only last layer's compact tensors: q, attention context, a temporary, and -- stream_lo -- the gathered
  // residual rows with their correction bytes)
  return x0 + act_slots_bytes(c, Tp, save_for_bwd) + 5 * al(n * c->hidden * esz(c->dtype));
}
static char* act_extra(const simx_bert_cfg* c, void* act, size_t Tp, int save) {
  return (char*)act + xs_bytes(c, Tp) + act_slots_bytes(c, Tp, save);
}
static char* act_x0(void* act) { return (char*)act; }
static ALayer carve_layer(const simx_bert_cfg* c, char* b, size_t T) {
  const size_t H = c->hidden, F = c->inter, e = esz(c->dtype);
  ALayer a;
  a.qkv = b; b += al(T * 3 * H * e);
  a.ctx = b; b += al(T * H * e);
  a.z1 = b; b += al(T * H * e);
  a.x1 = b; b += al(T * H * e);
  a.z2 = b; b += al(T * H * e);
  a.xout = b; b += al(T * H * e);
  a.x1l = a.xoutl = a.x1p = a.xoutp = nullptr;
  if (stream_lo(c)) { a.x1l = b; b += al(T * H); a.xoutl = b; b += al(T * H); }
  if (pl_layout(c, T)) { a.x1p = b; b += al(T * H * 4); a.xoutp = b; b += al(T * H * 4); }
  a.u = b; b += al(T * F * e);                 // gelu'(u) of the FFN pre-activation (SIMX_EPI_GELU writes it; only DGELU reads it)
  a.h = b; b += al(T * F * e);
  a.lse = (float*)b;
  return a;
}
// the slot layer l's FORWARD writes (checkpoint mode: ring slot l&1, its output redirected to the kept array)
static ALayer alayer(const simx_bert_cfg* c, void* act, size_t T, int l, int save) {
  const size_t xb = xs_bytes(c, T);
  char* base = (char*)act + xb;
  if (ckpt_mode(c, save)) {
    ALayer a = carve_layer(c, base + (size_t)c->layers * xb + (size_t)(l & 1) * act_layer_bytes(c, T), T);
    a.xout = base + (size_t)l * xb;
    a.xoutl = const_cast<char*>(xs_lo(c, a.xout, T));
    a.xoutp = const_cast<char*>(xs_pl(c, a.xout, T));
    return a;
  }
  return carve_layer(c, base + (size_t)(save ? l : (l & 1)) * act_layer_bytes(c, T), T);
}
// checkpoint mode, backward: the slot layer l is RE-COMPUTED into (ring slot 0, its own xout)
static ALayer alayer_recompute(const simx_bert_cfg* c, void* act, size_t T) {
  const size_t xb = xs_bytes(c, T);
  return carve_layer(c, (char*)act + xb + (size_t)c->layers * xb, T);
}
// input of layer l (= output of layer l-1, or the embedding output)
static const char* layer_input(const simx_bert_cfg* c, const void* act, size_t T, int l) {
  if (l == 0) return act_x0(const_cast<void*>(act));
  return alayer(c, const_cast<void*>(act), T, l - 1, 1).xout;
}
// its fp16 plane pair (NULL without pl_layout)
static const char* layer_input_pl(const simx_bert_cfg* c, const void* act, size_t T, int l) {
  if (l == 0) return xs_pl(c, act_x0(const_cast<void*>(act)), T);
  return alayer(c, const_cast<void*>(act), T, l - 1, 1).xoutp;
}
// its stream correction (NULL without stream_lo)
static const char* layer_input_lo(const simx_bert_cfg* c, const void* act, size_t T, int l) {
  if (l == 0) return xs_lo(c, act_x0(const_cast<void*>(act)), T);
  return alayer(c, const_cast<void*>(act), T, l - 1, 1).xoutl;
}

static size_t tn_ws_max(const simx_bert_cfg* c, int T) {
  const int H = c->hidden, F = c->inter;
  size_t m = simx_gemm_tn_workspace_bytes(3 * H, H, T);
  size_t v = simx_gemm_tn_workspace_bytes(H, H, T); if (v > m) m = v;
  v = simx_gemm_tn_workspace_bytes(F, H, T); if (v > m) m = v;
  v = simx_gemm_tn_workspace_bytes(H, F, T); if (v > m) m = v;
  if (pl_weights(c)) {
    v = simx_gemm_tn_planes_workspace_bytes(3 * H, H, T); if (v > m) m = v;
    v = simx_gemm_tn_planes_workspace_bytes(F, H, T); if (v > m) m = v;
    v = simx_gemm_tn_planes_workspace_bytes(H, F, T); if (v > m) m = v;
  }
  return al(m);
}
extern "C" size_t simx_bert_bwd_scratch_bytes(const simx_bert_cfg* c, int T, int nseq) {
  if (!cfg_ok(c) || T <= 0) return 0;
  const size_t H = c->hidden, F = c->inter, e = esz(c->dtype);
  const size_t Tp = (size_t)rows_cap(T);
  const size_t n = nseq > 0 ? (size_t)nseq : Tp;
  // (+ operand planes: the bf16 plane pair of the activation a wgrad GEMM contracts against, converted per use)
  return 3 * al(Tp * H * e) + al(Tp * F * e) + al(Tp * 3 * H * e) + tn_ws_max(c, T) + 3 * al(n * H * e) + (pl_layout(c, Tp) ? al(Tp * F * 4) : 0);
}

// ------------------------------------------------------------------------------------------ driver
#define RUN(call)            \
  do {                       \
    int rc__ = (call);       \
    if (rc__) return rc__;   \
  } while (0)

extern "C" int simx_bert_cast_weights(simx_stream_t stream, const simx_bert_cfg* c, const float* params, void* wcache) {
  SIMX_REQUIRE(cfg_ok(c), SIMX_ERR_BAD_SHAPE, "bert_cast_weights: bad config");
  SIMX_REQUIRE(params && wcache, SIMX_ERR_BAD_SHAPE, "bert_cast_weights: NULL buffer");
  const int H = c->hidden, F = c->inter;
  const bool f32 = c->dtype == SIMX_F32;
  if (pl_weights(c)) {                            // f32 transposes + both plane pairs of every dense weight, grouped launches
    static thread_local SimxSplitGroup sg;
    sg.n = 0;
    auto addp = [&](const float* w, int rows, int cols, const char* wT, const char* ph, const char* ptb) -> int {
      if (sg.n == SIMX_SPLIT_GROUP_MAX) { RUN(simx_split_weight_group((hipStream_t)stream, &sg)); sg.n = 0; }
      sg.job[sg.n++] = SimxSplitJob{w, (void*)ph, (void*)ptb, (float*)wT, rows, cols, 0, 0};
      return SIMX_OK;
    };
    for (int l = 0; l < c->layers; ++l) {
      const WLayer w = wlayer(c, params, wcache, l);
      RUN(addp(params + simx_bert_param_offset(c, l, SIMX_P_WQKV), 3 * H, H, w.wqkvT, w.wqkvP, w.wqkvTP));
      RUN(addp(params + simx_bert_param_offset(c, l, SIMX_P_WO), H, H, w.woT, w.woP, w.woTP));
      RUN(addp(params + simx_bert_param_offset(c, l, SIMX_P_W1), F, H, w.w1T, w.w1P, w.w1TP));
      RUN(addp(params + simx_bert_param_offset(c, l, SIMX_P_W2), H, F, w.w2T, w.w2P, w.w2TP));
    }
    if (sg.n) RUN(simx_split_weight_group((hipStream_t)stream, &sg));
    return SIMX_OK;
  }
  static thread_local SimxCastGroup g;              // (built in place, passed by value at the launch)
  g.n = 0;
  auto add = [&](const float* w, int rows, int cols, const char* out, const char* outT) -> int {
    if (g.n == SIMX_CAST_GROUP_MAX) { RUN(simx_transpose_cast_group((hipStream_t)stream, c->dtype, &g)); g.n = 0; }
    g.job[g.n++] = SimxCastJob{w, f32 ? nullptr : (void*)out, (void*)outT, rows, cols, 0, 0};
    return SIMX_OK;
  };
  for (int l = 0; l < c->layers; ++l) {
    const WLayer w = wlayer(c, params, wcache, l);
    RUN(add(params + simx_bert_param_offset(c, l, SIMX_P_WQKV), 3 * H, H, w.wqkv, w.wqkvT));
    RUN(add(params + simx_bert_param_offset(c, l, SIMX_P_WO), H, H, w.wo, w.woT));
    RUN(add(params + simx_bert_param_offset(c, l, SIMX_P_W1), F, H, w.w1, w.w1T));
    RUN(add(params + simx_bert_param_offset(c, l, SIMX_P_W2), H, F, w.w2, w.w2T));
  }
  if (g.n) RUN(simx_transpose_cast_group((hipStream_t)stream, c->dtype, &g));
  return SIMX_OK;
}

// dropout descriptor of (layer, site); layer -1 = embeddings
static simx_dropout drop_of(const simx_bert_cfg* c, int layer, int site) {
  simx_dropout d;
  d.p = site == 3 ? c->attn_dropout : c->hidden_dropout;
  d.seed = c->dropout_seed;
  d.stream = (uint32_t)((layer + 1) * 8 + site);
  return d;
}

static int check_io(const simx_bert_cfg* c, int nseq, int T, int max_len, const char* who) {
  SIMX_REQUIRE(cfg_ok(c), SIMX_ERR_BAD_SHAPE, "%s: bad config", who);
  SIMX_REQUIRE(c->hidden_dropout >= 0.f && c->hidden_dropout < 1.f && c->attn_dropout >= 0.f && c->attn_dropout < 1.f,
               SIMX_ERR_BAD_SHAPE, "%s: dropout probabilities must be in [0,1)", who);
  SIMX_REQUIRE(nseq > 0 && T >= nseq && max_len > 0 && max_len <= c->max_pos, SIMX_ERR_BAD_SHAPE,
               "%s: bad batch (nseq=%d T=%d max_len=%d max_pos=%d)", who, nseq, T, max_len, c->max_pos);
  return SIMX_OK;
}

// Head-major q/k/v (include/simx.h "head-major q / k / v"): chosen per tower when every kernel that touches the tensor has the
// form -- a 16-bit dtype, head size 64, sequences that the LDS-resident attention backward takes (<= 256), and a tower large
// enough for the persistent GEMMs (simx_gemm_hm_ok).  cfg->qkv_layout = 1 pins the token-major form (A/B measurements, tests);
// the choice is a function of (cfg, T, max_len) only, so a forward and its backward -- which receive the same per-call
// config copy -- always agree.
static int hm_rows_for(const simx_bert_cfg* c, int T, int Tp, int max_len) {
  if (c->qkv_layout == 1) return 0;
  const int d = c->hidden / c->heads;
  if (!simx_is16(c->dtype) || d != 64 || max_len > 256) return 0;
  return simx_gemm_hm_ok(Tp, c->hidden, T) ? Tp : 0;
}

// One encoder layer's forward: x_in [Tp,H] -> a.xout (a = the slot it writes).  `keep`: backward will read this slot
// (the pre-activation u is stored); the [CLS]-only form of the last layer writes [nseq, .] tensors into the slot's usual
// buffers and q / attention context of the [CLS] rows into `extra` (kept for backward).
static int layer_fwd(hipStream_t stream, const simx_bert_cfg* c, const float* params, const void* wcache, int l, const char* x,
                     const char* xl, const char* xp, const ALayer& a, char* extra, const int32_t* cu, int nseq, int T, int Tp, int max_len, int keep,
                     bool cls_form, float* cls_out) {
  const int H = c->hidden, F = c->inter, dt = c->dtype, d = H / c->heads;
  auto off = [&](int ll, int w) { return params + simx_bert_param_offset(c, ll, w); };
  const WLayer w = wlayer(c, params, wcache, l);
  const simx_dropout d1 = drop_of(c, l, 1), d2 = drop_of(c, l, 2), d3 = drop_of(c, l, 3);
  const int hm = hm_rows_for(c, T, Tp, max_len);
  // fp32 engine: the dense GEMMs run on the 16-bit matrix cores from fp16 hi + lo splits of their f32 operands unless the
  // config asks for exact f32 products (simx.h SIMX_F32_SPLIT_H; every other kernel of the layer is plain f32)
  const int gdt = (dt == SIMX_F32 && c->f32_gemm == 0) ? SIMX_F32_SPLIT_H : dt;
  if (cls_form) {
    // Last layer, [CLS] rows only: row s of every [nseq, .] buffer below is the sequence's token 0 (row cu[s] of the
    // full tensors).  Only K and V are projected for every token; Q, the attention core and everything after it run
    // for the one query per sequence that is read.  Same arithmetic as the full path -- dropout masks stay keyed by
    // the ORIGINAL row index.  z1, x1, u, h, z2, xout of this layer hold [nseq, .] tensors in their usual slots.
    const size_t e = esz(dt);
    char* qc = extra;                                            // kept for backward
    char* ctxc = qc + al((size_t)nseq * H * e);                  // kept for backward
    char* ytmp = ctxc + al((size_t)nseq * H * e);
    if (hm)      // K, V planes [heads, 3*heads) of the head-major tensor
      RUN(simx_gemm_nt_hm(stream, dt, Tp, 2 * H, H, x, H, w.wqkv + (size_t)H * H * e, H, a.qkv + (size_t)c->heads * hm * 64 * e, 64,
                          off(l, SIMX_P_BQKV) + H, nullptr, 0, nullptr, 0, hm));
    else
      RUN(simx_gemm_nt(stream, gdt, Tp, 2 * H, H, x, H, w.wqkv + (size_t)H * H * e, H, a.qkv + (size_t)H * e, 3 * H,
                       off(l, SIMX_P_BQKV) + H, nullptr, 0, SIMX_EPI_NONE, nullptr, 0, nullptr, 0));
    RUN(simx_rows_copy(stream, dt, dt, nseq, H, cu, nullptr, x, ytmp));
    RUN(simx_gemm_nt(stream, gdt, nseq, H, H, ytmp, H, w.wqkv, H, qc, H, off(l, SIMX_P_BQKV), nullptr, 0, SIMX_EPI_NONE, nullptr, 0,
                     nullptr, 0));
    RUN(simx_mha_cls_fwd_hm(stream, dt, nseq, c->heads, d, cu, max_len, T, qc, a.qkv, ctxc, &d3, hm));
    RUN(simx_gemm_nt(stream, gdt, nseq, H, H, ctxc, H, w.wo, H, ytmp, H, off(l, SIMX_P_BO), nullptr, 0, SIMX_EPI_NONE, nullptr, 0,
                     nullptr, 0));
    if (stream_lo(c)) {
      // residual stream with corrections: the [CLS] rows of the layer input (16-bit values + correction bytes) are gathered
      // behind the compact q / context / temporary tensors (kept for backward), the LayerNorm kernels add them in f32, and the embeddings
      // leave as f32(hi + correction) -- never rounded to 16 bits
      char* xg = ytmp + al((size_t)nseq * H * e);
      char* xgl = xg + al((size_t)nseq * H * e);
      RUN(simx_stream_rows(stream, dt, nseq, H, cu, x, xl, xg, xgl, nullptr));
      RUN(simx_drop_residual_rows(stream, dt, nseq, H, ytmp, nullptr, nullptr, cu, &d1, a.z1));
      RUN(simx_ln_fwd_res(stream, dt, nseq, H, a.z1, xg, xgl, off(l, SIMX_P_LN1_G), off(l, SIMX_P_LN1_B), c->eps, a.x1, a.x1l));
      RUN(simx_gemm_nt(stream, gdt, nseq, F, H, a.x1, H, w.w1, H, a.u, F, off(l, SIMX_P_B1), nullptr, 0, SIMX_EPI_GELU, nullptr, 0, a.h, F));
      RUN(simx_gemm_nt(stream, gdt, nseq, H, F, a.h, F, w.w2, F, ytmp, H, off(l, SIMX_P_B2), nullptr, 0, SIMX_EPI_NONE, nullptr, 0,
                       nullptr, 0));
      RUN(simx_drop_residual_rows(stream, dt, nseq, H, ytmp, nullptr, nullptr, cu, &d2, a.z2));
      RUN(simx_ln_fwd_res(stream, dt, nseq, H, a.z2, a.x1, a.x1l, off(l, SIMX_P_LN2_G), off(l, SIMX_P_LN2_B), c->eps, a.xout, a.xoutl));
      if (cls_out) RUN(simx_stream_rows(stream, dt, nseq, H, nullptr, a.xout, a.xoutl, nullptr, nullptr, cls_out));
      return SIMX_OK;
    }
    RUN(simx_drop_residual_rows(stream, dt, nseq, H, ytmp, x, cu, cu, &d1, a.z1));
    RUN(simx_ln_fwd(stream, dt, nseq, H, a.z1, off(l, SIMX_P_LN1_G), off(l, SIMX_P_LN1_B), c->eps, a.x1));
    RUN(simx_gemm_nt(stream, gdt, nseq, F, H, a.x1, H, w.w1, H, a.u, F, off(l, SIMX_P_B1), nullptr, 0, SIMX_EPI_GELU, nullptr, 0, a.h, F));
    RUN(simx_gemm_nt(stream, gdt, nseq, H, F, a.h, F, w.w2, F, ytmp, H, off(l, SIMX_P_B2), nullptr, 0, SIMX_EPI_NONE, nullptr, 0,
                     nullptr, 0));
    RUN(simx_drop_residual_rows(stream, dt, nseq, H, ytmp, a.x1, nullptr, cu, &d2, a.z2));
    RUN(simx_ln_fwd(stream, dt, nseq, H, a.z2, off(l, SIMX_P_LN2_G), off(l, SIMX_P_LN2_B), c->eps, a.xout));
    if (cls_out) RUN(simx_rows_copy(stream, dt, SIMX_F32, nseq, H, nullptr, nullptr, a.xout, cls_out));
    return SIMX_OK;
  }
  if (pl_run(c, Tp, max_len)) {
    // fp32 engine on operand planes: every GEMM operand is a 16-bit plane pair written by its producer -- xp by the previous
    // LayerNorm, ctx by the attention kernel, x1p by LayerNorm, h by the FFN-in epilogue; f32 forms exist where an
    // elementwise consumer needs them (residuals x / x1, q/k/v, the pre-LayerNorm sums, gelu')
    const long psH = (long)Tp * H, psF = (long)Tp * F;
    if (simx_mha_x3_ok(d, max_len)) {
      // q / k / v leave the projection as an fp16 plane pair (no f32 form) and the attention products run on the 16-bit
      // matrix cores from pairs (csrc/attention_x3.hip); longer sequences keep f32 q / k / v and the chunked f32 MFMA kernels
      RUN(simx_gemm_nt_planes(stream, SIMX_F16, SIMX_EPI_NONE_PLANES, Tp, 3 * H, H, xp, H, psH, w.wqkvP, H, 3L * H * H, nullptr, 3 * H,
                              off(l, SIMX_P_BQKV), nullptr, 0, a.qkv, 3 * H, (long)Tp * 3 * H, nullptr));
      RUN(simx_mha_fwd_x3(stream, nseq, c->heads, d, cu, max_len, T, a.qkv, (long)Tp * 3 * H, a.ctx, psH, a.lse, &d3));
    } else {
      RUN(simx_gemm_nt_planes(stream, SIMX_F16, SIMX_EPI_NONE, Tp, 3 * H, H, xp, H, psH, w.wqkvP, H, 3L * H * H, (float*)a.qkv, 3 * H,
                              off(l, SIMX_P_BQKV), nullptr, 0, nullptr, 0, 0, nullptr));
      RUN(simx_mha_fwd_planes(stream, nseq, c->heads, d, cu, max_len, T, (const float*)a.qkv, a.ctx, psH, a.lse, &d3));
    }
    RUN(simx_gemm_nt_planes(stream, SIMX_F16, SIMX_EPI_NONE, Tp, H, H, a.ctx, H, psH, w.woP, H, (long)H * H, (float*)a.z1, H, off(l, SIMX_P_BO),
                            (const float*)x, H, nullptr, 0, 0, &d1));
    RUN(simx_ln_fwd_planes(stream, T, H, (const float*)a.z1, off(l, SIMX_P_LN1_G), off(l, SIMX_P_LN1_B), c->eps, (float*)a.x1, a.x1p, psH));
    RUN(simx_gemm_nt_planes(stream, SIMX_F16, keep ? SIMX_EPI_GELU : SIMX_EPI_GELU_INFER, Tp, F, H, a.x1p, H, psH, w.w1P, H, (long)F * H,
                            (float*)a.u, F, off(l, SIMX_P_B1), nullptr, 0, a.h, F, psF, nullptr));
    RUN(simx_gemm_nt_planes(stream, SIMX_F16, SIMX_EPI_NONE, Tp, H, F, a.h, F, psF, w.w2P, F, (long)H * F, (float*)a.z2, H, off(l, SIMX_P_B2),
                            (const float*)a.x1, H, nullptr, 0, 0, &d2));
    RUN(simx_ln_fwd_planes(stream, T, H, (const float*)a.z2, off(l, SIMX_P_LN2_G), off(l, SIMX_P_LN2_B), c->eps, (float*)a.xout, a.xoutp, psH));
    return SIMX_OK;
  }
  if (hm)
    RUN(simx_gemm_nt_hm(stream, dt, Tp, 3 * H, H, x, H, w.wqkv, H, a.qkv, 64, off(l, SIMX_P_BQKV), nullptr, 0, nullptr, 0, hm));
  else
    RUN(simx_gemm_nt(stream, gdt, Tp, 3 * H, H, x, H, w.wqkv, H, a.qkv, 3 * H, off(l, SIMX_P_BQKV), nullptr, 0, SIMX_EPI_NONE,
                     nullptr, 0, nullptr, 0));
  RUN(simx_mha_fwd_hm(stream, dt, nseq, c->heads, d, cu, max_len, T, a.qkv, a.ctx, a.lse, &d3, hm));
  // stream_lo: the dense output leaves without the residual (a plain bias + dropout epilogue) and the LayerNorm kernel sums
  // dense + x_hi + x_lo in f32; otherwise the residual rides in the GEMM epilogue and z1 is the LayerNorm input itself
  const bool sl = stream_lo(c);
  RUN(simx_gemm_nt_ex(stream, gdt, Tp, H, H, a.ctx, H, w.wo, H, a.z1, H, off(l, SIMX_P_BO), sl ? nullptr : x, H, SIMX_EPI_NONE, nullptr, 0,
                      nullptr, 0, &d1));
  RUN(simx_ln_fwd_res(stream, dt, T, H, a.z1, sl ? x : nullptr, sl ? xl : nullptr, off(l, SIMX_P_LN1_G), off(l, SIMX_P_LN1_B), c->eps, a.x1,
                      a.x1l));
  RUN(simx_gemm_nt(stream, gdt, Tp, F, H, a.x1, H, w.w1, H, a.u, F, off(l, SIMX_P_B1), nullptr, 0,
                   keep ? SIMX_EPI_GELU : SIMX_EPI_GELU_INFER, nullptr, 0, a.h, F));   // no backward from this slot: u has no reader
  RUN(simx_gemm_nt_ex(stream, gdt, Tp, H, F, a.h, F, w.w2, F, a.z2, H, off(l, SIMX_P_B2), sl ? nullptr : a.x1, H, SIMX_EPI_NONE, nullptr, 0,
                      nullptr, 0, &d2));
  RUN(simx_ln_fwd_res(stream, dt, T, H, a.z2, sl ? a.x1 : nullptr, sl ? a.x1l : nullptr, off(l, SIMX_P_LN2_G), off(l, SIMX_P_LN2_B), c->eps,
                      a.xout, a.xoutl));
  return SIMX_OK;
}

extern "C" int simx_bert_fwd(simx_stream_t stream, const simx_bert_cfg* c, const float* params, const void* wcache,
                             const int32_t* ids, const int32_t* pos_ids, const int32_t* cu, int nseq, int T, int max_len,
                             void* act, size_t act_bytes, int save, float* cls_out, void* hidden_out) {
  RUN(check_io(c, nseq, T, max_len, "bert_fwd"));
  SIMX_REQUIRE(params && wcache && ids && pos_ids && cu && act, SIMX_ERR_BAD_SHAPE, "bert_fwd: NULL buffer");
  SIMX_REQUIRE(act_bytes >= simx_bert_act_bytes(c, T, nseq, save), SIMX_ERR_WORKSPACE, "bert_fwd: activation buffer %zu < %zu",
               act_bytes, simx_bert_act_bytes(c, T, nseq, save));
  const int H = c->hidden, dt = c->dtype;
  const int Tp = rows_cap(T);
  auto off = [&](int l, int w) { return params + simx_bert_param_offset(c, l, w); };
  const char* x = act_x0(act);
  const char* xl = xs_lo(c, x, Tp);
  const char* xp = xs_pl(c, x, Tp);
  {
    const simx_dropout d0 = drop_of(c, -1, 0);
    if (xp)
      RUN(simx_embed_ln_fwd_planes(stream, T, H, ids, pos_ids, off(-1, SIMX_P_WORD), off(-1, SIMX_P_POS), off(-1, SIMX_P_TYPE),
                                   off(-1, SIMX_P_EMB_LN_G), off(-1, SIMX_P_EMB_LN_B), c->eps, (float*)act_x0(act), const_cast<char*>(xp),
                                   (long)Tp * H, &d0));
    else
      RUN(simx_embed_ln_fwd_lo(stream, dt, T, H, ids, pos_ids, off(-1, SIMX_P_WORD), off(-1, SIMX_P_POS), off(-1, SIMX_P_TYPE),
                               off(-1, SIMX_P_EMB_LN_G), off(-1, SIMX_P_EMB_LN_B), c->eps, act_x0(act), const_cast<char*>(xl), &d0));
  }
  const bool cls_only = c->cls_only_last_layer != 0;
  const bool ckpt = ckpt_mode(c, save);
  SIMX_REQUIRE(!(cls_only && hidden_out), SIMX_ERR_BAD_SHAPE, "bert_fwd: cls_only_last_layer leaves no full hidden state to return");
  for (int l = 0; l < c->layers; ++l) {
    const ALayer a = alayer(c, act, Tp, l, save);
    const bool cls_form = cls_only && l == c->layers - 1;
    // (checkpoint mode: the slot is recomputed by backward, so the forward runs it in its inference form)
    RUN(layer_fwd((hipStream_t)stream, c, params, wcache, l, x, xl, xp, a, act_extra(c, act, Tp, save), cu, nseq, T, Tp, max_len,
                  save && !ckpt, cls_form, cls_out));
    if (cls_form) return SIMX_OK;
    x = a.xout;
    xl = a.xoutl;
    xp = a.xoutp;
  }
  if (cls_out) {
    // (a stream_lo tower: the embeddings are f32(hi + correction byte), as the [CLS]-only last layer returns them -- the same
    // precision whichever form the last layer ran in)
    if (stream_lo(c) && xl) RUN(simx_stream_rows(stream, dt, nseq, H, cu, x, xl, nullptr, nullptr, cls_out));
    else RUN(simx_cls_gather(stream, dt, nseq, H, cu, x, cls_out));
  }
  if (hidden_out) {
    if (hipMemcpyAsync(hidden_out, x, (size_t)T * H * esz(dt), hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
      simx_set_error("bert_fwd: copy of last hidden state failed");
      return SIMX_ERR_HIP;
    }
  }
  return SIMX_OK;
}

extern "C" int simx_bert_bwd(simx_stream_t stream, const simx_bert_cfg* c, const float* params, const void* wcache,
                             const int32_t* ids, const int32_t* pos_ids, const int32_t* cu, int nseq, int T, int max_len,
                             const void* act, size_t act_bytes, const float* dcls, float* grads, void* scratch,
                             size_t scratch_bytes) {
  return simx_bert_bwd_ex(stream, c, params, wcache, ids, pos_ids, cu, nseq, T, max_len, act, act_bytes, dcls, nullptr, grads, scratch,
                          scratch_bytes);
}

extern "C" int simx_bert_bwd_ex(simx_stream_t stream, const simx_bert_cfg* c, const float* params, const void* wcache,
                                const int32_t* ids, const int32_t* pos_ids, const int32_t* cu, int nseq, int T, int max_len,
                                const void* act, size_t act_bytes, const float* dcls, const void* dhidden, float* grads,
                                void* scratch, size_t scratch_bytes) {
  return simx_bert_bwd_range(stream, c, params, wcache, ids, pos_ids, cu, nseq, T, max_len, const_cast<void*>(act), act_bytes, dcls,
                             dhidden, grads, scratch, scratch_bytes, c ? c->layers - 1 : 0, 0);
}

extern "C" int simx_bert_bwd_range(simx_stream_t stream, const simx_bert_cfg* c, const float* params, const void* wcache,
                                   const int32_t* ids, const int32_t* pos_ids, const int32_t* cu, int nseq, int T, int max_len,
                                   void* act, size_t act_bytes, const float* dcls, const void* dhidden, float* grads,
                                   void* scratch, size_t scratch_bytes, int layer_hi, int layer_lo) {
  RUN(check_io(c, nseq, T, max_len, "bert_bwd"));
  SIMX_REQUIRE(params && wcache && ids && pos_ids && cu && act && grads && scratch, SIMX_ERR_BAD_SHAPE, "bert_bwd: NULL buffer");
  SIMX_REQUIRE(layer_hi < c->layers && layer_lo >= 0 && layer_lo <= layer_hi, SIMX_ERR_BAD_SHAPE, "bert_bwd: bad layer range %d..%d",
               layer_hi, layer_lo);
  const bool top = layer_hi == c->layers - 1;
  SIMX_REQUIRE(!top || ((dcls != nullptr) != (dhidden != nullptr)), SIMX_ERR_BAD_SHAPE, "bert_bwd: pass exactly one of dcls / dhidden");
  SIMX_REQUIRE(!(dhidden && c->cls_only_last_layer), SIMX_ERR_BAD_SHAPE,
               "bert_bwd: a gradient for the whole hidden state needs the full last layer (cls_only_last_layer = 0)");
  SIMX_REQUIRE(act_bytes >= simx_bert_act_bytes(c, T, nseq, 1), SIMX_ERR_WORKSPACE, "bert_bwd: activation buffer too small");
  SIMX_REQUIRE(scratch_bytes >= simx_bert_bwd_scratch_bytes(c, T, nseq), SIMX_ERR_WORKSPACE, "bert_bwd: scratch %zu < %zu",
               scratch_bytes, simx_bert_bwd_scratch_bytes(c, T, nseq));
  const int H = c->hidden, F = c->inter, dt = c->dtype, d = H / c->heads;
  const int Tp = rows_cap(T);
  const size_t e = esz(dt);
  const bool ckpt = c->grad_checkpoint != 0;
  auto off = [&](int l, int w) { return params + simx_bert_param_offset(c, l, w); };
  auto goff = [&](int l, int w) { return grads + simx_bert_param_offset(c, l, w); };
  // scratch: bufB carries the gradient w.r.t. the current layer's OUTPUT from one layer to the next -- and from one
  // simx_bert_bwd_range call to the next (the caller keeps `scratch` alive and untouched between the parts)
  char* bufA = (char*)scratch;
  char* bufB = bufA + al((size_t)Tp * H * e);
  char* bufC = bufB + al((size_t)Tp * H * e);                    // dropout-masked copy of dz (only with hidden dropout)
  char* du = bufC + al((size_t)Tp * H * e);
  const bool hd = c->hidden_dropout > 0.f;
  char* dqkv = du + al((size_t)Tp * F * e);
  char* tnws = dqkv + al((size_t)Tp * 3 * H * e);
  const size_t tnws_bytes = tn_ws_max(c, T);
  char* xconv = tnws + tnws_bytes + 3 * al((size_t)(nseq > 0 ? nseq : Tp) * H * e);   // operand planes: bf16 pair of a wgrad's activation operand
  const bool pl = pl_run(c, Tp, max_len);
  const int hm = hm_rows_for(c, T, Tp, max_len);        // layout of qkv / dqkv (the forward's choice: same config copy, same inputs)
  // fp16 engine: activation gradients travel multiplied by the loss scale S (gs = {S, 1/S} on the device); the kernels that
  // accumulate into `grads` multiply by 1/S, so `grads` holds true gradients (simx.h "gradient scale")
  const float* gs = dt == SIMX_F16 ? c->grad_scale : nullptr;
  // fp32 engine: dgrad / wgrad GEMMs from bf16 hi + lo splits (gradients need f32's exponent range; simx.h SIMX_F32_SPLIT_B)
  const int gdb = (dt == SIMX_F32 && c->f32_gemm == 0) ? SIMX_F32_SPLIT_B : dt;

  int l_top = layer_hi;
  if (top && c->cls_only_last_layer) {
    // mirror of the forward's [CLS]-only last layer: everything up to the attention core runs on nseq rows
    const int l = l_top;
    const WLayer w = wlayer(c, params, wcache, l);
    const char* xin = layer_input(c, act, Tp, l);
    const ALayer a = ckpt ? alayer_recompute(c, act, Tp) : alayer(c, act, Tp, l, 1);
    if (ckpt) RUN(layer_fwd((hipStream_t)stream, c, params, wcache, l, xin, layer_input_lo(c, act, Tp, l), layer_input_pl(c, act, Tp, l), a, act_extra(c, act, Tp, 1), cu, nseq, T, Tp, max_len, 1, true, nullptr));
    const simx_dropout d1 = drop_of(c, l, 1), d2 = drop_of(c, l, 2), d3 = drop_of(c, l, 3);
    char* dzm = hd ? bufC : bufA;
    char* e0 = tnws + tnws_bytes;                        // three [nseq,H] temporaries
    char* e1 = e0 + al((size_t)nseq * H * e);
    char* e2 = e1 + al((size_t)nseq * H * e);
    const char* qc = act_extra(c, act, Tp, 1);                                     // saved by the forward
    const char* ctxc = qc + al((size_t)nseq * H * e);
    RUN(simx_rows_copy_gs(stream, SIMX_F32, dt, nseq, H, nullptr, nullptr, dcls, bufB, gs));
    const bool slc = stream_lo(c);                           // (the compact rows kept by the forward: see layer_fwd)
    const char* xg = ctxc + 2 * al((size_t)nseq * H * e);
    const char* xgl = xg + al((size_t)nseq * H * e);
    RUN(simx_ln_bwd_res(stream, dt, nseq, H, a.z2, slc ? a.x1 : nullptr, slc ? a.x1l : nullptr, off(l, SIMX_P_LN2_G), c->eps, bufB, bufA,
                        hd ? bufC : nullptr, goff(l, SIMX_P_LN2_G), goff(l, SIMX_P_LN2_B), goff(l, SIMX_P_B2), &d2, cu, gs));
    RUN(simx_gemm_nt(stream, gdb, nseq, F, H, dzm, H, w.w2T, H, du, F, nullptr, nullptr, 0, SIMX_EPI_DGELU, a.u, F, nullptr, 0));
    RUN(simx_gemm_tn_gs(stream, gdb, H, F, nseq, dzm, H, a.h, F, goff(l, SIMX_P_W2), F, 1, tnws, tnws_bytes, nullptr, 0, gs));
    RUN(simx_gemm_nt(stream, gdb, nseq, H, F, du, F, w.w1T, F, bufB, H, nullptr, bufA, H, SIMX_EPI_NONE, nullptr, 0, nullptr, 0));
    RUN(simx_gemm_tn_gs(stream, gdb, F, H, nseq, du, F, a.x1, H, goff(l, SIMX_P_W1), H, 1, tnws, tnws_bytes, goff(l, SIMX_P_B1), 0, gs));
    RUN(simx_ln_bwd_res(stream, dt, nseq, H, a.z1, slc ? xg : nullptr, slc ? xgl : nullptr, off(l, SIMX_P_LN1_G), c->eps, bufB, bufA,
                        hd ? bufC : nullptr, goff(l, SIMX_P_LN1_G), goff(l, SIMX_P_LN1_B), goff(l, SIMX_P_BO), &d1, cu, gs));
    // e1 = dctx (gradient of the attention context, [CLS] rows), bufA[0:nseq] = gradient of the residual branch
    RUN(simx_gemm_nt(stream, gdb, nseq, H, H, dzm, H, w.woT, H, e1, H, nullptr, nullptr, 0, SIMX_EPI_NONE, nullptr, 0, nullptr, 0));
    RUN(simx_gemm_tn_gs(stream, gdb, H, H, nseq, dzm, H, ctxc, H, goff(l, SIMX_P_WO), H, 1, tnws, tnws_bytes, nullptr, 0, gs));
    // attention core for the one query per sequence: e0 = dq [nseq,H]; dK, dV for every token -> dqkv[:, H:3H]
    RUN(simx_mha_cls_bwd_hm(stream, dt, nseq, c->heads, d, cu, max_len, T, qc, a.qkv, e1, e0, dqkv, &d3, hm));
    // Q projection ([CLS] rows): dWq, dbq, and dx = dq . Wq + (residual-branch gradient), still compact
    RUN(simx_rows_copy(stream, dt, dt, nseq, H, cu, nullptr, xin, e2));
    RUN(simx_gemm_tn_gs(stream, gdb, H, H, nseq, e0, H, e2, H, goff(l, SIMX_P_WQKV), H, 1, tnws, tnws_bytes, goff(l, SIMX_P_BQKV), 0, gs));
    RUN(simx_gemm_nt(stream, gdb, nseq, H, H, e0, H, w.wqkvT, 3 * H, e1, H, nullptr, bufA, H, SIMX_EPI_NONE, nullptr, 0, nullptr, 0));
    // back to full tensors: that gradient is zero outside the [CLS] rows
    if (hipMemsetAsync(bufA, 0, (size_t)T * H * e, (hipStream_t)stream) != hipSuccess) {
      simx_set_error("bert_bwd: memset failed");
      return SIMX_ERR_HIP;
    }
    RUN(simx_rows_copy(stream, dt, dt, nseq, H, nullptr, cu, e1, bufA));
    // K, V projections (every token): dx += dkv . Wkv ; dWkv, dbkv
    if (hm) {
      const char* dkv = dqkv + (size_t)c->heads * hm * 64 * e;                  // planes [heads, 3*heads)
      RUN(simx_gemm_nt_hm(stream, dt, Tp, H, 2 * H, dkv, 64, w.wqkvT + (size_t)H * e, 3 * H, bufB, H, nullptr, bufA, H, nullptr, hm, 0));
      RUN(simx_gemm_tn_gs(stream, gdb, 2 * H, H, T, dkv, 0, xin, H, goff(l, SIMX_P_WQKV) + (size_t)H * H, H, 1, tnws, tnws_bytes,
                          goff(l, SIMX_P_BQKV) + H, hm, gs));
    } else {
      RUN(simx_gemm_nt(stream, gdb, Tp, H, 2 * H, dqkv + (size_t)H * e, 3 * H, w.wqkvT + (size_t)H * e, 3 * H, bufB, H, nullptr, bufA, H,
                       SIMX_EPI_NONE, nullptr, 0, nullptr, 0));
      RUN(simx_gemm_tn_gs(stream, gdb, 2 * H, H, T, dqkv + (size_t)H * e, 3 * H, xin, H, goff(l, SIMX_P_WQKV) + (size_t)H * H, H, 1, tnws, tnws_bytes, goff(l, SIMX_P_BQKV) + H, 0, gs));
    }
    --l_top;
  } else if (top && dhidden) {
    if (hipMemcpyAsync(bufB, dhidden, (size_t)T * H * e, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) {
      simx_set_error("bert_bwd: copy of the hidden-state gradient failed");
      return SIMX_ERR_HIP;
    }
  } else if (top) {
    RUN(simx_cls_scatter_gs(stream, dt, nseq, H, T, cu, dcls, bufB, gs));            // g_x = d(loss)/d(last hidden)
  }
  for (int l = l_top; l >= layer_lo; --l) {
    const WLayer w = wlayer(c, params, wcache, l);
    const char* xin = layer_input(c, act, Tp, l);
    const ALayer a = ckpt ? alayer_recompute(c, act, Tp) : alayer(c, act, Tp, l, 1);
    const char* xinl = layer_input_lo(c, act, Tp, l);
    const bool sl = stream_lo(c);
    if (ckpt) RUN(layer_fwd((hipStream_t)stream, c, params, wcache, l, xin, xinl, layer_input_pl(c, act, Tp, l), a, nullptr, cu, nseq, T, Tp, max_len, 1, false, nullptr));
    const simx_dropout d1 = drop_of(c, l, 1), d2 = drop_of(c, l, 2), d3 = drop_of(c, l, 3);
    if (pl) {
      // fp32 engine on operand planes (see layer_fwd): gradients that feed a GEMM leave their producer as bf16 plane pairs --
      // bufC = dz (x dropout mask) from the LayerNorm backward, du from the DGELU epilogue, dqkv from the attention backward;
      // the activation operand of each wgrad GEMM is converted to a bf16 pair into xconv right before its use
      const long psH = (long)Tp * H, psF = (long)Tp * F, ps3 = (long)Tp * 3 * H;
      RUN(simx_ln_bwd_planes(stream, T, H, (const float*)a.z2, off(l, SIMX_P_LN2_G), c->eps, (const float*)bufB, (float*)bufA, bufC, psH,
                             goff(l, SIMX_P_LN2_G), goff(l, SIMX_P_LN2_B), goff(l, SIMX_P_B2), &d2));
      // (the bias gradient of B1 = column sums of du comes out of this epilogue, so the W1 wgrad runs on the four-plane-stage
      // kernel, which carries no fused bias pass; the deterministic mode keeps the wgrad kernel's ordered pass)
      const bool b1_here = !simx_det();
      RUN(simx_gemm_nt_planes_cs(stream, SIMX_BF16, SIMX_EPI_DGELU, Tp, F, H, bufC, H, psH, w.w2TP, H, (long)F * H, nullptr, F, nullptr,
                                 (const float*)a.u, F, du, F, psF, nullptr, b1_here ? goff(l, SIMX_P_B1) : nullptr, T));
      RUN(simx_planes_from(stream, SIMX_F16, SIMX_BF16, T, F, a.h, F, psF, xconv, F, psF));
      RUN(simx_gemm_tn_planes(stream, H, F, T, bufC, H, psH, xconv, F, psF, goff(l, SIMX_P_W2), F, 1, tnws, tnws_bytes, nullptr));
      RUN(simx_gemm_nt_planes(stream, SIMX_BF16, SIMX_EPI_NONE, Tp, H, F, du, F, psF, w.w1TP, F, (long)H * F, (float*)bufB, H, nullptr,
                              (const float*)bufA, H, nullptr, 0, 0, nullptr));
      RUN(simx_planes_from(stream, SIMX_F32, SIMX_BF16, T, H, a.x1, H, 0, xconv, H, psH));
      RUN(simx_gemm_tn_planes(stream, F, H, T, du, F, psF, xconv, H, psH, goff(l, SIMX_P_W1), H, 1, tnws, tnws_bytes,
                              b1_here ? nullptr : goff(l, SIMX_P_B1)));
      RUN(simx_ln_bwd_planes(stream, T, H, (const float*)a.z1, off(l, SIMX_P_LN1_G), c->eps, (const float*)bufB, (float*)bufA, bufC, psH,
                             goff(l, SIMX_P_LN1_G), goff(l, SIMX_P_LN1_B), goff(l, SIMX_P_BO), &d1));
      RUN(simx_gemm_nt_planes(stream, SIMX_BF16, SIMX_EPI_NONE, Tp, H, H, bufC, H, psH, w.woTP, H, (long)H * H, (float*)bufB, H, nullptr, nullptr, 0,
                              nullptr, 0, 0, nullptr));
      RUN(simx_planes_from(stream, SIMX_F16, SIMX_BF16, T, H, a.ctx, H, psH, xconv, H, psH));
      RUN(simx_gemm_tn_planes(stream, H, H, T, bufC, H, psH, xconv, H, psH, goff(l, SIMX_P_WO), H, 1, tnws, tnws_bytes, nullptr));
      // (x3 attention: the QKV bias gradient = column sums of dq | dk | dv comes out of the attention backward, so the Wqkv wgrad
      // runs on the four-plane-stage kernel too; the deterministic mode keeps the wgrad kernel's ordered pass)
      const bool bqkv_here = simx_mha_x3_ok(d, max_len) && !simx_det();
      if (simx_mha_x3_ok(d, max_len))
        RUN(simx_mha_bwd_x3_bias(stream, nseq, c->heads, d, cu, max_len, T, a.qkv, ps3, a.ctx, psH, a.lse, (const float*)bufB, dqkv, ps3, &d3,
                                 bqkv_here ? goff(l, SIMX_P_BQKV) : nullptr));
      else
        RUN(simx_mha_bwd_planes(stream, nseq, c->heads, d, cu, max_len, T, (const float*)a.qkv, a.ctx, psH, a.lse, (const float*)bufB, dqkv, ps3, &d3));
      RUN(simx_gemm_nt_planes(stream, SIMX_BF16, SIMX_EPI_NONE, Tp, H, 3 * H, dqkv, 3 * H, ps3, w.wqkvTP, 3 * H, 3L * H * H, (float*)bufB, H, nullptr,
                              (const float*)bufA, H, nullptr, 0, 0, nullptr));
      RUN(simx_planes_from(stream, SIMX_F32, SIMX_BF16, T, H, xin, H, 0, xconv, H, psH));
      RUN(simx_gemm_tn_planes(stream, 3 * H, H, T, dqkv, 3 * H, ps3, xconv, H, psH, goff(l, SIMX_P_WQKV), H, 1, tnws, tnws_bytes,
                              bqkv_here ? nullptr : goff(l, SIMX_P_BQKV)));
      continue;
    }
    char* dzm = hd ? bufC : bufA;        // gradient of the (dropped) dense output; bufA = gradient of the residual branch
    // output LayerNorm : dz2, dgamma2, dbeta2, db2
    RUN(simx_ln_bwd_res(stream, dt, T, H, a.z2, sl ? a.x1 : nullptr, sl ? a.x1l : nullptr, off(l, SIMX_P_LN2_G), c->eps, bufB, bufA,
                        hd ? bufC : nullptr, goff(l, SIMX_P_LN2_G), goff(l, SIMX_P_LN2_B), goff(l, SIMX_P_B2), &d2, nullptr, gs));
    // du = (dz2m . W2) * gelu'(u)
    RUN(simx_gemm_nt(stream, gdb, Tp, F, H, dzm, H, w.w2T, H, du, F, nullptr, nullptr, 0, SIMX_EPI_DGELU, a.u, F, nullptr, 0));
    RUN(simx_gemm_tn_gs(stream, gdb, H, F, T, dzm, H, a.h, F, goff(l, SIMX_P_W2), F, 1, tnws, tnws_bytes, nullptr, 0, gs));
    // dx1 = du . W1 + dz2
    RUN(simx_gemm_nt(stream, gdb, Tp, H, F, du, F, w.w1T, F, bufB, H, nullptr, bufA, H, SIMX_EPI_NONE, nullptr, 0, nullptr, 0));
    RUN(simx_gemm_tn_gs(stream, gdb, F, H, T, du, F, a.x1, H, goff(l, SIMX_P_W1), H, 1, tnws, tnws_bytes, goff(l, SIMX_P_B1), 0, gs));
    // attention-output LayerNorm : dz1, dgamma1, dbeta1, dbo
    RUN(simx_ln_bwd_res(stream, dt, T, H, a.z1, sl ? xin : nullptr, sl ? xinl : nullptr, off(l, SIMX_P_LN1_G), c->eps, bufB, bufA,
                        hd ? bufC : nullptr, goff(l, SIMX_P_LN1_G), goff(l, SIMX_P_LN1_B), goff(l, SIMX_P_BO), &d1, nullptr, gs));
    // dctx = dz1m . Wo
    RUN(simx_gemm_nt(stream, gdb, Tp, H, H, dzm, H, w.woT, H, bufB, H, nullptr, nullptr, 0, SIMX_EPI_NONE, nullptr, 0, nullptr, 0));
    RUN(simx_gemm_tn_gs(stream, gdb, H, H, T, dzm, H, a.ctx, H, goff(l, SIMX_P_WO), H, 1, tnws, tnws_bytes, nullptr, 0, gs));
    RUN(simx_mha_bwd_hm(stream, dt, nseq, c->heads, d, cu, max_len, T, a.qkv, a.ctx, a.lse, bufB, dqkv, &d3, hm));
    // dx = dqkv . Wqkv + dz1
    if (hm) {
      RUN(simx_gemm_nt_hm(stream, dt, Tp, H, 3 * H, dqkv, 64, w.wqkvT, 3 * H, bufB, H, nullptr, bufA, H, nullptr, hm, 0));
      RUN(simx_gemm_tn_gs(stream, gdb, 3 * H, H, T, dqkv, 0, xin, H, goff(l, SIMX_P_WQKV), H, 1, tnws, tnws_bytes, goff(l, SIMX_P_BQKV), hm, gs));
    } else {
      RUN(simx_gemm_nt(stream, gdb, Tp, H, 3 * H, dqkv, 3 * H, w.wqkvT, 3 * H, bufB, H, nullptr, bufA, H, SIMX_EPI_NONE, nullptr, 0,
                       nullptr, 0));
      RUN(simx_gemm_tn_gs(stream, gdb, 3 * H, H, T, dqkv, 3 * H, xin, H, goff(l, SIMX_P_WQKV), H, 1, tnws, tnws_bytes,
                          goff(l, SIMX_P_BQKV), 0, gs));
    }
  }
  if (layer_lo > 0) return SIMX_OK;               // the next part continues from bufB
  const simx_dropout d0 = drop_of(c, -1, 0);
  RUN(simx_embed_ln_bwd_seq_gs(stream, dt, nseq, max_len, T, H, cu, ids, pos_ids, off(-1, SIMX_P_WORD), off(-1, SIMX_P_POS),
                               off(-1, SIMX_P_TYPE), off(-1, SIMX_P_EMB_LN_G), c->eps, bufB, goff(-1, SIMX_P_WORD),
                               goff(-1, SIMX_P_POS), goff(-1, SIMX_P_TYPE), goff(-1, SIMX_P_EMB_LN_G), goff(-1, SIMX_P_EMB_LN_B), &d0, gs));
  return SIMX_OK;
}

// ------------------------------------------------------------------------------------------ profiler
#include <vector>
#include "prof.h"
namespace {
struct ProfRec { int id; double work; hipEvent_t a, b; bool done; };
bool g_prof_on = false;
std::vector<ProfRec> g_prof;
std::vector<hipEvent_t> g_pool;
size_t g_pool_next = 0;
}  // namespace

int simx_prof_mark(int id, hipStream_t s, double work, int end_index) {
  if (end_index < 0) {
    if (!g_prof_on || g_pool_next + 2 > g_pool.size()) return -1;
    ProfRec r{id, work, g_pool[g_pool_next], g_pool[g_pool_next + 1], false};
    g_pool_next += 2;
    (void)hipEventRecord(r.a, s);
    g_prof.push_back(r);
    return (int)g_prof.size() - 1;
  }
  if ((size_t)end_index < g_prof.size() && !g_prof[end_index].done) {      // (scopes nest: each closes its own record)
    (void)hipEventRecord(g_prof[end_index].b, s);
    g_prof[end_index].done = true;
  }
  return end_index;
}

void simx_prof_retag(int id) {
  if (!g_prof_on) return;
  for (size_t i = g_prof.size(); i-- > 0;)
    if (!g_prof[i].done) { g_prof[i].id = id; return; }
}

extern "C" int simx_prof_begin(int max_launches) {
  while ((int)g_pool.size() < 2 * max_launches) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) { simx_set_error("prof_begin: hipEventCreate failed"); return SIMX_ERR_HIP; }
    g_pool.push_back(e);
  }
  g_prof.clear();
  g_pool_next = 0;
  g_prof_on = true;
  return SIMX_OK;
}

// Stops recording, waits for the recorded events and aggregates per kernel class:
// counts[k], total_ms[k], total_work[k] for k < SIMX_K_COUNT (host arrays).
extern "C" int simx_prof_end(int32_t* counts_host, double* total_ms_host, double* total_work_host) {
  g_prof_on = false;
  for (int k = 0; k < SIMX_K_COUNT; ++k) { counts_host[k] = 0; total_ms_host[k] = 0; total_work_host[k] = 0; }
  for (auto& r : g_prof) {
    if (!r.done) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    counts_host[r.id] += 1;
    total_ms_host[r.id] += ms;
    total_work_host[r.id] += r.work;
  }
  g_prof.clear();
  (void)hipGetLastError();      // a failed event query must not surface as the next launch's error
  return SIMX_OK;
}
extern "C" int simx_prof_kernel_count(void) { return SIMX_K_COUNT; }
