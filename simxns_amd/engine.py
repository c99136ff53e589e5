"""Host side of the MI355X encoder engine: owns the flat f32 parameter / gradient buffers, the
bf16 weight cache and the per-call activation buffers, and drives the C-ABI encoder
(simx_bert_fwd / simx_bert_bwd) from PyTorch autograd.

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream and the
autograd edge between the [CLS] embeddings and the loss.  All arithmetic runs in
libsimx_hip.so; there is no eager fallback.
"""
import ctypes as C
import json
import logging
import os
from collections import OrderedDict

import torch

from . import _lib as L


class BertConfigLite(object):
    """The few BertConfig fields the path needs (config.json compatible with HF)."""

    def __init__(self, vocab_size=30522, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
                 intermediate_size=3072, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12,
                 hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, **kw):
        self.vocab_size, self.hidden_size = vocab_size, hidden_size
        self.num_hidden_layers, self.num_attention_heads = num_hidden_layers, num_attention_heads
        self.intermediate_size, self.max_position_embeddings = intermediate_size, max_position_embeddings
        self.type_vocab_size, self.layer_norm_eps = type_vocab_size, layer_norm_eps
        self.hidden_dropout_prob, self.attention_probs_dropout_prob = hidden_dropout_prob, attention_probs_dropout_prob
        self.gradient_checkpointing = kw.get("gradient_checkpointing", False)
        # RoBERTa (E4): position ids start at pad_token_id + 1 (HF create_position_ids_from_input_ids on right-padded
        # rows) and the model is built without a pooler (RobertaModel(config, add_pooling_layer=False), models.py:340)
        roberta = kw.get("model_type") == "roberta"
        self.pad_token_id = kw.get("pad_token_id", 1 if roberta else 0)
        self.position_offset = kw.get("position_offset", self.pad_token_id + 1 if roberta else 0)
        self.add_pooling_layer = kw.get("add_pooling_layer", True)
        self.extra = kw

    # architectures of the hub names the reference's recipes pass as --model_type / --teacher_model_type
    # (SimANS/train_*_AR2.sh, PROD/README.md); this image has no network, so a name resolves to its published config and
    # the weights are random unless a local directory with the same name holds them
    KNOWN = {
        "bert-base-uncased": {},
        "bert-large-uncased": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096),
        "Luyu/co-condenser-marco": {},
        "Luyu/co-condenser-wiki": {},
        "nghuyong/ernie-2.0-en": dict(type_vocab_size=4),
        "nghuyong/ernie-2.0-base-en": dict(type_vocab_size=4),
        "nghuyong/ernie-2.0-large-en": dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16,
                                            intermediate_size=4096, type_vocab_size=4),
        "roberta-base": dict(vocab_size=50265, max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5,
                             model_type="roberta", pad_token_id=1),
    }

    @classmethod
    def from_pretrained(cls, path):
        f = path if str(path).endswith(".json") else os.path.join(str(path), "config.json")
        if not os.path.exists(f):
            name = str(path).rstrip("/")
            for key in (name, os.path.basename(name)):
                for k, kw in cls.KNOWN.items():
                    if key == k or key == os.path.basename(k):
                        return cls(**kw)
            raise FileNotFoundError(
                "model_type %r: no config.json under that path and not one of the known architectures %s. Provide a local "
                "directory holding config.json (+ model.safetensors / pytorch_model.bin): this image has no network access "
                "to the HF hub." % (path, sorted(cls.KNOWN)))
        with open(f) as fh:
            return cls(**json.load(fh))

    def to_dict(self):
        return dict(vocab_size=self.vocab_size, hidden_size=self.hidden_size, num_hidden_layers=self.num_hidden_layers,
                    num_attention_heads=self.num_attention_heads, intermediate_size=self.intermediate_size,
                    max_position_embeddings=self.max_position_embeddings, type_vocab_size=self.type_vocab_size,
                    layer_norm_eps=self.layer_norm_eps, model_type="bert")


def _dtype_code(name):
    if name in ("fp32_exact", "fp32-exact"):          # fp32 engine with exact f32 products in every GEMM (simx.h f32_gemm = 1)
        return L.SIMX_F32
    if name in ("fp16", "float16", "f16", "half", "fp16_plain", torch.float16):
        return L.SIMX_F16
    if name in ("bf16", "bfloat16", torch.bfloat16):
        return L.SIMX_BF16
    if name in ("fp32", "float32", "f32", torch.float32):
        return L.SIMX_F32
    raise ValueError("compute dtype must be 'fp32', 'fp16' or 'bf16', got %r" % (name,))


class LossScaler(object):
    """Loss scale of the fp16 engine: apex.amp's dynamic loss scaling (amp.initialize(opt_level='O1') /
    amp.scale_loss, SimANS/co_training/co_training_marco_train.py:97-104, 218-220) kept ENTIRELY on the device.
    Activation gradients of an fp16 tower travel multiplied by S; every kernel that adds into the f32 parameter
    gradients multiplies by 1/S, so ``param.grad`` is always the true gradient (what apex leaves behind when the
    ``scale_loss`` context exits).  FusedAdamW owns a dynamic one per optimiser (apex defaults: 2^16, /2 on inf / nan with
    the step skipped, x2 after 2000 clean steps); an fp16 engine used without an optimiser (tests, inference-time
    gradients) falls back to a static 2^10.  State layout: include/simx.h "Dynamic loss scaler"."""

    def __init__(self, init_scale=2.0 ** 16, growth_interval=2000, max_scale=2.0 ** 24):
        self.init_scale, self.growth_interval, self.max_scale = float(init_scale), float(growth_interval), float(max_scale)
        self._state = None

    def state(self, device):
        if self._state is None or self._state.device != torch.device(device):
            old = None if self._state is None else self._state.cpu()
            self._state = torch.empty(8, dtype=torch.float32, device=device)
            if old is None:
                L.call("simx_scaler_init", L.stream_ptr(), L.ptr(self._state), self.init_scale, self.growth_interval, self.max_scale)
            else:
                self._state.copy_(old)
        return self._state

    def update(self, sqnorm):
        """One optimiser step's verdict from the squared norm of ALL its gradient buffers (device scalar)."""
        L.call("simx_scaler_update", L.stream_ptr(), L.ptr(self.state(sqnorm.device)), L.ptr(sqnorm))

    def snapshot(self):
        """Host copy (synchronises): {'scale', 'applied_steps', 'skipped_steps', 'clean_steps'}."""
        if self._state is None:
            return {"scale": self.init_scale, "applied_steps": 0, "skipped_steps": 0, "clean_steps": 0}
        v = self._state.tolist()
        return {"scale": v[0], "applied_steps": int(v[4]), "skipped_steps": int(v[5]), "clean_steps": int(v[2])}

    def state_dict(self):
        d = self.snapshot()
        d.update(growth_interval=self.growth_interval, max_scale=self.max_scale)
        return d

    def load_state_dict(self, d, device):
        st = self.state(device)
        S = float(d["scale"])
        st.copy_(torch.tensor([S, 1.0 / S, float(d.get("clean_steps", 0)), 0.0, float(d.get("applied_steps", 0)),
                               float(d.get("skipped_steps", 0)), float(d.get("growth_interval", self.growth_interval)),
                               float(d.get("max_scale", self.max_scale))], dtype=torch.float32))


def hf_param_layout(cfg, ccfg):
    """[(hf_key, flat_offset, shape)] in HF registration order (state_dict schema)."""
    lib = L.load()
    H, F = cfg.hidden_size, cfg.intermediate_size
    off = lambda layer, which: lib.simx_bert_param_offset(C.byref(ccfg), layer, which)
    out = [("embeddings.word_embeddings.weight", off(-1, L.P_WORD), (cfg.vocab_size, H)),
           ("embeddings.position_embeddings.weight", off(-1, L.P_POS), (cfg.max_position_embeddings, H)),
           ("embeddings.token_type_embeddings.weight", off(-1, L.P_TYPE), (cfg.type_vocab_size, H)),
           ("embeddings.LayerNorm.weight", off(-1, L.P_EMB_LN_G), (H,)),
           ("embeddings.LayerNorm.bias", off(-1, L.P_EMB_LN_B), (H,))]
    for i in range(cfg.num_hidden_layers):
        p = "encoder.layer.%d." % i
        wq, bq = off(i, L.P_WQKV), off(i, L.P_BQKV)
        for j, nm in enumerate(("query", "key", "value")):
            out.append((p + "attention.self.%s.weight" % nm, wq + j * H * H, (H, H)))
            out.append((p + "attention.self.%s.bias" % nm, bq + j * H, (H,)))
        out += [(p + "attention.output.dense.weight", off(i, L.P_WO), (H, H)),
                (p + "attention.output.dense.bias", off(i, L.P_BO), (H,)),
                (p + "attention.output.LayerNorm.weight", off(i, L.P_LN1_G), (H,)),
                (p + "attention.output.LayerNorm.bias", off(i, L.P_LN1_B), (H,)),
                (p + "intermediate.dense.weight", off(i, L.P_W1), (F, H)),
                (p + "intermediate.dense.bias", off(i, L.P_B1), (F,)),
                (p + "output.dense.weight", off(i, L.P_W2), (H, F)),
                (p + "output.dense.bias", off(i, L.P_B2), (H,)),
                (p + "output.LayerNorm.weight", off(i, L.P_LN2_G), (H,)),
                (p + "output.LayerNorm.bias", off(i, L.P_LN2_B), (H,))]
    n = cfg.num_hidden_layers
    if getattr(cfg, "add_pooling_layer", True):      # (the slots stay in the flat buffer either way; never computed)
        out += [("pooler.dense.weight", off(n, L.P_POOL_W), (H, H)), ("pooler.dense.bias", off(n, L.P_POOL_B), (H,))]
    return out


class PackedBatch(object):
    """Right-padded (input_ids, attention_mask) -> packed real tokens + cu_seqlens (device, int32)."""

    def __init__(self, input_ids, attention_mask, pos_offset=0):
        assert input_ids.dim() == 2 and input_ids.shape == attention_mask.shape
        n, S = input_ids.shape
        m = attention_mask != 0
        lens = m.sum(1)
        stats = torch.stack([lens.sum(), lens.max(), m[:, 0].min().to(lens.dtype)]).tolist()   # one host sync
        self.T, self.max_len = int(stats[0]), int(stats[1])
        if int(stats[2]) == 0:
            raise ValueError("attention_mask[:, 0] must be 1: the [CLS] position is the embedding (models.py:81)")
        self.nseq, self.S = n, S
        cu = torch.zeros(n + 1, dtype=torch.int32, device=input_ids.device)
        cu[1:] = torch.cumsum(lens, 0).to(torch.int32)
        self.cu = cu
        if self.T == n * S:
            self.index = None
            self.ids = input_ids.reshape(-1).to(torch.int32)
            self.pos = (torch.arange(S, device=input_ids.device, dtype=torch.int32) + pos_offset).repeat(n)
        else:
            self.index = m.reshape(-1).nonzero(as_tuple=False).squeeze(1)
            self.ids = input_ids.reshape(-1)[self.index].to(torch.int32)
            self.pos = (self.index % S + pos_offset).to(torch.int32)

    def unpack(self, packed, fill=0.0):
        """packed [T,H] -> padded [n,S,H] (pad rows = fill)."""
        H = packed.shape[1]
        if self.index is None:
            return packed.reshape(self.nseq, self.S, H)
        out = packed.new_full((self.nseq * self.S, H), fill)
        out[self.index] = packed
        return out.reshape(self.nseq, self.S, H)


class _EncoderFn(torch.autograd.Function):
    """pool: "cls" -> [CLS] embeddings [n,H] f32; "hidden" -> (cls, packed last hidden state [T,H]), both
    differentiable; "mean" -> masked mean over every sequence's real tokens [n,H] f32 (models.py:296-299)."""

    @staticmethod
    def forward(ctx, anchor, engine, pb, pool, ccfg):
        cls, hidden, act = engine._run_forward(pb, True, pool != "cls", ccfg)
        engine._open_graphs += 1                  # graphs that will still add into flat_grad (shared towers: more than one)
        ctx.engine, ctx.pb, ctx.act, ctx.ccfg, ctx.pool = engine, pb, act, ccfg, pool
        if pool == "hidden":
            return cls, hidden
        if pool == "mean":
            return engine._seq_mean(pb, hidden)
        return cls

    @staticmethod
    def backward(ctx, g0, g1=None):
        e, pb = ctx.engine, ctx.pb
        if ctx.pool == "cls":
            e._run_backward(pb, ctx.act, dcls=g0, ccfg=ctx.ccfg)      # same config copy: same dropout seed
        elif ctx.pool == "mean":
            e._run_backward(pb, ctx.act, dhidden=e._seq_mean_bwd(pb, g0), ccfg=ctx.ccfg)
        else:
            H = e.cfg.hidden_size
            dh = torch.zeros(pb.T, H, dtype=torch.float32, device=e.flat.device) if g1 is None else g1.to(torch.float32).clone()
            if g0 is not None:
                dh.index_add_(0, pb.cu[:-1].long(), g0.to(torch.float32))
            e._run_backward(pb, ctx.act, dhidden=e._enter_backward(dh), ccfg=ctx.ccfg)
        ctx.act = None
        return None, None, None, None, None


class BertEngine(object):
    """One BERT tower on one GPU."""

    def __init__(self, cfg, compute_dtype="bf16"):
        self.cfg = cfg
        self.lib = L.load()
        self.set_compute_dtype(compute_dtype)
        self.layout = hf_param_layout(cfg, self.ccfg)
        self.n_params = int(self.lib.simx_bert_param_count(C.byref(self.ccfg)))
        self.flat = torch.zeros(self.n_params, dtype=torch.float32)
        self.flat_grad = None
        self.wcache = None
        self._wcache_version = None
        self._dirty = True
        self.anchor = torch.zeros((), requires_grad=True)
        self.dropout_seed = 0                # base seed of the stateless dropout masks (set_seed / manual)
        # data parallelism: hook(engine, lo, hi) is called as soon as flat_grad[lo:hi] is final; with bwd_parts > 1 the
        # backward runs in that many layer ranges (simx_bert_bwd_range) and the hook fires after each, so the all-reduce of
        # the upper layers' gradients overlaps the lower layers' backward (FusedAdamW.enable_overlap sets both)
        self.grad_ready_hook = None
        self.bwd_parts = 1
        self._open_graphs = 0
        self._bwd_serial = 0                 # counts backward passes (the optimiser tells parts of one pass from a new pass)
        self._reduced_this_step = False      # set by the data-parallel hook once slices of flat_grad are on the wire
        self.last_stream = None              # stream of the last forward / backward (the optimiser joins it)
        self.after_backward = None           # module callback: expose flat_grad as param.grad views
        self.scaler = None                   # fp16 only: the optimiser's LossScaler (FusedAdamW attaches it); None -> static 2^10

    # ---- configuration -----------------------------------------------------------------
    def set_compute_dtype(self, name):
        c = self.cfg
        self.dtype_code = _dtype_code(name)
        # cfg.gradient_checkpointing (models.py:73-74): per-layer recompute in the native backward; SIMX_GRAD_CKPT=0/1 overrides
        ck = os.environ.get("SIMX_GRAD_CKPT")
        ckpt = bool(getattr(c, "gradient_checkpointing", False)) if ck is None else ck == "1"
        # fp32: the dense GEMMs run on the 16-bit matrix cores from hi + lo splits of their f32 operands (simx.h
        # SIMX_F32_SPLIT_H / _B); "fp32_exact" (or SIMX_GEMM_F32=exact) keeps exact f32 products everywhere
        exact = name in ("fp32_exact", "fp32-exact") or os.environ.get("SIMX_GEMM_F32", "")[:1] == "e"
        self.ccfg = L.BertCfg(self.dtype_code, c.num_hidden_layers, c.hidden_size, c.num_attention_heads,
                              c.intermediate_size, c.vocab_size, c.max_position_embeddings, c.type_vocab_size,
                              float(c.layer_norm_eps), 0.0, 0.0, 0, 0, 1 if ckpt else 0, 0, 1 if exact else 0, self._stream_lo(name), None)
        self.wcache = None
        self._dirty = True
        # (one line per engine: which arithmetic a tower actually runs is decided by three inputs -- the argument, SIMX_DTYPE,
        # SIMX_GEMM_F32 / SIMX_STREAM_LO -- and a log is where a user can see the outcome)
        logging.getLogger("simxns_amd").info(
            "BertEngine: compute dtype %s (code %d), f32 GEMMs %s, residual stream %s, gradient checkpointing %s", name, self.dtype_code,
            "exact" if exact else "16-bit hi+lo splits on the matrix cores", "16-bit + correction byte" if self.ccfg.stream_lo else "plain",
            "on" if ckpt else "off")

    def _stream_lo(self, name):
        """fp16: the residual stream carries one correction byte beside every 16-bit value and the LayerNorm kernels add the
        residual in f32 (simx.h stream_lo) -- the f32 residual stream / f32 LayerNorm of apex O1, which the reference's --fp16
        mode is.  "fp16_plain" (or SIMX_STREAM_LO=0) keeps a plain 16-bit stream; bf16 defaults to plain, SIMX_STREAM_LO=1 opts in."""
        if self.dtype_code == L.SIMX_F32:
            return 0
        env = os.environ.get("SIMX_STREAM_LO")
        if env is not None:
            return 1 if env == "1" else 0
        return 1 if (self.dtype_code == L.SIMX_F16 and name != "fp16_plain") else 0

    @property
    def act_torch_dtype(self):
        return {L.SIMX_BF16: torch.bfloat16, L.SIMX_F16: torch.float16}.get(self.dtype_code, torch.float32)

    def grad_scale_state(self):
        """fp16: the device tensor {S, 1/S, ...} the backward kernels read (None for the other dtypes)."""
        if self.dtype_code != L.SIMX_F16:
            return None
        if self.scaler is None:
            self.scaler = LossScaler(init_scale=float(os.environ.get("SIMX_LOSS_SCALE", 1024.0)), growth_interval=0)
        return self.scaler.state(self.flat.device)

    def views(self, base):
        out = OrderedDict()
        for name, off, shape in self.layout:
            n = 1
            for s in shape:
                n *= s
            out[name] = base[off:off + n].view(*shape)
        return out

    def to(self, device):
        if self.flat.device != torch.device(device):
            self.flat = self.flat.to(device)
            if self.flat_grad is not None:
                self.flat_grad = self.flat_grad.to(device)
            self.wcache = None
            self._dirty = True
        return self

    def ensure_grad(self):
        if self.flat_grad is None or self.flat_grad.device != self.flat.device:
            self.flat_grad = torch.zeros_like(self.flat)
        return self.flat_grad

    def mark_weights_dirty(self):
        self._dirty = True

    # ---- device work ---------------------------------------------------------------------
    def _require_gpu(self):
        if not self.flat.is_cuda:
            raise L.SimxError("the encoder runs only on a HIP device: move the module to 'cuda' "
                              "(there is no CPU fallback in the product path)")

    def _refresh_wcache(self):
        if self.wcache is not None and not self._dirty and self._wcache_version == self.flat._version:
            return
        nbytes = int(self.lib.simx_bert_wcache_bytes(C.byref(self.ccfg)))
        if self.wcache is None or self.wcache.numel() != nbytes or self.wcache.device != self.flat.device:
            self.wcache = torch.empty(nbytes, dtype=torch.uint8, device=self.flat.device)
        L.call("simx_bert_cast_weights", L.stream_ptr(), C.byref(self.ccfg), L.ptr(self.flat), L.ptr(self.wcache))
        self._dirty = False
        self._wcache_version = self.flat._version

    def call_cfg(self, training, want_hidden=True):
        """Per-call copy of the C config: dropout on only in training mode (nn.Dropout semantics), fresh seed per call.
        Callers that only read the [CLS] embedding (BiBertEncoder / Reranker / RobertaDot, models.py:81) let the last
        layer's post-attention blocks run on the [CLS] rows alone (SIMX_FULL_LAST_LAYER=1 computes every row anyway)."""
        c = L.BertCfg.from_buffer_copy(self.ccfg)
        c.cls_only_last_layer = 0 if (want_hidden or os.environ.get("SIMX_FULL_LAST_LAYER", "0") == "1") else 1
        # decided HERE, once per forward: the backward receives this same copy (SIMX_QKV_LAYOUT=token pins token-major)
        c.qkv_layout = 1 if os.environ.get("SIMX_QKV_LAYOUT", "")[:1] == "t" else 0
        if training and (self.cfg.hidden_dropout_prob > 0 or self.cfg.attention_probs_dropout_prob > 0):
            self._drop_calls = getattr(self, "_drop_calls", 0) + 1
            c.hidden_dropout = float(self.cfg.hidden_dropout_prob)
            c.attn_dropout = float(self.cfg.attention_probs_dropout_prob)
            c.dropout_seed = (self.dropout_seed * 0x9E3779B1 + self._drop_calls * 0x85EBCA6B) & 0xFFFFFFFF
        return c

    def _run_forward(self, pb, save, want_hidden, ccfg=None):
        self._require_gpu()
        self._refresh_wcache()
        ccfg = ccfg if ccfg is not None else self.ccfg
        dev = self.flat.device
        H = self.cfg.hidden_size
        # sized for the padded token count nseq*S, not for this batch's T: ragged batches change T every step and a 50+ GB
        # request of a new size makes the caching allocator free and re-malloc the block (measured: +245 ms per step)
        nbytes = int(self.lib.simx_bert_act_bytes(C.byref(self.ccfg), pb.nseq * pb.S, pb.nseq, 1 if save else 0))
        act = self._alloc_act(nbytes, dev, save)
        cls = torch.empty(pb.nseq, H, dtype=torch.float32, device=dev)
        hidden = torch.empty(pb.T, H, dtype=self.act_torch_dtype, device=dev) if want_hidden else None
        L.call("simx_bert_fwd", L.stream_ptr(), C.byref(ccfg), L.ptr(self.flat), L.ptr(self.wcache),
               L.ptr(pb.ids), L.ptr(pb.pos), L.ptr(pb.cu), pb.nseq, pb.T, pb.max_len, L.ptr(act), nbytes,
               1 if save else 0, L.ptr(cls), L.ptr(hidden))
        return cls, hidden, (act if save else None)

    def _alloc_act(self, nbytes, dev, save):
        """The activation buffer of one forward.  Fails loudly -- before the allocator does -- when the kept activations
        cannot fit, and says what to do about it (the reference's answer is --gradient_checkpointing, models.py:73-74)."""
        try:
            return torch.empty(nbytes, dtype=torch.uint8, device=dev)
        except torch.cuda.OutOfMemoryError:
            free, total = torch.cuda.mem_get_info(dev)
            hint = ("" if self.ccfg.grad_checkpoint or not save else
                    "; pass --gradient_checkpointing (cfg.gradient_checkpointing=True / SIMX_GRAD_CKPT=1): only the layer "
                    "inputs are kept and each layer is recomputed in backward")
            raise L.SimxError("activations of this batch need %.1f GB (%d layers, %s) but only %.1f of %.1f GB of HBM are free%s"
                              % (nbytes / 1e9, self.cfg.num_hidden_layers, "checkpointed" if self.ccfg.grad_checkpoint else
                                 "all kept for backward" if save else "inference ring", free / 1e9, total / 1e9, hint))

    def _seq_mean(self, pb, hidden):
        out = torch.empty(pb.nseq, self.cfg.hidden_size, dtype=torch.float32, device=hidden.device)
        L.call("simx_seq_mean_fwd", L.stream_ptr(), self.dtype_code, pb.nseq, self.cfg.hidden_size, L.ptr(pb.cu), L.ptr(hidden), L.ptr(out))
        return out

    def _enter_backward(self, dh):
        """f32 gradient of the whole hidden state [T,H] -> the activation dtype (fp16: multiplied by the loss scale)."""
        dh = dh.contiguous()
        out = torch.empty(dh.shape, dtype=self.act_torch_dtype, device=dh.device)
        L.call("simx_rows_copy_gs", L.stream_ptr(), L.SIMX_F32, self.dtype_code, dh.shape[0], dh.shape[1], None, None, L.ptr(dh), L.ptr(out),
               L.ptr(self.grad_scale_state()))
        return out

    def _seq_mean_bwd(self, pb, dmean):
        dmean = dmean.contiguous().to(torch.float32)
        dh = torch.empty(pb.T, self.cfg.hidden_size, dtype=self.act_torch_dtype, device=dmean.device)
        L.call("simx_seq_mean_bwd_gs", L.stream_ptr(), self.dtype_code, pb.nseq, self.cfg.hidden_size, L.ptr(pb.cu), L.ptr(dmean), L.ptr(dh),
               L.ptr(self.grad_scale_state()))
        return dh

    def _run_backward(self, pb, act, dcls=None, ccfg=None, dhidden=None):
        ccfg = ccfg if ccfg is not None else self.ccfg
        if self._reduced_this_step:
            raise L.SimxError("backward into a gradient buffer whose slices were already all-reduced for this optimiser step: "
                              "the sum would be reduced twice (or mixed with un-reduced gradients).  With gradient accumulation "
                              "set optimizer.armed = False for every micro-step but the last, or call optimizer.step() first")
        self._bwd_serial += 1
        gs = self.grad_scale_state()                 # fp16: the loss scale as it is NOW (the optimiser may have been attached,
        ccfg.grad_scale = None if gs is None else gs.data_ptr()     # or moved its state, after the forward)
        self.last_stream = torch.cuda.current_stream()
        g = self.ensure_grad()
        dev = self.flat.device
        if dcls is not None:
            dcls = dcls.contiguous().to(torch.float32)
        nbytes = int(self.lib.simx_bert_bwd_scratch_bytes(C.byref(self.ccfg), pb.nseq * pb.S, pb.nseq))
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        nl = self.cfg.num_hidden_layers
        self._open_graphs = max(0, self._open_graphs - 1)
        # the hook fires only for the LAST backward that adds into this buffer (an in-place accumulated buffer is reduced once)
        hook = self.grad_ready_hook if self._open_graphs == 0 else None
        parts = max(1, min(int(self.bwd_parts), nl)) if hook is not None else 1
        # layer ranges [hi, lo], top-down; after a part the gradient slice [offset(lo), end of the previous slice) is final
        bounds = [nl - 1 - (i * nl) // parts for i in range(parts)] + [-1]
        end = self.n_params
        for i in range(parts):
            hi, lo = bounds[i], bounds[i + 1] + 1
            L.call("simx_bert_bwd_range", L.stream_ptr(), C.byref(ccfg), L.ptr(self.flat), L.ptr(self.wcache),
                   L.ptr(pb.ids), L.ptr(pb.pos), L.ptr(pb.cu), pb.nseq, pb.T, pb.max_len, L.ptr(act), act.numel(),
                   L.ptr(dcls), L.ptr(dhidden), L.ptr(g), L.ptr(scratch), nbytes, hi, lo)
            if hook is not None:
                start = 0 if lo == 0 else int(self.lib.simx_bert_param_offset(C.byref(self.ccfg), lo, L.P_WQKV))
                hook(self, start, end)
                end = start
        if self.after_backward is not None:
            self.after_backward()

    def encode(self, input_ids, attention_mask, want_hidden=False, requires_grad=None, training=False, pool=None):
        """-> cls [n,H] f32; with want_hidden also the packed last hidden state + PackedBatch (both outputs carry
        gradient); pool="mean" -> the masked mean over each sequence's real tokens [n,H] f32 instead of [CLS]."""
        pool = pool or ("hidden" if want_hidden else "cls")
        pb = PackedBatch(input_ids, attention_mask, getattr(self.cfg, "position_offset", 0))
        ccfg = self.call_cfg(training, pool != "cls")
        if requires_grad is None:
            requires_grad = torch.is_grad_enabled()
        if requires_grad and torch.is_grad_enabled():
            if self.anchor.device != self.flat.device:
                self.anchor = torch.zeros((), requires_grad=True, device=self.flat.device)
            out = _EncoderFn.apply(self.anchor, self, pb, pool, ccfg)
            return (out[0], out[1], pb) if pool == "hidden" else out
        cls, hidden, _ = self._run_forward(pb, False, pool != "cls", ccfg)
        if pool == "mean":
            return self._seq_mean(pb, hidden)
        return (cls, hidden, pb) if pool == "hidden" else cls
