"""Drop-in module API of SimANS/model/models.py on the MI355X engine.

Same class names, constructor arguments, method signatures, return conventions and
``state_dict`` key schema as the reference (file:line cited per symbol), so that
``co_training_*_train.py`` and released AR2 / SimANS checkpoints work unchanged; the arithmetic
runs in libsimx_hip.so (include/simx.h).  Out of scope: Reader/MML, *_daya,
Cross_Encoder, Reranker_2, adv_* forward variants.
"""
import os

import numpy as np
import torch
from torch import nn

from .. import ops
from ..engine import BertConfigLite, BertEngine


class _Holder(nn.Module):
    """Name-space node so that parameters carry HF state_dict keys."""


class HFBertEncoder(nn.Module):
    """SimANS/model/models.py:58-82.  forward(**kwargs) -> (sequence_output, pooled_output, None) where
    pooled_output = sequence_output[:, 0, :] (raw last-layer [CLS], no projection; the pooler is kept in the
    state_dict but its output is multiplied by 0 in the reference, so its gradient is exactly 0 and it is
    never computed here)."""

    def __init__(self, config, compute_dtype=None):
        super(HFBertEncoder, self).__init__()
        assert config.hidden_size > 0, 'Encoder hidden_size can\'t be zero'
        self.config = config
        # one default everywhere: the reference's arithmetic (fp32) unless the caller / SIMX_DTYPE says otherwise
        self.engine = BertEngine(config, compute_dtype or os.environ.get("SIMX_DTYPE", "fp32"))
        for name, view in self.engine.views(self.engine.flat).items():
            parts = name.split(".")
            mod = self
            for p in parts[:-1]:
                if p not in mod._modules:
                    mod.add_module(p, _Holder())
                mod = mod._modules[p]
            mod.register_parameter(parts[-1], nn.Parameter(view))
        self.engine.after_backward = self.attach_grads
        self.init_weights()

    # -- parameters live in ONE flat f32 buffer; nn.Parameters are views of it ------------------
    def _rebind(self):
        views = self.engine.views(self.engine.flat)
        gviews = self.engine.views(self.engine.flat_grad) if self.engine.flat_grad is not None else None
        for name, p in self.named_parameters():
            p.data = views[name]
            if gviews is not None:
                p.grad = gviews[name]
        self.engine.mark_weights_dirty()

    def _apply(self, fn, recurse=True):
        new_flat = fn(self.engine.flat)
        if new_flat.dtype != torch.float32:
            raise TypeError("master parameters stay float32; choose the compute dtype with set_compute_dtype()")
        self.engine.flat = new_flat
        if self.engine.flat_grad is not None:
            self.engine.flat_grad = fn(self.engine.flat_grad)
        self.engine.wcache = None
        self._rebind()
        return self

    def set_compute_dtype(self, name):
        self.engine.set_compute_dtype(name)
        return self

    def attach_grads(self):
        """Expose the flat gradient buffer as ``param.grad`` views (done after every backward)."""
        if self.engine.flat_grad is None:
            return
        gviews = self.engine.views(self.engine.flat_grad)
        for name, p in self.named_parameters():
            if p.grad is None or p.grad.data_ptr() != gviews[name].data_ptr():
                p.grad = gviews[name]

    def zero_grad(self, set_to_none=False):
        # a data-parallel optimiser may already have this tower's gradient slices on the wire (armed backward, then the step is
        # dropped): its collectives are waited for and its per-step bookkeeping reset before the buffer is zeroed
        owner = getattr(getattr(self.engine, "grad_ready_hook", None), "__self__", None)
        if owner is not None and hasattr(owner, "_discard_pending"):
            owner._discard_pending()
        if self.engine.flat_grad is not None:
            self.engine.flat_grad.zero_()
        self.engine._open_graphs = 0          # graphs that never got a backward must not hold back the data-parallel hooks

    def init_weights(self):
        # HF BertPreTrainedModel._init_weights == reference init_weights (models.py:452-465)
        with torch.no_grad():
            for name, p in self.named_parameters():
                if name.endswith("LayerNorm.weight"):
                    p.fill_(1.0)
                elif name.endswith(".bias"):
                    p.zero_()
                else:
                    p.normal_(mean=0.0, std=0.02)
        self.engine.mark_weights_dirty()

    def _load_from_state_dict(self, *a, **k):
        super(HFBertEncoder, self)._load_from_state_dict(*a, **k)
        self.engine.mark_weights_dirty()

    def load_numpy_state(self, state):
        with torch.no_grad():
            own = dict(self.named_parameters())
            for k, v in state.items():
                own[k].copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
        self.engine.mark_weights_dirty()

    @classmethod
    def init_encoder(cls, args, dropout: float = 0.1, model_type=None, compute_dtype=None, num_hidden_layers=None):
        """models.py:65-75.  ``model_type`` is a local directory with config.json (+ model.safetensors or
        pytorch_model.bin) or a known name; without weights on disk the encoder is randomly initialised
        (this image has no network)."""
        if model_type is None:
            model_type = args.model_type
        cfg = BertConfigLite.from_pretrained(model_type)
        if dropout != 0:
            cfg.attention_probs_dropout_prob = dropout
            cfg.hidden_dropout_prob = dropout
        cfg.gradient_checkpointing = bool(getattr(args, "gradient_checkpointing", False))
        if num_hidden_layers:                      # PROD: the depth is a property of the role, not of the checkpoint
            cfg.num_hidden_layers = int(num_hidden_layers)
        if compute_dtype is None:
            # the reference computes in fp32 unless --fp16 (apex O1: fp16 GEMM operands, f32 master weights, dynamic loss
            # scaling) is passed (co_training_marco_train.py:97-104, 218-220); here --fp16 selects the fp16 engine (same operand
            # width, f32 accumulation / statistics / master weights, FusedAdamW's device-side dynamic loss scale), and
            # SIMX_DTYPE=fp16|bf16|fp32 switches any recipe explicitly
            compute_dtype = os.environ.get("SIMX_DTYPE") or ("fp16" if getattr(args, "fp16", False) else "fp32")
        enc = cls(cfg, compute_dtype=compute_dtype)
        if os.path.isdir(str(model_type)):
            st = os.path.join(model_type, "model.safetensors")
            pt = os.path.join(model_type, "pytorch_model.bin")
            sd = None
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu")
            if sd is not None:
                sd = {(k[5:] if k.startswith("bert.") else k): v for k, v in sd.items()}
                enc.load_state_dict({k: v for k, v in sd.items() if k in dict(enc.named_parameters())}, strict=False)
        return enc

    def forward(self, **kwargs):
        input_ids = kwargs.get("input_ids")
        attention_mask = kwargs.get("attention_mask")
        if input_ids is None:
            raise NotImplementedError("inputs_embeds is not supported by the MI355X engine; pass input_ids")
        tt = kwargs.get("token_type_ids")
        if tt is not None and bool((tt != 0).any()):
            raise NotImplementedError("token_type_ids other than 0 are not used on this path (models.py:654)")
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        cls_vec, hidden, pb = self.engine.encode(input_ids, attention_mask, want_hidden=True, training=self.training)
        if self.engine.flat_grad is not None:
            self.attach_grads()
        seq = pb.unpack(hidden.to(torch.float32))
        # every row carries gradient back into the encoder; row 0 is the f32 [CLS] output (exact in bf16 mode too)
        seq = seq.index_copy(1, torch.zeros(1, dtype=torch.long, device=seq.device), cls_vec.unsqueeze(1))
        return seq, seq[:, 0, :], None

    def embed(self, input_ids, attention_mask):
        """Fast path used by BiBertEncoder / Reranker: [CLS] embeddings [n,H] f32 only."""
        return self.engine.encode(input_ids, attention_mask, training=self.training)

    def embed_mean(self, input_ids, attention_mask):
        """EmbeddingMixin.masked_mean (models.py:296-299): mean of the last hidden state over each sequence's real tokens,
        [n,H] f32, computed on the packed layout (no padded sequence_output is materialised)."""
        return self.engine.encode(input_ids, attention_mask, training=self.training, pool="mean")


class RobertaDot(nn.Module):
    """SimANS/model/models.py:277-359 (E4, the MS-Doc student of co_training_doc_train.py:203-208): ONE shared RoBERTa
    encoder without pooler for queries and documents, emb = LayerNorm(Linear(H -> output_embedding_size)(seq[:, 0])).
    ``model_argobj.use_mean`` selects the masked mean over the real tokens instead of row 0 (models.py:283-286, 296-305;
    False on the from_pretrained path the reference's scripts use).  state_dict keys: roberta.*, embeddingHead.*, norm.*"""

    def __init__(self, config, model_argobj=None, compute_dtype=None):
        super(RobertaDot, self).__init__()
        self.use_mean = False if model_argobj is None else bool(model_argobj.use_mean)
        if not isinstance(config, BertConfigLite):
            d = config.to_dict() if hasattr(config, "to_dict") else dict(config)
            d.setdefault("model_type", "roberta")
            config = BertConfigLite(**d)
        config.add_pooling_layer = False
        if config.position_offset == 0:                 # a RoBERTa config without model_type
            config.pad_token_id, config.position_offset = 1, 2
        self.config = config
        self.roberta = HFBertEncoder(config, compute_dtype=compute_dtype)
        self.output_embedding_size = getattr(config, "output_embedding_size", None) or config.extra.get("output_embedding_size", config.hidden_size)
        self.embeddingHead = nn.Linear(config.hidden_size, self.output_embedding_size)
        self.norm = nn.LayerNorm(self.output_embedding_size)
        with torch.no_grad():                            # EmbeddingMixin._init_weights: N(0, 0.02) on Linear weights
            self.embeddingHead.weight.normal_(mean=0.0, std=0.02)

    def query_emb(self, input_ids, attention_mask):
        full_emb = self.roberta.embed_mean(input_ids, attention_mask) if self.use_mean else self.roberta.embed(input_ids, attention_mask)
        z = ops.linear_f32(full_emb, self.embeddingHead.weight, self.embeddingHead.bias)
        return ops.layer_norm_f32(z, self.norm.weight, self.norm.bias, self.norm.eps)

    def body_emb(self, input_ids, attention_mask):
        return self.query_emb(input_ids, attention_mask)

    def forward(self, input_ids, attention_mask, is_query, *args):
        assert len(args) == 0
        return self.query_emb(input_ids, attention_mask) if is_query else self.body_emb(input_ids, attention_mask)

    def zero_grad(self, set_to_none=False):
        self.roberta.zero_grad()
        for p in list(self.embeddingHead.parameters()) + list(self.norm.parameters()):
            if p.grad is not None:
                p.grad.zero_()


class BiBertEncoder(nn.Module):
    """ Bi-Encoder model component. Encapsulates query/question and context/passage encoders.
    (SimANS/model/models.py:85-118) """

    def __init__(self, args):
        super(BiBertEncoder, self).__init__()
        self.question_model = HFBertEncoder.init_encoder(args)
        if hasattr(args, 'share_weight') and args.share_weight:
            self.ctx_model = self.question_model
        else:
            self.ctx_model = HFBertEncoder.init_encoder(args)

    def query_emb(self, input_ids, attention_mask):
        return self.question_model.embed(input_ids, attention_mask)

    def body_emb(self, input_ids, attention_mask):
        return self.ctx_model.embed(input_ids, attention_mask)

    def forward(self, query_ids, attention_mask_q, input_ids_a=None, attention_mask_a=None, input_ids_b=None,
                attention_mask_b=None):
        if input_ids_b is None:
            if self._overlap_towers(query_ids):
                # The query tower (B x 32 tokens) cannot fill 256 CUs: run it on a side stream beside the passage tower.
                # Autograd replays each tower's backward on the stream its forward ran on, so the backward overlaps as
                # well; FusedAdamW.step() joins the streams before it touches the gradients.
                main = torch.cuda.current_stream()
                side = self._side_stream(query_ids.device)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    q_embs = self.query_emb(query_ids, attention_mask_q)
                a_embs = self.body_emb(input_ids_a, attention_mask_a)
                main.wait_stream(side)
                q_embs.record_stream(main)
                return (q_embs, a_embs)
            q_embs = self.query_emb(query_ids, attention_mask_q)
            a_embs = self.body_emb(input_ids_a, attention_mask_a)
            return (q_embs, a_embs)
        # triplet form (models.py:111-118): -log_softmax([q.a, q.b])[0], mean over the batch
        q_embs = self.query_emb(query_ids, attention_mask_q)
        a_embs = self.body_emb(input_ids_a, attention_mask_a)
        b_embs = self.body_emb(input_ids_b, attention_mask_b)
        B = q_embs.shape[0]
        ab = torch.stack([a_embs, b_embs], dim=1).reshape(2 * B, -1)
        loss, _ = ops.pair_ce_loss(q_embs, ab)
        return (loss,)

    def _overlap_towers(self, t):
        return (t.is_cuda and self.question_model is not self.ctx_model and os.environ.get("SIMX_OVERLAP_TOWERS", "1") != "0")

    def _side_stream(self, device):
        st = getattr(self, "_side", None)
        if st is None or st.device != device:
            st = torch.cuda.Stream(device=device)
            object.__setattr__(self, "_side", st)
        return st

    def zero_grad(self, set_to_none=False):
        self.question_model.zero_grad()
        self.ctx_model.zero_grad()


class Reranker(nn.Module):
    """SimANS/model/models.py:638-659: cross-encoder, Linear(H,1) on [CLS]."""

    def __init__(self, encoder: nn.Module, hidden_size):
        super(Reranker, self).__init__()
        self.encoder = encoder
        self.qa_classifier = nn.Linear(hidden_size, 1)

    def forward(self, input_ids, attention_mask):
        # notations: N - number of questions in a batch, M - number of passages per questions, L - sequence length
        N, M, L = input_ids.size()
        relevance_logits = self._forward(input_ids.view(N * M, L), attention_mask.view(N * M, L))
        return relevance_logits.view(N, M)

    def _forward(self, input_ids, attention_mask):
        cls_vec = self.encoder.embed(input_ids, attention_mask)
        return ops.linear_f32(cls_vec, self.qa_classifier.weight, self.qa_classifier.bias)

    def zero_grad(self, set_to_none=False):
        self.encoder.zero_grad()
        for p in self.qa_classifier.parameters():
            if p.grad is not None:
                p.grad.zero_()


class BiEncoderNllLoss(object):
    """SimANS/model/models.py:468-514."""

    def calc(self, q_vectors, ctx_vectors, positive_idx_per_question: list, hard_negative_idx_per_question: list = None,
             loss_scale: float = None, local_q=None, local_ctx=None):
        """-> (loss, correct_predictions_count).  ``local_q`` / ``local_ctx`` = (row_offset, rows): the rows of
        the rank-ordered global concatenation that belong to this rank and carry gradient
        (PROD/ProD_base/train_DE_model_marco.py:251-264); default: all rows."""
        loss, correct = ops.inbatch_nll_loss(q_vectors, ctx_vectors, positive_idx_per_question, loss_scale,
                                             local_q, local_ctx)
        return loss, correct.to(torch.long)

    @staticmethod
    def get_scores(q_vector, ctx_vectors):
        f = BiEncoderNllLoss.get_similarity_function()
        return f(q_vector, ctx_vectors)

    @staticmethod
    def get_similarity_function():
        return dot_product_scores


class CrossBERTKDLoss(object):
    """PROD/ProD_KD/model/models.py:668-760 (L3): CE_WEIGHT * NLL(log_softmax(sim), 0) + KD_WEIGHT * kd_loss(sim, teacher)
    on the per-query block similarity; KD_type "KD_softmax" (the shipped configuration)."""

    def calc(self, args, q_vectors, ctx_vectors, relevance_logits, hard_negative_idx_per_question: list = None,
             loss_scale: float = None, LwF=False, ori_q_vector=None, ori_ctx_vectors=None):
        if getattr(args, "KD_type", "KD_softmax") != "KD_softmax":
            raise NotImplementedError("KD_type %r: only KD_softmax runs on the HIP path" % args.KD_type)
        loss, correct, _, _ = ops.cross_kd_loss(q_vectors, ctx_vectors, relevance_logits, args.TEMPERATURE,
                                                args.CE_WEIGHT, args.KD_WEIGHT)
        if LwF:                                      # + LwF_WEIGHT * kd_loss(scores, scores of the frozen student) (:687-690, 748-750)
            ori_scores = ops.block_scores(ori_q_vector, ori_ctx_vectors)
            lwf, _, _, _ = ops.cross_kd_loss(q_vectors, ctx_vectors, ori_scores, args.TEMPERATURE, 0.0, args.LwF_WEIGHT)
            loss = loss + lwf
        if loss_scale:
            loss = loss * loss_scale
        return loss, correct.to(torch.long)


class BiEncoderKDLoss(object):
    """PROD/ProD_KD/model/models.py:970-1038 (L4): the same loss on all-pairs scores of the student embeddings against
    all-pairs scores of the teacher (dual-encoder) embeddings."""

    def calc(self, args, q_vectors, ctx_vectors, teacher_q_vector, teacher_ctxs_vector, positive_idx_per_question: list,
             hard_negative_idx_per_question: list = None, loss_scale: float = None, local_q=None, local_ctx=None):
        if teacher_q_vector is None or teacher_ctxs_vector is None:
            loss, correct = ops.inbatch_nll_loss(q_vectors, ctx_vectors, positive_idx_per_question, loss_scale, local_q, local_ctx)
            return loss, correct.to(torch.long)
        if getattr(args, "KD_type", "KD_softmax") != "KD_softmax":
            raise NotImplementedError("KD_type %r: only KD_softmax runs on the HIP path" % args.KD_type)
        loss, _, _, correct = ops.bi_kd_loss(q_vectors, ctx_vectors, teacher_q_vector, teacher_ctxs_vector,
                                             positive_idx_per_question, args.TEMPERATURE, args.CE_WEIGHT, args.KD_WEIGHT,
                                             loss_scale, local_q, local_ctx)
        return loss, correct.to(torch.long)

    @staticmethod
    def get_scores(q_vector, ctx_vectors):
        return dot_product_scores(q_vector, ctx_vectors)

    @staticmethod
    def get_similarity_function():
        return dot_product_scores


def dot_product_scores(q_vectors, ctx_vectors):
    """SimANS/model/models.py:564-572: q->ctx scores for every row in ctx_vector."""
    return ops.dot_product_scores(q_vectors, ctx_vectors)
