"""Drop-in module API of PROD/ProD_KD/model/models.py for the cross-encoder -> dual-encoder distillation step
(BASELINE configs[3]: 12-layer cross-encoder teacher -> 6-layer bi-encoder student, PROD/README.md:208-224).

Same class names, constructor arguments and return conventions as the reference; the arithmetic runs in
libsimx_hip.so.  What differs from the SimANS classes (simxns_amd/model/models.py):
  * HFBertEncoder.init_encoder(args, role, ..., number_layers): the number of layers is a ROLE property
    (args.student_num_hidden_layers / args.teacher_num_hidden_layers, models.py:52-64) -- a 6-layer student is the first
    6 layers of the 12-layer checkpoint, which is what loading a deeper state_dict non-strictly gives;
  * BiBertEncoder(args, role) (models.py:208-256);
  * Reranker carries a second head `binary = Linear(H, 2)` and returns (binary_logits [N,M,2], relevance_logits [N,M],
    None) (models.py:1232-1254);
  * CrossBERTKDLoss / BiEncoderKDLoss / BiEncoderNllLoss are the loss classes of the same file (re-exported).
Out of scope: ColBERT, DistilBERT students, the KD_logit / DKD loss variants.
"""
import torch
from torch import nn

from ... import ops
from ...model.models import (HFBertEncoder as _SimansHFBertEncoder, BiBertEncoder as _SimansBiBertEncoder,
                             CrossBERTKDLoss, BiEncoderKDLoss, BiEncoderNllLoss, dot_product_scores)  # noqa: F401


class HFBertEncoder(_SimansHFBertEncoder):
    """PROD/ProD_KD/model/models.py:13-81."""

    @classmethod
    def init_encoder(cls, args, role=None, dropout: float = 0.1, model_type=None, number_layers=0, compute_dtype=None):
        if model_type is None:
            if role == 'student':
                model_type = args.model_type
            elif role == 'teacher':
                model_type = args.teacher_model_type
            elif role == 'double_teacher':
                model_type = args.double_teacher_pretrain
            else:
                raise ValueError("no such type role: %r" % (role,))
        if role is None:
            layers = number_layers
        elif role == 'student':
            layers = args.student_num_hidden_layers
        elif role == 'teacher':
            layers = args.teacher_num_hidden_layers
        elif role == 'double_teacher':
            layers = args.double_teacher_num_hidden_layers
        else:
            raise ValueError("no such type role: %r" % (role,))
        return super(HFBertEncoder, cls).init_encoder(args, dropout=dropout, model_type=model_type,
                                                      compute_dtype=compute_dtype, num_hidden_layers=layers)


class BiBertEncoder(_SimansBiBertEncoder):
    """PROD/ProD_KD/model/models.py:208-256 (forward / query_emb / body_emb are the SimANS ones)."""

    def __init__(self, args, role=None):
        nn.Module.__init__(self)
        if role is not None:
            if role == 'student' and getattr(args, "model_type", None) == 'distilbert-base-uncased':
                raise NotImplementedError("DistilBERT students are not on the MI355X path")
            self.role = role
            mk = lambda: HFBertEncoder.init_encoder(args, role)
        else:
            mk = lambda: HFBertEncoder.init_encoder(args, model_type='nghuyong/ernie-2.0-base-en', number_layers=12)
        self.question_model = mk()
        self.ctx_model = self.question_model if getattr(args, 'share_weight', False) else mk()


class Reranker(nn.Module):
    """PROD/ProD_KD/model/models.py:1232-1254: cross-encoder with a binary head and a ranking head on [CLS]."""

    def __init__(self, encoder: nn.Module, hidden_size):
        super(Reranker, self).__init__()
        self.encoder = encoder
        self.binary = nn.Linear(hidden_size, 2)
        self.qa_classifier = nn.Linear(hidden_size, 1)
        with torch.no_grad():                      # init_weights (models.py:300-308): N(0, 0.02), zero bias
            for m in (self.binary, self.qa_classifier):
                m.weight.normal_(mean=0.0, std=0.02)
                m.bias.zero_()

    def forward(self, input_ids, attention_mask):
        N, M, L = input_ids.size()
        binary_logits, relevance_logits, _ = self._forward(input_ids.view(N * M, L), attention_mask.view(N * M, L))
        return binary_logits.view(N, M, 2), relevance_logits.view(N, M), None

    def _forward(self, input_ids, attention_mask):
        cls_vec = self.encoder.embed(input_ids, attention_mask)
        binary_logits = ops.linear_f32(cls_vec, self.binary.weight, self.binary.bias)
        rank_logits = ops.linear_f32(cls_vec, self.qa_classifier.weight, self.qa_classifier.bias)
        return binary_logits, rank_logits, None

    def zero_grad(self, set_to_none=False):
        self.encoder.zero_grad()
        for p in list(self.binary.parameters()) + list(self.qa_classifier.parameters()):
            if p.grad is not None:
                p.grad.zero_()
