"""How far is the REFERENCE'S OWN 16-bit mode from its fp64 arithmetic?  (test infrastructure, build container only.)

The reference's optional --fp16 mode is apex.amp O1 (SimANS/co_training/co_training_marco_train.py:97-104, 218-220): the
"FP16_FUNCS" (nn.functional.linear, torch.matmul / bmm -- every GEMM of BERT) run on fp16-cast inputs and return fp16,
everything else (LayerNorm, softmax, GELU, the residual additions by type promotion, the loss) runs in fp32, master weights
are fp32, the loss is scaled.  apex is not installed here and fp16 GEMMs on this container's CPU are not practical, so this
script EMULATES that arithmetic on the imported reference modules: F.linear / torch.matmul are wrapped to round their
inputs and their result to fp16 (products accumulated in fp32, as the matrix cores do), gradients round to fp16 on the same
edges through autograd, the loss is scaled by 2^10 and the gradients unscaled.  It then measures that run against the
committed fp64 golden (tests/golden/step_base_hot.npz) with the SAME error function the product's fp16 engine is measured
with (simxns_amd/utils/parity.py::golden_errors) -- the yardstick for "at least as wide as the reference's own 16-bit mode".

    python -m oracle.o1_emulation            # -> profiles/r03_o1_emulation.json   (a few minutes of CPU)
"""
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

from . import make_golden as MG
from .weights import BertCfg, make_bert_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _r16(t):
    return t.half().float() if t.dtype == torch.float32 else t


class o1_autocast(object):
    """apex O1's cast lists, restricted to what BERT uses: linear / matmul on fp16-rounded operands with an fp16-rounded result."""

    def __enter__(self):
        import torch.nn.functional as F
        self.F, self.lin, self.mm = F, F.linear, torch.matmul
        lin, mm = self.lin, self.mm
        F.linear = lambda x, w, b=None: _r16(lin(_r16(x), _r16(w), None if b is None else _r16(b)))
        torch.matmul = lambda a, b: _r16(mm(_r16(a), _r16(b)))
        return self

    def __exit__(self, *a):
        self.F.linear, torch.matmul = self.lin, self.mm
        return False


def main():
    torch.manual_seed(0)
    torch.set_num_threads(int(os.environ.get("SIMX_CPU_THREADS", "16")))
    MG.RM = MG._ref_models()
    RM = MG.RM
    G = np.load(os.path.join(ROOT, "tests", "golden", "step_base_hot.npz"))
    cfg = BertCfg(**json.loads(str(G["cfg"])) if False else {})          # the hot fixture is BERT-base (default BertCfg)
    seeds, std = [int(s) for s in G["seeds"]], float(G["std"])
    Pq, Pc, Pt = (make_bert_params(cfg, s, std=std) for s in seeds)
    tt = lambda k: torch.from_numpy(G[k])
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        args = types.SimpleNamespace(model_type=MG._hf_dir(tmp, cfg, Pq, "o1_q"), gradient_checkpointing=True, share_weight=False)
        model = RM.BiBertEncoder(args)
        model.ctx_model.load_state_dict({k: torch.from_numpy(v) for k, v in Pc.items()}, strict=False)
        MG._no_dropout(model)
        teacher = RM.Reranker(RM.HFBertEncoder.init_encoder(types.SimpleNamespace(gradient_checkpointing=False),
                                                            model_type=MG._hf_dir(tmp, cfg, Pt, "o1_t")), cfg.hidden)
        with torch.no_grad():
            teacher.qa_classifier.weight.copy_(tt("qa_w"))
            teacher.qa_classifier.bias.copy_(tt("qa_b"))
        MG._no_dropout(teacher)
    model.train()                                  # (HF checkpoints only in training mode; every Dropout has p = 0)
    S = 1024.0
    with o1_autocast():
        model.zero_grad()
        q, c = model(query_ids=tt("q_ids"), attention_mask_q=tt("q_mask"), input_ids_a=tt("c_ids"), attention_mask_a=tt("c_mask"))
        sim = torch.einsum("bh,bdh->bd", q.float(), c.float().reshape(q.size(0), c.size(0) // q.size(0), -1))
        p_s = torch.nn.functional.softmax(sim, dim=1)
        with torch.no_grad():
            z = teacher(input_ids=tt("t_ids"), attention_mask=tt("t_mask")).float()
            p_t = torch.nn.functional.softmax(z, dim=1)
        loss = torch.nn.KLDivLoss(reduction="batchmean")((p_s + 1e-7).log(), p_t)
        (loss * S).backward()
    grads = {k: p.grad.detach().numpy().astype(np.float64) / S for k, p in model.named_parameters() if p.grad is not None}
    R = dict(q=q.detach().numpy(), c=c.detach().numpy(), z=z.numpy(), sim=sim.detach().numpy(), loss=float(loss.item()), grads=grads)
    sys.path.insert(0, ROOT)
    from simxns_amd.utils.parity import golden_errors
    e = golden_errors(R, G)
    out = {"what": "imported reference (SimANS modules) under an emulation of apex.amp O1 -- fp16-rounded linear / matmul I/O, fp32 "
                   "accumulation, everything else fp32, loss scale 2^10 -- against its own fp64 run (tests/golden/step_base_hot.npz)",
           "errors": e, "seconds": round(time.time() - t0, 1),
           "summary": {"logits_rel_err": e["sim_abs"] / e["sim_scale"], "embeddings_max_abs_err": max(e["q_abs"], e["c_abs"]),
                       "teacher_logits_max_abs_err": e["z_abs"], "loss_abs_err": e["loss_abs"], "grad_norm_rel_err_median": e["gnorm_rel_median"],
                       "grad_norm_rel_err_max": e["gnorm_rel_max"], "grad_slice_cosine_min_dense_weights": e["gslice_cos_min"],
                       "grad_slice_cosine_median_all_tensors": e["gslice_cos_median_all"]}}
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r03_o1_emulation.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)
    print(json.dumps(out["summary"], indent=1))


if __name__ == "__main__":
    main()
