"""Build-container check of the CPU baseline: times the IMPORTED reference's retriever step
(SimANS BiBertEncoder on HF BertModel, fp32, eval() with grads, the literal step body of
co_training_marco_train.py:198-217) and oracle/torch_cpu.py's restatement of it on the same shapes and thread count, and
prints both.  The restatement must be within ~10 % of the reference.  Needs /root/reference.

    python -m oracle.time_reference [--threads 8] [--config 1|2r]
"""
import argparse
import json
import tempfile
import time
import types

import numpy as np
import torch

from . import make_golden as MG
from . import torch_cpu as TC
from .weights import BertCfg, make_batch, make_bert_params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--config", default="1", choices=["1", "2r"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    B, N = (4, 1) if a.config == "1" else (8, 15)
    MG.RM = MG._ref_models()
    cfg = BertCfg()
    with tempfile.TemporaryDirectory() as tmp:
        Pq, Pc = make_bert_params(cfg, 1, perturb=False), make_bert_params(cfg, 2, perturb=False)
        args = types.SimpleNamespace(model_type=MG._hf_dir(tmp, cfg, Pq, "q"), gradient_checkpointing=False, share_weight=False)
        model = MG.RM.BiBertEncoder(args)
        model.ctx_model.load_state_dict({k: torch.from_numpy(v) for k, v in Pc.items()}, strict=False)
        MG._no_dropout(model)
    P = B * (1 + N)
    q_ids, q_mask, _ = make_batch(1, B, 32, cfg.vocab, 9, 3, 4, full=True)
    c_ids, c_mask, _ = make_batch(2, P, 128, cfg.vocab, 80, 25, 16, full=True)
    z = torch.from_numpy(np.linspace(-2, 2, P).reshape(B, 1 + N).astype(np.float32))
    tt = lambda x: torch.from_numpy(x)

    def ref_step():
        model.zero_grad()
        q, c = model(query_ids=tt(q_ids), attention_mask_q=tt(q_mask), input_ids_a=tt(c_ids), attention_mask_a=tt(c_mask))
        sim = torch.einsum("bh,bdh->bd", q, c.reshape(q.size(0), c.size(0) // q.size(0), -1))
        loss = torch.nn.KLDivLoss(reduction="batchmean")((torch.softmax(sim, 1) + 1e-7).log(), torch.softmax(z, 1))
        loss.backward()
        return float(loss.item())
    for _ in range(a.warmup):
        lref = ref_step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ref_step()
    t_ref = (time.perf_counter() - t0) / a.steps
    t_port, _, thr = TC.time_step(B, N, with_teacher=False, warmup=a.warmup, steps=a.steps, threads=a.threads)
    # same weights, same inputs -> same loss
    Tq, Tc = TC.to_torch_params(Pq), TC.to_torch_params(Pc)
    r = TC.retriever_step(Tq, Tc, tt(q_ids), tt(q_mask), tt(c_ids), tt(c_mask), cfg.heads, teacher_logits=z)
    print(json.dumps({"config": a.config, "B": B, "N": N, "threads": thr, "reference_s_per_step": round(t_ref, 4),
                      "restatement_s_per_step": round(t_port, 4), "ratio": round(t_port / t_ref, 3),
                      "reference_pairs_per_s": round(P / t_ref, 2), "restatement_pairs_per_s": round(P / t_port, 2),
                      "loss_reference": lref, "loss_restatement": r["loss"]}))


if __name__ == "__main__":
    main()
