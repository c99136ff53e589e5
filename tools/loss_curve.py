"""Training-trajectory check on the GPU (not part of bench.py's metric): the SAME retriever job -- two BERT-base towers,
frozen cross-encoder teacher, SimANS draw, KL-distill loss, clip 2.0 + AdamW + warm-up, dropout 0.1 -- run for N optimiser
steps in the fp16 (apex-O1 form, dynamic loss scale), bf16 and fp32 engines from identical weights, data and dropout masks.
Prints the loss curves, their distances from the fp32 curve and the fp16 engine's scaler state; the committed outputs are
profiles/r02_loss_curve.json (bf16 / fp32), profiles/r03_loss_curve.json (all three) and profiles/r04_loss_curve.json (+ the exact-f32
engine as the reference curve: the fp32 engine runs on 16-bit plane pairs since round 4).
usage: python tools/loss_curve.py [steps=60] [B=32] [N=15]"""
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from simxns_amd import ops                                                     # noqa: E402
from simxns_amd.engine import BertConfigLite                                   # noqa: E402
from simxns_amd.model.models import HFBertEncoder, BiBertEncoder, Reranker     # noqa: E402
from simxns_amd.optim import FusedAdamW, LinearWarmupSchedule                  # noqa: E402
from simxns_amd.utils import synth                                             # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
N = int(sys.argv[3]) if len(sys.argv) > 3 else 15
Cn, QL, PL, CL = 64, 32, 128, 160
dev = torch.device("cuda:0")
cfg = BertConfigLite(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)


def run(dtype):
    torch.manual_seed(1234)

    def tower():
        return HFBertEncoder(cfg, compute_dtype=dtype)
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model, bi.ctx_model = tower(), tower()
    teacher = Reranker(tower(), cfg.hidden_size)
    bi.to(dev).train()
    teacher.to(dev).eval()
    for m in (bi.question_model, bi.ctx_model, teacher.encoder):
        m.engine.dropout_seed = 7                       # same stateless masks in both engines
    opt = FusedAdamW(bi, lr=2e-5, eps=1e-8)
    sch = LinearWarmupSchedule(opt, 10, 10 * steps, last_step=1)
    q_ids, _, q_lens = synth.make_batch(100, B, QL, cfg.vocab_size, 9, 3, 4, full=False)
    p_ids, _, p_lens = synth.make_batch(200, B * (1 + Cn), PL, cfg.vocab_size, 80, 25, 16, full=False)
    pool_q = torch.from_numpy(q_ids.astype(np.int32)).to(dev)
    pool_p = torch.from_numpy(p_ids.astype(np.int32)).to(dev)
    q_rows = torch.arange(B, dtype=torch.int32, device=dev)
    row_base = (torch.arange(B, device=dev) * (1 + Cn)).unsqueeze(1)
    rs = np.random.RandomState(7)
    s_pos = 70.0 + 20.0 * rs.rand(B)
    scores = np.sort(s_pos[:, None] - np.abs(rs.randn(B, Cn)) * 1.5, axis=1)[:, ::-1].copy()
    d_scores, d_spos = torch.from_numpy(scores).to(dev), torch.from_numpy(s_pos).to(dev)
    zero_col = torch.zeros(B, 1, dtype=torch.long, device=dev)
    losses = []
    for it in range(steps):
        neg = ops.simans_sample(d_scores, d_spos, N, form=ops.LAPLACE, tau=3.0, seed=42, offset=1 + it % 4)   # 4 batches, revisited
        sel = torch.cat([zero_col, neg.long() + 1], dim=1)
        batch = ops.assemble_batch(pool_q, pool_p, q_rows, (row_base + sel).to(torch.int32), 1 + N, pad_id=0, sep_id=102, ce_len=CL)
        q_i, q_m, c_i, c_m, _ = batch["student"]
        q, c = bi(q_i, q_m, c_i, c_m)
        with torch.no_grad():
            z = teacher(batch["teacher"][0], batch["teacher"][1])
        loss, _, _ = ops.kl_distill_loss(q, c, z, 1.0, False, 1)
        loss.backward()
        opt.step(max_grad_norm=2.0, world_size=1)
        sch.step()
        losses.append(float(loss.item()))
    scaler = opt.scaler.snapshot() if getattr(opt, "scaler", None) is not None else None
    del bi, teacher, opt
    torch.cuda.empty_cache()
    return losses, scaler


curves, scalers = {}, {}
ENGINES = ("fp16", "bf16", "fp32", "fp32_exact")        # fp32_exact: exact f32 products everywhere -- the curve the others are measured against
for dt in ENGINES:
    curves[dt], scalers[dt] = run(dt)
ref = np.array(curves["fp32_exact"])
out = {"job": "retriever step x %d, B=%d, %d negatives of %d candidates, 4 batches revisited, lr 2e-5 (warm-up 10), dropout 0.1, "
              "clip 2.0; identical weights / data / dropout masks in all engines" % (steps, B, N, Cn),
       "fp16_loss_scaler": scalers["fp16"]}
for dt in ENGINES:
    c = np.array(curves[dt])
    d = np.abs(c - ref)
    out["loss_" + dt] = [round(float(x), 4) for x in c]
    out["summary_" + dt] = {"first": round(float(c[0]), 4), "last5_mean": round(float(c[-5:].mean()), 4),
                            "max_abs_diff_vs_fp32_exact": round(float(d.max()), 4), "mean_abs_diff_vs_fp32_exact": round(float(d.mean()), 4),
                            "rel_diff_of_last5_mean_vs_fp32_exact": round(float(abs(c[-5:].mean() - ref[-5:].mean()) / abs(ref[-5:].mean())), 4)}
print(json.dumps(out))
