"""SIMX_DETERMINISTIC=1: two runs of the same training steps in two fresh processes leave bit-identical gradients, losses and
weights; the ordered reductions agree with the default (atomic) ones to f32 rounding.  The switch is read once per process,
hence the worker subprocesses."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(dtype, out_path):
    import torch
    from simxns_amd import _lib, ops
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import BiBertEncoder, HFBertEncoder
    from simxns_amd.optim import FusedAdamW
    dev = torch.device("cuda:0")
    cfg = BertConfigLite(vocab_size=300, hidden_size=256, num_hidden_layers=2, num_attention_heads=4, intermediate_size=1024,
                         max_position_embeddings=64, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model, bi.ctx_model = HFBertEncoder(cfg, dtype), HFBertEncoder(cfg, dtype)
    bi.to(dev)
    bi.train()
    B, N = 16, 7
    g = torch.Generator().manual_seed(1)
    # many repeated ids (the embedding scatter's contended rows) and ragged lengths
    q_ids = torch.randint(1, 40, (B, 24), generator=g).to(dev)
    c_ids = torch.randint(1, 300, (B * (1 + N), 48), generator=g).to(dev)
    qm, cm = torch.ones_like(q_ids), torch.ones_like(c_ids)
    for i in range(B):
        qm[i, 8 + i:] = 0
    for i in range(B * (1 + N)):
        cm[i, 16 + (i * 5) % 32:] = 0
    z = torch.linspace(-2, 2, B * (1 + N)).reshape(B, 1 + N).to(dev)
    opt = FusedAdamW(bi, lr=1e-3)
    rec = {"det": int(_lib.load().simx_deterministic()), "loss": [], "grad_sha": [], "w_sha": []}
    grads = None
    for step in range(3):
        bi.zero_grad()
        q, c = bi(q_ids, qm, c_ids, cm)
        loss, _, _ = ops.kl_distill_loss(q, c, z, 1.0, False, 1)
        loss.backward()
        grads = torch.cat([p.grad.reshape(-1) for p in bi.parameters()]).float().cpu().numpy()
        rec["loss"].append(float(loss.item()).hex())
        rec["grad_sha"].append(hashlib.sha256(grads.tobytes()).hexdigest())
        opt.step(max_grad_norm=2.0)
        w = torch.cat([e.engine.flat.reshape(-1) for e in (bi.question_model, bi.ctx_model)]).cpu().numpy()
        rec["w_sha"].append(hashlib.sha256(w.tobytes()).hexdigest())
    np.save(out_path + ".npy", grads)
    with open(out_path, "w") as f:
        json.dump(rec, f)


def _run(tmp_path, tag, dtype, det):
    out = str(tmp_path / ("%s.json" % tag))
    env = dict(os.environ)
    env["SIMX_DETERMINISTIC"] = "1" if det else "0"
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    code = "import sys; sys.path.insert(0, %r); from tests.test_det_gpu import _worker; _worker(%r, %r)" % (ROOT, dtype, out)
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    with open(out) as f:
        return json.load(f), np.load(out + ".npy")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_two_runs_are_bit_identical(tmp_path, dtype):
    a, ga = _run(tmp_path, "a", dtype, True)
    b, gb = _run(tmp_path, "b", dtype, True)
    assert a["det"] == 1 and b["det"] == 1
    assert a["loss"] == b["loss"], (a["loss"], b["loss"])
    assert a["grad_sha"] == b["grad_sha"]
    assert a["w_sha"] == b["w_sha"]
    assert np.array_equal(ga, gb)
    # the ordered sums are the same sums: against the default build's atomics only the rounding order differs
    c, gc = _run(tmp_path, "c", dtype, False)
    assert c["det"] == 0
    l_det, l_def = float.fromhex(a["loss"][0]), float.fromhex(c["loss"][0])
    assert abs(l_det - l_def) <= 1e-5 * max(1.0, abs(l_def))
    cos = float(np.dot(ga.astype(np.float64), gc.astype(np.float64)) / (np.linalg.norm(ga.astype(np.float64)) * np.linalg.norm(gc.astype(np.float64))))
    assert cos >= 0.999, cos                    # third step of two trajectories that differ by rounding only
