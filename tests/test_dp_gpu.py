"""Data-parallel step on the GPU (C1 + C2): two ranks -- two processes sharing cuda:0, gloo rendezvous on
127.0.0.1 (a 1-GPU box cannot host two RCCL ranks; the collectives' arithmetic is the same) -- each run their half of a
global batch through the SAME code path the 8-GPU job uses:
    BiBertEncoder fwd -> KL-distill loss + 0.2 * in-batch NLL over the all-gathered embeddings (local-slot gradient)
    -> backward in two layer ranges with the gradient slices all-reduced asynchronously by FusedAdamW's hooks
    -> FusedAdamW.step(world_size=2)  (clip 2.0, AdamW, 1/W folded into the update)
and must reproduce the single-rank result on the global batch (the reference's DistributedDataParallel averaging,
SimANS/co_training/co_training_marco_train.py:107-114, and the gather semantics of
PROD/ProD_base/train_DE_model_marco.py:224-278): mean over ranks of the local gradients
 = grad of [ KL over the global batch + (0.2 / W) * NLL_global ]."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["SIMX_ROOT"])
from simxns_amd import ops, parallel
from simxns_amd.engine import BertConfigLite
from simxns_amd.model.models import BiBertEncoder, HFBertEncoder
from simxns_amd.optim import FusedAdamW
from simxns_amd.utils import synth

W, BQ, D = 2, 3, 4                      # ranks, queries per rank, passages per query
def build(dev, dtype):
    cfg = BertConfigLite(vocab_size=1000, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                         max_position_embeddings=192, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model, bi.ctx_model = HFBertEncoder(cfg, dtype), HFBertEncoder(cfg, dtype)
    for m, seed in ((bi.question_model, 71), (bi.ctx_model, 72)):
        m.load_numpy_state(synth.fill_bert_state_dict([(k, tuple(p.shape)) for k, p in m.named_parameters()], seed, std=0.08))
    return bi.to(dev)
def batch(dev):
    q_ids, q_mask, _ = synth.make_batch(81, W * BQ, 32, 1000, 9, 3, 4)
    c_ids, c_mask, _ = synth.make_batch(82, W * BQ * D, 128, 1000, 80, 25, 16)
    z = np.random.RandomState(5).randn(W * BQ, D).astype(np.float32) * 2
    t = lambda a: torch.from_numpy(a).to(dev)
    return t(q_ids), t(q_mask), t(c_ids), t(c_mask), t(z)
def flat_state(bi):
    return torch.cat([bi.question_model.engine.flat, bi.ctx_model.engine.flat]).detach().cpu().numpy()
'''

WORKER = COMMON + r'''
import torch.distributed as dist
rank = int(os.environ["RANK"])
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("gloo")
dtype = os.environ["SIMX_TEST_DTYPE"]
bi = build(dev, dtype).train()
opt = FusedAdamW(bi, lr=1e-3, eps=1e-8).enable_overlap(W, parts=2)
q_ids, q_mask, c_ids, c_mask, z = batch(dev)
qs, cs = slice(rank * BQ, (rank + 1) * BQ), slice(rank * BQ * D, (rank + 1) * BQ * D)
fired = []
for m in (bi.question_model, bi.ctx_model):
    hook = m.engine.grad_ready_hook
    m.engine.grad_ready_hook = (lambda e, lo, hi, hook=hook: (fired.append((lo, hi)), hook(e, lo, hi)))
q, c = bi(q_ids[qs], q_mask[qs], c_ids[cs], c_mask[cs])
loss, _, _ = ops.kl_distill_loss(q, c, z[qs])
loss = loss + 0.2 * parallel.inbatch_nll_allgather(q, c, D)
loss.backward()
assert len(fired) == 4 and all(hi > lo for lo, hi in fired), fired            # two layer ranges per tower
assert len(opt._pending) == 2                                                   # slices already in flight before step()
scale = opt.sync_grads()
torch.cuda.synchronize()
g = torch.cat([bi.question_model.engine.flat_grad, bi.ctx_model.engine.flat_grad]).cpu().numpy() * scale
opt.step(max_grad_norm=2.0, world_size=W)
torch.cuda.synchronize()
np.savez(os.path.join(os.environ["SIMX_OUT"], "rank%d.npz" % rank), grad=g, params=flat_state(bi), loss=loss.item())
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_two_rank_step_equals_single_rank_global_batch(dev, tmp_path, dtype):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, SIMX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", SIMX_OUT=str(tmp_path),
               SIMX_TEST_DTYPE=dtype, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK="0"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o
    # single rank, global batch
    ns = {}
    os.environ["SIMX_ROOT"] = ROOT
    exec(COMMON, ns)
    from simxns_amd import ops
    from simxns_amd.optim import FusedAdamW
    W, D = ns["W"], ns["D"]
    bi = ns["build"](dev, dtype).train()
    opt = FusedAdamW(bi, lr=1e-3, eps=1e-8)
    q_ids, q_mask, c_ids, c_mask, z = ns["batch"](dev)
    q, c = bi(q_ids, q_mask, c_ids, c_mask)
    loss, _, _ = ops.kl_distill_loss(q, c, z)
    nll, _ = ops.inbatch_nll_loss(q, c, [i * D for i in range(q.shape[0])])
    (loss + (0.2 / W) * nll).backward()
    torch.cuda.synchronize()
    g = torch.cat([bi.question_model.engine.flat_grad, bi.ctx_model.engine.flat_grad]).cpu().numpy()
    p0 = ns["flat_state"](bi)
    opt.step(max_grad_norm=2.0)
    torch.cuda.synchronize()
    p1 = ns["flat_state"](bi)
    R = [np.load(str(tmp_path / ("rank%d.npz" % r))) for r in range(2)]
    assert np.array_equal(R[0]["grad"], R[1]["grad"]) and np.array_equal(R[0]["params"], R[1]["params"]), "ranks diverged"
    # both ranks see the same global NLL term; their KL terms average to the global KL
    want_loss = loss.item() + 0.2 * nll.item()
    got_loss = 0.5 * (R[0]["loss"] + R[1]["loss"])
    tol = 2e-5 if dtype == "fp32" else 2e-2
    assert abs(got_loss - want_loss) <= tol * max(1.0, abs(want_loss))
    err = np.abs(R[0]["grad"] - g).max() / np.abs(g).max()
    assert err <= (5e-5 if dtype == "fp32" else 5e-2), "averaged DP gradient vs global-batch gradient: rel-to-max err %.3e" % err
    # the update: same direction everywhere the gradient is not at the noise floor (Adam's first step is lr * g/(|g|+eps))
    d_dp, d_1 = R[0]["params"] - p0, p1 - p0
    big = np.abs(g) > 1e-3 * np.abs(g).max()
    assert np.abs(d_dp[big] - d_1[big]).max() <= (0.02 if dtype == "fp32" else 0.5) * 1e-3
    assert np.abs(d_1).max() > 0.5e-3
