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
opt = FusedAdamW(bi, lr=1e-3, eps=1e-8).enable_overlap(W, parts=2, payload=os.environ.get("SIMX_TEST_PAYLOAD", "fp32"))
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
assert len(opt._pending) == 4                                                   # slices already in flight before step()
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


@pytest.mark.parametrize("dtype,payload", [("fp32", "fp32"), ("bf16", "fp32"), ("fp16", "fp32"), ("fp32", "bf16")])
def test_two_rank_step_equals_single_rank_global_batch(dev, tmp_path, dtype, payload):
    script = tmp_path / "w.py"
    script.write_text(WORKER)
    env = dict(os.environ, SIMX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2", SIMX_OUT=str(tmp_path),
               SIMX_TEST_DTYPE=dtype, SIMX_TEST_PAYLOAD=payload, SIMX_LOSS_SCALE_INIT="1024", HSA_ENABLE_IPC_MODE_LEGACY="0")
    os.environ["SIMX_LOSS_SCALE_INIT"] = "1024"
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
    tol = {"fp32": 2e-5, "fp16": 3e-3, "bf16": 2e-2}[dtype]
    assert abs(got_loss - want_loss) <= tol * max(1.0, abs(want_loss))
    err = np.abs(R[0]["grad"] - g).max() / np.abs(g).max()
    # (a bf16 payload rounds each rank's slice to 8 bits before the sum: 2^-9 relative per element)
    gtol = {"fp32": 5e-5, "fp16": 8e-3, "bf16": 5e-2}[dtype] if payload == "fp32" else 6e-3
    assert err <= gtol, "averaged DP gradient vs global-batch gradient: rel-to-max err %.3e" % err
    # the update: same direction everywhere the gradient is not at the noise floor (Adam's first step is lr * g/(|g|+eps))
    d_dp, d_1 = R[0]["params"] - p0, p1 - p0
    big = np.abs(g) > 1e-3 * np.abs(g).max()
    assert np.abs(d_dp[big] - d_1[big]).max() <= (0.02 if dtype == "fp32" and payload == "fp32" else 0.5) * 1e-3
    os.environ.pop("SIMX_LOSS_SCALE_INIT", None)
    assert np.abs(d_1).max() > 0.5e-3


RCCL1 = COMMON + r'''
import torch.distributed as dist
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)          # RCCL, one rank
from simxns_amd import retrieval
assert parallel.FORCE_COLLECTIVES
dtype = os.environ["SIMX_TEST_DTYPE"]
q_ids, q_mask, c_ids, c_mask, z = batch(dev)
res = {}
for mode in ("plain", "rccl", "rccl_bf16"):
    bi = build(dev, dtype).train()
    opt = FusedAdamW(bi, lr=1e-3, eps=1e-8)
    fired = []
    if mode != "plain":
        opt.enable_overlap(1, parts=2, force=True, payload="bf16" if mode == "rccl_bf16" else "fp32")
        opt.profile_comm = True
        for m in (bi.question_model, bi.ctx_model):
            hook = m.engine.grad_ready_hook
            m.engine.grad_ready_hook = (lambda e, lo, hi, hook=hook: (fired.append((lo, hi)), hook(e, lo, hi)))
    parallel.FORCE_COLLECTIVES = mode != "plain"
    q, c = bi(q_ids, q_mask, c_ids, c_mask)
    loss, _, _ = ops.kl_distill_loss(q, c, z)
    loss = loss + 0.2 * parallel.inbatch_nll_allgather(q, c, D)     # all_gather_into_tensor through RCCL when forced
    loss.backward()
    if mode != "plain":
        assert len(fired) == 4 and len(opt._pending) == 4, (fired, len(opt._pending))   # async all-reduces in flight on the comm stream
    scale = opt.sync_grads()
    torch.cuda.synchronize()
    g = torch.cat([bi.question_model.engine.flat_grad, bi.ctx_model.engine.flat_grad]).clone()
    opt.step(max_grad_norm=2.0)
    torch.cuda.synchronize()
    res[mode] = (loss.item(), g, torch.from_numpy(flat_state(bi)))
    if mode != "plain":
        st = opt.comm_stats()
        assert st and st["allreduce_bytes_per_step"] > 0 and st["overlapped_slices_per_step"] == 2.0, st
        print("comm", mode, st)
    # a second backward into already-reduced buffers must refuse (it would be reduced twice)
    if mode == "rccl":
        q, c = bi(q_ids, q_mask, c_ids, c_mask)
        l2, _, _ = ops.kl_distill_loss(q, c, z)
        l2.backward()
        q, c = bi(q_ids, q_mask, c_ids, c_mask)
        l3, _, _ = ops.kl_distill_loss(q, c, z)
        try:
            l3.backward()
            raise SystemExit("second backward after the slices were reduced did not raise")
        except Exception as e:
            assert "already all-reduced" in str(e), e
# a DROPPED step (armed backward, slices on the wire, then zero_grad() without step() -- e.g. a non-finite loss): zero_grad must
# wait for the pending collectives, reset the bookkeeping, and the next backward + step must equal a plain step
bi = build(dev, dtype).train()
opt = FusedAdamW(bi, lr=1e-3, eps=1e-8).enable_overlap(1, parts=2, force=True, payload="bf16")
parallel.FORCE_COLLECTIVES = True
q, c = bi(q_ids, q_mask, c_ids, c_mask)
(ops.kl_distill_loss(q, c, z)[0] * 3.0).backward()                 # a different gradient than the one that will count
assert len(opt._pending) == 4
bi.zero_grad()                                                     # HFBertEncoder.zero_grad -> FusedAdamW._discard_pending
assert not opt._pending and not opt._reduced and not any(m.engine._reduced_this_step for m in (bi.question_model, bi.ctx_model))
q, c = bi(q_ids, q_mask, c_ids, c_mask)
loss, _, _ = ops.kl_distill_loss(q, c, z)
loss = loss + 0.2 * parallel.inbatch_nll_allgather(q, c, D)
loss.backward()
opt.sync_grads()
torch.cuda.synchronize()
gd = torch.cat([bi.question_model.engine.flat_grad, bi.ctx_model.engine.flat_grad]).clone()
opt.step(max_grad_norm=2.0)
torch.cuda.synchronize()
rel_d = ((gd - res["plain"][1]).abs().max() / res["plain"][1].abs().max()).item()
assert rel_d <= 4e-3, ("gradient after a dropped step", rel_d)    # (bf16 payload: 2^-9 per element; stale slices would be O(1) off)
l0, g0, p0 = res["plain"]
l1, g1, p1 = res["rccl"]
# (equal up to the order of the f32 atomic sums in the LayerNorm / bias / embedding gradients, which differs run to run)
tolg = 1e-5 if dtype == "fp32" else 2e-3
assert abs(l0 - l1) <= 1e-6 * max(1.0, abs(l0)) and ((g0 - g1).abs().max() / g0.abs().max()).item() <= tolg, "one-rank RCCL step differs from the plain step"
assert torch.isfinite(g0).all() and (p0 - p1).abs().max().item() <= 2.5e-3          # (Adam's first step moves every weight by <= lr)
l2, g2, p2 = res["rccl_bf16"]
rel = ((g2 - g0).abs().max() / g0.abs().max()).item()
assert rel <= 4e-3, rel                                   # bf16 payload: 2^-9 per element
# the search's cross-shard merge through RCCL all_gather
Dm = torch.randn(5, 7, device=dev).sort(1, descending=True)[0].contiguous()
Im = torch.arange(35, device=dev, dtype=torch.int64).reshape(5, 7).contiguous()
D2, I2 = retrieval.merge_topk(Dm, Im, 7, None)
assert torch.equal(D2, Dm) and torch.equal(I2, Im)
dist.destroy_process_group()
print("rccl one-rank ok")
'''


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_rccl_one_rank_drives_the_whole_dp_path(dev, tmp_path, dtype):
    """backend="nccl" (RCCL) with one rank on the box's GPU and the overlap hooks forced on: the sliced asynchronous
    all-reduce on the communication stream, Work.wait(), the bf16 payload, all_gather_into_tensor of the [CLS] embeddings
    with its local-slot backward and the search's top-k merge all EXECUTE through RCCL; the result equals the plain step."""
    script = tmp_path / "r.py"
    script.write_text(RCCL1)
    env = dict(os.environ, SIMX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0",
               SIMX_TEST_DTYPE=dtype, SIMX_FORCE_COLLECTIVES="1", SIMX_LOSS_SCALE_INIT="1024", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = p.stdout.decode()
    assert p.returncode == 0 and "rccl one-rank ok" in out, out


@pytest.mark.parametrize("launcher", ["self", "torch.distributed.run"])
def test_bench_two_ranks_share_one_gpu(dev, launcher):
    """`bench.py --gpus 2` end to end, both ways it is launched: `python bench.py --gpus 2` (self-spawned ranks) and the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2`
    (the SCALE run at N = 2, 4, 8).  Rendezvous on
    127.0.0.1, sharded queries, overlapped gradient all-reduce with the default bf16 payload, replica check after the first step,
    barrier + max-over-ranks timing, ONE JSON line from rank 0 with a `comm` block.  SIMX_BENCH_SHARE_GPU=1 puts both ranks on
    cuda:0 over gloo (a 1-GPU box cannot host two RCCL ranks): the control flow is what is checked here, never a number."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SIMX_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("SIMX_GRAD_PAYLOAD", None)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable]
    if launcher != "self":
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)]
    r = subprocess.run(cmd + [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "8",
                              "--no-cpu-baseline", "--no-realistic", "--no-parity", "--no-fp32-side"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.stdout.strip().splitlines()[-1] == lines[0], r.stdout[-2000:]
    assert len(lines[0]) < 6000                      # the driver's parser takes the LAST stdout line; it must stay small with `comm` present
    d = json.loads(lines[0])
    assert "roofline" in d and d["roofline"]["frac"] > 0
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16 and d["config"]["parallelism"] == "dp2"
    c = d["comm"]
    assert c["payload"] == "bf16" and c["allreduce_bytes_per_step"] > 2 * 100e6          # two BERT-base towers, 2 B per parameter
    import math
    assert c["allreduce_ms_on_comm_stream_per_step"] > 0
    assert "exposed_wait_ms_per_step" in c and math.isfinite(c["exposed_wait_ms_per_step"]) and c["exposed_wait_ms_per_step"] >= 0   # a first SCALE record cannot lack it
