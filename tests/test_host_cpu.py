"""CPU: host-side logic and the C-ABI surface (no compute calls -- those need a GPU)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from simxns_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["bash", os.path.join(ROOT, "simxns_amd", "csrc", "build.sh")])
    return _lib


def test_library_exports_every_header_symbol():
    L = _lib()
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "simx.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(simx_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"simx_bert_cfg", "simx_loss_params", "simx_stream_t"}
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), "libsimx_hip.so does not export %s" % name
        assert name in L.SIGNATURES, "ctypes stub has no signature for %s" % name
    assert set(L.SIGNATURES) <= declared
    assert lib.simx_version() >= 100


def test_param_layout_matches_bert_base():
    L = _lib()
    lib = L.load()
    cfg = L.BertCfg(L.SIMX_BF16, 12, 768, 12, 3072, 30522, 512, 2, 1e-12)
    n = lib.simx_bert_param_count(C.byref(cfg))
    assert n == 109482240                                # BertModel(bert-base-uncased) incl. pooler
    offs = []
    for layer, kinds in [(-1, range(0, 5))] + [(l, range(5, 17)) for l in range(12)] + [(12, range(17, 19))]:
        for k in kinds:
            o = lib.simx_bert_param_offset(C.byref(cfg), layer, k)
            assert o != C.c_size_t(-1).value and o % 4 == 0
            offs.append(o)
    assert offs == sorted(offs) and len(set(offs)) == len(offs) and offs[-1] < n
    assert lib.simx_bert_param_offset(C.byref(cfg), 3, 0) == C.c_size_t(-1).value      # embeddings id on a layer: rejected
    bad = L.BertCfg(L.SIMX_BF16, 12, 770, 12, 3072, 30522, 512, 2, 1e-12)
    assert lib.simx_bert_param_count(C.byref(bad)) == 0
    assert lib.simx_bert_act_bytes(C.byref(cfg), 262144, 2048, 1) > 70 * 2 ** 30       # cfg2 activations ~77 GB, kept


def test_module_api_and_state_dict_schema(golden_dir):
    """Key schema of BiBertEncoder / Reranker == the imported reference's (names recorded in the golden file)."""
    _lib()
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import BiBertEncoder, HFBertEncoder, Reranker, BiEncoderNllLoss, dot_product_scores  # noqa
    G = np.load(os.path.join(golden_dir, "step_base_cfg1.npz"))
    ref_names = set(str(n) for n in G["grad_names"])
    cfg = BertConfigLite(vocab_size=100, hidden_size=64, num_hidden_layers=12, num_attention_heads=4, intermediate_size=128,
                         max_position_embeddings=64)
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model, bi.ctx_model = HFBertEncoder(cfg), HFBertEncoder(cfg)
    names = set(k for k, _ in bi.named_parameters())
    assert names == ref_names
    assert set(bi.state_dict().keys()) == ref_names
    # optimiser grouping of the reference works on the names (co_training_marco_train.py:59)
    nd = [n for n in names if any(x in n for x in ("bias", "LayerNorm.weight"))]
    assert len(nd) == 2 * (2 + 12 * 10 + 1)
    # parameters are views of ONE flat buffer; load_state_dict writes through
    enc = bi.question_model
    assert sum(p.numel() for p in enc.parameters()) == enc.engine.n_params
    w = enc.state_dict()["encoder.layer.3.attention.self.key.weight"]
    sd = {k: torch.full_like(v, 0.5) for k, v in enc.state_dict().items()}
    enc.load_state_dict(sd)
    assert float(enc.engine.flat.min()) == 0.5 and float(enc.engine.flat.max()) == 0.5
    assert w.data_ptr() != 0
    r = Reranker(HFBertEncoder(cfg), 64)
    assert {"qa_classifier.weight", "qa_classifier.bias", "encoder.pooler.dense.weight"} <= set(r.state_dict().keys())
    # share_weight aliasing (models.py:92-95)
    import types
    d = os.path.join(str(pytest.importorskip("tempfile").mkdtemp()), "m")
    os.makedirs(d)
    import json
    json.dump(cfg.to_dict(), open(os.path.join(d, "config.json"), "w"))
    bi2 = BiBertEncoder(types.SimpleNamespace(model_type=d, share_weight=True, gradient_checkpointing=False))
    assert bi2.ctx_model is bi2.question_model


def test_product_path_fails_loudly_without_gpu():
    _lib()
    from simxns_amd import _lib as L
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import HFBertEncoder
    from simxns_amd import ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    enc = HFBertEncoder(BertConfigLite(vocab_size=50, hidden_size=32, num_hidden_layers=1, num_attention_heads=2,
                                       intermediate_size=64, max_position_embeddings=16))
    ids = torch.randint(1, 50, (2, 8))
    with pytest.raises(L.SimxError):
        enc(input_ids=ids, attention_mask=torch.ones_like(ids))
    with pytest.raises(L.SimxError):
        ops.kl_distill_loss(torch.randn(2, 8), torch.randn(4, 8), torch.randn(2, 2))
    with pytest.raises(L.SimxError):
        ops.simans_sample(torch.randn(2, 8, dtype=torch.float64), torch.ones(2, dtype=torch.float64), 2)


def test_missing_library_is_an_error(tmp_path, monkeypatch):
    from simxns_amd import _lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.SimxError):
        L.load()


def test_packed_batch_layout():
    _lib()
    from simxns_amd.engine import PackedBatch
    ids = torch.tensor([[101, 5, 6, 102, 0, 0], [101, 7, 102, 0, 0, 0], [101, 1, 2, 3, 4, 102]])
    mask = (ids != 0).long()
    pb = PackedBatch(ids, mask)
    assert pb.T == 13 and pb.max_len == 6 and pb.nseq == 3
    assert pb.cu.tolist() == [0, 4, 7, 13]
    assert pb.ids.tolist() == [101, 5, 6, 102, 101, 7, 102, 101, 1, 2, 3, 4, 102]
    assert pb.pos.tolist() == [0, 1, 2, 3, 0, 1, 2, 0, 1, 2, 3, 4, 5]
    x = torch.arange(13.0).unsqueeze(1).repeat(1, 2)
    padded = pb.unpack(x)
    assert padded.shape == (3, 6, 2) and padded[1, 2, 0] == 6 and padded[1, 3, 0] == 0
    full = PackedBatch(torch.ones(2, 4, dtype=torch.long), torch.ones(2, 4, dtype=torch.long))
    assert full.index is None and full.pos.tolist() == [0, 1, 2, 3, 0, 1, 2, 3]
    with pytest.raises(ValueError):
        PackedBatch(ids, torch.tensor([[0, 1, 1, 1, 0, 0], [1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 1, 1]]))


def test_synth_generator_equals_oracle_generator():
    from oracle import weights as ow
    from simxns_amd.utils import synth
    cfg = ow.BertCfg(**ow.TINY)
    a = ow.make_bert_params(cfg, 77, std=0.08)
    b = synth.fill_bert_state_dict(ow.bert_param_shapes(cfg), 77, std=0.08)
    assert a.keys() == b.keys() and all(np.array_equal(a[k], b[k]) for k in a)
    x, y = ow.make_batch(5, 7, 32, 1000, 9, 3, 4), synth.make_batch(5, 7, 32, 1000, 9, 3, 4)
    assert all(np.array_equal(p, q) for p, q in zip(x, y))


def test_schedule_and_checkpoint_state(tmp_path):
    _lib()
    from oracle import optim as oo
    from simxns_amd.optim import LinearWarmupSchedule
    from simxns_amd.utils.dpr_utils import CheckpointState, load_states_from_checkpoint, get_model_obj

    class O(object):
        base_lr = 5e-6
        param_groups = [{"lr": 5e-6}]
    s = LinearWarmupSchedule(O(), 5400, 54000)
    assert s.get_last_lr()[0] == 0.0
    for t in range(1, 40):
        s.step()
        assert abs(s.get_last_lr()[0] - 5e-6 * oo.linear_schedule(t, 5400, 54000)) < 1e-18
    assert CheckpointState._fields == ('model_dict', 'optimizer_dict', 'scheduler_dict', 'offset', 'epoch', 'encoder_params')
    p = str(tmp_path / "checkpoint-5")
    torch.save(CheckpointState({"w": torch.ones(2)}, {}, {"t": 3}, 0, 0, None)._asdict(), p)
    st = load_states_from_checkpoint(p)
    assert st.scheduler_dict == {"t": 3} and torch.equal(st.model_dict["w"], torch.ones(2))
    m = torch.nn.Linear(2, 2)

    class W(object):
        module = m
    assert get_model_obj(W()) is m and get_model_obj(m) is m


DIST_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.environ["SIMX_ROOT"])
from simxns_amd import parallel
rank, W = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
torch.manual_seed(rank)
B, D, H = 3, 4, 8
q = (torch.arange(B * H, dtype=torch.float32).view(B, H) + 100 * rank).requires_grad_(True)
c = (torch.arange(B * D * H, dtype=torch.float32).view(B * D, H) + 1000 * rank).requires_grad_(True)
gq, gc = parallel.gather_with_local_grad(q), parallel.gather_with_local_grad(c)
assert gq.shape == (W * B, H) and gc.shape == (W * B * D, H)
for r in range(W):                                           # rank order, exact payload
    assert torch.equal(gq[r * B:(r + 1) * B], torch.arange(B * H, dtype=torch.float32).view(B, H) + 100 * r)
wq = torch.arange(W * B, dtype=torch.float32).view(-1, 1) + 1
(gq * wq).sum().backward()                                    # gradient reaches ONLY the local slot
assert torch.equal(q.grad, wq[rank * B:(rank + 1) * B].expand(B, H))
assert parallel.global_positive_indices(W, B, D) == [r * B * D + j * D for r in range(W) for j in range(B)]
objs = parallel.all_gather_list({"rank": rank, "t": torch.ones(2) * rank}, max_size=640000000)
assert [o["rank"] for o in objs] == list(range(W)) and float(objs[1]["t"][0]) == 1.0
g1, g2 = torch.full((10,), float(rank + 1)), torch.full((6,), 2.0 * (rank + 1))
hs, scale = parallel.allreduce_flat_grads([g1, g2])
assert scale == 1.0 / W and float(g1[0]) == sum(range(1, W + 1)) and float(g2[0]) == 2.0 * sum(range(1, W + 1))
# DDP's role on the flat gradient buffers: slices reduced by the backward hooks as they become final + the rest at step()
from simxns_amd.engine import BertConfigLite
from simxns_amd.model.models import HFBertEncoder, Reranker
from simxns_amd.optim import FusedAdamW
cfg = BertConfigLite(vocab_size=50, hidden_size=8, num_hidden_layers=2, num_attention_heads=2, intermediate_size=16, max_position_embeddings=16)
model = Reranker(HFBertEncoder(cfg, "fp32"), 8)
# default payload follows the engine arithmetic: f32 engines reduce in f32 (the reference's DDP), all-16-bit engines send bf16
assert FusedAdamW(model, lr=1e-3).enable_overlap(W, parts=2).payload == "fp32"
m16 = Reranker(HFBertEncoder(cfg, "fp16"), 8)
assert FusedAdamW(m16, lr=1e-3).enable_overlap(W, parts=2).payload == "bf16"
os.environ["SIMX_GRAD_PAYLOAD"] = "bf16"
assert FusedAdamW(model, lr=1e-3).enable_overlap(W, parts=2).payload == "bf16"     # opt-in for an f32 engine
del os.environ["SIMX_GRAD_PAYLOAD"]
# what the bf16 payload costs: each rank's slice rounded to 8 significand bits, summed in bf16 by the backend -> |err| <= 2^-7 * sum|g_r| per element
# (2^-9 per rank for the rounding of its slice + the backend's bf16 add, which need not round to nearest)
ob = FusedAdamW(m16, lr=1e-3).enable_overlap(W, parts=2)
eb = m16.encoder.engine
gen = torch.Generator().manual_seed(5)
gr = [torch.randn(eb.n_params, generator=gen) * 10 ** torch.randint(-6, 2, (eb.n_params,), generator=gen).float() for _ in range(W)]
eb.ensure_grad().copy_(gr[rank])
for p_ in m16.qa_classifier.parameters():
    p_.grad = torch.zeros_like(p_)
eb.grad_ready_hook(eb, 0, eb.n_params)
assert ob.sync_grads() == 1.0 / W
exact = sum(g.double() for g in gr)
bound = 2.0 ** -7 * sum(g.abs().double() for g in gr)
errb = (eb.flat_grad.double() - exact).abs()
assert bool((errb <= bound + 1e-30).all()) and float(errb.max() / exact.abs().max()) < 8e-3, float((errb / bound.clamp_min(1e-30)).max())
opt = FusedAdamW(model, lr=1e-3).enable_overlap(W, parts=2)                         # f32 engine -> f32 payload: exact sums below
assert opt.payload == "fp32"
e = model.encoder.engine
assert e.grad_ready_hook is not None and e.bwd_parts == 2
n = e.n_params
e.ensure_grad().copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
for p_ in model.qa_classifier.parameters():
    p_.grad = torch.full_like(p_, float(rank + 1))
cut = n // 3
e.grad_ready_hook(e, cut, n)                                 # what _run_backward does after the upper layer range ...
e.grad_ready_hook(e, 0, cut)                                 # ... and after the lower one
scale = opt.sync_grads()
tot = sum(range(1, W + 1))
assert scale == 1.0 / W and torch.equal(e.flat_grad, torch.arange(n, dtype=torch.float32) * tot)
assert float(opt.state["extra"]["g"][0]) == tot and opt.sync_grads() == 1.0 / W     # idempotent until step()
assert torch.equal(e.flat_grad, torch.arange(n, dtype=torch.float32) * tot)
opt._synced = False                                          # a step whose backward never fired the hooks: synchronous path
opt.state["extra"]["g_ready"] = False
e.flat_grad.copy_(torch.ones(n) * (rank + 1))
opt.armed = False
e.grad_ready_hook(e, 0, n)                                   # un-armed (accumulating micro-step): nothing may be reduced
assert not opt._pending
opt.armed = True
assert opt.sync_grads() == 1.0 / W and float(e.flat_grad[5]) == tot
dist.barrier()
dist.destroy_process_group()
print("rank %d ok" % rank)
'''


def test_optimizer_state_interchange_with_torch_adamw():
    """K1 / (f)3: optimizer_dict and scheduler_dict in the reference's on-disk format (torch Optimizer.state_dict of
    transformers.AdamW with the two parameter groups of co_training_marco_train.py:57-65; LambdaLR.state_dict) load into
    FusedAdamW / LinearWarmupSchedule, and what these write loads back into torch's classes."""
    _lib()
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import HFBertEncoder, Reranker
    from simxns_amd.optim import FusedAdamW, LinearWarmupSchedule
    cfg = BertConfigLite(vocab_size=60, hidden_size=8, num_hidden_layers=2, num_attention_heads=2, intermediate_size=16,
                         max_position_embeddings=16)
    torch.manual_seed(0)
    model = Reranker(HFBertEncoder(cfg, "fp32"), 8)

    def ref_optimizer(mod):
        nd = ['bias', 'LayerNorm.weight']
        groups = [{'params': [p for n, p in mod.named_parameters() if not any(k in n for k in nd)], 'weight_decay': 0.01},
                  {'params': [p for n, p in mod.named_parameters() if any(k in n for k in nd)], 'weight_decay': 0.0}]
        return torch.optim.AdamW(groups, lr=3e-5, eps=1e-8)
    ref = ref_optimizer(model)
    sched = torch.optim.lr_scheduler.LambdaLR(ref, lambda t: min(1.0, t / 10.0))
    for it in range(3):
        for p in model.parameters():
            p.grad = torch.randn_like(p)
        ref.step(); sched.step()
    sd, ssd = ref.state_dict(), sched.state_dict()
    opt = FusedAdamW(model, lr=3e-5, eps=1e-8, weight_decay=0.01)
    sch = LinearWarmupSchedule(opt, 10, 100)
    opt.load_state_dict(sd)
    sch.load_state_dict(ssd)
    assert opt.step_count == 3 and sch.t == 3 and abs(sch.get_last_lr()[0] - 3e-5 * 0.3) < 1e-12
    views = opt._moment_views()
    order = opt._ref_param_order()
    assert len(order) == len(list(model.parameters()))
    for i, (n, p, g) in enumerate(order):
        assert torch.equal(views[id(p)][0], sd["state"][i]["exp_avg"]), n
        assert torch.equal(views[id(p)][1], sd["state"][i]["exp_avg_sq"]), n
    # ... and back: what FusedAdamW / LinearWarmupSchedule write is a torch state_dict
    out = opt.state_dict()
    assert [len(g["params"]) for g in out["param_groups"]] == [len(g["params"]) for g in sd["param_groups"]]
    ref2 = ref_optimizer(model)
    ref2.load_state_dict(out)
    for i in range(len(order)):
        assert torch.equal(ref2.state_dict()["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
        assert float(ref2.state_dict()["state"][i]["step"]) == 3.0
    sched2 = torch.optim.lr_scheduler.LambdaLR(ref2, lambda t: min(1.0, t / 10.0))
    sched2.load_state_dict(sch.state_dict())
    assert sched2.last_epoch == 3
    # a dict that is neither format is an error, never skipped
    with pytest.raises(ValueError):
        opt.load_state_dict({"foo": 1})
    with pytest.raises(ValueError):
        opt.load_state_dict({"state": {}, "param_groups": [{"params": [0, 1]}]})


def test_distributed_gather_semantics_gloo_world2(tmp_path):
    """N>1 path on CPU: gloo, world_size 2 (127.0.0.1 rendezvous)."""
    script = tmp_path / "w.py"
    script.write_text(DIST_WORKER)
    env = dict(os.environ, SIMX_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=240)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("rank %d ok" % r) in o, o


def test_host_samplers_replay_reference_picks(golden_dir):
    """simxns_amd's host-side SimANS draws (the DataLoader path of the drop-in datasets) on CPython's ``random``
    reproduce what the imported reference datasets picked (recorded seeds, tests/golden/sampler_ref.json)."""
    import json
    import random
    from simxns_amd.utils.MARCO_until_new import simans_draw
    from simxns_amd.utils.util_wiki import simans_draw_gauss
    meta = json.load(open(os.path.join(golden_dir, "sampler_ref.json")))
    N = meta["N"]
    random.seed(meta["marco_seed"])
    for m in meta["queries"]:
        random.choice([0])
        got = simans_draw(list(zip(m["cand"], m["scores"])), m["s_pos"], N, tau=3.0)
        assert got == m["picked"]
    random.seed(meta["wiki_seed"])
    for i, want in enumerate(meta["wiki_picked"]):
        m = meta["queries"][i]
        sp = m["s_pos"] if m["s_pos"] else 75.0
        order = list(range(len(m["cand"])))
        random.shuffle(order)
        cand, sc = [m["cand"][j] for j in order], [m["scores"][j] for j in order]
        chosen = simans_draw_gauss(cand, sc, sp, N, a=0.5, b=1.0)
        assert [c for c in cand if c in chosen][0:N] == want


def test_generate_job_metrics_and_tsv_writer(tmp_path):
    """compute_metrics / write_to_file of the generate job (co_training_generate.py:153-266) on a hand-checked case."""
    from simxns_amd.co_training.co_training_generate import compute_metrics, load_reference_from_stream, write_to_file
    gold = tmp_path / "qrels.tsv"
    gold.write_text("1 0 10 1\n1 0 11 1\n2 0 20 1\n3 0 30 1\n")
    rel = load_reference_from_stream(str(gold))
    assert rel == {1: [10, 11], 2: [20], 3: [30]}
    cand = {1: [5, 10, 6] + list(range(100, 160)), 2: list(range(200, 260)) + [20], 3: [7, 8, 9]}
    m = compute_metrics(rel, cand)
    assert abs(m["MRR @10"] - (0.5 / 3)) < 1e-12                 # q1 hit at rank 2, q2 at rank 61 (> 10), q3 none
    assert m["recall@1"] == 0 and abs(m["recall@50"] - 1 / 3) < 1e-12 and abs(m["recall@all"] - 2 / 3) < 1e-12
    scores = {q: [100.0 - i for i in range(len(c))] for q, c in cand.items()}
    path = write_to_file(cand, scores, [[1, "q one"], [2, "q two"], [3, "q three"]], rel, {}, "train", str(tmp_path), 7)
    lines = open(path).read().splitlines()
    assert os.path.basename(path) == "train_ce_7.tsv" and len(lines) == 3
    f = lines[0].split("\t")
    assert f[0] == "1" and f[1] == "q one" and f[2] == "10 99.0,11 0"          # 11 was not retrieved: score 0
    assert f[3].split(",")[:2] == ["5 100.0", "6 98.0"] and len(f[3].split(",")) == 62
    # the train job's parser reads it back (MARCO_until_new.py:170-172 format)
    from simxns_amd.utils.MARCO_until_new import read_sharded_tsv
    rows = read_sharded_tsv(path)
    assert len(rows) == 3
    # load_pos_examples('train') also reads qrels.train.addition.tsv (literal-match positives, :139-150); write_to_file merges
    # them into the POSITIVE column (temp_pos, :163-166), so they never reach the hard-negative column SimANS samples from
    from simxns_amd.co_training.co_training_generate import load_pos_examples
    (tmp_path / "qrels.train.tsv").write_text("1\t10\n1\t11\n2\t20\n3\t30\n")
    (tmp_path / "qrels.train.addition.tsv").write_text("1\t6\n3\t9\n3\t4242\n")
    pos, pos_add = load_pos_examples(str(tmp_path / "qrels.train.tsv"), "train", str(tmp_path))
    assert pos == rel and pos_add == {1: [6], 3: [9, 4242]}
    assert load_pos_examples(str(tmp_path / "qrels.train.tsv"), "dev", str(tmp_path))[1] == {}     # dev: no additions (:151)
    path = write_to_file(cand, scores, [[1, "q one"], [2, "q two"], [3, "q three"]], pos, pos_add, "train", str(tmp_path), 8)
    lines = open(path).read().splitlines()
    f = lines[0].split("\t")
    assert f[2] == "10 99.0,11 0,6 98.0"                                          # 6 is now a positive, with its retrieved score
    assert f[3].split(",")[0] == "5 100.0" and "6 98.0" not in f[3].split(",") and len(f[3].split(",")) == 61
    f3 = lines[2].split("\t")
    assert f3[2] == "30 0,9 98.0,4242 0" and f3[3] == "7 100.0,8 99.0"


def test_recipe_launcher_command_lines_parse():
    """simxns_amd.launch (the loop of SimANS/train_*_AR2.sh): every recipe's train command line is accepted by the
    train script's own argument parser, and carries the hyper-parameters of record."""
    from simxns_amd import launch
    from simxns_amd.co_training import co_training_marco_train as marco
    from simxns_amd.wiki import co_training_wiki_train as wiki
    parsers = {"simxns_amd/co_training/co_training_marco_train.py": marco.get_arguments,
               "simxns_amd/wiki/co_training_wiki_train.py": wiki.get_arguments,
               "simxns_amd/Doc_training/co_training_doc_train.py": marco.get_arguments}
    want = {"MS_Pas": (5000, 500, 60000, 16, 2, 5e-6), "NQ": (2000, 500, 30000, 8, 1, 1e-5),
            "TQ": (2000, 500, 10000, 8, 1, 5e-6), "MS_Doc": (5000, 1000, 40000, 32, 1, 5e-6)}
    for name, make in launch.RECIPES.items():
        r = make()
        script, flags = r["train"]
        argv = launch.job_argv(script, dict(flags, global_step=0, max_steps=r["max_steps"], iteration_step=r["iteration_step"],
                                            iteration_reranker_step=r["iteration_reranker_step"]), 8, 9539)
        assert argv[argv.index(script) - 1].startswith("--master_port") and "127.0.0.1" in argv
        a = parsers[script](argv[argv.index(script) + 1:])
        it, rr, ms, bs, acc, lr = want[name]
        assert (a.iteration_step, a.iteration_reranker_step, a.max_steps) == (it, rr, ms), name
        assert (a.per_gpu_train_batch_size, a.gradient_accumulation_steps, a.number_neg) == (bs, acc, 15), name
        assert abs(a.learning_rate - lr) < 1e-12 and a.gradient_checkpointing, name
        # every recipe has its generate job (second command of the round), parsed by the same flag set, one step ahead
        assert r["generate"] is not None and os.path.exists(os.path.join(ROOT, r["generate"][0])), name
        gscript, gflags = r["generate"]
        gargv = launch.job_argv(gscript, dict(gflags, global_step=r["iteration_step"], max_steps=r["max_steps"]), 8, 9539)
        g = marco.get_arguments(gargv[gargv.index(gscript) + 1:])
        assert g.global_step == it and g.ann_dir == a.ann_dir and g.output_dir == a.output_dir and g.train_qa_path, name
        if name in ("NQ", "TQ"):
            assert g.passage_path == "data/psgs_w100.tsv" and g.test_qa_path and g.origin_data_dir == a.origin_data_dir
    for sh in ("train_MS_Pas_AR2.sh", "train_NQ_AR2.sh", "train_TQ_AR2.sh", "train_MS_Doc_AR2.sh"):
        assert "simxns_amd.launch" in open(os.path.join(ROOT, sh)).read()


def test_answer_match_and_ranking_metrics():
    """SimpleTokenizer / has_answer / Eval_Tool (SimANS/utils/dpr_utils.py:91-164, 309-420) against literal restatements of
    the reference's loops (its module cannot be imported here: it imports faiss)."""
    import math
    import random
    import unicodedata
    from simxns_amd.utils.dpr_utils import Eval_Tool, SimpleTokenizer, has_answer
    tok = SimpleTokenizer()
    assert tok.tokenize("Hello, w\u00f6rld! it's 3.5\u00a0km").words(uncased=True) == ["hello", ",", "w\u00f6rld", "!", "it", "'", "s", "3", ".", "5", "km"]

    def literal(answers, text):                       # the reference's window compare
        t = tok.tokenize(unicodedata.normalize('NFD', text)).words(uncased=True)
        for a in answers:
            w = tok.tokenize(unicodedata.normalize('NFD', a)).words(uncased=True)
            for i in range(0, len(t) - len(w) + 1):
                if w == t[i:i + len(w)]:
                    return True
        return False
    rng = random.Random(0)
    vocab = ["new", "york", "New", "York", "city", "1999", ",", "the", "caf\u00e9", "cafe\u0301", "a", "b"]
    for _ in range(300):
        text = " ".join(rng.choice(vocab) for _ in range(rng.randint(0, 12)))
        answers = [" ".join(rng.choice(vocab) for _ in range(rng.randint(0, 3))) for _ in range(rng.randint(0, 3))]
        assert has_answer(answers, text, tok) == literal(answers, text), (answers, text)
    assert has_answer(["New  York"], "He moved to new york, in 1999.", tok) and not has_answer(["york new"], "new york", tok)
    assert has_answer([r"19\d\d"], "in 1999", tok, match_type="regex") and not has_answer(["(("], "((", tok, match_type="regex")
    R = [[rng.random() < 0.1 for _ in range(rng.choice([3, 100, 50]))] for _ in range(40)] + [[False] * 100, [True] + [False] * 99]

    def avg(f, n):
        return sum(f(r[:n], n) for r in R) / len(R)
    lit = {"MRR_n": lambda r, n: next((1.0 / (i + 1.0) for i, x in enumerate(r) if x), 0),
           "DCG_n": lambda r, n: sum(1 / math.log2(i + 2) for i, x in enumerate(r) if x),
           "nDCG_n": lambda r, n: sum(1 / math.log2(i + 2) for i, x in enumerate(r) if x) / sum(math.log2(i + 2) for i in range(n)),
           "P_n": lambda r, n: sum(1 for x in r if x) / n}

    def ap(r, n):
        s_, h = 0.0, 1
        for i, x in enumerate(r):
            if x:
                s_ += h / (i + 1.0)
                h += 1
        return s_ / n
    lit["MAP_n"] = ap
    m = Eval_Tool.get_matrics(R)
    assert len(m) == 30
    for name, f in lit.items():
        for n in (1, 5, 10, 20, 50, 100):
            assert abs(m[name + "@_" + str(n)] - avg(f, n)) < 1e-12, (name, n)


def test_wiki_generate_output_files(tmp_path):
    """reform_out / read_train_pos / load_passage / load_qa of the NQ / TQ generate job (co_training_generate_new_train_wiki.py
    :184-226, :317-343, :448-459): the gold passage stays positive_ctxs[0] and takes the retrieval score when retrieved, other
    answer-bearing passages become further positives, the rest hard negatives -- and TraditionDataset reads the file."""
    from simxns_amd.utils.MARCO_until_new import HashTokenizer
    from simxns_amd.utils.util_wiki import TraditionDataset
    from simxns_amd.wiki import co_training_generate_new_train_wiki as W
    psg = tmp_path / "psgs.tsv"
    psg.write_text("id\ttext\ttitle\n1\tparis is the capital of france\tFrance\n2\tberlin is in germany\tGermany\n"
                   "3\tthe eiffel tower stands in paris\tEiffel\n4\trome is old\tItaly\nbroken\n")
    P = W.load_passage(str(psg))
    assert P == [(0, "paris is the capital of france", "France"), (1, "berlin is in germany", "Germany"),
                 (2, "the eiffel tower stands in paris", "Eiffel"), (3, "rome is old", "Italy")]
    qa = tmp_path / "qa.csv"
    qa.write_text("capital of france?\t['Paris']\nwhere is berlin\t[\"Germany\", 'Deutschland']\n")
    Q, A = W.load_qa(str(qa))
    assert Q == ["capital of france?", "where is berlin"] and A == [["Paris"], ["Germany", "Deutschland"]]
    text = {p[0]: (p[1], p[2]) for p in P}
    top, hits, metrics, rd = W.validate(Q, text, A, [[2, 1, 0, 3], [3, 1, 0, 2]], [[9.0, 8.0, 7.0, 6.0], [5.0, 4.0, 3.0, 2.0]])
    assert hits == [[True, False, True, False], [False, True, False, False]] and top == [0.5, 1.0, 1.0, 1.0]
    assert abs(metrics["MRR_n@_5"] - 0.75) < 1e-12 and rd[0]["ctxs"][0] == {"d_id": "2", "text": P[2][1], "title": "Eiffel", "score": "9.0", "hit": "True"}
    gold = tmp_path / "train_ce_0.json"
    json.dump([{"question": Q[0], "positive_ctxs": [{"title": "France", "text": P[0][1], "passage_id": "1", "score": "55"}]},
               {"question": "unused", "positive_ctxs": []}], open(gold, "w"))
    pos = W.read_train_pos(str(gold))
    assert list(pos) == [Q[0]]
    out = W.reform_out(rd, pos)
    e0, e1 = out
    assert [c["passage_id"] for c in e0["positive_ctxs"]] == ["1", "2"] and e0["positive_ctxs"][0]["score"] == "7.0"   # gold retrieved at rank 3
    assert [c["passage_id"] for c in e0["hard_negative_ctxs"]] == ["1", "3"] and e0["q_id"] == "0" and e0["negative_ctxs"] == []
    assert [c["passage_id"] for c in e1["positive_ctxs"]] == ["1"] and len(e1["hard_negative_ctxs"]) == 3           # no gold: hits only
    path = tmp_path / "train_ce_5.json"
    json.dump(out, open(path, "w"))
    ds = TraditionDataset(str(path), HashTokenizer(), num_hard_negatives=2, a=0.5, b=0.0, max_seq_length=32)
    assert len(ds) == 2
    q_ids, ctx_ids, ce, answers, se = ds[0]
    assert len(ctx_ids) == 3 and len(ce) == 3


def test_launch_fold_accumulation_dry_run(capsys):
    """`launch.py --fold-accumulation`: the micro-batches of an optimizer step as one batch (per-query losses only)."""
    from simxns_amd import launch
    launch.main(["MS_Pas", "--dry-run", "--nproc", "8", "--last-step", "0"])
    plain = capsys.readouterr().out.splitlines()[0]
    launch.main(["MS_Pas", "--dry-run", "--nproc", "8", "--last-step", "0", "--fold-accumulation"])
    folded = capsys.readouterr().out.splitlines()[0]
    assert "--per_gpu_train_batch_size=16" in plain and "--gradient_accumulation_steps=2" in plain
    assert "--per_gpu_train_batch_size=32" in folded and "--gradient_accumulation_steps=1" in folded
    assert plain.replace("--per_gpu_train_batch_size=16", "").replace("--gradient_accumulation_steps=2", "") == \
        folded.replace("--per_gpu_train_batch_size=32", "").replace("--gradient_accumulation_steps=1", "")


def test_bench_line_is_compact(golden_dir):
    """The driver parses bench.py's LAST stdout line; round 4's 24 KB line was not taken (BENCH_r04.parsed == null).  The
    formatter must turn the largest record ever produced (profiles/r04_bench.json, every side line present) -- with an N > 1
    `comm` block added -- into one line under bench.LINE_LIMIT that round-trips through json.loads and still carries the
    headline, `roofline` and `cpu_baseline`."""
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    assert len(json.dumps(full)) > 20000
    full["comm"] = {"allreduce_bytes_per_step": 437928960, "allreduce_ms_on_comm_stream_per_step": 9.87, "exposed_wait_ms_per_step": 0.42,
                    "payload": "f32", "parts": 2, "replica_check": "parameter checksums of all ranks agree to 1e-6 after the first optimiser step"}
    line = bench.compact_line(full, "gpurun_out/bench_full.json")
    assert "\n" not in line and len(line) < bench.LINE_LIMIT < 8000
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert d[k] == full[k], k
    assert d["config"]["workload"].startswith("SimANS MS-MARCO Passage retriever step (BASELINE configs[1])")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert d["roofline"][k] == full["roofline"][k]
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    assert {"value", "unit", "cores", "kind", "sample"} <= set(d["cpu_baseline"])
    assert d["comm"]["allreduce_bytes_per_step"] == 437928960 and "replica_check" not in d["comm"]
    for name in ("fp32_mode", "recipe_of_record", "cfg3_inbatch", "teacher_train_step"):
        assert set(d["sides"][name]) <= {"value", "ms_per_step", "frac", "step_mfma_util"} and d["sides"][name]["value"] == full[name]["value"]
    assert d["sides"]["cfg4_prod"]["B8"]["value"] == full["cfg4_prod"]["B8"]["value"]
    # a failed side line degrades to a short error stub, and an absurdly long record still yields a parseable line
    full["fp32_mode"] = {"value": None, "error": "x" * 5000}
    full["config"]["workload"] = "w" * 20000
    d2 = json.loads(bench.compact_line(full))
    assert len(json.dumps(d2)) < bench.LINE_LIMIT and len(d2["sides"]["fp32_mode"]["error"]) <= 120


def test_bench_marks_foreign_counters_stale(tmp_path, monkeypatch):
    """Counters quoted from profiles/ are marked stale unless they were taken at the kernel sources in the tree."""
    import bench
    dg = bench.csrc_digest()
    assert re.fullmatch(r"[0-9a-f]{16}", dg)
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_digest", lambda: dg)
    json.dump({"source": {"commit": "abc", "csrc_digest": "0" * 16}, "kernels": {}}, open(tmp_path / "profiles" / "r09_traffic.json", "w"))
    assert bench.pmc_source("r*_traffic.json")["stale"] is True
    json.dump({"source": {"commit": "abc", "csrc_digest": dg}, "kernels": {}}, open(tmp_path / "profiles" / "r09_traffic.json", "w"))
    assert bench.pmc_source("r*_traffic.json")["stale"] is False


def test_no_spills_on_the_hot_path():
    """ISA gate (tools/check_isa.py): the compiler's resource report of every gfx950 kernel in libsimx_hip.so shows no VGPR
    spill and no scratch.  Round 4 shipped mha_bwd_long_kernel<.., DKV, DROP, 128> with 12-23 spilled VGPRs and
    gemm_tn_xq_kernel<bf16> with one; nothing failed."""
    _lib()
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_isa
    if not check_isa.collect():                     # objects built before build.sh wrote resource reports: rebuild once
        for f in os.listdir(os.path.join(ROOT, "simxns_amd", "csrc")):
            if f.endswith(".o"):
                os.remove(os.path.join(ROOT, "simxns_amd", "csrc", f))
        subprocess.check_call(["bash", os.path.join(ROOT, "simxns_amd", "csrc", "build.sh")])
    ks = check_isa.collect()
    names = " ".join(k["name"] for k in ks)
    assert len(ks) > 250 and "gemm_nt_p3_kernel" in names and "mha_bwd_long_kernel" in names and "gemm_tn_xq_kernel" in names
    assert check_isa.offenders(ks) == []
    assert not check_isa.ALLOW                      # nothing is exempt today; an entry needs a reason next to it
