"""CPU: the NumPy oracle against the golden vectors captured from the IMPORTED REFERENCE
(tests/golden/*, written by oracle/make_golden.py in the build container).  This is the oracle's pin:
it runs on any box, without /root/reference."""
import json
import os
import random

import numpy as np
import pytest

from oracle import bert as ob
from oracle import losses as ol
from oracle import optim as oo
from oracle import sampler as osamp
from oracle.weights import BertCfg, make_bert_params


def _cfg(G):
    return BertCfg(**json.loads(str(G["cfg"])))


def _params(G):
    cfg = _cfg(G)
    seeds = [int(s) for s in G["seeds"]]
    std = float(G["std"])
    return cfg, [make_bert_params(cfg, s, std=std) for s in seeds]


def test_tiny_step_forward_backward_matches_reference(golden_dir):
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    cfg, (Pq, Pc, Pt) = _params(G)
    _, q, cq = ob.bert_forward(Pq, G["q_ids"], G["q_mask"], cfg.heads)
    _, c, cc = ob.bert_forward(Pc, G["c_ids"], G["c_mask"], cfg.heads)
    np.testing.assert_allclose(q, G["q_emb"], atol=1e-11)
    np.testing.assert_allclose(c, G["ctx_emb"], atol=1e-11)
    # the reference as shipped (fp32) agrees to fp32 round-off
    np.testing.assert_allclose(q, G["q_emb_fp32"], atol=2e-5)
    Pt2 = {"encoder." + k: v for k, v in Pt.items()}
    Pt2["qa_classifier.weight"], Pt2["qa_classifier.bias"] = G["qa_w"], G["qa_b"]
    z, _, _ = ob.reranker_forward(Pt2, G["t_ids"], G["t_mask"], cfg.heads, keep=False)
    np.testing.assert_allclose(z, G["teacher_logits"], atol=1e-11)
    sim = ol.sim_block(q, c)
    np.testing.assert_allclose(sim, G["sim"], atol=1e-10)
    loss, _, ds = ol.kl_distill(sim, z)
    assert abs(loss - float(G["loss_kl"])) < 1e-12
    for lam in (0.0, 0.5):
        l2, _, _, _ = ol.wiki_normal_adv(sim, z, 1.0, lam)
        assert abs(l2 - float(G["loss_wiki_lam%g" % lam])) < 1e-11
    dq, dc = ol.sim_block_bwd(q, c, ds)
    Gq = ob.bert_backward(Pq, G["q_ids"], G["q_mask"], cfg.heads, cq, dq)
    Gc = ob.bert_backward(Pc, G["c_ids"], G["c_mask"], cfg.heads, cc, dc)
    for pre, Go in (("question_model.", Gq), ("ctx_model.", Gc)):
        for k, g in Go.items():
            ref = G["grad." + pre + k]
            assert np.abs(g - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-3), k
    assert np.abs(G["grad.question_model.pooler.dense.weight"]).max() == 0.0


def test_tiny_teacher_step_matches_reference(golden_dir):
    """teacher (reranker) train step: CE(target 0) through Linear(H,1) and the cross-encoder (co_training_marco_train.py:225-245)."""
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    cfg, (_, _, Pt) = _params(G)
    Pt2 = {"encoder." + k: v for k, v in Pt.items()}
    Pt2["qa_classifier.weight"], Pt2["qa_classifier.bias"] = G["qa_w"], G["qa_b"]
    z, cls, caches = ob.reranker_forward(Pt2, G["t_ids"], G["t_mask"], cfg.heads, keep=True)
    loss, dz = ol.teacher_ce(z)
    assert abs(loss - float(G["teacher_ce_loss"])) < 1e-12
    TG = ob.reranker_backward(Pt2, G["t_ids"], G["t_mask"], cfg.heads, caches, cls, dz)
    n = 0
    for k, g in TG.items():
        ref = G["tgrad." + k]
        assert np.abs(g.reshape(ref.shape) - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-3), k
        n += 1
    assert n == 41


def test_loss_restatements_match_reference_autograd(golden_dir):
    G = np.load(os.path.join(golden_dir, "losses.npz"))
    s, z = G["s"], G["z"]
    l, _, ds = ol.kl_distill(s, z, 1.0, 1.0, 1)
    assert abs(l - float(G["L1a_loss"])) < 1e-12 and np.abs(ds - G["L1a_ds"]).max() < 1e-12
    l, _, ds = ol.kl_distill(s, z, 2.0, 1.0 / np.sqrt(768.0), 2)
    assert abs(l - float(G["L1b_loss"])) < 1e-12 and np.abs(ds - G["L1b_ds"]).max() < 1e-12
    sim = ol.sim_block(G["q"], G["c"])
    l, _, _, corr, ds = ol.cross_kd(sim, z, 4.0, 0.1, 0.9)
    dq, dc = ol.sim_block_bwd(G["q"], G["c"], ds)
    assert abs(l - float(G["L3_loss"])) < 1e-11 and corr == int(G["L3_correct"])
    assert np.abs(dq - G["L3_dq"]).max() < 1e-11 and np.abs(dc - G["L3_dc"]).max() < 1e-11
    l, _, _, _, ds = ol.cross_kd(sim, z, 4.0, 0.1, 0.9, ol.sim_block(G["qo"], G["co"]), 1.0)
    assert abs(l - float(G["L3lwf_loss"])) < 1e-11
    pos = [int(v) for v in G["pos"]]
    l, corr, dq, dc, _ = ol.nll_inbatch(G["q2"], G["c2"], pos)
    assert abs(l - float(G["M2_loss"])) < 1e-12 and corr == int(G["M2_correct"])
    assert np.abs(dq - G["M2_dq"]).max() < 1e-12 and np.abs(dc - G["M2_dc"]).max() < 1e-12
    l, _, _, corr, dq, dc = ol.bi_kd(G["q2"], G["c2"], G["qT"], G["cT"], pos)
    assert abs(l - float(G["L4_loss"])) < 1e-11 and np.abs(dq - G["L4_dq"]).max() < 1e-11
    l, dz = ol.teacher_ce(z)
    assert abs(l - float(G["L6_loss"])) < 1e-12 and np.abs(dz - G["L6_dz"]).max() < 1e-12
    qr, cr = [G["dist_q0"], G["dist_q1"]], [G["dist_c0"], G["dist_c1"]]
    for r in range(2):
        l, _, dq, dc = ol.nll_inbatch_distributed(qr, cr, r)
        assert abs(l - float(G["dist_loss%d" % r])) < 1e-12
        assert np.abs(dq - G["dist_dq%d" % r]).max() < 1e-12 and np.abs(dc - G["dist_dc%d" % r]).max() < 1e-12


def test_sampler_literal_replay_and_weights(golden_dir):
    """reference_draw on CPython's random with the recorded seed reproduces what the imported
    Rocketqa_v2Dataset picked; S1 weights are the reference's math.exp values."""
    meta = json.load(open(os.path.join(golden_dir, "sampler_ref.json")))
    N = meta["N"]
    random.seed(meta["marco_seed"])
    for m in meta["queries"]:
        random.choice([0])                      # the reference's random.choice(pos_pairs_list)
        _, negs = osamp.reference_draw(random, m["cand"], m["scores"], m["s_pos"], N, osamp.LAPLACE, tau=3.0)
        assert negs == m["picked"]
        if m["s_pos"] != 0:
            assert osamp.weights(m["scores"], m["s_pos"], osamp.LAPLACE, tau=3.0) == m["weights_laplace"]
            assert osamp.weights(m["scores"], m["s_pos"], osamp.GAUSS, a=0.5, b=1.0) == m["weights_gauss_nq"]
        else:
            assert negs == m["cand"][-N:]       # positive not retrieved -> last N candidates
    assert meta["collate_shapes"]["q"] == [4, 32] and meta["collate_shapes"]["ctx"] == [4 * (1 + N), 128]
    assert meta["collate_shapes"]["ce"] == [4, 1 + N, 160] and meta["collate_shapes"]["pos"] == [0, 16, 32, 48]


def test_scheme_draw_has_the_reference_law():
    """The GPU scheme (Philox + fixed summation order + draw-order truncation) against the literal reference
    algorithm: the PRE-truncation union has the same distribution.  Compared on inclusion frequencies and on
    the mean union size over 6000 trials each (binomial 5-sigma bands)."""
    rs = np.random.RandomState(0)
    Cn, N, trials = 24, 6, 6000
    s_pos = 80.0
    scores = list(np.sort(s_pos - np.abs(rs.randn(Cn)) * 0.8)[::-1])
    cand = list(range(Cn))
    rng = random.Random(123)
    f_ref, f_sch = np.zeros(Cn), np.zeros(Cn)
    size_ref = size_sch = 0
    for t in range(trials):
        u, _ = osamp.reference_draw(rng, cand, scores, s_pos, N, osamp.LAPLACE, tau=3.0)
        for i in u:
            f_ref[i] += 1
        size_ref += len(u)
        negs, union, _ = osamp.scheme_draw(scores, s_pos, N, osamp.LAPLACE, 0.5, 0.0, 3.0, seed=99, offset=0, q=t)
        assert len(set(negs)) == N and negs == union[:N]
        for i in union:
            f_sch[i] += 1
        size_sch += len(union)
    p = (f_ref + f_sch) / (2 * trials)
    sigma = np.sqrt(np.maximum(p * (1 - p), 1e-4) * 2 / trials)
    assert (np.abs(f_ref - f_sch) / trials <= 5 * sigma + 1e-3).all()
    assert abs(size_ref - size_sch) / trials < 0.08


def test_philox_known_answer():
    # Random123 known-answer vectors for philox4x32-10
    assert osamp.philox4x32((0, 0, 0, 0), (0, 0)) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert osamp.philox4x32((0xffffffff,) * 4, (0xffffffff, 0xffffffff)) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert osamp.philox4x32((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0)) == \
        (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_adamw_restatement_vs_torch():
    """transformers.AdamW is gone from transformers 5: the restated update is cross-checked against
    torch.optim.AdamW, which differs only in the eps placement (O(eps/sqrt(v)))."""
    import torch
    rs = np.random.RandomState(1)
    p0, g = rs.randn(1000), rs.randn(1000)
    tp = torch.tensor(p0, dtype=torch.float64, requires_grad=True)
    opt = torch.optim.AdamW([tp], lr=1e-3, eps=1e-8, weight_decay=0.0)
    P, M, V = p0.copy(), np.zeros(1000), np.zeros(1000)
    for step in (1, 2, 3):
        tp.grad = torch.tensor(g * step)
        opt.step()
        oo.adamw_hf_step(P, g * step, M, V, step, 1e-3)
        # eps enters as eps/sqrt(1-beta2^t) in the HF form: up to lr*31.6*eps/|g| at t=1
        np.testing.assert_allclose(P, tp.detach().numpy(), rtol=0, atol=2e-6)
    assert oo.linear_schedule(0, 10, 100) == 0.0 and oo.linear_schedule(5, 10, 100) == 0.5
    assert oo.linear_schedule(10, 10, 100) == 1.0 and abs(oo.linear_schedule(55, 10, 100) - 0.5) < 1e-12
    tot, coef = oo.clip_coef([np.ones(4) * 3.0], 2.0)
    assert abs(tot - 6.0) < 1e-12 and abs(coef - 2.0 / (6.0 + 1e-6)) < 1e-12


def test_collate_oracle_matches_reference(golden_dir):
    """oracle/collate.py reproduces the imported Rocketqa_v2Dataset + create_biencoder_input2 tensors bit for bit."""
    from oracle import collate as oc
    G = np.load(os.path.join(golden_dir, "collate_ref.npz"))
    B, D = G["ce_ids"].shape[:2]
    o = oc.assemble(G["q_ids"], G["ctx_ids"], list(range(B)), list(range(B * D)), D)
    for k in ("q_ids", "q_mask", "ctx_ids", "ctx_mask", "ce_ids", "ce_mask", "tgt"):
        assert (o[k] == G[k]).all(), k
    assert o["positive_ctx_indices"] == [int(v) for v in G["pos"]]
    # both remove_special_token branches are present in the fixture
    last = np.array([G["ctx_ids"][r, oc.row_len(G["ctx_ids"][r], 0) - 1] for r in range(B * D)])
    assert (last == 102).any() and (last != 102).any()


@pytest.mark.parametrize("fixture", ["roberta_dot_tiny.npz", "roberta_dot_mean_tiny.npz"])
def test_roberta_dot_oracle_matches_reference(golden_dir, fixture):
    """E4: oracle RobertaDot restatement vs the imported reference (fp64 golden), forward + gradients; the second
    fixture is the use_mean=True (masked-mean pooling) variant."""
    import json
    from oracle.weights import BertCfg
    G = np.load(os.path.join(golden_dir, fixture))
    um = bool(int(G["use_mean"]))
    cfg = BertCfg(**json.loads(str(G["cfg"])))
    P = {"roberta." + k: v for k, v in make_bert_params(cfg, int(G["seed"]), std=0.08).items()}
    for k in ("embeddingHead.weight", "embeddingHead.bias", "norm.weight", "norm.bias"):
        P[k] = G["param." + k]
    q, cq = ob.roberta_dot_forward(P, G["q_ids"], G["q_mask"], cfg.heads, use_mean=um)
    d, cd = ob.roberta_dot_forward(P, G["d_ids"], G["d_mask"], cfg.heads, use_mean=um)
    np.testing.assert_allclose(q, G["q_emb"], atol=1e-11)
    np.testing.assert_allclose(d, G["d_emb"], atol=1e-11)
    np.testing.assert_allclose(q, G["q_emb_fp32"], atol=2e-5)
    sim = ol.sim_block(q, d)
    loss, _, ds = ol.kl_distill(sim, G["teacher"])
    assert abs(loss - float(G["loss"])) < 1e-12
    dq, dd = ol.sim_block_bwd(q, d, ds)
    Gq = ob.roberta_dot_backward(P, G["q_ids"], G["q_mask"], cfg.heads, cq, dq)
    Gd = ob.roberta_dot_backward(P, G["d_ids"], G["d_mask"], cfg.heads, cd, dd)
    checked = 0
    for k in Gq:
        if "grad." + k not in G.files:              # the mean-pooling fixture keeps a subset of the gradients
            continue
        ref = G["grad." + k]
        assert np.abs((Gq[k] + Gd[k]).reshape(ref.shape) - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-3), k
        checked += 1
    assert checked >= 15


def test_torch_cpu_restatement_matches_reference_goldens(golden_dir):
    """oracle/torch_cpu.py (the CPU-baseline leg of bench.py: the reference's torch operators without transformers) against the
    reference goldens: tiny step in f64 (all gradients) and BASELINE configs[0] (BERT-base, B=4, N=1) in the f32 it is
    timed in."""
    import torch
    from oracle import torch_cpu as tc
    tt = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    cfg, (Pq, Pc, Pt) = _params(G)
    Tq, Tc = tc.to_torch_params(Pq, torch.float64), tc.to_torch_params(Pc, torch.float64)
    Tt = tc.to_torch_params(Pt, torch.float64, requires_grad=False)
    teacher = (Tt, tt(G["qa_w"]).double(), tt(G["qa_b"]).double(), tt(G["t_ids"]), tt(G["t_mask"]))
    r = tc.retriever_step(Tq, Tc, tt(G["q_ids"]), tt(G["q_mask"]), tt(G["c_ids"]), tt(G["c_mask"]), cfg.heads, teacher=teacher)
    np.testing.assert_allclose(r["q"].numpy(), G["q_emb"], atol=1e-10)
    np.testing.assert_allclose(r["ctx"].numpy(), G["ctx_emb"], atol=1e-10)
    np.testing.assert_allclose(r["z"].numpy(), G["teacher_logits"], atol=1e-10)
    assert abs(r["loss"] - float(G["loss_kl"])) < 1e-11
    for pre, T in (("question_model.", Tq), ("ctx_model.", Tc)):
        for k, v in T.items():
            ref = G["grad." + pre + k]
            assert np.abs(v.grad.numpy() - ref).max() <= 1e-9 * max(np.abs(ref).max(), 1e-3), k
    G = np.load(os.path.join(golden_dir, "step_base_cfg1.npz"))
    cfg, (Pq, Pc, Pt) = _params(G)
    Tq, Tc = tc.to_torch_params(Pq), tc.to_torch_params(Pc)
    r = tc.retriever_step(Tq, Tc, tt(G["q_ids"]), tt(G["q_mask"]), tt(G["c_ids"]), tt(G["c_mask"]), cfg.heads,
                          teacher_logits=tt(G["teacher_logits"]).float())
    np.testing.assert_allclose(r["q"].numpy(), G["q_emb"], atol=5e-5)
    np.testing.assert_allclose(r["sim"].numpy(), G["sim"], rtol=0, atol=1e-3 * np.abs(G["sim"]).max())
    assert abs(r["loss"] - float(G["loss_kl"])) < 1e-3
    names = [str(n) for n in G["grad_names"]]
    norms = dict(zip(names, G["grad_norms"]))
    for pre, T in (("question_model.", Tq), ("ctx_model.", Tc)):
        for k in ("encoder.layer.3.intermediate.dense.weight", "embeddings.word_embeddings.weight", "encoder.layer.11.output.dense.bias"):
            got = float(T[k].grad.double().norm())
            assert abs(got - norms[pre + k]) <= 1e-3 * norms[pre + k] + 1e-7, (pre + k, got, norms[pre + k])


def test_dropout_mask_definition_statistics():
    """The stateless dropout mask the kernels and the oracle share (one 32-bit hash per four columns, bytes against an 8-bit
    threshold): realised drop rate = round(256 p) / 256, E[multiplier] = 1 exactly by construction, the four byte lanes of a
    hash behave alike, neighbouring columns / rows / streams are uncorrelated.  (The reference fixes only p = 0.1,
    SimANS/model/models.py:70-72; its torch RNG stream cannot be replayed by a stateless GPU mask.)"""
    from oracle import bert as ob
    for p, thr in ((0.1, 26), (0.5, 128), (0.002, 1), (0.999, 255), (0.25, 64)):
        t, sc = ob.drop_threshold(p)
        assert t == thr and abs(sc * (256 - thr) / 256.0 - 1.0) < 1e-6
    # outside the realisable range nothing is clamped INTO it: below 1/512 no dropout, p >= 1 drops everything (torch's p = 1)
    assert ob.drop_threshold(0.001) == (0, 1.0) and ob.drop_threshold(1.0) == (256, 0.0)
    assert np.array_equal(ob.drop_multipliers(0.001, 7, 3, np.arange(8), np.arange(16)), np.ones((8, 16)))
    assert not ob.drop_multipliers(1.0, 7, 3, np.arange(8), np.arange(16)).any()
    rows, cols = np.arange(3000), np.arange(512)
    m = ob.drop_multipliers(0.1, 1234, 19, rows, cols)
    keep = m > 0
    n = keep.size
    assert abs(keep.mean() - 230.0 / 256.0) < 4 * np.sqrt(0.09 / n) and abs(m.mean() - 1.0) < 4 * np.sqrt(0.09 / n) * 1.12
    for lane in range(4):                                  # each byte lane of the hash
        k = keep[:, lane::4]
        assert abs(k.mean() - 230.0 / 256.0) < 5 * np.sqrt(0.09 / k.size)
    d = keep.astype(np.float64) - keep.mean()
    for a, b in ((d[:, :-1], d[:, 1:]), (d[:-1], d[1:]), (d[:, 0::4], d[:, 3::4])):       # column / row neighbours, lanes of one hash
        assert abs((a * b).mean() / d.var()) < 5 / np.sqrt(a.size)
    m2 = ob.drop_multipliers(0.1, 1234, 20, rows, cols)    # another stream (site / layer): an independent mask
    assert abs(((m2 > 0) == keep).mean() - (0.8984375 ** 2 + 0.1015625 ** 2)) < 0.003
    assert np.array_equal(ob.drop_multipliers(0.0, 1, 1, rows[:4], cols[:8]), np.ones((4, 8)))
