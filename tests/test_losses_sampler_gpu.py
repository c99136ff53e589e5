"""Loss / similarity / sampler kernels vs the committed golden vectors (generated from the imported
reference, oracle/make_golden.py) and vs the oracle on fresh inputs.  f32 kernels: tolerance 2e-5
relative to max(1,|x|); sampler indices: bit-exact against the oracle's scheme restatement."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import losses as ol
from oracle import sampler as osamp

pytestmark = pytest.mark.gpu


def _close(got, ref, tol=3e-5, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    err = np.abs(got - ref).max()
    lim = tol * max(1.0, np.abs(ref).max())
    assert err <= lim, "%s: err %.3e > %.3e" % (what, err, lim)


def _t(a, dev, req=False):
    t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    return t.requires_grad_(req)


@pytest.fixture(scope="module")
def G(golden_dir):
    return np.load(os.path.join(golden_dir, "losses.npz"))


def test_kl_distill_golden(dev, G):
    from simxns_amd import ops
    s, z = G["s"], G["z"]
    B, D = s.shape
    # feed logits through q=I-like trick: q [B,H=D*?]: use H = D with ctx = one-hot rows so that sim == s
    for tag, temp, scale_flag, ga in (("a", 1.0, False, 1),):
        H = D
        q = np.zeros((B, H), np.float32); q[:] = 1.0
        c = np.zeros((B * D, H), np.float32)
        for b in range(B):
            for d in range(D):
                c[b * D + d, d] = s[b, d]
        tq, tc = _t(q, dev, True), _t(c, dev, True)
        loss, distill, sim = ops.kl_distill_loss(tq, tc, _t(z, dev), temp, scale_flag, ga)
        _close(sim.cpu().numpy(), s, what="sim")
        _close(loss.item(), G["L1a_loss"], what="L1a loss")
        loss.backward()
        # dctx[b*D+d, d] = ds[b,d] * q = ds
        ds = np.array([[tc.grad[b * D + d, d].item() for d in range(D)] for b in range(B)])
        _close(ds, G["L1a_ds"], what="L1a ds")


@pytest.mark.parametrize("temp,scale,ga", [(1.0, False, 1), (2.0, True, 2)])
def test_kl_distill_oracle(dev, temp, scale, ga):
    from simxns_amd import ops
    rs = np.random.RandomState(3)
    B, D, H = 9, 16, 768
    q, c, z = rs.randn(B, H) * 0.4, rs.randn(B * D, H) * 0.4, rs.randn(B, D) * 2
    tq, tc = _t(q, dev, True), _t(c, dev, True)
    loss, distill, sim = ops.kl_distill_loss(tq, tc, _t(z, dev), temp, scale, ga)
    loss.backward()
    q32, c32 = q.astype(np.float32).astype(np.float64), c.astype(np.float32).astype(np.float64)
    osim = ol.sim_block(q32, c32)
    l, dl, ods = ol.kl_distill(osim, z.astype(np.float32).astype(np.float64), temp, 1.0 / np.sqrt(H) if scale else 1.0, ga)
    dq, dc = ol.sim_block_bwd(q32, c32, ods)
    _close(sim.cpu().numpy(), osim, what="sim")
    _close(loss.item(), l, what="loss")
    _close(distill.item(), dl, what="distill")
    _close(tq.grad.cpu().numpy(), dq, what="dq")
    _close(tc.grad.cpu().numpy(), dc, what="dc")


@pytest.mark.parametrize("lam", [0.0, 0.5])
def test_wiki_loss_oracle(dev, lam):
    from simxns_amd import ops
    rs = np.random.RandomState(4)
    B, D, H = 5, 16, 64
    q, c, z = rs.randn(B, H) * 0.4, rs.randn(B * D, H) * 0.4, rs.randn(B, D) * 2
    tq, tc = _t(q, dev, True), _t(c, dev, True)
    loss, normal, adv, sim = ops.wiki_normal_adv_loss(tq, tc, _t(z, dev), 1.0, lam)
    loss.backward()
    q32, c32 = q.astype(np.float32).astype(np.float64), c.astype(np.float32).astype(np.float64)
    osim = ol.sim_block(q32, c32)
    l, n_, a_, ods = ol.wiki_normal_adv(osim, z.astype(np.float32).astype(np.float64), 1.0, lam)
    dq, dc = ol.sim_block_bwd(q32, c32, ods)
    _close(loss.item(), l, what="loss"); _close(normal.item(), n_, what="normal"); _close(adv.item(), a_, what="adv")
    _close(tq.grad.cpu().numpy(), dq, what="dq"); _close(tc.grad.cpu().numpy(), dc, what="dc")


def test_cross_kd_golden(dev, G):
    from simxns_amd import ops
    tq, tc = _t(G["q"], dev, True), _t(G["c"], dev, True)
    loss, correct, hard, soft = ops.cross_kd_loss(tq, tc, _t(G["z"], dev), 4.0, 0.1, 0.9)
    loss.backward()
    _close(loss.item(), G["L3_loss"], what="L3 loss")
    assert int(correct.item()) == int(G["L3_correct"])
    _close(tq.grad.cpu().numpy(), G["L3_dq"], what="L3 dq")
    _close(tc.grad.cpu().numpy(), G["L3_dc"], what="L3 dc")


def test_teacher_ce_golden(dev, G):
    from simxns_amd import ops
    tz = _t(G["z"], dev, True)
    loss, contr = ops.teacher_ce_loss(tz)
    loss.backward()
    _close(loss.item(), G["L6_loss"], what="L6 loss")
    _close(tz.grad.cpu().numpy(), G["L6_dz"], what="L6 dz")


def test_inbatch_nll_golden(dev, G):
    from simxns_amd.model.models import BiEncoderNllLoss, dot_product_scores
    tq, tc = _t(G["q2"], dev, True), _t(G["c2"], dev, True)
    loss, correct = BiEncoderNllLoss().calc(tq, tc, [int(v) for v in G["pos"]])
    loss.backward()
    _close(loss.item(), G["M2_loss"], what="M2 loss")
    assert int(correct.item()) == int(G["M2_correct"])
    _close(tq.grad.cpu().numpy(), G["M2_dq"], what="M2 dq")
    _close(tc.grad.cpu().numpy(), G["M2_dc"], what="M2 dc")
    _close(dot_product_scores(tq, tc).cpu().numpy(), G["q2"].astype(np.float32) @ G["c2"].astype(np.float32).T, what="scores")


def test_inbatch_nll_distributed_semantics_golden(dev, G):
    """caculate_cont_loss as seen by each of W=2 ranks: gradient only on the local slot."""
    from simxns_amd.model.models import BiEncoderNllLoss
    W, Bq, Np = 2, 3, 4
    qr = [G["dist_q%d" % r] for r in range(W)]
    cr = [G["dist_c%d" % r] for r in range(W)]
    pos = []
    for r in range(W):
        pos += [r * Bq * Np + j * Np for j in range(Bq)]
    for r in range(W):
        tq, tc = _t(np.concatenate(qr), dev, True), _t(np.concatenate(cr), dev, True)
        loss, _ = BiEncoderNllLoss().calc(tq, tc, pos, local_q=(r * Bq, Bq), local_ctx=(r * Bq * Np, Bq * Np))
        loss.backward()
        _close(loss.item(), G["dist_loss%d" % r], what="dist loss")
        gq, gc = tq.grad.cpu().numpy(), tc.grad.cpu().numpy()
        _close(gq[r * Bq:(r + 1) * Bq], G["dist_dq%d" % r], what="dist dq")
        _close(gc[r * Bq * Np:(r + 1) * Bq * Np], G["dist_dc%d" % r], what="dist dc")
        o = 1 - r
        assert np.abs(gq[o * Bq:(o + 1) * Bq]).max() == 0.0 and np.abs(gc[o * Bq * Np:(o + 1) * Bq * Np]).max() == 0.0


def test_inbatch_nll_big(dev):
    from simxns_amd import ops
    rs = np.random.RandomState(5)
    Q, Cn, H = 96, 1536, 768
    q, c = rs.randn(Q, H) * 0.3, rs.randn(Cn, H) * 0.3
    pos = [i * 16 for i in range(Q)]
    tq, tc = _t(q, dev, True), _t(c, dev, True)
    loss, corr = ops.inbatch_nll_loss(tq, tc, pos)
    loss.backward()
    q32, c32 = q.astype(np.float32).astype(np.float64), c.astype(np.float32).astype(np.float64)
    l, cc, dq, dc, _ = ol.nll_inbatch(q32, c32, pos)
    _close(loss.item(), l, tol=1e-4, what="loss"); assert int(corr.item()) == cc
    _close(tq.grad.cpu().numpy(), dq, tol=1e-4, what="dq"); _close(tc.grad.cpu().numpy(), dc, tol=1e-4, what="dc")


def test_inbatch_nll_config3_full_shape_each_local_slot(dev):
    """BASELINE configs[2] at its real shapes: 8 ranks x (128 queries, 2048 passages) -> global [1024, 16384] score matrix
    (PROD/ProD_base/train_DE_model_marco.py:224-278); for EACH of the 8 local slots the loss, the argmax count and the
    local-slot gradients against the oracle (f64 on the f32-rounded inputs).  Embedding scale: post-LayerNorm [CLS]
    vectors are O(1) per component, logits O(H)."""
    from simxns_amd import ops
    rs = np.random.RandomState(17)
    W, B, D, H = 8, 128, 16, 768
    Q, Cn = W * B, W * B * D
    c = (rs.randn(Cn, H) * 0.35).astype(np.float32)
    pos = [r * B * D + j * D for r in range(W) for j in range(B)]
    q = (0.12 * c[pos] + rs.randn(Q, H) * 0.3).astype(np.float32)          # queries lean towards their positives (~half are top-1)
    q64, c64 = q.astype(np.float64), c.astype(np.float64)
    S = q64 @ c64.T
    m = S.max(1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(S - m).sum(1))
    want_loss = float((lse - S[np.arange(Q), pos]).mean())
    want_correct = int((S.argmax(1) == np.asarray(pos)).sum())
    assert 0 < want_correct < Q                                            # a non-degenerate case
    dS = np.exp(S - lse[:, None])
    dS[np.arange(Q), pos] -= 1.0
    dS /= Q
    dq_all, dc_all = dS @ c64, dS.T @ q64
    tq0, tc0 = _t(q, dev), _t(c, dev)
    for r in range(W):
        tq, tc = tq0.clone().requires_grad_(True), tc0.clone().requires_grad_(True)
        loss, corr = ops.inbatch_nll_loss(tq, tc, pos, None, (r * B, B), (r * B * D, B * D))
        loss.backward()
        assert abs(loss.item() - want_loss) <= 1e-3, "rank %d loss %.6f vs %.6f" % (r, loss.item(), want_loss)
        assert int(corr.item()) == want_correct
        gq, gc = tq.grad.cpu().numpy(), tc.grad.cpu().numpy()
        lq, lc = slice(r * B, (r + 1) * B), slice(r * B * D, (r + 1) * B * D)
        for got, ref, what in ((gq[lq], dq_all[lq], "dq"), (gc[lc], dc_all[lc], "dctx")):      # relative to the largest entry
            err = np.abs(got - ref).max()
            assert err <= 2e-4 * np.abs(ref).max(), "cfg3 %s rank %d: err %.3e (scale %.3e)" % (what, r, err, np.abs(ref).max())
        gq[lq] = 0; gc[lc] = 0
        assert np.abs(gq).max() == 0.0 and np.abs(gc).max() == 0.0, "remote rows must not receive gradient"


# ------------------------------------------------------------------------------------------ sampler
def test_sampler_bit_exact_vs_oracle_scheme(dev, golden_dir):
    from simxns_amd import ops
    meta = json.load(open(os.path.join(golden_dir, "sampler_ref.json")))
    qs = meta["queries"]
    N = meta["N"]
    scores = np.array([m["scores"] for m in qs], dtype=np.float64)
    spos = np.array([m["s_pos"] for m in qs], dtype=np.float64)
    for form, kw, wkey in ((osamp.LAPLACE, dict(tau=3.0), "weights_laplace"), (osamp.GAUSS, dict(a=0.5, b=1.0), "weights_gauss_nq")):
        neg, uni, cnt, wts = ops.simans_sample(torch.from_numpy(scores).to(dev), torch.from_numpy(spos).to(dev), N, form=form,
                                               seed=0x1234ABCD5678, offset=7, return_union=True, return_weights=True, **kw)
        neg, uni, cnt, wts = neg.cpu().numpy(), uni.cpu().numpy(), cnt.cpu().numpy(), wts.cpu().numpy()
        for qi, m in enumerate(qs):
            a, b, tau = kw.get("a", 0.5), kw.get("b", 0.0), kw.get("tau", 3.0)
            want, union, _ = osamp.scheme_draw(m["scores"], m["s_pos"], N, form, a, b, tau, 0x1234ABCD5678, 7, qi)
            assert list(neg[qi]) == want, (form, qi)
            assert list(uni[qi][:cnt[qi]]) == union[:2 * N]
            assert len(set(want)) == N
            if m["s_pos"] != 0:
                # S1: weights pinned to the reference's math.exp values
                np.testing.assert_allclose(wts[qi], np.array(m[wkey]), rtol=1e-13, atol=1e-300)


def test_sampler_big_batch_properties(dev):
    """cfg2 shape: 128 queries x 200 candidates x 15 negatives; structural properties + replay determinism."""
    from simxns_amd import ops
    rs = np.random.RandomState(0)
    nq, Cn, N = 128, 200, 15
    spos = 70 + 20 * rs.rand(nq)
    scores = np.sort(spos[:, None] - np.abs(rs.randn(nq, Cn)) * 1.5, axis=1)[:, ::-1].copy()
    ts, tp = torch.from_numpy(scores).to(dev), torch.from_numpy(spos).to(dev)
    a1 = ops.simans_sample(ts, tp, N, seed=11, offset=1).cpu().numpy()
    a2 = ops.simans_sample(ts, tp, N, seed=11, offset=1).cpu().numpy()
    a3 = ops.simans_sample(ts, tp, N, seed=11, offset=2).cpu().numpy()
    assert (a1 == a2).all() and (a1 != a3).any()
    assert a1.min() >= 0 and a1.max() < Cn
    assert all(len(set(r)) == N for r in a1)
    # ambiguous negatives: chosen candidates are much closer to the positive's score than the pool average
    gap_sel = np.abs(np.take_along_axis(scores, a1.astype(np.int64), 1) - spos[:, None]).mean()
    assert gap_sel < 0.5 * np.abs(scores - spos[:, None]).mean()
    for qi in (0, 77):
        want, _, _ = osamp.scheme_draw(list(scores[qi]), float(spos[qi]), N, osamp.LAPLACE, 0.5, 0.0, 3.0, 11, 1, qi)
        assert list(a1[qi]) == want


# ------------------------------------------------------------------------------------------ L4 / L5
def test_bi_kd_golden(dev, G):
    """BiEncoderKDLoss.calc (KD_softmax) against the imported reference's autograd (tests/golden/losses.npz, L4_*)."""
    import types
    from simxns_amd.model.models import BiEncoderKDLoss
    tq, tc = _t(G["q2"], dev, True), _t(G["c2"], dev, True)
    args = types.SimpleNamespace(KD_type="KD_softmax", TEMPERATURE=4.0, CE_WEIGHT=0.1, KD_WEIGHT=0.9)
    loss, correct = BiEncoderKDLoss().calc(args, tq, tc, _t(G["qT"], dev), _t(G["cT"], dev), [int(v) for v in G["pos"]])
    loss.backward()
    _close(loss.item(), G["L4_loss"], what="L4 loss")
    assert int(correct.item()) == int(G["L4_correct"])
    _close(tq.grad.cpu().numpy(), G["L4_dq"], what="L4 dq")
    _close(tc.grad.cpu().numpy(), G["L4_dc"], what="L4 dc")


def test_bi_kd_big_local_slot(dev):
    """cfg-3-like shapes, gradient only on the local slot, against the oracle."""
    from simxns_amd import ops
    rs = np.random.RandomState(9)
    Q, Cn, H, HT = 64, 1024, 768, 256
    q, c = rs.randn(Q, H) * 0.3, rs.randn(Cn, H) * 0.3
    qT, cT = rs.randn(Q, HT) * 0.5, rs.randn(Cn, HT) * 0.5
    pos = [i * 16 for i in range(Q)]
    tq, tc = _t(q, dev, True), _t(c, dev, True)
    loss, hard, soft, corr = ops.bi_kd_loss(tq, tc, _t(qT, dev), _t(cT, dev), pos, 4.0, 0.1, 0.9, None, (16, 32), (256, 512))
    loss.backward()
    f = lambda a: a.astype(np.float32).astype(np.float64)
    l, h_, s_, cc, dq, dc = ol.bi_kd(f(q), f(c), f(qT), f(cT), pos, 4.0, 0.1, 0.9)
    _close(loss.item(), l, tol=1e-4, what="loss"); _close(hard.item(), h_, tol=1e-4, what="hard"); _close(soft.item(), s_, tol=1e-4, what="soft")
    assert int(corr.item()) == cc
    gq, gc = tq.grad.cpu().numpy(), tc.grad.cpu().numpy()
    _close(gq[16:48], dq[16:48], tol=1e-4, what="dq local"); _close(gc[256:768], dc[256:768], tol=1e-4, what="dc local")
    assert np.abs(gq[:16]).max() == 0.0 and np.abs(gq[48:]).max() == 0.0 and np.abs(gc[:256]).max() == 0.0 and np.abs(gc[768:]).max() == 0.0


def test_fused_normal_inbatch_loss_oracle(dev, G):
    """L5 (MASTER/finetune/MS/co_training_model.py:249-270): normal loss + 0.2 * in-batch NLL, gradients add up."""
    from simxns_amd import ops
    q, c, z = G["q"], G["c"], G["z"]
    B, D = z.shape
    pos = [i * D for i in range(B)]
    tq, tc = _t(q, dev, True), _t(c, dev, True)
    loss, normal, nll, correct = ops.fused_normal_inbatch_loss(tq, tc, _t(z, dev), pos, grad_accum=2)
    loss.backward()
    f = lambda a: a.astype(np.float32).astype(np.float64)
    sim = ol.sim_block(f(q), f(c))
    l2, n_, _, ds = ol.wiki_normal_adv(sim, f(z), 1.0, 0.0, 1.0, 2)
    dq1, dc1 = ol.sim_block_bwd(f(q), f(c), ds)
    l3, cc, dq2, dc2, _ = ol.nll_inbatch(f(q), f(c), pos)
    _close(loss.item(), l2 + 0.2 * l3 / 2, what="L5 loss")
    assert int(correct.item()) == cc
    _close(tq.grad.cpu().numpy(), dq1 + 0.2 * dq2 / 2, what="L5 dq"); _close(tc.grad.cpu().numpy(), dc1 + 0.2 * dc2 / 2, what="L5 dc")


def test_cross_kd_lwf_golden(dev, G):
    """CrossBERTKDLoss.calc with the LwF term (kd_loss against the frozen student's scores) vs the imported reference."""
    import types
    from simxns_amd.model.models import CrossBERTKDLoss
    tq, tc = _t(G["q"], dev, True), _t(G["c"], dev, True)
    args = types.SimpleNamespace(KD_type="KD_softmax", TEMPERATURE=4.0, CE_WEIGHT=0.1, KD_WEIGHT=0.9, LwF_WEIGHT=1.0)
    loss, correct = CrossBERTKDLoss().calc(args, tq, tc, _t(G["z"], dev), LwF=True, ori_q_vector=_t(G["qo"], dev),
                                           ori_ctx_vectors=_t(G["co"], dev))
    loss.backward()
    _close(loss.item(), G["L3lwf_loss"], what="L3+LwF loss")
    assert int(correct.item()) == int(G["L3lwf_correct"])
    _close(tq.grad.cpu().numpy(), G["L3lwf_dq"], what="L3+LwF dq")
    _close(tc.grad.cpu().numpy(), G["L3lwf_dc"], what="L3+LwF dc")
