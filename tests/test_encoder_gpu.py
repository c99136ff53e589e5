"""End-to-end parity of the retriever step (BiBertEncoder -> similarity -> KL-distill loss -> backward)
against golden vectors produced by the IMPORTED REFERENCE (tests/golden/step_*.npz, see
oracle/make_golden.py).  Weights are regenerated on the box from the seeded integer generator.

Tolerances
  f32 parity mode : |x - ref| <= 1e-3 on loss / log-probs and 1e-3 * max(1,|x|) on embeddings / logits
                    (BASELINE.json north_star: "loss/logits within 1e-3 of reference"); measured errors
                    are ~1e-5.  Parameter gradients: 2e-4 of each tensor's largest entry + 1e-5 of the model's
                    largest gradient entry (absolute floor for analytically-zero gradients); measured 4e-5.
  fp16 engine     : (the benchmarked engine: fp16 operands, f32 accumulation, 16-bit + correction-byte residual stream) the
                    measured x3 bounds of simxns_amd/utils/parity.py FP16_HOT_TOL -- at the distance from the fp64 reference at
                    which an emulation of the reference's own apex-O1 mode sits (profiles/r03_o1_emulation.json): logits
                    ~2.4e-3 of their scale, loss ~1e-3, gradient cosine >= 0.999 on the dense weights.
  bf16 (experimental): documented looser bound (bf16 has 8 mantissa bits): embeddings 6e-2 abs on O(1)
                    values, loss 5e-2, gradients checked by cosine similarity >= 0.98.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["cls_only_last_layer", "full_last_layer"], autouse=True)
def last_layer_mode(request, monkeypatch):
    """Every test of this file runs twice: with the [CLS]-only last layer the towers use by default (the reference's
    models.py:81 reads row 0 only) and with every row of the last layer computed (SIMX_FULL_LAST_LAYER=1).  Both must
    match the reference's goldens."""
    monkeypatch.setenv("SIMX_FULL_LAST_LAYER", "1" if request.param == "full_last_layer" else "0")
    return request.param


def _cfg_from(G):
    from simxns_amd.engine import BertConfigLite
    c = json.loads(str(G["cfg"]))
    return BertConfigLite(vocab_size=c["vocab"], hidden_size=c["hidden"], num_hidden_layers=c["layers"],
                          num_attention_heads=c["heads"], intermediate_size=c["inter"],
                          max_position_embeddings=c["max_pos"], type_vocab_size=c["type_vocab"], layer_norm_eps=c["eps"],
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)   # goldens are dropout-free


def _named_shapes(enc):
    return [(k, tuple(p.shape)) for k, p in enc.named_parameters()]


def build_models(G, dev, dtype):
    from simxns_amd.model.models import HFBertEncoder, BiBertEncoder, Reranker
    from simxns_amd.utils import synth
    cfg = _cfg_from(G)
    std = float(G["std"])
    seeds = [int(s) for s in G["seeds"]]
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model = HFBertEncoder(cfg, compute_dtype=dtype)
    bi.ctx_model = HFBertEncoder(cfg, compute_dtype=dtype)
    bi.question_model.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(bi.question_model), seeds[0], std=std))
    bi.ctx_model.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(bi.ctx_model), seeds[1], std=std))
    tenc = HFBertEncoder(cfg, compute_dtype=dtype)
    tenc.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(tenc), seeds[2], std=std))
    teacher = Reranker(tenc, cfg.hidden_size)
    with torch.no_grad():
        teacher.qa_classifier.weight.copy_(torch.from_numpy(G["qa_w"]))
        teacher.qa_classifier.bias.copy_(torch.from_numpy(G["qa_b"]))
    return bi.to(dev), teacher.to(dev)


def run_step(G, dev, dtype):
    from simxns_amd import ops
    bi, teacher = build_models(G, dev, dtype)
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    bi.zero_grad()
    q, c = bi(t("q_ids"), t("q_mask"), t("c_ids"), t("c_mask"))
    with torch.no_grad():
        z = teacher(t("t_ids"), t("t_mask"))
    loss, distill, sim = ops.kl_distill_loss(q, c, z, 1.0, False, 1)
    loss.backward()
    torch.cuda.synchronize()
    grads = {}
    for pre, m in (("question_model.", bi.question_model), ("ctx_model.", bi.ctx_model)):
        for k, p in m.named_parameters():
            grads[pre + k] = p.grad.detach().cpu().numpy().astype(np.float64)
    return dict(q=q.detach().cpu().numpy(), c=c.detach().cpu().numpy(), z=z.cpu().numpy(), sim=sim.cpu().numpy(),
                loss=loss.item(), grads=grads, bi=bi, teacher=teacher)


def _close(got, ref, tol, what):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    err = np.abs(got - ref).max()
    lim = tol * max(1.0, np.abs(ref).max())
    assert err <= lim, "%s: max err %.3e > %.3e (scale %.3e)" % (what, err, lim, np.abs(ref).max())
    return err


def test_tiny_step_fp32_vs_reference_golden(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    R = run_step(G, dev, "fp32")
    _close(R["q"], G["q_emb"], 1e-3, "q_emb")
    _close(R["c"], G["ctx_emb"], 1e-3, "ctx_emb")
    _close(R["z"], G["teacher_logits"], 1e-3, "teacher logits")
    _close(R["sim"], G["sim"], 1e-3, "student_simila")
    assert abs(R["loss"] - float(G["loss_kl"])) <= 1e-3
    # tighter: what f32 should really deliver
    _close(R["q"], G["q_emb"], 2e-5, "q_emb (tight)")
    _close(R["sim"], G["sim"], 5e-5, "sim (tight)")
    assert abs(R["loss"] - float(G["loss_kl"])) <= 5e-5
    # Per tensor: |g - ref| <= 2e-4 * max|ref| + 1e-5 * (largest gradient entry of the model).  The absolute term
    # covers the gradients that are analytically 0 (key biases: softmax shift invariance; last-layer LayerNorm bias
    # of the passage tower: sum_d ds[b,d] = 0), where f32 leaves ~1e-7 of round-off.  Measured: <= 4e-5 relative.
    gmax = max(np.abs(G["grad." + k]).max() for k in R["grads"])
    worst = 0.0
    for k, g in R["grads"].items():
        ref = G["grad." + k]
        err = np.abs(g - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 1e-5 * gmax, "grad %s: err %.3e (scale %.3e)" % (k, err, np.abs(ref).max())
        worst = max(worst, err / max(np.abs(ref).max(), 1e-3 * gmax))
    assert np.abs(R["grads"]["question_model.pooler.dense.weight"]).max() == 0.0      # pooler grads exactly 0
    print("worst grad rel-to-max err", worst)


def test_tiny_step_bf16_vs_reference_golden(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    R = run_step(G, dev, "bf16")
    _close(R["q"], G["q_emb"], 6e-2, "q_emb bf16")
    _close(R["c"], G["ctx_emb"], 6e-2, "ctx_emb bf16")
    assert abs(R["loss"] - float(G["loss_kl"])) <= 5e-2
    for k in ("ctx_model.encoder.layer.1.output.dense.weight", "ctx_model.encoder.layer.0.attention.self.query.weight",
              "question_model.encoder.layer.0.intermediate.dense.weight", "ctx_model.embeddings.word_embeddings.weight",
              "ctx_model.encoder.layer.1.attention.output.LayerNorm.weight"):
        g, ref = R["grads"][k].ravel(), G["grad." + k].ravel()
        cos = float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
        assert cos >= 0.98, "grad %s cosine %.4f" % (k, cos)


def test_base_cfg1_step_fp32_vs_reference_golden(dev, golden_dir):
    """BASELINE config 1: BERT-base, B=4, 1 hard negative, q32/p128 -- the reference's CPU-runnable case."""
    G = np.load(os.path.join(golden_dir, "step_base_cfg1.npz"))
    R = run_step(G, dev, "fp32")
    _close(R["q"], G["q_emb"], 1e-3, "q_emb")
    _close(R["c"], G["ctx_emb"], 1e-3, "ctx_emb")
    _close(R["z"], G["teacher_logits"], 1e-3, "teacher logits")
    _close(R["sim"], G["sim"], 1e-3, "student_simila (logits, rel to max)")
    assert abs(R["loss"] - float(G["loss_kl"])) <= 1e-3
    names = [str(n) for n in G["grad_names"]]
    norms = G["grad_norms"]
    for n, ref in zip(names, norms):       # same rule on the per-tensor L2 norms (398 tensors)
        got = np.sqrt((R["grads"][n] ** 2).sum())
        assert abs(got - ref) <= 2e-4 * ref + 1e-6 * norms.max(), "grad norm %s: %.6e vs %.6e" % (n, got, ref)
    for k in G.files:
        if k.startswith("gslice."):
            name = k[len("gslice."):]
            g = R["grads"][name]
            ref = G[k]
            got = g[:8, :64] if ref.ndim == 2 else g
            assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-6 * norms.max(), "grad slice %s" % name


def test_base_cfg1_step_bf16(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "step_base_cfg1.npz"))
    R = run_step(G, dev, "bf16")
    _close(R["q"], G["q_emb"], 8e-2, "q_emb bf16")
    _close(R["c"], G["ctx_emb"], 8e-2, "ctx_emb bf16")
    # logits are O(50): bf16 carries ~3 significant digits
    _close(R["sim"], G["sim"], 3e-2, "sim bf16 (rel to max)")
    names = [str(n) for n in G["grad_names"]]
    sel = [i for i, n in enumerate(names) if n.endswith("dense.weight") and "pooler" not in n]
    got = np.array([np.sqrt((R["grads"][names[i]] ** 2).sum()) for i in sel])
    ref = G["grad_norms"][sel]
    rel = np.abs(got - ref) / ref
    print("cfg1 bf16 grad-norm rel err: median %.4f max %.4f" % (np.median(rel), rel.max()))
    # measured on MI355X: median 0.027-0.029, max 0.039 -- a common factor, not noise: logits are O(100) here, so bf16's
    # ~0.3 % logit error moves the softmax / KL gradient d(loss)/d(sim) by a few per cent and every gradient scales with it
    assert np.median(rel) < 0.08 and rel.max() < 0.12


# ------------------------------------------------------------------------------------------ the HOT kernels under the reference golden
from simxns_amd.utils.parity import golden_errors as hot_errors, BF16_HOT_TOL, HOT_TOL  # noqa: E402




def test_base_hot_step_fp32_vs_reference_golden(dev, golden_dir):
    """The shapes the benchmarked kernels need (>= 16k tokens) in the f32 parity mode: loss / logits / embeddings within
    1e-3 (north_star), all 398 gradient norms and a slice of every gradient within 2e-4 / 1e-3 of their scale."""
    G = np.load(os.path.join(golden_dir, "step_base_hot.npz"))
    R = run_step(G, dev, "fp32")
    e = hot_errors(R, G)
    print("hot fp32 errors:", json.dumps(e))
    assert e["q_abs"] <= 1e-3 and e["c_abs"] <= 1e-3 and e["loss_abs"] <= 1e-3
    assert e["z_abs"] <= 1e-3 * max(1.0, e["z_scale"]) and e["sim_abs"] <= 1e-3 * max(1.0, e["sim_scale"])
    assert e["gnorm_rel_max"] <= 5e-4, e
    assert e["gslice_rel_to_max"] <= 2e-3 and e["gslice_cos_min"] >= 0.99999, e


def test_base_hot_step_bf16_vs_reference_golden(dev, golden_dir):
    """The BENCHMARKED kernels end to end against the reference: bf16 engine on the hot fixture.  Tolerances are three
    times the measured errors (BF16_HOT_TOL), not a generic bf16 allowance."""
    G = np.load(os.path.join(golden_dir, "step_base_hot.npz"))
    R = run_step(G, dev, "bf16")
    e = hot_errors(R, G)
    print("hot bf16 errors:", json.dumps(e))
    t = BF16_HOT_TOL
    assert e["q_abs"] <= t["emb_abs"] and e["c_abs"] <= t["emb_abs"], e
    assert e["sim_abs"] <= t["logits_rel"] * e["sim_scale"] and e["z_abs"] <= t["teacher_logits_abs"], e
    assert e["loss_abs"] <= t["loss_abs"], e
    assert e["gnorm_rel_median"] <= t["gnorm_rel_median"] and e["gnorm_rel_max"] <= t["gnorm_rel_max"], e
    assert e["gslice_cos_min"] >= t["gslice_cos_min"] and e["gslice_cos_median_all"] >= t["gslice_cos_median_all"], e


@pytest.mark.parametrize("mode", ["fp16", "fp16_plain"])
def test_base_hot_step_fp16_vs_reference_golden(dev, golden_dir, mode):
    """The fp16 engine (apex-O1 operand width, loss-scaled backward) on the hot fixture.  "fp16" = with the f32-grade residual
    stream (the default, apex O1's arithmetic): the round-2 verdict's targets -- logits within 1 % of their scale, loss within
    5e-3, worst dense-weight gradient slice cosine >= 0.97; "fp16_plain" = plain 16-bit stream, residual in the GEMM epilogue."""
    G = np.load(os.path.join(golden_dir, "step_base_hot.npz"))
    R = run_step(G, dev, mode)
    e = hot_errors(R, G)
    print("hot %s errors:" % mode, json.dumps(e))
    t = HOT_TOL[mode]
    assert e["q_abs"] <= t["emb_abs"] and e["c_abs"] <= t["emb_abs"], e
    assert e["sim_abs"] <= t["logits_rel"] * e["sim_scale"] and e["z_abs"] <= t["teacher_logits_abs"], e
    assert e["loss_abs"] <= t["loss_abs"], e
    assert e["gnorm_rel_median"] <= t["gnorm_rel_median"] and e["gnorm_rel_max"] <= t["gnorm_rel_max"], e
    assert e["gslice_cos_min"] >= t["gslice_cos_min"] and e["gslice_cos_median_all"] >= t["gslice_cos_median_all"], e


def test_tiny_and_cfg1_step_fp16(dev, golden_dir):
    """fp16 engine on the small goldens (generic kernels, ragged shapes): embeddings / logits ~8x closer than bf16."""
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    R = run_step(G, dev, "fp16")
    _close(R["q"], G["q_emb"], 8e-3, "q_emb fp16")
    _close(R["c"], G["ctx_emb"], 8e-3, "ctx_emb fp16")
    assert abs(R["loss"] - float(G["loss_kl"])) <= 8e-3
    for k in ("ctx_model.encoder.layer.1.output.dense.weight", "ctx_model.encoder.layer.0.attention.self.query.weight",
              "question_model.encoder.layer.0.intermediate.dense.weight", "ctx_model.embeddings.word_embeddings.weight",
              "ctx_model.encoder.layer.1.attention.output.LayerNorm.weight"):
        g, ref = R["grads"][k].ravel(), G["grad." + k].ravel()
        cos = float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
        assert cos >= 0.9995, "grad %s cosine %.5f" % (k, cos)
    G = np.load(os.path.join(golden_dir, "step_base_cfg1.npz"))
    R = run_step(G, dev, "fp16")
    _close(R["q"], G["q_emb"], 1e-2, "q_emb fp16 cfg1")
    _close(R["c"], G["ctx_emb"], 1e-2, "ctx_emb fp16 cfg1")
    _close(R["sim"], G["sim"], 4e-3, "sim fp16 cfg1 (rel to max)")
    names = [str(n) for n in G["grad_names"]]
    sel = [i for i, n in enumerate(names) if n.endswith("dense.weight") and "pooler" not in n]
    got = np.array([np.sqrt((R["grads"][names[i]] ** 2).sum()) for i in sel])
    rel = np.abs(got - G["grad_norms"][sel]) / G["grad_norms"][sel]
    print("cfg1 fp16 grad-norm rel err: median %.4f max %.4f" % (np.median(rel), rel.max()))
    assert np.median(rel) < 0.01 and rel.max() < 0.02


@pytest.mark.parametrize("mode", ["fp32", "fp16", "fp16+ckpt"])
def test_large_cfg5_step_vs_reference_golden(dev, golden_dir, mode, monkeypatch):
    """BASELINE config 5's geometry at full depth under a reference golden (oracle/make_golden.py --only-large): BERT-large,
    24 layers, H = 1024, 16 heads, F = 4096; one query (<= 128 tokens), two documents (<= 512: the chunked attention
    backward), cross-encoder rows <= 512; fp64 run of the imported reference.  fp32 within north_star's 1e-3; the fp16 engine
    (config 5 says "gradient checkpointing + fp16") at its measured distance, with and without per-layer recompute."""
    G = np.load(os.path.join(golden_dir, "step_large_cfg5.npz"))
    if mode.endswith("+ckpt"):
        monkeypatch.setenv("SIMX_GRAD_CKPT", "1")
    R = run_step(G, dev, mode.split("+")[0])
    e = hot_errors(R, G)
    print("large cfg5 %s errors:" % mode, json.dumps(e))
    if mode == "fp32":
        assert e["q_abs"] <= 1e-3 and e["c_abs"] <= 1e-3 and e["loss_abs"] <= 1e-3
        assert e["z_abs"] <= 1e-3 * max(1.0, e["z_scale"]) and e["sim_abs"] <= 1e-3 * max(1.0, e["sim_scale"])
        assert e["gnorm_rel_max"] <= 1e-3 and e["gslice_cos_min"] >= 0.9999, e
    else:
        # 24 layers of fp16 operand rounding, logits of O(30); measured: DESIGN.md 2 (bounds ~3x)
        assert e["q_abs"] <= 0.03 and e["c_abs"] <= 0.03 and e["sim_abs"] <= 0.015 * e["sim_scale"] and e["loss_abs"] <= 0.02, e
        assert e["gnorm_rel_median"] <= 0.01 and e["gslice_cos_min"] >= 0.99, e


def test_module_api_and_sequence_output(dev, golden_dir):
    """HFBertEncoder.forward(**kwargs) -> (sequence_output, pooled_output, None); padded view, CLS row == pooled."""
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    bi, teacher = build_models(G, dev, "fp32")
    ids, mask = torch.from_numpy(G["q_ids"]).to(dev), torch.from_numpy(G["q_mask"]).to(dev)
    seq, pooled, hs = bi.question_model(input_ids=ids, attention_mask=mask)
    assert hs is None and seq.shape == (ids.shape[0], ids.shape[1], 64) and pooled.shape == (ids.shape[0], 64)
    assert torch.equal(seq[:, 0, :], pooled)
    np.testing.assert_allclose(pooled.detach().cpu().numpy(), G["q_emb"], atol=2e-5 * 4)
    emb = bi.query_emb(ids, mask)        # [CLS]-only last layer: single-query attention sums in a different order
    assert torch.allclose(emb, pooled, atol=2e-5, rtol=1e-5)
    # share_weight aliasing + state_dict schema
    keys = list(bi.state_dict().keys())
    assert "question_model.encoder.layer.0.attention.self.query.weight" in keys
    assert "ctx_model.pooler.dense.bias" in keys
    tk = list(teacher.state_dict().keys())
    assert "encoder.embeddings.word_embeddings.weight" in tk and "qa_classifier.weight" in tk


def test_training_step_fused_optimizer(dev, golden_dir):
    """Two optimiser steps with clip 2.0 and warm-up: loss moves, weights change, grads are zeroed,
    pooler stays untouched (exact-zero grads) -- the clip/AdamW path on the flat buffers."""
    from simxns_amd import ops
    from simxns_amd.optim import FusedAdamW, LinearWarmupSchedule
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    bi, teacher = build_models(G, dev, "fp32")
    opt = FusedAdamW(bi, lr=1e-3, eps=1e-8)
    sch = LinearWarmupSchedule(opt, 1, 10, last_step=1)
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    pool0 = bi.question_model.pooler.dense.weight.detach().clone()
    losses = []
    with torch.no_grad():
        z = teacher(t("t_ids"), t("t_mask"))
    for it in range(3):
        q, c = bi(t("q_ids"), t("q_mask"), t("c_ids"), t("c_mask"))
        loss, _, _ = ops.kl_distill_loss(q, c, z)
        loss.backward()
        sq = opt.step(max_grad_norm=2.0)
        sch.step()
        losses.append(loss.item())
        assert float(bi.ctx_model.engine.flat_grad.abs().max()) == 0.0
    assert losses[2] < losses[0]
    assert torch.equal(pool0, bi.question_model.pooler.dense.weight)
    assert np.isfinite(losses).all()


# ------------------------------------------------------------------------------------------ teacher (reranker) train step
def run_teacher_step(G, dev, dtype):
    """co_training_marco_train.py:225-245: Reranker forward -> CrossEntropy(target 0) -> backward."""
    from simxns_amd import ops
    _, teacher = build_models(G, dev, dtype)
    teacher.train()                                     # dropout probabilities of the golden config are 0
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    teacher.zero_grad()
    logits = teacher(t("t_ids"), t("t_mask"))
    loss, contr = ops.teacher_ce_loss(logits, 1)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in teacher.named_parameters() if p.grad is not None}
    return loss.item(), grads


def test_tiny_teacher_step_fp32_vs_reference_golden(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    loss, grads = run_teacher_step(G, dev, "fp32")
    assert abs(loss - float(G["teacher_ce_loss"])) <= 5e-5
    names = [k[len("tgrad."):] for k in G.files if k.startswith("tgrad.")]
    assert len(names) == 41
    gmax = max(np.abs(G["tgrad." + k]).max() for k in names)
    for k in names:
        ref = G["tgrad." + k]
        got = grads[k].reshape(ref.shape)
        err = np.abs(got - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 1e-5 * gmax, "teacher grad %s: err %.3e (scale %.3e)" % (k, err, np.abs(ref).max())


def test_base_cfg1_teacher_step_fp32_vs_reference_golden(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "step_base_cfg1.npz"))
    loss, grads = run_teacher_step(G, dev, "fp32")
    assert abs(loss - float(G["teacher_ce_loss"])) <= 1e-3
    names = [str(n) for n in G["tgrad_names"]]
    norms = G["tgrad_norms"]
    for n, ref in zip(names, norms):
        got = np.sqrt((grads[n] ** 2).sum())
        assert abs(got - ref) <= 2e-4 * ref + 1e-6 * norms.max(), "teacher grad norm %s: %.6e vs %.6e" % (n, got, ref)
    for k in G.files:
        if k.startswith("tgslice."):
            name = k[len("tgslice."):]
            ref = G[k]
            g = grads[name]
            got = g[:8, :64] if (ref.ndim == 2 and ref.shape != g.shape) else g.reshape(ref.shape)
            assert np.abs(got - ref).max() <= 1e-3 * np.abs(ref).max() + 1e-6 * norms.max(), "teacher grad slice %s" % name


def test_tiny_teacher_step_bf16(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    loss, grads = run_teacher_step(G, dev, "bf16")
    assert abs(loss - float(G["teacher_ce_loss"])) <= 5e-2
    for k in ("encoder.encoder.layer.1.output.dense.weight", "encoder.encoder.layer.0.attention.self.query.weight", "qa_classifier.weight"):
        g, ref = grads[k].ravel(), G["tgrad." + k].ravel()
        cos = float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
        assert cos >= 0.98, "teacher grad %s cosine %.4f" % (k, cos)


def _teacher_step_16bit_errors(G, grads, loss):
    names = [k[len("tgrad."):] for k in G.files if k.startswith("tgrad.")]
    cos, rel = [], []
    for k in names:
        ref = G["tgrad." + k].ravel().astype(np.float64)
        g = grads[k].ravel()
        if np.linalg.norm(ref) <= 1e-9 * max(np.linalg.norm(G["tgrad." + n]) for n in names):
            continue
        rel.append(abs(np.linalg.norm(g) - np.linalg.norm(ref)) / np.linalg.norm(ref))
        if k.endswith("dense.weight") or k.endswith("query.weight") or k.endswith("key.weight") or k.endswith("value.weight") or k == "qa_classifier.weight":
            cos.append(float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30)))
    return dict(loss_abs=abs(loss - float(G["teacher_ce_loss"])), cos_min=min(cos), gnorm_rel_med=float(np.median(rel)), gnorm_rel_max=float(max(rel)))


def test_tiny_teacher_step_fp16(dev, golden_dir):
    """The reranker TRAIN step (co_training_marco_train.py:225-245) on the benchmarked engine (fp16, apex-O1 form) against the
    reference golden.  Measured on MI355X: loss error 1e-6..3e-5, slice cosine min 0.99988, gradient norms 0.05-0.06 % median /
    0.7-1.1 % max; bounds = 3x."""
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    loss, grads = run_teacher_step(G, dev, "fp16")
    e = _teacher_step_16bit_errors(G, grads, loss)
    print("teacher step fp16 errors:", json.dumps(e))
    assert e["loss_abs"] <= 1e-4 and e["cos_min"] >= 0.9996 and e["gnorm_rel_med"] <= 2e-3 and e["gnorm_rel_max"] <= 3.5e-2, e


# ------------------------------------------------------------------------------------------ BERT-large geometry (configs 4 / 5)
def test_large_hidden_geometry_against_oracle(dev):
    """H=1024, 16 heads, F=4096 (ernie-large / BERT-large layer geometry, two layers), passages of 512 tokens and
    queries of 128: the f32 engine against the oracle (forward, loss, a few gradients), and bf16 against f32."""
    from oracle import bert as ob
    from oracle import losses as ol
    from oracle.weights import BertCfg, make_bert_params, make_batch
    from simxns_amd import ops
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import BiBertEncoder, HFBertEncoder
    ocfg = BertCfg(vocab=2000, hidden=1024, layers=2, heads=16, inter=4096, max_pos=512)
    cfg = BertConfigLite(vocab_size=2000, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=4096,
                         max_position_embeddings=512, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    Pq, Pc = make_bert_params(ocfg, 31, std=0.03), make_bert_params(ocfg, 32, std=0.03)
    B, N = 2, 2
    q_ids, q_mask, _ = make_batch(41, B, 128, ocfg.vocab, 40, 20, 8)
    c_ids, c_mask, _ = make_batch(42, B * (1 + N), 512, ocfg.vocab, 300, 120, 64)
    z = np.linspace(-1, 1, B * (1 + N)).reshape(B, 1 + N)
    _, oq, cq = ob.bert_forward(Pq, q_ids, q_mask, ocfg.heads)
    _, oc, cc = ob.bert_forward(Pc, c_ids, c_mask, ocfg.heads)
    osim = ol.sim_block(oq, oc)
    oloss, _, ods = ol.kl_distill(osim, z)
    dq, dc = ol.sim_block_bwd(oq, oc, ods)
    Gc = ob.bert_backward(Pc, c_ids, c_mask, ocfg.heads, cc, dc)
    res = {}
    for dtype in ("fp32", "bf16"):
        bi = BiBertEncoder.__new__(BiBertEncoder)
        torch.nn.Module.__init__(bi)
        bi.question_model, bi.ctx_model = HFBertEncoder(cfg, dtype), HFBertEncoder(cfg, dtype)
        bi.question_model.load_numpy_state(Pq)
        bi.ctx_model.load_numpy_state(Pc)
        bi.to(dev)
        t = lambda a: torch.from_numpy(a).to(dev)
        q, c = bi(t(q_ids), t(q_mask), t(c_ids), t(c_mask))
        loss, _, sim = ops.kl_distill_loss(q, c, t(z).float())
        loss.backward()
        torch.cuda.synchronize()
        g = dict(bi.ctx_model.named_parameters())
        res[dtype] = (q.detach().cpu().numpy(), c.detach().cpu().numpy(), loss.item(),
                      {k: g[k].grad.cpu().numpy().astype(np.float64) for k in ("encoder.layer.1.output.dense.weight",
                                                                               "encoder.layer.0.attention.self.query.weight",
                                                                               "embeddings.position_embeddings.weight")})
    q32, c32, l32, g32 = res["fp32"]
    _close(q32, oq, 1e-3, "q_emb H=1024"); _close(c32, oc, 1e-3, "ctx_emb H=1024 S=512")
    assert abs(l32 - oloss) <= 1e-3
    gmax = max(np.abs(v).max() for v in Gc.values())
    for k, g in g32.items():
        assert np.abs(g - Gc[k]).max() <= 5e-4 * np.abs(Gc[k]).max() + 1e-5 * gmax, k
    q16, c16, l16, g16 = res["bf16"]
    _close(q16, oq, 1e-1, "q_emb bf16"); _close(c16, oc, 1e-1, "ctx_emb bf16")
    for k, g in g16.items():
        ref = Gc[k].ravel()
        cos = float(g.ravel() @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
        assert cos >= 0.97, "%s cosine %.4f" % (k, cos)


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_sequence_output_carries_gradient(dev, dtype):
    """HFBertEncoder.forward returns the whole sequence_output (models.py:77-82) and, as in the reference, every row
    of it backpropagates: loss = sum(w * seq) over all real tokens + sum(v * pooled), against the oracle's backward
    with d_seq."""
    from oracle import bert as ob
    from oracle.weights import BertCfg, make_bert_params, make_batch
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import HFBertEncoder
    ocfg = BertCfg(vocab=400, hidden=64, layers=2, heads=4, inter=128, max_pos=64)
    cfg = BertConfigLite(vocab_size=400, hidden_size=64, num_hidden_layers=2, num_attention_heads=4, intermediate_size=128,
                         max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    P = make_bert_params(ocfg, 9, std=0.08)
    ids, mask, _ = make_batch(77, 5, 48, 400, 20, 10, 3)
    rs = np.random.RandomState(3)
    w = rs.randn(5, 48, 64) * mask[..., None]
    v = rs.randn(5, 64)
    enc = HFBertEncoder(cfg, dtype)
    enc.load_numpy_state(P)
    enc.to(dev).eval()
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    seq, pooled, _ = enc(input_ids=t(ids), attention_mask=t(mask))
    loss = (seq * t(w).float()).sum() + (pooled * t(v).float()).sum()
    loss.backward()
    oseq, ocls, cache = ob.bert_forward(P, ids, mask, 4)
    tol = 2e-5 if dtype == "fp32" else 6e-2
    assert np.abs(seq.detach().cpu().numpy() - oseq * mask[..., None]).max() <= tol * 4
    d_seq = w.copy()
    d_cls = v + w[:, 0, :]                      # row 0 of seq IS pooled: both terms reach it
    d_seq[:, 0, :] = 0
    G = ob.bert_backward(P, ids, mask, 4, cache, d_cls, d_seq=d_seq)
    own = dict(enc.named_parameters())
    gmax = max(np.abs(g).max() for g in G.values())
    for k, g in G.items():
        got = own[k].grad.cpu().numpy().astype(np.float64)
        if dtype == "fp32":
            assert np.abs(got - g).max() <= 2e-4 * np.abs(g).max() + 1e-5 * gmax, k
        elif k.endswith("dense.weight") and "pooler" not in k:
            cos = float(got.ravel() @ g.ravel() / (np.linalg.norm(got) * np.linalg.norm(g) + 1e-30))
            assert cos > 0.98, (k, cos)


@pytest.mark.parametrize("dtype", ["fp16", "fp32", "bf16"])
def test_full_size_config2_properties(dev, dtype):
    """BASELINE config 2 at its FULL size (BERT-base passage tower, 2048 passages x 128 tokens = 262144 tokens) in each engine
    bench.py times (fp16 = the headline, fp32 = the recipes' arithmetic; bf16 experimental) through size-independent properties
    (the oracle cannot run this size; SimANS/train_MS_Pas_AR2.sh:8-13 x the B = 128 of configs[1]):
      * batch independence: a sequence's embedding does not depend on its batch mates (16 sequences re-encoded alone --
        a shape the small-size oracle tests cover -- against their rows of the full batch), all-max and ragged lengths;
      * the [CLS]-only last layer against every-row computation on the full batch;
      * additivity of the backward pass over the batch: the gradients of the 2048-sequence batch equal the SUM of the gradients
        of its two 1024-sequence halves.  The halves run different wgrad split plans (131072 instead of 262144 tokens: other
        split counts and token ranges per workgroup, tn_plan / xp_tn_plan) on the same per-token operands, so a token range
        dropped, doubled or mis-summed by a plan shows as >= 1/56 of a tensor's gradient; summation order alone is ~1e-6;
      * linearity of the backward pass in the upstream gradient (kept from round 3; blind to the plans, sees everything else)."""
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import HFBertEncoder
    from simxns_amd.utils import synth
    cfg = BertConfigLite(hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)      # BERT-base defaults
    enc = HFBertEncoder(cfg, compute_dtype=dtype)
    enc.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(enc), 1235, std=0.02))
    enc.to(dev).eval()
    P, S = 2048, 128
    pick = np.array([0, 1, 17, 255, 256, 511, 777, 1024, 1025, 1300, 1599, 1600, 1900, 2000, 2046, 2047])
    # distance between two shapes of the same arithmetic: f32 round-off for the fp32 engine, a few 16-bit ulps of the row otherwise
    tol_e = {"fp32": 2e-5, "fp16": 5e-3, "bf16": 4e-2}[dtype]
    for full in (True, False):
        ids, mask, lens = synth.make_batch(4242, P, S, cfg.vocab_size, 80, 25, 16, full=full)
        ti, tm = torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev)
        with torch.no_grad():
            emb = enc.embed(ti, tm)                                  # [CLS]-only last layer, persistent GEMMs
            alone = enc.embed(ti[pick], tm[pick])                    # 16 sequences: the small-shape kernels
            _, pooled, _ = enc(input_ids=ti, attention_mask=tm)      # every row of the last layer
        scale = float(emb.abs().max())
        assert torch.isfinite(emb).all() and scale > 0.1
        assert float((emb[pick] - alone).abs().max()) <= tol_e * scale, "batch independence (full=%s)" % full
        assert float((emb - pooled).abs().max()) <= tol_e * scale, "[CLS]-only vs all rows (full=%s)" % full
    rs = np.random.RandomState(5)
    d1 = torch.from_numpy(rs.randn(P, cfg.hidden_size).astype(np.float32)).to(dev)
    d2 = torch.from_numpy(rs.randn(P, cfg.hidden_size).astype(np.float32)).to(dev)
    own = dict(enc.named_parameters())

    def grads(d, rows=slice(None), keys=None):
        enc.zero_grad()
        enc.train(False)
        e = enc.embed(ti[rows], tm[rows])
        (e * d[rows]).sum().backward()
        torch.cuda.synchronize()
        return {k: own[k].grad.detach().float().clone() for k in (keys or own) if own[k].grad is not None}
    # ---- additivity over the batch, EVERY tensor (the ragged batch of the last iteration: the halves also differ in token count)
    for which, (ti, tm) in (("ragged", (ti, tm)),
                            ("all-max", tuple(torch.from_numpy(a).to(dev) for a in synth.make_batch(4242, P, S, cfg.vocab_size, 80, 25, 16, full=True)[:2]))):
        gf = grads(d1)
        ga, gb = grads(d1, slice(0, P // 2)), grads(d1, slice(P // 2, P))
        add_tol = {"fp32": 2e-5, "fp16": 3e-4, "bf16": 3e-4}[dtype]      # same operands, other summation order (f32 accumulators / slabs)
        worst = (0.0, None)
        for k, g in gf.items():
            n = float(g.norm())
            if n == 0.0:
                assert float(ga[k].norm()) == 0.0 and float(gb[k].norm()) == 0.0, k       # pooler: exactly zero everywhere
                continue
            if k.endswith("self.key.bias"):        # softmax is invariant to the key bias: this gradient IS round-off (sum of dK rows = 0
                n = float(gf[k.replace("self.key.", "self.query.")].norm())      # in exact arithmetic) -- measured on the query bias' scale
            rel = float((g - (ga[k] + gb[k])).norm()) / n
            if rel > worst[0]:
                worst = (rel, k)
        assert worst[0] <= add_tol, "backward additivity over the batch (%s, %s): %s off by %.3e of its norm" % (dtype, which, worst[1], worst[0])
    # ---- linearity in the upstream gradient
    keys = ("encoder.layer.11.output.dense.weight", "encoder.layer.5.attention.self.query.weight",
            "encoder.layer.0.intermediate.dense.bias", "embeddings.position_embeddings.weight")
    g1, g2, g12 = grads(d1, keys=keys), grads(d2, keys=keys), grads(0.5 * d1 - 2.0 * d2, keys=keys)
    lin = {"fp32": (0.999999, 6e-4),          # (bf16 gradient pairs: 2^-17 per operand; measured 1.7e-4)
            "fp16": (0.99999, 6e-3), "bf16": (0.995, 0.08)}[dtype]
    for k in keys:
        want = 0.5 * g1[k] - 2.0 * g2[k]
        cos = float((g12[k] * want).sum() / (g12[k].norm() * want.norm() + 1e-30))
        rel = float((g12[k] - want).norm() / (want.norm() + 1e-30))
        assert cos >= lin[0] and rel <= lin[1], "backward linearity %s: cos %.7f rel %.5f" % (k, cos, rel)


# ------------------------------------------------------------------------------------------ gradient checkpointing / layer-range backward
@pytest.mark.parametrize("dtype,dropout", [("fp32", 0.0), ("bf16", 0.1)])
def test_gradient_checkpointing_and_ranged_backward_change_nothing(dev, golden_dir, dtype, dropout):
    """cfg.gradient_checkpointing (SimANS/model/models.py:73-74; every train_*_AR2.sh passes it): the native backward re-runs
    each layer's forward from the kept layer input -- with the same stateless dropout masks -- so embeddings and every
    gradient must equal the keep-everything mode up to the atomics' summation order; and the backward split into layer
    ranges (simx_bert_bwd_range, used to overlap the gradient all-reduce) must equal the single call.  Activation memory
    shrinks accordingly."""
    import ctypes as C
    from simxns_amd import _lib as L
    from simxns_amd.model.models import HFBertEncoder
    from simxns_amd.utils import synth
    G = np.load(os.path.join(golden_dir, "step_tiny.npz"))
    cfg = _cfg_from(G)
    cfg.num_hidden_layers = 4
    cfg.hidden_dropout_prob = cfg.attention_probs_dropout_prob = dropout
    ids, mask = torch.from_numpy(G["c_ids"]).to(dev), torch.from_numpy(G["c_mask"]).to(dev)
    d = torch.from_numpy(np.random.RandomState(3).randn(ids.shape[0], cfg.hidden_size).astype(np.float32)).to(dev)
    res = {}
    for mode in ("keep", "ckpt", "ranges"):
        cfg.gradient_checkpointing = mode == "ckpt"
        enc = HFBertEncoder(cfg, compute_dtype=dtype)
        enc.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(enc), 77, std=0.08))
        enc.to(dev).train()
        enc.engine.dropout_seed = 123
        assert enc.engine.ccfg.grad_checkpoint == (1 if mode == "ckpt" else 0)
        if mode == "ranges":
            fired = []
            enc.engine.grad_ready_hook = lambda e, lo, hi: fired.append((lo, hi))
            enc.engine.bwd_parts = 3
        e = enc.embed(ids, mask)
        (e * d).sum().backward()
        torch.cuda.synchronize()
        res[mode] = (e.detach().cpu().numpy(), enc.engine.flat_grad.detach().cpu().numpy().copy(),
                     int(L.load().simx_bert_act_bytes(C.byref(enc.engine.ccfg), 65536, 512, 1)))
        if mode == "ranges":                  # slices tile the flat buffer top-down, no gap, no overlap
            assert len(fired) == 3 and fired[0][1] == enc.engine.n_params and fired[-1][0] == 0
            assert all(fired[i][0] == fired[i + 1][1] for i in range(2))
    # (fp32: the largest gradient is the token-type row, an atomic f32 sum over every token of the batch: two runs of the SAME
    # mode differ by up to ~1e-6 of it in the summation order; SIMX_DETERMINISTIC=1 removes that, tests/test_det_gpu.py)
    tol = 5e-6 if dtype == "fp32" else 2e-2
    gscale = np.abs(res["keep"][1]).max()
    for mode in ("ckpt", "ranges"):
        assert np.array_equal(res[mode][0], res["keep"][0]), "embeddings differ (%s)" % mode
        err = np.abs(res[mode][1] - res["keep"][1]).max()
        assert err <= tol * gscale, "%s: gradient differs from the keep-everything mode by %.3e (scale %.3e)" % (mode, err, gscale)
    assert res["ckpt"][2] < 0.65 * res["keep"][2]         # 4 layers: 4 layer inputs + a 2-slot ring instead of 4 slots
    big = L.BertCfg.from_buffer_copy(enc.engine.ccfg)     # BERT-large depth: 24 inputs + 2 slots instead of 24 slots
    big.layers, big.grad_checkpoint = 24, 0
    keep24 = int(L.load().simx_bert_act_bytes(C.byref(big), 65536, 512, 1))
    big.grad_checkpoint = 1
    assert int(L.load().simx_bert_act_bytes(C.byref(big), 65536, 512, 1)) < 0.2 * keep24


# ------------------------------------------------------------------------------------------ BASELINE configs[4]: MS-Doc, BERT-large, S=512
def test_config5_bert_large_s512_step_with_gradient_checkpointing(dev):
    """BASELINE configs[4] as a step: 24-layer H=1024 (coCondenser-large / BERT-large geometry), queries of 128 and documents
    of 512 tokens, 7 hard negatives, bf16 engine (the role of the reference's fp16, co_training_marco_train.py:97-104) with
    dropout 0.1 and gradient checkpointing (models.py:73-74) -- 8 queries x 8 documents = 32768 document tokens per step.
    Checks: the step runs and trains (finite loss that goes down over three optimiser steps on the same batch); the
    checkpointed step needs a fraction of the activation memory and gives the same embeddings / gradients as the
    keep-everything step; the memory both need is printed."""
    from simxns_amd import ops
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import BiBertEncoder, HFBertEncoder
    from simxns_amd.optim import FusedAdamW, LinearWarmupSchedule
    from simxns_amd.utils import synth
    B, N, QL, DL = 8, 7, 128, 512
    q_ids, q_mask, _ = synth.make_batch(501, B, QL, 30522, 40, 20, 8)
    d_ids, d_mask, dl = synth.make_batch(502, B * (1 + N), DL, 30522, 400, 120, 64)
    z = np.random.RandomState(9).randn(B, 1 + N).astype(np.float32) * 2
    t = lambda a: torch.from_numpy(a).to(dev)
    stats = {}
    for mode in ("ckpt", "keep"):
        cfg = BertConfigLite(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                             hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, gradient_checkpointing=mode == "ckpt")
        bi = BiBertEncoder.__new__(BiBertEncoder)
        torch.nn.Module.__init__(bi)
        bi.question_model, bi.ctx_model = HFBertEncoder(cfg, "bf16"), HFBertEncoder(cfg, "bf16")
        for m, seed in ((bi.question_model, 31), (bi.ctx_model, 32)):
            m.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(m), seed, std=0.02))
            m.engine.dropout_seed = 5
        bi.to(dev).train()
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        q, c = bi(t(q_ids), t(q_mask), t(d_ids), t(d_mask))
        loss, _, _ = ops.kl_distill_loss(q, c, t(z))
        loss.backward()
        torch.cuda.synchronize()
        stats[mode] = dict(peak_gb=(torch.cuda.max_memory_allocated() - base) / 2 ** 30, loss=loss.item(),
                           c=c.detach().float().cpu().numpy(), g=bi.ctx_model.engine.flat_grad.detach().cpu().numpy().copy())
        if mode == "ckpt":                                        # ... and it trains
            opt = FusedAdamW(bi, lr=2e-5, eps=1e-8)
            sch = LinearWarmupSchedule(opt, 0, 100, last_step=1)
            losses = [loss.item()]
            opt.step(max_grad_norm=2.0)
            for _ in range(2):
                for m in (bi.question_model, bi.ctx_model):
                    m.engine._drop_calls = 0                      # same masks every step: the loss is comparable
                q, c = bi(t(q_ids), t(q_mask), t(d_ids), t(d_mask))
                l2, _, _ = ops.kl_distill_loss(q, c, t(z))
                l2.backward()
                opt.step(max_grad_norm=2.0); sch.step()
                losses.append(l2.item())
            assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
        del bi
        torch.cuda.empty_cache()
    print("config 5 (B=%d x %d docs, %d doc tokens): peak step memory %.1f GB with gradient checkpointing, %.1f GB keeping all activations"
          % (B, 1 + N, int(dl.sum()), stats["ckpt"]["peak_gb"], stats["keep"]["peak_gb"]))
    assert stats["ckpt"]["peak_gb"] < 0.45 * stats["keep"]["peak_gb"]
    assert np.array_equal(stats["ckpt"]["c"], stats["keep"]["c"])
    assert abs(stats["ckpt"]["loss"] - stats["keep"]["loss"]) <= 1e-5           # (the loss kernel sums the rows with f32 atomics)
    gs = np.abs(stats["keep"]["g"]).max()
    assert np.abs(stats["ckpt"]["g"] - stats["keep"]["g"]).max() <= 2e-3 * gs      # (f32 atomics in the LN / bias-gradient sums)


# ------------------------------------------------------------------------------------------ optimiser trajectory (last row)
def test_optimizer_trajectory_vs_reference_golden(dev, golden_dir):
    """Four optimiser steps of the retriever job against the trajectory the imported reference produced (tests/golden/
    trajectory_tiny.npz: literal loop body, get_optimizer's groups, transformers' linear warm-up schedule, clip 2.0 -- active
    in steps 1-3, inactive in step 4 --, transformers-4 AdamW update): per step the loss, the pre-clip gradient norm, the
    learning rate, the L2 norm of EVERY tensor's cumulative update and the updates of six tensors element by element.
    f32 engine; the update is lr * m / (sqrt(v) + eps), so relative gradient errors of 1e-5 carry over one-to-one."""
    from simxns_amd import ops
    from simxns_amd.model.models import BiBertEncoder, HFBertEncoder
    from simxns_amd.optim import FusedAdamW, LinearWarmupSchedule
    from simxns_amd.utils import synth
    G = np.load(os.path.join(golden_dir, "trajectory_tiny.npz"))
    cfg = _cfg_from(G)
    seeds, std = [int(s) for s in G["seeds"]], float(G["std"])
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model, bi.ctx_model = HFBertEncoder(cfg, "fp32"), HFBertEncoder(cfg, "fp32")
    bi.question_model.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(bi.question_model), seeds[0], std=std))
    bi.ctx_model.load_numpy_state(synth.fill_bert_state_dict(_named_shapes(bi.ctx_model), seeds[1], std=std))
    bi.to(dev).train()
    opt = FusedAdamW(bi, lr=float(G["lr"]), eps=float(G["eps"]))
    sch = LinearWarmupSchedule(opt, int(G["warmup"]), int(G["total"]))
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    p0 = {k: v.detach().clone() for k, v in bi.named_parameters()}
    names = [str(n) for n in G["names"]]
    assert names == [k for k, _ in bi.named_parameters()]              # same registration order as the reference's modules
    lr_peak = float(G["lr"])
    for it in range(len(G["losses"])):
        q, c = bi(t("q_ids"), t("q_mask"), t("c_ids"), t("c_mask"))
        loss, _, _ = ops.kl_distill_loss(q, c, t("teacher").float(), 1.0, False, 1)
        loss.backward()
        assert abs(opt.param_groups[0]["lr"] - float(G["lrs"][it])) <= 1e-12
        sq = opt.step(max_grad_norm=float(G["max_grad_norm"]))
        sch.step()
        assert abs(loss.item() - float(G["losses"][it])) <= 2e-5, (it, loss.item(), float(G["losses"][it]))
        gn = float(sq.sqrt().item())
        assert abs(gn - float(G["grad_norms"][it])) <= 1e-4 * float(G["grad_norms"][it]), (it, gn)
        cur = dict(bi.named_parameters())
        dn = np.array([float((cur[k].detach() - p0[k]).double().norm()) for k in names])
        ref = G["dnorm%d" % it]
        # tensors whose gradient is ANALYTICALLY zero (key biases: softmax shift invariance; ~1e-13 in the fp64 reference) get
        # f32 round-off (~1e-8, the size of Adam's eps) instead, and Adam turns noise of that size into updates of up to lr per
        # element -- in the reference's own fp32 run as well.  They are held to that bound; every other tensor to 2e-3.
        live = ref > 1e-6 * max(ref.max(), 1e-30)
        if it > 0:
            # per tower: one key bias per layer + the two pooler tensors; + the passage tower's last LayerNorm bias (sum_d ds[b,d] = 0)
            assert live.sum() >= len(names) - 2 * (cfg.num_hidden_layers + 2) - 1
        assert np.abs(dn - ref)[live].max() <= 2e-3 * ref.max() + 1e-9 if live.any() else True, \
            "step %d: update norms differ by %.3e (scale %.3e)" % (it, np.abs(dn - ref)[live].max(), ref.max())
        numel = np.array([cur[k].numel() for k in names], np.float64)
        assert (dn[~live] <= lr_peak * (it + 1) * np.sqrt(numel[~live]) + 1e-12).all()
        for key in G.files:
            if key.startswith("delta%d." % it):
                k = key.split(".", 1)[1]
                d = (cur[k].detach() - p0[k]).cpu().numpy().astype(np.float64)
                d = d[:8, :64] if d.ndim == 2 else d
                assert np.abs(d - G[key]).max() <= 0.02 * lr_peak, "step %d %s: update differs by %.3e (lr %.1e)" % (it, k, np.abs(d - G[key]).max(), lr_peak)
    assert float((cur["ctx_model.pooler.dense.weight"] - p0["ctx_model.pooler.dense.weight"]).abs().max()) == 0.0
