"""Check of an installation against the committed reference goldens (tests/golden/step_*.npz, written by the imported
reference in the build container: oracle/make_golden.py).  Product-side: no oracle import -- the fixture is data.
Used by tests/test_encoder_gpu.py and by bench.py's `parity_16bit` block (the measured distance of the benchmarked 16-bit
engine from the reference on the shapes its hot kernels need).  The fixtures are looked up in $SIMX_GOLDEN_DIR, else in
<repository>/tests/golden next to an in-tree package."""
import json
import os

import numpy as np
import torch


def golden_dir():
    env = os.environ.get("SIMX_GOLDEN_DIR")
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")


GOLDEN_DIR = golden_dir()

# bf16 tolerances = 3x the errors MEASURED on MI355X with this fixture (tests print them; docs/history.md "bf16 parity").  Measured,
# [CLS]-only / full last layer: embeddings max |err| 0.070 / 0.085 (values up to 4.4), student logits 1.65 / 1.57 on a scale
# of 28 (5.9 %), teacher logits 0.044, loss 0.0064 / 0.0009, gradient norms median 0.87 % / 0.78 % and max 4.6 % / 3.8 % over
# 396 tensors, cosine of the [:8,:64] slices of the dense-layer weight gradients min 0.85-0.89 (a query-tower FFN slice: ~150
# query tokens, and d(loss)/d(logit) is ill-conditioned at logits of O(30): a logit error of 1.6 moves a softmax weight 5x),
# median over all tensors 0.98.  The f32 engine on the same fixture: embeddings 8e-6, logits 1.5e-4 (5e-6 rel.), loss 4e-6,
# gradient norms 1e-5, slices 6e-5 of their maximum.
# The scalar quantities are ONE draw of the rounding noise each: a change of rounding points anywhere in the forward (e.g. gelu
# of the f32 pre-activation instead of its bf16 rounding) redraws them -- loss 0.0009 ... 0.025 and gradient-norm median
# 0.8 % ... 1.8 % (a common-mode factor: every gradient scales with the softmax weights the logit errors move) over the builds
# and boxes measured so far, with the direction measures unchanged or better (slice cosine min 0.92).  Bounds = 3x the largest.
# Round 3: the direction bound is what would catch a regression (a doubled error roughly halves 1 - cos), so it sits at the
# worst box's measurement minus a third of its distance to 1, not at 3x: cosine min >= 0.80 (measured 0.845-0.92).
BF16_HOT_TOL = dict(emb_abs=0.2, logits_rel=0.12, teacher_logits_abs=0.1, loss_abs=0.06, gnorm_rel_median=0.04, gnorm_rel_max=0.13,
                    gslice_cos_min=0.80, gslice_cos_median_all=0.96)
# fp16 engine (IEEE half operands = the operand width of the reference's own optional apex-O1 mode; f32 accumulation,
# statistics, master weights; loss-scaled backward).  11-bit significands instead of 8: every figure above shrinks ~8x.
# With the f32-grade residual stream (the default, apex O1's arithmetic).  Measured ([CLS]-only / full last layer): embeddings
# 0.0061 / 0.0042, student logits 0.075 / 0.071 on a scale of 28 (0.27 % / 0.25 %), teacher logits 0.0029 / 0.0026, loss 0.0028,
# gradient norms median 0.11 % / 0.03 %, max 0.31 % / 0.23 %, slice cosine min 0.99957 / 0.99983, median 0.99994 -- the
# distance of the reference's OWN fp16 mode from its fp64 run is 0.0036 / 0.24 % / 0.0023 / 0.0019 / 0.18 % / 0.44 % / 0.99979
# (apex O1 emulated on the imported modules, oracle/o1_emulation.py -> profiles/r03_o1_emulation.json).  Bounds ~3x the
# measurement, never looser than the round-2 verdict's targets (logits <= 1 %, loss <= 5e-3 x 1.6 for its one-draw noise,
# cosine >= 0.97).
FP16_HOT_TOL = dict(emb_abs=0.02, logits_rel=0.008, teacher_logits_abs=0.009, loss_abs=8e-3, gnorm_rel_median=0.004, gnorm_rel_max=0.01,
                    gslice_cos_min=0.998, gslice_cos_median_all=0.9998)
# plain 16-bit residual stream (residual added in the GEMM epilogue, the pre-LayerNorm sum rounded to fp16): measured logits
# 0.69 %, embeddings 0.013, loss 0.010, gradient norms median 0.6 % / max 0.96 %, slice cosine min 0.9986
FP16_PLAIN_HOT_TOL = dict(emb_abs=0.04, logits_rel=0.02, teacher_logits_abs=0.025, loss_abs=0.03, gnorm_rel_median=0.018, gnorm_rel_max=0.03,
                          gslice_cos_min=0.995, gslice_cos_median_all=0.999)
HOT_TOL = {"bf16": BF16_HOT_TOL, "fp16": FP16_HOT_TOL, "fp16_plain": FP16_PLAIN_HOT_TOL}


def _cfg_from(G):
    from ..engine import BertConfigLite
    c = json.loads(str(G["cfg"]))
    return BertConfigLite(vocab_size=c["vocab"], hidden_size=c["hidden"], num_hidden_layers=c["layers"],
                          num_attention_heads=c["heads"], intermediate_size=c["inter"],
                          max_position_embeddings=c["max_pos"], type_vocab_size=c["type_vocab"], layer_norm_eps=c["eps"],
                          hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)   # goldens are dropout-free


def run_golden_step(G, dev, dtype):
    """The retriever step of co_training_marco_train.py:198-217 on a golden's inputs, weights regenerated from its seeds."""
    from .. import ops
    from ..model.models import HFBertEncoder, BiBertEncoder, Reranker
    from . import synth
    cfg, std, seeds = _cfg_from(G), float(G["std"]), [int(s) for s in G["seeds"]]
    shapes = lambda enc: [(k, tuple(p.shape)) for k, p in enc.named_parameters()]
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model, bi.ctx_model = HFBertEncoder(cfg, compute_dtype=dtype), HFBertEncoder(cfg, compute_dtype=dtype)
    bi.question_model.load_numpy_state(synth.fill_bert_state_dict(shapes(bi.question_model), seeds[0], std=std))
    bi.ctx_model.load_numpy_state(synth.fill_bert_state_dict(shapes(bi.ctx_model), seeds[1], std=std))
    tenc = HFBertEncoder(cfg, compute_dtype=dtype)
    tenc.load_numpy_state(synth.fill_bert_state_dict(shapes(tenc), seeds[2], std=std))
    teacher = Reranker(tenc, cfg.hidden_size)
    with torch.no_grad():
        teacher.qa_classifier.weight.copy_(torch.from_numpy(G["qa_w"]))
        teacher.qa_classifier.bias.copy_(torch.from_numpy(G["qa_b"]))
    bi.to(dev)
    teacher.to(dev)
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
                loss=loss.item(), grads=grads)


def golden_errors(R, G):
    """Measured distance of one run from the reference golden (step_base_hot.npz: BERT-base, 16 queries x 16 passages,
    ~20k passage tokens -> gemm_nt_bf16_p3 / gemm_tn2 / mha<8> / the wide LayerNorm kernels all dispatch)."""
    e = {}
    for k, g in (("q", "q_emb"), ("c", "ctx_emb"), ("z", "teacher_logits"), ("sim", "sim")):
        e[k + "_abs"] = float(np.abs(np.asarray(R[k], np.float64) - G[g]).max())
        e[k + "_scale"] = float(np.abs(G[g]).max())
    e["loss_abs"] = abs(R["loss"] - float(G["loss_kl"]))
    names = [str(n) for n in G["grad_names"]]
    norms = G["grad_norms"]
    got = np.array([np.sqrt((R["grads"][n] ** 2).sum()) for n in names])
    live = norms > 1e-6 * norms.max()                    # (pooler / analytically-zero gradients excluded)
    rel = np.abs(got - norms)[live] / norms[live]
    e["gnorm_rel_max"], e["gnorm_rel_median"] = float(rel.max()), float(np.median(rel))
    e["gnorm_worst"] = np.asarray(names)[live][int(rel.argmax())]
    # element-wise: the [:8, :64] slice of every matrix gradient and every vector gradient.  Cosines are quoted for the
    # dense-layer weight matrices (the GEMM outputs; every element carries signal) and, as a median, for all tensors --
    # bias / LayerNorm / embedding-row slices of 64-768 elements include near-cancelling entries whose cosine is noise
    cos_dense, cos_all, sl_rel, worst = [], [], 0.0, None
    for k in G.files:
        if not k.startswith("gslice."):
            continue
        name, ref = k[len("gslice."):], G[k]
        if np.abs(ref).max() <= 1e-6 * norms.max():
            continue
        g = R["grads"][name]
        got_s = (g[:8, :64] if ref.ndim == 2 else g).ravel()
        r = ref.ravel()
        c = float(got_s @ r / (np.linalg.norm(got_s) * np.linalg.norm(r) + 1e-300))
        cos_all.append(c)
        if name.endswith(("dense.weight", "query.weight", "key.weight", "value.weight")) and "pooler" not in name:
            cos_dense.append(c)
            rel_ = float(np.abs(got_s - r).max() / np.abs(r).max())
            if rel_ > sl_rel:
                sl_rel, worst = rel_, name
    e["gslice_rel_to_max"], e["gslice_cos_min"] = sl_rel, float(min(cos_dense))
    e["gslice_cos_median_all"], e["gslice_worst"] = float(np.median(cos_all)), worst
    return e


def parity_report(dev, dtype="bf16", fixture="step_base_hot.npz"):
    """-> dict of measured errors of `dtype` against the reference golden, or None when the fixture is absent."""
    path = os.path.join(golden_dir(), fixture)
    if not os.path.exists(path):
        return None
    G = np.load(path)
    e = golden_errors(run_golden_step(G, dev, dtype), G)
    return {"fixture": fixture, "dtype": dtype,
            "logits_max_abs_err": round(e["sim_abs"], 4), "logits_scale": round(e["sim_scale"], 2),
            "logits_rel_err": round(e["sim_abs"] / e["sim_scale"], 5),
            "embeddings_max_abs_err": round(max(e["q_abs"], e["c_abs"]), 5), "loss_abs_err": round(e["loss_abs"], 5),
            "teacher_logits_max_abs_err": round(e["z_abs"], 5),
            "grad_norm_rel_err_median": round(e["gnorm_rel_median"], 5), "grad_norm_rel_err_max": round(e["gnorm_rel_max"], 5),
            "grad_slice_cosine_min_dense_weights": round(e["gslice_cos_min"], 6),
            "grad_slice_cosine_median_all_tensors": round(e["gslice_cos_median_all"], 6),
            "reference": "imported SimANS modules, fp64, same inputs (oracle/make_golden.py)"}
