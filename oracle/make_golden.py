"""Generate tests/golden/* by running the IMPORTED REFERENCE in the build container.

TEST INFRASTRUCTURE.  Needs /root/reference (not present on the GPU box); only
the data files it writes travel.  Run:  python -m oracle.make_golden

What is imported from the reference and driven with synthetic inputs:
  SimANS/model/models.py        HFBertEncoder, BiBertEncoder, Reranker, BiEncoderNllLoss
  SimANS/utils/MARCO_until_new.py  Rocketqa_v2Dataset (sampler + collate)
  SimANS/utils/util_wiki.py        TraditionDataset   (sampler + collate)
  PROD/ProD_KD/model/models.py  CrossBERTKDLoss, BiEncoderKDLoss
and the literal step bodies co_training_marco_train.py:198-217 /
co_training_wiki_train.py:198-228 (the scripts themselves are not importable:
faiss / apex / transformers.AdamW are absent).

Each golden is also asserted against the NumPy oracle here, so a drift between
oracle and reference is caught at generation time, and again (oracle vs file)
by tests/test_oracle_golden.py on any box.
"""
import importlib.util
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

from . import bert as obert
from . import losses as oloss
from . import sampler as osamp
from .weights import BertCfg, TINY, make_bert_params, make_batch, normal

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _ref_models():
    sys.path.insert(0, os.path.join(REF, "SimANS"))
    return _load("ref_simans_models", os.path.join(REF, "SimANS/model/models.py"))


def _hf_dir(tmp, cfg, params, tag):
    """Write config.json + weights so that HFBertEncoder.init_encoder(args) can from_pretrained it."""
    from transformers import BertConfig
    d = os.path.join(tmp, tag)
    hf = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                    num_attention_heads=cfg.heads, intermediate_size=cfg.inter,
                    max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.type_vocab,
                    layer_norm_eps=cfg.eps, hidden_act="gelu")
    m = RM.HFBertEncoder(hf)
    sd = {k: torch.from_numpy(np.asarray(v, dtype=np.float32)) for k, v in params.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not [k for k in missing if "position_ids" not in k], missing
    assert not unexpected, unexpected
    m.save_pretrained(d)
    return d


def _no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    m.eval()
    return m


def _grads(model):
    return {k: p.grad.detach().numpy().astype(np.float64) for k, p in model.named_parameters()
            if p.grad is not None}


def _cmp(tag, a, b, tol):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    err = np.abs(a - b).max() if a.size else 0.0
    ref = max(1.0, np.abs(b).max()) if b.size else 1.0
    print("   %-42s max|diff| %.3e  (scale %.3e)" % (tag, err, ref))
    assert err <= tol * ref, (tag, err, tol * ref)


def gen_encoder_step(tmp, cfg_kw, tag, B, N, q_len, p_len, ce_len, seeds, full_grads, tol, std=0.02, extras=True,
                     q_stats=(9, 3, 4), p_stats=(80, 25, 16), t_stats=(90, 25, 20)):
    """Config-1 shaped retriever step of co_training_marco_train.py / co_training_wiki_train.py.
    extras=False keeps only the retriever step proper (L1): no NQ/TQ loss variants, no teacher train step -- used for
    the larger "hot" fixture whose only purpose is to put the persistent GEMM / wgrad / attention kernels, which need
    >= 16k tokens to dispatch, under a reference-generated golden."""
    cfg = BertCfg(**cfg_kw)
    Pq = make_bert_params(cfg, seeds[0], std=std)
    Pc = make_bert_params(cfg, seeds[1], std=std)
    Pt = make_bert_params(cfg, seeds[2], std=std)
    # extras=False (the big fixture): the imported model runs with ITS gradient checkpointing (models.py:73-74) -- same values,
    # and the fp64 autograd graph of ~20k tokens x 12 layers would not fit this container's 62 GB otherwise
    args = types.SimpleNamespace(model_type=_hf_dir(tmp, cfg, Pq, tag + "_q"), gradient_checkpointing=not extras,
                                 share_weight=False)
    model = RM.BiBertEncoder(args)
    # ctx tower gets its own weights
    sd = {k: torch.from_numpy(v) for k, v in Pc.items()}
    model.ctx_model.load_state_dict(sd, strict=False)
    _no_dropout(model)
    teacher = RM.Reranker(RM.HFBertEncoder.init_encoder(
        types.SimpleNamespace(gradient_checkpointing=False), model_type=_hf_dir(tmp, cfg, Pt, tag + "_t")), cfg.hidden)
    wcls = normal(seeds[2], "qa_classifier.weight", (1, cfg.hidden), 0.05).astype(np.float32)
    bcls = normal(seeds[2], "qa_classifier.bias", (1,), 0.05).astype(np.float32)
    with torch.no_grad():
        teacher.qa_classifier.weight.copy_(torch.from_numpy(wcls))
        teacher.qa_classifier.bias.copy_(torch.from_numpy(bcls))
    _no_dropout(teacher)

    P = B * (1 + N)
    q_ids, q_mask, _ = make_batch(seeds[0] + 100, B, q_len, cfg.vocab, *q_stats)
    c_ids, c_mask, _ = make_batch(seeds[1] + 100, P, p_len, cfg.vocab, *p_stats)
    t_ids, t_mask, _ = make_batch(seeds[2] + 100, P, ce_len, cfg.vocab, *t_stats)
    t_ids3, t_mask3 = t_ids.reshape(B, 1 + N, ce_len), t_mask.reshape(B, 1 + N, ce_len)
    tt = lambda a: torch.from_numpy(a)

    # fp32 run of the imported reference (as shipped) -- kept for the record
    with torch.no_grad():
        q32, c32 = model(query_ids=tt(q_ids), attention_mask_q=tt(q_mask), input_ids_a=tt(c_ids), attention_mask_a=tt(c_mask))
        z32 = teacher(input_ids=tt(t_ids3), attention_mask=tt(t_mask3))
    # golden = the same imported modules in float64 (removes fp32 cancellation noise from grads)
    model.double(); teacher.double()
    if not extras:
        model.train()                      # HF checkpoints only in training mode; every Dropout has p = 0 (_no_dropout)
        assert model.ctx_model.is_gradient_checkpointing

    # --- literal step body, co_training_marco_train.py:198-217 (L1) -----------------
    model.zero_grad()
    local_q, local_ctx = model(query_ids=tt(q_ids), attention_mask_q=tt(q_mask),
                               input_ids_a=tt(c_ids), attention_mask_a=tt(c_mask))
    ctx3 = local_ctx.reshape(local_q.size(0), local_ctx.size(0) // local_q.size(0), -1)
    sim = torch.einsum("bh,bdh->bd", local_q, ctx3)
    p_s = torch.nn.functional.softmax(sim, dim=1)
    with torch.no_grad():
        z = teacher(input_ids=tt(t_ids3), attention_mask=tt(t_mask3))
        p_t = torch.nn.functional.softmax(z / 1.0, dim=1)
    loss = torch.nn.KLDivLoss(reduction="batchmean")((p_s + 1e-7).log(), p_t)
    loss.backward()
    G = _grads(model)
    local_q, local_ctx, sim, z, loss_v = local_q.detach(), local_ctx.detach(), sim.detach(), z.detach(), loss.item()
    del loss, p_s, p_t, ctx3                # drop the autograd graph before the oracle allocates its own caches
    loss = types.SimpleNamespace(item=lambda: loss_v)

    out = dict(q_ids=q_ids, q_mask=q_mask, c_ids=c_ids, c_mask=c_mask, t_ids=t_ids3, t_mask=t_mask3,
               qa_w=wcls, qa_b=bcls,
               q_emb_fp32=q32.numpy(), ctx_emb_fp32=c32.numpy(), teacher_logits_fp32=z32.numpy(), std=np.float64(std),
               q_emb=local_q.detach().numpy(), ctx_emb=local_ctx.detach().numpy(),
               sim=sim.detach().numpy(), teacher_logits=z.numpy(), loss_kl=np.float64(loss.item()),
               cfg=json.dumps(cfg.as_dict()), seeds=np.asarray(seeds), shape=np.asarray([B, N, q_len, p_len, ce_len]))

    # --- oracle agreement (fp64) ------------------------------------------------------
    print(" [%s] oracle vs imported reference" % tag)
    # (the big fixture checks the oracle's FORWARD only: its fp64 backward caches for ~33k padded tokens x 12 layers do not
    # fit this container's memory; the oracle's backward is pinned by the small fixtures)
    _, oq, cq = obert.bert_forward(Pq, q_ids, q_mask, cfg.heads, keep=extras)
    _, oc, cc = obert.bert_forward(Pc, c_ids, c_mask, cfg.heads, keep=extras)
    Pt2 = {"encoder." + k: v for k, v in Pt.items()}
    Pt2["qa_classifier.weight"], Pt2["qa_classifier.bias"] = wcls, bcls
    oz, _, _ = obert.reranker_forward(Pt2, t_ids3, t_mask3, cfg.heads, keep=False)
    _cmp("q_emb (vs fp32 reference)", oq, out["q_emb_fp32"], 2e-5)
    _cmp("ctx_emb (vs fp32 reference)", oc, out["ctx_emb_fp32"], 2e-5)
    _cmp("teacher_logits (vs fp32 reference)", oz, out["teacher_logits_fp32"], 2e-5)
    _cmp("q_emb", oq, out["q_emb"], tol)
    _cmp("ctx_emb", oc, out["ctx_emb"], tol)
    _cmp("teacher_logits", oz, out["teacher_logits"], tol)
    osim = oloss.sim_block(oq, oc)
    _cmp("sim", osim, out["sim"], tol)
    ol, od, ods = oloss.kl_distill(osim, oz)
    _cmp("loss_kl", ol, out["loss_kl"], tol)
    if extras:
        dq, dc = oloss.sim_block_bwd(oq, oc, ods)
        Gq = obert.bert_backward(Pq, q_ids, q_mask, cfg.heads, cq, dq)
        Gc = obert.bert_backward(Pc, c_ids, c_mask, cfg.heads, cc, dc)
        worst = 0.0
        for pre, Go in (("question_model.", Gq), ("ctx_model.", Gc)):
            for k, g in Go.items():
                r = G[pre + k]
                scale = max(np.abs(r).max(), 1e-6)
                worst = max(worst, np.abs(g - r).max() / scale)
        print("   %-42s worst rel-to-max grad diff %.3e" % ("all %d parameter grads" % (len(Gq) + len(Gc)), worst))
        assert worst < 1e3 * tol, worst

    # --- NQ/TQ loss on the same embeddings, co_training_wiki_train.py:198-228 (L2) ----
    for lam in ((0.0, 0.5) if extras else ()):
        model.zero_grad()
        lq, lc = model(query_ids=tt(q_ids), attention_mask_q=tt(q_mask), input_ids_a=tt(c_ids), attention_mask_a=tt(c_mask))
        rs = torch.einsum("bh,bdh->bd", lq, lc.reshape(lq.size(0), lc.size(0) // lq.size(0), -1))
        rp = torch.nn.functional.softmax(rs, dim=1)
        with torch.no_grad():
            probs = torch.nn.functional.softmax(z / 1.0, dim=1)
            pos = z[:, :1]
            reward_logits = torch.stack((pos.expand(z.size()), z), -1)
            reward = torch.log(torch.nn.functional.softmax(reward_logits, dim=2)[:, :, 0] + 1e-7)
        normal_loss = (-probs * torch.log(rp + 1e-7)).sum() / rp.size(0)
        adv_loss = (reward * torch.log(rp + 1e-7)).sum()
        l2 = lam * adv_loss + (1 - lam) * normal_loss
        out["loss_wiki_lam%g" % lam] = np.float64(l2.item())
        o2, _, _, _ = oloss.wiki_normal_adv(osim, oz, 1.0, lam)
        _cmp("loss_wiki lam=%g" % lam, o2, l2.item(), tol)

    # --- teacher (reranker) train step, co_training_marco_train.py:225-245 (L6 + E3 backward) ---------
    if not extras:
        names = sorted(G.keys())
        out["grad_names"] = np.asarray(names)
        out["grad_norms"] = np.asarray([np.sqrt((G[k] ** 2).sum()) for k in names])
        for k in names:                      # a slice of EVERY 2-D gradient and every vector: element-wise bf16 error statistics
            out["gslice." + k] = G[k][:8, :64] if G[k].ndim == 2 else G[k]
        np.savez_compressed(os.path.join(OUT, "step_%s.npz" % tag), **out)
        return
    teacher.zero_grad()
    rl = teacher(input_ids=tt(t_ids3), attention_mask=tt(t_mask3))
    contr_loss = torch.nn.CrossEntropyLoss()(rl, torch.zeros(rl.size(0), dtype=torch.long))
    contr_loss.backward()
    TG = _grads(teacher)
    out["teacher_ce_loss"] = np.float64(contr_loss.item())
    ozk, ocls, ocache = obert.reranker_forward(Pt2, t_ids3, t_mask3, cfg.heads, keep=True)
    otl, otd = oloss.teacher_ce(ozk)
    _cmp("teacher_ce_loss", otl, contr_loss.item(), tol)
    OTG = obert.reranker_backward(Pt2, t_ids3, t_mask3, cfg.heads, ocache, ocls, otd)
    worst = 0.0
    for k, g in OTG.items():
        r = TG[k]
        worst = max(worst, np.abs(g.reshape(r.shape) - r).max() / max(np.abs(r).max(), 1e-6))
    print("   %-42s worst rel-to-max grad diff %.3e" % ("all %d teacher grads" % len(OTG), worst))
    assert worst < 1e3 * tol, worst
    if full_grads:
        for k, g in TG.items():
            out["tgrad." + k] = g
    else:
        tn = sorted(TG.keys())
        out["tgrad_names"] = np.asarray(tn)
        out["tgrad_norms"] = np.asarray([np.sqrt((TG[k] ** 2).sum()) for k in tn])
        out["tgslice.encoder.encoder.layer.7.attention.output.dense.weight"] = TG["encoder.encoder.layer.7.attention.output.dense.weight"][:8, :64]
        out["tgslice.encoder.encoder.layer.0.intermediate.dense.bias"] = TG["encoder.encoder.layer.0.intermediate.dense.bias"]
        out["tgslice.qa_classifier.weight"] = TG["qa_classifier.weight"]

    # --- gradients kept in the fixture ---------------------------------------------------
    if full_grads:
        for k, g in G.items():
            out["grad." + k] = g
    else:
        # BERT-base: norms of every grad + a few slices
        names = sorted(G.keys())
        out["grad_names"] = np.asarray(names)
        out["grad_norms"] = np.asarray([np.sqrt((G[k] ** 2).sum()) for k in names])
        for k in ("question_model.encoder.layer.0.attention.self.query.weight",
                  "ctx_model.encoder.layer.11.output.dense.weight",
                  "ctx_model.encoder.layer.5.intermediate.dense.weight"):
            out["gslice." + k] = G[k][:8, :64]
        for k in ("ctx_model.embeddings.LayerNorm.weight", "question_model.encoder.layer.11.output.LayerNorm.bias",
                  "ctx_model.encoder.layer.3.attention.self.key.bias"):
            out["gslice." + k] = G[k]
    np.savez_compressed(os.path.join(OUT, "step_%s.npz" % tag), **out)


def gen_losses():
    """Loss classes driven on random [B,1+N] / [Q,C] tensors with torch autograd."""
    PR = _load("ref_prod_models", os.path.join(REF, "PROD/ProD_KD/model/models.py"))
    rs = np.random.RandomState(7)
    out = {}
    B, D, H = 6, 16, 32
    s = rs.randn(B, D) * 3.0
    z = rs.randn(B, D) * 2.0
    sf = rs.randn(B, D) * 3.0
    out.update(s=s, z=z, s_frozen=sf)
    T = torch.tensor

    def run(fn, *xs):
        ts = [T(x, dtype=torch.float64, requires_grad=True) for x in xs]
        l = fn(*ts)
        l.backward()
        return l.item(), [t.grad.numpy() for t in ts]

    # L1 (with and without scale_simmila, temperature_distill 1 and 2, grad_accum 2)
    for tag, temp, scale, ga in (("a", 1.0, 1.0, 1), ("b", 2.0, 1.0 / np.sqrt(768.0), 2)):
        def f(st):
            p = torch.nn.functional.softmax(st * scale, dim=1)
            pt = torch.nn.functional.softmax(T(z) / temp, dim=1)
            return torch.nn.KLDivLoss(reduction="batchmean")((p + 1e-7).log(), pt) / ga
        l, (g,) = run(f, s)
        out["L1%s_loss" % tag], out["L1%s_ds" % tag] = l, g
        ol, _, ods = oloss.kl_distill(s, z, temp, scale, ga)
        _cmp("L1%s loss" % tag, ol, l, 1e-12); _cmp("L1%s ds" % tag, ods, g, 1e-12)
    # L3 CrossBERTKDLoss (KD_softmax), without and with LwF
    a = types.SimpleNamespace(KD_type="KD_softmax", TEMPERATURE=4.0, CE_WEIGHT=0.1, KD_WEIGHT=0.9, LwF_WEIGHT=1.0)
    q = rs.randn(B, H); c = rs.randn(B * D, H); qo = rs.randn(B, H); co = rs.randn(B * D, H)
    out.update(q=q, c=c, qo=qo, co=co)
    import warnings
    warnings.simplefilter("ignore")
    for lwf in (False, True):
        def f(qt, ct):
            l, corr = PR.CrossBERTKDLoss().calc(a, qt, ct, T(z), LwF=lwf, ori_q_vector=T(qo), ori_ctx_vectors=T(co))
            f.corr = int(corr)
            return l
        l, (gq, gc) = run(f, q, c)
        key = "L3lwf" if lwf else "L3"
        out[key + "_loss"], out[key + "_dq"], out[key + "_dc"], out[key + "_correct"] = l, gq, gc, f.corr
        sim = oloss.sim_block(q, c)
        ol, _, _, corr, ods = oloss.cross_kd(sim, z, 4.0, 0.1, 0.9, oloss.sim_block(qo, co) if lwf else None, 1.0)
        odq, odc = oloss.sim_block_bwd(q, c, ods)
        _cmp(key + " loss", ol, l, 1e-12); _cmp(key + " dq", odq, gq, 1e-12); _cmp(key + " dc", odc, gc, 1e-12)
        assert corr == f.corr
    # M2 BiEncoderNllLoss (SimANS copy)
    Q, C = 5, 20
    q2 = rs.randn(Q, H); c2 = rs.randn(C, H); pos = [0, 4, 8, 12, 16]
    out.update(q2=q2, c2=c2, pos=np.asarray(pos))
    def f(qt, ct):
        l, corr = RM.BiEncoderNllLoss().calc(qt, ct, pos)
        f.corr = int(corr)
        return l
    l, (gq, gc) = run(f, q2, c2)
    out.update(M2_loss=l, M2_dq=gq, M2_dc=gc, M2_correct=f.corr)
    ol, corr, odq, odc, _ = oloss.nll_inbatch(q2, c2, pos)
    _cmp("M2 loss", ol, l, 1e-12); _cmp("M2 dq", odq, gq, 1e-12); _cmp("M2 dc", odc, gc, 1e-12)
    assert corr == f.corr
    # L4 BiEncoderKDLoss
    qT = rs.randn(Q, H); cT = rs.randn(C, H)
    out.update(qT=qT, cT=cT)
    def f(qt, ct):
        l, corr = PR.BiEncoderKDLoss().calc(a, qt, ct, T(qT), T(cT), pos)
        f.corr = int(corr)
        return l
    l, (gq, gc) = run(f, q2, c2)
    out.update(L4_loss=l, L4_dq=gq, L4_dc=gc, L4_correct=f.corr)
    ol, _, _, corr, odq, odc = oloss.bi_kd(q2, c2, qT, cT, pos, 4.0, 0.1, 0.9)
    _cmp("L4 loss", ol, l, 1e-12); _cmp("L4 dq", odq, gq, 1e-12); _cmp("L4 dc", odc, gc, 1e-12)
    # L6 teacher CE
    def f(zt):
        return torch.nn.CrossEntropyLoss()(zt, torch.zeros(B, dtype=torch.long))
    l, (g,) = run(f, z)
    out.update(L6_loss=l, L6_dz=g)
    ol, odz = oloss.teacher_ce(z)
    _cmp("L6 loss", ol, l, 1e-12); _cmp("L6 dz", odz, g, 1e-12)
    # distributed gather semantics (train_DE_model_marco.py:224-278), simulated W=2 in-process
    W, Bq, Np = 2, 3, 4
    qr = [rs.randn(Bq, H) for _ in range(W)]
    cr = [rs.randn(Bq * Np, H) for _ in range(W)]
    for r in range(W):
        lq = T(qr[r], dtype=torch.float64, requires_grad=True)
        lc = T(cr[r], dtype=torch.float64, requires_grad=True)
        gq_, gc_, posi, tot = [], [], [], 0
        for i in range(W):
            if i != r:
                gq_.append(T(qr[i])); gc_.append(T(cr[i]))
            else:
                gq_.append(lq); gc_.append(lc)
            posi += [v + tot for v in [j * Np for j in range(Bq)]]
            tot += cr[i].shape[0]
        l, corr = RM.BiEncoderNllLoss().calc(torch.cat(gq_, 0), torch.cat(gc_, 0), posi)
        l.backward()
        out["dist_q%d" % r], out["dist_c%d" % r] = qr[r], cr[r]
        out["dist_loss%d" % r], out["dist_dq%d" % r], out["dist_dc%d" % r] = l.item(), lq.grad.numpy(), lc.grad.numpy()
        ol, _, odq, odc = oloss.nll_inbatch_distributed(qr, cr, r)
        _cmp("dist loss r%d" % r, ol, l.item(), 1e-12); _cmp("dist dq r%d" % r, odq, lq.grad.numpy(), 1e-12)
        _cmp("dist dc r%d" % r, odc, lc.grad.numpy(), 1e-12)
    np.savez_compressed(os.path.join(OUT, "losses.npz"), **out)


class _StubTok:
    """Minimal tokenizer: passage text is its pid as a decimal string; encodes to [101, 1000+pid, 102]."""
    sep_token_id, pad_token_id = 102, 0

    def encode(self, text, text_pair=None, add_special_tokens=True, max_length=None, truncation=True,
               pad_to_max_length=False):
        body = text_pair if text_pair is not None else text
        try:
            v = 1000 + int(str(body).strip())
        except ValueError:
            v = 999
        return [101, v, 102] if add_special_tokens else [v]


def gen_sampler(tmp):
    MU = _load("ref_marco_until_new", os.path.join(REF, "SimANS/utils/MARCO_until_new.py"))
    rs = np.random.RandomState(11)
    n_q, C, N = 12, 40, 15
    para = os.path.join(tmp, "corpus")
    os.makedirs(para, exist_ok=True)
    with open(os.path.join(para, "para.txt"), "w") as f, open(os.path.join(para, "para.title.txt"), "w") as g:
        for pid in range(5000):
            f.write("%d\t%d\n" % (pid, pid)); g.write("%d\t-\n" % pid)
    rows, meta = [], []
    for qi in range(n_q):
        s_pos = float(np.round(70 + 20 * rs.rand(), 4))
        if qi == 3:
            s_pos = 0.0                       # positive not retrieved -> fallback branch
        pids = rs.choice(np.arange(1, 5000), size=C + 1, replace=False)
        scores = np.sort(np.round((s_pos if s_pos else 80.0) - np.abs(rs.randn(C)) * 1.5, 4))[::-1]
        pos = "%d %s" % (pids[0], repr(s_pos))
        neg = ",".join("%d %s" % (p, repr(float(s))) for p, s in zip(pids[1:], scores))
        rows.append("%d\tq%d\t%s\t%s" % (qi, qi, pos, neg))
        meta.append(dict(s_pos=s_pos, cand=[int(p) for p in pids[1:]], scores=[float(s) for s in scores]))
    tsv = os.path.join(tmp, "train.tsv")
    with open(tsv, "w") as f:
        f.write("\n".join(rows) + "\n")
    ds = MU.Rocketqa_v2Dataset(tsv, _StubTok(), num_hard_negatives=N, corpus_path=para)
    random.seed(1234)
    picked = []
    for i in range(n_q):
        qtok, ctx, ce = ds[i]
        picked.append([int(v) - 1000 for v in ctx[1:, 1].tolist()])
    # oracle literal replay with the same seed
    random.seed(1234)
    for i in range(n_q):
        m = meta[i]
        random.choice([0])                                     # the reference's random.choice(pos_pairs_list)
        _, negs = osamp.reference_draw(random, m["cand"], m["scores"], m["s_pos"], N, osamp.LAPLACE, tau=3.0)
        assert negs == picked[i], (i, negs, picked[i])
        w = osamp.weights(m["scores"], m["s_pos"], osamp.LAPLACE, tau=3.0)
        m["weights_laplace"] = w
        m["weights_gauss_nq"] = osamp.weights(m["scores"], m["s_pos"], osamp.GAUSS, a=0.5, b=1.0)
        m["picked"] = picked[i]
    print(" [sampler] literal replay == imported Rocketqa_v2Dataset for %d queries (seed 1234)" % n_q)
    # collate shape check of the reference
    coll = MU.Rocketqa_v2Dataset.get_collate_fn(None)
    random.seed(5)
    batch = coll([ds[i] for i in range(4)])
    shapes = dict(q=list(batch["student"][0].shape), ctx=list(batch["student"][2].shape),
                  ce=list(batch["teacher"][0].shape), pos=batch["student"][4])
    # TraditionDataset (NQ/TQ gaussian form)
    UW = _load("ref_util_wiki", os.path.join(REF, "SimANS/utils/util_wiki.py"))
    data = []
    for qi in range(6):
        m = meta[qi]
        sp = m["s_pos"] if m["s_pos"] else 75.0
        data.append(dict(question="q %d" % qi, answers=["a"],
                         positive_ctxs=[dict(text="%d" % (9000 + qi), title="-", score=sp, passage_id=9000 + qi)],
                         hard_negative_ctxs=[dict(text="%d" % p, title="-", score=s, passage_id=p)
                                             for p, s in zip(m["cand"], m["scores"])]))
    js = os.path.join(tmp, "nq.json")
    with open(js, "w") as f:
        json.dump(data, f)
    td = UW.TraditionDataset(js, _StubTok(), num_hard_negatives=N, a=0.5, b=1.0)
    random.seed(77)
    wiki_picked = []
    for i in range(6):
        _, ctx_tok, _, _, _ = td[i]
        wiki_picked.append([t[1] - 1000 for t in ctx_tok[1:]])
    # literal replay
    random.seed(77)
    for i in range(6):
        m = meta[i]
        sp = m["s_pos"] if m["s_pos"] else 75.0
        order = list(range(len(m["cand"])))
        random.shuffle(order)
        cand = [m["cand"][j] for j in order]
        sc = [m["scores"][j] for j in order]
        union, _ = osamp.reference_draw(random, cand, sc, sp, N, osamp.GAUSS, a=0.5, b=1.0)
        sel = [c for c in cand if c in union][0:N]
        assert sel == wiki_picked[i], (i, sel, wiki_picked[i])
    print(" [sampler] literal replay == imported TraditionDataset for 6 queries (seed 77)")
    with open(os.path.join(OUT, "sampler_ref.json"), "w") as f:
        json.dump(dict(N=N, tau=3.0, queries=meta, collate_shapes=shapes, wiki_picked=wiki_picked,
                       wiki_seed=77, marco_seed=1234), f)


class _VarTok(object):
    """Variable-length stand-in for the HF tokenizer (third-party, no vocab in this image): deterministic token ids from
    the text, [CLS] a [SEP] b [SEP], truncated to max_length keeping the final [SEP] -- except that every 5th passage
    is returned WITHOUT the trailing [SEP] so the reference's other remove_special_token branch is exercised."""
    sep_token_id, pad_token_id, cls_token_id = 102, 0, 101

    @staticmethod
    def _toks(text, n):
        h = 1469598103934665603
        out = []
        for i in range(n):
            for ch in (str(text) + "/%d" % i):
                h = ((h ^ ord(ch)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            out.append(1000 + h % 29000)
        return out

    def encode(self, text, text_pair=None, add_special_tokens=True, max_length=None, truncation=True, pad_to_max_length=False):
        key = sum(ord(c) for c in (str(text) + "|" + str(text_pair)))
        a = self._toks(text, 2 + key % 7 if text_pair is not None else 3 + key % 40)
        ids = [101] + a + [102]
        if text_pair is not None:
            ids += self._toks(text_pair, 5 + (key * 7) % 170) + [102]
        if max_length is not None and len(ids) > max_length:
            ids = ids[:max_length - 1] + [102]
        if text_pair is not None and key % 5 == 0:
            ids = ids[:-1]
        return ids


def gen_collate(tmp):
    """Batch assembly of the imported Rocketqa_v2Dataset + its collate (MARCO_until_new.py:204-258) -> collate_ref.npz."""
    from . import collate as ocoll
    MU = _load("ref_marco_until_new2", os.path.join(REF, "SimANS/utils/MARCO_until_new.py"))
    rs = np.random.RandomState(21)
    n_q, C, N = 6, 30, 15
    para = os.path.join(tmp, "corpus2")
    os.makedirs(para, exist_ok=True)
    with open(os.path.join(para, "para.txt"), "w") as f, open(os.path.join(para, "para.title.txt"), "w") as g:
        for pid in range(3000):
            f.write("%d\tpassage text number %d\n" % (pid, pid * 7919)); g.write("%d\ttitle %d\n" % (pid, pid % 97))
    rows = []
    for qi in range(n_q):
        s_pos = float(np.round(70 + 20 * rs.rand(), 4))
        pids = rs.choice(np.arange(1, 3000), size=C + 1, replace=False)
        scores = np.sort(np.round(s_pos - np.abs(rs.randn(C)) * 1.5, 4))[::-1]
        rows.append("%d\tquery words %d %s\t%d %s\t%s" % (qi, qi, "x " * (qi * 9), pids[0], repr(s_pos),
                                                        ",".join("%d %s" % (p, repr(float(s))) for p, s in zip(pids[1:], scores))))
    tsv = os.path.join(tmp, "train2.tsv")
    with open(tsv, "w") as f:
        f.write("\n".join(rows) + "\n")
    ds = MU.Rocketqa_v2Dataset(tsv, _VarTok(), num_hard_negatives=N, corpus_path=para)
    random.seed(99)
    batch = MU.Rocketqa_v2Dataset.get_collate_fn(None)([ds[i] for i in range(n_q)])
    q, qm, doc, dm, pos = batch["student"]
    ce, cem, tgt = batch["teacher"]
    q, doc, ce = q.numpy(), doc.numpy(), ce.numpy()
    assert (doc[:, -1] == 102).any() and (np.array([doc[r, ocoll.row_len(doc[r], 0) - 1] for r in range(len(doc))]) != 102).any()
    # the oracle, fed the reference's own padded rows as the pre-tokenised pool, reproduces the teacher tensors exactly
    o = ocoll.assemble(q, doc, list(range(n_q)), list(range(len(doc))), 1 + N)
    assert (o["ce_ids"] == ce).all() and (o["ce_mask"] == cem.numpy()).all() and (o["q_mask"] == qm.numpy()).all()
    assert (o["ctx_mask"] == dm.numpy()).all() and o["positive_ctx_indices"] == pos and (o["tgt"] == tgt.numpy()).all()
    print(" [collate] oracle == imported Rocketqa_v2Dataset + create_biencoder_input2 (%d queries x %d passages, max ce len %d)"
          % (n_q, 1 + N, int(o["ce_len"].max())))
    np.savez_compressed(os.path.join(OUT, "collate_ref.npz"), q_ids=q, q_mask=qm.numpy(), ctx_ids=doc, ctx_mask=dm.numpy(),
                        ce_ids=ce, ce_mask=cem.numpy(), tgt=tgt.numpy(), pos=np.asarray(pos))


def gen_roberta_dot(tmp, use_mean=False):
    """E4: the imported RobertaDot (SimANS/model/models.py:334-359) on a tiny RoBERTa config, shared encoder for queries
    and documents, with the MS-Doc step's KL-distill loss (co_training_doc_train.py:209-224) -> roberta_dot_tiny.npz;
    use_mean=True: the masked-mean pooling of EmbeddingMixin (models.py:296-305) -> roberta_dot_mean_tiny.npz (the
    embeddings, the loss and a subset of the gradients: the whole hidden state carries gradient on that path)."""
    import transformers
    cfgd = dict(vocab=1000, hidden=64, layers=2, heads=4, inter=128, max_pos=140, type_vocab=1, eps=1e-5, pooler=False)
    cfg = BertCfg(**cfgd)
    hf = transformers.RobertaConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers,
                                    num_attention_heads=cfg.heads, intermediate_size=cfg.inter,
                                    max_position_embeddings=cfg.max_pos, type_vocab_size=1, layer_norm_eps=1e-5,
                                    hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, pad_token_id=1)
    hf.return_dict = False                      # the reference only sets this for transformers 4 (models.py:338-339)
    import types
    model = RM.RobertaDot(hf, types.SimpleNamespace(use_mean=True)) if use_mean else RM.RobertaDot(hf)
    assert bool(model.use_mean) == use_mean
    P = {"roberta." + k: v for k, v in make_bert_params(cfg, 4321, std=0.08).items()}
    P["embeddingHead.weight"] = normal(4321, "embeddingHead.weight", (cfg.hidden, cfg.hidden), 0.08).astype(np.float32)
    P["embeddingHead.bias"] = normal(4321, "embeddingHead.bias", (cfg.hidden,), 0.02).astype(np.float32)
    P["norm.weight"] = (1.0 + normal(4321, "norm.weight", (cfg.hidden,), 0.05)).astype(np.float32)
    P["norm.bias"] = normal(4321, "norm.bias", (cfg.hidden,), 0.02).astype(np.float32)
    missing, unexpected = model.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=False)
    assert not unexpected and all("position_ids" in m or "token_type_ids" in m for m in missing), (missing, unexpected)
    model.eval()
    B, N = 4, 3
    def batch(seed, n, S, mean, std, lo):
        ids, mask, lens = make_batch(seed, n, S, cfg.vocab, mean, std, lo)
        ids = np.where(mask == 1, ids, 1)               # RoBERTa: pad id 1, mask = ids != 1 (MARCO_until_Doc.py:200-203)
        ids[:, 0] = 0                                   # <s>
        ids[np.arange(n), lens - 1] = 2                 # </s>
        return ids, (ids != 1).astype(np.int64)
    q_ids, q_mask = batch(501, B, 32, 9, 3, 4)
    d_ids, d_mask = batch(502, B * (1 + N), 128, 80, 25, 16)
    z = normal(503, "teacher", (B, 1 + N), 2.0)
    tt = lambda a: torch.from_numpy(a)
    with torch.no_grad():
        q32 = model(tt(q_ids), tt(q_mask), True).numpy()
        d32 = model(tt(d_ids), tt(d_mask), False).numpy()
    model.double()
    model.zero_grad()
    q = model(tt(q_ids), tt(q_mask), True)
    d = model(tt(d_ids), tt(d_mask), False)
    sim = torch.einsum("bh,bdh->bd", q, d.reshape(B, 1 + N, -1))
    loss = torch.nn.KLDivLoss(reduction="batchmean")((torch.softmax(sim, 1) + 1e-7).log(), torch.softmax(tt(z), 1))
    loss.backward()
    G = _grads(model)
    print(" [roberta_dot] oracle vs imported reference")
    oq, cq = obert.roberta_dot_forward(P, q_ids, q_mask, cfg.heads, use_mean=use_mean)
    od, cd = obert.roberta_dot_forward(P, d_ids, d_mask, cfg.heads, use_mean=use_mean)
    _cmp("q_emb (vs fp32 reference)", oq, q32, 2e-5)
    _cmp("doc_emb (vs fp32 reference)", od, d32, 2e-5)
    _cmp("q_emb", oq, q.detach().numpy(), 1e-11)
    _cmp("doc_emb", od, d.detach().numpy(), 1e-11)
    osim = oloss.sim_block(oq, od)
    ol, _, ods = oloss.kl_distill(osim, z)
    _cmp("loss", ol, loss.item(), 1e-11)
    dq, dd = oloss.sim_block_bwd(oq, od, ods)
    Gq = obert.roberta_dot_backward(P, q_ids, q_mask, cfg.heads, cq, dq)
    Gd = obert.roberta_dot_backward(P, d_ids, d_mask, cfg.heads, cd, dd)
    worst = 0.0
    for k in Gq:
        g = Gq[k] + Gd[k]                              # shared encoder: both passes accumulate
        r = G[k]
        worst = max(worst, np.abs(g.reshape(r.shape) - r).max() / max(np.abs(r).max(), 1e-6))
    print("   %-42s worst rel-to-max grad diff %.3e" % ("all %d parameter grads" % len(Gq), worst))
    assert worst < 1e-8, worst
    out = dict(q_ids=q_ids, q_mask=q_mask, d_ids=d_ids, d_mask=d_mask, teacher=z, cfg=json.dumps(cfgd), seed=np.int64(4321),
               q_emb=q.detach().numpy(), d_emb=d.detach().numpy(), q_emb_fp32=q32, d_emb_fp32=d32, sim=sim.detach().numpy(),
               loss=np.float64(loss.item()))
    for k in ("embeddingHead.weight", "embeddingHead.bias", "norm.weight", "norm.bias"):
        out["param." + k] = P[k]
    keep = None if not use_mean else ("embeddingHead.", "norm.", "roberta.embeddings.", "roberta.encoder.layer.0.attention.self.query",
                                      "roberta.encoder.layer.1.output.", "roberta.encoder.layer.1.attention.output.")
    for k, g in G.items():
        if keep is None or k.startswith(keep):
            out["grad." + k] = g
    out["use_mean"] = np.int64(use_mean)
    np.savez_compressed(os.path.join(OUT, "roberta_dot_mean_tiny.npz" if use_mean else "roberta_dot_tiny.npz"), **out)


def gen_prod_step(tmp):
    """BASELINE configs[3] as ONE step, from the imported PROD modules (PROD/ProD_KD/model/models.py: HFBertEncoder,
    BiBertEncoder, Reranker with its binary head, CrossBERTKDLoss) driven by the literal step body of
    PROD/ProD_KD/run_progressive_distill_marco.py:288-314 with the README recipe (PROD/README.md:208-224): 6-layer
    bi-encoder student, 12-layer cross-encoder teacher, frozen student copy (--open_LwF), KD_softmax, T=4,
    CE_WEIGHT 0.1, KD_WEIGHT 0.9, LwF_WEIGHT 1.0, B=8 queries x (1+15) passages, q32 / p128 / cross-encoder 160.
    PROD's init_encoder hard-codes /colab_space paths (models.py:34-43), so the encoders are built by HFBertEncoder(cfg)
    -- the same class, the same forward.  -> step_prod_cfg4.npz (grad norms + slices)."""
    from transformers import BertConfig
    PR = _load("ref_prod_models2", os.path.join(REF, "PROD/ProD_KD/model/models.py"))
    B, N, q_len, p_len, ce_len = 8, 15, 32, 128, 160
    seeds = (3234, 3235, 3236, 3237, 3238)           # student q / ctx, teacher, frozen copy q / ctx
    scfg, tcfg = BertCfg(layers=6), BertCfg(layers=12)

    def enc(cfg, P):
        hf = BertConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads,
                        intermediate_size=cfg.inter, max_position_embeddings=cfg.max_pos, type_vocab_size=cfg.type_vocab,
                        layer_norm_eps=cfg.eps, hidden_act="gelu")
        m = PR.HFBertEncoder(hf)
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(v) for k, v in P.items()}, strict=False)
        assert not [k for k in missing if "position_ids" not in k] and not unexpected, (missing, unexpected)
        return m

    def bi(Pq, Pc):
        m = PR.BiBertEncoder.__new__(PR.BiBertEncoder)
        torch.nn.Module.__init__(m)
        m.question_model, m.ctx_model = enc(scfg, Pq), enc(scfg, Pc)
        return _no_dropout(m)
    Ps = [make_bert_params(scfg, s_) for s_ in (seeds[0], seeds[1], seeds[3], seeds[4])]
    Pt = make_bert_params(tcfg, seeds[2])
    model, student_copy = bi(Ps[0], Ps[1]), bi(Ps[2], Ps[3])
    teacher = PR.Reranker(enc(tcfg, Pt), tcfg.hidden)
    wcls = normal(seeds[2], "qa_classifier.weight", (1, tcfg.hidden), 0.05).astype(np.float32)
    bcls = normal(seeds[2], "qa_classifier.bias", (1,), 0.05).astype(np.float32)
    with torch.no_grad():
        teacher.qa_classifier.weight.copy_(torch.from_numpy(wcls))
        teacher.qa_classifier.bias.copy_(torch.from_numpy(bcls))
    _no_dropout(teacher)
    P = B * (1 + N)
    q_ids, q_mask, _ = make_batch(seeds[0] + 100, B, q_len, scfg.vocab, 9, 3, 4)
    c_ids, c_mask, _ = make_batch(seeds[1] + 100, P, p_len, scfg.vocab, 80, 25, 16)
    t_ids, t_mask, _ = make_batch(seeds[2] + 100, P, ce_len, scfg.vocab, 90, 25, 20)
    t_ids3, t_mask3 = t_ids.reshape(B, 1 + N, ce_len), t_mask.reshape(B, 1 + N, ce_len)
    tt = lambda a: torch.from_numpy(a)
    args = types.SimpleNamespace(KD_type="KD_softmax", TEMPERATURE=4.0, CE_WEIGHT=0.1, KD_WEIGHT=0.9, LwF_WEIGHT=1.0,
                                 open_LwF=True, gradient_accumulation_steps=1)
    inputs_retriever = dict(query_ids=tt(q_ids), attention_mask_q=tt(q_mask), input_ids_a=tt(c_ids), attention_mask_a=tt(c_mask))
    inputs_reranker = dict(input_ids=tt(t_ids3), attention_mask=tt(t_mask3))
    model.double(); teacher.double(); student_copy.double()
    import warnings
    warnings.simplefilter("ignore")
    # --- literal step body, run_progressive_distill_marco.py:288-314 -------------------------------------------------
    model.zero_grad()
    teacher.eval()
    local_q_vector, local_ctx_vectors = model(**inputs_retriever)
    with torch.no_grad():
        binary_logits, relevance_logits, _ = teacher(**inputs_reranker)
    student_copy.eval()
    ori_q_vector, ori_ctx_vectors = student_copy(**inputs_retriever)
    loss, is_correct = PR.CrossBERTKDLoss().calc(args, local_q_vector, local_ctx_vectors, relevance_logits, LwF=True,
                                                 ori_q_vector=ori_q_vector, ori_ctx_vectors=ori_ctx_vectors)
    loss = loss / args.gradient_accumulation_steps
    loss.backward()
    G = _grads(model)
    # the same step without LwF (the `else` branch, :305-313)
    with torch.no_grad():
        loss_nolwf, _ = PR.CrossBERTKDLoss().calc(args, local_q_vector.detach(), local_ctx_vectors.detach(), relevance_logits)
    print(" [prod_cfg4] oracle vs imported PROD modules")
    _, oq, cq = obert.bert_forward(Ps[0], q_ids, q_mask, scfg.heads)
    _, oc, cc = obert.bert_forward(Ps[1], c_ids, c_mask, scfg.heads)
    _, oq0, _ = obert.bert_forward(Ps[2], q_ids, q_mask, scfg.heads)
    _, oc0, _ = obert.bert_forward(Ps[3], c_ids, c_mask, scfg.heads)
    Pt2 = {"encoder." + k: v for k, v in Pt.items()}
    Pt2["qa_classifier.weight"], Pt2["qa_classifier.bias"] = wcls, bcls
    oz, _, _ = obert.reranker_forward(Pt2, t_ids3, t_mask3, tcfg.heads, keep=False)
    _cmp("q_emb", oq, local_q_vector.detach().numpy(), 1e-10)
    _cmp("ctx_emb", oc, local_ctx_vectors.detach().numpy(), 1e-10)
    _cmp("relevance_logits", oz, relevance_logits.numpy(), 1e-10)
    osim = oloss.sim_block(oq, oc)
    ol, _, _, ocorr, ods = oloss.cross_kd(osim, oz, 4.0, 0.1, 0.9, oloss.sim_block(oq0, oc0), 1.0)
    _cmp("loss (LwF)", ol, loss.item(), 1e-10)
    assert ocorr == int(is_correct)
    dq, dc = oloss.sim_block_bwd(oq, oc, ods)
    Gq = obert.bert_backward(Ps[0], q_ids, q_mask, scfg.heads, cq, dq)
    Gc = obert.bert_backward(Ps[1], c_ids, c_mask, scfg.heads, cc, dc)
    worst = 0.0
    for pre, Go in (("question_model.", Gq), ("ctx_model.", Gc)):
        for k, g in Go.items():
            r = G[pre + k]
            worst = max(worst, np.abs(g - r).max() / max(np.abs(r).max(), 1e-6))
    print("   %-42s worst rel-to-max grad diff %.3e" % ("all %d parameter grads" % (len(Gq) + len(Gc)), worst))
    assert worst < 1e-7, worst
    names = sorted(G.keys())
    out = dict(q_ids=q_ids, q_mask=q_mask, c_ids=c_ids, c_mask=c_mask, t_ids=t_ids3, t_mask=t_mask3, qa_w=wcls, qa_b=bcls,
               seeds=np.asarray(seeds), shape=np.asarray([B, N, q_len, p_len, ce_len]), layers=np.asarray([6, 12]),
               q_emb=local_q_vector.detach().numpy(), ctx_emb=local_ctx_vectors.detach().numpy(),
               ori_q_emb=ori_q_vector.detach().numpy(), ori_ctx_emb=ori_ctx_vectors.detach().numpy(),
               relevance_logits=relevance_logits.numpy(), loss=np.float64(loss.item()), loss_nolwf=np.float64(loss_nolwf.item()),
               correct=np.int64(int(is_correct)), grad_names=np.asarray(names),
               grad_norms=np.asarray([np.sqrt((G[k] ** 2).sum()) for k in names]))
    for k in names:
        out["gslice." + k] = G[k][:8, :64] if G[k].ndim == 2 else G[k]
    np.savez_compressed(os.path.join(OUT, "step_prod_cfg4.npz"), **out)


class _HFAdamW(torch.optim.Optimizer):
    """transformers.AdamW as published in transformers 2.x-4.x (`optimization.py`, class AdamW, correct_bias=True) -- the
    optimizer the reference builds (co_training_marco_train.py:21-25, 57-69).  It was REMOVED in transformers 5 (this
    container has 5.15.0), so its update rule is restated here as a torch Optimizer and driven exactly like the reference
    drives the original: per parameter  exp_avg = b1*exp_avg + (1-b1)*g ; exp_avg_sq = b2*exp_avg_sq + (1-b2)*g*g ;
    step_size = lr*sqrt(1-b2^t)/(1-b1^t) ; p -= step_size * exp_avg/(sqrt(exp_avg_sq)+eps) ; then p -= lr*wd*p."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"], st["exp_avg"], st["exp_avg_sq"] = 0, torch.zeros_like(p), torch.zeros_like(p)
                b1, b2 = group["betas"]
                st["step"] += 1
                st["exp_avg"].mul_(b1).add_(p.grad, alpha=1.0 - b1)
                st["exp_avg_sq"].mul_(b2).addcmul_(p.grad, p.grad, value=1.0 - b2)
                denom = st["exp_avg_sq"].sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * (1.0 - b2 ** st["step"]) ** 0.5 / (1.0 - b1 ** st["step"])
                p.addcdiv_(st["exp_avg"], denom, value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])


def gen_trajectory(tmp):
    """SURVEY 8(c) last row: the optimiser trajectory of the retriever job -- 4 steps of the literal loop body
    (co_training_marco_train.py:198-217 step, :246-254 clip_grad_norm_ -> optimizer.step() -> scheduler.step() ->
    zero_grad) on the tiny model in fp64: get_optimizer's two parameter groups (:57-69), transformers'
    get_linear_schedule_with_warmup (the imported function) with 2 warm-up steps, clipping ACTIVE.  Recorded: loss and
    pre-clip gradient norm of every step, the L2 norm of every tensor's cumulative update after every step, and the updates
    themselves for a few tensors.  -> trajectory_tiny.npz"""
    from transformers import get_linear_schedule_with_warmup
    cfg = BertCfg(**TINY)
    seeds, std = (1234, 1235, 1236), 0.08
    Pq, Pc = make_bert_params(cfg, seeds[0], std=std), make_bert_params(cfg, seeds[1], std=std)
    args = types.SimpleNamespace(model_type=_hf_dir(tmp, cfg, Pq, "traj_q"), gradient_checkpointing=False, share_weight=False)
    model = RM.BiBertEncoder(args)
    model.ctx_model.load_state_dict({k: torch.from_numpy(v) for k, v in Pc.items()}, strict=False)
    _no_dropout(model)
    model.double()
    B, N = 4, 3
    q_ids, q_mask, _ = make_batch(seeds[0] + 100, B, 32, cfg.vocab, 9, 3, 4)
    c_ids, c_mask, _ = make_batch(seeds[1] + 100, B * (1 + N), 128, cfg.vocab, 80, 25, 16)
    z = normal(77, "traj_teacher", (B, 1 + N), 2.0)
    tt = lambda a: torch.from_numpy(a)
    lr, eps, max_norm, warm, total, steps = 1e-3, 1e-8, 2.0, 2, 10, 4
    no_decay = ['bias', 'LayerNorm.weight']
    groups = [{'params': [p for n, p in model.named_parameters() if p.requires_grad and not any(nd in n for nd in no_decay)], 'weight_decay': 0.0},
              {'params': [p for n, p in model.named_parameters() if p.requires_grad and any(nd in n for nd in no_decay)], 'weight_decay': 0.0}]
    optimizer = _HFAdamW(groups, lr=lr, eps=eps)
    scheduler = get_linear_schedule_with_warmup(optimizer, num_warmup_steps=warm, num_training_steps=total)
    p0 = {k: v.detach().clone() for k, v in model.named_parameters()}
    names = [k for k, _ in model.named_parameters()]
    keep = ("question_model.encoder.layer.0.attention.self.query.weight", "ctx_model.encoder.layer.1.output.dense.weight",
            "ctx_model.embeddings.LayerNorm.weight", "ctx_model.encoder.layer.0.intermediate.dense.bias",
            "question_model.embeddings.position_embeddings.weight", "ctx_model.pooler.dense.weight")
    out = dict(q_ids=q_ids, q_mask=q_mask, c_ids=c_ids, c_mask=c_mask, teacher=z, cfg=json.dumps(cfg.as_dict()), seeds=np.asarray(seeds),
               std=np.float64(std), lr=np.float64(lr), eps=np.float64(eps), max_grad_norm=np.float64(max_norm), warmup=np.int64(warm),
               total=np.int64(total), names=np.asarray(names))
    losses, gnorms, lrs = [], [], []
    for it in range(steps):
        model.train()                                  # (:196; every Dropout has p = 0)
        local_q, local_ctx = model(query_ids=tt(q_ids), attention_mask_q=tt(q_mask), input_ids_a=tt(c_ids), attention_mask_a=tt(c_mask))
        sim = torch.einsum("bh,bdh->bd", local_q, local_ctx.reshape(local_q.size(0), local_ctx.size(0) // local_q.size(0), -1))
        loss = torch.nn.KLDivLoss(reduction="batchmean")((torch.nn.functional.softmax(sim, dim=1) + 1e-7).log(),
                                                         torch.nn.functional.softmax(tt(z) / 1.0, dim=1))
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        lrs.append(optimizer.param_groups[0]["lr"])
        optimizer.step()
        scheduler.step()
        model.zero_grad()
        losses.append(loss.item()); gnorms.append(float(gn))
        cur = dict(model.named_parameters())
        out["dnorm%d" % it] = np.asarray([float((cur[k].detach() - p0[k]).norm()) for k in names])
        for k in keep:
            d = (cur[k].detach() - p0[k]).numpy()
            out["delta%d.%s" % (it, k)] = d[:8, :64] if d.ndim == 2 else d
    out.update(losses=np.asarray(losses), grad_norms=np.asarray(gnorms), lrs=np.asarray(lrs))
    print(" [trajectory] losses %s  pre-clip grad norms %s  lrs %s" % (np.round(losses, 6), np.round(gnorms, 4), lrs))
    assert max(gnorms) > max_norm > min(gnorms) and lrs[0] == 0.0 and losses[-1] < losses[0]     # clipping active AND inactive

    np.savez_compressed(os.path.join(OUT, "trajectory_tiny.npz"), **out)


def main():
    global RM
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    RM = _ref_models()
    with tempfile.TemporaryDirectory() as tmp:
        if "--only-traj" in sys.argv:
            gen_trajectory(tmp)
            return
        if "--only-prod" in sys.argv:
            gen_prod_step(tmp)
            return
        if "--only-large" in sys.argv:
            # BASELINE config 5's geometry at full depth: BERT-large (24 layers, H = 1024, 16 heads, F = 4096), one query of up
            # to 128 tokens and two documents of up to 512 (MS-Doc lengths), cross-encoder rows of up to 512; fp64 reference
            gen_encoder_step(tmp, dict(hidden=1024, layers=24, heads=16, inter=4096), "large_cfg5", B=1, N=1, q_len=128, p_len=512,
                             ce_len=512, seeds=(3234, 3235, 3236), full_grads=False, tol=1e-9, extras=False,
                             q_stats=(40, 20, 8), p_stats=(470, 40, 300), t_stats=(480, 30, 300))
            return
        if "--only-hot" in sys.argv:
            gen_encoder_step(tmp, {}, "base_hot", B=16, N=15, q_len=32, p_len=128, ce_len=160,
                             seeds=(2234, 2235, 2236), full_grads=False, tol=1e-10, extras=False)
            return
        gen_losses()
        gen_sampler(tmp)
        gen_collate(tmp)
        gen_roberta_dot(tmp)
        gen_roberta_dot(tmp, use_mean=True)
        gen_trajectory(tmp)
        if "--only-small" in sys.argv:
            return
        gen_encoder_step(tmp, TINY, "tiny", B=4, N=3, q_len=32, p_len=128, ce_len=160,
                         seeds=(1234, 1235, 1236), full_grads=True, tol=1e-11, std=0.08)
        if "--no-base" not in sys.argv:
            gen_encoder_step(tmp, {}, "base_cfg1", B=4, N=1, q_len=32, p_len=128, ce_len=160,
                             seeds=(1234, 1235, 1236), full_grads=False, tol=1e-10)
        if "--hot" in sys.argv or "--all" in sys.argv:
            gen_prod_step(tmp)
            # BERT-base, B=16 queries x 16 passages: ~20k passage tokens -> the persistent NT GEMM, the 256x256 wgrad kernel,
            # mha<8> and the wide LayerNorm kernels all dispatch (minutes of fp64 CPU time; not part of the default run)
            gen_encoder_step(tmp, {}, "base_hot", B=16, N=15, q_len=32, p_len=128, ce_len=160,
                             seeds=(2234, 2235, 2236), full_grads=False, tol=1e-10, extras=False)
    print("golden vectors written to", OUT)


if __name__ == "__main__":
    main()
