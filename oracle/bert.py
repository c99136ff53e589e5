"""NumPy restatement of the BERT encoder forward/backward used by
``HFBertEncoder`` / ``BiBertEncoder`` / ``Reranker``.

TEST INFRASTRUCTURE (see oracle/__init__.py).

Spec followed (file:line under /root/reference):
  * embeddings: word + token_type(0) + position(0..S-1) -> LayerNorm(eps 1e-12)
    LEAD/modeling_bert.py:181-240
  * self-attention: QKV linears, QK^T / sqrt(d) + (1-mask)*finfo.min, softmax, PV
    LEAD/modeling_bert.py:243-374
  * attention output / FFN output: LN(dense(x) + residual)
    LEAD/modeling_bert.py:377-388, 455-466
  * intermediate: erf-GELU(dense(x))     LEAD/modeling_bert.py:440-452
  * pooler output multiplied by 0 and [CLS] slice of last hidden returned
    SimANS/model/models.py:77-82  (so pooler grads are exact zeros)
Dropout is OFF (reference parity is taken in eval()-semantics with grads on,
SURVEY 8c).  Works on the padded [n,S] layout exactly like the reference; the
GPU product uses a packed (varlen) layout, so agreement is a real check.
"""
import numpy as np
from scipy.special import erf

SQRT1_2 = 0.7071067811865476
INV_SQRT_2PI = 0.3989422804014327


def _ln_fwd(x, g, b, eps):
    mu = x.mean(-1, keepdims=True)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True)
    rstd = 1.0 / np.sqrt(var + eps)
    xhat = xc * rstd
    return xhat * g + b, (xhat, rstd)


def _ln_bwd(dy, cache, g):
    xhat, rstd = cache
    dg = (dy * xhat).reshape(-1, xhat.shape[-1]).sum(0)
    db = dy.reshape(-1, xhat.shape[-1]).sum(0)
    dxh = dy * g
    dx = rstd * (dxh - dxh.mean(-1, keepdims=True) - xhat * (dxh * xhat).mean(-1, keepdims=True))
    return dx, dg, db


def gelu(u):
    return 0.5 * u * (1.0 + erf(u * SQRT1_2))


def gelu_grad(u):
    return 0.5 * (1.0 + erf(u * SQRT1_2)) + u * np.exp(-0.5 * u * u) * INV_SQRT_2PI


def _mix32(seed, stream, row, colquad):
    """uint32 hash of the product's dropout mask (simxns_amd/csrc/common.h drop_mix), vectorised."""
    M = np.uint64(0xFFFFFFFF)
    row = np.asarray(row, dtype=np.uint64)
    cq = np.asarray(colquad, dtype=np.uint64)
    h = ((row * np.uint64(0x9E3779B1)) & M) ^ ((((cq + np.uint64((stream * 0x632BE5AB) & 0xFFFFFFFF)) & M) * np.uint64(0x85EBCA77)) & M) ^ np.uint64(seed & 0xFFFFFFFF)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x7FEB352D)) & M
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x846CA68B)) & M
    h ^= h >> np.uint64(16)
    return h


def drop_threshold(p):
    """-> (thr, scale) of the stateless mask: one hash serves four columns (its bytes against an 8-bit threshold), so the drop
    probability is realised in steps of 1/256 -- thr = round(256 p), 26/256 for p = 0.1 -- and kept values are scaled by the
    reciprocal of the REALISED keep rate, 256 / (256 - thr): E[multiplier] = 1 exactly (common.h make_drop)."""
    thr = int(np.float32(p) * np.float32(256.0) + np.float32(0.5))
    if np.float32(p) >= np.float32(1.0):
        return 256, 0.0                                                         # p >= 1: everything dropped, like torch (no 256 / 0)
    thr = min(255, thr) if np.float32(p) >= np.float32(1.0 / 512.0) else 0      # p < 1/512: no dropout (never clamped UP to 1/256)
    return thr, float(np.float32(256.0) / np.float32(256 - thr))


def drop_multipliers(p, seed, stream, rows, cols):
    """[len(rows), len(cols)] multipliers (0 or 256/(256-thr)) of the stateless dropout mask (include/simx.h simx_dropout):
    element (row, col) is kept iff byte (col & 3) of mix32(seed, stream, row, col >> 2) is >= thr."""
    rows = np.asarray(rows, dtype=np.uint64)[:, None]
    cols = np.asarray(cols, dtype=np.uint64)[None, :]
    if p <= 0:
        return np.ones((rows.shape[0], cols.shape[1]))
    thr, scale = drop_threshold(p)
    h = _mix32(seed, stream, rows, cols >> np.uint64(2))
    lane = (h >> ((cols & np.uint64(3)) * np.uint64(8))) & np.uint64(0xFF)
    return np.where(lane >= thr, scale, 0.0)


def _drop_masks(drop, ids, mask, heads, layers):
    """Multipliers in the PADDED layout for every dropout site, from the packed-row hash definition.
    -> dict: ('h', layer, site) -> [n,S,H] ; ('a', layer) -> [n,heads,S,S]"""
    n, S = ids.shape
    lens = mask.sum(1).astype(np.int64)
    cu = np.concatenate([[0], np.cumsum(lens)])
    T = int(cu[-1])
    out = {}
    H = drop["H"]
    trow = np.zeros((n, S), dtype=np.int64)          # packed token index (pad positions: 0, never used)
    for s_ in range(n):
        trow[s_, :lens[s_]] = cu[s_] + np.arange(lens[s_])
    for layer in range(-1, layers):
        for site in ((0,) if layer == -1 else (1, 2)):
            m = drop_multipliers(drop["p_hidden"], drop["seed"], (layer + 1) * 8 + site, trow.reshape(-1), np.arange(H))
            out[("h", layer, site)] = m.reshape(n, S, H)
        if layer >= 0:
            a = np.ones((n, heads, S, S))
            if drop["p_attn"] > 0:
                for hh in range(heads):
                    m = drop_multipliers(drop["p_attn"], drop["seed"], (layer + 1) * 8 + 3, (hh * T + trow).reshape(-1), np.arange(S))
                    a[:, hh] = m.reshape(n, S, S)
            out[("a", layer)] = a
    return out


def bert_forward(P, ids, mask, heads, eps=1e-12, dtype=np.float64, keep=True, prefix="", drop=None, pos_offset=0):
    """Returns (seq [n,S,H], cls [n,H], caches).  ``P``: HF-keyed dict.
    ``drop`` = dict(p_hidden, p_attn, seed) enables training-mode dropout with the product's stateless masks.
    ``pos_offset``: position id of the first token (RoBERTa: padding_idx + 1 = 2, HF create_position_ids_from_input_ids
    on right-padded rows; BERT: 0)."""
    g = lambda k: np.asarray(P[prefix + k], dtype=dtype)
    n, S = ids.shape
    H = g("embeddings.word_embeddings.weight").shape[1]
    d = H // heads
    fmin = float(np.finfo(np.float32).min)
    emb = (g("embeddings.word_embeddings.weight")[ids]
           + g("embeddings.position_embeddings.weight")[None, pos_offset:pos_offset + S]
           + g("embeddings.token_type_embeddings.weight")[0][None, None])
    x, ln0 = _ln_fwd(emb, g("embeddings.LayerNorm.weight"), g("embeddings.LayerNorm.bias"), eps)
    bias = ((1.0 - mask.astype(dtype)) * fmin)[:, None, None, :]          # [n,1,1,S]
    caches = {"ln0": ln0, "layers": []} if keep else None
    L = 0
    while (prefix + "encoder.layer.%d.attention.self.query.weight" % L) in P:
        L += 1
    DM = None
    if drop is not None:
        DM = _drop_masks(dict(drop, H=H), ids, mask, heads, L)
        x = x * DM[("h", -1, 0)]
        if keep:
            caches["DM"] = DM
    for i in range(L):
        p = "encoder.layer.%d." % i
        def lin(t, name):
            return t @ g(p + name + ".weight").T + g(p + name + ".bias")
        q = lin(x, "attention.self.query").reshape(n, S, heads, d).transpose(0, 2, 1, 3)
        k = lin(x, "attention.self.key").reshape(n, S, heads, d).transpose(0, 2, 1, 3)
        v = lin(x, "attention.self.value").reshape(n, S, heads, d).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) / np.sqrt(d) + bias
        s = s - s.max(-1, keepdims=True)
        e = np.exp(s)
        pr = e / e.sum(-1, keepdims=True)
        prd = pr * DM[("a", i)] if DM is not None else pr
        ctx = (prd @ v).transpose(0, 2, 1, 3).reshape(n, S, H)
        a = lin(ctx, "attention.output.dense")
        if DM is not None:
            a = a * DM[("h", i, 1)]
        x1, ln1 = _ln_fwd(a + x, g(p + "attention.output.LayerNorm.weight"),
                          g(p + "attention.output.LayerNorm.bias"), eps)
        u = lin(x1, "intermediate.dense")
        h = gelu(u)
        y = lin(h, "output.dense")
        if DM is not None:
            y = y * DM[("h", i, 2)]
        x2, ln2 = _ln_fwd(y + x1, g(p + "output.LayerNorm.weight"), g(p + "output.LayerNorm.bias"), eps)
        if keep:
            caches["layers"].append(dict(x=x, q=q, k=k, v=v, pr=pr, prd=prd, ctx=ctx, ln1=ln1, x1=x1, u=u, h=h, ln2=ln2))
        x = x2
    return x, x[:, 0, :].copy(), caches


def bert_backward(P, ids, mask, heads, caches, d_cls, d_seq=None, dtype=np.float64, prefix="", pos_offset=0):
    """Gradients of sum(cls*d_cls) (+ sum(seq*d_seq)) w.r.t. every parameter.
    Pad positions receive zero upstream gradient (they never reach the loss in
    the reference either: only [CLS] is used, and real tokens do not attend to
    pad keys)."""
    g = lambda k: np.asarray(P[prefix + k], dtype=dtype)
    n, S = ids.shape
    H = g("embeddings.word_embeddings.weight").shape[1]
    d = H // heads
    G = {}
    dx = np.zeros((n, S, H), dtype=dtype)
    dx[:, 0, :] += d_cls
    if d_seq is not None:
        dx += d_seq
    L = len(caches["layers"])
    DM = caches.get("DM")
    flat = lambda t: t.reshape(-1, t.shape[-1])

    def lin_bwd(dy, x_in, name, p):
        G[prefix + p + name + ".weight"] = flat(dy).T @ flat(x_in)
        G[prefix + p + name + ".bias"] = flat(dy).sum(0)
        return dy @ g(p + name + ".weight")

    for i in reversed(range(L)):
        p = "encoder.layer.%d." % i
        c = caches["layers"][i]
        dz, dg_, db_ = _ln_bwd(dx, c["ln2"], g(p + "output.LayerNorm.weight"))
        G[prefix + p + "output.LayerNorm.weight"], G[prefix + p + "output.LayerNorm.bias"] = dg_, db_
        dzd = dz * DM[("h", i, 2)] if DM is not None else dz
        dh = lin_bwd(dzd, c["h"], "output.dense", p)
        du = dh * gelu_grad(c["u"])
        dx1 = dz + lin_bwd(du, c["x1"], "intermediate.dense", p)
        dz1, dg_, db_ = _ln_bwd(dx1, c["ln1"], g(p + "attention.output.LayerNorm.weight"))
        G[prefix + p + "attention.output.LayerNorm.weight"] = dg_
        G[prefix + p + "attention.output.LayerNorm.bias"] = db_
        dz1d = dz1 * DM[("h", i, 1)] if DM is not None else dz1
        dctx = lin_bwd(dz1d, c["ctx"], "attention.output.dense", p)
        dctx = dctx.reshape(n, S, heads, d).transpose(0, 2, 1, 3)
        dpr = dctx @ c["v"].transpose(0, 1, 3, 2)
        if DM is not None:
            dpr = dpr * DM[("a", i)]
        dv = c["prd"].transpose(0, 1, 3, 2) @ dctx
        ds = c["pr"] * (dpr - (dpr * c["pr"]).sum(-1, keepdims=True))
        ds = ds / np.sqrt(d)
        dq = ds @ c["k"]
        dk = ds.transpose(0, 1, 3, 2) @ c["q"]
        back = lambda t: t.transpose(0, 2, 1, 3).reshape(n, S, H)
        dxa = (lin_bwd(back(dq), c["x"], "attention.self.query", p)
               + lin_bwd(back(dk), c["x"], "attention.self.key", p)
               + lin_bwd(back(dv), c["x"], "attention.self.value", p))
        dx = dz1 + dxa
    if DM is not None:
        dx = dx * DM[("h", -1, 0)]
    demb, dg_, db_ = _ln_bwd(dx, caches["ln0"], g("embeddings.LayerNorm.weight"))
    G[prefix + "embeddings.LayerNorm.weight"], G[prefix + "embeddings.LayerNorm.bias"] = dg_, db_
    # only real tokens contribute in the product (packed layout); in the padded
    # reference pad rows have zero upstream grad as well when only CLS is used.
    demb = demb * mask[..., None].astype(dtype) if d_seq is None else demb
    gw = np.zeros_like(g("embeddings.word_embeddings.weight"))
    np.add.at(gw, ids.reshape(-1), flat(demb))
    G[prefix + "embeddings.word_embeddings.weight"] = gw
    gp = np.zeros_like(g("embeddings.position_embeddings.weight"))
    gp[pos_offset:pos_offset + S] = demb.sum(0)
    G[prefix + "embeddings.position_embeddings.weight"] = gp
    gt = np.zeros_like(g("embeddings.token_type_embeddings.weight"))
    gt[0] = flat(demb).sum(0)
    G[prefix + "embeddings.token_type_embeddings.weight"] = gt
    if (prefix + "pooler.dense.weight") in P:
        G[prefix + "pooler.dense.weight"] = np.zeros_like(g("pooler.dense.weight"))
        G[prefix + "pooler.dense.bias"] = np.zeros_like(g("pooler.dense.bias"))
    return G


def reranker_forward(P, ids3, mask3, heads, dtype=np.float64, keep=True):
    """Reranker.forward (SimANS/model/models.py:647-659): [N,M,L] -> logits [N,M]."""
    N, M, Lq = ids3.shape
    seq, cls, caches = bert_forward(P, ids3.reshape(N * M, Lq), mask3.reshape(N * M, Lq), heads,
                                    dtype=dtype, keep=keep, prefix="encoder.")
    w = np.asarray(P["qa_classifier.weight"], dtype=dtype)
    b = np.asarray(P["qa_classifier.bias"], dtype=dtype)
    logits = (cls @ w.T + b).reshape(N, M)
    return logits, cls, caches


def reranker_backward(P, ids3, mask3, heads, caches, cls, dlogits, dtype=np.float64):
    """Backward of Reranker.forward: d logits [N,M] -> gradients of `encoder.*` and `qa_classifier.*`
    (teacher train step, co_training_marco_train.py:225-245)."""
    N, M, Lq = ids3.shape
    w = np.asarray(P["qa_classifier.weight"], dtype=dtype)
    dl = np.asarray(dlogits, dtype=dtype).reshape(N * M, 1)
    d_cls = dl @ w
    G = bert_backward(P, ids3.reshape(N * M, Lq), mask3.reshape(N * M, Lq), heads, caches, d_cls, dtype=dtype,
                      prefix="encoder.")
    G["qa_classifier.weight"] = dl.T @ cls
    G["qa_classifier.bias"] = dl.sum(0)
    return G


# ---- E4: RobertaDot (SimANS/model/models.py:277-359) -----------------------------------------
def roberta_dot_forward(P, ids, mask, heads, eps=1e-5, head_eps=1e-5, pad_id=1, dtype=np.float64, keep=True, use_mean=False):
    """emb = LayerNorm(embeddingHead(pool(roberta(ids, mask)[0]))) with pool = row 0 (use_mean == False, the
    from_pretrained default, models.py:283-286) or the masked mean over the real tokens (models.py:296-305).
    ``P`` keys: roberta.* (no pooler), embeddingHead.{weight,bias}, norm.{weight,bias}."""
    seq, cls, caches = bert_forward(P, ids, mask, heads, eps=eps, dtype=dtype, keep=keep, prefix="roberta.",
                                    pos_offset=pad_id + 1)
    if use_mean:
        m = np.asarray(mask, dtype)[..., None]
        cls = (seq * m).sum(1) / m.sum(1)
    w, b = np.asarray(P["embeddingHead.weight"], dtype), np.asarray(P["embeddingHead.bias"], dtype)
    z = cls @ w.T + b
    y, ln = _ln_fwd(z, np.asarray(P["norm.weight"], dtype), np.asarray(P["norm.bias"], dtype), head_eps)
    return y, dict(enc=caches, cls=cls, ln=ln, use_mean=use_mean)


def roberta_dot_backward(P, ids, mask, heads, cache, d_emb, pad_id=1, dtype=np.float64):
    dz, dg, db = _ln_bwd(np.asarray(d_emb, dtype), cache["ln"], np.asarray(P["norm.weight"], dtype))
    w = np.asarray(P["embeddingHead.weight"], dtype)
    d_in = dz @ w
    if cache.get("use_mean"):
        m = np.asarray(mask, dtype)[..., None]
        d_seq = d_in[:, None, :] * m / m.sum(1, keepdims=True)
        G = bert_backward(P, ids, mask, heads, cache["enc"], np.zeros_like(d_in), d_seq=d_seq, dtype=dtype, prefix="roberta.",
                          pos_offset=pad_id + 1)
    else:
        G = bert_backward(P, ids, mask, heads, cache["enc"], d_in, dtype=dtype, prefix="roberta.", pos_offset=pad_id + 1)
    G["norm.weight"], G["norm.bias"] = dg, db
    G["embeddingHead.weight"], G["embeddingHead.bias"] = dz.T @ cache["cls"], dz.sum(0)
    return G
