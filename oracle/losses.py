"""NumPy restatement of the similarity + loss step bodies (forward and closed-form
backward).  TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference lines followed (under /root/reference):
  M1  local sim            SimANS/co_training/co_training_marco_train.py:199-202
  L1  MS-Pas KL distill    SimANS/co_training/co_training_marco_train.py:203-217
  L2  NQ/TQ normal+adv     SimANS/wiki/co_training_wiki_train.py:203-228
  L3  CrossBERTKDLoss      PROD/ProD_KD/model/models.py:668-781
  M2  dot_product_scores + BiEncoderNllLoss.calc   SimANS/model/models.py:468-505,564-572
  L4  BiEncoderKDLoss      PROD/ProD_KD/model/models.py:970-1059
  L5  MASTER fused         MASTER/finetune/MS/co_training_model.py:249-270
  L6  teacher CE           SimANS/co_training/co_training_marco_train.py:228-236
"""
import numpy as np

EPS = 1e-7


def softmax(x, axis=-1):
    x = x - x.max(axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis, keepdims=True)


def log_softmax(x, axis=-1):
    x = x - x.max(axis, keepdims=True)
    return x - np.log(np.exp(x).sum(axis, keepdims=True))


# ---- M1 -------------------------------------------------------------------
def sim_block(q, ctx):
    """q [B,H], ctx [B*(1+N),H] -> s [B,1+N] (einsum bh,bdh->bd)."""
    B, H = q.shape
    c = ctx.reshape(B, -1, H)
    return np.einsum("bh,bdh->bd", q, c)


def sim_block_bwd(q, ctx, ds):
    B, H = q.shape
    c = ctx.reshape(B, -1, H)
    dq = np.einsum("bd,bdh->bh", ds, c)
    dc = ds[:, :, None] * q[:, None, :]
    return dq, dc.reshape(-1, H)


# ---- L1 -------------------------------------------------------------------
def kl_distill(s, z, temperature=1.0, scale=1.0, grad_accum=1):
    """loss = KLDivLoss(batchmean)(log(softmax(s*scale)+eps), softmax(z/temperature)) / grad_accum.
    Returns (loss_after_accum_division, distill_loss, ds)."""
    B = s.shape[0]
    p = softmax(s * scale, 1)
    t = softmax(z / temperature, 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        tlogt = np.where(t > 0, t * np.log(np.where(t > 0, t, 1.0)), 0.0)
    distill = (tlogt - t * np.log(p + EPS)).sum() / B
    g = -t / (B * (p + EPS))
    ds = p * (g - (g * p).sum(1, keepdims=True)) * scale / grad_accum
    return distill / grad_accum, distill, ds


# ---- L2 -------------------------------------------------------------------
def wiki_normal_adv(s, z, temperature_normal=1.0, adv_lambda=0.0, scale=1.0, grad_accum=1):
    B = s.shape[0]
    p = softmax(s * scale, 1)
    t = softmax(z / temperature_normal, 1)
    # reward[b,d] = log(softmax([z[b,0], z[b,d]])[0] + eps)      (:214-219)
    m = np.maximum(z[:, :1], z)
    r = np.log(np.exp(z[:, :1] - m) / (np.exp(z[:, :1] - m) + np.exp(z - m)) + EPS)
    lp = np.log(p + EPS)
    normal = -(t * lp).sum() / B
    adv = (r * lp).sum()
    loss = adv_lambda * adv + (1.0 - adv_lambda) * normal
    g = (adv_lambda * r - (1.0 - adv_lambda) * t / B) / (p + EPS)
    ds = p * (g - (g * p).sum(1, keepdims=True)) * scale / grad_accum
    return loss / grad_accum, normal, adv, ds


# ---- L3 -------------------------------------------------------------------
def cross_kd(s, z, T=4.0, ce_w=0.1, kd_w=0.9, s_frozen=None, lwf_w=1.0):
    """CrossBERTKDLoss.calc with KD_type == 'KD_softmax', hard target 0."""
    B = s.shape[0]
    lsm = log_softmax(s, 1)
    hard = -lsm[:, 0].mean()
    correct = int((lsm.argmax(1) == 0).sum())
    p = np.exp(lsm)
    e0 = np.zeros_like(s)
    e0[:, 0] = 1.0

    def kd(zz):
        lpT = log_softmax(s / T, 1)
        u = softmax(zz / T, 1)
        with np.errstate(divide="ignore", invalid="ignore"):
            ulogu = np.where(u > 0, u * np.log(np.where(u > 0, u, 1.0)), 0.0)
        val = (ulogu - u * lpT).sum(1).mean() * T * T
        grad = (T / B) * (np.exp(lpT) - u)
        return val, grad

    soft, gsoft = kd(z)
    loss = ce_w * hard + kd_w * soft
    ds = ce_w * (p - e0) / B + kd_w * gsoft
    if s_frozen is not None:
        lw, glw = kd(s_frozen)
        loss = loss + lwf_w * lw
        ds = ds + lwf_w * glw
    return loss, hard, soft, correct, ds


# ---- M2 -------------------------------------------------------------------
def nll_inbatch(q, ctx, pos_idx, loss_scale=None):
    """BiEncoderNllLoss.calc.  Returns (loss, correct, dq, dctx, scores)."""
    S = q @ ctx.T
    lsm = log_softmax(S, 1)
    nq = q.shape[0]
    pos = np.asarray(pos_idx, dtype=np.int64)
    loss = -lsm[np.arange(nq), pos].mean()
    correct = int((lsm.argmax(1) == pos).sum())
    dS = np.exp(lsm)
    dS[np.arange(nq), pos] -= 1.0
    dS /= nq
    if loss_scale:
        loss = loss * loss_scale
        dS = dS * loss_scale
    return loss, correct, dS @ ctx, dS.T @ q, S


def nll_inbatch_distributed(q_ranks, ctx_ranks, rank, loss_scale=None):
    """caculate_cont_loss (PROD/ProD_base/train_DE_model_marco.py:224-278) as seen by
    ``rank``: global concat in rank order, positives at r*P + j*(P/B); gradient
    reaches only the local rows.  Returns (loss, correct, dq_local, dctx_local)."""
    W = len(q_ranks)
    pos, off = [], 0
    for r in range(W):
        B, Pn = q_ranks[r].shape[0], ctx_ranks[r].shape[0]
        pos += [off + j * (Pn // B) for j in range(B)]
        off += Pn
    Q = np.concatenate(q_ranks, 0)
    C = np.concatenate(ctx_ranks, 0)
    loss, correct, dQ, dC, _ = nll_inbatch(Q, C, pos, loss_scale)
    qo = sum(x.shape[0] for x in q_ranks[:rank])
    co = sum(x.shape[0] for x in ctx_ranks[:rank])
    return (loss, correct, dQ[qo:qo + q_ranks[rank].shape[0]],
            dC[co:co + ctx_ranks[rank].shape[0]])


# ---- L4 -------------------------------------------------------------------
def bi_kd(q, ctx, qT, ctxT, pos_idx, T=4.0, ce_w=0.1, kd_w=0.9):
    """BiEncoderKDLoss.calc, KD_softmax: all-pairs student vs teacher-embedding scores."""
    S = q @ ctx.T
    Z = qT @ ctxT.T
    nq = q.shape[0]
    pos = np.asarray(pos_idx, dtype=np.int64)
    lsm = log_softmax(S, 1)
    hard = -lsm[np.arange(nq), pos].mean()
    correct = int((lsm.argmax(1) == pos).sum())
    lpT = log_softmax(S / T, 1)
    u = softmax(Z / T, 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        ulogu = np.where(u > 0, u * np.log(np.where(u > 0, u, 1.0)), 0.0)
    soft = (ulogu - u * lpT).sum(1).mean() * T * T
    loss = ce_w * hard + kd_w * soft
    dS = np.exp(lsm)
    dS[np.arange(nq), pos] -= 1.0
    dS = ce_w * dS / nq + kd_w * (T / nq) * (np.exp(lpT) - u)
    return loss, hard, soft, correct, dS @ ctx, dS.T @ q


# ---- L6 -------------------------------------------------------------------
def teacher_ce(z, grad_accum=1):
    B = z.shape[0]
    lsm = log_softmax(z, 1)
    loss = -lsm[:, 0].mean()
    dz = np.exp(lsm)
    dz[:, 0] -= 1.0
    return loss / grad_accum, dz / (B * grad_accum)
