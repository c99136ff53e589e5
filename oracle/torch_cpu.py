"""torch-CPU restatement of the retriever step -- the CPU BASELINE leg of bench.py, and a second checker.

TEST / MEASUREMENT INFRASTRUCTURE (see oracle/__init__.py): imported only by tests/, by bench.py's `cpu_baseline` leg and by
oracle/time_reference.py; never by the product.

Why a second restatement beside oracle/bert.py (NumPy): the reference's CPU path IS torch (HF BertModel on torch CPU
kernels: oneDNN GEMMs, fused SDPA, vectorised LayerNorm / GELU), and a NumPy port of it is ~5x slower than the real thing
on the same cores, which would flatter the GPU.  This module issues the same torch operators the reference's modules
issue -- embedding gathers, F.linear, scaled_dot_product_attention with the additive key mask, erf-GELU, LayerNorm(eps
1e-12), the pooler-times-zero term (SimANS/model/models.py:77-82), einsum similarity, softmax / KLDivLoss(batchmean)
(co_training_marco_train.py:198-217), autograd backward -- without importing transformers or the reference, so it travels to
the GPU box.  oracle/time_reference.py times it against the IMPORTED reference in the build container (same step, same
threads); tests/test_oracle_golden.py checks its outputs against the reference goldens.

Spec followed: LEAD/modeling_bert.py:181-240 (embeddings), :243-374 (self-attention), :377-388 / :455-466 (dense +
residual + LayerNorm), :440-452 (erf-GELU), :649-662 (pooler).
"""
import math
import time

import numpy as np
import torch
import torch.nn.functional as F


def usable_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (cpu.max / cfs_quota).  The GPU
    boxes show 256 hardware threads but a 16-CPU quota: 256 OpenMP threads on 16 CPUs spin against the throttle and run
    orders of magnitude slower than 16 threads."""
    import math
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.ceil(int(q) / int(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except Exception:
            pass
    return n


def to_torch_params(P, dtype=torch.float32, requires_grad=True, prefix=""):
    return {k[len(prefix):]: torch.from_numpy(np.asarray(v)).to(dtype).requires_grad_(requires_grad)
            for k, v in P.items() if k.startswith(prefix)}


def bert_forward(P, ids, mask, heads, eps=1e-12):
    """P: HF-keyed dict of torch tensors.  ids / mask: int64 [n,S].  -> (sequence_output, pooled = seq[:, 0])."""
    n, S = ids.shape
    H = P["embeddings.word_embeddings.weight"].shape[1]
    d = H // heads
    x = (F.embedding(ids, P["embeddings.word_embeddings.weight"])
         + P["embeddings.position_embeddings.weight"][:S].unsqueeze(0)
         + P["embeddings.token_type_embeddings.weight"][0])
    x = F.layer_norm(x, (H,), P["embeddings.LayerNorm.weight"], P["embeddings.LayerNorm.bias"], eps)
    # additive key mask (1 - mask) * finfo.min, broadcast over heads and queries (modeling_bert.py:349-354)
    bias = ((1.0 - mask.to(x.dtype)) * torch.finfo(x.dtype).min)[:, None, None, :]
    L = 0
    while ("encoder.layer.%d.attention.self.query.weight" % L) in P:
        L += 1
    for i in range(L):
        p = "encoder.layer.%d." % i
        lin = lambda t, name: F.linear(t, P[p + name + ".weight"], P[p + name + ".bias"])
        sp = lambda t: t.view(n, S, heads, d).transpose(1, 2)
        q, k, v = sp(lin(x, "attention.self.query")), sp(lin(x, "attention.self.key")), sp(lin(x, "attention.self.value"))
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=bias)         # softmax(QK^T/sqrt(d) + bias) V
        ctx = ctx.transpose(1, 2).reshape(n, S, H)
        x = F.layer_norm(lin(ctx, "attention.output.dense") + x, (H,), P[p + "attention.output.LayerNorm.weight"],
                         P[p + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(lin(x, "intermediate.dense"))
        x = F.layer_norm(lin(h, "output.dense") + x, (H,), P[p + "output.LayerNorm.weight"], P[p + "output.LayerNorm.bias"], eps)
    if "pooler.dense.weight" in P:         # models.py:80: last_hidden_state + 0 * pooler_output.sum()
        pooled = torch.tanh(F.linear(x[:, 0], P["pooler.dense.weight"], P["pooler.dense.bias"]))
        x = x + 0 * pooled.sum()
    return x, x[:, 0, :]


def retriever_step(Pq, Pc, q_ids, q_mask, c_ids, c_mask, heads, teacher_logits=None, teacher=None, temperature=1.0):
    """co_training_marco_train.py:198-217 + backward.  teacher = (Pt, qa_w, qa_b, t_ids3, t_mask3): cross-encoder forward
    under no_grad (models.py:647-659).  -> dict(q, ctx, sim, loss, z)"""
    _, q = bert_forward(Pq, q_ids, q_mask, heads)
    _, c = bert_forward(Pc, c_ids, c_mask, heads)
    sim = torch.einsum("bh,bdh->bd", q, c.reshape(q.size(0), c.size(0) // q.size(0), -1))
    p_s = F.softmax(sim, dim=1)
    with torch.no_grad():
        if teacher is not None:
            Pt, w, b, t_ids3, t_mask3 = teacher
            N, M, Lq = t_ids3.shape
            _, tcls = bert_forward(Pt, t_ids3.view(N * M, Lq), t_mask3.view(N * M, Lq), heads)
            teacher_logits = F.linear(tcls, w, b).view(N, M)
        p_t = F.softmax(teacher_logits / temperature, dim=1)
    loss = torch.nn.KLDivLoss(reduction="batchmean")((p_s + 1e-7).log(), p_t)
    loss.backward()
    return dict(q=q.detach(), ctx=c.detach(), sim=sim.detach(), loss=float(loss.item()), z=teacher_logits)


def time_step(B, N, q_len=32, p_len=128, ce_len=160, with_teacher=True, warmup=3, steps=10, threads=None, seed=1):
    """Times the retriever step on synthetic all-max-length inputs (BERT-base, random init), fp32.
    -> (seconds per step, pairs per step, threads)."""
    from .weights import BertCfg, make_batch, make_bert_params
    if threads:
        torch.set_num_threads(int(threads))
    cfg = BertCfg()
    P = B * (1 + N)
    Pq, Pc = (to_torch_params(make_bert_params(cfg, s, perturb=False)) for s in (seed, seed + 1))
    tt = lambda a: torch.from_numpy(a)
    q_ids, q_mask, _ = make_batch(seed, B, q_len, cfg.vocab, 9, 3, 4, full=True)
    c_ids, c_mask, _ = make_batch(seed + 1, P, p_len, cfg.vocab, 80, 25, 16, full=True)
    teacher, z = None, None
    if with_teacher:
        Pt = to_torch_params(make_bert_params(cfg, seed + 2, perturb=False), requires_grad=False)
        t_ids, t_mask, _ = make_batch(seed + 2, P, ce_len, cfg.vocab, 90, 25, 20, full=True)
        teacher = (Pt, torch.full((1, cfg.hidden), 0.01), torch.zeros(1), tt(t_ids).view(B, 1 + N, -1), tt(t_mask).view(B, 1 + N, -1))
    else:
        z = torch.from_numpy(np.linspace(-2, 2, P).reshape(B, 1 + N).astype(np.float32))

    def one():
        for Pp in (Pq, Pc):
            for v in Pp.values():
                v.grad = None
        retriever_step(Pq, Pc, tt(q_ids), tt(q_mask), tt(c_ids), tt(c_mask), cfg.heads, teacher_logits=z, teacher=teacher)
    for _ in range(warmup):
        one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    return (time.perf_counter() - t0) / steps, P, torch.get_num_threads()


def main():
    """`python -m oracle.torch_cpu` -- the CPU-baseline measurement of bench.py, run in its own process (its own OpenMP pool,
    a hard timeout on the caller's side): prints one JSON line."""
    import json
    import os
    import sys
    thr = usable_cpus()
    torch.set_num_threads(thr)
    t0 = time.time()
    s1, p1, _ = time_step(4, 1, with_teacher=False, warmup=3, steps=10)
    w2, n2 = int(os.environ.get("SIMX_CPU_BASE_WARMUP", 3)), int(os.environ.get("SIMX_CPU_BASE_STEPS", 5))
    s2, p2, _ = time_step(8, 15, with_teacher=True, warmup=w2, steps=n2)
    print(json.dumps({"threads": thr, "cfg0_s_per_step": s1, "cfg0_pairs": p1, "cfg1r_s_per_step": s2, "cfg1r_pairs": p2,
                      "cfg1r_warmup": w2, "cfg1r_steps": n2, "wall_s": time.time() - t0}))
    sys.stdout.flush()


if __name__ == "__main__":
    main()
