"""Optimiser / clipping / schedule restatement.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference: ``transformers.AdamW`` (imported at
SimANS/co_training/co_training_marco_train.py:21-25, built at :57-69), third-party
and removed from transformers >= 5 -- its published 4.x update rule
(``correct_bias=True``) is restated here;
``clip_grad_norm_`` (:251) and ``get_linear_schedule_with_warmup`` (:126-134).
Parity for AdamW is therefore "unpinned by the reference"; it is cross-checked
against ``torch.optim.AdamW`` (eps placement differs by O(eps/sqrt(v))).
"""
import numpy as np


def clip_coef(grads, max_norm=2.0):
    tot = np.sqrt(sum(float((g.astype(np.float64) ** 2).sum()) for g in grads))
    return tot, min(1.0, max_norm / (tot + 1e-6))


def adamw_hf_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, wd=0.0):
    """In the transformers-4 order: moments, bias-corrected step size, then decoupled decay."""
    m[...] = beta1 * m + (1.0 - beta1) * g
    v[...] = beta2 * v + (1.0 - beta2) * g * g
    step_size = lr * np.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    p[...] = p - step_size * m / (np.sqrt(v) + eps)
    if wd > 0.0:
        p[...] = p - lr * wd * p
    return p, m, v


def linear_schedule(step, warmup, total):
    if step < warmup:
        return float(step) / float(max(1, warmup))
    return max(0.0, float(total - step) / float(max(1, total - warmup)))
