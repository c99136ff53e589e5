"""Oracle-checked wgrad kernels at the token count bench.py runs them at (BASELINE configs[1]: 2048 passages x 128 tokens =
262144 tokens per launch; SimANS/train_MS_Pas_AR2.sh:8-13 x the B = 128 of configs[1]) -- the per-kernel tests of
tests/test_kernels_gpu.py / tests/test_planes_gpu.py stop at 33000 tokens, where the split plans (tn_plan / xp_tn_plan: token
ranges per workgroup, slab count, slab reduction) are different ones.

Checker: a float64 NumPy product of the SAME 16-bit operands on the host (155-620 GFLOP per shape), computed in token chunks.
Both split rules are covered: the default (one round of the chip) and SIMX_TN_ROUNDS=2 (two rounds; rounds 2-4's rule).  The
library reads the switch per call, so one process can run both.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
F32, BF16, F16 = 0, 1, 2
TOKENS = 262144
SHAPES = [(768, 768), (3072, 768)]          # (M, N) of dW = dY^T . X: attention-output / FFN-out weight gradients of BERT-base
CHUNK = 16384


def L():
    from simxns_amd import _lib
    return _lib


def _ref_tn(a_chunks, b_chunks, M, N):
    """float64 A^T . B and the column sums of A, accumulated over host chunks (a_chunks / b_chunks yield float64 arrays)."""
    ref = np.zeros((M, N), np.float64)
    col = np.zeros((M,), np.float64)
    for a, b in zip(a_chunks, b_chunks):
        ref += a.T @ b
        col += a.sum(0)
    return ref, col


class _rounds:
    def __init__(self, v):
        self.v = v

    def __enter__(self):
        self.old = os.environ.pop("SIMX_TN_ROUNDS", None)
        if self.v is not None:
            os.environ["SIMX_TN_ROUNDS"] = self.v

    def __exit__(self, *a):
        os.environ.pop("SIMX_TN_ROUNDS", None)
        if self.old is not None:
            os.environ["SIMX_TN_ROUNDS"] = self.old


@pytest.mark.parametrize("M,N", SHAPES)
def test_gemm_tn_fp16_at_the_benchmarked_token_count(dev, M, N):
    """simx_gemm_tn_bias, fp16 operands, K = 262144 tokens: product and fused bias gradient against float64, both split rules,
    accumulate on (the engine accumulates into the flat gradient buffer)."""
    lib = L()
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + M)
    K = TOKENS
    A = (torch.randn(K, M, device=dev, generator=g) * 0.5).to(torch.float16)           # dY
    B = (torch.randn(K, N, device=dev, generator=g) * 0.5).to(torch.float16)           # X
    C0 = torch.randn(M, N, device=dev, generator=g)
    db0 = torch.randn(M, device=dev, generator=g)
    ref, col = _ref_tn((A[i:i + CHUNK].cpu().numpy().astype(np.float64) for i in range(0, K, CHUNK)),
                       (B[i:i + CHUNK].cpu().numpy().astype(np.float64) for i in range(0, K, CHUNK)), M, N)
    ref += C0.cpu().numpy().astype(np.float64)
    col += db0.cpu().numpy().astype(np.float64)
    plans = set()
    for rounds in (None, "2"):
        with _rounds(rounds):
            wsb = int(lib.load().simx_gemm_tn_workspace_bytes(M, N, K))
            plans.add(wsb)
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            dC, db = C0.clone(), db0.clone()
            lib.call("simx_gemm_tn_bias", lib.stream_ptr(), F16, M, N, K, lib.ptr(A), M, lib.ptr(B), N, lib.ptr(dC), N, 1, lib.ptr(ws), wsb,
                     lib.ptr(db))
            torch.cuda.synchronize()
        got = dC.cpu().numpy().astype(np.float64)
        err = np.abs(got - ref)
        lim = 2e-5 * math.sqrt(K) + 2e-5 * np.abs(ref)               # f32 accumulation of K products of O(0.25); a lost 4681-token range is ~17
        assert (err <= lim).all(), "gemm_tn fp16 %dx%d rounds=%s: worst %.3e (limit %.3e)" % (M, N, rounds, err.max(), lim.flat[err.argmax()])
        gb = db.cpu().numpy().astype(np.float64)
        assert (np.abs(gb - col) <= 1e-5 * np.abs(col) + 2e-5 * math.sqrt(K)).all(), "fused bias gradient, rounds=%s: %.3e" % (rounds, np.abs(gb - col).max())
    assert len(plans) == 2, "the two split rules must be different plans at this size (workspace bytes %s)" % sorted(plans)


@pytest.mark.parametrize("M,N", SHAPES)
def test_gemm_tn_planes_at_the_benchmarked_token_count(dev, M, N):
    """simx_gemm_tn_planes (the fp32 engine's wgrad: bf16 plane pairs, hi.lo + lo.hi + hi.hi in f32) at K = 262144 tokens against
    the float64 product of the pairs' values, both split rules, fused bias gradient."""
    lib = L()
    g = torch.Generator(device=dev)
    g.manual_seed(4321 + M)
    K = TOKENS
    A = torch.randn(K, M, device=dev, generator=g) * 1e-3 * torch.exp(torch.randn(K, 1, device=dev, generator=g) * 2.0)   # dY: rows of very different size
    B = torch.randn(K, N, device=dev, generator=g) * 0.7
    Ap = torch.empty(2, K, M, device=dev, dtype=torch.int16)
    Bp = torch.empty(2, K, N, device=dev, dtype=torch.int16)
    lib.call("simx_planes_from", lib.stream_ptr(), F32, BF16, K, M, lib.ptr(A), M, 0, lib.ptr(Ap), M, K * M)
    lib.call("simx_planes_from", lib.stream_ptr(), F32, BF16, K, N, lib.ptr(B), N, 0, lib.ptr(Bp), N, K * N)
    torch.cuda.synchronize()
    del A, B

    def val(p, i):
        v = p[:, i:i + CHUNK].view(torch.bfloat16).to(torch.float64)
        return (v[0] + v[1]).cpu().numpy()
    ref = np.zeros((M, N), np.float64)
    col = np.zeros((M,), np.float64)
    acol = np.zeros((M,), np.float64)
    sa = np.zeros((M,), np.float64)
    sb = np.zeros((N,), np.float64)
    for i in range(0, K, CHUNK):
        a, b = val(Ap, i), val(Bp, i)
        ref += a.T @ b
        col += a.sum(0)
        acol += np.abs(a).sum(0)
        sa += (a ** 2).sum(0)
        sb += (b ** 2).sum(0)
    C0 = torch.randn(M, N, device=dev, generator=g) * 0.01
    db0 = torch.randn(M, device=dev, generator=g) * 0.01
    ref += C0.cpu().numpy().astype(np.float64)
    col += db0.cpu().numpy().astype(np.float64)
    scale = np.sqrt(sa)[:, None] * np.sqrt(sb)[None, :]
    plans = set()
    for rounds in (None, "2"):
        with _rounds(rounds):
            wsb = int(lib.load().simx_gemm_tn_planes_workspace_bytes(M, N, K))
            plans.add(wsb)
            ws = torch.empty(max(wsb, 16) // 4, device=dev)
            dC, db = C0.clone(), db0.clone()
            lib.call("simx_gemm_tn_planes", lib.stream_ptr(), M, N, K, lib.ptr(Ap), M, K * M, lib.ptr(Bp), N, K * N, lib.ptr(dC), N, 1,
                     lib.ptr(ws), wsb, lib.ptr(db))
            torch.cuda.synchronize()
        got = dC.cpu().numpy().astype(np.float64)
        # the bound of tests/test_planes_gpu.py::test_gemm_tn_planes (2^-17 per element of the dropped lo.lo term + f32 accumulation)
        assert np.all(np.abs(got - ref) <= 3e-5 * scale + 1e-6 * np.abs(ref)), \
            "gemm_tn_planes %dx%d rounds=%s: %.3e of the row scale" % (M, N, rounds, (np.abs(got - ref) / (scale + 1e-30)).max())
        gb = db.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(gb - col) <= 1e-5 * acol + 1e-7), "bias gradient, rounds=%s: %.3e" % (rounds, np.abs(gb - col).max())
    assert len(plans) == 2, "the two split rules must be different plans at this size (workspace bytes %s)" % sorted(plans)
