"""The fp32 engine's plane-operand GEMMs (csrc/gemm_xp.hip) through the C ABI against float64 NumPy on the same inputs.

A plane pair is hi = rnd16(x), lo = rnd16(x - hi); the kernels compute hi.lo + lo.hi + hi.hi in f32 accumulators, i.e. the
product of the f32 operands up to the split's 2^-22 (fp16 planes, forward) / 2^-17 (bf16 planes, backward) per element and f32
accumulation -- the tolerances below are those figures times the row scale, NOT 16-bit tolerances.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import bert as obert

pytestmark = pytest.mark.gpu
F32, BF16, F16 = 0, 1, 2


def L():
    from simxns_amd import _lib
    return _lib


def rnd(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32)


def planes_of(x, fmt, dev):
    """plane pair of an f32 device matrix via simx_planes_from: int16 tensor [2, rows, cols]"""
    lib = L()
    rows, cols = x.shape
    p = torch.empty(2, rows, cols, device=dev, dtype=torch.int16)
    lib.call("simx_planes_from", lib.stream_ptr(), F32, fmt, rows, cols, lib.ptr(x), cols, 0, lib.ptr(p), cols, rows * cols)
    return p


def planes_value(p, fmt):
    """float64 value hi + lo of a plane pair"""
    t = torch.float16 if fmt == F16 else torch.bfloat16
    v = p.view(t).to(torch.float64)
    return (v[0] + v[1]).cpu().numpy()


@pytest.mark.parametrize("fmt", [F16, BF16])
def test_planes_roundtrip(dev, fmt):
    lib = L()
    x = torch.from_numpy(rnd((300, 136), 1) * np.exp(rnd((300, 136), 2) * 2)).to(dev)
    p = planes_of(x, fmt, dev)
    v = planes_value(p, fmt)
    xr = x.cpu().numpy().astype(np.float64)
    rel = 2.0 ** -21 if fmt == F16 else 2.0 ** -16
    assert np.all(np.abs(v - xr) <= rel * np.abs(xr) + (2.0 ** -24 if fmt == F16 else 0.0))
    back = torch.empty_like(x)
    lib.call("simx_planes_join", lib.stream_ptr(), fmt, 300, 136, lib.ptr(p), 136, 300 * 136, lib.ptr(back), 136)
    assert np.array_equal(back.cpu().numpy().astype(np.float64), v.astype(np.float32).astype(np.float64))
    # pair -> pair of the other format
    other = BF16 if fmt == F16 else F16
    q = torch.empty_like(p)
    lib.call("simx_planes_from", lib.stream_ptr(), fmt, other, 300, 136, lib.ptr(p), 136, 300 * 136, lib.ptr(q), 136, 300 * 136)
    w = planes_value(q, other)
    rel2 = 2.0 ** -21 if other == F16 else 2.0 ** -16
    assert np.all(np.abs(w - v) <= rel2 * np.abs(v) + 2.0 ** -24)


NT_CASES = [  # (variant, fmt)
    ("bias", F16), ("bias+res", F16), ("bias+res+drop", F16), ("gelu", F16), ("gelu_infer", F16),
    ("plain", BF16), ("res", BF16), ("dgelu", BF16)]


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (512, 768, 768), (1024, 256, 192), (2048, 2304, 768), (768, 768, 3072),
                                   (66048, 768, 256)])     # the last: 774 tiles on 256 workgroups -- the persistent walk, ragged last round
@pytest.mark.parametrize("variant,fmt", NT_CASES)
def test_gemm_nt_planes(dev, M, N, K, variant, fmt):
    lib = L()
    sa, sb = (0.5, 0.05) if fmt == F16 else (1e-3, 0.05)        # backward operands are small: gradients
    A = torch.from_numpy(rnd((M, K), 1, sa)).to(dev)
    B = torch.from_numpy(rnd((N, K), 2, sb)).to(dev)
    Ap, Bp = planes_of(A, fmt, dev), planes_of(B, fmt, dev)
    bias = torch.from_numpy(rnd((N,), 3, 0.5)).to(dev) if variant.startswith(("bias", "gelu")) else None
    need_in = "res" in variant or variant == "dgelu"
    inn = torch.from_numpy(rnd((M, N), 4)).to(dev) if need_in else None
    Cc = torch.full((M, N), float("nan"), device=dev) if variant not in ("dgelu", "gelu_infer") else None
    Cp = torch.zeros(2, M, N, device=dev, dtype=torch.int16) if variant in ("gelu", "gelu_infer", "dgelu") else None
    epi = {"gelu": 1, "gelu_infer": 3, "dgelu": 2}.get(variant, 0)
    drop = lib.Dropout(0.1, 1234, 7) if variant.endswith("drop") else None
    lib.call("simx_gemm_nt_planes", lib.stream_ptr(), fmt, epi, M, N, K, lib.ptr(Ap), K, M * K, lib.ptr(Bp), K, N * K, lib.ptr(Cc), N,
             lib.ptr(bias), lib.ptr(inn), N, lib.ptr(Cp), N, M * N, C.byref(drop) if drop else None)
    torch.cuda.synchronize()
    Av, Bv = planes_value(Ap, fmt), planes_value(Bp, fmt)
    acc = Av @ Bv.T
    scale = np.sqrt((Av ** 2).sum(1))[:, None] * np.sqrt((Bv ** 2).sum(1))[None, :]      # |a||b| bounds every partial sum
    if bias is not None:
        acc = acc + bias.cpu().numpy().astype(np.float64)
    tol = 2e-6 * scale + 3e-7 * np.abs(acc) + 1e-30           # f32 accumulation over K + the dropped lo.lo term (2^-22 x 2^-22)
    if variant in ("bias", "plain"):
        got = Cc.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - acc) <= tol), np.abs(got - acc).max()
    elif variant in ("bias+res", "res"):
        got = Cc.cpu().numpy().astype(np.float64)
        ref = acc + inn.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - ref) <= tol + 2e-7 * np.abs(ref)), np.abs(got - ref).max()
    elif variant == "bias+res+drop":
        got = Cc.cpu().numpy().astype(np.float64)
        mask = obert.drop_multipliers(0.1, 1234, 7, np.arange(M), np.arange(N))
        ref = acc * mask + inn.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(got - ref) <= tol * 1.2 + 2e-7 * np.abs(ref)), np.abs(got - ref).max()
    elif variant in ("gelu", "gelu_infer"):
        h = planes_value(Cp, F16)
        assert np.all(np.abs(h - obert.gelu(acc)) <= tol * 1.2 + 6e-7 * np.abs(acc) + 3e-7), np.abs(h - obert.gelu(acc)).max()
        if variant == "gelu":
            d = Cc.cpu().numpy().astype(np.float64)
            assert np.all(np.abs(d - obert.gelu_grad(acc)) <= tol * 1.5 + 1e-6), np.abs(d - obert.gelu_grad(acc)).max()
    else:
        du = planes_value(Cp, BF16)
        ref = acc * inn.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(du - ref) <= (tol + 3e-7 * np.abs(acc)) * np.abs(inn.cpu().numpy()) + 2.0 ** -15 * np.abs(ref)), np.abs(du - ref).max()


@pytest.mark.parametrize("M,N,K", [(256, 256, 2048), (768, 768, 4096), (2304, 768, 4100), (768, 3072, 8192 + 33), (520, 264, 1000),
                                   (768, 768, 40)])
@pytest.mark.parametrize("with_bias", [False, True])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_gemm_tn_planes(dev, M, N, K, with_bias, accumulate):
    lib = L()
    A = torch.from_numpy(rnd((K, M), 1, 1e-3) * np.exp(rnd((K, 1), 5, 2.0))).to(dev)          # dY: rows of very different size
    B = torch.from_numpy(rnd((K, N), 2, 0.7)).to(dev)
    Ap, Bp = planes_of(A, BF16, dev), planes_of(B, BF16, dev)
    C0 = rnd((M, N), 3, 0.01)
    Cc = torch.from_numpy(C0.copy()).to(dev)
    db0 = rnd((M,), 4, 0.01)
    db = torch.from_numpy(db0.copy()).to(dev) if with_bias else None
    wsb = lib.load().simx_gemm_tn_planes_workspace_bytes(M, N, K)
    ws = torch.empty(max(wsb, 16) // 4, device=dev)
    lib.call("simx_gemm_tn_planes", lib.stream_ptr(), M, N, K, lib.ptr(Ap), M, K * M, lib.ptr(Bp), N, K * N, lib.ptr(Cc), N, accumulate,
             lib.ptr(ws), wsb, lib.ptr(db))
    torch.cuda.synchronize()
    Av, Bv = planes_value(Ap, BF16), planes_value(Bp, BF16)
    ref = Av.T @ Bv + (C0.astype(np.float64) if accumulate else 0.0)
    scale = np.sqrt((Av ** 2).sum(0))[:, None] * np.sqrt((Bv ** 2).sum(0))[None, :]
    got = Cc.cpu().numpy().astype(np.float64)
    assert np.all(np.abs(got - ref) <= 3e-5 * scale + 1e-6 * np.abs(ref)), (np.abs(got - ref) / (scale + 1e-30)).max()
    if with_bias:
        rb = Av.sum(0) + db0.astype(np.float64)
        gb = db.cpu().numpy().astype(np.float64)
        assert np.all(np.abs(gb - rb) <= 1e-5 * np.abs(Av).sum(0) + 1e-7), np.abs(gb - rb).max()


@pytest.mark.parametrize("M,N,K,valid", [(512, 768, 768, 512), (2048, 3072, 768, 1999), (66048, 768, 256, 65900)])
def test_gemm_nt_planes_dgelu_colsum(dev, M, N, K, valid):
    """SIMX_EPI_DGELU with the fused bias gradient: colsum[N] += column sums of (acc * in) over the rows below `valid` (rows past
    the real tokens hold garbage in the engine: here NaN, which must not leak)."""
    lib = L()
    A = torch.from_numpy(rnd((M, K), 1, 1e-3)).to(dev)
    A[valid:] = float("nan")
    B = torch.from_numpy(rnd((N, K), 2, 0.05)).to(dev)
    Ap, Bp = planes_of(A, BF16, dev), planes_of(B, BF16, dev)
    inn = torch.from_numpy(rnd((M, N), 4)).to(dev)
    Cp = torch.zeros(2, M, N, device=dev, dtype=torch.int16)
    cs0 = rnd((N,), 6, 1e-3)
    cs = torch.from_numpy(cs0.copy()).to(dev)
    lib.call("simx_gemm_nt_planes_cs", lib.stream_ptr(), BF16, 2, M, N, K, lib.ptr(Ap), K, M * K, lib.ptr(Bp), K, N * K, None, N,
             None, lib.ptr(inn), N, lib.ptr(Cp), N, M * N, None, lib.ptr(cs), valid)
    torch.cuda.synchronize()
    Av, Bv = planes_value(Ap, BF16)[:valid], planes_value(Bp, BF16)
    ref = (Av @ Bv.T) * inn.cpu().numpy().astype(np.float64)[:valid]
    du = planes_value(Cp, BF16)[:valid]
    scale = np.sqrt((Av ** 2).sum(1))[:, None] * np.sqrt((Bv ** 2).sum(1))[None, :] * np.abs(inn.cpu().numpy().astype(np.float64)[:valid])
    assert np.all(np.abs(du - ref) <= 2.0 ** -15 * np.abs(ref) + 2e-6 * scale + 1e-12)
    got = cs.cpu().numpy().astype(np.float64)
    want = ref.sum(0) + cs0
    assert np.isfinite(got).all()
    assert np.all(np.abs(got - want) <= 2e-5 * np.abs(ref).sum(0) + 1e-7), np.abs(got - want).max()


@pytest.mark.parametrize("rows,cols", [(768, 3072), (2304, 768), (64, 64), (96, 40), (33, 7)])
def test_split_weight_forms(dev, rows, cols):
    """simx_split_weight (what simx_bert_cast_weights runs per dense weight of an fp32 tower): the fp16 plane pair of W, the bf16
    plane pair of W^T and W^T in f32 are BIT-identical to the elementwise definition hi = rnd16(w), lo = rnd16(w - hi) -- on the
    64 x 64-tile kernel (rows, cols multiples of 64) and on the 32 x 32 kernel for every other shape; and the rate of the former."""
    lib = L()
    w = torch.from_numpy(rnd((rows, cols), 7, 0.05) * np.exp(rnd((rows, cols), 8) * 1.5)).to(dev)
    ph = torch.full((2, rows, cols), -1, device=dev, dtype=torch.int16)
    pt = torch.full((2, cols, rows), -1, device=dev, dtype=torch.int16)
    wT = torch.full((cols, rows), float("nan"), device=dev)
    run = lambda: lib.call("simx_split_weight", lib.stream_ptr(), lib.ptr(w), rows, cols, lib.ptr(ph), lib.ptr(pt), lib.ptr(wT))
    run()
    torch.cuda.synchronize()
    assert torch.equal(wT, w.t().contiguous())
    for planes, src, t in ((ph, w, torch.float16), (pt, w.t().contiguous(), torch.bfloat16)):
        hi = src.to(t)
        lo = (src - hi.float()).to(t)
        assert torch.equal(planes[0].view(t), hi) and torch.equal(planes[1].view(t), lo)
    # any output may be absent
    ph.fill_(-1)
    lib.call("simx_split_weight", lib.stream_ptr(), lib.ptr(w), rows, cols, lib.ptr(ph), None, None)
    torch.cuda.synchronize()
    assert torch.equal(ph[0].view(torch.float16), w.to(torch.float16))
    if rows * cols >= 768 * 3072:
        from tests.test_attention_x3_gpu import _ms
        ms = _ms(run, 20)
        gbs = rows * cols * 16 / ms / 1e6
        print("split_weight %d x %d: %.3f ms = %.0f GB/s" % (rows, cols, ms, gbs))
        assert gbs > 1000, gbs                       # (the 32 x 32 scalar-store kernel ran at 220 GB/s)
