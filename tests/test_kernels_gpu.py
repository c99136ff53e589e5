"""Per-kernel parity: HIP kernels (through the C ABI) vs the NumPy oracle on the same seeded inputs.

16-bit kernels (bf16, fp16) are compared against the oracle evaluated on the SAME rounded inputs in float64, so the
tolerance only has to cover f32 accumulation order and the final rounding of the output (2^-9 relative for bf16,
2^-12 for fp16); f32 kernels are compared at f32 round-off.  The `bf16` parameter of the tests is the element format:
False = f32, True = bf16, "f16" = IEEE half.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import bert as obert
from oracle import optim as ooptim

pytestmark = pytest.mark.gpu


def L():
    from simxns_amd import _lib
    return _lib


def rnd(shape, seed, scale=1.0):
    rs = np.random.RandomState(seed)
    return (rs.randn(*shape) * scale).astype(np.float32)


def code(fmt):
    """C-ABI dtype code of an element format (SIMX_F32 / SIMX_BF16 / SIMX_F16)."""
    return 2 if fmt == "f16" else int(bool(fmt))


def tdt(fmt):
    return torch.float16 if fmt == "f16" else torch.bfloat16 if fmt else torch.float32


def to_dev(a, dev, bf16=False):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return t.to(tdt(bf16)) if bf16 else t


def back(t):
    return t.detach().to(torch.float32).cpu().numpy().astype(np.float64)


def rounded(a, bf16):
    """what the kernel actually sees, as float64"""
    if not bf16:
        return a.astype(np.float64)
    return torch.from_numpy(a).to(tdt(bf16)).to(torch.float32).numpy().astype(np.float64)


def assert_close(got, ref, rtol, atol, what=""):
    err = np.abs(got - ref)
    lim = atol + rtol * np.abs(ref)
    bad = err > lim
    assert not bad.any(), "%s: %d/%d out of tolerance, worst err %.3e (ref %.3e) at %s" % (
        what, bad.sum(), bad.size, err.max(), np.abs(ref).max(), np.unravel_index(err.argmax(), err.shape))


TOL = {False: dict(rtol=2e-5, atol=2e-5), True: dict(rtol=1.2e-2, atol=2e-2), "f16": dict(rtol=1.5e-3, atol=2.5e-3)}
FMTS = [False, True, "f16"]
H16 = [True, "f16"]


def pick(fmt, f32, bf16, f16):
    return f16 if fmt == "f16" else bf16 if fmt else f32


# ------------------------------------------------------------------------------------------ GEMM NT
@pytest.mark.parametrize("bf16", FMTS)
@pytest.mark.parametrize("M,N,K", [(200, 192, 64), (300, 768, 768), (129, 64, 128), (77, 100, 40), (1000, 3072, 768),
                                   (8300, 768, 768), (4100, 1536, 128), (8300, 768, 192), (6200, 1024, 3072),
                                   (8300, 1536, 128), (16500, 768, 64), (16500, 700, 192), (16400, 768, 768)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_nt(dev, bf16, M, N, K, epi):
    lib = L()
    A, B = rnd((M, K), 1, 0.5), rnd((N, K), 2, 0.5)
    bias = rnd((N,), 3, 0.5) if epi != 2 else None
    res = rnd((M, N), 4) if epi == 0 else None
    aux = rnd((M, N), 5) if epi == 2 else None
    dA, dB = to_dev(A, dev, bf16), to_dev(B, dev, bf16)
    dbias = to_dev(bias, dev) if bias is not None else None
    dres = to_dev(res, dev, bf16) if res is not None else None
    daux = to_dev(aux, dev, bf16) if aux is not None else None
    dC = torch.empty(M, N, device=dev, dtype=tdt(bf16))
    dC2 = torch.empty_like(dC) if epi == 1 else None
    lib.call("simx_gemm_nt", lib.stream_ptr(), code(bf16), M, N, K, lib.ptr(dA), K, lib.ptr(dB), K, lib.ptr(dC), N,
             lib.ptr(dbias), lib.ptr(dres), N, epi, lib.ptr(daux), N, lib.ptr(dC2), N)
    torch.cuda.synchronize()
    acc = rounded(A, bf16) @ rounded(B, bf16).T
    if bias is not None:
        acc = acc + bias.astype(np.float64)
    t = dict(TOL[bf16])
    t["atol"] *= max(1.0, math.sqrt(K) * 0.25)
    if epi == 0:
        assert_close(back(dC), acc + rounded(res, bf16), what="gemm_nt none", **t)
    elif epi == 1:
        # C = gelu'(u) -- the factor backward multiplies by, stored where the pre-activation used to be -- and C2 = gelu(u)
        assert_close(back(dC), obert.gelu_grad(acc), what="gemm_nt gelu derivative output", **t)
        assert_close(back(dC2), obert.gelu(acc), what="gemm_nt gelu", **t)
    else:
        assert_close(back(dC), acc * rounded(aux, bf16), what="gemm_nt dgelu (x stored derivative)", **t)


@pytest.mark.parametrize("M,N,K", [(32768, 768, 128), (16384, 1536, 192), (24576, 768, 768), (16384, 768, 3072), (49152, 256, 64 * 5)])
@pytest.mark.parametrize("variant", ["bias", "bias+res", "res", "plain", "gelu", "dgelu", "bias+res+drop"])
@pytest.mark.parametrize("fmt", H16)
def test_gemm_nt_persistent(dev, M, N, K, variant, fmt):
    """Full-tile bf16 problems with >= 192 256x256 tiles run the persistent kernel (several tiles per workgroup when
    there are more tiles than CUs, odd and even stage counts, every epilogue variant)."""
    lib = L()
    A, B = rnd((M, K), 1, 0.5), rnd((N, K), 2, 0.5)
    epi = 1 if variant == "gelu" else 2 if variant == "dgelu" else 0
    bias = rnd((N,), 3, 0.5) if "bias" in variant or epi == 1 else None
    res = rnd((M, N), 4) if "res" in variant else None
    aux = rnd((M, N), 5) if epi == 2 else None
    dA, dB = to_dev(A, dev, fmt), to_dev(B, dev, fmt)
    dbias = to_dev(bias, dev) if bias is not None else None
    dres = to_dev(res, dev, fmt) if res is not None else None
    daux = to_dev(aux, dev, fmt) if aux is not None else None
    dC = torch.full((M, N), float("nan"), device=dev, dtype=tdt(fmt))
    dC2 = torch.full((M, N), float("nan"), device=dev, dtype=tdt(fmt)) if epi == 1 else None
    acc = rounded(A, fmt) @ rounded(B, fmt).T
    if bias is not None:
        acc = acc + bias.astype(np.float64)
    t = dict(TOL[fmt])
    t["atol"] *= max(1.0, math.sqrt(K) * 0.25)
    if "drop" in variant:
        from simxns_amd._lib import Dropout
        d = Dropout(0.1, 77, 9)
        lib.call("simx_gemm_nt_ex", lib.stream_ptr(), code(fmt), M, N, K, lib.ptr(dA), K, lib.ptr(dB), K, lib.ptr(dC), N,
                 lib.ptr(dbias), lib.ptr(dres), N, epi, lib.ptr(daux), N, lib.ptr(dC2), N, C.byref(d))
        torch.cuda.synchronize()
        mult = obert.drop_multipliers(0.1, 77, 9, np.arange(M), np.arange(N))
        assert_close(back(dC), acc * mult + rounded(res, fmt), what="gemm_nt persistent dropout", **t)
        return
    lib.call("simx_gemm_nt", lib.stream_ptr(), code(fmt), M, N, K, lib.ptr(dA), K, lib.ptr(dB), K, lib.ptr(dC), N,
             lib.ptr(dbias), lib.ptr(dres), N, epi, lib.ptr(daux), N, lib.ptr(dC2), N)
    torch.cuda.synchronize()
    if epi == 0:
        assert_close(back(dC), acc + (rounded(res, fmt) if res is not None else 0.0), what="gemm_nt persistent " + variant, **t)
    elif epi == 1:
        assert_close(back(dC), obert.gelu_grad(acc), what="gemm_nt persistent gelu derivative output", **t)
        assert_close(back(dC2), obert.gelu(acc), what="gemm_nt persistent gelu", **t)
    else:
        assert_close(back(dC), acc * rounded(aux, fmt), what="gemm_nt persistent dgelu (x stored derivative)", **t)


# ------------------------------------------------------------------------------------------ f32 GEMMs on the 16-bit matrix cores
@pytest.mark.parametrize("code_", [3, 4])
@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (1000, 3072, 768), (8300, 768, 3072), (2048, 768, 772), (513, 260, 64), (4100, 2304, 768)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm_nt_f32_split(dev, code_, M, N, K, epi):
    """SIMX_F32_SPLIT_H / _B (csrc/gemm_x3.hip): f32 operands split into 16-bit hi + lo on the fly, three MFMAs per product.
    fp16 halves: the exact kernel's tolerance holds; bf16 halves: 2^-17 per operand."""
    lib = L()
    A, B = rnd((M, K), 1, 0.5), rnd((N, K), 2, 0.5)
    B[:, :7] *= 0.03                                   # weight-like magnitudes (fp16 lo parts go subnormal: absolute floor)
    bias = rnd((N,), 3, 0.5) if epi != 2 else None
    res = rnd((M, N), 4) if epi == 0 else None
    aux = rnd((M, N), 5) if epi == 2 else None
    dA, dB = to_dev(A, dev), to_dev(B, dev)
    dbias = to_dev(bias, dev) if bias is not None else None
    dres = to_dev(res, dev) if res is not None else None
    daux = to_dev(aux, dev) if aux is not None else None
    dC = torch.full((M, N), float("nan"), device=dev)
    dC2 = torch.full((M, N), float("nan"), device=dev) if epi == 1 else None
    lib.call("simx_gemm_nt", lib.stream_ptr(), code_, M, N, K, lib.ptr(dA), K, lib.ptr(dB), K, lib.ptr(dC), N,
             lib.ptr(dbias), lib.ptr(dres), N, epi, lib.ptr(daux), N, lib.ptr(dC2), N)
    torch.cuda.synchronize()
    acc = A.astype(np.float64) @ B.astype(np.float64).T
    if bias is not None:
        acc = acc + bias.astype(np.float64)
    t = dict(rtol=2e-5, atol=2e-5 * max(1.0, math.sqrt(K) * 0.25)) if code_ == 3 else dict(rtol=1e-4, atol=4e-5 * math.sqrt(K))
    if epi == 0:
        assert_close(back(dC), acc + res.astype(np.float64), what="split gemm_nt none", **t)
    elif epi == 1:
        assert_close(back(dC), obert.gelu_grad(acc), what="split gemm_nt gelu derivative output", **t)
        assert_close(back(dC2), obert.gelu(acc), what="split gemm_nt gelu", **t)
    else:
        assert_close(back(dC), acc * aux.astype(np.float64), what="split gemm_nt dgelu", **t)


@pytest.mark.parametrize("code_", [3, 4])
@pytest.mark.parametrize("M,N,K,accumulate", [(768, 768, 5000, 1), (3072, 768, 2100, 0), (768, 3072, 33000, 1), (128, 256, 1000, 0), (772, 260, 4100, 1)])
def test_gemm_tn_f32_split(dev, code_, M, N, K, accumulate):
    lib = L()
    A, B = rnd((K, M), 1, 0.01), rnd((K, N), 2, 0.5)     # A: gradient-like magnitudes
    if code_ == 3:
        A *= 50.0                                          # (fp16 halves are for O(1) operands; the engine uses bf16 halves for gradients)
    C0 = rnd((M, N), 3)
    dA, dB, dC = to_dev(A, dev), to_dev(B, dev), to_dev(C0, dev)
    wsb = int(lib.load().simx_gemm_tn_workspace_bytes(M, N, K))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    outs = []
    for rep in range(2):
        dC = to_dev(C0, dev)
        lib.call("simx_gemm_tn", lib.stream_ptr(), code_, M, N, K, lib.ptr(dA), M, lib.ptr(dB), N, lib.ptr(dC), N, accumulate, lib.ptr(ws), wsb)
        torch.cuda.synchronize()
        outs.append(dC)
    assert torch.equal(outs[0], outs[1]), "split-K slabs are added in slice order: run-to-run identical"
    ref = A.astype(np.float64).T @ B.astype(np.float64) + (C0 if accumulate else 0.0)
    scale = np.abs(A).mean() * np.abs(B).mean() * math.sqrt(K)
    t = dict(rtol=2e-5, atol=3e-5 * scale) if code_ == 3 else dict(rtol=1e-4, atol=1e-4 * scale)
    assert_close(back(outs[0]), ref, what="split gemm_tn", **t)
    # the fused bias gradient (column sums of A taken from the staging registers): exact f32 sums, accumulated into dbias
    db0 = rnd((M,), 4)
    ddb = to_dev(db0, dev)
    dC = to_dev(C0, dev)
    lib.call("simx_gemm_tn_bias", lib.stream_ptr(), code_, M, N, K, lib.ptr(dA), M, lib.ptr(dB), N, lib.ptr(dC), N, accumulate, lib.ptr(ws), wsb,
             lib.ptr(ddb))
    torch.cuda.synchronize()
    assert torch.equal(dC, outs[0])
    assert_close(back(ddb), A.astype(np.float64).sum(0) + db0, rtol=1e-5, atol=1e-5 * np.abs(A).mean() * K, what="split gemm_tn bias gradient")


# ------------------------------------------------------------------------------------------ GEMM TN (wgrad)
@pytest.mark.parametrize("bf16", FMTS)
@pytest.mark.parametrize("M,N,K", [(64, 64, 200), (192, 64, 333), (768, 768, 5000), (3072, 768, 2100), (128, 256, 64), (72, 40, 130)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_gemm_tn(dev, bf16, M, N, K, accumulate):
    lib = L()
    A, B = rnd((K, M), 1, 0.5), rnd((K, N), 2, 0.5)
    C0 = rnd((M, N), 3)
    dA, dB = to_dev(A, dev, bf16), to_dev(B, dev, bf16)
    dC = to_dev(C0, dev)
    wsb = int(lib.load().simx_gemm_tn_workspace_bytes(M, N, K))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    lib.call("simx_gemm_tn", lib.stream_ptr(), code(bf16), M, N, K, lib.ptr(dA), M, lib.ptr(dB), N, lib.ptr(dC), N,
             accumulate, lib.ptr(ws), wsb)
    torch.cuda.synchronize()
    ref = rounded(A, bf16).T @ rounded(B, bf16) + (C0 if accumulate else 0.0)
    assert_close(back(dC), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K), what="gemm_tn")


@pytest.mark.parametrize("M,N,K,acc", [(128, 768, 16384, 0), (96, 772, 5000, 1), (2048, 768, 1024, 0)])
def test_gemm_f32_split_k(dev, M, N, K, acc):
    """M2's backward products (dQ of the local slot over all gathered passages: few output tiles, long K) run split over K
    into f32 slabs added in slice order -- same answer as the unsplit kernel within f32 round-off, and run-to-run identical."""
    lib = L()
    A, B, C0 = rnd((M, K), 1, 0.5), rnd((K, N), 2, 0.5), rnd((M, N), 3)
    dA, dB = to_dev(A, dev), to_dev(B, dev)
    wsb = int(lib.load().simx_gemm_f32_workspace_bytes(M, N, K))
    assert wsb > 0 or M * N >= 256 * 128 * 128, "this shape is meant to split"
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    outs = []
    for rep in range(2):
        dC = to_dev(C0, dev)
        lib.call("simx_gemm_f32_strided_ws", lib.stream_ptr(), M, N, K, lib.ptr(dA), K, 1, lib.ptr(dB), N, 1, lib.ptr(dC), N, acc,
                 lib.ptr(ws), wsb)
        torch.cuda.synchronize()
        outs.append(dC)
    assert torch.equal(outs[0], outs[1]), "split-K result must not depend on scheduling"
    ref = A.astype(np.float64) @ B.astype(np.float64) + (C0 if acc else 0.0)
    assert_close(back(outs[0]), ref, rtol=2e-5, atol=2e-5 * math.sqrt(K), what="gemm_f32 split-K")
    dU = to_dev(C0, dev)
    lib.call("simx_gemm_f32_strided", lib.stream_ptr(), M, N, K, lib.ptr(dA), K, 1, lib.ptr(dB), N, 1, lib.ptr(dU), N, acc)
    torch.cuda.synchronize()
    assert_close(back(outs[0]), back(dU), rtol=1e-5, atol=1e-5 * math.sqrt(K), what="split vs unsplit")


@pytest.mark.parametrize("bf16", FMTS)
@pytest.mark.parametrize("M,N,K", [(768, 768, 4100), (2304, 768, 2048), (512, 256, 2111), (192, 64, 333)])
def test_gemm_tn_fused_bias_grad(dev, bf16, M, N, K):
    """wgrad + bias gradient (column sums of dY) in one call; large shapes take the 256x256 transpose-read kernel."""
    lib = L()
    A, B = rnd((K, M), 1, 0.5), rnd((K, N), 2, 0.5)
    dA, dB = to_dev(A, dev, bf16), to_dev(B, dev, bf16)
    dC = torch.zeros(M, N, device=dev)
    db = torch.full((M,), 0.25, device=dev)
    wsb = int(lib.load().simx_gemm_tn_workspace_bytes(M, N, K))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    lib.call("simx_gemm_tn_bias", lib.stream_ptr(), code(bf16), M, N, K, lib.ptr(dA), M, lib.ptr(dB), N, lib.ptr(dC), N,
             1, lib.ptr(ws), wsb, lib.ptr(db))
    torch.cuda.synchronize()
    assert_close(back(dC), rounded(A, bf16).T @ rounded(B, bf16), rtol=2e-5, atol=2e-5 * math.sqrt(K), what="gemm_tn")
    assert_close(back(db), rounded(A, bf16).sum(0) + 0.25, rtol=1e-5, atol=2e-3, what="fused bias grad")


def test_colsum_and_cast(dev):
    lib = L()
    x = rnd((1234, 200), 1)
    for bf16 in FMTS:
        dx = to_dev(x, dev, bf16)
        out = torch.full((200,), 1.0, device=dev)
        lib.call("simx_colsum", lib.stream_ptr(), code(bf16), 1234, 200, lib.ptr(dx), 200, lib.ptr(out), 1)
        assert_close(back(out), rounded(x, bf16).sum(0) + 1.0, rtol=1e-5, atol=1e-3, what="colsum")
    w = rnd((100, 36), 2)
    dw = to_dev(w, dev)
    o = torch.empty(100, 36, device=dev, dtype=torch.bfloat16)
    ot = torch.empty(36, 100, device=dev, dtype=torch.bfloat16)
    lib.call("simx_cast_weight", lib.stream_ptr(), lib.ptr(dw), 100, 36, lib.ptr(o), lib.ptr(ot))
    assert torch.equal(o, dw.to(torch.bfloat16)) and torch.equal(ot, dw.to(torch.bfloat16).t().contiguous())
    # multiples of 64: the 64 x 64-tile kernel with 16-byte accesses (what simx_bert_cast_weights runs for the BERT geometries)
    for rows, cols, dt, tdt in ((768, 3072, 1, torch.bfloat16), (2304, 768, 2, torch.float16), (64, 128, 2, torch.float16)):
        dw = to_dev(rnd((rows, cols), 3), dev)
        o = torch.full((rows, cols), -1.0, device=dev, dtype=tdt)
        ot = torch.full((cols, rows), -1.0, device=dev, dtype=tdt)
        lib.call("simx_transpose_cast", lib.stream_ptr(), dt, lib.ptr(dw), rows, cols, lib.ptr(o), lib.ptr(ot))
        assert torch.equal(o, dw.to(tdt)) and torch.equal(ot, dw.to(tdt).t().contiguous())
        lib.call("simx_transpose_cast", lib.stream_ptr(), dt, lib.ptr(dw), rows, cols, None, lib.ptr(ot))      # either output may be absent
        torch.cuda.synchronize()


# ------------------------------------------------------------------------------------------ LayerNorm family
@pytest.mark.parametrize("bf16", FMTS)
@pytest.mark.parametrize("T,H", [(37, 64), (300, 768), (50, 1024)])
def test_ln_fwd_bwd(dev, bf16, T, H):
    lib = L()
    z, dy = rnd((T, H), 1, 2.0) + 0.3, rnd((T, H), 2)
    g, b = 1.0 + rnd((H,), 3, 0.1), rnd((H,), 4, 0.1)
    dz_, dy_, dg_, db_ = to_dev(z, dev, bf16), to_dev(dy, dev, bf16), to_dev(g, dev), to_dev(b, dev)
    y = torch.empty_like(dz_)
    lib.call("simx_ln_fwd", lib.stream_ptr(), code(bf16), T, H, lib.ptr(dz_), lib.ptr(dg_), lib.ptr(db_), 1e-12, lib.ptr(y))
    zr = rounded(z, bf16)
    yr, cache = obert._ln_fwd(zr, g.astype(np.float64), b.astype(np.float64), 1e-12)
    assert_close(back(y), yr, what="ln_fwd", **TOL[bf16])
    dzo = torch.empty_like(dz_)
    dgam, dbet, dbias = (torch.zeros(H, device=dev) for _ in range(3))
    lib.call("simx_ln_bwd", lib.stream_ptr(), code(bf16), T, H, lib.ptr(dz_), lib.ptr(dg_), 1e-12, lib.ptr(dy_), lib.ptr(dzo),
             lib.ptr(dgam), lib.ptr(dbet), lib.ptr(dbias))
    dx, dg, db = obert._ln_bwd(rounded(dy, bf16), cache, g.astype(np.float64))
    assert_close(back(dzo), dx, what="ln_bwd dz", **TOL[bf16])
    assert_close(back(dgam), dg, rtol=1e-4, atol=1e-3, what="ln_bwd dgamma")
    assert_close(back(dbet), db, rtol=1e-4, atol=1e-3, what="ln_bwd dbeta")
    assert_close(back(dbias), dx.sum(0), rtol=1e-4, atol=2e-3, what="ln_bwd dbias")


@pytest.mark.parametrize("bf16", FMTS)
def test_embed_ln(dev, bf16):
    lib = L()
    T, H, V, P = 211, 64, 300, 40
    rs = np.random.RandomState(0)
    ids = rs.randint(0, V, size=T).astype(np.int32)
    ids[:20] = 7                                    # collisions on one row (the [CLS] case)
    pos = rs.randint(0, P, size=T).astype(np.int32)
    word, posw, typew = rnd((V, H), 1), rnd((P, H), 2), rnd((2, H), 3)
    g, b = 1.0 + rnd((H,), 4, 0.1), rnd((H,), 5, 0.1)
    dy = rnd((T, H), 6)
    d = lambda a, bf=False: to_dev(a, dev, bf)
    dids, dpos, dword, dposw, dtype_, dg, db = d(ids), d(pos), d(word), d(posw), d(typew), d(g), d(b)
    out = torch.empty(T, H, device=dev, dtype=tdt(bf16))
    lib.call("simx_embed_ln_fwd", lib.stream_ptr(), code(bf16), T, H, lib.ptr(dids), lib.ptr(dpos), lib.ptr(dword), lib.ptr(dposw),
             lib.ptr(dtype_), lib.ptr(dg), lib.ptr(db), 1e-12, lib.ptr(out))
    e = (word[ids] + posw[pos] + typew[0][None]).astype(np.float64)
    yr, cache = obert._ln_fwd(e, g.astype(np.float64), b.astype(np.float64), 1e-12)
    assert_close(back(out), yr, what="embed_ln_fwd", **TOL[bf16])
    ddy = d(dy, bf16)
    gw, gp, gt = torch.zeros(V, H, device=dev), torch.zeros(P, H, device=dev), torch.zeros(2, H, device=dev)
    gg, gb = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    lib.call("simx_embed_ln_bwd", lib.stream_ptr(), code(bf16), T, H, lib.ptr(dids), lib.ptr(dpos), lib.ptr(dword), lib.ptr(dposw),
             lib.ptr(dtype_), lib.ptr(dg), 1e-12, lib.ptr(ddy), lib.ptr(gw), lib.ptr(gp), lib.ptr(gt), lib.ptr(gg), lib.ptr(gb))
    de, dgr, dbr = obert._ln_bwd(rounded(dy, bf16), cache, g.astype(np.float64))
    rw, rp = np.zeros((V, H)), np.zeros((P, H))
    np.add.at(rw, ids, de)
    np.add.at(rp, pos, de)
    assert_close(back(gw), rw, rtol=1e-4, atol=1e-4, what="dword")
    assert_close(back(gp), rp, rtol=1e-4, atol=1e-4, what="dpos")
    assert_close(back(gt)[0], de.sum(0), rtol=1e-4, atol=1e-3, what="dtype0")
    assert np.abs(back(gt)[1]).max() == 0.0
    assert_close(back(gg), dgr, rtol=1e-4, atol=1e-3, what="dgamma")
    assert_close(back(gb), dbr, rtol=1e-4, atol=1e-3, what="dbeta")


@pytest.mark.parametrize("bf16", FMTS)
@pytest.mark.parametrize("lens,off", [([128, 1, 17, 33, 128, 100, 7], 0), ([40] * 70, 2), ([5, 300, 64], 0)])
def test_embed_ln_bwd_position_major(dev, bf16, lens, off):
    """simx_embed_ln_bwd_seq (one wave per in-sequence position, register accumulation of the position gradient) against
    the float64 reference; ragged lengths, more sequences than chunks, RoBERTa-style position offset."""
    lib = L()
    H, V = 64, 300
    T, P = sum(lens), max(lens) + off
    rs = np.random.RandomState(len(lens))
    ids = rs.randint(0, V, size=T).astype(np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    for s_ in range(len(lens)):
        ids[cu[s_]] = 7                              # [CLS] collisions
    pos = np.concatenate([np.arange(n) + off for n in lens]).astype(np.int32)
    word, posw, typew = rnd((V, H), 1), rnd((P, H), 2), rnd((2, H), 3)
    g = 1.0 + rnd((H,), 4, 0.1)
    dy = rnd((T, H), 6)
    d = lambda a, bf=False: to_dev(a, dev, bf)
    gw, gp, gt = torch.zeros(V, H, device=dev), torch.zeros(P, H, device=dev), torch.zeros(2, H, device=dev)
    gg, gb = torch.zeros(H, device=dev), torch.zeros(H, device=dev)
    keep = [d(cu), d(ids), d(pos), d(word), d(posw), d(typew), d(g), d(dy, bf16)]     # must outlive the launch
    lib.call("simx_embed_ln_bwd_seq", lib.stream_ptr(), code(bf16), len(lens), max(lens), T, H, *[lib.ptr(t) for t in keep[:7]], 1e-12,
             lib.ptr(keep[7]), lib.ptr(gw), lib.ptr(gp), lib.ptr(gt), lib.ptr(gg), lib.ptr(gb), None)
    torch.cuda.synchronize()
    e = (word[ids] + posw[pos] + typew[0][None]).astype(np.float64)
    _, cache = obert._ln_fwd(e, g.astype(np.float64), np.zeros(H), 1e-12)
    de, dgr, dbr = obert._ln_bwd(rounded(dy, bf16), cache, g.astype(np.float64))
    rw, rp = np.zeros((V, H)), np.zeros((P, H))
    np.add.at(rw, ids, de)
    np.add.at(rp, pos, de)
    assert_close(back(gw), rw, rtol=1e-4, atol=2e-4, what="dword")
    assert_close(back(gp), rp, rtol=1e-4, atol=2e-4, what="dpos")
    assert_close(back(gt)[0], de.sum(0), rtol=1e-4, atol=2e-3, what="dtype0")
    assert_close(back(gg), dgr, rtol=1e-4, atol=2e-3, what="dgamma")
    assert_close(back(gb), dbr, rtol=1e-4, atol=2e-3, what="dbeta")


# ------------------------------------------------------------------------------------------ attention
def _mha_ref(qkv, lens, heads, d, dctx=None):
    """float64 reference on the packed layout."""
    T = qkv.shape[0]
    H = heads * d
    ctx = np.zeros((T, H))
    lse = np.zeros((heads, T))
    dqkv = np.zeros_like(qkv) if dctx is not None else None
    t0 = 0
    for n in lens:
        for h in range(heads):
            q = qkv[t0:t0 + n, h * d:(h + 1) * d]
            k = qkv[t0:t0 + n, H + h * d:H + (h + 1) * d]
            v = qkv[t0:t0 + n, 2 * H + h * d:2 * H + (h + 1) * d]
            s = q @ k.T / math.sqrt(d)
            m = s.max(1, keepdims=True)
            e = np.exp(s - m)
            p = e / e.sum(1, keepdims=True)
            ctx[t0:t0 + n, h * d:(h + 1) * d] = p @ v
            lse[h, t0:t0 + n] = (m + np.log(e.sum(1, keepdims=True)))[:, 0]
            if dctx is not None:
                do = dctx[t0:t0 + n, h * d:(h + 1) * d]
                dp = do @ v.T
                ds = p * (dp - (dp * p).sum(1, keepdims=True)) / math.sqrt(d)
                dqkv[t0:t0 + n, h * d:(h + 1) * d] = ds @ k
                dqkv[t0:t0 + n, H + h * d:H + (h + 1) * d] = ds.T @ q
                dqkv[t0:t0 + n, 2 * H + h * d:2 * H + (h + 1) * d] = p.T @ do
        t0 += n
    return ctx, lse, dqkv


@pytest.mark.parametrize("bf16,heads,d,lens", [
    (False, 4, 16, [5, 32, 17, 1]),
    (False, 2, 64, [40, 7]),                    # f32 MFMA kernels (head size 64, <= 256 tokens): NKT = 4 tiles of 32
    (False, 3, 64, [9, 32, 4, 31]),             # NKT = 1
    (False, 2, 64, [128, 1, 17, 33, 16, 100]),
    (False, 2, 64, [160, 129, 45]),             # NKT = 5
    (False, 1, 64, [250, 200, 256]),            # NKT = 8
    (False, 1, 64, [300, 33]),                  # > 256: the chunked f32 MFMA kernels (128-token chunks, online softmax)
    (False, 2, 64, [512, 257, 384, 40, 511, 129]),      # whole and ragged chunks, sequences shorter than one chunk
    (False, 1, 64, [1000, 700]),
    (True, 4, 16, [5, 32, 17]),                 # generic kernel in bf16
    (True, 2, 64, [128, 1, 17, 33, 16, 100]),   # MFMA kernel, NKT=8
    (True, 3, 64, [9, 32, 4, 31]),              # NKT=2
    (True, 2, 64, [160, 129, 45]),              # NKT=10
    (True, 1, 64, [250, 200]),                  # NKT=16
    (True, 1, 64, [300, 512, 33]),              # fwd NKT=32, bwd chunked (2 chunks of 256)
    (True, 2, 64, [700, 257, 1024, 40, 256]),   # fwd generic, bwd chunked (up to 4 chunks, ragged tails, exact multiples)
    ("f16", 4, 16, [5, 32, 17]),                # the same kernels on IEEE half
    ("f16", 2, 64, [128, 1, 17, 33, 16, 100]),
    ("f16", 3, 64, [9, 32, 4, 31]),
    ("f16", 2, 64, [160, 129, 45]),
    ("f16", 1, 64, [250, 200]),
    ("f16", 1, 64, [300, 512, 33]),
    ("f16", 2, 64, [700, 257, 1024, 40, 256]),
    # >= 1024 (sequence, head) items of <= 128 tokens: the persistent double-buffered backward (mha_bwd3), ragged lengths,
    # more items than workgroups so that every workgroup walks several
    (True, 12, 64, [1 + (37 * i) % 128 for i in range(90)] + [128, 127, 113]),
    ("f16", 12, 64, [128] * 30 + [1 + (53 * i) % 128 for i in range(70)]),
])
def test_mha_fwd_bwd(dev, bf16, heads, d, lens):
    lib = L()
    T, H = sum(lens), heads * d
    qkv, dctx = rnd((T, 3 * H), 1, 1.0), rnd((T, H), 2)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    dq_, dd_, dcu = to_dev(qkv, dev, bf16), to_dev(dctx, dev, bf16), to_dev(cu, dev)
    adt = tdt(bf16)
    ctx = torch.zeros(T, H, device=dev, dtype=adt)
    lse = torch.zeros(heads, T, device=dev)
    lib.call("simx_mha_fwd", lib.stream_ptr(), code(bf16), len(lens), heads, d, lib.ptr(dcu), max(lens), T, lib.ptr(dq_), lib.ptr(ctx), lib.ptr(lse))
    rc, rl, rdq = _mha_ref(rounded(qkv, bf16), lens, heads, d, rounded(dctx, bf16))
    tol = pick(bf16, dict(rtol=2e-5, atol=2e-5), dict(rtol=2e-2, atol=2e-2), dict(rtol=2.5e-3, atol=2.5e-3))
    assert_close(back(ctx), rc, what="mha ctx", **tol)
    assert_close(back(lse), rl, rtol=1e-5, atol=pick(bf16, 2e-5, 5e-3, 1e-3), what="mha lse")
    dqkv = torch.zeros(T, 3 * H, device=dev, dtype=adt)
    lib.call("simx_mha_bwd", lib.stream_ptr(), code(bf16), len(lens), heads, d, lib.ptr(dcu), max(lens), T, lib.ptr(dq_), lib.ptr(ctx),
             lib.ptr(lse), lib.ptr(dd_), lib.ptr(dqkv))
    tolb = pick(bf16, dict(rtol=1e-4, atol=1e-4), dict(rtol=3e-2, atol=6e-2), dict(rtol=4e-3, atol=8e-3))
    assert_close(back(dqkv), rdq, what="mha dqkv", **tolb)


@pytest.mark.parametrize("bf16,heads,d,lens", [
    (False, 4, 16, [5, 32, 17, 1]),
    (False, 2, 64, [40, 7, 130]),
    (True, 2, 64, [128, 1, 17, 33, 16, 100]),
    (True, 12, 64, [160, 129, 45, 300]),
    ("f16", 2, 64, [128, 1, 17, 33, 16, 100]),
    ("f16", 12, 64, [160, 129, 45, 300]),
])
def test_mha_single_query(dev, bf16, heads, d, lens):
    """The [CLS]-only last layer's attention (simx_mha_cls_fwd / _bwd): token 0 of every sequence is the only query.
    Reference = the float64 full attention with dctx zero outside the [CLS] rows."""
    lib = L()
    T, H, n = sum(lens), heads * d, len(lens)
    qkv, dcc = rnd((T, 3 * H), 11, 1.0), rnd((n, H), 12)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    first = cu[:-1]
    qc = qkv[first, :H].copy()
    qkv_in = qkv.copy()
    qkv_in[:, :H] = 7.0                          # the Q columns of the packed rows must not be read
    adt = tdt(bf16)
    d_qkv, d_qc, d_dcc, dcu = to_dev(qkv_in, dev, bf16), to_dev(qc, dev, bf16), to_dev(dcc, dev, bf16), to_dev(cu, dev)
    ctxc = torch.zeros(n, H, device=dev, dtype=adt)
    lib.call("simx_mha_cls_fwd", lib.stream_ptr(), code(bf16), n, heads, d, lib.ptr(dcu), max(lens), T, lib.ptr(d_qc), lib.ptr(d_qkv),
             lib.ptr(ctxc), None)
    dctx = np.zeros((T, H))
    dctx[first] = rounded(dcc, bf16)
    rc, _, rdq = _mha_ref(rounded(qkv, bf16), lens, heads, d, dctx)
    tol = pick(bf16, dict(rtol=2e-5, atol=2e-5), dict(rtol=2e-2, atol=2e-2), dict(rtol=2.5e-3, atol=2.5e-3))
    assert_close(back(ctxc), rc[first], what="cls ctx", **tol)
    dqc = torch.zeros(n, H, device=dev, dtype=adt)
    dqkv = torch.full((T, 3 * H), 3.0, device=dev, dtype=adt)
    lib.call("simx_mha_cls_bwd", lib.stream_ptr(), code(bf16), n, heads, d, lib.ptr(dcu), max(lens), T, lib.ptr(d_qc), lib.ptr(d_qkv),
             lib.ptr(d_dcc), lib.ptr(dqc), lib.ptr(dqkv), None)
    tolb = pick(bf16, dict(rtol=1e-4, atol=1e-4), dict(rtol=3e-2, atol=6e-2), dict(rtol=4e-3, atol=8e-3))
    got = back(dqkv)
    assert_close(back(dqc), rdq[first, :H], what="cls dq", **tolb)
    assert_close(got[:, H:], rdq[:, H:], what="cls dk dv", **tolb)
    assert np.all(got[:, :H] == 3.0)             # the Q columns of dqkv are left alone


def test_mha_bf16_spiked_scores(dev):
    """large-magnitude logits: one key dominates each row (softmax saturation / max subtraction)."""
    lib = L()
    heads, d, lens = 1, 64, [64]
    T, H = 64, 64
    qkv = rnd((T, 3 * H), 3, 1.0)
    qkv[:, :H] *= 6.0
    qkv[:, H:2 * H] *= 6.0
    cu = np.array([0, 64], dtype=np.int32)
    dq_, dcu = to_dev(qkv, dev, True), to_dev(cu, dev)
    ctx = torch.zeros(T, H, device=dev, dtype=torch.bfloat16)
    lse = torch.zeros(1, T, device=dev)
    lib.call("simx_mha_fwd", lib.stream_ptr(), 1, 1, heads, d, lib.ptr(dcu), 64, T, lib.ptr(dq_), lib.ptr(ctx), lib.ptr(lse))
    rc, rl, _ = _mha_ref(rounded(qkv, True), lens, heads, d)
    assert np.isfinite(back(ctx)).all()
    assert_close(back(ctx), rc, rtol=2e-2, atol=3e-2, what="spiked ctx")
    assert_close(back(lse), rl, rtol=1e-4, atol=2e-2, what="spiked lse")


# ------------------------------------------------------------------------------------------ optimiser
def test_adamw_clip(dev):
    lib = L()
    n = 10007
    p, g = rnd((n,), 1), rnd((n,), 2, 3.0)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    dp, dg, dm, dv = (to_dev(a.copy(), dev) for a in (p, g, m, v))
    pad = lambda t: t          # kernels take any n (tail handled by block 0)
    P, M, V = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in (1, 2, 3):
        sq = torch.zeros(1, device=dev)
        lib.call("simx_sqnorm_accum", lib.stream_ptr(), lib.ptr(dg), n, lib.ptr(sq))
        tot, coef = ooptim.clip_coef([g], 2.0)
        assert abs(math.sqrt(float(sq)) - tot) < 1e-3 * tot
        lib.call("simx_adamw_step", lib.stream_ptr(), lib.ptr(dp), lib.ptr(dg), lib.ptr(dm), lib.ptr(dv), n, 1e-3, 0.9, 0.999, 1e-8, 0.01,
                 step, lib.ptr(sq), 2.0, 1.0, 0)
        ooptim.adamw_hf_step(P, g.astype(np.float64) * coef, M, V, step, 1e-3, wd=0.01)
        assert_close(back(dp), P, rtol=1e-5, atol=1e-6, what="adamw p step %d" % step)
    lib.call("simx_adamw_step", lib.stream_ptr(), lib.ptr(dp), lib.ptr(dg), lib.ptr(dm), lib.ptr(dv), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 4,
             None, 0.0, 0.5, 1)
    assert float(dg.abs().max()) == 0.0


# ------------------------------------------------------------------------------------------ p5: epilogue under the next tile's main loop
@pytest.mark.parametrize("fmt", ["f16", True])
@pytest.mark.parametrize("M,N,K,with_bias,hm", [(65536, 768, 768, True, False), (49152, 2304, 768, True, True), (49152, 2304, 768, False, False),
                                                (65536, 768, 3072, True, False), (65536, 768, 576, True, False), (262144, 768, 768, True, False),
                                                (12800, 1024, 768, True, False),      # 200 tiles: fewer than CUs, one tile per workgroup (tail drain only)
                                                (49152, 1536, 1536, False, True),     # 1152 tiles on 256 workgroups: 4 or 5 tiles each
                                                (327680, 768, 3072, True, False)])    # the launch the headline routes to p5: the teacher's FFN-out, 2048 x 160 tokens
def test_gemm_nt_p5_bit_identical_to_p3(dev, fmt, M, N, K, with_bias, hm):
    """csrc/gemm_p5.hip (one wave per SIMD, accumulators in AGPRs, the finished tile parked in registers and drained through LDS
    inside the next tile's main loop) against gemm_nt_p3_kernel on the same operands: same k order, bias as the accumulators'
    initial value, one rounding -> the outputs must be EQUAL, and both equal the float64 product within the output rounding.
    Shapes: 3 to 36 tiles per workgroup, 8 / 12 / 48 stages per tile, row-major and head-major (QKV projection) outputs."""
    import os
    lib = L()
    A, B = rnd((M, K), 11, 0.5), rnd((N, K), 12, 0.5)
    bias = rnd((N,), 13, 0.5) if with_bias else None
    dA, dB = to_dev(A, dev, fmt), to_dev(B, dev, fmt)
    dbias = to_dev(bias, dev) if with_bias else None
    outs = {}
    old = os.environ.get("SIMX_P5")
    try:
        for mode in ("0", "1"):
            os.environ["SIMX_P5"] = mode
            dC = torch.full((M, N), float("nan"), device=dev, dtype=tdt(fmt))
            if hm:
                lib.call("simx_gemm_nt_hm", lib.stream_ptr(), code(fmt), M, N, K, lib.ptr(dA), K, lib.ptr(dB), K, lib.ptr(dC), 64, lib.ptr(dbias),
                         None, 0, None, 0, M)
            else:
                lib.call("simx_gemm_nt", lib.stream_ptr(), code(fmt), M, N, K, lib.ptr(dA), K, lib.ptr(dB), K, lib.ptr(dC), N,
                         lib.ptr(dbias), None, N, 0, None, N, None, N)
            torch.cuda.synchronize()
            outs[mode] = dC
    finally:
        if old is None:
            os.environ.pop("SIMX_P5", None)
        else:
            os.environ["SIMX_P5"] = old
    assert torch.isfinite(outs["1"].float()).all(), "p5 left output elements unwritten"
    assert torch.equal(outs["0"], outs["1"]), "p5 differs from p3: %d elements, worst %.3e" % (
        int((outs["0"] != outs["1"]).sum()), float((outs["0"].float() - outs["1"].float()).abs().max()))
    rows = np.r_[0:256, M // 2:M // 2 + 256, M - 256:M]                 # three 256-row tiles against float64
    acc = rounded(A[rows], fmt) @ rounded(B, fmt).T + (bias.astype(np.float64) if with_bias else 0.0)
    got = outs["1"]
    if hm:                                                              # [N/64][M][64] -> [M, N]
        got = got.view(N // 64, M, 64).permute(1, 0, 2).reshape(M, N)
    t = dict(TOL[fmt])
    t["atol"] *= max(1.0, math.sqrt(K) * 0.25)
    assert_close(back(got[torch.from_numpy(rows).to(dev)]), acc, what="gemm_nt p5 vs float64", **t)


# ------------------------------------------------------------------------------------------ tn5: the one-wave-per-SIMD wgrad kernel
@pytest.mark.parametrize("fmt", ["f16", True])
@pytest.mark.parametrize("M,N,K", [(768, 768, 16384), (2304, 768, 32768), (768, 3072, 65536), (3072, 768, 24576), (256, 256, 4096),
                                   (768, 768, 16421), (3072, 768, 24576 + 63), (768, 3072, 40000 + 1), (2304, 768, 20000 + 32)])   # ragged token counts
def test_gemm_tn5_vs_float64_and_tn2(dev, fmt, M, N, K):
    """csrc/gemm_tn5.hip (4 waves, 128 x 128 wave tiles, AGPR accumulators, 3 + 2 slot ring, fused bias gradient) through
    simx_gemm_tn_bias against the float64 product of the same 16-bit operands, and against gemm_tn2_kernel (SIMX_TN5=0) on the
    same split plan: the two differ only in the order of f32 additions inside a token slice."""
    import os
    lib = L()
    g = torch.Generator(device=dev)
    g.manual_seed(99 + M + K)
    A = (torch.randn(K, M, device=dev, generator=g) * 0.5).to(tdt(fmt))
    B = (torch.randn(K, N, device=dev, generator=g) * 0.5).to(tdt(fmt))
    C0 = torch.randn(M, N, device=dev, generator=g)
    db0 = torch.randn(M, device=dev, generator=g)
    wsb = int(lib.load().simx_gemm_tn_workspace_bytes(M, N, K))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    outs = {}
    old = os.environ.get("SIMX_TN5")
    try:
        for mode in ("0", "1"):
            os.environ["SIMX_TN5"] = mode
            dC, db = C0.clone(), db0.clone()
            lib.call("simx_gemm_tn_bias", lib.stream_ptr(), code(fmt), M, N, K, lib.ptr(A), M, lib.ptr(B), N, lib.ptr(dC), N, 1, lib.ptr(ws), wsb,
                     lib.ptr(db))
            torch.cuda.synchronize()
            outs[mode] = (dC, db)
    finally:
        if old is None:
            os.environ.pop("SIMX_TN5", None)
        else:
            os.environ["SIMX_TN5"] = old
    Ad, Bd = A.double(), B.double()
    ref = (Ad.t() @ Bd + C0.double()).cpu().numpy()                       # (float64 on the device: a checker, not the product path)
    col = (Ad.sum(0) + db0.double()).cpu().numpy()
    for mode in ("0", "1"):
        got, gb = back(outs[mode][0]), back(outs[mode][1])
        assert_close(got, ref, rtol=2e-5, atol=2e-5 * math.sqrt(K), what="gemm_tn (SIMX_TN5=%s)" % mode)
        assert_close(gb, col, rtol=1e-5, atol=2e-5 * math.sqrt(K), what="fused bias gradient (SIMX_TN5=%s)" % mode)
    d = float((outs["0"][0] - outs["1"][0]).abs().max())
    assert d <= 1e-5 * math.sqrt(K), "tn5 vs tn2: %.3e" % d


def test_compute_cu_budget_changes_the_partition_not_the_result(dev):
    """simx_set_compute_cus (the budget the data-parallel step can give the persistent kernels while an RCCL ring holds CUs,
    profiles/r06_cu_steal.json): 240 instead of 256 workgroups walk the same tiles -- p3 / p5 outputs are EQUAL, the wgrad (another
    split plan: one round of 240) agrees within f32 summation order."""
    lib = L()
    M, N, K = 65536, 768, 768
    A, B, bias = rnd((M, K), 21, 0.5), rnd((N, K), 22, 0.5), rnd((N,), 23, 0.5)
    dA, dB, db = to_dev(A, dev, "f16"), to_dev(B, dev, "f16"), to_dev(bias, dev)
    X = to_dev(rnd((M, 3072), 24, 0.5), dev, "f16")
    W2 = to_dev(rnd((N, 3072), 25, 0.5), dev, "f16")
    outs = {}
    try:
        for budget in (0, 240):
            lib.call("simx_set_compute_cus", budget)
            c1 = torch.empty(M, N, device=dev, dtype=torch.float16)
            c2 = torch.empty(M, N, device=dev, dtype=torch.float16)
            lib.call("simx_gemm_nt", lib.stream_ptr(), 2, M, N, K, lib.ptr(dA), K, lib.ptr(dB), K, lib.ptr(c1), N, lib.ptr(db), None, N, 0, None, N, None, N)
            lib.call("simx_gemm_nt", lib.stream_ptr(), 2, M, N, 3072, lib.ptr(X), 3072, lib.ptr(W2), 3072, lib.ptr(c2), N, lib.ptr(db), None, N, 0, None, N, None, N)
            wsb = int(lib.load().simx_gemm_tn_workspace_bytes(N, K, M))
            ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
            g = torch.zeros(N, K, device=dev)
            lib.call("simx_gemm_tn", lib.stream_ptr(), 2, N, K, M, lib.ptr(c1), N, lib.ptr(dA), K, lib.ptr(g), K, 0, lib.ptr(ws), wsb)
            torch.cuda.synchronize()
            outs[budget] = (c1, c2, g)
    finally:
        lib.call("simx_set_compute_cus", 0)
    assert torch.equal(outs[0][0], outs[240][0]) and torch.equal(outs[0][1], outs[240][1])
    d = float((outs[0][2] - outs[240][2]).abs().max())
    assert d <= 1e-5 * math.sqrt(M) * float(outs[0][2].abs().max()) / 100 + 1e-2, d
