"""fp32-engine attention on the 16-bit matrix cores from fp16 plane pairs (csrc/attention_x3.hip) against float64 NumPy on the
plane values, at f32-grade tolerances (NOT 16-bit ones), and against the f32 MFMA kernels it replaces (timing printed)."""
import math

import numpy as np
import pytest
import torch

from tests.test_kernels_gpu import _mha_ref
from tests.test_planes_gpu import planes_of, planes_value, rnd, L, F16, BF16, F32

pytestmark = pytest.mark.gpu

LENS = [[40, 7], [9, 32, 4, 31], [128, 1, 17, 33, 16, 100], [160, 129, 45], [250, 200, 256], [128] * 6,
        [512, 300, 257, 5], [640, 129, 384], [1030, 77]]        # > 256 tokens: the chunked kernels (128-token chunks, online softmax)


def _ms(fn, n=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


@pytest.mark.parametrize("heads", [1, 3])
@pytest.mark.parametrize("lens", LENS)
@pytest.mark.parametrize("scale_q", [1.0, 4.0])                 # 4.0: peaked softmax rows (scores up to ~100)
def test_mha_fwd_x3(dev, heads, lens, scale_q):
    lib = L()
    d, T, H = 64, sum(lens), heads * 64
    qkv = rnd((T, 3 * H), 1, 1.0)
    qkv[:, :H] *= scale_q
    dq = torch.from_numpy(qkv).to(dev)
    qp = planes_of(dq, F16, dev)
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
    ctxp = torch.zeros(2, T, H, device=dev, dtype=torch.int16)
    lse = torch.zeros(heads, T, device=dev)
    lib.call("simx_mha_fwd_x3", lib.stream_ptr(), len(lens), heads, d, lib.ptr(cu), max(lens), T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
             lib.ptr(lse), None)
    torch.cuda.synchronize()
    rc, rl, _ = _mha_ref(planes_value(qp, F16), lens, heads, d)
    got = planes_value(ctxp, F16)
    assert np.abs(got - rc).max() <= 3e-6 * max(1.0, np.abs(rc).max()), np.abs(got - rc).max()
    assert np.abs(lse.cpu().numpy() - rl).max() <= 2e-5 * max(1.0, np.abs(rl).max())


def test_mha_fwd_x3_dropout_and_speed(dev):
    """dropout on the probabilities (same stateless mask as the f32 kernels: equal results), and the rate against the f32 MFMA
    kernel on the benchmark's passage-tower shape."""
    import ctypes as C
    lib = L()
    heads, d, S, nseq = 12, 64, 128, 512
    T, H = nseq * S, heads * 64
    qkv = torch.from_numpy(rnd((T, 3 * H), 3, 1.0)).to(dev)
    qp = planes_of(qkv, F16, dev)
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
    drop = lib.Dropout(0.1, 99, 11)
    ctxp = torch.zeros(2, T, H, device=dev, dtype=torch.int16)
    ctxq = torch.zeros(2, T, H, device=dev, dtype=torch.int16)
    lse, lse2 = torch.zeros(heads, T, device=dev), torch.zeros(heads, T, device=dev)
    x3 = lambda dr: lib.call("simx_mha_fwd_x3", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
                             lib.ptr(lse), C.byref(dr) if dr else None)
    f32 = lambda dr: lib.call("simx_mha_fwd_planes", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qkv), lib.ptr(ctxq), T * H,
                              lib.ptr(lse2), C.byref(dr) if dr else None)
    x3(drop)
    f32(drop)
    torch.cuda.synchronize()
    a, b = planes_value(ctxp, F16), planes_value(ctxq, F16)
    assert np.abs(a - b).max() <= 5e-6 * max(1.0, np.abs(b).max()), np.abs(a - b).max()
    assert np.abs(lse.cpu().numpy() - lse2.cpu().numpy()).max() <= 2e-5 * 20
    t3, t32 = _ms(lambda: x3(None)), _ms(lambda: f32(None))
    print("mha_fwd S=128 x %d seqs x 12 heads: x3 %.3f ms, f32 MFMA %.3f ms" % (nseq, t3, t32))
    # probabilities travel as 2^10 p / (1 - p_drop) in fp16 halves: p_drop = 0.9 is finite, above it the call refuses (it used to
    # return inf / NaN silently from p_drop ~ 0.984 on)
    from simxns_amd._lib import SimxError
    x3(lib.Dropout(0.9, 99, 11))
    torch.cuda.synchronize()
    assert np.isfinite(planes_value(ctxp, F16)).all()
    with pytest.raises(SimxError):
        x3(lib.Dropout(0.95, 99, 11))


@pytest.mark.parametrize("heads", [1, 3])
@pytest.mark.parametrize("lens", LENS)
@pytest.mark.parametrize("gscale,spread", [(1.0, 0.0), (1.0, 2.0), (1e-7, 2.0), (3e4, 0.0)])   # gradients of any magnitude: the per-block
def test_mha_bwd_x3(dev, heads, lens, gscale, spread):                                          # power-of-two scaling; rows of equal / very different size
    lib = L()
    d, T, H = 64, sum(lens), heads * 64
    qkv = rnd((T, 3 * H), 1, 1.0)
    dctx = (rnd((T, H), 2) * np.exp(rnd((T, 1), 3, spread)) * gscale).astype(np.float32)
    dq = torch.from_numpy(qkv).to(dev)
    qp = planes_of(dq, F16, dev)
    cu = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)).to(dev)
    ctxp = torch.zeros(2, T, H, device=dev, dtype=torch.int16)
    lse = torch.zeros(heads, T, device=dev)
    lib.call("simx_mha_fwd_x3", lib.stream_ptr(), len(lens), heads, d, lib.ptr(cu), max(lens), T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
             lib.ptr(lse), None)
    dd = torch.from_numpy(dctx).to(dev)
    dqkvp = torch.zeros(2, T, 3 * H, device=dev, dtype=torch.int16)
    db0 = rnd((3 * H,), 7, float(gscale))
    db = torch.from_numpy(db0.copy()).to(dev)
    lib.call("simx_mha_bwd_x3_bias", lib.stream_ptr(), len(lens), heads, d, lib.ptr(cu), max(lens), T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
             lib.ptr(lse), lib.ptr(dd), lib.ptr(dqkvp), T * 3 * H, None, lib.ptr(db))
    torch.cuda.synchronize()
    _, _, rdq = _mha_ref(planes_value(qp, F16), lens, heads, d, dctx.astype(np.float64))
    got = planes_value(dqkvp, BF16)
    assert np.isfinite(got).all()
    # the fused QKV bias gradient: column sums over the tokens, accumulated into the buffer
    # (measured: 1e-6 of the column's L1 norm with rows of one magnitude -- the f32 MFMA kernels give 7e-7 --; with row magnitudes
    # spread over e^+-6 inside a block the per-block scale leaves the small rows ~1e-5 of relative precision and the sum 1e-4)
    gb = db.cpu().numpy().astype(np.float64)
    # (the absolute floor sits below the BLOCK's largest dO element, the column's L1 norm is carried by its few largest rows: the
    # ratio grows with the rows of a block -- 1030 here against 256: measured 8e-4)
    ctol = (8e-6 if spread == 0.0 else 4e-4) * max(1.0, max(lens) / 256.0)
    assert np.all(np.abs(gb - (rdq.sum(0) + db0)) <= ctol * np.abs(rdq).sum(0) + 1e-6 * np.abs(db0) + 1e-30), np.abs(gb - (rdq.sum(0) + db0)).max()
    # per (sequence, head) block the error is bounded relative to the block's gradient scale: bf16 pair output (2^-16) + the products
    t0 = 0
    for n in lens:
        for hh in range(heads):
            sls = [(slice(t0, t0 + n), slice(w * H + hh * 64, w * H + (hh + 1) * 64)) for w in range(3)]
            gmax = max(np.abs(rdq[sl]).max() for sl in sls)          # (a one-token sequence has dq = dk = 0 exactly: scale by the block)
            for w, sl in enumerate(sls):
                err = np.abs(got[sl] - rdq[sl]).max()
                # (sequences past 256 tokens: the floor below the block's largest dO element is summed over more rows -- 1030 here:
                # measured 2.3e-5 with rows of one magnitude, 5.2e-5 with magnitudes spread over e^+-6)
                assert err <= 2e-5 * max(1.0, n / 256.0) * gmax + 1e-30, (n, hh, w, err, gmax)
        t0 += n


def test_mha_bwd_x3_dropout_and_speed(dev):
    import ctypes as C
    lib = L()
    heads, d, S, nseq = 12, 64, 128, 512
    T, H = nseq * S, heads * 64
    qkv = torch.from_numpy(rnd((T, 3 * H), 3, 1.0)).to(dev)
    dctx = torch.from_numpy(rnd((T, H), 4, 1e-3)).to(dev)
    qp = planes_of(qkv, F16, dev)
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
    drop = lib.Dropout(0.1, 99, 11)
    ctxp = torch.zeros(2, T, H, device=dev, dtype=torch.int16)
    lse = torch.zeros(heads, T, device=dev)
    lib.call("simx_mha_fwd_x3", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H, lib.ptr(lse), C.byref(drop))
    g3 = torch.zeros(2, T, 3 * H, device=dev, dtype=torch.int16)
    g32 = torch.zeros(2, T, 3 * H, device=dev, dtype=torch.int16)
    x3 = lambda dr: lib.call("simx_mha_bwd_x3", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
                             lib.ptr(lse), lib.ptr(dctx), lib.ptr(g3), T * 3 * H, C.byref(dr) if dr else None)
    f32 = lambda dr: lib.call("simx_mha_bwd_planes", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qkv), lib.ptr(ctxp), T * H,
                              lib.ptr(lse), lib.ptr(dctx), lib.ptr(g32), T * 3 * H, C.byref(dr) if dr else None)
    x3(drop)
    f32(drop)
    torch.cuda.synchronize()
    a, b = planes_value(g3, BF16), planes_value(g32, BF16)
    assert np.abs(a - b).max() <= 4e-5 * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())
    t3, t32 = _ms(lambda: x3(None)), _ms(lambda: f32(None))
    print("mha_bwd S=128 x %d seqs x 12 heads: x3 %.3f ms, f32 MFMA %.3f ms" % (nseq, t3, t32))


@pytest.mark.parametrize("S,nseq", [(512, 96), (400, 40)])
def test_mha_x3_long_dropout_and_speed(dev, S, nseq):
    """The chunked kernels with dropout against the chunked f32 MFMA kernels they replace (same stateless mask), and both rates on the
    MS-MARCO Document shape (BASELINE configs[4]: 512-token documents, 16 heads)."""
    import ctypes as C
    lib = L()
    heads, d = 16, 64
    T, H = nseq * S, heads * 64
    qkv = torch.from_numpy(rnd((T, 3 * H), 3, 1.0)).to(dev)
    dctx = torch.from_numpy(rnd((T, H), 4, 1e-3)).to(dev)
    qp = planes_of(qkv, F16, dev)
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
    drop = lib.Dropout(0.1, 99, 11)
    ctxp, ctxq = torch.zeros(2, T, H, device=dev, dtype=torch.int16), torch.zeros(2, T, H, device=dev, dtype=torch.int16)
    lse, lse2 = torch.zeros(heads, T, device=dev), torch.zeros(heads, T, device=dev)
    f3 = lambda dr: lib.call("simx_mha_fwd_x3", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
                             lib.ptr(lse), C.byref(dr) if dr else None)
    f32 = lambda dr: lib.call("simx_mha_fwd_planes", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qkv), lib.ptr(ctxq), T * H,
                              lib.ptr(lse2), C.byref(dr) if dr else None)
    f3(drop)
    f32(drop)
    torch.cuda.synchronize()
    a, b = planes_value(ctxp, F16), planes_value(ctxq, F16)
    assert np.abs(a - b).max() <= 5e-6 * max(1.0, np.abs(b).max()), np.abs(a - b).max()
    assert np.abs(lse.cpu().numpy() - lse2.cpu().numpy()).max() <= 2e-5 * 20
    g3, g32 = torch.zeros(2, T, 3 * H, device=dev, dtype=torch.int16), torch.zeros(2, T, 3 * H, device=dev, dtype=torch.int16)
    b3 = lambda dr: lib.call("simx_mha_bwd_x3", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
                             lib.ptr(lse), lib.ptr(dctx), lib.ptr(g3), T * 3 * H, C.byref(dr) if dr else None)
    b32 = lambda dr: lib.call("simx_mha_bwd_planes", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qkv), lib.ptr(ctxp), T * H,
                              lib.ptr(lse), lib.ptr(dctx), lib.ptr(g32), T * 3 * H, C.byref(dr) if dr else None)
    b3(drop)
    b32(drop)
    torch.cuda.synchronize()
    a, b = planes_value(g3, BF16), planes_value(g32, BF16)
    assert np.isfinite(a).all()
    assert np.abs(a - b).max() <= 4e-5 * np.abs(b).max(), (np.abs(a - b).max(), np.abs(b).max())
    print("mha S=%d x %d seqs x 16 heads: forward x3 %.3f ms, f32 MFMA %.3f ms; backward x3 %.3f ms, f32 MFMA %.3f ms"
          % (S, nseq, _ms(lambda: f3(None)), _ms(lambda: f32(None)), _ms(lambda: b3(None)), _ms(lambda: b32(None))))
