"""Head-major q / k / v layout (include/simx.h): every kernel that reads or writes the packed tensor in the
[3][heads][R][64] form must give EXACTLY what its token-major form gives -- only addresses change.  The kernels' arithmetic
itself is pinned by tests/test_kernels_gpu.py (vs the oracle) and the end-to-end fixtures (the hot-shape golden runs the
head-major path: 32768 passage tokens)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def L():
    from simxns_amd import _lib
    return _lib


def to_hm(x, planes, R):
    """[T, planes*64] token-major -> [planes, R, 64] head-major (rows >= T stay zero)."""
    T = x.shape[0]
    out = torch.zeros(planes, R, 64, dtype=x.dtype, device=x.device)
    out[:, :T] = x.view(T, planes, 64).permute(1, 0, 2)
    return out


def from_hm(x, T):
    planes = x.shape[0]
    return x[:, :T].permute(1, 0, 2).reshape(T, planes * 64).contiguous()


def bf(shape, seed, dev, scale=0.5):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(dev)


@pytest.mark.parametrize("M,N,K", [(16384, 2304, 768), (24576, 1536, 768), (49152, 1024, 256)])
def test_gemm_nt_writes_head_major(dev, M, N, K):
    lib = L()
    A, B = bf((M, K), 1, dev), bf((N, K), 2, dev)
    bias = torch.randn(N, device=dev)
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("simx_gemm_nt", lib.stream_ptr(), 1, M, N, K, lib.ptr(A), K, lib.ptr(B), K, lib.ptr(ref), N, lib.ptr(bias), None, 0, 0,
             None, 0, None, 0)
    R = M + 256                                            # a plane may be longer than the row count in use
    out = torch.full((N // 64, R, 64), float("nan"), dtype=torch.bfloat16, device=dev)
    lib.call("simx_gemm_nt_hm", lib.stream_ptr(), 1, M, N, K, lib.ptr(A), K, lib.ptr(B), K, lib.ptr(out), 64, lib.ptr(bias), None, 0,
             None, 0, R)
    torch.cuda.synchronize()
    assert torch.equal(from_hm(out, M), ref)
    assert torch.isnan(out[:, M:]).all(), "rows beyond M must not be written"


@pytest.mark.parametrize("M,N,K", [(16384, 768, 2304), (32768, 768, 1536), (16384, 1024, 3072)])
def test_gemm_nt_reads_head_major(dev, M, N, K):
    lib = L()
    A, B, res = bf((M, K), 3, dev), bf((N, K), 4, dev), bf((M, N), 5, dev)
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("simx_gemm_nt", lib.stream_ptr(), 1, M, N, K, lib.ptr(A), K, lib.ptr(B), K, lib.ptr(ref), N, None, lib.ptr(res), N, 0,
             None, 0, None, 0)
    R = M
    Ah = to_hm(A, K // 64, R)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    lib.call("simx_gemm_nt_hm", lib.stream_ptr(), 1, M, N, K, lib.ptr(Ah), 64, lib.ptr(B), K, lib.ptr(out), N, None, lib.ptr(res), N,
             None, R, 0)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K,Kcap", [(2304, 768, 16384, 16384), (1536, 768, 20000, 20224), (768, 1024, 4100, 4352)])
def test_gemm_tn_reads_head_major(dev, M, N, K, Kcap):
    """wgrad with dq/dk/dv planes as the token-contracted operand (ragged token count, fused bias gradient)."""
    lib = L()
    A, B = bf((K, M), 6, dev), bf((K, N), 7, dev)
    wsb = int(lib.load().simx_gemm_tn_workspace_bytes(M, N, K))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    ref, dbr = torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)
    lib.call("simx_gemm_tn_bias", lib.stream_ptr(), 1, M, N, K, lib.ptr(A), M, lib.ptr(B), N, lib.ptr(ref), N, 0, lib.ptr(ws), wsb,
             lib.ptr(dbr))
    Ah = to_hm(A, M // 64, Kcap)
    out, db = torch.zeros(M, N, device=dev), torch.zeros(M, device=dev)
    lib.call("simx_gemm_tn_hm", lib.stream_ptr(), 1, M, N, K, lib.ptr(Ah), Kcap, lib.ptr(B), N, lib.ptr(out), N, 0, lib.ptr(ws), wsb,
             lib.ptr(db))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert torch.allclose(db, dbr, rtol=1e-5, atol=1e-3)          # (the bias sums are flushed with f32 atomics)


@pytest.mark.parametrize("heads,lens,p", [(2, [128, 1, 17, 33, 16, 100], 0.0), (3, [160, 129, 45], 0.1), (12, [128] * 8, 0.1),
                                          (1, [250, 200], 0.0), (2, [31, 9, 4], 0.1),
                                          (12, [128] * 40 + [1 + (41 * i) % 128 for i in range(60)], 0.1)])   # persistent backward
def test_attention_head_major(dev, heads, lens, p):
    """simx_mha_fwd_hm / bwd_hm and the [CLS]-row pair: bit-identical to the token-major calls."""
    lib = L()
    from simxns_amd._lib import Dropout
    T, H, d = sum(lens), heads * 64, 64
    R = ((T + 255) // 256) * 256
    qkv, dctx = bf((T, 3 * H), 8, dev, 1.0), bf((T, H), 9, dev, 1.0)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
    drop = Dropout(p, 11, 3) if p else None
    dp = C.byref(drop) if drop else None
    n, S = len(lens), max(lens)

    def run(hm):
        q_in = to_hm(qkv, 3 * heads, R) if hm else qkv
        ctx = torch.zeros(T, H, dtype=torch.bfloat16, device=dev)
        lse = torch.zeros(heads, T, device=dev)
        dq = torch.zeros_like(q_in)
        lib.call("simx_mha_fwd_hm", lib.stream_ptr(), 1, n, heads, d, lib.ptr(cu), S, T, lib.ptr(q_in), lib.ptr(ctx), lib.ptr(lse), dp,
                 R if hm else 0)
        lib.call("simx_mha_bwd_hm", lib.stream_ptr(), 1, n, heads, d, lib.ptr(cu), S, T, lib.ptr(q_in), lib.ptr(ctx), lib.ptr(lse),
                 lib.ptr(dctx), lib.ptr(dq), dp, R if hm else 0)
        # [CLS]-row pair
        qc = qkv[cu[:-1].long(), :H].contiguous()
        cc = torch.zeros(n, H, dtype=torch.bfloat16, device=dev)
        lib.call("simx_mha_cls_fwd_hm", lib.stream_ptr(), 1, n, heads, d, lib.ptr(cu), S, T, lib.ptr(qc), lib.ptr(q_in), lib.ptr(cc), dp,
                 R if hm else 0)
        dcc = dctx[cu[:-1].long()].contiguous()
        dqc = torch.zeros(n, H, dtype=torch.bfloat16, device=dev)
        dkv = torch.zeros_like(q_in)
        lib.call("simx_mha_cls_bwd_hm", lib.stream_ptr(), 1, n, heads, d, lib.ptr(cu), S, T, lib.ptr(qc), lib.ptr(q_in), lib.ptr(dcc),
                 lib.ptr(dqc), lib.ptr(dkv), dp, R if hm else 0)
        torch.cuda.synchronize()
        return ctx, lse, (from_hm(dq, T) if hm else dq), cc, dqc, (from_hm(dkv, T) if hm else dkv)

    a, b = run(False), run(True)
    for x, y, name in zip(a, b, ("ctx", "lse", "dqkv", "ctx_cls", "dq_cls", "dkv_cls")):
        assert torch.equal(x, y), name


@pytest.mark.parametrize("ragged", [False, True])
def test_encoder_layouts_agree(dev, ragged):
    """One tower large enough for the head-major path (16384 tokens): SIMX_QKV_LAYOUT=token vs the default give the same
    embeddings bit for bit and the same gradients up to the order of the f32 atomic sums (LayerNorm / bias gradients)."""
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import HFBertEncoder
    cfg = BertConfigLite(num_hidden_layers=2, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    enc = HFBertEncoder(cfg, compute_dtype="bf16").to(dev).train()
    assert int(L().load().simx_gemm_hm_ok(16384, cfg.hidden_size, 16384)) == 1
    assert int(L().load().simx_gemm_hm_ok(4096, cfg.hidden_size, 4096)) == 0      # (a query tower stays token-major)
    nseq = 224 if ragged else 128
    ids = torch.randint(1000, 20000, (nseq, 128), device=dev)
    mask = torch.ones_like(ids)
    if ragged:                                   # real lengths 40..128: ~19 k real tokens, planes longer than the token count
        lens = torch.randint(40, 129, (nseq,), device=dev)
        mask = (torch.arange(128, device=dev)[None, :] < lens[:, None]).long()
        ids = ids * mask
        assert int(mask.sum()) >= 16384
    res = {}
    for mode in ("token", "head"):
        if mode == "token":
            os.environ["SIMX_QKV_LAYOUT"] = "token"
        else:
            os.environ.pop("SIMX_QKV_LAYOUT", None)
        try:
            enc.zero_grad()
            enc.engine._drop_calls = 0                      # same stateless dropout masks in both runs
            emb = enc.embed(ids, mask) if hasattr(enc, "embed") else enc(ids, mask)[1]
            (emb.float() ** 2).sum().backward()
            torch.cuda.synchronize()
            res[mode] = (emb.detach().float().cpu().numpy().copy(), enc.engine.flat_grad.detach().cpu().numpy().copy())
        finally:
            os.environ.pop("SIMX_QKV_LAYOUT", None)
    assert np.array_equal(res["token"][0], res["head"][0])
    g0, g1 = res["token"][1], res["head"][1]
    assert np.abs(g0 - g1).max() <= 2e-3 * np.abs(g0).max()


def test_plane_blocked_entry_rejects_what_is_not_built(dev):
    """simx_gemm_nt_pb: only the two q/k/v forms exist; everything else must fail loudly, never fall back."""
    lib = L()
    from simxns_amd._lib import SimxError
    M, N, K = 16384, 768, 768
    A, B, C = bf((M, K), 1, dev), bf((N, K), 2, dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for epi, flags, with_in in ((1, 2, False), (2, 6, True), (0, 3, True), (0, 1, False), (0, 2, True)):
        with pytest.raises(SimxError):
            lib.call("simx_gemm_nt_pb", lib.stream_ptr(), 1, M, N, K, lib.ptr(A), K, lib.ptr(B), K, lib.ptr(C), N, None,
                     lib.ptr(C) if with_in else None, N, epi, lib.ptr(C) if epi == 1 else None, N, None, flags, M)
    with pytest.raises(SimxError):                     # a tower too small for the persistent kernel
        lib.call("simx_gemm_nt_pb", lib.stream_ptr(), 1, 4096, N, K, lib.ptr(A), K, lib.ptr(B), K, lib.ptr(C), 64, None, None, 0, 0,
                 None, 0, None, 2, 4096)
    with pytest.raises(SimxError):                     # fp32 has no plane-blocked form
        lib.call("simx_gemm_nt_pb", lib.stream_ptr(), 0, M, N, K, lib.ptr(A), K, lib.ptr(B), K, lib.ptr(C), 64, None, None, 0, 0,
                 None, 0, None, 2, M)


def test_head_major_attention_limit_is_the_same_forward_and_backward(dev):
    """simx_mha_fwd_hm must not accept a head-major layout its backward rejects (sequences above 256 tokens run on the
    token-major chunked kernels)."""
    lib = L()
    from simxns_amd._lib import SimxError
    heads, d, S, n = 2, 64, 320, 2
    T = n * S
    R = 768
    qkv = torch.zeros(3 * heads * R * 64, dtype=torch.bfloat16, device=dev)
    ctx = torch.zeros(T, heads * 64, dtype=torch.bfloat16, device=dev)
    lse = torch.zeros(heads, T, device=dev)
    cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
    with pytest.raises(SimxError):
        lib.call("simx_mha_fwd_hm", lib.stream_ptr(), 1, n, heads, d, lib.ptr(cu), S, T, lib.ptr(qkv), lib.ptr(ctx), lib.ptr(lse), None, R)
    with pytest.raises(SimxError):
        lib.call("simx_mha_bwd_hm", lib.stream_ptr(), 1, n, heads, d, lib.ptr(cu), S, T, lib.ptr(qkv), lib.ptr(ctx), lib.ptr(lse),
                 lib.ptr(ctx), lib.ptr(qkv), None, R)
