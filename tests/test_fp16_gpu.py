"""The fp16 engine's own machinery: the gradient scale {S, 1/S} through every kernel that carries it, the device-side
dynamic loss scaler, and the optimiser step under it (include/simx.h "gradient scale", "Dynamic loss scaler").

S is a power of two, so scaling is EXACT as long as nothing leaves fp16's range: a kernel fed S x dY with gs = {S, 1/S}
must return the bits it returns for dY with gs = NULL (atomically accumulated outputs: the same values up to the order
of the f32 additions)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def L():
    from simxns_amd import _lib
    return _lib


def rnd(shape, seed, scale=1.0):
    return (np.random.RandomState(seed).randn(*shape) * scale).astype(np.float32)


def h(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(torch.float16)


def gs_of(S, dev):
    return torch.tensor([S, 1.0 / S], dtype=torch.float32, device=dev)


@pytest.mark.parametrize("M,N,K", [(768, 768, 4100), (2304, 768, 2048), (192, 64, 333), (72, 40, 130), (768, 3072, 16384)])
def test_gemm_tn_gs_exact(dev, M, N, K):
    """wgrad with the scale: every path (256x256 split-K + slab reduction + fused bias gradient, 128x128, generic)."""
    lib = L()
    S = 1024.0
    A, B = rnd((K, M), 1, 0.05), rnd((K, N), 2, 0.5)
    dA, dAs, dB = h(A, dev), h(A, dev) * S, h(B, dev)
    assert torch.isfinite(dAs).all()
    wsb = int(lib.load().simx_gemm_tn_workspace_bytes(M, N, K))
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    outs = []
    for a, gs in ((dA, None), (dAs, gs_of(S, dev))):
        c = torch.full((M, N), 0.5, device=dev)
        db = torch.full((M,), 0.25, device=dev)
        lib.call("simx_gemm_tn_gs", lib.stream_ptr(), 2, M, N, K, lib.ptr(a), M, lib.ptr(dB), N, lib.ptr(c), N, 1, lib.ptr(ws), wsb,
                 lib.ptr(db), 0, lib.ptr(gs))
        torch.cuda.synchronize()
        outs.append((c, db))
    assert torch.equal(outs[0][0], outs[1][0]), "scaled wgrad must be bit-identical to the unscaled one"
    ref = dA.double().t() @ dB.double() + 0.5
    assert (outs[0][0].double() - ref).abs().max().item() <= 2e-5 * math.sqrt(K) + 2e-5 * ref.abs().max().item()
    refb = dA.double().sum(0) + 0.25
    for _, db in outs:                             # (atomic accumulation order differs run to run)
        assert (db.double() - refb).abs().max().item() <= 2e-3


def test_ln_bwd_and_embed_bwd_gs(dev):
    lib = L()
    S = 4096.0
    T, H = 300, 768
    z, dy = rnd((T, H), 1, 2.0) + 0.3, rnd((T, H), 2, 0.01)
    g = 1.0 + rnd((H,), 3, 0.1)
    dz_, dy_, dys, dg_ = h(z, dev), h(dy, dev), h(dy, dev) * S, torch.from_numpy(g).to(dev)
    res = []
    for d_in, gs in ((dy_, None), (dys, gs_of(S, dev))):
        dzo = torch.empty(T, H, device=dev, dtype=torch.float16)
        dgam, dbet, dbias = (torch.zeros(H, device=dev) for _ in range(3))
        lib.call("simx_ln_bwd_gs", lib.stream_ptr(), 2, T, H, lib.ptr(dz_), lib.ptr(dg_), 1e-12, lib.ptr(d_in), lib.ptr(dzo), None,
                 lib.ptr(dgam), lib.ptr(dbet), lib.ptr(dbias), None, None, lib.ptr(gs))
        torch.cuda.synchronize()
        res.append((dzo, dgam, dbet, dbias))
    # activation gradient: S x (up to fp16 rounding of a differently scaled value: exact unless subnormal)
    a, b = res[0][0].float() * S, res[1][0].float()
    assert (a - b).abs().max().item() <= 2e-3 * b.abs().max().item()
    for i in (1, 2, 3):                            # parameter gradients: unscaled
        x, y = res[0][i], res[1][i]
        assert (x - y).abs().max().item() <= 1e-3 * x.abs().max().item() + 1e-6
    # embedding backward
    lens = [128, 1, 17, 33, 128, 100, 7]
    Hh, V = 64, 300
    Tt, P = sum(lens), max(lens)
    rs = np.random.RandomState(5)
    ids = rs.randint(0, V, size=Tt).astype(np.int32)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    pos = np.concatenate([np.arange(n) for n in lens]).astype(np.int32)
    word, posw, typew = rnd((V, Hh), 1), rnd((P, Hh), 2), rnd((2, Hh), 3)
    gam = 1.0 + rnd((Hh,), 4, 0.1)
    dyy = rnd((Tt, Hh), 6, 0.01)
    t = lambda x: torch.from_numpy(x).to(dev)
    keep = [t(cu), t(ids), t(pos), t(word), t(posw), t(typew), t(gam)]
    out = []
    for d_in, gs in ((h(dyy, dev), None), (h(dyy, dev) * S, gs_of(S, dev))):
        gw, gp, gt = torch.zeros(V, Hh, device=dev), torch.zeros(P, Hh, device=dev), torch.zeros(2, Hh, device=dev)
        gg, gb = torch.zeros(Hh, device=dev), torch.zeros(Hh, device=dev)
        lib.call("simx_embed_ln_bwd_seq_gs", lib.stream_ptr(), 2, len(lens), max(lens), Tt, Hh, *[lib.ptr(x) for x in keep], 1e-12,
                 lib.ptr(d_in), lib.ptr(gw), lib.ptr(gp), lib.ptr(gt), lib.ptr(gg), lib.ptr(gb), None, lib.ptr(gs))
        torch.cuda.synchronize()
        out.append((gw, gp, gt, gg, gb))
    for x, y in zip(*out):
        assert (x - y).abs().max().item() <= 1e-4 * x.abs().max().item() + 1e-7


def test_entry_points_scale(dev):
    """where an f32 gradient enters the fp16 backward: x S on the way in (rows_copy_gs, cls_scatter_gs, seq_mean_bwd_gs)."""
    lib = L()
    S = 512.0
    n, H = 9, 64
    x = rnd((n, H), 1, 1e-4)                       # would lose most of its bits in fp16 without the scale
    dx, gs = torch.from_numpy(x).to(dev), gs_of(S, dev)
    out = torch.empty(n, H, device=dev, dtype=torch.float16)
    lib.call("simx_rows_copy_gs", lib.stream_ptr(), 0, 2, n, H, None, None, lib.ptr(dx), lib.ptr(out), lib.ptr(gs))
    assert torch.equal(out, (dx * S).to(torch.float16))
    cu = torch.tensor([0, 3, 8, 9, 20, 21, 22, 30, 31, 40], dtype=torch.int32, device=dev)
    full = torch.empty(40, H, device=dev, dtype=torch.float16)
    lib.call("simx_cls_scatter_gs", lib.stream_ptr(), 2, n, H, 40, lib.ptr(cu), lib.ptr(dx), lib.ptr(full), lib.ptr(gs))
    ref = torch.zeros(40, H, device=dev, dtype=torch.float16)
    ref[cu[:-1].long()] = (dx * S).to(torch.float16)
    assert torch.equal(full, ref)
    lib.call("simx_seq_mean_bwd_gs", lib.stream_ptr(), 2, n, H, lib.ptr(cu), lib.ptr(dx), lib.ptr(full), lib.ptr(gs))
    lens = (cu[1:] - cu[:-1]).float()
    ref = torch.repeat_interleave(dx * (S / lens)[:, None], (cu[1:] - cu[:-1]).long(), dim=0).to(torch.float16)
    assert torch.equal(full, ref)


def test_scaler_and_adamw_skip(dev):
    lib = L()
    st = torch.empty(8, device=dev)
    lib.call("simx_scaler_init", lib.stream_ptr(), lib.ptr(st), 65536.0, 3.0, 2.0 ** 24)
    assert st.tolist() == [65536.0, 1.0 / 65536.0, 0, 0, 0, 0, 3.0, 2.0 ** 24]
    n = 1003
    p0, g0 = rnd((n,), 1), rnd((n,), 2)
    p, g = torch.from_numpy(p0).to(dev), torch.from_numpy(g0).to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    for bad in (float("inf"), float("nan")):
        sq = torch.tensor([bad], device=dev)
        S0 = st[0].item()
        lib.call("simx_scaler_update", lib.stream_ptr(), lib.ptr(st), lib.ptr(sq))
        assert st[0].item() == S0 / 2 and st[1].item() == 2 / S0 and st[3].item() == 1.0 and st[4].item() == 0
        g.copy_(torch.from_numpy(g0))
        lib.call("simx_adamw_step_sc", lib.stream_ptr(), lib.ptr(p), lib.ptr(g), lib.ptr(m), lib.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 7,
                 lib.ptr(sq), 2.0, 1.0, 1, lib.ptr(st))
        assert torch.equal(p, torch.from_numpy(p0).to(dev)) and m.abs().max().item() == 0 and v.abs().max().item() == 0
        assert g.abs().max().item() == 0, "a skipped step still drops its gradients"
    assert st[5].item() == 2 and st[0].item() == 16384.0
    # clean steps: applied, bias correction from the APPLIED count (1, 2, ...), growth after 3 clean steps
    from oracle import optim as ooptim
    P, M, V = p0.astype(np.float64), np.zeros(n), np.zeros(n)
    for k in range(1, 5):
        g.copy_(torch.from_numpy(g0))
        sq = torch.zeros(1, device=dev)
        lib.call("simx_sqnorm_accum", lib.stream_ptr(), lib.ptr(g), n, lib.ptr(sq))
        lib.call("simx_scaler_update", lib.stream_ptr(), lib.ptr(st), lib.ptr(sq))
        assert st[3].item() == 0 and st[4].item() == k
        lib.call("simx_adamw_step_sc", lib.stream_ptr(), lib.ptr(p), lib.ptr(g), lib.ptr(m), lib.ptr(v), n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 99,
                 None, 0.0, 1.0, 1, lib.ptr(st))
        ooptim.adamw_hf_step(P, g0.astype(np.float64), M, V, k, 1e-3, wd=0.0)
        assert np.abs(p.cpu().numpy() - P).max() <= 1e-5 * np.abs(P).max()
        assert st[0].item() == (16384.0 if k < 3 else 32768.0)


def test_fp16_training_steps_with_dynamic_scale(dev):
    """FusedAdamW on an fp16 bi-encoder: the first steps at S = 2^16 may overflow and are skipped (apex behaviour), the scale
    settles, applied steps move the parameters, gradients exposed as .grad are the TRUE gradients whatever S is."""
    from simxns_amd import ops
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import BiBertEncoder, HFBertEncoder
    from simxns_amd.optim import FusedAdamW
    cfg = BertConfigLite(vocab_size=200, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                         max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)

    def build(dtype):
        bi = BiBertEncoder.__new__(BiBertEncoder)
        torch.nn.Module.__init__(bi)
        bi.question_model, bi.ctx_model = HFBertEncoder(cfg, dtype), HFBertEncoder(cfg, dtype)
        return bi

    ref, bi = build("fp32"), build("fp16")
    bi.load_state_dict(ref.state_dict())
    ref.to(dev), bi.to(dev)
    B, N = 4, 3
    g = torch.Generator().manual_seed(1)
    q_ids = torch.randint(1, 200, (B, 16), generator=g).to(dev)
    c_ids = torch.randint(1, 200, (B * (1 + N), 32), generator=g).to(dev)
    qm, cm = torch.ones_like(q_ids), torch.ones_like(c_ids)
    z = torch.linspace(-2, 2, B * (1 + N)).reshape(B, 1 + N).to(dev)
    opt = FusedAdamW(bi, lr=1e-3)
    assert opt.scaler is not None and bi.ctx_model.engine.scaler is opt.scaler

    def grads_of(model):
        model.zero_grad()
        q, c = model(q_ids, qm, c_ids, cm)
        loss, _, _ = ops.kl_distill_loss(q, c, z, 1.0, False, 1)
        loss.backward()
        return loss.item(), torch.cat([p.grad.reshape(-1) for p in model.parameters()]).clone()

    l32, g32 = grads_of(ref)
    l16, g16 = grads_of(bi)
    assert abs(l32 - l16) <= 5e-3
    cos = torch.dot(g32, g16) / (g32.norm() * g16.norm())
    assert torch.isfinite(g16).all() and cos.item() >= 0.999 and abs(g16.norm().item() / g32.norm().item() - 1) <= 0.02, \
        "fp16 .grad must be the true (unscaled) gradient: cos %.5f" % cos.item()
    w0 = bi.ctx_model.engine.flat.clone()
    losses = []
    for _ in range(12):
        losses.append(grads_of(bi)[0])
        opt.step(max_grad_norm=2.0)
    snap = opt.scaler.snapshot()
    assert snap["applied_steps"] + snap["skipped_steps"] == 12 and snap["applied_steps"] >= 6, snap
    assert snap["scale"] >= 64.0, snap
    assert not torch.equal(w0, bi.ctx_model.engine.flat) and torch.isfinite(bi.ctx_model.engine.flat).all()
    assert losses[-1] < losses[0], losses
    sd = opt.state_dict()
    assert sd["state"][0]["step"] == snap["applied_steps"] and sd["loss_scaler"]["scale"] == snap["scale"]


@pytest.mark.parametrize("fmt", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("T,H", [(300, 768), (37, 64), (50, 1024)])
def test_ln_residual_stream(dev, fmt, T, H):
    """simx_ln_fwd_res / simx_ln_bwd_res (simx.h stream_lo): y = LN(d + r_hi + r_lo) summed in f32, y leaves as a 16-bit
    value plus a one-byte correction x = hi + (b - 128) ulp(hi) / 256: hi + lo carries the f32 result to 2^-16 (bf16) /
    2^-19 (fp16), and the backward rebuilds the LayerNorm input from the same three tensors."""
    from oracle import bert as obert
    lib = L()
    code = 2 if fmt == torch.float16 else 1
    t16 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(fmt)
    d, r = rnd((T, H), 1, 0.7), rnd((T, H), 2, 1.5) + 0.3
    g, b = 1.0 + rnd((H,), 3, 0.1), rnd((H,), 4, 0.1)
    dd, rh = t16(d), t16(r)

    def ulp256(hi):                                   # spacing of the 16-bit format at hi's exponent, / 256 (float64 [T,H])
        a = np.abs(hi.double().cpu().numpy())
        mant, emin = (10, -14) if fmt == torch.float16 else (7, -111)
        ex = np.maximum(np.floor(np.log2(np.maximum(a, 2.0 ** -200))), emin)
        return 2.0 ** (ex - mant - 8)

    def decode(hi, lo8):
        return hi.double().cpu().numpy() + (lo8.double().cpu().numpy() - 128.0) * ulp256(hi)
    # the correction bytes a producer would have written for the f32 value r
    rq = np.clip(np.rint((r.astype(np.float64) - rh.double().cpu().numpy()) / ulp256(rh) + 128.0), 1, 255)
    rl = torch.from_numpy(rq.astype(np.uint8)).to(dev)
    dg, db = torch.from_numpy(g).to(dev), torch.from_numpy(b).to(dev)
    y, ylo = torch.empty(T, H, device=dev, dtype=fmt), torch.empty(T, H, device=dev, dtype=torch.uint8)
    lib.call("simx_ln_fwd_res", lib.stream_ptr(), code, T, H, lib.ptr(dd), lib.ptr(rh), lib.ptr(rl), lib.ptr(dg), lib.ptr(db), 1e-12,
             lib.ptr(y), lib.ptr(ylo))
    z = dd.double().cpu().numpy() + decode(rh, rl)
    assert np.abs(decode(rh, rl) - r).max() <= (2.0 ** -18 if fmt == torch.float16 else 2.0 ** -15) * np.abs(r).max()
    yr, cache = obert._ln_fwd(z, g.astype(np.float64), b.astype(np.float64), 1e-12)
    got = decode(y, ylo)
    assert int(ylo.min()) >= 1
    eps2 = 2.0 ** -18 if fmt == torch.float16 else 2.0 ** -15
    assert np.abs(got - yr).max() <= eps2 * max(1.0, np.abs(yr).max()) + 1e-6, np.abs(got - yr).max()
    assert torch.equal(y, torch.from_numpy(yr).to(dev).to(fmt)) or (y.double().cpu().numpy() - yr).__abs__().max() <= 2.0 ** -7 * np.abs(yr).max()
    # without the corrections: plain 16-bit residual (res_lo = y_lo = NULL)
    y2 = torch.empty_like(y)
    lib.call("simx_ln_fwd_res", lib.stream_ptr(), code, T, H, lib.ptr(dd), lib.ptr(rh), None, lib.ptr(dg), lib.ptr(db), 1e-12, lib.ptr(y2), None)
    yr2, _ = obert._ln_fwd(dd.double().cpu().numpy() + rh.double().cpu().numpy(), g.astype(np.float64), b.astype(np.float64), 1e-12)
    tol = 2.0 ** -10 if fmt == torch.float16 else 2.0 ** -7
    assert np.abs(y2.double().cpu().numpy() - yr2).max() <= tol * max(1.0, np.abs(yr2).max())
    # backward
    dy = rnd((T, H), 5, 0.3)
    ddy = t16(dy)
    dz = torch.empty(T, H, device=dev, dtype=fmt)
    dgam, dbet, dbias = (torch.zeros(H, device=dev) for _ in range(3))
    lib.call("simx_ln_bwd_res", lib.stream_ptr(), code, T, H, lib.ptr(dd), lib.ptr(rh), lib.ptr(rl), lib.ptr(dg), 1e-12, lib.ptr(ddy),
             lib.ptr(dz), None, lib.ptr(dgam), lib.ptr(dbet), lib.ptr(dbias), None, None, None)
    dx, dgr, dbr = obert._ln_bwd(ddy.double().cpu().numpy(), cache, g.astype(np.float64))
    assert np.abs(dz.double().cpu().numpy() - dx).max() <= tol * max(1.0, np.abs(dx).max())
    assert np.abs(dgam.double().cpu().numpy() - dgr).max() <= 1e-3 * max(1.0, np.abs(dgr).max())
    assert np.abs(dbet.double().cpu().numpy() - dbr).max() <= 1e-3 * max(1.0, np.abs(dbr).max())
    assert np.abs(dbias.double().cpu().numpy() - dx.sum(0)).max() <= 3e-3 * max(1.0, np.abs(dx.sum(0)).max())
