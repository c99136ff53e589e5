"""Attention kernel timing on the bench shape (not part of the product):  python tools/att_bench.py [nseq] [len] [p]
(the A/B of tools/experiments/mha_bwd3.hip was run with it)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simxns_amd import _lib as L       # noqa: E402
from simxns_amd._lib import Dropout    # noqa: E402

if os.environ.get("ATT_LIB"):          # A/B against a variant build (tools/variants/<name>/libsimx_hip.so)
    L.LIB_PATH = os.environ["ATT_LIB"]

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
ragged = len(sys.argv) > 4
heads, d = 12, 64
dev = torch.device("cuda:0")
lens = [S] * nseq if not ragged else [max(1, int(S * (0.3 + 0.7 * ((i * 37) % 100) / 100))) for i in range(nseq)]
T, H = sum(lens), heads * d
R = ((T + 255) // 256) * 256
g = torch.Generator(device="cpu").manual_seed(0)
for code, dt in ((1, torch.bfloat16), (2, torch.float16)):
    qkv = torch.randn(3 * heads, R, 64, generator=g).to(dev).to(dt)
    dctx = (torch.randn(T, H, generator=g) * 0.01).to(dev).to(dt)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]).astype(np.int32), device=dev)
    ctx = torch.zeros(T, H, dtype=dt, device=dev)
    lse = torch.zeros(heads, T, device=dev)
    dq = torch.zeros_like(qkv)
    drop = Dropout(p, 11, 3) if p else None
    dp = C.byref(drop) if drop else None

    def fwd():
        L.call("simx_mha_fwd_hm", L.stream_ptr(), code, nseq, heads, d, L.ptr(cu), S, T, L.ptr(qkv), L.ptr(ctx), L.ptr(lse), dp, R)

    def bwd():
        L.call("simx_mha_bwd_hm", L.stream_ptr(), code, nseq, heads, d, L.ptr(cu), S, T, L.ptr(qkv), L.ptr(ctx), L.ptr(lse), L.ptr(dctx),
               L.ptr(dq), dp, R)

    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        byt = T * H * 2 * ((3 + 1) if name == "fwd" else (3 + 1 + 1 + 3)) + heads * T * 4
        print("%s %s T=%d: %.4f ms  %.2f TB/s" % (str(dt).split(".")[1], name, T, ms,
                                                       byt / ms / 1e9))
