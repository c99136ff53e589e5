"""Timing of the plane-pair attention kernels (csrc/attention_x3.hip):  python tools/x3_bench.py [nseq] [len] [heads] [p]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_planes_gpu import planes_of, rnd, L, F16      # noqa: E402

nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 12
p = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
dev = torch.device("cuda:0")
lib = L()
d, T, H = 64, nseq * S, heads * 64
qkv = torch.randn(T, 3 * H, device=dev)
dctx = torch.randn(T, H, device=dev) * 1e-3
qp = planes_of(qkv, F16, dev)
cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
drop = lib.Dropout(p, 99, 11) if p else None
dp = C.byref(drop) if drop else None
ctxp = torch.zeros(2, T, H, device=dev, dtype=torch.int16)
lse = torch.zeros(heads, T, device=dev)
g = torch.zeros(2, T, 3 * H, device=dev, dtype=torch.int16)
fwd = lambda: lib.call("simx_mha_fwd_x3", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H, lib.ptr(lse), dp)
bwd = lambda: lib.call("simx_mha_bwd_x3", lib.stream_ptr(), nseq, heads, d, lib.ptr(cu), S, T, lib.ptr(qp), T * 3 * H, lib.ptr(ctxp), T * H,
                       lib.ptr(lse), lib.ptr(dctx), lib.ptr(g), T * 3 * H, dp)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("x3 %s S=%d x %d seqs x %d heads p=%g: %.3f ms" % (name, S, nseq, heads, p, a.elapsed_time(b) / 20))
