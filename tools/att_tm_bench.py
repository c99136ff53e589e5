"""Token-major 16-bit attention timing: python tools/att_tm_bench.py nseq S heads p"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simxns_amd import _lib as L
from simxns_amd._lib import Dropout
nseq, S, heads, p = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
dev = torch.device("cuda:0"); d = 64; T = nseq * S; H = heads * d
qkv = torch.randn(T, 3 * H, device=dev).half(); dctx = (torch.randn(T, H, device=dev) * 0.01).half()
cu = torch.arange(0, T + 1, S, dtype=torch.int32, device=dev)
ctx = torch.zeros(T, H, dtype=torch.half, device=dev); lse = torch.zeros(heads, T, device=dev); dq = torch.zeros_like(qkv)
drop = Dropout(p, 11, 3) if p else None; dp = C.byref(drop) if drop else None
fwd = lambda: L.call("simx_mha_fwd_ex", L.stream_ptr(), 2, nseq, heads, d, L.ptr(cu), S, T, L.ptr(qkv), L.ptr(ctx), L.ptr(lse), dp)
bwd = lambda: L.call("simx_mha_bwd_ex", L.stream_ptr(), 2, nseq, heads, d, L.ptr(cu), S, T, L.ptr(qkv), L.ptr(ctx), L.ptr(lse), L.ptr(dctx), L.ptr(dq), dp)
for name, fn in (("fwd", fwd), ("bwd", bwd)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    print("f16 %s S=%d x %d x %d heads p=%g: %.3f ms" % (name, S, nseq, heads, p, a.elapsed_time(b) / 20))
