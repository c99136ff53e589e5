"""Does the attention kernels' bandwidth depend on the qkv row stride?  Same number of (sequence, head) work items and the
same bytes, once as 2048 sequences x 12 heads (K/V rows of a head are 128-B pieces 4608 B apart) and once as 24576
sequences x 1 head (384-B rows: the pieces of one item are adjacent).  Not part of the product."""
import ctypes as C
import sys
import time
import torch
sys.path.insert(0, ".")
from simxns_amd import _lib as L

dev = torch.device("cuda", 0)
S = 128
for heads, nseq in ((12, 2048), (1, 2048 * 12)):
    H = heads * 64
    T = nseq * S
    qkv = (torch.randn(T, 3 * H, device=dev) * 0.5).to(torch.bfloat16)
    dctx = (torch.randn(T, H, device=dev) * 0.5).to(torch.bfloat16)
    ctx = torch.empty(T, H, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(heads, T, device=dev)
    dqkv = torch.empty_like(qkv)
    cu = (torch.arange(nseq + 1, device=dev, dtype=torch.int32) * S)
    f = lambda: L.call("simx_mha_fwd", L.stream_ptr(), 1, nseq, heads, 64, L.ptr(cu), S, T, L.ptr(qkv), L.ptr(ctx), L.ptr(lse))
    b = lambda: L.call("simx_mha_bwd", L.stream_ptr(), 1, nseq, heads, 64, L.ptr(cu), S, T, L.ptr(qkv), L.ptr(ctx), L.ptr(lse), L.ptr(dctx), L.ptr(dqkv))
    for name, fn, nbytes in (("fwd", f, T * H * 2 * 4), ("bwd", b, T * H * 2 * 8)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print("heads=%2d nseq=%5d %s: %.3f ms  %.2f TB/s" % (heads, nseq, name, dt * 1e3, nbytes / dt / 1e12))
