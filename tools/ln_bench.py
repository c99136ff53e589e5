"""LayerNorm kernel timing on the bench shape (not part of the product):  python tools/ln_bench.py [T] [H]
plain and residual-stream forms, forward and backward, with and without dropout; algorithmic TB/s."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simxns_amd import _lib as L       # noqa: E402
from simxns_amd._lib import Dropout    # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
H = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
dt, code = torch.float16, 2
d = torch.randn(T, H, generator=g).to(dev).to(dt)
rh = torch.randn(T, H, generator=g).to(dev).to(dt)
rl = torch.randint(0, 256, (T, H), generator=g, dtype=torch.uint8).to(dev)
dy = (torch.randn(T, H, generator=g) * 0.01).to(dev).to(dt)
y, ylo = torch.empty_like(d), torch.empty_like(rl)
dz, dzm = torch.empty_like(d), torch.empty_like(d)
gamma, beta = torch.ones(H, device=dev), torch.zeros(H, device=dev)
dgam, dbet, dbias = torch.zeros(H, device=dev), torch.zeros(H, device=dev), torch.zeros(H, device=dev)
drop = Dropout(0.1, 11, 3)


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


cases = {
    "fwd plain   (z -> y)                 4 B": (lambda: L.call("simx_ln_fwd", L.stream_ptr(), code, T, H, L.ptr(d), L.ptr(gamma), L.ptr(beta), 1e-12, L.ptr(y)), 4),
    "fwd stream  (d, rh, rl -> y, ylo)    8 B": (lambda: L.call("simx_ln_fwd_res", L.stream_ptr(), code, T, H, L.ptr(d), L.ptr(rh), L.ptr(rl), L.ptr(gamma), L.ptr(beta), 1e-12, L.ptr(y), L.ptr(ylo)), 8),
    "bwd plain   (z, dy -> dz)            6 B": (lambda: L.call("simx_ln_bwd_res", L.stream_ptr(), code, T, H, L.ptr(d), None, None, L.ptr(gamma), 1e-12, L.ptr(dy), L.ptr(dz), None, L.ptr(dgam), L.ptr(dbet), L.ptr(dbias), None, None, None), 6),
    "bwd plain + dropout (.. -> dz, dzm)  8 B": (lambda: L.call("simx_ln_bwd_res", L.stream_ptr(), code, T, H, L.ptr(d), None, None, L.ptr(gamma), 1e-12, L.ptr(dy), L.ptr(dz), L.ptr(dzm), L.ptr(dgam), L.ptr(dbet), L.ptr(dbias), C.byref(drop), None, None), 8),
    "bwd stream  (d, rh, rl, dy -> dz)    9 B": (lambda: L.call("simx_ln_bwd_res", L.stream_ptr(), code, T, H, L.ptr(d), L.ptr(rh), L.ptr(rl), L.ptr(gamma), 1e-12, L.ptr(dy), L.ptr(dz), None, L.ptr(dgam), L.ptr(dbet), L.ptr(dbias), None, None, None), 9),
    "bwd stream + dropout (-> dz, dzm)   11 B": (lambda: L.call("simx_ln_bwd_res", L.stream_ptr(), code, T, H, L.ptr(d), L.ptr(rh), L.ptr(rl), L.ptr(gamma), 1e-12, L.ptr(dy), L.ptr(dz), L.ptr(dzm), L.ptr(dgam), L.ptr(dbet), L.ptr(dbias), C.byref(drop), None, None), 11),
}
for name, (fn, bpe) in cases.items():
    ms = timeit(fn)
    print("%s  %.4f ms  %.2f TB/s" % (name, ms, T * H * bpe / ms / 1e9))
