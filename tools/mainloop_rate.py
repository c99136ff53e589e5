"""Main-loop rate of the two persistent NT kernels: gemm_nt_p3 (two waves per SIMD, 128 x 64 wave tiles) against gemm_nt_p5 (one wave
per SIMD, 128 x 128 wave tiles, AGPR accumulators) at contraction lengths where the epilogue is < 2 % of a tile (K = 6144, 12288),
fp16 random operands, plain bias epilogue.  Same process, alternating (SIMX_P5 is read per call).  DESIGN.md section 6, item 1."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simxns_amd import _lib as L  # noqa: E402

dev = torch.device("cuda", 0)
L.load()
F16 = 2
for M, N, K in ((65536, 768, 12288), (131072, 768, 6144), (65536, 2304, 6144), (262144, 768, 3072), (262144, 768, 768)):
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    A = (torch.randn(M, K, device=dev, generator=g) * 0.5).half()
    B = (torch.randn(N, K, device=dev, generator=g) * 0.5).half()
    bias = torch.randn(N, device=dev, generator=g)
    C = torch.empty(M, N, device=dev, dtype=torch.float16)
    res = {}
    for rep in range(3):
        for mode in ("0", "1"):
            os.environ["SIMX_P5"] = mode
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for it in range(3 + 10):
                if it == 3:
                    e0.record()
                L.call("simx_gemm_nt", L.stream_ptr(), F16, M, N, K, L.ptr(A), K, L.ptr(B), K, L.ptr(C), N, L.ptr(bias), None, N, 0, None, N, None, N)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(mode, []).append(e0.elapsed_time(e1) / 10)
    fl = 2.0 * M * N * K
    p3, p5 = min(res["0"]), min(res["1"])
    print("M=%d N=%d K=%d: p3 %.3f ms = %.0f TFLOP/s | p5 %.3f ms = %.0f TFLOP/s | p5 / p3 = %.3f" % (M, N, K, p3, fl / p3 / 1e9, p5, fl / p5 / 1e9, p3 / p5))
    del A, B, C
