"""Micro-benchmark of M2 (dot_product_scores + in-batch NLL, SimANS/model/models.py:468-505, 564-572) at BASELINE
configs[2] shapes: the gathered global batch of 8 ranks, Q = 1024 queries x C = 16384 passages x H = 768, one rank's local
slot carrying gradient (PROD/ProD_base/train_DE_model_marco.py:224-278).  Not part of bench.py's metric.
usage: python tools/m2_bench.py [Q C H world]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from simxns_amd import ops       # noqa: E402

Q, Cn, H, W = (int(a) for a in sys.argv[1:5]) if len(sys.argv) >= 5 else (1024, 16384, 768, 8)
dev = torch.device("cuda:0")
torch.manual_seed(0)
q = (torch.randn(Q, H, device=dev) * 0.5).requires_grad_(True)
c = (torch.randn(Cn, H, device=dev) * 0.5).requires_grad_(True)
pos = torch.arange(Q, dtype=torch.int32, device=dev) * (Cn // Q)
ql, cl = Q // W, Cn // W


def step():
    q.grad = c.grad = None
    loss, correct = ops.inbatch_nll_loss(q, c, pos, None, (3 * ql, ql), (3 * cl, cl))
    loss.backward()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
flop = 2.0 * Q * Cn * H + 2.0 * ql * Cn * H + 2.0 * cl * Q * H        # scores + dQ_local + dC_local
byts = 4.0 * (Q * H + Cn * H) + 3 * 4.0 * Q * Cn                         # operands + scores written, read, dS written
print(json.dumps({"Q": Q, "C": Cn, "H": H, "world": W, "ms": round(ms, 3), "tflops_f32": round(flop / ms / 1e9, 1),
                  "score_matrix_GBps": round(byts / ms / 1e6, 1),
                  "note": "scores [Q,C] f32 materialised once (67 MB), softmax/NLL/argmax + dS in one pass over it, "
                          "dQ/dC for the local slot only; all three GEMMs on gemm_f32_mfma_kernel"}))
