"""Micro-benchmark of the generate job's search on one GPU (not part of bench.py's metric): f32 inner-product scores
+ exact top-k over a corpus shard resident in HBM.  usage: python tools/search_bench.py [nq nc k]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from simxns_amd.retrieval import FlatIPIndex      # noqa: E402

nq, nc, k = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2048, 1000000, 200)
H = 768
dev = torch.device("cuda:0")
torch.manual_seed(0)
index = FlatIPIndex(H)
for lo in range(0, nc, 250000):
    index.add(torch.randn(min(250000, nc - lo), H, device=dev))
q = torch.randn(nq, H, device=dev)
index.search(q[:64], k)
torch.cuda.synchronize()
t0 = time.perf_counter()
D, I = index.search(q, k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"nq": nq, "nc": nc, "k": k, "seconds": round(dt, 4), "queries_per_s": round(nq / dt, 1),
                  "tflops_f32": round(2.0 * nq * nc * H / dt / 1e12, 2),
                  "full_msmarco_train_job_s_per_gpu_of_8": round(502939 * 8841823 / 8 * H * 2 / (2.0 * nq * nc * H / dt), 1)}))
