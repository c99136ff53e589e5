"""Encoder forward + backward at S = 512 (MS-Doc / config-5 geometry): where the time goes."""
import ctypes as C
import sys
import time

import torch

sys.path.insert(0, ".")
from simxns_amd import _lib as L
from simxns_amd.engine import BertConfigLite
from simxns_amd.model.models import HFBertEncoder

dev = torch.device("cuda:0")
nseq, S = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 512
large = len(sys.argv) > 2 and sys.argv[2] == "large"
cfg = (BertConfigLite(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                      hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=514) if large else
       BertConfigLite(hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, max_position_embeddings=514))
enc = HFBertEncoder(cfg, "bf16").to(dev).train()
ids = torch.randint(1000, 30000, (nseq, S), device=dev)
mask = torch.ones_like(ids)
for it in range(3):
    if it == 1:
        torch.cuda.synchronize()
        L.call("simx_prof_begin", 8192)
        t0 = time.perf_counter()
    cls = enc.embed(ids, mask)
    cls.sum().backward()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
nk = L.load().simx_prof_kernel_count()
cnt, ms, wk = (C.c_int32 * nk)(), (C.c_double * nk)(), (C.c_double * nk)()
L.call("simx_prof_end", cnt, ms, wk)
H, Lh, F = cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size
fl = 3.0 * nseq * Lh * S * (8.0 * H * H + 4.0 * H * F + 4.0 * S * H)
print("S=512 nseq=%d %s: %.1f ms per fwd+bwd = %.0f TFLOP/s algorithmic (%.1f%% of 2.5 PF)" % (nseq, "BERT-large" if large else "BERT-base", dt * 1e3, fl / dt / 1e12, fl / dt / 2.5e13))
print({L.PROF_NAMES[k]: round(ms[k] / 2, 2) for k in range(nk) if cnt[k]})
