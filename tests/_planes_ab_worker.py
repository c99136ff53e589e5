"""Worker of tests/test_planes_engine_gpu.py: one encoder forward + backward in the fp32 engine, results to an .npz.
The plane switch (SIMX_F32_PLANES / SIMX_F32_PLANES_MIN_TILES) is read once per process by the library, hence a process per mode."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))


def main(out, ckpt, cls_only, dropout):
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import HFBertEncoder
    from simxns_amd.utils import synth
    dev = torch.device("cuda:0")
    cfg = BertConfigLite(vocab_size=3000, hidden_size=256, num_hidden_layers=3, num_attention_heads=4, intermediate_size=1024,
                         max_position_embeddings=160, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    cfg.gradient_checkpointing = bool(ckpt)
    enc = HFBertEncoder(cfg, compute_dtype="fp32")
    enc.load_numpy_state(synth.fill_bert_state_dict([(k, tuple(p.shape)) for k, p in enc.named_parameters()], 77, std=0.08))
    enc.to(dev).train()
    enc.engine.dropout_seed = 123
    enc.engine.ccfg.cls_only_last_layer = int(cls_only)
    rs = np.random.RandomState(5)
    n, S = 40, 128
    lens = rs.randint(20, S + 1, size=n)
    lens[0] = S
    ids = np.zeros((n, S), np.int64)
    mask = np.zeros((n, S), np.int64)
    for i, L in enumerate(lens):
        ids[i, :L] = rs.randint(1000, 3000, size=L)
        mask[i, :L] = 1
    d = torch.from_numpy(rs.randn(n, 256).astype(np.float32)).to(dev)
    e = enc.embed(torch.from_numpy(ids).to(dev), torch.from_numpy(mask).to(dev))
    (e * d).sum().backward()
    torch.cuda.synchronize()
    np.savez(out, emb=e.detach().cpu().numpy(), grad=enc.engine.flat_grad.detach().cpu().numpy())


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]))
