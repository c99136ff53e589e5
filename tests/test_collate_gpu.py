"""D1 device-side batch assembly (simx_assemble_batch) -- integer work, bit-exact:
against the fixture written by the imported reference's dataset + collate (tests/golden/collate_ref.npz), and against
the oracle at BASELINE config-2 sizes with ragged / empty / maximum-length rows and a RoBERTa-style pad id."""
import os

import numpy as np
import pytest
import torch

from oracle import collate as oc

pytestmark = pytest.mark.gpu


def _check(out, ref):
    s, t = out["student"], out["teacher"]
    got = dict(q_ids=s[0], q_mask=s[1], ctx_ids=s[2], ctx_mask=s[3], ce_ids=t[0], ce_mask=t[1], tgt=t[2])
    for k, v in got.items():
        assert v.dtype == torch.int64
        assert (v.cpu().numpy() == ref[k]).all(), k
    assert s[4] == list(ref["positive_ctx_indices"])


def test_collate_vs_reference_golden(dev, golden_dir):
    from simxns_amd import ops
    G = np.load(os.path.join(golden_dir, "collate_ref.npz"))
    B, D = G["ce_ids"].shape[:2]
    q_tok = torch.from_numpy(G["q_ids"].astype(np.int32)).to(dev)
    p_tok = torch.from_numpy(G["ctx_ids"].astype(np.int32)).to(dev)
    out = ops.assemble_batch(q_tok, p_tok, list(range(B)), list(range(B * D)), D)
    torch.cuda.synchronize()
    ref = {k: G[k] for k in ("q_ids", "q_mask", "ctx_ids", "ctx_mask", "ce_ids", "ce_mask", "tgt")}
    ref["positive_ctx_indices"] = [int(v) for v in G["pos"]]
    _check(out, ref)


@pytest.mark.parametrize("pad,sep,B,D,NQ,NP", [(0, 102, 128, 16, 500, 20000), (1, 2, 32, 16, 64, 3000), (0, 102, 3, 1, 3, 7)])
def test_collate_vs_oracle_full_size(dev, pad, sep, B, D, NQ, NP):
    from simxns_amd import ops
    rs = np.random.RandomState(B + D)
    QL, PL, CL = 32, 128, 160

    def table(n, S, lo):
        t = np.full((n, S), pad, np.int32)
        lens = rs.randint(lo, S + 1, size=n)
        lens[0], lens[-1] = S, lo                                  # maximum-length and shortest rows
        for i in range(n):
            t[i, :lens[i]] = rs.randint(1000, 30000, size=lens[i])
            t[i, 0] = 101
            if lens[i] > 1 and i % 3:                              # two thirds end with [SEP]
                t[i, lens[i] - 1] = sep
        return t
    q_tok, p_tok = table(NQ, QL, 2), table(NP, PL, 1)
    p_tok[1, :] = pad                                              # an empty passage row
    q_rows = rs.randint(0, NQ, size=B)
    p_rows = rs.randint(0, NP, size=B * D)
    p_rows[:3] = [0, 1, NP - 1]
    out = ops.assemble_batch(torch.from_numpy(q_tok).to(dev), torch.from_numpy(p_tok).to(dev), q_rows, p_rows.reshape(B, D), D,
                             pad_id=pad, sep_id=sep, ce_len=CL)
    torch.cuda.synchronize()
    ref = oc.assemble(q_tok, p_tok, q_rows, p_rows, D, pad_id=pad, sep_id=sep, ce_len=CL)
    _check(out, ref)
    assert (out["lens"]["ce"].cpu().numpy() == ref["ce_len"]).all()
    assert (out["lens"]["ctx"].cpu().numpy() == np.array([oc.row_len(p_tok[r], pad) for r in p_rows])).all()
    # size-independent property: masks are exactly ids != pad and every ce row is (question prefix, passage infix)
    ce = out["teacher"][0].cpu().numpy().reshape(B * D, CL)
    ql = out["lens"]["q"].cpu().numpy()
    for r in (0, 1, 2, B * D - 1):
        b = r // D
        assert (ce[r, :ql[b]] == q_tok[q_rows[b], :ql[b]]).all()
