"""Training-mode dropout (nn.Dropout p=0.1 in BertEmbeddings / attention probabilities / BertSelfOutput / BertOutput,
forced on by SimANS/model/models.py:70-72): the product's stateless hash masks are reproduced by the oracle, so a
dropout-ON step is compared number for number (forward, loss, every gradient).  The reference's torch RNG stream cannot
be replayed on a GPU kernel; what is pinned is the placement / scaling / backward of each dropout site and the keep rate."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import bert as ob
from oracle import losses as ol
from oracle.weights import BertCfg, make_bert_params, make_batch

pytestmark = pytest.mark.gpu


def _fixed_seed(enc, seed, full_last_layer=False):
    from simxns_amd import _lib as L
    def call_cfg(training, want_hidden=True):
        c = L.BertCfg.from_buffer_copy(enc.engine.ccfg)
        c.cls_only_last_layer = 0 if (want_hidden or full_last_layer) else 1
        if training:
            c.hidden_dropout, c.attn_dropout, c.dropout_seed = enc.config.hidden_dropout_prob, enc.config.attention_probs_dropout_prob, seed
        return c
    enc.engine.call_cfg = call_cfg


@pytest.mark.parametrize("full_last_layer", [False, True])     # False: the [CLS]-only last layer (what the towers run); True: every row
@pytest.mark.parametrize("dtype,heads,hidden,p_len", [("fp32", 4, 64, 128), ("bf16", 1, 64, 128), ("fp32", 1, 64, 128),
                                                    ("bf16", 2, 128, 600),      # 600: chunked long-sequence attention backward
                                                    ("fp32", 2, 128, 600),      # the chunked f32 MFMA attention (with dropout keys)
                                                    ("fp16", 1, 64, 128), ("fp16", 2, 128, 600)])   # the benchmarked engine (apex-O1 form)
def test_step_with_dropout_matches_oracle(dev, dtype, heads, hidden, p_len, full_last_layer):
    from simxns_amd import ops
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import BiBertEncoder, HFBertEncoder
    max_pos = max(160, p_len + 8)
    ocfg = BertCfg(vocab=500, hidden=hidden, layers=2, heads=heads, inter=128, max_pos=max_pos)
    cfg = BertConfigLite(vocab_size=500, hidden_size=hidden, num_hidden_layers=2, num_attention_heads=heads, intermediate_size=128,
                         max_position_embeddings=max_pos, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    Pq, Pc = make_bert_params(ocfg, 5, std=0.08), make_bert_params(ocfg, 6, std=0.08)
    B, N = 3, 3
    q_ids, q_mask, _ = make_batch(31, B, 32, 500, 9, 3, 4)
    c_ids, c_mask, _ = make_batch(32, B * (1 + N), p_len, 500, 0.45 * p_len, 0.2 * p_len, 16)
    z = np.linspace(-1.5, 1.5, B * (1 + N)).reshape(B, 1 + N).astype(np.float32)
    bi = BiBertEncoder.__new__(BiBertEncoder)
    torch.nn.Module.__init__(bi)
    bi.question_model, bi.ctx_model = HFBertEncoder(cfg, dtype), HFBertEncoder(cfg, dtype)
    bi.question_model.load_numpy_state(Pq)
    bi.ctx_model.load_numpy_state(Pc)
    bi.to(dev).train()
    _fixed_seed(bi.question_model, 1111, full_last_layer)
    _fixed_seed(bi.ctx_model, 2222, full_last_layer)
    t = lambda a: torch.from_numpy(a).to(dev)
    q, c = bi(t(q_ids), t(q_mask), t(c_ids), t(c_mask))
    loss, _, _ = ops.kl_distill_loss(q, c, t(z))
    loss.backward()
    dq_ = dict(p_hidden=0.1, p_attn=0.1, seed=1111)
    dc_ = dict(p_hidden=0.1, p_attn=0.1, seed=2222)
    _, oq, cq = ob.bert_forward(Pq, q_ids, q_mask, heads, drop=dq_)
    _, oc, cc = ob.bert_forward(Pc, c_ids, c_mask, heads, drop=dc_)
    # dropout really is on: the no-dropout embeddings differ
    _, oq0, _ = ob.bert_forward(Pq, q_ids, q_mask, heads, keep=False)
    assert np.abs(oq0 - oq).max() > 1e-2
    osim = ol.sim_block(oq, oc)
    ol_, _, ods = ol.kl_distill(osim, z.astype(np.float64))
    # (NOTE: "matches the oracle" here is a consistency check of the mask plumbing -- the oracle restates the product's own
    # stateless hash; what is pinned to the REFERENCE's dropout semantics is the keep rate -- p realised in steps of 1/256 -- and the
    # 1/keep-rate scaling)
    tol = 2e-5 if dtype == "fp32" else 1.5e-3 if dtype == "fp16" else 8e-2       # fp16 measured on MI355X: embeddings 0.9-1.8e-3, loss 0.4-1.4e-3
    print("dropout step %s: q err %.2e c err %.2e loss err %.2e" % (dtype, np.abs(q.detach().cpu().numpy() - oq).max(),
                                                                   np.abs(c.detach().cpu().numpy() - oc).max(), abs(loss.item() - ol_)))
    assert np.abs(q.detach().cpu().numpy() - oq).max() <= tol * 4
    assert np.abs(c.detach().cpu().numpy() - oc).max() <= tol * 4
    assert abs(loss.item() - ol_) <= (1e-4 if dtype == "fp32" else 5e-3 if dtype == "fp16" else 8e-2)
    dq, dc = ol.sim_block_bwd(oq, oc, ods)
    Gq = ob.bert_backward(Pq, q_ids, q_mask, heads, cq, dq)
    Gc = ob.bert_backward(Pc, c_ids, c_mask, heads, cc, dc)
    gmax = max(np.abs(v).max() for v in Gc.values())
    for pre, Go, m in (("q", Gq, bi.question_model), ("c", Gc, bi.ctx_model)):
        own = dict(m.named_parameters())
        for k, g in Go.items():
            got = own[k].grad.cpu().numpy().astype(np.float64)
            if dtype == "fp32":
                assert np.abs(got - g).max() <= 2e-4 * np.abs(g).max() + 1e-5 * gmax, (pre, k)
            elif k.endswith("dense.weight") and "pooler" not in k:
                cos = float(got.ravel() @ g.ravel() / (np.linalg.norm(got) * np.linalg.norm(g) + 1e-30))
                assert cos > (0.999 if dtype == "fp16" else 0.97), (pre, k, cos)
    # eval() switches dropout off again
    bi.eval()
    q2, _ = bi(t(q_ids), t(q_mask), t(c_ids), t(c_mask))
    assert np.abs(q2.detach().cpu().numpy() - oq0).max() <= tol * 4


def test_large_gemm_epilogue_dropout(dev):
    """the 256x256 LDS-epilogue kernel: drop(acc + bias) + residual, keep rate and exact mask placement."""
    from simxns_amd import _lib as L
    M, N, K = 8300, 768, 128
    rs = np.random.RandomState(0)
    A, B = (rs.randn(M, K) * 0.5).astype(np.float32), (rs.randn(N, K) * 0.5).astype(np.float32)
    bias, res = rs.randn(N).astype(np.float32), rs.randn(M, N).astype(np.float32)
    bf = lambda a: torch.from_numpy(a).to(dev).to(torch.bfloat16)
    dA, dB, dres, dbias = bf(A), bf(B), bf(res), torch.from_numpy(bias).to(dev)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    d = L.Dropout(0.1, 777, 17)
    L.call("simx_gemm_nt_ex", L.stream_ptr(), 1, M, N, K, L.ptr(dA), K, L.ptr(dB), K, L.ptr(out), N, L.ptr(dbias), L.ptr(dres), N, 0,
           None, 0, None, 0, C.byref(d))
    r = lambda x: x.to(torch.float32).cpu().numpy().astype(np.float64)
    mult = ob.drop_multipliers(0.1, 777, 17, np.arange(M), np.arange(N))
    # p = 0.1 is realised as 26/256 (one hash per four columns, 8-bit threshold); kept values carry 256/230, so E[mask] = 1
    assert abs((mult == 0).mean() - 26.0 / 256.0) < 0.002 and abs(mult.mean() - 1.0) < 0.005 and abs(mult.max() - 256.0 / 230.0) < 1e-6
    ref = (r(dA) @ r(dB).T + bias) * mult + r(dres)
    err = np.abs(r(out) - ref)
    assert (err <= 2e-2 + 1.2e-2 * np.abs(ref)).all()
