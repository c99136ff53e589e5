"""`launch.py --fold-accumulation` (docs/history.md 6.7): the micro-batches of an optimizer step run as ONE batch.  The claim behind it:
the MS-Pas recipe's loss is a mean of per-query terms (co_training_marco_train.py:198-217: every query scores its OWN 1 + N
passages, no in-batch negatives), so  sum_k grad(loss_k / accum)  over the micro-batches equals the gradient of the folded batch.
Checked here on the engine itself, dropout off, fp32 arithmetic: two towers, KL-distillation loss, 4 queries x (1 + 3) passages as
one batch against 2 micro-batches of 2."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _towers(dev, seed):
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import HFBertEncoder
    from simxns_amd.utils import synth
    out = []
    for s in (seed, seed + 1):
        cfg = BertConfigLite(vocab_size=2000, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=512,
                             max_position_embeddings=64, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
        enc = HFBertEncoder(cfg, compute_dtype="fp32")
        enc.load_numpy_state(synth.fill_bert_state_dict([(k, tuple(p.shape)) for k, p in enc.named_parameters()], s, std=0.08))
        out.append(enc.to(dev).train())
    return out


def _grads(dev, nq, npas, batches, accum):
    from simxns_amd import ops
    rs = np.random.RandomState(3)
    B, S, H = nq, 48, 128
    qi = rs.randint(5, 2000, size=(B, 16))
    pi = rs.randint(5, 2000, size=(B, npas, S))
    plen = rs.randint(10, S + 1, size=(B, npas))
    pm = (np.arange(S)[None, None, :] < plen[:, :, None]).astype(np.int64)
    z = rs.randn(B, npas).astype(np.float32) * 2.0                      # teacher logits of every (query, passage)
    qenc, penc = _towers(dev, 11)
    losses = []
    for rows in batches:
        r = np.asarray(rows)
        q = qenc.embed(torch.from_numpy(qi[r]).to(dev), torch.ones(len(r), 16, dtype=torch.long, device=dev))
        c = penc.embed(torch.from_numpy((pi[r] * pm[r]).reshape(-1, S)).to(dev), torch.from_numpy(pm[r].reshape(-1, S)).to(dev))
        loss, _, _ = ops.kl_distill_loss(q, c, torch.from_numpy(z[r]).to(dev), 1.0, False, accum)     # c: [B * (1 + N), H]
        loss.backward()
        losses.append(float(loss.detach()))
    torch.cuda.synchronize()
    return sum(losses), qenc.engine.flat_grad.detach().cpu().numpy().copy(), penc.engine.flat_grad.detach().cpu().numpy().copy()


def test_folded_step_equals_accumulated_micro_batches():
    dev = torch.device("cuda:0")
    l1, gq1, gp1 = _grads(dev, 4, 4, [[0, 1, 2, 3]], 1)
    l2, gq2, gp2 = _grads(dev, 4, 4, [[0, 1], [2, 3]], 2)
    assert abs(l1 - l2) <= 1e-5 * max(1.0, abs(l1)), (l1, l2)
    for a, b in ((gq1, gq2), (gp1, gp2)):
        assert np.abs(a).max() > 0
        assert np.abs(a - b).max() <= 2e-5 * np.abs(a).max(), (np.abs(a - b).max(), np.abs(a).max())
