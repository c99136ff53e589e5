"""E4 RobertaDot (SimANS/model/models.py:277-359): shared RoBERTa encoder (pad id 1, positions from 2, no pooler) +
Linear + LayerNorm head, against the golden written by the IMPORTED reference class (tests/golden/roberta_dot_tiny.npz).
fp32 parity mode: embeddings / loss within 1e-3 (measured ~1e-6), every gradient within 2e-4 of its scale."""
import json
import os

import numpy as np
import pytest
import torch

from oracle.weights import BertCfg, make_bert_params

pytestmark = pytest.mark.gpu


def _build(G, dev, dtype):
    from simxns_amd.engine import BertConfigLite
    from simxns_amd.model.models import RobertaDot
    c = json.loads(str(G["cfg"]))
    cfg = BertConfigLite(vocab_size=c["vocab"], hidden_size=c["hidden"], num_hidden_layers=c["layers"],
                         num_attention_heads=c["heads"], intermediate_size=c["inter"], max_position_embeddings=c["max_pos"],
                         type_vocab_size=c["type_vocab"], layer_norm_eps=c["eps"], hidden_dropout_prob=0.0,
                         attention_probs_dropout_prob=0.0, model_type="roberta", pad_token_id=1)
    import types
    use_mean = bool(int(G["use_mean"])) if "use_mean" in G.files else False
    m = RobertaDot(cfg, types.SimpleNamespace(use_mean=use_mean), compute_dtype=dtype) if use_mean else RobertaDot(cfg, compute_dtype=dtype)
    assert m.use_mean == use_mean
    P = make_bert_params(BertCfg(**c), int(G["seed"]), std=0.08)
    m.roberta.load_numpy_state(P)
    with torch.no_grad():
        m.embeddingHead.weight.copy_(torch.from_numpy(G["param.embeddingHead.weight"]))
        m.embeddingHead.bias.copy_(torch.from_numpy(G["param.embeddingHead.bias"]))
        m.norm.weight.copy_(torch.from_numpy(G["param.norm.weight"]))
        m.norm.bias.copy_(torch.from_numpy(G["param.norm.bias"]))
    return m.to(dev)


def _step(G, dev, dtype):
    from simxns_amd import ops
    m = _build(G, dev, dtype)
    assert "roberta.pooler.dense.weight" not in dict(m.named_parameters())       # add_pooling_layer=False
    t = lambda k: torch.from_numpy(G[k]).to(dev)
    m.zero_grad()
    q = m(t("q_ids"), t("q_mask"), True)
    d = m(t("d_ids"), t("d_mask"), False)
    loss, _, sim = ops.kl_distill_loss(q, d, t("teacher").float(), 1.0, False, 1)
    loss.backward()
    torch.cuda.synchronize()
    grads = {k: p.grad.detach().cpu().numpy().astype(np.float64) for k, p in m.named_parameters()}
    return q.detach().cpu().numpy(), d.detach().cpu().numpy(), loss.item(), grads


def test_roberta_dot_fp32_vs_reference_golden(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "roberta_dot_tiny.npz"))
    q, d, loss, grads = _step(G, dev, "fp32")
    assert np.abs(q - G["q_emb"]).max() <= 2e-5 and np.abs(d - G["d_emb"]).max() <= 2e-5
    assert abs(loss - float(G["loss"])) <= 5e-5
    names = [k[len("grad."):] for k in G.files if k.startswith("grad.")]
    assert len(names) == 41
    gmax = max(np.abs(G["grad." + k]).max() for k in names)
    for k in names:
        ref = G["grad." + k]
        err = np.abs(grads[k].reshape(ref.shape) - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 1e-5 * gmax, "grad %s: err %.3e (scale %.3e)" % (k, err, np.abs(ref).max())


def test_roberta_dot_bf16(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "roberta_dot_tiny.npz"))
    q, d, loss, grads = _step(G, dev, "bf16")
    assert np.abs(q - G["q_emb"]).max() <= 8e-2 and np.abs(d - G["d_emb"]).max() <= 8e-2
    assert abs(loss - float(G["loss"])) <= 5e-2
    for k in ("roberta.encoder.layer.1.output.dense.weight", "embeddingHead.weight", "roberta.embeddings.position_embeddings.weight"):
        g, ref = grads[k].ravel(), G["grad." + k].ravel()
        cos = float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
        assert cos >= 0.97, "grad %s cosine %.4f" % (k, cos)


@pytest.mark.parametrize("fixture", ["roberta_dot_tiny.npz", "roberta_dot_mean_tiny.npz"])
def test_roberta_dot_fp16(dev, golden_dir, fixture):
    """E4 on the benchmarked engine (fp16, apex-O1 form), [CLS] and masked-mean pooling.  Measured on MI355X: embeddings 6e-4..1e-3, loss
    7e-5..1.1e-4, gradient cosines >= 0.99998; bounds = 3x."""
    G = np.load(os.path.join(golden_dir, fixture))
    q, d, loss, grads = _step(G, dev, "fp16")
    eq, ed, el = np.abs(q - G["q_emb"]).max(), np.abs(d - G["d_emb"]).max(), abs(loss - float(G["loss"]))
    cos = {}
    for k in ("roberta.encoder.layer.1.output.dense.weight", "embeddingHead.weight", "roberta.embeddings.position_embeddings.weight"):
        g, ref = grads[k].ravel(), G["grad." + k].ravel()
        cos[k] = float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
    print("roberta_dot fp16 %s: q %.2e d %.2e loss %.2e cos %s" % (fixture, eq, ed, el, {k.split(".")[-2]: round(v, 6) for k, v in cos.items()}))
    assert eq <= 3e-3 and ed <= 3e-3 and el <= 4e-4
    assert min(cos.values()) >= 0.99994, cos


def test_roberta_dot_mean_pooling_fp32_vs_reference_golden(dev, golden_dir):
    """use_mean=True (EmbeddingMixin.masked_mean, models.py:296-305): the embedding is the mean of the last hidden state
    over the real tokens, so EVERY row of the hidden state carries gradient (simx_seq_mean_* + simx_bert_bwd_ex)."""
    G = np.load(os.path.join(golden_dir, "roberta_dot_mean_tiny.npz"))
    assert int(G["use_mean"]) == 1
    q, d, loss, grads = _step(G, dev, "fp32")
    assert np.abs(q - G["q_emb"]).max() <= 2e-5 and np.abs(d - G["d_emb"]).max() <= 2e-5
    assert abs(loss - float(G["loss"])) <= 5e-5
    names = [k[len("grad."):] for k in G.files if k.startswith("grad.")]
    assert len(names) >= 15
    gmax = max(np.abs(G["grad." + k]).max() for k in names)
    for k in names:
        ref = G["grad." + k]
        err = np.abs(grads[k].reshape(ref.shape) - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 1e-5 * gmax, "grad %s: err %.3e (scale %.3e)" % (k, err, np.abs(ref).max())


def test_roberta_dot_mean_pooling_bf16(dev, golden_dir):
    G = np.load(os.path.join(golden_dir, "roberta_dot_mean_tiny.npz"))
    q, d, loss, grads = _step(G, dev, "bf16")
    assert np.abs(q - G["q_emb"]).max() <= 8e-2 and np.abs(d - G["d_emb"]).max() <= 8e-2
    assert abs(loss - float(G["loss"])) <= 5e-2
    for k in ("roberta.encoder.layer.1.output.dense.weight", "embeddingHead.weight", "roberta.embeddings.position_embeddings.weight"):
        g, ref = grads[k].ravel(), G["grad." + k].ravel()
        cos = float(g @ ref / (np.linalg.norm(g) * np.linalg.norm(ref) + 1e-30))
        assert cos >= 0.97, "grad %s cosine %.4f" % (k, cos)
