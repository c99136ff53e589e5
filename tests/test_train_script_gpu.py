"""The drop-in train job (co_training_marco_train.py) end to end on a synthetic MS-MARCO-shaped corpus: tiny BERT
config, hash tokenizer, 6 optimiser steps through the retriever phase, checkpoint files in CheckpointState layout,
then a resume from that checkpoint (the shell loop's second iteration)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write_corpus(root, n_pass=400, n_q=24, n_cand=30):
    rs = np.random.RandomState(0)
    words = ["w%d" % i for i in range(300)]
    os.makedirs(root, exist_ok=True)
    with open(os.path.join(root, "para.txt"), "w") as f, open(os.path.join(root, "para.title.txt"), "w") as g:
        for pid in range(n_pass):
            f.write("%d\t%s\n" % (pid, " ".join(rs.choice(words, size=rs.randint(10, 60)))))
            g.write("%d\t%s\n" % (pid, " ".join(rs.choice(words, size=3))))
    with open(os.path.join(root, "train_ce_0.tsv"), "w") as f:
        for q in range(n_q):
            pids = rs.choice(n_pass, size=n_cand + 1, replace=False)
            sp = 70 + 20 * rs.rand()
            sc = np.sort(sp - np.abs(rs.randn(n_cand)) * 1.5)[::-1]
            f.write("%d\t%s\t%d %.4f\t%s\n" % (q, " ".join(rs.choice(words, size=6)), pids[0], sp,
                                               ",".join("%d %.4f" % (p, s) for p, s in zip(pids[1:], sc))))
    for name in ("student", "teacher"):
        d = os.path.join(root, name)
        os.makedirs(d, exist_ok=True)
        json.dump(dict(vocab_size=30522, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128,
                       max_position_embeddings=192, type_vocab_size=2, layer_norm_eps=1e-12), open(os.path.join(d, "config.json"), "w"))


def test_train_job_runs_and_resumes(dev, tmp_path):
    from simxns_amd.co_training import co_training_marco_train as T
    from simxns_amd.utils.dpr_utils import load_states_from_checkpoint
    root = str(tmp_path / "data")
    _write_corpus(root)
    out = str(tmp_path / "ckpt")
    common = ["--model_type", os.path.join(root, "student"), "--teacher_model_type", os.path.join(root, "teacher"),
              "--tokenizer_name", "hash", "--per_gpu_train_batch_size", "4", "--gradient_accumulation_steps", "1",
              "--number_neg", "7", "--learning_rate", "1e-3", "--teacher_learning_rate", "1e-4", "--output_dir", out,
              "--log_dir", str(tmp_path / "tb"), "--origin_data_dir", os.path.join(root, "train_ce_0.tsv"),
              "--passage_path", root, "--logging_steps", "2", "--save_steps", "1000", "--max_steps", "12",
              "--iteration_step", "6", "--iteration_reranker_step", "2", "--temperature_distill", "1",
              "--ann_dir", root, "--num_workers", "0", "--fp16"]
    gs = T.main(common + ["--global_step", "0"])
    assert gs == 6                                                   # breaks at the iteration boundary (:283-297)
    st = load_states_from_checkpoint(os.path.join(out, "checkpoint-6"))
    assert set(st._fields) == {"model_dict", "optimizer_dict", "scheduler_dict", "offset", "epoch", "encoder_params"}
    assert "question_model.encoder.layer.1.output.dense.weight" in st.model_dict
    tst = load_states_from_checkpoint(os.path.join(out, "checkpoint-reranker6"))
    assert "qa_classifier.weight" in tst.model_dict and "encoder.embeddings.word_embeddings.weight" in tst.model_dict
    assert all(torch.isfinite(v).all() for v in st.model_dict.values())
    # optimizer / scheduler state in the reference's on-disk formats (torch Optimizer.state_dict, LambdaLR.state_dict)
    assert set(st.optimizer_dict) == {"state", "param_groups"} and len(st.optimizer_dict["param_groups"]) == 2
    assert st.optimizer_dict["state"][0]["exp_avg"].shape == st.model_dict["question_model.embeddings.word_embeddings.weight"].shape
    assert st.scheduler_dict["last_epoch"] == 6 and "base_lrs" in st.scheduler_dict         # first iteration: 6 student steps (:289-290)
    # second shell-loop iteration: resume from checkpoint-6 with the mined file train_ce_6.tsv; goes through a teacher phase
    os.replace(os.path.join(root, "train_ce_0.tsv"), os.path.join(root, "train_ce_6.tsv"))
    gs2 = T.main(common + ["--global_step", "6"])
    assert gs2 == 12 and os.path.exists(os.path.join(out, "checkpoint-reranker12"))
    st2 = load_states_from_checkpoint(os.path.join(out, "checkpoint-12"))
    assert st2.scheduler_dict["last_epoch"] > st.scheduler_dict["last_epoch"]          # torch LambdaLR's state_dict keys
    w0, w1 = st.model_dict["ctx_model.encoder.layer.0.output.dense.weight"], st2.model_dict["ctx_model.encoder.layer.0.output.dense.weight"]
    assert not torch.equal(w0, w1)


def test_wiki_train_job(dev, tmp_path):
    """NQ/TQ variant: JSON data, Gaussian SimANS sampler (--a/--b), dynamic padding, reranker phase first."""
    from simxns_amd.wiki import co_training_wiki_train as W
    root = str(tmp_path / "data")
    _write_corpus(root)
    rs = np.random.RandomState(1)
    words = ["w%d" % i for i in range(300)]
    data = []
    for q in range(20):
        sp = 70 + 20 * rs.rand()
        sc = np.sort(sp - np.abs(rs.randn(25)) * 1.5)[::-1]
        mk = lambda pid, s_: dict(text=" ".join(rs.choice(words, size=rs.randint(8, 40))), title="t %d" % pid, score=float(s_), passage_id=int(pid))
        data.append(dict(question="what is %s?" % " ".join(rs.choice(words, size=5)), answers=["w1"],
                         positive_ctxs=[mk(1000 + q, sp)], hard_negative_ctxs=[mk(2000 + q * 30 + j, sc[j]) for j in range(25)]))
    json.dump(data, open(os.path.join(root, "train_ce_0.json"), "w"))
    out = str(tmp_path / "ckpt")
    gs = W.main(["--model_type", os.path.join(root, "student"), "--reranker_model_type", os.path.join(root, "teacher"),
                 "--tokenizer_name", "hash", "--per_gpu_train_batch_size", "4", "--number_neg", "7", "--learning_rate", "1e-3",
                 "--reranker_learning_rate", "1e-4", "--output_dir", out, "--log_dir", str(tmp_path / "tb"),
                 "--origin_data_dir", os.path.join(root, "train_ce_0.json"), "--logging_steps", "2", "--max_steps", "12",
                 "--iteration_step", "6", "--iteration_reranker_step", "2", "--temperature_normal", "1", "--adv_lambda", "0",
                 "--b", "1.0", "--ann_dir", root, "--num_workers", "0", "--fp16", "--max_seq_length", "128"])
    assert gs == 6 and os.path.exists(os.path.join(out, "checkpoint-6")) and os.path.exists(os.path.join(out, "checkpoint-reranker6"))


def test_generate_job_mines_hard_negatives_file(dev, tmp_path):
    """The generate half of the iteration: embed corpus + queries, search, metrics, train_ce_<step>.tsv that the train
    job's dataset parses; candidate ids equal the oracle's exhaustive search over the same embeddings."""
    import types
    from oracle import retrieval as orr
    from simxns_amd.co_training import co_training_generate as Gn
    from simxns_amd.model.models import BiBertEncoder
    from simxns_amd.utils.MARCO_until_new import HashTokenizer, Rocketqa_v2Dataset
    root = str(tmp_path / "data")
    _write_corpus(root, n_pass=700, n_q=24)
    rs = np.random.RandomState(1)
    with open(os.path.join(root, "train.query.txt"), "w") as f, open(os.path.join(root, "qrels.train.tsv"), "w") as g:
        for q in range(24):
            f.write("%d\t%s\n" % (q, " ".join("w%d" % w for w in rs.randint(0, 300, size=6))))
            g.write("%d 0 %d 1\n" % (q, rs.randint(0, 700)))
    args = types.SimpleNamespace(model_type=os.path.join(root, "student"), gradient_checkpointing=False, share_weight=False, fp16=True)
    model = BiBertEncoder(args).to(dev).eval()
    tok = HashTokenizer(model.question_model.config.vocab_size)
    out = str(tmp_path / "gen")
    os.makedirs(out)
    tools = Gn.RenewTools(os.path.join(root, "para.txt"), tok, out, os.path.join(root, "para.title.txt"))
    index = tools.build_index(model, dev)
    assert index.ntotal == 700
    pos, pos_add = Gn.load_pos_examples(os.path.join(root, "qrels.train.tsv"), "train", root)
    assert pos_add == {}                                   # (no qrels.train.addition.tsv in this corpus: warned, not fatal)
    result, path = tools.get_question_topk(model, dev, index, os.path.join(root, "train.query.txt"),
                                           os.path.join(root, "qrels.train.tsv"), pos, pos_add, "train", 6)
    assert os.path.basename(path) == "train_ce_6.tsv" and 0.0 <= result["MRR @10"] <= 1.0 and result["QueriesRanked"] == 24
    # ids in the file == oracle exhaustive search over the very same embeddings
    qids, qtab = Gn.tokenize_table([[q, l.split("\t")[1].strip()] for q, l in enumerate(open(os.path.join(root, "train.query.txt")))], tok, 32)
    qemb = Gn.embed_table(model.query_emb, qtab, dev).cpu().numpy()
    pemb = Gn.embed_table(model.body_emb, tools.passage_table, dev).cpu().numpy()
    _, ref_ids = orr.search(qemb, pemb, 200)
    lines = open(path).read().splitlines()
    assert len(lines) == 24
    for r, line in enumerate(lines):
        f = line.split("\t")
        posid = pos[r][0]
        want = [int(i) for i in ref_ids[r] if int(i) != posid]
        got = [int(p.split(" ")[0]) for p in f[3].split(",")]
        assert got == want and int(f[2].split(" ")[0]) == posid
    # and the train job's dataset consumes it
    ds = Rocketqa_v2Dataset(path, tok, num_hard_negatives=7, corpus_path=root)
    q, ctx, ce = ds[0]
    assert tuple(ctx.shape) == (8, 128) and tuple(ce.shape) == (8, 160)


def test_ms_doc_train_job(dev, tmp_path):
    """MS-MARCO Document job: shared RobertaDot student + RoBERTa cross-encoder teacher + Doc_v2Dataset (Gaussian SimANS
    weights, pad id 1, q128 / d512) through the retriever and teacher phases, CheckpointState files with the
    reference's key schema (roberta.* / embeddingHead.* / norm.*)."""
    from simxns_amd.Doc_training import co_training_doc_train as D
    from simxns_amd.utils.dpr_utils import load_states_from_checkpoint
    root = str(tmp_path / "doc")
    os.makedirs(root)
    rs = np.random.RandomState(2)
    words = ["w%d" % i for i in range(400)]
    with open(os.path.join(root, "msmarco-docs.tsv"), "w") as f:
        for pid in range(200):
            f.write("D%d\thttp://u/%d\t%s\t%s\n" % (pid, pid, " ".join(rs.choice(words, size=4)), " ".join(rs.choice(words, size=rs.randint(30, 400)))))
    with open(os.path.join(root, "train_ce_0.tsv"), "w") as f:
        for q in range(16):
            pids = rs.choice(200, size=21, replace=False)
            sp = 70 + 20 * rs.rand()
            sc = np.sort(sp - np.abs(rs.randn(20)) * 1.5)[::-1]
            f.write("%d\t%s\t%d %.4f\t%s\n" % (q, " ".join(rs.choice(words, size=7)), pids[0], sp,
                                               ",".join("%d %.4f" % (p, s) for p, s in zip(pids[1:], sc))))
    for name in ("student", "teacher"):
        d = os.path.join(root, name)
        os.makedirs(d)
        json.dump(dict(vocab_size=50265, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128,
                       max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, model_type="roberta", pad_token_id=1),
                  open(os.path.join(d, "config.json"), "w"))
    out = str(tmp_path / "ckpt")
    gs = D.main(["--model_type", os.path.join(root, "student"), "--teacher_model_type", os.path.join(root, "teacher"),
                 "--tokenizer_name", "hash", "--per_gpu_train_batch_size", "2", "--number_neg", "3", "--learning_rate", "1e-3",
                 "--teacher_learning_rate", "1e-4", "--output_dir", out, "--log_dir", str(tmp_path / "tb"),
                 "--origin_data_dir", os.path.join(root, "train_ce_0.tsv"), "--passage_path", root, "--logging_steps", "2",
                 "--save_steps", "1000", "--max_steps", "12", "--iteration_step", "6", "--iteration_reranker_step", "2",
                 "--temperature_distill", "1", "--ann_dir", root, "--num_workers", "0", "--fp16", "--global_step", "0",
                 "--a", "0.5", "--b", "0"])
    assert gs == 6
    st = load_states_from_checkpoint(os.path.join(out, "checkpoint-6"))
    assert "roberta.encoder.layer.1.output.dense.weight" in st.model_dict and "embeddingHead.weight" in st.model_dict
    assert "norm.bias" in st.model_dict and not any("pooler" in k for k in st.model_dict)
    assert all(torch.isfinite(v).all() for v in st.model_dict.values())
    tst = load_states_from_checkpoint(os.path.join(out, "checkpoint-reranker6"))
    assert "qa_classifier.weight" in tst.model_dict
