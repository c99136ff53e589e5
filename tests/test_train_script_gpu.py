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
    # (+ "loss_scaler" with --fp16: an extra top-level key, which torch's Optimizer.load_state_dict ignores)
    assert set(st.optimizer_dict) >= {"state", "param_groups"} and len(st.optimizer_dict["param_groups"]) == 2
    assert st.optimizer_dict["loss_scaler"]["applied_steps"] + st.optimizer_dict["loss_scaler"]["skipped_steps"] == 6
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


def test_gpu_sampler_matches_host_collate_and_law(dev, tmp_path):
    """--sampler gpu (Rocketqa_v2Dataset.build_device_pool / device_batch; reference: SimANS/utils/MARCO_until_new.py:165-258):
    (1) fed the host draw's own picks, the device batch equals the host collate bit for bit; (2) left to its own Philox
    draws, every pick is a distinct real candidate and the inclusion frequencies match the host sampler's law; (3) the
    train job runs on it."""
    import random
    from simxns_amd.utils.MARCO_until_new import Rocketqa_v2Dataset, HashTokenizer, simans_draw
    root = str(tmp_path / "data")
    _write_corpus(root, n_q=16, n_cand=24)
    # ragged candidate lists + several positives on some rows
    lines = open(os.path.join(root, "train_ce_0.tsv")).read().splitlines()
    rows = [l.split("\t") for l in lines]
    rows[3][3] = ",".join(rows[3][3].split(",")[:18])
    rows[5][2] = rows[5][2] + "," + "7 77.5000"
    open(os.path.join(root, "train_ce_0.tsv"), "w").write("\n".join("\t".join(r) for r in rows) + "\n")
    N = 7
    ds = Rocketqa_v2Dataset(os.path.join(root, "train_ce_0.tsv"), HashTokenizer(), num_hard_negatives=N, corpus_path=root)
    ds.build_device_pool(dev)
    collate = Rocketqa_v2Dataset.get_collate_fn(None)
    # (1) same picks -> same batch.  Replay the host __getitem__ while recording its picks.
    idx = [0, 3, 5, 9]
    feats, pos_choice, neg_choice = [], [], []
    for i in idx:
        sample = ds.data[i]
        pos_pairs = sample.pos_id.split(",")
        negs = [(int(p.split()[0]), float(p.split()[1])) for p in sample.neg_id.split(",")]
        random.seed(100 + i)
        pc = random.randrange(len(pos_pairs))
        pos_score = float(pos_pairs[pc].split()[1])
        picked = simans_draw(negs, pos_score, N, 3)
        pos_choice.append(pc)
        neg_choice.append([[p for p, _ in negs].index(pid) for pid in picked])
        # the host __getitem__ with exactly these picks
        ctx = [ds._encode_ctx(int(pos_pairs[pc].split()[0]))] + [ds._encode_ctx(n) for n in picked]
        q = ds.tokenizer.encode(sample.query_string, add_special_tokens=True, max_length=32, truncation=True)
        strip = lambda t: t[1:-1] if t[-1] == 102 else t[1:]
        ce = [q + strip(c) for c in ctx]
        feats.append((torch.LongTensor(q + [0] * (32 - len(q))), torch.LongTensor([c + [0] * (128 - len(c)) for c in ctx]),
                      torch.LongTensor([c + [0] * (160 - len(c)) for c in ce])))
    host = collate(feats)
    devb = ds.device_batch(idx, pos_choice=pos_choice, neg_choice=neg_choice)
    for k in ("student", "teacher"):
        for a, b in zip(host[k], devb[k]):
            if torch.is_tensor(a):
                assert torch.equal(a, b.cpu()), k
            else:
                assert a == b
    # (2) the device draw: distinct real candidates; inclusion frequencies vs the host sampler (same law, different RNG)
    q = 9
    negs = [(int(p.split()[0]), float(p.split()[1])) for p in ds.data[q].neg_id.split(",")]
    pos_score = float(ds.data[q].pos_id.split(",")[0].split()[1])
    C = len(negs)
    trials = 3000
    cnt_dev = np.zeros(C)
    cmax = ds.pool["cand_rows"].shape[1]
    for t in range(0, trials, 100):
        b = ds.device_batch([q] * 100, seed=5, step=t, pos_choice=[0] * 100)
        tab = b["picks"]["neg_table_index"].cpu().numpy() - (cmax - C)
        assert tab.min() >= 0 and all(len(set(r)) == N for r in tab)
        np.add.at(cnt_dev, tab.ravel(), 1)
    # host side of the same law: the reference's rounds (weights, N with-replacement draws, dedupe, remove, repeat) with the
    # surplus of the last round dropped in DRAW order -- the device rule; the reference drops it in CPython set order, a
    # pid-hash-dependent subset (the documented deviation, SURVEY App. B / DESIGN.md section 8), so its frequencies are not comparable
    import math
    cnt_host = np.zeros(C)
    rng = random.Random(1)
    w0 = [math.exp(-abs(s_ - pos_score) * 3) for _, s_ in negs]
    for _ in range(trials):
        cand, w, chosen = list(range(C)), list(w0), []
        while len(chosen) < N:
            for c in rng.choices(cand, weights=w, k=N):
                if c not in chosen:
                    chosen.append(c)
            keep = [(c, wi) for c, wi in zip(cand, w) if c not in chosen]
            cand, w = [c for c, _ in keep], [wi for _, wi in keep]
        for c in chosen[:N]:
            cnt_host[c] += 1
    fd, fh = cnt_dev / trials, cnt_host / trials
    assert np.abs(fd - fh).max() <= 0.05, (fd, fh)
    # (3) the job itself
    from simxns_amd.co_training import co_training_marco_train as T
    gs = T.main(["--model_type", os.path.join(root, "student"), "--teacher_model_type", os.path.join(root, "teacher"),
                 "--tokenizer_name", "hash", "--per_gpu_train_batch_size", "4", "--number_neg", "7", "--learning_rate", "1e-3",
                 "--teacher_learning_rate", "1e-4", "--output_dir", str(tmp_path / "ckpt"), "--log_dir", str(tmp_path / "tb"),
                 "--origin_data_dir", os.path.join(root, "train_ce_0.tsv"), "--passage_path", root, "--logging_steps", "2",
                 "--save_steps", "1000", "--max_steps", "6", "--iteration_step", "6", "--iteration_reranker_step", "2",
                 "--temperature_distill", "1", "--ann_dir", root, "--num_workers", "0", "--sampler", "gpu", "--global_step", "0"])
    assert gs == 6


def test_wiki_generate_job(dev, tmp_path):
    """NQ / TQ generate job end to end (wiki/co_training_wiki_generate.py): passages tsv + qa csv -> top-k by the HIP index ==
    the oracle's exhaustive search over the same embeddings, hit flags == has_answer, the three files of the round, and the
    next train job's dataset reads train_ce_<step>.json."""
    from oracle import retrieval as orr
    from simxns_amd.co_training.co_training_generate import embed_table
    from simxns_amd.utils.dpr_utils import SimpleTokenizer, has_answer, load_states_from_checkpoint, save_checkpoint_state  # noqa: F401
    from simxns_amd.utils.util_wiki import TraditionDataset
    from simxns_amd.utils.MARCO_until_new import HashTokenizer
    from simxns_amd.wiki import co_training_wiki_generate as G
    root = str(tmp_path / "data")
    _write_corpus(root)
    rs = np.random.RandomState(3)
    words = ["w%d" % i for i in range(300)]
    n_pass, n_q = 300, 12
    texts = [" ".join(rs.choice(words, size=rs.randint(8, 40))) for _ in range(n_pass)]
    with open(os.path.join(root, "psgs.tsv"), "w") as f:
        f.write("id\ttext\ttitle\n")
        for i, t in enumerate(texts):
            f.write("%d\t%s\tt%d\n" % (i + 1, t, i))
    questions = ["what about %s?" % " ".join(rs.choice(words, size=4)) for _ in range(n_q)]
    answers = [[texts[rs.randint(n_pass)].split()[2], "zzz-never"] for _ in range(n_q)]
    for mode in ("train", "dev", "test"):
        with open(os.path.join(root, "%s.qa.csv" % mode), "w") as f:
            for q, a in zip(questions, answers):
                f.write("%s\t%s\n" % (q, repr(a)))
    gold = [dict(question=q, answers=a, positive_ctxs=[dict(title="t%d" % i, text=texts[i], passage_id=str(i + 1), score="1")],
                 hard_negative_ctxs=[]) for i, (q, a) in enumerate(zip(questions, answers))]
    for name in ("train_ce_0.json", "dev_ce_0.json"):
        json.dump(gold, open(os.path.join(root, name), "w"))
    ann = str(tmp_path / "ann")
    out = str(tmp_path / "ckpt")
    argv = ["--model_type", os.path.join(root, "student"), "--tokenizer_name", "hash", "--max_seq_length", "64", "--output_dir", out,
            "--origin_data_dir", os.path.join(root, "train_ce_0.json"), "--origin_data_dir_dev", os.path.join(root, "dev_ce_0.json"),
            "--train_qa_path", os.path.join(root, "train.qa.csv"), "--dev_qa_path", os.path.join(root, "dev.qa.csv"),
            "--test_qa_path", os.path.join(root, "test.qa.csv"), "--passage_path", os.path.join(root, "psgs.tsv"), "--ann_dir", ann,
            "--global_step", "0", "--max_steps", "10", "--fp16"]
    # one set of weights for the job and for the check below: the job loads --model_name_or_path
    from simxns_amd.utils.dpr_utils import CheckpointState
    args = G.M.get_arguments(argv)
    args.device = dev
    torch.manual_seed(5)
    tok, model = G.load_model(args)
    os.makedirs(out, exist_ok=True)
    torch.save(CheckpointState(model.state_dict(), {}, {}, 0, 0, None)._asdict(), os.path.join(out, "init.pkl"))
    argv += ["--model_name_or_path", os.path.join(out, "init.pkl")]
    assert G.main(argv) == 0
    for name in ("train_result_dict_list_0.json", "train_eval_result0.json", "train_ce_0.json", "dev_ce_0.json",
                 "test_result_dict_list_0.json", "test_eval_result0.json"):
        assert os.path.exists(os.path.join(ann, name)), name
    assert not os.path.exists(os.path.join(ann, "test_ce_0.json"))
    rd = json.load(open(os.path.join(ann, "train_result_dict_list_0.json")))
    ev = json.load(open(os.path.join(ann, "train_eval_result0.json")))
    assert len(rd) == n_q and all(len(r["ctxs"]) == 100 for r in rd) and set(ev) == {"top1", "top5", "top20", "top100", "result_dict"}
    # ranking == exhaustive search over the same embeddings; hits == has_answer on the passage text
    model.eval()
    tools = G.RenewTools(os.path.join(root, "psgs.tsv"), tok, str(tmp_path / "ann2"), max_seq_length=64)
    _, _, qemb = tools.get_question_embedding(model, dev, os.path.join(root, "train.qa.csv"))
    pemb = embed_table(model.body_emb, tools.passage_table, dev)
    _, ref_ids = orr.search(qemb.cpu().numpy(), pemb.cpu().numpy(), 100)
    st = SimpleTokenizer()
    first_hit = []
    for r, res in enumerate(rd):
        assert [int(c["d_id"]) for c in res["ctxs"]] == [int(i) for i in ref_ids[r]]
        flags = [has_answer(answers[r], texts[int(c["d_id"])], st) for c in res["ctxs"]]
        assert [c["hit"] for c in res["ctxs"]] == [str(x) for x in flags]
        first_hit.append(next((i for i, x in enumerate(flags) if x), None))
    assert abs(ev["top1"] - sum(1 for h in first_hit if h == 0) / n_q) < 1e-12
    assert abs(ev["top100"] - sum(1 for h in first_hit if h is not None) / n_q) < 1e-12
    ce = json.load(open(os.path.join(ann, "train_ce_0.json")))
    for r, e in enumerate(ce):
        assert e["positive_ctxs"][0]["passage_id"] == str(r + 1) and e["question"] == questions[r]
        got = {c["passage_id"] for c in e["positive_ctxs"][1:]} | {c["passage_id"] for c in e["hard_negative_ctxs"]}
        assert got == {c["d_id"] for c in rd[r]["ctxs"] if not (c["hit"] == "True" and int(c["d_id"]) == r)}   # (the retrieved gold is positive 0)
    ds = TraditionDataset(os.path.join(ann, "train_ce_0.json"), HashTokenizer(), num_hard_negatives=7, a=0.5, b=1.0, max_seq_length=64)
    assert len(ds) == n_q and len(ds[0][1]) == 8


def test_ms_doc_generate_job(dev, tmp_path):
    """MS-MARCO Document generate job end to end (Doc_training/co_training_doc_generate.py): the checkpoint of a train round,
    D-prefixed corpus / qrels, top-200 == the oracle's search, only queries whose positive was retrieved are written, and
    Doc_v2Dataset reads the file."""
    from oracle import retrieval as orr
    from simxns_amd.Doc_training import co_training_doc_generate as G
    from simxns_amd.Doc_training import co_training_doc_train as D
    from simxns_amd.co_training.co_training_generate import embed_table
    from simxns_amd.utils.MARCO_until_Doc import Doc_v2Dataset
    from simxns_amd.utils.dpr_utils import CheckpointState
    root = str(tmp_path / "doc")
    os.makedirs(root)
    rs = np.random.RandomState(4)
    words = ["w%d" % i for i in range(400)]
    n_doc, n_q = 260, 10
    with open(os.path.join(root, "msmarco-docs.tsv"), "w") as f:
        for pid in range(n_doc):
            f.write("D%d\thttp://u/%d\t%s\t%s\n" % (pid, pid, " ".join(rs.choice(words, size=4)), " ".join(rs.choice(words, size=rs.randint(30, 300)))))
    with open(os.path.join(root, "msmarco-doctrain-queries.tsv"), "w") as f, open(os.path.join(root, "msmarco-doctrain-qrels.tsv"), "w") as g:
        for q in range(n_q):
            f.write("%d\t%s\n" % (100 + q, " ".join(rs.choice(words, size=7))))
            g.write("%d 0 D%d 1\n" % (100 + q, rs.randint(n_doc)))
    d = os.path.join(root, "student")
    os.makedirs(d)
    json.dump(dict(vocab_size=50265, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128,
                   max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, model_type="roberta", pad_token_id=1),
              open(os.path.join(d, "config.json"), "w"))
    out, ann = str(tmp_path / "ckpt"), str(tmp_path / "ann")
    os.makedirs(out)
    argv = ["--model_type", d, "--tokenizer_name", "hash", "--max_seq_length", "512", "--output_dir", out,
            "--train_qa_path", os.path.join(root, "msmarco-doctrain-queries.tsv"), "--passage_path", root, "--ann_dir", ann,
            "--global_step", "6", "--max_steps", "12", "--fp16"]
    args = G.M.get_arguments(argv)
    args.teacher_model_type = d
    tok, model, _ = D.load_model(args)
    torch.save(CheckpointState(model.state_dict(), {}, {}, 0, 0, None)._asdict(), os.path.join(out, "checkpoint-6"))
    res = G.main(argv)
    result, path = res["train"]
    assert os.path.basename(path) == "train_ce_6.tsv" and result["QueriesRanked"] == n_q and os.path.exists(os.path.join(ann, "train_eval_result6.json"))
    model = model.to(dev).eval()
    tools = G.RenewTools(os.path.join(root, "msmarco-docs.tsv"), tok, str(tmp_path / "ann2"))
    _, qids, qemb = tools.get_question_embedding(model, dev, os.path.join(root, "msmarco-doctrain-queries.tsv"))
    pemb = embed_table(model.body_emb, tools.passage_table, dev, batch_size=256, pad_id=1)
    _, ref_ids = orr.search(qemb.cpu().numpy(), pemb.cpu().numpy(), 200)
    qrels = {int(l.split()[0]): int(l.split()[2][1:]) for l in open(os.path.join(root, "msmarco-doctrain-qrels.tsv"))}
    want = {}
    for r, q in enumerate(qids):
        ids = [int(i) for i in ref_ids[r]]
        if qrels[int(q)] in ids:
            want[int(q)] = [i for i in ids if i != qrels[int(q)]]
    lines = [l.rstrip("\n").split("\t") for l in open(path)]
    assert {int(f[0]) for f in lines} == set(want) and len(lines) >= 1
    for f in lines:
        assert [int(p.split(" ")[0]) for p in f[3].split(",")] == want[int(f[0])] and int(f[2].split(" ")[0]) == qrels[int(f[0])]
        assert float(f[2].split(" ")[1]) != 0.0
    ds = Doc_v2Dataset(path, tok, num_hard_negatives=3, corpus_path=root)
    q, ctx, ce = ds[0]
    assert tuple(q.shape) == (128,) and tuple(ctx.shape) == (4, 512)


def test_wiki_iteration_train_generate_train(dev, tmp_path):
    """One full round of train_NQ_AR2.sh on synthetic data: train job to the first iteration boundary (checkpoint-6), generate job
    at that step (reads the checkpoint, mines train_ce_6.json into --ann_dir), train job resumed from step 6 on the mined file."""
    from simxns_amd.utils.dpr_utils import load_states_from_checkpoint
    from simxns_amd.wiki import co_training_wiki_generate as G
    from simxns_amd.wiki import co_training_wiki_train as W
    root = str(tmp_path / "data")
    _write_corpus(root)
    rs = np.random.RandomState(5)
    words = ["w%d" % i for i in range(300)]
    n_pass, n_q = 240, 16
    texts = [" ".join(rs.choice(words, size=rs.randint(8, 40))) for _ in range(n_pass)]
    with open(os.path.join(root, "psgs.tsv"), "w") as f:
        f.write("id\ttext\ttitle\n")
        for i, t in enumerate(texts):
            f.write("%d\t%s\tt%d\n" % (i + 1, t, i))
    questions = ["what about %s?" % " ".join(rs.choice(words, size=4)) for _ in range(n_q)]
    answers = [[texts[i].split()[1]] for i in range(n_q)]                  # the gold passage of question i contains its answer
    for mode in ("train", "dev", "test"):
        with open(os.path.join(root, "%s.qa.csv" % mode), "w") as f:
            for q, a in zip(questions, answers):
                f.write("%s\t%s\n" % (q, repr(a)))
    mk = lambda pid, s_: dict(text=texts[pid], title="t%d" % pid, score=str(s_), passage_id=str(pid + 1))
    gold = [dict(question=q, answers=a, positive_ctxs=[mk(i, 80.0)],
                 hard_negative_ctxs=[mk(int(j), 79.0 - k) for k, j in enumerate(rs.choice(n_pass, size=20, replace=False)) if int(j) != i])
            for i, (q, a) in enumerate(zip(questions, answers))]
    for name in ("train_ce_0.json", "dev_ce_0.json"):
        json.dump(gold, open(os.path.join(root, name), "w"))
    out, ann = str(tmp_path / "ckpt"), str(tmp_path / "ckpt" / "temp")
    common = ["--model_type", os.path.join(root, "student"), "--tokenizer_name", "hash", "--max_seq_length", "64", "--output_dir", out,
              "--origin_data_dir", os.path.join(root, "train_ce_0.json"), "--ann_dir", ann, "--max_steps", "12", "--fp16"]
    train = common + ["--reranker_model_type", os.path.join(root, "teacher"), "--per_gpu_train_batch_size", "4", "--number_neg", "7",
                      "--learning_rate", "1e-3", "--reranker_learning_rate", "1e-4", "--log_dir", str(tmp_path / "tb"), "--logging_steps", "2",
                      "--iteration_step", "6", "--iteration_reranker_step", "2", "--temperature_normal", "1", "--adv_lambda", "0", "--b", "1.0",
                      "--num_workers", "0"]
    assert W.main(train + ["--global_step", "0"]) == 6
    w6 = load_states_from_checkpoint(os.path.join(out, "checkpoint-6")).model_dict
    gen = common + ["--origin_data_dir_dev", os.path.join(root, "dev_ce_0.json"), "--train_qa_path", os.path.join(root, "train.qa.csv"),
                    "--dev_qa_path", os.path.join(root, "dev.qa.csv"), "--test_qa_path", os.path.join(root, "test.qa.csv"),
                    "--passage_path", os.path.join(root, "psgs.tsv"), "--global_step", "6"]
    assert G.main(gen) == 6
    mined = json.load(open(os.path.join(ann, "train_ce_6.json")))
    assert len(mined) == n_q and all(e["positive_ctxs"][0]["passage_id"] == str(i + 1) for i, e in enumerate(mined))
    assert all(len(e["hard_negative_ctxs"]) >= 7 for e in mined)
    assert W.main(train + ["--global_step", "6"]) == 12
    w12 = load_states_from_checkpoint(os.path.join(out, "checkpoint-12")).model_dict
    k = "ctx_model.encoder.layer.0.output.dense.weight"
    assert torch.isfinite(w12[k]).all() and not torch.equal(w6[k], w12[k])


def test_ms_pas_iteration_train_generate_train_gpu_sampler(dev, tmp_path):
    """One full round of train_MS_Pas_AR2.sh on synthetic data with the recipe's sampler (--sampler gpu): train to the iteration
    boundary, generate at that step (checkpoint-6 -> train_ce_6.tsv in --ann_dir), train resumed on the mined file."""
    from simxns_amd.co_training import co_training_generate as Gn
    from simxns_amd.co_training import co_training_marco_train as T
    root = str(tmp_path / "data")
    _write_corpus(root, n_pass=500, n_q=24)
    rs = np.random.RandomState(6)
    with open(os.path.join(root, "train.query.txt"), "w") as f, open(os.path.join(root, "qrels.train.tsv"), "w") as g:
        for q, line in enumerate(open(os.path.join(root, "train_ce_0.tsv"))):
            fld = line.rstrip("\n").split("\t")
            f.write("%s\t%s\n" % (fld[0], fld[1]))
            g.write("%s 0 %s 1\n" % (fld[0], fld[2].split(" ")[0]))
    out, ann = str(tmp_path / "ckpt"), str(tmp_path / "ckpt" / "temp")
    os.makedirs(ann)
    common = ["--model_type", os.path.join(root, "student"), "--tokenizer_name", "hash", "--output_dir", out, "--passage_path", root,
              "--ann_dir", ann, "--max_steps", "12", "--fp16", "--train_qa_path", os.path.join(root, "train.query.txt")]
    train = common + ["--teacher_model_type", os.path.join(root, "teacher"), "--per_gpu_train_batch_size", "4", "--number_neg", "7",
                      "--learning_rate", "1e-3", "--teacher_learning_rate", "1e-4", "--log_dir", str(tmp_path / "tb"),
                      "--origin_data_dir", os.path.join(root, "train_ce_0.tsv"), "--logging_steps", "2", "--save_steps", "1000",
                      "--iteration_step", "6", "--iteration_reranker_step", "2", "--temperature_distill", "1", "--num_workers", "0",
                      "--sampler", "gpu"]
    assert T.main(train + ["--global_step", "0"]) == 6
    import sys
    argv0 = sys.argv
    try:
        sys.argv = ["co_training_generate.py"] + common + ["--global_step", "6"]
        Gn.main()
    finally:
        sys.argv = argv0
    mined = os.path.join(ann, "train_ce_6.tsv")
    lines = open(mined).read().splitlines()
    assert len(lines) == 24 and all(len(l.split("\t")) == 4 and len(l.split("\t")[3].split(",")) >= 100 for l in lines)
    assert T.main(train + ["--global_step", "6"]) == 12 and os.path.exists(os.path.join(out, "checkpoint-12"))


def test_ms_doc_iteration_train_generate_train(dev, tmp_path):
    """One full round of train_MS_Doc_AR2.sh on synthetic data: RobertaDot train job to the iteration boundary, generate job at that
    step (D-prefixed corpus / qrels beside the query file), train job resumed on the mined train_ce_6.tsv."""
    from simxns_amd.Doc_training import co_training_doc_generate as G
    from simxns_amd.Doc_training import co_training_doc_train as D
    root = str(tmp_path / "doc")
    os.makedirs(root)
    rs = np.random.RandomState(8)
    words = ["w%d" % i for i in range(400)]
    n_doc, n_q = 230, 16
    with open(os.path.join(root, "msmarco-docs.tsv"), "w") as f:
        for pid in range(n_doc):
            f.write("D%d\thttp://u/%d\t%s\t%s\n" % (pid, pid, " ".join(rs.choice(words, size=4)), " ".join(rs.choice(words, size=rs.randint(30, 200)))))
    with open(os.path.join(root, "train_ce_0.tsv"), "w") as f, open(os.path.join(root, "msmarco-doctrain-queries.tsv"), "w") as qf, \
            open(os.path.join(root, "msmarco-doctrain-qrels.tsv"), "w") as g:
        for q in range(n_q):
            pids = rs.choice(n_doc, size=21, replace=False)
            sp = 70 + 20 * rs.rand()
            sc = np.sort(sp - np.abs(rs.randn(20)) * 1.5)[::-1]
            text = " ".join(rs.choice(words, size=7))
            f.write("%d\t%s\t%d %.4f\t%s\n" % (q, text, pids[0], sp, ",".join("%d %.4f" % (p, s) for p, s in zip(pids[1:], sc))))
            qf.write("%d\t%s\n" % (q, text))
            g.write("%d 0 D%d 1\n" % (q, pids[0]))
    for name in ("student", "teacher"):
        d = os.path.join(root, name)
        os.makedirs(d)
        json.dump(dict(vocab_size=50265, hidden_size=64, num_hidden_layers=2, num_attention_heads=1, intermediate_size=128,
                       max_position_embeddings=514, type_vocab_size=1, layer_norm_eps=1e-5, model_type="roberta", pad_token_id=1),
                  open(os.path.join(d, "config.json"), "w"))
    out, ann = str(tmp_path / "ckpt"), str(tmp_path / "ckpt" / "temp")
    common = ["--model_type", os.path.join(root, "student"), "--tokenizer_name", "hash", "--output_dir", out, "--passage_path", root,
              "--ann_dir", ann, "--max_steps", "12", "--fp16", "--train_qa_path", os.path.join(root, "msmarco-doctrain-queries.tsv")]
    train = common + ["--teacher_model_type", os.path.join(root, "teacher"), "--per_gpu_train_batch_size", "2", "--number_neg", "3",
                      "--learning_rate", "1e-3", "--teacher_learning_rate", "1e-4", "--log_dir", str(tmp_path / "tb"),
                      "--origin_data_dir", os.path.join(root, "train_ce_0.tsv"), "--logging_steps", "2", "--save_steps", "1000",
                      "--iteration_step", "6", "--iteration_reranker_step", "2", "--temperature_distill", "1", "--num_workers", "0",
                      "--a", "0.5", "--b", "0"]
    assert D.main(train + ["--global_step", "0"]) == 6
    res = G.main(common + ["--global_step", "6", "--max_seq_length", "512"])
    result, path = res["train"]
    assert path == os.path.join(ann, "train_ce_6.tsv") and result["QueriesRanked"] == n_q
    kept = open(path).read().splitlines()
    assert 2 <= len(kept) <= n_q
    assert D.main(train + ["--global_step", "6"]) == 12 and os.path.exists(os.path.join(out, "checkpoint-12"))


def test_ms_pas_recipe_defaults_job_and_resume_equals_uninterrupted(dev, tmp_path, caplog, monkeypatch, capsys):
    """The job a drop-in user of train_MS_Pas_AR2.sh gets: `python -m simxns_amd.launch MS_Pas --dry-run` prints it, and its flags
    of record -- fp32 (no --fp16), --gradient_checkpointing, --sampler=gpu, accumulation 2, --distill_loss
    (SimANS/train_MS_Pas_AR2.sh:8-26) -- are run here through co_training_marco_train.py on a synthetic shard (paths, model size
    and step counts are the only substitutions) across ONE iteration boundary:
      * relaunched: job A stops at the boundary (checkpoint-4 / checkpoint-reranker4), job B resumes with --global_step 4 from the
        checkpoint files and runs the next iteration (one student step, two reranker steps, one student step);
      * uninterrupted: the same two iterations in one process on the live model / optimiser / scheduler objects, no save / load
        round trip (SIMX_CONTINUE_AT_BOUNDARY=1).
    The losses the two log for steps 5..8 must agree (f32 atomics' summation order aside) -- i.e. the checkpoint carries
    everything a resumed job needs: weights, Adam moments, both schedulers, the sampler's counter."""
    import logging
    import re
    from simxns_amd import launch
    from simxns_amd.co_training import co_training_marco_train as T
    launch.main(["MS_Pas", "--dry-run", "--nproc", "8", "--last-step", "0"])
    line = [l for l in capsys.readouterr().out.splitlines() if "co_training_marco_train.py" in l][0]
    assert "--gradient_checkpointing" in line and "--sampler=gpu" in line and "--gradient_accumulation_steps=2" in line
    assert "--fp16" not in line and "--distill_loss" in line and "--per_gpu_train_batch_size=16" in line
    flags = dict(launch.RECIPES["MS_Pas"]()["train"][1])
    root = str(tmp_path / "data")
    _write_corpus(root)
    import shutil
    shutil.copy(os.path.join(root, "train_ce_0.tsv"), os.path.join(root, "train_ce_4.tsv"))     # (the mined file of the next iteration)

    def argv(out, global_step):
        f = dict(flags, model_type=os.path.join(root, "student"), teacher_model_type=os.path.join(root, "teacher"), model_name_or_path="",
                 teacher_model_path="", per_gpu_train_batch_size=4, number_neg=7, learning_rate=1e-3, teacher_learning_rate=1e-4,
                 output_dir=out, log_dir=str(tmp_path / "tb"), origin_data_dir=os.path.join(root, "train_ce_0.tsv"),
                 passage_path=root, ann_dir=root, logging_steps=1, save_steps=1000, max_steps=8, iteration_step=4,
                 iteration_reranker_step=2, global_step=global_step, train_qa_path="", dev_qa_path="")
        assert f["gradient_checkpointing"] is True and f["sampler"] == "gpu" and f["gradient_accumulation_steps"] == 2 and "fp16" not in f
        a = ["--tokenizer_name", "hash", "--num_workers", "0"]
        for k, v in f.items():
            if v is True:
                a.append("--" + k)
            elif v not in (False, None, ""):
                a += ["--" + k, str(v)]
        return a

    def losses(records):
        out = {}
        for r in records:
            m = re.match(r"^\{.*\"step\": (\d+)\}$", r.getMessage())
            if m:
                out[int(m.group(1))] = json.loads(r.getMessage())["loss"]
        return out

    caplog.set_level(logging.INFO)
    monkeypatch.delenv("SIMX_CONTINUE_AT_BOUNDARY", raising=False)
    outA = str(tmp_path / "relaunched")
    assert T.main(argv(outA, 0)) == 4 and os.path.exists(os.path.join(outA, "checkpoint-4"))
    assert T.main(argv(outA, 4)) == 8 and os.path.exists(os.path.join(outA, "checkpoint-reranker8"))
    rel = losses(caplog.records)
    caplog.clear()
    monkeypatch.setenv("SIMX_CONTINUE_AT_BOUNDARY", "1")
    outB = str(tmp_path / "uninterrupted")
    assert T.main(argv(outB, 0)) == 8
    unint = losses(caplog.records)
    assert sorted(rel) == sorted(unint) == list(range(1, 9)), (sorted(rel), sorted(unint))
    for st in range(1, 9):
        assert np.isfinite(rel[st]) and abs(rel[st] - unint[st]) <= 2e-5 * max(1.0, abs(unint[st])), (st, rel[st], unint[st])
    assert len({round(v, 6) for v in rel.values()}) > 4                     # (the losses do move: not a degenerate comparison)
