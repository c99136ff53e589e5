"""Generate job of the AR2/SimANS iteration on the MI355X engine -- the other half of every ``train_*_AR2.sh`` round
(SimANS/co_training/co_training_generate.py): embed the corpus with ``body_emb`` and the train / dev queries with
``query_emb`` (:97-121, :334-357), exhaustive inner-product search (top-200 for train, top-1000 for dev, :415-421),
MRR@10 / recall (:217-266) and the ``<mode>_ce_<step>.tsv`` hard-negative file the train job samples from (:153-193).

What changed against the reference, all of it behind the same file formats:
  * FAISS is replaced by simxns_amd.retrieval.FlatIPIndex (HIP kernels, corpus shard resident in HBM, shard merge over
    RCCL); the corpus is tokenised ONCE into an int32 table in HBM and embedded in large batches, instead of 20
    DataLoader workers re-tokenising per batch;
  * each rank embeds and indexes the contiguous corpus slice [r*N/W, (r+1)*N/W) -- no pickled embedding shards on disk.
"""
import argparse
import json
import logging
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from simxns_amd.model.models import BiBertEncoder                                     # noqa: E402
from simxns_amd.retrieval import FlatIPIndex                                          # noqa: E402
from simxns_amd.utils.MARCO_until_new import HashTokenizer, load_id_text              # noqa: E402
from simxns_amd.utils.dpr_utils import get_model_obj, load_states_from_checkpoint     # noqa: E402
from simxns_amd.utils.util import is_first_worker                                     # noqa: E402

logger = logging.getLogger("__main__")


def tokenize_table(rows, tokenizer, max_length, pair=False, pad_id=0):
    """[(id, text[, title])] -> (ids int64 [n], int32 table [n, max_length]); the TextDataset / Question_dataset
    encoders (:48-94): questions alone, passages as (title, text) pairs, padded to max_length."""
    ids = np.empty(len(rows), np.int64)
    tab = np.full((len(rows), max_length), pad_id, np.int32)
    for i, r in enumerate(rows):
        ids[i] = int(r[0])
        toks = (tokenizer.encode(r[2], text_pair=r[1], add_special_tokens=True, max_length=max_length, truncation=True)
                if pair else tokenizer.encode(r[1], add_special_tokens=True, max_length=max_length, truncation=True))
        tab[i, :len(toks)] = toks
    return ids, tab


@torch.no_grad()
def embed_table(embed_fn, table, device, batch_size=2048, pad_id=0):
    """int32 token table -> [n,H] f32 embeddings on the device (eval mode, no grad)."""
    out = []
    for lo in range(0, table.shape[0], batch_size):
        t = torch.from_numpy(table[lo:lo + batch_size].astype(np.int64)).to(device)
        out.append(embed_fn(t, (t != pad_id).long()).float())
    return torch.cat(out, 0) if out else torch.empty(0, 0, device=device)


def load_reference_from_stream(path):
    """qid -> [relevant pids] from a qrels-style file 'qid 0 pid rel' or 'qid\\tpid' (:196-214)."""
    rel = {}
    with open(path) as f:
        for line in f:
            p = line.strip().split()
            if not p:
                continue
            qid, pid = int(p[0]), int(p[2] if len(p) >= 4 else p[1])
            rel.setdefault(qid, []).append(pid)
    return rel


def compute_metrics(qids_to_relevant_passageids, qids_to_ranked_candidate_passages):
    """MRR@10 and recall@1/@50/@all exactly as :217-266 (denominator = number of queries WITH relevance labels)."""
    mrr, ranking = 0.0, []
    r1, r50, rall = set(), set(), set()
    for qid, cand in qids_to_ranked_candidate_passages.items():
        if qid not in qids_to_relevant_passageids:
            continue
        ranking.append(0)
        target = qids_to_relevant_passageids[qid]
        for i in range(0, min(10, len(cand))):
            if cand[i] in target:
                mrr += 1.0 / (i + 1)
                ranking[-1] = i + 1
                break
        for i, pid in enumerate(cand):
            if pid in target:
                rall.add(qid)
                if i < 50:
                    r50.add(qid)
                if i == 0:
                    r1.add(qid)
                break
    if len(ranking) == 0:
        raise IOError("No matching QIDs found. Are you sure you are scoring the evaluation set?")
    n = len(qids_to_relevant_passageids)
    return {"MRR @10": mrr / n, "recall@1": len(r1) / n, "recall@50": len(r50) / n, "recall@all": len(rall) / n,
            "QueriesRanked": len(qids_to_ranked_candidate_passages)}


def write_to_file(qids_to_ranked_candidate_passages, qids_to_ranked_candidate_scores, q_text, pos_qp, pos_qp_add, q_type,
                  save_path, global_step=0):
    """:153-193 -- one line per query: qid \\t question \\t 'pid score,...' positives \\t 'pid score,...' hard negatives
    (the top-200 ranked passages that are not positives; a positive that was not retrieved keeps score 0)."""
    q_text_dict = {item[0]: item[1] for item in q_text}
    out_path = os.path.join(save_path, q_type + "_ce_" + str(global_step) + ".tsv")
    with open(out_path, "w", encoding="utf-8") as f:
        for q_id, p_id_list in qids_to_ranked_candidate_passages.items():
            temp_pos = {ele: 0 for ele in pos_qp.get(q_id, []) + pos_qp_add.get(q_id, [])}
            negs = []
            for doc_id, doc_score in zip(p_id_list[:200], qids_to_ranked_candidate_scores[q_id][:200]):
                if doc_id in temp_pos:
                    temp_pos[doc_id] = doc_score
                else:
                    negs.append((str(doc_id), str(doc_score)))
            f.write("%s\t%s\t%s\t%s\n" % (str(q_id), q_text_dict[q_id],
                                          ",".join(str(d) + " " + str(s) for d, s in temp_pos.items()),
                                          ",".join(a + " " + b for a, b in negs)))
    return out_path


class RenewTools(object):
    """The reference's driver object (:269-460) reduced to what the job needs."""

    def __init__(self, passages_path, tokenizer, output_dir, passage_title_path=None, max_passage_length=128, rank=0, world=1):
        """Reads the corpus file once (ids of every row: the search returns global corpus rows), but TOKENISES only this
        rank's contiguous slice [rank*n/world, (rank+1)*n/world) -- the only rows this rank embeds."""
        self.tokenizer, self.output_dir = tokenizer, output_dir
        titles = load_id_text(passage_title_path) if passage_title_path and os.path.exists(passage_title_path) else {}
        rows = []
        with open(passages_path) as inp:
            for line in inp:
                pid, text = line.rstrip("\n").split("\t")[:2]
                rows.append((int(pid), text, titles.get(pid, titles.get(int(pid), "-"))))
        n = len(rows)
        self.passage_ids = np.fromiter((r[0] for r in rows), np.int64, n)
        self.slice = (rank * n // world, (rank + 1) * n // world)
        _, self.passage_table = tokenize_table(rows[self.slice[0]:self.slice[1]], tokenizer, max_passage_length, pair=True)

    def build_index(self, model, device, rank=0, world=1):
        """embed this rank's contiguous corpus slice and hold it in HBM; ids are global corpus rows."""
        n = len(self.passage_ids)
        lo, hi = rank * n // world, (rank + 1) * n // world
        if (lo, hi) != self.slice:
            raise ValueError("RenewTools was built for corpus rows %s, build_index asked for %s" % (self.slice, (lo, hi)))
        emb = embed_table(get_model_obj(model).body_emb, self.passage_table, device)
        index = FlatIPIndex(emb.shape[1] if emb.numel() else 768, id_base=lo)
        if emb.numel():
            index.add(emb)
        return index

    def get_question_topk(self, model, device, index, qa_path, golden_path, pos_qp, pos_qp_add=None, mode="train", step_num=0,
                          group=None):
        questions = []
        with open(qa_path, "r", encoding="utf-8") as f:
            for line in f:
                qid, text = line.rstrip("\n").split("\t")[:2]
                questions.append([int(qid), text])
        qids, qtab = tokenize_table(questions, self.tokenizer, 32)
        qemb = embed_table(get_model_obj(model).query_emb, qtab, device)
        k = 200 if mode == "train" else 1000
        D, I = index.search(qemb, k, group=group)
        D, I = D.cpu().numpy(), I.cpu().numpy()
        cand = {int(q): [int(self.passage_ids[j]) for j in I[r] if j >= 0] for r, q in enumerate(qids)}
        scores = {int(q): [float(s) for s, j in zip(D[r], I[r]) if j >= 0] for r, q in enumerate(qids)}
        result = None
        if golden_path and os.path.exists(golden_path):
            result = compute_metrics(load_reference_from_stream(golden_path), cand)
            logger.info("***** Done %s validate ***** %s", mode, result)
            with open(os.path.join(self.output_dir, mode + "_eval_result" + str(step_num) + ".json"), "w") as f:
                json.dump(result, f, indent=2)
        path = None
        if is_first_worker():
            path = write_to_file(cand, scores, questions, pos_qp, pos_qp_add or {}, mode, self.output_dir, step_num)
        return result, path


def load_pos_examples(path, q_type="train", data_dir=None):
    """-> (pos_qp, pos_qp_add), load_pos_examples (:120-152): qid -> [positive pids] from the qrels file, and for the train
    queries the ADDITIONAL positives collected by literal match, `<data_dir>/qrels.train.addition.tsv` ('qid\tpid').  They are
    merged into the positive column by write_to_file, i.e. kept OUT of the hard-negative column SimANS samples from."""
    pos_qp = load_reference_from_stream(path) if path and os.path.exists(path) else {}
    pos_qp_add = {}
    if q_type == "train":
        add = os.path.join(data_dir if data_dir is not None else os.path.dirname(path), "qrels.train.addition.tsv")
        if os.path.exists(add):
            pos_qp_add = load_reference_from_stream(add)
        else:
            logger.warning("%s not found: no literal-match positives are excluded from the mined hard negatives", add)
    return pos_qp, pos_qp_add


def main():
    """Flags of the reference's generate jobs (co_training_marco_generate.py:222-252, 327-362): --passage_path is the data
    DIRECTORY (para.txt, para.title.txt, qrels.{train,dev}.tsv inside), the model is output_dir/checkpoint-<global_step>,
    the mined file goes to --ann_dir."""
    ap = argparse.ArgumentParser()
    A = ap.add_argument
    A("--model_type", required=True); A("--model_name_or_path", default=None); A("--output_dir", required=True)
    A("--passage_path", required=True); A("--ann_dir", default="")
    A("--train_qa_path", default=None); A("--dev_qa_path", default=None)
    A("--global_step", type=int, default=0); A("--fp16", action="store_true")
    A("--local_rank", "--local-rank", dest="local_rank", type=int, default=-1)
    A("--tokenizer_name", default=""); A("--share_weight", action="store_true")
    for ignored in ("--max_seq_length", "--max_steps", "--adv_step", "--iteration_step", "--iteration_reranker_step"):
        A(ignored, type=int, default=0)
    A("--log_dir", default=None); A("--gradient_checkpointing", action="store_true")
    args = ap.parse_args()
    logging.basicConfig(level=logging.INFO)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", max(args.local_rank, 0)))
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    group = None
    if world > 1:
        dist.init_process_group(backend="nccl", device_id=device)
        group = dist.group.WORLD
    ann_dir = args.ann_dir or args.output_dir
    os.makedirs(ann_dir, exist_ok=True)
    model = BiBertEncoder(args).to(device).eval()
    ckpt = os.path.join(args.output_dir, "checkpoint-" + str(args.global_step))
    for path in (ckpt, args.model_name_or_path):
        if path and os.path.exists(path):
            get_model_obj(model).load_state_dict(load_states_from_checkpoint(path).model_dict, strict=False)
            break
    if args.tokenizer_name == "hash":
        tok = HashTokenizer(model.question_model.config.vocab_size)
    else:
        from transformers import BertTokenizer
        tok = BertTokenizer.from_pretrained(args.tokenizer_name or "bert-base-uncased", do_lower_case=True)
    data_dir = args.passage_path
    tools = RenewTools(os.path.join(data_dir, "para.txt"), tok, ann_dir, os.path.join(data_dir, "para.title.txt"), rank=rank, world=world)
    index = tools.build_index(model, device, rank, world)
    for mode, qa in (("train", args.train_qa_path), ("dev", args.dev_qa_path)):
        if qa:
            gold = os.path.join(data_dir, "qrels.%s.tsv" % mode)
            pos, pos_add = load_pos_examples(gold, mode, data_dir)
            tools.get_question_topk(model, device, index, qa, gold, pos, pos_add, mode, args.global_step, group)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
