"""Hard-negative mining of the NQ / TriviaQA AR2+SimANS iteration on the MI355X engine -- RenewTools of
SimANS/wiki/co_training_generate_new_train_wiki.py: embed the Wikipedia passages (``psgs_w100.tsv``: id, text, title) with
``body_emb`` and the questions of a ``*.qa.csv`` file (question \\t answer list) with ``query_emb`` (:77-101, :284-309),
exhaustive inner-product top-100 (:371), answer-string validation of every retrieved passage (``has_answer``, :108-140,
:153-181) and the three files of a round (:375-400): ``<mode>_result_dict_list_<step>.json`` (the raw ranking),
``<mode>_eval_result<step>.json`` (top-1/5/20/100 accuracy + Eval_Tool metrics) and ``<mode>_ce_<step>.json`` -- the DPR-style
training file (positive_ctxs = the gold passage of the original file first, then every retrieved passage that contains an
answer; hard_negative_ctxs = the retrieved passages that do not; every entry with its retrieval score) that
TraditionDataset's SimANS sampler reads in the next train job.

Against the reference: FAISS -> simxns_amd.retrieval.FlatIPIndex (corpus shard resident in HBM, per-rank top-k merged over
RCCL); no pickled embedding shards on disk -- rank r embeds and indexes the contiguous corpus slice itself; passages and
questions are tokenised once into int32 tables and embedded in large batches."""
import ast
import csv
import json
import logging
import os

import numpy as np

from ..co_training.co_training_generate import embed_table, tokenize_table
from ..retrieval import FlatIPIndex
from ..utils.dpr_utils import Eval_Tool, SimpleTokenizer, get_model_obj, has_answer
from ..utils.util import is_first_worker

logger = logging.getLogger("__main__")
TOP_K = 100


def load_passage(passage_path):
    """psgs_w100.tsv -> [(row = id - 1, text, title)] (:317-331); the header row and malformed rows are skipped."""
    passages = []
    with open(passage_path) as fin:
        for row in csv.reader(fin, delimiter='\t'):
            if row and row[0] != 'id':
                try:
                    passages.append((int(row[0]) - 1, row[1], row[2]))
                except (ValueError, IndexError):
                    logger.warning('The following input line has not been correctly loaded: %s', row)
    return passages


def load_qa(qa_path):
    """``question \\t ['answer', ...]`` rows (:332-343) -> (questions, answers)."""
    questions, answers = [], []
    with open(qa_path, "r", encoding="utf-8") as ifile:
        for row in csv.reader(ifile, delimiter='\t'):
            questions.append(row[0])
            answers.append(list(ast.literal_eval(row[1])))
    return questions, answers


def validate(questions, passages, answers, closest_docs, similar_scores):
    """:153-181 -- per question the hit flags of its ranked passages and the result dictionary; -> (top-k accuracy list,
    hit lists, Eval_Tool metrics, result dictionaries).  ``passages``: row -> (text, title)."""
    tok = SimpleTokenizer()
    hit_lists, result_dicts = [], []
    for qi, (docs, scores) in enumerate(zip(closest_docs, similar_scores)):
        hits, ctxs = [], []
        for d, s in zip(docs, scores):
            text, title = passages[int(d)]
            hits.append(has_answer(answers[qi], text, tok))
            ctxs.append({'d_id': str(int(d)), 'text': text, 'title': title, 'score': str(s), 'hit': str(hits[-1])})
        hit_lists.append(hits)
        result_dicts.append({'id': str(qi), 'question': questions[qi], 'answers': answers[qi], 'ctxs': ctxs})
    n_docs = len(closest_docs[0]) if len(closest_docs) else 0
    top_k_hits = [0] * n_docs
    for hits in hit_lists:
        best = next((i for i, x in enumerate(hits) if x), None)
        if best is not None:
            for i in range(best, n_docs):
                top_k_hits[i] += 1
    top_k_hits = [v / max(1, len(closest_docs)) for v in top_k_hits]
    return top_k_hits, hit_lists, Eval_Tool.get_matrics(hit_lists), result_dicts


def reform_out(result_dict_list, q_pos_dict):
    """:184-226 -- ranking -> training entries.  The gold passage of the ORIGINAL training file stays positive_ctxs[0]; when
    the ranking contains it (row = its 1-based passage_id - 1) it takes that retrieval score, otherwise score 0 (SimANS then
    falls back to the last negatives, util_wiki.py:617)."""
    out = []
    for r in result_dict_list:
        pos, neg, gold_row = [], [], None
        gold = q_pos_dict.get(r["question"])
        if gold is not None:
            gold = dict(gold)
            if 'passage_id' not in gold and 'id' in gold:
                gold['passage_id'] = gold['id']
            elif 'psg_id' in gold:
                gold['passage_id'] = gold['psg_id']
            gold['score'] = str(0)
            gold_row = int(gold['passage_id']) - 1
            pos.append(gold)
        for doc in r['ctxs']:
            entry = {'title': doc['title'], 'text': doc['text'], 'passage_id': doc['d_id'], 'score': str(doc['score'])}
            if doc['hit'] == "True":
                if gold_row is not None and int(doc['d_id']) == gold_row:
                    pos[0]['score'] = str(doc['score'])
                else:
                    pos.append(entry)
            else:
                neg.append(entry)
        out.append({"q_id": str(r.get("passage_id", r["id"])), "question": r["question"], "answers": r["answers"],
                    "positive_ctxs": pos, "hard_negative_ctxs": neg, "negative_ctxs": []})
    return out


def read_train_pos(ground_truth_path):
    """question -> its first positive of a DPR-style training file (:448-459, :388-392)."""
    with open(ground_truth_path, "r", encoding="utf-8") as ifile:
        return {e['question']: e['positive_ctxs'][0] for e in json.load(ifile) if e.get('positive_ctxs')}


class RenewTools(object):
    def __init__(self, passages_path, tokenizer, output_dir, temp_dir=None, max_seq_length=128, rank=0, world=1):
        """Reads the whole passage file (texts are needed for the answer match and the output), tokenises only this rank's
        contiguous slice (title, text pairs as TextCollator :38-52)."""
        self.passages = load_passage(passages_path)
        self.tokenizer, self.output_dir, self.temp_dir = tokenizer, output_dir, temp_dir
        if is_first_worker():
            os.makedirs(output_dir, exist_ok=True)
        n = len(self.passages)
        self.rank, self.world = rank, world
        self.slice = (rank * n // world, (rank + 1) * n // world)
        rows = self.passages[self.slice[0]:self.slice[1]]
        self.passage_rows = np.fromiter((p[0] for p in self.passages), np.int64, n)
        _, self.passage_table = tokenize_table(rows, tokenizer, max_seq_length, pair=True)
        self.passage_text = {p[0]: (p[1], p[2]) for p in self.passages}

    def get_new_faiss_index(self, model, device):
        """this rank's shard of the exact inner-product index (:311-337); ids are positions in the passage file."""
        emb = embed_table(get_model_obj(model).body_emb, self.passage_table, device)
        index = FlatIPIndex(emb.shape[1] if emb.numel() else 768, id_base=self.slice[0])
        if emb.numel():
            index.add(emb)
        return index

    def get_question_embedding(self, model, device, qa_path):
        """:332-343 + Question_dataset (:55-75): typographic apostrophes replaced, no truncation below the model's limit."""
        questions, answers = load_qa(qa_path)
        rows = [(i, q.replace("’", "'")) for i, q in enumerate(questions)]
        longest = max([len(self.tokenizer.encode(r[1], add_special_tokens=True)) for r in rows] + [2])
        _, table = tokenize_table(rows, self.tokenizer, min(longest, 512))
        return questions, answers, embed_table(get_model_obj(model).query_emb, table, device)

    def get_question_topk(self, questions, answers, question_embedding, golden_path, index, mode='train', step_num=0, group=None):
        """:345-400.  Every rank takes part in the search (its shard + the merge); rank 0 validates and writes."""
        D, I = index.search(question_embedding, min(TOP_K, len(self.passages)), group=group)
        if not is_first_worker():
            return None
        D, I = D.cpu().numpy(), I.cpu().numpy()
        rows = self.passage_rows[np.clip(I, 0, None)]                       # file position -> passage row (= id - 1)
        top_k_hits, _, metrics, result_dicts = validate(questions, self.passage_text, answers, rows, D)
        with open(os.path.join(self.output_dir, mode + "_result_dict_list_" + str(step_num) + ".json"), 'w') as f:
            json.dump(result_dicts, f, indent=2)
        pick = lambda k: top_k_hits[min(k, len(top_k_hits)) - 1] if top_k_hits else 0.0
        with open(os.path.join(self.output_dir, mode + "_eval_result" + str(step_num) + ".json"), 'w') as f:
            json.dump({'top1': pick(1), 'top5': pick(5), 'top20': pick(20), 'top100': pick(100), 'result_dict': metrics}, f, indent=2)
        logger.info('%s validation: top-1 %.4f top-5 %.4f top-20 %.4f top-100 %.4f', mode, pick(1), pick(5), pick(20), pick(100))
        if mode != 'test':
            with open(os.path.join(self.output_dir, mode + '_ce_' + str(step_num) + '.json'), 'w') as f:
                json.dump(reform_out(result_dicts, read_train_pos(golden_path)), f, indent=2)
        return result_dicts
