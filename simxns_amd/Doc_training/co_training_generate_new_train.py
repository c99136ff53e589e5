"""Hard-negative mining of the MS-MARCO Document AR2+SimANS iteration on the MI355X engine -- RenewTools of
SimANS/Doc_training/co_training_generate_new_train.py: documents ``D<id> \\t url \\t title \\t body`` joined as
url<sep>title<sep>body and cut at 10000 characters (:382-400), RoBERTa tokenisation to 512 (documents, :88-114) / 128
(queries, :47-70) with pad id 1, one shared RobertaDot for both sides, top-200 (train) / top-1000 (dev) exhaustive search
(:421-426), MRR@10 / recall against ``msmarco-doc<mode>-qrels.tsv`` (``qid 0 D<pid> rel``, :143-158, :220-268) and the
``<mode>_ce_<step>.tsv`` file -- which, unlike the MS-Pas job, keeps only the queries whose positive WAS retrieved with a
non-zero score (:161-196: ``if sum(temp_pos.values()) > 0``).

The compute side is the MS-Pas job's (co_training/co_training_generate.py): int32 token tables, large embedding batches,
FlatIPIndex shards in HBM merged over RCCL."""
import json
import logging
import os

import numpy as np

from ..co_training.co_training_generate import compute_metrics, embed_table, tokenize_table
from ..retrieval import FlatIPIndex
from ..utils.MARCO_until_Doc import load_docs
from ..utils.dpr_utils import get_model_obj
from ..utils.util import is_first_worker

logger = logging.getLogger("__main__")
PAD = 1                                  # RoBERTa <pad>


def load_doc_qrels(path):
    """qid -> [doc ids] from 'qid 0 D<pid> rel' lines."""
    rel = {}
    with open(path) as inp:
        for line in inp:
            p = line.split()
            if len(p) >= 3:
                rel.setdefault(int(p[0]), []).append(int(p[2][1:]))
    return rel


def write_to_file(cand, scores, q_text, pos_qp, pos_qp_add, q_type, save_path, global_step=0):
    q_text_dict = {q[0]: q[1] for q in q_text}
    out_path = os.path.join(save_path, q_type + '_ce_' + str(global_step) + '.tsv')
    kept = 0
    with open(out_path, 'w', encoding='utf-8') as f:
        for q_id, p_id_list in cand.items():
            temp_pos = {p: 0 for p in pos_qp.get(q_id, []) + pos_qp_add.get(q_id, [])}
            negs = []
            for doc_id, doc_score in zip(p_id_list[:200], scores[q_id][:200]):
                if doc_id in temp_pos:
                    temp_pos[doc_id] = doc_score
                else:
                    negs.append(str(doc_id) + ' ' + str(doc_score))
            if sum(temp_pos.values()) > 0:
                kept += 1
                f.write('%s\t%s\t%s\t%s\n' % (str(q_id), q_text_dict[q_id], ",".join(str(d) + ' ' + str(s) for d, s in temp_pos.items()),
                                              ",".join(negs)))
    logger.info("%s: %d of %d queries kept (positive retrieved in the top 200)", out_path, kept, len(cand))
    return out_path


class RenewTools(object):
    def __init__(self, passages_ctx_path, tokenizer, output_dir, temp_dir=None, max_doc_character=10000, max_doc_length=512,
                 max_query_length=128, rank=0, world=1):
        docs = load_docs(passages_ctx_path)                      # id -> url<sep>title<sep>body[:10000]
        if max_doc_character != 10000:
            docs = {k: v[:max_doc_character] for k, v in docs.items()}
        self.tokenizer, self.output_dir, self.max_query_length = tokenizer, output_dir, max_query_length
        if is_first_worker():
            os.makedirs(output_dir, exist_ok=True)
        ids = list(docs)
        n = len(ids)
        self.passage_ids = np.asarray(ids, np.int64)
        self.slice = (rank * n // world, (rank + 1) * n // world)
        rows = [(i, docs[i]) for i in ids[self.slice[0]:self.slice[1]]]
        _, self.passage_table = tokenize_table(rows, tokenizer, max_doc_length, pad_id=PAD)

    def get_new_faiss_index(self, model, device, batch_size=256):
        emb = embed_table(get_model_obj(model).body_emb, self.passage_table, device, batch_size=batch_size, pad_id=PAD)
        index = FlatIPIndex(emb.shape[1] if emb.numel() else 768, id_base=self.slice[0])
        if emb.numel():
            index.add(emb)
        return index

    def get_question_embedding(self, model, device, qa_path):
        questions = []
        with open(qa_path, "r", encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if line:
                    qid, text = line.split('\t')[:2]
                    questions.append([int(qid), text])
        qids, table = tokenize_table(questions, self.tokenizer, self.max_query_length, pad_id=PAD)
        return questions, qids, embed_table(get_model_obj(model).query_emb, table, device, pad_id=PAD)

    def get_question_topk(self, questions, qids, question_embedding, golden_path, index, mode='train', step_num=0, group=None):
        k = min(200 if mode == 'train' else 1000, len(self.passage_ids))
        D, I = index.search(question_embedding, k, group=group)
        D, I = D.cpu().numpy(), I.cpu().numpy()
        cand = {int(q): [int(self.passage_ids[j]) for j in I[r] if j >= 0] for r, q in enumerate(qids)}
        scores = {int(q): [float(s) for s, j in zip(D[r], I[r]) if j >= 0] for r, q in enumerate(qids)}
        rel = load_doc_qrels(golden_path)
        result = compute_metrics(rel, cand)
        logger.info("***** Done %s validate ***** %s", mode, result)
        path = None
        if is_first_worker():
            with open(os.path.join(self.output_dir, mode + "_eval_result" + str(step_num) + ".json"), 'w') as f:
                json.dump(result, f, indent=2)
            path = write_to_file(cand, scores, questions, rel, {}, mode, self.output_dir, step_num)
        return result, path
