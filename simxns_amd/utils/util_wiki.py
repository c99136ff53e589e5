"""TraditionDataset -- the NQ / TriviaQA training set of the AR2/SimANS step (SimANS/utils/util_wiki.py:558-701):
DPR-style JSON ({question, answers, positive_ctxs, hard_negative_ctxs[{text,title,score,passage_id}]}), SimANS Gaussian
weights exp(-(s_i - s_pos + b)^2 * a) (util_wiki.py:620-626; the code ADDS b), rounds of random.choices, selection kept
in the order of the shuffled negative list (:605,:639), tiling when fewer than N negatives (:614-616), dynamic-padding
collate (:667-699).  set_seed / is_first_worker live in utils/util.py."""
import json
import math
import random

import torch
from torch.utils.data import Dataset

from .util import is_first_worker, set_seed  # noqa: F401  (same names as util_wiki.py:198-208)


def normalize_question(question: str) -> str:
    return question[:-1] if question.endswith("?") else question


def simans_draw_gauss(cands, scores, pos_score, num_neg, a=0.5, b=0.0, rng=random):
    """util_wiki.py:620-639 -> set of chosen candidate ids (pre-truncation union)."""
    w = [math.exp(-(s - pos_score + b) ** 2 * a) for s in scores]
    cand = list(cands)
    chosen = set()
    while len(chosen) < num_neg:
        chosen = chosen.union(rng.choices(cand, weights=w, k=num_neg))
        keep = [(c, wi) for c, wi in zip(cand, w) if c not in chosen]
        cand, w = [c for c, _ in keep], [wi for _, wi in keep]
    return chosen


class TraditionDataset(Dataset):
    def __init__(self, file_path, tokenizer, num_hard_negatives=1, is_training=True, a=0.5, b=0, max_seq_length=256,
                 max_q_length=32, shuffle_positives=False):
        self.file_path, self.tokenizer = file_path, tokenizer
        with open(file_path, 'r', encoding="utf-8") as f:
            data = json.load(f)
        self.data = [r for r in data if len(r["positive_ctxs"]) > 0 and len(r['hard_negative_ctxs']) > 0]
        self.is_training, self.num_hard_negatives = is_training, num_hard_negatives
        self.max_seq_length, self.max_q_length, self.shuffle_positives = max_seq_length, max_q_length, shuffle_positives
        self.a, self.b = a, b

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        s = self.data[index]
        query = normalize_question(s["question"])
        pos = s["positive_ctxs"]
        negs = list(s.get("hard_negative_ctxs", []))
        if self.is_training:
            random.shuffle(negs)
        p = random.choice(pos) if self.shuffle_positives else pos[0]
        pos_score = float(p["score"])
        N = self.num_hard_negatives
        if len(negs) < N:
            negs = negs * N
            sel = negs[-N:]
        elif pos_score == 0:
            sel = negs[-N:]
        else:
            chosen = simans_draw_gauss([c["passage_id"] for c in negs], [float(c["score"]) for c in negs], pos_score, N,
                                       self.a, self.b)
            sel = [c for c in negs if c["passage_id"] in chosen][0:N]
        ctxs = [p] + sel
        enc = self.tokenizer.encode
        ctx_ids = [enc(c.get("title"), text_pair=c["text"].strip(), add_special_tokens=True, max_length=self.max_seq_length,
                       truncation=True) for c in ctxs]
        q_ids = enc(query)
        sep = self.tokenizer.sep_token_id
        ce = [q_ids + (c[1:-1] if c[-1] == sep else c[1:]) for c in ctx_ids]
        answers = [enc(a_, add_special_tokens=False) for a_ in s.get('answers', [])]
        return q_ids, ctx_ids, ce, answers, [[len(q_ids), len(c)] for c in ce]

    @classmethod
    def get_collate_fn(cls, args):
        def create_biencoder_input2(features):
            q_list, d_list, ce_list, pos_idx, se = [], [], [], [], []
            for f in features:
                pos_idx.append(len(d_list))
                q_list.append(f[0]); d_list.extend(f[1]); ce_list.extend(f[2]); se.append(f[4])
            pad = lambda rows: torch.LongTensor([r + [0] * (max(len(x) for x in rows) - len(r)) for r in rows])
            q, d, ce = pad(q_list), pad(d_list), pad(ce_list)
            qn, dn = q.size(0), d.size(0)
            tgt = torch.zeros(dn, dtype=torch.long)
            tgt[pos_idx] = 1
            ce = ce.reshape(qn, dn // qn, -1)
            return {'reranker': [ce, (ce != 0).long(), tgt.reshape(qn, dn // qn)],
                    'retriever': [q, (q != 0).long(), d, (d != 0).long(), pos_idx],
                    'answers': [f[3] for f in features], 'reranker_ctx_start_end': se}
        return create_biencoder_input2
