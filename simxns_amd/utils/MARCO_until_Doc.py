"""MS-MARCO Document data path of SimANS (SimANS/utils/MARCO_until_Doc.py): Doc_v2Dataset = the MS-Pas dataset with
the Gaussian SimANS weights exp(-(s_i - s+ + b)^2 a) (:127-133), RoBERTa tokenisation (pad id 1), queries of 128 and
documents of 512 tokens, the cross-encoder row truncated to 512 (:166) and masks `ids != 1` (:200-203)."""
import math
import os
import random

import torch
from torch.utils.data import Dataset

from .MARCO_until_new import hash_str, read_sharded_tsv


def load_docs(file_name):
    """msmarco-docs.tsv: 'D<id> \\t url \\t title \\t body' -> {id: url<sep>title<sep>body[:10000]}  (:89-108)."""
    id_text = {}
    with open(file_name) as inp:
        for line in inp:
            a = line.split('\t')
            id_text[int(a[0][1:])] = (a[1].rstrip() + "<sep>" + a[2].rstrip() + "<sep>" + a[3].rstrip())[:10000]
    return id_text


def simans_draw_doc(neg_pairs, pos_score, num_neg, a=0.5, b=0.0, rng=random):
    """:122-148: Gaussian weights, rounds of num_neg draws with replacement, union / remove / repeat, set-order cut."""
    if pos_score == 0:
        return [p for p, _ in neg_pairs[-num_neg:]]
    cand = [p for p, _ in neg_pairs]
    w = [math.exp(-(s - pos_score + b) ** 2 * a) for _, s in neg_pairs]
    chosen = set()
    while len(chosen) < num_neg:
        chosen = chosen.union(rng.choices(cand, weights=w, k=num_neg))
        keep = [(c, wi) for c, wi in zip(cand, w) if c not in chosen]
        cand, w = [c for c, _ in keep], [wi for _, wi in keep]
    return list(chosen)[0:num_neg]


class Doc_v2Dataset(Dataset):
    Q_LEN, D_LEN, CE_LEN = 128, 512, 512

    def __init__(self, file_path, tokenizer, num_hard_negatives=1, a=0.5, b=0, trainer_id=0, trainer_num=1, is_training=True,
                 corpus_path='', rand_pool=50, p_text=None, p_title=None):
        self.file_path, self.tokenizer = file_path, tokenizer
        self.data = read_sharded_tsv(file_path, trainer_id, trainer_num)
        self.is_training, self.num_hard_negatives, self.rand_pool, self.a, self.b = is_training, num_hard_negatives, rand_pool, a, b
        self.p_text = load_docs(os.path.join(corpus_path, 'msmarco-docs.tsv')) if p_text is None else p_text

    def __len__(self):
        return len(self.data)

    def __getitem__(self, index):
        sample = self.data[index]
        pos_pairs = sample.pos_id.split(',')
        neg_pairs = [(int(p.split()[0]), float(p.split()[1])) for p in sample.neg_id.split(',')]
        pos_id, pos_score = (random.choice(pos_pairs) if self.is_training else pos_pairs[0]).split()
        pos_id, pos_score = int(pos_id), float(pos_score)
        neg_ids = simans_draw_doc(neg_pairs, pos_score, self.num_hard_negatives, self.a, self.b)
        enc = lambda text, n: self.tokenizer.encode(text, add_special_tokens=True, max_length=n, truncation=True)
        ctx_token_ids = [enc(self.p_text[pos_id], self.D_LEN)] + [enc(self.p_text[int(n)], self.D_LEN) for n in neg_ids]
        q_ids = enc(sample.query_string, self.Q_LEN)
        sep, pad = self.tokenizer.sep_token_id, self.tokenizer.pad_token_id

        def strip(t):
            return t[1:-1] if t[-1] == sep else t[1:]
        ce = [(q_ids + strip(c))[:self.CE_LEN] for c in ctx_token_ids]
        q = torch.LongTensor(q_ids + [pad] * (self.Q_LEN - len(q_ids)))
        ctx = torch.LongTensor([c + [pad] * (self.D_LEN - len(c)) for c in ctx_token_ids])
        ce = torch.LongTensor([c + [pad] * (self.CE_LEN - len(c)) for c in ce])
        return q, ctx, ce

    @classmethod
    def get_collate_fn(cls, args):
        def create_biencoder_input2(features):
            doc_per_question = features[0][1].size(0)
            q_tensor = torch.stack([f[0] for f in features], dim=0)
            doc_tensor = torch.cat([f[1] for f in features])
            ce = torch.cat([f[2] for f in features])
            positive_ctx_indices = [i * doc_per_question for i in range(len(features))]
            q_num, d_num = q_tensor.size(0), doc_tensor.size(0)
            tgt = torch.zeros((d_num), dtype=torch.long)
            tgt[positive_ctx_indices] = 1
            ce = ce.reshape(q_num, d_num // q_num, -1)
            return {'student': [q_tensor, (q_tensor != 1).long(), doc_tensor, (doc_tensor != 1).long(), positive_ctx_indices],
                    'teacher': [ce, (ce != 1).long(), tgt.reshape(q_num, d_num // q_num)]}
        return create_biencoder_input2


class RobertaHashTokenizer(object):
    """Offline stand-in for RobertaTokenizer (no vocab in this image): <s>=0, <pad>=1, </s>=2, words hashed into the rest."""
    cls_token_id, pad_token_id, sep_token_id = 0, 1, 2

    def __init__(self, vocab_size=50265):
        self.vocab_size = vocab_size

    def encode(self, text, text_pair=None, add_special_tokens=True, max_length=None, truncation=True, **kw):
        toks = [4 + (hash_str(w) % (self.vocab_size - 4)) for w in str(text).lower().replace("<sep>", " ").split()]
        ids = [self.cls_token_id] + toks + [self.sep_token_id] if add_special_tokens else toks
        if max_length is not None and len(ids) > max_length:
            ids = ids[:max_length - 1] + [self.sep_token_id] if add_special_tokens else ids[:max_length]
        return ids
