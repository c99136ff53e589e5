"""Rocketqa_v2Dataset -- the MS-MARCO passage training set of the AR2/SimANS step, same file formats, constructor
and batch layout as SimANS/utils/MARCO_until_new.py:125-260:

  train_ce_<step>.tsv rows   qid \\t query \\t "pid score,..." (positives) \\t "pid score,..." (mined negatives, rank order)
  para.txt / para.title.txt  pid \\t text

``__getitem__`` draws the SimANS negatives on the host exactly like the reference (CPython ``random``:
weights exp(-|s_i - s_pos| * tau), rounds of ``random.choices(k=N)``, dedupe, remove, repeat; ``pos_score == 0`` ->
the last N candidates) so a seeded run replays the reference's picks.  ``sampler="gpu"`` (``--sampler gpu`` of the train job)
moves the whole of it to the device: ``build_device_pool`` tokenises this rank's queries and the passages its rows refer to
ONCE per iteration into int32 tables resident in HBM (+ the candidate scores, right-aligned), and ``device_batch`` then costs
the host a list of B row numbers per step -- positive pick, SimANS draw (``ops.simans_sample``: the same weights, rounds of N
with-replacement draws, dedupe, remove, repeat; Philox instead of CPython's Mersenne twister, so the picks follow the same
law but not the same sequence) and collate (``ops.assemble_batch``, bit-identical to the host collate) run on the GPU and
return the collate's dictionary with device tensors.
Collate -> {'student': [q[B,32], q_mask, ctx[B(1+N),128], ctx_mask, positive_ctx_indices],
            'teacher': [ce[B,1+N,160], ce_mask, tgt]}  (MARCO_until_new.py:241-258).
"""
import math
import os
import random
from collections import namedtuple

import torch
from torch.utils.data import Dataset

Example = namedtuple("Example", "query_id query_string pos_id neg_id".split())


def read_sharded_tsv(path, trainer_id=0, trainer_num=1):
    """line i belongs to rank i % trainer_num (MARCO_until_new.py:8-18)."""
    out = []
    with open(path, "r", encoding="utf8") as f:
        for i, line in enumerate(f):
            if i % trainer_num == trainer_id:
                out.append(Example(*line.rstrip("\n").split("\t")))
    return out


def load_id_text(file_name):
    id_text = {}
    with open(file_name) as inp:
        for line in inp:
            pid, text = line.strip().split("\t")
            id_text[int(pid)] = text
    return id_text


def simans_draw(neg_pairs, pos_score, num_neg, tau=3.0, rng=random):
    """MARCO_until_new.py:179-202 (Laplace form)."""
    if pos_score == 0:
        return [p for p, _ in neg_pairs[-num_neg:]]
    cand = [p for p, _ in neg_pairs]
    w = [math.exp(-abs(s - pos_score) * tau) for _, s in neg_pairs]
    chosen = set()
    while len(chosen) < num_neg:
        chosen = chosen.union(rng.choices(cand, weights=w, k=num_neg))
        keep = [(c, wi) for c, wi in zip(cand, w) if c not in chosen]
        cand, w = [c for c, _ in keep], [wi for _, wi in keep]
    return list(chosen)[0:num_neg]


class Rocketqa_v2Dataset(Dataset):
    def __init__(self, file_path, tokenizer, num_hard_negatives=1, trainer_id=0, trainer_num=1, is_training=True,
                 corpus_path='', rand_pool=50, p_text=None, p_title=None, sampler="host"):
        self.file_path, self.tokenizer = file_path, tokenizer
        self.data = read_sharded_tsv(file_path, trainer_id, trainer_num)
        self.is_training, self.num_hard_negatives, self.rand_pool, self.tau = is_training, num_hard_negatives, rand_pool, 3
        self.sampler = sampler
        self.p_text = load_id_text(os.path.join(corpus_path, 'para.txt')) if p_text is None else p_text
        self.p_title = load_id_text(os.path.join(corpus_path, 'para.title.txt')) if p_title is None else p_text  # sic (:139)

    def __len__(self):
        return len(self.data)

    # ---- sampler="gpu": SimANS draw + collate on the device (MARCO_until_new.py:165-258 without the DataLoader workers) -------
    def build_device_pool(self, device, q_len=32, p_len=128):
        """Tokenise this rank's queries and every passage its rows mention once; keep them, the candidate row numbers and the
        candidate scores in HBM.  Candidates are RIGHT-aligned in [NQ, Cmax] tables (pad score = +inf -> weight exp(-inf) = 0;
        the ``pos_score == 0`` rule "last N candidates" then still means the last N real ones)."""
        import numpy as np
        pad = self.tokenizer.pad_token_id
        rows, ptok = {}, []

        def row_of(pid):
            r = rows.get(pid)
            if r is None:
                r = rows[pid] = len(ptok)
                t = self._encode_ctx(pid)
                ptok.append(t + [pad] * (p_len - len(t)))
            return r
        NQ = len(self.data)
        qtok = np.full((NQ, q_len), pad, dtype=np.int32)
        parsed = []
        for i, sample in enumerate(self.data):
            q = self.tokenizer.encode(sample.query_string, add_special_tokens=True, max_length=q_len, truncation=True)
            qtok[i, :len(q)] = q
            pos = [(int(p.split()[0]), float(p.split()[1])) for p in sample.pos_id.split(',')]
            neg = [(int(p.split()[0]), float(p.split()[1])) for p in sample.neg_id.split(',')]
            if len(neg) < self.num_hard_negatives:
                raise ValueError("query %s has %d mined negatives, fewer than --number_neg %d" % (sample.query_id, len(neg), self.num_hard_negatives))
            parsed.append((pos, neg))
        cmax = max(len(n) for _, n in parsed)
        pmax = max(len(p) for p, _ in parsed)
        cand_rows = np.zeros((NQ, cmax), dtype=np.int32)
        cand_scores = np.full((NQ, cmax), np.inf, dtype=np.float64)
        pos_rows = np.zeros((NQ, pmax), dtype=np.int32)
        pos_scores = np.zeros((NQ, pmax), dtype=np.float64)
        pos_cnt = np.zeros(NQ, dtype=np.int64)
        for i, (pos, neg) in enumerate(parsed):
            k = len(neg)
            cand_rows[i, cmax - k:] = [row_of(pid) for pid, _ in neg]
            cand_scores[i, cmax - k:] = [s for _, s in neg]
            pos_rows[i, :len(pos)] = [row_of(pid) for pid, _ in pos]
            pos_scores[i, :len(pos)] = [s for _, s in pos]
            pos_cnt[i] = len(pos)
        t = lambda a: torch.from_numpy(a).to(device)
        self.pool = {"q_tok": t(qtok), "p_tok": t(np.asarray(ptok, dtype=np.int32).reshape(-1, p_len)), "cand_rows": t(cand_rows),
                     "cand_scores": t(cand_scores), "pos_rows": t(pos_rows), "pos_scores": t(pos_scores), "pos_cnt": t(pos_cnt),
                     "row_of": rows, "device": torch.device(device)}
        return self.pool

    def device_batch(self, indices, seed=0, step=0, pos_choice=None, neg_choice=None):
        """One training batch of rows `indices` assembled on the device -> the collate's dictionary (device tensors).
        pos_choice [B] (index into the row's positives) / neg_choice [B,N] (index into its candidate list) override the random
        picks (tests: the same picks as a host-side draw must give the collate's batch bit for bit)."""
        from .. import ops
        P = self.pool
        dev = P["device"]
        idx = torch.as_tensor(indices, dtype=torch.int64, device=dev)
        B, N = idx.numel(), self.num_hard_negatives
        cmax = P["cand_rows"].shape[1]
        if pos_choice is None:
            if self.is_training:                 # random.choice(pos_pairs) (:170) as a counter-based device draw
                g = torch.Generator(device=dev)
                g.manual_seed((int(seed) * 1000003 + int(step)) & 0x7FFFFFFF)
                pos_choice = (torch.rand(B, device=dev, generator=g) * P["pos_cnt"][idx]).long().clamp_(max=P["pos_rows"].shape[1] - 1)
            else:
                pos_choice = torch.zeros(B, dtype=torch.int64, device=dev)
        else:
            pos_choice = torch.as_tensor(pos_choice, dtype=torch.int64, device=dev)
        pos_row = P["pos_rows"][idx].gather(1, pos_choice[:, None])
        pos_score = P["pos_scores"][idx].gather(1, pos_choice[:, None]).squeeze(1)
        if neg_choice is None:
            neg = ops.simans_sample(P["cand_scores"][idx], pos_score, N, form=ops.LAPLACE, tau=float(self.tau), seed=seed, offset=step).long()
        else:                                    # indices into the row's OWN candidate list (0 = best ranked) -> right-aligned table
            nc = torch.as_tensor(neg_choice, dtype=torch.int64, device=dev)
            neg = nc + (cmax - torch.isfinite(P["cand_scores"][idx]).sum(1, keepdim=True))
        p_rows = torch.cat([pos_row, P["cand_rows"][idx].gather(1, neg).long()], dim=1).to(torch.int32)
        out = ops.assemble_batch(P["q_tok"], P["p_tok"], idx.to(torch.int32), p_rows, 1 + N, pad_id=self.tokenizer.pad_token_id,
                                 sep_id=self.tokenizer.sep_token_id, ce_len=160)
        out["picks"] = {"pos_choice": pos_choice, "neg_table_index": neg}
        return out

    def _encode_ctx(self, pid):
        title, para = self.p_title.get(int(pid), '-'), self.p_text[int(pid)]
        return self.tokenizer.encode(title, text_pair=para, add_special_tokens=True, max_length=128, truncation=True)

    def __getitem__(self, index):
        sample = self.data[index]
        pos_pairs = sample.pos_id.split(',')
        neg_pairs = [(int(p.split()[0]), float(p.split()[1])) for p in sample.neg_id.split(',')]
        pos_id, pos_score = (random.choice(pos_pairs) if self.is_training else pos_pairs[0]).split()
        pos_id, pos_score = int(pos_id), float(pos_score)
        neg_ids = simans_draw(neg_pairs, pos_score, self.num_hard_negatives, self.tau)
        ctx_token_ids = [self._encode_ctx(pos_id)] + [self._encode_ctx(n) for n in neg_ids]
        q_ids = self.tokenizer.encode(sample.query_string, add_special_tokens=True, max_length=32, truncation=True)
        sep, pad = self.tokenizer.sep_token_id, self.tokenizer.pad_token_id

        def strip(t):
            return t[1:-1] if t[-1] == sep else t[1:]
        ce = [q_ids + strip(c) for c in ctx_token_ids]
        q = torch.LongTensor(q_ids + [pad] * (32 - len(q_ids)))
        ctx = torch.LongTensor([c + [pad] * (128 - len(c)) for c in ctx_token_ids])
        ce = torch.LongTensor([c + [pad] * (160 - len(c)) for c in ce])
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
            return {'student': [q_tensor, (q_tensor != 0).long(), doc_tensor, (doc_tensor != 0).long(), positive_ctx_indices],
                    'teacher': [ce, (ce != 0).long(), tgt.reshape(q_num, d_num // q_num)]}
        return create_biencoder_input2


class HashTokenizer(object):
    """Offline stand-in for BertTokenizer (this image ships no vocab): whitespace tokens hashed into the vocab.
    For plumbing tests and synthetic runs only; pass a real tokenizer directory with --tokenizer_name in production."""
    cls_token_id, sep_token_id, pad_token_id = 101, 102, 0

    def __init__(self, vocab_size=30522):
        self.vocab_size = vocab_size

    def _tok(self, text):
        return [1000 + (hash_str(w) % (self.vocab_size - 1000)) for w in str(text).lower().split()]

    def encode(self, text, text_pair=None, add_special_tokens=True, max_length=None, truncation=True, **kw):
        a = self._tok(text)
        b = self._tok(text_pair) if text_pair is not None else None
        if add_special_tokens:
            ids = [self.cls_token_id] + a + [self.sep_token_id] + ((b + [self.sep_token_id]) if b is not None else [])
        else:
            ids = a + (b or [])
        if max_length is not None and len(ids) > max_length:
            ids = ids[:max_length - 1] + [self.sep_token_id] if add_special_tokens else ids[:max_length]
        return ids


def hash_str(w):
    h = 2166136261
    for ch in w.encode("utf8"):
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h
