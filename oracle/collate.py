"""CPU restatement of the reference's batch assembly on PRE-TOKENISED rows.  TEST INFRASTRUCTURE.

Follows SimANS/utils/MARCO_until_new.py:204-258 (Rocketqa_v2Dataset.__getitem__ tail + create_biencoder_input2):
  * question ids padded to 32, passage ids (title [SEP] text, special tokens included) padded to 128;
  * cross-encoder input = question ids + passage ids without their first token and, when the passage ends with
    [SEP], without that last token (remove_special_token, :220-224), padded to 160;
  * masks are `ids != pad` (:249-252; pad id 1 for RoBERTa, MARCO_until_Doc.py:200-203);
  * positive_ctx_indices = i * docs_per_question, tgt one-hot on them.
The tokenizer itself is third-party (HF) and out of scope: rows arrive tokenised, padded with `pad_id`."""
import numpy as np


def row_len(ids, pad_id):
    """number of leading non-pad tokens (rows are `tokens + [pad]*k`)."""
    ids = np.asarray(ids)
    nz = np.nonzero(ids == pad_id)[0]
    return int(nz[0]) if len(nz) else len(ids)


def assemble(q_tok, p_tok, q_rows, p_rows, docs_per_question, pad_id=0, sep_id=102, ce_len=160):
    """q_tok [NQ,QL], p_tok [NP,PL] int; q_rows [B], p_rows [B*D] -> dict of int64 arrays."""
    q_tok, p_tok = np.asarray(q_tok, np.int64), np.asarray(p_tok, np.int64)
    B, D = len(q_rows), docs_per_question
    assert len(p_rows) == B * D
    q_ids = q_tok[np.asarray(q_rows)]
    c_ids = p_tok[np.asarray(p_rows)]
    ce = np.full((B * D, ce_len), pad_id, np.int64)
    ce_n = np.zeros(B * D, np.int64)
    for r in range(B * D):
        q = q_ids[r // D]
        ql = row_len(q, pad_id)
        p = c_ids[r]
        pl = row_len(p, pad_id)
        body = p[1:pl - 1] if (pl > 0 and p[pl - 1] == sep_id) else p[1:pl]
        row = np.concatenate([q[:ql], body])[:ce_len]
        ce[r, :len(row)] = row
        ce_n[r] = len(row)
    tgt = np.zeros(B * D, np.int64)
    pos = [i * D for i in range(B)]
    tgt[pos] = 1
    return dict(q_ids=q_ids, q_mask=(q_ids != pad_id).astype(np.int64), ctx_ids=c_ids,
                ctx_mask=(c_ids != pad_id).astype(np.int64), ce_ids=ce.reshape(B, D, ce_len),
                ce_mask=(ce != pad_id).astype(np.int64).reshape(B, D, ce_len), ce_len=ce_n,
                positive_ctx_indices=pos, tgt=tgt.reshape(B, D))
