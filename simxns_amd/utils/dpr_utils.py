"""Runtime utilities of the path with the reference's names (SimANS/utils/dpr_utils.py): checkpoint state,
model unwrapping, the gather helper, and what the NQ / TriviaQA generate job needs on the host: answer matching
(SimpleTokenizer / has_answer, :309-420) and the ranking metrics (Eval_Tool, :91-164).  The FAISS indexers are replaced by
simxns_amd.retrieval.FlatIPIndex."""
import collections
import logging
import math
import unicodedata

import numpy as np
import torch

from ..parallel import all_gather_list  # noqa: F401  (same name / call signature as dpr_utils.py:166)

logger = logging.getLogger()

# dpr_utils.py:22-24 -- on-disk layout: torch.save(state._asdict(), path)
CheckpointState = collections.namedtuple("CheckpointState",
                                         ['model_dict', 'optimizer_dict', 'scheduler_dict', 'offset', 'epoch',
                                          'encoder_params'])


def get_model_obj(model):
    """dpr_utils.py:57-58"""
    return model.module if hasattr(model, 'module') else model


def load_states_from_checkpoint(model_file: str) -> CheckpointState:
    """dpr_utils.py:73-77"""
    logger.info('Reading saved model from %s', model_file)
    state_dict = torch.load(model_file, map_location='cpu', weights_only=False)
    logger.info('model_state_dict keys %s', state_dict.keys())
    return CheckpointState(**state_dict)


def save_checkpoint_state(path, model, optimizer, scheduler, offset=0, epoch=0, encoder_params=None):
    """co_training_marco_train.py:310-345 (_save_checkpoint / _save_teacher_checkpoint)."""
    model_to_save = get_model_obj(model)
    state = CheckpointState(model_to_save.state_dict(), optimizer.state_dict(), scheduler.state_dict(), offset, epoch,
                            encoder_params)
    torch.save(state._asdict(), path)
    return path


class Tokens(object):
    """Token list of SimpleTokenizer.tokenize (dpr_utils.py:391-420, the part the answer match uses)."""

    def __init__(self, data):
        self.data = data

    def __len__(self):
        return len(self.data)

    def words(self, uncased=False):
        return [t.lower() for t in self.data] if uncased else list(self.data)


class SimpleTokenizer(object):
    """DPR's answer-matching tokeniser (dpr_utils.py:345-385): a maximal run of letters / digits / combining marks is one
    token, every other character that is neither a separator nor a control character is a token of its own."""

    def __init__(self, **kwargs):
        import regex
        self._pattern = regex.compile(r"[\p{L}\p{N}\p{M}]+|[^\p{Z}\p{C}]", flags=regex.UNICODE)

    def tokenize(self, text):
        return Tokens(self._pattern.findall(text))


def _key(words):
    # token lists compared as one string: NUL never occurs inside a token (control characters are not tokens)
    return "\0" + "\0".join(words) + "\0"


def has_answer(answers, text, tokenizer, match_type='string') -> bool:
    """dpr_utils.py:309-343: does the passage contain one of the answers?  'string': the answer's token sequence occurs
    contiguously in the passage's (NFD-normalised, lower-cased); 'regex': the answer is a pattern searched in the text."""
    text = unicodedata.normalize('NFD', text)
    if match_type == 'string':
        hay = _key(tokenizer.tokenize(text).words(uncased=True))
        for a in answers:
            words = tokenizer.tokenize(unicodedata.normalize('NFD', a)).words(uncased=True)
            if not words or _key(words) in hay:          # (an answer without tokens matches anywhere, as the reference's slice compare)
                return True
    elif match_type == 'regex':
        import re
        for a in answers:
            try:
                pat = re.compile(unicodedata.normalize('NFD', a), flags=re.IGNORECASE + re.UNICODE + re.MULTILINE)
            except BaseException:
                continue
            if pat.search(text) is not None:
                return True
    return False


class Eval_Tool(object):
    """Ranking metrics over per-question hit lists (dpr_utils.py:91-164), each averaged over the questions.  MAP_n divides by
    n and nDCG_n by sum_{i<n} log2(i+2) -- the reference's definitions, kept so that eval_result files stay comparable."""

    @staticmethod
    def _hits(results_list, n):
        h = np.zeros((len(results_list), n), dtype=bool)
        for i, r in enumerate(results_list):
            r = list(r)[:n]
            h[i, :len(r)] = [bool(x) for x in r]
        return h

    @classmethod
    def MRR_n(cls, results_list, n):
        h = cls._hits(results_list, n)
        first = np.where(h.any(1), h.argmax(1) + 1.0, np.inf)
        return float((1.0 / first).sum() / len(results_list))

    @classmethod
    def MAP_n(cls, results_list, n):
        h = cls._hits(results_list, n)
        return float(((np.cumsum(h, 1) * h) / np.arange(1, n + 1)).sum() / n / len(results_list))

    @classmethod
    def DCG_n(cls, results_list, n):
        h = cls._hits(results_list, n)
        return float((h / np.log2(np.arange(n) + 2.0)).sum() / len(results_list))

    @classmethod
    def nDCG_n(cls, results_list, n):
        return cls.DCG_n(results_list, n) / sum(math.log2(i + 2) for i in range(n))

    @classmethod
    def P_n(cls, results_list, n):
        return float(cls._hits(results_list, n).sum() / n / len(results_list))

    @classmethod
    def get_matrics(cls, results_list):
        out = {}
        for name in ('MRR_n', 'MAP_n', 'DCG_n', 'nDCG_n', 'P_n'):
            for p in (1, 5, 10, 20, 50, 100):
                out[name + '@_' + str(p)] = getattr(cls, name)(results_list, p)
        return out
