"""Seeded synthetic weights and batches for bench / smoke / tests (product side).

Counter-based recipe (splitmix64 -> Box-Muller) so that the same tensors are generated on any
box without shipping checkpoints and without depending on a torch / numpy RNG stream.  Init
distribution follows the reference's ``init_weights`` (SimANS/model/models.py:452-465).
Synthetic batch shape follows SURVEY 8d: ids ~ U{1000..vocab-1}, [CLS]=101, [SEP]=102, pad=0.
"""
import numpy as np


def _mix(x):
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _key(seed, name):
    h = 0xCBF29CE484222325
    for ch in name.encode("utf8"):
        h = ((h ^ ch) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return (h ^ ((seed * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)) & 0xFFFFFFFFFFFFFFFF


def uniform(seed, name, n):
    with np.errstate(over="ignore"):
        ctr = np.arange(n, dtype=np.uint64) + np.uint64(_key(seed, name))
    return (_mix(ctr) >> np.uint64(11)).astype(np.float64) / 9007199254740992.0


def normal(seed, name, shape, std=1.0):
    n = int(np.prod(shape))
    r = np.sqrt(-2.0 * np.log(1.0 - uniform(seed, name + "#a", n)))
    return (r * np.cos(2.0 * np.pi * uniform(seed, name + "#b", n)) * std).reshape(shape)


def randint(seed, name, lo, hi, shape):
    return (lo + np.floor(uniform(seed, name, int(np.prod(shape))) * (hi - lo))).astype(np.int64).reshape(shape)


def fill_bert_state_dict(named_shapes, seed, perturb=True, std=0.02):
    """{hf_key: np.float32 array} for the given [(name, shape)] list."""
    out = {}
    for name, shape in named_shapes:
        if name.endswith("LayerNorm.weight"):
            w = 1.0 + (normal(seed, name, shape, 0.05) if perturb else np.zeros(shape))
        elif name.endswith(".bias"):
            w = normal(seed, name, shape, 0.02) if perturb else np.zeros(shape)
        else:
            w = normal(seed, name, shape, std)
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def make_batch(seed, n, S, vocab, len_mean, len_std, len_min, full=False):
    lo = 1000 if vocab > 2000 else 110
    ids = randint(seed, "ids", lo, vocab, (n, S))
    if full:
        lens = np.full((n,), S, dtype=np.int64)
    else:
        lens = np.clip(np.rint(len_mean + len_std * normal(seed, "lens", (n,), 1.0)), len_min, S).astype(np.int64)
    mask = (np.arange(S)[None, :] < lens[:, None]).astype(np.int64)
    ids = ids * mask
    ids[:, 0] = 101
    ids[np.arange(n), lens - 1] = 102
    return ids, mask, lens
