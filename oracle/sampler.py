"""SimANS ambiguous-negative sampler, restated.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference (under /root/reference):
  weights  MS-Pas (Laplace, tau=3)   SimANS/utils/MARCO_until_new.py:136,183-188
           NQ/TQ  (Gaussian a,b)     SimANS/utils/util_wiki.py:620-626
           MS-Doc (Gaussian a,b)     SimANS/utils/MARCO_until_Doc.py:127-133
  draw     rounds of N with-replacement draws, dedupe, remove, repeat
           SimANS/utils/MARCO_until_new.py:174-202, util_wiki.py:628-639

Two restatements:
  * ``reference_draw``  -- literal algorithm on CPython's ``random`` module; with the
    same ``random.seed`` it replays the imported reference exactly (pinned in
    tests/golden/sampler_ref.json by oracle/make_golden.py).
  * ``scheme_draw``     -- the same rounds scheme as a pure function of explicit
    Philox4x32-10 uniforms with a fixed f64 summation order; this is what the HIP
    kernel ``simx_simans_sample`` implements bit-for-bit.  It differs from the
    reference only in the RNG stream and in truncating the surplus of the last
    round in *draw order* instead of CPython ``set`` iteration order (documented
    deviation); the pre-truncation union has the same law, which
    tests check statistically.
"""
import math
import numpy as np

LAPLACE, GAUSS = 0, 1


def weights(scores, s_pos, form, a=0.5, b=0.0, tau=3.0):
    """Exact f64 SimANS weights (math.exp like the reference)."""
    if form == LAPLACE:
        return [math.exp(-abs(s - s_pos) * tau) for s in scores]
    return [math.exp(-(s - s_pos + b) ** 2 * a) for s in scores]


def reference_draw(rng, cand_ids, scores, s_pos, N, form, a=0.5, b=0.0, tau=3.0):
    """Literal MARCO_until_new.py:179-202.  Returns (union_set, negs_list)."""
    if s_pos == 0:
        negs = list(cand_ids[-N:])
        return set(negs), negs
    cand = list(cand_ids)
    w = weights(scores, s_pos, form, a, b, tau)
    chosen = set()
    while len(chosen) < N:
        chosen = chosen.union(rng.choices(cand, weights=w, k=N))
        nc, nw = [], []
        for c, wi in zip(cand, w):
            if c not in chosen:
                nc.append(c)
                nw.append(wi)
        cand, w = nc, nw
    return chosen, list(chosen)[0:N]


# ---------------------------------------------------------------- Philox4x32-10
_M0, _M1 = 0xD2511F53, 0xCD9E8D57
_W0, _W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32(ctr, key):
    """ctr: 4 python ints (32-bit), key: 2 python ints -> 4 uint32 (10 rounds)."""
    c0, c1, c2, c3 = [int(x) & 0xFFFFFFFF for x in ctr]
    k0, k1 = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> 32, p0 & 0xFFFFFFFF
        hi1, lo1 = p1 >> 32, p1 & 0xFFFFFFFF
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & 0xFFFFFFFF, lo1, (hi0 ^ c3 ^ k1) & 0xFFFFFFFF, lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def philox_uniform(seed, offset, q, rnd, j):
    """The j-th uniform double in [0,1) of round ``rnd`` for query ``q``.
    counter = (q, rnd, j>>1, offset) ; key = (seed_lo, seed_hi);
    words (0,1) for even j, (2,3) for odd j; 53 bits = (hi>>5)*2^26 + (lo>>6)."""
    r = philox4x32((q, rnd, j >> 1, offset), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    lo, hi = (r[0], r[1]) if (j & 1) == 0 else (r[2], r[3])
    return ((hi >> 5) * 67108864.0 + (lo >> 6)) * (1.0 / 9007199254740992.0)


def _wave_cumsum(w):
    """Inclusive f64 prefix sum in the kernel's summation order: 64 lanes, lane l owns the
    contiguous chunk [l*K,(l+1)*K), serial local sums, Hillis-Steele scan of the
    lane totals (offsets 1,2,..,32).  Returns (cum[C], total)."""
    C = len(w)
    K = (C + 63) // 64
    pad = np.zeros(64 * K, dtype=np.float64)
    pad[:C] = w
    loc = pad.reshape(64, K)
    inc = np.empty_like(loc)
    acc = np.zeros(64, dtype=np.float64)
    for k in range(K):
        acc = acc + loc[:, k]
        inc[:, k] = acc
    tot = inc[:, K - 1].copy()
    scan = tot.copy()
    off = 1
    while off < 64:
        sh = np.zeros(64, dtype=np.float64)
        sh[off:] = scan[:-off]
        scan = np.where(np.arange(64) >= off, scan + sh, scan)
        off *= 2
    excl = np.zeros(64, dtype=np.float64)
    excl[1:] = scan[:-1]
    cum = (excl[:, None] + inc).reshape(-1)[:C]
    return cum, float(scan[63])


def scheme_draw(scores, s_pos, N, form, a, b, tau, seed, offset, q, max_rounds=64):
    """Returns (negs[N] candidate indices, union list in draw order, rounds used)."""
    C = len(scores)
    if C < N:
        raise ValueError("need at least N candidates")
    if s_pos == 0:
        idx = list(range(C - N, C))
        return idx, idx, 0
    s = np.asarray(scores, dtype=np.float64)
    if form == LAPLACE:
        w = np.exp(-np.abs(s - s_pos) * tau)
    else:
        d = s - s_pos + b
        w = np.exp(-(d * d) * a)
    chosen, taken = [], np.zeros(C, dtype=bool)
    rnd = 0
    while len(chosen) < N and rnd < max_rounds:
        cum, total = _wave_cumsum(w)
        if not (total > 0.0):
            break
        for j in range(N):
            x = philox_uniform(seed, offset, q, rnd, j) * total
            idx = min(int((cum <= x).sum()), C - 1)
            if not taken[idx]:
                taken[idx] = True
                chosen.append(idx)
        w = np.where(taken, 0.0, w)
        rnd += 1
    if len(chosen) < N:                      # degenerate weights: fill with lowest free indices
        for i in range(C):
            if len(chosen) >= N:
                break
            if not taken[i]:
                taken[i] = True
                chosen.append(i)
    return chosen[:N], chosen, rnd
