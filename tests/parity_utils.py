"""Helpers shared by the parity tests (test infrastructure)."""
import numpy as np


def edit_distance(a: bytes, b: bytes) -> int:
    """Levenshtein distance, one numpy row per character of a (the j-1 dependency is a running minimum)."""
    if len(a) < len(b):
        a, b = b, a
    bb = np.frombuffer(b, np.uint8)
    ar = np.arange(len(b) + 1)
    prev = ar.copy()
    for i, ca in enumerate(a, 1):
        cur = np.empty_like(prev)
        cur[0] = i
        np.minimum(prev[1:] + 1, prev[:-1] + (bb != ca), out=cur[1:])
        cur = np.minimum.accumulate(cur - ar) + ar
        prev = cur
    return int(prev[-1])


def identity(a: str, b: str) -> float:
    if not a and not b:
        return 1.0
    return 1.0 - edit_distance(a.encode(), b.encode()) / max(len(a), len(b))


def align_matches(a: bytes, b: bytes):
    """Global unit-cost alignment of a against b (full DP matrix, numpy row updates, traceback).  Returns (ok, d):
    ok[j] is True when position j of b is aligned to an equal character of a; d = edit distance."""
    aa = np.frombuffer(a, np.uint8)
    bb = np.frombuffer(b, np.uint8)
    n, m = len(aa), len(bb)
    D = np.empty((n + 1, m + 1), np.int32)
    ar = np.arange(m + 1, dtype=np.int32)
    D[0] = ar
    for i in range(1, n + 1):
        prev, cur = D[i - 1], D[i]
        cur[0] = i
        np.minimum(prev[1:] + 1, prev[:-1] + (bb != aa[i - 1]), out=cur[1:])
        cur[:] = np.minimum.accumulate(cur - ar) + ar
    ok = np.zeros(m, bool)
    i, j = n, m
    while i > 0 and j > 0:
        d = D[i, j]
        if d == D[i - 1, j - 1] + (aa[i - 1] != bb[j - 1]):
            ok[j - 1] = aa[i - 1] == bb[j - 1]
            i -= 1
            j -= 1
        elif d == D[i - 1, j] + 1:
            i -= 1
        else:
            j -= 1
    return ok, int(D[n, m])


def confident_identity(calls, ref_calls, qmin: int):
    """Identity restricted to the bases the REFERENCE calls with q >= qmin: (matched confident reference bases,
    confident reference bases, all reference bases) summed over the chunks.  calls / ref_calls: (seq, qstring, ...)."""
    good = tot = allb = 0
    for c, r in zip(calls, ref_calls):
        q = np.frombuffer(r[1].encode(), np.uint8).astype(int) - 33
        ok, _ = align_matches(c[0].encode(), r[0].encode())
        sel = q >= qmin
        good += int((ok & sel).sum())
        tot += int(sel.sum())
        allb += len(r[0])
    return good, tot, allb


def structured_scores(state_len: int, n: int, T: int, seed: int, gain: float = 1.0, clip: float = 5.0):
    """CRF transition scores [n, T, 4^(state_len+1)] (f16) that decode like a real chunk instead of like noise: every chunk
    follows a hidden path of stays and steps (a step every ~2.2 blocks) through three kinds of stretches of 60-250 blocks —
    SHARP (the step actually taken scores high, everything else sits below the fixed stay score of 2.0; weak steps that barely
    beat a stay and competing bases with almost the true score keep the beam honest), BLURRED (the same path at low contrast:
    dozens of candidates survive the beam cut) and FLAT (no signal at all, as the dec_s* fixtures: the cut-off bisection runs on
    every block and the beam stays full).  Over a full-length chunk (T = 1666 / 2048) this produces thousands of equal-hash
    stay / step folds, the presence filter, hundreds of bisections and full beams (basecall/decode/beam_search.cpp:236-409) —
    counted by orc_beam_stats for the fixtures.  Deterministic in (arguments, numpy's PCG64 stream); the fixtures store a
    CRC-32 of the bytes.  Score index of a step prev -> new: new * 4 + (prev >> 2 (state_len - 1)) (beam_search.cpp:209-222).
    gain scales everything (the transformer's unclamped head), clip = 0: no clipping."""
    rng = np.random.default_rng(seed)
    S = 4 ** state_len
    K = 4 * S
    mask = S - 1
    out = np.empty((n, T, K), np.float16)
    for c in range(n):
        s = np.empty((T, K), np.float32)
        kindseg = np.empty(T, np.int8)
        t = 0
        while t < T:
            ln = int(rng.integers(60, 250))
            k = int(rng.choice(3, p=[0.5, 0.3, 0.2]))
            kindseg[t:t + ln] = k
            m = slice(t, min(T, t + ln))
            rows = m.stop - m.start
            if k == 0:
                s[m] = rng.normal(-2.0, 1.2, (rows, K))
            elif k == 1:
                s[m] = rng.normal(-0.3, 1.3, (rows, K))
            else:
                s[m] = rng.normal(0.0, 2.0, (rows, K))
            t += ln
        state = int(rng.integers(0, S))
        step = rng.random(T) < 0.45
        base = rng.integers(0, 4, T)
        kind = rng.random(T)
        for t in range(T):
            if not step[t]:
                continue
            new = ((state << 2) & mask) | int(base[t])
            idx = new * 4 + (state >> (2 * (state_len - 1)))
            if kindseg[t] == 0:
                if kind[t] < 0.12:                         # weak step: about as good as staying
                    s[t, idx] = rng.normal(2.0, 0.4)
                else:
                    s[t, idx] = rng.normal(4.0, 0.8)
                if 0.12 <= kind[t] < 0.34:                 # a competing base, nearly as good
                    other = ((state << 2) & mask) | int((base[t] + 1 + rng.integers(0, 3)) % 4)
                    s[t, other * 4 + (state >> (2 * (state_len - 1)))] = rng.normal(3.4, 0.6)
            elif kindseg[t] == 1:
                s[t, idx] = rng.normal(2.6, 0.7)
            state = new
        s *= gain
        if clip > 0:
            np.clip(s, -clip, clip, out=s)
        out[c] = s.astype(np.float16)
    return out
