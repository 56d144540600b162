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
