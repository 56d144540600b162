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
