"""CPU checks of the parity helpers."""
import numpy as np

from parity_utils import align_matches, confident_identity, edit_distance, identity


def test_edit_distance_known_answers():
    assert edit_distance(b"kitten", b"sitting") == 3
    assert edit_distance(b"", b"abc") == 3 and edit_distance(b"abc", b"") == 3
    assert edit_distance(b"ACGT", b"ACGT") == 0
    assert edit_distance(b"AAAA", b"TTTT") == 4


def test_identity():
    assert identity("", "") == 1.0
    assert identity("ACGT", "ACGT") == 1.0
    assert abs(identity("ACGTACGT", "ACGAACG") - 0.75) < 1e-9


def test_align_matches_and_confident_identity():
    ok, d = align_matches(b"ACGTTGCA", b"ACGTGCA")       # one base deleted from b
    assert d == 1 and ok.all()
    ok, d = align_matches(b"ACGAACGT", b"ACGTACGT")      # one substitution
    assert d == 1 and ok.sum() == 7 and not ok[3]
    rng = np.random.default_rng(0)
    for _ in range(20):
        a = bytes(rng.choice(list(b"ACGT"), size=rng.integers(0, 60)).tolist())
        b = bytes(rng.choice(list(b"ACGT"), size=rng.integers(0, 60)).tolist())
        ok, d = align_matches(a, b)
        assert d == edit_distance(a, b) and ok.sum() <= min(len(a), len(b))
    # reference qstring: only the q >= 20 bases ('5' = q20) count
    good, tot, allb = confident_identity([("ACGAACGT", "")], [("ACGTACGT", "5555!!!5")], 20)
    assert (good, tot, allb) == (4, 5, 8)
