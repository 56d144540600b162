"""CPU checks of the parity helpers."""
from parity_utils import edit_distance, identity


def test_edit_distance_known_answers():
    assert edit_distance(b"kitten", b"sitting") == 3
    assert edit_distance(b"", b"abc") == 3 and edit_distance(b"abc", b"") == 3
    assert edit_distance(b"ACGT", b"ACGT") == 0
    assert edit_distance(b"AAAA", b"TTTT") == 4


def test_identity():
    assert identity("", "") == 1.0
    assert identity("ACGT", "ACGT") == 1.0
    assert abs(identity("ACGTACGT", "ACGAACG") - 0.75) < 1e-9
