#!/usr/bin/env python3
"""tests/golden/dec_full_s{4,5}.npz (round 5): FULL-LENGTH decoder fixtures from the compiled reference.  The round 1-4 decoder
fixtures (dec_s3 / s4 / s5) stop at T = 120 ... 300; at the BASELINE chunk lengths (T = 1666 hac / sup@v4.3, 2048 sup@v5) the
device used to be compared with oracle.c only.  Here the reference's own CPUDecoder + beam_search (oracle/_ref) decode
structured f16 scores (tests/parity_utils.structured_scores: a hidden stay / step path with weak steps and competing bases, so
that hash merges, the presence filter and the cut-off bisection are exercised along the whole chunk):
    dec_full_s4: 4 chunks x 1666 steps x 1024 transitions, clamped range (+-5)        — hac@v4.3.0's decoder shape
    dec_full_s5: 2 chunks x 2048 steps x 4096 transitions, gain 1.5, no clamp (+-17)     — sup@v5.0.0's decoder shape
Scores are regenerated from the seed by the tests (CRC-32 stored); the fixture holds the reference's moves / bases / qstring.
    python tests/golden/make_golden_decoder_full.py"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))
from parity_utils import structured_scores  # noqa: E402

# name -> (state_len, N, T, seed, gain, clip, q_shift, q_scale)
CASES = {
    "dec_full_s4": (4, 4, 1666, 0xDEC4, 1.0, 5.0, -1.1, 1.1),
    "dec_full_s5": (5, 2, 2048, 0xDEC5, 1.5, 0.0, -0.3, 0.95),
}


def planes(dec, T):
    n = len(dec)
    seq, qs, mv = np.zeros((n, T), np.uint8), np.zeros((n, T), np.uint8), np.zeros((n, T), np.uint8)
    ln = np.zeros((n,), np.int32)
    for i, (s, q, m) in enumerate(dec):
        seq[i, : len(s)] = np.frombuffer(s.encode(), np.uint8)
        qs[i, : len(q)] = np.frombuffer(q.encode(), np.uint8)
        mv[i] = m
        ln[i] = len(s)
    return seq, qs, mv, ln


def main():
    from oracle import oracle_py as O
    assert O.have_ref(), "build oracle/_ref first: make -C oracle -f Makefile.ref"
    for name, (L, n, T, seed, gain, clip, qsh, qsc) in CASES.items():
        s16 = structured_scores(L, n, T, seed, gain, clip)
        dec = O.decode(s16.astype(np.float32), q_shift=qsh, q_scale=qsc, use_ref=True)
        seq, qs, mv, ln = planes(dec, T)
        # what the decode exercised, counted by the C restatement (which must agree with the reference on every output)
        import ctypes as C
        O.lib().orc_beam_stats(None, 1)
        dec_o = O.decode(s16.astype(np.float32), q_shift=qsh, q_scale=qsc, det=0)
        st = (C.c_long * 5)()
        O.lib().orc_beam_stats(st, 1)
        assert all(a[0] == b[0] and (a[2] == b[2]).all() for a, b in zip(dec, dec_o)), "oracle.c differs from the reference"
        np.savez_compressed(os.path.join(HERE, name + ".npz"), params=np.array([L, n, T, seed], np.int64),
                            fparams=np.array([gain, clip, qsh, qsc], np.float32), scores_crc=np.uint32(zlib.crc32(s16.tobytes())),
                            seq=seq, qstr=qs, moves=mv, seqlen=ln, beam_stats=np.array(list(st), np.int64))
        print("   blocks, equal-hash folds, bisected blocks, exhausted bisections, full-beam blocks:", list(st))
        print(name, s16.shape, "range", float(s16.min()), float(s16.max()), "seqlens", ln.tolist(),
              "mean q", [round(float(np.mean(np.frombuffer(d[1].encode(), np.uint8)) - 33), 1) for d in dec])


if __name__ == "__main__":
    main()
