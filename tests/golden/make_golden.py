"""Generates tests/golden/*.npz by running the REFERENCE ITSELF (oracle/_ref/libdorado_ref.so =
the reference's own CPU sources compiled in place, see oracle/Makefile.ref) on seeded synthetic
inputs.  Run here (needs /root/reference + libtorch); the fixtures are committed so that the
oracle can be pinned on boxes where /root/reference does not exist.

    python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dorado_amd import config, synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def wsum(ws):
    c = 0
    for w in ws:
        c = zlib.crc32(np.ascontiguousarray(w).tobytes(), c)
    return np.uint32(c)


def planes(dec, T):
    n = len(dec)
    seq = np.zeros((n, T), np.uint8)
    qs = np.zeros((n, T), np.uint8)
    mv = np.zeros((n, T), np.uint8)
    ln = np.zeros((n,), np.int32)
    for i, (s, q, m) in enumerate(dec):
        seq[i, : len(s)] = np.frombuffer(s.encode(), np.uint8)
        qs[i, : len(q)] = np.frombuffer(q.encode(), np.uint8)
        mv[i] = m
        ln[i] = len(s)
    return seq, qs, mv, ln


def network_case(name, cfg, N, T_in, seed):
    ws = synth.make_weights(cfg, seed=seed)
    x16 = synth.make_signal(N, T_in, seed=seed + 1)
    x = x16.astype(np.float32)[:, None, :]
    scores = O.forward(cfg, ws, x, use_ref=True)
    fwd, bwd, posts = O.scans(scores, use_ref=True)
    dec = O.decode(scores, q_shift=cfg.qbias, q_scale=cfg.qscale, use_ref=True)
    seq, qs, mv, ln = planes(dec, scores.shape[1])
    np.savez_compressed(
        os.path.join(OUT, name + ".npz"),
        signal_f16=x16, scores=scores.astype(np.float32), bwd=bwd, posts_sample=posts[:, ::16],
        seq=seq, qstr=qs, moves=mv, seqlen=ln, weights_crc=wsum(ws), seed=np.int64(seed),
        N=np.int64(N), T_in=np.int64(T_in),
    )
    print(name, scores.shape, "seqlens", ln.tolist(), "score range", scores.min(), scores.max())


def decoder_case(name, state_len, N, T, seed, sigma=2.0):
    """Decoder alone on random f16-representable scores (what the GPU head emits)."""
    rng = np.random.default_rng(seed)
    K = 4 ** (state_len + 1)
    s = np.clip(rng.standard_normal((N, T, K)) * sigma, -5, 5).astype(np.float16).astype(np.float32)
    # make it structured: favour a random walk so that the beam is not flat
    dec = O.decode(s, q_shift=-1.1, q_scale=1.1, use_ref=True)
    _, bwd, _ = O.scans(s, use_ref=True)
    seq, qs, mv, ln = planes(dec, T)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), scores_f16=s.astype(np.float16),
                        bwd_t0=bwd[:, 0], seq=seq, qstr=qs, moves=mv, seqlen=ln)
    print(name, s.shape, "seqlens", ln.tolist())


def chunk_cases():
    rng = np.random.default_rng(7)
    rows = []
    for cs, st, ov in [(9996, 6, 498), (9996, 12, 492), (12288, 12, 600), (555, 5, 25), (83, 1, 13),
                       (4998, 6, 498)]:
        for _ in range(8):
            n = int(rng.integers(1, 300000))
            offs = O.generate_chunks(n, cs, st, ov, use_ref=True)
            rows.append((n, cs, st, ov, offs))
    np.savez_compressed(
        os.path.join(OUT, "chunks.npz"),
        args=np.array([r[:4] for r in rows], np.int64),
        counts=np.array([len(r[4]) for r in rows], np.int64),
        offsets=np.concatenate([np.array(r[4], np.int64) for r in rows]),
    )
    print("chunks", len(rows))


def scaler_cases():
    """outputs of the reference's own utils (compiled in oracle/_ref) on seeded reads,
    the expressions its tests pin them to (tests/TensorUtilsTest.cpp:44-63: torch.quantile 'lower';
    :121-139: ((x.float() - shift) / scale).half()), and the TrimTest.cpp:31-93 signal + answers."""
    import torch

    rng = np.random.default_rng(31)
    reads, q_ref, q_torch, mm_ref, ss, ss_ref = [], [], [], [], [], []
    for i in range(12):
        n = int(rng.integers(1, 40000)) if i else 1
        base = rng.integers(200, 900)
        x = (base + 90.0 * rng.standard_normal(n) + 40.0 * np.sin(np.arange(n) / 37.0)).astype(np.int16)
        if i == 3:
            x[::97] = 32767   # extreme values: wide counting range, int16 wrap in |x - med|
            x[5::101] = -32768
        qs = np.array([0.2, 0.9], np.float32)
        reads.append(x)
        q_ref.append(O.quantile_counting(x, qs, use_ref=True))
        q_torch.append(torch.quantile(torch.from_numpy(x).float(), torch.from_numpy(qs), 0, False,
                                      interpolation="lower").numpy())
        mm_ref.append(O.med_mad(x, use_ref=True))
        shift = float(rng.uniform(-100, 1000))
        scale = float(rng.uniform(0.1, 200))
        ss.append((shift, scale))
        ss_ref.append(O.shift_scale_i16_to_f16(x, shift, scale, use_ref=True).view(np.uint16))
        want = ((torch.from_numpy(x).float() - shift) / scale).half().numpy().view(np.uint16)
        assert (ss_ref[-1] == want).all()
    sig = O.trimtest_signal(2000)
    np.savez_compressed(
        os.path.join(OUT, "scaler.npz"),
        lens=np.array([len(r) for r in reads], np.int64), signal=np.concatenate(reads),
        quantiles_ref=np.array(q_ref, np.float32), quantiles_torch_lower=np.array(q_torch, np.float32),
        med_mad_ref=np.array(mm_ref, np.float32), shift_scale=np.array(ss, np.float32),
        scaled_f16_bits_ref=np.concatenate(ss_ref),
        trim_signal=sig, trim_expected=np.array([90, 60, 10, 10], np.int32),
    )
    print("scaler", len(reads), "reads")


if __name__ == "__main__":
    assert O.have_ref(), "build oracle/_ref first: make -C oracle -f Makefile.ref"
    chunk_cases()
    scaler_cases()
    network_case("net_tiny64_s3", config.tiny(64, 3), N=3, T_in=1200, seed=11)
    network_case("net_tiny128_s4", config.tiny(128, 4), N=2, T_in=900, seed=12)
    network_case("net_tx_tiny", config.tiny_tx(), N=2, T_in=1536, seed=13)
    decoder_case("dec_s3", 3, N=4, T=300, seed=21)
    decoder_case("dec_s4", 4, N=3, T=250, seed=22)
    decoder_case("dec_s5", 5, N=2, T=120, seed=23)
