#!/usr/bin/env python3
"""tests/golden/pipeline_hac.npz (round 5): WHOLE raw reads through the reference's own simplex hot path on the CPU, end to end —
ScalerNode.cpp -> BasecallerNode.cpp (chunk.cpp, stitch.cpp) -> basecall/ModelRunner.cpp (CRFModel, f32) -> CPUDecoder, all
compiled in place (oracle/_ref/libdorado_ref_pipeline.so, oracle/ref_pipeline.cpp) — at the hac@v4.3.0 BASELINE configuration
(chunk 9996, overlap 498, PA scaling + standardisation of the model's config.toml).  Reads, calibrations and weights are
regenerated from seeds by the test (pipeline_reads below is imported by it); the fixture holds the reference's outputs per read:
sequence, qstring, move table, read_common.scale / shift (pA), num_trimmed_samples, the number of samples that reached the
basecaller, the chunk offsets (utils::generate_chunks of the compiled reference on that length) — and, for the identity floor, the
calls of the C restatement in f16-storage emulation on the same scaled signal (what an ideal f16 pipeline calls; see
make_golden_baseline.py).  Needs /root/reference (oracle/Makefile.ref); the fixture travels.
    python tests/golden/make_golden_pipeline.py"""
import os
import sys
import time
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from dorado_amd import config, synth  # noqa: E402

STANDARDISATION = (True, 91.88, 22.65)     # dna_r10.4.1_e8.2_400bps_hac@v4.3.0/config.toml [standardisation]
FLOW_CELL = "FLO-PRO114M"
EXPECTED_OPEN_PORE = 199.21                  # ScalerNode.cpp:112-139 for that flow cell
WEIGHT_SEED, READ_SEED = 42, 0x91BE


def model_weights(cfg):
    """Round 6: the synthetic model with decision margins (synth.make_margin_weights) on base-level reads."""
    return synth.make_margin_weights(cfg, seed=WEIGHT_SEED)


def pipeline_reads():
    """32 raw int16 reads + (scaling, offset, open_pore_level) each: 26 of about five chunks (38 k - 48 k samples), and six edge
    lengths (shorter than a chunk, exactly a chunk after the 10-sample trim, a chunk + a few samples, two and three chunks)."""
    rng = np.random.default_rng(READ_SEED)
    lens = [int(v) for v in rng.integers(38000, 48000, 26)] + [1500, 9996 + 10, 9996 + 10 + 7, 19494 + 10, 20500, 29000]
    raws, cal = [], []
    for i, n in enumerate(lens):
        scaling = float(rng.uniform(0.14, 0.2))
        offset = float(rng.integers(-260, -200))
        opl = float(rng.uniform(190.0, 210.0))
        x = synth.make_base_signal(1, n, seed=READ_SEED + 1 + i)[0].astype(np.float32)
        # a pore whose open-pore level sits above / below the flow cell's expected one (FLO-PRO114M: 199.21 pA) shifts every level
        # by the same amount — the shift ScalerNode's open-pore adjustment takes out again (ScalerNode.cpp:205-213); round 6: the
        # margin model reads LEVELS, so the synthetic read has to obey that physics (the random model of round 5 did not care)
        pa = STANDARDISATION[1] + STANDARDISATION[2] * x + (opl - EXPECTED_OPEN_PORE)
        raws.append(np.clip(np.round(pa / scaling - offset), -32768, 32767).astype(np.int16))
        cal.append((scaling, offset, opl))
    return raws, np.array(cal, np.float32)


def pack(strs, dtype=np.uint8):
    ln = np.array([len(s) for s in strs], np.int64)
    flat = np.frombuffer("".join(strs).encode(), dtype) if isinstance(strs[0], str) else np.concatenate(strs).astype(dtype)
    return flat, ln


def main():
    from oracle import oracle_py as O
    assert O.have_ref_pipeline(), "build oracle/_ref first: make -C oracle -f Makefile.ref"
    cfg = config.hac_v43()
    ws = model_weights(cfg)
    raws, cal = pipeline_reads()
    t0 = time.time()
    ref = O.ref_pipeline(cfg, ws, raws, cal, "pa", standardisation=STANDARDISATION, flow_cell_product_code=FLOW_CELL,
                         batch_size=16, num_runners=min(8, os.cpu_count() or 1))
    t1 = time.time()
    print(f"reference pipeline: {len(raws)} reads, {sum(len(r) for r in raws)} samples, {t1 - t0:.0f}s", flush=True)
    # chunk offsets of the compiled reference for the scaled lengths
    offs = [np.array(O.generate_chunks(r["scaled_len"], cfg.chunk_size, cfg.stride, cfg.overlap, use_ref=True), np.int64) for r in ref]
    # identity floor: the same reads through the C restatement in f16-storage emulation (scaler restatement -> chunks ->
    # network (f16 emulation) -> decoder -> stitch), i.e. what an ideal f16 pipeline calls
    f16_seq = []
    for i, (raw, c, r) in enumerate(zip(raws, cal, ref)):
        sn = O.scaler_node(raw, "pa", standardisation=STANDARDISATION, scaling=float(c[0]), offset=float(c[1]),
                           open_pore_level=float(c[2]), flow_cell_product_code=FLOW_CELL)
        assert sn["num_trimmed_samples"] == r["num_trimmed_samples"] and len(sn["signal"]) == r["scaled_len"]
        sig = sn["signal"]
        n = len(sig)
        chunks = np.zeros((len(offs[i]), cfg.chunk_size), np.float16)
        sizes = []
        for k, o in enumerate(offs[i]):
            seg = sig[o:o + cfg.chunk_size]
            sizes.append(len(seg))
            if len(seg) < cfg.chunk_size:      # BasecallerNode.cpp: a short (single) chunk is padded by repeating the signal
                seg = np.resize(seg, cfg.chunk_size)
            chunks[k] = seg
        with O.f16_emulation():
            sc = O.forward(cfg, ws, chunks.astype(np.float32)[:, None, :])
        dec = O.decode(sc, q_shift=cfg.qbias, q_scale=cfg.qscale, det=1)
        seq, _, _ = O.stitch_chunks(offs[i].tolist(), sizes, [d[2] for d in dec], [d[0] for d in dec], [d[1] for d in dec], n, cfg.stride)
        f16_seq.append(seq)
        print(f"  read {i}: {n} samples, ref {len(r['seq'])} bases, f16 emulation {len(seq)} bases", flush=True)
    t2 = time.time()
    seq, seq_len = pack([r["seq"] for r in ref])
    qs, _ = pack([r["qstr"] for r in ref])
    mv, mv_len = pack([r["moves"] for r in ref])
    fs, fs_len = pack(f16_seq)
    wcrc = 0
    for w in ws:
        wcrc = zlib.crc32(np.ascontiguousarray(w).tobytes(), wcrc)
    np.savez_compressed(
        os.path.join(HERE, "pipeline_hac.npz"),
        raw_len=np.array([len(r) for r in raws], np.int64), raw_crc=np.uint32(zlib.crc32(np.concatenate(raws).tobytes())),
        weights_crc=np.uint32(wcrc), calibration=cal,
        seq=seq, seq_len=seq_len, qstr=qs, moves=mv, moves_len=mv_len, f16_seq=fs, f16_seq_len=fs_len,
        scale_shift_pa=np.array([(r["scale_pa"], r["shift_pa"]) for r in ref], np.float32),
        num_trimmed=np.array([r["num_trimmed_samples"] for r in ref], np.int32),
        scaled_len=np.array([r["scaled_len"] for r in ref], np.int64),
        chunk_offsets=np.concatenate(offs), chunk_counts=np.array([len(o) for o in offs], np.int64))
    print(f"f16 emulation {t2 - t1:.0f}s; {int(seq_len.sum())} reference bases in {len(ref)} reads, "
          f"{int(np.array([len(o) for o in offs]).sum())} chunks")


if __name__ == "__main__":
    main()
