"""POD5 reader without libpod5 + VBZ decode.

CPU part: container / footer / Arrow tables / zstd / the oracle's svb16 restatement, pinned on the
reference's own POD5 fixtures (tests/golden/pod5/, see README there) through the known answers its tests
hold (read counts, sample rates, read ids, file-name encoded chemistry) and through structural invariants
(every inflated row is consumed exactly and yields exactly `samples` values; values inside the ADC range).
GPU part: mibc_svb16_decode == oracle bit for bit; POD5 -> device decode -> device scaling -> calls."""
import glob
import os

import numpy as np
import pytest

from dorado_amd import capi, config, hostapi, pod5, synth
from oracle import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))
P5 = os.path.join(HERE, "golden", "pod5")
FILES = sorted(glob.glob(os.path.join(P5, "*.pod5")))


def test_fixture_set_present():
    assert len(FILES) == 6


def test_footer_and_tables():
    f = pod5.Pod5File(os.path.join(P5, "single_na24385.pod5"))
    assert f.footer["pod5_version"] == "0.1.5" and len(f.footer["contents"]) == 3
    assert f.signal_table.schema.names == ["read_id", "signal", "samples"]
    assert f.num_reads == 1                                             # FileInfoTest.cpp:35-37
    r = f.reads()[0]
    assert r.sample_rate == 4000                                        # FileInfoTest.cpp:51-54
    assert r.read_id == "002bd127-db82-436f-b828-28567c3d505d"
    assert r.num_samples == 47062 and r.offset == -254.0 and abs(r.scaling - 0.14620706) < 1e-7
    assert r.flow_cell_product_code == "FLO-PRO112" and r.is_end_reason_mux_change
    assert f.reads(allowed_read_ids=set()) == []                        # FileInfoTest.cpp:39-42
    assert len(f.reads(allowed_read_ids={"1", "2"})) == 0
    with pytest.raises(pod5.Pod5Error):
        pod5.parse_footer(b"not a pod5 file" * 10)


def test_multi_read_file_and_filters():
    f = pod5.Pod5File(os.path.join(P5, "filtered.pod5"))
    assert f.num_reads == 4
    ids = [r.read_id for r in f.reads()]
    assert "0007f755-bc82-432c-82be-76220b107ec5" in ids                # FileInfoTest.cpp:69-73
    assert len(f.reads(ignored_read_ids={"0007f755-bc82-432c-82be-76220b107ec5"})) == 3
    assert len(f.reads(allowed_read_ids={"0007f755-bc82-432c-82be-76220b107ec5"},
                       ignored_read_ids={"0007f755-bc82-432c-82be-76220b107ec5"})) == 0


@pytest.mark.parametrize("path", [p for p in FILES if "dna_r10" in p])
def test_metadata_matches_file_name(path):
    """The reference names these fixtures <model>-<flowcell>-<kit>-<sample rate>.pod5."""
    stem = os.path.basename(path)[:-5]
    _, fc, kit, sr = stem.rsplit("-", 3)
    r = pod5.Pod5File(path).reads()[0]
    assert r.flow_cell_product_code == fc.replace("_", "-")
    assert r.sequencing_kit.upper() == kit.replace("_", "-")
    assert r.sample_rate == int(sr) and r.num_samples == 2048


@pytest.mark.parametrize("path", FILES)
def test_vbz_rows_decode_exactly(path):
    """zstd frame -> svb16 stream consumed exactly, `samples` values, inside the ADC range."""
    f = pod5.Pod5File(path)
    for r in f.reads():
        streams, ns = f.inflated_rows(r.signal_rows)
        assert sum(ns) == r.num_samples
        parts = []
        for s, n in zip(streams, ns):
            x, used = O.svb16_decode(np.frombuffer(s, np.uint8), n)
            assert used == len(s)
            parts.append(x)
        x = np.concatenate(parts)
        assert x.size == r.num_samples
        if "single_na24385" in path:
            assert x.min() >= 0 and x.max() <= 2047                    # run_info adc_min / adc_max
            pa = (x.astype(np.float64) + r.offset) * r.scaling
            assert 30 < np.median(pa) < 150 and np.percentile(pa, 95) < 250   # picoampere scale of a nanopore read


def test_svb16_round_trip_property():
    rng = np.random.default_rng(3)
    for n in [1, 2, 7, 8, 9, 15, 16, 17, 4095, 4096, 4097, 100000]:
        x = (rng.integers(-300, 300, n).cumsum() % 4000 - 500).astype(np.int16)
        if n > 50:
            x[::37] = 32767
            x[5::41] = -32768
        s = O.svb16_encode(x)
        y, used = O.svb16_decode(s, n)
        assert used == len(s) and (x == y).all()
        assert O.svb16_decode(s[:-1], n)[1] == -1                      # truncated stream is detected


def test_zstd_frame_header():
    f = pod5.Pod5File(os.path.join(P5, "single_na24385.pod5"))
    raw = f.signal_table.column("signal")[0].as_py()
    n = pod5.zstd_frame_content_size(raw)
    assert n == len(pod5.zstd_inflate(raw)) == 53704
    with pytest.raises(pod5.Pod5Error):
        pod5.zstd_frame_content_size(b"\x00" * 16)
    # a frame that claims 2^64-1 bytes is refused against the row's own bound, before any allocation
    hostile = b"\x28\xb5\x2f\xfd" + bytes([0xC0]) + b"\x00" + b"\xff" * 8 + b"\x00" * 8
    assert pod5.zstd_frame_content_size(hostile) == 2 ** 64 - 1
    with pytest.raises(pod5.Pod5Error, match="at most"):
        pod5.zstd_inflate(hostile, pod5.svb16_max_bytes(47062))
    assert len(pod5.zstd_inflate(raw, pod5.svb16_max_bytes(int(f.signal_table.column("samples")[0].as_py())))) == n
    with pytest.raises(pod5.Pod5Error):
        pod5.zstd_inflate(raw[:40] + bytes(len(raw) - 40))            # corrupt payload -> Pod5Error, not a crash


def test_corrupt_footer_is_a_pod5_error():
    buf = bytearray(open(os.path.join(P5, "single_na24385.pod5"), "rb").read())
    good = pod5.parse_footer(bytes(buf))
    assert len(good["contents"]) == 3
    flen = int.from_bytes(buf[len(buf) - 32:len(buf) - 24], "little")
    start = len(buf) - 32 - flen
    rng = np.random.default_rng(3)
    n_err = 0
    for trial in range(300):
        b = bytearray(buf)
        for _ in range(1 + trial % 3):
            b[start + int(rng.integers(0, flen))] = int(rng.integers(0, 256))
        try:
            pod5.parse_footer(bytes(b))
        except pod5.Pod5Error:
            n_err += 1
    assert n_err > 20


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_svb16_equals_oracle_on_fixtures_and_random():
    cfg = config.tiny(128, 3)
    eng = capi.Engine(cfg, synth.make_weights(cfg, seed=1))
    streams, ns, want = [], [], []
    for path in FILES:
        f = pod5.Pod5File(path)
        for r in f.reads():
            s, n = f.inflated_rows(r.signal_rows)
            streams += s
            ns += n
    rng = np.random.default_rng(11)
    for n in [1, 7, 8, 9, 16, 17, 255, 4095, 4096, 4097, 65536, 300001]:
        x = (rng.integers(-200, 200, n).cumsum() % 3000).astype(np.int16)
        if n > 100:
            x[::53] = 32767
            x[7::59] = -32768
        streams.append(O.svb16_encode(x).tobytes())
        ns.append(n)
    for s, n in zip(streams, ns):
        x, used = O.svb16_decode(np.frombuffer(s, np.uint8), n)
        assert used == len(s)
        want.append(x)
    got, status = eng.svb16_decode(streams, ns)
    assert not status.any()
    for g, w in zip(got, want):
        assert (g == w).all()
    # corrupt rows are flagged, the others still decode
    bad = list(streams)
    bad[0] = bad[0][:-3]
    bad[2] = bad[2] + b"\x00\x00"
    got2, status2 = eng.svb16_decode(bad, ns)
    assert status2[0] == 1 and status2[2] == 1 and status2[1] == 0 and not status2[3:].any()
    assert (got2[1] == want[1]).all()
    eng.close()


@pytest.mark.gpu
def test_pod5_to_calls_device_pipeline():
    """POD5 file -> device VBZ decode -> PA scaling parameters (host formula) -> raw int16 chunks scaled
    inside conv1 -> calls; equals the reference order of operations done with the oracle pieces
    (oracle svb16 decode, oracle shift/scale on the CPU, trim 10, f16 chunks)."""
    cfg = config.tiny(128, 4)
    cfg.lstm_layers = 5
    cfg.chunk_size, cfg.overlap = 1200, 120
    cfg.normalise_basecaller_params()
    ws = synth.make_weights(cfg, seed=51)
    eng = capi.Engine(cfg, ws)
    reads = []
    for path in FILES:
        if "trimming_bomb" in path:
            continue                                                   # 10 samples: shorter than the trim
        f = pod5.Pod5File(path)
        reads += f.load_signals(f.reads(), eng)
    eng.close()
    assert sum(r.raw.size for r in reads) == sum(r.num_samples for r in reads) > 400000
    raws, ss, pre = [], [], []
    for r in reads:
        sc = hostapi.pa_read_scaling(True, 93.69, 23.5, r.scaling, r.offset, r.open_pore_level,
                                     r.flow_cell_product_code)
        ss.append((sc["shift"] + sc["open_pore_adjustment"], sc["scale"]))
        raws.append(r.raw)
        f = pod5.Pod5File(os.path.join(P5, r.filename))
        streams, ns = f.inflated_rows(r.signal_rows)
        x = np.concatenate([O.svb16_decode(np.frombuffer(s, np.uint8), n)[0] for s, n in zip(streams, ns)])
        pre.append(O.shift_scale_i16_to_f16(x, float(ss[-1][0]), float(ss[-1][1]))[10:])
    ss = np.array(ss, np.float32)
    got, _ = hostapi.basecall_raw_reads(cfg, ws, raws, ss, [10] * len(raws), num_runners=2, batch_size=64)
    want, _ = hostapi.basecall_reads(cfg, ws, pre, num_runners=2, batch_size=64)
    for g, w in zip(got, want):
        assert g[0] == w[0] and g[1] == w[1] and (g[2] == w[2]).all() and g[3] == w[3]


REF_DATA = "/root/reference/tests/data"


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="reference tree not present (GPU box)")
def test_every_reference_pod5_file_parses_and_decodes():
    """All POD5 files of the reference's test data (pod5 writer versions 0.1.5 - 0.3.35): footer, tables, read
    metadata, and every signal row inflates and is consumed exactly by the svb16 restatement."""
    files = sorted(glob.glob(os.path.join(REF_DATA, "**", "*.pod5"), recursive=True))
    assert len(files) >= 30
    versions, n_reads, n_samples = set(), 0, 0
    for path in files:
        f = pod5.Pod5File(path)
        versions.add(f.footer["pod5_version"])
        for r in f.reads():
            streams, ns = f.inflated_rows(r.signal_rows)
            assert sum(ns) == r.num_samples and r.sample_rate > 0 and r.scaling > 0
            for s, n in zip(streams, ns):
                x, used = O.svb16_decode(np.frombuffer(s, np.uint8), n)
                assert used == len(s) and x.size == n
            n_reads += 1
            n_samples += r.num_samples
    assert len(versions) >= 5 and n_reads >= 45 and n_samples > 3_000_000
