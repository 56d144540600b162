"""f1, whole ScalerNode (read_pipeline/nodes/ScalerNode.cpp:144-267) incl. the RNA adapter cut (:58-107) that rounds 1-3 left out.

Pinned by the reference itself: tests/golden/scaler_node.npz holds 86 reads x configurations run through the reference's OWN
ScalerNode (compiled in place, a real node fed through push_message: oracle/ref_scaler.cpp, oracle/Makefile.ref;
tests/golden/make_golden_scaler_node.py).  The contract is bit-exact: CRC-32 of the scaled + trimmed f16 signal, its length,
read_common.scale / shift, num_trimmed_samples, rna_adapter_end_signal_pos.

  not gpu: the oracle's restatement vs the fixture; the product's host orchestration (host::scaler_node through the ScalerOps seam,
           with the oracle's sample passes plugged in — no device) vs the fixture; determine_rna_adapter_pos: host == oracle ==
           compiled reference on 200 synthetic dRNA-like reads incl. ties.
  gpu:     the product path proper — statistics and sample map on the device (mibc_scaler_stats / mibc_scale_reads through
           HipCaller) — vs the fixture."""
import os
import zlib

import numpy as np
import pytest

from dorado_amd import config, hostapi, synth
from oracle import oracle_py as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CONFIGS = [  # tests/golden/make_golden_scaler_node.py CONFIGS
    ("quantile", (0.2, 0.9, 0.51, 0.53), (False, 0.0, 1.0), 0.1755, -243.0, float("nan"), ""),
    ("med_mad", (0.2, 0.9, 0.51, 0.53), (False, 0.0, 1.0), 0.1755, -243.0, float("nan"), ""),
    ("pa", (0.2, 0.9, 0.51, 0.53), (True, 93.69, 23.51), 0.1462, -228.0, 204.7, "FLO-PRO114M"),
    ("pa", (0.2, 0.9, 0.51, 0.53), (False, 0.0, 1.0), 0.1462, -228.0, float("nan"), "FLO-MIN114"),
    ("pa", (0.2, 0.9, 0.51, 0.53), (True, 79.2, 16.9), 0.1612, -251.0, 190.1, "FLO-PRO004RA"),
]


def _cases():
    g = np.load(os.path.join(GOLDEN, "scaler_node.npz"))
    offs = np.concatenate([[0], np.cumsum(g["raw_len"])])
    reads = [g["raw"][offs[i]:offs[i + 1]] for i in range(len(g["raw_len"]))]
    for k in range(len(g["case_read"])):
        strat, q, std, scaling, offset, opl, fc = CONFIGS[int(g["case_cfg"][k])]
        kw = dict(strategy=strat, quantile=q, standardisation=std, is_rna_model=bool(g["case_rna"][k]), scaling=scaling,
                  offset=offset, open_pore_level=opl, flow_cell_product_code=fc)
        want = dict(crc=int(g["out_crc32"][k]), head=g["out_head"][k], n=int(g["out_len"][k]),
                    scale_pa=g["scale_shift_pa"][k, 0], shift_pa=g["scale_shift_pa"][k, 1],
                    trimmed=int(g["trimmed_rna_end"][k, 0]), rna_end=int(g["trimmed_rna_end"][k, 1]))
        yield str(g["read_names"][int(g["case_read"][k])]), reads[int(g["case_read"][k])], kw, want


def _check(name, kw, got, want):
    tag = f"{name} {kw['strategy']} rna={kw['is_rna_model']} std={kw['standardisation'][0]}"
    bits = np.ascontiguousarray(got["signal"]).view(np.uint16)
    assert bits.size == want["n"], tag
    assert got["num_trimmed_samples"] == want["trimmed"] and got["rna_adapter_end_signal_pos"] == want["rna_end"], tag
    m = min(bits.size, want["head"].size)
    assert (bits[:m] == want["head"][:m]).all(), tag
    assert zlib.crc32(bits.tobytes()) == want["crc"], tag
    assert np.float32(got["scale_pa"]) == want["scale_pa"] and np.float32(got["shift_pa"]) == want["shift_pa"], tag


def test_fixture_covers_the_branches():
    seen = {(kw["is_rna_model"], kw["strategy"], w["trimmed"] > 10, w["trimmed"] == 0) for _, _, kw, w in _cases()}
    assert sum(1 for s in seen if s[0]) >= 5 and sum(1 for s in seen if not s[0]) >= 5
    assert any(s[0] and s[2] for s in seen)          # an RNA adapter was found and cut
    assert any(s[0] and s[3] for s in seen)          # none found
    assert any(not s[0] and s[2] for s in seen)      # the DNA trim heuristic found a peak
    assert any(not s[0] and s[3] for s in seen)      # the trim would swallow the read


def test_oracle_restatement_vs_reference_fixture():
    for name, x, kw, want in _cases():
        _check(name, kw, O.scaler_node(x, **kw), want)


def _oracle_stats(x, strategy, p4):
    return O.quantile_shift_scale(x, *p4) if strategy == "quantile" else O.med_mad(x)


def test_host_orchestration_vs_reference_fixture():
    """host::scaler_node (the product's C++ orchestration: adapter cut -> statistics -> sample map -> DNA trim, in the reference's
    order) with the two sample passes supplied through the ScalerOps seam — here by the oracle, on the GPU by HipCaller."""
    for name, x, kw, want in _cases():
        got = hostapi.scaler_node_ops(_oracle_stats, O.shift_scale_i16_to_f16, x, **kw)
        _check(name, kw, got, want)
        short = hostapi.scaler_node_ops(_oracle_stats, O.shift_scale_i16_to_f16, x, want_signal=False, **kw)
        assert short["n_out"] == want["n"] and short["num_trimmed_samples"] == want["trimmed"]
        assert short["first_sample"] == x.size - want["n"]
        assert (short["shift"], short["scale"]) == (got["shift"], got["scale"])


def test_scaler_node_rejects_an_empty_read():
    with pytest.raises(ValueError, match="empty read"):
        hostapi.scaler_node_ops(_oracle_stats, O.shift_scale_i16_to_f16, np.zeros(0, np.int16))


def test_rna_trim_decision():
    rng = np.random.default_rng(2)
    x = np.concatenate([rng.normal(480, 30, 3000), rng.normal(830, 90, 6000)]).round().astype(np.int16)
    pos = hostapi.rna_adapter_pos(x)
    assert 2700 <= pos <= 3100
    assert hostapi.rna_trim(x) == {"trim_start": pos, "rna_adapter_end_signal_pos": 0}
    assert hostapi.rna_trim(x, has_rna_based_adapters=True) == {"trim_start": 0, "rna_adapter_end_signal_pos": 0}
    assert hostapi.rna_trim(x[:1200]) == {"trim_start": 0, "rna_adapter_end_signal_pos": 0}     # shorter than the search start
    assert hostapi.rna_adapter_pos(np.zeros(0, np.int16)) == 0


def test_rna_adapter_pos_host_oracle_reference():
    """determine_rna_adapter_pos: host == oracle == the reference's own function (anonymous namespace of ScalerNode.cpp, reachable
    because ref_scaler.cpp #includes the file in place), incl. quantised signals whose window medians tie."""
    rng = np.random.default_rng(31)
    cases = []
    for _ in range(160):
        n = int(rng.integers(900, 30000))
        cut = int(rng.integers(0, n))
        lo, hi = float(rng.uniform(300, 800)), float(rng.uniform(300, 1000))
        x = np.concatenate([rng.normal(lo, rng.uniform(5, 60), cut), rng.normal(hi, rng.uniform(20, 120), n - cut)])
        cases.append(np.clip(np.round(x), -32768, 32767).astype(np.int16))
    for n in (5000, 9000, 12000):
        for lo, hi in ((500, 626), (500, 625), (600, 751), (600, 750), (700, 826), (576, 701)):
            x = np.full(n, lo, np.int16)
            x[n // 3:] = hi
            cases.append(x)
        cases.append((500 + (np.arange(n) // 50) % 3 * 60).astype(np.int16))
        cases.append((500 + (np.arange(n) // 250) % 5 * 40).astype(np.int16))
    have_ref = os.path.exists(O.REF_SCALER_SO)
    found = 0
    for x in cases:
        a, c = O.rna_adapter_pos(x), hostapi.rna_adapter_pos(x)
        assert a == c
        if have_ref:
            assert a == O.rna_adapter_pos(x, use_ref=True)
        found += a > 0
    assert found >= 30 and found <= len(cases) - 30


@pytest.mark.gpu
def test_scaler_node_on_device_vs_reference_fixture():
    """The product path: statistics (mibc_scaler_stats) and sample map (mibc_scale_reads) on the device through HipCaller, host
    logic around them — every fixture case bit-exact against the reference's ScalerNode."""
    cfg = config.tiny(128, 3)
    ws = synth.make_weights(cfg, seed=1)
    for k, (name, x, kw, want) in enumerate(_cases()):
        if k % 2 and kw["strategy"] == "pa":
            continue        # PA needs no statistics pass: half of those cases is enough on the device
        _check(name, kw, hostapi.scaler_node(cfg, ws, x, device="hip:0", **kw), want)
    name, x, kw, want = next(c for c in _cases() if c[0] == "dna_peak" and c[2]["strategy"] == "quantile" and not c[2]["is_rna_model"])
    short = hostapi.scaler_node(cfg, ws, x, device="hip:0", want_signal=False, **kw)
    assert short["n_out"] == want["n"] and short["num_trimmed_samples"] == want["trimmed"] > 10


@pytest.mark.skipif(not os.path.isdir("/root/reference/tests/data/model_configs"), reason="reference tree not present")
def test_scaler_parameters_come_from_the_model_config():
    """config.toml -> SignalNormalisationParams + sample type -> ScalerNode: the parsed hac@v4.3.0 (pa, standardised) and
    rna004 sup@v3.0.1 (quantile with edited multipliers, RNA) configs drive the host orchestration exactly as the explicit
    parameters do in the reference run."""
    base = "/root/reference/tests/data/model_configs"
    rng = np.random.default_rng(8)
    x = np.concatenate([rng.normal(480, 30, 2600), rng.normal(830, 90, 5000)]).round().astype(np.int16)
    cal = dict(scaling=0.1462, offset=-228.0, open_pore_level=204.7, flow_cell_product_code="FLO-PRO114M")
    for name, want_strategy, want_rna in (("dna_r10.4.1_e8.2_400bps_hac@v4.3.0", "pa", False), ("rna004_130bps_sup@v3.0.1", "quantile", True)):
        kw = hostapi.scaler_kwargs(config.load_model_config(os.path.join(base, name)))
        assert kw["strategy"] == want_strategy and kw["is_rna_model"] == want_rna
        got = hostapi.scaler_node_ops(_oracle_stats, O.shift_scale_i16_to_f16, x, **kw, **cal)
        ref = O.ref_scaler_node(x, kw["strategy"], kw["quantile"], kw["standardisation"], kw["is_rna_model"], cal["scaling"],
                                cal["offset"], cal["open_pore_level"], cal["flow_cell_product_code"]) if os.path.exists(O.REF_SCALER_SO) \
            else O.scaler_node(x, **kw, **cal)
        assert (got["signal"].view(np.uint16) == ref["signal"].view(np.uint16)).all()
        assert got["num_trimmed_samples"] == ref["num_trimmed_samples"] and np.float32(got["scale_pa"]) == np.float32(ref["scale_pa"])
        assert (got["num_trimmed_samples"] > 1000) == want_rna          # the RNA model cuts the adapter, the DNA model trims 10
    # an RNA002 model is NOT an "rna model" for ScalerNode (ScalerNode.cpp:157: only SampleType::RNA004 is): no adapter cut,
    # the DNA trim heuristic runs.  The two rna002 models are refused by name unless allow_deprecated.
    import copy
    c2 = copy.deepcopy(config.load_model_config(os.path.join(base, "rna004_130bps_sup@v3.0.1")))
    c2.sample_type = "RNA002"
    assert hostapi.scaler_kwargs(c2)["is_rna_model"] is False
    c2.sample_type = "RNA004"
    assert hostapi.scaler_kwargs(c2)["is_rna_model"] is True
