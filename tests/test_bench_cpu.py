"""bench.py's host-side helpers (no GPU): the numbers the line derives its roofline / identity objects from."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dorado_amd import config  # noqa: E402


def test_network_flops_per_sample_match_the_survey_figures():
    """SURVEY.md 8d / DESIGN.md 4: hac 2.139 MFLOP, sup@v4.3 15.49 MFLOP, sup@v5 14.35 MFLOP per raw sample."""
    assert abs(bench.network_flops_per_sample(config.hac_v43()) / 1e6 - 2.139) < 0.01
    assert abs(bench.network_flops_per_sample(config.sup_v43()) / 1e6 - 15.49) < 0.05
    assert abs(bench.network_flops_per_sample(config.sup_v50()) / 1e6 - 14.35) < 0.05
    # the dominant kernel's algorithmic work per launch at the bench batch: N T 2 (4C)(2C)
    assert abs(bench.lstm_flops_per_launch(config.hac_v43(), 16384, 1666) - 6.44e13) < 1e11


def test_headline_arithmetic_follows_the_reference_rule():
    assert config.hac_v43().reference_gpu_lstm_int8() and config.sup_v43().reference_gpu_lstm_int8()
    name, sub = bench.dominant_kernel(_q(config.hac_v43()), 16384)
    assert "lstm_layer_q8_kernel<384>" in name
    name, sub = bench.dominant_kernel(_q(config.sup_v43()), 8192)
    assert "lstm_layer_cl_kernel<1024, int8>" in name
    assert "lstm_layer_x8_kernel" in bench.dominant_kernel(config.hac_v43(), 16384)[0]


def _q(cfg):
    cfg.lstm_quant = True
    return cfg


def test_committed_profiles_resolve_for_the_bench_workloads():
    """The line quotes PMC traffic and identity from committed files: they must exist for the configurations it runs."""
    for model, n, t_in, sub in (("hac_q8", 16384, 9996, "lstm_layer_q8_kernel<384, 4, false"), ("hac", 16384, 9996, "lstm_layer_x8"),
                                ("sup_q8", 8192, 9996, "lstm_layer_cl_kernelILi1024ELb0ELi0ELi1E"), ("sup", 8192, 9996, "lstm_layer_cl")):
        tr = bench.pmc_traffic(sub, model, n, t_in)
        assert tr and tr["hbm_bytes"] > 1e9 and os.path.exists(os.path.join(ROOT, tr["source"])), model
    assert bench.pmc_traffic(None, "sup5", 1024, 12288, total=True)["hbm_bytes"] > 1e11
    for key, quant, lo in (("hac", True, 0.99), ("hac", False, 0.995), ("sup", True, 0.99), ("sup", False, 0.995), ("sup5", False, 0.97)):
        ident = bench.committed_identity(key, quant)
        assert ident and ident["per_chunk_identity_median"] >= lo, (key, quant, ident)
        assert json.load(open(os.path.join(ROOT, ident["source"])))["identity_vs_reference"]["median"] == ident["per_chunk_identity_median"]
    assert bench.committed_identity("tiny", False) is None


def test_full_length_cpu_baselines_are_committed_for_all_three_models():
    import glob
    for key in ("hac", "sup", "sup5"):
        files = glob.glob(os.path.join(ROOT, "profiles", f"r*_cpu_baseline_full_{key}.json"))
        assert files, key
        cb = json.load(open(sorted(files)[-1]))["cpu_baseline"]
        assert cb["kind"] == "reference" and cb["value"] > 1e4 and "full: real T_in" in cb["sample"]
