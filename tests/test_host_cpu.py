"""CPU tests of the host layer and the boundary: C-ABI symbols, C++ chunk/stitch/device-string
mirrors against the reference's own known answers and the oracle, multi-process (gloo, world 2)
plumbing.  No compute call is made without a GPU."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from dorado_amd import capi, config, dist, hostapi
from oracle import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cabi_library_loads_and_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mibc.h")).read()
    declared = sorted(set(re.findall(r"\b(mibc_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    L = capi.lib()
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert set(declared) == set(capi.EXPORTS)
    # ... and NOTHING else leaves the product library (-fvisibility=hidden + csrc/mibc.map): no mibc_launch_* internals,
    # no mibc_debug_* test hooks (those live in libmibc_dbg.so only), no kernel handles, no libstdc++ instantiations
    nm = subprocess.run(["nm", "-D", "--defined-only", capi.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted(l.split()[-1] for l in nm.splitlines() if l.strip())
    assert exported == declared, sorted(set(exported) ^ set(declared))
    nm = subprocess.run(["nm", "-D", "--defined-only", capi.DBG_LIB_PATH], capture_output=True, text=True, check=True).stdout
    dbg = sorted(l.split()[-1] for l in nm.splitlines() if l.strip())
    assert set(declared) <= set(dbg) and all(n.startswith("mibc_debug_") for n in set(dbg) - set(declared))


def test_library_build_id_is_the_hash_of_the_tree():
    """libmibc.so / libmibc_dbg.so carry a hash of every source they were compiled from (mibc_build_id,
    tools/build_id.py via the Makefile); tests/conftest.py ends the session when it is not the tree's.  Here: the ids agree, and
    the hash really covers the sources (a changed byte changes it)."""
    import ctypes
    import importlib.util
    spec = importlib.util.spec_from_file_location("build_id", os.path.join(ROOT, "tools", "build_id.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    want = mod.build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", want)
    assert capi.lib().mibc_build_id().decode() == want
    dbg = ctypes.CDLL(capi.DBG_LIB_PATH)
    dbg.mibc_build_id.restype = ctypes.c_char_p
    assert dbg.mibc_build_id().decode() == want + "-dbg"
    files = mod.source_files()
    assert any(f.endswith("engine.hip") for f in files) and any(f.endswith("mibc.h") for f in files) and len(files) >= 20
    real_open = open

    def patched(path, *a, **k):      # the same tree with one byte appended to one kernel source
        fh = real_open(path, *a, **k)
        if str(path).endswith("decode.hip") and "b" in (a[0] if a else k.get("mode", "")):
            import io
            return io.BytesIO(fh.read() + b" ")
        return fh
    import builtins
    builtins.open, saved = patched, builtins.open
    try:
        assert mod.build_id() != want
    finally:
        builtins.open = saved


def test_reference_rule_for_the_lstm_arithmetic():
    """nn/ConvStack.cpp:60-89 get_koi_lstm_input_layout as config.reference_gpu_lstm_int8 (bench.py's headline arithmetic; the
    adapter restates it in C++ with the DORADO_LSTM_MODE override): int8 iff the last convolution ends in tanh and
    128 < lstm_size <= 1024, lstm_size % 128 == 0."""
    assert config.hac_v43().reference_gpu_lstm_int8() and config.sup_v43().reference_gpu_lstm_int8()
    assert not config.sup_v50().reference_gpu_lstm_int8()            # transformer
    assert not config.fast_v40().reference_gpu_lstm_int8()           # lstm_size 96, swish front end
    assert not config.tiny(128, 4).reference_gpu_lstm_int8()         # 128 is not > 128
    for c in (256, 384, 512, 768, 1024):
        assert config.tiny(c, 4).reference_gpu_lstm_int8()
    sw = config.tiny(384, 4)
    sw.convs[2].activation = config.ACT_SWISH_CLAMP
    assert not sw.reference_gpu_lstm_int8()                          # CUTLASS_TNC_F16 for a swish front end
    hdr = open(os.path.join(ROOT, "integration", "HipModelRunnerAdapter.h")).read()
    assert "mibc_reference_lstm_int8" in hdr and "DORADO_LSTM_MODE" in hdr and "d.lstm_quant = mibc_reference_lstm_int8(c)" in hdr


def test_no_gpu_means_loud_failure_not_fallback():
    if capi.device_count() > 0:
        pytest.skip("GPU present")
    from dorado_amd import synth
    cfg = config.tiny(128, 3)
    with pytest.raises(capi.MibcError):
        capi.Engine(cfg, synth.make_weights(cfg))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "dorado_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle_py" not in src and "liboracle" not in src and "libdorado_ref" not in src, f


def test_host_generate_chunks_known_answers_and_oracle():
    # /root/reference/tests/ChunkTest.cpp:28-39
    assert hostapi.generate_chunks(9996 // 2, 9996, 6, 498) == [0]
    assert hostapi.generate_chunks(9996 + 1, 9996, 6, 498) == [0, 6]
    assert hostapi.generate_chunks(3 * 9996, 9996, 6, 498) == [0, 9498, 18996, 19992]
    for bad in [(0, 9996, 6, 498), (12345, 0, 6, 498), (12345, 9996, 0, 498), (12345, 9996, 10, 498),
                (12345, 9996, 7, 498), (12345, 9996, 6, 9996)]:
        with pytest.raises(ValueError):
            hostapi.generate_chunks(*bad)
    rng = np.random.default_rng(3)
    for cs, st, ov in [(9996, 6, 498), (12288, 12, 600), (555, 5, 25), (83, 1, 13)]:
        for n in rng.integers(1, 400000, size=24):
            assert hostapi.generate_chunks(int(n), cs, st, ov) == O.generate_chunks(int(n), cs, st, ov)


def test_host_stitch_known_answer_and_oracle():
    # /root/reference/tests/StitchTest.cpp:10-99
    moves = [[1, 0, 0, 1, 0, 0, 1, 0, 1, 0], [1, 0, 0, 1, 0, 0, 0, 1, 0, 1], [1, 0, 0, 1, 0, 1, 1, 0, 0, 0],
             [1, 0, 0, 1, 0, 0, 1, 0, 1, 0], [0, 1, 0, 1, 0, 0, 1, 0, 1, 0], [1, 0, 0, 0, 0, 0, 1, 0, 1, 1],
             [1, 0, 0, 1, 0, 0, 1, 0, 1, 0]]
    offsets = [0, 7, 14, 21, 28, 35, 40]
    seq, qs, mv = hostapi.stitch_chunks(offsets, [10] * 7, moves, ["ACGT"] * 7, ["!&.-"] * 7, 0, 1)
    assert seq == "ACGTCGCGTCGTCGTCCGT" and qs == "!&.-&.&.-&.-&.-&&.-" and len(mv) == 49
    # random chunkings against the oracle restatement
    rng = np.random.default_rng(9)
    for _ in range(40):
        stride = int(rng.choice([1, 5, 6]))
        cs, ov = 60 * stride, 6 * stride
        raw = int(rng.integers(cs // 3, 6 * cs))
        offs = O.generate_chunks(raw, cs, stride, ov)
        mvs, seqs, qss = [], [], []
        for _o in offs:
            m = (rng.random(cs // stride) < 0.4).astype(np.uint8)
            m[0] = 1
            nb = int(m.sum())
            mvs.append(m)
            seqs.append("".join(rng.choice(list("ACGT"), nb)))
            qss.append("".join(chr(int(c)) for c in rng.integers(34, 80, nb)))
        a = hostapi.stitch_chunks(offs, [cs] * len(offs), mvs, seqs, qss, raw, stride)
        b = O.stitch_chunks(offs, [cs] * len(offs), mvs, seqs, qss, raw, stride)
        assert a[0] == b[0] and a[1] == b[1] and (a[2] == b[2]).all()


def test_device_string_table_from_reference_tests():
    # /root/reference/tests/cuda_utils_test.cpp:51-80 (try_parse_device_ids), "cuda:" kept as alias
    table = [("cpu", 0, True, []), ("cpu", 1, True, []), ("cuda:all", 1, True, [0]),
             ("cuda:all", 0, False, []), ("cuda:all", 4, True, [0, 1, 2, 3]), ("cuda:2", 2, False, []),
             ("cuda:-1", 1, False, []), ("cuda:2", 3, True, [2]), ("cuda:2,0,3", 4, True, [0, 2, 3]),
             ("cuda:0,0", 4, False, []), ("cuda:0,1,2,1", 4, False, []), ("cuda:a", 4, False, []),
             ("cuda:a,0", 4, False, []), ("cuda:0,a", 4, False, []), ("cuda:1-3", 4, False, []),
             ("cuda:1.3", 4, False, [])]
    for s, n, ok, ids in table:
        for prefix in ("cuda:", "hip:"):
            ss = s.replace("cuda:", prefix)
            got_ok, got_ids = hostapi.parse_device_ids(ss, n)
            assert got_ok == ok, ss
            if ok:
                assert sorted(got_ids) == ids, ss


def test_config_matches_reference_config_tests():
    # /root/reference/tests/BasecallModelConfigTest.cpp (hac@v4.3.0 expectations) — parsed from the
    # same config.toml when the reference tree is present, else from the built-in mirror
    p = "/root/reference/tests/data/model_configs/dna_r10.4.1_e8.2_400bps_hac@v4.3.0"
    cfg = config.load_model_config(p) if os.path.isdir(p) else config.hac_v43()
    assert [(c.insize, c.size, c.winlen, c.stride, c.activation) for c in cfg.convs] == \
        [(1, 16, 5, 1, 0), (16, 16, 5, 1, 0), (16, 384, 19, 6, 2)]
    assert (cfg.lstm_size, cfg.lstm_layers, cfg.state_len, cfg.outsize, cfg.stride) == (384, 5, 4, 1024, 6)
    assert cfg.clamp and not cfg.bias and cfg.out_features is None
    assert abs(cfg.qscale - 1.1) < 1e-6 and abs(cfg.qbias + 1.1) < 1e-6
    assert (cfg.chunk_size, cfg.overlap) == (9996, 498)  # BatchParams.cpp:89-105 normalisation


def test_fast_config_matches_reference_toml():
    p = "/root/reference/tests/data/model_configs/dna_r10.4.1_e8.2_260bps_fast@v4.0.0"
    if not os.path.isdir(p):
        pytest.skip("reference tree not present")
    with pytest.raises(ValueError, match="Deprecated model"):      # BasecallModelConfigTest.cpp:159-163: refused by name
        config.load_model_config(p)
    a, b = config.load_model_config(p, allow_deprecated=True), config.fast_v40()
    assert [(c.insize, c.size, c.winlen, c.stride, c.activation) for c in a.convs] == \
        [(c.insize, c.size, c.winlen, c.stride, c.activation) for c in b.convs]
    assert (a.lstm_size, a.lstm_layers, a.state_len, a.clamp, a.out_features) == \
        (b.lstm_size, b.lstm_layers, b.state_len, b.clamp, b.out_features)
    assert abs(a.scale - b.scale) < 1e-6 and abs(a.qscale - b.qscale) < 1e-6 and abs(a.qbias - b.qbias) < 1e-6


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 1000):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = dist.shard_range(n, r, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(n))


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from dorado_amd import dist as D
rank, local, world = D.env_world()
dist.init_process_group(backend="gloo")
lo, hi = D.shard_range(101, rank, world)
tot = D.sum_over_ranks(hi - lo)
mx = D.max_over_ranks(1.0 + rank)
dist.barrier()
assert tot == 101 and mx == float(world), (tot, mx)
if rank == 0:
    print("OK", world, tot, mx)
dist.destroy_process_group()
"""


def test_two_process_gloo_replicas(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29517", str(script), ROOT]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OK 2 101.0 2.0" in out.stdout


# ---------------------------------------------------------------- ScalerNode host half
def test_scaler_host_pa_formula_matches_oracle():
    """ScalerNode.cpp:186-227: host mirror == oracle restatement, incl. the open-pore adjustment table."""
    from oracle import oracle_py as O

    for std, mean, stdev, scaling, offset, opl, fc, expected in [
        (1, 93.69, 23.5, 0.1755, -243.0, float("nan"), "", None),
        (1, 93.69, 23.5, 0.1755, -243.0, 205.0, "FLO-PRO114M", 199.21),
        (0, 0.0, 1.0, 0.2, 12.0, 190.0, "flo-min114", 197.61),      # case-insensitive lookup
        (1, 90.0, 20.0, 0.15, -250.0, 201.0, "FLO-UNKNOWN", None),
    ]:
        got = hostapi.pa_read_scaling(std, mean, stdev, scaling, offset, opl, fc)
        sh, sc, adj = O.pa_shift_scale(scaling, offset, std, mean, stdev, opl, expected or 0.0)
        assert got["shift"] == np.float32(sh) and got["scale"] == np.float32(sc)
        assert got["open_pore_adjustment"] == np.float32(adj)
        assert got["scale_pa"] == np.float32(np.float32(scaling) * np.float32(sc))
        assert got["shift_pa"] == np.float32(np.float32(scaling) * (np.float32(sh) + np.float32(offset)))


def test_scaler_host_trim_known_answers():
    """tests/TrimTest.cpp:31-93 through the host mirror (operating on the f16 signal the pipeline holds)."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scaler.npz"))
    sig = g["trim_signal"].astype(np.float16)
    from oracle import oracle_py as O

    for args in [(2.4, 40, 3), (2.4, 10, 3), (24.0, 40, 3)]:
        assert hostapi.trim_signal(sig, *args) == O.trim(sig.astype(np.float32), *args)
    assert hostapi.trim_signal(sig) == 90
    assert hostapi.trim_signal(sig, 2.4, 10, 3) == 60
    assert hostapi.trim_signal(sig, 24.0, 40, 3) == 10
    assert hostapi.trim_signal(np.full(2000, 100.0, np.float16), 24.0, 40, 3) == 10
    assert hostapi.dna_trim_start(True, sig) == 10                      # standardised models: constant
    assert hostapi.dna_trim_start(False, sig) == 90
    assert hostapi.dna_trim_start(True, sig[:8]) == 0                   # trim would swallow the read


def test_simplex_chunk_sizes_and_queue_routing():
    """CudaCaller.cpp:382-413 batch dimensions and BasecallerNode.cpp:81-94 routing (restated; known answers worked
    from the reference code: hac 9996 -> {9996, 4998}; sup@v5 12288 -> {12288, 6144} with granularity 12*16)."""
    from dorado_amd import config
    hac, s5 = config.hac_v43(), config.sup_v50()
    assert hostapi.model_stride(hac) == 6 and hostapi.model_stride(s5) == 6
    assert hostapi.simplex_chunk_sizes(hac, 9996, 498) == [9996, 4998]
    assert hostapi.simplex_chunk_sizes(hac, 10000, 498) == [9996, 4998]       # x / 6 * 6
    assert hostapi.simplex_chunk_sizes(s5, 12288, 600) == [12288, 6144]
    assert hostapi.simplex_chunk_sizes(s5, 12000, 600) == [11904, 5952]       # granularity 192
    assert hostapi.simplex_chunk_sizes(hac, 600, 498) == [600, 504]           # never below pad_to(overlap + 1)
    assert hostapi.simplex_chunk_sizes(hac, 504, 498) == [504]                # duplicates collapse
    q = hostapi.get_chunk_queue_idx
    sizes = [9996, 4998]
    # smallest size STRICTLY larger than the read (the reference compares with <), else the largest
    assert q(sizes, 100) == 1 and q(sizes, 4997) == 1 and q(sizes, 4998) == 0 and q(sizes, 4999) == 0
    assert q(sizes, 9996) == 0 and q(sizes, 50000) == 0                           # else the largest
    assert q([4998, 9996], 50000) == 1 and q([4998, 9996], 10) == 0
    assert q([9996], 5) == 0


def test_fused_layer_kernel_has_no_unpadded_asm_mfma_hazard():
    """csrc/txlayer.hip pins two accumulator tiles to VGPRs with inline-asm MFMAs, which hipcc's hazard recogniser does not
    see.  The one hazard that can arise around them — a VALU-written register (a compiler v_mov / v_accvgpr_read) read as an
    MFMA operand within two states — must not occur in the compiled product kernel: scan the gfx950 ISA (no GPU needed)."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_asm_hazards.py")
    out = subprocess.run([sys.executable, tool], capture_output=True, text=True)
    product = [l for l in out.stdout.splitlines() if "tx_layer_kernelILi3ELi0E" in l]
    assert not product, "\n".join(product)
    assert "VALU -> asm-MFMA operand hazards" in out.stdout, out.stdout + out.stderr


def test_asm_hazard_checker_detects_and_clears(tmp_path):
    """The gate itself: a VALU write one / two states in front of an asm MFMA that reads the register is reported, three
    states (or an s_nop 1 in between) is not; loads and builtin (non-asm) MFMAs are not its business."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def listing(body):
        p = tmp_path / "k.s"
        p.write_text("_Z1kv:\n" + body)
        return mod.scan(str(p))

    mf = "\t;;#ASMSTART\n\tv_mfma_f32_32x32x16_f16 v[0:15], v[20:23], v[24:27], v[0:15]\n\t;;#ASMEND\n"
    assert len(listing("\tv_mov_b32_e32 v21, v40\n" + mf)) == 1                       # one state
    assert len(listing("\tv_accvgpr_read_b32 v24, a3\n\ts_add_i32 s1, s1, 1\n" + mf)) == 1   # two states
    assert len(listing("\tv_permlane32_swap_b32_e32 v50, v26\n" + mf)) == 1           # the swap writes its source too
    assert listing("\tv_mov_b32_e32 v21, v40\n\ts_nop 1\n" + mf) == []              # padded
    assert listing("\tv_mov_b32_e32 v21, v40\n\ts_add_i32 s1, s1, 1\n\ts_add_i32 s2, s2, 1\n" + mf) == []
    assert listing("\tv_mov_b32_e32 v99, v40\n" + mf) == []                           # unrelated register
    assert listing("\tds_read_b128 v[20:23], v60\n\ts_waitcnt lgkmcnt(0)\n" + mf) == []   # a load, waited for
    assert listing("\tv_mov_b32_e32 v21, v40\n\tv_mfma_f32_32x32x16_f16 a[0:15], v[20:23], v[24:27], a[0:15]\n") == []


def test_decoder_inline_asm_has_no_unpadded_valu_hazard():
    """csrc/decode.hip holds inline-asm reductions (v_max_f32_dpp / v_add_f32_dpp chains) and the beam search's match bits
    (v_cmp into an SGPR pair + v_addc reading it as carry-in), none of which hipcc's hazard recogniser sees.  Scan the gfx950
    ISA of the whole translation unit for the four VALU -> VALU wait-state rules they are subject to (no GPU needed)."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "check_asm_hazards.py")
    out = subprocess.run([sys.executable, tool, "--valu"], capture_output=True, text=True)
    assert out.returncode == 0 and "0 VALU -> VALU hazards around inline asm" in out.stdout, out.stdout + out.stderr


def test_valu_hazard_checker_detects_and_clears(tmp_path):
    """The gate itself, on hand-written listings: each rule fires at too short a distance and clears with padding."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)

    def listing(body):
        p = tmp_path / "k.s"
        p.write_text("_Z1kv:\n" + body)
        return [b[4] for b in mod.scan_valu(str(p))]

    A, E = "\t;;#ASMSTART\n", "\t;;#ASMEND\n"
    dpp = "\tv_max_f32_dpp v9, v9, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
    # DPP: compiler VALU write, asm DPP read
    assert listing("\tv_max_f32_e32 v9, v1, v2\n" + A + dpp + E) == ["VALU write -> DPP read"]
    assert listing("\tv_max_f32_e32 v9, v1, v2\n" + A + "\ts_nop 0\n" + dpp + E) == ["VALU write -> DPP read"]
    assert listing("\tv_max_f32_e32 v9, v1, v2\n" + A + "\ts_nop 1\n" + dpp + E) == []
    assert listing(A + dpp + dpp + E) == ["VALU write -> DPP read"]                       # two chained stages, unpadded
    assert listing(A + dpp + "\ts_nop 1\n" + dpp + E) == []
    # SGPR mask: v_cmp then v_addc reading it
    cmp_ = "\tv_cmp_eq_u32_e64 s[12:13], v41, v10\n"
    addc = "\tv_addc_co_u32_e64 v11, s[12:13], v11, v11, s[12:13]\n"
    other = "\tv_cmp_eq_u32_e64 s[14:15], v40, v10\n"
    assert listing(A + cmp_ + addc + E) == ["VALU write of an SGPR mask -> carry-in / select read"]
    assert listing(A + cmp_ + other + addc + E) == ["VALU write of an SGPR mask -> carry-in / select read"]
    assert listing(A + cmp_ + other + other + addc + E) == []
    assert listing(A + "\tv_cmp_eq_u32_e32 vcc, v1, v2\n\tv_addc_co_u32_e32 v3, vcc, v3, v3, vcc\n" + E) != []
    # asm VALU write, compiler swap / readlane behind it
    w = A + "\tv_add_f32_dpp v5, v5, v5 row_mirror row_mask:0xf bank_mask:0xf\n" + E
    assert listing("\ts_nop 1\n" + w + "\tv_permlane32_swap_b32_e32 v5, v6\n") == ["VALU write -> v_permlane swap read"]
    assert listing("\ts_nop 1\n" + w + "\tv_readlane_b32 s3, v5, 63\n") == ["VALU write -> v_readlane read"]
    w1 = A + "\tv_add_f32_dpp v5, v5, v5 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 1\n" + E
    assert listing("\ts_nop 1\n" + w1 + "\tv_permlane32_swap_b32_e32 v5, v6\n") == []
    assert listing("\ts_nop 1\n" + w1 + "\tv_readlane_b32 s3, v5, 63\n") == []
    # compiler-to-compiler pairs are not its business
    assert listing("\tv_max_f32_e32 v9, v1, v2\n\tv_mov_b32_dpp v3, v9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n") == []
    # a label starts a new basic block: the producer may sit on the taken edge, so the textual predecessors do not
    # clear an asm consumer in the first wait states of a block — padding inside the block does
    far = "\ts_add_i32 s1, s1, 1\n" * 3
    assert listing("\tv_max_f32_e32 v9, v1, v2\n" + far + ".LBB0_7:\n" + A + dpp + E) == ["VALU write -> DPP read"]
    assert listing("\tv_max_f32_e32 v9, v1, v2\n" + far + ".LBB0_7:\n" + A + "\ts_nop 1\n" + dpp + E) == []
    assert listing(".LBB0_7:\n\ts_add_i32 s1, s1, 1\n\ts_add_i32 s2, s2, 1\n" + A + dpp + E) == []
    assert listing(".LBB0_8:\n" + A + addc + E) == ["VALU write of an SGPR mask -> carry-in / select read"]


def test_fused_layer_weight_image_layout():
    """csrc/txlayer.hip streams one weight image per layer in consumption order (host function tx_layer_image, no device):
    (16 + 3 FF/32) stages of 32 KB; 16 x Wo | W1(0) | W1(1) W2(0) | ... | W2(NJ-1); every weight exactly once."""
    import ctypes as C
    from dorado_amd import capi
    L = capi.dbg_lib()
    L.mibc_debug_tx_layer_image.restype = C.c_long
    L.mibc_debug_tx_layer_image.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_long]
    FF, D = 256, 512
    NJ = FF // 32
    rng = np.random.default_rng(5)
    wo = rng.standard_normal((D, D)).astype(np.float32)
    w1 = rng.standard_normal((2 * FF, D)).astype(np.float32)
    w2 = rng.standard_normal((D, FF)).astype(np.float32)
    w1[0:32] = 7.0            # y rows of slab 0
    w1[FF:FF + 32] = 9.0      # gate rows of slab 0
    w2[:, FF - 32:] = 11.0    # slab NJ - 1 of W2
    n = L.mibc_debug_tx_layer_image(wo.ctypes.data, w1.ctypes.data, w2.ctypes.data, FF, None, 0)
    assert n == (16 + 3 * NJ) * 16384
    img = np.zeros(n, np.uint16)
    assert L.mibc_debug_tx_layer_image(wo.ctypes.data, w1.ctypes.data, w2.ctypes.data, FF, img.ctypes.data, n) == n
    img = img.view(np.float16).reshape(16 + 3 * NJ, 32, 64, 8)       # [stage][fragment][lane][e]
    want = np.sort(np.concatenate([wo.ravel(), w1.ravel(), w2.ravel()]).astype(np.float16))
    assert (np.sort(img.ravel()) == want).all()                      # a permutation of the f16-rounded weights
    # Wo: stage st, fragment (s, c), lane, e  ->  Wo[32 c + l31][32 st + 16 s + 8 lhi + e]
    l = np.arange(64)
    for st, s, c in ((0, 0, 0), (5, 1, 9), (15, 1, 15)):
        k0 = 32 * st + 16 * s
        ref = np.stack([wo[32 * c + (ln & 31), k0 + 8 * (ln >> 5):k0 + 8 * (ln >> 5) + 8] for ln in l])
        assert (img[st, s * 16 + c] == ref.astype(np.float16)).all()
    # stages 16, 17 = W1 of slab 0: fragments alternate y (7) / gate (9)
    assert (img[16:18, 0::2] == np.float16(7.0)).all() and (img[16:18, 1::2] == np.float16(9.0)).all()
    assert (img[18:20] != np.float16(7.0)).all()                     # stage 18, 19: W1 of slab 1
    # W2(0) follows W1(1); the last stage is W2(NJ - 1)
    assert (img[-1] == np.float16(11.0)).all() and (img[-2] != np.float16(11.0)).all()
    # W2(j) fragment (s, c): k index permuted inside a 16-group: 4 lhi + (e & 3) + 8 (e >> 2)
    j, s, c = 0, 1, 3
    ref = np.stack([[w2[32 * c + (ln & 31), 32 * j + 16 * s + 4 * (ln >> 5) + (e & 3) + 8 * (e >> 2)] for e in range(8)] for ln in l])
    assert (img[20, s * 16 + c] == ref.astype(np.float16)).all()     # stage 20 = W2(0)


def test_variable_chunk_row_packer_properties():
    """HipModelRunner's placement of variable-length chunks (the role of CudaModelRunner.cpp:21-32 + the step budget of
    BasecallerNode.cpp:408-430), CPU only: chunks never overlap, at least `gap` samples (2 output steps) lie between the
    chunks of a row, no chunk crosses a row end, placement order is accept order, and once a chunk does not fit every later
    one is deferred too (the second engine batch keeps result order trivial)."""
    import ctypes as C
    from dorado_amd import hostapi
    L = hostapi.lib()
    L.mibch_debug_pack_rows.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    stride, cs = 6, 9996
    gap = 2 * stride
    rng = np.random.default_rng(9)
    for rows, n in ((64, 300), (8, 40), (4, 4), (1, 3)):
        lens = (rng.integers(2, cs // stride + 1, n) * stride).astype(np.int32)
        lens[0] = cs                                    # a full-size chunk owns its row
        out_row = np.zeros(n, np.int32)
        out_start = np.zeros(n, np.int32)
        placed = L.mibch_debug_pack_rows(lens.ctypes.data, n, rows, cs, gap, out_row.ctypes.data, out_start.ctypes.data)
        ok = out_row >= 0
        assert placed == int(ok.sum()) and placed >= 1
        first_bad = int(np.argmin(ok)) if not ok.all() else n
        assert ok[:first_bad].all() and not ok[first_bad:].any()          # a prefix is placed, the rest deferred
        assert out_row[0] == 0 and out_start[0] == 0
        for r in range(rows):
            idx = np.nonzero(out_row[:first_bad] == r)[0]
            spans = sorted((int(out_start[i]), int(out_start[i] + lens[i])) for i in idx)
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert b0 - a1 >= gap                                      # >= 2 output steps between chunks of a row
            assert all(a1 <= cs and a0 % stride == 0 for a0, a1 in spans)
        # first fit over ALL rows: when chunk i was placed, no row in front of its own had room for it
        fill = np.zeros(rows, np.int64)
        for i in range(first_bad):
            for r in range(int(out_row[i])):
                assert fill[r] + lens[i] > cs
            assert out_start[i] == fill[out_row[i]]
            fill[out_row[i]] = out_start[i] + lens[i] + gap
        if first_bad < n:                                                  # it really did not fit anywhere
            assert (fill + lens[first_bad] > cs).all()


def test_host_layer_under_sanitizers():
    """Race / memory checking of the host layer without a GPU (the reference runs TSan / ASan builds in CI):
    tools/sanitize_host.sh builds dorado_amd/host/*.cpp against the C-ABI test double (tools/fake_mibc.cpp) with -fsanitize=thread
    and with -fsanitize=address,undefined (+ leak check) and runs (a) the node over stand-in runners (8 worker threads, 9000 reads,
    both chunk-size queues, the parallel chunking path), scaler_node from eight threads, the row packer; (b) the whole layer:
    HipCaller's GPU thread, device FIFO and two asynchronous slots, variable / raw int16 batches, two devices, scaler_node beside
    the node — every leg cross-checked against a direct evaluation.  Any report or failed check fails the script."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "tools", "sanitize_host.sh")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitize_host: thread clean" in r.stdout and "sanitize_host: address clean" in r.stdout


REF_CONFIGS = "/root/reference/tests/data/model_configs"
needs_ref_configs = pytest.mark.skipif(not os.path.isdir(REF_CONFIGS), reason="reference tree not present")


@needs_ref_configs
def test_tx_config_matches_reference_config_test():
    """BasecallModelConfigTest.cpp:47-152 (sup@v5.0.0 transformer model load) and :21-45 (normalise BatchParams)."""
    cfg = config.load_model_config(os.path.join(REF_CONFIGS, "dna_r10.4.1_e8.2_400bps_sup@v5.0.0"))
    assert cfg.is_tx and cfg.bias is True and cfg.num_features == 1 and cfg.stride == 6 and cfg.scale == 1.0
    assert cfg.state_len == 5 and cfg.outsize == 4096 and cfg.clamp is False and cfg.out_features == 4096
    assert cfg.sample_type == "DNA" and cfg.qbias == 0.0 and cfg.qscale == 1.0 and cfg.sample_rate == 5000
    assert cfg.stride * cfg.tx.up_scale_factor == 12 and cfg.tx.up_scale_factor == 2          # stride_inner, scale_factor
    sn = cfg.signal_norm
    assert (sn.strategy, sn.standardise) == ("pa", True)
    assert np.float32(sn.mean) == np.float32(93.6376) and np.float32(sn.stdev) == np.float32(22.6004)
    assert (sn.quantile_a, sn.quantile_b, sn.shift_multiplier, sn.scale_multiplier) == (0.2, 0.9, 0.51, 0.53)
    assert [(c.insize, c.size, c.winlen, c.stride, c.activation) for c in cfg.convs] == \
        [(1, 64, 5, 1, 0), (64, 64, 5, 1, 0), (64, 128, 9, 3, 0), (128, 128, 9, 2, 0), (128, 512, 5, 2, 0)]
    t = cfg.tx
    assert (t.d_model, t.depth, t.dim_feedforward, t.attn_window, t.nhead) == (512, 18, 2048, (127, 128), 8)
    assert (t.crf_insize, t.crf_n_base, t.crf_blank_score, t.crf_scale, t.crf_expand_blanks) == (512, 4, 2.0, 5.0, True)
    assert (t.up_scale_factor, t.up_size) == (2, 512)
    assert (cfg.chunk_size, cfg.overlap) == (12288, 600) and cfg.has_normalised_basecaller_params()
    # the engine's own description of the same model (what bench.py runs) agrees with the parsed file
    ref = config.sup_v50()
    assert cfg.to_desc().tx_d_model == ref.to_desc().tx_d_model and cfg.n_weights() == ref.n_weights()
    for f in ("tx_nhead", "tx_depth", "tx_dim_ff", "tx_win_upper", "tx_win_lower", "up_scale_factor", "state_len", "outsize"):
        assert getattr(cfg.to_desc(), f) == getattr(ref.to_desc(), f), f
    # :30-45 known non-normalised chunk size
    cfg.chunk_size = 1921
    assert not cfg.has_normalised_basecaller_params()
    cfg.normalise_basecaller_params()
    assert cfg.has_normalised_basecaller_params() and cfg.chunk_size == 1920


@needs_ref_configs
def test_lstm_configs_match_reference_config_tests():
    """BasecallModelConfigTest.cpp:165-224 (hac@v4.2.0), :226-283 (hac@v4.3.0 pa), :285-343 (hac@v4.3.0 quantile),
    :350-408 (rna004 sup@v3.0.1, a pre-v4 flat encoder table), :410-423 (sample_type), :154-163 / :345-348 (deprecated)."""
    c = config.load_model_config(os.path.join(REF_CONFIGS, "dna_r10.4.1_e8.2_400bps_hac@v4.2.0"))
    assert (c.bias, c.stride, c.lstm_size, c.blank_score, c.scale, c.state_len, c.outsize, c.clamp, c.out_features) == \
        (True, 6, 384, 2.0, 1.0, 4, 1024, True, 128)
    assert np.float32(c.qbias) == np.float32(-0.2) and np.float32(c.qscale) == np.float32(0.95) and c.sample_rate == 5000
    assert c.signal_norm.strategy == "quantile" and not c.signal_norm.standardise and c.sample_type == "DNA"
    assert [(v.insize, v.size, v.winlen, v.stride, v.activation) for v in c.convs] == \
        [(1, 16, 5, 1, 1), (16, 16, 5, 1, 1), (16, 384, 19, 6, 1)]            # swish_clamp: every conv is followed by a clamp

    c = config.load_model_config(os.path.join(REF_CONFIGS, "dna_r10.4.1_e8.2_400bps_hac@v4.3.0"))
    assert (c.bias, c.lstm_size, c.out_features, c.clamp) == (False, 384, None, True)
    sn = c.signal_norm
    assert (sn.strategy, sn.standardise) == ("pa", True) and np.float32(sn.mean) == np.float32(91.88) and np.float32(sn.stdev) == np.float32(22.65)

    c = config.load_model_config(os.path.join(REF_CONFIGS, "dna_r10.4.1_e8.2_400bps_hac@v4.3.0_quantile"))
    assert (c.qbias, c.qscale, c.sample_rate) == (0.0, 1.0, -1) and c.signal_norm.strategy == "quantile" and not c.signal_norm.standardise

    c = config.load_model_config(os.path.join(REF_CONFIGS, "rna004_130bps_sup@v3.0.1"))
    assert (c.bias, c.num_features, c.stride, c.lstm_size, c.blank_score, c.scale, c.state_len, c.outsize, c.clamp, c.out_features) == \
        (True, 1, 5, 768, 2.0, 5.0, 5, 4096, False, None)
    assert c.sample_type == "RNA004" and c.sample_rate == 4000
    assert np.float32(c.qbias) == np.float32(-0.1) and np.float32(c.qscale) == np.float32(0.9)
    sn = c.signal_norm
    assert sn.strategy == "quantile" and [np.float32(v) for v in (sn.quantile_a, sn.quantile_b, sn.scale_multiplier, sn.shift_multiplier)] == \
        [np.float32(v) for v in (0.22, 0.88, 0.595, 0.485)]
    assert [(v.insize, v.size, v.winlen, v.stride, v.activation) for v in c.convs] == [(1, 4, 5, 1, 0), (4, 16, 5, 1, 0), (16, 768, 19, 5, 0)]

    assert config.load_model_config(os.path.join(REF_CONFIGS, "sample_type_d_e8.2_400bps_sup@v5.0.0")).sample_type == "DNA"
    assert config.load_model_config(os.path.join(REF_CONFIGS, "sample_type_130bps_sup@v3.0.1")).sample_type == "RNA004"
    for dep in ("dna_r9.4.1_e8_hac@v3.3", "dna_r10.4.1_e8.2_260bps_fast@v4.0.0", "rna002_70bps_fast@v3"):
        with pytest.raises(ValueError, match="Deprecated model"):
            config.load_model_config(os.path.join(REF_CONFIGS, dep))


def test_config_loader_errors(tmp_path):
    """The checks of parse_signal_normalisation_params (:176-193), parse_tx_encoder_params (:383-391), parse_convs
    (common.cpp:53-87), load_lstm_model_config (:305-314) and parse_run_info (:131-138)."""
    base = """[input]\nfeatures = 1\n[global_norm]\nstate_len = 3\n[run_info]\nsample_rate = 5000\n[encoder]\nstride = 5\nfeatures = 96\nscale = 5.0\nblank_score = 2.0\n"""

    def write(name, text):
        d = tmp_path / name
        d.mkdir()
        (d / "config.toml").write_text(text)
        return str(d)
    assert config.load_model_config(write("dna_ok@v1", base)).lstm_size == 96
    with pytest.raises(ValueError, match="sample type"):
        config.load_model_config(write("mystery@v1", base))                               # neither run_info.sample_type nor the name
    assert config.load_model_config(write("mystery@v2", base.replace("sample_rate = 5000", 'sample_rate = 5000\nsample_type = "rna004"'))).sample_type == "RNA004"
    with pytest.raises(ValueError, match="only for `scaling.strategy = pa`"):
        config.load_model_config(write("dna_a@v1", base + "[standardisation]\nstandardise = 1\nmean = 90.0\nstdev = 20.0\n"))
    with pytest.raises(ValueError, match="must be greater than 0"):
        config.load_model_config(write("dna_b@v1", base + '[scaling]\nstrategy = "pa"\n[standardisation]\nstandardise = 1\nmean = 90.0\nstdev = 0.0\n'))
    with pytest.raises(ValueError, match="Unknown scaling strategy"):
        config.load_model_config(write("dna_c@v1", base + '[scaling]\nstrategy = "zscore"\n'))
    with pytest.raises(ValueError, match="first convolution layer must be size 4 or 16"):
        config.load_model_config(write("dna_d@v1", base.replace("blank_score = 2.0", "blank_score = 2.0\nfirst_conv_size = 8")))
