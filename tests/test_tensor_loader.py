"""libtorch-free ".tensor" loader (dorado_amd/host/tensor_loader.cpp) against torch's own
reader — the reference loads these files with torch::load (torch_utils/tensor_utils.cpp:153-163) — on the
reference's fixtures and on model directories written here with torch.jit in the same archive format,
named as basecall/crf_utils.cpp:26-150 expects."""
import json
import os
import zipfile
import zlib

import numpy as np
import pytest
import torch

from dorado_amd import capi, config, hostapi, synth

HERE = os.path.dirname(os.path.abspath(__file__))
GT = os.path.join(HERE, "golden", "tensor")


def _save_tensor(path, t):
    """One tensor as a TorchScript module attribute "0" — the layout of the reference's fixtures."""
    m = torch.nn.Module()
    m.register_parameter("0", torch.nn.Parameter(t, requires_grad=False))
    torch.jit.script(m).save(str(path))


def test_reference_fixtures_bit_exact():
    exp = json.load(open(os.path.join(GT, "expected.json")))
    for name, e in exp.items():
        (n, a), = hostapi.load_tensor_file(os.path.join(GT, name + ".tensor"))
        assert n == "0" and str(a.dtype) == e["dtype"] and list(a.shape) == e["shape"]
        assert zlib.crc32(a.tobytes()) == e["crc32"]
        t = dict(torch.jit.load(os.path.join(GT, name + ".tensor")).named_parameters())["0"].detach().numpy()
        assert (a.view(np.uint8) == t.view(np.uint8)).all()


def test_dtypes_views_and_multi_tensor(tmp_path):
    g = torch.Generator().manual_seed(1)
    base = torch.randn(7, 12, 5, generator=g)
    cases = {
        "f32": base.clone(), "f16": base.half(), "bf16": base.bfloat16(), "f64": base.double(),
        "i16": (base * 100).short(), "i32": (base * 1000).int(), "i64": (base * 1000).long(),
        "u8": (base.abs() * 20).byte(), "i8": (base * 20).char(), "bool": base > 0,
        "view_offset": base.flatten()[17:17 + 60].reshape(6, 10),          # storage offset
        "transposed": base.permute(2, 0, 1),                               # non-contiguous strides
        "scalar_like": torch.tensor([3.5]), "empty": torch.zeros(0, 4),
    }
    for k, t in cases.items():
        p = tmp_path / f"{k}.tensor"
        _save_tensor(p, t)
        (n, a), = hostapi.load_tensor_file(p)
        want = dict(torch.jit.load(str(p)).named_parameters())["0"].detach()
        assert tuple(a.shape) == tuple(want.shape), k
        if k == "bf16":
            assert (a == want.float().numpy()).all()
        else:
            assert a.dtype == want.numpy().dtype and (a == want.numpy()).all(), k
        af = hostapi.load_tensor_file(p, as_float=True)[0][1]
        assert np.allclose(af, want.float().numpy(), rtol=0, atol=0) or k in ("f64", "i64")
    # several tensors in one archive, in attribute order
    m = torch.nn.Module()
    for i in range(3):
        m.register_parameter(str(i), torch.nn.Parameter(torch.full((2, i + 1), float(i)), requires_grad=False))
    torch.jit.script(m).save(str(tmp_path / "multi.tensor"))
    got = hostapi.load_tensor_file(tmp_path / "multi.tensor")
    assert [n for n, _ in got] == ["0", "1", "2"] and [a.shape for _, a in got] == [(2, 1), (2, 2), (2, 3)]


def test_errors_are_loud(tmp_path):
    with pytest.raises(ValueError, match="cannot open"):
        hostapi.load_tensor_file(tmp_path / "missing.tensor")
    (tmp_path / "junk.tensor").write_bytes(b"\x00" * 100)
    with pytest.raises(ValueError, match="not a zip"):
        hostapi.load_tensor_file(tmp_path / "junk.tensor")
    _save_tensor(tmp_path / "ok.tensor", torch.arange(1000.0))
    raw = (tmp_path / "ok.tensor").read_bytes()
    (tmp_path / "cut.tensor").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(ValueError):
        hostapi.load_tensor_file(tmp_path / "cut.tensor")
    with zipfile.ZipFile(tmp_path / "deflated.tensor", "w", zipfile.ZIP_DEFLATED) as z:
        with zipfile.ZipFile(tmp_path / "ok.tensor") as src:
            for info in src.infolist():
                z.writestr(info.filename, src.read(info.filename))
    with pytest.raises(ValueError, match="compressed"):
        hostapi.load_tensor_file(tmp_path / "deflated.tensor")


def test_malformed_archives_never_crash(tmp_path):
    """File-format parser hygiene: byte flips, truncations and hostile pickles (empty-stack BINPUT, stride-0
    blow-up, offsets beyond the file) either load or raise ValueError — never touch memory out of bounds."""
    _save_tensor(tmp_path / "ok.tensor", torch.arange(64, dtype=torch.float16).reshape(4, 16))
    raw = bytearray((tmp_path / "ok.tensor").read_bytes())
    rng = np.random.default_rng(7)
    bad = tmp_path / "bad.tensor"
    outcomes = {"ok": 0, "err": 0}
    eocd = raw.rfind(b"PK\x05\x06")
    cd = int.from_bytes(raw[eocd + 16:eocd + 20], "little")
    pk = raw.find(b"data.pkl") + len("data.pkl")
    regions = [(cd, len(raw)), (pk, pk + 400), (0, len(raw))]
    for trial in range(600):
        b = bytearray(raw)
        lo, hi = regions[trial % 3]
        for _ in range(1 + trial % 4):
            b[int(rng.integers(lo, min(hi, len(b))))] = int(rng.integers(0, 256))
        if trial % 10 == 9:
            b = b[:int(rng.integers(1, len(b)))]
        bad.write_bytes(bytes(b))
        try:
            hostapi.load_tensor_file(bad)
            outcomes["ok"] += 1
        except ValueError:
            outcomes["err"] += 1
    assert outcomes["err"] > 100 and outcomes["ok"] > 0, outcomes

    def archive(pickle_bytes, storage=b"\x00" * 128):
        with zipfile.ZipFile(bad, "w", zipfile.ZIP_STORED) as z:
            z.writestr("m/data.pkl", pickle_bytes)
            z.writestr("m/data/0", storage)

    archive(b"\x80\x02q\x00.")                       # BINPUT on an empty stack
    with pytest.raises(ValueError, match="underflow"):
        hostapi.load_tensor_file(bad)
    archive(b"\x80\x02b.")                            # BUILD on an empty stack
    with pytest.raises(ValueError, match="underflow"):
        hostapi.load_tensor_file(bad)

    def rebuild(shape, stride, offset=0):
        def tup(v):
            return b"(" + b"".join(b"J" + int(x).to_bytes(4, "little", signed=True) for x in v) + b"t"
        return (b"\x80\x02ctorch._utils\n_rebuild_tensor_v2\n((X\x07\x00\x00\x00storagectorch\nFloatStorage\n"
                b"X\x01\x00\x00\x000X\x03\x00\x00\x00cpuK\x20tQJ" + int(offset).to_bytes(4, "little", signed=True)
                + tup(shape) + tup(stride) + b"\x89tR.")

    archive(rebuild((4, 8), (8, 1)))
    (_, a), = hostapi.load_tensor_file(bad)
    assert a.shape == (4, 8)
    for shape, stride, off in [((2 ** 31 - 1, 2 ** 31 - 1), (0, 0), 0),      # stride-0 view: numel overflow / giant resize
                               ((4, 8), (8, 1), 1),                          # one element past the storage
                               ((4, 8), (2 ** 31 - 1, 2 ** 31 - 1), 0),      # index overflow
                               ((100000,), (0,), 0)]:                        # 100000x expansion of a 32-element storage
        archive(rebuild(shape, stride, off))
        with pytest.raises(ValueError):
            hostapi.load_tensor_file(bad)


def test_model_tensor_names_follow_the_reference():
    """basecall/crf_utils.cpp:26-88 for hac@v4.3.0 (3 convs, 5 LSTMs, no bias, no decomposition) and
    :90-150 for sup@v5.0.0 (5 convs, 18 encoder layers)."""
    names = hostapi.model_tensor_names(config.hac_v43())
    assert names[:6] == ["0.conv.weight.tensor", "0.conv.bias.tensor", "1.conv.weight.tensor", "1.conv.bias.tensor",
                         "2.conv.weight.tensor", "2.conv.bias.tensor"]
    assert names[6:10] == ["4.rnn.weight_ih_l0.tensor", "4.rnn.weight_hh_l0.tensor", "4.rnn.bias_ih_l0.tensor",
                           "4.rnn.bias_hh_l0.tensor"]
    assert names[-1] == "9.linear.weight.tensor" and len(names) == 27
    tx = hostapi.model_tensor_names(config.sup_v50())
    assert tx[0] == "conv.0.conv.weight.tensor" and tx[10] == "transformer_encoder.0.self_attn.Wqkv.weight.tensor"
    assert tx[-3:] == ["upsample.linear.weight.tensor", "upsample.linear.bias.tensor", "crf.linear.weight.tensor"]
    assert len(tx) == 10 + 18 * 7 + 3


def _write_model_dir(d, cfg, ws, dtype=torch.float16):
    os.makedirs(d, exist_ok=True)
    for name, w in zip(hostapi.model_tensor_names(cfg), ws):
        _save_tensor(os.path.join(d, name), torch.from_numpy(np.ascontiguousarray(w)).to(dtype))


@pytest.mark.parametrize("which", ["lstm", "tx"])
def test_model_directory_round_trip(tmp_path, which):
    cfg = config.tiny(128, 4) if which == "lstm" else config.tiny_tx()
    ws = synth.make_weights(cfg, seed=9)
    _write_model_dir(tmp_path / "model", cfg, ws)
    got = hostapi.load_model_weights(tmp_path / "model", cfg)
    assert len(got) == len(ws)
    for g, w in zip(got, ws):
        assert g.dtype == np.float32 and g.shape == w.shape
        assert (g == w.astype(np.float16).astype(np.float32)).all()     # files hold f16, as ONT's models do


@pytest.mark.gpu
def test_engine_from_model_directory(tmp_path):
    """Model directory -> weights through the libtorch-free loader -> engine == engine built from the
    in-memory (f16-rounded) weights, bit for bit."""
    cfg = config.tiny(128, 4)
    ws = [w.astype(np.float16).astype(np.float32) for w in synth.make_weights(cfg, seed=10)]
    _write_model_dir(tmp_path / "model", cfg, ws)
    loaded = hostapi.load_model_weights(tmp_path / "model", cfg)
    x = synth.make_signal(64, 1200, seed=3)
    a = capi.Engine(cfg, ws)
    b = capi.Engine(cfg, loaded)
    assert (a.forward(x).view(np.uint16) == b.forward(x).view(np.uint16)).all()
    a.close()
    b.close()
