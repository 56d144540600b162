"""Host-side mirror of the reference's model configuration for the hot path.

Follows dorado/config/BasecallModelConfig.cpp:214-323 (load_lstm_model_config),
dorado/config/common.cpp:53-87 (conv parsing, swish->swish_clamp when followed by a clamp),
dorado/config/BatchParams.cpp:89-105 (normalise) and
dorado/utils/include/utils/parameters.h:7-15 (defaults: chunk 10000, overlap 500).
"""
from __future__ import annotations

import ctypes
import dataclasses
import os
from typing import List, Optional

ACT_SWISH, ACT_SWISH_CLAMP, ACT_TANH = 0, 1, 2  # config/common.h Activation order

DEFAULT_CHUNK_SIZE = 10000
DEFAULT_OVERLAP = 500


class ModelDescC(ctypes.Structure):
    """C layout of `mibc_model_desc` (include/mibc.h). oracle/ restates the same layout."""

    _fields_ = [
        ("n_convs", ctypes.c_int),
        ("conv_insize", ctypes.c_int * 8),
        ("conv_size", ctypes.c_int * 8),
        ("conv_winlen", ctypes.c_int * 8),
        ("conv_stride", ctypes.c_int * 8),
        ("conv_act", ctypes.c_int * 8),
        ("lstm_size", ctypes.c_int),
        ("lstm_layers", ctypes.c_int),
        ("state_len", ctypes.c_int),
        ("outsize", ctypes.c_int),
        ("bias", ctypes.c_int),
        ("clamp", ctypes.c_int),
        ("scale", ctypes.c_float),
        ("out_features", ctypes.c_int),
        ("num_features", ctypes.c_int),
        ("tx_d_model", ctypes.c_int),
        ("tx_nhead", ctypes.c_int),
        ("tx_depth", ctypes.c_int),
        ("tx_dim_ff", ctypes.c_int),
        ("tx_win_upper", ctypes.c_int),
        ("tx_win_lower", ctypes.c_int),
        ("tx_max_seq_len", ctypes.c_int),
        ("tx_deepnorm_alpha", ctypes.c_float),
        ("tx_theta", ctypes.c_float),
        ("up_size", ctypes.c_int),
        ("up_scale_factor", ctypes.c_int),
        ("crf_scale", ctypes.c_float),
        ("crf_blank_score", ctypes.c_float),
        ("crf_expand_blanks", ctypes.c_int),
        ("lstm_quant", ctypes.c_int),
    ]


@dataclasses.dataclass
class ConvParams:
    insize: int
    size: int
    winlen: int
    stride: int = 1
    activation: int = ACT_SWISH


@dataclasses.dataclass
class TxParams:
    """config::TxEncoderParams + LinearUpsampleParams + CRFEncoderParams
    (config/include/config/BasecallModelConfig.h:46-97)."""

    d_model: int = 512
    nhead: int = 8
    depth: int = 18
    dim_feedforward: int = 2048
    attn_window: tuple = (127, 128)
    deepnorm_alpha: float = 2.4494897
    theta: float = 10000.0
    max_seq_len: int = 2048
    up_scale_factor: int = 2
    crf_scale: float = 5.0
    crf_blank_score: float = 2.0


@dataclasses.dataclass
class ModelConfig:
    """The subset of BasecallModelConfig (config/include/config/BasecallModelConfig.h:99-160)
    the hot path reads."""

    convs: List[ConvParams]
    lstm_size: int
    lstm_layers: int = 5
    state_len: int = 4
    bias: bool = False
    clamp: bool = True
    scale: float = 1.0
    blank_score: float = 2.0
    out_features: Optional[int] = None
    num_features: int = 1
    qscale: float = 1.0
    qbias: float = 0.0
    sample_rate: int = 5000
    chunk_size: int = DEFAULT_CHUNK_SIZE
    overlap: int = DEFAULT_OVERLAP
    name: str = "synthetic"
    tx: Optional[TxParams] = None
    lstm_quant: bool = False   # opt-in: the reference's int8 LSTM path (nn/LSTMStack.cpp:127-211), csrc/lstm_q8.hip
    # synthetic weights only (dorado_amd/synth.py; no effect on a loaded model): gain on the transformer's CRF projection.
    # With gain 1 the random-init sup@v5 model calls NO base with q >= 10 (nothing discriminating to compare identities
    # on); gain 3 gives decision margins: 48 % of the reference's bases at q >= 10, 18 % at q >= 20 (scores +-27).
    synth_crf_gain: float = 1.0

    @property
    def is_tx(self) -> bool:
        return self.tx is not None

    @property
    def conv_stride(self) -> int:
        s = 1
        for c in self.convs:
            s *= c.stride
        return s

    @property
    def stride(self) -> int:
        """Samples per OUTPUT step (BasecallModelConfig.cpp:447-454: conv strides / upsample)."""
        s = self.conv_stride
        return s // self.tx.up_scale_factor if self.tx else s

    @property
    def outsize(self) -> int:
        return 4 ** (self.state_len + 1)

    @property
    def num_states(self) -> int:
        return 4 ** self.state_len

    def normalise_basecaller_params(self) -> None:
        """BatchParams::normalise (BatchParams.cpp:89-105): overlap -> multiple of stride,
        chunk -> multiple of the granularity (= stride for LSTM models)."""
        # BasecallModelConfig.h:152-159: stride_inner = stride * scale_factor; granularity x16 for Tx
        stride = self.stride * (self.tx.up_scale_factor if self.tx else 1)
        gran = stride * (16 if self.tx else 1)
        self.overlap = (self.overlap // stride) * stride
        self.chunk_size = (self.chunk_size // gran) * gran
        if self.chunk_size <= self.overlap:
            raise ValueError("chunk_size must be greater than overlap")

    def n_weights(self) -> int:
        if self.tx:
            return 2 * len(self.convs) + 7 * self.tx.depth + 3
        n = 2 * len(self.convs) + 4 * self.lstm_layers + 1
        if self.out_features is not None:
            n += 1 + (1 if self.bias else 0)
        elif not (self.convs[0].size > 4 and self.num_features == 1):
            n += 1  # pre-v4: linear bias
        return n

    def to_desc(self) -> ModelDescC:
        d = ModelDescC()
        d.n_convs = len(self.convs)
        for i, c in enumerate(self.convs):
            d.conv_insize[i] = c.insize
            d.conv_size[i] = c.size
            d.conv_winlen[i] = c.winlen
            d.conv_stride[i] = c.stride
            d.conv_act[i] = c.activation
        d.lstm_size = self.lstm_size
        d.lstm_layers = self.lstm_layers
        d.state_len = self.state_len
        d.outsize = self.outsize
        d.bias = int(self.bias)
        d.clamp = int(self.clamp)
        d.scale = self.scale
        d.out_features = self.out_features if self.out_features is not None else -1
        d.num_features = self.num_features
        d.tx_d_model = 0
        if self.tx:
            t = self.tx
            d.tx_d_model, d.tx_nhead, d.tx_depth, d.tx_dim_ff = t.d_model, t.nhead, t.depth, t.dim_feedforward
            d.tx_win_upper, d.tx_win_lower = t.attn_window
            d.tx_max_seq_len = t.max_seq_len
            d.tx_deepnorm_alpha, d.tx_theta = t.deepnorm_alpha, t.theta
            d.up_size, d.up_scale_factor = t.d_model, t.up_scale_factor
            d.crf_scale, d.crf_blank_score, d.crf_expand_blanks = t.crf_scale, t.crf_blank_score, 1
        d.lstm_quant = 1 if self.lstm_quant else 0
        return d


def hac_v43() -> ModelConfig:
    """dna_r10.4.1_e8.2_400bps_hac@v4.3.0 (tests/data/model_configs/.../config.toml)."""
    cfg = ModelConfig(
        convs=[
            ConvParams(1, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 384, 19, 6, ACT_TANH),
        ],
        lstm_size=384,
        lstm_layers=5,
        state_len=4,
        clamp=True,
        qscale=1.1,
        qbias=-1.1,
        name="dna_r10.4.1_e8.2_400bps_hac@v4.3.0",
    )
    cfg.normalise_basecaller_params()
    return cfg


def fast_v40() -> ModelConfig:
    """dna_r10.4.1_e8.2_260bps_fast@v4.0.0 (tests/data/model_configs/.../config.toml): C = 96,
    conv3 stride 5 + swish, state_len 3, plain linear head, no clamp; qscore scale 1.04, bias -3."""
    cfg = ModelConfig(
        convs=[
            ConvParams(1, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 96, 19, 5, ACT_SWISH),
        ],
        lstm_size=96, lstm_layers=5, state_len=3, clamp=False, qscale=1.04, qbias=-3.0,
        name="dna_r10.4.1_e8.2_260bps_fast@v4.0.0",
    )
    cfg.normalise_basecaller_params()
    return cfg


def fast_v43() -> ModelConfig:
    """dna_r10.4.1_e8.2_400bps_fast@v4.3.0 — config NOT in the reference tree; inferred in
    SURVEY.md §8 (C = 96, stride 6, state_len 3)."""
    cfg = ModelConfig(
        convs=[
            ConvParams(1, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 96, 19, 6, ACT_TANH),
        ],
        lstm_size=96, lstm_layers=5, state_len=3, clamp=True,
        name="dna_r10.4.1_e8.2_400bps_fast@v4.3.0(inferred)",
    )
    cfg.normalise_basecaller_params()
    return cfg


def sup_v43() -> ModelConfig:
    """dna_r10.4.1_e8.2_400bps_sup@v4.3.0 — config NOT in the reference tree; topology
    inferred in SURVEY.md §8 (C=1024, state_len 5)."""
    cfg = ModelConfig(
        convs=[
            ConvParams(1, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 1024, 19, 6, ACT_TANH),
        ],
        lstm_size=1024,
        lstm_layers=5,
        state_len=5,
        clamp=True,
        name="dna_r10.4.1_e8.2_400bps_sup@v4.3.0(inferred)",
    )
    cfg.normalise_basecaller_params()
    return cfg


def sup_v50() -> ModelConfig:
    """dna_r10.4.1_e8.2_400bps_sup@v5.0.0 (tests/data/model_configs/.../config.toml)."""
    cfg = ModelConfig(
        convs=[
            ConvParams(1, 64, 5, 1, ACT_SWISH),
            ConvParams(64, 64, 5, 1, ACT_SWISH),
            ConvParams(64, 128, 9, 3, ACT_SWISH),
            ConvParams(128, 128, 9, 2, ACT_SWISH),
            ConvParams(128, 512, 5, 2, ACT_SWISH),
        ],
        lstm_size=0, lstm_layers=0, state_len=5, clamp=False, tx=TxParams(),
        chunk_size=12288, overlap=600, name="dna_r10.4.1_e8.2_400bps_sup@v5.0.0", synth_crf_gain=3.0,
    )
    cfg.normalise_basecaller_params()
    return cfg


def tiny_tx(d_model: int = 128, nhead: int = 2, depth: int = 2, ff: int = 256, state_len: int = 3,
            window=(15, 16)) -> ModelConfig:
    """Small same-topology transformer model for fast parity tests."""
    cfg = ModelConfig(
        convs=[
            ConvParams(1, 64, 5, 1, ACT_SWISH),
            ConvParams(64, 64, 5, 1, ACT_SWISH),
            ConvParams(64, 128, 9, 3, ACT_SWISH),
            ConvParams(128, 128, 9, 2, ACT_SWISH),
            ConvParams(128, d_model, 5, 2, ACT_SWISH),
        ],
        lstm_size=0, lstm_layers=0, state_len=state_len, clamp=False,
        tx=TxParams(d_model=d_model, nhead=nhead, depth=depth, dim_feedforward=ff, attn_window=window),
        chunk_size=1536, overlap=192, name=f"tiny-tx-{d_model}-{depth}",
    )
    cfg.normalise_basecaller_params()
    return cfg


def tiny(C: int = 64, state_len: int = 3, stride: int = 6) -> ModelConfig:
    """Small same-topology model for fast parity tests."""
    cfg = ModelConfig(
        convs=[
            ConvParams(1, 16, 5, 1, ACT_SWISH),
            ConvParams(16, 16, 5, 1, ACT_SWISH),
            ConvParams(16, C, 19, stride, ACT_TANH),
        ],
        lstm_size=C,
        lstm_layers=5,
        state_len=state_len,
        clamp=True,
        name=f"tiny-{C}-{state_len}",
    )
    cfg.normalise_basecaller_params()
    return cfg


def load_model_config(path: str) -> ModelConfig:
    """Parse `<path>/config.toml` (v4-type LSTM models) the way
    BasecallModelConfig.cpp:214-323 does."""
    import tomli

    with open(os.path.join(path, "config.toml"), "rb") as f:
        t = tomli.load(f)
    enc = t["encoder"]
    if "type" not in enc:
        raise NotImplementedError("pre-v4 model configs are not supported yet")
    if any(s.get("type") in ("upsample",) for s in enc["sublayers"]) or "transformer_encoder" in enc:
        raise NotImplementedError("transformer configs are handled in a later round")
    subs = enc["sublayers"]
    convs: List[ConvParams] = []
    lstm_layers = 0
    out_features = None
    bias = False
    scale = 1.0
    blank_score = 2.0
    clamp = any(s["type"] == "clamp" for s in subs)
    for i, s in enumerate(subs):
        ty = s["type"]
        if ty == "convolution":
            nxt_clamp = i + 1 < len(subs) and subs[i + 1]["type"] == "clamp"
            act = s["activation"]
            if act == "swish":
                a = ACT_SWISH_CLAMP if nxt_clamp else ACT_SWISH
            elif act == "tanh":
                a = ACT_TANH
            else:
                raise ValueError(f"Unknown activation: `{act}`")
            convs.append(ConvParams(s["insize"], s["size"], s["winlen"], s["stride"], a))
        elif ty == "lstm":
            lstm_layers += 1
        elif ty == "linear":
            out_features = int(s["out_features"])
        elif ty == "linearcrfencoder":
            blank_score = float(s["blank_score"])
            scale = float(s.get("scale", 1.0))
    lstm_size = convs[-1].size
    for s in subs:
        if s["type"] == "linear":
            bias = bool(s.get("bias", lstm_size > 128))
    if len(convs) != 3:
        raise ValueError(f"Expected 3 convolution layers but found: {len(convs)}")
    q = t.get("qscore", {})
    cfg = ModelConfig(
        convs=convs,
        lstm_size=lstm_size,
        lstm_layers=lstm_layers,
        state_len=int(t["global_norm"]["state_len"]),
        bias=bias,
        clamp=clamp,
        scale=scale,
        blank_score=blank_score,
        out_features=out_features,
        num_features=int(t["input"]["features"]),
        qscale=float(q.get("scale", 1.0)),
        qbias=float(q.get("bias", 0.0)),
        sample_rate=int(t.get("run_info", {}).get("sample_rate", -1)),
        name=os.path.basename(os.path.normpath(path)),
    )
    bc = t.get("basecaller", {})
    if "chunksize" in bc:
        cfg.chunk_size = int(bc["chunksize"])
    if "overlap" in bc:
        cfg.overlap = int(bc["overlap"])
    cfg.normalise_basecaller_params()
    return cfg
