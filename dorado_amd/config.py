"""Host-side mirror of the reference's model configuration for the hot path.

Follows dorado/config/BasecallModelConfig.cpp:214-323 (load_lstm_model_config: v4-type and pre-v4 LSTM configs), :366-462
(load_tx_model_config), :153-197 (signal normalisation), :63-148 (qscore, run_info / sample type),
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
    up_size: int = 512                  # upsample.d_model
    crf_insize: int = 512
    crf_n_base: int = 4
    crf_expand_blanks: bool = True
    crf_permute: tuple = (1, 0, 2)


@dataclasses.dataclass
class SignalNorm:
    """config::SignalNormalisationParams (config/include/config/BasecallModelConfig.h:13-44): what ScalerNode is created with."""

    strategy: str = "quantile"          # ScalingStrategy: "med_mad" | "quantile" | "pa"
    quantile_a: float = 0.2
    quantile_b: float = 0.9
    shift_multiplier: float = 0.51
    scale_multiplier: float = 0.53
    standardise: bool = False
    mean: float = 0.0
    stdev: float = 1.0


# models/models.cpp simplex::deprecated (load_model_config refuses them by NAME: BasecallModelConfig.cpp:503-507,
# models.cpp:1792-1814); the names are data of the reference's model catalogue
DEPRECATED_SIMPLEX_MODELS = frozenset("""
dna_r9.4.1_e8_fast@v3.4 dna_r9.4.1_e8_hac@v3.3 dna_r9.4.1_e8_sup@v3.3 dna_r9.4.1_e8_sup@v3.6
dna_r10.4.1_e8.2_260bps_fast@v3.5.2 dna_r10.4.1_e8.2_260bps_hac@v3.5.2 dna_r10.4.1_e8.2_260bps_sup@v3.5.2
dna_r10.4.1_e8.2_400bps_fast@v3.5.2 dna_r10.4.1_e8.2_400bps_hac@v3.5.2 dna_r10.4.1_e8.2_400bps_sup@v3.5.2
dna_r10.4.1_e8.2_260bps_fast@v4.0.0 dna_r10.4.1_e8.2_260bps_hac@v4.0.0 dna_r10.4.1_e8.2_260bps_sup@v4.0.0
dna_r10.4.1_e8.2_400bps_fast@v4.0.0 dna_r10.4.1_e8.2_400bps_hac@v4.0.0 dna_r10.4.1_e8.2_400bps_sup@v4.0.0
dna_r10.4.1_e8.2_260bps_fast@v4.1.0 dna_r10.4.1_e8.2_260bps_hac@v4.1.0 dna_r10.4.1_e8.2_260bps_sup@v4.1.0
dna_r10.4.1_e8.2_400bps_fast@v4.1.0 dna_r10.4.1_e8.2_400bps_hac@v4.1.0 dna_r10.4.1_e8.2_400bps_sup@v4.1.0
rna002_70bps_fast@v3 rna002_70bps_hac@v3
""".split())


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
    signal_norm: SignalNorm = dataclasses.field(default_factory=SignalNorm)
    sample_type: str = "DNA"            # models::SampleType: "DNA" | "RNA002" | "RNA004" (run_info.sample_type or the model name)
    mean_qscore_start_pos: int = -1     # qscore.mean_qscore_start_pos, else 60 (BasecallModelConfig.cpp:23-39)
    lstm_quant: bool = False   # the reference's int8 LSTM path (nn/LSTMStack.cpp:127-211; csrc/lstm_q8.hip, int8 cluster kernel); bench.py and the
                               # adapter set it by the reference's own rule (reference_gpu_lstm_int8 below); False = f16 throughout
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

    def reference_gpu_lstm_int8(self) -> bool:
        """The LSTM arithmetic the reference's GPU build picks for this model (nn/ConvStack.cpp:60-89 get_koi_lstm_input_layout):
        the convolution in front of the LSTM stack writes the int8 layout — every LSTM layer int8, nn/LSTMStack.cpp:127-211 — when
        it ends in tanh and 128 < lstm_size <= 1024, lstm_size % 128 == 0.  (integration/HipModelRunnerAdapter.h restates it in
        C++ incl. the DORADO_LSTM_MODE override; bench.py uses it to pick the headline arithmetic.)"""
        return (self.tx is None and len(self.convs) >= 3 and self.lstm_layers >= 2 and self.convs[-1].activation == ACT_TANH
                and 128 < self.lstm_size <= 1024 and self.lstm_size % 128 == 0)

    def normalise_basecaller_params(self) -> None:
        """BatchParams::normalise (BatchParams.cpp:89-105): overlap -> multiple of stride,
        chunk -> multiple of the granularity (= stride for LSTM models)."""
        # BasecallModelConfig.h:152-159: stride_inner = stride * scale_factor; granularity x16 for Tx
        stride = self.stride * (self.tx.up_scale_factor if self.tx else 1)
        gran = stride * (16 if self.tx else 1)
        self.overlap = max(1, self.overlap // stride) * stride            # a multiple of the stride, greater than 0
        self.chunk_size = (max(self.overlap + gran - 1, self.chunk_size) // gran) * gran   # ... and greater than the overlap

    def has_normalised_basecaller_params(self) -> bool:
        """BasecallModelConfig::has_normalised_basecaller_params (BasecallModelConfig.cpp:475-499)."""
        stride = self.stride * (self.tx.up_scale_factor if self.tx else 1)
        gran = stride * (16 if self.tx else 1)
        return self.chunk_size % gran == 0 and self.overlap % stride == 0 and self.chunk_size > self.overlap

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


def _signal_norm(t) -> SignalNorm:
    """parse_signal_normalisation_params (BasecallModelConfig.cpp:153-197)."""
    sn = SignalNorm()
    if "scaling" in t:
        st = t["scaling"]["strategy"]
        if st not in ("med_mad", "quantile", "pa"):
            raise ValueError(f"Unknown scaling strategy: `{st}`")
        sn.strategy = st
    if "normalisation" in t:
        n = t["normalisation"]
        sn.quantile_a, sn.quantile_b = float(n["quantile_a"]), float(n["quantile_b"])
        sn.shift_multiplier, sn.scale_multiplier = float(n["shift_multiplier"]), float(n["scale_multiplier"])
    if "standardisation" in t:
        n = t["standardisation"]
        sn.standardise = int(n["standardise"]) > 0
        if sn.standardise:
            sn.mean, sn.stdev = float(n["mean"]), float(n["stdev"])
        if sn.standardise and sn.strategy != "pa":
            raise ValueError("Signal standardisation is implemented only for `scaling.strategy = pa`")
        if sn.stdev <= 0.0:
            raise ValueError(f"Config error: `standardisation.stdev` must be greater than 0, got: {sn.stdev}")
    return sn


def _sample_type(t, model_name: str) -> str:
    """parse_run_info (BasecallModelConfig.cpp:117-139) + models::get_sample_type_from_model_name (kits.cpp:448-458)."""
    st = "UNKNOWN"
    ri = t.get("run_info", {})
    if "sample_type" in ri:
        up = str(ri["sample_type"]).upper()
        st = up if up in ("DNA", "RNA002", "RNA004") else "UNKNOWN"
    if st == "UNKNOWN":
        st = "RNA004" if "rna004" in model_name else "RNA002" if "rna002" in model_name else "DNA" if "dna" in model_name else "UNKNOWN"
    if st == "UNKNOWN":
        raise ValueError("Failed to determine model sample type from model name or config")
    return st


def _conv(seg, clamp_follows: bool) -> ConvParams:
    """parse_conv_params (config/common.cpp:53-87): swish followed by a clamp sublayer is swish_clamp."""
    act = seg["activation"]
    if act == "swish":
        a = ACT_SWISH_CLAMP if clamp_follows else ACT_SWISH
    elif act == "tanh":
        a = ACT_TANH
    else:
        raise ValueError(f"Unknown activation: `{act}`")
    return ConvParams(int(seg["insize"]), int(seg["size"]), int(seg["winlen"]), int(seg["stride"]), a)


def is_tx_model_config(path: str) -> bool:
    """BasecallModelConfig.cpp:343-347."""
    import tomli

    with open(os.path.join(path, "config.toml"), "rb") as f:
        t = tomli.load(f)
    return isinstance(t.get("model", {}).get("encoder", {}), dict) and "transformer_encoder" in t.get("model", {}).get("encoder", {})


def load_model_config(path: str, allow_deprecated: bool = False) -> ModelConfig:
    """Parse `<path>/config.toml` the way config::load_model_config does (BasecallModelConfig.cpp:501-507): models of the
    reference's deprecated catalogue are refused by name (allow_deprecated=True parses them anyway — the v4.0 / v4.1 LSTM
    configs are still well-formed), transformer configs go through the load_tx_model_config rules (:410-462), everything else
    through load_lstm_model_config (:214-323; v4-type sublayer lists and the pre-v4 flat encoder table).
    The transformer's lstm_size is 0 here (the reference stores -1 "to force a downstream issue"); `bias` and `scale` keep the
    struct defaults (true, 1.0) for transformer models exactly as there."""
    import tomli

    name = os.path.basename(os.path.normpath(os.path.realpath(path)))
    if name in DEPRECATED_SIMPLEX_MODELS and not allow_deprecated:
        raise ValueError(f"Deprecated model: '{name}'. Its chemistry has been deprecated since Dorado version 1.0.0.")
    with open(os.path.join(path, "config.toml"), "rb") as f:
        t = tomli.load(f)
    q = t.get("qscore")
    common = dict(
        qscale=float(q["scale"]) if q else 1.0,
        qbias=float(q["bias"]) if q else 0.0,
        mean_qscore_start_pos=(int(q["mean_qscore_start_pos"]) if "mean_qscore_start_pos" in q else 60) if q else -1,
        sample_rate=int(t["run_info"]["sample_rate"]) if "run_info" in t else -1,
        signal_norm=_signal_norm(t),
        sample_type=_sample_type(t, name),
        name=name,
    )
    if common["mean_qscore_start_pos"] < 0 and q:
        raise ValueError("model config error - qscore.mean_qscore_start_pos cannot be < 0")
    menc = t.get("model", {}).get("encoder", {})
    if isinstance(menc, dict) and "transformer_encoder" in menc:
        # ---- load_tx_model_config
        enc, layer, ups, crf = menc["transformer_encoder"], menc["transformer_encoder"]["layer"], menc["upsample"], menc["crf"]
        if "rotary_base" in layer and "theta" in layer:
            raise ValueError("Model Config Error. [model.encoder.transformer_encoder] 'rotary_base' and 'theta' are mutually exclusive.")
        tx = TxParams(
            d_model=int(layer["d_model"]), nhead=int(layer["nhead"]), depth=int(enc["depth"]),
            dim_feedforward=int(layer["dim_feedforward"]), attn_window=(int(layer["attn_window"][0]), int(layer["attn_window"][1])),
            deepnorm_alpha=float(layer["deepnorm_alpha"]),
            theta=float(layer.get("theta", layer.get("rotary_base", TxParams.theta))),
            max_seq_len=int(layer.get("max_seq_len", TxParams.max_seq_len)),
            up_scale_factor=int(ups["scale_factor"]), up_size=int(ups["d_model"]),
            crf_scale=float(crf["scale"]), crf_blank_score=float(crf["blank_score"]), crf_insize=int(crf["insize"]),
            crf_n_base=int(crf["n_base"]), crf_expand_blanks=bool(crf["expand_blanks"]), crf_permute=tuple(int(v) for v in crf["permute"]),
        )
        convs = [_conv(s_, False) for s_ in menc["conv"]["sublayers"] if s_["type"] == "convolution"]   # no swish clamp in Tx models
        state_len = int(crf["state_len"])
        cfg = ModelConfig(convs=convs, lstm_size=0, lstm_layers=0, state_len=state_len, bias=True, clamp=False, scale=1.0,
                          out_features=tx.crf_n_base ** (state_len + 1), num_features=convs[0].insize, tx=tx, **common)
    else:
        # ---- load_lstm_model_config
        enc = t["encoder"]
        num_features = int(t["input"]["features"])
        bias, clamp, scale, blank_score, out_features = True, False, 1.0, 2.0, None    # the struct's defaults
        if "type" in enc:       # v4-type model
            subs = enc["sublayers"]
            bias = False
            clamp = any(s_["type"] == "clamp" for s_ in subs)
            convs = [_conv(s_, i + 1 < len(subs) and subs[i + 1]["type"] == "clamp") for i, s_ in enumerate(subs)
                     if s_["type"] == "convolution"]
            lstm_size = convs[-1].size
            lstm_layers = sum(1 for s_ in subs if s_["type"] == "lstm")
            if any(s_["type"] == "flstm" for s_ in subs):
                raise NotImplementedError("factorised-LSTM model configs: the engine has no FLSTM stack (not a BASELINE model)")
            for s_ in subs:
                if s_["type"] == "linear":
                    out_features = int(s_["out_features"])
                    bias = bool(s_.get("bias", lstm_size > 128))
                elif s_["type"] == "linearcrfencoder":
                    blank_score = float(s_["blank_score"])
                    scale = float(s_.get("scale", 1.0))
        else:                   # pre-v4 model
            stride, lstm_size = int(enc["stride"]), int(enc["features"])
            blank_score, scale = float(enc["blank_score"]), float(enc["scale"])
            first_conv = int(enc.get("first_conv_size", 4))
            convs = [ConvParams(num_features, first_conv, 5, 1, ACT_SWISH), ConvParams(first_conv, 16, 5, 1, ACT_SWISH),
                     ConvParams(16, lstm_size, 19, stride, ACT_SWISH)]
            lstm_layers = 5
        if len(convs) != 3:
            raise ValueError(f"Expected 3 convolution layers but found: {len(convs)}")
        if convs[0].size not in (4, 16):
            raise ValueError(f"Invalid CRF model configuration - first convolution layer must be size 4 or 16. Got: {convs[0].size}")
        cfg = ModelConfig(convs=convs, lstm_size=lstm_size, lstm_layers=lstm_layers, state_len=int(t["global_norm"]["state_len"]),
                          bias=bias, clamp=clamp, scale=scale, blank_score=blank_score, out_features=out_features,
                          num_features=num_features, **common)
    bc = t.get("basecaller", {})          # BatchParams::update(path): chunksize / overlap; batchsize is ignored (BatchParams.cpp:25-55)
    if "chunksize" in bc:
        cfg.chunk_size = int(bc["chunksize"])
    if "overlap" in bc:
        cfg.overlap = int(bc["overlap"])
    cfg.normalise_basecaller_params()
    return cfg
