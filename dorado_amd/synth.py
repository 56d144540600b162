"""Seeded synthetic weights and signal (there are no model files or datasets offline).

Recipe after SURVEY.md §8(d): torch-default-shaped uniform init with gains chosen (probed with
the oracle) so that the output depends on the input signal, pre-clamp scores span about +-5 and
the beam search emits ~0.5-0.7 bases per output step: conv weights x3, W_ih x8, W_hh x1, CRF head
x10; signal = unit noise + piecewise-constant level process, clipped to +-5.
"""
from __future__ import annotations

import numpy as np

from .config import ModelConfig


def make_weights(cfg: ModelConfig, seed: int = 42, conv_gain: float = 3.0, ih_gain: float = 8.0,
                 hh_gain: float = 1.0, head_gain: float = 10.0, bias_hh: bool = True):
    """Returns a list of f32 arrays in module.parameters() order
    (dorado/basecall/crf_utils.cpp:34-88): conv{w,b}*, rnn{w_ih,w_hh,b_ih,b_hh}*, linear*."""
    rng = np.random.default_rng(seed)
    if cfg.tx is not None:
        return _make_tx_weights(cfg, rng)
    ws = []
    for c in cfg.convs:
        k = 1.0 / np.sqrt(c.insize * c.winlen)
        ws.append((rng.uniform(-k, k, size=(c.size, c.insize, c.winlen)) * conv_gain).astype(np.float32))
        ws.append(rng.uniform(-k, k, size=(c.size,)).astype(np.float32))
    C = cfg.lstm_size
    k = 1.0 / np.sqrt(C)
    for _ in range(cfg.lstm_layers):
        ws.append((rng.uniform(-k, k, size=(4 * C, C)) * ih_gain).astype(np.float32))
        ws.append((rng.uniform(-k, k, size=(4 * C, C)) * hh_gain).astype(np.float32))
        ws.append(rng.uniform(-k, k, size=(4 * C,)).astype(np.float32))
        if bias_hh:
            ws.append(rng.uniform(-k, k, size=(4 * C,)).astype(np.float32))
        else:
            ws.append(np.zeros((4 * C,), np.float32))
    K = cfg.outsize
    if cfg.out_features is not None:
        D = cfg.out_features
        ws.append((rng.uniform(-k, k, size=(D, C)) * 2.0).astype(np.float32))
        if cfg.bias:
            ws.append(rng.uniform(-k, k, size=(D,)).astype(np.float32))
        kd = 1.0 / np.sqrt(D)
        ws.append((rng.uniform(-kd, kd, size=(K, D)) * head_gain).astype(np.float32))
    elif cfg.convs[0].size > 4 and cfg.num_features == 1:
        ws.append((rng.uniform(-k, k, size=(K, C)) * head_gain).astype(np.float32))
    else:
        ws.append((rng.uniform(-k, k, size=(K, C)) * head_gain).astype(np.float32))
        ws.append(rng.uniform(-k, k, size=(K,)).astype(np.float32))
    return ws


def make_signal(n_chunks: int, t_in: int, seed: int = 0xD0AD0, mean_dwell: float = 9.0):
    """[n_chunks, t_in] f16 PA-standardised-looking signal."""
    rng = np.random.default_rng(seed)
    out = np.empty((n_chunks, t_in), np.float32)
    for i in range(n_chunks):
        n_levels = int(t_in / mean_dwell * 1.5) + 16
        dwell = rng.geometric(1.0 / mean_dwell, size=n_levels)
        levels = rng.standard_normal(n_levels).astype(np.float32)
        sig = np.repeat(levels, dwell)[:t_in]
        if sig.size < t_in:
            sig = np.pad(sig, (0, t_in - sig.size), mode="edge")
        out[i] = sig + 0.35 * rng.standard_normal(t_in).astype(np.float32)
    return np.clip(out, -5.0, 5.0).astype(np.float16)


def _make_tx_weights(cfg: ModelConfig, rng):
    """Parameter order of TxModel (dorado/basecall/crf_utils.cpp:100-147): conv{1..5}.{w,b};
    per layer {wqkv.w, out_proj.w, out_proj.b, fc1.w, fc2.w, norm1.w, norm2.w}; upsample.{w,b}; crf.w.
    torch-default-shaped init (the SURVEY probe found default init already gives ~0.6 bases/step)."""
    t = cfg.tx
    ws = []
    for c in cfg.convs:
        k = 1.0 / np.sqrt(c.insize * c.winlen)
        ws.append((rng.uniform(-k, k, size=(c.size, c.insize, c.winlen)) * 2.0).astype(np.float32))
        ws.append(rng.uniform(-k, k, size=(c.size,)).astype(np.float32))
    C, F = t.d_model, t.dim_feedforward
    k = 1.0 / np.sqrt(C)
    kf = 1.0 / np.sqrt(F)
    for _ in range(t.depth):
        ws.append(rng.uniform(-k, k, size=(3 * C, C)).astype(np.float32) * 2.0)
        ws.append(rng.uniform(-k, k, size=(C, C)).astype(np.float32))
        ws.append(rng.uniform(-k, k, size=(C,)).astype(np.float32))
        ws.append(rng.uniform(-k, k, size=(2 * F, C)).astype(np.float32))
        ws.append(rng.uniform(-kf, kf, size=(C, F)).astype(np.float32))
        ws.append((1.0 + 0.1 * rng.standard_normal(C)).astype(np.float32))
        ws.append((1.0 + 0.1 * rng.standard_normal(C)).astype(np.float32))
    ws.append(rng.uniform(-k, k, size=(t.up_scale_factor * C, C)).astype(np.float32))
    ws.append(rng.uniform(-k, k, size=(t.up_scale_factor * C,)).astype(np.float32))
    ws.append(rng.uniform(-k, k, size=(cfg.outsize, C)).astype(np.float32) * np.float32(getattr(cfg, "synth_crf_gain", 1.0)))
    return [np.ascontiguousarray(w, np.float32) for w in ws]
