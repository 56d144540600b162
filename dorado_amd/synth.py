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


# ------------------------------------------------------------------------------------------------------------------------
# A synthetic LSTM-CRF model WITH DECISION MARGINS (round 6).
#
# Random weights (make_weights) give a model whose beam search sits on near-ties everywhere: 95 % of its bases are called
# below q20 and one f16 ulp of score noise flips 3 % of them, so per-read identity cannot hold any data path to the 0.995 a
# trained model gives.  No trained weights exist offline, so this recipe BUILDS the function a trained model computes, for a
# synthetic pore whose level is set by the newest base (make_base_signal):
#   conv1 / conv2   pass the signal and its negation (swish(u) - swish(-u) = u) and smooth it over 3 samples;
#   conv3           twelve tanh threshold units per output step: is the level in each of four 3-sample windows — two LEFT of
#                   the step (samples 6t-9 .. 6t-4), two RIGHT of it (6t+2 .. 6t+7) — above each of the three thresholds
#                   between the four base levels;
#   LSTM layer 1    twelve units e[p][n] (p != n): the AND of "both left windows are base p" and "both right windows are base
#                   n" — base n was entered, after base p, at this output step (input gate open, forget gate shut: no memory
#                   needed; a window that straddles a level change disagrees with its clean neighbour, so no event fires for
#                   an in-between level);
#   LSTM layers 2-5 carry those units through (h = tanh(tanh(2 x)));
#   linear head     transition k (new state * 4 + dropped base, dorado/basecall/decode/beam_search.cpp:209-222) scores
#                   +4.3 when e[previous base of k][newest base of k] fired and -4.3 otherwise (homopolymer steps: -4.3).
# A path that calls a wrong base, or none, is in a state whose outgoing transitions never fire again: the posterior is as
# concentrated as a trained model's.  EVERY other weight of the architecture is dense seeded noise (torch-default-shaped,
# gains below) and all remaining units are a random reservoir feeding the head through small weights, so the scores carry a
# continuous input-dependent component, every MFMA lane does real work and quantisation / rounding noise propagates as in any
# dense network.  The criteria this recipe was tuned to (fixed by VERDICT r5 before any device output was looked at; measured
# on the CPU with tools/margin_sweep.py against the f32 oracle): 0.40-0.55 bases per output step, >= 40 % of them at
# q >= 20, f16-storage emulation vs f32 median identity >= 0.995.
BASE_LEVEL_STEP = 0.9
BASE_LEVEL_SHIFT = 0.225


def base_levels():
    return (BASE_LEVEL_STEP * (np.arange(4) - 1.5) + BASE_LEVEL_SHIFT).astype(np.float32)


def make_base_signal(n_chunks: int, t_in: int, seed: int = 0xBA5E, mean_dwell: float = 12.5, min_dwell: int = 10,
                     noise: float = 0.2, want_truth: bool = False):
    """[n_chunks, t_in] f16: a base sequence without homopolymers (5 kHz / 400 bases per second = 12.5 samples per base), one
    level per base + white noise.  want_truth: also the list of (bases, start sample) per chunk."""
    rng = np.random.default_rng(seed)
    lv = base_levels()
    out = np.empty((n_chunks, t_in), np.float32)
    truth = []
    for i in range(n_chunks):
        nb = int(t_in / mean_dwell * 1.6) + 16
        dw = min_dwell + rng.geometric(1.0 / (mean_dwell - min_dwell + 1.0), size=nb) - 1
        b = np.cumsum(np.concatenate([rng.integers(0, 4, 1), rng.integers(1, 4, nb - 1)])) % 4
        sig = np.repeat(lv[b], dw)[:t_in]
        if sig.size < t_in:
            sig = np.pad(sig, (0, t_in - sig.size), mode="edge")
        out[i] = sig + noise * rng.standard_normal(t_in).astype(np.float32)
        if want_truth:
            st = np.concatenate([[0], np.cumsum(dw)[:-1]])
            truth.append((b[st < t_in], st[st < t_in]))
    x = np.clip(out, -5.0, 5.0).astype(np.float16)
    return (x, truth) if want_truth else x


def make_margin_weights(cfg: ModelConfig, seed: int = 42, noise_gain: float = 1.0, on_score: float = 4.3,
                        beta: float = 6.0, gamma: float = 3.0, alpha: float = 2.0, head_noise: float = 0.15):
    """Weights in module.parameters() order (as make_weights) for the LSTM-CRF models with three convolutions (stride 6
    in the last), lstm_size >= 64 and the plain linear head."""
    assert cfg.tx is None and len(cfg.convs) == 3 and cfg.out_features is None
    c1, c2, c3 = cfg.convs
    assert c1.insize == 1 and c1.winlen == 5 and c2.winlen == 5 and c3.winlen == 19 and c3.stride == 6 and c3.activation == 2
    rng = np.random.default_rng(seed)
    C = cfg.lstm_size
    K = cfg.outsize
    f32 = np.float32

    def uni(shape, fan_in, gain):
        k = 1.0 / np.sqrt(fan_in)
        return (rng.uniform(-k, k, size=shape) * gain * noise_gain).astype(f32)

    ws = []
    # conv1: channel 0 = swish(x), channel 1 = swish(-x); the rest dense noise
    w = uni((c1.size, 1, 5), 5, 0.5)
    b = uni((c1.size,), 5, 0.5)
    w[0, 0], w[1, 0] = [0, 0, 1, 0, 0], [0, 0, -1, 0, 0]
    b[0] = b[1] = 0.0
    ws += [w, b]
    # conv2: channels 0 / 1 = swish(+- 3-sample mean of x)
    w = uni((c2.size, c2.insize, 5), c2.insize * 5, 0.5)
    b = uni((c2.size,), c2.insize * 5, 0.5)
    w[0], w[1], b[0], b[1] = 0.0, 0.0, 0.0, 0.0
    w[0, 0, 1:4], w[0, 1, 1:4], w[1, 0, 1:4], w[1, 1, 1:4] = 1 / 3, -1 / 3, -1 / 3, 1 / 3
    ws += [w, b]
    # conv3: unit 3 win + j = tanh(beta (window mean - theta_j)) for the four 3-sample windows L2, L1, R1, R2 centred on samples
    # 6t-8, 6t-5, 6t+3, 6t+6 (tap k of output step t = sample 6t + k - 9): a level change at sample 6t-3 .. 6t+2 belongs to step t
    theta = base_levels()[:3] + 0.5 * BASE_LEVEL_STEP
    w = uni((C, c3.insize, 19), c3.insize * 19, 1.0)
    b = uni((C,), c3.insize * 19, 1.0)
    for win, tap in enumerate((1, 4, 12, 15)):
        for j in range(3):
            u = 3 * win + j
            w[u] = 0.0
            w[u, 0, tap], w[u, 1, tap] = beta, -beta
            b[u] = -beta * theta[j]
    ws += [w, b]
    # LSTM: structured units first, the rest a random reservoir (W_ih x2, W_hh x1 on torch's default init)
    pairs = [(p, n) for p in range(4) for n in range(4) if p != n]
    NS = len(pairs) + 1                       # 12 event units + one always-on unit
    for layer in range(cfg.lstm_layers):
        wih = uni((4 * C, C), C, 2.0)
        whh = uni((4 * C, C), C, 1.0)
        bih = uni((4 * C,), C, 1.0)
        bhh = uni((4 * C,), C, 1.0)
        for u in range(NS):
            for g in range(4):
                wih[g * C + u] *= 0.05        # the structured units listen to the reservoir only faintly
                whh[g * C + u] *= 0.05
                bhh[g * C + u] = 0.0
            bih[0 * C + u], bih[1 * C + u], bih[3 * C + u] = 6.0, -6.0, 6.0   # input / output gates open, forget gate shut
            row = wih[2 * C + u]
            if u == NS - 1:
                bih[2 * C + u] = 3.0          # always on
            elif layer == 0:
                p, n = pairs[u]
                terms = []
                for win, cls in ((0, p), (1, p), (2, n), (3, n)):     # both left windows are base p, both right windows base n
                    if cls > 0:
                        terms.append((3 * win + cls - 1, 1.0))
                    if cls < 3:
                        terms.append((3 * win + cls, -1.0))
                for idx, sgn in terms:
                    row[idx] = gamma * sgn
                bih[2 * C + u] = -gamma * (len(terms) - 1)
            else:
                row[u] = alpha
                bih[2 * C + u] = 0.0
        ws += [wih, whh, bih, bhh]
    # head: transition k = new_state * 4 + dropped base; newest base = (k >> 2) & 3, the one before = (k >> 4) & 3
    hstar = np.tanh(np.tanh(alpha * 0.73))    # the carried value of a fired unit (fixed point of h = tanh(tanh(alpha h)))
    w = uni((K, C), C, 1.0)
    w *= f32(head_noise * np.sqrt(3.0) / 0.35)             # reservoir rms ~ 0.35: the noise term of a score has rms ~ head_noise
    w[:, :NS] = 0.0
    ks = np.arange(K)
    new, prev = (ks >> 2) & 3, (ks >> 4) & 3
    idx = {pn: u for u, pn in enumerate(pairs)}
    for k in range(K):
        if new[k] != prev[k]:
            w[k, idx[(int(prev[k]), int(new[k]))]] = on_score / hstar
        else:
            w[k, NS - 1] = -on_score / hstar
    ws.append(w)
    if cfg.convs[0].size <= 4 or cfg.num_features != 1:
        ws.append(np.zeros((K,), f32))
    return [np.ascontiguousarray(a, f32) for a in ws]
