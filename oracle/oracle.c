/* oracle/oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C (f32) restatement of the reference's simplex-basecalling hot path, function by
 * function, each citing the reference file:line it follows (paths relative to
 * /root/reference/dorado).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the product (dorado_amd/) never does.
 *
 * Pinning: every function here is checked (tests/test_oracle_*.py, `-m "not gpu"`) against
 *   (1) the golden vectors the reference's own tests hold (tests/ChunkTest.cpp:28-39,
 *       tests/StitchTest.cpp:10-99) and
 *   (2) outputs of the reference itself compiled in place (oracle/_ref/libdorado_ref.so),
 *       committed as fixtures under tests/golden/ by tests/golden/make_golden.py.
 * The reference holds NO golden vectors for the network/decoder, so for those
 * rows the pin is (2) only.
 *
 * Two transcendental modes (last argument `det` of the decode functions):
 *   det = 0: glibc expf/logf/log1pf — the natural restatement.
 *   det = 1: a fixed fmaf-only polynomial exp/log (det_expf/det_logf below) that the HIP
 *            decoder evaluates with the identical operation sequence, so that the integer
 *            outputs (moves, bases) can be compared BIT-EXACTLY between CPU and GPU.
 * Build with -ffp-contract=off (see oracle/Makefile) so no a*b+c is fused behind our back.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------
 * Deterministic transcendental kernel (shared definition with dorado_amd/csrc/detmath.h —
 * restated there, not included from here).
 * ---------------------------------------------------------------------------------------- */
static inline float bits_to_f(uint32_t u) {
    float f;
    memcpy(&f, &u, 4);
    return f;
}
static inline uint32_t f_to_bits(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return u;
}

/* exp(x) for x <= 0; arguments below -86 are clamped (the result, < 5e-38, vanishes in any sum
 * that also holds the exp(0) = 1 term of a log-sum-exp). */
static float det_expf(float x) {
    x = fmaxf(x, -86.0f);
    const float n = rintf(x * 1.44269504088896341f);
    const float r = fmaf(n, -0.693147182464599609375f, x); /* one-constant ln2 */
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    /* p in [0.70, 1.42], n in [-124, 0]: 2^n by adding n to the exponent field */
    return bits_to_f(f_to_bits(p) + ((uint32_t)(int)n << 23));
}

/* log(x) for finite x > 0 (normal). */
static float det_logf(float x) {
    /* mantissa into [sqrt(1/2), sqrt(2)): re-bias the exponent field around sqrt(1/2) */
    const uint32_t ix = f_to_bits(x) + (0x3f800000u - 0x3f3504f3u);
    const int e = (int)(ix >> 23) - 127;
    const float m = bits_to_f((ix & 0x007fffffu) + 0x3f3504f3u);
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = fmaf(p, f, -1.1514610310e-1f);
    p = fmaf(p, f, 1.1676998740e-1f);
    p = fmaf(p, f, -1.2420140846e-1f);
    p = fmaf(p, f, 1.4249322787e-1f);
    p = fmaf(p, f, -1.6668057665e-1f);
    p = fmaf(p, f, 2.0000714765e-1f);
    p = fmaf(p, f, -2.4999993993e-1f);
    p = fmaf(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    y = fmaf(-0.5f, z, y);
    const float r = f + y;
    return fmaf((float)e, 0.693147182464599609375f, r);
}

static inline float m_exp(float x, int det) { return det ? det_expf(x) : expf(x); }
static inline float m_log(float x, int det) { return det ? det_logf(x) : logf(x); }

ORC_API float orc_det_expf(float x) { return det_expf(x); }
ORC_API float orc_det_logf(float x) { return det_logf(x); }

/* ------------------------------------------------------------------------------------------
 * f16-storage emulation (checker-side only).  The device path stores weights of every MFMA
 * GEMM, the activations between kernels and the recurrent state h in IEEE half precision and
 * accumulates in f32 (the reference's own GPU path does the same, CRFModel.cpp:111).  With
 * orc_set_f16_emulation(1) the network functions below round at exactly those points, so that
 * "device vs this oracle" isolates kernel errors from the expected precision noise, while
 * "this oracle vs the f32 reference" measures the precision noise alone.  Default 0 = the plain
 * f32 restatement that is pinned to the compiled reference.
 * ---------------------------------------------------------------------------------------- */
#include <immintrin.h>
static int g_f16 = 0;
ORC_API void orc_set_f16_emulation(int on) { g_f16 = on; }
ORC_API int orc_get_f16_emulation(void) { return g_f16; }
static inline float rf16(float v) { return _cvtsh_ss(_cvtss_sh(v, _MM_FROUND_TO_NEAREST_INT)); }
static void round_f16_inplace(float *x, size_t n) {
#pragma omp parallel for schedule(static)
    for (long i = 0; i < (long)n; ++i) x[i] = rf16(x[i]);
}
static float *rounded_copy(const float *w, size_t n) {
    float *r = (float *)malloc(n * sizeof(float));
    for (size_t i = 0; i < n; ++i) r[i] = rf16(w[i]);
    return r;
}

/* ------------------------------------------------------------------------------------------
 * int8-LSTM emulation (checker-side only).  The reference's GPU path runs the LSTM stack of the
 * tanh-conv models in int8 (nn/ConvStack.cpp:69-73, nn/LSTMStack.cpp:127-211 forward_cutlass with
 * KOI_I8): [W_ih | W_hh] quantised per output row by utils::quantize_tensor on the f16 weights
 * (torch_utils/tensor_utils.cpp:293-300 as called by LSTMStack.cpp:160-168), int8 activations,
 * integer accumulation, floating-point bias / gates / cell state.  Koi is closed, so the
 * activation scale is the device path's own (round(127 v), v in [-1, 1]).  With
 * orc_set_q8_emulation(1) (which implies the f16-storage roundings everywhere else) the LSTM
 * layers below follow that arithmetic: "device int8 path vs this" isolates kernel error from the
 * quantisation noise, "this vs the f32 reference" is the quantisation noise alone.
 * ---------------------------------------------------------------------------------------- */
static int g_q8 = 0;
ORC_API void orc_set_q8_emulation(int on) { g_q8 = on; }
ORC_API int orc_get_q8_emulation(void) { return g_q8; }
static inline int q8q(float v) { /* round(127 clamp(v, -1, 1)), ties to even */
    v = v < -1.0f ? -1.0f : (v > 1.0f ? 1.0f : v);
    return (int)nearbyintf(v * 127.0f);
}
/* utils::quantize_tensor(cat(W_ih, W_hh, 1).half(), 1): every elementwise result is an f16 tensor
 * (tensor_utils.cpp:293-300): scale = f16(128 / max|row|), q = clip(round(f16(w scale)), +-127).
 * q [4C][2C], scale [4C]. */
ORC_API void orc_quantize_lstm_weights(const float *wih, const float *whh, int C, int8_t *q, float *scale) {
    for (int row = 0; row < 4 * C; ++row) {
        float amax = 0.0f;
        for (int k = 0; k < C; ++k) {
            amax = fmaxf(amax, fabsf(rf16(wih[(size_t)row * C + k])));
            amax = fmaxf(amax, fabsf(rf16(whh[(size_t)row * C + k])));
        }
        const float s = amax > 0.0f ? rf16(128.0f / amax) : 1.0f;
        scale[row] = s;
        for (int k = 0; k < 2 * C; ++k) {
            const float w = rf16((k < C) ? wih[(size_t)row * C + k] : whh[(size_t)row * C + (k - C)]);
            float v = nearbyintf(rf16(w * s));
            v = fminf(127.0f, fmaxf(-127.0f, v));
            q[(size_t)row * 2 * C + k] = (int8_t)v;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a12: chunking and stitching (integer, bit-exact contract)
 * ---------------------------------------------------------------------------------------- */

/* read_pipeline/base/chunk.cpp:11-47.  Returns the number of offsets (written to out, at most
 * max_out of them), or -1 for the argument combinations on which the reference throws. */
ORC_API long orc_generate_chunks(uint64_t num_samples, uint64_t chunk_size, uint64_t stride,
                                 uint64_t overlap, uint64_t *out, long max_out) {
    if (num_samples == 0 || stride == 0) {
        return -1;
    }
    if (chunk_size == 0 || (chunk_size % stride) != 0 || chunk_size <= overlap) {
        return -1;
    }
    if ((overlap % stride) != 0) {
        return -1;
    }
    long n = 0;
    if (n < max_out) {
        out[n] = 0;
    }
    ++n;
    uint64_t last = (num_samples > chunk_size) ? (num_samples - chunk_size) : 0;
    const uint64_t mis = last % stride;
    if (mis != 0) {
        last += stride - mis;
    }
    const uint64_t step = chunk_size - overlap;
    uint64_t off = 0;
    while (off + chunk_size < num_samples) {
        off = (off + step < last) ? (off + step) : last;
        if (n < max_out) {
            out[n] = off;
        }
        ++n;
    }
    return n;
}

/* read_pipeline/base/chunk.cpp:49-107 (f3): nearly equal chunks, every interior edge
 * aligned to the stride.  out = pairs (begin, end).  Returns the number of chunks or -1 where the
 * reference throws. */
ORC_API long orc_generate_variable_chunks(uint64_t num_samples, uint64_t chunk_size, uint64_t stride,
                                          uint64_t overlap, uint64_t *out, long max_out) {
    if (num_samples == 0 || stride == 0) return -1;
    if (chunk_size == 0 || (chunk_size % stride) != 0 || chunk_size == stride || chunk_size <= overlap) return -1;
    if ((overlap % stride) != 0 || (stride != 1 && overlap == 0)) return -1;
    uint64_t num_chunks = 1;
    if (num_samples > chunk_size) {
        num_chunks += (uint64_t)ceil((double)(num_samples - chunk_size) / (double)(chunk_size - overlap));
    }
    const uint64_t with_overlaps = num_samples + (num_chunks - 1) * overlap;
    const uint64_t num_longer = with_overlaps % num_chunks;
    const uint64_t adjusted = with_overlaps / num_chunks;
    uint64_t start = 0;
    for (uint64_t i = 0; i < num_chunks; ++i) {
        const uint64_t end = start + adjusted + (i < num_longer ? 1 : 0);
        if ((long)i < max_out) {
            out[2 * i] = start;
            out[2 * i + 1] = end;
        }
        start = end - overlap;
    }
    const uint64_t n = num_chunks < (uint64_t)max_out ? num_chunks : (uint64_t)max_out;
    for (uint64_t i = 1; i < n; ++i) {
        const uint64_t mis = out[2 * i] % stride;
        if (mis != 0) out[2 * i] += stride - mis;
    }
    for (uint64_t i = 0; i + 1 < n; ++i) out[2 * i + 1] -= out[2 * i + 1] % stride;
    return (long)num_chunks;
}

/* read_pipeline/base/stitch.cpp:12-96.  Chunks i = 0..n-1 with input_offset[i],
 * raw_chunk_size[i], per-chunk moves (T_i each, concatenated; moves_off[i] = start), seq/qstr
 * (concatenated; seq_off[i], seq_len[i]).  Outputs the stitched read; returns the stitched
 * sequence length, *n_moves_out the number of moves. */
ORC_API long orc_stitch_chunks(int n_chunks, const int64_t *input_offset,
                               const int64_t *raw_chunk_size, const uint8_t *moves,
                               const int64_t *moves_off, const int64_t *moves_len, const char *seq,
                               const char *qstr, const int64_t *seq_off, const int64_t *seq_len,
                               int64_t raw_samples, int model_stride, char *seq_out,
                               char *qstr_out, uint8_t *moves_out, int64_t *n_moves_out) {
    int64_t so = 0, mo = 0;
    int start_pos = 0, mid_front = 0;
    for (int i = 0; i < n_chunks - 1; ++i) {
        const int overlap_size =
                (int)((raw_chunk_size[i] + input_offset[i]) - input_offset[i + 1]);
        const int overlap_ds = overlap_size / model_stride;
        const int mid_rear = overlap_ds / 2;
        const uint8_t *mv = moves + moves_off[i];
        const int64_t ml = moves_len[i];
        int trim = 0;
        for (int64_t j = ml - mid_rear; j < ml; ++j) {
            trim += mv[j];
        }
        const int end_pos = (int)seq_len[i] - trim;
        const int trimmed = end_pos - start_pos;
        memcpy(seq_out + so, seq + seq_off[i] + start_pos, (size_t)trimmed);
        memcpy(qstr_out + so, qstr + seq_off[i] + start_pos, (size_t)trimmed);
        so += trimmed;
        for (int64_t j = mid_front; j < ml - mid_rear; ++j) {
            moves_out[mo++] = mv[j];
        }
        mid_front = overlap_ds - mid_rear;
        start_pos = 0;
        const uint8_t *nmv = moves + moves_off[i + 1];
        for (int j = 0; j < mid_front; ++j) {
            start_pos += nmv[j];
        }
    }
    const int L = n_chunks - 1;
    const uint8_t *lmv = moves + moves_off[L];
    for (int64_t j = mid_front; j < moves_len[L]; ++j) {
        moves_out[mo++] = lmv[j];
    }
    if (n_chunks == 1) {
        const int64_t keep = raw_samples / model_stride;
        if (mo > keep) {
            mo = keep;
        }
        int end = 0;
        for (int64_t j = 0; j < mo; ++j) {
            end += moves_out[j];
        }
        /* substr(start_pos, end) clamps to the string length */
        int64_t avail = seq_len[L] - start_pos;
        int64_t take = end < avail ? end : avail;
        if (take < 0) take = 0;
        memcpy(seq_out + so, seq + seq_off[L] + start_pos, (size_t)take);
        memcpy(qstr_out + so, qstr + seq_off[L] + start_pos, (size_t)take);
        so += take;
    } else {
        const int64_t take = seq_len[L] - start_pos;
        memcpy(seq_out + so, seq + seq_off[L] + start_pos, (size_t)take);
        memcpy(qstr_out + so, qstr + seq_off[L] + start_pos, (size_t)take);
        so += take;
    }
    /* remove partial stride overhang (stitch.cpp:85-95) */
    if (mo > raw_samples / model_stride) {
        if (moves_out[mo - 1] == 1) {
            --so;
        }
        --mo;
    }
    *n_moves_out = mo;
    return so;
}

/* ------------------------------------------------------------------------------------------
 * a2-a4: network layers (f32, activations NTC)
 * ---------------------------------------------------------------------------------------- */
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/* nn/ConvStack.cpp:103-113 (Conv1d, padding = winlen/2) and :146-163 (activation).
 * in [N,T_in,Cin], W [Cout,Cin,win] (torch layout), out [N,T_out,Cout].
 * act: 0 swish, 1 swish clamp(<=3.5), 2 tanh.  Returns T_out. */
ORC_API int orc_conv1d(const float *in, int N, int T_in, int Cin, const float *W, const float *b,
                       int Cout, int win, int stride, int act, float *out) {
    const int pad = win / 2;
    const int T_out = (T_in + 2 * pad - win) / stride + 1;
    if (!out) {
        return T_out;
    }
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n) {
        for (int t = 0; t < T_out; ++t) {
            const float *xin = in + (size_t)n * T_in * Cin;
            float *y = out + ((size_t)n * T_out + t) * Cout;
            for (int co = 0; co < Cout; ++co) {
                float acc = b ? b[co] : 0.0f;
                const float *w = W + (size_t)co * Cin * win;
                for (int k = 0; k < win; ++k) {
                    const int ti = t * stride + k - pad;
                    if (ti < 0 || ti >= T_in) {
                        continue;
                    }
                    const float *xv = xin + (size_t)ti * Cin;
                    for (int ci = 0; ci < Cin; ++ci) {
                        acc += w[ci * win + k] * xv[ci];
                    }
                }
                float v;
                if (act == 2) {
                    v = tanhf(acc);
                } else {
                    v = acc * sigmoidf_(acc);
                    if (act == 1 && v > 3.5f) {
                        v = 3.5f;
                    }
                }
                y[co] = v;
            }
        }
    }
    return T_out;
}

/* One uni-directional LSTM layer; nn/LSTMStack.cpp:19-27 (torch::nn::LSTM(size,size),
 * batch_first) with torch's gate order i,f,g,o, zero initial state, bias_ih + bias_hh.
 * `reverse` = process t from T-1 down to 0 (the flip()s of LSTMStack.cpp:29-41 expressed in
 * original time).  in/out [N,T,C]. */
ORC_API void orc_lstm_layer(const float *in, int N, int T, int C, const float *Wih,
                            const float *Whh, const float *bih, const float *bhh, int reverse,
                            float *out) {
    /* transposed copies Wt[k][4C]: the loop over the gate index j vectorises while every gate
     * keeps the plain left-to-right sum over k of the scalar formulation */
    const int G = 4 * C;
    float *WiT = (float *)malloc((size_t)C * G * sizeof(float));
    float *WhT = (float *)malloc((size_t)C * G * sizeof(float));
    for (int j = 0; j < G; ++j)
        for (int k = 0; k < C; ++k) {
            WiT[(size_t)k * G + j] = Wih[(size_t)j * C + k];
            WhT[(size_t)k * G + j] = Whh[(size_t)j * C + k];
        }
    const int f16 = g_f16 || g_q8;
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        float *h = (float *)calloc((size_t)C, sizeof(float));
        float *c = (float *)calloc((size_t)C, sizeof(float));
        float *a0 = (float *)malloc((size_t)G * sizeof(float));
        float *a1 = (float *)malloc((size_t)G * sizeof(float));
        for (int step = 0; step < T; ++step) {
            const int t = reverse ? (T - 1 - step) : step;
            const float *x = in + ((size_t)n * T + t) * C;
            for (int j = 0; j < G; ++j) a0[j] = 0.f, a1[j] = 0.f;
            for (int k = 0; k < C; ++k) {
                const float xv = x[k], hv = h[k];
                const float *wi = WiT + (size_t)k * G;
                const float *wh = WhT + (size_t)k * G;
#pragma omp simd
                for (int j = 0; j < G; ++j) {
                    a0[j] += wi[j] * xv;
                    a1[j] += wh[j] * hv;
                }
            }
            float *y = out + ((size_t)n * T + t) * C;
            for (int j = 0; j < C; ++j) {
                const float ig = sigmoidf_((a0[j] + bih[j]) + (a1[j] + bhh[j]));
                const float fg = sigmoidf_((a0[C + j] + bih[C + j]) + (a1[C + j] + bhh[C + j]));
                const float gg = tanhf((a0[2 * C + j] + bih[2 * C + j]) + (a1[2 * C + j] + bhh[2 * C + j]));
                const float og = sigmoidf_((a0[3 * C + j] + bih[3 * C + j]) + (a1[3 * C + j] + bhh[3 * C + j]));
                c[j] = fg * c[j] + ig * gg;
                h[j] = og * tanhf(c[j]);
                if (f16) h[j] = rf16(h[j]); /* the device keeps h_t (state and output) in f16, c in f32 */
                y[j] = h[j];
            }
        }
        free(h);
        free(c);
        free(a0);
        free(a1);
    }
    free(WiT);
    free(WhT);
}

/* The int8 instance of the layer above (nn/LSTMStack.cpp:127-211, KOI_I8): x_t and h_{t-1} enter as
 * round(127 v) int8, the weights as orc_quantize_lstm_weights, the gate pre-activation is
 * float(acc_i32) / (127 scale[row]) + (b_ih + b_hh), gates and cell state are f32.  in [N,T,C] holds values
 * in [-1, 1] (f16 conv / layer outputs or q / 127 of the previous int8 layer); out = q(h) / 127 — or, out_f16
 * (the last layer hands f16 to the CRF head, LSTMStack.cpp:199-207), f16(h) while the recurrence keeps q(h). */
ORC_API void orc_lstm_layer_q8(const float *in, int N, int T, int C, const float *Wih, const float *Whh,
                               const float *bih, const float *bhh, int reverse, int out_f16, float *out) {
    const int G = 4 * C;
    int8_t *q = (int8_t *)malloc((size_t)G * 2 * C);
    float *scale = (float *)malloc((size_t)G * sizeof(float));
    orc_quantize_lstm_weights(Wih, Whh, C, q, scale);
    /* transposed int16 copy Wt[k][4C] so the loop over the gate index vectorises (integer sums are exact in any order) */
    int16_t *Wt = (int16_t *)malloc((size_t)2 * C * G * sizeof(int16_t));
    for (int j = 0; j < G; ++j)
        for (int k = 0; k < 2 * C; ++k) Wt[(size_t)k * G + j] = q[(size_t)j * 2 * C + k];
    float *deq = (float *)malloc((size_t)G * sizeof(float));
    float *bias = (float *)malloc((size_t)G * sizeof(float));
    for (int j = 0; j < G; ++j) {
        deq[j] = 1.0f / (127.0f * scale[j]);
        bias[j] = bih[j] + bhh[j];
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        int16_t *hq = (int16_t *)calloc((size_t)C, sizeof(int16_t));
        int16_t *xq = (int16_t *)malloc((size_t)C * sizeof(int16_t));
        float *c = (float *)calloc((size_t)C, sizeof(float));
        int32_t *acc = (int32_t *)malloc((size_t)G * sizeof(int32_t));
        for (int step = 0; step < T; ++step) {
            const int t = reverse ? (T - 1 - step) : step;
            const float *x = in + ((size_t)n * T + t) * C;
            for (int k = 0; k < C; ++k) xq[k] = (int16_t)q8q(x[k]);
            for (int j = 0; j < G; ++j) acc[j] = 0;
            for (int k = 0; k < 2 * C; ++k) {
                const int32_t v = k < C ? xq[k] : hq[k - C];
                if (v == 0) continue;
                const int16_t *w = Wt + (size_t)k * G;
#pragma omp simd
                for (int j = 0; j < G; ++j) acc[j] += v * (int32_t)w[j];
            }
            float *y = out + ((size_t)n * T + t) * C;
            for (int j = 0; j < C; ++j) {
                const float ig = sigmoidf_(fmaf((float)acc[j], deq[j], bias[j]));
                const float fg = sigmoidf_(fmaf((float)acc[C + j], deq[C + j], bias[C + j]));
                const float gg = tanhf(fmaf((float)acc[2 * C + j], deq[2 * C + j], bias[2 * C + j]));
                const float og = sigmoidf_(fmaf((float)acc[3 * C + j], deq[3 * C + j], bias[3 * C + j]));
                c[j] = fmaf(fg, c[j], ig * gg);
                const float h = og * tanhf(c[j]);
                const int hqv = q8q(h);
                hq[j] = (int16_t)hqv;
                y[j] = out_f16 ? rf16(h) : (float)hqv / 127.0f;
            }
        }
        free(hq);
        free(xq);
        free(c);
        free(acc);
    }
    free(q);
    free(scale);
    free(Wt);
    free(deq);
    free(bias);
}

/* nn/CRFModules.cpp:24-34: y = x W^T (+ b); optional tanh * scale.  W [Cout,Cin]. */
ORC_API void orc_linear(const float *in, long rows, int Cin, const float *W, const float *b,
                        int Cout, int use_tanh, float scale, float *out) {
    /* Wt[k][Cout]: vectorises over the output index, each output keeps the left-to-right k sum */
    float *Wt = (float *)malloc((size_t)Cin * Cout * sizeof(float));
    for (int j = 0; j < Cout; ++j)
        for (int k = 0; k < Cin; ++k) Wt[(size_t)k * Cout + j] = W[(size_t)j * Cin + k];
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const float *x = in + (size_t)r * Cin;
        float *y = out + (size_t)r * Cout;
        for (int j = 0; j < Cout; ++j) y[j] = 0.f;
        for (int k = 0; k < Cin; ++k) {
            const float xv = x[k];
            const float *w = Wt + (size_t)k * Cout;
#pragma omp simd
            for (int j = 0; j < Cout; ++j) y[j] += w[j] * xv;
        }
        for (int j = 0; j < Cout; ++j) {
            float a = y[j];
            if (b) {
                a += b[j];
            }
            if (use_tanh) {
                a = tanhf(a) * scale;
            }
            y[j] = a;
        }
    }
    free(Wt);
}

/* nn/CRFModules.cpp:128-134 */
ORC_API void orc_clamp(float *x, long n, float lo, float hi) {
    for (long i = 0; i < n; ++i) {
        x[i] = x[i] < lo ? lo : (x[i] > hi ? hi : x[i]);
    }
}

/* Same field layout as RefModelDesc in oracle/ref_driver.cpp. */
typedef struct {
    int n_convs;
    int conv_insize[8], conv_size[8], conv_winlen[8], conv_stride[8];
    int conv_act[8];
    int lstm_size, lstm_layers;
    int state_len, outsize;
    int bias;
    int clamp;
    float scale;
    int out_features;
    int num_features;
    int tx_d_model, tx_nhead, tx_depth, tx_dim_ff, tx_win_upper, tx_win_lower, tx_max_seq_len;
    float tx_deepnorm_alpha, tx_theta;
    int up_size, up_scale_factor;
    float crf_scale, crf_blank_score;
    int crf_expand_blanks;
} orc_model_desc;

/* basecall/model/CRFModel.cpp:29-62 (assembly; three head variants) + :127 (Sequential
 * forward).  weights in module.parameters() order = crf_utils.cpp:34-88:
 *   conv{i}.weight, conv{i}.bias ...; rnn{l}.{weight_ih,weight_hh,bias_ih,bias_hh} ...;
 *   linear1.weight [, linear1.bias] [, linear2.weight]
 * in [N, num_features, T_in] f32 (NCT, as the runner hands it over); scores_out [N,T,K].
 * If layer_out != NULL it receives the activations entering the linear head [N,T,C].
 * Returns T. */
ORC_API int orc_lstm_crf_forward(const orc_model_desc *d, const float *const *weights,
                                 const float *in_NCT, int N, int T_in, float *scores_out,
                                 float *layer_out) {
    int wi = 0;
    const int F = d->num_features;
    const int q8 = g_q8;
    const int f16 = g_f16 || q8;
    float *cur = (float *)malloc((size_t)N * T_in * F * sizeof(float));
    for (int n = 0; n < N; ++n)
        for (int f = 0; f < F; ++f)
            for (int t = 0; t < T_in; ++t)
                cur[((size_t)n * T_in + t) * F + f] = in_NCT[((size_t)n * F + f) * T_in + t];
    int T = T_in, C = F;
    for (int i = 0; i < d->n_convs; ++i) {
        const int To = orc_conv1d(cur, N, T, C, NULL, NULL, d->conv_size[i], d->conv_winlen[i],
                                  d->conv_stride[i], d->conv_act[i], NULL);
        float *nxt = (float *)malloc((size_t)N * To * d->conv_size[i] * sizeof(float));
        /* f16 emulation: conv1/conv2 run in f32 with f32 weights (conv1's output never leaves the CU);
         * the last conv is an MFMA GEMM with f16 weights; conv2's and the last conv's outputs are stored f16 */
        const int last = (i == d->n_convs - 1);
        float *wq = (f16 && last) ? rounded_copy(weights[wi], (size_t)d->conv_size[i] * C * d->conv_winlen[i]) : NULL;
        orc_conv1d(cur, N, T, C, wq ? wq : weights[wi], weights[wi + 1], d->conv_size[i], d->conv_winlen[i],
                   d->conv_stride[i], d->conv_act[i], nxt);
        free(wq);
        if (f16 && i >= 1) round_f16_inplace(nxt, (size_t)N * To * d->conv_size[i]);
        wi += 2;
        free(cur);
        cur = nxt;
        T = To;
        C = d->conv_size[i];
    }
    /* LSTMStack(layers, size, reverse_first = true): layer 0 reversed, then alternating. */
    float *buf = (float *)malloc((size_t)N * T * C * sizeof(float));
    /* int8 emulation: every layer int8 when the last convolution ends in tanh and 128 < C (nn/ConvStack.cpp:69-73: the
     * convolution writes the CUTLASS_TNC_I8 layout), otherwise the first layer runs in f16 and its output is converted
     * (LSTMStack.cpp:199-207); the last layer always hands f16 to the head */
    const int q_all = q8 && d->n_convs >= 3 && d->conv_act[d->n_convs - 1] == 2 && C > 128;
    for (int l = 0; l < d->lstm_layers; ++l) {
        const int reverse = (l % 2 == 0);
        if (q8 && (l >= 1 || q_all)) {
            orc_lstm_layer_q8(cur, N, T, C, weights[wi], weights[wi + 1], weights[wi + 2], weights[wi + 3], reverse,
                              l == d->lstm_layers - 1, buf);
            wi += 4;
            float *tmp = cur;
            cur = buf;
            buf = tmp;
            continue;
        }
        float *wq0 = f16 ? rounded_copy(weights[wi], (size_t)4 * C * C) : NULL;
        float *wq1 = f16 ? rounded_copy(weights[wi + 1], (size_t)4 * C * C) : NULL;
        orc_lstm_layer(cur, N, T, C, f16 ? wq0 : weights[wi], f16 ? wq1 : weights[wi + 1], weights[wi + 2],
                       weights[wi + 3], reverse, buf);
        free(wq0);
        free(wq1);
        wi += 4;
        float *tmp = cur;
        cur = buf;
        buf = tmp;
    }
    free(buf);
    if (layer_out) {
        memcpy(layer_out, cur, (size_t)N * T * C * sizeof(float));
    }
    const long rows = (long)N * T;
    const int tanh_x5 = (d->scale == 5.0f);
    if (d->out_features > 0) {
        const int D = d->out_features;
        float *mid = (float *)malloc((size_t)rows * D * sizeof(float));
        const float *w1 = weights[wi++];
        const float *b1 = d->bias ? weights[wi++] : NULL;
        const float *w2 = weights[wi++];
        float *q1 = f16 ? rounded_copy(w1, (size_t)D * C) : NULL;
        float *q2 = f16 ? rounded_copy(w2, (size_t)d->outsize * D) : NULL;
        orc_linear(cur, rows, C, f16 ? q1 : w1, b1, D, 0, 1.0f, mid);
        if (f16) round_f16_inplace(mid, (size_t)rows * D);
        orc_linear(mid, rows, D, f16 ? q2 : w2, NULL, d->outsize, tanh_x5, 5.0f, scores_out);
        free(q1);
        free(q2);
        free(mid);
        if (f16) round_f16_inplace(scores_out, (size_t)rows * d->outsize);
        if (d->clamp) {
            orc_clamp(scores_out, rows * d->outsize, -5.0f, 5.0f);
        }
    } else if (d->conv_size[0] > 4 && d->num_features == 1) {
        const float *w1 = weights[wi++];
        float *q1 = f16 ? rounded_copy(w1, (size_t)d->outsize * C) : NULL;
        orc_linear(cur, rows, C, f16 ? q1 : w1, NULL, d->outsize, tanh_x5, 5.0f, scores_out);
        free(q1);
        if (f16) round_f16_inplace(scores_out, (size_t)rows * d->outsize);
        if (d->clamp) {
            orc_clamp(scores_out, rows * d->outsize, -5.0f, 5.0f);
        }
    } else {
        const float *w1 = weights[wi++];
        const float *b1 = weights[wi++];
        float *q1 = f16 ? rounded_copy(w1, (size_t)d->outsize * C) : NULL;
        orc_linear(cur, rows, C, f16 ? q1 : w1, b1, d->outsize, 1, 5.0f, scores_out);
        free(q1);
        if (f16) round_f16_inplace(scores_out, (size_t)rows * d->outsize);
    }
    free(cur);
    return T;
}

/* ------------------------------------------------------------------------------------------
 * a7-a8: CRF scans and posteriors (one chunk; scores [T,K] with K = 4S, index s*4 + b)
 * ---------------------------------------------------------------------------------------- */
static inline float lse5(float v0, float v1, float v2, float v3, float v4, int det) {
    /* at::logsumexp over {stay, step0..3} (CPUDecoder.cpp:28-34): max, sum of exp, log */
    float m = v0;
    m = v1 > m ? v1 : m;
    m = v2 > m ? v2 : m;
    m = v3 > m ? v3 : m;
    m = v4 > m ? v4 : m;
    float s = m_exp(v0 - m, det);
    s += m_exp(v1 - m, det);
    s += m_exp(v2 - m, det);
    s += m_exp(v3 - m, det);
    s += m_exp(v4 - m, det);
    return m + m_log(s, det);
}

/* decode/CPUDecoder.cpp:43-64 + scan :17-38.  alpha[0] = 0;
 * alpha[t+1][s] = LSE(alpha[t][s] + stay, alpha[t][pred_b(s)] + M[t][s*4+b]),
 * pred_b(s) = (s >> 2) + b*(S/4)  (idx = arange(S).repeat_interleave(4).reshape(4,-1).t()). */
ORC_API void orc_forward_scores(const float *scores, int T, int S, float stay, float *fwd,
                                int det) {
    for (int s = 0; s < S; ++s) {
        fwd[s] = 0.0f;
    }
    const int Q = S / 4;
    for (int t = 0; t < T; ++t) {
        const float *M = scores + (size_t)t * S * 4;
        const float *a = fwd + (size_t)t * S;
        float *o = fwd + (size_t)(t + 1) * S;
        for (int s = 0; s < S; ++s) {
            const int p = s >> 2;
            o[s] = lse5(a[s] + stay, a[p] + M[s * 4 + 0], a[p + Q] + M[s * 4 + 1],
                        a[p + 2 * Q] + M[s * 4 + 2], a[p + 3 * Q] + M[s * 4 + 3], det);
        }
    }
}

/* decode/CPUDecoder.cpp:66-92.  beta[T] = 0;
 * beta[t][s] = LSE(beta[t+1][s] + stay, beta[t+1][succ_b(s)] + M[t][succ_b(s)*4 + (s >> 2(L-1))]),
 * succ_b(s) = ((s << 2) & (S-1)) | b   (idx_T = idx.flatten().argsort(); states idx_T >> 2). */
ORC_API void orc_backward_scores(const float *scores, int T, int S, float stay, float *bwd,
                                 int det) {
    float *last = bwd + (size_t)T * S;
    for (int s = 0; s < S; ++s) {
        last[s] = 0.0f;
    }
    const int Q = S / 4;
    for (int t = T - 1; t >= 0; --t) {
        const float *M = scores + (size_t)t * S * 4;
        const float *a = bwd + (size_t)(t + 1) * S;
        float *o = bwd + (size_t)t * S;
        for (int s = 0; s < S; ++s) {
            const int hi = s / Q;            /* base that falls off the front */
            const int n0 = (s << 2) & (S - 1);
            o[s] = lse5(a[s] + stay, a[n0] + M[(n0 + 0) * 4 + hi], a[n0 + 1] + M[(n0 + 1) * 4 + hi],
                        a[n0 + 2] + M[(n0 + 2) * 4 + hi], a[n0 + 3] + M[(n0 + 3) * 4 + hi], det);
        }
    }
}

/* decode/CPUDecoder.cpp:130: posts = softmax(fwd + bwd, -1) per timestep. */
ORC_API void orc_posts(const float *fwd, const float *bwd, int T, int S, float *posts, int det) {
    for (int t = 0; t <= T; ++t) {
        const float *f = fwd + (size_t)t * S, *b = bwd + (size_t)t * S;
        float *p = posts + (size_t)t * S;
        float m = f[0] + b[0];
        for (int s = 1; s < S; ++s) {
            const float v = f[s] + b[s];
            m = v > m ? v : m;
        }
        float sum = 0.f;
        for (int s = 0; s < S; ++s) {
            p[s] = m_exp((f[s] + b[s]) - m, det);
            sum += p[s];
        }
        for (int s = 0; s < S; ++s) {
            p[s] = p[s] / sum;
        }
    }
}

/* ------------------------------------------------------------------------------------------
 * a9: beam search (decode/beam_search.cpp:125-520), float scores / float posts instantiation
 * ---------------------------------------------------------------------------------------- */
#define ORC_MAX_BEAM 256
#define ORC_FLT_LOWEST (-3.402823466e+38f)

/* beam_search.cpp:104-121, reversed CRC32C polynomial, LSB first */
static uint32_t crc32c_bits(uint32_t crc, uint32_t bits, int nbits) {
    for (int i = 0; i < nbits; ++i) {
        const uint32_t b = (bits ^ crc) & 1u;
        crc >>= 1;
        if (b) {
            crc ^= 0x82f63b78u;
        }
        bits >>= 1;
    }
    return crc;
}

/* beam_search.cpp:42-45 */
static float log_sum_exp2(float x, float y, int det) {
    const float d = fabsf(x - y);
    const float m = x > y ? x : y;
    if (!(d < 17.0f)) {
        return m + 0.0f;
    }
    if (det) {
        return m + det_logf(1.0f + det_expf(-d));
    }
    return m + log1pf(expf(-d));
}

static int cmp_desc(const void *a, const void *b) {
    const float x = *(const float *)a, y = *(const float *)b;
    return (x < y) - (x > y);
}

/* What a decode exercised (for the fixtures' descriptions, tests/golden/make_golden_decoder_full.py): blocks, equal-hash stay /
 * step folds, blocks whose first cut-off kept more than W candidates (bisection), bisections that ran out of guesses, blocks that
 * ended with a full beam.  Summed over every call since the last orc_beam_stats(out, reset = 1); not thread-safe (the batch
 * decode below is: it adds under a critical section). */
static long g_bs_stats[5];
ORC_API void orc_beam_stats(long *out5, int reset) {
    for (int i = 0; i < 5; ++i) {
        if (out5) out5[i] = g_bs_stats[i];
        if (reset) g_bs_stats[i] = 0;
    }
}

ORC_API float orc_beam_search(const float *scores, long block_stride, const float *back_guide,
                              const float *posts, int num_state_bits, int num_blocks,
                              int max_beam_width, float beam_cut, float fixed_stay_score,
                              int32_t *states, uint8_t *moves, float *qual_data, int det) {
    const int S = 1 << num_state_bits;
    const uint32_t mask = (uint32_t)(S - 1);
    const int W = max_beam_width;
    if (W > ORC_MAX_BEAM) {
        return NAN;
    }
    const float log_cut = (beam_cut > 0.0f) ? logf(beam_cut) : 3.402823466e+38f;
    const int CAND = 5 * W;

    /* persistent beam: (T+1) x W of {state, prev, stay} (beam_search.cpp:153) */
    uint16_t *bv_state = (uint16_t *)calloc((size_t)W * (num_blocks + 1), sizeof(uint16_t));
    uint8_t *bv_prev = (uint8_t *)calloc((size_t)W * (num_blocks + 1), 1);
    uint8_t *bv_stay = (uint8_t *)calloc((size_t)W * (num_blocks + 1), 1);

    uint32_t *c_hash = (uint32_t *)calloc((size_t)CAND, 4), *p_hash = (uint32_t *)calloc((size_t)CAND, 4);
    uint16_t *c_state = (uint16_t *)calloc((size_t)CAND, 2), *p_state = (uint16_t *)calloc((size_t)CAND, 2);
    uint8_t *c_prev = (uint8_t *)calloc((size_t)CAND, 1), *p_prev = (uint8_t *)calloc((size_t)CAND, 1);
    uint8_t *c_stay = (uint8_t *)calloc((size_t)CAND, 1), *p_stay = (uint8_t *)calloc((size_t)CAND, 1);
    float *c_score = (float *)calloc((size_t)CAND, 4), *p_score = (float *)calloc((size_t)CAND, 4);

    /* seed (beam_search.cpp:165-198) */
    float thr = ORC_FLT_LOWEST;
    if (W < S) {
        float *tmp = (float *)malloc((size_t)S * 4);
        memcpy(tmp, back_guide, (size_t)S * 4);
        qsort(tmp, (size_t)S, 4, cmp_desc);
        thr = tmp[W - 1];
        free(tmp);
    }
    int ne = 0;
    for (int s = 0; s < S && ne < W; ++s) {
        if (back_guide[s] >= thr) {
            p_hash[ne] = crc32c_bits(0x12345678u, (uint32_t)s, 32);
            p_state[ne] = (uint16_t)s;
            p_prev[ne] = 0;
            p_stay[ne] = 0;
            p_score[ne] = 0.0f;
            ++ne;
        }
    }
    int width = W < S ? W : S;
    for (int i = 0; i < width; ++i) {
        bv_state[i] = p_state[i];
        bv_prev[i] = p_prev[i];
        bv_stay[i] = p_stay[i];
    }

    uint8_t present[4096 / 8];
    long st_merges = 0, st_bisect = 0, st_exhausted = 0, st_full = 0;
    for (int blk = 0; blk < num_blocks; ++blk) {
        const float *bs = scores + (size_t)blk * block_stride;
        const float *bg = back_guide + ((size_t)(blk + 1) << num_state_bits);
        float max_score = ORC_FLT_LOWEST;
        memset(present, 0, sizeof(present));

        /* steps (beam_search.cpp:236-260): slot 4e+b */
        int cnt = 0;
        for (int e = 0; e < width; ++e) {
            const uint32_t ps = p_state[e];
            for (int b = 0; b < 4; ++b) {
                const uint16_t ns = (uint16_t)(((ps << 2) & mask) | (uint32_t)b);
                const uint16_t mi = (uint16_t)(((uint32_t)ns << 2) + ((ps << 2) >> num_state_bits));
                const float sc = p_score[e] + bs[mi] + bg[ns];
                const uint32_t h = crc32c_bits(p_hash[e], (uint32_t)b, 2);
                present[(h & 4095u) >> 3] |= (uint8_t)(1u << (h & 7u));
                c_hash[cnt] = h;
                c_state[cnt] = ns;
                c_prev[cnt] = (uint8_t)e;
                c_stay[cnt] = 0;
                c_score[cnt] = sc;
                max_score = sc > max_score ? sc : max_score;
                ++cnt;
            }
        }
        /* stays + merge (beam_search.cpp:262-308): slot 4*width + e */
        for (int e = 0; e < width; ++e) {
            const float sc = p_score[e] + fixed_stay_score + bg[p_state[e]];
            c_hash[cnt] = p_hash[e];
            c_state[cnt] = p_state[e];
            c_prev[cnt] = (uint8_t)e;
            c_stay[cnt] = 1;
            c_score[cnt] = sc;
            max_score = sc > max_score ? sc : max_score;
            const uint32_t hb = p_hash[e] & 4095u;
            if (present[hb >> 3] & (1u << (hb & 7u))) {
                const int si = (width << 2) + e;
                const int lb = p_state[e] & 3;
                for (int e2 = 0; e2 < width; ++e2) {
                    const int ti = (e2 << 2) | lb;
                    if (c_hash[si] == c_hash[ti]) {
                        const float folded = log_sum_exp2(c_score[si], c_score[ti], det);
                        if (c_score[si] > c_score[ti]) {
                            c_score[si] = folded;
                            c_score[ti] = ORC_FLT_LOWEST;
                        } else {
                            c_score[ti] = folded;
                            c_score[si] = ORC_FLT_LOWEST;
                        }
                        max_score = folded > max_score ? folded : max_score;
                        ++st_merges;
                    }
                }
            }
            ++cnt;
        }

        /* cut-off (beam_search.cpp:310-396) */
        float cutoff = max_score - log_cut;
        int ec = 0;
        for (int i = 0; i < cnt; ++i) ec += (c_score[i] >= cutoff);
        if (ec > W) {
            const int minw = (W * 8) / 10;
            ++st_bisect;
            float lo = cutoff, hi = max_score;
            int guesses = 1;
            while ((ec > W || ec < minw) && guesses < 10) {
                if (ec > W) {
                    lo = cutoff;
                    cutoff = (cutoff + hi) / 2.0f;
                } else {
                    hi = cutoff;
                    cutoff = (cutoff + lo) / 2.0f;
                }
                ec = 0;
                for (int i = 0; i < cnt; ++i) ec += (c_score[i] >= cutoff);
                ++guesses;
            }
            if (guesses == 10) {
                ++st_exhausted;
                cutoff = hi;
                ec = 0;
                for (int i = 0; i < cnt; ++i) ec += (c_score[i] >= cutoff);
            }
            if (ec > W) ec = W;
        }

        /* compaction in slot order (beam_search.cpp:398-409) */
        int wr = 0;
        for (int i = 0; i < cnt; ++i) {
            if (c_score[i] >= cutoff) {
                if (wr < W) {
                    p_hash[wr] = c_hash[i];
                    p_state[wr] = c_state[i];
                    p_prev[wr] = c_prev[i];
                    p_stay[wr] = c_stay[i];
                    p_score[wr] = c_score[i];
                    ++wr;
                } else {
                    break;
                }
            }
        }

        /* last block: best into slot 0 (beam_search.cpp:413-424) */
        if (blk == num_blocks - 1) {
            float best = ORC_FLT_LOWEST;
            int bi = 0;
            for (int i = 0; i < ec; ++i) {
                if (p_score[i] > best) {
                    best = p_score[i];
                    bi = i;
                }
            }
#define SWP(T_, a_, b_) do { T_ t_ = (a_); (a_) = (b_); (b_) = t_; } while (0)
            SWP(uint32_t, p_hash[0], p_hash[bi]);
            SWP(uint16_t, p_state[0], p_state[bi]);
            SWP(uint8_t, p_prev[0], p_prev[bi]);
            SWP(uint8_t, p_stay[0], p_stay[bi]);
            SWP(float, p_score[0], p_score[bi]);
        }

        /* store (beam_search.cpp:426-437) */
        const size_t off = (size_t)(blk + 1) * W;
        for (int i = 0; i < ec; ++i) {
            p_score[i] -= bg[p_state[i]];
            bv_state[off + i] = p_state[i];
            bv_prev[off + i] = p_prev[i];
            bv_stay[off + i] = p_stay[i];
        }
        width = ec;
        st_full += (ec == W);
    }
#pragma omp critical(orc_bs_stats)
    {
        g_bs_stats[0] += num_blocks;
        g_bs_stats[1] += st_merges;
        g_bs_stats[2] += st_bisect;
        g_bs_stats[3] += st_exhausted;
        g_bs_stats[4] += st_full;
    }
    const float final_score = p_score[0];

    /* trace back (beam_search.cpp:448-455) */
    uint8_t ei = 0;
    for (int bi = num_blocks; bi != 0; --bi) {
        const size_t a = (size_t)bi * W + ei;
        states[bi - 1] = (int32_t)bv_state[a];
        moves[bi - 1] = bv_stay[a] ? 0 : 1;
        ei = bv_prev[a];
    }
    moves[0] = 1;

    /* per-block quality (beam_search.cpp:459-517) */
    for (int blk = 0; blk < num_blocks; ++blk) {
        const int state = states[blk];
        states[blk] = state % 4;
        const int base = states[blk];
        const float *tp = posts + ((size_t)(blk + 1) << num_state_bits);
        float prob = tp[state];
        const int l_idx = state >> 2;
        const int r_idx = (state << 2) % S;
        const int msb = S >> 2;
        int sh[8];
        for (int b = 0; b < 4; ++b) {
            sh[2 * b] = l_idx + msb * b;
            sh[2 * b + 1] = r_idx + b;
        }
        for (int i = 0; i < 8; ++i) {
            const int cs = sh[i];
            int count = (cs != state);
            if (count) {
                for (int j = 0; j < i; ++j) {
                    if (sh[j] == cs) {
                        count = 0;
                        break;
                    }
                }
            }
            if (count) {
                prob += tp[cs];
            }
        }
        prob = prob < 0.0f ? 0.0f : (prob > 1.0f ? 1.0f : prob);
        prob = powf(prob, 0.4f);
        const float wrong = (1.0f - prob) / 3.0f;
        for (int b = 0; b < 4; ++b) {
            qual_data[blk * 4 + b] = (b == base) ? prob : wrong;
        }
    }

    free(bv_state); free(bv_prev); free(bv_stay);
    free(c_hash); free(p_hash); free(c_state); free(p_state);
    free(c_prev); free(p_prev); free(c_stay); free(p_stay);
    free(c_score); free(p_score);
    return final_score;
}

/* a10: decode/beam_search.cpp:54-102.  Returns the sequence length; seq/qstr get that many
 * bytes (no terminator). */
ORC_API int orc_generate_sequence(const uint8_t *moves, const int32_t *states, const float *qual,
                                  int num_blocks, float shift, float scale, char *seq,
                                  char *qstr) {
    int len = 0;
    for (int i = 0; i < num_blocks; ++i) len += moves[i];
    float *bp = (float *)calloc((size_t)len + 1, 4), *tp = (float *)calloc((size_t)len + 1, 4);
    static const char alphabet[4] = {'A', 'C', 'G', 'T'};
    int pos = 0;
    for (int blk = 0; blk < num_blocks; ++blk) {
        const int base = states[blk] & 3;
        const int mv = moves[blk];
        const int ppos = pos + (blk == 0 ? 0 : mv - 1);
        bp[ppos] += qual[blk * 4 + base];
        for (int k = 0; k < 4; ++k) tp[ppos] += qual[blk * 4 + k];
        if (blk == 0) {
            seq[pos++] = alphabet[base];
        } else {
            for (int j = 0; j < mv; ++j) seq[pos++] = alphabet[base];
        }
    }
    for (int i = 0; i < len; ++i) {
        float e = 1.0f - (bp[i] / tp[i]);
        e = -10.0f * log10f(e);
        float q = e * scale + shift;
        q = q < 1.0f ? 1.0f : (q > 50.0f ? 50.0f : q);
        qstr[i] = (char)(33.5f + q);
    }
    free(bp);
    free(tp);
    return len;
}

/* One chunk end to end: CPUDecoder.cpp:127-144 + beam_search.cpp:522-606.
 * scores [T,K] f32.  moves[T]; seq/qstr capacity T.  Optional fwd/bwd/posts out ([T+1,S]). */
ORC_API int orc_decode_chunk(const float *scores, int T, int K, int beam_width, float beam_cut,
                             float blank, float q_shift, float q_scale, uint8_t *moves, char *seq,
                             char *qstr, float *bwd_out, float *posts_out, int det) {
    const int S = K / 4;
    int bits = 0;
    while ((1 << bits) < S) ++bits;
    float *fwd = (float *)malloc((size_t)(T + 1) * S * 4);
    float *bwd = (float *)malloc((size_t)(T + 1) * S * 4);
    float *posts = (float *)malloc((size_t)(T + 1) * S * 4);
    orc_forward_scores(scores, T, S, blank, fwd, det);
    orc_backward_scores(scores, T, S, blank, bwd, det);
    orc_posts(fwd, bwd, T, S, posts, det);
    int32_t *states = (int32_t *)malloc((size_t)T * 4);
    float *qual = (float *)malloc((size_t)T * 16);
    orc_beam_search(scores, K, bwd, posts, bits, T, beam_width, beam_cut, blank, states, moves,
                    qual, det);
    const int len = orc_generate_sequence(moves, states, qual, T, q_shift, q_scale, seq, qstr);
    if (bwd_out) memcpy(bwd_out, bwd, (size_t)(T + 1) * S * 4);
    if (posts_out) memcpy(posts_out, posts, (size_t)(T + 1) * S * 4);
    free(fwd); free(bwd); free(posts); free(states); free(qual);
    return len;
}

/* Batch: scores [N,T,K]; outputs [N,T] planes (NUL padded) + seqlen[N]. */
ORC_API void orc_decode_batch(const float *scores, int N, int T, int K, int beam_width,
                              float beam_cut, float blank, float q_shift, float q_scale,
                              uint8_t *moves, char *seq, char *qstr, int *seqlen, int det) {
    memset(seq, 0, (size_t)N * T);
    memset(qstr, 0, (size_t)N * T);
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        seqlen[n] = orc_decode_chunk(scores + (size_t)n * T * K, T, K, beam_width, beam_cut,
                                     blank, q_shift, q_scale, moves + (size_t)n * T,
                                     seq + (size_t)n * T, qstr + (size_t)n * T, NULL, NULL, det);
    }
}

/* ------------------------------------------------------------------------------------------
 * f1: signal scaling in front of the path — ScalerNode.
 * ---------------------------------------------------------------------------------------- */

/* f32 -> f16 bit pattern, round to nearest even (what at::Half / _mm256_cvtps_ph do). */
static uint16_t f32_to_f16_bits(float f) {
    const uint32_t x = f_to_bits(f);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? 0x0200u : 0u));
    }
    if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x33000001u) { /* <= 2^-25: rounds to zero (ties-to-even at exactly 2^-25) */
        return (uint16_t)sign;
    }
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007fffffu) | 0x00800000u; /* 24-bit significand */
    int shift;                                    /* bits to drop */
    uint32_t he;
    if (e < -14) { /* subnormal half */
        shift = 13 + (-14 - e);
        he = 0;
    } else {
        shift = 13;
        he = (uint32_t)(e + 15);
    }
    const uint32_t kept = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    uint32_t r = kept;
    if (rem > half || (rem == half && (kept & 1u))) r += 1u;
    uint32_t h;
    if (he == 0) {
        h = r; /* may carry into the exponent field: correct (becomes the smallest normal) */
    } else {
        h = ((he - 1u) << 10) + r; /* r includes the hidden bit (0x400) */
    }
    return (uint16_t)(sign | h);
}

ORC_API uint16_t orc_f32_to_f16_bits(float f) { return f32_to_f16_bits(f); }

/* torch_utils/tensor_utils.cpp:89-142: x -> f16((float(x) - shift) / scale), f32 arithmetic. */
ORC_API void orc_shift_scale_i16_to_f16(const int16_t *x, long n, float shift, float scale,
                                        uint16_t *out) {
    for (long i = 0; i < n; ++i) {
        const float v = ((float)x[i] - shift) / scale;
        out[i] = f32_to_f16_bits(v);
    }
}

/* torch_utils/tensor_utils.cpp:217-245: "lower" quantiles of an int16 signal by counting.
 * out[i] = smallest value v with #(x <= v) > int(q[i] * (n - 1)). */
ORC_API void orc_quantile_counting(const int16_t *x, long n, const float *q, int nq, float *out) {
    int lo = x[0], hi = x[0];
    for (long i = 1; i < n; ++i) {
        lo = x[i] < lo ? x[i] : lo;
        hi = x[i] > hi ? x[i] : hi;
    }
    const int nb = hi - lo + 1;
    int *cnt = (int *)calloc((size_t)nb, sizeof(int));
    for (long i = 0; i < n; ++i) cnt[x[i] - lo]++;
    for (int i = 1; i < nb; ++i) cnt[i] += cnt[i - 1];
    for (int k = 0; k < nq; ++k) {
        const int thr = (int)(q[k] * (float)(n - 1)); /* float * size_t -> float, truncated */
        for (int i = 0; i < nb; ++i) {
            if (cnt[i] > thr) {
                out[k] = (float)(i + lo);
                break;
            }
        }
    }
    free(cnt);
}

/* read_pipeline/nodes/ScalerNode.cpp:32-40 (med_mad on the int16 tensor): torch's median is the
 * LOWER median (sorted[(n-1)/2]); |x - med| is evaluated in int16 (wraps), its median again lower;
 * mad = that * 1.4826f + 1e-9f in f32. */
static int16_t lower_median_i16(const int16_t *x, long n) {
    int *cnt = (int *)calloc(65536, sizeof(int));
    for (long i = 0; i < n; ++i) cnt[(int)x[i] + 32768]++;
    const long k = (n - 1) / 2;
    long acc = 0;
    int v = 0;
    for (int i = 0; i < 65536; ++i) {
        acc += cnt[i];
        if (acc > k) {
            v = i - 32768;
            break;
        }
    }
    free(cnt);
    return (int16_t)v;
}

ORC_API void orc_med_mad(const int16_t *x, long n, float *med_out, float *mad_out) {
    const int16_t med = lower_median_i16(x, n);
    int16_t *d = (int16_t *)malloc((size_t)n * sizeof(int16_t));
    for (long i = 0; i < n; ++i) {
        const int16_t diff = (int16_t)((int)x[i] - (int)med); /* int16 arithmetic wraps */
        d[i] = (int16_t)(diff < 0 ? -diff : diff);
    }
    const int16_t mad_i = lower_median_i16(d, n);
    free(d);
    *med_out = (float)med;
    *mad_out = (float)mad_i * 1.4826f + 1e-9f;
}

/* ScalerNode.cpp:42-52 (normalisation): shift/scale from two quantiles. */
ORC_API void orc_quantile_shift_scale(const int16_t *x, long n, float quantile_a, float quantile_b,
                                      float shift_multiplier, float scale_multiplier, float *shift,
                                      float *scale) {
    const float q[2] = {quantile_a, quantile_b};
    float r[2];
    orc_quantile_counting(x, n, q, 2, r);
    const float sh = shift_multiplier * (r[0] + r[1]);
    const float sc = scale_multiplier * (r[1] - r[0]);
    *shift = sh > 10.0f ? sh : 10.0f;
    *scale = sc > 1.0f ? sc : 1.0f;
}

/* ScalerNode.cpp:186-215 (strategy PA): the read's calibration (scaling, offset) and the model's
 * standardisation (mean, stdev) give the affine map x -> (x - shift) / scale; the open-pore
 * adjustment (added to shift) is (open_pore_level - expected) / scaling when both are known.
 * expected_open_pore_level <= 0 or a NaN open_pore_level mean "not available". */
ORC_API void orc_pa_shift_scale(float scaling, float offset, int standardise, float mean, float stdev,
                                float open_pore_level, float expected_open_pore_level, float *shift,
                                float *scale, float *open_pore_adjustment) {
    float sc, sh;
    if (standardise) {
        sc = stdev / scaling;
        sh = (mean / scaling) - offset;
    } else {
        sc = 1.f / scaling;
        sh = -1.f * offset;
    }
    float adj = 0.0f;
    if (!isnan(open_pore_level) && expected_open_pore_level > 0.0f) {
        adj = (open_pore_level - expected_open_pore_level) / scaling;
    }
    *shift = sh;
    *scale = sc;
    *open_pore_adjustment = adj;
}

/* torch_utils/trim.cpp:23-60: first window end after the initial peak. */
ORC_API int orc_trim(const float *signal, int n, float threshold, int window_size, int min_elements) {
    const int min_trim = 10;
    const int num_samples = n - min_trim;
    const int num_windows = num_samples / window_size;
    int seen_peak = 0;
    for (int pos = 0; pos < num_windows; ++pos) {
        const int start = pos * window_size + min_trim;
        const int end = start + window_size;
        int cnt = 0;
        for (int i = start; i < end; ++i) cnt += signal[i] > threshold;
        if (cnt > min_elements || seen_peak) {
            seen_peak = 1;
            if (signal[end - 1] > threshold) {
                continue;
            }
            if (end >= num_samples) {
                return min_trim;
            }
            return end;
        }
    }
    return min_trim;
}

/* read_pipeline/nodes/ScalerNode.cpp:58-107 (determine_rna_adapter_pos): end of the DNA adapter of a dRNA read on the raw
 * int16 samples.  Window median = at::median = the LOWER middle element; ties of the five-median buffer resolve as
 * std::minmax_element does (first smallest, last largest). */
ORC_API int orc_rna_adapter_pos(const int16_t *signal, long signal_len) {
    const int kWindowSize = 250, kStride = 50;
    const int kMedianDiff = 125, kMedianDiffForDiffOnlyCheck = 150, kMinMedianForRNASignal = 700;
    int medians[5] = {0, 0, 0, 0, 0};
    int window_pos[5] = {0, 0, 0, 0, 0};
    int median_pos = 0;
    const long signal_start = 1000, signal_end = 3 * signal_len / 4;
    for (long i = signal_start; i < signal_end; i += kStride) {
        int16_t w[250];
        const int len = (int)(signal_len - i < kWindowSize ? signal_len - i : kWindowSize);
        for (int k = 0; k < len; ++k) {         /* insertion sort of the window */
            const int16_t v = signal[i + k];
            int j = k;
            while (j > 0 && w[j - 1] > v) {
                w[j] = w[j - 1];
                --j;
            }
            w[j] = v;
        }
        medians[median_pos % 5] = w[(len - 1) / 2];
        window_pos[median_pos % 5] = median_pos;
        int mn = 0, mx = 0;
        for (int k = 1; k < 5; ++k) {
            if (medians[k] < medians[mn]) mn = k;    /* first smallest */
            if (medians[k] >= medians[mx]) mx = k;   /* last largest */
        }
        const int diff = medians[mx] - medians[mn];
        if (median_pos >= 5 && window_pos[mx] > window_pos[mn] &&
            ((medians[mx] > kMinMedianForRNASignal && diff > kMedianDiff) || diff > kMedianDiffForDiffOnlyCheck))
            return (int)i;
        ++median_pos;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * f2: POD5 signal decompression.  The reference reads signals through
 * pod5_get_read_complete_signal (data_loader/DataLoader.cpp:163-170) of the un-vendored pod5 library
 * (pod5-file-format 0.3.36, cmake/Pod5.cmake).  Its published "VBZ" signal encoding is
 *     int16 samples -> delta (x[i] - x[i-1], x[-1] = 0) -> zig-zag ((d << 1) ^ (d >> 15)) ->
 *     StreamVByte-16 (one control BIT per value, LSB first, 0 = one byte, 1 = two bytes little
 *     endian; the (n + 7) / 8 control bytes come first, the data bytes follow) -> zstd frame.
 * The zstd stage stays on the CPU (libzstd); these two functions restate the svb16 stage, which
 * is what the device kernel replaces.  Parity is anchored on the reference's own POD5 fixtures
 * (tests/data/pod5/ **) through structural invariants: every stream must be consumed exactly and give
 * exactly `samples` values — see tests/test_oracle_pinned.py::test_pod5_*.
 * ---------------------------------------------------------------------------------------- */

/* Returns the number of stream bytes consumed, or -1 if the stream is too short. */
ORC_API long orc_svb16_decode(const uint8_t *stream, long stream_len, long n, int16_t *out) {
    const long nk = (n + 7) / 8;
    if (nk > stream_len) return -1;
    long pos = nk;
    uint16_t prev = 0;
    for (long i = 0; i < n; ++i) {
        const int two = (stream[i >> 3] >> (i & 7)) & 1;
        if (pos + 1 + two > stream_len) return -1;
        uint16_t u = stream[pos];
        if (two) u |= (uint16_t)stream[pos + 1] << 8;
        pos += 1 + two;
        const uint16_t d = (uint16_t)((u >> 1) ^ (uint16_t)(-(int16_t)(u & 1)));  /* zig-zag decode */
        prev = (uint16_t)(prev + d);                                              /* wraps mod 2^16 */
        out[i] = (int16_t)prev;
    }
    return pos;
}

/* Inverse (test fixture generator for round trips); `stream` needs (n + 7) / 8 + 2 n bytes. */
ORC_API long orc_svb16_encode(const int16_t *in, long n, uint8_t *stream) {
    const long nk = (n + 7) / 8;
    memset(stream, 0, (size_t)nk);
    long pos = nk;
    uint16_t prev = 0;
    for (long i = 0; i < n; ++i) {
        const uint16_t d = (uint16_t)((uint16_t)in[i] - prev);
        prev = (uint16_t)in[i];
        const uint16_t z = (uint16_t)((d << 1) ^ (uint16_t)(-(int16_t)(d >> 15)));
        if (z < 256) {
            stream[pos++] = (uint8_t)z;
        } else {
            stream[i >> 3] |= (uint8_t)(1u << (i & 7));
            stream[pos++] = (uint8_t)(z & 0xff);
            stream[pos++] = (uint8_t)(z >> 8);
        }
    }
    return pos;
}

/* ------------------------------------------------------------------------------------------
 * a5-a6: transformer model (sup@v5): conv stack -> TxEncoder x depth -> LinearUpsample ->
 * LinearScaledCRF.  Follows basecall/model/TxModel.cpp:20-41 and nn/TxModules.cpp.
 * ---------------------------------------------------------------------------------------- */
static inline float siluf_(float x) { return x / (1.0f + expf(-x)); }

/* nn/RMSNorm.cpp:14-18 applied to (in + alpha * x): x <- (in + alpha x) * rsqrt(mean(.^2) + 1e-5) * w
 * (nn/TxModules.cpp:881: norm(in + (x * deepnorm_alpha))) */
static void residual_rmsnorm(float *x, const float *in, const float *w, long rows, int C, float alpha) {
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        float *xr = x + (size_t)r * C;
        const float *ir = in + (size_t)r * C;
        float ss = 0.f;
        for (int c = 0; c < C; ++c) {
            xr[c] = ir[c] + xr[c] * alpha;
            ss += xr[c] * xr[c];
        }
        const float rstd = 1.0f / sqrtf(ss / (float)C + 1e-5f);
        for (int c = 0; c < C; ++c) {
            xr[c] = (xr[c] * rstd) * w[c];
            if (g_f16) xr[c] = rf16(xr[c]);
        }
    }
}

/* nn/TxModules.cpp:184-250 (RotaryEmbedding, "evens/odds" = first/second half of head_dim) +
 * :279-426 (MultiHeadAttention CPU path: wqkv, rotary, windowed SDPA evaluated in num_splits = 12
 * query splits whose K/V slice is [qb - win_lower, qe + win_upper) — :398-411 — out_proj). */
static void tx_attention(const float *x, int N, int T, int C, int H, const float *Wqkv, const float *Wo,
                         const float *bo, int win_upper, int win_lower, float theta, float *out) {
    const int D = C / H;
    const long rows = (long)N * T;
    float *qkv = (float *)malloc((size_t)rows * 3 * C * sizeof(float));
    const int f16 = g_f16;
    float *wq = f16 ? rounded_copy(Wqkv, (size_t)3 * C * C) : NULL;
    float *woq = f16 ? rounded_copy(Wo, (size_t)C * C) : NULL;
    orc_linear(x, rows, C, f16 ? wq : Wqkv, NULL, 3 * C, 0, 1.0f, qkv);
    free(wq);
    /* rotary tables */
    float *cs = (float *)malloc((size_t)T * (D / 2) * sizeof(float));
    float *sn = (float *)malloc((size_t)T * (D / 2) * sizeof(float));
    for (int i = 0; i < D / 2; ++i) {
        const float fi = (float)(2 * i);
        const double p = pow((double)theta, (double)(fi / (float)D));
        const float inv = 1.0f / (float)p;
        for (int t = 0; t < T; ++t) {
            const float f = (float)t * inv;
            cs[(size_t)t * (D / 2) + i] = cosf(f);
            sn[(size_t)t * (D / 2) + i] = sinf(f);
        }
    }
#pragma omp parallel for schedule(static)
    for (long r = 0; r < rows; ++r) {
        const int t = (int)(r % T);
        for (int which = 0; which < 2; ++which) {
            for (int h = 0; h < H; ++h) {
                float *v = qkv + (size_t)r * 3 * C + (size_t)which * C + (size_t)h * D;
                for (int i = 0; i < D / 2; ++i) {
                    const float e = v[i], o = v[D / 2 + i];
                    const float c = cs[(size_t)t * (D / 2) + i], s = sn[(size_t)t * (D / 2) + i];
                    v[i] = c * e - s * o;
                    v[D / 2 + i] = s * e + c * o;
                }
            }
        }
    }
    /* device: the QKV GEMM epilogue applies the rotation to the f32 accumulators and stores q, k, v as f16 */
    if (f16) round_f16_inplace(qkv, (size_t)rows * 3 * C);
    const int num_splits = 12;
    int es = (T + num_splits - 1) / num_splits;
    es = (es + 3) / 4 * 4; /* utils::pad_to(div_round_up(T, 12), 4) */
    const float scale = 1.0f / sqrtf((float)D);
    float *att = (float *)malloc((size_t)rows * C * sizeof(float));
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n) {
        for (int i = 0; i < T; ++i) {
            float w[1024];
            const int qe = ((i / es + 1) * es < T) ? (i / es + 1) * es : T;
            const int qb = (i / es) * es;
            int jlo = i - win_upper, jhi = i + win_lower;           /* band: -win_upper <= j - i <= win_lower */
            const int kvb = (qb - win_lower > 0) ? qb - win_lower : 0; /* slice [kvb, kve) */
            const int kve = (qe + win_upper < T) ? qe + win_upper : T;
            if (jlo < kvb) jlo = kvb;
            if (jhi > kve - 1) jhi = kve - 1;
            const int nk = jhi - jlo + 1;
            for (int h = 0; h < H; ++h) {
                const float *q = qkv + ((size_t)n * T + i) * 3 * C + (size_t)h * D;
                float m = -3.0e38f;
                for (int jj = 0; jj < nk; ++jj) {
                    const float *k = qkv + ((size_t)n * T + jlo + jj) * 3 * C + C + (size_t)h * D;
                    float a = 0.f;
                    for (int d = 0; d < D; ++d) a += q[d] * k[d];
                    a *= scale;
                    w[jj] = a;
                    m = a > m ? a : m;
                }
                float sum = 0.f;
                for (int jj = 0; jj < nk; ++jj) {
                    w[jj] = expf(w[jj] - m);
                    sum += w[jj];
                }
                float *o = att + ((size_t)n * T + i) * C + (size_t)h * D;
                for (int d = 0; d < D; ++d) o[d] = 0.f;
                for (int jj = 0; jj < nk; ++jj) {
                    const float *v = qkv + ((size_t)n * T + jlo + jj) * 3 * C + 2 * C + (size_t)h * D;
                    float p = w[jj] / sum;
                    if (f16) p = rf16(p); /* probabilities are the f16 B operand of the PV MFMA */
                    for (int d = 0; d < D; ++d) o[d] += p * v[d];
                }
            }
        }
    }
    if (f16) round_f16_inplace(att, (size_t)rows * C);
    orc_linear(att, rows, C, f16 ? woq : Wo, bo, C, 0, 1.0f, out);
    if (f16) round_f16_inplace(out, (size_t)rows * C);
    free(woq);
    free(qkv); free(cs); free(sn); free(att);
}

/* weights in module.parameters() order (basecall/crf_utils.cpp:100-147): conv{1..n}.{w,b};
 * per layer {wqkv.w, out_proj.w, out_proj.b, fc1.w, fc2.w, norm1.w, norm2.w}; upsample.{w,b}; crf.w.
 * in [N, F, T_in]; scores_out [N, up_scale*T, outsize]; tokens_out (optional) = encoder-stack
 * output [N, T, d_model].  Returns the number of output steps (up_scale * T). */
ORC_API int orc_tx_forward(const orc_model_desc *d, const float *const *weights, const float *in_NCT,
                           int N, int T_in, float *scores_out, float *tokens_out) {
    int wi = 0;
    const int F = d->num_features;
    float *cur = (float *)malloc((size_t)N * T_in * F * sizeof(float));
    for (int n = 0; n < N; ++n)
        for (int f = 0; f < F; ++f)
            for (int t = 0; t < T_in; ++t)
                cur[((size_t)n * T_in + t) * F + f] = in_NCT[((size_t)n * F + f) * T_in + t];
    int T = T_in, C = F;
    for (int i = 0; i < d->n_convs; ++i) {
        const int To = orc_conv1d(cur, N, T, C, NULL, NULL, d->conv_size[i], d->conv_winlen[i],
                                  d->conv_stride[i], d->conv_act[i], NULL);
        float *nxt = (float *)malloc((size_t)N * To * d->conv_size[i] * sizeof(float));
        /* f16 emulation: conv1 is a direct f32 kernel, conv2.. are MFMA GEMMs with f16 weights; every output f16 */
        float *wq = (g_f16 && i >= 1) ? rounded_copy(weights[wi], (size_t)d->conv_size[i] * C * d->conv_winlen[i]) : NULL;
        orc_conv1d(cur, N, T, C, wq ? wq : weights[wi], weights[wi + 1], d->conv_size[i], d->conv_winlen[i],
                   d->conv_stride[i], d->conv_act[i], nxt);
        free(wq);
        if (g_f16) round_f16_inplace(nxt, (size_t)N * To * d->conv_size[i]);
        wi += 2;
        free(cur);
        cur = nxt;
        T = To;
        C = d->conv_size[i];
    }
    if (T > 1024 || C != d->tx_d_model) {
        free(cur);
        return -1;
    }
    const long rows = (long)N * T;
    const int FF = d->tx_dim_ff;
    float *attn = (float *)malloc((size_t)rows * C * sizeof(float));
    float *t1 = (float *)malloc((size_t)rows * 2 * FF * sizeof(float));
    float *t2 = (float *)malloc((size_t)rows * FF * sizeof(float));
    for (int l = 0; l < d->tx_depth; ++l) {
        const float *Wqkv = weights[wi++], *Wo = weights[wi++], *bo = weights[wi++];
        const float *Wfc1 = weights[wi++], *Wfc2 = weights[wi++];
        const float *n1 = weights[wi++], *n2 = weights[wi++];
        tx_attention(cur, N, T, C, d->tx_nhead, Wqkv, Wo, bo, d->tx_win_upper, d->tx_win_lower,
                     d->tx_theta, attn);
        residual_rmsnorm(cur, attn, n1, rows, C, d->tx_deepnorm_alpha);
        /* GatedMLP (nn/TxModules.cpp:140-182): y = first half, gate = second half */
        float *w1q = g_f16 ? rounded_copy(Wfc1, (size_t)2 * FF * C) : NULL;
        float *w2q = g_f16 ? rounded_copy(Wfc2, (size_t)C * FF) : NULL;
        orc_linear(cur, rows, C, g_f16 ? w1q : Wfc1, NULL, 2 * FF, 0, 1.0f, t1);
#pragma omp parallel for schedule(static)
        for (long r = 0; r < rows; ++r)
            for (int j = 0; j < FF; ++j)
                t2[(size_t)r * FF + j] = siluf_(t1[(size_t)r * 2 * FF + FF + j]) * t1[(size_t)r * 2 * FF + j];
        if (g_f16) round_f16_inplace(t2, (size_t)rows * FF); /* SwiGLU output = f16 operand of FC2 */
        orc_linear(t2, rows, FF, g_f16 ? w2q : Wfc2, NULL, C, 0, 1.0f, attn);
        if (g_f16) round_f16_inplace(attn, (size_t)rows * C);
        free(w1q);
        free(w2q);
        residual_rmsnorm(cur, attn, n2, rows, C, d->tx_deepnorm_alpha);
    }
    if (tokens_out) memcpy(tokens_out, cur, (size_t)rows * C * sizeof(float));
    free(attn); free(t1); free(t2);
    /* LinearUpsample (nn/LinearUpsample.cpp:17-23): linear C -> sf*C (+bias), reshape [N, sf*T, C] */
    const int sf = d->up_scale_factor;
    float *up = (float *)malloc((size_t)rows * sf * C * sizeof(float));
    {
        float *wuq = g_f16 ? rounded_copy(weights[wi], (size_t)sf * C * C) : NULL;
        orc_linear(cur, rows, C, g_f16 ? wuq : weights[wi], weights[wi + 1], sf * C, 0, 1.0f, up);
        free(wuq);
        if (g_f16) round_f16_inplace(up, (size_t)rows * sf * C);
    }
    wi += 2;
    free(cur);
    /* LinearScaledCRF (nn/TxModules.cpp:1010-1016): weight *= scale once, then linear, no bias */
    const int K = d->outsize;
    float *ws = (float *)malloc((size_t)K * C * sizeof(float));
    for (size_t i = 0; i < (size_t)K * C; ++i) {
        ws[i] = weights[wi][i] * d->crf_scale;
        if (g_f16) ws[i] = rf16(ws[i]);
    }
    orc_linear(up, rows * sf, C, ws, NULL, K, 0, 1.0f, scores_out);
    if (g_f16) round_f16_inplace(scores_out, (size_t)rows * sf * K);
    free(ws); free(up);
    return T * sf;
}
