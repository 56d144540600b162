// dorado_amd/csrc/engine_tx.hip — transformer-model (sup@v5) half of the engine: weight layout
// conversion, workspace and launch sequence of
//   conv1 (direct) -> conv2..5 (implicit-im2col MFMA GEMMs) -> depth x [QKV GEMM + RoPE epilogue ->
//   sliding-window attention -> out-proj GEMM (+bias) -> residual RMSNorm -> FC1 GEMM + SwiGLU
//   epilogue -> FC2 GEMM -> residual RMSNorm] -> upsample GEMM (+bias) -> CRF GEMM (x scale)
// i.e. basecall/model/TxModel.cpp:20-41 + nn/TxModules.cpp:859-906 without libtorch / Koi.
#include "engine.h"

extern "C" int mibc_launch_conv1_tx(hipStream_t s, const half_t *x, const float *w, const float *b,
                                    half_t *out, const float *ss, int N, int T_in, int Tpitch, int pad_out,
                                    int C1, int act);
extern "C" int mibc_launch_window_attention(hipStream_t s, const half_t *qkv, half_t *out, int N, int T,
                                            int C, int H, int win_upper, int win_lower);
extern "C" int mibc_launch_window_attention_v2(hipStream_t s, const half_t *qk, const half_t *vT, half_t *out,
                                               int N, int T, int C, int H, int ld, int win_upper,
                                               int win_lower);
extern "C" int mibc_launch_residual_rmsnorm(hipStream_t s, const half_t *in, half_t *x, const float *w,
                                            long rows, int C, float alpha);

static std::vector<half_t> f2h(const float *p, size_t n, float mul = 1.0f) {
    std::vector<half_t> v(n);
    for (size_t i = 0; i < n; ++i) v[i] = (half_t)(p[i] * mul);
    return v;
}

int tx_create(mibc_engine *e, const mibc_model_desc &d, const float *const *weights, int n_weights) {
    const int C = d.tx_d_model;
    if (d.n_convs < 2 || d.num_features != 1 || d.conv_insize[0] != 1 || d.conv_size[0] != 64 ||
        d.conv_winlen[0] != 5 || d.conv_stride[0] != 1)
        return fail(e, MIBC_NOT_SUPPORTED, "tx conv front-end must start with 1->64 (w5,s1)");
    if (d.conv_size[d.n_convs - 1] != C || C % 128 != 0 || d.tx_nhead * 64 != C)
        return fail(e, MIBC_NOT_SUPPORTED, "tx: d_model must be nhead*64 and a multiple of 128");
    if ((C != 128 && C != 256 && C != 512) || d.tx_dim_ff % 64 != 0 || d.up_size != C)
        return fail(e, MIBC_NOT_SUPPORTED, "tx: d_model must be 128/256/512");
    if (64 + d.tx_win_upper + d.tx_win_lower > 320)
        return fail(e, MIBC_NOT_SUPPORTED, "tx: attention window too large");
    if (n_weights != 2 * d.n_convs + 7 * d.tx_depth + 3)
        return fail(e, MIBC_ERR_ARG, "tx: unexpected number of weight tensors");
    auto &tx = e->tx;
    e->is_tx = true;
    tx.D = C;
    tx.H = d.tx_nhead;
    tx.FF = d.tx_dim_ff;
    tx.depth = d.tx_depth;
    tx.sf = d.up_scale_factor;
    tx.conv_stride = 1;
    for (int i = 0; i < d.n_convs; ++i) tx.conv_stride *= d.conv_stride[i];
    HIP_OK(e, hipEventCreate(&tx.ev_conv));
    HIP_OK(e, hipEventCreate(&tx.ev_layers));
    int wi = 0;
    {  // conv1 [C1][1][5] -> [5][C1]
        const int C1 = d.conv_size[0];
        const float *W = weights[wi++], *B = weights[wi++];
        std::vector<float> w((size_t)5 * C1), b(B, B + C1);
        for (int co = 0; co < C1; ++co)
            for (int k = 0; k < 5; ++k) w[(size_t)k * C1 + co] = W[co * 5 + k];
        if (mibc_upload(e, &tx.c1w, w) || mibc_upload(e, &tx.c1b, b)) return MIBC_ERR_HIP;
    }
    for (int i = 1; i < d.n_convs; ++i) {
        mibc_engine::TxConv cv;
        cv.cin = d.conv_insize[i];
        cv.cout = d.conv_size[i];
        cv.cout_pad = (cv.cout + 127) / 128 * 128;
        cv.w = d.conv_winlen[i];
        cv.stride = d.conv_stride[i];
        cv.pad = cv.w / 2;
        cv.act = d.conv_act[i];
        const int K = cv.w * cv.cin;
        if (K % 32 != 0 || cv.cin != d.conv_size[i - 1])
            return fail(e, MIBC_NOT_SUPPORTED, "tx conv: W*Cin must be a multiple of 32");
        const float *W = weights[wi++], *B = weights[wi++];
        std::vector<half_t> w((size_t)cv.cout_pad * K, (half_t)0.0f);
        for (int co = 0; co < cv.cout; ++co)
            for (int ci = 0; ci < cv.cin; ++ci)
                for (int k = 0; k < cv.w; ++k)
                    w[(size_t)co * K + k * cv.cin + ci] = (half_t)W[((size_t)co * cv.cin + ci) * cv.w + k];
        std::vector<float> b((size_t)cv.cout_pad, 0.0f);
        for (int co = 0; co < cv.cout; ++co) b[co] = B[co];
        if (mibc_upload(e, &cv.wB, w) || mibc_upload(e, &cv.bias, b)) return MIBC_ERR_HIP;
        tx.convs.push_back(cv);
    }
    const int FF = tx.FF;
    for (int l = 0; l < tx.depth; ++l) {
        mibc_engine::TxLayer L;
        const float *wqkv = weights[wi++], *wo = weights[wi++], *bo = weights[wi++];
        const float *wfc1 = weights[wi++], *wfc2 = weights[wi++];
        const float *n1 = weights[wi++], *n2 = weights[wi++];
        if (mibc_upload(e, &L.wqkv, f2h(wqkv, (size_t)3 * C * C))) return MIBC_ERR_HIP;
        if (mibc_upload(e, &L.wo, f2h(wo, (size_t)C * C))) return MIBC_ERR_HIP;
        if (mibc_upload(e, &L.bo, std::vector<float>(bo, bo + C))) return MIBC_ERR_HIP;
        // fc1 [2FF][C]: rows 0..FF-1 = y, FF..2FF-1 = gate (nn/TxModules.cpp:171-175).  Interleave
        // in blocks of 64 so a 128-column GEMM tile holds 64 y + the matching 64 gate features.
        std::vector<half_t> w1((size_t)2 * FF * C);
        for (int blk = 0; blk < FF / 64; ++blk)
            for (int j = 0; j < 64; ++j)
                for (int k = 0; k < C; ++k) {
                    w1[((size_t)blk * 128 + j) * C + k] = (half_t)wfc1[((size_t)blk * 64 + j) * C + k];
                    w1[((size_t)blk * 128 + 64 + j) * C + k] = (half_t)wfc1[((size_t)FF + blk * 64 + j) * C + k];
                }
        if (mibc_upload(e, &L.wfc1, w1)) return MIBC_ERR_HIP;
        if (mibc_upload(e, &L.wfc2, f2h(wfc2, (size_t)C * FF))) return MIBC_ERR_HIP;
        if (mibc_upload(e, &L.n1, std::vector<float>(n1, n1 + C))) return MIBC_ERR_HIP;
        if (mibc_upload(e, &L.n2, std::vector<float>(n2, n2 + C))) return MIBC_ERR_HIP;
        if (tx_layer_supported(C, FF))
            if (mibc_upload(e, &L.wimg, tx_layer_image(wo, wfc1, wfc2, FF))) return MIBC_ERR_HIP;
        tx.layers.push_back(L);
    }
    {
        const float *wup = weights[wi++], *bup = weights[wi++], *wcrf = weights[wi++];
        if (mibc_upload(e, &tx.wup, f2h(wup, (size_t)tx.sf * C * C))) return MIBC_ERR_HIP;
        if (mibc_upload(e, &tx.bup, std::vector<float>(bup, bup + (size_t)tx.sf * C))) return MIBC_ERR_HIP;
        // LinearScaledCRF: weight *= scale once (nn/TxModules.cpp:1010-1016)
        if (mibc_upload(e, &tx.wcrf, f2h(wcrf, (size_t)e->K * C, d.crf_scale))) return MIBC_ERR_HIP;
    }
    {  // rotary table [max_seq_len][32] {cos, sin}  (nn/TxModules.cpp:184-220, head_dim 64)
        const int L = d.tx_max_seq_len > 0 ? d.tx_max_seq_len : 2048;
        std::vector<float> tab((size_t)L * 32 * 2);
        for (int i = 0; i < 32; ++i) {
            const float fi = (float)(2 * i);
            const double p = pow((double)d.tx_theta, (double)(fi / 64.0f));
            const float inv = 1.0f / (float)p;
            for (int t = 0; t < L; ++t) {
                const float f = (float)t * inv;
                tab[((size_t)t * 32 + i) * 2] = cosf(f);
                tab[((size_t)t * 32 + i) * 2 + 1] = sinf(f);
            }
        }
        if (mibc_upload(e, &tx.rope, tab)) return MIBC_ERR_HIP;
    }
    return MIBC_OK;
}

void tx_free_ws(mibc_engine *e) {
    auto &tx = e->tx;
    for (auto p : tx.cbuf)
        if (p) (void)hipFree(p);
    tx.cbuf.clear();
    tx.cbuf_bytes.clear();
    tx.ctp.clear();
    tx.ct.clear();
    void *ptrs[] = {tx.x, tx.qkv, tx.attn, tx.tmp, tx.ff, tx.up, tx.vT};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    tx.x = tx.qkv = tx.attn = tx.tmp = tx.ff = tx.up = tx.vT = nullptr;
}

void tx_destroy(mibc_engine *e) {
    auto &tx = e->tx;
    tx_free_ws(e);
    void *ptrs[] = {tx.c1w, tx.c1b, tx.wup, tx.wcrf, tx.bup, tx.rope};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto &c : tx.convs) {
        if (c.wB) (void)hipFree(c.wB);
        if (c.bias) (void)hipFree(c.bias);
    }
    for (auto &l : tx.layers) {
        void *lp[] = {l.wqkv, l.wo, l.wfc1, l.wfc2, l.bo, l.n1, l.n2, l.wimg};
        for (void *p : lp)
            if (p) (void)hipFree(p);
    }
    if (tx.ev_conv) (void)hipEventDestroy(tx.ev_conv);
    if (tx.ev_layers) (void)hipEventDestroy(tx.ev_layers);
}

// number of tokens (encoder time steps) for a chunk of T_in samples
int tx_tokens(const mibc_engine *e, int T_in) {
    int T = T_in;  // conv1: w5 s1 pad2 keeps T
    for (const auto &c : e->tx.convs) T = (T + 2 * c.pad - c.w) / c.stride + 1;
    return T;
}

size_t tx_bytes_per_chunk(const mibc_engine *e, int T_in) {
    const auto &tx = e->tx;
    size_t b = (size_t)T_in * 2;
    int T = T_in, C = e->d.conv_size[0];
    for (size_t i = 0; i <= tx.convs.size(); ++i) {
        const int pad_next = (i < tx.convs.size()) ? tx.convs[i].pad : 0;
        b += (size_t)(T + 2 * pad_next) * C * 2;
        if (i < tx.convs.size()) {
            T = (T + 2 * tx.convs[i].pad - tx.convs[i].w) / tx.convs[i].stride + 1;
            C = tx.convs[i].cout;
        }
    }
    const size_t R = (size_t)T;
    b += R * (tx.D /* x */ + 3 * tx.D /* qkv */ + tx.D /* vT */ + tx.D /* attn */ + tx.D /* tmp */ + tx.FF + tx.sf * tx.D) * 2;   // == tx_reserve
    b += 3 * R * tx.sf;
    return b;
}

int tx_reserve(mibc_engine *e, int N_max, int T_in, size_t *total_out) {
    auto &tx = e->tx;
    const size_t N = (size_t)N_max;
    size_t total = 0;
    auto alloc = [&](half_t **p, size_t halfs) -> int {
        HIP_OK(e, hipMalloc((void **)p, halfs * 2));
        total += halfs * 2;
        return 0;
    };
    // conv buffers: buffer i holds conv(i+1)'s output with the NEXT conv's padding rows (zeroed by tx_set_geometry)
    int T = T_in, C = e->d.conv_size[0];
    for (size_t i = 0; i < tx.convs.size(); ++i) {
        const int tp = T + 2 * tx.convs[i].pad;
        half_t *p = nullptr;
        if (alloc(&p, (N * tp + 64) * (size_t)C)) return MIBC_ERR_MEM;
        tx.cbuf.push_back(p);
        tx.cbuf_bytes.push_back((N * tp + 64) * (size_t)C * 2);
        T = (T + 2 * tx.convs[i].pad - tx.convs[i].w) / tx.convs[i].stride + 1;
        C = tx.convs[i].cout;
    }
    if (T > (e->d.tx_max_seq_len > 0 ? e->d.tx_max_seq_len : 2048))
        return fail(e, MIBC_ERR_ARG, "RotE - maximum sequence length exceeded - chunksize too large");
    const size_t R = N * (size_t)T;
    if (alloc(&tx.x, R * tx.D)) return MIBC_ERR_MEM;
    if (alloc(&tx.qkv, R * 3 * tx.D)) return MIBC_ERR_MEM;
    if (alloc(&tx.vT, R * tx.D)) return MIBC_ERR_MEM;      // used when the call's token count is a multiple of 128
    if (alloc(&tx.attn, R * tx.D)) return MIBC_ERR_MEM;
    if (alloc(&tx.tmp, R * tx.D)) return MIBC_ERR_MEM;
    if (alloc(&tx.ff, R * tx.FF)) return MIBC_ERR_MEM;
    if (alloc(&tx.up, R * tx.sf * tx.D)) return MIBC_ERR_MEM;
    *total_out = total;
    return MIBC_OK;
}

int tx_set_geometry(mibc_engine *e, int T_in) {
    auto &tx = e->tx;
    tx.ctp.clear();
    tx.ct.clear();
    int T = T_in;
    for (size_t i = 0; i < tx.convs.size(); ++i) {
        tx.ctp.push_back(T + 2 * tx.convs[i].pad);
        tx.ct.push_back(T);
        T = (T + 2 * tx.convs[i].pad - tx.convs[i].w) / tx.convs[i].stride + 1;
        // the pad rows sit where this chunk length puts them, in stream order: the whole buffer the first time, afterwards
        // only the rows that are padding in the new pitch (N strips of 2 pad rows + the slack rows; the data rows are
        // rewritten by every call) — see set_geometry in engine.hip
        const size_t Tc = (size_t)tx.ct[i], pad = (size_t)tx.convs[i].pad, tp = Tc + 2 * pad;
        const size_t rowb = (size_t)(i == 0 ? e->d.conv_size[0] : tx.convs[i - 1].cout) * sizeof(half_t);
        if (e->T_in_res == 0 || pad == 0) {
            HIP_OK(e, hipMemsetAsync(tx.cbuf[i], 0, tx.cbuf_bytes[i], e->stream));
        } else {
            char *base = (char *)tx.cbuf[i];
            HIP_OK(e, hipMemsetAsync(base, 0, pad * rowb, e->stream));
            HIP_OK(e, hipMemset2DAsync(base + (pad + Tc) * rowb, tp * rowb, 0, 2 * pad * rowb, (size_t)e->N_res, e->stream));
            HIP_OK(e, hipMemsetAsync(base + (size_t)e->N_res * tp * rowb, 0, 64 * rowb, e->stream));
        }
    }
    if (T > (e->d.tx_max_seq_len > 0 ? e->d.tx_max_seq_len : 2048))
        return fail(e, MIBC_ERR_ARG, "RotE - maximum sequence length exceeded - chunksize too large");
    tx.T_tok = T;
    return MIBC_OK;
}

static int gemm(mibc_engine *e, const half_t *A, const half_t *B, const float *bias, half_t *out, long M,
                int cols, int K, int act, int epi = 0, int ncols_valid = 0, long out_stride = 0) {
    GemmArgs g{};
    g.A = A;
    g.B = B;
    g.bias = bias;
    g.out = out;
    g.M = (int)M;
    g.Ncols = cols;
    g.K = K;
    g.a_div = 1 << 30;  // m / a_div == 0
    g.a_outer = 0;
    g.a_inner = K;
    g.o_div = 1 << 30;
    g.o_outer = 0;
    g.o_inner = out_stride ? out_stride : cols;
    g.act = act;
    g.epi_mode = epi;
    g.ncols_valid = ncols_valid;
    if (epi == 1) {
        g.rope = e->tx.rope;
        g.rope_T = e->tx.T_tok;
        g.rope_cols = 2 * e->tx.D;
        g.vT = (e->tx.T_tok % 128 == 0) ? e->tx.vT : nullptr;
    }
    return mibc_launch_gemm_tn(e->stream, &g);
}

int tx_run_network(mibc_engine *e, const half_t *in_dev, int N, int T_in) {
    auto &tx = e->tx;
    const mibc_model_desc &d = e->d;
    const bool prof = e->profile > 0;
    if (prof) e->ps ^= 1;
    if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_START], e->stream));
    // conv1 -> cbuf[0] (with conv2's pad rows)
    if (mibc_launch_conv1_tx(e->stream, in_dev, tx.c1w, tx.c1b, tx.cbuf[0], e->in_ss, N, T_in, tx.ctp[0],
                             tx.convs[0].pad, d.conv_size[0], d.conv_act[0]) != 0)
        return fail(e, MIBC_NOT_SUPPORTED, "tx conv1 shape");
    for (size_t i = 0; i < tx.convs.size(); ++i) {
        const auto &cv = tx.convs[i];
        const int T_inp = tx.ct[i];
        const int T_out = (T_inp + 2 * cv.pad - cv.w) / cv.stride + 1;
        const bool last = (i + 1 == tx.convs.size());
        GemmArgs g{};
        g.A = tx.cbuf[i];
        g.B = cv.wB;
        g.bias = cv.bias;
        g.M = N * T_out;
        g.Ncols = cv.cout_pad;
        g.K = cv.w * cv.cin;
        g.a_div = T_out;
        g.a_outer = (long)tx.ctp[i] * cv.cin;
        g.a_inner = (long)cv.stride * cv.cin;
        g.o_div = T_out;
        if (last) {
            g.out = tx.x;
            g.o_outer = (long)T_out * cv.cout;
        } else {
            g.out = tx.cbuf[i + 1] + (size_t)tx.convs[i + 1].pad * cv.cout;
            g.o_outer = (long)tx.ctp[i + 1] * cv.cout;
        }
        g.o_inner = cv.cout;
        g.act = cv.act;
        g.ncols_valid = (cv.cout_pad != cv.cout) ? cv.cout : 0;
        if (mibc_launch_gemm_tn(e->stream, &g) != 0) return fail(e, MIBC_NOT_SUPPORTED, "tx conv gemm shape");
    }
    if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_CONV], e->stream));
    const int T = tx.T_tok, C = tx.D;
    const long R = (long)N * T;
    for (int l = 0; l < tx.depth; ++l) {
        const auto &L = tx.layers[l];
        MibcRange r_layer(e, "TxEncoder");
        if (gemm(e, tx.x, L.wqkv, nullptr, tx.qkv, R, 3 * C, C, -1, /*rope*/ 1) != 0)
            return fail(e, MIBC_NOT_SUPPORTED, "tx qkv gemm");
        int arc;
        if (tx.vT && T % 128 == 0)
            arc = mibc_launch_window_attention_v2(e->stream, tx.qkv, tx.vT, tx.attn, N, T, C, tx.H, 3 * C,
                                                  d.tx_win_upper, d.tx_win_lower);
        else
            arc = mibc_launch_window_attention(e->stream, tx.qkv, tx.attn, N, T, C, tx.H, d.tx_win_upper,
                                               d.tx_win_lower);
        if (arc != 0) return fail(e, MIBC_NOT_SUPPORTED, "tx attention shape");
        // everything behind the attention in ONE launch (txlayer.hip) when the model has the sup@v5 width
        if (L.wimg != nullptr && tx.fused &&
            mibc_launch_tx_layer(e->stream, tx.attn, tx.x, L.wimg, L.bo, L.n1, L.n2, d.tx_deepnorm_alpha, R, tx.FF, 3) == 0)
            continue;
        if (gemm(e, tx.attn, L.wo, L.bo, tx.tmp, R, C, C, -1) != 0) return fail(e, MIBC_NOT_SUPPORTED, "tx out_proj");
        if (mibc_launch_residual_rmsnorm(e->stream, tx.tmp, tx.x, L.n1, R, C, d.tx_deepnorm_alpha) != 0)
            return fail(e, MIBC_NOT_SUPPORTED, "tx rmsnorm");
        if (gemm(e, tx.x, L.wfc1, nullptr, tx.ff, R, 2 * tx.FF, C, -1, /*swiglu*/ 2, 0, tx.FF) != 0)
            return fail(e, MIBC_NOT_SUPPORTED, "tx fc1");
        if (gemm(e, tx.ff, L.wfc2, nullptr, tx.tmp, R, C, tx.FF, -1) != 0) return fail(e, MIBC_NOT_SUPPORTED, "tx fc2");
        if (mibc_launch_residual_rmsnorm(e->stream, tx.tmp, tx.x, L.n2, R, C, d.tx_deepnorm_alpha) != 0)
            return fail(e, MIBC_NOT_SUPPORTED, "tx rmsnorm");
    }
    // upsample: [R][C] -> [R][sf*C] == [N][sf*T][C]
    if (gemm(e, tx.x, tx.wup, tx.bup, tx.up, R, tx.sf * C, C, -1) != 0) return fail(e, MIBC_NOT_SUPPORTED, "tx upsample");
    if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_LSTM0], e->stream));
    e->lstm_out = tx.x;
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

// CRF head for chunks [n0, n0+ns): scores_out [ns][sf*T][K]
int tx_run_head(mibc_engine *e, int N, int n0, int ns, half_t *scores_out) {
    (void)N;
    auto &tx = e->tx;
    const long rows = (long)ns * tx.sf * tx.T_tok;
    const half_t *A = tx.up + (size_t)n0 * tx.sf * tx.T_tok * tx.D;
    if (gemm(e, A, tx.wcrf, nullptr, scores_out, rows, e->K, tx.D, -1) != 0)
        return fail(e, MIBC_NOT_SUPPORTED, "tx crf gemm");
    return MIBC_OK;
}

#ifdef MIBC_DEBUG_KERNELS
// Test entry (debug library only): the fused layer tail (txlayer.hip) against the five-launch path it replaces
// (gemm256 out-proj -> residual_rmsnorm -> gemm256 FC1 + SwiGLU -> gemm256 FC2 -> residual_rmsnorm) on the same
// pseudo-random attn / x / weights.  mode 3 = whole tail, 1 = out-proj + norm 1 only (contract: bit-identical), 2 = MLP +
// norm 2 only (x holds x1).  Reports the number of differing output halfs, the largest difference, the rms difference,
// the largest |reference| and both run times.
MIBC_HOOK int mibc_debug_txlayer_compare(long R, int FF, int mode, int iters, long long *ndiff, float *maxdiff, float *rmsdiff,
                                          float *amax_out, float *ms_fused, float *ms_unfused, uint16_t *dump_fused,
                                          uint16_t *dump_ref) {
    const int C = 512;
    const int dbg_bits = mode >> 8;      // timing ablations of the fused kernel (results wrong), see TxLayerArgs::dbg
    mode &= 0xff;
    if (!tx_layer_supported(C, FF) || (mode != 1 && mode != 2 && mode != 3 && mode != 6)) return 1;
    auto lcg = [](uint32_t &s) {
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 9) & 0x7fff) / 16384.0f - 1.0f;   // [-1, 1)
    };
    uint32_t seed = 4242u + (uint32_t)(R + 3 * FF + mode);
    std::vector<float> wo((size_t)C * C), w1((size_t)2 * FF * C), w2((size_t)C * FF), bo(C), n1(C), n2(C);
    for (auto &v : wo) v = lcg(seed) * 0.06f;
    for (auto &v : w1) v = lcg(seed) * 0.06f;
    for (auto &v : w2) v = lcg(seed) * 0.03f;
    for (auto &v : bo) v = lcg(seed) * 0.1f;
    for (auto &v : n1) v = 1.0f + 0.1f * lcg(seed);
    for (auto &v : n2) v = 1.0f + 0.1f * lcg(seed);
    std::vector<half_t> hattn((size_t)R * C), hx((size_t)R * C);
    for (auto &v : hattn) v = (half_t)lcg(seed);
    for (auto &v : hx) v = (half_t)(lcg(seed) * 0.7f);
    mibc_engine eng{};
    mibc_engine *e = &eng;
    e->tx.D = C;
    e->tx.FF = FF;
    half_t *attn = nullptr, *x0 = nullptr, *xa = nullptr, *xb = nullptr, *tmp = nullptr, *ff = nullptr;
    half_t *dwo = nullptr, *dw1 = nullptr, *dw2 = nullptr, *dimg = nullptr;
    float *dbo = nullptr, *dn1 = nullptr, *dn2 = nullptr;
    const size_t xb_bytes = (size_t)R * C * 2;
    auto cleanup = [&]() {
        for (void *q : {(void *)attn, (void *)x0, (void *)xa, (void *)xb, (void *)tmp, (void *)ff, (void *)dwo, (void *)dw1,
                        (void *)dw2, (void *)dimg, (void *)dbo, (void *)dn1, (void *)dn2})
            if (q) (void)hipFree(q);
    };
    std::vector<half_t> w1i((size_t)2 * FF * C);   // the unfused path's 64 y | 64 gate interleave (tx_create)
    for (int blk = 0; blk < FF / 64; ++blk)
        for (int j = 0; j < 64; ++j)
            for (int k = 0; k < C; ++k) {
                w1i[((size_t)blk * 128 + j) * C + k] = (half_t)w1[((size_t)blk * 64 + j) * C + k];
                w1i[((size_t)blk * 128 + 64 + j) * C + k] = (half_t)w1[((size_t)FF + blk * 64 + j) * C + k];
            }
    if (hipMalloc((void **)&attn, xb_bytes) != hipSuccess || hipMalloc((void **)&x0, xb_bytes) != hipSuccess ||
        hipMalloc((void **)&xa, xb_bytes) != hipSuccess || hipMalloc((void **)&xb, xb_bytes) != hipSuccess ||
        hipMalloc((void **)&tmp, xb_bytes) != hipSuccess || hipMalloc((void **)&ff, (size_t)R * FF * 2) != hipSuccess ||
        mibc_upload(e, &dwo, f2h(wo.data(), wo.size())) || mibc_upload(e, &dw1, w1i) ||
        mibc_upload(e, &dw2, f2h(w2.data(), w2.size())) || mibc_upload(e, &dimg, tx_layer_image(wo.data(), w1.data(), w2.data(), FF)) ||
        mibc_upload(e, &dbo, bo) || mibc_upload(e, &dn1, n1) || mibc_upload(e, &dn2, n2)) {
        cleanup();
        return -1;
    }
    (void)hipMemcpy(attn, hattn.data(), xb_bytes, hipMemcpyHostToDevice);
    (void)hipMemcpy(x0, hx.data(), xb_bytes, hipMemcpyHostToDevice);
    const float alpha = 2.4494897f;
    auto unfused = [&](half_t *x) -> int {
        if (mode & 1) {
            if (gemm(e, attn, dwo, dbo, tmp, R, C, C, -1) != 0) return 1;
            if (mibc_launch_residual_rmsnorm(e->stream, tmp, x, dn1, R, C, alpha) != 0) return 1;
        }
        if (mode & 2) {
            if (gemm(e, x, dw1, nullptr, ff, R, 2 * FF, C, -1, 2, 0, FF) != 0) return 1;
            if (gemm(e, ff, dw2, nullptr, tmp, R, C, FF, -1) != 0) return 1;
            if (mode & 4) {   // raw FC2 result
                if (hipMemcpyAsync(x, tmp, xb_bytes, hipMemcpyDeviceToDevice, e->stream) != hipSuccess) return 1;
            } else if (mibc_launch_residual_rmsnorm(e->stream, tmp, x, dn2, R, C, alpha) != 0) return 1;
        }
        return 0;
    };
    auto fused = [&](half_t *x) -> int {
        return mibc_launch_tx_layer(e->stream, attn, x, dimg, dbo, dn1, dn2, alpha, R, FF, mode | (dbg_bits << 8));
    };
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms[2] = {0, 0};
    int rc = 0;
    for (int which = 0; which < 2 && rc == 0; ++which) {
        half_t *x = which ? xb : xa;
        (void)hipMemcpy(x, x0, xb_bytes, hipMemcpyDeviceToDevice);
        rc = which ? unfused(x) : fused(x);                       // the compared result: exactly one application
        if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) rc = 3;
        (void)hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters && rc == 0; ++i) rc = which ? unfused(x0) : fused(x0);   // timing only (x0 is scratch here)
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[which], e0, e1);
        ms[which] /= (float)(iters > 0 ? iters : 1);
        if (which == 0) (void)hipMemcpy(x0, hx.data(), xb_bytes, hipMemcpyHostToDevice);   // restore for the reference run
    }
    if (rc != 0 || hipDeviceSynchronize() != hipSuccess) {
        cleanup();
        return -2;
    }
    std::vector<half_t> oa((size_t)R * C), ob((size_t)R * C);
    (void)hipMemcpy(oa.data(), xa, xb_bytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(ob.data(), xb, xb_bytes, hipMemcpyDeviceToHost);
    if (dump_fused) memcpy(dump_fused, oa.data(), xb_bytes);
    if (dump_ref) memcpy(dump_ref, ob.data(), xb_bytes);
    long long nd = 0;
    float md = 0.0f, amax = 0.0f;
    double sq = 0.0;
    for (size_t i = 0; i < oa.size(); ++i) {
        uint16_t a, b;
        memcpy(&a, &oa[i], 2);
        memcpy(&b, &ob[i], 2);
        const float d = fabsf((float)oa[i] - (float)ob[i]);
        if (a != b) ++nd;
        if (!(d <= md)) md = d;       // NaN-propagating max
        sq += (double)d * d;
        amax = fmaxf(amax, fabsf((float)ob[i]));
    }
    *ndiff = nd;
    *maxdiff = md;
    *rmsdiff = (float)sqrt(sq / (double)oa.size());
    *amax_out = amax;
    *ms_fused = ms[0];
    *ms_unfused = ms[1];
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    cleanup();
    return 0;
}
#endif   // MIBC_DEBUG_KERNELS
