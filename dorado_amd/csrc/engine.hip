// dorado_amd/csrc/engine.hip — the C-ABI (include/mibc.h): engine lifetime, weight layout
// conversion, device workspace, stage orchestration on one HIP stream, timing and parity taps.
//
// This is the device-side half of what basecall::CudaCaller does in the reference
// (dorado/basecall/CudaCaller.cpp:149-200 ctor, :224-271 call_chunks, :323-369 memory model,
// :552-569 forward timing) minus its libtorch/Koi dependencies.  The thread/queue half
// (per-device FIFO, runners, pinned batch buffers) lives in dorado_amd/host/.
#include "engine.h"

static int check_cluster_error(mibc_engine *e, int slot = 2);
#ifndef MIBC_CL_WROW
#define MIBC_CL_WROW 0     // layout of the cluster LSTM's weight slices: must equal lstm_cluster.hip's
#endif
static int set_geometry(mibc_engine *e, int T_in);

#include <dlfcn.h>
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        // whatever the process already carries first (a profiler may preload its own), then rocprofv3's implementation
        // (rocprofiler-sdk: the only one its --marker-trace records), then roctracer's
        push = (int (*)(const char *))dlsym(RTLD_DEFAULT, "roctxRangePushA");
        pop = (int (*)())dlsym(RTLD_DEFAULT, "roctxRangePop");
        if (push && pop) return;
        void *h = nullptr;
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (h) {
            push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
            pop = (int (*)())dlsym(h, "roctxRangePop");
        }
    }
};
Roctx &roctx() {
    static Roctx r;
    return r;
}
}  // namespace
// utils::quantize_tensor(cat(W_ih, W_hh, 1).half(), 1) (torch_utils/tensor_utils.cpp:293-300 as called by
// nn/LSTMStack.cpp:160-168), with the reference's arithmetic: everything happens on f16 tensors, i.e. every elementwise
// result is rounded to f16 — scale = f16(128 / max|row|), q = clip(round_half_even(f16(w * scale)), +-127).  Host only (no
// device needed): pinned bit for bit against the compiled reference in tests/test_oracle_pinned.py.
extern "C" int mibc_quantize_lstm_weights(const float *wih, const float *whh, int C, int8_t *q, float *scale) {
    if (!wih || !whh || !q || !scale || C <= 0) return MIBC_ERR_ARG;
    for (int row = 0; row < 4 * C; ++row) {
        float amax = 0.0f;
        for (int k = 0; k < C; ++k) {
            amax = fmaxf(amax, fabsf((float)(half_t)wih[(size_t)row * C + k]));
            amax = fmaxf(amax, fabsf((float)(half_t)whh[(size_t)row * C + k]));
        }
        const float s = amax > 0.0f ? (float)(half_t)(128.0f / amax) : 1.0f;
        scale[row] = s;
        for (int k = 0; k < 2 * C; ++k) {
            const float w = (float)(half_t)((k < C) ? wih[(size_t)row * C + k] : whh[(size_t)row * C + (k - C)]);
            float v = nearbyintf((float)(half_t)(w * s));
            v = fminf(127.0f, fmaxf(-127.0f, v));
            q[(size_t)row * 2 * C + k] = (int8_t)v;
        }
    }
    return MIBC_OK;
}

MibcRange::MibcRange(const mibc_engine *e, const char *name) : on(false) {
    if (e && e->profile >= 2 && roctx().push && roctx().pop) {
        (void)roctx().push(name);
        on = true;
    }
}
MibcRange::~MibcRange() {
    if (on) (void)roctx().pop();
}

std::string &mibc_gerr() {
    static thread_local std::string g;
    return g;
}

extern "C" int mibc_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        return 0;
    }
    return n;
}

// free / total bytes of a device (utils::available_memory, torch_utils/cuda_utils.cpp:250-262)
extern "C" int mibc_device_memory(int device_id, size_t *free_bytes, size_t *total_bytes) {
    if (hipSetDevice(device_id) != hipSuccess) return MIBC_ERR_HIP;
    size_t f = 0, t = 0;
    if (hipMemGetInfo(&f, &t) != hipSuccess) return MIBC_ERR_HIP;
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return MIBC_OK;
}

extern "C" const char *mibc_last_error(const mibc_engine *e) {
    return e ? e->err.c_str() : g_err.c_str();
}

// [cols][K] row-major -> [cols/16][K/32][64 lanes][8]: lane holds W[16*ct + (lane & 15)][32*ks + 8*(lane >> 4) ..]
static std::vector<half_t> to_frag16(const std::vector<half_t> &w, int cols, int K) {
    std::vector<half_t> f((size_t)cols * K);
    const int KT = K / 32;
    for (int ct = 0; ct < cols / 16; ++ct)
        for (int ks = 0; ks < KT; ++ks)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i)
                    f[(((size_t)ct * KT + ks) * 64 + lane) * 8 + i] =
                            w[(size_t)(16 * ct + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + i];
    return f;
}

template <typename T>
static int upload(mibc_engine *e, T **dst, const std::vector<T> &src) {
    HIP_OK(e, hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_OK(e, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// common tail of mibc_create: hand the engine over, or destroy everything that was built so far
static int finish_create(mibc_engine *e, int rc, mibc_engine **out) {
    if (rc != MIBC_OK) {
        g_err = e->err.empty() ? g_err : e->err;
        mibc_destroy(e);
        return rc;
    }
    *out = e;
    return MIBC_OK;
}

extern "C" int mibc_create(int device_id, const mibc_model_desc *desc, const float *const *weights,
                           int n_weights, mibc_engine **out) {
    if (!desc || !weights || !out) {
        return fail(nullptr, MIBC_ERR_ARG, "null argument");
    }
    const mibc_model_desc &d = *desc;
    if (d.tx_d_model > 0) {
        const int S_ = 1 << (2 * d.state_len);
        if (S_ != 64 && S_ != 256 && S_ != 1024) return fail(nullptr, MIBC_NOT_SUPPORTED, "state_len must be 3, 4 or 5");
        if (d.outsize != 4 * S_) return fail(nullptr, MIBC_ERR_ARG, "outsize must be 4^(state_len+1)");
        if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, MIBC_ERR_HIP, "hipSetDevice failed");
        mibc_engine *e = new mibc_engine();
        e->device = device_id;
        e->d = d;
        e->C = d.tx_d_model;
        e->S = S_;
        e->K = 4 * S_;
        auto body = [&]() -> int {
            HIP_OK(e, hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
            for (auto &set : e->ev)
                for (auto &ev : set) HIP_OK(e, hipEventCreate(&ev));
            const int rc = tx_create(e, d, weights, n_weights);
            if (rc != MIBC_OK) return rc;
            const char *tp = getenv("MIBC_TAPS");
            e->taps = tp ? atoi(tp) : 0;
            // weight uploads are null-stream copies from pageable memory: hipMemcpy may return once the data
            // sits in the staging buffer, and the engine's stream is non-blocking, so finish them here
            HIP_OK(e, hipDeviceSynchronize());
            return MIBC_OK;
        };
        return finish_create(e, body(), out);
    }
    if (d.n_convs != 3 || d.num_features != 1 || d.conv_insize[0] != 1 || d.conv_size[0] != 16 ||
        d.conv_size[1] != 16 || d.conv_winlen[0] != 5 || d.conv_winlen[1] != 5 ||
        d.conv_stride[0] != 1 || d.conv_stride[1] != 1 || d.conv_insize[2] != 16) {
        return fail(nullptr, MIBC_NOT_SUPPORTED,
                    "conv front-end must be 1->16 (w5,s1) ->16 (w5,s1) ->C (v4 LSTM-CRF models)");
    }
    const int C = d.lstm_size;
    if (d.conv_size[2] != C || mibc_lstm_rows_per_wg(C) == 0) {
        return fail(nullptr, MIBC_NOT_SUPPORTED,
                    "lstm_size must be one of 96/128/256/384/512/768/1024 for now");
    }
    const int S = 1 << (2 * d.state_len);
    if (S != 64 && S != 256 && S != 1024) {
        return fail(nullptr, MIBC_NOT_SUPPORTED, "state_len must be 3, 4 or 5");
    }
    if (d.outsize != 4 * S) {
        return fail(nullptr, MIBC_ERR_ARG, "outsize must be 4^(state_len+1)");
    }
    const bool two_stage = d.out_features > 0;
    const bool v4_single = !two_stage && (d.conv_size[0] > 4 && d.num_features == 1);
    int expect = 6 + 4 * d.lstm_layers + 1;
    if (two_stage) expect += 1 + (d.bias ? 1 : 0);
    else if (!v4_single) expect += 1;
    if (n_weights != expect) {
        return fail(nullptr, MIBC_ERR_ARG, "unexpected number of weight tensors: got " +
                                                   std::to_string(n_weights) + ", expected " +
                                                   std::to_string(expect));
    }
    if (d.lstm_quant && ((C != 128 && C != 256 && C != 384 && C != 512 && C != 768 && C != 1024) || d.lstm_layers < 2))
        return fail(nullptr, MIBC_NOT_SUPPORTED, "lstm_quant: lstm_size 128 / 256 / 384 / 512 / 768 / 1024 and at least two layers");
    if (two_stage && (d.out_features % 128 != 0)) {
        return fail(nullptr, MIBC_NOT_SUPPORTED, "out_features must be a multiple of 128");
    }
    if (hipSetDevice(device_id) != hipSuccess) {
        return fail(nullptr, MIBC_ERR_HIP, "hipSetDevice failed");
    }
    mibc_engine *e = new mibc_engine();
    e->device = device_id;
    e->d = d;
    e->C = C;
    e->S = S;
    e->K = 4 * S;
    e->stride = d.conv_stride[2];
    e->pad3 = d.conv_winlen[2] / 2;
    // every failure past this point goes through finish_create -> mibc_destroy (no leaked stream / events / weights)
    auto body = [&]() -> int {
    HIP_OK(e, hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    for (auto &set : e->ev)
                for (auto &ev : set) HIP_OK(e, hipEventCreate(&ev));

    int wi = 0;
    // conv1 [16][1][5] -> [k][co]; conv2 [16][16][5] -> [k][ci][co]
    {
        const float *W = weights[wi++], *B = weights[wi++];
        std::vector<float> w(5 * 16), b(B, B + 16);
        for (int co = 0; co < 16; ++co)
            for (int k = 0; k < 5; ++k) w[k * 16 + co] = W[co * 5 + k];
        if (upload(e, &e->w1, w) || upload(e, &e->b1, b)) return MIBC_ERR_HIP;
    }
    {
        const float *W = weights[wi++], *B = weights[wi++];
        std::vector<float> w(5 * 16 * 16), b(B, B + 16);
        for (int co = 0; co < 16; ++co)
            for (int ci = 0; ci < 16; ++ci)
                for (int k = 0; k < 5; ++k) w[(k * 16 + ci) * 16 + co] = W[(co * 16 + ci) * 5 + k];
        if (upload(e, &e->w2, w) || upload(e, &e->b2, b)) return MIBC_ERR_HIP;
    }
    // conv3 [C][16][W3] -> GEMM B matrix [C][K3pad], k = w*16 + ci (im2col row order)
    {
        const int W3 = d.conv_winlen[2];
        e->K3 = W3 * 16;
        e->K3pad = (e->K3 + 31) / 32 * 32;
        const float *W = weights[wi++], *B = weights[wi++];
        const int Cpad = (C + 127) / 128 * 128;  // GEMM column tiles are 128 wide (C = 96: masked)
        std::vector<half_t> w((size_t)Cpad * e->K3pad, (half_t)0.0f);
        for (int co = 0; co < C; ++co)
            for (int ci = 0; ci < 16; ++ci)
                for (int k = 0; k < W3; ++k)
                    w[(size_t)co * e->K3pad + k * 16 + ci] = (half_t)W[((size_t)co * 16 + ci) * W3 + k];
        std::vector<float> b((size_t)Cpad, 0.0f);
        for (int co = 0; co < C; ++co) b[co] = B[co];
        if (upload(e, &e->w3, w) || upload(e, &e->b3, b)) return MIBC_ERR_HIP;
        if (e->K3pad % 32 == 0 && C % 64 == 0)
            if (upload(e, &e->w3f, to_frag16(w, C, e->K3pad))) return MIBC_ERR_HIP;
    }
    // LSTM layers: [W_ih | W_hh] in MFMA-fragment order + summed biases in D-register order
    // the reference quantises EVERY LSTM layer when the convolution in front hands over tanh outputs (nn/ConvStack.cpp:72:
    // CUTLASS_TNC_I8 for Activation::TANH — the v4.3 LSTM-CRF models), and keeps the first layer in f16 otherwise (:73, LSTMStack.cpp:199-207)
    // (the reference selects that layout for 128 < lstm_size <= 1024 only, nn/ConvStack.cpp:69-73; at lstm_size <= 128 its
    // quantised path is the NTC forward_quantized scheme, a different one: first layer f16 here)
    const bool q_all = d.lstm_quant && d.n_convs >= 3 && d.conv_act[d.n_convs - 1] == MIBC_ACT_TANH && e->C > 128;
    for (int l = 0; l < d.lstm_layers; ++l) {
        const float *Wih = weights[wi++], *Whh = weights[wi++], *bih = weights[wi++],
                    *bhh = weights[wi++];
        const int KS = 2 * C / 16;
        std::vector<half_t> wf((size_t)4 * C * 2 * C);
        for (int j = 0; j < C / 32; ++j)
            for (int ks = 0; ks < KS; ++ks)
                for (int g = 0; g < 4; ++g)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int i = 0; i < 8; ++i) {
                            const int row = g * C + 32 * j + (lane & 31);
                            const int k = ks * 16 + 8 * (lane >> 5) + i;
                            const float v = (k < C) ? Wih[(size_t)row * C + k]
                                                    : Whh[(size_t)row * C + (k - C)];
                            wf[((((size_t)j * KS + ks) * 4 + g) * 64 + lane) * 8 + i] = (half_t)v;
                        }
        half_t *dw16 = nullptr;
        if (C <= 384) {
            const int KS32 = 2 * C / 32;
            std::vector<half_t> w16((size_t)4 * C * 2 * C);
            for (int j = 0; j < C / 16; ++j)
                for (int ks = 0; ks < KS32; ++ks)
                    for (int g = 0; g < 4; ++g)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int i = 0; i < 8; ++i) {
                                const int row = g * C + 16 * j + (lane & 15);
                                const int k = ks * 32 + 8 * (lane >> 4) + i;
                                const float v = (k < C) ? Wih[(size_t)row * C + k]
                                                        : Whh[(size_t)row * C + (k - C)];
                                w16[((((size_t)j * KS32 + ks) * 4 + g) * 64 + lane) * 8 + i] = (half_t)v;
                            }
            if (upload(e, &dw16, w16)) return MIBC_ERR_HIP;
        }
        e->lstm_w16.push_back(dw16);
        std::vector<float> bn((size_t)4 * C);
        for (int j = 0; j < C / 32; ++j)
            for (int g = 0; g < 4; ++g)
                for (int h = 0; h < 32; ++h)
                    bn[((size_t)j * 4 + g) * 32 + h] = bih[g * C + 32 * j + h] + bhh[g * C + 32 * j + h];
        half_t *dw = nullptr;
        float *dbn = nullptr;
        if (upload(e, &dw, wf) || upload(e, &dbn, bn)) return MIBC_ERR_HIP;
        e->lstm_w.push_back(dw);
        e->lstm_bn.push_back(dbn);
        int8_t *dwq = nullptr;
        float *ddeq = nullptr;
        if (d.lstm_quant && (l >= 1 || q_all) && C <= 384) {
            // utils::quantize_tensor(cat(W_ih, W_hh, 1), 1) (torch_utils/tensor_utils.cpp:293-300, LSTMStack.cpp:165-172):
            // per output row scale = 128 / max|row|, round to nearest even, clip +-127
            const int KS64 = 2 * C / 64;
            std::vector<float> scale((size_t)4 * C);
            std::vector<int8_t> qrow((size_t)4 * C * 2 * C);      // [4C][2C], k < C: W_ih, else W_hh
            mibc_quantize_lstm_weights(Wih, Whh, C, qrow.data(), scale.data());
            std::vector<int8_t> wq((size_t)4 * C * 2 * C);
            for (int j = 0; j < C / 16; ++j)
                for (int ks = 0; ks < KS64; ++ks)
                    for (int g = 0; g < 4; ++g)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int i = 0; i < 16; ++i) {
                                const int row = g * C + 16 * j + (lane & 15);
                                const int k = ks * 64 + 16 * (lane >> 4) + i;
                                wq[((((size_t)j * KS64 + ks) * 4 + g) * 64 + lane) * 16 + i] = qrow[(size_t)row * 2 * C + k];
                            }
            std::vector<float> deq((size_t)4 * C);
            for (int j = 0; j < C / 32; ++j)
                for (int g = 0; g < 4; ++g)
                    for (int h = 0; h < 32; ++h) deq[((size_t)j * 4 + g) * 32 + h] = 1.0f / (127.0f * scale[g * C + 32 * j + h]);
            if (upload(e, &dwq, wq) || upload(e, &ddeq, deq)) return MIBC_ERR_HIP;
        }
        e->lstm_wq.push_back(dwq);
        e->lstm_deq.push_back(ddeq);
        if (C == 512 || C == 768 || C == 1024) {
            // cluster kernel (lstm_cluster.hip): member j of a cluster owns hidden units [128 j, 128 j + 128); its
            // weight slab (pass p, k-slab ks) is stored as the exact LDS image it is DMA'd into: 256 gate rows
            // [hidden group hg][gate g][32 units] x 32 k, 64-byte rows with the 16-byte column XOR-swizzled by
            // (-(row >> 2)) & 3 (the read side applies the same XOR; round 6: the swizzle that is conflict-free for the lane groups
            // of a ds_read_b128 under the 16x16x32 fragment mapping — (row >> 2) & 3, right for the 32x32x16 mapping, is 2-way
            // conflicted there: SQ_LDS_BANK_CONFLICT 0.22 of the CU's cycles, profiles/r06_e_pmc_clock_sup_n8192.json)
            const int KCL = C / 128, KSL = 2 * C / 32;
            std::vector<half_t> wcl((size_t)4 * C * 2 * C);
            std::vector<float> bcl((size_t)4 * C);
            for (int jm = 0; jm < KCL; ++jm)
                for (int p = 0; p < 2; ++p) {
                    for (int row = 0; row < 256; ++row) {
                        const int hgi = row >> 7, g = (row >> 5) & 3, hl = row & 31;
                        const int hidden = jm * 128 + p * 64 + hgi * 32 + hl;
                        const size_t G = (size_t)g * C + hidden;
                        bcl[(((size_t)(jm * 2 + p) * 2 + hgi) * 4 + g) * 32 + hl] = bih[G] + bhh[G];
                        for (int ks = 0; ks < KSL; ++ks) {
                            for (int kk = 0; kk < 32; ++kk) {
                                const int k = ks * 32 + kk;
                                const float v = (k < C) ? Wih[G * C + k] : Whh[G * C + (k - C)];
#if MIBC_CL_WROW   // row-major [pass][256 rows][2C]: the kernel's DMA pieces apply the swizzle (lstm_cluster.hip)
                                wcl[((size_t)(jm * 2 + p) * 256 + row) * 2 * C + k] = (half_t)v;
#else
                                half_t *dst = wcl.data() + ((((size_t)(jm * 2 + p)) * KSL + ks) * 256 + row) * 32;
                                dst[(((kk >> 3) ^ ((0 - (row >> 2)) & 3)) << 3) + (kk & 7)] = (half_t)v;
#endif
                            }
                        }
                    }
                }
            half_t *dwcl = nullptr;
            float *dbcl = nullptr;
            if (upload(e, &dwcl, wcl) || upload(e, &dbcl, bcl)) return MIBC_ERR_HIP;
            e->lstm_wcl.push_back(dwcl);
            e->lstm_bcl.push_back(dbcl);
            int8_t *dwclq = nullptr;
            float *dbclq = nullptr, *ddqcl = nullptr;
            if (d.lstm_quant && (l >= 1 || q_all)) {
                // the quantised instance of the cluster kernel: the same per-row quantisation (utils::quantize_tensor,
                // LSTMStack.cpp:165-172), slab images of 256 gate rows x 64 k (64-byte rows, same XOR swizzle of the 16-byte
                // column); accumulators start from round(bias / deq[row]) (an int32: the rounding is < deq / 2 ~ 1e-5 of
                // a pre-activation), deq[row] = 1 / (127 * row scale)
                const int KSQ = 2 * C / 64;
                std::vector<float> scale((size_t)4 * C);
                std::vector<int8_t> qrow((size_t)4 * C * 2 * C);      // [4C][2C], k < C: W_ih, else W_hh
                mibc_quantize_lstm_weights(Wih, Whh, C, qrow.data(), scale.data());
                std::vector<int8_t> wclq((size_t)4 * C * 2 * C);
                std::vector<float> bclq((size_t)4 * C), dqcl((size_t)4 * C);
                for (int jm = 0; jm < KCL; ++jm)
                    for (int p = 0; p < 2; ++p)
                        for (int row = 0; row < 256; ++row) {
                            const int hgi = row >> 7, g = (row >> 5) & 3, hl = row & 31;
                            const int hidden = jm * 128 + p * 64 + hgi * 32 + hl;
                            const size_t G = (size_t)g * C + hidden;
                            const float deq = 1.0f / (127.0f * scale[G]);
                            const size_t bi = (((size_t)(jm * 2 + p) * 2 + hgi) * 4 + g) * 32 + hl;
                            dqcl[bi] = deq;
                            const int bq = (int)lrintf(fminf(fmaxf((bih[G] + bhh[G]) / deq, -2.0e9f), 2.0e9f));
                            memcpy(&bclq[bi], &bq, 4);
                            for (int ks = 0; ks < KSQ; ++ks) {
#if MIBC_CL_WROW
                                for (int kk = 0; kk < 64; ++kk)
                                    wclq[((size_t)(jm * 2 + p) * 256 + row) * 2 * C + (size_t)ks * 64 + kk] = qrow[G * 2 * C + (size_t)ks * 64 + kk];
#else
                                int8_t *dst = wclq.data() + ((((size_t)(jm * 2 + p)) * KSQ + ks) * 256 + row) * 64;
                                for (int kk = 0; kk < 64; ++kk)
                                    dst[(((kk >> 4) ^ ((0 - (row >> 2)) & 3)) << 4) + (kk & 15)] = qrow[G * 2 * C + (size_t)ks * 64 + kk];
#endif
                            }
                        }
                if (upload(e, &dwclq, wclq) || upload(e, &dbclq, bclq) || upload(e, &ddqcl, dqcl)) return MIBC_ERR_HIP;
            }
            e->lstm_wclq.push_back(dwclq);
            e->lstm_bclq.push_back(dbclq);
            e->lstm_dqcl.push_back(ddqcl);
        }
    }
    if (!e->lstm_wcl.empty()) {
        HIP_OK(e, hipMalloc((void **)&e->lstm_zero, (size_t)256 * C * 2));
        HIP_OK(e, hipMemset(e->lstm_zero, 0, (size_t)256 * C * 2));
        HIP_OK(e, hipMalloc((void **)&e->cl_err, 16));
        HIP_OK(e, hipMemset(e->cl_err, 0, 16));
        HIP_OK(e, hipHostMalloc((void **)&e->cl_err_host, 3 * 16, hipHostMallocDefault));
        memset(e->cl_err_host, 0, 3 * 16);
    }
    // head (basecall/model/CRFModel.cpp:43-61)
    const int tanh_x5 = (d.scale == 5.0f) ? 3 : -1;
    auto to_half = [](const float *p, size_t n) {
        std::vector<half_t> v(n);
        for (size_t i = 0; i < n; ++i) v[i] = (half_t)p[i];
        return v;
    };
    if (two_stage) {
        const int D = d.out_features;
        if (upload(e, &e->head_w1, to_half(weights[wi++], (size_t)D * C))) return MIBC_ERR_HIP;
        if (d.bias) {
            const float *B = weights[wi++];
            if (upload(e, &e->head_b1, std::vector<float>(B, B + D))) return MIBC_ERR_HIP;
        }
        if (upload(e, &e->head_w2, to_half(weights[wi++], (size_t)e->K * D))) return MIBC_ERR_HIP;
        e->head_act1 = -1;
        e->head_act2 = tanh_x5;
    } else if (v4_single) {
        const auto hw = to_half(weights[wi++], (size_t)e->K * C);
        if (upload(e, &e->head_w1, hw)) return MIBC_ERR_HIP;
        if (C % 32 == 0 && C <= 512)
            if (upload(e, &e->head_w1f, to_frag16(hw, e->K, C))) return MIBC_ERR_HIP;
        e->head_act1 = tanh_x5;
    } else {
        const auto hw = to_half(weights[wi++], (size_t)e->K * C);
        if (upload(e, &e->head_w1, hw)) return MIBC_ERR_HIP;
        if (C % 32 == 0 && C <= 512)
            if (upload(e, &e->head_w1f, to_frag16(hw, e->K, C))) return MIBC_ERR_HIP;
        const float *B = weights[wi++];
        if (upload(e, &e->head_b1, std::vector<float>(B, B + e->K))) return MIBC_ERR_HIP;
        e->head_act1 = 3;
    }
    e->use_ws = MIBC_ENV_INT("MIBC_WSGEMM", 1);
    e->fuse_q8 = MIBC_ENV_INT("MIBC_FUSE_Q8", 1);
    e->use_cluster = MIBC_ENV_INT("MIBC_LSTM_CLUSTER", 1);
    const char *tp = getenv("MIBC_TAPS");
    e->taps = tp ? atoi(tp) : 0;
    // weight uploads are null-stream copies from pageable memory: hipMemcpy may return once the data sits
    // in the staging buffer, and the engine's stream is non-blocking, so finish them before first use
    HIP_OK(e, hipDeviceSynchronize());
    return MIBC_OK;
    };
    return finish_create(e, body(), out);
}

static void free_ws(mibc_engine *e) {
    if (e->is_tx) tx_free_ws(e);
    void *ptrs[] = {e->in_stage, e->a2p, e->xa, e->xb, e->scores, e->scores2, e->mid, e->a1_tap, e->bwd,
                    e->prob_tap, e->trace, e->path_state, e->out3, e->ss_stage, e->cl_flags, e->cl_cstate};
    e->scores2 = nullptr;
    e->dec_pending[0] = e->dec_pending[1] = false;
    e->timed[0] = e->timed[1] = false;     // the stage-event sets are recreated below: nothing recorded yet
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    e->cl_flags = nullptr;
    e->cl_cstate = nullptr;
    e->in_stage = e->a2p = e->xa = e->xb = e->scores = e->mid = e->a1_tap = nullptr;
    e->bwd = e->prob_tap = nullptr;
    e->trace = nullptr;
    e->path_state = nullptr;
    e->out3 = nullptr;
    e->ss_stage = nullptr;
    for (auto &set : e->sub_ev) {
        for (auto ev : set) (void)hipEventDestroy(ev);
        set.clear();
    }
    e->N_res = 0;
    e->T_in_cap = 0;
    e->T_in_res = 0;
    e->ws_bytes = 0;
}

extern "C" void mibc_destroy(mibc_engine *e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->s_dec) {
        (void)hipStreamSynchronize(e->s_dec);
        (void)hipStreamDestroy(e->s_dec);
        for (int p = 0; p < 2; ++p) {
            if (e->ev_head[p]) (void)hipEventDestroy(e->ev_head[p]);
            if (e->ev_dec[p]) (void)hipEventDestroy(e->ev_dec[p]);
        }
    }
    free_ws(e);
    if (e->is_tx) tx_destroy(e);
    void *ptrs[] = {e->w1, e->b1, e->w2, e->b2, e->b3, e->w3, e->head_w1, e->head_w2, e->head_b1, e->w3f, e->head_w1f,
                    e->stats_scratch, e->var_scratch};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    for (auto p : e->lstm_w) (void)hipFree(p);
    for (auto p : e->lstm_w16)
        if (p) (void)hipFree(p);
    for (auto p : e->lstm_bn) (void)hipFree(p);
    for (auto p : e->lstm_wq)
        if (p) (void)hipFree(p);
    for (auto p : e->lstm_deq)
        if (p) (void)hipFree(p);
    for (auto &a : e->aslot) {
        if (a.in) (void)hipFree(a.in);
        if (a.ss) (void)hipFree(a.ss);
        if (a.out3) (void)hipFree(a.out3);
        if (a.var_dev) (void)hipFree(a.var_dev);
        if (a.var_host) (void)hipHostFree(a.var_host);
        if (a.ev_in) (void)hipEventDestroy(a.ev_in);
        if (a.ev_done) (void)hipEventDestroy(a.ev_done);
        if (a.ev_out) (void)hipEventDestroy(a.ev_out);
    }
    if (e->s_in) (void)hipStreamDestroy(e->s_in);
    if (e->s_out) (void)hipStreamDestroy(e->s_out);
    for (auto p : e->lstm_wcl) (void)hipFree(p);
    for (auto p : e->lstm_bcl) (void)hipFree(p);
    for (auto p : e->lstm_wclq)
        if (p) (void)hipFree(p);
    for (auto p : e->lstm_bclq)
        if (p) (void)hipFree(p);
    for (auto p : e->lstm_dqcl)
        if (p) (void)hipFree(p);
    if (e->lstm_zero) (void)hipFree(e->lstm_zero);
    if (e->cl_err) (void)hipFree(e->cl_err);
    if (e->cl_err_host) (void)hipHostFree(e->cl_err_host);
    for (auto &set : e->ev)
        for (auto &ev : set)
            if (ev) (void)hipEventDestroy(ev);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

extern "C" int mibc_output_steps(const mibc_engine *e, int T_in) {
    if (e->is_tx) return tx_tokens(e, T_in) * e->tx.sf;
    const int W3 = e->d.conv_winlen[2];
    return (T_in + 2 * e->pad3 - W3) / e->stride + 1;
}

extern "C" int mibc_batch_granularity(const mibc_engine *e) {
    if (e->is_tx) return 1;
    // the quantised wide layers exist only as the 256-row cluster kernel (no per-workgroup int8 instance for C >= 512): every
    // batch size derived from the granularity (mibc_reserve, HipCaller::choose_batch_size) is then a whole number of clusters
    if (e->d.lstm_quant && e->C >= 512) return 256;
    return mibc_lstm_rows_per_wg(e->C);
}

static int decode_sub_default(const mibc_engine *e) {
    if (e->is_tx) return 2048;          // 25 MB per chunk (T = 2048, K = 4096)
    return e->K > 1024 ? 4096 : 16384;  // sup@v4.3: 20.5 MB per chunk ; hac: 5.3 MB ; fast: 1.3 MB
}

static int decode_sub(const mibc_engine *e, int N) {
    // Largest sub-batch whose scores + back-guides + trace stay under ~100 GB of the 288 GB: every
    // decode kernel is one wave / workgroup per chunk, so the more chunks per launch the better the
    // latency hiding (hac: 16384 chunks in one launch 70 ms vs 81 ms in four launches of 4096).
    int nd = MIBC_ENV_INT("MIBC_DECODE_SUB", decode_sub_default(e));
    if (nd < 64) nd = 64;
    nd = (nd / 64) * 64;
    return N < nd ? N : nd;
}

// Bytes per chunk (linear in N) and fixed part — the analogue of CudaCaller.cpp:323-369.
extern "C" int mibc_query_memory(const mibc_engine *e, int T_in, size_t *bytes_per_chunk,
                                 size_t *bytes_fixed) {
    if (!e || T_in <= 0) return MIBC_ERR_ARG;
    const size_t T = (size_t)mibc_output_steps(e, T_in);
    if (e->is_tx) {
        // (decode overlap: a second scores buffer per decode sub-batch, mibc_set_decode_overlap — ADVICE r5)
        const size_t per_dec = T * e->K * 2 * (e->decode_overlap ? 2 : 1) + (T + 1) * e->S * 4 + (T + 1) * 32 * 4 + T * 2 + T * 4;
        if (bytes_per_chunk) *bytes_per_chunk = tx_bytes_per_chunk(e, T_in);
        if (bytes_fixed) *bytes_fixed = per_dec * (size_t)decode_sub_default(e);
        return MIBC_OK;
    }
    const size_t Tpitch = (size_t)T_in + 2 * e->pad3 + 2;
    size_t per = 0;
    per += (size_t)T_in * 2;                 // staged input
    per += Tpitch * 16 * 2;                  // conv2 out (padded)
    per += 2 * T * e->C * 2;                 // LSTM ping-pong
    per += 3 * T;                            // out planes
    const size_t per_dec = T * e->K * 2 * (e->decode_overlap ? 2 : 1) + (T + 1) * e->S * 4 + (T + 1) * 32 * 4 + T * 2 + T * 4 +
                           (e->d.out_features > 0 ? T * e->d.out_features * 2 : 0);
    if (bytes_per_chunk) *bytes_per_chunk = per;
    if (bytes_fixed) *bytes_fixed = per_dec * (size_t)decode_sub_default(e);  // decode scratch is per sub-batch
    return MIBC_OK;
}

// Geometry of the call in flight inside the reserved workspace.  A change of chunk length moves the zero padding
// rows of the convolution buffers: they are re-zeroed on the engine's stream (a2p: 5 GB at the hac batch = ~1.5 ms,
// paid only when consecutive batches differ in chunk size).
static int set_geometry(mibc_engine *e, int T_in) {
    if (e->T_in_res == T_in) return MIBC_OK;
    const int T = mibc_output_steps(e, T_in);
    if (T < 1) return fail(e, MIBC_ERR_ARG, "chunk too short");
    if (e->is_tx) {
        const int rc = tx_set_geometry(e, T_in);
        if (rc != MIBC_OK) return rc;
    } else {
        // on the engine's stream: a null-stream memset is not ordered with this non-blocking stream.  First geometry of a
        // workspace: everything.  Later switches (the two chunk-size queues of a device alternate on one engine): only the
        // rows that are padding in the NEW pitch — row n's data is rewritten by conv12 on every call, its pad rows
        // [n Tp, +pad) and [n Tp + pad + T_in, (n + 1) Tp) never are — i.e. N strips of 2 pad + 2 rows (back pad of n +
        // front pad of n + 1 are adjacent) plus the 64 slack rows behind the last chunk: 0.2 % of the buffer.
        const size_t Tp = (size_t)T_in + 2 * e->pad3 + 2, rowb = 16 * sizeof(half_t), pad = (size_t)e->pad3;
        if (e->T_in_res == 0) {
            HIP_OK(e, hipMemsetAsync(e->a2p, 0, e->a2p_bytes, e->stream));
        } else {
            char *base = (char *)e->a2p;
            HIP_OK(e, hipMemsetAsync(base, 0, pad * rowb, e->stream));
            HIP_OK(e, hipMemset2DAsync(base + (pad + (size_t)T_in) * rowb, Tp * rowb, 0, (2 * pad + 2) * rowb, (size_t)e->N_res, e->stream));
            HIP_OK(e, hipMemsetAsync(base + (size_t)e->N_res * Tp * rowb, 0, 64 * rowb, e->stream));
        }
    }
    e->Tpitch = T_in + 2 * e->pad3 + 2;
    e->T_in_res = T_in;
    e->T_res = T;
    return MIBC_OK;
}

extern "C" int mibc_reserve(mibc_engine *e, int N_max, int T_in) {
    if (!e || N_max <= 0 || T_in <= 0) return MIBC_ERR_ARG;
    if (N_max % mibc_batch_granularity(e) != 0)
        return fail(e, MIBC_ERR_ARG, "N_max must be a multiple of mibc_batch_granularity()");
    HIP_OK(e, hipSetDevice(e->device));
    if (e->N_res >= N_max && e->T_in_cap >= T_in) return set_geometry(e, T_in);
    HIP_OK(e, hipStreamSynchronize(e->stream));
    if (e->s_dec) HIP_OK(e, hipStreamSynchronize(e->s_dec));
    free_ws(e);
    const size_t T = (size_t)mibc_output_steps(e, T_in);
    if (T < 1) return fail(e, MIBC_ERR_ARG, "chunk too short");
    const size_t N = (size_t)N_max;
    e->Tpitch = T_in + 2 * e->pad3 + 2;
    e->Nd = decode_sub(e, N_max);
    const size_t Nd = (size_t)e->Nd;
    size_t total = 0;
    auto alloc = [&](void **p, size_t bytes) -> int {
        HIP_OK(e, hipMalloc(p, bytes));
        total += bytes;
        return 0;
    };
    if (alloc((void **)&e->in_stage, N * T_in * 2)) return MIBC_ERR_MEM;
    if (alloc((void **)&e->ss_stage, N * 2 * sizeof(float))) return MIBC_ERR_MEM;
    if (e->is_tx) {
        size_t txb = 0;
        const int rc = tx_reserve(e, N_max, T_in, &txb);
        if (rc != MIBC_OK) return rc;
        total += txb;
    } else {
        e->a2p_bytes = (N * e->Tpitch + 64) * 16 * 2;   // zeroed by set_geometry
        if (alloc((void **)&e->a2p, e->a2p_bytes)) return MIBC_ERR_MEM;
        if (alloc((void **)&e->xa, T * N * e->C * 2)) return MIBC_ERR_MEM;
        if (alloc((void **)&e->xb, T * N * e->C * 2)) return MIBC_ERR_MEM;
        if (!e->lstm_wcl.empty() && N >= 256) {
            if (alloc((void **)&e->cl_flags, (N / 256) * (size_t)(e->C / 128) * 16 * sizeof(unsigned))) return MIBC_ERR_MEM;
            if (alloc((void **)&e->cl_cstate, (N / 256) * 256 * (size_t)e->C * sizeof(float))) return MIBC_ERR_MEM;
        }
    }
    if (alloc((void **)&e->scores, Nd * T * e->K * 2)) return MIBC_ERR_MEM;
    if (e->decode_overlap && alloc((void **)&e->scores2, Nd * T * e->K * 2)) return MIBC_ERR_MEM;
    if (e->d.out_features > 0)
        if (alloc((void **)&e->mid, Nd * T * e->d.out_features * 2)) return MIBC_ERR_MEM;
    if (alloc((void **)&e->bwd, Nd * (T + 1) * e->S * 4)) return MIBC_ERR_MEM;
    if (alloc((void **)&e->trace, Nd * (T + 1) * 32 * 4)) return MIBC_ERR_MEM;
    if (alloc((void **)&e->path_state, Nd * T * 2)) return MIBC_ERR_MEM;
    if (alloc((void **)&e->out3, 3 * N * T)) return MIBC_ERR_MEM;
    if (e->taps) {
        if (alloc((void **)&e->a1_tap, N * T_in * 16 * 2)) return MIBC_ERR_MEM;
        if (alloc((void **)&e->prob_tap, Nd * T * 4)) return MIBC_ERR_MEM;
    }
    const int nsub = (N_max + e->Nd - 1) / e->Nd;
    for (auto &set : e->sub_ev) {
        set.resize((size_t)nsub * 3);
        for (auto &ev : set) HIP_OK(e, hipEventCreate(&ev));
    }
    e->N_res = N_max;
    e->T_in_cap = T_in;
    e->T_in_res = 0;
    e->ws_bytes = total;
    const int rc = set_geometry(e, T_in);
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipStreamSynchronize(e->stream));  // the zero fills
    return MIBC_OK;
}

extern "C" void *mibc_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
extern "C" void mibc_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}
extern "C" void *mibc_device_alloc(mibc_engine *e, size_t bytes) {
    void *p = nullptr;
    if (e) (void)hipSetDevice(e->device);
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
extern "C" void mibc_device_free(mibc_engine *e, void *p) {
    if (e) (void)hipSetDevice(e->device);
    if (p) (void)hipFree(p);
}
extern "C" int mibc_memcpy_h2d(mibc_engine *e, void *dst, const void *src, size_t bytes) {
    HIP_OK(e, hipSetDevice(e->device));
    HIP_OK(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, e->stream));
    HIP_OK(e, hipStreamSynchronize(e->stream));
    return MIBC_OK;
}
static int overlap_join(mibc_engine *e, hipStream_t st);
extern "C" int mibc_memcpy_d2h(mibc_engine *e, void *dst, const void *src, size_t bytes) {
    HIP_OK(e, hipSetDevice(e->device));
    // decode overlap: the source may be output planes the decoder stream is still writing: `mibc_call_device;
    // mibc_memcpy_d2h(out)` stays a valid sequence without a mibc_sync in between
    const int jrc = overlap_join(e, e->stream);
    if (jrc != MIBC_OK) return jrc;
    HIP_OK(e, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_OK(e, hipStreamSynchronize(e->stream));
    return check_cluster_error(e);
}
// decode overlap: everything the decoder stream still has in flight becomes a dependency of `st` (the engine's stream before a
// serial stage touches the scores buffers / the caller's output; a copy stream before it reads the output planes)
static int overlap_join(mibc_engine *e, hipStream_t st) {
    for (int p = 0; p < 2; ++p)
        if (e->dec_pending[p]) HIP_OK(e, hipStreamWaitEvent(st, e->ev_dec[p], 0));
    return MIBC_OK;
}

extern "C" int mibc_sync(mibc_engine *e) {
    HIP_OK(e, hipSetDevice(e->device));
    HIP_OK(e, hipStreamSynchronize(e->stream));
    if (e->s_dec) {
        HIP_OK(e, hipStreamSynchronize(e->s_dec));
        e->dec_pending[0] = e->dec_pending[1] = false;
    }
    return check_cluster_error(e);
}

extern "C" int mibc_set_decode_overlap(mibc_engine *e, int on) {
    if (!e) return MIBC_ERR_ARG;
    HIP_OK(e, hipSetDevice(e->device));
    int rc = mibc_sync(e);
    if (rc != MIBC_OK) return rc;
    if (on && !e->s_dec) {
        HIP_OK(e, hipStreamCreateWithFlags(&e->s_dec, hipStreamNonBlocking));
        for (int p = 0; p < 2; ++p) {
            HIP_OK(e, hipEventCreateWithFlags(&e->ev_head[p], hipEventDisableTiming));
            HIP_OK(e, hipEventCreateWithFlags(&e->ev_dec[p], hipEventDisableTiming));
        }
    }
    if (on && !e->scores2 && e->scores) {      // a workspace is already reserved: add the second scores buffer of the same size
        const size_t T = (size_t)mibc_output_steps(e, e->T_in_cap);
        const size_t Nd = (size_t)((e->N_res < e->Nd) ? e->N_res : e->Nd);
        if (hipMalloc((void **)&e->scores2, Nd * T * e->K * 2) != hipSuccess) return fail(e, MIBC_ERR_MEM, "decode overlap: second scores buffer");
    }
    e->decode_overlap = on ? 1 : 0;
    return MIBC_OK;
}

extern "C" int mibc_set_profile(mibc_engine *e, int level) {
    e->profile = level;
    return MIBC_OK;
}

// conv1+conv2 -> conv3 (implicit GEMM) -> LSTM stack.  Leaves e->lstm_out.
static int run_encoder(mibc_engine *e, const half_t *in_dev, int N, int T_in) {
    const mibc_model_desc &d = e->d;
    const int T = mibc_output_steps(e, T_in);
    const bool prof = e->profile > 0;
    MibcRange r_enc(e, "mibc:encoder");
    if (prof) e->ps ^= 1;
    if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_START], e->stream));
    if (mibc_launch_conv12(e->stream, in_dev, e->w1, e->b1, e->w2, e->b2, e->a2p, e->a1_tap, e->in_ss, e->in_smask, N, T_in,
                           e->Tpitch, e->pad3, d.conv_act[0], d.conv_act[1]) != 0)
        return fail(e, MIBC_NOT_SUPPORTED, "conv activation combination not supported");
    bool conv3_done = false;
    // lstm_quant: every layer int8 when conv3 hands over tanh outputs (the reference's CUTLASS_TNC_I8 layout, nn/ConvStack.cpp:72),
    // else the first layer in f16 (LSTMStack.cpp:199-207)
    // (the reference selects that layout for 128 < lstm_size <= 1024 only, nn/ConvStack.cpp:69-73; at lstm_size <= 128 its
    // quantised path is the NTC forward_quantized scheme, a different one: first layer f16 here)
    const bool q_all = d.lstm_quant && d.n_convs >= 3 && d.conv_act[d.n_convs - 1] == MIBC_ACT_TANH && e->C > 128;
    bool conv3_int8 = false;   // round 6: conv3's epilogue wrote the int8 rows itself (no separate conversion pass)
    if (e->use_ws && e->w3f) {
        WsArgs w{};
        w.A = e->a2p;
        w.Wf = e->w3f;
        w.bias = e->b3;
        w.out = e->xa;
        w.cols = e->C;
        w.act = d.conv_act[2];
        w.N = N;
        w.T = T;
        w.Tpitch = e->Tpitch;
        w.stride = e->stride;
        if (q_all && e->fuse_q8) {
            // the reference converts conv3's f16 output in a separate pass (nn/ConvStack.cpp:243,324-329 host_convert); here the
            // GEMM epilogue emits round(127 f16(tanh)) rows — bit-identical to that pass, minus its 3 bytes per element of HBM traffic
            w.act = 4;
            conv3_int8 = conv3_done = (mibc_launch_wsgemm(e->stream, &w, e->K3pad, 1) == 0);
            w.act = d.conv_act[2];
        }
        if (!conv3_done) conv3_done = (mibc_launch_wsgemm(e->stream, &w, e->K3pad, 1) == 0);
    }
    GemmArgs g{};
    g.A = e->a2p;
    g.B = e->w3;
    g.bias = e->b3;
    g.out = e->xa;
    g.M = N * T;
    g.Ncols = (e->C + 127) / 128 * 128;
    g.ncols_valid = (g.Ncols != e->C) ? e->C : 0;
    g.K = e->K3pad;
    g.a_div = T;
    g.a_outer = (long)e->Tpitch * 16;
    g.a_inner = (long)e->stride * 16;
    g.o_div = T;
    g.o_outer = e->C;             // n
    g.o_inner = (long)N * e->C;   // t
    g.act = d.conv_act[2];
    if (!conv3_done && mibc_launch_gemm_tn(e->stream, &g) != 0)
        return fail(e, MIBC_NOT_SUPPORTED, "conv3 gemm shape");
    if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_CONV], e->stream));
    half_t *cur = e->xa, *nxt = e->xb;
    // conv3's f16 output is converted once (round(127 v), v in (-1, 1)) unless its epilogue already wrote int8
    if (q_all && !conv3_int8) {
        if (mibc_launch_q8_convert(e->stream, cur, (int8_t *)nxt, (size_t)T * N * e->C) != 0) return fail(e, MIBC_NOT_SUPPORTED, "lstm shape");
        half_t *t = cur;
        cur = nxt;
        nxt = t;
    }
#ifdef MIBC_DEBUG_KERNELS
    if (MIBC_ENV_INT("MIBC_STOP_AFTER_CONV3", 0)) {   // debug library only: tap 3 then returns what the LSTM stack would read
        e->lstm_out = cur;
        return MIBC_OK;
    }
#endif
    for (int l = 0; l < d.lstm_layers; ++l) {
        // LSTMStack(layers, size, reverse_first = true): nn/LSTMStack.cpp:29-41, CRFModel.cpp:41
        const int reverse = (l % 2 == 0) ? 1 : 0;
        MibcRange r_layer(e, "lstm_layer");   // the reference's range name (nn/LSTMStack.cpp:100,148)
        // wide layers: the hidden-split cluster kernel whenever the batch is a whole number of 256-row clusters
        // (same arithmetic, element for element, as the per-workgroup kernel it replaces)
        const bool wide_q = d.lstm_quant && e->C >= 512;
        const bool cl_ok = !(wide_q && (l >= 1 || q_all)) && (!d.lstm_quant || wide_q) &&
                           e->use_cluster && !e->lstm_wcl.empty() && e->cl_flags != nullptr && N % 256 == 0 &&
                           mibc_launch_lstm_layer_cl(e->stream, e->C, cur, nxt, e->lstm_wcl[l], e->lstm_bcl[l],
                                                     e->lstm_zero, e->cl_cstate, e->cl_flags, e->cl_err, T, N, reverse,
                                                     e->in_tmask) == 0;
        if (wide_q) {
            // quantised path of the wide (cluster) layers: layer 0 in f16 (cluster kernel) + conversion of its output to int8
            // (nn/LSTMStack.cpp:199-207), layers 1 .. L-1 on the int8 instance of the cluster kernel; the last one writes f16
            // for the head and exchanges an int8 copy of h kept behind its int8 input (the input occupies only the first
            // T N C bytes of its f16-sized buffer)
            // (variable chunks: the masked instances of the same kernels — the reference's default GPU mode is both at once)
            if (N % 256 != 0 || e->cl_flags == nullptr)
                return fail(e, MIBC_NOT_SUPPORTED, "lstm_quant with lstm_size >= 512 needs batches that are multiples of 256");
            if (l == 0 && !q_all) {
                if (!cl_ok) return fail(e, MIBC_NOT_SUPPORTED, "lstm shape");
                e->cl_used = true;
                if (mibc_launch_q8_convert(e->stream, nxt, (int8_t *)cur, (size_t)T * N * e->C) != 0)
                    return fail(e, MIBC_NOT_SUPPORTED, "lstm shape");
                if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_LSTM0 + l], e->stream));
                continue;      // layer 1 reads `cur` (int8): the ping-pong is not swapped after layer 0
            }
            const bool last = (l + 1 == d.lstm_layers);
            signed char *hx = last ? (signed char *)cur + (size_t)T * N * e->C : nullptr;
            if (mibc_launch_lstm_layer_cl(e->stream, e->C, cur, nxt, (const half_t *)e->lstm_wclq[l], e->lstm_bclq[l], e->lstm_zero,
                                          e->cl_cstate, e->cl_flags, e->cl_err, T, N, reverse, e->in_tmask, last ? 2 : 1,
                                          e->lstm_dqcl[l], hx) != 0)
                return fail(e, MIBC_NOT_SUPPORTED, "lstm_quant: cluster launch");
            e->cl_used = true;
        } else if (d.lstm_quant) {
            // the reference's quantised path: first layer f16 + conversion of its output (LSTMStack.cpp:199-207), then int8
            int qrc;
            if (l == 0 && !q_all) {
                qrc = e->in_tmask != nullptr
                              ? mibc_launch_lstm_layer_masked(e->stream, e->C, cur, nxt, e->lstm_w16[l], e->lstm_bn[l], T, N, reverse, e->in_tmask)
                              : mibc_launch_lstm_layer(e->stream, e->C, cur, nxt, e->lstm_w[l], e->lstm_w16[l], e->lstm_bn[l], T, N, reverse);
                // f16 output of layer 0 (nxt) -> int8 into the buffer layer 0 read from (cur): layer 1 then reads `cur`,
                // so the ping-pong is NOT swapped after this layer
                if (qrc == 0) qrc = mibc_launch_q8_convert(e->stream, nxt, (int8_t *)cur, (size_t)T * N * e->C);
                if (qrc != 0) return fail(e, MIBC_NOT_SUPPORTED, "lstm shape");
                if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_LSTM0 + l], e->stream));
                continue;
            }
            qrc = mibc_launch_lstm_layer_q8(e->stream, e->C, (const int8_t *)cur, nxt, e->lstm_wq[l], e->lstm_bn[l], e->lstm_deq[l],
                                            T, N, reverse, l + 1 == d.lstm_layers ? 1 : 0, e->in_tmask);
            if (qrc != 0) return fail(e, MIBC_NOT_SUPPORTED, "lstm_quant: batch must be a multiple of 64");
        } else if (cl_ok) {
            e->cl_used = true;
        } else if (e->in_tmask != nullptr) {
            if (mibc_launch_lstm_layer_masked(e->stream, e->C, cur, nxt, e->C >= 512 ? e->lstm_w[l] : e->lstm_w16[l],
                                              e->lstm_bn[l], T, N, reverse, e->in_tmask) != 0)
                return fail(e, MIBC_NOT_SUPPORTED, "variable chunks need an LSTM model and N % 64 == 0");
        } else if (mibc_launch_lstm_layer(e->stream, e->C, cur, nxt, e->lstm_w[l], e->lstm_w16[l], e->lstm_bn[l], T, N,
                                          reverse) != 0)
            return fail(e, MIBC_NOT_SUPPORTED, "lstm shape");
        if (prof && l < 8) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_LSTM0 + l], e->stream));
        half_t *t = cur;
        cur = nxt;
        nxt = t;
    }
    e->lstm_out = cur;
    if (e->cl_used) {
        // hand-off time-outs of the cluster kernel: the sticky device word is copied to THIS call's host slot and
        // cleared, both in stream order, so a time-out is reported by the call it happened in and by no other
        HIP_OK(e, hipMemcpyAsync(e->cl_err_host + 4 * e->err_slot, e->cl_err, 16, hipMemcpyDeviceToHost, e->stream));
        HIP_OK(e, hipMemsetAsync(e->cl_err, 0, 16, e->stream));
    }
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

// after the call that reports into `slot` has completed: did a cluster hand-off of its LSTM kernels time out?
static int check_cluster_error(mibc_engine *e, int slot) {
    if (!e->cl_err_host) return MIBC_OK;
    unsigned *h = e->cl_err_host + 4 * slot;
#ifdef MIBC_DEBUG_KERNELS
    if (h[1] | h[2] | h[3]) {
        fprintf(stderr, "[mibc dbg] cluster hand-off: slow-path entries %u, worst lag %u, real waits %u\n", h[1], h[2], h[3]);
        h[1] = h[2] = h[3] = 0;
    }
#endif
    if (h[0] != 0) {
        const unsigned w = h[0];
        h[0] = 0;
        return fail(e, MIBC_ERR_HIP, "LSTM cluster kernel: hand-off between workgroups timed out (cluster " +
                                             std::to_string((w >> 16) & 0x7fff) + ", step " + std::to_string(w & 0xffff) +
                                             "); results of this call are invalid");
    }
    return MIBC_OK;
}

// head for chunk rows [n0, n0+ns) of the LSTM output -> scores_out [ns][T][K]
static int run_head(mibc_engine *e, int N, int T, int n0, int ns, half_t *scores_out) {
    const mibc_model_desc &d = e->d;
    MibcRange r_head(e, "linear");
    GemmArgs g{};
    g.A = e->lstm_out + (size_t)n0 * e->C;
    g.M = ns * T;
    g.K = e->C;
    g.a_div = ns;
    g.a_outer = (long)N * e->C;  // t
    g.a_inner = e->C;            // n'
    g.o_div = ns;
    if (d.out_features > 0) {
        const int D = d.out_features;
        g.B = e->head_w1;
        g.bias = e->head_b1;
        g.out = e->mid;
        g.Ncols = D;
        g.o_outer = (long)ns * D;  // t   (mid laid out [T][ns][D])
        g.o_inner = D;             // n'
        g.act = e->head_act1;
        if (mibc_launch_gemm_tn(e->stream, &g) != 0) return fail(e, MIBC_NOT_SUPPORTED, "head gemm 1");
        GemmArgs h{};
        h.A = e->mid;
        h.B = e->head_w2;
        h.bias = nullptr;
        h.out = scores_out;
        h.M = ns * T;
        h.Ncols = e->K;
        h.K = D;
        h.a_div = ns;
        h.a_outer = (long)ns * D;
        h.a_inner = D;
        h.o_div = ns;
        h.o_outer = e->K;             // t
        h.o_inner = (long)T * e->K;   // n'
        h.act = e->head_act2;
        if (mibc_launch_gemm_tn(e->stream, &h) != 0) return fail(e, MIBC_NOT_SUPPORTED, "head gemm 2");
    } else {
        if (e->use_ws && e->head_w1f) {
            WsArgs w{};
            w.A = e->lstm_out;
            w.Wf = e->head_w1f;
            w.bias = e->head_b1;
            w.out = scores_out;
            w.cols = e->K;
            w.act = e->head_act1;
            w.N = N;
            w.Ns = ns;
            w.n0 = n0;
            w.T = T;
            if (mibc_launch_wsgemm(e->stream, &w, e->C, 0) == 0) return MIBC_OK;
        }
        g.B = e->head_w1;
        g.bias = e->head_b1;
        g.out = scores_out;
        g.Ncols = e->K;
        g.o_outer = e->K;             // t
        g.o_inner = (long)T * e->K;   // n'
        g.act = e->head_act1;
        if (mibc_launch_gemm_tn(e->stream, &g) != 0) return fail(e, MIBC_NOT_SUPPORTED, "head gemm");
    }
    return MIBC_OK;
}

static int check_call(mibc_engine *e, int N, int T_in) {
    if (!e) return MIBC_ERR_ARG;
    if (N <= 0 || N % mibc_batch_granularity(e) != 0)
        return fail(e, MIBC_ERR_ARG, "N must be a positive multiple of mibc_batch_granularity()");
    HIP_OK(e, hipSetDevice(e->device));
    if (e->N_res < N || e->T_in_cap < T_in || e->T_in_res != T_in) {
        const int rc = mibc_reserve(e, N, T_in);   // grows the workspace or only switches the geometry
        if (rc != MIBC_OK) return rc;
    }
    return MIBC_OK;
}

extern "C" int mibc_forward(mibc_engine *e, const uint16_t *in_dev, int N, int T_in,
                            uint16_t *scores_dev) {
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    const int T = mibc_output_steps(e, T_in);
    rc = e->is_tx ? tx_run_network(e, (const half_t *)in_dev, N, T_in)
                  : run_encoder(e, (const half_t *)in_dev, N, T_in);
    if (rc != MIBC_OK) return rc;
    // the two-stage head uses the sub-batch sized `mid` buffer
    for (int n0 = 0; n0 < N; n0 += e->Nd) {
        const int ns = (N - n0 < e->Nd) ? (N - n0) : e->Nd;
        half_t *so = (half_t *)scores_dev + (size_t)n0 * T * e->K;
        rc = e->is_tx ? tx_run_head(e, N, n0, ns, so) : run_head(e, N, T, n0, ns, so);
        if (rc != MIBC_OK) return rc;
    }
    if (e->profile > 0) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_END], e->stream));
    e->last_N = N;
    e->last_T = T;
    e->last_T_in = T_in;
    e->timed[e->ps] = false;
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

static float clamp_value(const mibc_engine *e) { return e->d.clamp ? 5.0f : 0.0f; }

extern "C" int mibc_decode(mibc_engine *e, const uint16_t *scores_dev, int N, int T,
                           const mibc_decode_opts *o, int8_t *out_dev) {
    if (!e || !o || N <= 0) return MIBC_ERR_ARG;
    if (e->N_res <= 0 || T != e->T_res) return fail(e, MIBC_ERR_ARG, "mibc_reserve first (T mismatch)");
    if (N > e->N_res) return fail(e, MIBC_ERR_ARG, "mibc_decode: N exceeds the reserved batch");
    HIP_OK(e, hipSetDevice(e->device));
    {
        const int jrc = overlap_join(e, e->stream);   // the decoder's scratch (guides, trace) is shared with the decoder stream
        if (jrc != MIBC_OK) return jrc;
    }
    for (int n0 = 0; n0 < N; n0 += e->Nd) {
        const int ns = (N - n0 < e->Nd) ? (N - n0) : e->Nd;
        const int rc = mibc_launch_decode(e->stream, (const half_t *)scores_dev + (size_t)n0 * T * e->K,
                                          ns, T, e->S, o->beam_width, o->beam_cut, o->blank_score,
                                          clamp_value(e), o->q_shift, o->q_scale, e->bwd, e->trace,
                                          e->path_state, out_dev + (size_t)n0 * T, (size_t)N * T,
                                          e->prob_tap);
        if (rc != 0) return fail(e, MIBC_NOT_SUPPORTED, "decoder: beam_width must be 1..32");
    }
    e->last_N = N;
    e->last_T = T;
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

// head of sub-batch [n0, n0 + ns) on the engine's stream into the scores buffer of the current parity, its decoder on the
// decoder stream.  The head waits for the decoder that last read this buffer (two sub-batches ago).
static int head_and_decode_overlapped(mibc_engine *e, int N, int T, int n0, int ns, const mibc_decode_opts *o, int8_t *out_dev) {
    const int p = (int)(e->dec_parity++ & 1u);
    half_t *sc = p ? e->scores2 : e->scores;
    if (e->dec_pending[p]) HIP_OK(e, hipStreamWaitEvent(e->stream, e->ev_dec[p], 0));
    const int rc = e->is_tx ? tx_run_head(e, N, n0, ns, sc) : run_head(e, N, T, n0, ns, sc);
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipEventRecord(e->ev_head[p], e->stream));
    HIP_OK(e, hipStreamWaitEvent(e->s_dec, e->ev_head[p], 0));
    if (mibc_launch_decode(e->s_dec, sc, ns, T, e->S, o->beam_width, o->beam_cut, o->blank_score, clamp_value(e), o->q_shift,
                           o->q_scale, e->bwd, e->trace, e->path_state, out_dev + (size_t)n0 * T, (size_t)N * T, e->prob_tap) != 0)
        return fail(e, MIBC_NOT_SUPPORTED, "decoder: beam_width must be 1..32");
    HIP_OK(e, hipEventRecord(e->ev_dec[p], e->s_dec));
    e->dec_pending[p] = true;
    return MIBC_OK;
}

extern "C" int mibc_call_device(mibc_engine *e, const uint16_t *in_dev, int N, int T_in,
                                const mibc_decode_opts *o, int8_t *out_dev) {
    if (!o) return MIBC_ERR_ARG;
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    const int T = mibc_output_steps(e, T_in);
    const bool ovl = e->decode_overlap && e->scores2 && !e->taps;
    const bool prof = e->profile > 0 && !ovl;
    rc = e->is_tx ? tx_run_network(e, (const half_t *)in_dev, N, T_in)
                  : run_encoder(e, (const half_t *)in_dev, N, T_in);
    if (rc != MIBC_OK) return rc;
    if (ovl) {
        // (the caller's out_dev is written on the decoder stream: mibc_sync, or overlap_join on the stream that reads it)
        for (int n0 = 0; n0 < N; n0 += e->Nd) {
            const int ns = (N - n0 < e->Nd) ? (N - n0) : e->Nd;
            rc = head_and_decode_overlapped(e, N, T, n0, ns, o, out_dev);
            if (rc != MIBC_OK) return rc;
        }
        e->last_N = N;
        e->last_T = T;
        e->last_T_in = T_in;
        e->timed[e->ps] = false;
        HIP_OK(e, hipGetLastError());
        return MIBC_OK;
    }
    rc = overlap_join(e, e->stream);   // overlap switched off with decoders in flight
    if (rc != MIBC_OK) return rc;
    int si = 0;
    for (int n0 = 0; n0 < N; n0 += e->Nd, ++si) {
        const int ns = (N - n0 < e->Nd) ? (N - n0) : e->Nd;
        if (prof) HIP_OK(e, hipEventRecord(e->sub_ev[e->ps][si * 3 + 0], e->stream));
        rc = e->is_tx ? tx_run_head(e, N, n0, ns, e->scores) : run_head(e, N, T, n0, ns, e->scores);
        if (rc != MIBC_OK) return rc;
        if (prof) HIP_OK(e, hipEventRecord(e->sub_ev[e->ps][si * 3 + 1], e->stream));
        MibcRange r_dec(e, "beam_search");
        if (mibc_launch_decode(e->stream, e->scores, ns, T, e->S, o->beam_width, o->beam_cut,
                               o->blank_score, clamp_value(e), o->q_shift, o->q_scale, e->bwd, e->trace,
                               e->path_state, out_dev + (size_t)n0 * T, (size_t)N * T,
                               e->prob_tap) != 0)
            return fail(e, MIBC_NOT_SUPPORTED, "decoder: beam_width must be 1..32");
        if (prof) HIP_OK(e, hipEventRecord(e->sub_ev[e->ps][si * 3 + 2], e->stream));
    }
    if (prof) HIP_OK(e, hipEventRecord(e->ev[e->ps][mibc_engine::EV_END], e->stream));
    e->last_N = N;
    e->last_T = T;
    e->last_T_in = T_in;
    e->timed[e->ps] = prof;
    e->prof_N[e->ps] = N;
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

extern "C" int mibc_call(mibc_engine *e, const uint16_t *in_host, int N, int T_in,
                         const mibc_decode_opts *o, int8_t *out_host) {
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    const int T = mibc_output_steps(e, T_in);
    HIP_OK(e, hipMemcpyAsync(e->in_stage, in_host, (size_t)N * T_in * 2, hipMemcpyHostToDevice,
                             e->stream));
    rc = mibc_call_device(e, (const uint16_t *)e->in_stage, N, T_in, o, e->out3);
    if (rc != MIBC_OK) return rc;
    rc = overlap_join(e, e->stream);
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipMemcpyAsync(out_host, e->out3, (size_t)3 * N * T, hipMemcpyDeviceToHost, e->stream));
    HIP_OK(e, hipStreamSynchronize(e->stream));
    return check_cluster_error(e);
}

// ---------------------------------------------------------------------------------------------
// Two-phase calls: copies on their own streams, ordered with the engine's stream by events.
// ---------------------------------------------------------------------------------------------
static int async_prepare(mibc_engine *e, int slot, int N, int T_in) {
    if (!e->s_in) {
        HIP_OK(e, hipStreamCreateWithFlags(&e->s_in, hipStreamNonBlocking));
        HIP_OK(e, hipStreamCreateWithFlags(&e->s_out, hipStreamNonBlocking));
    }
    auto &a = e->aslot[slot];
    if (!a.ev_in) {
        HIP_OK(e, hipEventCreateWithFlags(&a.ev_in, hipEventDisableTiming));
        HIP_OK(e, hipEventCreateWithFlags(&a.ev_done, hipEventDisableTiming));
        HIP_OK(e, hipEventCreateWithFlags(&a.ev_out, hipEventDisableTiming));
    }
    const size_t T = (size_t)mibc_output_steps(e, T_in);
    const size_t inb = (size_t)N * T_in * 2, outb = (size_t)3 * N * T;
    if (a.in_bytes < inb || a.out_bytes < outb) {
        HIP_OK(e, hipStreamSynchronize(e->stream));
        if (a.in) (void)hipFree(a.in);
        if (a.ss) (void)hipFree(a.ss);
        if (a.out3) (void)hipFree(a.out3);
        a.in = nullptr; a.ss = nullptr; a.out3 = nullptr; a.in_bytes = a.out_bytes = 0;
        HIP_OK(e, hipMalloc((void **)&a.in, inb));
        HIP_OK(e, hipMalloc((void **)&a.ss, (size_t)N * 2 * sizeof(float)));
        HIP_OK(e, hipMalloc((void **)&a.out3, outb));
        a.in_bytes = inb;
        a.out_bytes = outb;
    }
    return MIBC_OK;
}

extern "C" int mibc_call_async(mibc_engine *e, int slot, const void *in_host, const float *shift_scale_host, int N,
                               int T_in, const mibc_decode_opts *o, int8_t *out_host) {
    if (!e || !in_host || !out_host || !o || slot < 0 || slot > 1) return MIBC_ERR_ARG;
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    rc = async_prepare(e, slot, N, T_in);
    if (rc != MIBC_OK) return rc;
    auto &a = e->aslot[slot];
    const int T = mibc_output_steps(e, T_in);
    HIP_OK(e, hipMemcpyAsync(a.in, in_host, (size_t)N * T_in * 2, hipMemcpyHostToDevice, e->s_in));
    if (shift_scale_host)
        HIP_OK(e, hipMemcpyAsync(a.ss, shift_scale_host, (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice, e->s_in));
    HIP_OK(e, hipEventRecord(a.ev_in, e->s_in));
    HIP_OK(e, hipStreamWaitEvent(e->stream, a.ev_in, 0));
    e->err_slot = slot;
    rc = shift_scale_host ? mibc_call_device_i16(e, (const int16_t *)a.in, a.ss, N, T_in, o, a.out3)
                          : mibc_call_device(e, (const uint16_t *)a.in, N, T_in, o, a.out3);
    e->err_slot = 2;
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipEventRecord(a.ev_done, e->stream));
    HIP_OK(e, hipStreamWaitEvent(e->s_out, a.ev_done, 0));
    rc = overlap_join(e, e->s_out);          // decode overlap: the output planes are complete when the decoder stream says so
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipMemcpyAsync(out_host, a.out3, (size_t)3 * N * T, hipMemcpyDeviceToHost, e->s_out));
    HIP_OK(e, hipEventRecord(a.ev_out, e->s_out));
    a.n = N;
    return MIBC_OK;
}

extern "C" int mibc_call_poll(mibc_engine *e, int slot) {
    if (!e || slot < 0 || slot > 1 || !e->aslot[slot].ev_out) return 1;
    return hipEventQuery(e->aslot[slot].ev_out) == hipErrorNotReady ? 0 : 1;
}

extern "C" int mibc_call_wait(mibc_engine *e, int slot) {
    if (!e || slot < 0 || slot > 1) return MIBC_ERR_ARG;
    if (!e->aslot[slot].ev_out) return fail(e, MIBC_ERR_ARG, "mibc_call_wait: nothing was submitted on this slot");
    HIP_OK(e, hipSetDevice(e->device));
    HIP_OK(e, hipEventSynchronize(e->aslot[slot].ev_out));
    return check_cluster_error(e, slot);
}

// ---------------------------------------------------------------------------------------------
// f1: ScalerNode on the device.  The *_i16 entry points take raw int16 chunks
// plus one (shift, scale) pair per chunk and apply f16((x - shift) / scale)
// (torch_utils/tensor_utils.cpp:89-142) inside the first convolution's input read.
// ---------------------------------------------------------------------------------------------
extern "C" int mibc_forward_i16(mibc_engine *e, const int16_t *in_dev, const float *shift_scale_dev, int N,
                                int T_in, uint16_t *scores_dev) {
    if (!e || !shift_scale_dev) return MIBC_ERR_ARG;
    e->in_ss = shift_scale_dev;
    const int rc = mibc_forward(e, (const uint16_t *)in_dev, N, T_in, scores_dev);
    e->in_ss = nullptr;
    return rc;
}

extern "C" int mibc_call_device_i16(mibc_engine *e, const int16_t *in_dev, const float *shift_scale_dev, int N,
                                    int T_in, const mibc_decode_opts *o, int8_t *out_dev) {
    if (!e || !shift_scale_dev) return MIBC_ERR_ARG;
    e->in_ss = shift_scale_dev;
    const int rc = mibc_call_device(e, (const uint16_t *)in_dev, N, T_in, o, out_dev);
    e->in_ss = nullptr;
    return rc;
}

extern "C" int mibc_call_i16(mibc_engine *e, const int16_t *in_host, const float *shift_scale_host, int N,
                             int T_in, const mibc_decode_opts *o, int8_t *out_host) {
    if (!shift_scale_host) return MIBC_ERR_ARG;
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    const int T = mibc_output_steps(e, T_in);
    HIP_OK(e, hipMemcpyAsync(e->in_stage, in_host, (size_t)N * T_in * 2, hipMemcpyHostToDevice, e->stream));
    HIP_OK(e, hipMemcpyAsync(e->ss_stage, shift_scale_host, (size_t)N * 2 * sizeof(float),
                             hipMemcpyHostToDevice, e->stream));
    rc = mibc_call_device_i16(e, (const int16_t *)e->in_stage, e->ss_stage, N, T_in, o, e->out3);
    if (rc != MIBC_OK) return rc;
    rc = overlap_join(e, e->stream);          // decode overlap: the planes are complete when the decoder stream says so
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipMemcpyAsync(out_host, e->out3, (size_t)3 * N * T, hipMemcpyDeviceToHost, e->stream));
    HIP_OK(e, hipStreamSynchronize(e->stream));
    return check_cluster_error(e);
}

// Per-read shift / scale of the QUANTILE (strategy 0; params = quantile_a, quantile_b, shift_multiplier,
// scale_multiplier) and MED_MAD (strategy 1) strategies, ScalerNode.cpp:32-52.  Reads are concatenated in
// sig_dev, read r = [offsets_dev[r], offsets_dev[r+1]).  raw_dev (optional) receives (q_a, q_b) or
// (median, median |x - median|).
extern "C" int mibc_scaler_stats(mibc_engine *e, const int16_t *sig_dev, const int64_t *offsets_dev, int n_reads,
                                 int strategy, const float *params4, float *shift_scale_dev, float *raw_dev) {
    if (!e || !sig_dev || !offsets_dev || !shift_scale_dev || n_reads < 0) return MIBC_ERR_ARG;
    if (strategy != 0 && strategy != 1) return fail(e, MIBC_ERR_ARG, "strategy: 0 quantile, 1 med_mad");
    if (strategy == 0 && !params4) return MIBC_ERR_ARG;
    HIP_OK(e, hipSetDevice(e->device));
    constexpr int GROUP = 256;  // bounds the wide-range scratch at 128 MiB
    if (!e->stats_scratch) HIP_OK(e, hipMalloc((void **)&e->stats_scratch, (size_t)GROUP * 2 * 65536 * 4));
    const float qa = params4 ? params4[0] : 0.f, qb = params4 ? params4[1] : 0.f;
    const float shm = params4 ? params4[2] : 0.f, scm = params4 ? params4[3] : 0.f;
    for (int r0 = 0; r0 < n_reads; r0 += GROUP) {
        const int nr = (n_reads - r0 < GROUP) ? (n_reads - r0) : GROUP;
        mibc_launch_read_stats(e->stream, sig_dev, (const long long *)offsets_dev + r0, nr, strategy, qa, qb, shm,
                               scm, shift_scale_dev + 2 * (size_t)r0, raw_dev ? raw_dev + 2 * (size_t)r0 : nullptr,
                               e->stats_scratch);
    }
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

// out[i] = f16((float(x[i]) - shift_r) / scale_r) for every sample of every read
// (utils::shift_scale_tensor_i16_to_f16_inplace, tensor_utils.cpp:89-142).
extern "C" int mibc_scale_reads(mibc_engine *e, const int16_t *sig_dev, const int64_t *offsets_dev, int n_reads,
                                const float *shift_scale_dev, uint16_t *out_f16_dev) {
    if (!e || !sig_dev || !offsets_dev || !shift_scale_dev || !out_f16_dev || n_reads < 0) return MIBC_ERR_ARG;
    if (n_reads == 0) return MIBC_OK;
    HIP_OK(e, hipSetDevice(e->device));
    for (int r0 = 0; r0 < n_reads; r0 += 32768) {  // grid.y limit
        const int nr = (n_reads - r0 < 32768) ? (n_reads - r0) : 32768;
        mibc_launch_scale_reads(e->stream, sig_dev, (const long long *)offsets_dev + r0, nr,
                                shift_scale_dev + 2 * (size_t)r0, (half_t *)out_f16_dev, n_reads >= 256 ? 8 : 64);
    }
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

// f2: svb16 + zig-zag + delta stage of the POD5 VBZ signal codec on the device; replaces that
// half of pod5_get_read_complete_signal (data_loader/DataLoader.cpp:163-170).  Row r of the signal table (after
// zstd) = streams_dev[stream_off[r] .. stream_off[r+1]), decodes to out_dev[sample_off[r] .. sample_off[r+1]).
// status_dev[r] = 0 ok, 1 = the stream was not consumed exactly (corrupt row).
extern "C" int mibc_svb16_decode(mibc_engine *e, const uint8_t *streams_dev, const int64_t *stream_off_dev,
                                 const int64_t *sample_off_dev, int n_rows, int16_t *out_dev, int *status_dev) {
    if (!e || !streams_dev || !stream_off_dev || !sample_off_dev || !out_dev || !status_dev || n_rows < 0)
        return MIBC_ERR_ARG;
    if (n_rows == 0) return MIBC_OK;
    HIP_OK(e, hipSetDevice(e->device));
    mibc_launch_svb16_decode(e->stream, streams_dev, (const long long *)stream_off_dev,
                             (const long long *)sample_off_dev, n_rows, out_dev, status_dev);
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

// ---------------------------------------------------------------------------------------------
// f3: variable chunk sizes.  The reference's CUDA path packs ragged chunks back to back
// (basecall/CudaModelRunner.cpp:21-49, nn/AuxiliaryData.cpp) so that short reads do not pay for padding.
// Here the batch keeps its [N][T_in] rows, but a row may hold SEVERAL chunks of any stride-multiple length,
// separated by >= 2 steps: convolutions see zero padding at every chunk edge (sample bitmap), the LSTM state
// is forced to zero in the gaps (= fresh initial state for the neighbour in either direction), and the
// decoder runs once per chunk on its own step interval.  Each chunk's result equals running the network on
// that chunk alone.  LSTM models with lstm_size 128 / 256 / 384 (the x8 kernels).
// ---------------------------------------------------------------------------------------------
// Builds the masks and the decoder's chunk table of a variable-chunk call.  The decoder runs per decode sub-batch of
// e->Nd rows like the fixed path: the table is ordered by sub-batch (stable), sub_begin[k] .. sub_begin[k+1] are the
// chunks of sub-batch k, score offsets are relative to the sub-batch's first row and back-guide rows restart at 0.
struct VarPlan {
    size_t b0 = 0, b1 = 0, b2 = 0, off1 = 0, off2 = 0, need = 0;   // smask | tmask | idx blob layout (16-byte aligned parts)
    int Tmax = 0;
    std::vector<int> sub_begin;
};
static int var_layout(mibc_engine *e, int N, int T_in, int n_chunks, VarPlan *vp) {
    if (n_chunks <= 0) return fail(e, MIBC_ERR_ARG, "no chunks");
    if (e->is_tx) return fail(e, MIBC_NOT_SUPPORTED, "variable chunks: LSTM models only");
    const int stride = e->stride, T = mibc_output_steps(e, T_in);
    if (T_in % stride != 0 || T != T_in / stride) return fail(e, MIBC_ERR_ARG, "variable chunks: T_in must be a stride multiple");
    const size_t mw = (size_t)(T_in + 31) / 32, G = (size_t)N / 64;
    vp->b0 = (size_t)N * mw * 4;
    vp->b1 = (size_t)T * G * 8;
    vp->b2 = (size_t)3 * n_chunks * 4;
    vp->off1 = (vp->b0 + 15) & ~size_t(15);
    vp->off2 = vp->off1 + ((vp->b1 + 15) & ~size_t(15));
    vp->need = vp->off2 + vp->b2;
    return MIBC_OK;
}
// Fills the blob (host memory of vp->need bytes, laid out by var_layout): sample bitmap, row-mask words, decoder table.
static int var_build(mibc_engine *e, int N, int T_in, const mibc_var_chunk *ch, int n_chunks, VarPlan *vp, char *blob) {
    if (!ch) return fail(e, MIBC_ERR_ARG, "no chunks");
    const int stride = e->stride, T = mibc_output_steps(e, T_in);
    const int mw = (T_in + 31) / 32, G = N / 64;
    const int nsub = (N + e->Nd - 1) / e->Nd;
    memset(blob, 0, vp->off2);
    uint32_t *smask = (uint32_t *)blob;
    unsigned long long *tmask = (unsigned long long *)(blob + vp->off1);
    int *idx = (int *)(blob + vp->off2);
    std::vector<int> row_end((size_t)N, -2);   // last occupied step per row (chunks of a row must come in order)
    std::vector<int> count((size_t)nsub + 1, 0);
    for (int c = 0; c < n_chunks; ++c) {
        const int r = ch[c].row, s0 = ch[c].sample_start, L = ch[c].n_samples;
        if (r < 0 || r >= N || s0 < 0 || L <= 0 || s0 % stride || L % stride || s0 + L > T_in)
            return fail(e, MIBC_ERR_ARG, "variable chunks: chunk outside its row or not stride aligned");
        ++count[(size_t)(r / e->Nd) + 1];
    }
    for (int k = 0; k < nsub; ++k) count[(size_t)k + 1] += count[(size_t)k];
    vp->sub_begin = count;
    std::vector<int> next(count.begin(), count.end() - 1);
    std::vector<long> brow((size_t)nsub, 0);
    int Tmax = 0;
    for (int c = 0; c < n_chunks; ++c) {
        const int r = ch[c].row, s0 = ch[c].sample_start, L = ch[c].n_samples;
        const int t0 = s0 / stride, Tc = L / stride;
        if (t0 < row_end[r] + 3 && row_end[r] >= 0)
            return fail(e, MIBC_ERR_ARG, "variable chunks: chunks of a row must be ordered and >= 2 steps apart");
        row_end[r] = t0 + Tc - 1;
        // bits [s0, s0 + L) of the row's sample bitmap, a word at a time
        uint32_t *sm = smask + (size_t)r * mw;
        const int p1 = s0 + L, w0 = s0 >> 5, w1 = (p1 - 1) >> 5;
        const uint32_t first = 0xffffffffu << (s0 & 31), last = 0xffffffffu >> (31 - ((p1 - 1) & 31));
        if (w0 == w1) {
            sm[w0] |= first & last;
        } else {
            sm[w0] |= first;
            for (int w = w0 + 1; w < w1; ++w) sm[w] = 0xffffffffu;
            sm[w1] |= last;
        }
        const unsigned long long rbit = 1ull << (r & 63);
        unsigned long long *tm = tmask + (size_t)t0 * G + (r >> 6);
        for (int t = 0; t < Tc; ++t) tm[(size_t)t * G] |= rbit;
        const int k = r / e->Nd, slot = next[(size_t)k]++;
        idx[slot] = (r - k * e->Nd) * T + t0;
        idx[n_chunks + slot] = (int)brow[(size_t)k];
        idx[2 * n_chunks + slot] = Tc;
        brow[(size_t)k] += Tc + 1;
        Tmax = Tc > Tmax ? Tc : Tmax;
    }
    for (int k = 0; k < nsub; ++k)
        if (brow[(size_t)k] > (long)e->Nd * (T + 1))
            return fail(e, MIBC_ERR_ARG, "variable chunks: too many chunks for the decode workspace");
    vp->Tmax = Tmax;
    return MIBC_OK;
}

static void var_clear(mibc_engine *e) {
    e->in_smask = nullptr;
    e->in_tmask = nullptr;
    e->var_idx = nullptr;
    e->in_ss = nullptr;
}
static void var_point(mibc_engine *e, const VarPlan &vp, const char *dev_blob, const float *shift_scale_dev) {
    e->in_smask = (const uint32_t *)dev_blob;
    e->in_tmask = (const unsigned long long *)(dev_blob + vp.off1);
    e->var_idx = (const int *)(dev_blob + vp.off2);
    e->in_ss = shift_scale_dev;
}

// Synchronous set-up on the engine's stream (mibc_forward_var / mibc_call_device_var / mibc_call_var).
static int var_setup(mibc_engine *e, int N, int T_in, const mibc_var_chunk *ch, int n_chunks, VarPlan *vp) {
    int rc = var_layout(e, N, T_in, n_chunks, vp);
    if (rc != MIBC_OK) return rc;
    std::vector<char> blob(vp->need);
    rc = var_build(e, N, T_in, ch, n_chunks, vp, blob.data());
    if (rc != MIBC_OK) return rc;
    if (vp->need > e->var_scratch_bytes) {
        HIP_OK(e, hipStreamSynchronize(e->stream));
        if (e->var_scratch) (void)hipFree(e->var_scratch);
        e->var_scratch = nullptr;
        HIP_OK(e, hipMalloc(&e->var_scratch, vp->need));
        e->var_scratch_bytes = vp->need;
    }
    HIP_OK(e, hipMemcpyAsync(e->var_scratch, blob.data(), vp->need, hipMemcpyHostToDevice, e->stream));
    HIP_OK(e, hipStreamSynchronize(e->stream));   // pageable source: it dies with this frame
    return MIBC_OK;
}

extern "C" int mibc_forward_var(mibc_engine *e, const void *in_dev, const float *shift_scale_dev, int N, int T_in,
                                const mibc_var_chunk *chunks, int n_chunks, uint16_t *scores_dev) {
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    VarPlan vp;
    rc = var_setup(e, N, T_in, chunks, n_chunks, &vp);
    if (rc != MIBC_OK) return rc;
    var_point(e, vp, (const char *)e->var_scratch, shift_scale_dev);
    rc = mibc_forward(e, (const uint16_t *)in_dev, N, T_in, scores_dev);
    var_clear(e);
    return rc;
}

// Network + per-chunk decode of a variable-chunk batch whose masks / decoder table are already on their way to dev_blob
// (in stream order).  (async)
static int var_run(mibc_engine *e, const void *in_dev, const float *shift_scale_dev, int N, int T_in, int n_chunks,
                   const VarPlan &vp, const char *dev_blob, const mibc_decode_opts *o, int8_t *out_dev) {
    const int T = mibc_output_steps(e, T_in);
    var_point(e, vp, dev_blob, shift_scale_dev);
    int rc = overlap_join(e, e->stream);      // variable batches stay serial: they use the first scores buffer on this stream
    if (rc != MIBC_OK) return rc;
    rc = run_encoder(e, (const half_t *)in_dev, N, T_in);
    if (rc == MIBC_OK) {
        // gaps of the output planes stay zero
        if (hipMemsetAsync(out_dev, 0, (size_t)3 * N * T, e->stream) != hipSuccess) rc = MIBC_ERR_HIP;
    }
    // head + per-chunk decode, one decode sub-batch of rows at a time (the scores / guide workspace is sized for e->Nd rows)
    int si = 0;
    for (int n0 = 0; n0 < N && rc == MIBC_OK; n0 += e->Nd, ++si) {
        const int ns = (N - n0 < e->Nd) ? (N - n0) : e->Nd;
        const int c0 = vp.sub_begin[(size_t)si], nc = vp.sub_begin[(size_t)si + 1] - c0;
        if (nc == 0) continue;
        rc = run_head(e, N, T, n0, ns, e->scores);
        if (rc == MIBC_OK &&
            mibc_launch_decode_var(e->stream, e->scores, nc, vp.Tmax, e->S, o->beam_width, o->beam_cut, o->blank_score,
                                   clamp_value(e), o->q_shift, o->q_scale, e->bwd, e->trace, e->path_state,
                                   out_dev + (size_t)n0 * T, (size_t)N * T, nullptr, e->var_idx + c0,
                                   e->var_idx + n_chunks + c0, e->var_idx + 2 * n_chunks + c0) != 0)
            rc = fail(e, MIBC_NOT_SUPPORTED, "decoder: beam_width must be 1..32");
    }
    var_clear(e);
    if (rc != MIBC_OK) return rc;
    e->last_N = N;
    e->last_T = T;
    e->last_T_in = T_in;
    e->timed[e->ps] = false;
    HIP_OK(e, hipGetLastError());
    return MIBC_OK;
}

extern "C" int mibc_call_device_var(mibc_engine *e, const void *in_dev, const float *shift_scale_dev, int N, int T_in,
                                    const mibc_var_chunk *chunks, int n_chunks, const mibc_decode_opts *o,
                                    int8_t *out_dev) {
    if (!o) return MIBC_ERR_ARG;
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    VarPlan vp;
    rc = var_setup(e, N, T_in, chunks, n_chunks, &vp);
    if (rc != MIBC_OK) return rc;
    return var_run(e, in_dev, shift_scale_dev, N, T_in, n_chunks, vp, (const char *)e->var_scratch, o, out_dev);
}

extern "C" int mibc_call_var(mibc_engine *e, const void *in_host, const float *shift_scale_host, int N, int T_in,
                             const mibc_var_chunk *chunks, int n_chunks, const mibc_decode_opts *o, int8_t *out_host) {
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    const int T = mibc_output_steps(e, T_in);
    HIP_OK(e, hipMemcpyAsync(e->in_stage, in_host, (size_t)N * T_in * 2, hipMemcpyHostToDevice, e->stream));
    if (shift_scale_host)
        HIP_OK(e, hipMemcpyAsync(e->ss_stage, shift_scale_host, (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice,
                                 e->stream));
    rc = mibc_call_device_var(e, e->in_stage, shift_scale_host ? e->ss_stage : nullptr, N, T_in, chunks, n_chunks, o,
                              e->out3);
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipMemcpyAsync(out_host, e->out3, (size_t)3 * N * T, hipMemcpyDeviceToHost, e->stream));
    HIP_OK(e, hipStreamSynchronize(e->stream));
    return check_cluster_error(e);
}

// Two-phase form of mibc_call_var on the slot machinery of mibc_call_async (CudaModelRunner's variable-chunk batches run
// on the same stream pipeline as fixed ones: basecall/CudaModelRunner.cpp:21-49, CudaCaller.cpp:645-719): the chunk table
// is turned into the masks / decoder table in a PER-SLOT pinned buffer and copied on the copy stream next to the input
// rows, so a variable batch overlaps its copies with the kernels of the batch in front of it exactly like a fixed one and
// two variable batches can be in flight.  chunks_host need not outlive the call.  Completion: mibc_call_poll / _wait.
extern "C" int mibc_call_var_async(mibc_engine *e, int slot, const void *in_host, const float *shift_scale_host, int N,
                                   int T_in, const mibc_var_chunk *chunks_host, int n_chunks, const mibc_decode_opts *o,
                                   int8_t *out_host) {
    if (!e || !in_host || !out_host || !o || slot < 0 || slot > 1) return MIBC_ERR_ARG;
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    rc = async_prepare(e, slot, N, T_in);
    if (rc != MIBC_OK) return rc;
    auto &a = e->aslot[slot];
    VarPlan vp;
    rc = var_layout(e, N, T_in, n_chunks, &vp);
    if (rc != MIBC_OK) return rc;
    if (a.var_bytes < vp.need) {
        // (the slot was waited for before this re-submission: nothing of it is in flight)
        if (a.var_dev) (void)hipFree(a.var_dev);
        if (a.var_host) (void)hipHostFree(a.var_host);
        a.var_dev = nullptr; a.var_host = nullptr; a.var_bytes = 0;
        const size_t cap = vp.need + vp.need / 4;       // the table grows with the number of chunks
        HIP_OK(e, hipMalloc((void **)&a.var_dev, cap));
        HIP_OK(e, hipHostMalloc((void **)&a.var_host, cap, hipHostMallocDefault));
        a.var_bytes = cap;
    }
    rc = var_build(e, N, T_in, chunks_host, n_chunks, &vp, a.var_host);
    if (rc != MIBC_OK) return rc;
    const int T = mibc_output_steps(e, T_in);
    HIP_OK(e, hipMemcpyAsync(a.var_dev, a.var_host, vp.need, hipMemcpyHostToDevice, e->s_in));
    HIP_OK(e, hipMemcpyAsync(a.in, in_host, (size_t)N * T_in * 2, hipMemcpyHostToDevice, e->s_in));
    if (shift_scale_host)
        HIP_OK(e, hipMemcpyAsync(a.ss, shift_scale_host, (size_t)N * 2 * sizeof(float), hipMemcpyHostToDevice, e->s_in));
    HIP_OK(e, hipEventRecord(a.ev_in, e->s_in));
    HIP_OK(e, hipStreamWaitEvent(e->stream, a.ev_in, 0));
    e->err_slot = slot;
    rc = var_run(e, a.in, shift_scale_host ? a.ss : nullptr, N, T_in, n_chunks, vp, a.var_dev, o, a.out3);
    e->err_slot = 2;
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipEventRecord(a.ev_done, e->stream));
    HIP_OK(e, hipStreamWaitEvent(e->s_out, a.ev_done, 0));
    HIP_OK(e, hipMemcpyAsync(out_host, a.out3, (size_t)3 * N * T, hipMemcpyDeviceToHost, e->s_out));
    HIP_OK(e, hipEventRecord(a.ev_out, e->s_out));
    a.n = N;
    return MIBC_OK;
}

static int stage_ms_of(mibc_engine *e, int set, mibc_stage_ms *out) {
    if (!e || !out) return MIBC_ERR_ARG;
    memset(out, 0, sizeof(*out));
    if (!e->timed[set]) return fail(e, MIBC_ERR_ARG, "no profiled mibc_call_device yet (mibc_set_profile(1))");
    HIP_OK(e, hipSetDevice(e->device));
    HIP_OK(e, hipEventSynchronize(e->ev[set][mibc_engine::EV_END]));
    float ms = 0;
    HIP_OK(e, hipEventElapsedTime(&ms, e->ev[set][mibc_engine::EV_START], e->ev[set][mibc_engine::EV_CONV]));
    out->conv = ms;
    hipEvent_t prev = e->ev[set][mibc_engine::EV_CONV];
    if (e->is_tx) {  // the whole encoder stack + upsample is reported in the "lstm" slot
        HIP_OK(e, hipEventElapsedTime(&ms, prev, e->ev[set][mibc_engine::EV_LSTM0]));
        out->lstm = out->lstm_layer[0] = ms;
    }
    for (int l = 0; !e->is_tx && l < e->d.lstm_layers && l < 8; ++l) {
        HIP_OK(e, hipEventElapsedTime(&ms, prev, e->ev[set][mibc_engine::EV_LSTM0 + l]));
        out->lstm_layer[l] = ms;
        out->lstm += ms;
        prev = e->ev[set][mibc_engine::EV_LSTM0 + l];
    }
    const int nsub = (e->prof_N[set] + e->Nd - 1) / e->Nd;
    for (int si = 0; si < nsub; ++si) {
        HIP_OK(e, hipEventElapsedTime(&ms, e->sub_ev[set][si * 3 + 0], e->sub_ev[set][si * 3 + 1]));
        out->head += ms;
        HIP_OK(e, hipEventElapsedTime(&ms, e->sub_ev[set][si * 3 + 1], e->sub_ev[set][si * 3 + 2]));
        out->decode += ms;
    }
    HIP_OK(e, hipEventElapsedTime(&ms, e->ev[set][mibc_engine::EV_START], e->ev[set][mibc_engine::EV_END]));
    out->total = ms;
    return MIBC_OK;
}
extern "C" int mibc_get_stage_ms(mibc_engine *e, mibc_stage_ms *out) { return e ? stage_ms_of(e, e->ps, out) : MIBC_ERR_ARG; }
// stage times of the profiled call BEFORE the one enqueued last (blocks only until that earlier call has finished)
extern "C" int mibc_get_stage_ms_prev(mibc_engine *e, mibc_stage_ms *out) { return e ? stage_ms_of(e, e->ps ^ 1, out) : MIBC_ERR_ARG; }

// min of 2 timed forward runs (network only), like CudaCaller.cpp:552-569
extern "C" int mibc_time_forward(mibc_engine *e, int N, int T_in, float *ms_out) {
    int rc = check_call(e, N, T_in);
    if (rc != MIBC_OK) return rc;
    const int T = mibc_output_steps(e, T_in);
    rc = overlap_join(e, e->stream);
    if (rc != MIBC_OK) return rc;
    HIP_OK(e, hipMemsetAsync(e->in_stage, 0, (size_t)N * T_in * 2, e->stream));
    float best = 1e30f;
    hipEvent_t a, b;
    HIP_OK(e, hipEventCreate(&a));
    HIP_OK(e, hipEventCreate(&b));
    const int save = e->profile;
    e->profile = 0;
    for (int it = 0; it < 3; ++it) {
        HIP_OK(e, hipEventRecord(a, e->stream));
        rc = e->is_tx ? tx_run_network(e, e->in_stage, N, T_in) : run_encoder(e, e->in_stage, N, T_in);
        if (rc == MIBC_OK)
            for (int n0 = 0; n0 < N && rc == MIBC_OK; n0 += e->Nd) {
                const int ns = (N - n0 < e->Nd) ? (N - n0) : e->Nd;
                rc = e->is_tx ? tx_run_head(e, N, n0, ns, e->scores) : run_head(e, N, T, n0, ns, e->scores);
            }
        HIP_OK(e, hipEventRecord(b, e->stream));
        HIP_OK(e, hipEventSynchronize(b));
        float ms = 0;
        HIP_OK(e, hipEventElapsedTime(&ms, a, b));
        if (it > 0 && ms < best) best = ms;  // first run is warm-up
    }
    e->profile = save;
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    if (rc != MIBC_OK) return rc;
    *ms_out = best;
    return MIBC_OK;
}

extern "C" int mibc_debug_tap(mibc_engine *e, int tap, void *host_dst, size_t bytes) {
    if (!e || !host_dst) return MIBC_ERR_ARG;
    HIP_OK(e, hipSetDevice(e->device));
    HIP_OK(e, hipStreamSynchronize(e->stream));
    const void *src = nullptr;
    size_t have = 0;
    const size_t N = (size_t)e->last_N, T = (size_t)e->last_T, Tin = (size_t)e->last_T_in;
    const size_t Nd = N < (size_t)e->Nd ? N : (size_t)e->Nd;
    switch (tap) {
        case 0: src = e->a1_tap; have = N * Tin * 16 * 2; break;
        case 1: src = e->a2p; have = N * e->Tpitch * 16 * 2; break;
        case 2: src = e->xa; have = T * N * e->C * 2; break;
        case 3:
            src = e->lstm_out;
            have = e->is_tx ? (size_t)N * e->tx.T_tok * e->C * 2 : T * N * e->C * 2;
            break;
        case 4: src = e->bwd; have = Nd * (T + 1) * e->S * 4; break;
        case 5: src = e->prob_tap; have = Nd * T * 4; break;
        default: return fail(e, MIBC_ERR_ARG, "unknown tap");
    }
    if (!src) return fail(e, MIBC_ERR_ARG, "tap not recorded (set MIBC_TAPS=1 before mibc_create)");
    if (bytes > have) return fail(e, MIBC_ERR_ARG, "tap: requested more bytes than recorded");
    HIP_OK(e, hipMemcpy(host_dst, src, bytes, hipMemcpyDeviceToHost));
    return MIBC_OK;
}


#ifdef MIBC_DEBUG_KERNELS
// Microbenchmark (debug library only): C[M][N] = A[M][K] . B[N][K]^T, avg ms over `iters`.
MIBC_HOOK int mibc_debug_gemm(int M, int N, int K, int dbg, int iters, float *ms_out) {
    half_t *A = nullptr, *B = nullptr, *C = nullptr;
    if (hipMalloc((void **)&A, (size_t)M * K * 2) != hipSuccess) return -1;
    if (hipMalloc((void **)&B, (size_t)N * K * 2) != hipSuccess) return -1;
    if (hipMalloc((void **)&C, (size_t)M * N * 2) != hipSuccess) return -1;
    // RANDOM operands (cdna_hip_programming.md 5.4 rule 25: constant fills run at a 15-20 % higher clock): a 32 MB block of
    // uniform [-1, 1) halfs (B scaled by 1/16) replicated over the operands
    {
        const size_t blk = 16u << 20;
        std::vector<half_t> h(blk);
        uint32_t sd = 4242u;
        for (auto &v : h) {
            sd = sd * 1664525u + 1013904223u;
            v = (half_t)((float)((sd >> 9) & 0x7fff) / 16384.0f - 1.0f);
        }
        for (size_t o = 0; o < (size_t)M * K; o += blk)
            (void)hipMemcpy(A + o, h.data(), std::min(blk, (size_t)M * K - o) * 2, o == 0 ? hipMemcpyHostToDevice : hipMemcpyHostToDevice);
        for (auto &v : h) v = (half_t)((float)v * 0.0625f);
        for (size_t o = 0; o < (size_t)N * K; o += blk)
            (void)hipMemcpy(B + o, h.data(), std::min(blk, (size_t)N * K - o) * 2, hipMemcpyHostToDevice);
    }
    GemmArgs g{};
    g.A = A; g.B = B; g.out = C; g.M = M; g.Ncols = N; g.K = K;
    g.a_div = 1 << 30; g.a_inner = K; g.o_div = 1 << 30; g.o_inner = N; g.act = -1; g.dbg = dbg;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    mibc_launch_gemm_tn(nullptr, &g);
    (void)hipEventRecord(a, nullptr);
    for (int i = 0; i < iters; ++i) mibc_launch_gemm_tn(nullptr, &g);
    (void)hipEventRecord(b, nullptr);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    *ms_out = ms / iters;
    (void)hipFree(A); (void)hipFree(B); (void)hipFree(C);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}


// Test entry (not part of the public ABI): runs one GEMM shape through gemm256_kernel and through gemm_dma_kernel on
// the same pseudo-random operands and reports how many output halfs differ (contract: none — same arithmetic) and
// both run times.  epi: 0 plain (act, optional bias), 1 rotary + transposed V (N = 3 * d_model, T = rope_T), 2 SwiGLU.
// dbg0: which kernel runs as the first of the two: 0 = the production choice (gemm256x_kernel for the plain / rotary
// epilogues with K <= 1024, else gemm256_kernel), 0x1000 = gemm256_kernel (32 x 32 x 16: bit-identical to gemm_dma_kernel).
// err_f64 (plain epilogue only, else -1): largest deviation of the first kernel from an f64 host product on 4096 sampled
// outputs.
MIBC_HOOK int mibc_debug_gemm_compare(int M, int N, int K, int epi, int act, int use_bias, int rope_T, int iters, int dbg0,
                                       long long *ndiff, float *maxdiff, float *ms_256, float *ms_128, float *err_f64) {
    auto lcg = [](uint32_t &s) {
        s = s * 1664525u + 1013904223u;
        return (float)((s >> 9) & 0x7fff) / 16384.0f - 1.0f;   // [-1, 1)
    };
    uint32_t seed = 12345u + (uint32_t)(M + 3 * N + 7 * K + epi);
    std::vector<half_t> hA((size_t)M * K), hB((size_t)N * K);
    for (auto &v : hA) v = (half_t)lcg(seed);
    for (auto &v : hB) v = (half_t)(lcg(seed) * 0.08f);
    std::vector<float> hbias((size_t)N), hrope((size_t)(rope_T > 0 ? rope_T : 1) * 64);
    for (auto &v : hbias) v = lcg(seed);
    for (size_t i = 0; i < hrope.size() / 2; ++i) {
        const float ang = 3.0f * lcg(seed);
        hrope[2 * i] = cosf(ang);
        hrope[2 * i + 1] = sinf(ang);
    }
    half_t *A = nullptr, *B = nullptr, *C1 = nullptr, *C2 = nullptr, *V1 = nullptr, *V2 = nullptr;
    float *bias = nullptr, *rope = nullptr;
    const size_t ocols = (epi == 2) ? (size_t)N / 2 : (size_t)N;
    const size_t obytes = (size_t)M * ocols * 2, vbytes = (size_t)M * (N / 3) * 2;
    if (hipMalloc((void **)&A, hA.size() * 2) != hipSuccess || hipMalloc((void **)&B, hB.size() * 2) != hipSuccess ||
        hipMalloc((void **)&C1, obytes) != hipSuccess || hipMalloc((void **)&C2, obytes) != hipSuccess ||
        hipMalloc((void **)&bias, hbias.size() * 4) != hipSuccess || hipMalloc((void **)&rope, hrope.size() * 4) != hipSuccess ||
        hipMalloc((void **)&V1, vbytes) != hipSuccess || hipMalloc((void **)&V2, vbytes) != hipSuccess)
        return -1;
    (void)hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(bias, hbias.data(), hbias.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(rope, hrope.data(), hrope.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemset(C1, 0, obytes);
    (void)hipMemset(C2, 0, obytes);
    (void)hipMemset(V1, 0, vbytes);
    (void)hipMemset(V2, 0, vbytes);
    GemmArgs g{};
    g.A = A; g.B = B; g.bias = use_bias ? bias : nullptr; g.M = M; g.Ncols = N; g.K = K;
    g.a_div = 1 << 30; g.a_inner = K; g.o_div = 1 << 30; g.o_inner = (long)ocols; g.act = act; g.epi_mode = epi;
    if (epi == 1) { g.rope = rope; g.rope_T = rope_T; g.rope_cols = 2 * (N / 3); }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
        g.out = which ? C2 : C1;
        g.vT = (epi == 1) ? (which ? V2 : V1) : nullptr;
        g.dbg = which ? 0x100 : dbg0;
        if (mibc_launch_gemm_tn(nullptr, &g) != 0) return -2;
        (void)hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) mibc_launch_gemm_tn(nullptr, &g);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms[which], e0, e1);
        ms[which] /= (float)(iters > 0 ? iters : 1);
    }
    if (hipDeviceSynchronize() != hipSuccess) return -3;
    std::vector<half_t> o1(obytes / 2), o2(obytes / 2), v1(vbytes / 2), v2(vbytes / 2);
    (void)hipMemcpy(o1.data(), C1, obytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(o2.data(), C2, obytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(v1.data(), V1, vbytes, hipMemcpyDeviceToHost);
    (void)hipMemcpy(v2.data(), V2, vbytes, hipMemcpyDeviceToHost);
    long long nd = 0;
    float md = 0.0f, amax = 0.0f;
    auto cmp = [&](const std::vector<half_t> &x, const std::vector<half_t> &y) {
        for (size_t i = 0; i < x.size(); ++i) {
            uint16_t a, b;
            memcpy(&a, &x[i], 2);
            memcpy(&b, &y[i], 2);
            if (a != b) {
                ++nd;
                const float d = fabsf((float)x[i] - (float)y[i]);
                if (!(d <= md)) md = d;
            }
            amax = fmaxf(amax, fabsf((float)y[i]));
        }
    };
    cmp(o1, o2);
    if (epi == 1) cmp(v1, v2);
    if (amax == 0.0f) nd = -1;   // nothing was written: the comparison is void
    double ef = -1.0;
    if (epi == 0 && (act == -1 || act == 3)) {
        ef = 0.0;
        uint32_t sd = 99u;
        for (int it = 0; it < 4096; ++it) {
            sd = sd * 1664525u + 1013904223u;
            const size_t m = (size_t)(sd >> 8) % (size_t)M;
            sd = sd * 1664525u + 1013904223u;
            const size_t n = (size_t)(sd >> 8) % (size_t)N;
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += (double)(float)hA[m * K + k] * (double)(float)hB[n * K + k];
            if (use_bias) acc += (double)hbias[n];
            if (act == 3) acc = 5.0 * tanh(acc);
            ef = std::max(ef, fabs(acc - (double)(float)o1[m * ocols + n]));
        }
    }
    if (err_f64) *err_f64 = (float)ef;
    *ndiff = nd;
    *maxdiff = md;
    *ms_256 = ms[0];
    *ms_128 = ms[1];
    for (void *q : {(void *)A, (void *)B, (void *)C1, (void *)C2, (void *)V1, (void *)V2, (void *)bias, (void *)rope}) (void)hipFree(q);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}
#endif   // MIBC_DEBUG_KERNELS
