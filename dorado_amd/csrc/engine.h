// dorado_amd/csrc/engine.h — internal declarations shared by engine.hip (C-ABI, LSTM-CRF path)
// and engine_tx.hip (transformer path).  Not part of the public ABI (that is include/mibc.h).
#pragma once
#include "../../include/mibc.h"
#include "common.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

struct GemmArgs {
    const half_t *A;
    const half_t *B;    // [Ncols][K]
    const float *bias;  // [Ncols] or nullptr
    half_t *out;
    int M, Ncols, K;    // K multiple of 32, Ncols multiple of 128
    int a_div;
    long a_outer, a_inner;
    int o_div;
    long o_outer, o_inner;
    int act;            // -1 identity, 0/1/2 as MIBC_ACT_*, 3 = 5*tanh
    int ncols_valid;    // 0 = all; else columns >= ncols_valid are computed (zero weights) but not stored
    // epilogue fusions of the transformer path (tx.hip):
    //  mode 1: rotary embedding on q and k (columns < rope_cols), head_dim 64, half-split pairs
    //          (c, c+32); table rope[t][32] = {cos, sin} interleaved as float2, t = m % rope_T
    //  mode 2: SwiGLU: each 128-column tile holds 64 "y" then 64 "gate" features; writes
    //          silu(gate) * y to 64 output columns (out row stride = Ncols / 2)
    int epi_mode;
    const float *rope;
    int rope_T, rope_cols;
    half_t *vT;         // epi_mode 1: if set, columns >= 2*rope_cols/2.. (the V third) go to vT[n][h][64][rope_T]
    int dbg;            // debug ablation bits (microbenchmark only): 1 no stores, 2 no MFMA, 4 no DMA
};
extern "C" int mibc_launch_gemm_tn(hipStream_t s, const GemmArgs *a);
struct WsArgs {
    const half_t *A;
    const half_t *Wf;
    const float *bias;
    half_t *out;
    int cols;
    int act;
    int N, Ns, n0, T;    // head: full batch, sub-batch, first chunk of the sub-batch, steps
    int Tpitch, stride;  // conv3: a2p rows per chunk, conv stride
    int dbg;             // timing ablations (MIBC_WS_DBG: 1 no activation, 2 no stores, 4 no k-loop, 8 non-temporal tile loads); 0 in production
};
extern "C" int mibc_launch_wsgemm(hipStream_t s, const WsArgs *a, int K, int mode);
extern "C" int mibc_launch_conv12(hipStream_t s, const half_t *x, const float *w1, const float *b1,
                                  const float *w2, const float *b2, half_t *a2p, half_t *a1_tap, const float *ss,
                                  const uint32_t *smask, int N, int T_in, int Tpitch, int pad, int act1, int act2);
extern "C" int mibc_launch_read_stats(hipStream_t s, const int16_t *sig, const long long *off, int n_reads,
                                      int strategy, float qa, float qb, float shift_mult, float scale_mult,
                                      float *out_ss, float *out_raw, uint32_t *scratch);
extern "C" int mibc_launch_svb16_decode(hipStream_t s, const uint8_t *streams, const long long *stream_off,
                                        const long long *sample_off, int n_rows, int16_t *out, int *status);
extern "C" int mibc_launch_scale_reads(hipStream_t s, const int16_t *sig, const long long *off, int n_reads,
                                       const float *ss, half_t *out, int blocks_per_read);
extern "C" int mibc_launch_lstm_layer_masked(hipStream_t s, int C, const half_t *Xin, half_t *Xout,
                                             const half_t *Wf16, const float *biasn, int T, int N, int reverse,
                                             const unsigned long long *tmask);
extern "C" int mibc_launch_lstm_layer(hipStream_t s, int C, const half_t *Xin, half_t *Xout,
                                      const half_t *Wf, const half_t *Wf16, const float *biasn, int T, int N,
                                      int reverse);
extern "C" int mibc_lstm_rows_per_wg(int C);
// lstm_q8.hip: int8 layer (Xin int8 [T][N][C]; Xout int8 or f16) and the f16 -> int8 conversion behind the first layer
extern "C" int mibc_launch_lstm_layer_q8(hipStream_t s, int C, const int8_t *Xin, void *Xout, const int8_t *Wq,
                                         const float *biasn, const float *deqn, int T, int N, int reverse, int out_f16,
                                         const unsigned long long *tmask = nullptr);
extern "C" int mibc_launch_q8_convert(hipStream_t s, const half_t *in, int8_t *out, size_t n);
// lstm_cluster.hip: hidden-split cluster kernel (C = 512 / 768 / 1024, N a multiple of 256); 1 = shape not covered
extern "C" int mibc_launch_lstm_layer_cl(hipStream_t s, int C, const half_t *Xin, half_t *Xout, const half_t *Wt,
                                         const float *biascl, const half_t *zeros, float *cbuf, unsigned *flags,
                                         unsigned *err, int T, int N, int reverse,
                                         const unsigned long long *tmask, int q8 = 0, const float *deqcl = nullptr,
                                         signed char *hx = nullptr);
extern "C" int mibc_launch_decode_var(hipStream_t st, const half_t *scores, int N, int T, int S, int W,
                                      float beam_cut, float stay, float clampv, float q_shift, float q_scale,
                                      float *bwd, uint32_t *trace, uint16_t *path_state, int8_t *out3,
                                      size_t plane_stride, float *prob_tap, const int *coff, const int *boff,
                                      const int *clen);
extern "C" int mibc_launch_decode(hipStream_t st, const half_t *scores, int N, int T, int S, int W,
                                  float beam_cut, float stay, float clampv, float q_shift,
                                  float q_scale, float *bwd, uint32_t *trace,
                                  uint16_t *path_state, int8_t *out3, size_t plane_stride,
                                  float *prob_tap);

// Stage markers for rocprofv3 traces (the role of utils::ScopedProfileRange, torch_utils/include/torch_utils/
// gpu_profiling.h:32-99, which wraps nvtx ranges): roctx push / pop around the stages of a call when the engine's profile
// level is >= 2 (mibc_set_profile).  libroctx64 is looked up at run time (dlopen), so the product has no link dependency.
struct MibcRange {
    bool on;
    MibcRange(const struct mibc_engine *e, const char *name);
    ~MibcRange();
};

std::string &mibc_gerr();
#define g_err (mibc_gerr())

struct mibc_engine {
    int device = 0;
    mibc_model_desc d{};
    hipStream_t stream = nullptr;
    std::string err;
    // weights (device)
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *b3 = nullptr;
    half_t *w3 = nullptr;  // [C][K3pad]
    half_t *w3f = nullptr, *head_w1f = nullptr;  // 16x32 MFMA-fragment order (wsgemm.hip)
    int use_ws = 1;
    int fuse_q8 = 1;    // conv3 epilogue writes the int8 rows of the quantised LSTM itself (debug library: MIBC_FUSE_Q8=0 -> separate pass)
    int K3 = 0, K3pad = 0;
    std::vector<half_t *> lstm_w;    // 32-unit tiles, k-steps of 16 (v_mfma 32x32x16)
    std::vector<half_t *> lstm_w16;  // 16-unit tiles, k-steps of 32 (v_mfma 16x16x32); C <= 384 only
    std::vector<float *> lstm_bn;  // b_ih + b_hh, [C/32][4][32]
    // quantised path (lstm_q8.hip): int8 [W_ih | W_hh] in 16x64 fragment order, 1 / (127 * row scale) in bias order
    std::vector<int8_t *> lstm_wq;
    std::vector<float *> lstm_deq;
    // cluster kernel (lstm_cluster.hip), C = 512 / 768 / 1024 only
    std::vector<half_t *> lstm_wcl;  // [C/128 members][2 passes][2C/32 slabs][256 gate rows][32] swizzled LDS images
    std::vector<float *> lstm_bcl;   // [C/128][2][2][4][32]
    // quantised cluster layers (lstm_quant, layers >= 1): int8 slab images [C/128][2][2C/64][256][64], int32 round(bias / deq)
    // and the dequantisation factors, both in lstm_bcl order
    std::vector<int8_t *> lstm_wclq;
    std::vector<float *> lstm_bclq, lstm_dqcl;
    half_t *lstm_zero = nullptr;     // [256][C] zeros (h_{-1})
    unsigned *cl_flags = nullptr;    // [N_res/256][C/128][16] completed-step counters (zeroed per launch)
    float *cl_cstate = nullptr;      // [N_res][C] f32 cell state of the layer in flight (workspace)
    unsigned *cl_err = nullptr;      // device [4]: sticky hand-off time-out word
    unsigned *cl_err_host = nullptr; // pinned [3][4]: copy per call slot (async slot 0 / 1, synchronous calls 2),
                                     // taken in stream order behind the LSTM stack of that call
    int err_slot = 2;                // where the call being enqueued reports
    bool cl_used = false;
    int use_cluster = 1;             // debug build: MIBC_LSTM_CLUSTER=0 forces the per-workgroup kernels
    half_t *head_w1 = nullptr, *head_w2 = nullptr;
    float *head_b1 = nullptr;
    int head_act1 = -1, head_act2 = -1;
    // f1 (ScalerNode): per-chunk (shift, scale) of the int16 input of the call in flight, or nullptr
    const float *in_ss = nullptr;
    // f3 (variable chunks): masks of the call in flight (device), or nullptr
    const uint32_t *in_smask = nullptr;     // [N][(T_in+31)/32] sample bitmap
    const unsigned long long *in_tmask = nullptr;  // [T][N/64] valid-step bitmap per 64-row workgroup
    const int *var_idx = nullptr;           // coff | boff | clen, n_chunks each
    void *var_scratch = nullptr;            // smask | tmask | var_idx, grown on demand
    size_t var_scratch_bytes = 0;
    float *ss_stage = nullptr;          // device copy of host-provided pairs (mibc_call_i16)
    uint32_t *stats_scratch = nullptr;  // [256][2][65536] wide-range histograms, allocated on first use
    // geometry
    int C = 0, S = 0, K = 0, stride = 1, pad3 = 0;
    // workspace
    // N_res / T_in_cap: what the workspace was allocated for (capacity); T_in_res / T_res / Tpitch: geometry of the
    // call in flight — one engine serves every chunk size of its device (CudaCaller's m_batch_dims,
    // CudaCaller.cpp:382-413), any (N <= N_res, T_in <= T_in_cap) runs in the same buffers
    int N_res = 0, T_in_cap = 0, T_in_res = 0, T_res = 0, Tpitch = 0, Nd = 0;
    size_t a2p_bytes = 0;
    half_t *in_stage = nullptr, *a2p = nullptr, *xa = nullptr, *xb = nullptr, *scores = nullptr,
           *mid = nullptr, *a1_tap = nullptr;
    float *bwd = nullptr, *prob_tap = nullptr;
    uint32_t *trace = nullptr;
    uint16_t *path_state = nullptr;
    int8_t *out3 = nullptr;
    size_t ws_bytes = 0;
    // two-phase calls (mibc_call_async / mibc_call_wait): per-slot device staging + copy streams, so that the
    // H2D copy of batch i+1 and the D2H copy of batch i-1 run beside the kernels of batch i
    struct AsyncSlot {
        half_t *in = nullptr;
        float *ss = nullptr;
        int8_t *out3 = nullptr;
        hipEvent_t ev_in = nullptr, ev_done = nullptr, ev_out = nullptr;
        size_t in_bytes = 0, out_bytes = 0;
        int n = 0;
        char *var_dev = nullptr, *var_host = nullptr;   // mibc_call_var_async: masks + decoder table (device / pinned host)
        size_t var_bytes = 0;
    } aslot[2];
    hipStream_t s_in = nullptr, s_out = nullptr;
    // mibc_set_decode_overlap: decoder stream, second scores buffer, per-parity events (head done -> decoder may read;
    // decoder done -> the head two (sub-)batches later may overwrite)
    int decode_overlap = 0;
    hipStream_t s_dec = nullptr;
    half_t *scores2 = nullptr;
    hipEvent_t ev_head[2] = {nullptr, nullptr}, ev_dec[2] = {nullptr, nullptr};
    bool dec_pending[2] = {false, false};
    unsigned dec_parity = 0;
    // last call
    half_t *lstm_out = nullptr;
    int last_N = 0, last_T = 0, last_T_in = 0;
    int profile = 0, taps = 0;
    enum { EV_START, EV_CONV, EV_LSTM0, EV_HEAD_BASE = EV_LSTM0 + 8, EV_END = EV_HEAD_BASE + 1, EV_N };
    // two sets of stage events, alternating per profiled call: the host reads the set of call i - 1 (mibc_get_stage_ms_prev)
    // while call i is already enqueued, so per-step stage times cost no pipeline bubble
    hipEvent_t ev[2][16] = {};
    float head_ms = 0, dec_ms = 0;
    std::vector<hipEvent_t> sub_ev[2];  // per decode sub-batch: head start, head end, decode end
    bool timed[2] = {false, false};
    int prof_N[2] = {0, 0};
    int ps = 0;                         // set of the profiled call enqueued last
    // ---- transformer model (engine_tx.hip) ----
    bool is_tx = false;
    struct TxConv {
        int cin, cout, cout_pad, w, stride, pad, act;
        half_t *wB = nullptr;   // [cout_pad][w*cin] f16 (im2col order k = tap*cin + ci)
        float *bias = nullptr;  // [cout_pad]
    };
    struct TxLayer {
        half_t *wqkv = nullptr, *wo = nullptr, *wfc1 = nullptr, *wfc2 = nullptr;
        half_t *wimg = nullptr;   // txlayer.hip: out-proj + MLP weights as one fragment-ordered stream (d_model 512)
        float *bo = nullptr, *n1 = nullptr, *n2 = nullptr;
    };
    struct Tx {
        float *c1w = nullptr, *c1b = nullptr;  // conv1 [5][C1], [C1]
        std::vector<TxConv> convs;             // conv2..n as implicit-im2col GEMMs
        std::vector<TxLayer> layers;
        half_t *wup = nullptr, *wcrf = nullptr;
        float *bup = nullptr, *rope = nullptr;
        int D = 0, H = 0, FF = 0, depth = 0, sf = 1, conv_stride = 1;
        int fused = 1;                // layer tail through txlayer.hip (tests compare against the five-launch path)
        // workspace
        std::vector<half_t *> cbuf;   // padded conv outputs (NTC)
        std::vector<size_t> cbuf_bytes;
        std::vector<int> ctp, ct;     // row pitch and valid steps per conv buffer
        half_t *x = nullptr, *qkv = nullptr, *attn = nullptr, *tmp = nullptr, *ff = nullptr, *up = nullptr;
        half_t *vT = nullptr;         // V transposed [N][H][64][T] (when T % 128 == 0)
        int T_tok = 0;
        hipEvent_t ev_conv = nullptr, ev_layers = nullptr;
    } tx;
};

#define HIP_OK(e_, call)                                                                  \
    do {                                                                                  \
        hipError_t rc_ = (call);                                                          \
        if (rc_ != hipSuccess) {                                                          \
            std::string m_ = std::string(#call) + ": " + hipGetErrorString(rc_);          \
            if (e_) (e_)->err = m_;                                                       \
            g_err = m_;                                                                   \
            return MIBC_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

inline int fail(mibc_engine *e, int code, const std::string &m) {
    if (e) e->err = m;
    g_err = m;
    return code;
}


// txlayer.hip: fused layer tail (out-proj + RMSNorm + gated MLP + RMSNorm), d_model 512
std::vector<half_t> tx_layer_image(const float *wo, const float *w1, const float *w2, int FF);
bool tx_layer_supported(int d_model, int ff);
extern "C" int mibc_launch_tx_layer(hipStream_t s, const half_t *attn, half_t *x, const half_t *wimg, const float *bo,
                                    const float *n1, const float *n2, float alpha, long R, int FF, int mode);

// engine_tx.hip
int tx_create(mibc_engine *e, const mibc_model_desc &d, const float *const *weights, int n_weights);
void tx_destroy(mibc_engine *e);
void tx_free_ws(mibc_engine *e);
int tx_tokens(const mibc_engine *e, int T_in);
int tx_reserve(mibc_engine *e, int N, int T_in, size_t *total);
int tx_set_geometry(mibc_engine *e, int T_in);   // row pitches / token count of the call in flight; re-zeroes pad rows
size_t tx_bytes_per_chunk(const mibc_engine *e, int T_in);
int tx_run_network(mibc_engine *e, const half_t *in_dev, int N, int T_in);
int tx_run_head(mibc_engine *e, int N, int n0, int ns, half_t *scores_out);

template <typename T>
inline int mibc_upload(mibc_engine *e, T **dst, const std::vector<T> &src) {
    HIP_OK(e, hipMalloc((void **)dst, src.size() * sizeof(T)));
    HIP_OK(e, hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

