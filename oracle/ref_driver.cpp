// oracle/ref_driver.cpp — TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Thin extern "C" driver around the *reference's own* CPU hot path, compiled in place from
// /root/reference/dorado (see oracle/Makefile.ref): CRFModel / TxModel (libtorch CPU, f32),
// CPUDecoder (scans + posts + beam_search) and utils::generate_chunks.  This file is our code;
// it contains no reference source, it only *calls* the reference classes:
//   dorado/basecall/model/CRFModel.cpp:29-62,117-128     (model assembly + forward)
//   dorado/basecall/model/TxModel.cpp:10-41
//   dorado/basecall/decode/CPUDecoder.cpp:43-92,100-157  (scans, posts, beam search)
//   dorado/basecall/ModelRunner.cpp:32-45               (NTC -> TNC, decode options)
//   dorado/read_pipeline/base/chunk.cpp:11-47
// Used to (a) pin oracle/oracle.c (the C restatement) and (b) generate tests/golden/*.
#include "basecall/decode/CPUDecoder.h"
#include "basecall/model/CRFModel.h"
#include "basecall/model/TxModel.h"
#include "config/BasecallModelConfig.h"
#include "read_pipeline/base/chunk.h"
#include "torch_utils/tensor_utils.h"

#include <torch/torch.h>

#include <cstdint>
#include <cstring>
#include <random>
#include <string>
#include <vector>

using namespace dorado;

#include "ref_common.h"

extern "C" {
static thread_local std::string g_err;
const char *ref_last_error() { return g_err.c_str(); }
}  // extern "C"

static config::BasecallModelConfig make_config(const RefModelDesc &d) { return ref_make_config(d); }

template <typename M>
static int load_flat(M &module, const float *const *weights, int n_weights) {
    auto params = module->parameters();
    if (int(params.size()) != n_weights) {
        g_err = "weight count mismatch: module has " + std::to_string(params.size()) + ", got " +
                std::to_string(n_weights);
        return -1;
    }
    std::vector<at::Tensor> ws;
    for (size_t i = 0; i < params.size(); ++i) {
        at::Tensor t = torch::empty_like(params[i], torch::kFloat32);
        std::memcpy(t.data_ptr(), weights[i], sizeof(float) * t.numel());
        ws.push_back(t);
    }
    module->load_state_dict(ws);
    return 0;
}

extern "C" {

// Number of parameters and their element counts, in module.parameters() order (the order
// basecall/crf_utils.cpp:34-88 / :100-147 loads them in).
int ref_param_numels(const RefModelDesc *d, int64_t *numels, int max_n) {
    try {
        torch::InferenceMode guard;
        auto cfg = make_config(*d);
        std::vector<at::Tensor> params;
        if (cfg.is_tx_model()) {
            basecall::model::TxModel m(cfg, at::TensorOptions().dtype(torch::kFloat32));
            params = m->parameters();
        } else {
            basecall::model::CRFModel m(cfg);
            params = m->parameters();
        }
        int n = int(params.size());
        for (int i = 0; i < n && i < max_n; ++i) {
            numels[i] = params[i].numel();
        }
        return n;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Forward pass exactly as basecall/ModelRunner.cpp:32-36 drives it (f32, CPU).  in: [N, C, T_in]
// f32. out: scores [N, T, K] f32 (NTC); returns T (or <0 on error).
int ref_forward(const RefModelDesc *d,
                const float *const *weights,
                int n_weights,
                const float *in_NCT,
                int N,
                int T_in,
                float *scores_NTK,
                int64_t scores_capacity) {
    try {
        torch::InferenceMode guard;
        torch::set_num_threads(1);  // torch_utils/torch_utils.cpp:20
        auto cfg = make_config(*d);
        auto x = torch::from_blob(const_cast<float *>(in_NCT), {N, d->num_features, T_in},
                                  torch::kFloat32)
                         .clone();
        at::Tensor y;
        if (cfg.is_tx_model()) {
            basecall::model::TxModel m(cfg, at::TensorOptions().dtype(torch::kFloat32));
            if (load_flat(m, weights, n_weights) != 0) {
                return -1;
            }
            m->eval();
            y = m->forward(x, nullptr);
        } else {
            basecall::model::CRFModel m(cfg);
            if (load_flat(m, weights, n_weights) != 0) {
                return -1;
            }
            m->eval();
            y = m->forward(x, nullptr);
        }
        y = y.contiguous();
        if (y.numel() > scores_capacity) {
            g_err = "scores buffer too small";
            return -2;
        }
        std::memcpy(scores_NTK, y.data_ptr<float>(), sizeof(float) * y.numel());
        return int(y.size(1));
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// CRF scans + posteriors (decode/CPUDecoder.cpp:43-92,130). scores [N,T,K] f32 ->
// fwd/bwd/posts [N,T+1,S] f32 (any may be null).
int ref_scans(const float *scores_NTK, int N, int T, int K, float blank, float *fwd, float *bwd,
              float *posts) {
    try {
        torch::InferenceMode guard;
        torch::set_num_threads(1);
        auto s_TNC = torch::from_blob(const_cast<float *>(scores_NTK), {N, T, K}, torch::kFloat32)
                             .transpose(0, 1)
                             .contiguous();
        auto f = basecall::decode::inner::forward_scores(s_TNC, blank);
        auto b = basecall::decode::inner::backward_scores(s_TNC, blank);
        auto p = at::softmax(f + b, -1);
        auto put = [&](const at::Tensor &t, float *dst) {
            if (dst) {
                auto c = t.transpose(0, 1).contiguous();
                std::memcpy(dst, c.data_ptr<float>(), sizeof(float) * c.numel());
            }
        };
        put(f, fwd);
        put(b, bwd);
        put(p, posts);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Full CPU decode of a batch (decode/CPUDecoder.cpp:100-157 -> beam_search.cpp:522-606).
// scores [N,T,K] f32. Outputs are [N,T] byte planes: moves (0/1), seq (ASCII, NUL padded),
// qstr (ASCII, NUL padded); seqlen[N].
int ref_decode(const float *scores_NTK, int N, int T, int K, int beam_width, float beam_cut,
               float blank, float q_shift, float q_scale, uint8_t *moves, char *seq, char *qstr,
               int *seqlen) {
    try {
        torch::InferenceMode guard;
        torch::set_num_threads(1);
        auto s_TNC = torch::from_blob(const_cast<float *>(scores_NTK), {N, T, K}, torch::kFloat32)
                             .transpose(0, 1)
                             .contiguous();
        basecall::decode::DecoderOptions o;
        o.beam_width = size_t(beam_width);
        o.beam_cut = beam_cut;
        o.blank_score = blank;
        o.q_shift = q_shift;
        o.q_scale = q_scale;
        basecall::decode::CPUDecoder dec;
        auto chunks = dec.beam_search_part_2(dec.beam_search_part_1({s_TNC, N, o}));
        std::memset(seq, 0, size_t(N) * T);
        std::memset(qstr, 0, size_t(N) * T);
        for (int i = 0; i < N; ++i) {
            const auto &c = chunks[i];
            std::memcpy(moves + size_t(i) * T, c.moves.data(), T);
            std::memcpy(seq + size_t(i) * T, c.sequence.data(), c.sequence.size());
            std::memcpy(qstr + size_t(i) * T, c.qstring.data(), c.qstring.size());
            seqlen[i] = int(c.sequence.size());
        }
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// read_pipeline/base/chunk.cpp:11-47. Returns count, or -1 if the reference throws.
int ref_generate_chunks(uint64_t num_samples, uint64_t chunk_size, uint64_t stride,
                        uint64_t overlap, uint64_t *out, int max_out) {
    try {
        auto v = utils::generate_chunks(num_samples, chunk_size, stride, overlap);
        for (size_t i = 0; i < v.size() && int(i) < max_out; ++i) {
            out[i] = v[i];
        }
        return int(v.size());
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// read_pipeline/base/chunk.cpp:49-107. Returns count, or -1 if the reference throws.
int ref_generate_variable_chunks(uint64_t num_samples, uint64_t chunk_size, uint64_t stride, uint64_t overlap,
                                 uint64_t *out_pairs, int max_out) {
    try {
        auto v = utils::generate_variable_chunks(num_samples, chunk_size, stride, overlap);
        for (size_t i = 0; i < v.size() && int(i) < max_out; ++i) {
            out_pairs[2 * i] = v[i].first;
            out_pairs[2 * i + 1] = v[i].second;
        }
        return int(v.size());
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// ---- signal scaling: the reference's own utils, torch_utils/tensor_utils.cpp ----

// utils::quantile_counting (tensor_utils.cpp:217-245)
int ref_quantile_counting(const int16_t *x, long n, const float *q, int nq, float *out) {
    try {
        auto t = at::from_blob(const_cast<int16_t *>(x), {n}, at::kShort).clone();
        auto qt = at::from_blob(const_cast<float *>(q), {nq}, at::kFloat).clone();
        auto r = utils::quantile_counting(t, qt);
        for (int i = 0; i < nq; ++i) out[i] = r[i].item<float>();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// utils::shift_scale_tensor_i16_to_f16_inplace (tensor_utils.cpp:89-142); out = f16 bit patterns
int ref_shift_scale_i16_to_f16(const int16_t *x, long n, float shift, float scale, uint16_t *out) {
    try {
        auto t = at::from_blob(const_cast<int16_t *>(x), {n}, at::kShort).clone();
        utils::shift_scale_tensor_i16_to_f16_inplace(t, shift, scale);
        std::memcpy(out, t.data_ptr(), size_t(n) * 2);
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// The expression of the file-local med_mad() in read_pipeline/nodes/ScalerNode.cpp:32-40, evaluated by
// libtorch on the same dtypes (int16 tensor).  (ScalerNode.cpp itself needs the whole pipeline to link.)
int ref_med_mad_expr(const int16_t *x, long n, float *med_out, float *mad_out) {
    try {
        auto t = at::from_blob(const_cast<int16_t *>(x), {n}, at::kShort).clone();
        constexpr float factor = 1.4826f;
        auto med = t.median();
        auto mad = at::median(at::abs(t - med)) * factor + 1e-9f;
        *med_out = med.item<float>();
        *mad_out = mad.item<float>();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// utils::quantize_tensor(t, 1) (tensor_utils.cpp:293-300) exactly as LSTMStack.cpp:160-168 calls it: on the f16 copy of
// the [rows][cols] weight matrix.  q = int8 values, scale = the per-row float scales.
int ref_quantize_tensor_f16_rows(const float *w, int rows, int cols, int8_t *q, float *scale) {
    try {
        auto t = at::from_blob(const_cast<float *>(w), {rows, cols}, at::kFloat).clone().to(at::kHalf);
        auto st = utils::quantize_tensor(t, 1);
        auto qt = st.t.contiguous();
        auto sc = st.scale.contiguous();
        std::memcpy(q, qt.data_ptr(), size_t(rows) * size_t(cols));
        std::memcpy(scale, sc.data_ptr(), size_t(rows) * sizeof(float));
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// The input signal of tests/TrimTest.cpp:31-42 ("Test trim signal"): mt19937{42}, N(0,1), +5 on [1,55).
void ref_trimtest_signal(float *out, int n) {
    std::mt19937 gen{42};
    std::normal_distribution<float> rng{0, 1};
    for (int i = 0; i < n; ++i) out[i] = rng(gen);
    for (int i = 1; i < 55 && i < n; ++i) out[i] += 5;
}

}  // extern "C"
