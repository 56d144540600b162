// integration/HipModelRunnerAdapter.h — the reference-side binding of the MI355X engine.
//
// This is the one file a Dorado maintainer adds next to dorado/basecall/CudaModelRunner.h: it derives from the
// reference's own dorado::basecall::ModelRunnerBase (basecall/include/basecall/ModelRunnerBase.h:20-38) and
// forwards to dorado_amd::host::HipModelRunner (dorado_amd/host/mibc_host.h -> libmibc_host.so -> libmibc.so).
// It is COMPILED against the reference's real headers by oracle/Makefile.ref (target adapter) and exercised
// through ModelRunnerBase::accept_chunk(at::Tensor) / call_chunks by tests/test_adapter.py — not documentation.
//
//   accept_chunk(at::Tensor)  the only libtorch type that crosses the boundary: the [1, T_in] f16 slice
//                             BasecallerNode hands over (BasecallerNode.cpp:397-444) -> raw pointer + length
//   create_hip_basecall_runners  the branch api::create_basecall_runners gains beside its "cuda" branch
//                             (api/runner_creation.cpp:85-124): ONE caller per device, same
//                             [devices][runners][batch dims] order, same {chunk, 0.5 x chunk} batch dimensions for
//                             PipelineType::simplex; every field of BasecallerCreationParams is honoured
//                             (memory_limit_fraction, batch_size_time_penalty, run_/emit_batchsize_benchmarks,
//                             variable_chunk_sizes)
//   variable chunk sizes      accept_chunk packs variable-length slices behind each other exactly as
//                             CudaModelRunner::accept_chunk does (CudaModelRunner.cpp:21-32); variable_chunk_sizes()
//                             reports what the caller was created with
#pragma once
#include "basecall/ModelRunnerBase.h"
#include "config/BasecallModelConfig.h"
#include "mibc_host.h"   // this repo: dorado_amd/host

#include <ATen/ATen.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <cstdlib>
#include <string>
#include <vector>

namespace dorado::basecall {

// The LSTM arithmetic the reference's own GPU build runs this model in (round 6: the default here too — the int8 path has a stated
// identity bound): nn/ConvStack.cpp:60-89 get_koi_lstm_input_layout — the convolution in front of the LSTM stack writes
// the CUTLASS_TNC_I8 layout (=> every LSTM layer int8, nn/LSTMStack.cpp:127-211) when it ends in tanh and 128 < lstm_size <= 1024,
// lstm_size % 128 == 0; the reference's override DORADO_LSTM_MODE is honoured the same way (CUTLASS_TNC_F16 / CUBLAS_TN2C: f16;
// CUTLASS_TNC_I8 on a model the rule would run in f16: int8 from the second layer on).
inline bool mibc_reference_lstm_int8(const config::BasecallModelConfig &c) {
    if (c.tx.has_value() || c.convs.size() < 3 || c.lstm_layers < 2) return false;
    const int C = c.lstm_size;
    const bool cutlass_shape = C <= 1024 && C > 128 && (C % 128) == 0;
    bool int8 = cutlass_shape && c.convs.back().activation == config::Activation::TANH;
    if (const char *env = std::getenv("DORADO_LSTM_MODE")) {
        const std::string m(env);
        if (m == "CUBLAS_TN2C" || m == "CUTLASS_TNC_F16") int8 = false;
        else if (m == "CUTLASS_TNC_I8" && cutlass_shape) int8 = true;
    }
    return int8;
}

// config::BasecallModelConfig -> the plain C descriptor of include/mibc.h
inline mibc_model_desc mibc_desc_from_config(const config::BasecallModelConfig &c) {
    mibc_model_desc d{};
    d.n_convs = int(c.convs.size());
    if (d.n_convs > 8) throw std::runtime_error("mibc: more than 8 convolution layers");
    for (int i = 0; i < d.n_convs; ++i) {
        d.conv_insize[i] = c.convs[size_t(i)].insize;
        d.conv_size[i] = c.convs[size_t(i)].size;
        d.conv_winlen[i] = c.convs[size_t(i)].winlen;
        d.conv_stride[i] = c.convs[size_t(i)].stride;
        d.conv_act[i] = int(c.convs[size_t(i)].activation);   // MIBC_ACT_* == config::Activation order
    }
    d.lstm_size = c.lstm_size;
    d.lstm_layers = c.lstm_layers;
    d.state_len = c.state_len;
    d.outsize = c.outsize;
    d.bias = c.bias ? 1 : 0;
    d.clamp = c.clamp ? 1 : 0;
    d.scale = c.scale;
    d.out_features = c.out_features.value_or(-1);
    d.num_features = c.num_features;
    if (c.tx.has_value()) {
        const auto &t = *c.tx;
        d.tx_d_model = t.tx.d_model;
        d.tx_nhead = t.tx.nhead;
        d.tx_depth = t.tx.depth;
        d.tx_dim_ff = t.tx.dim_feedforward;
        d.tx_win_upper = t.tx.attn_window.first;
        d.tx_win_lower = t.tx.attn_window.second;
        d.tx_max_seq_len = t.tx.max_seq_len;
        d.tx_deepnorm_alpha = t.tx.deepnorm_alpha;
        d.tx_theta = t.tx.theta;
        d.up_size = t.upsample.size;
        d.up_scale_factor = t.upsample.scale_factor;
        d.crf_scale = t.crf.scale;
        d.crf_blank_score = t.crf.blank_score;
        d.crf_expand_blanks = t.crf.expand_blanks ? 1 : 0;
        d.state_len = t.crf.state_len;
        d.outsize = t.crf.outsize();
    }
    d.lstm_quant = mibc_reference_lstm_int8(c) ? 1 : 0;
    return d;
}

class HipModelRunnerAdapter final : public ModelRunnerBase {
public:
    // cfg: the caller's OWN copy of the model configuration, shared by its runners (CudaCaller keeps m_config by value,
    // CudaCaller.h; CudaModelRunner::config() returns that copy) — never a reference into the creation parameters
    HipModelRunnerAdapter(std::unique_ptr<dorado_amd::host::ModelRunnerBase> impl,
                          std::shared_ptr<const config::BasecallModelConfig> cfg, bool low_latency)
            : m_impl(std::move(impl)), m_cfg(std::move(cfg)), m_low_latency(low_latency) {}

    void accept_chunk(int chunk_idx, const at::Tensor &chunk) override {
        // [1, T_in] (or [T_in]) f16 view of the read's signal, possibly shorter than chunk_size for variable chunks
        const at::Tensor c = chunk.to(at::kHalf).contiguous();
        m_impl->accept_chunk(chunk_idx, reinterpret_cast<const uint16_t *>(c.data_ptr()), size_t(c.numel()));
    }
    std::vector<decode::DecodedChunk> call_chunks(int num_chunks) override {
        auto r = m_impl->call_chunks(num_chunks);
        std::vector<decode::DecodedChunk> out(r.size());
        for (size_t i = 0; i < r.size(); ++i)
            out[i] = {std::move(r[i].sequence), std::move(r[i].qstring), std::move(r[i].moves)};
        return out;
    }
    const config::BasecallModelConfig &config() const override { return *m_cfg; }
    size_t chunk_size() const override { return m_impl->chunk_size(); }
    size_t batch_size() const override { return m_impl->batch_size(); }
    // ModelRunnerBase.h:29: true when the caller was created with BasecallerCreationParams::variable_chunk_sizes and
    // the model supports it; the node then cuts reads with generate_variable_chunks and accept_chunk packs the slices
    bool variable_chunk_sizes() const override { return m_impl->variable_chunk_sizes(); }
    std::pair<int, int> batch_timeouts_ms() const override {
        // CudaCaller.cpp:126-138: low latency 350 ms / 350 ms, else 300 s first chunk / 30 s last chunk
        return m_low_latency ? std::pair<int, int>{350, 350} : m_impl->batch_timeouts_ms();
    }
    bool is_low_latency() const override { return m_low_latency; }   // ModelRunnerBase.h:34
    void terminate() override { m_impl->terminate(); }
    void restart() override { m_impl->restart(); }
    std::string get_name() const override { return m_impl->get_name(); }
    stats::NamedStats sample_stats() const override {
        stats::NamedStats s;
        for (const auto &kv : m_impl->sample_stats()) s[kv.first] = kv.second;
        return s;
    }

private:
    std::unique_ptr<dorado_amd::host::ModelRunnerBase> m_impl;
    std::shared_ptr<const config::BasecallModelConfig> m_cfg;
    bool m_low_latency;
};

// api::check_variable_chunk_sizes_supported (api/runner_creation.cpp:24-44) for this engine: the same model range (LSTM
// models, 128 < lstm_size <= 1024, multiples of 128); the device condition (koi_can_use_cutlass) has no counterpart.
inline bool hip_variable_chunk_sizes_supported(const config::BasecallModelConfig &c) {
    return c.is_lstm_model() && !c.is_flstm_model() && c.lstm_size > 128 && c.lstm_size <= 1024 && c.lstm_size % 128 == 0;
}

// The "hip:" branch of api::create_basecall_runners (api/runner_creation.cpp:85-124).  weights: host f32 tensors in
// module.parameters() order (what basecall::load_crf_model_weights returns, crf_utils.cpp:26-150).
// Returns the runners in [devices][runners][chunk_sizes] order and the number of devices.
inline std::pair<std::vector<RunnerPtr>, size_t> create_hip_basecall_runners(const BasecallerCreationParams &params,
                                                                            const std::vector<at::Tensor> &weights,
                                                                            size_t num_gpu_runners) {
    auto cfg_owned = std::make_shared<const config::BasecallModelConfig>(params.model_config);
    const config::BasecallModelConfig &cfg = *cfg_owned;
    const mibc_model_desc desc = mibc_desc_from_config(cfg);
    std::vector<at::Tensor> keep;
    std::vector<const float *> wp;
    for (const auto &w : weights) {
        keep.push_back(w.to(at::kCPU).to(at::kFloat).contiguous());
        wp.push_back(keep.back().data_ptr<float>());
    }
    const mibc_decode_opts opts{32, 100.0f, 2.0f, cfg.qbias, cfg.qscale};   // DecodedChunk.h:15-23, ModelRunner.cpp:17-18
    const int chunk = int(cfg.basecaller.chunk_size()), overlap = int(cfg.basecaller.overlap());
    // batch dimensions: {chunk, 0.5 x chunk} for high-throughput simplex, one size otherwise (CudaCaller.cpp:388-413)
    const std::vector<int> sizes = (params.pipeline_type == PipelineType::simplex)
                                           ? dorado_amd::host::simplex_chunk_sizes(desc, chunk, overlap)
                                           : std::vector<int>{chunk};
    dorado_amd::host::CallerParams cp;
    cp.memory_limit_fraction = params.memory_limit_fraction;       // CudaCaller.cpp:434-439
    cp.batch_size_time_penalty = params.batch_size_time_penalty;   // :603-627
    cp.run_batchsize_benchmarks = params.run_batchsize_benchmarks;
    cp.emit_batchsize_benchmarks = params.emit_batchsize_benchmarks;
    cp.variable_chunk_sizes = params.variable_chunk_sizes && hip_variable_chunk_sizes_supported(cfg);
    auto per_device = dorado_amd::host::create_basecall_runners(desc, wp.data(), int(wp.size()), params.device,
                                                                int(num_gpu_runners), sizes,
                                                                int(cfg.basecaller.batch_size()), opts, cp);
    std::vector<RunnerPtr> runners;
    const bool low_latency = params.pipeline_type == PipelineType::simplex_low_latency;
    for (auto &dev : per_device)
        for (auto &r : dev) runners.push_back(std::make_unique<HipModelRunnerAdapter>(std::move(r), cfg_owned, low_latency));
    return {std::move(runners), per_device.size()};
}

}  // namespace dorado::basecall
