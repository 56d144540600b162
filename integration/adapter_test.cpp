// integration/adapter_test.cpp — TEST DRIVER for integration/HipModelRunnerAdapter.h, compiled against the reference's
// REAL headers (oracle/Makefile.ref, target adapter): builds a config::BasecallModelConfig, creates the runners through
// create_hip_basecall_runners and drives them exclusively through the reference's own interface,
// dorado::basecall::ModelRunnerBase (accept_chunk(int, const at::Tensor&), call_chunks, chunk_size, ...).
#include "HipModelRunnerAdapter.h"

#include <cstring>
#include <string>

using namespace dorado;

// ModelRunnerBase's key function (its vtable / typeinfo are emitted with it) lives in basecall/ModelRunnerBase.cpp, which
// pulls in BasecallModelConfig.cpp -> toml11 (absent offline).  The test driver supplies the simplex value of that
// one-liner (ModelRunnerBase.cpp:6-9: duplex 5000 / 5000, else 100 / 100); in the reference tree the real one links.
namespace dorado::basecall {
std::pair<int, int> ModelRunnerBase::batch_timeouts_ms() const { return std::make_pair(100, 100); }
}  // namespace dorado::basecall

namespace {
struct ForcedBatchParams : config::BatchParams {   // set_value() lives in BatchParams.cpp (needs toml11): write the fields
    ForcedBatchParams(int chunk, int overlap, int batch) {
        m_chunk_size = {chunk, Priority::FORCE};
        m_overlap = {overlap, Priority::FORCE};
        m_batch_size = {batch, Priority::FORCE};
    }
};
thread_local std::string g_err;

// the reference's configuration object built FROM the C descriptor (the adapter converts it back: a round trip through
// config::BasecallModelConfig)
static config::BasecallModelConfig make_cfg(const mibc_model_desc *md, float qscale, float qbias, int chunk_size, int overlap,
                                            int batch_size) {
    config::BasecallModelConfig cfg;
    for (int i = 0; i < md->n_convs; ++i) {
        config::ConvParams p;
        p.insize = md->conv_insize[i];
        p.size = md->conv_size[i];
        p.winlen = md->conv_winlen[i];
        p.stride = md->conv_stride[i];
        p.activation = static_cast<config::Activation>(md->conv_act[i]);
        cfg.convs.push_back(p);
    }
    cfg.lstm_size = md->lstm_size;
    cfg.lstm_layers = md->lstm_layers;
    cfg.state_len = md->state_len;
    cfg.outsize = md->outsize;
    cfg.bias = md->bias != 0;
    cfg.clamp = md->clamp != 0;
    cfg.scale = md->scale;
    cfg.blank_score = 2.0f;
    cfg.num_features = md->num_features;
    cfg.qscale = qscale;
    cfg.qbias = qbias;
    if (md->out_features > 0) cfg.out_features = md->out_features;
    cfg.stride = 1;
    for (int i = 0; i < md->n_convs; ++i) cfg.stride *= md->conv_stride[i];
    if (md->tx_d_model > 0) {
        config::TxStack tx;
        tx.tx.d_model = md->tx_d_model;
        tx.tx.nhead = md->tx_nhead;
        tx.tx.depth = md->tx_depth;
        tx.tx.dim_feedforward = md->tx_dim_ff;
        tx.tx.attn_window = {md->tx_win_upper, md->tx_win_lower};
        tx.tx.deepnorm_alpha = md->tx_deepnorm_alpha;
        tx.tx.theta = md->tx_theta;
        tx.tx.max_seq_len = md->tx_max_seq_len;
        tx.upsample.size = md->up_size;
        tx.upsample.scale_factor = md->up_scale_factor;
        tx.crf.insize = md->up_size;
        tx.crf.n_base = 4;
        tx.crf.state_len = md->state_len;
        tx.crf.scale = md->crf_scale;
        tx.crf.blank_score = md->crf_blank_score;
        tx.crf.expand_blanks = md->crf_expand_blanks != 0;
        cfg.tx = tx;
        cfg.stride /= md->up_scale_factor;   // BasecallModelConfig.cpp:447-454
    }
    cfg.basecaller = ForcedBatchParams(chunk_size, overlap, batch_size);

    return cfg;
}

static std::vector<at::Tensor> make_weights(const float *const *weights, const int64_t *wnumel, int n_weights) {
    std::vector<at::Tensor> ws;
    for (int i = 0; i < n_weights; ++i)
        ws.push_back(at::from_blob(const_cast<float *>(weights[i]), {wnumel[i]}, at::kFloat).clone());
    return ws;
}
}  // namespace

// shared with basecaller_node_test.cpp
config::BasecallModelConfig adapter_test_make_cfg(const mibc_model_desc *md, float qscale, float qbias, int chunk_size, int overlap,
                                                  int batch_size) {
    return make_cfg(md, qscale, qbias, chunk_size, overlap, batch_size);
}
std::vector<at::Tensor> adapter_test_make_weights(const float *const *weights, const int64_t *wnumel, int n_weights) {
    return make_weights(weights, wnumel, n_weights);
}
void adapter_test_set_error(const std::string &e) { g_err = e; }

extern "C" {

const char *adapter_last_error() { return g_err.c_str(); }

// desc: include/mibc.h descriptor of the model (the test builds the reference config FROM it and the adapter converts it
// back — a round trip through config::BasecallModelConfig).  chunks: [n_chunks][chunk_size] f16 for the FIRST chunk size.
// Outputs for those chunks: seq/qstr [n_chunks][T] (NUL padded), moves [n_chunks][T].
// info: [0] number of runners, [1] number of devices, [2..5] chunk sizes in runner order (first device, first runner),
//       [6] variable_chunk_sizes(), [7] is_low_latency(), [8] batch size of runner 0, [9] batch timeouts first, [10] last.
int adapter_run(const mibc_model_desc *md, const float *const *weights, const int64_t *wnumel, int n_weights,
                const char *device, int pipeline_type, int num_runners, int chunk_size, int overlap, int batch_size,
                float qscale, float qbias, const uint16_t *chunks, int n_chunks, int T, char *seq_out, char *qstr_out,
                uint8_t *moves_out, int *info, char *name_out, int name_cap) {
    try {
        // a TEMPORARY configuration: the runners must own their copy (config() is read after it is gone)
        auto cfg_tmp = std::make_unique<config::BasecallModelConfig>(make_cfg(md, qscale, qbias, chunk_size, overlap, batch_size));
        std::vector<at::Tensor> ws = make_weights(weights, wnumel, n_weights);

        const std::string dev(device);
        const basecall::BasecallerCreationParams params{*cfg_tmp, dev, 1.0f, static_cast<basecall::PipelineType>(pipeline_type),
                                                        0.0f, false, false, false};
        auto [runners, num_devices] = basecall::create_hip_basecall_runners(params, ws, size_t(num_runners));
        cfg_tmp.reset();
        info[0] = int(runners.size());
        info[1] = int(num_devices);
        for (int i = 0; i < 4; ++i) info[2 + i] = i < int(runners.size()) ? int(runners[size_t(i)]->chunk_size()) : 0;
        basecall::ModelRunnerBase &r0 = *runners.at(0);        // the REFERENCE's interface from here on
        info[6] = r0.variable_chunk_sizes() ? 1 : 0;
        info[7] = r0.is_low_latency() ? 1 : 0;
        info[8] = int(r0.batch_size());
        info[9] = r0.batch_timeouts_ms().first;
        info[10] = r0.batch_timeouts_ms().second;
        info[11] = (r0.config().lstm_size == md->lstm_size && int(r0.config().convs.size()) == md->n_convs &&
                    r0.config().qscale == qscale) ? 1 : 0;   // config() after the creation parameters died
        std::strncpy(name_out, r0.get_name().c_str(), size_t(name_cap) - 1);
        if (int(r0.chunk_size()) != chunk_size) throw std::runtime_error("runner 0 does not have the requested chunk size");
        if (n_chunks > int(r0.batch_size())) throw std::runtime_error("more chunks than the batch holds");
        for (int i = 0; i < n_chunks; ++i) {
            // what BasecallerNode hands over: a [1, T_in] f16 slice of the read tensor
            at::Tensor t = at::from_blob(const_cast<uint16_t *>(chunks + size_t(i) * size_t(chunk_size)), {1, chunk_size},
                                         at::kHalf);
            r0.accept_chunk(i, t);
        }
        std::vector<basecall::decode::DecodedChunk> dec = r0.call_chunks(n_chunks);
        if (int(dec.size()) != n_chunks) throw std::runtime_error("call_chunks returned the wrong number of chunks");
        std::memset(seq_out, 0, size_t(n_chunks) * size_t(T));
        std::memset(qstr_out, 0, size_t(n_chunks) * size_t(T));
        for (int i = 0; i < n_chunks; ++i) {
            if (int(dec[size_t(i)].moves.size()) != T) throw std::runtime_error("unexpected move table length");
            std::memcpy(seq_out + size_t(i) * size_t(T), dec[size_t(i)].sequence.data(), dec[size_t(i)].sequence.size());
            std::memcpy(qstr_out + size_t(i) * size_t(T), dec[size_t(i)].qstring.data(), dec[size_t(i)].qstring.size());
            std::memcpy(moves_out + size_t(i) * size_t(T), dec[size_t(i)].moves.data(), size_t(T));
        }
        const auto stats = r0.sample_stats();
        if (stats.find("batches_called") == stats.end()) throw std::runtime_error("sample_stats lacks batches_called");
        for (auto &r : runners) r->terminate();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Variable chunk sizes through the reference interface (ModelRunnerBase::variable_chunk_sizes() == true): the caller is
// created with BasecallerCreationParams::variable_chunk_sizes, every chunk — any stride multiple up to chunk_size — is
// handed over with accept_chunk exactly as BasecallerNode does for variable chunks (BasecallerNode.cpp:397-430; the chunk
// index is ignored, CudaModelRunner.cpp:21-32) and all of them come back from ONE call_chunks, in order.
// samples: the chunks back to back (f16), lens[n_chunks].  Outputs with row pitch T; moves_len_out[i] = steps of chunk i.
// info: [0] variable_chunk_sizes(), [1] batch size, [2] chunk size.
int adapter_run_variable(const mibc_model_desc *md, const float *const *weights, const int64_t *wnumel, int n_weights,
                         const char *device, int chunk_size, int overlap, int batch_size, float qscale, float qbias,
                         const uint16_t *samples, const int64_t *lens, int n_chunks, int T, char *seq_out, char *qstr_out,
                         uint8_t *moves_out, int64_t *moves_len_out, int *info) {
    try {
        const config::BasecallModelConfig cfg = make_cfg(md, qscale, qbias, chunk_size, overlap, batch_size);
        std::vector<at::Tensor> ws = make_weights(weights, wnumel, n_weights);
        const std::string dev(device);
        const basecall::BasecallerCreationParams params{cfg, dev, 1.0f, basecall::PipelineType::simplex_low_latency,
                                                        0.0f, false, false, /*variable_chunk_sizes*/ true};
        auto [runners, num_devices] = basecall::create_hip_basecall_runners(params, ws, 1);
        basecall::ModelRunnerBase &r0 = *runners.at(0);
        info[0] = r0.variable_chunk_sizes() ? 1 : 0;
        info[1] = int(r0.batch_size());
        info[2] = int(r0.chunk_size());
        size_t pos = 0;
        for (int i = 0; i < n_chunks; ++i) {
            at::Tensor t = at::from_blob(const_cast<uint16_t *>(samples + pos), {1, lens[i]}, at::kHalf);
            r0.accept_chunk(0, t);
            pos += size_t(lens[i]);
        }
        std::vector<basecall::decode::DecodedChunk> dec = r0.call_chunks(n_chunks);
        if (int(dec.size()) != n_chunks) throw std::runtime_error("call_chunks returned the wrong number of chunks");
        std::memset(seq_out, 0, size_t(n_chunks) * size_t(T));
        std::memset(qstr_out, 0, size_t(n_chunks) * size_t(T));
        std::memset(moves_out, 0, size_t(n_chunks) * size_t(T));
        for (int i = 0; i < n_chunks; ++i) {
            const auto &d = dec[size_t(i)];
            if (int(d.moves.size()) > T) throw std::runtime_error("move table longer than the chunk size allows");
            moves_len_out[i] = int64_t(d.moves.size());
            std::memcpy(seq_out + size_t(i) * size_t(T), d.sequence.data(), d.sequence.size());
            std::memcpy(qstr_out + size_t(i) * size_t(T), d.qstring.data(), d.qstring.size());
            std::memcpy(moves_out + size_t(i) * size_t(T), d.moves.data(), d.moves.size());
        }
        for (auto &r : runners) r->terminate();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}

// Automatic batch size under the creation parameters (batch_size 0 in the configuration): memory_limit_fraction caps it
// (CudaCaller.cpp:434-439), run_batchsize_benchmarks selects the timing sweep, batch_size_time_penalty its tolerance.
int adapter_auto_batch(const mibc_model_desc *md, const float *const *weights, const int64_t *wnumel, int n_weights,
                       const char *device, int chunk_size, int overlap, float memory_limit_fraction, float time_penalty,
                       int run_benchmarks, int *batch_out) {
    try {
        const config::BasecallModelConfig cfg = make_cfg(md, 1.0f, 0.0f, chunk_size, overlap, 0);
        std::vector<at::Tensor> ws = make_weights(weights, wnumel, n_weights);
        const std::string dev(device);
        const basecall::BasecallerCreationParams params{cfg, dev, memory_limit_fraction,
                                                        basecall::PipelineType::simplex_low_latency, time_penalty,
                                                        run_benchmarks != 0, false, false};
        auto [runners, num_devices] = basecall::create_hip_basecall_runners(params, ws, 1);
        *batch_out = int(runners.at(0)->batch_size());
        for (auto &r : runners) r->terminate();
        return 0;
    } catch (const std::exception &e) {
        g_err = e.what();
        return -1;
    }
}
}
