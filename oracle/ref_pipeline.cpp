// oracle/ref_pipeline.cpp — TEST INFRASTRUCTURE ONLY (never linked or loaded by the product).
// The reference's simplex hot path chained END TO END on the CPU, every stage the reference's own source compiled in place
// (oracle/Makefile.ref -> oracle/_ref/libdorado_ref_pipeline.so):
//     raw int16 read -> ScalerNode (read_pipeline/nodes/ScalerNode.cpp, #included here as in ref_scaler.cpp)
//                    -> BasecallerNode (read_pipeline/nodes/BasecallerNode.cpp: chunking, batching, stitching; with the reference's
//                       MessageSink.cpp, chunk.cpp, stitch.cpp, thread_utils.cpp)
//                    -> basecall::ModelRunner (basecall/ModelRunner.cpp: the CPU runner, f32) -> CRFModel / TxModel -> CPUDecoder
//                    -> called read (sequence, qstring, move table) out of a capturing sink.
// This is what north_star's "match the reference's own CPU path on the same input" means for a whole read; the fixture it
// produces (tests/golden/make_golden_pipeline.py -> tests/golden/pipeline_hac.npz) is compared on the GPU box with raw reads ->
// mibc_*_i16 -> this repo's host node (tests/test_gpu_baseline_parity.py).
//
// What is NOT the reference's code in this translation unit, and why (same list as ref_scaler.cpp / basecaller_node_test.cpp):
//   * basecall::load_crf_model: crf_utils.cpp reads the weights from a model directory of .tensor files; the weights here are
//     synthetic and arrive in memory, so the function ModelRunner.cpp calls is supplied below — it builds the reference's
//     CRFModel / TxModel and load_state_dict()s the tensors exactly as crf_utils.cpp:157-185 does, minus the file reads.
//   * is_read_message / get_read_common_data / materialise_read_raw_data (messages.cpp needs htslib + modbase), BamDestructor,
//     utils::mux_change_trim_read (read_utils.cpp -> htslib; acts only on reads that ended with a mux change), utils::trim
//     (torch_utils/trim.cpp:23-60 needs htslib; pinned by the reference's TrimTest answers), config::is_rna_model and
//     config::to_string(ScalingStrategy) / config::is_duplex_model (BasecallModelConfig.cpp needs toml11), the Pipeline friend that connects sinks.
#include "read_pipeline/nodes/ScalerNode.cpp"

#include "basecall/ModelRunner.h"
#include "basecall/crf_utils.h"
#include "basecall/model/CRFModel.h"
#include "basecall/model/TxModel.h"
#include "read_pipeline/nodes/BasecallerNode.h"
#include "ref_common.h"

#include <torch/torch.h>

#include <cstdlib>
#include <cstring>
#include <limits>
#include <mutex>

namespace dorado {

void BamDestructor::operator()(bam1_t *) {}
bool is_read_message(const Message &message) {
    return std::holds_alternative<SimplexReadPtr>(message) || std::holds_alternative<DuplexReadPtr>(message);
}
const ReadCommon &get_read_common_data(const Message &message) {
    if (std::holds_alternative<SimplexReadPtr>(message)) return std::get<SimplexReadPtr>(message)->read_common;
    throw std::invalid_argument("ref_pipeline: not a simplex read");
}
ReadCommon &get_read_common_data(Message &message) {
    return const_cast<ReadCommon &>(get_read_common_data(const_cast<const Message &>(message)));
}
void materialise_read_raw_data(Message &) {}   // duplex reads only

namespace config {
std::string to_string(const ScalingStrategy &s) {
    return s == ScalingStrategy::MED_MAD ? "med_mad" : s == ScalingStrategy::QUANTILE ? "quantile" : "pa";
}
bool is_rna_model(const BasecallModelConfig &c) {
    return c.sample_type == models::SampleType::RNA002 || c.sample_type == models::SampleType::RNA004;
}
bool is_duplex_model(const BasecallModelConfig &c) { return c.num_features > 1; }   // BasecallModelConfig.cpp:364-366
}  // namespace config

namespace utils {
void mux_change_trim_read(ReadCommon &) {}
int trim(const at::Tensor &signal, float threshold, int window_size, int min_elements) {
    // torch_utils/trim.cpp:23-60
    const int min_trim = 10;
    const auto sig = signal.to(at::kFloat).contiguous();
    const float *p = sig.data_ptr<float>();
    const int num_samples = int(sig.size(0)) - min_trim;
    const int num_windows = num_samples / window_size;
    bool seen_peak = false;
    for (int pos = 0; pos < num_windows; ++pos) {
        const int start = pos * window_size + min_trim, end = start + window_size;
        int cnt = 0;
        for (int i = start; i < end; ++i) cnt += p[i] > threshold;
        if (cnt > min_elements || seen_peak) {
            seen_peak = true;
            if (p[end - 1] > threshold) continue;
            return end >= num_samples ? min_trim : end;
        }
    }
    return min_trim;
}
}  // namespace utils

// ---- the weights ModelRunner's constructor asks load_crf_model for (set by the driver call below, one call at a time)
namespace {
std::mutex g_weights_mutex;
std::vector<at::Tensor> g_weights;
}  // namespace

namespace basecall {
// crf_utils.cpp:157-185,204 with the state dict taken from memory instead of <model dir>/*.tensor
torch::nn::ModuleHolder<torch::nn::AnyModule> load_crf_model(const config::BasecallModelConfig &model_config,
                                                             const torch::TensorOptions &options) {
    using namespace torch::nn;
    std::vector<at::Tensor> state;
    {
        std::lock_guard<std::mutex> lk(g_weights_mutex);
        for (const auto &t : g_weights) state.push_back(t.clone());
    }
    if (model_config.is_tx_model()) {
        auto model = model::TxModel(model_config, options);
        const auto params = model->parameters();
        if (params.size() != state.size()) throw std::runtime_error("ref_pipeline: weight count mismatch");
        for (size_t i = 0; i < state.size(); ++i) state[i] = state[i].reshape(params[i].sizes());
        model->load_state_dict(state);
        model->to(options.dtype().toScalarType());
        model->to(options.device());
        model->eval();
        return ModuleHolder<AnyModule>(AnyModule(model));
    }
    auto model = model::CRFModel(model_config);
    const auto params = model->parameters();
    if (params.size() != state.size()) throw std::runtime_error("ref_pipeline: weight count mismatch");
    for (size_t i = 0; i < state.size(); ++i) state[i] = state[i].reshape(params[i].sizes());
    model->load_state_dict(state);
    model->to(options.dtype().toScalarType());
    model->to(options.device());
    model->eval();
    return ModuleHolder<AnyModule>(AnyModule(model));
}
}  // namespace basecall

class Pipeline {
public:
    static void connect(MessageSink &from, MessageSink &to) { from.add_sink(to); }
};

namespace {
struct ForcedBatchParams : config::BatchParams {   // set_value() lives in BatchParams.cpp (needs toml11): write the fields
    ForcedBatchParams(int chunk, int overlap, int batch) {
        m_chunk_size = {chunk, Priority::FORCE};
        m_overlap = {overlap, Priority::FORCE};
        m_batch_size = {batch, Priority::FORCE};
    }
};

class CaptureSink final : public MessageSink {
public:
    CaptureSink() : MessageSink(4096, 1) {}
    ~CaptureSink() override { stop_input_processing(utils::AsyncQueueTerminateFast::Yes); }
    std::string get_name() const override { return "ref_pipeline_capture"; }
    void terminate(const TerminateOptions &o) override { stop_input_processing(o.fast); }
    void restart() override {
        start_input_processing(
                [this] {
                    Message m;
                    while (get_input_message(m))
                        if (std::holds_alternative<SimplexReadPtr>(m)) got.push_back(std::get<SimplexReadPtr>(std::move(m)));
                },
                "ref_capture");
    }
    std::vector<SimplexReadPtr> got;
};
}  // namespace
}  // namespace dorado

static std::string g_pipeline_err;

extern "C" {

const char *ref_pipeline_last_error(void) { return g_pipeline_err.c_str(); }

// n_reads raw int16 reads back to back (read_len[n_reads]) -> ScalerNode -> BasecallerNode over `num_runners` CPU ModelRunners
// (f32, batch `batch_size`, one torch intra-op thread per runner as torch_utils.cpp:20 sets it) -> called reads.
// strategy / params7 / sample_type_rna004 / cal3 (per read: scaling, offset, open_pore_level) / flow_cell: as ref_scaler_node.
// Outputs per read (row pitch `pitch`): seq / qstr NUL padded, moves; seq_len / moves_len; f2 = {scale, shift} (pA);
// i2 = {num_trimmed_samples, rna_adapter_end_signal_pos}; scaled_len = samples that reached the basecaller.
int ref_pipeline_run(const RefModelDesc *d, const float *const *weights, const int64_t *wnumel, int n_weights, float qscale,
                     float qbias, int chunk_size, int overlap, int batch_size, int num_runners, int strategy, const float *params7,
                     int sample_type_rna004, const int16_t *raw, const int64_t *read_len, int n_reads, const float *cal3,
                     const char *flow_cell_product_code, int pitch, char *seq_out, char *qstr_out, uint8_t *moves_out,
                     int64_t *seq_len, int64_t *moves_len, float *f2, int *i2, int64_t *scaled_len) {
    try {
        using namespace dorado;
        torch::set_num_threads(1);   // torch_utils/torch_utils.cpp:20
        auto cfg = ref_make_config(*d);
        if (cfg.is_tx_model()) cfg.stride /= d->up_scale_factor;   // BasecallModelConfig.cpp:447-454
        cfg.qscale = qscale;
        cfg.qbias = qbias;
        cfg.sample_type = sample_type_rna004 ? models::SampleType::RNA004 : models::SampleType::DNA;
        cfg.basecaller = ForcedBatchParams(chunk_size, overlap, batch_size);
        cfg.signal_norm_params.strategy = static_cast<config::ScalingStrategy>(strategy);
        cfg.signal_norm_params.quantile = {params7[0], params7[1], params7[2], params7[3]};
        cfg.signal_norm_params.standardisation = {params7[4] != 0.0f, params7[5], params7[6]};
        {
            std::lock_guard<std::mutex> lk(g_weights_mutex);
            g_weights.clear();
            for (int i = 0; i < n_weights; ++i)
                g_weights.push_back(at::from_blob(const_cast<float *>(weights[i]), {wnumel[i]}, at::kFloat).clone());
        }
        std::vector<basecall::RunnerPtr> runners;
        for (int r = 0; r < num_runners; ++r) runners.push_back(std::make_unique<basecall::ModelRunner>(cfg, "cpu"));

        ScalerNode scaler(cfg.signal_norm_params, cfg.sample_type, 1, 4096);
        BasecallerNode node(std::move(runners), size_t(overlap), "ref_cpu_model", 4096, "BasecallerNode", 0);
        CaptureSink sink;
        Pipeline::connect(scaler, node);
        Pipeline::connect(node, sink);
        sink.restart();
        node.restart();
        scaler.restart();
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            auto read = std::make_unique<SimplexRead>();
            read->read_common.raw_data =
                    at::from_blob(const_cast<int16_t *>(raw + pos), {read_len[r]}, at::TensorOptions().dtype(at::kShort)).clone();
            read->read_common.read_id = "read_" + std::to_string(r);
            read->read_common.flow_cell_product_code = flow_cell_product_code ? flow_cell_product_code : "";
            read->scaling = cal3[3 * r + 0];
            read->offset = cal3[3 * r + 1];
            read->open_pore_level = cal3[3 * r + 2];
            pos += size_t(read_len[r]);
            scaler.push_message(std::move(read));
        }
        scaler.terminate(TerminateOptions{});
        node.terminate(TerminateOptions{});
        sink.terminate(TerminateOptions{});
        if (int(sink.got.size()) != n_reads)
            throw std::runtime_error("ref_pipeline: " + std::to_string(sink.got.size()) + " reads came out");
        std::memset(seq_out, 0, size_t(n_reads) * size_t(pitch));
        std::memset(qstr_out, 0, size_t(n_reads) * size_t(pitch));
        std::memset(moves_out, 0, size_t(n_reads) * size_t(pitch));
        for (auto &rd : sink.got) {
            const auto &rc = rd->read_common;
            const int r = std::stoi(rc.read_id.substr(5));
            if (int(rc.seq.size()) > pitch || int(rc.moves.size()) > pitch) throw std::runtime_error("output pitch too small");
            std::memcpy(seq_out + size_t(r) * size_t(pitch), rc.seq.data(), rc.seq.size());
            std::memcpy(qstr_out + size_t(r) * size_t(pitch), rc.qstring.data(), rc.qstring.size());
            std::memcpy(moves_out + size_t(r) * size_t(pitch), rc.moves.data(), rc.moves.size());
            seq_len[r] = int64_t(rc.seq.size());
            moves_len[r] = int64_t(rc.moves.size());
            f2[2 * r + 0] = rc.scale;
            f2[2 * r + 1] = rc.shift;
            i2[2 * r + 0] = int(rc.num_trimmed_samples);
            i2[2 * r + 1] = int(rc.rna_adapter_end_signal_pos);
            scaled_len[r] = int64_t(rc.get_raw_data_samples());
        }
        return 0;
    } catch (const std::exception &e) {
        g_pipeline_err = e.what();
        return -1;
    }
}

}  // extern "C"
