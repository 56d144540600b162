// integration/basecaller_node_test.cpp — TEST DRIVER (built by oracle/Makefile.ref into oracle/_ref/libmibc_adapter_test.so).
// The drop-in claim end to end: the reference's OWN BasecallerNode (read_pipeline/nodes/BasecallerNode.cpp compiled in place,
// with the reference's MessageSink.cpp, chunk.cpp, stitch.cpp, thread_utils.cpp) owns the chunking, batching, time-outs,
// stitching and read bookkeeping; its runners are integration/HipModelRunnerAdapter.h — i.e. the reference drives this engine
// through basecall::ModelRunnerBase exactly as `dorado basecaller` would.  Reads go in through push_message, called reads come
// out of a capturing sink.
//
// What this file supplies because the reference's own definition cannot be built offline (messages.cpp needs htslib + modbase,
// read_utils.cpp needs trim.cpp -> htslib, BasecallModelConfig.cpp needs toml11):
//   is_read_message / get_read_common_data / materialise_read_raw_data (messages.cpp:16-19, 425-450; simplex reads only here),
//   utils::mux_change_trim_read (read_utils.cpp: acts only on reads that ended with a mux change — none here), BamDestructor,
//   config::is_rna_model (BasecallModelConfig.cpp: sample_type is RNA002 / RNA004), and the Pipeline friend that connects sinks.
#include "HipModelRunnerAdapter.h"
#include "read_pipeline/base/MessageSink.h"
#include "read_pipeline/nodes/BasecallerNode.h"

#include <cstring>
#include <string>

namespace dorado {

void BamDestructor::operator()(bam1_t *) {}
bool is_read_message(const Message &message) {
    return std::holds_alternative<SimplexReadPtr>(message) || std::holds_alternative<DuplexReadPtr>(message);
}
const ReadCommon &get_read_common_data(const Message &message) {
    if (std::holds_alternative<SimplexReadPtr>(message)) return std::get<SimplexReadPtr>(message)->read_common;
    throw std::invalid_argument("basecaller_node_test: not a simplex read");
}
ReadCommon &get_read_common_data(Message &message) {
    return const_cast<ReadCommon &>(get_read_common_data(const_cast<const Message &>(message)));
}
void materialise_read_raw_data(Message &) {}   // duplex reads only
namespace utils {
void mux_change_trim_read(ReadCommon &) {}
}  // namespace utils
namespace config {
bool is_rna_model(const BasecallModelConfig &c) {
    return c.sample_type == models::SampleType::RNA002 || c.sample_type == models::SampleType::RNA004;
}
}  // namespace config

class Pipeline {   // also used by node_cpu_test.cpp
public:
    static void connect(MessageSink &from, MessageSink &to);
};
void Pipeline::connect(MessageSink &from, MessageSink &to) { from.add_sink(to); }

namespace {
class CaptureSink final : public MessageSink {
public:
    CaptureSink() : MessageSink(4096, 1) {}
    ~CaptureSink() override { stop_input_processing(utils::AsyncQueueTerminateFast::Yes); }
    std::string get_name() const override { return "capture"; }
    void terminate(const TerminateOptions &o) override { stop_input_processing(o.fast); }
    void restart() override {
        start_input_processing(
                [this] {
                    Message m;
                    while (get_input_message(m))
                        if (std::holds_alternative<SimplexReadPtr>(m)) got.push_back(std::get<SimplexReadPtr>(std::move(m)));
                },
                "capture");
    }
    std::vector<SimplexReadPtr> got;
};
}  // namespace
}  // namespace dorado

using namespace dorado;

// adapter_test.cpp
config::BasecallModelConfig adapter_test_make_cfg(const mibc_model_desc *md, float qscale, float qbias, int chunk_size, int overlap,
                                                  int batch_size);
std::vector<at::Tensor> adapter_test_make_weights(const float *const *weights, const int64_t *wnumel, int n_weights);
void adapter_test_set_error(const std::string &e);

extern "C" {

// reads: n_reads scaled f16 reads back to back, read_len[n_reads].  variable: BasecallerCreationParams::variable_chunk_sizes.
// Outputs per read (row pitch `pitch`): seq / qstr NUL padded, moves; seq_len / moves_len.  The reads come back in completion
// order; they are matched to their input by read_id.  stats8 = {batches called, partial batches called, samples processed,
// samples incl. padding} of the node's sample_stats(), and whether the runners report variable_chunk_sizes() (the adapter
// applies the reference's model rule, api/runner_creation.cpp:24-44: lstm_size in (128, 1024], else fixed chunks), then the
// runners' own counters summed: engine batches of variable runners, overflow batches (chunks that found no row), 0.
// restart_after > 0: the node is terminated and restarted after that many reads (NodeSmokeTest.cpp's restart case).
int adapter_run_basecaller_node(const mibc_model_desc *md, const float *const *weights, const int64_t *wnumel, int n_weights,
                                const char *device, int num_runners, int chunk_size, int overlap, int batch_size, int variable,
                                float qscale, float qbias, const uint16_t *reads, const int64_t *read_len, int n_reads, int pitch,
                                char *seq_out, char *qstr_out, uint8_t *moves_out, int64_t *seq_len, int64_t *moves_len,
                                double *stats8, int restart_after) {
    try {
        auto cfg = adapter_test_make_cfg(md, qscale, qbias, chunk_size, overlap, batch_size);
        std::vector<at::Tensor> ws = adapter_test_make_weights(weights, wnumel, n_weights);
        const basecall::BasecallerCreationParams params{cfg, std::string(device), 1.0f, basecall::PipelineType::simplex,
                                                        0.0f, false, false, variable != 0};
        auto [runners, num_devices] = basecall::create_hip_basecall_runners(params, ws, size_t(num_runners));
        (void)num_devices;
        stats8[4] = runners.at(0)->variable_chunk_sizes() ? 1.0 : 0.0;
        std::vector<basecall::ModelRunnerBase *> raw_runners;   // owned by the node from here on; alive until it is destroyed
        for (auto &r : runners) raw_runners.push_back(r.get());
        BasecallerNode node(std::move(runners), size_t(overlap), "hip_model", 1000, "BasecallerNode", 0);
        CaptureSink sink;
        Pipeline::connect(node, sink);
        sink.restart();
        node.restart();
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            if (restart_after > 0 && r == restart_after) {
                // NodeSmokeTest.cpp's restart case: drain and stop the node and its runners (BasecallerNode::terminate ->
                // ModelRunnerBase::terminate), start them again (restart()), carry on with the remaining reads
                node.terminate(TerminateOptions{});
                node.restart();
            }
            auto read = std::make_unique<SimplexRead>();
            read->read_common.raw_data =
                    at::from_blob(const_cast<uint16_t *>(reads + pos), {read_len[r]}, at::kHalf).clone();
            read->read_common.read_id = "read_" + std::to_string(r);
            pos += size_t(read_len[r]);
            node.push_message(std::move(read));
        }
        node.terminate(TerminateOptions{});
        sink.terminate(TerminateOptions{});
        if (int(sink.got.size()) != n_reads) throw std::runtime_error("BasecallerNode returned " + std::to_string(sink.got.size()) + " reads");
        std::memset(seq_out, 0, size_t(n_reads) * size_t(pitch));
        std::memset(qstr_out, 0, size_t(n_reads) * size_t(pitch));
        for (auto &rd : sink.got) {
            const auto &rc = rd->read_common;
            const int r = std::stoi(rc.read_id.substr(5));
            if (int(rc.seq.size()) > pitch || int(rc.moves.size()) > pitch) throw std::runtime_error("output pitch too small");
            std::memcpy(seq_out + size_t(r) * size_t(pitch), rc.seq.data(), rc.seq.size());
            std::memcpy(qstr_out + size_t(r) * size_t(pitch), rc.qstring.data(), rc.qstring.size());
            std::memcpy(moves_out + size_t(r) * size_t(pitch), rc.moves.data(), rc.moves.size());
            seq_len[r] = int64_t(rc.seq.size());
            moves_len[r] = int64_t(rc.moves.size());
        }
        const auto st = node.sample_stats();
        auto get = [&](const char *k) {
            const auto it = st.find(k);
            return it == st.end() ? -1.0 : it->second;
        };
        stats8[0] = get("batches_called");
        stats8[1] = get("partial_batches_called");
        stats8[2] = get("samples_processed");
        stats8[3] = get("samples_incl_padding");
        stats8[5] = stats8[6] = stats8[7] = 0.0;
        for (auto *r : raw_runners) {
            const auto rs = r->sample_stats();
            const auto a = rs.find("var_engine_batches"), b = rs.find("var_overflow_batches");
            if (a != rs.end()) stats8[5] += a->second;
            if (b != rs.end()) stats8[6] += b->second;
        }
        return 0;
    } catch (const std::exception &e) {
        adapter_test_set_error(e.what());
        return -1;
    }
}

}  // extern "C"
