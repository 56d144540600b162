// oracle/ref_scaler.cpp — TEST INFRASTRUCTURE ONLY (never linked or loaded by the product).
// Runs the reference's OWN ScalerNode on one read: read_pipeline/nodes/ScalerNode.cpp is compiled IN PLACE (it is
// #included here from /root/reference, nothing is copied), together with the reference's MessageSink.cpp, kits.cpp,
// tensor_utils.cpp (Makefile.ref).  Including the .cpp makes the functions of its anonymous namespace
// (determine_rna_adapter_pos, normalisation, med_mad, get_expected_open_pore_level) callable from the C driver below.
//
// What is NOT the reference's code in this translation unit, and why:
//   * dorado::Pipeline: the friend through which MessageSink::add_sink is reachable (the real ReadPipeline.cpp needs the whole
//     node graph); here it only connects the scaler to a capturing sink.
//   * is_read_message / get_read_common_data (messages.cpp:16-19, 425-439; that file needs htslib and modbase), restated in
//     five lines; BamDestructor (never called); config::to_string(ScalingStrategy) (BasecallModelConfig.cpp:513 needs toml11).
//   * utils::trim (torch_utils/trim.cpp:23-60 needs htslib): the restatement below; the function itself is pinned by the
//     reference's TrimTest answers in tests/test_oracle_pinned.py.
#include "read_pipeline/nodes/ScalerNode.cpp"

#include <cstdlib>
#include <cstring>
#include <limits>

namespace dorado {

void BamDestructor::operator()(bam1_t *) {}

bool is_read_message(const Message &message) {
    return std::holds_alternative<SimplexReadPtr>(message) || std::holds_alternative<DuplexReadPtr>(message);
}
const ReadCommon &get_read_common_data(const Message &message) {
    if (std::holds_alternative<SimplexReadPtr>(message)) return std::get<SimplexReadPtr>(message)->read_common;
    throw std::invalid_argument("ref_scaler: not a simplex read");
}
ReadCommon &get_read_common_data(Message &message) {
    return const_cast<ReadCommon &>(get_read_common_data(const_cast<const Message &>(message)));
}

namespace config {
std::string to_string(const ScalingStrategy &s) {
    return s == ScalingStrategy::MED_MAD ? "med_mad" : s == ScalingStrategy::QUANTILE ? "quantile" : "pa";
}
}  // namespace config

namespace utils {
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

class Pipeline {
public:
    static void connect(MessageSink &from, MessageSink &to) { from.add_sink(to); }
};

namespace {
class CaptureSink final : public MessageSink {
public:
    CaptureSink() : MessageSink(64, 1) {}
    ~CaptureSink() override { stop_input_processing(utils::AsyncQueueTerminateFast::Yes); }
    std::string get_name() const override { return "ref_scaler_capture"; }
    void terminate(const TerminateOptions &o) override { stop_input_processing(o.fast); }
    void restart() override {
        start_input_processing(
                [this] {
                    Message m;
                    while (get_input_message(m)) {
                        if (std::holds_alternative<SimplexReadPtr>(m)) got.push_back(std::get<SimplexReadPtr>(std::move(m)));
                    }
                },
                "ref_capture");
    }
    std::vector<SimplexReadPtr> got;
};
}  // namespace
}  // namespace dorado

static std::string g_scaler_err;

extern "C" {

const char *ref_scaler_last_error(void) { return g_scaler_err.c_str(); }

// ScalerNode.cpp:58-107 on a raw int16 read
int ref_determine_rna_adapter_pos(const int16_t *sig, long n) {
    dorado::SimplexRead read;
    read.read_common.raw_data =
            at::from_blob(const_cast<int16_t *>(sig), {n}, at::TensorOptions().dtype(at::kShort)).clone();
    return determine_rna_adapter_pos(read);
}

// ScalerNode.cpp:112-139; NaN if the flow cell has no expected level
float ref_expected_open_pore_level(const char *flow_cell_product_code) {
    const auto v = get_expected_open_pore_level(flow_cell_product_code ? flow_cell_product_code : "");
    return v ? *v : std::numeric_limits<float>::quiet_NaN();
}

// One read through the reference's ScalerNode (a real node with its worker thread, fed through push_message, its output
// captured by a sink).  strategy: 0 MED_MAD, 1 QUANTILE, 2 PA (config::ScalingStrategy order).  params7 = {quantile_a,
// quantile_b, shift_multiplier, scale_multiplier, standardise, mean, stdev}; cal3 = {scaling, offset, open_pore_level}.
// out_f16: the scaled (and trimmed) signal as f16 bits, capacity n; out_n its length.
// out_f4 = {read_common.scale, read_common.shift (both pA)}, out_i2 = {num_trimmed_samples, rna_adapter_end_signal_pos}.
int ref_scaler_node(const int16_t *raw, long n, int strategy, const float *params7, int sample_type_rna004, const float *cal3,
                    const char *flow_cell_product_code, uint16_t *out_f16, long *out_n, float *out_f2, int *out_i2) {
    try {
        using namespace dorado;
        config::SignalNormalisationParams p;
        p.strategy = static_cast<config::ScalingStrategy>(strategy);
        p.quantile = {params7[0], params7[1], params7[2], params7[3]};
        p.standardisation = {params7[4] != 0.0f, params7[5], params7[6]};
        ScalerNode node(p, sample_type_rna004 ? models::SampleType::RNA004 : models::SampleType::DNA, 1, 64);
        CaptureSink sink;
        Pipeline::connect(node, sink);
        sink.restart();
        node.restart();
        auto read = std::make_unique<SimplexRead>();
        read->read_common.raw_data =
                at::from_blob(const_cast<int16_t *>(raw), {n}, at::TensorOptions().dtype(at::kShort)).clone();
        read->read_common.read_id = "ref_scaler_read";
        read->read_common.flow_cell_product_code = flow_cell_product_code ? flow_cell_product_code : "";
        read->scaling = cal3[0];
        read->offset = cal3[1];
        read->open_pore_level = cal3[2];
        node.push_message(std::move(read));
        node.terminate(TerminateOptions{});
        sink.terminate(TerminateOptions{});
        if (sink.got.size() != 1) throw std::runtime_error("ref_scaler_node: no read came out");
        const auto &rc = sink.got[0]->read_common;
        const auto sig = rc.raw_data.contiguous();
        if (sig.dtype() != at::kHalf) throw std::runtime_error("ref_scaler_node: output is not f16");
        *out_n = sig.size(0);
        std::memcpy(out_f16, sig.data_ptr(), size_t(sig.size(0)) * 2);
        out_f2[0] = rc.scale;
        out_f2[1] = rc.shift;
        out_i2[0] = int(rc.num_trimmed_samples);
        out_i2[1] = int(rc.rna_adapter_end_signal_pos);
        return 0;
    } catch (const std::exception &e) {
        g_scaler_err = e.what();
        return -1;
    }
}

}  // extern "C"
