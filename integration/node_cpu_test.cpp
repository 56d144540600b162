// integration/node_cpu_test.cpp — TEST DRIVER, no GPU (built by oracle/Makefile.ref into oracle/_ref/libmibc_adapter_test.so).
// Host-logic equivalence under load: the reference's OWN BasecallerNode (BasecallerNode.cpp, MessageSink.cpp, chunk.cpp,
// stitch.cpp compiled in place) and this repo's node (dorado_amd::host::SimplexBasecaller) are given runners that "call" a
// chunk by the SAME pure function of its samples, and the same reads.  Whatever the two nodes do differently — chunk plan, queue
// choice between the chunk sizes, repeat-padding of short chunks, partial batches, stitching — shows up as a different read.
// The fake call has a receptive field (step t depends on the samples of steps t - 2 .. t + 2 of ITS chunk and on the chunk's
// length), like a network: the two chunks covering an overlap disagree near their edges, so the stitch points matter too.
#include "HipModelRunnerAdapter.h"
#include "read_pipeline/base/MessageSink.h"
#include "read_pipeline/nodes/BasecallerNode.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>

void adapter_test_set_error(const std::string &e);

namespace {

void fake_call(const uint16_t *x, size_t n, size_t stride, std::string &seq, std::string &qs, std::vector<uint8_t> &moves) {
    const size_t T = n / stride;
    moves.assign(T, 0);
    seq.clear();
    qs.clear();
    for (size_t t = 0; t < T; ++t) {
        // like a network, a step sees a window of its chunk (two steps either side, cut at the chunk's ends) and the chunk it
        // sits in: repeat-padding, the choice of chunk-size queue and the chunk plan all change what comes out
        uint32_t h = 2166136261u ^ uint32_t(n);
        const size_t a = t >= 2 ? (t - 2) * stride : 0, b = std::min(n, (t + 3) * stride);
        for (size_t k = a; k < b; ++k) {
            h = (h ^ (x[k] & 0xffu)) * 16777619u;
            h = (h ^ (x[k] >> 8)) * 16777619u;
        }
        if ((h >> 9) % 5 < 2) {
            moves[t] = 1;
            seq.push_back("ACGT"[(h >> 3) & 3]);
            qs.push_back(char('!' + (h >> 12) % 41));
        }
    }
}

using namespace dorado;

struct Counters {
    std::atomic<long> batches{0}, chunks{0};
};

// a runner behind the REFERENCE's interface
class FakeRefRunner final : public basecall::ModelRunnerBase {
public:
    FakeRefRunner(size_t chunk, size_t batch, int stride, int timeout_ms, Counters *c)
            : m_chunk(chunk), m_batch(batch), m_timeout(timeout_ms), m_rows(batch), m_c(c) {
        m_cfg.stride = stride;
        m_cfg.qscale = 1.0f;
        m_cfg.qbias = 0.0f;
    }
    void accept_chunk(int idx, const at::Tensor &chunk) override {
        const at::Tensor c = chunk.to(at::kHalf).contiguous();
        if (idx < 0 || size_t(idx) >= m_batch || size_t(c.numel()) != m_chunk) throw std::runtime_error("fake ref runner: bad chunk");
        const auto *p = reinterpret_cast<const uint16_t *>(c.data_ptr());
        m_rows[size_t(idx)].assign(p, p + m_chunk);
    }
    std::vector<basecall::decode::DecodedChunk> call_chunks(int n) override {
        std::vector<basecall::decode::DecodedChunk> out(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i)
            fake_call(m_rows[size_t(i)].data(), m_chunk, size_t(m_cfg.stride), out[size_t(i)].sequence, out[size_t(i)].qstring,
                      out[size_t(i)].moves);
        ++m_c->batches;
        m_c->chunks += n;
        return out;
    }
    const config::BasecallModelConfig &config() const override { return m_cfg; }
    size_t chunk_size() const override { return m_chunk; }
    size_t batch_size() const override { return m_batch; }
    std::pair<int, int> batch_timeouts_ms() const override { return {m_timeout, m_timeout}; }
    void terminate() override {}
    void restart() override {}
    std::string get_name() const override { return "FakeRefRunner"; }
    stats::NamedStats sample_stats() const override { return {}; }

private:
    config::BasecallModelConfig m_cfg;
    size_t m_chunk, m_batch;
    int m_timeout;
    std::vector<std::vector<uint16_t>> m_rows;
    Counters *m_c;
};

// the same runner behind this repo's mirror of that interface
class FakeHostRunner final : public dorado_amd::host::ModelRunnerBase {
public:
    FakeHostRunner(size_t chunk, size_t batch, int stride, Counters *c) : m_chunk(chunk), m_batch(batch), m_stride(stride), m_rows(batch), m_c(c) {
        std::memset(&m_desc, 0, sizeof(m_desc));
    }
    void accept_chunk(int idx, const uint16_t *f16, size_t n) override {
        if (idx < 0 || size_t(idx) >= m_batch || n != m_chunk) throw std::runtime_error("fake host runner: bad chunk");
        m_rows[size_t(idx)].assign(f16, f16 + n);
    }
    std::vector<dorado_amd::host::DecodedChunk> call_chunks(int n) override {
        std::vector<dorado_amd::host::DecodedChunk> out(static_cast<size_t>(n));
        for (int i = 0; i < n; ++i)
            fake_call(m_rows[size_t(i)].data(), m_chunk, size_t(m_stride), out[size_t(i)].sequence, out[size_t(i)].qstring,
                      out[size_t(i)].moves);
        ++m_c->batches;
        m_c->chunks += n;
        return out;
    }
    const mibc_model_desc &config() const override { return m_desc; }
    size_t chunk_size() const override { return m_chunk; }
    size_t batch_size() const override { return m_batch; }
    void terminate() override {}
    void restart() override {}
    std::string get_name() const override { return "FakeHostRunner"; }
    dorado_amd::host::NamedStats sample_stats() const override { return {}; }

private:
    mibc_model_desc m_desc;
    size_t m_chunk, m_batch;
    int m_stride;
    std::vector<std::vector<uint16_t>> m_rows;
    Counters *m_c;
};

class CaptureSink2 final : public MessageSink {
public:
    CaptureSink2() : MessageSink(1 << 16, 1) {}
    ~CaptureSink2() override { stop_input_processing(utils::AsyncQueueTerminateFast::Yes); }
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

namespace dorado {
class Pipeline {   // defined in basecaller_node_test.cpp (same library): the friend through which add_sink is reachable
public:
    static void connect(MessageSink &from, MessageSink &to);
};
}  // namespace dorado

extern "C" {

// reads: n_reads f16 reads back to back.  chunk_sizes[n_sizes] (descending), num_runners runners per size, in the reference's
// [runner][chunk size] order.  out6 = {reads that differ, first differing read (-1), bases called, chunks called by the
// reference node's runners, chunks called by the host node's runners, reference batches}.  Returns 0, or -1 with
// adapter_last_error() where either node throws.
int node_cpu_compare(const uint16_t *reads, const int64_t *read_len, int n_reads, const int *chunk_sizes, int n_sizes, int overlap,
                     int stride, int batch_size, int num_runners, int timeout_ms, long *out6) {
    try {
        using namespace dorado;
        Counters cref, chost;
        // ---- the reference's node
        std::vector<basecall::RunnerPtr> rr;
        for (int r = 0; r < num_runners; ++r)
            for (int s = 0; s < n_sizes; ++s)
                rr.push_back(std::make_unique<FakeRefRunner>(size_t(chunk_sizes[s]), size_t(batch_size), stride, timeout_ms, &cref));
        std::vector<std::string> ref_seq(static_cast<size_t>(n_reads)), ref_qs(static_cast<size_t>(n_reads));
        std::vector<std::vector<uint8_t>> ref_mv(static_cast<size_t>(n_reads));
        {
            BasecallerNode node(std::move(rr), size_t(overlap), "fake_model", 1000, "BasecallerNode", 0);
            CaptureSink2 sink;
            Pipeline::connect(node, sink);
            sink.restart();
            node.restart();
            size_t pos = 0;
            for (int r = 0; r < n_reads; ++r) {
                auto read = std::make_unique<SimplexRead>();
                read->read_common.raw_data = at::from_blob(const_cast<uint16_t *>(reads + pos), {read_len[r]}, at::kHalf).clone();
                read->read_common.read_id = "read_" + std::to_string(r);
                pos += size_t(read_len[r]);
                node.push_message(std::move(read));
            }
            node.terminate(TerminateOptions{});
            sink.terminate(TerminateOptions{});
            if (int(sink.got.size()) != n_reads) throw std::runtime_error("the reference node returned " + std::to_string(sink.got.size()) + " reads");
            for (auto &rd : sink.got) {
                const size_t r = size_t(std::stoi(rd->read_common.read_id.substr(5)));
                ref_seq[r] = rd->read_common.seq;
                ref_qs[r] = rd->read_common.qstring;
                ref_mv[r] = rd->read_common.moves;
            }
        }
        // ---- this repo's node
        std::vector<dorado_amd::host::RunnerPtr> hr;
        for (int r = 0; r < num_runners; ++r)
            for (int s = 0; s < n_sizes; ++s)
                hr.push_back(std::make_unique<FakeHostRunner>(size_t(chunk_sizes[s]), size_t(batch_size), stride, &chost));
        dorado_amd::host::SimplexBasecaller mine(std::move(hr), overlap, stride);
        std::vector<std::vector<uint16_t>> rv(static_cast<size_t>(n_reads));
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            rv[size_t(r)].assign(reads + pos, reads + pos + read_len[r]);
            pos += size_t(read_len[r]);
        }
        const auto called = mine.basecall(rv);
        long bad = 0, first = -1, bases = 0;
        for (int r = 0; r < n_reads; ++r) {
            const auto &c = called[size_t(r)];
            bases += long(ref_seq[size_t(r)].size());
            if (c.seq != ref_seq[size_t(r)] || c.qstring != ref_qs[size_t(r)] || c.moves != ref_mv[size_t(r)]) {
                if (first < 0) first = r;
                ++bad;
            }
        }
        out6[0] = bad;
        out6[1] = first;
        out6[2] = bases;
        out6[3] = cref.chunks.load();
        out6[4] = chost.chunks.load();
        out6[5] = cref.batches.load();
        return 0;
    } catch (const std::exception &e) {
        adapter_test_set_error(e.what());
        return -1;
    }
}

}  // extern "C"
