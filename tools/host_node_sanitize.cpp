// tools/host_node_sanitize.cpp — host-layer logic under ThreadSanitizer / AddressSanitizer + UBSan, no GPU
// (tools/sanitize_host.sh; tests/test_host_cpu.py runs the thread-sanitizer build).  The node (host::SimplexBasecaller: shared
// chunk queues, one worker thread per runner, stitch by the worker that delivers a read's last chunk) runs over stand-in runners
// that call a chunk by a pure function of its samples; the ScalerNode mirror (host::scaler_node) runs through its ScalerOps seam
// from eight threads; the row packer of the variable-chunk path places random chunk sets.
#include "mibc_host.h"
#include "tensor_loader.h"

#include <dirent.h>
#include <fstream>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <random>
#include <thread>
using namespace dorado_amd::host;
static void fake_call(const uint16_t *x, size_t n, size_t stride, std::string &seq, std::string &qs, std::vector<uint8_t> &moves) {
    const size_t T = n / stride; moves.assign(T, 0); seq.clear(); qs.clear();
    for (size_t t = 0; t < T; ++t) { uint32_t h = 2166136261u ^ uint32_t(n);
        const size_t a = t >= 2 ? (t - 2) * stride : 0, b = std::min(n, (t + 3) * stride);
        for (size_t k = a; k < b; ++k) { h = (h ^ (x[k] & 0xffu)) * 16777619u; h = (h ^ (x[k] >> 8)) * 16777619u; }
        if ((h >> 9) % 5 < 2) { moves[t] = 1; seq.push_back("ACGT"[(h >> 3) & 3]); qs.push_back(char('!' + (h >> 12) % 41)); } }
}
class FakeHostRunner final : public ModelRunnerBase {
public:
    FakeHostRunner(size_t chunk, size_t batch, int stride) : m_chunk(chunk), m_batch(batch), m_stride(stride), m_rows(batch) { std::memset(&m_desc, 0, sizeof(m_desc)); }
    void accept_chunk(int idx, const uint16_t *f16, size_t n) override { m_rows[size_t(idx)].assign(f16, f16 + n); }
    std::vector<DecodedChunk> call_chunks(int n) override { std::vector<DecodedChunk> out((size_t)n);
        for (int i = 0; i < n; ++i) fake_call(m_rows[size_t(i)].data(), m_chunk, size_t(m_stride), out[size_t(i)].sequence, out[size_t(i)].qstring, out[size_t(i)].moves); return out; }
    const mibc_model_desc &config() const override { return m_desc; }
    size_t chunk_size() const override { return m_chunk; }
    size_t batch_size() const override { return m_batch; }
    void terminate() override {} void restart() override {}
    std::string get_name() const override { return "fake"; }
    NamedStats sample_stats() const override { return {}; }
private: mibc_model_desc m_desc; size_t m_chunk, m_batch; int m_stride; std::vector<std::vector<uint16_t>> m_rows;
};
static void scaler_threads() {
    std::vector<std::thread> th;
    std::atomic<long> sum{0};
    for (int t = 0; t < 8; ++t)
        th.emplace_back([t, &sum] {
            std::mt19937 rng(100 + t);
            ScalerOps ops;
            ops.stats = [](const int16_t *x, size_t n, const SignalNormalisationParams &) {
                double m = 0; for (size_t i = 0; i < n; ++i) m += x[i];
                return std::make_pair(float(m / double(n ? n : 1)), 50.0f);
            };
            ops.scale = [](const int16_t *x, size_t n, float sh, float sc) {
                std::vector<uint16_t> o(n); for (size_t i = 0; i < n; ++i) o[i] = uint16_t(int((x[i] - sh) / sc * 64) & 0x3ff); return o;
            };
            for (int k = 0; k < 60; ++k) {
                const size_t n = 1 + rng() % 20000, cut = rng() % n;
                std::vector<int16_t> x(n);
                for (size_t i = 0; i < n; ++i) x[i] = int16_t((i < cut ? 480 : 830) + int(rng() % 120) - 60);
                SignalNormalisationParams p;
                p.strategy = k % 3 == 0 ? ScalingStrategy::PA : k % 3 == 1 ? ScalingStrategy::QUANTILE : ScalingStrategy::MED_MAD;
                p.standardisation.standardise = k % 2;
                const ScaledRead r = scaler_node(ops, p, k % 4 < 2, false, x.data(), n, ReadCalibration{0.17f, -240.0f, 201.0f, "FLO-PRO114M"}, k % 5 != 0);
                sum += r.num_trimmed_samples + long(r.signal_f16.size());
            }
        });
    for (auto &t : th) t.join();
    std::printf("scaler_node: 8 threads x 60 reads, checksum %ld\n", sum.load());
}

static void packer_cases() {
    std::mt19937 rng(9);
    long placed = 0;
    for (int it = 0; it < 300; ++it) {
        const size_t rows = 32 * (1 + rng() % 8), cs = 6 * (50 + rng() % 400), gap = 12;
        HipModelRunner::RowPacker pk;
        pk.reset(rows, cs, gap);
        for (int k = 0; k < 2000; ++k) {
            int row = -1, start = 0;
            const size_t n = 6 * (1 + rng() % (cs / 6));
            if (!pk.place(n, row, start)) break;
            if (row < 0 || size_t(row) >= rows || size_t(start) + n > cs) { std::printf("packer: chunk outside its row\n"); std::abort(); }
            ++placed;
        }
    }
    std::printf("row packer: %ld chunks placed\n", placed);
}

// the ".tensor" parser reads files from disk (ZIP central directory + a pickle subset): mutated copies of the reference's fixtures
// must either load or throw — never read outside the file (the address / undefined-behaviour sanitizers watch)
static void fuzz_tensor_loader(const std::string &root) {
    const std::string dir = root + "/tests/golden/tensor";
    const char *tmp = std::getenv("TMPDIR");
    const std::string scratch = std::string(tmp ? tmp : "/tmp") + "/mibc_fuzz_" + std::to_string(long(getpid())) + ".tensor";
    std::mt19937 rng(77);
    long ok = 0, threw = 0, files = 0;
    DIR *d = opendir(dir.c_str());
    if (!d) {
        std::printf("tensor loader fuzz: %s not found, skipped\n", dir.c_str());
        return;
    }
    while (dirent *e = readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() < 8 || name.substr(name.size() - 7) != ".tensor") continue;
        std::ifstream f(dir + "/" + name, std::ios::binary);
        std::vector<char> orig((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
        if (orig.empty()) continue;
        ++files;
        for (int it = 0; it < 400; ++it) {
            std::vector<char> m = orig;
            const int kind = int(rng() % 4);
            if (kind == 0) {
                for (int k = 0, nk = 1 + int(rng() % 8); k < nk; ++k) m[rng() % m.size()] = char(rng());
            } else if (kind == 1) {
                m.resize(rng() % m.size());
            } else if (kind == 2) {   // hit the tail: end-of-central-directory record and the directory itself
                const size_t span = std::min<size_t>(m.size(), 400);
                for (int k = 0; k < 4; ++k) m[m.size() - 1 - rng() % span] = char(rng());
            } else {                  // splice a block from elsewhere in the file
                const size_t len = 1 + rng() % std::min<size_t>(64, m.size()), a = rng() % (m.size() - len + 1), b = rng() % (m.size() - len + 1);
                std::memmove(m.data() + a, m.data() + b, len);
            }
            {
                std::ofstream o(scratch, std::ios::binary | std::ios::trunc);
                o.write(m.data(), std::streamsize(m.size()));
            }
            try {
                const auto ts = load_tensor_file(scratch);
                size_t bytes = 0;
                for (auto &t : ts) bytes += t.data.size() + t.to_float().size();
                ok += bytes >= 0;
            } catch (const std::exception &) {
                ++threw;
            }
        }
    }
    closedir(d);
    std::remove(scratch.c_str());
    std::printf("tensor loader fuzz: %ld files, %ld mutants loaded, %ld rejected\n", files, ok, threw);
    if (files == 0) { std::printf("FAILED: no .tensor fixtures\n"); std::exit(1); }
}

int main(int argc, char **argv) {
    fuzz_tensor_loader(argc > 1 ? argv[1] : ".");
    scaler_threads();
    packer_cases();
    std::mt19937 rng(5);
    for (int rep = 0; rep < 3; ++rep) {
        std::vector<RunnerPtr> hr;
        for (int r = 0; r < 4; ++r) for (size_t cs : {1200, 600}) hr.push_back(std::make_unique<FakeHostRunner>(cs, 16, 6));
        SimplexBasecaller node(std::move(hr), 120, 6);
        std::vector<std::vector<uint16_t>> reads(rep == 2 ? 6000 : 1500);
        for (auto &r : reads) { r.resize(1 + rng() % 7000); for (auto &v : r) v = uint16_t(rng()); }
        auto out = node.basecall(reads);
        size_t bases = 0; for (auto &c : out) bases += c.seq.size();
        auto st = node.sample_stats();
        std::printf("rep %d reads %zu bases %zu batches %.0f\n", rep, reads.size(), bases, st["batches_called"]);
    }
    return 0;
}
