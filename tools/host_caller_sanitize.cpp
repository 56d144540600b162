// tools/host_caller_sanitize.cpp — the WHOLE host layer over the C-ABI test double (tools/fake_mibc.cpp), no GPU: functional
// cross-checks + ThreadSanitizer / AddressSanitizer (tools/sanitize_host.sh).
//   1. fixed chunk sizes: SimplexBasecaller over HipModelRunner / HipCaller (one caller per device, two chunk-size batch
//      dimensions, per-device FIFO, GPU thread with two asynchronous slots) must return exactly what the same node returns over
//      plain stand-in runners applying the engine double's call function directly — for f16 reads and for raw int16 reads with
//      per-read (shift, scale);
//   2. variable chunk sizes (HipModelRunner row packing, mibc_call_var_async on the two slots): every read must equal a direct
//      evaluation of its chunk plan (generate_variable_chunks -> call each chunk alone -> stitch_chunks);
//   3. two devices ("hip:all" with FAKE_MIBC_DEVICES=2), and scaler_node(HipCaller&) from two threads WHILE the node is calling
//      (the engine mutex serialises it with the GPU thread);
//   3b. injected engine failures: retried batches give the same reads, a double failure reaches the caller as an exception;
//   4. the automatic batch size against hand-computed values.
#include "mibc_host.h"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>

using namespace dorado_amd::host;

static void fake_call(const uint16_t *x, size_t n, size_t stride, std::string &seq, std::string &qs, std::vector<uint8_t> &moves) {
    const size_t T = n / stride;
    moves.assign(T, 0);
    seq.clear();
    qs.clear();
    for (size_t t = 0; t < T; ++t) {
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

class PlainRunner final : public ModelRunnerBase {
public:
    PlainRunner(size_t chunk, size_t batch, int stride) : m_chunk(chunk), m_batch(batch), m_stride(stride), m_rows(batch) {
        std::memset(&m_desc, 0, sizeof(m_desc));
    }
    void accept_chunk(int idx, const uint16_t *f16, size_t n) override { m_rows[size_t(idx)].assign(f16, f16 + n); }
    std::vector<DecodedChunk> call_chunks(int n) override {
        std::vector<DecodedChunk> out((size_t)n);
        for (int i = 0; i < n; ++i)
            fake_call(m_rows[size_t(i)].data(), m_chunk, size_t(m_stride), out[size_t(i)].sequence, out[size_t(i)].qstring, out[size_t(i)].moves);
        return out;
    }
    const mibc_model_desc &config() const override { return m_desc; }
    size_t chunk_size() const override { return m_chunk; }
    size_t batch_size() const override { return m_batch; }
    void terminate() override {}
    void restart() override {}
    std::string get_name() const override { return "plain"; }
    NamedStats sample_stats() const override { return {}; }

private:
    mibc_model_desc m_desc;
    size_t m_chunk, m_batch;
    int m_stride;
    std::vector<std::vector<uint16_t>> m_rows;
};

static void die(const char *what) {
    std::printf("FAILED: %s\n", what);
    std::exit(1);
}

static bool same(const std::vector<CalledRead> &a, const std::vector<CalledRead> &b) {
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); ++i)
        if (a[i].seq != b[i].seq || a[i].qstring != b[i].qstring || a[i].moves != b[i].moves) return false;
    return true;
}

int main() {
    mibc_model_desc d;
    std::memset(&d, 0, sizeof(d));
    d.n_convs = 3;
    d.conv_stride[0] = 1; d.conv_stride[1] = 1; d.conv_stride[2] = 6;
    d.lstm_size = 256; d.lstm_layers = 5; d.state_len = 3; d.outsize = 320; d.tx_d_model = 0;
    const int stride = 6, cs = 1200, overlap = 120;
    const mibc_decode_opts opts{32, 100.0f, 2.0f, 0.0f, 1.0f};
    const float *noweights = nullptr;
    std::mt19937 rng(11);
    auto make_reads = [&](size_t n, size_t maxlen) {
        std::vector<std::vector<uint16_t>> reads(n);
        for (auto &r : reads) {
            r.resize(1 + rng() % maxlen);
            for (auto &v : r) v = uint16_t(rng());
        }
        return reads;
    };
    const std::vector<int> sizes = simplex_chunk_sizes(d, cs, overlap);
    std::printf("chunk sizes %d / %d\n", sizes[0], sizes.size() > 1 ? sizes[1] : 0);

    // ---- 1. fixed chunk sizes: HipCaller path == plain runners
    {
        auto reads = make_reads(2500, 7000);
        std::vector<RunnerPtr> plain;
        for (int r = 0; r < 2; ++r)
            for (int s : sizes) plain.push_back(std::make_unique<PlainRunner>(size_t(s), 64, stride));
        SimplexBasecaller ref_node(std::move(plain), overlap, stride);
        const auto want = ref_node.basecall(reads);

        auto per_dev = create_basecall_runners(d, &noweights, 0, "hip:0", 2, sizes, 64, opts);
        std::vector<RunnerPtr> flat;
        for (auto &dev : per_dev)
            for (auto &r : dev) flat.push_back(std::move(r));
        if (flat.size() != 2 * sizes.size()) die("runner count");
        SimplexBasecaller node(std::move(flat), overlap, stride);
        const auto got = node.basecall(reads);
        if (!same(got, want)) die("fixed: HipCaller path differs from plain runners");
        const auto st = node.sample_stats();
        std::printf("fixed: %zu reads identical, %.0f batches (%.0f partial)\n", reads.size(), st.at("batches_called"), st.at("partial_batches_called"));

        // raw int16 reads + (shift, scale): must equal the f16 path on the scaled reads... the double scales with the same
        // arithmetic the engine documents, so compare against plain runners fed the scaled reads
        std::vector<std::vector<int16_t>> raw(400);
        std::vector<SimplexBasecaller::RawRead> rr;
        std::vector<std::vector<uint16_t>> scaled(raw.size());
        std::vector<std::pair<float, float>> ss(raw.size());
        for (size_t i = 0; i < raw.size(); ++i) {
            raw[i].resize(1 + rng() % 6000);
            for (auto &v : raw[i]) v = int16_t(300 + rng() % 600);
            ss[i] = {400.0f + float(rng() % 100), 50.0f + float(rng() % 60)};
        }
        // scale through a caller (mibc_scale_reads of the double) to get the f16 reads
        {
            HipCaller c(d, &noweights, 0, 0, cs, 64, opts);
            for (size_t i = 0; i < raw.size(); ++i) scaled[i] = c.scale_reads({{raw[i].data(), raw[i].size()}}, {ss[i]})[0];
        }
        for (size_t i = 0; i < raw.size(); ++i) rr.push_back({raw[i].data(), raw[i].size(), ss[i].first, ss[i].second});
        std::vector<RunnerPtr> plain2;
        for (int r = 0; r < 2; ++r)
            for (int s : sizes) plain2.push_back(std::make_unique<PlainRunner>(size_t(s), 64, stride));
        SimplexBasecaller ref2(std::move(plain2), overlap, stride);
        const auto want2 = ref2.basecall(scaled);
        const auto got2 = node.basecall_raw(rr);
        if (!same(got2, want2)) die("raw int16 reads differ from the prescaled f16 reads");
        std::printf("raw int16: %zu reads identical to the prescaled path\n", raw.size());
    }

    // ---- 2. variable chunk sizes
    {
        auto reads = make_reads(1500, 9000);
        CallerParams cp;
        cp.variable_chunk_sizes = true;
        auto per_dev = create_basecall_runners(d, &noweights, 0, "hip:0", 2, std::vector<int>{cs}, 64, opts, cp);
        std::vector<RunnerPtr> flat;
        for (auto &dev : per_dev)
            for (auto &r : dev) flat.push_back(std::move(r));
        if (!flat.at(0)->variable_chunk_sizes()) die("variable: runners are not variable");
        SimplexBasecaller node(std::move(flat), overlap, stride);
        const auto got = node.basecall_variable(reads);
        size_t bases = 0;
        for (size_t r = 0; r < reads.size(); ++r) {
            const auto iv = generate_variable_chunks(reads[r].size(), size_t(cs), size_t(stride), size_t(overlap));
            std::vector<Chunk> chunks(iv.size());
            std::vector<const Chunk *> cc;
            for (size_t i = 0; i < iv.size(); ++i) {
                const size_t len = iv[i].second - iv[i].first, padded = (len + stride - 1) / stride * stride;
                std::vector<uint16_t> x(padded);
                for (size_t p = 0; p < padded; ++p) x[p] = reads[r][iv[i].first + p % len];
                chunks[i].input_offset = iv[i].first;
                chunks[i].raw_chunk_size = len;
                fake_call(x.data(), padded, size_t(stride), chunks[i].seq, chunks[i].qstring, chunks[i].moves);
                cc.push_back(&chunks[i]);
            }
            const StitchedRead st = stitch_chunks(cc, reads[r].size(), stride);
            if (st.seq != got[r].seq || st.qstring != got[r].qstring || st.moves != got[r].moves) die("variable: a read differs from its chunk plan evaluated directly");
            bases += st.seq.size();
        }
        const auto st = node.sample_stats();
        std::printf("variable: %zu reads identical, %zu bases, %.0f engine batches\n", reads.size(), bases, st.at("batches_called"));
    }

    // ---- 3. two devices + scaler_node beside the node
    {
        setenv("FAKE_MIBC_DEVICES", "2", 1);
        auto reads = make_reads(2000, 5000);
        auto per_dev = create_basecall_runners(d, &noweights, 0, "hip:all", 2, sizes, 64, opts);
        if (per_dev.size() != 2) die("hip:all did not give two devices");
        std::vector<RunnerPtr> flat;
        for (auto &dev : per_dev)
            for (auto &r : dev) flat.push_back(std::move(r));
        SimplexBasecaller node(std::move(flat), overlap, stride);
        std::vector<RunnerPtr> plain;
        for (int r = 0; r < 2; ++r)
            for (int s : sizes) plain.push_back(std::make_unique<PlainRunner>(size_t(s), 64, stride));
        SimplexBasecaller ref_node(std::move(plain), overlap, stride);
        const auto want = ref_node.basecall(reads);
        HipCaller side(d, &noweights, 0, 1, cs, 64, opts);      // a second caller on device 1, used for scaling beside the node
        std::atomic<bool> stop{false};
        std::atomic<long> scaled{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 2; ++t)
            th.emplace_back([&, t] {
                std::mt19937 r2(50 + t);
                while (!stop.load()) {
                    std::vector<int16_t> x(2000 + r2() % 9000);
                    for (size_t i = 0; i < x.size(); ++i) x[i] = int16_t((i < 2500 ? 480 : 830) + int(r2() % 100));
                    SignalNormalisationParams p;
                    p.strategy = (scaled.load() & 1) ? ScalingStrategy::QUANTILE : ScalingStrategy::MED_MAD;
                    const ScaledRead s = scaler_node(side, p, t == 0, false, x.data(), x.size(), ReadCalibration{0.17f, -240.0f}, true);
                    if (s.signal_f16.empty()) die("scaler_node returned nothing");
                    ++scaled;
                }
            });
        const auto got = node.basecall(reads);
        stop.store(true);
        for (auto &t : th) t.join();
        if (!same(got, want)) die("two devices: differs from plain runners");
        std::printf("two devices: %zu reads identical; %ld reads scaled beside the node\n", reads.size(), scaled.load());
    }
    // ---- 3a. scaler_node for a whole read set (one statistics launch + one sample-map launch) == read by read
    {
        HipCaller c(d, &noweights, 0, 0, cs, 64, opts);
        std::vector<std::vector<int16_t>> raw(150);
        std::vector<ScalerInput> in;
        for (size_t i = 0; i < raw.size(); ++i) {
            const size_t n = 1 + rng() % 15000, cut = rng() % n;
            raw[i].resize(n);
            for (size_t k = 0; k < n; ++k) raw[i][k] = int16_t((k < cut ? 480 : 800) + int(rng() % 140) - 70 + (k < 60 ? 500 : 0));
            in.push_back({raw[i].data(), n, ReadCalibration{0.15f + 0.0001f * float(i), -230.0f, 201.0f + float(i % 5), "FLO-PRO114M"}, i % 7 == 0});
        }
        long checked = 0;
        for (int strat = 0; strat < 3; ++strat)
            for (int rna = 0; rna < 2; ++rna)
                for (int sig = 0; sig < 2; ++sig) {
                    SignalNormalisationParams p;
                    p.strategy = strat == 0 ? ScalingStrategy::MED_MAD : strat == 1 ? ScalingStrategy::QUANTILE : ScalingStrategy::PA;
                    p.standardisation.standardise = strat == 2 && rna == 0;
                    p.standardisation.mean = 90.0f;
                    p.standardisation.stdev = 22.0f;
                    const auto all = scaler_node(c, p, rna != 0, in, sig != 0);
                    for (size_t i = 0; i < in.size(); ++i) {
                        const ScaledRead one = scaler_node(c, p, rna != 0, in[i].has_rna_based_adapters, in[i].raw, in[i].n_samples, in[i].cal, sig != 0);
                        const ScaledRead &b = all[i];
                        if (one.signal_f16 != b.signal_f16 || one.num_trimmed_samples != b.num_trimmed_samples ||
                            one.rna_adapter_end_signal_pos != b.rna_adapter_end_signal_pos || one.first_sample != b.first_sample ||
                            one.scaling.shift != b.scaling.shift || one.scaling.scale != b.scaling.scale ||
                            one.scaling.scale_pa != b.scaling.scale_pa || one.scaling.shift_pa != b.scaling.shift_pa ||
                            one.scaling.open_pore_adjustment != b.scaling.open_pore_adjustment)
                            die("scaler_node over a read set differs from the per-read form");
                        ++checked;
                    }
                }
        std::printf("scaler_node read sets: %ld read x configuration results identical to the per-read form\n", checked);
    }

    // ---- 3b. error paths (CudaCaller.cpp:698-704: a failed batch is retried once, synchronously; a second failure reaches the
    //          caller): every third asynchronous batch fails -> the retries must give the same reads; with the synchronous
    //          calls failing too the node must throw, not hang or return garbage
    {
        unsetenv("FAKE_MIBC_DEVICES");
        auto reads = make_reads(1200, 6000);
        std::vector<RunnerPtr> plain;
        for (int r = 0; r < 2; ++r)
            for (int s : sizes) plain.push_back(std::make_unique<PlainRunner>(size_t(s), 64, stride));
        SimplexBasecaller ref_node(std::move(plain), overlap, stride);
        const auto want = ref_node.basecall(reads);
        auto make_node2 = [&](bool variable) {
            CallerParams cp;
            cp.variable_chunk_sizes = variable;
            auto per_dev = create_basecall_runners(d, &noweights, 0, "hip:0", 2, variable ? std::vector<int>{cs} : sizes, 64, opts, cp);
            std::vector<RunnerPtr> flat;
            for (auto &dev : per_dev)
                for (auto &r : dev) flat.push_back(std::move(r));
            return std::make_unique<SimplexBasecaller>(std::move(flat), overlap, stride);
        };
        setenv("FAKE_MIBC_FAIL_ASYNC_EVERY", "3", 1);
        if (!same(make_node2(false)->basecall(reads), want)) die("retried batches differ");
        const auto var_ok = [&] {
            unsetenv("FAKE_MIBC_FAIL_ASYNC_EVERY");
            return make_node2(true)->basecall_variable(reads);
        }();
        setenv("FAKE_MIBC_FAIL_ASYNC_EVERY", "3", 1);
        if (!same(make_node2(true)->basecall_variable(reads), var_ok)) die("retried variable batches differ");
        setenv("FAKE_MIBC_FAIL_SYNC", "1", 1);
        bool threw = false;
        try {
            (void)make_node2(false)->basecall(reads);
        } catch (const std::exception &e) {
            threw = std::string(e.what()).find("mibc_call") != std::string::npos;
        }
        if (!threw) die("a batch that fails twice must reach the caller as an exception");
        unsetenv("FAKE_MIBC_FAIL_SYNC");
        unsetenv("FAKE_MIBC_FAIL_ASYNC_EVERY");
        std::printf("error paths: every third batch failed and was retried (fixed + variable): reads identical; double failure throws\n");
    }

    // ---- 4. automatic batch size (CudaCaller::determine_batch_dims, CudaCaller.cpp:382-627, as HipCaller::choose_batch_size
    //         restates it) against the double's memory figures (64 B x T_in per chunk + 64 MB fixed) and time model
    //         (0.05 ms + 1e-6 ms per row-sample, 20 % cheaper from 256 rows on)
    {
        auto chosen = [&](int requested, bool sweep, const char *free_mb) {
            if (free_mb) setenv("FAKE_MIBC_FREE_MB", free_mb, 1);
            else unsetenv("FAKE_MIBC_FREE_MB");
            CallerParams cp;
            cp.run_batchsize_benchmarks = sweep;
            HipCaller c(d, &noweights, 0, 0, std::vector<int>{cs}, requested, opts, cp);
            return std::make_pair(c.batch_size(), c.batch_timings().size());
        };
        // explicit request: rounded up to the granularity
        if (chosen(100, false, nullptr).first != 128) die("auto batch: explicit request not rounded to the granule");
        // plenty of memory, no sweep: the engine's knee (256 granules)
        if (chosen(0, false, nullptr).first != 256 * 64) die("auto batch: knee");
        // memory cap: 0.8 x 1200 MB - 1 GB - 64 MB fixed is negative -> one granule with a warning (the reference falls back too)
        if (chosen(0, false, "1200").first != 64) die("auto batch: fallback to one granule");
        // cap in between: 0.8 x 1500 MB - 1024 MB - 64 MB = 112 MB over 64 x 1200 B per chunk = 1529 -> 1472 (23 granules)
        if (chosen(0, false, "1500").first != 1472) die("auto batch: memory cap");
        // timing sweep: ladder 32768 .. 1024, all within 5 % of the best time per chunk -> the smallest of the ladder
        const auto sw = chosen(0, true, nullptr);
        if (sw.first != 1024 || sw.second != 6) die("auto batch: timing sweep did not pick the smallest batch within the penalty");
        // sweep under a cap of 640 rows: 640, 320, 128, 64 -> 320 is 5.5 % slower per chunk than 640 -> 640
        // (0.8 x F - 1088 MB = 640 x 76800 B  ->  F = 1418.6 MB)
        const auto sc = chosen(0, true, "1419");
        if (sc.first != 640) { std::printf("got %d\n", sc.first); die("auto batch: sweep under a memory cap"); }
        unsetenv("FAKE_MIBC_FREE_MB");
        // quantised wide layers (lstm_quant, lstm_size >= 512) run only as 256-row clusters: the granularity is 256, so a
        // request of 1100 becomes 1280 (not 1120), the knee is one cluster per lstm_size / 128 CUs, and a memory cap is a
        // whole number of clusters
        {
            mibc_model_desc dq = d;
            dq.lstm_size = 1024;
            dq.lstm_quant = 1;
            auto chosen_q = [&](int requested, const char *free_mb) {
                if (free_mb) setenv("FAKE_MIBC_FREE_MB", free_mb, 1);
                else unsetenv("FAKE_MIBC_FREE_MB");
                HipCaller c(dq, &noweights, 0, 0, std::vector<int>{cs}, requested, opts, CallerParams{});
                return c.batch_size();
            };
            if (chosen_q(1100, nullptr) != 1280) die("quantised wide layers: request not rounded to a 256-row cluster");
            if (chosen_q(0, nullptr) != 32 * 256) die("quantised wide layers: knee");
            if (chosen_q(0, "1500") != 1280) die("quantised wide layers: memory cap not a whole number of clusters");   // 1529 -> 1280
            unsetenv("FAKE_MIBC_FREE_MB");
        }
        std::printf("auto batch size: request / knee / fallback / memory cap / sweep / capped sweep / quantised clusters as specified\n");
    }
    std::printf("host_caller_sanitize: all checks passed\n");
    return 0;
}
