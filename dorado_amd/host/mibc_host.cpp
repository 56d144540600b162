// dorado_amd/host/mibc_host.cpp — see mibc_host.h for what mirrors what in the reference.
#include "mibc_host.h"

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <numeric>
#include <set>
#include <stdexcept>

namespace dorado_amd::host {

// ------------------------------------------------------------------ device strings
bool try_parse_device_ids(const std::string &device_string, size_t num_devices,
                          std::vector<int> &device_ids, std::string &error_message) {
    device_ids.clear();
    std::string prefix;
    if (device_string.rfind("hip:", 0) == 0) {
        prefix = "hip:";
    } else if (device_string.rfind("cuda:", 0) == 0) {
        prefix = "cuda:";
    } else {
        return true;  // not a GPU device string (e.g. "cpu"): not an error
    }
    const std::string rest = device_string.substr(prefix.size());
    if (rest == "all" || rest == "auto") {
        if (num_devices == 0) {
            error_message = "device string set to " + device_string + " but no GPU devices available.";
            return false;
        }
        for (size_t i = 0; i < num_devices; ++i) device_ids.push_back(int(i));
        return true;
    }
    std::set<int> unique;
    size_t n_tokens = 0;
    size_t pos = 0;
    while (pos <= rest.size()) {
        const size_t comma = rest.find(',', pos);
        const std::string tok = rest.substr(pos, comma == std::string::npos ? std::string::npos : comma - pos);
        pos = (comma == std::string::npos) ? rest.size() + 1 : comma + 1;
        ++n_tokens;
        if (tok.empty() || !std::all_of(tok.begin(), tok.end(), [](char c) { return c >= '0' && c <= '9'; })) {
            error_message = "Invalid device string: " + device_string;
            return false;
        }
        const long id = std::stol(tok);
        if (id < 0 || size_t(id) >= num_devices) {
            error_message = "Invalid device index " + tok + " in " + device_string + " (" +
                            std::to_string(num_devices) + " devices available)";
            return false;
        }
        unique.insert(int(id));
    }
    if (unique.size() != n_tokens) {
        error_message = "Duplicate device index in " + device_string;
        return false;
    }
    device_ids.assign(unique.begin(), unique.end());
    return true;
}

// ------------------------------------------------------------------ chunking (a12)
std::vector<size_t> generate_chunks(size_t num_samples, size_t chunk_size, size_t stride, size_t overlap) {
    if (num_samples == 0) throw std::runtime_error("generate_chunks: empty read");
    if (stride == 0) throw std::logic_error("generate_chunks: invalid stride 0");
    if (chunk_size == 0 || (chunk_size % stride) != 0 || chunk_size <= overlap)
        throw std::logic_error("generate_chunks: invalid chunk size " + std::to_string(chunk_size));
    if ((overlap % stride) != 0) throw std::logic_error("generate_chunks: invalid overlap " + std::to_string(overlap));
    std::vector<size_t> offsets{0};
    size_t last = num_samples > chunk_size ? num_samples - chunk_size : 0;
    if (const size_t mis = last % stride; mis != 0) last += stride - mis;
    const size_t step = chunk_size - overlap;
    size_t off = 0;
    while (off + chunk_size < num_samples) {
        off = std::min(off + step, last);
        offsets.push_back(off);
    }
    return offsets;
}

std::vector<std::pair<size_t, size_t>> generate_variable_chunks(size_t num_samples, size_t chunk_size, size_t stride,
                                                               size_t overlap) {
    if (num_samples == 0) throw std::runtime_error("generate_variable_chunks: empty read");
    if (stride == 0) throw std::logic_error("generate_variable_chunks: invalid stride 0");
    if (chunk_size == 0 || (chunk_size % stride) != 0 || chunk_size == stride || chunk_size <= overlap)
        throw std::logic_error("generate_variable_chunks: invalid chunk size " + std::to_string(chunk_size));
    if ((overlap % stride) != 0 || (stride != 1 && overlap == 0))
        throw std::logic_error("generate_variable_chunks: invalid overlap " + std::to_string(overlap));
    const size_t num_chunks =
            1 + (num_samples > chunk_size
                         ? size_t(std::ceil(double(num_samples - chunk_size) / double(chunk_size - overlap)))
                         : 0);
    const size_t with_overlaps = num_samples + (num_chunks - 1) * overlap;
    const size_t num_longer = with_overlaps % num_chunks, adjusted = with_overlaps / num_chunks;
    std::vector<std::pair<size_t, size_t>> iv;
    for (size_t i = 0, start = 0; i < num_chunks; ++i) {
        iv.emplace_back(start, start + adjusted + (i < num_longer ? 1 : 0));
        start = iv.back().second - overlap;
    }
    for (size_t i = 1; i < num_chunks; ++i)
        if (const size_t mis = iv[i].first % stride; mis != 0) iv[i].first += stride - mis;
    for (size_t i = 0; i + 1 < num_chunks; ++i) iv[i].second -= iv[i].second % stride;
    return iv;
}

StitchedRead stitch_chunks(const std::vector<const Chunk *> &cc, size_t raw_samples, int stride) {
    StitchedRead r;
    int start_pos = 0, mid_front = 0;
    for (size_t i = 0; i + 1 < cc.size(); ++i) {
        const Chunk &cur = *cc[i], &nxt = *cc[i + 1];
        const int overlap_size = int((cur.raw_chunk_size + cur.input_offset) - nxt.input_offset);
        const int overlap_ds = overlap_size / stride;
        const int mid_rear = overlap_ds / 2;
        const int trim = std::accumulate(cur.moves.end() - mid_rear, cur.moves.end(), 0);
        const int end_pos = int(cur.seq.size()) - trim;
        r.seq.append(cur.seq, size_t(start_pos), size_t(end_pos - start_pos));
        r.qstring.append(cur.qstring, size_t(start_pos), size_t(end_pos - start_pos));
        r.moves.insert(r.moves.end(), cur.moves.begin() + mid_front, cur.moves.end() - mid_rear);
        mid_front = overlap_ds - mid_rear;
        start_pos = std::accumulate(nxt.moves.begin(), nxt.moves.begin() + mid_front, 0);
    }
    const Chunk &last = *cc.back();
    r.moves.insert(r.moves.end(), last.moves.begin() + mid_front, last.moves.end());
    if (cc.size() == 1) {
        const size_t keep = raw_samples / size_t(stride);
        if (r.moves.size() > keep) r.moves.resize(keep);
        const int end = std::accumulate(r.moves.begin(), r.moves.end(), 0);
        r.seq += last.seq.substr(size_t(start_pos), size_t(end));
        r.qstring += last.qstring.substr(size_t(start_pos), size_t(end));
    } else {
        r.seq += last.seq.substr(size_t(start_pos));
        r.qstring += last.qstring.substr(size_t(start_pos));
    }
    if (r.moves.size() > raw_samples / size_t(stride)) {  // partial stride overhang
        if (r.moves.back() == 1) {
            r.seq.pop_back();
            r.qstring.pop_back();
        }
        r.moves.pop_back();
    }
    return r;
}

// ------------------------------------------------------------------ HipCaller
// "Global task queues, one per GPU.  This ensures that tasks from different clients are processed in the order they
// became ready" (CudaCaller.cpp:204-214).  Every caller of a device pushes here; a caller's GPU thread takes the front
// task only when it is its own, and only while no OTHER caller has batches in flight on the device (`owner`), so the
// kernels of two callers never interleave (the reference pops a task after it has completed for the same reason,
// :709-718).
struct HipCaller::DeviceQueue {
    std::mutex mut;
    std::condition_variable cv;
    std::deque<std::shared_ptr<NNTask>> q;   // front = oldest
    HipCaller *owner = nullptr;
};

// Timed wait on the steady clock — except in a ThreadSanitizer build: GCC 11's libtsan does not intercept
// pthread_cond_clockwait (what libstdc++'s wait_for uses for the steady clock), misses the unlock inside the wait and then
// reports every access under that mutex as a race ("double lock of a mutex"); the system-clock wait goes through the intercepted
// pthread_cond_timedwait.  (tools/sanitize_host.sh)
template <class Pred>
static bool cv_wait_for(std::condition_variable &cv, std::unique_lock<std::mutex> &lk, std::chrono::microseconds d, Pred pred) {
#if defined(__SANITIZE_THREAD__)
    return cv.wait_until(lk, std::chrono::system_clock::now() + d, pred);
#else
    return cv.wait_for(lk, d, pred);
#endif
}

static HipCaller::DeviceQueue &device_queue(int device) {
    static std::mutex m;
    static std::map<int, std::unique_ptr<HipCaller::DeviceQueue>> queues;
    std::lock_guard<std::mutex> lk(m);
    auto &q = queues[device];
    if (!q) q = std::make_unique<HipCaller::DeviceQueue>();
    return *q;
}

// Batch size for the largest chunk size.  requested > 0: as given.  Otherwise automatic — the role of
// CudaCaller::determine_batch_dims (CudaCaller.cpp:382-627): the memory cap is memory_limit_fraction of the free device
// memory minus 1 GB (:434-439) over mibc_query_memory's bytes per chunk (= the memory model of :323-369); inside the cap
// either the engine's known knee (one LSTM workgroup / cluster slot on every CU) or, with run_batchsize_benchmarks
// (or requested == -1), the reference's timing sweep (:552-627).
int HipCaller::choose_batch_size(int chunk_size, int requested) {
    const int g = mibc_batch_granularity(m_engine);
    if (requested > 0) return (requested + g - 1) / g * g;
    size_t per_chunk = 0, fixed = 0, free_b = 0, total_b = 0;
    if (mibc_query_memory(m_engine, chunk_size, &per_chunk, &fixed) != MIBC_OK ||
        mibc_device_memory(m_device, &free_b, &total_b) != MIBC_OK || per_chunk == 0)
        throw std::runtime_error(std::string("auto batch size: ") + mibc_last_error(m_engine));
    const double budget = double(m_params.memory_limit_fraction) * double(free_b) - double(1ull << 30) - double(fixed);
    const long cap = budget > 0 ? long(budget / double(per_chunk)) / g * g : 0;
    if (cap < g) {
        // the reference warns and falls back to its default batch instead of failing (CudaCaller.cpp:441-445); here the
        // smallest batch the engine accepts — mibc_reserve still fails loudly if even that does not fit
        fprintf(stderr, "[mibc] hip:%d auto batch size: less than one batch granule fits into the memory limit (%.2f of %zu MB free); "
                "falling back to batch %d\n", m_device, double(m_params.memory_limit_fraction), free_b >> 20, g);
        return g;
    }
    // the knee: one LSTM workgroup (g rows) on each of the 256 CUs; cluster kernels: one 256-row cluster per lstm_size / 128 CUs
    // (the quantised wide layers report g = 256 = one cluster, so 256 g would be 8 x too many)
    const long want = (m_desc.tx_d_model > 0) ? 1024
                      : (g >= 256 && m_desc.lstm_size >= 512) ? (256L / (m_desc.lstm_size / 128)) * 256L : 256L * g;
    long n = std::min(want, cap);
    if (m_params.run_batchsize_benchmarks || requested < 0) {
        // time the network alone on a short chunk (288 output steps, :497-503) for a descending ladder of batch sizes
        // under the cap, min of the timed runs, and take the SMALLEST batch whose time per chunk is within
        // batch_size_time_penalty of the best
        const int stride = dorado_amd::host::model_stride(m_desc);
        const int gran = chunk_size_granularity(m_desc);
        const int t_bench = std::max(gran, (288 * stride) / gran * gran);
        const long top = std::max<long>(g, std::min<long>(2 * want, cap));
        std::vector<long> ladder;
        for (long b = top; b >= g && ladder.size() < 6; b = (b / 2) / g * g) ladder.push_back(b);
        double best = 1e30;
        std::vector<std::pair<long, double>> timed;
        for (long b : ladder) {
            float ms = 0;
            if (mibc_time_forward(m_engine, int(b), t_bench, &ms) != MIBC_OK) continue;
            timed.push_back({b, double(ms) / double(b)});
            best = std::min(best, timed.back().second);
            m_batch_timings.push_back({int(b), double(ms) / double(b)});
            if (m_params.emit_batchsize_benchmarks)
                fprintf(stderr, "[mibc] hip:%d batch %ld chunk %d: %.6f ms per chunk\n", m_device, b, t_bench, double(ms) / double(b));
        }
        for (auto &tb : timed)
            if (tb.second <= best * (1.0 + double(m_params.batch_size_time_penalty))) n = tb.first;   // descending: ends at the smallest
        n = std::min(n, cap);
    }
    return int(std::max<long>(g, n));
}

HipCaller::HipCaller(const mibc_model_desc &desc, const float *const *weights, int n_weights, int device,
                     const std::vector<int> &chunk_sizes, int batch_size, const mibc_decode_opts &opts,
                     const CallerParams &params)
        : m_desc(desc), m_opts(opts), m_params(params), m_device(device) {
    if (chunk_sizes.empty()) throw std::invalid_argument("HipCaller: no chunk size");
    const int rc = mibc_create(device, &desc, weights, n_weights, &m_engine);
    if (rc != MIBC_OK) throw std::runtime_error(std::string("mibc_create: ") + mibc_last_error(nullptr));
    try {
        // batch dimensions, largest chunk size first (CudaCaller.cpp:408-410); one workspace, sized for the largest,
        // serves all of them (the engine switches geometry per call)
        std::vector<int> sizes(chunk_sizes);
        std::sort(sizes.rbegin(), sizes.rend());
        sizes.erase(std::unique(sizes.begin(), sizes.end()), sizes.end());
        const int n = choose_batch_size(sizes[0], batch_size);
        for (int cs : sizes) m_dims.push_back({n, cs, mibc_output_steps(m_engine, cs)});
        if (mibc_reserve(m_engine, n, sizes[0]) != MIBC_OK)
            throw std::runtime_error(std::string("mibc_reserve: ") + mibc_last_error(m_engine));
    } catch (...) {
        mibc_destroy(m_engine);
        throw;
    }
    m_queue = &device_queue(device);
    start_thread();
}

HipCaller::~HipCaller() {
    terminate();
    mibc_destroy(m_engine);
}

void HipCaller::start_thread() {
    m_terminate.store(false);
    m_thread = std::thread([this] { gpu_thread_fn(); });
}

void HipCaller::terminate() {  // idempotent (CudaCaller.cpp:273-281)
    m_terminate.store(true);
    {
        std::lock_guard<std::mutex> lk(m_queue->mut);
    }
    m_queue->cv.notify_all();
    if (m_thread.joinable()) m_thread.join();
}

void HipCaller::restart() {  // CudaCaller.cpp:283-287
    if (m_terminate.load()) start_thread();
}

void HipCaller::run_task(const std::shared_ptr<NNTask> &task) {
    task->caller = this;
    {
        std::lock_guard<std::mutex> lk(m_queue->mut);
        m_queue->q.push_back(task);
    }
    m_queue->cv.notify_all();
    {
        std::unique_lock<std::mutex> lk(task->mut);
        task->cv.wait(lk, [&] { return task->done; });
    }
    if (task->rc != MIBC_OK) throw std::runtime_error("mibc_call: " + task->error);
}

std::vector<DecodedChunk> HipCaller::call_chunks(size_t dims, const uint16_t *in, int8_t *out, int num_chunks) {
    return submit(dims, in, nullptr, out, num_chunks);
}

std::vector<DecodedChunk> HipCaller::call_chunks_i16(size_t dims, const int16_t *in, const float *ss, int8_t *out,
                                                     int num_chunks) {
    if (!ss) throw std::invalid_argument("call_chunks_i16: shift/scale pairs missing");
    return submit(dims, reinterpret_cast<const uint16_t *>(in), ss, out, num_chunks);
}

std::vector<DecodedChunk> HipCaller::call_chunks_var(size_t dims, const uint16_t *in, int8_t *out,
                                                     const std::vector<mibc_var_chunk> &chunks) {
    if (chunks.empty()) return {};
    auto task = std::make_shared<NNTask>();
    task->dims = dims;
    task->in = in;
    task->out = out;
    task->num_chunks = int(chunks.size());
    task->var = &chunks;
    run_task(task);
    const BatchDims &bd = m_dims.at(dims);
    const size_t N = size_t(bd.N), T = size_t(bd.T_out);
    const int stride = model_stride();
    std::vector<DecodedChunk> res(chunks.size());
    for (size_t i = 0; i < chunks.size(); ++i) {
        const size_t t0 = size_t(chunks[i].sample_start / stride), tc = size_t(chunks[i].n_samples / stride);
        const size_t base = size_t(chunks[i].row) * T + t0;
        const int8_t *mv = out + base, *sq = out + N * T + base, *qs = out + 2 * N * T + base;
        size_t nb = 0;
        for (size_t t = 0; t < tc; ++t) nb += size_t(mv[t]);
        res[i].moves.assign(mv, mv + tc);
        res[i].sequence.assign(reinterpret_cast<const char *>(sq), nb);
        res[i].qstring.assign(reinterpret_cast<const char *>(qs), nb);
    }
    return res;
}

std::vector<DecodedChunk> HipCaller::submit(size_t dims, const uint16_t *in, const float *ss, int8_t *out, int num_chunks) {
    if (num_chunks <= 0) return {};
    auto task = std::make_shared<NNTask>();
    task->dims = dims;
    task->in = in;
    task->ss = ss;
    task->out = out;
    task->num_chunks = num_chunks;
    run_task(task);
    // part 2: slice the [3][N][T] planes into strings (decode/CUDADecoder.cpp:115-173)
    const BatchDims &bd = m_dims.at(dims);
    const size_t N = size_t(bd.N), T = size_t(bd.T_out);
    std::vector<DecodedChunk> res(static_cast<size_t>(num_chunks));
    for (int i = 0; i < num_chunks; ++i) {
        const int8_t *mv = out + size_t(i) * T;
        const int8_t *sq = out + N * T + size_t(i) * T;
        const int8_t *qs = out + 2 * N * T + size_t(i) * T;
        size_t nb = 0;
        for (size_t t = 0; t < T; ++t) nb += size_t(mv[t]);
        res[size_t(i)].moves.assign(mv, mv + T);
        res[size_t(i)].sequence.assign(reinterpret_cast<const char *>(sq), nb);
        res[size_t(i)].qstring.assign(reinterpret_cast<const char *>(qs), nb);
    }
    return res;
}

// Quantile / med_mad parameters of whole reads on the device (mibc_scaler_stats).  Synchronous; uses the
// engine's stream, so it is serialised with the GPU thread by the engine mutex of this caller.
std::vector<std::pair<float, float>> HipCaller::scaler_stats(
        const std::vector<std::pair<const int16_t *, size_t>> &reads, const SignalNormalisationParams &p) {
    if (p.strategy == ScalingStrategy::PA)
        throw std::invalid_argument("scaler_stats: the PA strategy needs no pass over the samples");
    std::vector<int64_t> off(reads.size() + 1, 0);
    for (size_t i = 0; i < reads.size(); ++i) off[i + 1] = off[i] + int64_t(reads[i].second);
    const size_t total = size_t(off.back());
    std::vector<std::pair<float, float>> out(reads.size());
    if (reads.empty()) return out;
    std::lock_guard<std::mutex> lk(m_engine_mutex);
    void *d_sig = mibc_device_alloc(m_engine, std::max<size_t>(total * 2, 16));
    void *d_off = mibc_device_alloc(m_engine, off.size() * 8);
    void *d_ss = mibc_device_alloc(m_engine, reads.size() * 8);
    if (!d_sig || !d_off || !d_ss) throw std::runtime_error("scaler_stats: device allocation failed");
    int rc = MIBC_OK;
    for (size_t i = 0; i < reads.size() && rc == MIBC_OK; ++i)
        if (reads[i].second)
            rc = mibc_memcpy_h2d(m_engine, static_cast<char *>(d_sig) + off[i] * 2, reads[i].first, reads[i].second * 2);
    if (rc == MIBC_OK) rc = mibc_memcpy_h2d(m_engine, d_off, off.data(), off.size() * 8);
    const float params[4] = {p.quantile.quantile_a, p.quantile.quantile_b, p.quantile.shift_multiplier,
                             p.quantile.scale_multiplier};
    if (rc == MIBC_OK)
        rc = mibc_scaler_stats(m_engine, static_cast<const int16_t *>(d_sig), static_cast<const int64_t *>(d_off),
                               int(reads.size()),
                               p.strategy == ScalingStrategy::QUANTILE ? MIBC_SCALE_QUANTILE : MIBC_SCALE_MED_MAD,
                               params, static_cast<float *>(d_ss), nullptr);
    std::vector<float> ss(reads.size() * 2);
    if (rc == MIBC_OK) rc = mibc_memcpy_d2h(m_engine, ss.data(), d_ss, ss.size() * 4);
    mibc_device_free(m_engine, d_sig);
    mibc_device_free(m_engine, d_off);
    mibc_device_free(m_engine, d_ss);
    if (rc != MIBC_OK) throw std::runtime_error(std::string("mibc_scaler_stats: ") + mibc_last_error(m_engine));
    for (size_t i = 0; i < reads.size(); ++i) out[i] = {ss[2 * i], ss[2 * i + 1]};
    return out;
}

std::vector<std::vector<uint16_t>> HipCaller::scale_reads(const std::vector<std::pair<const int16_t *, size_t>> &reads,
                                                          const std::vector<std::pair<float, float>> &shift_scale) {
    if (shift_scale.size() != reads.size()) throw std::invalid_argument("scale_reads: one (shift, scale) pair per read");
    std::vector<int64_t> off(reads.size() + 1, 0);
    for (size_t i = 0; i < reads.size(); ++i) off[i + 1] = off[i] + int64_t(reads[i].second);
    const size_t total = size_t(off.back());
    std::vector<std::vector<uint16_t>> out(reads.size());
    if (reads.empty() || total == 0) return out;
    std::vector<float> ss(reads.size() * 2);
    for (size_t i = 0; i < reads.size(); ++i) {
        ss[2 * i] = shift_scale[i].first;
        ss[2 * i + 1] = shift_scale[i].second;
    }
    std::lock_guard<std::mutex> lk(m_engine_mutex);
    void *d_sig = mibc_device_alloc(m_engine, total * 2);
    void *d_out = mibc_device_alloc(m_engine, total * 2);
    void *d_off = mibc_device_alloc(m_engine, off.size() * 8);
    void *d_ss = mibc_device_alloc(m_engine, ss.size() * 4);
    int rc = (d_sig && d_out && d_off && d_ss) ? MIBC_OK : MIBC_ERR_MEM;
    for (size_t i = 0; i < reads.size() && rc == MIBC_OK; ++i)
        if (reads[i].second)
            rc = mibc_memcpy_h2d(m_engine, static_cast<char *>(d_sig) + off[i] * 2, reads[i].first, reads[i].second * 2);
    if (rc == MIBC_OK) rc = mibc_memcpy_h2d(m_engine, d_off, off.data(), off.size() * 8);
    if (rc == MIBC_OK) rc = mibc_memcpy_h2d(m_engine, d_ss, ss.data(), ss.size() * 4);
    if (rc == MIBC_OK)
        rc = mibc_scale_reads(m_engine, static_cast<const int16_t *>(d_sig), static_cast<const int64_t *>(d_off),
                              int(reads.size()), static_cast<const float *>(d_ss), static_cast<uint16_t *>(d_out));
    for (size_t i = 0; i < reads.size() && rc == MIBC_OK; ++i) {
        out[i].resize(reads[i].second);
        if (reads[i].second)
            rc = mibc_memcpy_d2h(m_engine, out[i].data(), static_cast<char *>(d_out) + off[i] * 2, reads[i].second * 2);
    }
    if (d_sig) mibc_device_free(m_engine, d_sig);
    if (d_out) mibc_device_free(m_engine, d_out);
    if (d_off) mibc_device_free(m_engine, d_off);
    if (d_ss) mibc_device_free(m_engine, d_ss);
    if (rc != MIBC_OK) throw std::runtime_error(std::string("mibc_scale_reads: ") + mibc_last_error(m_engine));
    return out;
}

void HipCaller::gpu_thread_fn() {
    // Two batches in flight (the reference gets the same overlap from its runners' own streams,
    // CudaCaller.cpp:645-719): a task is submitted to the engine as soon as one of the two slots is free — its H2D
    // copy runs beside the kernels of the batch in front of it, and the D2H copy + the runner's string slicing of
    // a finished batch run beside the kernels of the next one.  Completion is reported in submission order.  The two
    // batches may belong to different batch dimensions: the engine switches its geometry in stream order.
    using Clock = std::chrono::steady_clock;
    struct InFlight {
        std::shared_ptr<NNTask> task;
        int slot;
        int rc;
        bool waited;          // mibc_call_wait already consumed this slot's status (rc holds it)
        Clock::time_point t0;
    };
    DeviceQueue &dq = *m_queue;
    std::deque<InFlight> inflight;
    bool slot_busy[2] = {false, false};
    Clock::time_point last_done = Clock::now();
    auto finish = [&](InFlight &f, int rc) {
        // a batch is charged from its submission or the completion of the batch in front of it, whichever is later
        const auto now = Clock::now();
        m_model_decode_us += std::chrono::duration_cast<std::chrono::microseconds>(now - std::max(f.t0, last_done)).count();
        last_done = now;
        ++m_batches;
        {
            std::lock_guard<std::mutex> lk(f.task->mut);
            f.task->rc = rc;
            if (rc != MIBC_OK) f.task->error = mibc_last_error(m_engine);
            f.task->done = true;
        }
        f.task->cv.notify_one();
    };
    auto run_sync = [&](NNTask &t) {   // whole call on the engine's stream (the one synchronous retry of a failed batch)
        std::lock_guard<std::mutex> elk(m_engine_mutex);
        const BatchDims &bd = m_dims.at(t.dims);
        if (t.var)
            return mibc_call_var(m_engine, t.in, nullptr, bd.N, bd.T_in, t.var->data(), int(t.var->size()), &m_opts, t.out);
        return t.ss ? mibc_call_i16(m_engine, reinterpret_cast<const int16_t *>(t.in), t.ss, bd.N, bd.T_in, &m_opts, t.out)
                    : mibc_call(m_engine, t.in, bd.N, bd.T_in, &m_opts, t.out);
    };
    // with dq.mut held
    auto front_is_mine = [&] {
        return !dq.q.empty() && dq.q.front()->caller == this && (dq.owner == nullptr || dq.owner == this);
    };
    auto has_mine = [&] {
        for (auto &t : dq.q)
            if (t->caller == this) return true;
        return false;
    };
    while (true) {
        // 1. take new tasks while a slot is free
        {
            std::unique_lock<std::mutex> lk(dq.mut);
            if (inflight.empty()) {
                // nothing of this caller is on the device: hand the device to whoever is next in the FIFO before sleeping
                // (also on the way out: a stale owner would block every later caller of this device)
                if (dq.owner == this) {
                    dq.owner = nullptr;
                    dq.cv.notify_all();
                }
                dq.cv.wait(lk, [&] { return front_is_mine() || (m_terminate.load() && !has_mine()); });
                if (!front_is_mine()) return;   // terminate, and nothing of ours is queued
            }
            while (front_is_mine() && (!slot_busy[0] || !slot_busy[1])) {
                std::shared_ptr<NNTask> task = dq.q.front();
                dq.q.pop_front();
                dq.owner = this;
                lk.unlock();
                InFlight f{task, -1, MIBC_OK, false, Clock::now()};
                // the engine decodes all batch rows (stale rows included), the node uses the first n
                // (CudaCaller.cpp:269-270, BasecallerNode.cpp:185-189)
                {
                    // fixed and variable-chunk batches share the two slots (CudaModelRunner.cpp:21-49 runs both on the
                    // same stream pipeline): the chunk table of a variable batch travels in the slot's own pinned buffer
                    f.slot = slot_busy[0] ? 1 : 0;
                    const BatchDims &bd = m_dims.at(task->dims);
                    {
                        std::lock_guard<std::mutex> elk(m_engine_mutex);
                        f.rc = task->var ? mibc_call_var_async(m_engine, f.slot, task->in, nullptr, bd.N, bd.T_in, task->var->data(),
                                                               int(task->var->size()), &m_opts, task->out)
                                         : mibc_call_async(m_engine, f.slot, task->in, task->ss, bd.N, bd.T_in, &m_opts, task->out);
                    }
                    slot_busy[f.slot] = true;
                    inflight.push_back(std::move(f));
                }
                lk.lock();
            }
            if (inflight.empty()) continue;
        }
        // 2. oldest batch.  While the second slot is free and could be filled, poll (a task arriving meanwhile gets its
        //    copy started at once); otherwise block on the batch's completion event.
        InFlight &f = inflight.front();
        bool ready = (f.rc != MIBC_OK) || f.waited;
        if (!ready) {
            std::lock_guard<std::mutex> elk(m_engine_mutex);
            ready = mibc_call_poll(m_engine, f.slot) != 0;
        }
        if (!ready) {
            const bool slot_free = !slot_busy[0] || !slot_busy[1];
            bool could_take = false;
            if (slot_free) {
                std::unique_lock<std::mutex> lk(dq.mut);
                could_take = cv_wait_for(dq.cv, lk, std::chrono::microseconds(200), front_is_mine);
                if (!could_take) continue;   // keep polling
            }
            if (could_take) continue;
            // both slots busy: sleep on the event
            std::lock_guard<std::mutex> elk(m_engine_mutex);
            f.rc = mibc_call_wait(m_engine, f.slot);
            f.waited = true;
        }
        int rc = f.rc;
        if (rc == MIBC_OK && !f.waited) {
            std::lock_guard<std::mutex> elk(m_engine_mutex);
            rc = mibc_call_wait(m_engine, f.slot);
        }
        if (rc != MIBC_OK) {   // retry once, synchronously (:698-704) — after the other slot has drained
            if (inflight.size() > 1 && !inflight[1].waited && inflight[1].rc == MIBC_OK) {
                std::lock_guard<std::mutex> elk(m_engine_mutex);
                inflight[1].rc = mibc_call_wait(m_engine, inflight[1].slot);   // its own status: kept, not discarded
                inflight[1].waited = true;
            }
            rc = run_sync(*f.task);
        }
        slot_busy[f.slot] = false;
        finish(f, rc);
        inflight.pop_front();
    }
}

NamedStats HipCaller::sample_stats() const {  // CudaCaller.cpp:316-321
    return {{"batches_called", double(m_batches.load())},
            {"model_decode_ms", double(m_model_decode_us.load()) / 1000.0}};
}

// ------------------------------------------------------------------ ScalerNode, host half
std::optional<float> expected_open_pore_level(const std::string &code) {
    // read_pipeline/nodes/ScalerNode.cpp:112-139; lookup is case-insensitive (models/kits.cpp:17-30)
    std::string s = code;
    std::transform(s.begin(), s.end(), s.begin(), [](unsigned char c) { return char(std::toupper(c)); });
    static const std::map<std::string, float> table = {
            {"FLO-FLG114", 200.0f},   {"FLO-FLG114HD", 200.0f}, {"FLO-MIN004RA", 195.50f},
            {"FLO-PRO004RA", 194.97f}, {"FLO-MIN114", 197.61f},  {"FLO-MIN114HD", 197.61f},
            {"FLO-PRO114", 199.21f},  {"FLO-PRO114HD", 199.21f}, {"FLO-PRO114M", 199.21f},
    };
    const auto it = table.find(s);
    if (it == table.end()) return std::nullopt;
    return it->second;
}

ReadScaling finish_read_scaling(float shift, float scale, const ReadCalibration &cal) {
    ReadScaling r;
    r.shift = shift;
    r.scale = scale;
    r.scale_pa = cal.scaling * scale;                 // ScalerNode.cpp:226
    r.shift_pa = cal.scaling * (shift + cal.offset);  // :227
    return r;
}

ReadScaling pa_read_scaling(const SignalNormalisationParams &p, const ReadCalibration &cal) {
    if (p.strategy != ScalingStrategy::PA)
        throw std::invalid_argument("pa_read_scaling: strategy is data-driven, use HipCaller::scaler_stats");
    float scale, shift;
    if (p.standardisation.standardise) {  // ScalerNode.cpp:191-199
        scale = p.standardisation.stdev / cal.scaling;
        shift = (p.standardisation.mean / cal.scaling) - cal.offset;
    } else {
        scale = 1.f / cal.scaling;
        shift = -1.f * cal.offset;
    }
    ReadScaling r = finish_read_scaling(shift, scale, cal);
    if (!std::isnan(cal.open_pore_level)) {  // :205-213
        const auto expected = expected_open_pore_level(cal.flow_cell_product_code);
        if (expected.has_value() && *expected != 0) r.open_pore_adjustment = (cal.open_pore_level - *expected) / cal.scaling;
    }
    return r;
}

static float f16_bits_to_f32(uint16_t h) {
    const uint32_t sign = uint32_t(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu, m = h & 0x3ffu, bits;
    if (e == 0) {
        if (m == 0) {
            bits = sign;
        } else {
            int sh = 0;
            while (!(m & 0x400u)) {
                m <<= 1;
                ++sh;
            }
            bits = sign | uint32_t(127 - 15 - sh + 1) << 23 | (m & 0x3ffu) << 13;
        }
    } else if (e == 31) {
        bits = sign | 0x7f800000u | m << 13;
    } else {
        bits = sign | (e + 127 - 15) << 23 | m << 13;
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

int trim_signal(const uint16_t *sig, int n, float threshold, int window_size, int min_elements) {
    // torch_utils/trim.cpp:23-60
    const int min_trim = 10;
    const int num_samples = n - min_trim;
    const int num_windows = num_samples / window_size;
    bool seen_peak = false;
    for (int pos = 0; pos < num_windows; ++pos) {
        const int start = pos * window_size + min_trim;
        const int end = start + window_size;
        int cnt = 0;
        for (int i = start; i < end; ++i) cnt += f16_bits_to_f32(sig[i]) > threshold;
        if (cnt > min_elements || seen_peak) {
            seen_peak = true;
            if (f16_bits_to_f32(sig[end - 1]) > threshold) continue;
            return end >= num_samples ? min_trim : end;
        }
    }
    return min_trim;
}

int dna_trim_start(const SignalNormalisationParams &p, const uint16_t *scaled, size_t n_samples) {
    // ScalerNode.cpp:231-254 (DNA branch, no RNA adapter trim in front)
    int trim_start;
    if (p.standardisation.standardise) {
        trim_start = 10;
    } else {
        const int max_samples = std::min(8000, int(n_samples / 2));
        trim_start = trim_signal(scaled, max_samples);
    }
    return size_t(trim_start) < n_samples ? trim_start : 0;
}

int rna_adapter_pos(const int16_t *signal, int signal_len) {
    // ScalerNode.cpp:58-107.  The median of a window is at::median's: the LOWER middle element for even lengths.
    constexpr int kWindowSize = 250, kStride = 50;
    constexpr int kMedianDiff = 125, kMedianDiffForDiffOnlyCheck = 150, kMinMedianForRNASignal = 700;
    std::array<int16_t, 5> medians = {0, 0, 0, 0, 0};
    std::array<int32_t, 5> window_pos = {0, 0, 0, 0, 0};
    int median_pos = 0;
    const int signal_start = 1000;
    const int signal_end = 3 * signal_len / 4;
    std::vector<int16_t> w(kWindowSize);
    for (int i = signal_start; i < signal_end; i += kStride) {
        const int len = std::min(kWindowSize, signal_len - i);
        std::copy(signal + i, signal + i + len, w.begin());
        std::nth_element(w.begin(), w.begin() + (len - 1) / 2, w.begin() + len);
        medians[size_t(median_pos) % medians.size()] = w[size_t(len - 1) / 2];
        window_pos[size_t(median_pos) % window_pos.size()] = median_pos;
        // first smallest, LAST largest (std::minmax_element's tie rule, which the reference relies on)
        const auto mm = std::minmax_element(medians.begin(), medians.end());
        const int min_median = *mm.first, max_median = *mm.second;
        const auto min_at = size_t(mm.first - medians.begin()), max_at = size_t(mm.second - medians.begin());
        if (median_pos >= int(medians.size()) && window_pos[max_at] > window_pos[min_at] &&
            ((max_median > kMinMedianForRNASignal && max_median - min_median > kMedianDiff) ||
             max_median - min_median > kMedianDiffForDiffOnlyCheck))
            return i;
        ++median_pos;
    }
    return 0;
}

RnaTrim rna_trim(const int16_t *raw, size_t n_samples, bool has_rna_based_adapters) {
    // ScalerNode.cpp:157-184
    RnaTrim r;
    if (has_rna_based_adapters) return r;
    const int pos = rna_adapter_pos(raw, int(n_samples));
    if (size_t(pos) < n_samples) {
        r.trim_start = pos;
        r.rna_adapter_end_signal_pos = 0;
    } else {
        r.trim_start = 0;
        r.rna_adapter_end_signal_pos = pos;
    }
    return r;
}

ScaledRead scaler_node(HipCaller &caller, const SignalNormalisationParams &p, bool is_rna_model, bool has_rna_based_adapters,
                       const int16_t *raw, size_t n, const ReadCalibration &cal, bool want_signal) {
    ScalerOps ops;
    ops.stats = [&caller](const int16_t *x, size_t len, const SignalNormalisationParams &sp) {
        return caller.scaler_stats({{x, len}}, sp)[0];
    };
    ops.scale = [&caller](const int16_t *x, size_t len, float shift, float scale) {
        return std::move(caller.scale_reads({{x, len}}, {{shift, scale}})[0]);
    };
    return scaler_node(ops, p, is_rna_model, has_rna_based_adapters, raw, n, cal, want_signal);
}

ScaledRead scaler_node(const ScalerOps &ops, const SignalNormalisationParams &p, bool is_rna_model, bool has_rna_based_adapters,
                       const int16_t *raw, size_t n, const ReadCalibration &cal, bool want_signal) {
    if (!raw || n == 0) throw std::invalid_argument("scaler_node: empty read");   // (the reference's at::median throws on it too)
    ScaledRead r;
    int trim_start = 0;
    if (is_rna_model) {   // ScalerNode.cpp:157-184: trim the adapter of RNA reads first, before scaling
        const RnaTrim t = rna_trim(raw, n, has_rna_based_adapters);
        trim_start = t.trim_start;
        r.rna_adapter_end_signal_pos = t.rna_adapter_end_signal_pos;
        raw += trim_start;
        n -= size_t(trim_start);
    }
    if (p.strategy == ScalingStrategy::PA) {   // :190-215
        r.scaling = pa_read_scaling(p, cal);
    } else {                                   // :216-224: the statistics ignore an un-cut RNA adapter
        const size_t skip = std::min(size_t(r.rna_adapter_end_signal_pos), n);
        const auto ss = ops.stats(raw + skip, n - skip, p);
        r.scaling = finish_read_scaling(ss.first, ss.second, cal);
    }
    // :228-229 the sample map, on the device; without want_signal only the prefix the DNA trim looks at
    const bool dna_heuristic = !is_rna_model && trim_start == 0 && !p.standardisation.standardise;
    const size_t n_scaled = want_signal ? n : dna_heuristic ? size_t(std::min(8000, int(n / 2))) : 0;
    std::vector<std::vector<uint16_t>> scaled(1);
    if (n_scaled) scaled[0] = ops.scale(raw, n_scaled, r.scaling.device_shift(), r.scaling.scale);
    if (!is_rna_model) {   // :233-254: no DNA trimming on RNA
        if (trim_start == 0 && p.standardisation.standardise) {
            trim_start = 10;
        } else if (trim_start == 0) {
            trim_start = trim_signal(scaled[0].data(), std::min(8000, int(n / 2)));
        }
        if (size_t(trim_start) < n) {
            if (want_signal) scaled[0].erase(scaled[0].begin(), scaled[0].begin() + trim_start);
            r.first_sample = size_t(trim_start);
        } else {
            trim_start = 0;
        }
    } else {
        r.first_sample = size_t(trim_start);
    }
    r.num_trimmed_samples = trim_start;   // :256
    if (want_signal) r.signal_f16 = std::move(scaled[0]);
    return r;
}

std::vector<ScaledRead> scaler_node(HipCaller &caller, const SignalNormalisationParams &p, bool is_rna_model,
                                    const std::vector<ScalerInput> &reads, bool want_signal) {
    const size_t R = reads.size();
    std::vector<ScaledRead> out(R);
    std::vector<const int16_t *> base(R);     // the read behind the RNA adapter cut
    std::vector<size_t> len(R);
    std::vector<int> trim(R, 0);
    for (size_t r = 0; r < R; ++r) {
        if (!reads[r].raw || reads[r].n_samples == 0) throw std::invalid_argument("scaler_node: empty read");
        base[r] = reads[r].raw;
        len[r] = reads[r].n_samples;
        if (is_rna_model) {
            const RnaTrim t = rna_trim(reads[r].raw, reads[r].n_samples, reads[r].has_rna_based_adapters);
            trim[r] = t.trim_start;
            out[r].rna_adapter_end_signal_pos = t.rna_adapter_end_signal_pos;
            base[r] += t.trim_start;
            len[r] -= size_t(t.trim_start);
        }
    }
    if (p.strategy == ScalingStrategy::PA) {
        for (size_t r = 0; r < R; ++r) out[r].scaling = pa_read_scaling(p, reads[r].cal);
    } else {   // one launch over all reads
        std::vector<std::pair<const int16_t *, size_t>> spans(R);
        for (size_t r = 0; r < R; ++r) {
            const size_t skip = std::min(size_t(out[r].rna_adapter_end_signal_pos), len[r]);
            spans[r] = {base[r] + skip, len[r] - skip};
        }
        const auto ss = caller.scaler_stats(spans, p);
        for (size_t r = 0; r < R; ++r) out[r].scaling = finish_read_scaling(ss[r].first, ss[r].second, reads[r].cal);
    }
    // the sample map: whole reads (want_signal) or the prefixes the DNA trim looks at, one launch
    std::vector<std::pair<const int16_t *, size_t>> spans;
    std::vector<std::pair<float, float>> pairs;
    std::vector<size_t> slot(R, size_t(-1));
    for (size_t r = 0; r < R; ++r) {
        const bool dna_heuristic = !is_rna_model && !p.standardisation.standardise;
        const size_t n_scaled = want_signal ? len[r] : dna_heuristic ? size_t(std::min(8000, int(len[r] / 2))) : 0;
        if (n_scaled == 0) continue;
        slot[r] = spans.size();
        spans.push_back({base[r], n_scaled});
        pairs.push_back({out[r].scaling.device_shift(), out[r].scaling.scale});
    }
    auto scaled = caller.scale_reads(spans, pairs);
    static const std::vector<uint16_t> none;
    for (size_t r = 0; r < R; ++r) {
        std::vector<uint16_t> *sig = slot[r] != size_t(-1) ? &scaled[slot[r]] : nullptr;
        int t = trim[r];
        if (!is_rna_model) {
            if (p.standardisation.standardise) t = 10;
            else t = trim_signal(sig ? sig->data() : none.data(), std::min(8000, int(len[r] / 2)));
            if (size_t(t) < len[r]) {
                if (want_signal && sig) sig->erase(sig->begin(), sig->begin() + t);
                out[r].first_sample = size_t(t);
            } else {
                t = 0;
            }
        } else {
            out[r].first_sample = size_t(t);
        }
        out[r].num_trimmed_samples = t;
        if (want_signal && sig) out[r].signal_f16 = std::move(*sig);
    }
    return out;
}

// ------------------------------------------------------------------ HipModelRunner
static std::atomic<int> g_runner_id{0};

HipModelRunner::HipModelRunner(std::shared_ptr<HipCaller> caller, size_t batch_dims_idx)
        : m_caller(std::move(caller)), m_dims(batch_dims_idx), m_id(g_runner_id++) {
    const size_t N = rows();
    m_in = static_cast<uint16_t *>(mibc_host_alloc(N * chunk_size() * 2));
    m_out = static_cast<int8_t *>(mibc_host_alloc(3 * N * size_t(m_caller->output_steps(m_dims))));
    m_ss = static_cast<float *>(mibc_host_alloc(N * 2 * sizeof(float)));
    if (!m_in || !m_out || !m_ss) throw std::runtime_error("mibc_host_alloc failed");
    for (size_t i = 0; i < N; ++i) {
        m_ss[2 * i] = 0.0f;
        m_ss[2 * i + 1] = 1.0f;
    }
    std::memset(m_in, 0, N * chunk_size() * 2);
}

HipModelRunner::~HipModelRunner() {
    mibc_host_free(m_in);
    mibc_host_free(m_out);
    mibc_host_free(m_ss);
}

void HipModelRunner::RowPacker::reset(size_t rows, size_t chunk_size, size_t gap) {
    m_rows = rows;
    m_cs = chunk_size;
    m_gap = gap;
    m_used = 0;
    m_leaf0 = 1;
    while (m_leaf0 < rows) m_leaf0 <<= 1;
    m_fill.assign(rows, 0);
    m_free.assign(2 * m_leaf0, -1);
    for (size_t r = 0; r < rows; ++r) m_free[m_leaf0 + r] = int(chunk_size);
    for (size_t i = m_leaf0 - 1; i >= 1; --i) m_free[i] = std::max(m_free[2 * i], m_free[2 * i + 1]);
}

bool HipModelRunner::RowPacker::place(size_t n, int &row, int &start) {
    if (m_rows == 0 || m_free[1] < int(n)) return false;
    size_t i = 1;
    while (i < m_leaf0) i = (m_free[2 * i] >= int(n)) ? 2 * i : 2 * i + 1;   // leftmost row with room: first fit
    const size_t r = i - m_leaf0;
    if (m_fill[r] == 0) ++m_used;
    start = m_fill[r];
    row = int(r);
    m_fill[r] = int(size_t(start) + n + m_gap);                  // the next chunk of this row starts behind the gap
    m_free[i] = std::max(0, int(m_cs) - m_fill[r]);
    for (i >>= 1; i >= 1; i >>= 1) m_free[i] = std::max(m_free[2 * i], m_free[2 * i + 1]);
    return true;
}

size_t HipModelRunner::batch_size() const {
    const size_t N = rows();
    if (!variable_chunk_sizes()) return N;
    size_t b = size_t(double(N) * double(m_caller->variable_batch_fill()));
    // BasecallerNode fills whole 32-row spans and flushes on size only when chunks_size == batch_size * (cs / stride + 2)
    // exactly (:303-305, 421-426): always a multiple of 32, at least one span (rows() is a multiple of 32 for every LSTM width:
    // the engine's granularity is 32 or 64)
    b = std::max<size_t>(32, b / 32 * 32);
    return std::min(b, N);
}

void HipModelRunner::accept_chunk(int idx, const uint16_t *f16, size_t n) {
    if (m_mode == 2) throw std::runtime_error("accept_chunk: this batch already holds raw int16 chunks");
    if (variable_chunk_sizes()) {
        // CudaModelRunner::accept_chunk concatenates the chunks into one flat span (CudaModelRunner.cpp:21-32); here they
        // are packed into the batch ROWS, first-fit, 2 output steps between the chunks of a row (the engine's rows are
        // independent sequences: a chunk cannot continue on the next row)
        const size_t cs = chunk_size(), stride = size_t(m_caller->model_stride()), gap = 2 * stride;
        if (n == 0 || n > cs || n % stride != 0)
            throw std::runtime_error("accept_chunk: a variable chunk must be a stride multiple of at most chunk_size samples");
        m_mode = 1;
        if (m_var_table.empty() && m_var_overflow.empty()) m_packer.reset(rows(), cs, gap);
        if (m_var_overflow.empty()) {
            int row = -1, start = 0;
            if (m_packer.place(n, row, start)) {
                std::memcpy(m_in + size_t(row) * cs + size_t(start), f16, n * 2);
                m_var_table.push_back({row, start, int(n)});
                return;
            }
        }
        // no row has room although the node's budget (batch_size() rows' worth of len / stride + 2 steps) was not used up:
        // a chunk cannot straddle two rows here.  Kept aside and called in a second engine batch by call_chunks (counted:
        // var_overflow_batches); to keep the result order trivial everything accepted from now on follows it.
        m_var_overflow.push_back({0, 0, int(n)});
        m_var_overflow_data.emplace_back(f16, f16 + n);
        return;
    }
    if (idx < 0 || idx >= int(rows()) || n != chunk_size())
        throw std::runtime_error("accept_chunk: bad index or chunk length");
    m_mode = 1;
    std::memcpy(m_in + size_t(idx) * n, f16, n * 2);
}

void HipModelRunner::accept_chunk_i16(int idx, const int16_t *raw, size_t n, float shift, float scale) {
    if (idx < 0 || idx >= int(rows()) || n != chunk_size())
        throw std::runtime_error("accept_chunk_i16: bad index or chunk length");
    if (m_mode == 1) throw std::runtime_error("accept_chunk_i16: this batch already holds scaled f16 chunks");
    m_mode = 2;
    std::memcpy(m_in + size_t(idx) * n, raw, n * 2);
    m_ss[2 * idx] = shift;
    m_ss[2 * idx + 1] = scale;
}

std::vector<DecodedChunk> HipModelRunner::call_chunks(int num_chunks) {
    ++m_batches;
    const int mode = m_mode;
    m_mode = 0;
    if (variable_chunk_sizes()) {
        if (size_t(num_chunks) != m_var_table.size() + m_var_overflow.size())
            throw std::runtime_error("call_chunks: num_chunks differs from the number of accepted chunks");
        std::vector<DecodedChunk> res;
        while (true) {
            auto part = m_caller->call_chunks_var(m_dims, m_in, m_out, m_var_table);
            for (auto &d : part) res.push_back(std::move(d));
            ++m_var_batches;
            m_var_rows_used += int64_t(m_packer.rows_used());
            m_var_table.clear();
            if (m_var_overflow.empty()) break;
            // second engine batch for what did not fit (see accept_chunk and batch_size())
            ++m_var_overflow_batches;
            std::vector<std::vector<uint16_t>> data;
            data.swap(m_var_overflow_data);
            m_var_overflow.clear();
            for (auto &d : data) accept_chunk(0, d.data(), d.size());
            m_mode = 0;
        }
        return res;
    }
    if (mode == 2) return m_caller->call_chunks_i16(m_dims, reinterpret_cast<const int16_t *>(m_in), m_ss, m_out, num_chunks);
    return m_caller->call_chunks(m_dims, m_in, m_out, num_chunks);
}

std::vector<DecodedChunk> HipModelRunner::call_chunks_var(const std::vector<mibc_var_chunk> &chunks) {
    ++m_batches;
    m_mode = 0;
    return m_caller->call_chunks_var(m_dims, m_in, m_out, chunks);
}

std::string HipModelRunner::get_name() const {  // unique per instance (CudaModelRunner.cpp:61-67)
    return "HipModelRunner_" + std::to_string(m_id) + "_hip:" + std::to_string(m_caller->device());
}

NamedStats HipModelRunner::sample_stats() const {
    NamedStats s = m_caller->sample_stats();
    s["runner_batches_called"] = double(m_batches.load());
    if (variable_chunk_sizes()) {
        s["var_engine_batches"] = double(m_var_batches.load());
        s["var_overflow_batches"] = double(m_var_overflow_batches.load());
        s["var_rows_used"] = double(m_var_rows_used.load());
    }
    return s;
}

int model_stride(const mibc_model_desc &d) {
    int stride = 1;
    for (int i = 0; i < d.n_convs; ++i) stride *= d.conv_stride[i];
    if (d.tx_d_model > 0 && d.up_scale_factor > 1) stride /= d.up_scale_factor;
    return stride;
}

int chunk_size_granularity(const mibc_model_desc &d) {
    const bool tx = d.tx_d_model > 0;
    const int stride_inner = model_stride(d) * (tx && d.up_scale_factor > 1 ? d.up_scale_factor : 1);
    return stride_inner * (tx ? 16 : 1);
}

std::vector<int> simplex_chunk_sizes(const mibc_model_desc &d, int requested_chunk_size, int overlap) {
    const int gran = chunk_size_granularity(d);
    const int min_chunk = (overlap + 1 + gran - 1) / gran * gran;   // utils::pad_to(overlap + 1, granularity)
    auto norm = [&](int x) { return std::max(min_chunk, (x / gran) * gran); };
    std::set<int> sizes{norm(requested_chunk_size)};
    for (float fraction : {0.5f}) sizes.insert(norm(int(float(requested_chunk_size) * fraction)));
    return std::vector<int>(sizes.rbegin(), sizes.rend());
}

size_t get_chunk_queue_idx(const std::vector<size_t> &chunk_sizes, size_t read_raw_size) {
    size_t best_idx = 0;
    for (size_t i = 1; i < chunk_sizes.size(); ++i) {
        const size_t best_size = chunk_sizes[best_idx], this_size = chunk_sizes[i];
        if ((best_size < read_raw_size && best_size < this_size) || (read_raw_size < this_size && this_size < best_size))
            best_idx = i;
    }
    return best_idx;
}

std::vector<std::vector<RunnerPtr>> create_basecall_runners(const mibc_model_desc &desc,
                                                            const float *const *weights, int n_weights,
                                                            const std::string &device_string, int num_runners,
                                                            const std::vector<int> &chunk_sizes, int batch_size,
                                                            const mibc_decode_opts &opts, const CallerParams &params) {
    std::vector<int> ids;
    std::string err;
    if (!try_parse_device_ids(device_string, size_t(mibc_device_count()), ids, err)) throw std::runtime_error(err);
    if (ids.empty()) throw std::runtime_error("no GPU device in '" + device_string + "' (the HIP engine has no CPU fallback)");
    if (chunk_sizes.empty()) throw std::invalid_argument("create_basecall_runners: no chunk size");
    // ONE caller per device, built concurrently (api/runner_creation.cpp:85-113: one pool thread per device)
    std::vector<std::shared_ptr<HipCaller>> callers(ids.size());
    std::vector<std::thread> pool;
    std::exception_ptr first;
    std::mutex emut;
    for (size_t di = 0; di < ids.size(); ++di)
        pool.emplace_back([&, di] {
            try {
                callers[di] = std::make_shared<HipCaller>(desc, weights, n_weights, ids[di], chunk_sizes, batch_size, opts, params);
            } catch (...) {
                std::lock_guard<std::mutex> lk(emut);
                if (!first) first = std::current_exception();
            }
        });
    for (auto &t : pool) t.join();
    if (first) std::rethrow_exception(first);
    // [device][runner][batch dims] (api/runner_creation.cpp:115-123); the caller orders its batch dimensions largest
    // chunk size first, the runner list follows the caller's order
    std::vector<std::vector<RunnerPtr>> out;
    for (size_t di = 0; di < ids.size(); ++di) {
        std::vector<RunnerPtr> rs;
        for (int r = 0; r < num_runners; ++r)
            for (size_t bd = 0; bd < callers[di]->num_batch_dims(); ++bd)
                rs.push_back(std::make_unique<HipModelRunner>(callers[di], bd));
        out.push_back(std::move(rs));
    }
    return out;
}

std::vector<std::vector<RunnerPtr>> create_basecall_runners(const mibc_model_desc &desc,
                                                            const float *const *weights, int n_weights,
                                                            const std::string &device_string, int num_runners,
                                                            int chunk_size, int batch_size,
                                                            const mibc_decode_opts &opts) {
    return create_basecall_runners(desc, weights, n_weights, device_string, num_runners, std::vector<int>{chunk_size},
                                   batch_size, opts);
}

// ------------------------------------------------------------------ SimplexBasecaller
namespace {
// One thread per runner.  An exception must not leave a std::thread (std::terminate would take the whole
// host process down on a transient HIP / allocation error): the first one is kept, the other workers stop
// at their next batch boundary, and it is rethrown on the calling thread after join().
template <typename Worker>
void run_workers(std::vector<RunnerPtr> &runners, Worker &&worker) {
    std::atomic<bool> failed{false};
    std::exception_ptr first;
    std::mutex emut;
    std::vector<std::thread> threads;
    for (auto &r : runners)
        threads.emplace_back([&, runner = r.get()] {
            try {
                worker(runner, failed);
            } catch (...) {
                std::lock_guard<std::mutex> lk(emut);
                if (!first) first = std::current_exception();
                failed.store(true);
            }
        });
    for (auto &t : threads) t.join();
    if (first) std::rethrow_exception(first);
}
}  // namespace

SimplexBasecaller::SimplexBasecaller(std::vector<RunnerPtr> runners, int overlap, int model_stride)
        : m_runners(std::move(runners)), m_overlap(overlap), m_stride(model_stride) {
    // the runner list is [devices][runners][chunk_sizes]: the chunk sizes repeat, so collect until the first one
    // shows up again (BasecallerNode.cpp:494-501)
    for (auto &r : m_runners) {
        if (!m_chunk_sizes.empty() && r->chunk_size() == m_chunk_sizes[0]) break;
        m_chunk_sizes.push_back(r->chunk_size());
    }
}

std::vector<CalledRead> SimplexBasecaller::basecall(const std::vector<std::vector<uint16_t>> &reads) {
    std::vector<ReadView> v;
    for (const auto &r : reads) v.push_back({r.data(), r.size(), false, 0.0f, 1.0f});
    return basecall_views(v);
}

size_t SimplexBasecaller::basecall_repeated(const uint16_t *data, size_t n_distinct, size_t read_len, size_t n_reads,
                                            double *seconds_to_last_read) {
    std::vector<ReadView> v;
    v.reserve(n_reads);
    for (size_t i = 0; i < n_reads; ++i) v.push_back({data + (i % n_distinct) * read_len, read_len, false, 0.0f, 1.0f});
    size_t bases = 0;
    // timed: chunking of every read, batching, PCIe both ways, the engine, slicing, stitching — until the last read is
    // called.  Not timed: releasing the called reads (in a pipeline they leave one by one to the writer while the GPU
    // works on later reads; here 10^5..10^6 of them would be freed in one serial burst at the end).
    const auto t0 = std::chrono::steady_clock::now();
    auto called = basecall_views(v);
    if (seconds_to_last_read)
        *seconds_to_last_read = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (const auto &r : called) bases += r.seq.size();
    return bases;
}

std::vector<CalledRead> SimplexBasecaller::basecall_raw(const std::vector<RawRead> &reads) {
    std::vector<ReadView> v;
    for (const auto &r : reads)
        v.push_back({reinterpret_cast<const uint16_t *>(r.signal), r.n_samples, true, r.shift, r.scale});
    return basecall_views(v);
}

std::vector<CalledRead> SimplexBasecaller::basecall_views(const std::vector<ReadView> &reads) {
    struct Work {
        size_t read, idx, offset;
    };
    const size_t nq = m_chunk_sizes.size();
    std::vector<CalledRead> out(reads.size());
    std::vector<std::vector<Chunk>> chunks(reads.size());
    std::vector<std::deque<Work>> queues(nq);   // one chunk queue per chunk size (BasecallerNode.cpp:515-522)
    {
        // chunking of all reads up front (the reference chunks a read when it arrives, on the node's input thread); for
        // hundreds of thousands of reads the per-read allocations are worth a few threads, merged in read order
        const size_t nthr = reads.size() >= 4096 ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
        std::vector<std::vector<std::vector<Work>>> local(nthr, std::vector<std::vector<Work>>(nq));
        auto chunk_range = [&](size_t t) {
            const size_t r0 = reads.size() * t / nthr, r1 = reads.size() * (t + 1) / nthr;
            for (size_t r = r0; r < r1; ++r) {
                // a read goes to the queue with the smallest chunk size that fits it whole, else the largest (:81-94, :125)
                const size_t qi = get_chunk_queue_idx(m_chunk_sizes, reads[r].n);
                const size_t cs = m_chunk_sizes[qi];
                out[r].chunk_offsets = generate_chunks(reads[r].n, cs, size_t(m_stride), size_t(m_overlap));
                chunks[r].resize(out[r].chunk_offsets.size());
                for (size_t i = 0; i < out[r].chunk_offsets.size(); ++i) {
                    chunks[r][i].input_offset = out[r].chunk_offsets[i];
                    chunks[r][i].raw_chunk_size = cs;
                    local[t][qi].push_back({r, i, out[r].chunk_offsets[i]});
                }
            }
        };
        // an exception (bad_alloc) inside a std::thread would terminate the process: catch, join, rethrow
        std::exception_ptr first_error;
        std::mutex err_mut;
        auto guarded = [&](size_t t) {
            try {
                chunk_range(t);
            } catch (...) {
                std::lock_guard<std::mutex> lk(err_mut);
                if (!first_error) first_error = std::current_exception();
            }
        };
        std::vector<std::thread> thr;
        for (size_t t = 1; t < nthr; ++t) thr.emplace_back(guarded, t);
        guarded(0);
        for (auto &th : thr) th.join();
        if (first_error) std::rethrow_exception(first_error);
        for (size_t t = 0; t < nthr; ++t)
            for (size_t qi = 0; qi < nq; ++qi) queues[qi].insert(queues[qi].end(), local[t][qi].begin(), local[t][qi].end());
    }
    std::mutex qmut;
    // a read is stitched by the worker that delivers its last chunk (the reference's stitch threads run beside the
    // basecalling workers too: BasecallerNode.cpp:205-287, 535-548), not in a serial pass at the end
    std::vector<std::atomic<int>> remaining(reads.size());
    for (size_t r = 0; r < reads.size(); ++r) remaining[r].store(int(chunks[r].size()));
    auto stitch_read = [&](size_t r) {
        std::vector<const Chunk *> cc;
        for (auto &c : chunks[r]) cc.push_back(&c);
        StitchedRead st = stitch_chunks(cc, reads[r].n, m_stride);
        out[r].seq = std::move(st.seq);
        out[r].qstring = std::move(st.qstring);
        out[r].moves = std::move(st.moves);
        std::vector<Chunk>().swap(chunks[r]);
        m_samples_processed += int64_t(reads[r].n);
    };
    auto worker = [&](ModelRunnerBase *runner, std::atomic<bool> &failed) {
        const size_t batch = runner->batch_size();
        const size_t chunk_size = runner->chunk_size();
        // a worker serves the queue of its runner's chunk size (BasecallerNode.cpp:300-301: worker_id % num queues,
        // which is the same thing for a [devices][runners][chunk_sizes] runner list)
        size_t qi = 0;
        while (qi + 1 < nq && m_chunk_sizes[qi] != chunk_size) ++qi;
        std::deque<Work> &queue = queues[qi];
        std::vector<uint16_t> padded(chunk_size);
        while (!failed.load()) {
            std::vector<Work> mine;
            {
                std::lock_guard<std::mutex> lk(qmut);
                while (!queue.empty() && mine.size() < batch) {
                    mine.push_back(queue.front());
                    queue.pop_front();
                }
            }
            if (mine.empty()) return;
            for (size_t k = 0; k < mine.size(); ++k) {
                const ReadView &sig = reads[mine[k].read];
                const size_t avail = std::min(chunk_size, sig.n - mine[k].offset);
                const uint16_t *src = sig.data + mine[k].offset;
                if (avail != chunk_size) {  // repeat-pad non-full chunks (BasecallerNode.cpp:432-440)
                    for (size_t p = 0; p < chunk_size; ++p) padded[p] = src[p % avail];
                    src = padded.data();
                }
                if (sig.raw) {
                    auto *hip = dynamic_cast<HipModelRunner *>(runner);
                    if (!hip) throw std::runtime_error("raw int16 reads need a HipModelRunner");
                    hip->accept_chunk_i16(int(k), reinterpret_cast<const int16_t *>(src), chunk_size, sig.shift, sig.scale);
                } else {
                    runner->accept_chunk(int(k), src, chunk_size);
                }
            }
            auto decoded = runner->call_chunks(int(mine.size()));
            ++m_batches;
            if (mine.size() < batch) ++m_partial_batches;
            m_samples_incl_padding += int64_t(mine.size() * chunk_size);
            for (size_t k = 0; k < mine.size(); ++k) {
                Chunk &c = chunks[mine[k].read][mine[k].idx];
                c.seq = std::move(decoded[k].sequence);
                c.qstring = std::move(decoded[k].qstring);
                c.moves = std::move(decoded[k].moves);
                if (remaining[mine[k].read].fetch_sub(1) == 1) stitch_read(mine[k].read);
            }
        }
    };
    run_workers(m_runners, worker);
    return out;
}

std::vector<CalledRead> SimplexBasecaller::basecall_variable(const std::vector<std::vector<uint16_t>> &reads) {
    const size_t chunk_size = m_runners.at(0)->chunk_size(), stride = size_t(m_stride);
    const size_t gap = 2 * stride;
    struct Work {
        size_t read, idx, offset, len;   // len = raw interval length (before the stride top-up)
    };
    std::vector<CalledRead> out(reads.size());
    std::vector<std::vector<Chunk>> chunks(reads.size());
    std::deque<Work> queue;
    for (size_t r = 0; r < reads.size(); ++r) {
        const auto iv = generate_variable_chunks(reads[r].size(), chunk_size, stride, size_t(m_overlap));
        chunks[r].resize(iv.size());
        for (size_t i = 0; i < iv.size(); ++i) {
            out[r].chunk_offsets.push_back(iv[i].first);
            chunks[r][i].input_offset = iv[i].first;
            chunks[r][i].raw_chunk_size = iv[i].second - iv[i].first;
            queue.push_back({r, i, iv[i].first, iv[i].second - iv[i].first});
        }
    }
    std::mutex qmut;
    std::vector<std::atomic<int>> remaining(reads.size());
    for (size_t r = 0; r < reads.size(); ++r) remaining[r].store(int(chunks[r].size()));
    auto stitch_read = [&](size_t r) {
        std::vector<const Chunk *> cc;
        for (auto &c : chunks[r]) cc.push_back(&c);
        StitchedRead st = stitch_chunks(cc, reads[r].size(), m_stride);
        out[r].seq = std::move(st.seq);
        out[r].qstring = std::move(st.qstring);
        out[r].moves = std::move(st.moves);
        std::vector<Chunk>().swap(chunks[r]);
        m_samples_processed += int64_t(reads[r].size());
    };
    auto worker = [&](ModelRunnerBase *base, std::atomic<bool> &failed) {
        auto *runner = dynamic_cast<HipModelRunner *>(base);
        if (!runner) throw std::runtime_error("variable chunk sizes need a HipModelRunner");
        const size_t batch = runner->rows();   // this path packs the rows itself: all of them are its budget
        while (!failed.load()) {
            // fill the rows first-fit (RowPacker) in queue order until a chunk finds no row (one lock per batch)
            std::vector<Work> mine;
            std::vector<mibc_var_chunk> table;
            {
                std::lock_guard<std::mutex> lk(qmut);
                HipModelRunner::RowPacker pk;
                pk.reset(batch, chunk_size, gap);
                while (!queue.empty()) {
                    const Work w = queue.front();
                    const size_t padded = (w.len + stride - 1) / stride * stride;   // BasecallerNode.cpp:408-416
                    int row = -1, start = 0;
                    if (!pk.place(padded, row, start)) break;
                    queue.pop_front();
                    mine.push_back(w);
                    table.push_back({row, start, int(padded)});
                }
            }
            if (mine.empty()) return;
            // (no clearing of the pinned rows: samples outside every chunk are ignored by the engine's sample bitmap)
            for (size_t k = 0; k < mine.size(); ++k) {
                const uint16_t *src = reads[mine[k].read].data() + mine[k].offset;
                uint16_t *dst = runner->batch_row(table[k].row) + table[k].sample_start;
                for (size_t p = 0; p < size_t(table[k].n_samples); ++p) dst[p] = src[p % mine[k].len];
            }
            // the chunks of a row appear in the table in ascending start order (first fit appends to a row)
            auto decoded = runner->call_chunks_var(table);
            ++m_batches;
            m_samples_incl_padding += int64_t(batch * chunk_size);
            for (size_t k = 0; k < mine.size(); ++k) {
                Chunk &c = chunks[mine[k].read][mine[k].idx];
                c.seq = std::move(decoded[k].sequence);
                c.qstring = std::move(decoded[k].qstring);
                c.moves = std::move(decoded[k].moves);
                // the worker that delivers a read's last chunk stitches it (as basecall() does): no serial pass at the end
                if (remaining[mine[k].read].fetch_sub(1) == 1) stitch_read(mine[k].read);
            }
        }
    };
    run_workers(m_runners, worker);
    return out;
}

NamedStats SimplexBasecaller::sample_stats() const {  // BasecallerNode.cpp:597-616
    return {{"samples_processed", double(m_samples_processed.load())},
            {"samples_incl_padding", double(m_samples_incl_padding.load())},
            {"batches_called", double(m_batches.load())},
            {"partial_batches_called", double(m_partial_batches.load())}};
}

}  // namespace dorado_amd::host

// ------------------------------------------------------------------ C test/driver entry points
using namespace dorado_amd::host;
static thread_local std::string g_herr;

namespace {
struct ReadOutputs {
    char *seq_out, *qstr_out;
    int64_t *seq_len_out;
    uint8_t *moves_out;
    int64_t *moves_len_out, *offsets_out, *n_offsets_out;
    double *stats4;
};
void write_outputs(const std::vector<CalledRead> &called, SimplexBasecaller &node, const ReadOutputs &o) {
    size_t so = 0, mo = 0, oo = 0;
    for (size_t r = 0; r < called.size(); ++r) {
        const auto &c = called[r];
        std::memcpy(o.seq_out + so, c.seq.data(), c.seq.size());
        std::memcpy(o.qstr_out + so, c.qstring.data(), c.qstring.size());
        so += c.seq.size();
        o.seq_len_out[r] = int64_t(c.seq.size());
        std::memcpy(o.moves_out + mo, c.moves.data(), c.moves.size());
        mo += c.moves.size();
        o.moves_len_out[r] = int64_t(c.moves.size());
        for (size_t off : c.chunk_offsets) o.offsets_out[oo++] = int64_t(off);
        o.n_offsets_out[r] = int64_t(c.chunk_offsets.size());
    }
    auto st = node.sample_stats();
    o.stats4[0] = st["samples_processed"];
    o.stats4[1] = st["samples_incl_padding"];
    o.stats4[2] = st["batches_called"];
    o.stats4[3] = st["partial_batches_called"];
}
std::unique_ptr<SimplexBasecaller> make_node(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                                             const char *device_string, int num_runners, int chunk_size, int overlap,
                                             int batch_size, const mibc_decode_opts *opts, bool extra_chunk_sizes = false) {
    // extra_chunk_sizes: the reference's high-throughput simplex set {chunk, 0.5 x chunk} (CudaCaller.cpp:388-413)
    const std::vector<int> sizes = extra_chunk_sizes ? simplex_chunk_sizes(*desc, chunk_size, overlap)
                                                     : std::vector<int>{chunk_size};
    auto per_dev = create_basecall_runners(*desc, weights, n_weights, device_string, num_runners, sizes, batch_size,
                                           *opts);
    std::vector<RunnerPtr> flat;
    for (auto &d : per_dev)
        for (auto &r : d) flat.push_back(std::move(r));
    return std::make_unique<SimplexBasecaller>(std::move(flat), overlap, model_stride(*desc));
}
}  // namespace

extern "C" {

const char *mibch_last_error(void) { return g_herr.c_str(); }

// ---- ScalerNode host half, for the Python tests ----
// strategy PA: out = {shift, scale, open_pore_adjustment, scale_pa, shift_pa}
int mibch_pa_read_scaling(int standardise, float mean, float stdev, float scaling, float offset,
                          float open_pore_level, const char *flow_cell_product_code, float *out5) {
    try {
        SignalNormalisationParams p;
        p.strategy = ScalingStrategy::PA;
        p.standardisation = {standardise != 0, mean, stdev};
        ReadCalibration cal{scaling, offset, open_pore_level, flow_cell_product_code ? flow_cell_product_code : ""};
        const ReadScaling r = pa_read_scaling(p, cal);
        out5[0] = r.shift; out5[1] = r.scale; out5[2] = r.open_pore_adjustment; out5[3] = r.scale_pa; out5[4] = r.shift_pa;
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}
int mibch_trim_signal(const uint16_t *f16, int n, float threshold, int window_size, int min_elements) {
    return trim_signal(f16, n, threshold, window_size, min_elements);
}
int mibch_rna_adapter_pos(const int16_t *raw, int n) { return rna_adapter_pos(raw, n); }
// out2 = {trim_start, rna_adapter_end_signal_pos}
int mibch_rna_trim(const int16_t *raw, uint64_t n, int has_rna_based_adapters, int *out2) {
    const RnaTrim r = rna_trim(raw, size_t(n), has_rna_based_adapters != 0);
    out2[0] = r.trim_start;
    out2[1] = r.rna_adapter_end_signal_pos;
    return 0;
}
// One read through scaler_node (the tests' entry; builds a caller for `desc`).  strategy 0 MED_MAD, 1 QUANTILE, 2 PA
// (config::ScalingStrategy order); params7 = {quantile_a, quantile_b, shift_multiplier, scale_multiplier, standardise, mean,
// stdev}; cal3 = {scaling, offset, open_pore_level}.  out_f16 (capacity n, may be null) / out_n: the scaled, trimmed signal;
// out_f5 = {shift, scale, open_pore_adjustment, scale_pa, shift_pa}; out_i3 = {num_trimmed_samples, rna_adapter_end_signal_pos,
// first_sample}.
static SignalNormalisationParams scaler_params(int strategy, const float *params7) {
    SignalNormalisationParams p;
    p.strategy = strategy == 0 ? ScalingStrategy::MED_MAD : strategy == 1 ? ScalingStrategy::QUANTILE : ScalingStrategy::PA;
    p.quantile = {params7[0], params7[1], params7[2], params7[3]};
    p.standardisation = {params7[4] != 0.0f, params7[5], params7[6]};
    return p;
}
static void scaler_outputs(const ScaledRead &r, uint64_t n, uint16_t *out_f16, uint64_t *out_n, float *out_f5, int *out_i3) {
    if (out_f16 && !r.signal_f16.empty()) std::memcpy(out_f16, r.signal_f16.data(), r.signal_f16.size() * 2);
    if (out_n) *out_n = out_f16 ? r.signal_f16.size() : n - r.first_sample;
    out_f5[0] = r.scaling.shift; out_f5[1] = r.scaling.scale; out_f5[2] = r.scaling.open_pore_adjustment;
    out_f5[3] = r.scaling.scale_pa; out_f5[4] = r.scaling.shift_pa;
    out_i3[0] = r.num_trimmed_samples; out_i3[1] = r.rna_adapter_end_signal_pos; out_i3[2] = int(r.first_sample);
}
int mibch_scaler_node(const mibc_model_desc *desc, const float *const *weights, int n_weights, const char *device_string,
                      int strategy, const float *params7, int is_rna_model, int has_rna_based_adapters, const int16_t *raw,
                      uint64_t n, const float *cal3, const char *flow_cell_product_code, uint16_t *out_f16, uint64_t *out_n,
                      float *out_f5, int *out_i3) {
    try {
        const mibc_decode_opts opts{32, 100.0f, 2.0f, 0.0f, 1.0f};
        std::vector<int> ids;
        std::string err;
        if (!try_parse_device_ids(device_string ? device_string : "hip:0", size_t(mibc_device_count()), ids, err))
            throw std::runtime_error(err);
        if (ids.empty()) throw std::runtime_error("mibch_scaler_node: no GPU device (the HIP engine has no CPU fallback)");
        HipCaller caller(*desc, weights, n_weights, ids[0], 64 * model_stride(*desc), 32, opts);
        ReadCalibration cal{cal3[0], cal3[1], cal3[2], flow_cell_product_code ? flow_cell_product_code : ""};
        const ScaledRead r = scaler_node(caller, scaler_params(strategy, params7), is_rna_model != 0, has_rna_based_adapters != 0,
                                         raw, size_t(n), cal, out_f16 != nullptr);
        scaler_outputs(r, n, out_f16, out_n, out_f5, out_i3);
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}
// The same orchestration with the caller's own sample passes (ScalerOps): stats_fn(x, n, strategy, params4, out shift_scale[2]),
// scale_fn(x, n, shift, scale, out f16[n]).  No device involved.
typedef void (*mibch_scaler_stats_fn)(const int16_t *, uint64_t, int, const float *, float *);
typedef void (*mibch_scaler_scale_fn)(const int16_t *, uint64_t, float, float, uint16_t *);
int mibch_scaler_node_ops(mibch_scaler_stats_fn stats_fn, mibch_scaler_scale_fn scale_fn, int strategy, const float *params7,
                          int is_rna_model, int has_rna_based_adapters, const int16_t *raw, uint64_t n, const float *cal3,
                          const char *flow_cell_product_code, uint16_t *out_f16, uint64_t *out_n, float *out_f5, int *out_i3) {
    try {
        ScalerOps ops;
        ops.stats = [&](const int16_t *x, size_t len, const SignalNormalisationParams &sp) {
            const float p4[4] = {sp.quantile.quantile_a, sp.quantile.quantile_b, sp.quantile.shift_multiplier,
                                 sp.quantile.scale_multiplier};
            float ss[2] = {0.0f, 1.0f};
            stats_fn(x, len, sp.strategy == ScalingStrategy::MED_MAD ? 0 : 1, p4, ss);
            return std::make_pair(ss[0], ss[1]);
        };
        ops.scale = [&](const int16_t *x, size_t len, float shift, float scale) {
            std::vector<uint16_t> out(len);
            scale_fn(x, len, shift, scale, out.data());
            return out;
        };
        ReadCalibration cal{cal3[0], cal3[1], cal3[2], flow_cell_product_code ? flow_cell_product_code : ""};
        const ScaledRead r = scaler_node(ops, scaler_params(strategy, params7), is_rna_model != 0, has_rna_based_adapters != 0,
                                         raw, size_t(n), cal, out_f16 != nullptr);
        scaler_outputs(r, n, out_f16, out_n, out_f5, out_i3);
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}
int mibch_dna_trim_start(int standardise, const uint16_t *f16, uint64_t n) {
    SignalNormalisationParams p;
    p.standardisation.standardise = standardise != 0;
    return dna_trim_start(p, f16, size_t(n));
}

// Same as mibch_basecall_reads but with variable chunk sizes (SimplexBasecaller::basecall_variable).
int mibch_basecall_reads_variable(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                                  const char *device_string, int num_runners, int chunk_size, int overlap,
                                  int batch_size, const mibc_decode_opts *opts, const uint16_t *signals,
                                  const int64_t *read_len, int n_reads, char *seq_out, char *qstr_out,
                                  int64_t *seq_len_out, uint8_t *moves_out, int64_t *moves_len_out,
                                  int64_t *offsets_out, int64_t *n_offsets_out, double *stats4) {
    try {
        auto node = make_node(desc, weights, n_weights, device_string, num_runners, chunk_size, overlap, batch_size, opts);
        std::vector<std::vector<uint16_t>> reads(static_cast<size_t>(n_reads));
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            reads[size_t(r)].assign(signals + pos, signals + pos + read_len[r]);
            pos += size_t(read_len[r]);
        }
        auto called = node->basecall_variable(reads);
        write_outputs(called, *node, {seq_out, qstr_out, seq_len_out, moves_out, moves_len_out, offsets_out,
                                      n_offsets_out, stats4});
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// Throughput of the variable-chunk host path (bench.py through_host_variable): reads of the given lengths (cycled over
// n_distinct signals) through SimplexBasecaller::basecall_variable — generate_variable_chunks, first-fit row packing,
// mibc_call_var_async with two batches in flight per device, slicing, stitching.  n_warm reads first (untimed).
// out8 = {read samples/s, seconds, engine batches, bases, batch rows x chunk size per second, devices, 0, 0}.
// variable = 0: the SAME read set through the fixed-chunk path (SimplexBasecaller::basecall: generate_chunks, one chunk per batch
// row, short chunks repeat-padded) — the comparison that says whether variable chunk sizes pay on this engine.
int mibch_bench_through_host_mixed(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                                   const char *device_string, int num_runners, int chunk_size, int overlap, int batch_size,
                                   const mibc_decode_opts *opts, const uint16_t *signals, int n_distinct, int64_t sig_len,
                                   const int64_t *read_len, int64_t n_warm, int64_t n_reads, int variable, double *out8);
int mibch_bench_through_host_variable(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                                      const char *device_string, int num_runners, int chunk_size, int overlap, int batch_size,
                                      const mibc_decode_opts *opts, const uint16_t *signals, int n_distinct, int64_t sig_len,
                                      const int64_t *read_len, int64_t n_warm, int64_t n_reads, double *out8) {
    return mibch_bench_through_host_mixed(desc, weights, n_weights, device_string, num_runners, chunk_size, overlap, batch_size, opts,
                                          signals, n_distinct, sig_len, read_len, n_warm, n_reads, 1, out8);
}
int mibch_bench_through_host_mixed(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                                   const char *device_string, int num_runners, int chunk_size, int overlap, int batch_size,
                                   const mibc_decode_opts *opts, const uint16_t *signals, int n_distinct, int64_t sig_len,
                                   const int64_t *read_len, int64_t n_warm, int64_t n_reads, int variable, double *out8) {
    try {
        auto node = make_node(desc, weights, n_weights, device_string, num_runners, chunk_size, overlap, batch_size, opts);
        auto make_reads = [&](int64_t first, int64_t count) {
            std::vector<std::vector<uint16_t>> reads(static_cast<size_t>(count));
            for (int64_t r = 0; r < count; ++r) {
                const int64_t L = std::min<int64_t>(read_len[first + r], sig_len);
                const uint16_t *src = signals + size_t((first + r) % n_distinct) * size_t(sig_len);
                reads[size_t(r)].assign(src, src + L);
            }
            return reads;
        };
        if (n_warm > 0) (void)(variable ? node->basecall_variable(make_reads(0, n_warm)) : node->basecall(make_reads(0, n_warm)));
        auto reads = make_reads(n_warm, n_reads);
        double total = 0;
        for (auto &r : reads) total += double(r.size());
        auto st0 = node->sample_stats();
        const auto t0 = std::chrono::steady_clock::now();
        auto called = variable ? node->basecall_variable(reads) : node->basecall(reads);
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        auto st1 = node->sample_stats();
        size_t bases = 0;
        for (auto &c : called) bases += c.seq.size();
        std::vector<int> ids;
        std::string err;
        (void)try_parse_device_ids(device_string, size_t(mibc_device_count()), ids, err);
        out8[0] = total / sec;
        out8[1] = sec;
        out8[2] = st1["batches_called"] - st0["batches_called"];
        out8[3] = double(bases);
        out8[4] = (st1["samples_incl_padding"] - st0["samples_incl_padding"]) / sec;
        out8[5] = double(ids.size());
        out8[6] = double(variable);
        out8[7] = 0.0;
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// Whole raw reads (int16) with per-read (shift, scale, trim_start): ScalerNode's host half decides the
// parameters, the samples are scaled on the device.  Same outputs as mibch_basecall_reads.
int mibch_basecall_raw_reads(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                             const char *device_string, int num_runners, int chunk_size, int overlap,
                             int batch_size, const mibc_decode_opts *opts, const int16_t *signals,
                             const int64_t *read_len, const float *shift_scale, const int64_t *trim_start,
                             int n_reads, char *seq_out, char *qstr_out, int64_t *seq_len_out,
                             uint8_t *moves_out, int64_t *moves_len_out, int64_t *offsets_out,
                             int64_t *n_offsets_out, double *stats4) {
    try {
        auto node = make_node(desc, weights, n_weights, device_string, num_runners, chunk_size, overlap, batch_size, opts);
        std::vector<SimplexBasecaller::RawRead> reads;
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            const size_t ts = size_t(trim_start[r]);
            reads.push_back({signals + pos + ts, size_t(read_len[r]) - ts, shift_scale[2 * r], shift_scale[2 * r + 1]});
            pos += size_t(read_len[r]);
        }
        auto called = node->basecall_raw(reads);
        write_outputs(called, *node, {seq_out, qstr_out, seq_len_out, moves_out, moves_len_out, offsets_out,
                                      n_offsets_out, stats4});
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// Throughput of the whole host path (bench.py --through-host): create_basecall_runners + SimplexBasecaller on
// synthetic f16 reads held in host memory: chunking, pinned batch assembly, H2D, network + decode, D2H, string
// slicing and stitching, `num_runners` runners per device and batch dimension with two batches in flight per device.
// device_string may name several devices ("hip:all"): one process, one HipCaller per device, shared chunk queues.
// n_warm reads are called first (untimed), then n_reads are timed.  two_queues: the reference's extra 0.5x chunk queue.
// out8 = {read samples/s (what BasecallerNode counts as samples_processed: overlaps are called twice but counted once),
//         seconds, batches, bases, samples incl. overlap and padding per second (batch rows x chunk size), devices,
//         partial batches, 0}.
int mibch_bench_through_host(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                             const char *device_string, int num_runners, int chunk_size, int overlap, int batch_size,
                             const mibc_decode_opts *opts, const uint16_t *signals, int n_distinct, int64_t read_len,
                             int64_t n_warm, int64_t n_reads, int two_queues, double *out8) {
    try {
        auto node = make_node(desc, weights, n_weights, device_string, num_runners, chunk_size, overlap, batch_size, opts,
                              two_queues != 0);
        if (n_warm > 0) (void)node->basecall_repeated(signals, size_t(n_distinct), size_t(read_len), size_t(n_warm));
        auto st0 = node->sample_stats();
        double sec = 0.0;
        const size_t bases = node->basecall_repeated(signals, size_t(n_distinct), size_t(read_len), size_t(n_reads), &sec);
        auto st1 = node->sample_stats();
        std::vector<int> ids;
        std::string err;
        (void)try_parse_device_ids(device_string, size_t(mibc_device_count()), ids, err);
        out8[0] = double(n_reads) * double(read_len) / sec;
        out8[1] = sec;
        out8[2] = st1["batches_called"] - st0["batches_called"];
        out8[3] = double(bases);
        out8[4] = (st1["samples_incl_padding"] - st0["samples_incl_padding"]) / sec;
        out8[5] = double(ids.size());
        out8[6] = st1["partial_batches_called"] - st0["partial_batches_called"];
        out8[7] = 0.0;
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// mibch_basecall_reads with the reference's extra 0.5x chunk-size queue; sizes_out (up to 4) gets the chunk sizes.
int mibch_basecall_reads_two_queues(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                                    const char *device_string, int num_runners, int chunk_size, int overlap,
                                    int batch_size, const mibc_decode_opts *opts, const uint16_t *signals,
                                    const int64_t *read_len, int n_reads, char *seq_out, char *qstr_out,
                                    int64_t *seq_len_out, uint8_t *moves_out, int64_t *moves_len_out,
                                    int64_t *offsets_out, int64_t *n_offsets_out, double *stats4, int *sizes_out) {
    try {
        auto node = make_node(desc, weights, n_weights, device_string, num_runners, chunk_size, overlap, batch_size, opts,
                              true);
        const auto sizes = simplex_chunk_sizes(*desc, chunk_size, overlap);
        for (size_t i = 0; i < 4; ++i) sizes_out[i] = i < sizes.size() ? sizes[i] : 0;
        std::vector<std::vector<uint16_t>> reads(static_cast<size_t>(n_reads));
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            reads[size_t(r)].assign(signals + pos, signals + pos + read_len[r]);
            pos += size_t(read_len[r]);
        }
        auto called = node->basecall(reads);
        write_outputs(called, *node, {seq_out, qstr_out, seq_len_out, moves_out, moves_len_out, offsets_out,
                                      n_offsets_out, stats4});
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// Batch size a HipCaller picks: mode 0 = the known knee, -1 = the reference's timing sweep (CudaCaller.cpp:552-627).
// timings_out: up to max_t (batch, ms per chunk) pairs of the sweep.
int mibch_auto_batch_size(const mibc_model_desc *desc, const float *const *weights, int n_weights, int device,
                          int chunk_size, int mode, const mibc_decode_opts *opts, int *chosen, double *timings_out,
                          int max_t, int *n_t) {
    try {
        HipCaller c(*desc, weights, n_weights, device, chunk_size, mode, *opts);   // mode -1: timing sweep
        *chosen = c.batch_size();
        const auto &t = c.batch_timings();
        *n_t = int(t.size());
        for (size_t i = 0; i < t.size() && int(i) < max_t; ++i) {
            timings_out[2 * i] = double(t[i].first);
            timings_out[2 * i + 1] = t[i].second;
        }
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// CPU-only helpers for the tests
int mibch_simplex_chunk_sizes(const mibc_model_desc *desc, int requested, int overlap, int *out, int max) {
    const auto v = simplex_chunk_sizes(*desc, requested, overlap);
    for (size_t i = 0; i < v.size() && int(i) < max; ++i) out[i] = v[i];
    return int(v.size());
}
int mibch_get_chunk_queue_idx(const uint64_t *sizes, int n, uint64_t read_raw_size) {
    std::vector<size_t> v(sizes, sizes + n);
    return int(get_chunk_queue_idx(v, size_t(read_raw_size)));
}
int mibch_model_stride(const mibc_model_desc *desc) { return model_stride(*desc); }

long mibch_generate_variable_chunks(uint64_t num_samples, uint64_t chunk_size, uint64_t stride, uint64_t overlap,
                                    uint64_t *out_pairs, long max_out) {
    try {
        auto v = generate_variable_chunks(num_samples, chunk_size, stride, overlap);
        for (size_t i = 0; i < v.size() && long(i) < max_out; ++i) {
            out_pairs[2 * i] = v[i].first;
            out_pairs[2 * i + 1] = v[i].second;
        }
        return long(v.size());
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

long mibch_generate_chunks(uint64_t num_samples, uint64_t chunk_size, uint64_t stride, uint64_t overlap,
                           uint64_t *out, long max_out) {
    try {
        auto v = generate_chunks(num_samples, chunk_size, stride, overlap);
        for (size_t i = 0; i < v.size() && long(i) < max_out; ++i) out[i] = v[i];
        return long(v.size());
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// returns 1 ok / 0 failure; ids_out gets up to max ids, *n_out their count
int mibch_parse_device_ids(const char *s, uint64_t num_devices, int *ids_out, int max, int *n_out) {
    std::vector<int> ids;
    std::string err;
    const bool ok = try_parse_device_ids(s, size_t(num_devices), ids, err);
    g_herr = err;
    *n_out = int(ids.size());
    for (size_t i = 0; i < ids.size() && int(i) < max; ++i) ids_out[i] = ids[i];
    return ok ? 1 : 0;
}

long mibch_stitch_chunks(int n_chunks, const int64_t *input_offset, const int64_t *raw_chunk_size,
                         const uint8_t *moves, const int64_t *moves_off, const int64_t *moves_len,
                         const char *seq, const char *qstr, const int64_t *seq_off, const int64_t *seq_len,
                         int64_t raw_samples, int model_stride, char *seq_out, char *qstr_out,
                         uint8_t *moves_out, int64_t *n_moves_out) {
    try {
        std::vector<Chunk> cs(static_cast<size_t>(n_chunks));
        std::vector<const Chunk *> cp;
        for (int i = 0; i < n_chunks; ++i) {
            cs[size_t(i)].input_offset = size_t(input_offset[i]);
            cs[size_t(i)].raw_chunk_size = size_t(raw_chunk_size[i]);
            cs[size_t(i)].moves.assign(moves + moves_off[i], moves + moves_off[i] + moves_len[i]);
            cs[size_t(i)].seq.assign(seq + seq_off[i], size_t(seq_len[i]));
            cs[size_t(i)].qstring.assign(qstr + seq_off[i], size_t(seq_len[i]));
            cp.push_back(&cs[size_t(i)]);
        }
        StitchedRead r = stitch_chunks(cp, size_t(raw_samples), model_stride);
        std::memcpy(seq_out, r.seq.data(), r.seq.size());
        std::memcpy(qstr_out, r.qstring.data(), r.qstring.size());
        std::memcpy(moves_out, r.moves.data(), r.moves.size());
        *n_moves_out = int64_t(r.moves.size());
        return long(r.seq.size());
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

// Whole reads through create_basecall_runners + SimplexBasecaller.  signals: concatenated f16
// reads; outputs: concatenated seq/qstr/moves with per-read lengths; chunk offsets concatenated.
int mibch_basecall_reads(const mibc_model_desc *desc, const float *const *weights, int n_weights,
                         const char *device_string, int num_runners, int chunk_size, int overlap,
                         int batch_size, const mibc_decode_opts *opts, const uint16_t *signals,
                         const int64_t *read_len, int n_reads, char *seq_out, char *qstr_out,
                         int64_t *seq_len_out, uint8_t *moves_out, int64_t *moves_len_out,
                         int64_t *offsets_out, int64_t *n_offsets_out, double *stats4) {
    try {
        auto node = make_node(desc, weights, n_weights, device_string, num_runners, chunk_size, overlap, batch_size, opts);
        std::vector<std::vector<uint16_t>> reads(static_cast<size_t>(n_reads));
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            reads[size_t(r)].assign(signals + pos, signals + pos + read_len[r]);
            pos += size_t(read_len[r]);
        }
        auto called = node->basecall(reads);
        write_outputs(called, *node, {seq_out, qstr_out, seq_len_out, moves_out, moves_len_out, offsets_out,
                                      n_offsets_out, stats4});
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}


// test hook (CPU, no device): place `n` chunk lengths (samples) with HipModelRunner's packer into `rows` batch rows of
// `chunk_size` samples, `gap` samples between the chunks of a row; out_row / out_start per chunk (-1 = did not fit).
int mibch_debug_pack_rows(const int *lens, int n, int rows, int chunk_size, int gap, int *out_row, int *out_start) {
    HipModelRunner::RowPacker pk;
    pk.reset(size_t(rows), size_t(chunk_size), size_t(gap));
    int placed = 0;
    bool overflow = false;
    for (int i = 0; i < n; ++i) {
        int row = -1, start = 0;
        // as in accept_chunk: once one chunk has overflowed, everything behind it follows it (result order)
        if (!overflow && pk.place(size_t(lens[i]), row, start)) {
            out_row[i] = row;
            out_start[i] = start;
            ++placed;
        } else {
            overflow = true;
            out_row[i] = -1;
            out_start[i] = -1;
        }
    }
    return placed;
}
}  // extern "C"
