// dorado_amd/host/mibc_host.cpp — see mibc_host.h for what mirrors what in the reference.
#include "mibc_host.h"

#include <algorithm>
#include <chrono>
#include <cstring>
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
HipCaller::HipCaller(const mibc_model_desc &desc, const float *const *weights, int n_weights, int device,
                     int chunk_size, int batch_size, const mibc_decode_opts &opts)
        : m_desc(desc), m_opts(opts), m_device(device), m_chunk_size(chunk_size) {
    const int rc = mibc_create(device, &desc, weights, n_weights, &m_engine);
    if (rc != MIBC_OK) throw std::runtime_error(std::string("mibc_create: ") + mibc_last_error(nullptr));
    const int g = mibc_batch_granularity(m_engine);
    m_batch_size = (batch_size + g - 1) / g * g;
    m_T = mibc_output_steps(m_engine, chunk_size);
    if (mibc_reserve(m_engine, m_batch_size, chunk_size) != MIBC_OK) {
        const std::string msg = mibc_last_error(m_engine);
        mibc_destroy(m_engine);
        throw std::runtime_error("mibc_reserve: " + msg);
    }
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
    m_cv.notify_all();
    if (m_thread.joinable()) m_thread.join();
}

void HipCaller::restart() {  // CudaCaller.cpp:283-287
    if (m_terminate.load()) start_thread();
}

std::vector<DecodedChunk> HipCaller::call_chunks(const uint16_t *in, int8_t *out, int num_chunks) {
    if (num_chunks <= 0) return {};
    auto task = std::make_shared<NNTask>();
    task->in = in;
    task->out = out;
    task->num_chunks = num_chunks;
    {
        std::lock_guard<std::mutex> lk(m_mutex);
        m_queue.push_front(task);
    }
    m_cv.notify_one();
    {
        std::unique_lock<std::mutex> lk(task->mut);
        task->cv.wait(lk, [&] { return task->done; });
    }
    if (task->rc != MIBC_OK) throw std::runtime_error(std::string("mibc_call: ") + mibc_last_error(m_engine));
    // part 2: slice the [3][N][T] planes into strings (decode/CUDADecoder.cpp:115-173)
    const size_t N = size_t(m_batch_size), T = size_t(m_T);
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

void HipCaller::gpu_thread_fn() {
    while (true) {
        std::shared_ptr<NNTask> task;
        {
            std::unique_lock<std::mutex> lk(m_mutex);
            m_cv.wait(lk, [&] { return m_terminate.load() || !m_queue.empty(); });
            if (m_queue.empty()) return;  // terminate and drained
            task = m_queue.back();
            m_queue.pop_back();
        }
        const auto t0 = std::chrono::steady_clock::now();
        // the engine decodes all batch rows (stale rows included), the node uses the first n
        // (CudaCaller.cpp:269-270, BasecallerNode.cpp:185-189)
        int rc = mibc_call(m_engine, task->in, m_batch_size, m_chunk_size, &m_opts, task->out);
        if (rc != MIBC_OK) rc = mibc_call(m_engine, task->in, m_batch_size, m_chunk_size, &m_opts, task->out);  // retry once (:698-704)
        m_model_decode_us += std::chrono::duration_cast<std::chrono::microseconds>(
                                     std::chrono::steady_clock::now() - t0).count();
        ++m_batches;
        {
            std::lock_guard<std::mutex> lk(task->mut);
            task->rc = rc;
            task->done = true;
        }
        task->cv.notify_one();
    }
}

NamedStats HipCaller::sample_stats() const {  // CudaCaller.cpp:316-321
    return {{"batches_called", double(m_batches.load())},
            {"model_decode_ms", double(m_model_decode_us.load()) / 1000.0}};
}

// ------------------------------------------------------------------ HipModelRunner
static std::atomic<int> g_runner_id{0};

HipModelRunner::HipModelRunner(std::shared_ptr<HipCaller> caller) : m_caller(std::move(caller)), m_id(g_runner_id++) {
    const size_t N = size_t(m_caller->batch_size());
    m_in = static_cast<uint16_t *>(mibc_host_alloc(N * size_t(m_caller->chunk_size()) * 2));
    m_out = static_cast<int8_t *>(mibc_host_alloc(3 * N * size_t(m_caller->output_steps())));
    if (!m_in || !m_out) throw std::runtime_error("mibc_host_alloc failed");
    std::memset(m_in, 0, N * size_t(m_caller->chunk_size()) * 2);
}

HipModelRunner::~HipModelRunner() {
    mibc_host_free(m_in);
    mibc_host_free(m_out);
}

void HipModelRunner::accept_chunk(int idx, const uint16_t *f16, size_t n) {
    if (idx < 0 || idx >= m_caller->batch_size() || n != size_t(m_caller->chunk_size()))
        throw std::runtime_error("accept_chunk: bad index or chunk length");
    std::memcpy(m_in + size_t(idx) * n, f16, n * 2);
}

std::vector<DecodedChunk> HipModelRunner::call_chunks(int num_chunks) {
    ++m_batches;
    return m_caller->call_chunks(m_in, m_out, num_chunks);
}

std::string HipModelRunner::get_name() const {  // unique per instance (CudaModelRunner.cpp:61-67)
    return "HipModelRunner_" + std::to_string(m_id) + "_hip:" + std::to_string(m_caller->device());
}

NamedStats HipModelRunner::sample_stats() const {
    NamedStats s = m_caller->sample_stats();
    s["runner_batches_called"] = double(m_batches.load());
    return s;
}

std::vector<std::vector<RunnerPtr>> create_basecall_runners(const mibc_model_desc &desc,
                                                            const float *const *weights, int n_weights,
                                                            const std::string &device_string, int num_runners,
                                                            int chunk_size, int batch_size,
                                                            const mibc_decode_opts &opts) {
    std::vector<int> ids;
    std::string err;
    if (!try_parse_device_ids(device_string, size_t(mibc_device_count()), ids, err)) throw std::runtime_error(err);
    if (ids.empty()) throw std::runtime_error("no GPU device in '" + device_string + "' (the HIP engine has no CPU fallback)");
    std::vector<std::vector<RunnerPtr>> out;
    for (int id : ids) {
        auto caller = std::make_shared<HipCaller>(desc, weights, n_weights, id, chunk_size, batch_size, opts);
        std::vector<RunnerPtr> rs;
        for (int r = 0; r < num_runners; ++r) rs.push_back(std::make_unique<HipModelRunner>(caller));
        out.push_back(std::move(rs));
    }
    return out;
}

// ------------------------------------------------------------------ SimplexBasecaller
SimplexBasecaller::SimplexBasecaller(std::vector<RunnerPtr> runners, int overlap, int model_stride)
        : m_runners(std::move(runners)), m_overlap(overlap), m_stride(model_stride) {}

std::vector<CalledRead> SimplexBasecaller::basecall(const std::vector<std::vector<uint16_t>> &reads) {
    struct Work {
        size_t read, idx, offset;
    };
    const size_t chunk_size = m_runners.at(0)->chunk_size();
    std::vector<CalledRead> out(reads.size());
    std::vector<std::vector<Chunk>> chunks(reads.size());
    std::deque<Work> queue;
    for (size_t r = 0; r < reads.size(); ++r) {
        out[r].chunk_offsets = generate_chunks(reads[r].size(), chunk_size, size_t(m_stride), size_t(m_overlap));
        chunks[r].resize(out[r].chunk_offsets.size());
        for (size_t i = 0; i < out[r].chunk_offsets.size(); ++i) {
            chunks[r][i].input_offset = out[r].chunk_offsets[i];
            chunks[r][i].raw_chunk_size = chunk_size;
            queue.push_back({r, i, out[r].chunk_offsets[i]});
        }
    }
    std::mutex qmut;
    auto worker = [&](ModelRunnerBase *runner) {
        const size_t batch = runner->batch_size();
        std::vector<uint16_t> padded(chunk_size);
        while (true) {
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
                const auto &sig = reads[mine[k].read];
                const size_t avail = std::min(chunk_size, sig.size() - mine[k].offset);
                const uint16_t *src = sig.data() + mine[k].offset;
                if (avail == chunk_size) {
                    runner->accept_chunk(int(k), src, chunk_size);
                } else {  // repeat-pad non-full chunks (BasecallerNode.cpp:432-440)
                    for (size_t p = 0; p < chunk_size; ++p) padded[p] = src[p % avail];
                    runner->accept_chunk(int(k), padded.data(), chunk_size);
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
            }
        }
    };
    std::vector<std::thread> threads;
    for (auto &r : m_runners) threads.emplace_back(worker, r.get());
    for (auto &t : threads) t.join();
    for (size_t r = 0; r < reads.size(); ++r) {
        std::vector<const Chunk *> cc;
        for (auto &c : chunks[r]) cc.push_back(&c);
        StitchedRead s = stitch_chunks(cc, reads[r].size(), m_stride);
        out[r].seq = std::move(s.seq);
        out[r].qstring = std::move(s.qstring);
        out[r].moves = std::move(s.moves);
        m_samples_processed += int64_t(reads[r].size());
    }
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

extern "C" {

const char *mibch_last_error(void) { return g_herr.c_str(); }

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
        int stride = 1;
        for (int i = 0; i < desc->n_convs; ++i) stride *= desc->conv_stride[i];
        auto per_dev = create_basecall_runners(*desc, weights, n_weights, device_string, num_runners,
                                               chunk_size, batch_size, *opts);
        std::vector<RunnerPtr> flat;
        for (auto &d : per_dev)
            for (auto &r : d) flat.push_back(std::move(r));
        SimplexBasecaller node(std::move(flat), overlap, stride);
        std::vector<std::vector<uint16_t>> reads(static_cast<size_t>(n_reads));
        size_t pos = 0;
        for (int r = 0; r < n_reads; ++r) {
            reads[size_t(r)].assign(signals + pos, signals + pos + read_len[r]);
            pos += size_t(read_len[r]);
        }
        auto called = node.basecall(reads);
        size_t so = 0, mo = 0, oo = 0;
        for (int r = 0; r < n_reads; ++r) {
            const auto &c = called[size_t(r)];
            std::memcpy(seq_out + so, c.seq.data(), c.seq.size());
            std::memcpy(qstr_out + so, c.qstring.data(), c.qstring.size());
            so += c.seq.size();
            seq_len_out[r] = int64_t(c.seq.size());
            std::memcpy(moves_out + mo, c.moves.data(), c.moves.size());
            mo += c.moves.size();
            moves_len_out[r] = int64_t(c.moves.size());
            for (size_t o : c.chunk_offsets) offsets_out[oo++] = int64_t(o);
            n_offsets_out[r] = int64_t(c.chunk_offsets.size());
        }
        auto st = node.sample_stats();
        stats4[0] = st["samples_processed"];
        stats4[1] = st["samples_incl_padding"];
        stats4[2] = st["batches_called"];
        stats4[3] = st["partial_batches_called"];
        return 0;
    } catch (const std::exception &e) {
        g_herr = e.what();
        return -1;
    }
}

}  // extern "C"
