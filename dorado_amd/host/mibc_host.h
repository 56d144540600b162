// dorado_amd/host/mibc_host.h — C++ host layer above the C-ABI (include/mibc.h).
//
// Mirrors, tensor-free and libtorch-free, the reference classes on either side of the drop-in
// boundary:
//   ModelRunnerBase   dorado/basecall/include/basecall/ModelRunnerBase.h:20-38 (same methods; the
//                     only signature change is accept_chunk(idx, const uint16_t* f16, n) instead
//                     of an at::Tensor — INTEGRATION.md shows the one-line adapter)
//   HipModelRunner    dorado/basecall/CudaModelRunner.cpp:13-79 (pinned in/out, delegates to caller)
//   HipCaller         dorado/basecall/CudaCaller.cpp:149-287,634-720 (1 per device: engine, batch dimensions for
//                     every chunk size, the device's FIFO, one GPU thread, stats, terminate/restart)
//   create_basecall_runners  dorado/api/runner_creation.cpp:46-133 ([device][runner] order)
//   generate_chunks / stitch_chunks  dorado/read_pipeline/base/{chunk,stitch}.cpp
//   ScalerNode (host half)  dorado/read_pipeline/nodes/ScalerNode.cpp:144-269: scaling parameters of a
//                     read (PA strategy: closed formula; quantile / med_mad: mibc_scaler_stats on the
//                     device), open-pore table, signal trim.  The sample-wise map itself runs on the
//                     device, fused into conv1 (accept_chunk_i16 -> mibc_call_i16).
//   SimplexBasecaller  the chunk -> batch -> call -> stitch loop of
//                     dorado/read_pipeline/nodes/BasecallerNode.cpp:96-171,289-457,205-287
//                     (one worker thread per runner, repeat-padding of short tails)
#pragma once
#include "../../include/mibc.h"

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <map>
#include <functional>
#include <memory>
#include <mutex>
#include <optional>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace dorado_amd::host {

// basecall/include/basecall/DecodedChunk.h:9-13
struct DecodedChunk {
    std::string sequence;
    std::string qstring;
    std::vector<uint8_t> moves;
};

using NamedStats = std::map<std::string, double>;  // utils/include/utils/stats.h:25

class ModelRunnerBase {
public:
    virtual ~ModelRunnerBase() = default;
    virtual void accept_chunk(int chunk_idx, const uint16_t *chunk_f16, size_t n_samples) = 0;
    virtual std::vector<DecodedChunk> call_chunks(int num_chunks) = 0;
    virtual const mibc_model_desc &config() const = 0;
    virtual size_t chunk_size() const = 0;
    virtual size_t batch_size() const = 0;
    virtual bool variable_chunk_sizes() const { return false; }
    virtual std::pair<int, int> batch_timeouts_ms() const { return {100, 5000}; }  // ModelRunnerBase.cpp:6-9
    virtual bool is_low_latency() const { return false; }
    virtual void terminate() = 0;
    virtual void restart() = 0;
    virtual std::string get_name() const = 0;
    virtual NamedStats sample_stats() const = 0;
};
using RunnerPtr = std::unique_ptr<ModelRunnerBase>;

// torch_utils/cuda_utils.cpp:224-248,364-384 — accepts "hip:" and, for drop-in use, "cuda:".
// Returns false (and sets error) exactly where the reference's try_parse_device_ids does.
bool try_parse_device_ids(const std::string &device_string, size_t num_devices,
                          std::vector<int> &device_ids, std::string &error_message);

std::vector<size_t> generate_chunks(size_t num_samples, size_t chunk_size, size_t stride,
                                    size_t overlap);  // throws like chunk.cpp:11-30

// read_pipeline/base/chunk.cpp:49-107: nearly equal chunks, interior edges aligned to the stride (throws alike)
std::vector<std::pair<size_t, size_t>> generate_variable_chunks(size_t num_samples, size_t chunk_size, size_t stride,
                                                               size_t overlap);

struct Chunk {  // read_pipeline/base/include/read_pipeline/base/messages.h (utils::Chunk)
    size_t input_offset = 0;
    size_t raw_chunk_size = 0;
    std::string seq, qstring;
    std::vector<uint8_t> moves;
};
struct StitchedRead {
    std::string seq, qstring;
    std::vector<uint8_t> moves;
};
StitchedRead stitch_chunks(const std::vector<const Chunk *> &called_chunks, size_t raw_samples,
                           int model_stride);  // stitch.cpp:12-96

// ---- ScalerNode, host half ------------------------------------------------------
// config/include/config/BasecallModelConfig.h:13-44
enum class ScalingStrategy { MED_MAD, QUANTILE, PA };
struct StandardisationScalingParams {
    bool standardise = false;
    float mean = 0.0f;
    float stdev = 1.0f;
};
struct QuantileScalingParams {
    float quantile_a = 0.2f;
    float quantile_b = 0.9f;
    float shift_multiplier = 0.51f;
    float scale_multiplier = 0.53f;
};
struct SignalNormalisationParams {
    ScalingStrategy strategy = ScalingStrategy::QUANTILE;
    QuantileScalingParams quantile;
    StandardisationScalingParams standardisation;
};
// The calibration fields of SimplexRead that ScalerNode reads (messages.h: scaling, offset,
// open_pore_level, read_common.flow_cell_product_code).
struct ReadCalibration {
    float scaling = 1.0f;
    float offset = 0.0f;
    float open_pore_level = __builtin_nanf("");
    std::string flow_cell_product_code;
};
// What ScalerNode leaves on the read: x_scaled = f16((x - (shift + open_pore_adjustment)) / scale),
// read_common.scale / shift in pA, num_trimmed_samples.
struct ReadScaling {
    float shift = 0.0f, scale = 1.0f, open_pore_adjustment = 0.0f;
    float scale_pa = 1.0f, shift_pa = 0.0f;   // ScalerNode.cpp:226-227
    float device_shift() const { return shift + open_pore_adjustment; }  // the pair mibc_*_i16 takes
};
std::optional<float> expected_open_pore_level(const std::string &flow_cell_product_code);  // :112-139
// Strategy PA (ScalerNode.cpp:186-215).  Throws std::invalid_argument for the data-driven strategies,
// whose (shift, scale) come from mibc_scaler_stats on the device.
ReadScaling pa_read_scaling(const SignalNormalisationParams &p, const ReadCalibration &cal);
// ScalerNode.cpp:226-227 applied to a (shift, scale) pair obtained on the device.
ReadScaling finish_read_scaling(float shift, float scale, const ReadCalibration &cal);
// torch_utils/trim.cpp:23-60 on a scaled f16 signal prefix; defaults = trim.h:17-19.
int trim_signal(const uint16_t *signal_f16, int n, float threshold = 2.4f, int window_size = 40,
                int min_elements = 3);
// The DNA branch of ScalerNode.cpp:231-254: 10 for standardised models, else trim() over the first
// min(8000, n/2) scaled samples; 0 when the trim would swallow the read.
int dna_trim_start(const SignalNormalisationParams &p, const uint16_t *scaled_f16, size_t n_samples);
// RNA models, in front of the scaling (ScalerNode.cpp:58-107, determine_rna_adapter_pos): where the DNA adapter of a dRNA read
// ends, from the median of a 250-sample window sliding in steps of 50 over raw samples [1000, 3n/4) — the first window at which,
// among the last five medians, the largest came after the smallest and exceeds it by > 150, or by > 125 with the largest > 700.
// 0 if no such window.
int rna_adapter_pos(const int16_t *raw, int n_samples);
// What ScalerNode does with it (:157-184): with no RNA-based adapter info on the read, cut the signal at the position when it
// lies inside the read (then the scaling statistics see the whole remainder), else keep the signal and let the statistics skip
// the adapter.  has_rna_based_adapters: the read's AdapterInfo asks for RNA adapter trimming downstream -> nothing happens here.
struct RnaTrim {
    int trim_start = 0;                    // samples cut from the front (= num_trimmed_samples for RNA reads, :256)
    int rna_adapter_end_signal_pos = 0;    // first sample the data-driven scaling statistics use (:218-219)
};
RnaTrim rna_trim(const int16_t *raw, size_t n_samples, bool has_rna_based_adapters);

// ScalerNode::input_thread_fn for one read (ScalerNode.cpp:144-267), in the reference's order: RNA adapter cut (RNA models) ->
// shift / scale (PA: closed formula; QUANTILE / MED_MAD: mibc_scaler_stats on the device over the samples behind
// rna_adapter_end_signal_pos) -> the sample map on the device -> DNA trim on the scaled prefix (DNA models).
struct ScaledRead {
    ReadScaling scaling;                  // shift, scale, open-pore adjustment; read_common.scale / shift in pA
    int num_trimmed_samples = 0;          // read_common.num_trimmed_samples
    int rna_adapter_end_signal_pos = 0;   // read_common.rna_adapter_end_signal_pos
    size_t first_sample = 0;              // the samples that go on = raw[first_sample, n): adapter cut + DNA trim
    std::vector<uint16_t> signal_f16;     // want_signal: what the reference leaves in read_common.raw_data (scaled, trimmed)
};
// The two passes over the samples, as a seam: HipCaller supplies the device kernels (mibc_scaler_stats / mibc_scale_reads);
// the host-logic tests supply their own (tests/test_scaler_node.py drives the orchestration on a machine without a GPU).
struct ScalerOps {
    std::function<std::pair<float, float>(const int16_t *, size_t, const SignalNormalisationParams &)> stats;   // QUANTILE / MED_MAD
    std::function<std::vector<uint16_t>(const int16_t *, size_t, float shift, float scale)> scale;           // f16((x - shift) / scale)
};
ScaledRead scaler_node(const ScalerOps &ops, const SignalNormalisationParams &p, bool is_rna_model, bool has_rna_based_adapters,
                       const int16_t *raw, size_t n_samples, const ReadCalibration &cal, bool want_signal);
class HipCaller;
ScaledRead scaler_node(HipCaller &caller, const SignalNormalisationParams &p, bool is_rna_model, bool has_rna_based_adapters,
                       const int16_t *raw, size_t n_samples, const ReadCalibration &cal, bool want_signal);

// Many reads at once — the form the device wants: ONE statistics launch and ONE sample-map launch for the whole set (per-read
// calls cost two device round trips each); host decisions (adapter cut, trim) per read around them.  Same results as the
// per-read form, read by read.
struct ScalerInput {
    const int16_t *raw;
    size_t n_samples;
    ReadCalibration cal;
    bool has_rna_based_adapters = false;
};
std::vector<ScaledRead> scaler_node(HipCaller &caller, const SignalNormalisationParams &p, bool is_rna_model,
                                    const std::vector<ScalerInput> &reads, bool want_signal);

// The fields of basecall::BasecallerCreationParams that shape a caller (basecall/include/basecall/ModelRunnerBase.h:43-52;
// model_config / device / pipeline_type arrive as constructor arguments).
struct CallerParams {
    float memory_limit_fraction = 0.8f;    // of the free device memory, minus 1 GB (CudaCaller.cpp:434-439)
    float batch_size_time_penalty = 0.05f; // smallest batch within (1 + penalty) of the best time per chunk (:603-627)
    bool run_batchsize_benchmarks = false; // true: the reference's timing sweep; false: the engine's known knee
    bool emit_batchsize_benchmarks = false;// print the sweep's (batch, ms per chunk) table to stderr
    bool variable_chunk_sizes = false;     // runners pack several chunks per batch row (CudaModelRunner.cpp:21-49)
    // variable chunk sizes: fraction of the batch rows' samples a runner offers BasecallerNode as its batch budget.  The
    // node packs a 1-D span (a chunk may straddle rows), this engine packs rows: with first-fit packing 0.85 still fits on
    // read sets whose chunks are mostly 0.5-1.0 of chunk_size (simulated); what does not fit costs a second
    // engine call (counted in sample_stats: var_overflow_batches).
    float variable_batch_fill = 0.8f;
};

struct BatchDims {  // CudaCaller::BatchDims (basecall/include/basecall/CudaCaller.h): batch, samples, output steps
    int N = 0, T_in = 0, T_out = 0;
};

// One per DEVICE (CudaCaller.cpp:149-200): one engine, one workspace, one GPU thread; every chunk size of the pipeline
// is a "batch dimension" of this caller (m_batch_dims, :382-413) and all of them go through the device's one FIFO
// (:204-214 "Global task queues, one per GPU").
class HipCaller {
public:
    HipCaller(const mibc_model_desc &desc, const float *const *weights, int n_weights, int device,
              const std::vector<int> &chunk_sizes, int batch_size, const mibc_decode_opts &opts,
              const CallerParams &params = CallerParams());
    HipCaller(const mibc_model_desc &desc, const float *const *weights, int n_weights, int device,
              int chunk_size, int batch_size, const mibc_decode_opts &opts)
            : HipCaller(desc, weights, n_weights, device, std::vector<int>{chunk_size}, batch_size, opts) {}
    ~HipCaller();
    // Blocks until decoded (CudaCaller::call_chunks, CudaCaller.cpp:224-271).
    // in: pinned f16 [batch][chunk]; out: pinned int8 [3][batch][T] of batch dimension `dims`.
    std::vector<DecodedChunk> call_chunks(size_t dims, const uint16_t *in_pinned, int8_t *out_pinned, int num_chunks);
    // Raw int16 batch + pinned [batch][2] (shift, scale): scaling fused into conv1 (mibc_call_i16).
    std::vector<DecodedChunk> call_chunks_i16(size_t dims, const int16_t *in_pinned, const float *shift_scale_pinned,
                                              int8_t *out_pinned, int num_chunks);
    // Several chunks per row (mibc_call_var); returns one DecodedChunk per entry of `chunks`.
    std::vector<DecodedChunk> call_chunks_var(size_t dims, const uint16_t *in_pinned, int8_t *out_pinned,
                                              const std::vector<mibc_var_chunk> &chunks);
    int model_stride() const { return m_dims[0].T_in / m_dims[0].T_out; }
    // Per-read (shift, scale) of the QUANTILE / MED_MAD strategies on the device (mibc_scaler_stats).
    std::vector<std::pair<float, float>> scaler_stats(const std::vector<std::pair<const int16_t *, size_t>> &reads,
                                                      const SignalNormalisationParams &p);
    // f16((x - shift) / scale) of whole reads on the device (mibc_scale_reads == utils::shift_scale_tensor_i16_to_f16_inplace
    // bit for bit); shift_scale[r] = (ReadScaling::device_shift(), ReadScaling::scale).
    std::vector<std::vector<uint16_t>> scale_reads(const std::vector<std::pair<const int16_t *, size_t>> &reads,
                                                   const std::vector<std::pair<float, float>> &shift_scale);
    void terminate();
    void restart();
    const mibc_model_desc &config() const { return m_desc; }
    size_t num_batch_dims() const { return m_dims.size(); }
    const BatchDims &batch_dims(size_t i) const { return m_dims.at(i); }
    int chunk_size(size_t dims = 0) const { return m_dims.at(dims).T_in; }
    int batch_size(size_t dims = 0) const { return m_dims.at(dims).N; }
    int output_steps(size_t dims = 0) const { return m_dims.at(dims).T_out; }
    int device() const { return m_device; }
    bool variable_chunk_sizes() const { return m_params.variable_chunk_sizes; }
    float variable_batch_fill() const { return m_params.variable_batch_fill; }
    std::pair<int, int> batch_timeouts_ms() const { return {300000, 30000}; }  // CudaCaller.cpp:126-132
    NamedStats sample_stats() const;
    std::string get_name() const { return "HipCaller_hip:" + std::to_string(m_device); }
    // (batch size, ms per chunk) pairs of the timing sweep (run_batchsize_benchmarks / batch_size = -1); empty otherwise
    const std::vector<std::pair<int, double>> &batch_timings() const { return m_batch_timings; }
    const CallerParams &params() const { return m_params; }

    struct NNTask {
        HipCaller *caller = nullptr;
        size_t dims = 0;
        const uint16_t *in = nullptr;
        const float *ss = nullptr;   // non-null: `in` holds raw int16 samples
        const std::vector<mibc_var_chunk> *var = nullptr;   // non-null: several chunks per row
        int8_t *out = nullptr;
        int num_chunks = 0;
        int rc = 0;
        std::string error;
        bool done = false;
        std::mutex mut;
        std::condition_variable cv;
    };
    struct DeviceQueue;   // the per-device FIFO shared by every caller of that device

private:
    void start_thread();
    void gpu_thread_fn();
    void run_task(const std::shared_ptr<NNTask> &task);   // enqueue + wait + throw on error
    std::vector<DecodedChunk> submit(size_t dims, const uint16_t *in, const float *ss, int8_t *out, int num_chunks);
    int choose_batch_size(int chunk_size, int requested);
    std::mutex m_engine_mutex;  // the engine (one stream) is used by the GPU thread and by scaler_stats
    mibc_model_desc m_desc;
    mibc_decode_opts m_opts;
    CallerParams m_params;
    mibc_engine *m_engine = nullptr;
    int m_device;
    std::vector<BatchDims> m_dims;
    DeviceQueue *m_queue = nullptr;
    std::thread m_thread;
    std::atomic<bool> m_terminate{false};
    std::atomic<int64_t> m_batches{0};
    std::atomic<int64_t> m_model_decode_us{0};
    std::vector<std::pair<int, double>> m_batch_timings;
};

class HipModelRunner final : public ModelRunnerBase {
public:
    explicit HipModelRunner(std::shared_ptr<HipCaller> caller, size_t batch_dims_idx = 0);
    ~HipModelRunner() override;
    // Fixed chunks: row chunk_idx of the batch.  Variable chunk sizes (the caller was created with
    // CallerParams::variable_chunk_sizes): chunk_idx is ignored and the chunk — any stride-multiple length up to
    // chunk_size() — is packed behind the previous ones, several per batch row, exactly as
    // CudaModelRunner::accept_chunk does (CudaModelRunner.cpp:21-32); BasecallerNode budgets len/stride + 2 steps per
    // chunk (BasecallerNode.cpp:408-430), which is this engine's 2-step gap rule.
    void accept_chunk(int chunk_idx, const uint16_t *chunk_f16, size_t n_samples) override;
    // Raw ADC samples of one chunk + its read's (shift, scale): the batch is then scaled on the device.
    // A batch is either all-f16 or all-int16 (the mode resets after every call_chunks).
    void accept_chunk_i16(int chunk_idx, const int16_t *chunk_raw, size_t n_samples, float shift, float scale);
    std::vector<DecodedChunk> call_chunks(int num_chunks) override;
    // variable mode, explicit form: write samples anywhere into the pinned batch, then call with the chunk table
    uint16_t *batch_row(int row) { return m_in + size_t(row) * chunk_size(); }
    std::vector<DecodedChunk> call_chunks_var(const std::vector<mibc_var_chunk> &chunks);
    bool variable_chunk_sizes() const override { return m_caller->variable_chunk_sizes(); }
    const mibc_model_desc &config() const override { return m_caller->config(); }
    size_t chunk_size() const override { return size_t(m_caller->chunk_size(m_dims)); }
    // Fixed chunks: the rows of the engine batch.  Variable chunk sizes: the budget BasecallerNode may fill
    // (batch_size() * (chunk_size / stride + 2) steps, BasecallerNode.cpp:303) = rows() * CallerParams::variable_batch_fill
    // rounded down to the node's 32-row spans, so that the chunks it hands over fit the rows without a second engine call.
    size_t batch_size() const override;
    std::pair<int, int> batch_timeouts_ms() const override { return m_caller->batch_timeouts_ms(); }
    void terminate() override { m_caller->terminate(); }
    void restart() override { m_caller->restart(); }
    std::string get_name() const override;
    NamedStats sample_stats() const override;
    HipCaller &caller() { return *m_caller; }

private:
    std::shared_ptr<HipCaller> m_caller;
    size_t m_dims;
    uint16_t *m_in = nullptr;  // pinned [batch][chunk]
    float *m_ss = nullptr;     // pinned [batch][2]
    int8_t *m_out = nullptr;   // pinned [3][batch][T]
    int m_mode = 0;            // 0 undecided, 1 f16 chunks, 2 raw int16 chunks
    int m_id;
    std::atomic<int64_t> m_batches{0};
    // variable-chunk packing state of the batch being filled
  public:
    // Row placement of variable-length chunks: FIRST-FIT over all rows of the batch (a max-free-space tree makes it
    // O(log rows) per chunk).  A chunk cannot straddle two rows here (rows are independent sequences of the time-major
    // LSTM kernels), whereas BasecallerNode budgets the batch as 32-row spans of steps (BasecallerNode.cpp:303-305,
    // 421-426), so what the node calls a full batch only fits when the rows pack well: see batch_size().
    class RowPacker {
    public:
        void reset(size_t rows, size_t chunk_size, size_t gap);
        // places n samples; false when no row has room (the caller's overflow path).  start = first sample in the row.
        bool place(size_t n, int &row, int &start);
        size_t rows_used() const { return m_used; }

    private:
        std::vector<int> m_free;   // tree of max free samples (a row's free space counts the gap it must leave in front)
        std::vector<int> m_fill;   // samples used per row incl. its chunks' trailing gaps
        size_t m_rows = 0, m_leaf0 = 1, m_cs = 0, m_gap = 0, m_used = 0;
    };
    size_t rows() const { return size_t(m_caller->batch_size(m_dims)); }   // batch rows of the engine call

  private:
    std::vector<mibc_var_chunk> m_var_table, m_var_overflow;
    std::vector<std::vector<uint16_t>> m_var_overflow_data;
    RowPacker m_packer;
    std::atomic<int64_t> m_var_batches{0}, m_var_overflow_batches{0}, m_var_rows_used{0};
};

// Samples per output step: product of the conv strides, divided by the upsample factor of the transformer models
// (config/BasecallModelConfig.cpp:447-454); chunk granularity = stride_inner * 16 for transformer models
// (config/include/config/BasecallModelConfig.h:152-159).
int model_stride(const mibc_model_desc &desc);
int chunk_size_granularity(const mibc_model_desc &desc);
// The batch dimensions CudaCaller builds for high-throughput simplex calling (CudaCaller.cpp:382-413): the
// requested chunk size and 0.5x of it, each rounded down to the chunk granularity and kept above the overlap;
// largest first, duplicates removed.
std::vector<int> simplex_chunk_sizes(const mibc_model_desc &desc, int requested_chunk_size, int overlap);
// BasecallerNode::get_chunk_queue_idx (BasecallerNode.cpp:81-94): the queue with the smallest chunk size that
// fits the whole read, else the one with the largest chunk size.
size_t get_chunk_queue_idx(const std::vector<size_t> &chunk_sizes, size_t read_raw_size);

// [device][runner][chunk_size] (inner vector: runner-major, chunk sizes in the order given) — the order
// api::create_basecall_runners returns and BasecallerNode relies on (api/runner_creation.cpp:115-123,
// BasecallerNode.cpp:494-501).  ONE caller (engine + workspace + GPU thread) per device, serving every chunk size
// as a batch dimension; num_runners runners per batch dimension (utils/include/utils/parameters.h:11 num_runners = 2).
// batch_size 0 = automatic, sized against the device memory that is still free when the caller is created.
std::vector<std::vector<RunnerPtr>> create_basecall_runners(
        const mibc_model_desc &desc, const float *const *weights, int n_weights,
        const std::string &device_string, int num_runners, const std::vector<int> &chunk_sizes, int batch_size,
        const mibc_decode_opts &opts, const CallerParams &params = CallerParams());
// single chunk size
std::vector<std::vector<RunnerPtr>> create_basecall_runners(
        const mibc_model_desc &desc, const float *const *weights, int n_weights,
        const std::string &device_string, int num_runners, int chunk_size, int batch_size,
        const mibc_decode_opts &opts);

struct CalledRead {
    std::string seq, qstring;
    std::vector<uint8_t> moves;
    std::vector<size_t> chunk_offsets;
};

// Chunk -> batch -> call -> stitch over a set of reads, one worker thread per runner pulling
// from one shared chunk queue (BasecallerNode semantics without the message plumbing).
class SimplexBasecaller {
public:
    SimplexBasecaller(std::vector<RunnerPtr> runners, int overlap, int model_stride);
    std::vector<CalledRead> basecall(const std::vector<std::vector<uint16_t>> &reads_f16);
    // Raw reads: int16 ADC samples (already cut at the read's trim_start) + the read's device
    // (shift, scale) pair = ReadScaling::device_shift(), ReadScaling::scale; scaling happens inside the
    // engine's first convolution (HipModelRunner::accept_chunk_i16).
    struct RawRead {
        const int16_t *signal;
        size_t n_samples;
        float shift, scale;
    };
    std::vector<CalledRead> basecall_raw(const std::vector<RawRead> &reads);
    // Variable chunk sizes (BasecallerNode.cpp:397-430 with m_variable_chunk_sizes): reads are cut with
    // generate_variable_chunks, each chunk is topped up to a stride multiple by repeating its head, chunks of any
    // length share batch rows (2-step gaps) and go through mibc_call_var.  f16 reads.
    std::vector<CalledRead> basecall_variable(const std::vector<std::vector<uint16_t>> &reads_f16);
    // Measurement helper: n_reads reads of read_len samples each, read i = data + (i % n_distinct) * read_len (f16);
    // same path as basecall(); returns the total number of called bases.
    size_t basecall_repeated(const uint16_t *data, size_t n_distinct, size_t read_len, size_t n_reads,
                             double *seconds_to_last_read = nullptr);
    NamedStats sample_stats() const;

private:
    struct ReadView {
        const uint16_t *data;   // f16 bits or raw int16 bits
        size_t n;
        bool raw;
        float shift, scale;
    };
    std::vector<CalledRead> basecall_views(const std::vector<ReadView> &reads);
    std::vector<RunnerPtr> m_runners;
    std::vector<size_t> m_chunk_sizes;   // one chunk queue per size (BasecallerNode.cpp:494-501, 515-522)
    int m_overlap, m_stride;
    std::atomic<int64_t> m_samples_processed{0}, m_samples_incl_padding{0}, m_batches{0},
            m_partial_batches{0};
};

}  // namespace dorado_amd::host
