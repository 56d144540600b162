/* include/mibc.h — C-ABI of the MI355X-native simplex basecalling engine ("mibc").
 *
 * This is the drop-in boundary: plain C, raw pointers and sizes, no torch types.
 * It replaces what the reference reaches through `extern "C" { #include "koi.h" }` (closed
 * source; call sites only) plus the libtorch ops around it, for the GPU side of
 * basecall::ModelRunnerBase (dorado/basecall/include/basecall/ModelRunnerBase.h:20-38).
 * The host-side C++ mirror of CudaCaller / CudaModelRunner that sits on top of this header is
 * dorado_amd/host/ (HipCaller, HipModelRunner); INTEGRATION.md shows the reference-side binding.
 *
 * Conventions (Koi-like): every entry returns int: 0 = MIBC_OK, >0 = not supported (caller may
 * fall back), <0 = error (text via mibc_last_error).  No exceptions cross this boundary.
 * All device work of one engine is issued on that engine's own HIP stream; entries marked
 * (async) return before the work completes — call mibc_sync().
 */
#ifndef MIBC_H
#define MIBC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libmibc.so is built with -fvisibility=hidden: exactly the entry points declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#define MIBC_API __attribute__((visibility("default")))
#else
#define MIBC_API
#endif

#define MIBC_OK 0
#define MIBC_NOT_SUPPORTED 1
#define MIBC_ERR_ARG (-1)
#define MIBC_ERR_HIP (-2)
#define MIBC_ERR_MEM (-3)

/* Activation ids == dorado/config/include/config/common.h:19 (enum Activation order). */
#define MIBC_ACT_SWISH 0
#define MIBC_ACT_SWISH_CLAMP 1
#define MIBC_ACT_TANH 2

/* The fields of config::BasecallModelConfig the hot path reads
 * (dorado/config/include/config/BasecallModelConfig.h:99-160; ConvParams common.h:41-49). */
typedef struct mibc_model_desc {
    int n_convs;
    int conv_insize[8], conv_size[8], conv_winlen[8], conv_stride[8];
    int conv_act[8];
    int lstm_size, lstm_layers;
    int state_len, outsize;
    int bias;         /* linear1 has a bias (pre-v4 / decomposed head) */
    int clamp;        /* clamp scores to [-5,5] (folded into the decoder's score read, as
                         CUDADecoder does: basecall/model/CRFModel.cpp:107-109) */
    float scale;      /* 5.0 => head applies 5*tanh */
    int out_features; /* >0: two-stage head (linear1 -> linear2); -1: single */
    int num_features;
    /* transformer (sup@v5) fields; tx_d_model <= 0 selects the LSTM-CRF path */
    int tx_d_model, tx_nhead, tx_depth, tx_dim_ff, tx_win_upper, tx_win_lower, tx_max_seq_len;
    float tx_deepnorm_alpha, tx_theta;
    int up_size, up_scale_factor;
    float crf_scale, crf_blank_score;
    int crf_expand_blanks;
    /* 1: the reference's quantised LSTM path (nn/LSTMStack.cpp:127-211, KOI_I8): int8 weights (per-row scales,
     * utils::quantize_tensor) and int8 activations — for EVERY layer when the last convolution's activation is tanh (the
     * v4.3 LSTM-CRF models: nn/ConvStack.cpp:72 hands over CUTLASS_TNC_I8), else layers 2..L with the first layer in f16
     * (:73, LSTMStack.cpp:199-207).  LSTM models with
     * lstm_size 128 / 256 / 384 (any batch) or 512 / 768 / 1024 (batches that are multiples of 256: mibc_batch_granularity()
     * reports 256 then) and >= 2 layers; combinable with variable chunks (mibc_*_var: masked instances of the same kernels —
     * the reference's default GPU mode is both at once, basecall/CudaModelRunner.cpp:21-49).
     * 0: f16 throughout.  Which of the two a model gets by default is the CALLER's rule: the reference-side binding
     * (integration/HipModelRunnerAdapter.h mibc_desc_from_config) and bench.py apply the reference's own
     * (nn/ConvStack.cpp:60-89: int8 for tanh-conv models with 128 < lstm_size <= 1024), both arithmetics have a stated parity
     * contract. */
    int lstm_quant;
} mibc_model_desc;

/* basecall::decode::DecoderOptions (dorado/basecall/include/basecall/DecodedChunk.h:15-23). */
typedef struct mibc_decode_opts {
    int beam_width;    /* 32 */
    float beam_cut;    /* 100.0 */
    float blank_score; /* 2.0 (fixed stay score) */
    float q_shift;     /* config qbias */
    float q_scale;     /* config qscale */
} mibc_decode_opts;

/* Per-stage GPU time of the last mibc_call*/
typedef struct mibc_stage_ms {
    float conv, lstm, head, decode, total;
    float lstm_layer[8]; /* per LSTM layer */
    float h2d, d2h;
} mibc_stage_ms;

typedef struct mibc_engine mibc_engine;

/* ---- devices (replaces torch_utils/cuda_utils.cpp:224-248,364-384 device discovery) ---- */
MIBC_API int mibc_device_count(void);
MIBC_API int mibc_device_memory(int device_id, size_t *free_bytes, size_t *total_bytes); /* cuda_utils.cpp:250-262 */
MIBC_API const char *mibc_last_error(const mibc_engine *e); /* e may be NULL: last global error */
/* Identity of the binary: 16 hex digits of a sha256 over every source file the library was compiled from (tools/build_id.py;
 * "-dbg" appended in libmibc_dbg.so) — what a caller logs next to the model name, the role of the Koi / dorado version strings
 * (dorado/dorado_version.h DORADO_VERSION); the test suite refuses a library whose id is not the tree's. */
MIBC_API const char *mibc_build_id(void);

/* ---- lifetime (replaces CudaCaller ctor: basecall/CudaCaller.cpp:149-200) ----
 * weights: host f32 tensors in module.parameters() order (basecall/crf_utils.cpp:34-88):
 * conv{1..3}.{weight[Cout,Cin,W],bias}, rnn{1..L}.{weight_ih,weight_hh,bias_ih,bias_hh},
 * linear1.weight [,linear1.bias] [,linear2.weight].  Converted to f16 device layouts once. */
MIBC_API int mibc_create(int device_id, const mibc_model_desc *desc, const float *const *weights,
                int n_weights, mibc_engine **out);
MIBC_API void mibc_destroy(mibc_engine *e);

/* ---- memory (replaces CudaCaller memory model :323-369 and WorkingMemory arena) ---- */
MIBC_API int mibc_query_memory(const mibc_engine *e, int T_in, size_t *bytes_per_chunk, size_t *bytes_fixed);
MIBC_API int mibc_reserve(mibc_engine *e, int N_max, int T_in); /* (re)allocates the device workspace */
MIBC_API int mibc_output_steps(const mibc_engine *e, int T_in); /* T = number of output steps */
MIBC_API int mibc_batch_granularity(const mibc_engine *e);      /* N must be a multiple of this */
/* The weight quantisation of the lstm_quant path, host only (no device): utils::quantize_tensor(cat(W_ih, W_hh, 1).half(), 1)
 * (torch_utils/tensor_utils.cpp:293-300 as called by nn/LSTMStack.cpp:160-168), bit for bit — f16 arithmetic included.
 * wih, whh: [4C][C] f32 (module.parameters() layout); q: [4C][2C] int8 (columns < C from W_ih); scale: [4C]. */
MIBC_API int mibc_quantize_lstm_weights(const float *wih, const float *whh, int C, int8_t *q, float *scale);

MIBC_API void *mibc_host_alloc(size_t bytes); /* pinned host memory (CudaCaller.cpp:289-314) */
MIBC_API void mibc_host_free(void *p);
MIBC_API void *mibc_device_alloc(mibc_engine *e, size_t bytes);
MIBC_API void mibc_device_free(mibc_engine *e, void *p);
MIBC_API int mibc_memcpy_h2d(mibc_engine *e, void *dst_dev, const void *src_host, size_t bytes); /* sync */
MIBC_API int mibc_memcpy_d2h(mibc_engine *e, void *dst_host, const void *src_dev, size_t bytes); /* sync */

/* ---- the hot path ----
 * in:     f16 [N, 1, T_in]                (what BasecallerNode hands accept_chunk)
 * scores: f16 [N, T, K], K = 4^(state_len+1)   (CRFModel.cpp:111: NTC f16)
 * out:    int8 [3][N][T] = moves | bases (ASCII, packed at the front, NUL padded) | qstring
 *         (identical to the reference's CUDADecoder buffer: decode/CUDADecoder.cpp:66-71,153-168)
 */
MIBC_API int mibc_forward(mibc_engine *e, const uint16_t *in_dev, int N, int T_in,
                 uint16_t *scores_dev); /* (async) network only; replaces CRFModelImpl::run_koi */
MIBC_API int mibc_decode(mibc_engine *e, const uint16_t *scores_dev, int N, int T,
                const mibc_decode_opts *opts,
                int8_t *out_dev); /* (async) replaces CUDADecoder::beam_search_part_1 */
MIBC_API int mibc_call_device(mibc_engine *e, const uint16_t *in_dev, int N, int T_in,
                     const mibc_decode_opts *opts, int8_t *out_dev); /* (async) forward+decode */
MIBC_API int mibc_call(mibc_engine *e, const uint16_t *in_host, int N, int T_in,
              const mibc_decode_opts *opts,
              int8_t *out_host); /* H2D + forward + decode + D2H, synchronous
                                    (CudaCaller::call_chunks, CudaCaller.cpp:224-271) */
MIBC_API int mibc_sync(mibc_engine *e);
/* Opt-in: run the decoder (backward / forward scans, posteriors, beam search) on its own stream with the scores buffer
 * double-buffered, so that the decoder of one batch (HBM streaming, idle matrix pipes) runs under the network of the next
 * (matrix pipe / L2 bound, 0.7 TB/s of HBM traffic), and — where a batch is decoded in sub-batches — under the head of the next
 * sub-batch.  Results are unchanged.  Costs a second scores buffer; per-stage profiling (mibc_get_stage_ms) is not available
 * while it is on.  Applies to mibc_call_device, mibc_call_device_i16, mibc_call, mibc_call_i16, mibc_call_async (the
 * variable-chunk calls stay serial).  After the asynchronous device calls the output planes are complete once mibc_sync or
 * mibc_memcpy_d2h returns (both join the decoder stream); mibc_query_memory counts the second scores buffer while it is on. */
MIBC_API int mibc_set_decode_overlap(mibc_engine *e, int on);
/* Two-phase form of mibc_call (the overlap CudaCaller gets from its runners' own streams, CudaCaller.cpp:645-719 +
 * decode/CUDADecoder.h:13-15): mibc_call_async enqueues H2D (copy stream) -> network + decode (engine stream) ->
 * D2H (second copy stream) for `slot` (0 or 1) and returns; mibc_call_wait blocks until that slot's output has
 * arrived in out_host.  Two slots in flight: the copies of one batch run beside the kernels of the other.  A slot
 * must be waited for before it is submitted again; in_host / out_host must be pinned (mibc_host_alloc) and stay
 * valid until the wait returns.  shift_scale_host: NULL (in_host holds scaled f16) or float [N][2] (raw int16). */
MIBC_API int mibc_call_async(mibc_engine *e, int slot, const void *in_host, const float *shift_scale_host, int N, int T_in,
                    const mibc_decode_opts *opts, int8_t *out_host);
MIBC_API int mibc_call_wait(mibc_engine *e, int slot);
MIBC_API int mibc_call_poll(mibc_engine *e, int slot); /* 1 = finished (then call mibc_call_wait), 0 = still running */

/* ---- signal scaling in front of the path (the device side of ScalerNode,
 *      dorado/read_pipeline/nodes/ScalerNode.cpp:144-269) ----
 * The reference scales each read on the host (int16 ADC -> f16((x - shift) / scale),
 * torch_utils/tensor_utils.cpp:89-142) before chunking; these entry points take the raw int16 chunk
 * rows instead, with ONE (shift, scale) pair per chunk (its read's), and apply the identical map
 * (f32 subtract, IEEE divide, round-to-nearest-even to f16) inside the first convolution's input read.
 * shift_scale: float [N][2]. */
#define MIBC_SCALE_QUANTILE 0 /* ScalerNode.cpp:42-52 (normalisation), utils::quantile_counting */
#define MIBC_SCALE_MED_MAD 1  /* ScalerNode.cpp:32-40 (med_mad) */
MIBC_API int mibc_forward_i16(mibc_engine *e, const int16_t *in_dev, const float *shift_scale_dev, int N, int T_in,
                     uint16_t *scores_dev);
MIBC_API int mibc_call_device_i16(mibc_engine *e, const int16_t *in_dev, const float *shift_scale_dev, int N, int T_in,
                         const mibc_decode_opts *opts, int8_t *out_dev);
MIBC_API int mibc_call_i16(mibc_engine *e, const int16_t *in_host, const float *shift_scale_host, int N, int T_in,
                  const mibc_decode_opts *opts, int8_t *out_host);
/* Per-read (shift, scale) of the two data-driven strategies.  Reads are concatenated in sig_dev;
 * read r = [offsets_dev[r], offsets_dev[r+1]) (n_reads + 1 offsets).  params4 (quantile only) =
 * {quantile_a, quantile_b, shift_multiplier, scale_multiplier} (config::QuantileScalingParams).
 * raw_dev (optional, [n_reads][2]) receives (q_a, q_b) resp. (median, median |x - median|).
 * Integer work, bit-exact with the reference.  (The PA strategy, ScalerNode.cpp:186-215, is a closed
 * formula of the read's calibration and needs no pass over the samples: host side.) */
MIBC_API int mibc_scaler_stats(mibc_engine *e, const int16_t *sig_dev, const int64_t *offsets_dev, int n_reads,
                      int strategy, const float *params4, float *shift_scale_dev, float *raw_dev);
/* Whole reads: out[i] = f16((float(x[i]) - shift_r) / scale_r); replaces
 * utils::shift_scale_tensor_i16_to_f16_inplace for callers that want the scaled read back
 * (e.g. for the signal trim, torch_utils/trim.cpp). */
MIBC_API int mibc_scale_reads(mibc_engine *e, const int16_t *sig_dev, const int64_t *offsets_dev, int n_reads,
                     const float *shift_scale_dev, uint16_t *out_f16_dev);

/* ---- variable chunk sizes ----
 * The reference's CUDA path packs ragged chunks (utils::generate_variable_chunks, read_pipeline/base/chunk.cpp:
 * 49-107; basecall/CudaModelRunner.cpp:21-49; nn/AuxiliaryData.cpp) so that short reads are not repeat-padded to
 * chunk_size.  Here a batch row [T_in] may hold several chunks: chunk = (row, sample_start, n_samples), both
 * multiples of the model stride, chunks of one row in ascending order and >= 2 output steps (2 * stride samples)
 * apart; samples outside every chunk are ignored.  Every chunk is called exactly as if it stood alone
 * (zero padding at its edges, zero initial LSTM state, its own decode).  Output planes int8 [3][N][T]: chunk c
 * occupies steps [sample_start / stride, (sample_start + n_samples) / stride) of its row in each plane (bases and
 * qstring packed at the front of that interval, NUL padded); everything else is 0.
 * shift_scale: NULL (input rows are scaled f16) or float [N][2] (input rows are raw int16, see above).
 * LSTM models of every supported lstm_size (128 ... 1024; the reference's range: api/runner_creation.cpp:24-44);
 * transformer models return MIBC_NOT_SUPPORTED (the reference's Tx path has no variable chunks either). */
typedef struct mibc_var_chunk {
    int row;
    int sample_start;
    int n_samples;
} mibc_var_chunk;
MIBC_API int mibc_forward_var(mibc_engine *e, const void *in_dev, const float *shift_scale_dev, int N, int T_in,
                     const mibc_var_chunk *chunks_host, int n_chunks, uint16_t *scores_dev);
MIBC_API int mibc_call_device_var(mibc_engine *e, const void *in_dev, const float *shift_scale_dev, int N, int T_in,
                         const mibc_var_chunk *chunks_host, int n_chunks, const mibc_decode_opts *opts,
                         int8_t *out_dev);
MIBC_API int mibc_call_var(mibc_engine *e, const void *in_host, const float *shift_scale_host, int N, int T_in,
                  const mibc_var_chunk *chunks_host, int n_chunks, const mibc_decode_opts *opts, int8_t *out_host);
/* Two-phase form of mibc_call_var on the slots of mibc_call_async (same rules: pinned in_host / out_host, one submission
 * per slot until mibc_call_wait; fixed and variable batches may be mixed across the two slots).  The reference runs its
 * variable-chunk batches on the same stream pipeline as the fixed ones (basecall/CudaModelRunner.cpp:21-49,
 * CudaCaller.cpp:645-719).  chunks_host is consumed before the call returns. */
MIBC_API int mibc_call_var_async(mibc_engine *e, int slot, const void *in_host, const float *shift_scale_host, int N, int T_in,
                        const mibc_var_chunk *chunks_host, int n_chunks, const mibc_decode_opts *opts, int8_t *out_host);

/* ---- POD5 signal decode ----
 * The reference obtains a read's int16 samples from pod5_get_read_complete_signal
 * (dorado/data_loader/DataLoader.cpp:163-170; pod5-file-format 0.3.36, not vendored).  POD5 stores each
 * signal-table row as zstd(svb16(zigzag(delta(int16)))).  The zstd frame is inflated on the host (libzstd);
 * this entry point does the StreamVByte-16 + zig-zag + delta stage for a whole batch of rows on the device,
 * so the samples are born in HBM, next to mibc_scaler_stats / mibc_*_i16.
 * streams_dev: concatenated inflated rows; stream_off_dev / sample_off_dev: n_rows + 1 prefix offsets
 * (bytes / samples); status_dev[r] = 0 ok, 1 = row not consumed exactly (corrupt). */
MIBC_API int mibc_svb16_decode(mibc_engine *e, const uint8_t *streams_dev, const int64_t *stream_off_dev,
                      const int64_t *sample_off_dev, int n_rows, int16_t *out_dev, int *status_dev);

/* ---- measurement (replaces CudaCaller.cpp:552-569 timing + gpu_profiling.h ranges) ---- */
MIBC_API int mibc_time_forward(mibc_engine *e, int N, int T_in, float *ms); /* min of 2 runs, like :552-569 */
MIBC_API int mibc_get_stage_ms(mibc_engine *e, mibc_stage_ms *out);         /* hipEvent times, last call */
/* ... of the profiled call BEFORE the last one: blocks only until that earlier call has finished, so a caller that has
 * already enqueued call i can read the stage times of call i - 1 without draining the stream (two event sets alternate) */
MIBC_API int mibc_get_stage_ms_prev(mibc_engine *e, mibc_stage_ms *out);
MIBC_API int mibc_set_profile(mibc_engine *e, int level);                   /* 0 off, 1 per-stage events, 2 + roctx ranges around the
                                                                      stages (utils::ScopedProfileRange,
                                                                      torch_utils/gpu_profiling.h:32-99) for rocprofv3
                                                                      --marker-trace */

/* ---- parity taps (test-only; copy an intermediate of the LAST call to the host) ----
 * tap: 0 conv1 out [N,T_in,16] f16 | 1 conv2 out (padded rows) | 2 conv3 out [T,N,C] f16
 *      3 LSTM stack out [T,N,C] f16 | 4 back-guides [N,T+1,S] f32 (first decode sub-batch)
 *      5 per-block quality prob [N,T] f32 */
MIBC_API int mibc_debug_tap(mibc_engine *e, int tap, void *host_dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* MIBC_H */
