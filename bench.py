#!/usr/bin/env python3
"""bench.py — Samples/s of the simplex-basecalling hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--model hac|sup|tiny] [--also-sup]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the whole hot path (conv -> 5x LSTM -> CRF head -> beam-search decode)
over one batch of synthetic 5 kHz signal chunks that is ALREADY RESIDENT IN HBM when the timed
region starts (mibc_call_device).  Workload at N=1 = BASELINE.json configs[1]:
dna_r10.4.1_e8.2_400bps_hac@v4.3.0 topology, chunksize 10000 -> 9996 after normalisation.
Reads shard embarrassingly: every rank owns its GPU, its engine and its chunks; there is no
data-path collective (weak scaling).  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline      the dominant kernel (lstm_layer_kernel): algorithmic MFMA flops per launch
                (N*T*2*(4C*2C)) / mean launch duration measured with HIP events on the engine's
                stream inside the timed region, against the 2.5 PFLOP/s dense f16 MFMA peak.
  cpu_baseline  the REFERENCE's own CPU path (oracle/_ref = reference sources compiled in place,
                libtorch CPU f32, 1 intra-op thread per runner as torch_utils.cpp:20 does) timed on
                this host on a bounded sample of the same workload.  Rank 0, N=1 only.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F16_PEAK = 2.5e15  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def pmc_traffic(kernel_prefix, model, n, t_in):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r*_pmc_traffic_*.json: separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 correction on
    FETCH_SIZE).  Only reported when the profiled workload matches this run; else null."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_*.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        w = d.get("workload", {})
        if (w.get("model"), w.get("N"), w.get("T_in")) != (model, n, t_in):
            continue
        for k, v in d.get("kernels", {}).items():
            if kernel_prefix in k:
                best = {"hbm_bytes": v["hbm_bytes_corrected"], "source": os.path.relpath(path, ROOT)}
    return best


def lstm_flops_per_launch(cfg, n, t):
    c = cfg.lstm_size
    return float(n) * t * 2.0 * (4 * c) * (2 * c)


def tx_layer_flops(cfg):
    """2*MAC per token per encoder layer: qkv, attention over the window, out-proj, fc1, fc2."""
    t = cfg.tx
    c, f = t.d_model, t.dim_feedforward
    win = t.attn_window[0] + t.attn_window[1] + 1
    return 2.0 * (c * 3 * c + 2 * win * c + c * c + c * 2 * f + f * c)


def network_flops_per_sample(cfg):
    """2*MAC of conv + LSTM + head per raw sample."""
    if cfg.tx is not None:
        per_tok = 0.0
        up = cfg.conv_stride
        for cv in cfg.convs:
            up //= cv.stride
            per_tok += 2.0 * cv.insize * cv.size * cv.winlen * up
        t = cfg.tx
        per_tok += t.depth * tx_layer_flops(cfg)
        per_tok += 2.0 * t.d_model * t.up_scale_factor * t.d_model
        per_tok += 2.0 * t.d_model * cfg.outsize * t.up_scale_factor
        return per_tok / cfg.conv_stride
    per_step = 0.0
    s = 1
    for cv in cfg.convs:
        s *= cv.stride
    for cv, up in zip(cfg.convs, [cfg.stride, cfg.stride, 1]):
        per_step += 2.0 * cv.insize * cv.size * cv.winlen * up
    per_step += cfg.lstm_layers * 2.0 * (4 * cfg.lstm_size) * (2 * cfg.lstm_size)
    per_step += 2.0 * cfg.lstm_size * cfg.outsize
    return per_step / cfg.stride


def cpu_baseline(cfg, ws, t_in, budget_s=25.0):
    """Reference CPU basecaller on this host: R runner threads x (n_per chunks), 1 torch thread
    each (the reference's rule, dorado/basecall/crf_utils.cpp:208-233)."""
    from oracle import oracle_py as O
    from dorado_amd import synth

    kind = "reference" if O.have_ref() else "port"
    cores = os.cpu_count() or 1
    R = max(1, min(cores, 16))
    n_per = 2
    x = synth.make_signal(n_per, t_in, seed=99).astype(np.float32)[:, None, :]
    done = []

    def runner():
        s = O.lstm_crf_forward(cfg, ws, x, use_ref=(kind == "reference"))
        O.decode(s, q_shift=cfg.qbias, q_scale=cfg.qscale, use_ref=(kind == "reference"))
        done.append(n_per)

    if kind == "port":
        R = 1  # the C port parallelises internally with OpenMP
    # warm-up (page in libtorch) on a short chunk
    xs = x[:, :, : 6 * 100]
    O.decode(O.lstm_crf_forward(cfg, ws, xs, use_ref=(kind == "reference")),
             use_ref=(kind == "reference"))
    t0 = time.time()
    rounds = 0
    while True:
        th = [threading.Thread(target=runner) for _ in range(R)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        rounds += 1
        el = time.time() - t0
        if el > budget_s * 0.5 or rounds >= 4:
            break
    el = time.time() - t0
    samples = sum(done) * t_in
    return {"value": samples / el, "unit": "samples/s", "cores": R if kind == "reference" else cores,
            "kind": kind,
            "sample": f"{sum(done)} chunks x {t_in} samples, {R} runner threads x 1 torch thread, "
                      f"forward+decode, {el:.1f}s wall, host has {cores} logical cores"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="chunks per GPU per step (0 = auto)")
    ap.add_argument("--model", default="hac")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--also-sup", type=int, default=1,
                    help="N=1 only: after the headline (hac) run, time 2 steps of the sup@v4.3 shape and "
                         "report it under extra.sup_v43 (BASELINE metric names hac & sup)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from dorado_amd import capi, config, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")

    if args.model == "hac":
        cfg = config.hac_v43()
    elif args.model == "sup":
        cfg = config.sup_v43()
    elif args.model == "sup5":
        cfg = config.sup_v50()
    elif args.model == "tiny":
        cfg = config.tiny(128, 4)
    else:
        raise SystemExit(f"unknown model {args.model}")
    t_in = cfg.chunk_size
    ws = synth.make_weights(cfg, seed=42)
    eng = capi.Engine(cfg, ws, device=local_rank)
    T = eng.output_steps(t_in)

    # auto batch: fill all 256 CUs with 64-chunk LSTM workgroups, bounded by device memory
    n = args.batch
    if n <= 0:
        per_chunk, fixed = eng.query_memory(t_in)
        free_b, total_b = torch.cuda.mem_get_info(local_rank)
        cap = int((free_b * 0.8 - fixed) // per_chunk)
        g = eng.batch_granularity()
        n = max(g, min(256 * g, (cap // g) * g))   # one LSTM workgroup per CU
        if cfg.tx is not None:
            n = min(1024, cap)
    eng.reserve(n, t_in)

    # synthetic signal: 256 distinct seeded chunks tiled to the batch, resident in HBM
    base = synth.make_signal(min(n, 256), t_in, seed=0xD0AD0 + rank)
    x = np.tile(base, ((n + base.shape[0] - 1) // base.shape[0], 1))[:n]
    d_in = eng.device_alloc(x.nbytes)
    d_out = eng.device_alloc(3 * n * T)
    eng.h2d(d_in, x)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.set_profile(1)
    for _ in range(args.warmup):
        eng.call_device(d_in, n, t_in, d_out)
    eng.sync()
    barrier()
    t0 = time.perf_counter()
    lstm_ms = []
    stage = None
    for _ in range(args.steps):
        eng.call_device(d_in, n, t_in, d_out)
        # HIP-event stage times of THIS step on the engine's stream (waits for the step's last
        # event, which the next step would have to wait for anyway: one stream, in order)
        stage = eng.stage_ms()
        lstm_ms.extend(stage["lstm_layer"][: max(1, cfg.lstm_layers)])
    eng.sync()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())

    out = np.zeros((3, n, T), np.int8)
    eng.d2h(out, d_out)
    bases = int(out[0].sum())
    parity = bench_scale_parity(eng, out, d_in, n, t_in, T, base.shape[0])

    if rank == 0:
        total_samples = float(world) * n * t_in * args.steps
        value = total_samples / el
        if cfg.tx is None:
            k_ms = float(np.mean(lstm_ms))
            fl = lstm_flops_per_launch(cfg, n, T)
            kname = "lstm_layer_%s_kernel<%d>" % ("x8" if cfg.lstm_size <= 384 else "xg", cfg.lstm_size)
        else:
            k_ms = float(stage["lstm"])
            fl = n * (T // cfg.tx.up_scale_factor) * tx_layer_flops(cfg) * cfg.tx.depth
            kname = "transformer encoder stack (gemm_dma_kernel + window_attention_v2_kernel)"
        achieved = fl / (k_ms * 1e-3)
        tr = pmc_traffic("lstm_layer_x8", args.model, n, t_in)
        line = {
            "metric": "Samples/s (whole node), simplex basecalling hot path",
            "value": value,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": el / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {
                "workload": f"{cfg.name} (LSTM-CRF, random-init weights), chunksize {t_in}, "
                            f"overlap {cfg.overlap}, batch {n} chunks/GPU, beam 32, inputs resident in HBM",
                "chunks_per_gpu": n, "chunk_size": t_in, "output_steps": T,
                "parallelism": f"{world} independent per-GPU engines, no collective",
                "bases_per_step_emitted": bases / float(n * T),
            },
            "stage_ms_last_step": stage,
            "parity": parity,
            "network_tflops": value * network_flops_per_sample(cfg) / 1e12,
            "roofline": {
                "kernel": kname,
                "bound": "mfma", "achieved": achieved / 1e12, "peak": MFMA_F16_PEAK / 1e12,
                "unit": "TFLOP/s", "frac": achieved / MFMA_F16_PEAK,
                "traffic": (tr or {}).get("hbm_bytes"), "traffic_source": (tr or {}).get("source"),
                "traffic_algorithmic": (2.0 * n * T * cfg.lstm_size * 2) if cfg.tx is None else None,
                "launch_ms": k_ms, "flops_per_launch": fl,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, ws, t_in)
            except Exception as ex:  # the checker must never take the bench line down
                line["cpu_baseline"] = {"value": None, "error": repr(ex)}
    eng.device_free(d_in)
    eng.device_free(d_out)
    eng.close()
    if rank == 0:
        if world == 1 and args.also_sup and args.model == "hac":
            try:
                line["extra"] = {"sup_v43": side_run(capi, config.sup_v43(), synth, local_rank)}
            except Exception as ex:
                line["extra"] = {"sup_v43": {"error": repr(ex)}}
            try:
                line["extra"]["sup_v50"] = side_run(capi, config.sup_v50(), synth, local_rank, n=1024)
            except Exception as ex:
                line["extra"]["sup_v50"] = {"error": repr(ex)}
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


def bench_scale_parity(eng, out, d_in, n, t_in, T, period):
    """Bench-scale output check (outside the timed region).  The batch tiles `period` distinct chunks, so
    (1) every output row i must be byte-identical to row i % period (all three planes: moves, bases,
    qstring), which exercises every index beyond 2^31 elements of the batch-sized buffers, and
    (2) the first `period` rows must equal a separate call on just those `period` chunks (a batch small
    enough for the BASELINE-size parity tests to have validated against the reference)."""
    res = {"batch": n, "distinct_chunks": period}
    if n > period:
        reps = (n + period - 1) // period
        tiled = np.tile(out[:, :period], (1, reps, 1))[:, :n]
        bad_rows = np.nonzero((tiled != out).any(axis=(0, 2)))[0]
        res["tiled_rows_identical"] = bool(bad_rows.size == 0)
        res["tiled_rows_differing"] = int(bad_rows.size)
    g = eng.batch_granularity()
    m = (min(period, n) // g) * g
    if m > 0:
        d_small = eng.device_alloc(3 * m * T)
        eng.call_device(d_in, m, t_in, d_small)   # the first m input rows ARE the distinct chunks
        eng.sync()
        small = np.zeros((3, m, T), np.int8)
        eng.d2h(small, d_small)
        eng.device_free(d_small)
        res["equals_small_batch_call"] = bool((small == out[:, :m]).all())
        res["small_batch"] = m
    res["ok"] = bool(res.get("tiled_rows_identical", True) and res.get("equals_small_batch_call", True))
    return res


def side_run(capi, cfg, synth, device, steps=2, n=None):
    """Secondary measurement (not the headline value): same hot path, another model shape."""
    t_in = cfg.chunk_size
    eng = capi.Engine(cfg, synth.make_weights(cfg, seed=42), device=device)
    T = eng.output_steps(t_in)
    n = n or 256 * eng.batch_granularity()
    eng.reserve(n, t_in)
    base = synth.make_signal(64, t_in, seed=7)
    x = np.tile(base, (n // 64, 1))
    d_in, d_out = eng.device_alloc(x.nbytes), eng.device_alloc(3 * n * T)
    eng.h2d(d_in, x)
    eng.set_profile(1)
    eng.call_device(d_in, n, t_in, d_out)
    eng.sync()
    t0 = time.perf_counter()
    lstm = []
    for _ in range(steps):
        eng.call_device(d_in, n, t_in, d_out)
        st = eng.stage_ms()
        lstm.extend(st["lstm_layer"][: cfg.lstm_layers])
    eng.sync()
    el = time.perf_counter() - t0
    res = {"workload": f"{cfg.name}, chunksize {t_in}, batch {n}", "samples_per_s": n * t_in * steps / el,
           "ms_per_step": el / steps * 1e3, "stage_ms_last_step": st,
           "network_tflops": n * t_in * steps / el * network_flops_per_sample(cfg) / 1e12}
    if cfg.tx is None:
        k_ms = float(np.mean(lstm))
        fl = lstm_flops_per_launch(cfg, n, T)
        res["roofline"] = {"kernel": "lstm_layer_xg_kernel<%d>" % cfg.lstm_size, "bound": "mfma",
                           "achieved": fl / (k_ms * 1e-3) / 1e12, "peak": MFMA_F16_PEAK / 1e12,
                           "unit": "TFLOP/s", "frac": fl / (k_ms * 1e-3) / MFMA_F16_PEAK, "launch_ms": k_ms}
    else:  # encoder stack (18 layers of GEMMs + attention) reported as one stage
        enc_ms = st["lstm"]
        fl = n * (T // cfg.tx.up_scale_factor) * tx_layer_flops(cfg) * cfg.tx.depth
        res["roofline"] = {"kernel": "transformer encoder stack (gemm_dma_kernel + window_attention_v2_kernel)",
                           "bound": "mfma", "achieved": fl / (enc_ms * 1e-3) / 1e12, "peak": MFMA_F16_PEAK / 1e12,
                           "unit": "TFLOP/s", "frac": fl / (enc_ms * 1e-3) / MFMA_F16_PEAK, "stage_ms": enc_ms}
    eng.device_free(d_in)
    eng.device_free(d_out)
    eng.close()
    return res


if __name__ == "__main__":
    main()
