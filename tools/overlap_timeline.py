#!/usr/bin/env python3
"""Timeline of one `rocprofv3 --kernel-trace` run of bench.py --decode-overlap 1: are the decoder kernels of batch i
inside the network launches of batch i + 1, and what does that do to their durations?

    python tools/overlap_timeline.py <dir with *_kernel_trace.csv> <out.txt>

Prints, for the last complete step: every kernel with queue, start and end (ms, relative), and the totals — decoder time,
decoder time that lies inside a network kernel's [start, end), per-kernel averages to compare with the serial run.
"""
import csv
import glob
import os
import sys

DEC = ("bwd_scan2_kernel", "beam_search64_kernel", "beam_search_kernel", "posts_qual_kernel")
NET = ("lstm_layer", "wsgemm_kernel", "conv12_kernel", "gemm256x_kernel", "tx_layer_kernel", "window_attention", "gemm_dma_kernel")


def short(name):
    for k in DEC + NET:
        if k in name:
            return k
    return name[:40]


def main(d, out_path):
    files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit("no *kernel_trace.csv under " + d)
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
    rows.sort()
    dec = [r for r in rows if r[3] in DEC]
    net = [r for r in rows if r[3] in NET]
    lines = []
    # overlap of every decoder kernel with the union of network kernels (network kernels of ONE stream never overlap each other)
    inside_total = 0
    dec_total = 0
    per = {}
    for s, e, q, n in dec:
        inside = 0
        for ns, ne, nq, nn in net:
            lo, hi = max(s, ns), min(e, ne)
            if hi > lo:
                inside += hi - lo
        inside_total += inside
        dec_total += e - s
        a = per.setdefault(n, [0, 0, 0])
        a[0] += 1
        a[1] += e - s
        a[2] += inside
    netper = {}
    for s, e, q, n in net:
        a = netper.setdefault(n, [0, 0])
        a[0] += 1
        a[1] += e - s
    lines.append("kernel                         calls   avg ms   of which inside a network kernel")
    for n, (c, t, i) in sorted(per.items()):
        lines.append("%-30s %5d %8.3f   %5.1f %%" % (n, c, t / c / 1e6, 100.0 * i / max(t, 1)))
    for n, (c, t) in sorted(netper.items()):
        lines.append("%-30s %5d %8.3f" % (n, c, t / c / 1e6))
    lines.append("decoder kernels: %.1f ms in total, %.1f ms (%.1f %%) of it while a network kernel was running"
                 % (dec_total / 1e6, inside_total / 1e6, 100.0 * inside_total / max(dec_total, 1)))
    queues = sorted({r[2] for r in rows})
    lines.append("queues seen: " + ", ".join("%s (%d kernels)" % (q, sum(1 for r in rows if r[2] == q)) for q in queues))
    # timeline of the last 2 steps' worth of kernels: from the third-last first-conv launch on
    conv_starts = [r[0] for r in rows if r[3] in ("conv12_kernel",)]
    t0 = conv_starts[-2] if len(conv_starts) >= 2 else rows[0][0]
    lines.append("")
    lines.append("timeline from the second-last batch's first network kernel (ms):   queue   start      end   kernel")
    for s, e, q, n in rows:
        if s >= t0:
            lines.append("  q%-4s %9.3f %9.3f   %s%s" % (q, (s - t0) / 1e6, (e - t0) / 1e6, "    " if n in NET else "", n))
    txt = "\n".join(lines) + "\n"
    open(out_path, "w").write(txt)
    sys.stdout.write(txt[:3000])


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
