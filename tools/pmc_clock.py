#!/usr/bin/env python3
"""rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_* --kernel-trace pass of tools/stage_times.py -> per-kernel effective clock and wave-cycle split.
usage: pmc_clock.py <dir> <out.json> <model> <N>
GRBM_GUI_ACTIVE counts shader-engine-clock cycles while the GPU is busy; rocprofv3 reports it summed over the XCDs it sampled, so the
per-XCD value = counter / n_xcd where n_xcd is inferred as the integer that puts the clock between 0.5 and 2.6 GHz (8 on MI355X).
SQ counters are summed over all CUs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)."""
import collections
import csv
import glob
import json
import sys


def main():
    d, out, model, n = sys.argv[1:5]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (r["Kernel_Name"], float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    res = {}
    for k, v in acc.items():
        if "rocclr" in k or "mfma_ref" in k:
            continue
        nl = len(disp[k])
        ns = sum(dur[i][1] for i in disp[k] if i in dur) / max(1, nl)
        if ns < 2e5:          # kernels under 0.2 ms: the clock estimate is noise
            continue
        c = {a: b / nl for a, b in v.items()}
        g = c.get("GRBM_GUI_ACTIVE", 0.0)
        nx = 1
        for cand in (1, 2, 4, 8, 16, 32):
            if 0.5 <= g / cand / ns <= 2.6:
                nx = cand
                break
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        e = {"launches": nl, "avg_ms": ns / 1e6, "clock_ghz": g / nx / ns, "grbm_instances": nx,
             "counters_per_launch": c}
        if wc > 0:
            e["wave_cycle_split"] = {"wait_any": c.get("SQ_WAIT_ANY", 0) / wc, "wait_inst_any": c.get("SQ_WAIT_INST_ANY", 0) / wc,
                                     "active_inst_any": c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                                     "wait_inst_lds_of_wave_cycles": c.get("SQ_WAIT_INST_LDS", 0) / wc}
        if g > 0:
            cyc = g / nx     # shader cycles of the launch
            e["mfma_busy_frac_of_simd_cycles"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 256 * 4)
            e["lds_active_frac_of_cu_cycles"] = c.get("SQ_LDS_IDX_ACTIVE", 0) / (cyc * 256)
            e["lds_bank_conflict_frac_of_cu_cycles"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / (cyc * 256)
        res[k] = e
    json.dump({"_note": __doc__, "workload": {"model": model, "N": int(n)}, "kernels": res}, open(out, "w"), indent=1)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"]):
        print(f"{k[:44]:44s} x{e['launches']:<3d} {e['avg_ms']:8.2f} ms  {e['clock_ghz']:.2f} GHz  mfma {e.get('mfma_busy_frac_of_simd_cycles', 0):.2f}  "
              f"lds {e.get('lds_active_frac_of_cu_cycles', 0):.2f} (+conf {e.get('lds_bank_conflict_frac_of_cu_cycles', 0):.2f})  "
              f"wait {e.get('wave_cycle_split', {}).get('wait_any', 0):.2f} stall {e.get('wave_cycle_split', {}).get('wait_inst_any', 0):.2f}")


if __name__ == "__main__":
    main()
