#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/<dir>) into small, reviewable files under profiles/.

  python profiles/summarize.py stats gpurun_out/prof_r1 profiles/r1_bench_q1_kernel_stats.csv
  python profiles/summarize.py pmc   gpurun_out/pmc_fetch_r1 gpurun_out/pmc_write_r1 "hdb_scan_kernel<__half, 1, 1" \
         profiles/r1_bench_q1_hbm_traffic.json
  python profiles/summarize.py mfma  "Li16ELi2ELi384" 1.966e12 profiles/r2_q256_pmc.json gpurun_out/pmc_q256_a gpurun_out/pmc_q256_b ...

The mfma mode condenses one or more `rocprofv3 --pmc ...` passes (each run as `rocprofv3 --pmc <counters> -d <dir>
--output-format csv -- python3 tools/run_q256.py`, the program directly after `--`) over the dispatches whose kernel
name contains the needle: average of every counter, average dispatch duration (End - Start timestamp of the counter
rows), and the derived figures bench.py reports for the batched leg:
  clock_ghz      = GRBM_GUI_ACTIVE / 8 XCDs / duration        (MI355X_MICROARCH.md, DVFS give-back)
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES x 4): busy cycles of the matrix pipes over the cycles
                   of the busy CUs' 4 SIMDs; cross-checked against the instruction count (FLOP / 16384 per
                   v_mfma_f32_16x16x32_f16 x 16 cycles) over 1024 SIMDs x (GRBM_GUI_ACTIVE / 8).

Kernel names are cut to 110 characters (torch's RNG kernels have 6 KB names).  The pmc mode applies the
gfx950 corrections of MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB, and
FETCH_SIZE reports exactly half of the bytes of a wide coalesced streaming read -> doubled.
"""
import csv
import glob
import json
import os
import sys


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits[0]


def stats(src, dst):
    rows = list(csv.DictReader(open(find(src, "_kernel_stats.csv"))))
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"], r["StdDev"]])
    print("wrote", dst, len(rows), "kernels")


def legs(src, dst, ratio=1.35):
    """Per-launch-size statistics out of the kernel TRACE (one row per dispatch): the default bench command runs the same kernel
    on several matrices (headline 10M rows, the 1.25M-row shard leg, the small leg), which the plain --stats table averages together.
    The dispatches of each hdb_* kernel are sorted by duration and cut into clusters wherever two neighbours differ by more than
    `ratio`; every cluster is one line: name, calls, average / min / max ns.  bench.py's kernel_us of a leg is the average of the
    cluster with that leg's launch count (200 headline calls + warm-up and api legs on the same matrix share a cluster)."""
    import subprocess
    rows = []
    for path in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(path)))
    by = {}
    for r in rows:
        name = r["Kernel_Name"]
        if "hdb_" not in name:
            continue
        by.setdefault(name, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    names = sorted(by)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    with open(dst, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Cluster", "Calls", "AverageNs", "MinNs", "MaxNs"])
        for name, dn in zip(names, dem):
            d = sorted(by[name])
            cl, cur = [], [d[0]]
            for x in d[1:]:
                if x > cur[-1] * ratio:
                    cl.append(cur); cur = [x]
                else:
                    cur.append(x)
            cl.append(cur)
            short = (dn or name).split("(")[0][:110]
            for i, c in enumerate(cl):
                w.writerow([short, i, len(c), round(sum(c) / len(c), 1), c[0], c[-1]])
    print("wrote", dst)


def pmc(fetch_dir, write_dir, needle, dst):
    def avg(d, counter):
        vals = [float(r["Counter_Value"])
                for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)   # one file per process
                for r in csv.DictReader(open(path))
                if needle in r["Kernel_Name"] and r["Counter_Name"] == counter]
        return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)
    f, nf = avg(fetch_dir, "FETCH_SIZE")
    w, nw = avg(write_dir, "WRITE_SIZE")
    out = {
        "kernel": needle, "dispatches_fetch": nf, "dispatches_write": nw,
        "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
        "read_bytes_corrected": f * 1024 * 2, "write_bytes": w * 1024,
        "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024,
        "correction": "FETCH_SIZE x1024 x2 (gfx950 counts 128-B requests as 64 B on wide streaming reads), WRITE_SIZE x1024",
    }
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out))


def mfma(needle, flop, dst, dirs):
    counters, durs, regs = {}, [], {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(path)):
                if needle not in r["Kernel_Name"]:
                    continue
                counters.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    durs.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
                regs = {"vgpr": r["VGPR_Count"], "agpr": r["Accum_VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"],
                        "workgroup": r["Workgroup_Size"], "grid": r["Grid_Size"]}
    if not durs:
        raise SystemExit(f"no dispatch matching {needle!r}")
    avg = {k: sum(v) / len(v) for k, v in counters.items()}
    dur_ns = sum(durs) / len(durs)
    out = {"kernel_needle": needle, "dispatches": len(durs), "duration_us_profiled": dur_ns / 1e3, "launch": regs,
           "counters_avg_per_dispatch": avg, "passes": dirs}
    gui = avg.get("GRBM_GUI_ACTIVE")
    if gui:
        cyc = gui / 8.0
        out["clock_ghz"] = cyc / dur_ns
        out["clock_formula"] = "GRBM_GUI_ACTIVE / 8 / duration_ns"
        if flop:
            mfma_cycles = flop / 16384.0 * 16.0
            out["mfma_busy_frac_from_flop"] = mfma_cycles / (1024.0 * cyc)
            out["tflops_profiled"] = flop / dur_ns / 1e3
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and avg.get("SQ_BUSY_CU_CYCLES"):
        out["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (avg["SQ_BUSY_CU_CYCLES"] * 4.0)
        out["mfma_busy_formula"] = "SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES * 4)"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and gui:
        out["mfma_busy_frac_vs_gui"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * gui / 8.0)
    if avg.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
            if k in avg:
                out[k + "_share_of_wave_cycles"] = avg[k] / avg["SQ_WAVE_CYCLES"]
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "legs":
        legs(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "mfma":
        mfma(sys.argv[2], float(sys.argv[3]), sys.argv[4], sys.argv[5:])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5])
