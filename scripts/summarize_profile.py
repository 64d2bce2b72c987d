#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (scripts/profile_round.sh) into the committed summaries under profiles/.

usage: summarize_profile.py <tag> <name> <stats_dir> [<fetch_dir> <write_dir> [<sq_dir>]] [--steps N]
  -> profiles/<tag>_<name>_kernel_stats.csv   (the rocprofv3 --stats table, names truncated)
     profiles/<tag>_<name>_summary.json       (per vnp_ kernel: calls, average duration; HBM traffic per launch from the FETCH_SIZE /
                                               WRITE_SIZE passes; MFMA-pipe busy fraction, wave states and the clock held during
                                               the kernel from the SQ / GRBM pass; achieved TB/s where traffic is known)
Counter conventions (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are in KiB, and on gfx950 FETCH_SIZE reports half of a wide
coalesced streaming read, so it is doubled; SQ_VALU_MFMA_BUSY_CYCLES counts SIMD cycles summed over all SIMDs (16 per
v_mfma_f32_16x16x32_bf16); GRBM_GUI_ACTIVE is the sum over the 8 XCDs, so cycles = GRBM_GUI_ACTIVE / 8 and the clock held during a
dispatch is cycles / duration; the wave-state counters (SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_*) count quad-cycles.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

argv = sys.argv[1:]
steps = None
if "--steps" in argv:
    i = argv.index("--steps"); steps = float(argv[i + 1]); del argv[i:i + 2]
tag, name, stats_dir = argv[0], argv[1], argv[2]
fetch_dir = argv[3] if len(argv) > 4 else None
write_dir = argv[4] if len(argv) > 4 else None
sq_dir = argv[5] if len(argv) > 5 else None
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_dir = os.path.join(ROOT, "profiles")
os.makedirs(out_dir, exist_ok=True)
N_SIMD = 256 * 4


def one(pattern):
    f = glob.glob(pattern, recursive=True)
    return f[0] if f else None


def short(kernel_name):
    return kernel_name.split("(")[0][:90]


stats = one(os.path.join(stats_dir, "**", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(out_dir, f"{tag}_{name}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
summary = {"tag": tag, "workload": name, "source": "rocprofv3 --kernel-trace --stats (+ separate --pmc passes)", "kernels": {}}
total_ns = sum(float(r["TotalDurationNs"]) for r in rows)
summary["all_kernels_ms"] = total_ns / 1e6
if steps:
    summary["steps_profiled_incl_warmup"] = steps
    summary["all_kernels_ms_per_step"] = total_ns / 1e6 / steps
for r in rows:
    if "vnp_" in r["Name"].split("(")[0]:
        k = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "total_ms": float(r["TotalDurationNs"]) / 1e6,
             "pct": float(r["Percentage"])}
        if steps:
            k["ms_per_step"] = k["total_ms"] / steps
        summary["kernels"][short(r["Name"])] = k


def pmc(dirname):
    """kernel -> counter -> [sum of values, dispatches]; kernel -> [sum of dispatch durations ns, dispatches]"""
    f = one(os.path.join(dirname, "**", "*counter_collection.csv"))
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, set()])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        if r["Dispatch_Id"] not in dur[k][1]:
            dur[k][1].add(r["Dispatch_Id"])
            dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return acc, dur


if fetch_dir:
    fe, _ = pmc(fetch_dir)
    wr, _ = pmc(write_dir)
    traffic = {}
    for k in fe:
        if "vnp_" not in k:
            continue
        n = max(fe[k]["FETCH_SIZE"][1], 1)
        rd = 2.0 * fe[k]["FETCH_SIZE"][0] * 1024 / n
        wt = wr[k]["WRITE_SIZE"][0] * 1024 / max(wr[k]["WRITE_SIZE"][1], 1) if k in wr else 0.0
        t = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wt, "hbm_bytes_per_launch": rd + wt}
        if k in summary["kernels"]:
            t["TBps_at_avg_duration"] = (rd + wt) / (summary["kernels"][k]["avg_us"] * 1e-6) / 1e12
        traffic[k] = t
    summary["traffic"] = traffic
    # the whole step through the fabric: every kernel's bytes per launch x its launches per step (launch counts from the --stats pass)
    if steps:
        tot = sum(t["hbm_bytes_per_launch"] * summary["kernels"][k]["calls"] / steps for k, t in traffic.items() if k in summary["kernels"])
        summary["step_fabric_bytes"] = tot
        summary["step_fabric_note"] = ("sum over the step's kernels of (FETCH_SIZE*2 + WRITE_SIZE) per launch x launches per step; divided by the "
                                       "step time it is the step's average HBM-side rate (bench.py: hbm_roofline)")
    summary["traffic_note"] = ("FETCH_SIZE*2*1024 + WRITE_SIZE*1024 bytes per launch, separate --pmc passes (gfx950 correction per "
                               "MI355X_MICROARCH.md; L2-miss traffic: Infinity-Cache hits are included); TBps uses the un-profiled average "
                               "duration of the --stats pass")

if sq_dir:
    sq, dur = pmc(sq_dir)
    pipes = {}
    for k, c in sq.items():
        if "vnp_" not in k or "GRBM_GUI_ACTIVE" not in c:
            continue
        n = len(dur[k][1])
        t_ns = dur[k][0]
        cycles = c["GRBM_GUI_ACTIVE"][0] / 8.0                       # summed over dispatches
        e = {"launches": n, "avg_us_in_this_pass": t_ns / n / 1e3, "clock_GHz": cycles / t_ns if t_ns > 0 else None}
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and cycles > 0:
            e["mfma_pipe_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / (cycles * N_SIMD)
        wc = c.get("SQ_WAVE_CYCLES", [0.0])[0]
        if wc > 0:
            for key, cn in (("wave_frac_waiting_waitcnt_or_barrier", "SQ_WAIT_ANY"), ("wave_frac_issue_stalled", "SQ_WAIT_INST_ANY"),
                            ("wave_frac_issuing", "SQ_ACTIVE_INST_ANY"), ("wave_frac_issuing_valu", "SQ_ACTIVE_INST_VALU")):
                if cn in c:
                    e[key] = c[cn][0] / wc
        e["raw_per_launch"] = {cn: v[0] / max(n, 1) for cn, v in c.items()}
        pipes[k] = e
    summary["pipes"] = pipes
    summary["pipes_note"] = ("one --pmc pass: mfma_pipe_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs); clock_GHz = "
                             "(GRBM_GUI_ACTIVE / 8) / dispatch duration (reads high on dispatches under ~0.3 ms); wave fractions are "
                             "of SQ_WAVE_CYCLES (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ 1)")
json.dump(summary, open(os.path.join(out_dir, f"{tag}_{name}_summary.json"), "w"), indent=1)
print("wrote", os.path.join(out_dir, f"{tag}_{name}_summary.json"))
