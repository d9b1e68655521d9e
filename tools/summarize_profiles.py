#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/gpu_profile.sh (gpurun_out/prof/<workload>) into the files kept
under profiles/:  <tag>_<workload>_kernel_stats.csv, <tag>_<workload>_pmc_summary.json,
<tag>_<workload>_bench_under_rocprof.json and traffic.json (HBM bytes per launch, read by bench.py).

    python tools/summarize_profiles.py [tag] [workload ...]     (defaults: r2, every workload found)
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
SHAPES = {"headline": (8, 96, 4096, 128), "gpt3_1k": (4, 96, 1024, 128), "d64": (16, 32, 2048, 64)}
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
workloads = sys.argv[2:] or [w for w in SHAPES if os.path.isdir(os.path.join(SRC, w))]

traffic_path = os.path.join(DST, "traffic.json")
try:
    traffic = json.load(open(traffic_path))
except (OSError, ValueError):
    traffic = {}

for w in workloads:
    B, H, N, D = SHAPES[w]
    src = os.path.join(SRC, w)
    rows = list(csv.reader(open(os.path.join(src, "kt", "bench_kernel_stats.csv"))))
    with open(os.path.join(DST, f"{tag}_{w}_kernel_stats.csv"), "w", newline="") as f:
        wr = csv.writer(f)
        for r in rows:
            r = list(r)
            r[0] = r[0][:120]          # torch's RNG kernels have kilobyte-long names
            wr.writerow(r)

    pmc = {}
    kernel = None
    for d in ("fetch", "write", "sq", "grbm"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(src, d, "p_counter_collection.csv"))):
            if "attn_fwd" in r["Kernel_Name"]:
                kernel = r["Kernel_Name"]
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v), "pass": d}

    m = lambda k: pmc[k]["mean_per_launch"]
    waves_per_simd = 2 if "m16w" in kernel else 1
    fetch, write = m("FETCH_SIZE") * 1024, m("WRITE_SIZE") * 1024
    hbm = 2 * fetch + write
    bench = json.load(open(os.path.join(src, "bench_prof.json")))
    kernel_ms = bench["roofline"]["kernel_ms"]
    out = {
        "command": f"rocprofv3 --pmc <counters> --output-format csv -- tests/native/selftest bench {B} {H} {N} {D} 5 11"
                   "  (one pass per counter group, tools/gpu_profile.sh)",
        "kernel": kernel, "workload": f"({B},{H},{N},{D}) fp16", "counters": pmc,
        "derived": {
            "hbm_read_bytes (2 x FETCH_SIZE x 1024; gfx950 counts wide streaming reads at half)": 2 * fetch,
            "hbm_write_bytes (WRITE_SIZE x 1024)": write,
            "hbm_bytes_per_launch": hbm,
            "algorithmic_bytes (Q+K+V read once, O written once)": 4 * B * H * N * D * 2,
            "waves_per_simd": waves_per_simd,
            "mfma_pipe_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES / waves_per_simd))":
                m("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * m("SQ_WAVE_CYCLES") / waves_per_simd),
            "active_inst_frac": m("SQ_ACTIVE_INST_ANY") / m("SQ_WAVE_CYCLES"),
            "wait_inst_frac": m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES"),
            "wait_any_frac": m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"),
            "gui_active_cycles_per_xcd": m("GRBM_GUI_ACTIVE") / 8,
            "effective_clock_ghz (gui_active_cycles_per_xcd / kernel_ms of the bench run under the profiler)":
                m("GRBM_GUI_ACTIVE") / 8 / (kernel_ms * 1e6),
            "lds_bank_conflict_cycles": m("SQ_LDS_BANK_CONFLICT"),
        },
    }
    json.dump(out, open(os.path.join(DST, f"{tag}_{w}_pmc_summary.json"), "w"), indent=1)
    traffic[w] = {"hbm_bytes_per_launch": hbm, "source": f"profiles/{tag}_{w}_pmc_summary.json"}
    shutil.copy(os.path.join(src, "bench_prof.json"), os.path.join(DST, f"{tag}_{w}_bench_under_rocprof.json"))
    print(w, json.dumps(out["derived"], indent=1))

json.dump(traffic, open(traffic_path, "w"), indent=1)
full = os.path.join(ROOT, "gpurun_out", "full", "bench.json")
if os.path.exists(full):
    shutil.copy(full, os.path.join(DST, f"{tag}_bench.json"))
