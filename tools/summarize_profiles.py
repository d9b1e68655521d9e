#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/gpu_profile.sh (gpurun_out/prof) into the files kept under
profiles/:  <tag>_kernel_stats.csv, <tag>_pmc_summary.json, traffic.json.

    python tools/summarize_profiles.py [tag]        (tag defaults to r1)
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
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
B, H, N, D = 8, 96, 4096, 128

rows = list(csv.reader(open(os.path.join(SRC, "kt", "bench_kernel_stats.csv"))))
with open(os.path.join(DST, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r = list(r)
        r[0] = r[0][:120]          # torch's RNG kernels have kilobyte-long names
        w.writerow(r)

pmc = {}
kernel = None
for d in ("fetch", "write", "sq", "grbm"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(SRC, d, "p_counter_collection.csv"))):
        if "attn_fwd" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"]
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        pmc[k] = {"mean_per_launch": sum(v) / len(v), "launches": len(v), "pass": d}

m = lambda k: pmc[k]["mean_per_launch"]
fetch, write = m("FETCH_SIZE") * 1024, m("WRITE_SIZE") * 1024
hbm = 2 * fetch + write
out = {
    "command": "rocprofv3 --pmc <counters> --output-format csv -- tests/native/selftest bench 8 96 4096 128 5 11"
               "  (one pass per counter group, tools/gpu_profile.sh)",
    "kernel": kernel, "workload": f"({B},{H},{N},{D}) fp16", "counters": pmc,
    "derived": {
        "hbm_read_bytes (2 x FETCH_SIZE x 1024; gfx950 counts wide streaming reads at half)": 2 * fetch,
        "hbm_write_bytes (WRITE_SIZE x 1024)": write,
        "hbm_bytes_per_launch": hbm,
        "algorithmic_bytes (Q+K+V read once, O written once)": 4 * B * H * N * D * 2,
        "mfma_pipe_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_WAVE_CYCLES))":
            m("SQ_VALU_MFMA_BUSY_CYCLES") / (4 * m("SQ_WAVE_CYCLES")),
        "active_inst_frac": m("SQ_ACTIVE_INST_ANY") / m("SQ_WAVE_CYCLES"),
        "wait_inst_frac": m("SQ_WAIT_INST_ANY") / m("SQ_WAVE_CYCLES"),
        "wait_any_frac": m("SQ_WAIT_ANY") / m("SQ_WAVE_CYCLES"),
        "gui_active_cycles_per_xcd": m("GRBM_GUI_ACTIVE") / 8,
        "lds_bank_conflict_cycles": m("SQ_LDS_BANK_CONFLICT"),
    },
}
json.dump(out, open(os.path.join(DST, f"{tag}_pmc_summary.json"), "w"), indent=1)
json.dump({"headline": {"hbm_bytes_per_launch": hbm, "source": f"profiles/{tag}_pmc_summary.json"}},
          open(os.path.join(DST, "traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(SRC, "bench_prof.json"), os.path.join(DST, f"{tag}_bench_under_rocprof.json"))
full = os.path.join(ROOT, "gpurun_out", "full", "bench.json")
if os.path.exists(full):
    shutil.copy(full, os.path.join(DST, f"{tag}_bench.json"))
print(json.dumps(out["derived"], indent=1))
