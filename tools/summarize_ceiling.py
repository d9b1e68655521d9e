#!/usr/bin/env python3
"""Condense the output of `tools/ceiling.sh run` (gpurun_out/ceiling) into profiles/<tag>_ceiling.json:
per knock-out build and per wave layout the time per launch, the cycles the kernel was resident on the
chip (GRBM_GUI_ACTIVE / 8 XCDs) and the clock that follows from the two; the microbenchmark output as text.

    python tools/summarize_ceiling.py [tag]        (tag defaults to r2)
"""
import collections
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "ceiling")
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
FLOPS = 4.0 * 8 * 96 * 4096 * 4096 * 128

KNOCK = {0: "nothing (the shipped kernel, eight-wave layout)", 1: "exponentials", 2: "K/V staging (global -> LDS)", 4: "LDS fragment reads",
         8: "conversion of P to fp16", 16: "row sums", 32: "running maximum", 64: "score multiply",
         127: "all of the above (matrix instructions and barriers only)"}


def counters(d):
    agg = collections.defaultdict(list)
    dur = []
    for r in csv.DictReader(open(os.path.join(SRC, d, "p_counter_collection.csv"))):
        if "attn_fwd" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    out = {k: sum(v) / len(v) for k, v in agg.items()}
    out["profiled_ns_per_launch"] = sum(dur) / len(dur)
    return out


def entry(label, line, d):
    ms = float(re.search(r"launches ([\d.]+) ms", line).group(1))
    c = counters(d)
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    e = {"what": label, "ms_per_launch (40 back-to-back launches, HIP events)": ms, "tflops": round(FLOPS / ms / 1e9, 1),
         "resident_cycles_per_launch (GRBM_GUI_ACTIVE / 8)": round(cyc),
         "clock_ghz (cycles / profiled kernel duration)": round(cyc / c["profiled_ns_per_launch"], 3)}
    if "SQ_INSTS_LDS" in c:
        e["lds_instructions_per_launch"] = round(c["SQ_INSTS_LDS"])
    return e


knock = []
for line in open(os.path.join(SRC, "knockout.log")):
    m = int(re.match(r"\[knockout (\d+)\]", line).group(1))
    knock.append(dict(mask=m, **entry(KNOCK[m], line, f"ko{m}")))
base = knock[0]["tflops"]
for k in knock:
    k["speedup_vs_shipped"] = round(k["tflops"] / base, 3)
knock4w = []
p4 = os.path.join(SRC, "knockout4w.log")
if os.path.exists(p4):
    for line in open(p4):
        m = int(re.match(r"\[knockout4w (\d+)\]", line).group(1))
        knock4w.append(dict(mask=m, **entry(KNOCK[m].replace("eight-wave", "four-wave"), line, f"ko4w{m}")))
    for k in knock4w:
        k["speedup_vs_four_wave_baseline"] = round(k["tflops"] / knock4w[0]["tflops"], 3)
layouts = []
for line in open(os.path.join(SRC, "layouts.log")):
    n = int(re.match(r"\[wide_max_n (\d+)\]", line).group(1))
    layouts.append(entry("eight waves x 32 query rows (two per SIMD)" if n > 4096 else "four waves x 64 query rows (one per SIMD)",
                         line, f"wide{n}"))
out = {
    "workload": "(8,96,4096,128) fp16, mma_qreg_async",
    "commands": ["bash tools/ceiling.sh build   (CPU: -DFLASH_KNOCKOUT=<mask> libraries, microbenchmarks)",
                 "bash tools/ceiling.sh run     (GPU: tests/native/selftest bench 8 96 4096 128 40 11 per library, one rocprofv3 "
                 "--pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace pass each; FLASH_HIP_WIDE_MAX_N picks the wave layout)"],
    "wave_layouts": layouts,
    "wave_layouts_reading": "the eight-wave layout needs about a tenth fewer cycles for the same launch and the chip "
                            "answers with a clock about a tenth lower: on a box at its power cap the launch takes "
                            "about the same time (within 1-2 %).  Under this kernel the chip is limited by power, "
                            "not by issue slots.  (Boxes that hold their clock turn the saved cycles into time: up "
                            "to +4 % was measured for the eight-wave layout.)",
    "knockouts": knock,
    "knockouts_four_wave_layout (FLASH_HIP_WIDE_MAX_N=0; nothing, LDS fragment reads)": knock4w,
    "knockouts_reading": "timing only, results are wrong.  Builds that freeze matrix operands (4, 8, 64, 127: stale "
                         "fragments, constant or saturated P) also lower the switching activity of the matrix pipe "
                         "itself, so they overstate what the removed work costs; 127 runs above the rate of a bare "
                         "MFMA loop on random operands for that reason.",
    "knockouts_four_wave_reading": "the four-wave layout issues half as many LDS instructions per launch as the "
                                   "eight-wave one, yet leaving ALL fragment reads out buys almost the same there "
                                   "(compare speedup_vs_four_wave_baseline with mask 4 above, and the clocks): "
                                   "what this knock-out removes is mostly not the LDS traffic but the switching "
                                   "activity of matrix operands that no longer change, so the LDS reads themselves "
                                   "are worth at most the difference between the two layouts' gains, a few per cent.",
    "microbench_output": open(os.path.join(SRC, "microbench.log")).read().split("\n"),
}
dst = os.path.join(ROOT, "profiles", f"{tag}_ceiling.json")
json.dump(out, open(dst, "w"), indent=1)
for k in knock:
    print(k["mask"], k["tflops"], k["speedup_vs_shipped"], k["clock_ghz (cycles / profiled kernel duration)"], k["resident_cycles_per_launch (GRBM_GUI_ACTIVE / 8)"])
for l in layouts:
    print(l)
