#!/usr/bin/env python3
"""Scalability plot from a sweep file (counterpart of the reference's img/scalability.png, which is
drawn from the `data` dict its benchmarks/bench.py:89-111 prints): kernel time against sequence
length, log-log, one line per implementation.

    python benchmarks/plot_sweep.py [profiles/r1_sweep.json] [profiles/r1_scalability.png]
"""
import json
import os
import sys

import matplotlib

matplotlib.use("Agg")
import matplotlib.pyplot as plt  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r1_sweep.json")
dst = sys.argv[2] if len(sys.argv) > 2 else os.path.splitext(src)[0].replace("sweep", "scalability") + ".png"

res = json.load(open(src))
ms, tf = res["ms"], res["tflops"]
seq = ms["Sequence Length"]
fig, (a0, a1) = plt.subplots(1, 2, figsize=(11, 4.2))
for name, vals in ms.items():
    if name == "Sequence Length":
        continue
    pts = [(n, t) for n, t in zip(seq, vals) if t > 0]      # 0 = ran out of memory
    a0.plot([p[0] for p in pts], [p[1] for p in pts], marker="o", label=name)
    a1.plot([p[0] for p in pts], [tf[name][seq.index(p[0])] for p in pts], marker="o", label=name)
shape = res.get("shape", {})
for ax, ylab in ((a0, "time per call, ms"), (a1, "TFLOP/s (4 B H N^2 d / t)")):
    ax.set_xscale("log", base=2)
    ax.set_xlabel("sequence length")
    ax.set_ylabel(ylab)
    ax.grid(True, which="both", alpha=0.3)
a0.set_yscale("log")
a0.legend(fontsize=8)
fig.suptitle(f"attention forward on one MI355X, B={shape.get('B')} H={shape.get('H')} d={shape.get('d')}")
fig.tight_layout()
fig.savefig(dst, dpi=110)
print("wrote", dst)
