"""GPU: bench.py end to end, including the multi-rank path started from one plain invocation."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(*args, timeout=280):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=timeout)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_two_ranks_from_a_plain_invocation_share_the_gpu():
    """`python bench.py --gpus 2` (no launcher around it): the parent starts both ranks; with the gloo
    backend they rehearse the multi-rank path on one GPU.  Every rank owns its own batch, so the job's
    batch*heads doubles and `value` counts both."""
    one = run_bench("--gpus", "1", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline")
    two = run_bench("--gpus", "2", "--backend", "gloo", "--workload", "tiny", "--steps", "2", "--warmup", "1",
                    "--no-cpu-baseline")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["config"]["global_batch_heads"] == 2 * one["config"]["global_batch_heads"]
    assert two["scaling"] == "weak" and two["value"] > 0 and two["steps"] == 2
    assert one["parity"]["ok"] and one["roofline"]["bound"] == "mfma"
    assert one["roofline"]["kernel"].startswith("flash_hip::m16::attn_fwd_kernel_m16p<64")
    # every rank's own kernel time travels with the line (launch skew between GPUs shows on a real node)
    assert len(one["ranks"]["kernel_ms"]) == 1 and len(two["ranks"]["kernel_ms"]) == 2
    assert two["ranks"]["min"] <= two["ranks"]["max"] and two["ranks"]["min"] > 0


def test_headline_line_carries_roofline_baseline_and_secondary_figures():
    line = run_bench("--steps", "5", "--warmup", "2")
    assert line["metric"].startswith("attention fwd TFLOP/s") and line["unit"] == "TFLOP/s"
    assert line["parity"]["ok"]
    r = line["roofline"]
    assert r["peak"] == 2500.0 and 0.2 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the median of at least 20 individually timed launches beside the mean over the region
    assert r["kernel_ms_median_of"] >= 20 and 0.8 < r["kernel_ms_median"] / r["kernel_ms"] < 1.25
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0
    s = line["secondary"]
    assert s["sustained"]["launches"] * s["sustained"]["kernel_ms"] >= 1000.0      # at least a second
    assert s["gpt3_4k_b4"]["shape"] == [4, 96, 4096, 128] and s["gpt3_4k_b4"]["tflops"] > 0


def test_sequence_sweep_script(tmp_path):
    """benchmarks/sweep.py, the counterpart of the reference's scalability study (reference
    benchmarks/bench.py:89-111), over its three shortest sequence lengths: exits 0 (it gates every length
    against the CPU oracle itself), and writes the reference's `data` dict -- "Sequence Length" plus one
    list of milliseconds per implementation, under the reference's row names."""
    out_path = tmp_path / "sweep.json"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "benchmarks", "sweep.py"), "--max-n", "2048", "--reps", "3",
                          "--min-seconds", "0.02", "--out", str(out_path)], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    res = json.loads(out_path.read_text())
    ms = res["ms"]
    assert ms["Sequence Length"] == [512, 1024, 2048]
    for row in ("manual_attn", "mma_qreg_vld", "mma_qreg_async", "torch_flash_v2"):       # bench.py:98-109
        assert len(ms[row]) == 3 and all(t > 0 for t in ms[row]), (row, ms[row])
    for row in ("mma_qreg_async_fp32", "mma_qreg_async_bf16"):                            # this build's extra rows
        assert len(ms[row]) == 3 and all(t > 0 for t in ms[row]), (row, ms[row])
    assert [p["N"] for p in res["parity"]] == [512, 1024, 2048] and all(p["ok"] for p in res["parity"])
    assert set(res["profiler_ms"]) == set(ms) - {"Sequence Length"}
    # the native kernel beats the unfused formula at every length
    assert all(a < b for a, b in zip(ms["mma_qreg_async"], ms["manual_attn"]))
