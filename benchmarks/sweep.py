#!/usr/bin/env python3
"""Sequence-length sweep of the attention forward on one MI355X.

Counterpart of the reference's scalability study (reference benchmarks/bench.py:89-111, the data
behind its img/scalability.png): B=4, H=96, d=128 (bench.py:28-31), N = 512 ... 16384, one row per
implementation, times in milliseconds in a dict of the same shape
    {"Sequence Length": [...], "<kernel name>": [ms, ...], ...}
so the reference's plot can be regenerated from it.  Differences, on purpose: inputs are fp16 (the
native kernel's storage type; one extra row feeds it fp32 tensors as the reference does), every kernel
is warmed up and the average of several back-to-back calls is reported (the reference profiles one
cold call, bench.py:58-66), and TFLOP/s = 4*B*H*N^2*d / t is
printed next to the times.

Rows
  manual_attn        softmax(q @ k^T / sqrt(d)) @ v with torch ops in fp16 (0 where it runs out of memory)
  mma_qreg_vld       torch.ops.pyflash.mma_qreg_vld   -> native kernel
  mma_qreg_async     torch.ops.pyflash.mma_qreg_async -> native kernel
  torch_flash_v2     torch.nn.functional.scaled_dot_product_attention in fp16 (whatever backend
                     PyTorch-ROCm selects)
  mma_qreg_async_fp32  the same op on fp32 tensors (converted inside the attention kernel up to N = 2048, by a
                       conversion pass into scratch beyond; fp32 out)
  mma_qreg_async_bf16  the same op on bfloat16 tensors

Two timing columns per row: `ms` = average of back-to-back calls between two HIP events, and
`profiler_ms` = the reference's own method (benchmarks/bench.py:58-66): one call under
torch.autograd.profiler, sum of self_device_time_total over key_averages() — here after warm-up.

Every sequence length is also a parity check: the last head of the native output is compared with the
CPU oracle (fp32 softmax(QK^T/sqrt(d))V on the same fp16 inputs) and the script exits non-zero if
max-abs exceeds 2e-3 or any element leaves 2e-3 + 1e-2*|ref|.

  python benchmarks/sweep.py [--out profiles/sweep.json] [--max-n 16384] [--reps 10] [--min-seconds 0.25]
"""
from __future__ import annotations

import argparse
import importlib
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-n", type=int, default=16384)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--min-seconds", type=float, default=0.25, help="least length of a timed burst")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--heads", type=int, default=96)
    ap.add_argument("--head-dim", type=int, default=128)
    args = ap.parse_args()

    import torch
    import torch.nn.functional as F

    fa = importlib.import_module("flash-attention_amd")
    fa.load_pyflash()
    pyflash = torch.ops.pyflash
    if not torch.cuda.is_available():
        raise SystemExit("sweep.py needs a HIP device")
    dev = torch.device("cuda", 0)
    B, H, d = args.batch, args.heads, args.head_dim

    def manual_attn(q, k, v):
        att = torch.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(d)), dim=-1)
        return att @ v

    rows = [
        ("manual_attn", manual_attn),
        ("mma_qreg_vld", pyflash.mma_qreg_vld),
        ("mma_qreg_async", pyflash.mma_qreg_async),
        ("torch_flash_v2", F.scaled_dot_product_attention),
        # the reference's own tensor dtype: fp32 in, fp32 out
        ("mma_qreg_async_fp32", lambda q, k, v: pyflash.mma_qreg_async(q32, k32, v32)),
        # bfloat16 tensors: the same kernels with bfloat16 matrix instructions
        ("mma_qreg_async_bf16", lambda q, k, v: pyflash.mma_qreg_async(qb, kb, vb)),
    ]

    def time_ms(fn, q, k, v):
        """Average over back-to-back calls that fill at least `--min-seconds` (and at least `--reps`
        calls): the chip needs milliseconds to settle its clock under a new load, so a burst of ten short
        calls times the ramp (N = 512: 0.09 ms per call in such a burst against 0.055 ms sustained), and
        single calls separated by host synchronisation run at boost clocks."""
        def burst(n):
            e0 = torch.cuda.Event(enable_timing=True)
            e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn(q, k, v)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n

        for _ in range(3):
            fn(q, k, v)
        torch.cuda.synchronize(dev)
        est = burst(args.reps)
        n = max(args.reps, min(5000, int(args.min_seconds * 1e3 / max(est, 1e-3)) + 1))
        return burst(n) if n > args.reps else est

    def profiler_ms(fn, q, k, v):
        """The reference's profile_kernel (benchmarks/bench.py:58-66): sum of self device time of one call."""
        import torch.autograd.profiler as profiler

        with profiler.profile(use_device="cuda") as prof:
            fn(q, k, v)
            torch.cuda.synchronize(dev)
        return sum(item.self_device_time_total for item in prof.key_averages()) / 1e3

    from oracle import oracle as orc
    import numpy as np

    orc.build()
    seqs = [n for n in (512, 1024, 2048, 4096, 8192, 16384) if n <= args.max_n]
    data = {"Sequence Length": seqs}
    tflops = {}
    prof_data = {}
    parity = []
    failed = False
    gen = torch.Generator(device=dev).manual_seed(0)
    for N in seqs:
        q, k, v = (torch.randn(B, H, N, d, device=dev, generator=gen).half() for _ in range(3))
        q32, k32, v32 = q.float(), k.float(), v.float()
        qb, kb, vb = q.bfloat16(), k.bfloat16(), v.bfloat16()
        flops = 4.0 * B * H * N * N * d
        print(f"\nseq_len {N}", flush=True)
        ref = None
        for name, fn in rows:
            try:
                ms = time_ms(fn, q, k, v)
                pms = profiler_ms(fn, q, k, v)
                if name == "torch_flash_v2":
                    ref = fn(q, k, v)
            except torch.OutOfMemoryError as e:   # the N x N score matrix of manual_attn at long N
                ms = pms = 0.0
                print("  out of memory:", str(e).split("\n")[0])
                torch.cuda.empty_cache()
            data.setdefault(name, []).append(round(ms, 4))
            prof_data.setdefault(name, []).append(round(pms, 4))
            tflops.setdefault(name, []).append(round(flops / (ms * 1e-3) / 1e12, 1) if ms else 0.0)
            print(f"  {name}: {ms:.3f} ms  {tflops[name][-1]} TFLOP/s   (profiler sum {pms:.3f} ms)", flush=True)
        o_nat = pyflash.mma_qreg_async(q, k, v)
        if ref is not None:
            err = float((o_nat.float() - ref.float()).abs().max())
            print(f"  max |native - torch SDPA| = {err:.2e}")
        # parity gate: last head (the one at the highest byte offset) against the CPU oracle
        hq, hk, hv = (x[B - 1:B, H - 1:H].float().cpu().numpy() for x in (q, k, v))
        want = orc.attn_naive(hq, hk, hv)[0, 0]
        got = o_nat[B - 1, H - 1].float().cpu().numpy()
        diff = np.abs(got - want)
        ok = bool(diff.max() <= 2e-3 and (diff <= 2e-3 + 1e-2 * np.abs(want)).all())
        parity.append({"N": N, "head": [B - 1, H - 1], "max_abs_vs_cpu_oracle": float(diff.max()), "ok": ok})
        print(f"  max |native - CPU oracle| (head {B - 1},{H - 1}) = {diff.max():.2e}  {'ok' if ok else 'FAIL'}", flush=True)
        failed = failed or not ok
        del o_nat
        del q, k, v, q32, k32, v32, qb, kb, vb
        torch.cuda.empty_cache()

    print()
    print(data)
    result = {"shape": {"B": B, "H": H, "d": d, "dtype": "fp16"}, "ms": data, "tflops": tflops,
              "profiler_ms": prof_data, "parity": parity,
              "method": f"ms: average of {args.reps} back-to-back calls between two HIP events after 3 warm-up calls; "
                        "profiler_ms: sum of self_device_time_total of one warmed-up call under torch.autograd.profiler "
                        "(the reference's method); one MI355X",
              "torch": torch.__version__}
    if args.out:
        with open(args.out, "w") as f:
            json.dump(result, f, indent=1)
        print("wrote", args.out)
    if failed:
        raise SystemExit("parity against the CPU oracle failed at one or more sequence lengths")


if __name__ == "__main__":
    main()
