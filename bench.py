#!/usr/bin/env python3
"""Benchmark of the attention-forward hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--gather-chunks C]

One "step" = one forward pass of the native kernel over one synthetic batch that is already
resident in HBM.  The default workload is the headline configuration of BASELINE.json's metric
(fp16, B=8, H=96, N=4096, d=128 per GPU: B=64 over 8 GPUs is configs[4]); `--workload gpt3_1k` is configs[1]
(N=1024) and `--workload d64` is the small-head configuration (16, 32, 2048, 64).  With N > 1 GPUs there is
one rank process per GPU: `python bench.py --gpus N` starts them itself, and under a launcher such as
torch.distributed.run the script is a rank.  Every rank owns its own batch of the same shape -- the
batch x heads axis is what shards, there is no collective on the data path -- so scaling is weak and
`value` is the sum over ranks of FLOPs / max-over-ranks time; the line also carries every rank's own kernel
time (`ranks`), so that launch skew between GPUs shows.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline      dominant kernel vs. the dense fp16 MFMA peak (HIP events on the launch stream): the mean
                over the timed region and the median of individually timed launches
  cpu_baseline  the CPU oracle (OpenMP, fp32) timed on a bounded head sample of the same workload
  parity        max-abs error of one head of the timed output against the CPU oracle
"""
from __future__ import annotations

import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (B, H, N, d)
    # target shape of the metric (H=96, d=128, N=4096) with the per-GPU batch of configs[4]: B=64 over 8
    # GPUs is 8 per GPU, so the 8-GPU run of this weak-scaling benchmark is exactly that configuration
    "headline": (8, 96, 4096, 128),
    "gpt3_4k_b4": (4, 96, 4096, 128),  # the same head shape at the batch of configs[1] and [2]
    "gpt3_1k": (4, 96, 1024, 128),    # BASELINE.json configs[1]
    "gpt3_2k": (4, 96, 2048, 128),    # the reference runner's benchmark shape (main.cpp:197-202)
    "d64": (16, 32, 2048, 64),        # configs[3]
    "tiny": (2, 4, 256, 64),          # configs[0]
}
# Default (steps, warmup) per workload: a timed region of a few tenths of a second.  The chip takes
# milliseconds to settle its clock under a new load, so 50 launches of a 0.2 ms kernel would time the
# ramp and not the kernel (measured: 0.235 ms per launch in such a burst against 0.198 ms sustained).
DEFAULT_STEPS = {"headline": (50, 5), "gpt3_4k_b4": (100, 10), "gpt3_2k": (400, 40), "gpt3_1k": (1500, 150),
                 "d64": (600, 60), "tiny": (2000, 200)}
PEAK_TFLOPS_FP16_DENSE = 2500.0       # MI355X dense fp16/bf16 MFMA peak (256 CU x 4096 flop/clk x 2.4 GHz)


def attention_flops(B, H, N, d):
    return 4.0 * B * H * N * N * d    # QK^T + PV, non-causal


def cpu_baseline(oracle, shape, target_seconds=12.0):
    """Time the CPU oracle on a bounded head sample of the workload (about `target_seconds` of CPU
    work, sized from a 4-head calibration run); report TFLOP/s.  Heads are independent, so the
    whole-workload time is the sample time scaled by B*H / heads."""
    import numpy as np

    B, H, N, d = shape
    rng = np.random.default_rng(0)

    def timed(heads):
        q, k, v = (oracle.round_f16(rng.standard_normal((1, heads, N, d)).astype(np.float32)) for _ in range(3))
        t0 = time.perf_counter()
        oracle.attn_naive(q, k, v)
        return time.perf_counter() - t0

    timed(1)                                   # spin up the OpenMP pool
    probe = min(4, B * H)
    per_head = timed(probe) / probe
    heads = int(max(probe, min(B * H, round(target_seconds / per_head))))
    dt = timed(heads)
    return {
        "value": round(attention_flops(1, heads, N, d) / dt / 1e12, 4),
        "unit": "TFLOP/s",
        "cores": oracle.num_threads(),
        "kind": "port",
        "sample": f"{heads} of {B * H} heads of ({B},{H},{N},{d}), fp32 softmax(QK^T/sqrt(d))V on fp16-rounded inputs, "
                  f"{dt:.2f} s; whole workload would take {dt * B * H / heads:.1f} s",
    }


def measured_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction for wide streaming
    reads applied), or None when no pass was recorded for this workload."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            return json.load(f).get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def launch_ranks(n_ranks):
    """`python bench.py --gpus N` started as one plain process: start N fresh rank processes of this
    script (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, a free rendezvous port),
    relay rank 0's JSON line and return non-zero if any rank fails.  This parent never imports torch
    and never touches a GPU; the ranks are children, not replacements of this process."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # RCCL shares device buffers between the rank processes through IPC handles; the driver of these
        # machines only supports the dmabuf kind, which this selects (the legacy kind fails with
        # "hipIpcGetMemHandle: invalid argument").  The attention path itself shares nothing between ranks.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rc = procs[0].returncode
    deadline = time.time() + 120
    for pr in procs[1:]:
        try:
            pr.wait(timeout=max(1.0, deadline - time.time()))
        except subprocess.TimeoutExpired:
            pr.kill()          # this exact child; rank 0 is gone, so it can only be stuck in a collective
            pr.wait()
        rc = rc or pr.returncode
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: per workload, see DEFAULT_STEPS)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default: per workload)")
    ap.add_argument("--workload", default="headline", choices=sorted(WORKLOADS))
    ap.add_argument("--kernel-type", default="mma_qreg_async")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16"],
                    help="tensor type; the metric is defined on fp16, bf16 runs the same kernels with bfloat16 "
                         "matrix instructions (parity is then checked against the bf16 bar of 1.6e-2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary figures of the headline run (sustained >= 1 s rate, batch-4 rate)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo lets several ranks rehearse the "
                         "multi-rank path on a box with fewer GPUs than ranks)")
    ap.add_argument("--gather-chunks", type=int, default=0,
                    help="also time the forward with the optional all-gather of O overlapped in this many "
                         "head chunks (reported beside the metric, never part of `value`)")
    ap.add_argument("--gather-exchange", default="p2p", choices=["p2p", "all_gather"])
    args = ap.parse_args()

    if args.steps is None:
        args.steps = DEFAULT_STEPS[args.workload][0]
    if args.warmup is None:
        args.warmup = DEFAULT_STEPS[args.workload][1]

    # N > 1 without a launcher around us: become the launcher, before anything GPU-related is imported
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    fa = importlib.import_module("flash-attention_amd")
    fa.lib()  # fail loudly if the HIP library is missing

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world         # under a launcher the launcher's world size is the number of ranks
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device; there is no CPU path for the attention kernel")
    n_dev = torch.cuda.device_count()
    if args.backend == "nccl" and local_rank >= n_dev:
        raise SystemExit(f"rank {rank}: local rank {local_rank} has no GPU ({n_dev} visible)")
    dev = torch.device("cuda", local_rank % n_dev)
    torch.cuda.set_device(dev)
    if world > 1 or args.gather_chunks > 0:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # single process with --gather-chunks: any free port
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    shape = WORKLOADS[args.workload]
    B, H, N, d = shape
    kt = fa.KernelType[args.kernel_type]
    gen = torch.Generator(device=dev).manual_seed(rank)
    tdtype = torch.float16 if args.dtype == "fp16" else torch.bfloat16
    q, k, v = (torch.randn(shape, device=dev, generator=gen).to(tdtype) for _ in range(3))
    o = torch.empty_like(q)

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        fa.forward(q, k, v, kt, out=o, sync=False)
    barrier()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                       # same stream the kernel is launched on (torch's current stream)
    for _ in range(args.steps):
        fa.forward(q, k, v, kt, out=o, sync=False)
    ev1.record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps

    # the same launch timed one at a time (an event pair around each, all enqueued back to back, one
    # synchronisation at the end): the median is insensitive to a slow first launch or a clock dip
    n_single = max(20, min(args.steps, 200))
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_single)]
    for e0, e1 in pairs:
        e0.record()
        fa.forward(q, k, v, kt, out=o, sync=False)
        e1.record()
    torch.cuda.synchronize(dev)
    singles = sorted(e0.elapsed_time(e1) for e0, e1 in pairs)
    kernel_ms_median = singles[len(singles) // 2]

    rank_ms = [kernel_ms]
    if world > 1:
        cpu_or_dev = dev if args.backend == "nccl" else "cpu"
        t = torch.tensor([elapsed], device=cpu_or_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        mine = torch.tensor([kernel_ms], device=cpu_or_dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(x.item()) for x in every]

    gather = None
    if args.gather_chunks > 0:
        if args.backend != "nccl":
            raise SystemExit("--gather-chunks needs the nccl backend (device tensors)")
        # every rank's batch is one slab of the flattened batch*heads axis of a world-times larger job
        qs, ks, vs = (x.view(B * H, N, d) for x in (q, k, v))
        full = torch.empty((world * B * H, N, d), dtype=q.dtype, device=dev)
        for _ in range(max(2, args.warmup)):
            fa.forward_shard_gather(qs, ks, vs, world * B * H, kt, chunks=args.gather_chunks, out=full, sync=False,
                                    exchange=args.gather_exchange)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            fa.forward_shard_gather(qs, ks, vs, world * B * H, kt, chunks=args.gather_chunks, out=full, sync=False,
                                    exchange=args.gather_exchange)
        barrier()
        tg = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        same = bool(torch.equal(full[rank * B * H:(rank + 1) * B * H], o.view(B * H, N, d)))
        gather = {
            "chunks": args.gather_chunks,
            "exchange": args.gather_exchange,
            "ms_per_step": round(float(tg.item()) / args.steps * 1e3, 4),
            "gathered_bytes_per_gpu": (world - 1) * B * H * N * d * 2,
            "own_slab_identical_to_plain_forward": same,
        }

    # secondary figures, single-GPU headline runs only, outside the timed region of `value`:
    #  - the same launch repeated for at least a second (clocks under sustained matrix load settle
    #    below what a 0.1 s burst sees),
    #  - the same head shape at batch 4 (the configuration of the reference's own study).
    secondary = None
    if world == 1 and args.workload == "headline" and not args.no_secondary:
        def timed_launches(tq, tk, tv, to, n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fa.forward(tq, tk, tv, kt, out=to, sync=False)
            e1.record()
            torch.cuda.synchronize(dev)
            return e0.elapsed_time(e1) / n

        n_sus = max(args.steps, int(1.2e3 / max(kernel_ms, 1e-3)) + 1)
        ms_sus = timed_launches(q, k, v, o, n_sus)
        b4 = WORKLOADS["gpt3_4k_b4"]
        ms_b4 = timed_launches(q[:4], k[:4], v[:4], o[:4], max(args.steps, 20))
        secondary = {
            "sustained": {"launches": n_sus, "kernel_ms": round(ms_sus, 4),
                          "tflops": round(attention_flops(*shape) / ms_sus / 1e9, 2),
                          "frac": round(attention_flops(*shape) / ms_sus / 1e9 / PEAK_TFLOPS_FP16_DENSE, 4)},
            "gpt3_4k_b4": {"shape": list(b4), "kernel_ms": round(ms_b4, 4),
                           "tflops": round(attention_flops(*b4) / ms_b4 / 1e9, 2),
                           "frac": round(attention_flops(*b4) / ms_b4 / 1e9 / PEAK_TFLOPS_FP16_DENSE, 4)},
        }

    if rank == 0:
        info = fa.launch_info(B, H, N, d, kt, fa.ENTRY_BF16 if args.dtype == "bf16" else fa.ENTRY_F16)
        pers = info.persistent    # what the library routes this shape to right now (environment included)
        ns = "m16w" if info.block_dim[1] == 8 else ("m16x" if info.tile_size[1] == 512 else "m16")
        flops = attention_flops(*shape)
        value = world * flops * args.steps / elapsed / 1e12
        achieved = flops / (kernel_ms * 1e-3) / 1e12
        line = {
            "metric": "attention fwd TFLOP/s (fp16, H=96 d=128) + % of MFMA peak at 1/2/4/8 GPUs",
            "value": round(value, 2),
            "unit": "TFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,       # the reference publishes no fp16 / MI355X number
            "dtype": args.dtype,
            "data": "synthetic (standard normal, seeded per rank)",
            "config": {
                "workload": f"{args.workload}: attention forward B={B} H={H} N={N} d={d} per GPU, non-causal, "
                            f"kernel_type={kt.name}",
                "global_batch_heads": world * B * H,
                "seq_len": N,
                "head_dim": d,
                "parallelism": f"batch*heads sharded over {world} GPU(s), no collective",
            },
            "pct_of_mfma_peak": round(100.0 * value / (world * PEAK_TFLOPS_FP16_DENSE), 2),
            "roofline": {
                "bound": "mfma",
                "achieved": round(achieved, 2),
                "peak": PEAK_TFLOPS_FP16_DENSE,
                "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_TFLOPS_FP16_DENSE, 4),
                # HBM bytes per launch of this kernel from the committed rocprofv3 PMC passes of this workload
                # (profiles/traffic.json names the pass); not re-measured by this run
                "traffic": measured_traffic(args.workload),
                # ns: the wave layout the library picked (m16 = 4 waves x 64 query rows, m16w = 8 x 32, m16x = 4 x 128)
                # (third template argument: K and V are fp32 tensors converted inside the kernel; never here)
                "kernel": "flash_hip::{ns}::attn_fwd_kernel_m16{}<{}, {}, false>".format(
                    "p" if pers else "", 128 if d > 64 else 64,
                    "_Float16" if args.dtype == "fp16" else "flash_hip::{ns}::bf16_out", ns=ns).format(ns=ns),
                # the same kernel as rocprofv3's CSV output names it (mangled)
                "kernel_symbol": "_ZN9flash_hip{}{}{}attn_fwd_kernel_m16{}ILi{}E{}Lb0EEEvNS_9FwdParamsE".format(
                    len(ns), ns, 20 if pers else 19, "p" if pers else "", 128 if d > 64 else 64,
                    "DF16_" if args.dtype == "fp16" else "NS0_8bf16_outE"),
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_median": round(kernel_ms_median, 4),
                "kernel_ms_median_of": n_single,
                "frac_at_median": round(flops / (kernel_ms_median * 1e-3) / 1e12 / PEAK_TFLOPS_FP16_DENSE, 4),
                "algorithmic_flops_per_launch": flops,
            },
            # every rank's mean kernel time over the timed region: the spread is launch skew / device variation
            "ranks": {"kernel_ms": [round(x, 4) for x in rank_ms], "min": round(min(rank_ms), 4),
                      "max": round(max(rank_ms), 4)},
        }
        # parity of the timed output (one head) and the CPU baseline: rank 0, single-GPU runs only
        if world == 1:
            from oracle import oracle as orc

            orc.build()
            import numpy as np

            hq, hk, hv = (x[B - 1:B, H - 1:H].float().cpu().numpy() for x in (q, k, v))
            ref = orc.attn_naive(hq, hk, hv)
            err = float(np.abs(o[B - 1, H - 1].float().cpu().numpy() - ref[0, 0]).max())
            bar = 2e-3 if args.dtype == "fp16" else 1.6e-2
            line["parity"] = {"max_abs_vs_cpu_oracle": err, "bar": bar, "ok": bool(err <= bar)}
            if not err <= bar:
                # a rate measured on a wrong result is no measurement: no JSON line, non-zero exit
                sys.stderr.write(f"bench.py: parity FAILED: max-abs {err:.3e} against the CPU oracle exceeds {bar}\n")
                if dist.is_initialized():
                    dist.destroy_process_group()
                sys.exit(1)
            if not args.no_cpu_baseline:
                line["cpu_baseline"] = cpu_baseline(orc, shape)
        if secondary is not None:
            line["secondary"] = secondary
        if gather is not None:
            gather["exposed_ms_per_step"] = round(gather["ms_per_step"] - line["ms_per_step"], 4)
            line["forward_plus_overlapped_gather"] = gather
        print(json.dumps(line), flush=True)

    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
