"""A longer randomised parity campaign than the forty-case fuzz of the test suite, for spare GPU time:
shapes up to a few GFLOP (several work items per persistent workgroup, every wave layout, the padded
route of head dims that are not a multiple of 8), every entry point (fp16, fp32, fp32 with workspace, the
built-in-scratch entry, bfloat16), input scales 0.25 ... 1.5, and in a quarter of the cases a planted key
whose score towers over its row (the overflow / second-sweep path of the persistent kernels).  Every case is
checked against the CPU oracle at the parity bar (bfloat16: its own looser bar).
usage: python tools/fuzz_long.py [cases [seed [tiles]]]     (GPU)
`tiles`: whole query tiles only (N a multiple of 256, up to 192 heads): persistent kernels with several
work items per workgroup in every case."""
import ctypes
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
fa = importlib.import_module("flash-attention_amd")
from oracle import oracle as orc  # noqa: E402  (the checker, not the product)

ATOL, RTOL, BF16_ATOL = 2e-3, 1e-2, 1.6e-2


def allowed(kt, d):
    if kt == 9:
        return d % 2 == 0
    if kt in (10, 11):
        return d % 4 == 0
    return True


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tiles = len(sys.argv) > 3 and sys.argv[3] == "tiles"
    rng = np.random.default_rng(seed)
    orc.build()
    L = fa.lib()
    edges = [1, 63, 64, 65, 255, 256, 257, 511, 512, 513, 768, 1000, 1024, 1280, 1536, 2048, 2304]
    dims = [8, 16, 24, 32, 40, 48, 56, 64, 72, 80, 96, 104, 112, 120, 128]
    modes = ["f16", "f32", "f32_ws", "f32_static", "bf16"]
    worst = {m: 0.0 for m in modes}
    routes = {}
    t0 = time.time()
    done = fails = 0
    while done < cases:
        B, H = int(rng.integers(1, 5)), int(rng.integers(1, 25))
        N = int(rng.choice(edges)) if rng.random() < 0.6 else int(rng.integers(1, 2305))
        d = int(rng.choice(dims)) if rng.random() < 0.5 else int(rng.integers(1, 129))
        if tiles:
            B, H, N = int(rng.integers(1, 5)), int(rng.integers(8, 49)), 256 * int(rng.integers(1, 10))
        if B * H * N * N * d > 6e10 // 4:      # keep the oracle's share of a case under a second
            continue
        kt = int(rng.integers(0, 12))
        if not allowed(kt, d):
            continue
        mode = modes[int(rng.integers(0, len(modes)))]
        if mode == "bf16" and d % 8:
            continue
        scale = float(rng.choice([0.25, 0.5, 1.0, 1.5]))
        shape = (B, H, N, d)
        q, k, v = ((rng.standard_normal(shape) * scale).astype(np.float32) for _ in range(3))
        spike = rng.random() < 0.25 and N > 8
        if spike:       # one key per head whose scaled score has a spread of 9 scale^2 nats over the queries (others: scale^2)
            pos = rng.integers(0, N, size=(B, H))
            for b in range(B):
                for h in range(H):
                    k[b, h, pos[b, h]] = np.sign(rng.standard_normal(d)).astype(np.float32) * np.float32(9.0 * scale)
        tdt = {"f16": torch.float16, "bf16": torch.bfloat16}.get(mode, torch.float32)
        tq, tk, tv = (torch.from_numpy(x).to("cuda:0", tdt) for x in (q, k, v))
        out = torch.full_like(tq, float("nan"))
        args = [tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), out.data_ptr(), B, H, N, d, 1.0 / np.sqrt(d), kt]
        if mode == "f32_ws":
            need = L.flash_fwd_f32_workspace_bytes(B, H, N, d, kt)
            ws = torch.empty(max(need, 1), dtype=torch.uint8, device="cuda:0")
            rc = L.flash_fwd_f32_ws(*args, ws.data_ptr() if need else None, need, None)
        elif mode == "f32_static":
            rc = L.flash_fwd_f32_static(*args, None)
            if rc == -1 and b"built-in buffer" in L.flash_hip_last_error():
                continue        # a shape beyond the 48 MiB of the built-in scratch: refused as documented
        else:
            rc = {"f16": L.flash_fwd_f16, "f32": L.flash_fwd_f32, "bf16": L.flash_fwd_bf16}[mode](*args, None)
        assert rc == 0, (shape, mode, kt, L.flash_hip_last_error())
        torch.cuda.synchronize()
        o = out.float().cpu().numpy().astype(np.float64)
        # the oracle sees what the matrix cores see: inputs rounded to the operand type
        if mode == "bf16":
            rq, rk, rv = (t.float().cpu().numpy() for t in (tq, tk, tv))
        else:
            rq, rk, rv = (orc.round_f16(x) for x in (q, k, v))
        ref = orc.attn_naive(rq, rk, rv).astype(np.float64)
        err = np.abs(o - ref)
        if mode == "bf16":
            ok = not np.isnan(o).any() and err.max() <= BF16_ATOL * max(1.0, scale)
        elif scale <= 1.0 and not spike:
            ok = not np.isnan(o).any() and err.max() <= ATOL and (err <= ATOL + RTOL * np.abs(ref)).all()
        else:
            ok = not np.isnan(o).any() and (err <= ATOL + RTOL * np.abs(ref)).all()
        entry = {"f16": fa.ENTRY_F16, "bf16": fa.ENTRY_BF16, "f32": fa.ENTRY_F32}.get(mode, fa.ENTRY_F32_WS)
        info = fa.launch_info(B, H, N, d, kernel_type=kt, entry=entry)
        key = (mode, info.staging, info.block_dim[1], info.tile_size[1], int(info.persistent))
        routes[key] = routes.get(key, 0) + 1
        worst[mode] = max(worst[mode], float(err.max()) / (scale if scale > 1 else 1.0))
        if not ok:
            fails += 1
            print(f"FAIL case {done}: {shape} kt {kt} {mode} scale {scale} spike {spike}: max-abs {err.max():.3e}", flush=True)
        done += 1
        if done % 50 == 0:
            print(f"[{time.time() - t0:6.1f} s] {done} cases, {fails} failures", flush=True)
    print(f"seed {seed}: {done} cases, {fails} failures; worst max-abs per entry point (divided by the input scale above 1): "
          + ", ".join(f"{m} {w:.2e}" for m, w in worst.items()))
    print("launches by (entry, staging, waves, query rows per item, persistent): "
          + ", ".join(f"{k}: {n}" for k, n in sorted(routes.items())))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
