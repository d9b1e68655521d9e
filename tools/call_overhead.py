"""Where the time of one blocking call goes at a short sequence: the same launch through (a) the bare C ABI
followed by a stream synchronise, (b) torch.ops.pyflash (the reference's benchmark path) and (c) the
Python wrapper, against (d) back-to-back asynchronous launches (kernel time alone).   (GPU)"""
import ctypes
import importlib
import sys
import time

import torch

sys.path.insert(0, ".")
fa = importlib.import_module("flash-attention_amd")


def timed(fn, reps):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    fa.load_pyflash()
    L = fa.lib()
    for shape, dtype in (((4, 96, 512, 128), torch.float16), ((4, 96, 1024, 128), torch.float16),
                         ((4, 96, 512, 128), torch.float32), ((16, 32, 512, 64), torch.float16)):
        B, H, N, d = shape
        q, k, v = (torch.randn(shape, device="cuda:0", dtype=dtype) for _ in range(3))
        o = torch.empty_like(q)
        stream = torch.cuda.current_stream()
        sp = ctypes.c_void_p(stream.cuda_stream)
        scale = 1.0 / d ** 0.5
        fn = L.flash_fwd_f16 if dtype == torch.float16 else L.flash_fwd_f32
        args = (q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), B, H, N, d, scale, 11, sp)
        reps = 2000
        # settle the clock
        for _ in range(3000):
            fn(*args)
        torch.cuda.synchronize()
        t_async = timed(lambda: fn(*args), reps)
        t_abi = timed(lambda: (fn(*args), stream.synchronize()), reps)
        t_alloc = timed(lambda: (torch.empty_like(q), fn(*args), stream.synchronize()), reps)
        t_op = timed(lambda: torch.ops.pyflash.mma_qreg_async(q, k, v), reps)
        t_py = timed(lambda: fa.forward(q, k, v), reps)
        fl = 4.0 * B * H * N * N * d
        print(f"{shape} {str(dtype)[6:]}: async {t_async:.1f} us ({fl / t_async / 1e6:.0f} TFLOP/s) | C ABI + sync {t_abi:.1f} | "
              f"+ empty_like {t_alloc:.1f} | torch.ops.pyflash {t_op:.1f} ({fl / t_op / 1e6:.0f}) | wrapper {t_py:.1f}", flush=True)


main()
