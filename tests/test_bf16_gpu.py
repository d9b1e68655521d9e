"""bfloat16 entry point (flash_fwd_bf16): the fp16 kernels with bfloat16 matrix instructions.

Not part of the reference (its tensors are fp32).  Bar, stated here because bfloat16 keeps 8 mantissa
bits where fp16 keeps 11: against the fp32 CPU oracle evaluated on the bfloat16-rounded inputs,
max-abs <= 1.6e-2 and |res - ref| <= 1.6e-2 + 2e-2 |ref| (eight times the fp16 bar, for the three mantissa
bits less; P and O are rounded to bfloat16).  Measured on standard-normal data: 1e-3 ... 4.4e-3
on the shapes below, 8.4e-3 worst over a few hundred random shapes (at (3,7,64,96)).
"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ATOL_BF16 = 1.6e-2
RTOL_BF16 = 2e-2

SHAPES = [(2, 3, 512, 128), (1, 4, 1024, 64), (2, 2, 333, 128), (1, 3, 100, 72), (3, 5, 256, 64), (1, 2, 65, 8),
          (1, 2, 2048, 128)]


def run_bf16(fa, torch, q, k, v, kt=11):
    B, H, N, d = q.shape
    tq, tk, tv = (torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0").bfloat16() for x in (q, k, v))
    out = torch.full_like(tq, float("nan"))
    rc = fa.lib().flash_fwd_bf16(tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), out.data_ptr(), B, H, N, d,
                                 1.0 / np.sqrt(d), int(kt), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, fa.lib().flash_hip_last_error()
    torch.cuda.synchronize()
    return out.float().cpu().numpy()


@pytest.mark.parametrize("shape", SHAPES, ids=str)
def test_bf16_vs_oracle(fa, oracle, shape):
    import torch

    rng = np.random.default_rng(sum(shape))
    q, k, v = (oracle.round_bf16(rng.standard_normal(shape).astype(np.float32)) for _ in range(3))
    ref = oracle.attn_naive(q, k, v)
    o = run_bf16(fa, torch, q, k, v)
    assert not np.isnan(o).any()
    err = np.abs(o - ref)
    print(shape, "max_abs", err.max(), "mean_abs", err.mean())
    assert err.max() <= ATOL_BF16 and (err <= ATOL_BF16 + RTOL_BF16 * np.abs(ref)).all(), err.max()
    assert err.mean() <= 1.5e-3            # rounding errors, not a bias


def test_bf16_forced_rescale_and_wrapper(fa, oracle):
    import torch

    B, H, N, d = 1, 2, 640, 128
    rng = np.random.default_rng(7)
    q, k, v = (rng.standard_normal((B, H, N, d)).astype(np.float32) for _ in range(3))
    for (row, key, gain) in [(5, 200, 6.0), (37, 449, 9.0), (300, 639, 12.0), (301, 64, 8.0)]:
        k[0, :, key] = gain * q[0, :, row] / np.linalg.norm(q[0, :, row], axis=-1, keepdims=True)
    q, k, v = (oracle.round_bf16(x) for x in (q, k, v))
    ref = oracle.attn_naive(q, k, v)
    o = run_bf16(fa, torch, q, k, v)
    err = np.abs(o - ref)
    assert err.max() <= ATOL_BF16 and (err <= ATOL_BF16 + RTOL_BF16 * np.abs(ref)).all(), err.max()
    # the Python wrapper: bfloat16 in, bfloat16 out
    tq, tk, tv = (torch.from_numpy(x).cuda().bfloat16() for x in (q, k, v))
    o2 = fa.forward(tq, tk, tv)
    assert o2.dtype == torch.bfloat16 and np.array_equal(o2.float().cpu().numpy(), o)
    # head dims without 16-byte row chunks are refused for bfloat16
    t = torch.zeros((1, 1, 16, 12), device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ValueError, match="multiple of 8"):
        fa.forward(t, t, t, fa.KernelType.mma)


def test_bf16_many_items_per_workgroup_with_spikes(fa, oracle):
    """More work items than workgroups in each persistent layout the bfloat16 kernels have (eight waves at
    d = 128, 512-row tiles and 256-row tiles at d = 64), with keys that force the rescale path."""
    import torch

    for shape in ((2, 150, 768, 128), (2, 300, 1024, 64), (2, 300, 768, 64)):
        B, H, N, d = shape
        rng = np.random.default_rng(N + d)
        q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
        for (row, key, gain) in [(5, 200, 6.0), (300, N - 1, 10.0), (N - 1, 64, 8.0)]:
            k[:, :, key] = gain * q[:, :, row] / np.linalg.norm(q[:, :, row], axis=-1, keepdims=True)
        q, k, v = (oracle.round_bf16(x) for x in (q, k, v))
        o = run_bf16(fa, torch, q, k, v)
        assert not np.isnan(o).any()
        for (b, h) in sorted({(0, 0), (B - 1, H - 1), (B // 2, H // 2), (0, 1)}):
            ref = oracle.attn_naive(*(x[b:b + 1, h:h + 1] for x in (q, k, v)))[0, 0]
            err = np.abs(o[b, h] - ref)
            assert err.max() <= ATOL_BF16 and (err <= ATOL_BF16 + RTOL_BF16 * np.abs(ref)).all(), (shape, b, h, err.max())


def test_bf16_through_the_torch_op(fa, oracle):
    """torch.ops.pyflash.* -> flash::forward accepts bfloat16 tensors as well."""
    import torch

    fa.load_pyflash()
    rng = np.random.default_rng(11)
    q, k, v = (oracle.round_bf16(rng.standard_normal((1, 2, 300, 64)).astype(np.float32)) for _ in range(3))
    tq, tk, tv = (torch.from_numpy(x).cuda().bfloat16() for x in (q, k, v))
    o = torch.ops.pyflash.mma_qreg_async(tq, tk, tv)
    assert o.dtype == torch.bfloat16 and o.shape == tq.shape
    err = np.abs(o.float().cpu().numpy() - oracle.attn_naive(q, k, v))
    assert err.max() <= ATOL_BF16
