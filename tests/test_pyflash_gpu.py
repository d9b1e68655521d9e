"""GPU: the reference's operator surface — torch.ops.pyflash.* (reference pyflash/pyflash.cpp:10-35,
loaded as in benchmarks/bench.py:12-14) and the C++ runner's correctness tests (main.cpp:153-194)."""
import subprocess

import numpy as np
import pytest

from conftest import assert_close, load_golden

pytestmark = pytest.mark.gpu

OPS = ["naive", "scalar2d", "scalar2d_row_tile", "warp_wmma", "block_wmma", "block_wmma_row_block",
       "block_wmma_async", "mma", "mma_swizzle", "mma_qreg", "mma_qreg_vld", "mma_qreg_async"]


@pytest.fixture(scope="module")
def pyflash(fa):
    import torch

    fa.load_pyflash()
    return torch.ops.pyflash


def test_all_twelve_ops_exist_and_match_golden(fa, pyflash):
    import torch

    g = load_golden((1, 1, 100, 32))
    q32, k32, v32 = (torch.from_numpy(g[n]).cuda() for n in ("q32", "k32", "v32"))
    qh, kh, vh = (torch.from_numpy(g[n]).cuda() for n in "qkv")
    for name in OPS:
        op = getattr(pyflash, name)
        o = op(qh, kh, vh)                     # native fp16
        assert o.dtype == torch.float16
        assert_close(o.float().cpu().numpy(), g["o"], f"pyflash.{name} fp16")
        o = op(q32, k32, v32)                  # what the reference's bench.py passes (fp32 tensors)
        assert o.dtype == torch.float32
        assert_close(o.cpu().numpy(), g["o"], f"pyflash.{name} fp32")


def test_op_errors_surface_as_exceptions(pyflash):
    import torch

    q = torch.randn(1, 1, 64, 50).cuda()
    with pytest.raises(Exception, match="multiple of 4"):
        pyflash.mma_qreg_async(q, q, q)
    q = torch.randn(1, 1, 64, 136).cuda()
    with pytest.raises(Exception, match="Unsupported tile size"):
        pyflash.mma_qreg(q, q, q)


def test_runner_correctness_suite(fa):
    """`runner` with no arguments: the reference's five shapes through flash::forward, exit code 0."""
    out = subprocess.run([fa.RUNNER_PATH], capture_output=True, text=True, timeout=110)
    assert out.returncode == 0, out.stdout[-4000:] + out.stderr[-2000:]
    assert "Test failed" not in out.stdout and out.stdout.count("Test passed") >= 50


def test_odd_head_dim_through_the_operator_takes_the_padded_route(fa, oracle, pyflash):
    """fp32 tensors with d = 69 (a head dim of the reference's runner, main.cpp:177-183) and enough work:
    flash::forward sizes its scratch with flash_fwd_f32_workspace_bytes, so the operator and the Python
    wrapper run the padded route (rows copied to 16-byte chunks, the fast kernels, output copied out)."""
    import torch

    shape = (2, 8, 1024, 69)
    assert fa.lib().flash_fwd_f32_workspace_bytes(*shape, 0) > 0
    assert fa.launch_info(*shape, kernel_type=0, entry=fa.ENTRY_F32_WS).staging == 2
    torch.manual_seed(5)
    q, k, v = (torch.randn(shape, device="cuda") for _ in range(3))
    ref = oracle.attn_naive(*(oracle.round_f16(x.cpu().numpy()) for x in (q, k, v)))
    o = pyflash.naive(q, k, v)
    assert o.dtype == torch.float32 and o.shape == q.shape and o.is_contiguous()
    assert_close(o.cpu().numpy(), ref, "pyflash.naive fp32 d = 69")
    assert torch.equal(fa.forward(q, k, v, fa.KernelType.naive1D), o)
    assert torch.equal(pyflash.mma(q, k, v), o)
