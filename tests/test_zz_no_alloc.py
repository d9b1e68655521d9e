"""GPU: the allocation-free fp32 entry point (flash_fwd_f32_static: scratch in a device array built
into the library) against flash_fwd_f32_ws with caller scratch, through the C ABI.  Both enqueue the
same kernels, so the outputs must be bit-identical; checked on two calls in a row, the second on a
smaller input, and against the CPU oracle at the usual bar."""
import ctypes

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _call(L, torch, name, tensors, shape, ws=None):
    B, H, N, d = shape
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [t.data_ptr() for t in tensors] + [B, H, N, d, 1.0 / np.sqrt(d), 11]
    if name == "flash_fwd_f32_ws":
        rc = L.flash_fwd_f32_ws(*args, ws.data_ptr(), ws.numel(), stream)
    else:
        rc = L.flash_fwd_f32_static(*args, stream)
    return rc


def test_static_scratch_entry_matches_workspace_entry(fa, oracle):
    import torch

    assert torch.cuda.is_available()
    L = fa.lib()
    rng = np.random.default_rng(2024)
    # second call smaller than the first: stale contents of the shared buffer must not leak into it.  Both
    # shapes take the conversion pass (the route that uses the scratch: N > 2048, or N > 768 for head dims up
    # to 64 unless N is a multiple of 512, which the kernel converts itself at every length)
    for shape in ((1, 4, 4096, 128), (1, 3, 1280, 64)):
        B, H, N, d = shape
        host = [rng.standard_normal(shape).astype(np.float32) * np.float32(1.1) for _ in range(3)]
        q, k, v = (torch.from_numpy(x).to("cuda:0") for x in host)
        need = L.flash_fwd_f32_workspace_bytes(B, H, N, d, 11)
        assert 0 < need <= 48 << 20
        ws = torch.empty(need, dtype=torch.uint8, device="cuda:0")
        o_ws = torch.full_like(q, float("nan"))
        o_st = torch.full_like(q, float("nan"))
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        assert _call(L, torch, "flash_fwd_f32_static", (q, k, v, o_st), shape) == 0, L.flash_hip_last_error()
        torch.cuda.synchronize()
        assert torch.cuda.mem_get_info()[0] == free_before or shape[2] == 4096   # (first call may load code)
        assert _call(L, torch, "flash_fwd_f32_ws", (q, k, v, o_ws), shape, ws) == 0, L.flash_hip_last_error()
        torch.cuda.synchronize()
        assert torch.equal(o_st, o_ws), f"{shape}: static-scratch output differs from the workspace entry"
        ref = oracle.attn_naive(*(oracle.round_f16(x) for x in host))
        assert_close(o_st.cpu().numpy(), ref, f"{shape} static scratch")


def test_static_scratch_entry_rejects_what_does_not_fit(fa):
    import torch

    L = fa.lib()
    t = torch.zeros(16, device="cuda:0")
    # K and V copies at d = 128: (1, 24, 4096, 128) is exactly the limit (12 Mi elements); one more head is over
    over = (1, 25, 4096, 128)
    rc = L.flash_fwd_f32_static(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), *over, 0.1, 11, None)
    assert rc == -1
    msg = L.flash_hip_last_error().decode()
    assert "flash_fwd_f32_static" in msg and str(48 << 20) in msg and "flash_fwd_f32_ws" in msg
    assert L.flash_fwd_f32_workspace_bytes(1, 24, 4096, 128, 11) == 48 << 20
    # head dims up to 64: whole 512-row tiles need no scratch at any length, other lengths beyond 768 two copies
    assert L.flash_fwd_f32_workspace_bytes(1, 32, 4096, 64, 11) == 0
    assert L.flash_fwd_f32_workspace_bytes(1, 60, 3840, 64, 11) == 2 * 60 * 3840 * 64 * 2
    assert L.flash_fwd_f32_static(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 60, 3840, 64, 0.1, 11, None) == -1
    # sequences short enough to be converted inside the attention kernel use no scratch and have no limit
    assert L.flash_fwd_f32_workspace_bytes(64, 96, 2048, 128, 11) == 0
    # errors of the plain entry points are kept
    assert L.flash_fwd_f32_static(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 1, 64, 200, 0.1, 11, None) == -2


@pytest.mark.parametrize("shape", [(1, 3, 2304, 128), (2, 6, 1024, 68)], ids=str)
def test_static_scratch_entry_is_capturable_in_a_hip_graph(fa, oracle, shape):
    """What the entry point is for: recorded into a HIP graph (where hipMalloc is not allowed) after one
    plain call on the device, which resolves the scratch address and sets the kernels' LDS attribute, and
    replayed on new data in the same buffers.  Both of its kernels (conversion pass, attention) are in the
    graph; for the head dim that is not a multiple of 8 all three (padding pass, attention on the padded
    copies in the scratch, output copy)."""
    import torch

    L = fa.lib()
    assert 0 < L.flash_fwd_f32_workspace_bytes(*shape, 11) <= 48 << 20
    rng = np.random.default_rng(77)
    host = [rng.standard_normal(shape).astype(np.float32) for _ in range(6)]
    q, k, v = (torch.from_numpy(x).to("cuda:0") for x in host[:3])
    o = torch.empty_like(q)
    assert _call(L, torch, "flash_fwd_f32_static", (q, k, v, o), shape) == 0, L.flash_hip_last_error()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            assert _call(L, torch, "flash_fwd_f32_static", (q, k, v, o), shape) == 0, L.flash_hip_last_error()
    for base in (0, 3):
        for t, x in zip((q, k, v), host[base:base + 3]):
            t.copy_(torch.from_numpy(x).to("cuda:0"))
        o.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        ref = oracle.attn_naive(*(oracle.round_f16(x) for x in host[base:base + 3]))
        assert_close(o.cpu().numpy(), ref, f"graph replay of the static-scratch entry, inputs {base}")


def test_static_scratch_entry_takes_unaligned_tensors_of_any_size(fa, oracle):
    """Tensors that start off a 16-byte boundary never use the scratch (they are staged element by element
    inside the attention kernel), so the size limit does not apply to them: a shape over the limit is
    accepted and computed, like flash_fwd_f32 would."""
    import torch

    L = fa.lib()
    shape = (1, 25, 4096, 128)
    B, H, N, d = shape
    n = B * H * N * d
    gen = torch.Generator(device="cuda:0").manual_seed(3)
    bufs = [torch.randn(n + 8, device="cuda:0", generator=gen) for _ in range(3)] + [torch.zeros(n + 8, device="cuda:0")]
    views = [b[1:1 + n].view(shape) for b in bufs]
    assert all(t.data_ptr() % 16 == 4 for t in views)
    assert _call(L, torch, "flash_fwd_f32_static", views, shape) == 0, L.flash_hip_last_error()
    torch.cuda.synchronize()
    h = H - 1
    ref = oracle.attn_naive(*(oracle.round_f16(t[:, h:h + 1].cpu().numpy()) for t in views[:3]))[0, 0]
    assert_close(views[3][0, h].cpu().numpy(), ref, "unaligned tensors over the scratch limit, last head")
