"""fp32 callers with head dims that are not a multiple of 8 (the reference's runner tests d = 50 and 69,
reference main.cpp:169-183) through flash_fwd_f32_ws with a workspace: rows are padded to 16-byte chunks in
the workspace, the 16x16x32 kernels run on the copies, the first d columns of the padded output are copied
out.  Checked against the CPU oracle, in every wave layout the padded shape can route to, and bit for bit
against the same call on caller-padded tensors."""
import ctypes

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def call_ws(fa, torch, tq, tk, tv, out, shape, kt, need=None):
    L = fa.lib()
    B, H, N, d = shape
    if need is None:
        need = L.flash_fwd_f32_workspace_bytes(B, H, N, d, kt)
    ws = torch.full((max(need, 1),), 0xFF, dtype=torch.uint8, device="cuda:0")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = L.flash_fwd_f32_ws(tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), out.data_ptr(), B, H, N, d,
                            1.0 / np.sqrt(d), kt, ws.data_ptr() if need else None, need, stream)
    assert rc == 0, L.flash_hip_last_error()
    torch.cuda.synchronize()
    return need


# shape, kernel type, (waves, query rows per tile, persistent) of the padded launch
CASES = [
    ((2, 8, 1024, 69), 0, (8, 256, 1)),    # d_pad 72 in the 128 tile: eight waves, Q from an fp16 copy
    ((4, 8, 512, 100), 11, (8, 256, 1)),   # d_pad 104
    ((1, 16, 1000, 50), 9, (4, 256, 0)),   # d_pad 56 in the 64 tile; ragged N: one item per workgroup
    ((2, 8, 1024, 52), 11, (4, 512, 1)),   # d_pad 56; whole 512-row tiles: 128 rows per wave
    ((4, 8, 768, 36), 11, (4, 256, 1)),    # d_pad 40; 256-row tiles, four waves x 64 rows, Q read as fp32
    ((1, 3, 4001, 127), 0, (4, 256, 0)),   # d_pad 128, one padding column, ragged N
]


@pytest.mark.parametrize("shape,kt,layout", CASES, ids=str)
def test_padded_route_vs_oracle(fa, oracle, torch_gpu, shape, kt, layout):
    torch = torch_gpu
    L = fa.lib()
    B, H, N, d = shape
    d_pad = (d + 7) // 8 * 8
    info = fa.launch_info(*shape, kernel_type=kt, entry=fa.ENTRY_F32_WS)
    assert info.staging == 2 and not info.aligned_rows and info.tile_size[2] == (64 if d_pad <= 64 else 128)
    assert (info.block_dim[1], info.tile_size[1], int(info.persistent)) == layout
    rng = np.random.default_rng(N + d)
    # not fp16-representable: the padding pass rounds K, V (and Q, here or in the kernel) to nearest even
    q, k, v = (rng.standard_normal(shape).astype(np.float32) * np.float32(1.13) for _ in range(3))
    tq, tk, tv = (torch.from_numpy(x).to("cuda:0") for x in (q, k, v))
    out = torch.full_like(tq, float("nan"))
    need = call_ws(fa, torch, tq, tk, tv, out, shape, kt)
    assert need == 6 * ((B * H * N * d_pad * 2 + 255) // 256 * 256)
    o = out.cpu().numpy()
    assert np.isfinite(o).all()
    ref = oracle.attn_naive(*(oracle.round_f16(x) for x in (q, k, v)))
    assert_close(o, ref, f"padded rows {shape}")
    # the element-staging route (no workspace) computes the same function
    o1 = torch.full_like(tq, float("nan"))
    call_ws(fa, torch, tq, tk, tv, o1, shape, kt, need=0)
    assert_close(o1.cpu().numpy(), ref, f"element staging {shape}")
    # caller-padded tensors with the scale of the original head dim: bit-identical columns
    if d_pad <= 128:
        pq, pk, pv = (torch.nn.functional.pad(t, (0, d_pad - d)).contiguous() for t in (tq, tk, tv))
        po = torch.full_like(pq, float("nan"))
        shape_p = (B, H, N, d_pad)
        needp = L.flash_fwd_f32_workspace_bytes(*shape_p, 11)
        wsp = torch.empty(max(needp, 1), dtype=torch.uint8, device="cuda:0")
        rc = L.flash_fwd_f32_ws(pq.data_ptr(), pk.data_ptr(), pv.data_ptr(), po.data_ptr(), *shape_p,
                                1.0 / np.sqrt(d), 11, wsp.data_ptr() if needp else None, needp, None)
        assert rc == 0, L.flash_hip_last_error()
        torch.cuda.synchronize()
        assert float(po[..., d:].abs().max()) == 0.0
        assert_close(po[..., :d].cpu().numpy(), ref, f"caller-padded {shape}")


def test_padded_route_workspace_checks_unaligned_views_and_static_entry(fa, oracle, torch_gpu):
    torch = torch_gpu
    L = fa.lib()
    shape = (2, 6, 1024, 69)
    B, H, N, d = shape
    n = B * H * N * d
    need = L.flash_fwd_f32_workspace_bytes(B, H, N, d, 0)
    assert need == 6 * B * H * N * 72 * 2
    rng = np.random.default_rng(7)
    host = [oracle.round_f16(rng.standard_normal(shape).astype(np.float32)) for _ in range(3)]
    ref = oracle.attn_naive(*host)
    # views one element into their allocations: element loads and stores on the caller's side
    bufs = [torch.zeros(n + 8, dtype=torch.float32, device="cuda:0") for _ in range(4)]
    views = [b[1:1 + n].view(shape) for b in bufs]
    for t, x in zip(views, host):
        t.copy_(torch.from_numpy(x).to("cuda:0"))
    assert all(t.data_ptr() % 16 != 0 for t in views)
    ws = torch.empty(need, dtype=torch.uint8, device="cuda:0")
    args = [t.data_ptr() for t in views] + [B, H, N, d, 1.0 / np.sqrt(d), 0]
    assert L.flash_fwd_f32_ws(*args, ws.data_ptr(), need - 1, None) == -1
    assert b"too small" in L.flash_hip_last_error()
    assert L.flash_fwd_f32_ws(*args, ws.data_ptr() + 2, need, None) == -1
    assert L.flash_fwd_f32_ws(*args, ws.data_ptr(), need, None) == 0, L.flash_hip_last_error()
    torch.cuda.synchronize()
    assert_close(views[3].cpu().numpy(), ref, "padded route, unaligned views")
    assert float(bufs[3][0]) == 0.0 and float(bufs[3][n + 1]) == 0.0   # nothing written outside the view
    # the allocation-free entry point takes the same route while the padded copies fit its buffer
    assert need <= 48 << 20
    o2 = torch.full(shape, float("nan"), device="cuda:0")
    tq, tk, tv = (t.clone() for t in views[:3])
    rc = L.flash_fwd_f32_static(tq.data_ptr(), tk.data_ptr(), tv.data_ptr(), o2.data_ptr(), B, H, N, d,
                                1.0 / np.sqrt(d), 0, None)
    assert rc == 0, L.flash_hip_last_error()
    torch.cuda.synchronize()
    assert torch.equal(o2, views[3])
    # ... and falls back to element staging (no scratch) for a shape whose copies do not fit
    big = (1, 40, 2048, 69)
    assert L.flash_fwd_f32_workspace_bytes(*big, 0) > 48 << 20
    tb = [torch.randn(big, device="cuda:0") for _ in range(3)]
    ob = torch.full(big, float("nan"), device="cuda:0")
    rc = L.flash_fwd_f32_static(*(t.data_ptr() for t in tb), ob.data_ptr(), *big, 1.0 / np.sqrt(69), 0, None)
    assert rc == 0, L.flash_hip_last_error()
    torch.cuda.synchronize()
    h = 17
    refb = oracle.attn_naive(*(oracle.round_f16(t[:, h:h + 1].cpu().numpy()) for t in tb))
    assert_close(ob[:, h:h + 1].cpu().numpy(), refb, "static entry, odd head dim beyond the buffer")
