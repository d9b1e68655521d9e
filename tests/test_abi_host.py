"""CPU only: the C-ABI library loads, exports every declared symbol, and its host logic (geometry
query, argument errors, kernel-type names, head sharding) behaves like the reference's."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


def test_library_exports_every_declared_symbol(fa):
    header = open(os.path.join(ROOT, "include", "flash_hip.h")).read()
    declared = set(re.findall(r"\b(flash_[a-z0-9_]+)\s*\(", header))
    declared -= {"flash_hip_status", "flash_launch_info"}
    assert declared == set(fa.ABI_SYMBOLS), declared ^ set(fa.ABI_SYMBOLS)
    L = fa.lib()
    for name in declared:
        assert hasattr(L, name), f"libflash_hip.so does not export {name}"
    assert L.flash_hip_version() >= 100


def test_kernel_type_names_match_reference_order(fa):
    # reference flash/launch.hpp:7-20 (order) and :28-46 (names)
    want = ["naive1D", "scalar2D", "scalar2D_row_tile", "warp_wmma", "block_wmma", "wmma_row_block",
            "block_wmma_async", "mma", "mma_swizzle", "mma_qreg", "mma_qreg_f32x4load", "mma_qreg_async"]
    assert [kt.name for kt in fa.KernelType] == want
    assert [fa.to_string(kt) for kt in fa.KernelType] == want
    with pytest.raises(ValueError):
        fa.to_string(12)


def test_geometry_query(fa):
    # GPT-3 head shape: 16 query tiles of 256 rows, worked by 8 waves of 64 lanes x 32 query rows (two per
    # SIMD); whole tiles run on the persistent kernel: 4 K + 4 V tiles of 64 x 128 halves = 128 KiB of LDS
    info = fa.launch_info(4, 96, 4096, 128, fa.KernelType.mma_qreg_async)
    assert info.tile_size == (64, 256, 128)
    assert info.block_dim == (64, 8)
    assert info.grid_dim == (4, 96, 16)
    assert info.sram_size == 131072 and info.sram_size <= 160 * 1024
    assert info.staging == 2 and info.aligned_rows and info.persistent
    # ragged N: one item per workgroup, 4 waves x 64 query rows, three-slot rings
    info = fa.launch_info(4, 96, 4000, 128, fa.KernelType.mma_qreg_async)
    assert info.block_dim == (64, 4) and info.grid_dim == (4, 96, 16) and not info.persistent
    # small heads keep four waves; whole 512-row tiles give every wave 128 query rows (tile_q 512)
    info = fa.launch_info(16, 32, 2048, 64)
    assert info.block_dim == (64, 4) and info.tile_size == (64, 512, 64) and info.grid_dim == (16, 32, 4)
    assert info.sram_size == 2 * 4 * 64 * 128 and info.persistent
    assert fa.launch_info(16, 32, 768, 64).tile_size == (64, 256, 64)    # 256-row tiles only
    # small heads use the 64-wide tile; ragged N rounds the grid up
    info = fa.launch_info(16, 32, 2049, 64)
    assert info.tile_size == (64, 256, 64) and info.grid_dim == (16, 32, 9) and info.sram_size == 49152
    # head dims that are not a multiple of 8 halves fall back to element staging
    info = fa.launch_info(5, 12, 53, 69, fa.KernelType.naive1D)
    assert info.tile_size[2] == 128 and info.staging == 0 and not info.aligned_rows


def test_geometry_query_per_entry_point(fa):
    """The query describes what each entry point launches: whole-tile fp16 / bf16 shapes are walked by
    the persistent kernel (four-slot rings), ragged ones by one workgroup per item (three-slot rings),
    and fp32 callers without scratch always stage elements through registers."""
    f16 = fa.launch_info(4, 96, 1024, 128)
    assert f16.persistent and f16.sram_size == 2 * 4 * 64 * 256 and f16.staging == 2
    ragged = fa.launch_info(4, 96, 1000, 128)
    assert not ragged.persistent and ragged.sram_size == 2 * 3 * 64 * 256
    bf = fa.launch_info(4, 96, 1024, 128, entry=fa.ENTRY_BF16)
    assert bf == f16
    # fp32 tensors converted inside the attention kernel (flash_fwd_f32): rows of a multiple of 8 elements
    # are staged with 128-bit loads through registers (staging 1) by four waves x 64 query rows, whole
    # tiles on the persistent kernel; other head dims stage element by element (staging 0)
    f32 = fa.launch_info(4, 96, 1024, 128, entry=fa.ENTRY_F32)
    assert f32.persistent and f32.staging == 1 and f32.sram_size == 2 * 4 * 64 * 256
    assert f32.grid_dim == f16.grid_dim and f32.block_dim == (64, 4)
    f32r = fa.launch_info(4, 96, 1000, 128, entry=fa.ENTRY_F32)
    assert not f32r.persistent and f32r.staging == 1 and f32r.sram_size == 2 * 3 * 64 * 256
    f32e = fa.launch_info(5, 12, 53, 69, fa.KernelType.naive1D, entry=fa.ENTRY_F32)
    assert not f32e.persistent and f32e.staging == 0 and f32e.block_dim == (64, 4)
    # fp32 tensors with a workspace: up to N = 2048 (768 for head dims up to 64) the same in-kernel
    # conversion, beyond that the conversion pass and the fp16 kernels with fp32 output (four waves)
    assert fa.launch_info(4, 96, 1024, 128, entry=fa.ENTRY_F32_WS) == f32
    assert fa.launch_info(4, 96, 512, 128, entry=fa.ENTRY_F32_WS).staging == 1
    ws = fa.launch_info(4, 96, 4096, 128, entry=fa.ENTRY_F32_WS)
    assert ws.persistent and ws.staging == 2 and ws.block_dim == (64, 4)
    assert fa.launch_info(4, 96, 4096, 128).block_dim == (64, 8)
    assert fa.launch_info(16, 32, 512, 64, entry=fa.ENTRY_F32_WS).staging == 1
    assert fa.launch_info(16, 32, 2048, 64, entry=fa.ENTRY_F32_WS).tile_size == (64, 512, 64)
    with pytest.raises(ValueError, match="multiple of 8"):
        fa.launch_info(1, 1, 64, 50, fa.KernelType.mma_qreg, entry=fa.ENTRY_BF16)
    with pytest.raises(ValueError, match="entry point"):
        fa.launch_info(1, 1, 64, 64, entry=7)


def test_bench_parent_launches_ranks_and_reports_their_failure(fa):
    """`python bench.py --gpus 2` as one plain process starts two rank processes itself.  Without a GPU
    every rank stops with "needs a HIP device"; the parent must come back non-zero, print no JSON line,
    and never initialise a GPU itself."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("covered by the GPU test of the same path")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                          "--workload", "tiny", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert out.stdout.strip() == ""
    assert out.stderr.count("needs a HIP device") == 2, out.stderr[-2000:]


def test_argument_errors_mirror_reference(fa):
    # reference flash/launch.cu:267-269 (odd d), :284-286 / :301-303 (d % 4), :270-280 (d > 128), :317
    with pytest.raises(ValueError, match="even"):
        fa.launch_info(1, 1, 64, 51, fa.KernelType.mma_qreg)
    with pytest.raises(ValueError, match="multiple of 4"):
        fa.launch_info(1, 1, 64, 50, fa.KernelType.mma_qreg_f32x4load)
    with pytest.raises(ValueError, match="multiple of 4"):
        fa.launch_info(1, 1, 64, 50, fa.KernelType.mma_qreg_async)
    with pytest.raises(ValueError, match="Unsupported tile size"):
        fa.launch_info(1, 1, 64, 136, fa.KernelType.mma_qreg_async)
    with pytest.raises(ValueError, match="Unsupported kernel type"):
        fa.launch_info(1, 1, 64, 64, 12)
    with pytest.raises(ValueError):
        fa.launch_info(0, 1, 64, 64)
    fa.launch_info(1, 1, 64, 50, fa.KernelType.mma_qreg)      # allowed, as in the reference
    fa.launch_info(5, 12, 53, 69, fa.KernelType.mma_swizzle)  # ladder types take any head dim <= 128


def test_forward_rejects_cpu_tensors(fa):
    import torch

    q = torch.zeros(1, 1, 8, 8, dtype=torch.float16)
    with pytest.raises(ValueError, match="no CPU path"):
        fa.forward(q, q, q)
    with pytest.raises(ValueError):
        fa.forward(q[0], q[0], q[0])


def test_null_pointer_and_errors_through_raw_abi(fa):
    L = fa.lib()
    rc = L.flash_fwd_f16(None, None, None, None, 1, 1, 8, 8, 1.0, 11, None)
    assert rc == -1 and b"null" in L.flash_hip_last_error()
    rc = L.flash_fwd_f16(None, None, None, None, 1, 1, 8, 200, 1.0, 11, None)
    assert rc == -2
    # the scale must be positive and finite (checked before the pointers, on every entry point)
    for scale in (0.0, -0.125, float("inf"), float("nan")):
        for fn in (L.flash_fwd_f16, L.flash_fwd_f32, L.flash_fwd_bf16):
            assert fn(None, None, None, None, 1, 1, 8, 8, scale, 11, None) == -1
            assert b"softmax_scale" in L.flash_hip_last_error()


def test_fp32_workspace_size_query(fa):
    """No GPU needed: the fp16 copies the conversion pass writes (256-byte granules) -- K and V (and Q as
    well where a routing switch sends the 128-rows-per-wave layout behind the pass); 0 for shapes that are
    converted inside the attention kernel (rows without 16-byte chunks, sequences up to the register-staging
    limit, whole 512-row tiles of head dims up to 64); -1 with the reference's
    error text for shapes flash::forward rejects."""
    L = fa.lib()
    assert L.flash_fwd_f32_workspace_bytes(4, 96, 4096, 128, 11) == 2 * 4 * 96 * 4096 * 128 * 2
    assert L.flash_fwd_f32_workspace_bytes(16, 32, 1280, 64, 11) == 2 * 16 * 32 * 1280 * 64 * 2
    # whole 512-row query tiles of head dims up to 64 are converted inside the kernel at every length
    assert L.flash_fwd_f32_workspace_bytes(16, 32, 2048, 64, 11) == 0
    assert L.flash_fwd_f32_workspace_bytes(8, 32, 16384, 64, 11) == 0
    assert L.flash_fwd_f32_workspace_bytes(1, 1, 2049, 8, 0) == 2 * ((2049 * 8 * 2 + 255) // 256 * 256)
    assert L.flash_fwd_f32_workspace_bytes(4, 96, 2048, 128, 11) == 0
    assert L.flash_fwd_f32_workspace_bytes(16, 32, 768, 64, 11) == 0
    assert L.flash_fwd_f32_workspace_bytes(1, 1, 3, 8, 0) == 0
    assert L.flash_fwd_f32_workspace_bytes(5, 12, 53, 69, 0) == 0
    assert L.flash_fwd_f32_workspace_bytes(1, 1, 64, 50, 9) == 0
    # rows without 16-byte chunks, from a few GFLOP on: padded copies of K, V (fp16), Q (room for fp32) and
    # a padded fp32 output, i.e. six fp16 tensors of the padded shape
    assert L.flash_fwd_f32_workspace_bytes(4, 96, 1024, 69, 0) == 6 * 4 * 96 * 1024 * 72 * 2
    assert L.flash_fwd_f32_workspace_bytes(4, 96, 1024, 50, 9) == 6 * 4 * 96 * 1024 * 56 * 2
    assert L.flash_fwd_f32_workspace_bytes(1, 1, 1024, 50, 9) == 0      # 0.2 GFLOP: one element-staging kernel
    info = fa.launch_info(4, 96, 1024, 100, entry=fa.ENTRY_F32_WS)
    assert info.staging == 2 and not info.aligned_rows and info.tile_size == (64, 256, 128) and info.persistent
    info = fa.launch_info(4, 96, 1024, 100, entry=fa.ENTRY_F32)
    assert info.staging == 0 and not info.persistent
    assert L.flash_fwd_f32_workspace_bytes(1, 1, 64, 50, 11) == -1
    assert b"multiple of 4" in L.flash_hip_last_error()
    assert L.flash_fwd_f32_workspace_bytes(1, 1, 64, 256, 11) == -1
    assert L.flash_fwd_f32_workspace_bytes(1, 1, 64, 64, 12) == -1


def test_head_chunks_cover_the_slab(fa):
    for per in (1, 2, 3, 48, 768):
        for chunks in (1, 2, 4, 5, 1000):
            cs = fa.head_chunks(per, chunks)
            assert [h for a, b in cs for h in range(a, b)] == list(range(per))
            assert len(cs) <= min(chunks, per)
    assert fa.head_chunks(0, 4) == []
    assert fa.head_chunks(768, 4) == [(0, 192), (192, 384), (384, 576), (576, 768)]


def test_shard_heads_partitions_exactly(fa):
    for total in (1, 7, 384, 6144, 13):
        for world in (1, 2, 3, 4, 8):
            slabs = [fa.shard_heads(total, world, r) for r in range(world)]
            covered = [h for a, b in slabs for h in range(a, b)]
            assert covered == list(range(total))
            assert max(b - a for a, b in slabs) == -(-total // world)
    assert fa.shard_heads(6144, 8, 3) == (2304, 3072)  # 64 x 96 heads over 8 GPUs: 768 each
    with pytest.raises(ValueError):
        fa.shard_heads(8, 2, 2)


def test_runner_cpu_plumbing_mode(fa):
    """`runner cpu`: oracle expression + geometry + argument errors, no GPU (tiny configuration)."""
    if not os.path.exists(fa.RUNNER_PATH):
        pytest.skip("runner not built")
    out = subprocess.run([fa.RUNNER_PATH, "cpu"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CPU plumbing passed" in out.stdout


_GLOO_WORKER = r'''
import importlib, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
fa = importlib.import_module("flash-attention_amd")
from oracle import oracle as orc
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=int(sys.argv[3]), world_size=2)
rank = dist.get_rank()
B, H, N, d = 1, 5, 40, 16          # 5 heads over 2 ranks: slabs of 3 and 2
rng = np.random.default_rng(0)      # same data on both ranks
q, k, v = (rng.standard_normal((B * H, N, d)).astype(np.float32) for _ in range(3))
a, b = fa.shard_heads(B * H, 2, rank)
# CPU stand-in for the per-slab kernel call (the real one needs a GPU): the oracle on the slab
o_local = torch.from_numpy(orc.attn_naive(q[None, a:b], k[None, a:b], v[None, a:b])[0])
o_full = fa.gather_output(o_local, B * H)
want = torch.from_numpy(orc.attn_naive(q[None], k[None], v[None])[0])
assert o_full.shape == want.shape
assert torch.allclose(o_full, want, atol=1e-6), (o_full - want).abs().max()
# chunked gather overlapped with the per-chunk compute (same CPU stand-in for the kernel), for chunk
# counts that do and do not divide the slab, including more chunks than heads
def cpu_kernel(qc, kc, vc, out):
    out.copy_(torch.from_numpy(orc.attn_naive(qc.numpy()[None], kc.numpy()[None], vc.numpy()[None])[0]))
tq, tk, tv = (torch.from_numpy(x[a:b]) for x in (q, k, v))
for exchange in ("p2p", "all_gather"):
    for chunks in (1, 2, 3, 8):
        got = fa.forward_shard_gather(tq, tk, tv, B * H, chunks=chunks, compute=cpu_kernel, exchange=exchange)
        assert got.shape == want.shape
        assert torch.allclose(got, want, atol=1e-6), (exchange, chunks, (got - want).abs().max())
try:
    fa.forward_shard_gather(tq[:1], tk[:1], tv[:1], B * H, compute=cpu_kernel)
    raise SystemExit("a slab of the wrong size was accepted")
except ValueError:
    pass
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_rank_gloo_shard_and_gather(tmp_path):
    """World size 2 on CPU (gloo): each rank owns a contiguous slab of batch*heads, no collective on
    the data path, optional all-gather of O reproduces the unsharded result."""
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    script = tmp_path / "worker.py"
    script.write_text(_GLOO_WORKER)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r}:\n{o}"
        assert f"rank {r} ok" in o
