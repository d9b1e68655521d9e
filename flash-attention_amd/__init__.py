"""flash-attention for MI355X (gfx950): Python host side of the attention-forward hot path.

This package is the Python mirror of the reference's operator surface for the one path it
accelerates (reference flash/launch.hpp:7-24 ``KernelType`` / ``flash::forward``,
flash/KernelParameters.hpp:63-71,383-422 launch geometry, pyflash/pyflash.cpp:10-35 torch ops).
All arithmetic happens in ``lib/libflash_hip.so`` (hand-written HIP, C ABI in include/flash_hip.h);
torch is used for device memory and streams only.  There is no CPU fallback: if the library is
missing or there is no GPU, calls raise.

The directory name contains a hyphen, so import it with
``importlib.import_module("flash-attention_amd")`` with the repository root on ``sys.path``.
"""
from __future__ import annotations

import contextlib
import ctypes
import enum
import os
from dataclasses import dataclass
from typing import Optional, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "lib", "libflash_hip.so")
PYFLASH_PATH = os.path.join(REPO_ROOT, "build", "pyflash", "libpyflash.so")
RUNNER_PATH = os.path.join(REPO_ROOT, "build", "runner")


class KernelType(enum.IntEnum):
    """Same members, same order as the reference's ``enum class KernelType`` (flash/launch.hpp:7-20)."""

    naive1D = 0
    scalar2D = 1
    scalar2D_row_tile = 2
    warp_wmma = 3
    block_wmma = 4
    wmma_row_block = 5
    block_wmma_async = 6
    mma = 7
    mma_swizzle = 8
    mma_qreg = 9
    mma_qreg_f32x4load = 10
    mma_qreg_async = 11


# flash_hip_status (include/flash_hip.h)
_OK, _EINVAL, _EDIM, _ERUNTIME, _ELDS = 0, -1, -2, -3, -4


class _LaunchInfo(ctypes.Structure):
    _fields_ = [(n, ctypes.c_uint32) for n in (
        "tile_kv", "tile_q", "tile_d", "lds_bytes", "block_x", "block_y",
        "grid_x", "grid_y", "grid_z", "staging", "aligned_rows", "persistent")]


@dataclass(frozen=True)
class LaunchInfo:
    """What the reference's KernelParametersBase reports (flash/KernelParameters.hpp:63-71)."""

    tile_size: Tuple[int, int, int]   # (Bc, Br, padded head dim)  == tileSize()
    sram_size: int                    # dynamic LDS bytes          == sramSize()
    block_dim: Tuple[int, int]        # (64 lanes, waves)          == blockDim()
    grid_dim: Tuple[int, int, int]    # (B, H, ceil(N/Br))         == gridDim()
    staging: int
    aligned_rows: bool
    persistent: bool = False          # items walked by one persistent workgroup per compute unit


# every symbol include/flash_hip.h declares
ABI_SYMBOLS = (
    "flash_fwd_f16", "flash_fwd_bf16", "flash_fwd_f32", "flash_fwd_f32_ws", "flash_fwd_f32_static",
    "flash_fwd_f32_workspace_bytes", "flash_fwd_query",
    "flash_fwd_query_entry",
    "flash_kernel_type_name",
    "flash_hip_last_error", "flash_hip_version", "flash_hip_reload_switches",
)

_lib: Optional[ctypes.CDLL] = None


def lib() -> ctypes.CDLL:
    """Load libflash_hip.so (once).  Raises ImportError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the attention kernel.")
    # torch bundles its own HIP runtime (same soname); importing it first makes this process use
    # exactly one copy of it.
    import torch  # noqa: F401

    L = ctypes.CDLL(LIB_PATH)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    for name in ("flash_fwd_f16", "flash_fwd_bf16", "flash_fwd_f32", "flash_fwd_f32_static"):
        fn = getattr(L, name)
        fn.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, ctypes.c_float, ctypes.c_int, vp]
        fn.restype = ctypes.c_int
    L.flash_fwd_f32_ws.argtypes = [vp, vp, vp, vp, i64, i64, i64, i64, ctypes.c_float, ctypes.c_int, vp, i64, vp]
    L.flash_fwd_f32_ws.restype = ctypes.c_int
    L.flash_fwd_f32_workspace_bytes.argtypes = [i64, i64, i64, i64, ctypes.c_int]
    L.flash_fwd_f32_workspace_bytes.restype = i64
    L.flash_fwd_query.argtypes = [i64, i64, i64, i64, ctypes.c_int, ctypes.POINTER(_LaunchInfo)]
    L.flash_fwd_query.restype = ctypes.c_int
    L.flash_fwd_query_entry.argtypes = [i64, i64, i64, i64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_LaunchInfo)]
    L.flash_fwd_query_entry.restype = ctypes.c_int
    L.flash_kernel_type_name.argtypes = [ctypes.c_int]
    L.flash_kernel_type_name.restype = ctypes.c_char_p
    L.flash_hip_last_error.argtypes = []
    L.flash_hip_last_error.restype = ctypes.c_char_p
    L.flash_hip_version.argtypes = []
    L.flash_hip_version.restype = ctypes.c_int
    L.flash_hip_reload_switches.argtypes = []
    L.flash_hip_reload_switches.restype = ctypes.c_int
    _lib = L
    return L


def _raise(rc: int) -> None:
    msg = lib().flash_hip_last_error().decode("utf-8", "replace")
    # std::invalid_argument in the reference (flash/launch.cu:79-85,267-269,284-286,317) -> ValueError
    if rc in (_EINVAL, _EDIM):
        raise ValueError(msg)
    raise RuntimeError(msg)  # std::runtime_error / gpuErrchk in the reference (launch.cu:63-77,146)


def to_string(kernel_type) -> str:
    """Reference ``to_string(flash::KernelType)`` (flash/launch.hpp:28-46)."""
    name = lib().flash_kernel_type_name(int(kernel_type))
    if name is None:
        raise ValueError("wrong kernel type")
    return name.decode()


ENTRY_F16, ENTRY_BF16, ENTRY_F32, ENTRY_F32_WS = 0, 1, 2, 3  # flash_hip_entry (include/flash_hip.h)


def launch_info(B: int, H: int, N: int, d: int, kernel_type=KernelType.mma_qreg_async, entry: int = ENTRY_F16) -> LaunchInfo:
    """Counterpart of ``KernelParametersFactory::create`` (flash/KernelParameters.hpp:383-422).
    ``entry`` names the entry point described: fp16 (default; also fp32 with scratch), bf16, or fp32
    without scratch (element staging)."""
    info = _LaunchInfo()
    rc = lib().flash_fwd_query_entry(B, H, N, d, int(kernel_type), int(entry), ctypes.byref(info))
    if rc != _OK:
        _raise(rc)
    return LaunchInfo(
        tile_size=(info.tile_kv, info.tile_q, info.tile_d), sram_size=info.lds_bytes,
        block_dim=(info.block_x, info.block_y), grid_dim=(info.grid_x, info.grid_y, info.grid_z),
        staging=info.staging, aligned_rows=bool(info.aligned_rows), persistent=bool(info.persistent))


def forward(Q, K, V, kernel_type=KernelType.mma_qreg_async, *, out=None, sync: bool = True):
    """Counterpart of ``flash::forward(Q, K, V, kernel_type)`` (flash/launch.hpp:23-24).

    Q, K, V: 4-D ``[B, H, N, d]`` tensors on a HIP device, fp16 (native), bf16 (same kernels with
    bfloat16 matrix instructions, d a multiple of 8) or fp32 (the reference's
    dtype; rounded to fp16 inside the kernel or by one conversion pass into per-call scratch, which
    also pads rows to 16-byte chunks when d is not a multiple of 8; result returned as fp32).  Returns a new tensor O of
    the shape and dtype of Q.  Like the reference (flash/launch.cu:320-321) the call blocks until
    the kernel has finished unless ``sync=False``.
    """
    import torch

    if Q.dim() != 4:
        raise ValueError("Q must be [B, H, N, d]")
    if K.shape != Q.shape or V.shape != Q.shape:
        raise ValueError("K and V must have the shape of Q")
    if not Q.is_cuda or K.device != Q.device or V.device != Q.device:
        raise ValueError("Q, K, V must live on the same HIP device (there is no CPU path)")
    if Q.dtype not in (torch.float16, torch.bfloat16, torch.float32) or K.dtype != Q.dtype or V.dtype != Q.dtype:
        raise ValueError("Q, K, V must all be float16, all bfloat16 or all float32")
    Q, K, V = Q.contiguous(), K.contiguous(), V.contiguous()
    B, H, N, d = Q.shape
    if out is None:
        out = torch.empty_like(Q)  # write-only for this kernel; the reference needs zeros (launch.cu:195)
    elif out.shape != Q.shape or out.dtype != Q.dtype or not out.is_contiguous() or out.device != Q.device:
        raise ValueError("out must be a contiguous tensor like Q")
    scale = 1.0 / float(d) ** 0.5
    # (entering a device context costs 2-3 us per call, which shows at short sequences: only when needed)
    ctx = contextlib.nullcontext() if Q.device.index == torch.cuda.current_device() else torch.cuda.device(Q.device)
    with ctx:
        stream = torch.cuda.current_stream(Q.device)
        sp = ctypes.c_void_p(stream.cuda_stream)
        if Q.dtype in (torch.float16, torch.bfloat16):
            fn = lib().flash_fwd_f16 if Q.dtype == torch.float16 else lib().flash_fwd_bf16
            rc = fn(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), B, H, N, d, scale, int(kernel_type), sp)
        else:
            # scratch for the fp16 (or padded) copies of Q, K, V (the reference allocates per-call scratch too,
            # flash/launch.cu:150-157); torch's allocator re-issues the block in stream order
            # (a rejected shape reports -1 here; the call below then returns the matching status)
            ws_bytes = max(0, lib().flash_fwd_f32_workspace_bytes(B, H, N, d, int(kernel_type)))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=Q.device) if ws_bytes else None
            rc = lib().flash_fwd_f32_ws(Q.data_ptr(), K.data_ptr(), V.data_ptr(), out.data_ptr(), B, H, N, d,
                                        scale, int(kernel_type), ws.data_ptr() if ws is not None else None,
                                        ws_bytes, sp)
        if rc != _OK:
            _raise(rc)
        if sync:
            stream.synchronize()
    return out


def load_pyflash() -> None:
    """``torch.ops.load_library(build/pyflash/libpyflash.so)`` as the reference's benchmark does
    (benchmarks/bench.py:12-14); afterwards ``torch.ops.pyflash.<name>(q, k, v)`` is available."""
    import torch

    if not os.path.exists(PYFLASH_PATH):
        raise ImportError(f"{PYFLASH_PATH} not found: run __graft_entry__.build()")
    torch.ops.load_library(PYFLASH_PATH)


# ---- batch x heads sharding across GPUs (one process per GPU) -------------------------------------
def shard_heads(total_heads: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous slab [start, stop) of the flattened batch*heads axis owned by ``rank``.

    Heads are independent (reference flash/kernel_mma_qreg.cuh:30-34: one block per (batch, head),
    offset (b*H + h)*N*d), and the layout is [B, H, N, d], so a slab of consecutive flattened heads
    is one contiguous byte range.  Slab size is ceil(total/world); the last ranks may own fewer (or
    zero) heads.
    """
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size / rank")
    per = -(-total_heads // world_size)
    start = min(rank * per, total_heads)
    stop = min(start + per, total_heads)
    return start, stop


def forward_shard(Qs, Ks, Vs, kernel_type=KernelType.mma_qreg_async, *, sync: bool = True):
    """Forward of this rank's slab.  ``Qs`` etc. are ``[heads_local, N, d]`` (the slab of the
    flattened [B*H, N, d] tensor); no collective is involved."""
    hl, N, d = Qs.shape
    out = forward(Qs.view(1, hl, N, d), Ks.view(1, hl, N, d), Vs.view(1, hl, N, d), kernel_type, sync=sync)
    return out.view(hl, N, d)


def gather_output(O_local, total_heads: int, group=None):
    """Optional output gather: all ranks receive the full ``[total_heads, N, d]`` O.

    Uses ``torch.distributed.all_gather`` (RCCL over xGMI with the nccl backend, gloo on CPU) on
    equally sized, zero-padded slabs; it is outside the attention hot loop and only needed by
    callers that want O replicated.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = -(-total_heads // world)
    pad = per - O_local.shape[0]
    if pad:
        O_local = torch.cat([O_local, O_local.new_zeros((pad,) + tuple(O_local.shape[1:]))], dim=0)
    bufs = [torch.empty_like(O_local) for _ in range(world)]
    dist.all_gather(bufs, O_local.contiguous(), group=group)
    return torch.cat(bufs, dim=0)[:total_heads]


def head_chunks(per_rank: int, chunks: int):
    """Split a slab of ``per_rank`` heads into at most ``chunks`` consecutive ranges of
    ceil(per_rank/chunks) heads (the last one may be shorter).  Returns [(c0, c1), ...]."""
    if per_rank <= 0:
        return []
    chunks = max(1, min(int(chunks), per_rank))
    step = -(-per_rank // chunks)
    return [(c0, min(c0 + step, per_rank)) for c0 in range(0, per_rank, step)]


def forward_shard_gather(Qs, Ks, Vs, total_heads: int, kernel_type=KernelType.mma_qreg_async, *,
                         chunks: int = 4, group=None, sync: bool = True, out=None, compute=None,
                         exchange: str = "p2p"):
    """Forward of this rank's slab with the output all-gather overlapped with the attention kernels.

    The slab is cut into ``chunks`` ranges of heads.  Each range is computed straight into its final
    place in the replicated ``[total_heads, N, d]`` output; as soon as the kernel of one range has
    been queued, the all-gather of that range is started (``async_op=True``: with the nccl backend
    RCCL runs it on its own stream behind the kernel, over xGMI) while the compute stream goes on with
    the next range.  Only the gather of the last range is exposed.  Every rank returns the full O.

    Heads are independent (reference flash/kernel_mma_qreg.cuh:30-34), so any cut of the slab is
    valid; there is no collective inside the attention itself.  The slab layout is the one of
    :func:`shard_heads`: rank r owns flattened heads [r*per, r*per + heads_local), per =
    ceil(total_heads / world); ranks may own fewer heads than ``per`` (or none).

    ``exchange`` selects how a range travels.  ``"p2p"`` (default): every rank sends its range to every
    peer and receives the peers' ranges straight into their final places, as one batch of point-to-point
    operations -- on a node whose GPUs are fully connected by point-to-point xGMI links every transfer
    has a link of its own, and nothing is staged or copied.  ``"all_gather"``: one all-gather per range
    on a list of output views (the process group gathers into a flat buffer and copies from there).

    ``out`` may hold a preallocated ``[world * per, N, d]`` buffer (reused from step to step).
    ``compute(q, k, v, out)`` replaces the kernel call; it exists for hosts without a GPU (CPU tests
    of the chunking and the collective).  The default is the HIP kernel and raises without one.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    per = -(-total_heads // world)
    start, stop = shard_heads(total_heads, world, rank)
    hl, N, d = Qs.shape
    if hl != stop - start:
        raise ValueError(f"rank {rank} owns {stop - start} heads, got a slab of {hl}")
    if compute is None:
        def compute(q, k, v, out):
            n = q.shape[0]
            forward(q.view(1, n, N, d), k.view(1, n, N, d), v.view(1, n, N, d), kernel_type,
                    out=out.view(1, n, N, d), sync=False)

    if out is None:
        full = torch.empty((world * per, N, d), dtype=Qs.dtype, device=Qs.device)
    else:
        full = out
        if tuple(full.shape) != (world * per, N, d) or full.dtype != Qs.dtype or full.device != Qs.device \
                or not full.is_contiguous():
            raise ValueError(f"out must be a contiguous [{world * per}, {N}, {d}] tensor like the slab")
    works = []
    for c0, c1 in head_chunks(per, chunks):
        mine = full[rank * per + c0: rank * per + c1]
        n_own = max(0, min(c1, hl) - c0)
        if n_own:
            compute(Qs[c0:c0 + n_own], Ks[c0:c0 + n_own], Vs[c0:c0 + n_own], mine[:n_own])
        if n_own < c1 - c0:
            mine[n_own:].zero_()  # padding rows of a short slab
        views = [full[g * per + c0: g * per + c1] for g in range(world)]
        if exchange == "all_gather":
            works.append(dist.all_gather(views, mine, group=group, async_op=True))
        elif exchange == "p2p":
            peers = [g for g in range(world) if g != rank]
            if not peers:
                continue
            to_global = (lambda g: g) if group is None else (lambda g: dist.get_global_rank(group, g))
            ops = [dist.P2POp(dist.isend, mine, to_global(g), group) for g in peers]
            ops += [dist.P2POp(dist.irecv, views[g], to_global(g), group) for g in peers]
            works.extend(dist.batch_isend_irecv(ops))
        else:
            raise ValueError("exchange must be 'p2p' or 'all_gather'")
    for w in works:
        w.wait()
    if sync and full.is_cuda:
        torch.cuda.current_stream(full.device).synchronize()
    return full[:total_heads]


__all__ = [
    "KernelType", "LaunchInfo", "ABI_SYMBOLS", "ENTRY_F16", "ENTRY_BF16", "ENTRY_F32", "ENTRY_F32_WS", "lib", "to_string", "launch_info", "forward",
    "load_pyflash", "shard_heads", "forward_shard", "gather_output", "head_chunks",
    "forward_shard_gather",
    "LIB_PATH", "PYFLASH_PATH", "RUNNER_PATH", "REPO_ROOT",
]
