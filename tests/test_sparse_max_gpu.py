"""GPU tests of the persistent kernels' two-pass scheme.

The first pass over a work item takes the running maximum in a few steps only (both steps of K/V tile 0,
then the second step of every fourth tile from tile 2 on, each looking at the NEXT 32 keys: key blocks 1, 2,
6, 14, 22, ...).  Scores in any other block never move the reference; a probability that leaves the fp16
range on the way (a score more than 16 exp2 units above the reference) shows as an infinite row sum.  The
workgroup then stops its first sweep (two K/V tiles into the next item) and computes that item and the rest
of its list again from a cold start with the maximum taken in every step.  These tests plant scores where the
first pass does not look and require the result of the plain formula in every case.
"""
import numpy as np
import pytest

from conftest import ATOL, assert_close
from test_parity_gpu import routing, run_abi, torch_gpu  # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu

# (shape, routing environment): every persistent kernel the library has
LAYOUTS = [
    ((1, 3, 768, 128), {}),                                 # eight waves x 32 rows
    ((1, 3, 768, 128), {"FLASH_HIP_WIDE_MAX_N": "0"}),      # four waves x 64 rows
    ((1, 3, 768, 64), {}),                                  # four waves x 64 rows, d = 64
    ((1, 3, 1024, 64), {}),                                 # four waves x 128 rows, 512-row items
    ((2, 150, 768, 128), {}),                               # 900 items on at most 256 workgroups
    ((2, 300, 1024, 64), {}),                               # 1200 items of 512 rows
]


def plant(q, k, row, key, gain):
    """Key `key` of every head becomes `gain` times the unit vector of query `row`: that query's score for it
    is gain * |q_row|, about gain * sqrt(d) / sqrt(d) * 1.44 = 1.44 * gain exp2 units after scaling."""
    k[:, :, key] = gain * q[:, :, row] / np.linalg.norm(q[:, :, row], axis=-1, keepdims=True)


def check_heads(fa, oracle, torch, q, k, v, dtype, what):
    B, H = q.shape[:2]
    o = run_abi(fa, torch, q, k, v, fa.KernelType.mma_qreg_async, dtype)
    for (b, h) in sorted({(0, 0), (B - 1, H - 1), (B // 2, H // 2), (0, min(H - 1, 1))}):
        ref = oracle.attn_naive(*(x[b:b + 1, h:h + 1] for x in (q, k, v)))[0, 0]
        assert_close(o[b, h], ref, f"{what} head {(b, h)}")
    return o


@pytest.mark.parametrize("case", LAYOUTS, ids=str)
def test_peaks_in_blocks_the_first_pass_does_not_look_at(fa, oracle, torch_gpu, monkeypatch, case):
    """Moderate peaks (about 14 exp2 units above the rest: probabilities up to 2^14 relative to the
    reference, no overflow, the first pass's result stands) and large ones (about 40 and 70 units: the first
    pass overflows and the item is computed again), in blocks 3, 4, 9 and the last one."""
    shape, env = case
    B, H, N, d = shape
    for gains, label in (((10.0, 11.0, 9.5, 10.5), "moderate"), ((28.0, 50.0, 35.0, 30.0), "large")):
        rng = np.random.default_rng(N + d)
        q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
        spots = [(5, 100), (300, 150), (N - 41, 300), (N - 1, N - 7)]   # (query row, key): keys in blocks 3, 4, 9, last
        for (row, key), gain in zip(spots, gains):
            plant(q, k, row, key, gain)
        q, k, v = (oracle.round_f16(x) for x in (q, k, v))
        with routing(fa, monkeypatch, **env):
            check_heads(fa, oracle, torch_gpu, q, k, v, torch_gpu.float16, f"{label} peaks {shape} {env}")
            if N <= 768 and (B * H <= 16):
                # fp32 tensors, converted inside the same kernels (register staging)
                check_heads(fa, oracle, torch_gpu, q, k, v, torch_gpu.float32, f"{label} peaks fp32 {shape} {env}")


@pytest.mark.parametrize("case", LAYOUTS[:4], ids=str)
def test_scores_that_rise_along_the_sequence(fa, oracle, torch_gpu, monkeypatch, case):
    """Every query's scores drift upwards with the key index.  A gentle drift (about 6 exp2 units per 256
    keys) is followed by the sparse maximum; a steep one (about 25 per 256 keys) outruns it between two
    looks and the items go through the second pass."""
    shape, env = case
    B, H, N, d = shape
    for per_256, label in ((6.0, "gentle"), (25.0, "steep")):
        rng = np.random.default_rng(3 * N + d)
        q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
        u = np.zeros(d, np.float32)
        u[0] = 1.0
        q[..., 0] = 4.0                                    # every query has the same component along u
        # score drift = 4 * k_j[0] / sqrt(d) * log2(e) exp2 units: per_256 units every 256 keys
        slope = per_256 / 256.0 * np.sqrt(d) / (4.0 * np.log2(np.e))
        k[..., 0] = slope * np.arange(N, dtype=np.float32)
        q, k, v = (oracle.round_f16(x) for x in (q, k, v))
        with routing(fa, monkeypatch, **env):
            check_heads(fa, oracle, torch_gpu, q, k, v, torch_gpu.float16, f"{label} drift {shape} {env}")


def test_long_item_lists_with_overflow_at_any_place(fa, oracle, torch_gpu, monkeypatch):
    """Eight workgroups (FLASH_HIP_GRID) share 8400 one-tile items, 1050 each.  Peaks that overflow the first
    pass sit in heads all over the range: in a workgroup's first item, its second, its last, and in the middle
    of a list (the first sweep ends there and the second takes over for the rest); some workgroups meet none."""
    shape = (1, 8400, 256, 64)
    B, H, N, d = shape
    rng = np.random.default_rng(99)
    q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    # items are dealt to XCDs in ranges of 1050 heads and one workgroup serves each range here: the first finds
    # a peak in its first item, the second in its second, the third and the last in their last, the fourth and
    # fifth in the middle of their lists, the sixth and seventh nowhere
    hot = [0, 7, 1051, 3149, 4100, 5000, 5249, 8399]
    for h in hot:
        q1, k1 = q[:, h:h + 1], k[:, h:h + 1]
        plant(q1, k1, 17, 120, 40.0)
        plant(q1, k1, 200, 250, 60.0)
    q, k, v = (oracle.round_f16(x) for x in (q, k, v))
    with routing(fa, monkeypatch, FLASH_HIP_GRID=8):
        o = run_abi(fa, torch_gpu, q, k, v, fa.KernelType.mma_qreg_async, torch_gpu.float16)
    assert np.isfinite(o).all()
    for h in sorted(set(hot + [1, 6, 8, 1049, 1050, 1052, 2100, 3148, 3150, 4099, 4101, 5248, 5250, 6300, 7349, 7350, 8398])):
        ref = oracle.attn_naive(*(x[:, h:h + 1] for x in (q, k, v)))[0, 0]
        assert_close(o[0, h], ref, f"head {h} of {shape} on eight workgroups")


# (shape, dtype name, workspace, routing environment): one case per persistent kernel, ten items per workgroup
RESUME_CASES = [
    ((1, 20, 1024, 128), "float16", False, {}),                               # eight waves x 32 rows
    ((1, 20, 1024, 128), "float16", False, {"FLASH_HIP_WIDE_MAX_N": "0"}),    # four waves x 64 rows
    ((1, 40, 1024, 64), "float16", False, {}),                                # four waves x 128 rows (512-row items)
    ((1, 20, 1024, 64), "float16", False, {"FLASH_HIP_TALL": "0"}),           # four waves x 64 rows, d = 64
    ((1, 20, 1024, 128), "float32", False, {}),                               # fp32 K/V converted in the kernel
    ((1, 40, 1024, 64), "float32", False, {}),                                # the same, 128 rows per wave
    ((1, 20, 1024, 64), "float32", False, {"FLASH_HIP_TALL": "0"}),           # the same, 64 rows per wave
    ((1, 20, 1024, 128), "float32", True, {"FLASH_HIP_VEC_MAX_N": "0"}),      # conversion pass, fp32 Q and O in the kernel
    ((1, 20, 512, 128), "float32", True, {"FLASH_HIP_VEC_MAX_N": "0"}),       # the same in the eight-wave layout
]


@pytest.mark.parametrize("case", RESUME_CASES, ids=str)
def test_giving_up_and_resuming_in_every_persistent_kernel(fa, oracle, torch_gpu, monkeypatch, case):
    """Eight workgroups (one per XCD) walk lists of five to ten items.  Overflowing peaks sit in the second
    item of the first list, the last item of the second, in the middle of the third (two of them); the other
    lists meet none.  Every head is compared: the items before the one that overflowed keep the first sweep's
    result, that item and those behind it come from the second sweep."""
    shape, dtype_name, ws, env = case
    B, H, N, d = shape
    dtype = getattr(torch_gpu, dtype_name)
    rng = np.random.default_rng(H + N + d)
    q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    tile_q = 512 if (d <= 64 and N % 512 == 0 and env.get("FLASH_HIP_TALL") != "0") else 256
    nqt = N // tile_q
    items = H * nqt
    per_list = items // 8                      # items are dealt to the XCDs in contiguous ranges
    assert items % 8 == 0 and per_list >= 5
    hot_items = [1, 2 * per_list - 1, 2 * per_list + 2, 2 * per_list + 3]
    for it in hot_items:
        h, qt = divmod(it, nqt)
        row = qt * tile_q + 17                 # a query row of that item; key 100 lies in a block nobody looks at
        plant(q[:, h:h + 1], k[:, h:h + 1], row, 100, 40.0)
    q, k, v = (oracle.round_f16(x) for x in (q, k, v))
    with routing(fa, monkeypatch, FLASH_HIP_GRID=8, **env):
        o = run_abi(fa, torch_gpu, q, k, v, fa.KernelType.mma_qreg_async, dtype, workspace=ws)
    for h in range(H):
        ref = oracle.attn_naive(*(x[:, h:h + 1] for x in (q, k, v)))[0, 0]
        assert_close(o[0, h], ref, f"head {h} of {shape} {dtype_name} {env}")


RAGGED = [(1, 3, 1000, 128), (2, 5, 1000, 64), (1, 2, 4000, 128), (1, 3, 450, 128)]


@pytest.mark.parametrize("shape", RAGGED, ids=str)
def test_one_item_kernels_follow_peaks_anywhere(fa, oracle, torch_gpu, shape):
    """Ragged N runs one work item per workgroup, with the maximum taken in every step (a first pass that
    looks at one tile in three, with a repeat on overflow, was built for these kernels too and measured
    1.4 - 2.7 % slower).  The same planted scores as above: moderate peaks and large ones, fp16 and fp32 tensors."""
    B, H, N, d = shape
    assert not fa.launch_info(*shape).persistent
    for gains, label in (((10.0, 11.0, 9.5), "moderate"), ((28.0, 50.0, 35.0), "large")):
        rng = np.random.default_rng(N + d + 1)
        q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
        # keys 100 and 150 lie in tiles 1 and 2, key 300 in tile 4: tiles that do not look when N > 320
        for (row, key), gain in zip([(5, 100), (300, 150), (N - 41, 300)], gains):
            plant(q, k, row, key, gain)
        q, k, v = (oracle.round_f16(x) for x in (q, k, v))
        for dtype in (torch_gpu.float16, torch_gpu.float32):
            check_heads(fa, oracle, torch_gpu, q, k, v, dtype, f"{label} peaks, ragged {shape} {dtype}")


def test_nan_in_one_head_stays_there_and_the_kernel_returns(fa, oracle, torch_gpu):
    """A NaN in K poisons every query of its head (as it does in the plain formula); the first sweep stops at
    an item of that head, the second computes it again (NaN again) and goes on; every other head is untouched."""
    shape = (1, 6, 512, 128)
    rng = np.random.default_rng(5)
    q, k, v = (oracle.round_f16(rng.standard_normal(shape).astype(np.float32)) for _ in range(3))
    k[0, 2, 77, 5] = np.nan
    o = run_abi(fa, torch_gpu, q, k, v, fa.KernelType.mma_qreg_async, torch_gpu.float16)
    assert np.isnan(o[0, 2]).all()
    for h in (0, 1, 3, 5):
        ref = oracle.attn_naive(*(x[:, h:h + 1] for x in (q, k, v)))[0, 0]
        assert_close(o[0, h], ref, f"head {h} beside the NaN head")


def test_first_pass_alone_is_not_enough(fa, oracle, torch_gpu):
    """The large peaks of the first test do overflow fp16 probabilities relative to the first block's
    maximum (so a correct result there comes from the second pass): checked on the numbers themselves."""
    shape = (1, 1, 768, 128)
    rng = np.random.default_rng(768 + 128)
    q, k, v = (rng.standard_normal(shape).astype(np.float32) for _ in range(3))
    plant(q, k, 5, 100, 28.0)
    q, k = oracle.round_f16(q), oracle.round_f16(k)
    s = (q[0, 0, 5] @ k[0, 0].T) / np.sqrt(128.0) * np.log2(np.e)     # exp2 units
    assert s[100] - s[:96].max() > 16.5, "the planted score must exceed every score the first pass looks at by > 16"
    assert ATOL == 2e-3
