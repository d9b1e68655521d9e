"""Shared pytest fixtures.  GPU tests are marked ``gpu``; everything else runs on CPU only."""
import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

# the reference runner's shapes (reference main.cpp:157-193) and the tiny plumbing configuration
RUNNER_SHAPES = [(1, 1, 50, 32), (1, 1, 100, 32), (1, 1, 64, 50), (5, 12, 53, 69), (1, 1, 30, 128)]
TINY_SHAPE = (2, 4, 256, 64)

# parity bar: max-abs <= 2e-3, and |res - ref| <= ATOL + RTOL*|ref| (the reference's allclose form,
# main.cpp:49, with the fp16-operand tolerances)
ATOL = 2e-3
RTOL = 1e-2


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fa():
    """The product package (directory name has a hyphen)."""
    return importlib.import_module("flash-attention_amd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc

    orc.build()
    return orc


def golden_path(shape):
    B, H, N, d = shape
    return os.path.join(GOLDEN_DIR, f"attn_B{B}_H{H}_N{N}_d{d}.npz")


def load_golden(shape):
    return np.load(golden_path(shape), allow_pickle=False)


def assert_close(res, ref, what=""):
    res = np.asarray(res, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert res.shape == ref.shape, (res.shape, ref.shape)
    assert not np.isnan(res).any(), f"{what}: NaN in result"
    err = np.abs(res - ref)
    bad = err > ATOL + RTOL * np.abs(ref)
    assert err.max() <= ATOL and not bad.any(), (
        f"{what}: max_abs={err.max():.3e} (bar {ATOL}), {int(bad.sum())} elements outside "
        f"{ATOL}+{RTOL}*|ref|, first at {np.argwhere(bad)[:3].tolist()}")
