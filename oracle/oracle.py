"""ctypes + numpy front end of the CPU oracle (oracle/attention_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py``.  Nothing under ``flash-attention_amd/`` imports it.

Parity status: pinned — see the header of attention_oracle.c and tests/test_oracle.py (golden
vectors are outputs of the reference's own ``manual_attn`` definition, benchmarks/bench.py:39-43,
executed unmodified by tests/golden/make_golden.py; ``--check`` requires them bit-identical).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    """Compile attention_oracle.c into oracle/liboracle.so with gcc (OpenMP)."""
    src = os.path.join(_HERE, "attention_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(
            ["gcc", "-O3", "-std=c11", "-fopenmp", "-fPIC", "-shared", src, "-o", _LIB_PATH, "-lm"]
        )
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        i64 = ctypes.c_int64
        L.oracle_attn_naive.argtypes = [fp, fp, fp, fp, i64, i64, i64, i64, ctypes.c_float]
        L.oracle_attn_naive.restype = ctypes.c_int
        L.oracle_attn_tiled.argtypes = [fp, fp, fp, fp, i64, i64, i64, i64, ctypes.c_float,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_attn_tiled.restype = ctypes.c_int
        L.oracle_attn_tiled_q16.argtypes = [fp, fp, fp, fp, i64, i64, i64, i64, ctypes.c_float,
                                            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_attn_tiled_q16.restype = ctypes.c_int
        L.oracle_round_f16_array.argtypes = [fp, fp, i64]
        L.oracle_round_f16_array.restype = None
        L.oracle_num_threads.argtypes = []
        L.oracle_num_threads.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(x) -> np.ndarray:
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def round_f16(x) -> np.ndarray:
    """fp32 -> fp16 (round to nearest even) -> fp32 with the oracle's own software conversion."""
    a = _f32(x)
    out = np.empty_like(a)
    lib().oracle_round_f16_array(_ptr(a), _ptr(out), a.size)
    return out


def round_bf16(x) -> np.ndarray:
    """fp32 -> bfloat16 -> fp32, round to nearest even (what torch's .bfloat16() does); pure numpy."""
    a = _f32(x)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def attn_naive(q, k, v, scale: float | None = None) -> np.ndarray:
    """softmax(q k^T * scale) v in fp32, q/k/v of shape [B, H, N, d] (reference main.cpp:12-17)."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    B, H, N, d = q.shape
    assert k.shape == q.shape and v.shape == q.shape
    if scale is None:
        scale = 1.0 / np.sqrt(d)
    o = np.empty_like(q)
    rc = lib().oracle_attn_naive(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, d, float(scale))
    if rc != 0:
        raise MemoryError("oracle_attn_naive failed")
    return o


def attn_tiled(q, k, v, scale: float | None = None, Br: int = 128, Bc: int = 64, p_f16: bool = False,
               defer_norm: bool = False, return_ml: bool = False, q_prescale_f16: bool = False):
    """The hot kernel's tile loop (reference flash/kernel_mma_qreg.cuh:100-266) on the CPU.
    ``q_prescale_f16`` models the MI355X kernel's rounding of Q * scale * log2(e) to fp16 (exp2 domain;
    the returned m is then in log2 units)."""
    q, k, v = _f32(q), _f32(k), _f32(v)
    B, H, N, d = q.shape
    if scale is None:
        scale = 1.0 / np.sqrt(d)
    o = np.empty_like(q)
    m = np.empty((B, H, N), dtype=np.float32)
    l = np.empty((B, H, N), dtype=np.float32)
    rc = lib().oracle_attn_tiled_q16(_ptr(q), _ptr(k), _ptr(v), _ptr(o), B, H, N, d, float(scale), Br, Bc,
                                     int(p_f16), int(defer_norm), int(q_prescale_f16), _ptr(m), _ptr(l))
    if rc != 0:
        raise RuntimeError(f"oracle_attn_tiled failed rc={rc}")
    return (o, m, l) if return_ml else o


def attn_numpy(q, k, v, scale: float | None = None) -> np.ndarray:
    """Independent numpy statement of the same formula in float64 (cross-check of the C code)."""
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    d = q.shape[-1]
    if scale is None:
        scale = 1.0 / np.sqrt(d)
    s = np.einsum("bhid,bhjd->bhij", q, k) * scale
    s -= s.max(axis=-1, keepdims=True)
    p = np.exp(s)
    p /= p.sum(axis=-1, keepdims=True)
    return np.einsum("bhij,bhjd->bhid", p, v)
