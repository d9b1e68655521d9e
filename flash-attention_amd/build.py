"""In-tree build of every native artefact (plain compiler invocations, no cmake needed).

  flash-attention_amd/lib/libflash_hip.so   HIP kernels + C ABI            (hipcc --offload-arch=gfx950)
  build/libflash.a                          flash::forward on libtorch     (g++)
  build/pyflash/libpyflash.so               torch.ops.pyflash.*            (g++), path as in the reference
  build/runner                              correctness tests / benchmark  (g++)
  tests/native/selftest                     torch-free parity harness      (hipcc host + oracle)
  oracle/liboracle.so                       CPU oracle (test infrastructure only)
  build/flash_hip.s                         device assembly of the kernels, on request (`asm`), for
                                            tests/test_kernel_asm.py and tools/loopstats.py

Steps are skipped when the output is newer than all of its inputs.
"""
from __future__ import annotations

import concurrent.futures
import glob
import os
import shlex
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
BUILD = os.path.join(ROOT, "build")
ARCH = "gfx950"


def _run(cmd):
    print("+", " ".join(shlex.quote(c) for c in cmd), flush=True)
    subprocess.check_call(cmd)


def _stale(out, inputs):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(i) > t for i in inputs)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build_hip_library(force=False):
    out = os.path.join(HERE, "lib", "libflash_hip.so")
    srcs = glob.glob(os.path.join(HERE, "csrc", "*")) + [os.path.join(ROOT, "include", "flash_hip.h")]
    if force or _stale(out, srcs):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        _run([_hipcc(), f"--offload-arch={ARCH}"] + HIP_FLAGS + ["-fPIC", "-shared",
              os.path.join(HERE, "csrc", "flash_hip.hip"), "-o", out])
    return out


HIP_FLAGS = ["-O3", "-std=c++17", "-fno-honor-nans"]


def build_device_asm(force=False):
    """gfx950 assembly of every kernel in the library (same flags as the library itself)."""
    out = os.path.join(BUILD, "flash_hip.s")
    srcs = glob.glob(os.path.join(HERE, "csrc", "*")) + [os.path.join(ROOT, "include", "flash_hip.h")]
    if force or _stale(out, srcs):
        os.makedirs(BUILD, exist_ok=True)
        _run([_hipcc(), f"--offload-arch={ARCH}"] + HIP_FLAGS + ["--cuda-device-only", "-S",
             os.path.join(HERE, "csrc", "flash_hip.hip"), "-o", out])
    return out


def build_oracle(force=False):
    out = os.path.join(ROOT, "oracle", "liboracle.so")
    src = os.path.join(ROOT, "oracle", "attention_oracle.c")
    if force or _stale(out, [src]):
        _run(["gcc", "-O3", "-std=c11", "-fopenmp", "-fPIC", "-shared", src, "-o", out, "-lm"])
    return out


def build_selftest(force=False):
    out = os.path.join(ROOT, "tests", "native", "selftest")
    src = os.path.join(ROOT, "tests", "native", "selftest.cpp")
    deps = [src, build_hip_library(), build_oracle()]
    if force or _stale(out, deps):
        _run([_hipcc(), "-O2", "-std=c++17", src, "-o", out,
              "-L" + os.path.join(HERE, "lib"), "-lflash_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
              "-Wl,-rpath,$ORIGIN/../../flash-attention_amd/lib", "-Wl,-rpath,$ORIGIN/../../oracle"])
    return out


def _torch_flags():
    import torch
    from torch.utils import cpp_extension as ce

    inc = []
    for p in ce.include_paths():
        inc += ["-isystem", p]
    inc += ["-isystem", sysconfig.get_paths()["include"], "-isystem", "/opt/rocm/include"]
    abi = int(torch._C._GLIBCXX_USE_CXX11_ABI)
    cflags = ["-O2", "-std=c++17", "-fPIC", "-DUSE_ROCM=1", "-D__HIP_PLATFORM_AMD__=1",
              f"-D_GLIBCXX_USE_CXX11_ABI={abi}", "-Wno-deprecated-declarations"] + inc
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    # torch's lib directory comes first: it holds the HIP runtime the process must share
    ldflags = ["-L" + tlib, "-Wl,-rpath," + tlib, "-ltorch", "-ltorch_cpu", "-ltorch_hip", "-lc10", "-lc10_hip",
               "-lamdhip64", "-L" + os.path.join(HERE, "lib"), "-lflash_hip"]
    return cflags, ldflags, tlib


def build_torch_targets(force=False):
    """libflash.a, libpyflash.so, runner."""
    build_hip_library()
    cflags, ldflags, _ = _torch_flags()
    os.makedirs(os.path.join(BUILD, "pyflash"), exist_ok=True)
    os.makedirs(os.path.join(BUILD, "obj"), exist_ok=True)
    hdrs = glob.glob(os.path.join(HERE, "flash", "*.hpp")) + [os.path.join(ROOT, "include", "flash_hip.h")]
    units = {
        "launch": os.path.join(HERE, "flash", "launch.cpp"),
        "pyflash": os.path.join(HERE, "pyflash", "pyflash.cpp"),
        "main": os.path.join(HERE, "main.cpp"),
    }
    objs = {k: os.path.join(BUILD, "obj", k + ".o") for k in units}

    def compile_unit(k):
        if force or _stale(objs[k], [units[k]] + hdrs):
            _run(["g++", "-c", units[k], "-o", objs[k]] + cflags)

    with concurrent.futures.ThreadPoolExecutor(max_workers=3) as ex:
        list(ex.map(compile_unit, units))

    liba = os.path.join(BUILD, "libflash.a")
    if force or _stale(liba, [objs["launch"]]):
        if os.path.exists(liba):
            os.remove(liba)
        _run(["ar", "rcs", liba, objs["launch"]])
    origin_lib = "-Wl,-rpath,$ORIGIN/../flash-attention_amd/lib"
    # libtorch_hip asks for "libamdhip64.so" and finds the copy torch bundles; libflash_hip.so and the
    # runner ask for the soname "libamdhip64.so.7".  Point that name at torch's copy too, so that a
    # process never holds two HIP runtimes (in Python, importing torch first has the same effect).
    # DT_RPATH (not DT_RUNPATH) so that it also wins over an LD_LIBRARY_PATH that names /opt/rocm/lib.
    rtlib = os.path.join(BUILD, "rtlib")
    os.makedirs(rtlib, exist_ok=True)
    link = os.path.join(rtlib, "libamdhip64.so.7")
    if os.path.lexists(link):
        os.remove(link)
    os.symlink(os.path.join(_torch_flags()[2], "libamdhip64.so"), link)
    runner = os.path.join(BUILD, "runner")
    if force or _stale(runner, [objs["main"], liba]):
        _run(["g++", objs["main"], liba, "-o", runner, "-Wl,--disable-new-dtags", "-Wl,-rpath,$ORIGIN/rtlib"] + ldflags + [origin_lib])
    pyflash = os.path.join(BUILD, "pyflash", "libpyflash.so")
    if force or _stale(pyflash, [objs["pyflash"], liba]):
        _run(["g++", "-shared", objs["pyflash"], liba, "-o", pyflash] + ldflags +
             ["-ltorch_python", "-Wl,-rpath,$ORIGIN/../../flash-attention_amd/lib"])
    return runner, pyflash


def build_all(force=False, with_torch=True):
    build_hip_library(force)
    build_oracle(force)
    build_selftest(force)
    if with_torch:
        build_torch_targets(force)


if __name__ == "__main__":
    if "asm" in sys.argv:
        build_device_asm(force="--force" in sys.argv)
    else:
        build_all(force="--force" in sys.argv, with_torch="--no-torch" not in sys.argv)
