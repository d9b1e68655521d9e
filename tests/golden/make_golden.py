"""Generate (or check) the golden vectors in tests/golden/*.npz by running the reference's own oracle.

The reference has no golden files; its tests recompute `manual_attn` with torch on every run
(reference main.cpp:12-17, 67-79; benchmarks/bench.py:39-43).  This script does not restate that
function: it reads /root/reference/benchmarks/bench.py as text, takes the `manual_attn` function
definition out of its syntax tree (the module cannot be imported whole: line 14 loads a library that
is never built here and line 34 needs a GPU), compiles exactly that definition and executes it with
the names it uses bound to the real local modules (`math`, `torch`, `torch.nn.functional` as `F`,
bound as the reference's lines 1-3 bind them).  Inputs are the reference's test inputs —
torch.manual_seed(0), then randn for q, k, v in that order (main.cpp:24-34) — for the reference
runner's five shapes (main.cpp:157-193) and the tiny plumbing configuration (2, 4, 256, 64).

Stored per fixture
  q32/k32/v32   the fp32 inputs as the reference generates them
  q/k/v         the same rounded to fp16 (what the MI355X kernel consumes)
  o_f32         reference manual_attn(q32, k32, v32)                    on the reference's own inputs
  o             reference manual_attn(q.float(), k.float(), v.float())  on the fp16-rounded inputs
  m, l          row max of the scaled scores and row sum of exp(score - m) for `o` (torch ops; these
                two pin the online-softmax state of the tiled restatement, they are not produced by
                reference code)
  ref_lines     first and last source line of the executed definition in the reference file

For the two multi-head shapes only a few (batch, head) slices are kept (heads are independent), so
every fixture stays small.  Nothing of the reference's text is written into a fixture or this
repository; only numbers are.  Needs /root/reference, i.e. the build container:

    python tests/golden/make_golden.py           # rewrite the fixtures
    python tests/golden/make_golden.py --check   # regenerate in memory and compare with the files

--check requires the inputs (randn draws, their fp16 roundings) and the bookkeeping arrays bit for bit.
The outputs of the reference function and the row statistics are fp32 results of torch's CPU matmul and
softmax, whose summation order follows the vector width of the host: on the machine that wrote the
fixtures they reproduce bit for bit, another CPU model may differ in the last places (seen: 3.6e-7 on
outputs of magnitude 1).  They must agree within OUT_ATOL + OUT_RTOL * |value|, a few fp32 ulps, four
orders of magnitude below the parity bar the fixtures are used for.
"""
import argparse
import ast
import hashlib
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_BENCH = "/root/reference/benchmarks/bench.py"
# sha256 of ast.dump() of the reference's manual_attn definition (benchmarks/bench.py:39-43)
MANUAL_ATTN_AST_SHA256 = "e60a531a6ae571cfec1cafba1f249a3207275a44d66ab87116fda47b1bc3a73b"

# arrays computed in floating point by torch on the host, and how far another host may be off
FLOAT_RESULTS = ("o_f32", "o", "m", "l")
OUT_ATOL, OUT_RTOL = 1e-6, 1e-6

# (B, H, N, d), heads kept
CASES = [
    ((1, 1, 50, 32), None),
    ((1, 1, 100, 32), None),
    ((1, 1, 64, 50), None),
    ((5, 12, 53, 69), [(0, 0), (2, 5), (4, 11)]),
    ((1, 1, 30, 128), None),
    ((2, 4, 256, 64), [(0, 0), (1, 3)]),
]


def reference_available():
    return os.path.isfile(REFERENCE_BENCH)


def load_reference_manual_attn(path=REFERENCE_BENCH):
    """Return (function, (first_line, last_line)): the reference's `manual_attn`, compiled from the
    reference file's own text and nothing else."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    defs = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "manual_attn"]
    if len(defs) != 1:
        raise RuntimeError(f"expected one top-level manual_attn in {path}, found {len(defs)}")
    node = defs[0]
    # The text is executed, so it is pinned first: its syntax tree must be the one this script was written
    # against (digest below), and may hold nothing but arithmetic, attribute access and calls.  Anything
    # else in those lines is refused before it runs, and the function gets no builtins.
    digest = hashlib.sha256(ast.dump(node).encode()).hexdigest()
    if digest != MANUAL_ATTN_AST_SHA256:
        raise RuntimeError(f"{path}: manual_attn is not the definition the fixtures were made from "
                           f"(syntax-tree digest {digest}); review it and update MANUAL_ATTN_AST_SHA256")
    allowed = (ast.FunctionDef, ast.arguments, ast.arg, ast.Assign, ast.Return, ast.BinOp, ast.UnaryOp, ast.Call,
               ast.keyword, ast.Attribute, ast.Name, ast.Constant, ast.Load, ast.Store, ast.MatMult, ast.Mult,
               ast.Div, ast.USub)
    for sub in ast.walk(node):
        if not isinstance(sub, allowed):
            raise RuntimeError(f"{path}: unexpected {type(sub).__name__} inside manual_attn")
    module = ast.Module(body=[node], type_ignores=[])
    env = {"__builtins__": {}, "math": math, "torch": torch, "F": F}
    exec(compile(module, path, "exec"), env)
    return env["manual_attn"], (node.lineno, node.end_lineno)


def generate():
    """{file name: {array name: ndarray}} for every case."""
    manual_attn, lines = load_reference_manual_attn()
    torch.set_num_threads(1)
    out = {}
    for (B, H, N, d), keep in CASES:
        torch.manual_seed(0)
        q32 = torch.randn(B, H, N, d)
        k32 = torch.randn(B, H, N, d)
        v32 = torch.randn(B, H, N, d)
        q16, k16, v16 = q32.half(), k32.half(), v32.half()
        o_f32 = manual_attn(q32, k32, v32)
        qf, kf, vf = q16.float(), k16.float(), v16.float()
        o = manual_attn(qf, kf, vf)
        s = qf @ kf.transpose(-2, -1) * (1.0 / math.sqrt(d))
        m = s.max(dim=-1).values
        l = torch.exp(s - m.unsqueeze(-1)).sum(dim=-1)
        if keep is None:
            keep = [(b, h) for b in range(B) for h in range(H)]

        def pick(t):
            return np.stack([t[b, h].numpy() for (b, h) in keep])[None]  # [1, n_kept, ...]

        out[f"attn_B{B}_H{H}_N{N}_d{d}.npz"] = dict(
            shape=np.array([B, H, N, d], dtype=np.int64),
            heads=np.array(keep, dtype=np.int64),
            q32=pick(q32), k32=pick(k32), v32=pick(v32),
            q=pick(q16), k=pick(k16), v=pick(v16),
            o_f32=pick(o_f32), o=pick(o), m=pick(m), l=pick(l),
            ref_lines=np.array(lines, dtype=np.int64),
        )
    return out


def check(notes=None):
    """Regenerate every fixture in memory and compare with the committed files: inputs and bookkeeping bit
    for bit, floating-point results within a few ulps (see the module docstring).  Returns a list of
    differences (empty = the fixtures are what the reference function produces); `notes`, if given,
    collects the arrays that agree without being bit-identical."""
    bad = []
    for name, arrays in generate().items():
        path = os.path.join(HERE, name)
        if not os.path.exists(path):
            bad.append(f"{name}: missing")
            continue
        with np.load(path) as have:
            if sorted(have.files) != sorted(arrays):
                bad.append(f"{name}: arrays {sorted(have.files)} != {sorted(arrays)}")
                continue
            for key, want in arrays.items():
                got = have[key]
                if got.dtype != want.dtype or got.shape != want.shape:
                    bad.append(f"{name}[{key}] differs in type or shape")
                elif got.tobytes() == want.tobytes():
                    continue
                elif key in FLOAT_RESULTS and np.all(
                        np.abs(got.astype(np.float64) - want.astype(np.float64))
                        <= OUT_ATOL + OUT_RTOL * np.abs(want.astype(np.float64))):
                    if notes is not None:
                        off = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
                        notes.append(f"{name}[{key}] agrees to {off:.1e} (host-dependent last places)")
                else:
                    bad.append(f"{name}[{key}] differs")
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    if not reference_available():
        sys.exit(f"{REFERENCE_BENCH} not found: fixtures are generated in the build container only")
    if args.check:
        notes = []
        bad = check(notes)
        for b in bad:
            print("MISMATCH", b)
        for n in notes:
            print("note:", n)
        if bad:
            print("golden fixtures: DIFFER")
        elif notes:
            print("golden fixtures: inputs bit-identical, results of the reference's manual_attn run here within a few ulps")
        else:
            print("golden fixtures: bit-identical to the reference's manual_attn run here")
        sys.exit(1 if bad else 0)
    for name, arrays in generate().items():
        path = os.path.join(HERE, name)
        np.savez_compressed(path, **arrays)
        print(path, os.path.getsize(path), "bytes; torch", torch.__version__,
              "reference lines", tuple(arrays["ref_lines"]))


if __name__ == "__main__":
    main()
