"""Generate the golden vectors in tests/golden/*.npz.

The reference has no golden files; its tests recompute `manual_attn` with libtorch on every run
(reference main.cpp:12-17, 67-79).  This script runs that same expression with the local torch CPU
operators (matmul, softmax, matmul) on the reference's test inputs — torch.manual_seed(0), then
randn for q, k, v in that order (main.cpp:24-34) — for the reference runner's five shapes
(main.cpp:157-193) and the tiny plumbing configuration (2, 4, 256, 64), and stores

  q32/k32/v32   the fp32 inputs as the reference generates them
  q/k/v         the same rounded to fp16 (what the MI355X kernel consumes)
  o_f32         manual_attn(q32, k32, v32)            fp32 oracle on the reference's own inputs
  o             manual_attn(q.float(), k.float(), v.float())   fp32 oracle on the fp16-rounded inputs
  m, l          row max of the scaled scores and row sum of exp(score - m) for `o`

For the two multi-head shapes only a few (batch, head) slices are kept (heads are independent), so
every fixture stays small.  Run from the repository root:  python tests/golden/make_golden.py
"""
import math
import os

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))

# (B, H, N, d), heads kept
CASES = [
    ((1, 1, 50, 32), None),
    ((1, 1, 100, 32), None),
    ((1, 1, 64, 50), None),
    ((5, 12, 53, 69), [(0, 0), (2, 5), (4, 11)]),
    ((1, 1, 30, 128), None),
    ((2, 4, 256, 64), [(0, 0), (1, 3)]),
]


def manual_attn(q, k, v):
    att = q @ k.transpose(-2, -1) * (1.0 / math.sqrt(k.size(-1)))
    att = F.softmax(att, dim=-1)
    return att @ v


def main():
    torch.set_num_threads(1)
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

        out = dict(
            shape=np.array([B, H, N, d], dtype=np.int64),
            heads=np.array(keep, dtype=np.int64),
            q32=pick(q32), k32=pick(k32), v32=pick(v32),
            q=pick(q16), k=pick(k16), v=pick(v16),
            o_f32=pick(o_f32), o=pick(o), m=pick(m), l=pick(l),
        )
        path = os.path.join(HERE, f"attn_B{B}_H{H}_N{N}_d{d}.npz")
        np.savez_compressed(path, **out)
        print(path, os.path.getsize(path), "bytes; torch", torch.__version__)


if __name__ == "__main__":
    main()
