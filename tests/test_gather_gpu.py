"""Forward with the optional all-gather of O overlapped in head chunks, on the GPU through RCCL.

A one-GPU box can host a single nccl rank only (RCCL refuses two ranks on one device), so this
covers the device side of `forward_shard_gather` (kernels writing straight into the gathered
buffer, collectives queued behind them) with world size 1; the two-rank slab arithmetic and the
collective itself are covered on CPU with gloo (tests/test_abi_host.py).
"""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import importlib, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
fa = importlib.import_module("flash-attention_amd")
from oracle import oracle as orc
orc.build()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:" + sys.argv[2], rank=0, world_size=1, device_id=dev)
ATOL, RTOL = 2e-3, 1e-2
for (heads, N, d) in ((7, 512, 128), (5, 320, 64), (3, 100, 68)):
    g = torch.Generator(device=dev).manual_seed(heads)
    q, k, v = (torch.randn((heads, N, d), device=dev, generator=g).half() for _ in range(3))
    plain = fa.forward(q[None], k[None], v[None])[0]
    ref = orc.attn_naive(*(x[None].float().cpu().numpy() for x in (q, k, v)))[0]
    err = np.abs(plain.float().cpu().numpy() - ref)
    assert err.max() <= ATOL and (err <= ATOL + RTOL * np.abs(ref)).all(), err.max()
    for exchange in ("p2p", "all_gather"):       # one rank: no peers to exchange with / RCCL all-gather of one
        for chunks in (1, 2, 4, 16):
            got = fa.forward_shard_gather(q, k, v, heads, chunks=chunks, exchange=exchange)
            assert got.shape == plain.shape
            assert torch.equal(got, plain), (heads, N, d, exchange, chunks,
                                             (got.float() - plain.float()).abs().max().item())
    # preallocated output, no host sync inside the call
    buf = torch.empty_like(plain)
    got = fa.forward_shard_gather(q, k, v, heads, chunks=3, out=buf, sync=False)
    torch.cuda.synchronize(dev)
    assert got.data_ptr() == buf.data_ptr() and torch.equal(buf, plain)
dist.destroy_process_group()
print("gather ok")
'''


@pytest.mark.gpu
def test_chunked_gather_matches_plain_forward_on_gpu(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    script = tmp_path / "gather_worker.py"
    script.write_text(_WORKER)
    out = subprocess.run([sys.executable, str(script), ROOT, port], capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "gather ok" in out.stdout
