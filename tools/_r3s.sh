set -o pipefail
cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out/r3s; rm -f gpurun_out/r3s/*
timeout -k 10 600 python benchmarks/sweep.py --out gpurun_out/r3s/sweep.json > gpurun_out/r3s/sweep.log 2>&1 || { tail -20 gpurun_out/r3s/sweep.log; exit 1; }
grep -E "seq_len|mma_qreg_async|torch_flash" gpurun_out/r3s/sweep.log
timeout -k 10 400 python benchmarks/sweep.py --batch 16 --heads 32 --head-dim 64 --max-n 8192 --out gpurun_out/r3s/sweep_d64.json > gpurun_out/r3s/sweep_d64.log 2>&1 || { tail -20 gpurun_out/r3s/sweep_d64.log; exit 1; }
grep -E "seq_len|mma_qreg_async" gpurun_out/r3s/sweep_d64.log
