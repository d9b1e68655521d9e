set -o pipefail
cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out/r3b; rm -f gpurun_out/r3b/*
timeout -k 10 800 python -u -m pytest tests/test_bench_gpu.py tests/test_zz_no_alloc.py tests/test_pyflash_gpu.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3b/pytest.log 2>&1 || { tail -40 gpurun_out/r3b/pytest.log; exit 1; }
tail -3 gpurun_out/r3b/pytest.log
FLASH_FORWARD_SPIN=0 timeout -k 10 300 python benchmarks/sweep.py --max-n 1024 --out gpurun_out/r3b/sweep_park.json > gpurun_out/r3b/sweep_park.log 2>&1 || { tail gpurun_out/r3b/sweep_park.log; exit 1; }
timeout -k 10 300 python benchmarks/sweep.py --max-n 1024 --out gpurun_out/r3b/sweep_spin.json > gpurun_out/r3b/sweep_spin.log 2>&1 || { tail gpurun_out/r3b/sweep_spin.log; exit 1; }
grep -E "seq_len|mma_qreg_async" gpurun_out/r3b/sweep_park.log
echo ---- spin
grep -E "seq_len|mma_qreg_async" gpurun_out/r3b/sweep_spin.log
