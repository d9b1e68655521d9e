# the tests of the two-sweep scheme, the parity tests, then what a defeating input costs
mkdir -p gpurun_out/sparse && cd /root/repo
export TMPDIR=/tmp
timeout -k 10 600 python -u -m pytest tests/test_sparse_max_gpu.py tests/test_parity_gpu.py -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/sparse/pytest.log 2>&1 || { tail -30 gpurun_out/sparse/pytest.log; exit 1; }
tail -3 gpurun_out/sparse/pytest.log
timeout -k 10 300 python tools/overflow_cost.py > gpurun_out/sparse/overflow_cost.log 2>&1 || { tail -30 gpurun_out/sparse/overflow_cost.log; exit 1; }
cat gpurun_out/sparse/overflow_cost.log
