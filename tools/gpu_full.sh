mkdir -p gpurun_out/full && cd /root/repo
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/full/pytest.log 2>&1 || { tail -30 gpurun_out/full/pytest.log; exit 1; }
tail -3 gpurun_out/full/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/full/smoke.log 2>&1 || { tail -30 gpurun_out/full/smoke.log; exit 1; }
tail -2 gpurun_out/full/smoke.log
timeout -k 10 400 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err || { tail -30 gpurun_out/full/bench.err; exit 1; }
cat gpurun_out/full/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/full/prof -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/full/bench_prof.json 2> gpurun_out/full/prof.err || { tail -30 gpurun_out/full/prof.err; exit 1; }
cat gpurun_out/full/bench_prof.json
find gpurun_out/full/prof -name "*stats*" | head
