mkdir -p gpurun_out/prof && cd /root/repo
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/kt -o bench -- python3 bench.py --no-cpu-baseline > gpurun_out/prof/bench_prof.json 2> gpurun_out/prof/kt.err || { tail -30 gpurun_out/prof/kt.err; exit 1; }
cat gpurun_out/prof/bench_prof.json
find gpurun_out/prof/kt -type f | head
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -o p -- tests/native/selftest bench 8 96 4096 128 5 11 > gpurun_out/prof/fetch.log 2>&1 || { tail gpurun_out/prof/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -o p -- tests/native/selftest bench 8 96 4096 128 5 11 > gpurun_out/prof/write.log 2>&1 || { tail gpurun_out/prof/write.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/prof/sq -o p -- tests/native/selftest bench 8 96 4096 128 5 11 > gpurun_out/prof/sq.log 2>&1 || { tail gpurun_out/prof/sq.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA --output-format csv -d gpurun_out/prof/grbm -o p -- tests/native/selftest bench 8 96 4096 128 5 11 > gpurun_out/prof/grbm.log 2>&1 || { tail gpurun_out/prof/grbm.log; exit 1; }
find gpurun_out/prof -type f | head -40
