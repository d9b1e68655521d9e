export TMPDIR=/tmp
mkdir -p gpurun_out/abl2
for a in "$@"; do
  LD_LIBRARY_PATH=gpurun_tmp/abl$a:$LD_LIBRARY_PATH timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/abl2/a$a -o p -- tests/native/selftest bench 4 96 4096 128 3 11 > gpurun_out/abl2/a$a.log 2>&1
  rc=$?
  if [ $rc -ge 124 ]; then echo "pmc run $a died rc=$rc"; exit 1; fi
  test -f gpurun_out/abl2/a$a/p_counter_collection.csv || { echo "no csv for $a"; tail -5 gpurun_out/abl2/a$a.log; exit 1; }
  grep "bench" gpurun_out/abl2/a$a.log | sed "s/^/ABL $a /"
done
echo finished
