
for a in 512; do
  echo "== ABL $a"
  LD_LIBRARY_PATH=gpurun_tmp/abl$a:$LD_LIBRARY_PATH timeout -k 10 60 tests/native/selftest bench 4 96 4096 128 20 11
done
echo "== base"
timeout -k 10 60 tests/native/selftest bench 4 96 4096 128 20 11
LD_LIBRARY_PATH=gpurun_tmp/abl512:$LD_LIBRARY_PATH timeout -k 10 60 tests/native/selftest bench 16 32 2048 64 20 11
LD_LIBRARY_PATH=gpurun_tmp/abl512:$LD_LIBRARY_PATH timeout -k 10 120 tests/native/selftest > gpurun_out/st512.log 2>&1; tail -1 gpurun_out/st512.log
