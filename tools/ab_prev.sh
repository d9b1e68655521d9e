for round in 1 2; do
for lib in prev cur; do
  if [ $lib = prev ]; then export LD_LIBRARY_PATH=gpurun_tmp/ko/prev; else unset LD_LIBRARY_PATH; fi
  for s in "4 96 512 128 200" "4 96 1024 128 100" "4 96 2048 128 40" "16 32 512 64 200"; do
    timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep "flash_fwd_f32 " | sed "s|^|[$lib] |"
  done
  for s in "8 96 4096 128 40" "4 96 1024 128 400" "16 32 2048 64 300"; do
    timeout -k 10 120 tests/native/selftest bench $s 11 2>&1 | grep sustained | sed "s|^|[$lib] |"
  done
done
done
