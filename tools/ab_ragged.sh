# Interleaved A/B on ragged sequence lengths (one work item per workgroup): gpurun_tmp/ko/prev against the
# in-tree library.
for round in 1 2 3; do
for lib in prev cur; do
  if [ $lib = prev ]; then export LD_LIBRARY_PATH=gpurun_tmp/ko/prev; else unset LD_LIBRARY_PATH; fi
  for s in "4 96 4000 128 40" "4 96 1000 128 300" "4 96 500 128 600" "16 32 2000 64 200" "4 96 8100 128 10"; do
    timeout -k 10 120 tests/native/selftest bench $s 11 2>&1 | grep sustained | sed "s|^|[$lib] |" || exit 1
  done
  for s in "4 96 1000 128 100"; do
    timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep "flash_fwd_f32 " | sed "s|^|[$lib] |" || exit 1
  done
done
done
