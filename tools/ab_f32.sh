# Interleaved A/B of the fp32 routes: gpurun_tmp/ko/prev/libflash_hip.so against the in-tree library.
for round in 1 2 3; do
for lib in prev cur; do
  if [ $lib = prev ]; then export LD_LIBRARY_PATH=gpurun_tmp/ko/prev; else unset LD_LIBRARY_PATH; fi
  for s in "4 96 512 128 300" "4 96 1024 128 150" "4 96 2048 128 60" "16 32 512 64 300" "16 32 1024 64 150" "16 32 2048 64 60"; do
    timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep "flash_fwd_f32 " | sed "s|^|[$lib] |" || exit 1
  done
done
done
