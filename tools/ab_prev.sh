# Interleaved A/B of two builds of libflash_hip.so through the bare C ABI:
#   gpurun_tmp/ko/prev/libflash_hip.so (an earlier build, copied there by hand) against the in-tree library.
for round in 1 2 3; do
for lib in prev cur; do
  if [ $lib = prev ]; then export LD_LIBRARY_PATH=gpurun_tmp/ko/prev; else unset LD_LIBRARY_PATH; fi
  for s in "4 96 512 128 800" "4 96 1024 128 400" "8 96 4096 128 40" "16 32 2048 64 300" "16 32 512 64 800"; do
    timeout -k 10 120 tests/native/selftest bench $s 11 2>&1 | grep sustained | sed "s|^|[$lib] |" || exit 1
  done
  for s in "4 96 512 128 200" "4 96 1024 128 100"; do
    timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep "flash_fwd_f32 " | sed "s|^|[$lib] |" || exit 1
  done
done
done
