set -o pipefail
cd /root/repo
export TMPDIR=/tmp; mkdir -p gpurun_out/r3c; rm -f gpurun_out/r3c/*
timeout -k 10 300 tests/native/selftest > gpurun_out/r3c/selftest.log 2>&1 || { grep FAIL gpurun_out/r3c/selftest.log | head; exit 1; }
tail -1 gpurun_out/r3c/selftest.log
for i in 1 2 3; do
for lib in base new; do
  if [ $lib = new ]; then L=flash-attention_amd/lib; else L=gpurun_tmp/$lib; fi
  for shape in "4 96 512 128" "4 96 1024 128" "16 32 512 64" "16 32 2048 64"; do
  LD_LIBRARY_PATH=$L timeout -k 10 120 tests/native/selftest bench $shape 400 11 2>&1 | grep sustained | sed "s|^|[$lib] |"
  done
done
done | tee gpurun_out/r3c/ab.log
