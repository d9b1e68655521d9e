# Library builds under gpurun_tmp/ko/<dir> side by side on fp32 tensors (GPU), two rounds.
set -o pipefail
cd "$(dirname "$0")/.."
for round in 1 2; do
  for s in "4 96 512 128 200" "4 96 1024 128 100" "4 96 2048 128 40" "16 32 512 64 200"; do
    for m in ${1:-"cur 256"}; do
      { LD_LIBRARY_PATH=gpurun_tmp/ko/$m timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 || true; } | grep "flash_fwd_f32 " | sed "s|^|[$m] |"
    done
  done
done | tee gpurun_out/ko_f32.log
