# The two wave layouts of the d = 128 persistent kernel on a few shapes, interleaved (GPU).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2 3; do
  for shape in "8 96 4096 128 40" "4 96 1024 128 400" "4 96 512 128 800" "4 96 8192 128 12"; do
    for wide in 0 1000000; do
      FLASH_HIP_WIDE_MAX_N=$wide timeout -k 10 120 tests/native/selftest bench $shape 11 2>&1 | grep sustained | sed "s|^|[round $round wide_max_n $wide] |"
    done
  done
done | tee gpurun_out/layout_ab.log
