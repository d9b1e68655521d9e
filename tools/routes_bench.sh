# One throughput figure per route of the library through the bare C ABI (GPU).
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
for s in "4 96 512 128 800" "4 96 1024 128 400" "4 96 2048 128 120" "4 96 4096 128 40" "4 96 4000 128 40" \
         "16 32 512 64 800" "16 32 1024 64 400" "16 32 2048 64 300" "4 96 4096 69 10" "4 96 4096 50 10" "4 96 1024 69 40"; do
  timeout -k 10 120 tests/native/selftest bench $s 11 2>&1 | grep sustained
done
for s in "4 96 512 128 200" "4 96 1024 128 100" "4 96 2048 128 40" "4 96 4096 128 20" "4 96 1000 128 100" \
         "16 32 512 64 200" "16 32 1024 64 100" "16 32 2048 64 60"; do
  timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep -i "tflop"
  FLASH_HIP_VEC_MAX_N=0 timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep -i "tflop" | sed "s|^|[pass forced] |"
done
} | tee gpurun_out/routes_bench.log
