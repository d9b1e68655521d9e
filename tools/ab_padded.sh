# fp32 tensors whose head dim is not a multiple of 8, through the bare C ABI: flash_fwd_f32 (element staging,
# one kernel) against flash_fwd_f32_ws (rows padded into the workspace, 16x16x32 kernels, output copied out).
# usage: bash tools/ab_padded.sh   (GPU)
set -o pipefail
cd "$(dirname "$0")/.."
for s in "4 96 4096 69 3" "4 96 4096 50 3" "4 96 1024 69 20" "4 96 1024 100 20" "16 32 2048 52 10" "4 96 512 100 40" "4 96 4000 69 3"; do
  timeout -k 10 120 tests/native/selftest benchf32 $s 2>&1 | grep "fp32 tensors" || exit 1
done
