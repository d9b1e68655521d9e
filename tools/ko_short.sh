# Library builds under gpurun_tmp/ko/<dir> side by side on short sequences (GPU), interleaved over three rounds.
set -o pipefail
cd "$(dirname "$0")/.."
MASKS=${1:-"0 128"}
mkdir -p gpurun_out
for round in 1 2 3; do
  for shape in "4 96 512 128 1500" "4 96 1024 128 500" "16 32 512 64 1500" "8 96 4096 128 40"; do
    for m in $MASKS; do
      { LD_LIBRARY_PATH=gpurun_tmp/ko/$m timeout -k 10 120 tests/native/selftest bench $shape 11 2>&1 || true; } | grep sustained | sed "s|^|[round $round $m] |"
    done
  done
done | tee gpurun_out/ko_short.log
