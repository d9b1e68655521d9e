# Times the knock-out libraries under gpurun_tmp/ko/<mask> on the three benchmark shapes, interleaved
# (mask by mask inside each round, three rounds), so that drift of the box averages out.
# usage: bash tools/ko_shapes.sh "<masks>"   (GPU; libraries built by `tools/ceiling.sh build` or by hand)
set -o pipefail
cd "$(dirname "$0")/.."
MASKS=${1:-"0 32"}
OUT=gpurun_out/ko_shapes
mkdir -p $OUT
for round in 1 2 3; do
  for shape in "8 96 4096 128 40" "4 96 1024 128 400" "16 32 2048 64 300"; do
    for m in $MASKS; do
      { LD_LIBRARY_PATH=gpurun_tmp/ko/$m timeout -k 10 120 tests/native/selftest bench $shape 11 2>&1 || true; } | grep sustained | sed "s|^|[round $round mask $m] |"
    done
  done
done | tee $OUT/ko_shapes.log
