# rocprofv3 passes for one workload:  bash tools/gpu_profile.sh <workload> "<B H N d>" [outdir]
#   kernel trace + stats of `bench.py --workload <workload>`, then one PMC pass per counter group over
#   `tests/native/selftest bench <B H N d> 5 11` (the same kernel through the C ABI, no torch).
# LIBDIR=<dir> profiles another build of libflash_hip.so (PMC passes only).
set -o pipefail
W=${1:-headline}; SHAPE=${2:-"8 96 4096 128"}; OUT=${3:-gpurun_out/prof/$W}
mkdir -p $OUT && cd /root/repo
export TMPDIR=/tmp
[ -n "$LIBDIR" ] && export LD_LIBRARY_PATH=$LIBDIR
if [ -z "$LIBDIR" ]; then
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 bench.py --workload $W --no-cpu-baseline --no-secondary > $OUT/bench_prof.json 2> $OUT/kt.err || { tail -30 $OUT/kt.err; exit 1; }
cat $OUT/bench_prof.json
fi
pass() {  # name, counters...
  local name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -o p -- tests/native/selftest bench $SHAPE 5 11 > $OUT/$name.log 2>&1 || { tail $OUT/$name.log; exit 1; }
}
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU &&
pass grbm GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_MFMA &&
find $OUT -name "*.csv" | head -20
