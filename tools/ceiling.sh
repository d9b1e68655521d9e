# Evidence for what bounds the headline kernel.  Two halves:
#   bash tools/ceiling.sh build     (CPU) diagnostic libraries with one kind of per-step work left out
#                                   (-DFLASH_KNOCKOUT=<mask>, see attn_fwd_mfma16_body.hpp) and the microbenchmarks
#   bash tools/ceiling.sh run       (GPU) times them on the headline shape, counts the cycles the kernel
#                                   was resident (GRBM_GUI_ACTIVE) to get the clock the chip held, compares
#                                   the two wave layouts, runs the microbenchmarks; output in gpurun_out/ceiling
# tools/summarize_ceiling.py condenses the output into profiles/r2_ceiling.json.
set -o pipefail
cd "$(dirname "$0")/.."
# every loop below runs on the left of `| tee`: its exit status is PIPESTATUS[0], checked after each one, so
# that a failed or crashed pass stops the script instead of leaving a partial run to be summarized
check() { [ "$1" -eq 0 ] || { echo "ceiling.sh: $2 failed" >&2; exit 1; }; }
MASKS="0 1 2 4 8 16 32 64 127"
SHAPE="8 96 4096 128"
if [ "$1" = build ]; then
  mkdir -p gpurun_tmp/ko gpurun_tmp/mb
  for m in $MASKS; do
    mkdir -p gpurun_tmp/ko/$m
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -fPIC -shared -DFLASH_KNOCKOUT=$m \
      flash-attention_amd/csrc/flash_hip.hip -o gpurun_tmp/ko/$m/libflash_hip.so 2>/dev/null &
    while [ "$(jobs -r | wc -l)" -ge 4 ]; do sleep 1; done
  done
  for f in mfma_clock issue_cost lds_issue_cost softmax_issue_cost; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/microbench/$f.cpp -o gpurun_tmp/mb/$f 2>/dev/null &
  done
  wait
  ls -la gpurun_tmp/ko/*/libflash_hip.so gpurun_tmp/mb
  exit 0
fi
OUT=gpurun_out/ceiling
mkdir -p $OUT
export TMPDIR=/tmp
for m in $MASKS; do
  # (the harness's own parity check fails on purpose here: only the timing line is kept; a crash leaves no such line)
  { LD_LIBRARY_PATH=gpurun_tmp/ko/$m timeout -k 10 120 tests/native/selftest bench $SHAPE 40 11 2>&1 || true; } | grep sustained | sed "s|^|[knockout $m] |" || exit 1
  LD_LIBRARY_PATH=gpurun_tmp/ko/$m timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/ko$m -o p -- tests/native/selftest bench $SHAPE 5 11 > $OUT/ko$m.log 2>&1
  grep -q sustained $OUT/ko$m.log || { tail $OUT/ko$m.log >&2; exit 1; }
done | tee $OUT/knockout.log
check ${PIPESTATUS[0]} "a knock-out pass"
# the LDS-fragment-read knock-out in the FOUR-wave layout as well (half as many LDS instructions per launch):
# what the reads cost there, next to its own baseline
for m in 0 4; do
  { FLASH_HIP_WIDE_MAX_N=0 LD_LIBRARY_PATH=gpurun_tmp/ko/$m timeout -k 10 120 tests/native/selftest bench $SHAPE 40 11 2>&1 || true; } | grep sustained | sed "s|^|[knockout4w $m] |" || exit 1
  FLASH_HIP_WIDE_MAX_N=0 LD_LIBRARY_PATH=gpurun_tmp/ko/$m timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/ko4w$m -o p -- tests/native/selftest bench $SHAPE 5 11 > $OUT/ko4w$m.log 2>&1
  grep -q sustained $OUT/ko4w$m.log || { tail $OUT/ko4w$m.log >&2; exit 1; }
done | tee $OUT/knockout4w.log
check ${PIPESTATUS[0]} "a four-wave knock-out pass"
# the two wave layouts of the same kernel on the headline shape (FLASH_HIP_WIDE_MAX_N routes any N to the eight-wave one)
for wide in 2048 1000000; do
  FLASH_HIP_WIDE_MAX_N=$wide timeout -k 10 120 tests/native/selftest bench $SHAPE 40 11 2>&1 | grep sustained | sed "s|^|[wide_max_n $wide] |" || exit 1
  FLASH_HIP_WIDE_MAX_N=$wide timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/wide$wide -o p -- tests/native/selftest bench $SHAPE 5 11 > $OUT/wide$wide.log 2>&1 || { tail $OUT/wide$wide.log; exit 1; }
done | tee $OUT/layouts.log
check ${PIPESTATUS[0]} "a wave-layout pass"
for f in mfma_clock issue_cost lds_issue_cost softmax_issue_cost; do
  echo "== $f"
  timeout -k 10 200 gpurun_tmp/mb/$f 2>&1 || exit 1
done | tee $OUT/microbench.log
check ${PIPESTATUS[0]} "a microbenchmark"
