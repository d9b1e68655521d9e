export TMPDIR=/tmp
mkdir -p gpurun_out/pmc1
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc1/a -o p -- tests/native/selftest bench 4 96 4096 128 3 11 > gpurun_out/pmc1/a.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then echo died; exit 1; fi
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d gpurun_out/pmc1/b -o p -- tests/native/selftest bench 4 96 4096 128 3 11 > gpurun_out/pmc1/b.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then echo died; exit 1; fi
tail -3 gpurun_out/pmc1/b.log
