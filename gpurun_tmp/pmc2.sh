export TMPDIR=/tmp
mkdir -p gpurun_out/pmc2
LD_LIBRARY_PATH=gpurun_tmp/w8:$LD_LIBRARY_PATH timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc2/a -o p -- tests/native/selftest bench 4 96 4096 128 3 11 > gpurun_out/pmc2/a.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then echo died; exit 1; fi
LD_LIBRARY_PATH=gpurun_tmp/w8:$LD_LIBRARY_PATH timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS --output-format csv -d gpurun_out/pmc2/b -o p -- tests/native/selftest bench 4 96 4096 128 3 11 > gpurun_out/pmc2/b.log 2>&1
rc=$?; if [ $rc -ge 124 ]; then echo died; exit 1; fi
tail -3 gpurun_out/pmc2/b.log
