#!/bin/bash
# usage: tools/pmc2.sh <tag> <script>: kernel trace + two SQ counter passes (no TCC counters: they hung once)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc1 -- python3 "$@" > $OUT/pmc1.log 2>&1
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/pmc2 -- python3 "$@" > $OUT/pmc2.log 2>&1
python3 tools/pmc_summary.py $OUT
