#!/bin/bash
# usage: tools/pmc.sh <tag> <script> : kernel trace + two PMC passes (separate runs, as the guide prescribes)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc1 -- python3 "$@" > $OUT/pmc1.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/pmc2 -- python3 "$@" > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc3 -- python3 "$@" > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 "$@" > $OUT/pmc4.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
for i in 1 2 3 4; do find $OUT/pmc$i -name "*counter_collection.csv" -exec cp {} $OUT/pmc$i.csv \; ; done
ls -la $OUT; tail -3 $OUT/pmc1.log $OUT/pmc2.log
