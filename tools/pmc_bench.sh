#!/bin/bash
# Counter evidence for the kernels the bench runs: one kernel-trace pass and four --pmc passes of the SAME command
# (separate runs, no trace flags beside --pmc, the program directly after --), then a per-kernel summary.
# usage: tools/pmc_bench.sh <tag> [bench args]
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-extras $@"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1; echo "trace rc=$?"
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmc1 -- python3 $ARGS > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES SQ_INSTS_VALU --output-format csv -d $OUT/pmc2 -- python3 $ARGS > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 $ARGS > $OUT/pmc3.log 2>&1; echo "pmc3 rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 $ARGS > $OUT/pmc4.log 2>&1; echo "pmc4 rc=$?"
python3 tools/pmc_report.py $OUT > $OUT/summary.txt; cat $OUT/summary.txt
