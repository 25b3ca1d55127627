#!/bin/bash
# HBM-side traffic of the conv kernels: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slots), no trace flags.
# usage: tools/pmc_traffic.sh <tag>
TAG=${1:-traffic}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 tools/conv_prof.py > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 100 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 tools/conv_prof.py > $OUT/write.log 2>&1; echo "write rc=$?"
python3 - <<PY
import csv, glob, collections
for name in ("fetch", "write"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % name, recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            acc[(r["Kernel_Name"][:70], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print(f"{name:5s} {c:11s} n={len(v):3d} mean={sum(v)/len(v):12.1f} KB  {k}")
PY
