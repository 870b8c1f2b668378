#!/bin/bash
# HBM traffic of the dominant kernels via rocprofv3 PMC counters, collected exactly as MI355X_MICROARCH.md prescribes:
# separate passes for FETCH_SIZE and WRITE_SIZE, --kernel-trace only, program directly after `--`.
# FETCH_SIZE on gfx950 reports 1/2 of the bytes of wide coalesced streaming reads -> doubled in the summary script.
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_r02
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -o write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/write.log 2>&1 || echo "write pass failed"
find $OUT -name "*.csv" | head
python3 $R/tools/pmc_summary.py $OUT > $OUT/summary.json && cat $OUT/summary.json
