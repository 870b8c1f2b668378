#!/bin/bash
# per-kernel times of the search path (rocprofv3 kernel trace of tools/bench_search.py)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_search
mkdir -p $OUT
cd /tmp
QS=${QS:-100} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $R/tools/bench_search.py > $OUT/log.txt 2>&1
cat $OUT/log.txt | tail -3
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/s_kernel_stats.csv")))
for r in rows[:14]:
    if 'at::' in r['Name'] or 'rocclr' in r['Name']: continue
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), "avg %.1f us" % (float(r['AverageNs'])/1e3), "min %.1f" % (float(r['MinNs'])/1e3))
PY
