#!/bin/bash
# rocprofv3 kernel stats of a short default-config bench run (program directly after `--`)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_bench
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o t -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-search --no-sparse > $OUT/log.txt 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/t_kernel_trace.csv")))
import collections
d=collections.defaultdict(list)
for r in rows:
    n=r['Kernel_Name']
    if 'k_gemm' in n or 'k_attn' in n or 'k_rmsnorm' in n:
        d[n.split('(')[0][:40]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    big=[x for x in v if x>0.3*max(v)]
    print(k.ljust(42), len(v), "full launches", len(big), "avg %.1f us  min %.1f  max %.1f" % (sum(big)/len(big), min(big), max(big)))
PY
