export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_attn_fetch; rm -rf $OUT; mkdir -p $OUT; cd /tmp
SHAPES=32-8-128 REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -o f -- python3 $R/tools/bench_attn.py > $OUT/f.log 2>&1
SHAPES=32-8-128 REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/g -o g -- python3 $R/tools/bench_attn.py > $OUT/g.log 2>&1
python3 - <<PY
import csv, glob, collections
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attn" in r["Kernel_Name"]: cnt[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in cnt.items():
    print(k, {x: round(sum(v) / len(v)) for x, v in c.items()})
PY
tail -3 $OUT/g.log
