export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/loop_trace; rm -rf $OUT; mkdir -p $OUT; cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-sparse > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_embedding_bag" in r["Kernel_Name"]]
for a, b in zip(idx[1:7], idx[2:8]):
    span = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[a:b]) / 1e3
    print("span %.1f us busy %.1f us kernels %d  embedding_bag %.1f us" % (span, busy, b - a, (int(rows[a]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3))
a, b = idx[3], idx[4]
prev = None
for r in rows[a:b + 1]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-50s dur %7.1f gap_before %6.1f" % (r["Kernel_Name"][:50], (en - st) / 1e3, (st - prev) / 1e3 if prev else 0)); prev = en
PY
