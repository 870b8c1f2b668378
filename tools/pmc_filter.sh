#!/bin/bash
# Memory-path counters of the search filter kernel (rocprofv3 PMC, kernel-trace only, own passes): texture-addresser busy / stalls, L1 (TCP)
# pending-request stalls and read latency, L2 (TCC) DRAM credit stalls.  tools/pmc_filter.sh  ->  gpurun_out/pmc_filter/summary.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_filter
rm -rf $OUT; mkdir -p $OUT
cd /tmp
QS=100 timeout 200 rocprofv3 --kernel-trace --pmc TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 $R/tools/bench_search.py > $OUT/a.log 2>&1 || echo "pass a failed"
QS=100 timeout 200 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_GATE_EN1 --output-format csv -d $OUT/b -o b -- python3 $R/tools/bench_search.py > $OUT/b.log 2>&1 || echo "pass b failed"
QS=100 timeout 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_BUSY --output-format csv -d $OUT/c -o c -- python3 $R/tools/bench_search.py > $OUT/c.log 2>&1 || echo "pass c failed"
python3 - <<PY
import csv, glob, collections, json
cnt = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "scores_split<7, 1" in r["Kernel_Name"]:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in cnt.items()}
if out.get("TCP_TCC_READ_REQ"):
    out["avg_read_latency_cycles"] = out.get("TCP_TCC_READ_REQ_LATENCY", 0) / out["TCP_TCC_READ_REQ"]
if out.get("TCC_EA0_RDREQ"):
    out["avg_outstanding_dram_reads_per_request_cycle"] = out.get("TCC_EA0_RDREQ_LEVEL", 0) / out["TCC_EA0_RDREQ"]
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
tail -2 $OUT/a.log
