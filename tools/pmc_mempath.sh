#!/bin/bash
# Memory-path counters (TA / TCP / TCC) of one kernel of a tool run: tools/pmc_mempath.sh "<kernel substring>" <python tool> [tag]
#   e.g. tools/pmc_mempath.sh "k_gemm_bf16_nt<2>" tools/bench_gemm.py gemm   ->  gpurun_out/pmc_mempath_<tag>/summary.json
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
MATCH="$1"; TOOL="$2"; TAG="${3:-run}"
OUT=$R/gpurun_out/pmc_mempath_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 $R/$TOOL > $OUT/a.log 2>&1 || echo "pass a failed"
timeout 200 rocprofv3 --kernel-trace --pmc TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ TCP_GATE_EN1 --output-format csv -d $OUT/b -o b -- python3 $R/$TOOL > $OUT/b.log 2>&1 || echo "pass b failed"
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT TCC_MISS TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL --output-format csv -d $OUT/c -o c -- python3 $R/$TOOL > $OUT/c.log 2>&1 || echo "pass c failed"
python3 - <<PY
import csv, glob, collections, json
cnt = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if """$MATCH""" in r["Kernel_Name"]:
            cnt[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in cnt.items()}
if out.get("TCP_TCC_READ_REQ"):
    out["avg_l1_to_l2_read_latency_cycles"] = out.get("TCP_TCC_READ_REQ_LATENCY", 0) / out["TCP_TCC_READ_REQ"]
if out.get("TCP_GATE_EN1"):
    out["tcp_pending_stall_frac"] = out.get("TCP_PENDING_STALL_CYCLES", 0) / out["TCP_GATE_EN1"]
if out.get("TCC_HIT") is not None and (out.get("TCC_HIT", 0) + out.get("TCC_MISS", 0)) > 0:
    out["l2_hit_rate"] = out["TCC_HIT"] / (out["TCC_HIT"] + out["TCC_MISS"])
if out.get("TCC_EA0_RDREQ"):
    out["avg_l2_to_dram_read_latency_cycles"] = out.get("TCC_EA0_RDREQ_LEVEL", 0) / out["TCC_EA0_RDREQ"]
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
