#!/bin/bash
# SQ counters of the attention kernels (tools/bench_attn.py, SHAPES selects the head layout), two passes (PMC slot limits)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_attn${PMC_TAG:-}
rm -rf $OUT; mkdir -p $OUT
cd /tmp
SHAPES=${SHAPES:-32-8-128} REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o a -- python3 $R/tools/bench_attn.py > $OUT/a.log 2>&1 || echo "pass a failed"
SHAPES=${SHAPES:-32-8-128} REPS=3 timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o b -- python3 $R/tools/bench_attn.py > $OUT/b.log 2>&1 || echo "pass b failed"
python3 - <<PY
import csv, glob, collections, json
cnt = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for f in glob.glob("$OUT/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_attn" in n: cnt[n[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob("$OUT/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_attn" in r["Kernel_Name"]: dur[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
for k, c in cnt.items():
    a = {x: sum(v) / len(v) for x, v in c.items()}
    print(k, "avg us", round(sum(dur[k]) / max(1, len(dur[k])), 1))
    print("  ", {x: int(v) for x, v in sorted(a.items())})
    wc = a.get("SQ_WAVE_CYCLES", 0)
    if wc:
        print("   per wave-cycle: mfma busy %.3f  valu active %.3f  lds active %.3f  wait_any %.3f  wait_inst_any %.3f  wait_lds %.3f" % tuple(
            a.get(x, 0) / wc for x in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS")))
        print("   insts per wave: mfma %.0f valu %.0f lds %.0f ; wave cycles per wave %.0f" % (a.get("SQ_INSTS_MFMA", 0) / a["SQ_WAVES"], a.get("SQ_INSTS_VALU", 0) / a["SQ_WAVES"], a.get("SQ_INSTS_LDS", 0) / a["SQ_WAVES"], wc / a["SQ_WAVES"]))
PY
