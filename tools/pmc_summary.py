#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes per kernel: average bytes per launch.
FETCH_SIZE/WRITE_SIZE are in KiB-like units of 1024 B?  rocprofv3 reports FETCH_SIZE in KB (1024 B) per the counter
definition (TCC_EA0_RDREQ_32B*32 + ...)/1024; gfx950 correction: FETCH_SIZE under-reports wide coalesced reads by 2x."""
import csv, glob, json, os, sys, collections

def load(d, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows

import re

def short(name):
    # the tiled-shadow filter kernels: persistent emitting main pass / one-workgroup-per-block form (emit or score stores)
    if "k_filter_xreg_emit<" in name:
        return "k_filter_xreg<emit>"
    m = re.search(r"k_filter_xreg<(\d+), (\d+), (true|false)(, \d+)?>", name)
    if m:
        return "k_filter_xreg<%s>" % ("emit" if m.group(3) == "true" else "scores")
    m = re.search(r"k_flat_ip_scores_split<(\d+), (\d+)([^>]*)>", name)
    if m:
        # NP=1: filter pass (emit = the score-free main pass, scores = its sample pass / the score-matrix filter), NP=3: six-product
        # pass (gated fallback launches are dropped below)
        if m.group(2) == "1":
            return "k_flat_ip_scores_split<NP=1,%s>" % ("emit" if m.group(3).strip().endswith("true, true, true") or m.group(3).strip().endswith("false, false, true") else "scores")
        return "k_flat_ip_scores_split<NP=%s>" % m.group(2)
    for k in ("k_sample_threshold", "k_refine_band", "k_shard_rows", "k_shard_bounds", "k_pool_norm", "k_gemm_bf16_nt<6>", "k_gemm_bf16_nt<4>", "k_refine_topk", "k_refine_merge", "k_topk_select_rescore", "k_gemm_bf16_nt<3>", "k_gemm_bf16_nt<2>", "k_gemm_bf16_nt<1>", "k_gemm_bf16_nt<0>", "k_attn_resident64", "k_attn_varlen_causal", "k_attn_stream", "k_attn_build_items", "k_rmsnorm", "k_rope", "k_flat_ip_scores_split", "k_flat_ip_scores", "k_topk_select"):
        if k in name:
            return k
    return None

out = {}
base = sys.argv[1]
for counter, sub, corr in (("FETCH_SIZE", "fetch", 2.0), ("WRITE_SIZE", "write", 1.0)):
    for name, vals in load(os.path.join(base, sub), counter).items():
        s = short(name)
        if not s:
            continue
        vals = [v for v in vals if v >= 0.01 * max(vals)] if max(vals) > 0 else vals     # device-gated launches that returned at once
        e = out.setdefault(s, {})
        e[counter + "_KiB_avg_raw"] = sum(vals) / len(vals)
        e[counter + "_bytes_avg_corrected"] = sum(vals) / len(vals) * 1024 * corr
        e["launches_" + sub] = len(vals)
for e in out.values():
    e["hbm_bytes_per_launch"] = e.get("FETCH_SIZE_bytes_avg_corrected", 0) + e.get("WRITE_SIZE_bytes_avg_corrected", 0)
print(json.dumps(out, indent=1))
