#!/usr/bin/env python
"""gpurun_out/pmc_tail/*counter_collection.csv (tools/pmc_tail.sh) -> per kernel: every counter per dispatch and a few ratios."""
import collections, csv, glob, json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_tail"
pat = sys.argv[3] if len(sys.argv) > 3 else "tail"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").strip()
        if pat not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {}
for k, v in agg.items():
    per = {c: v[c] / max(len(n[k][c]), 1) for c in v}
    wc, cyc = per.get("SQ_WAVE_CYCLES", 0), per.get("GRBM_GUI_ACTIVE", 0) / 8
    o = dict(per)
    if wc:
        o["parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)"] = per.get("SQ_WAIT_ANY", 0) / wc
        o["issue_stall_frac (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)"] = per.get("SQ_WAIT_INST_ANY", 0) / wc
        o["issuing_frac (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)"] = per.get("SQ_ACTIVE_INST_ANY", 0) / wc
    if cyc:
        o["gpu_cycles_per_dispatch"] = cyc
        o["mfma_busy_frac"] = per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024)
        o["ta_busy_frac (TA_TA_BUSY_sum / (cycles x 256 TAs))"] = per.get("TA_TA_BUSY_sum", 0) / (cyc * 256)
        o["ta_addr_stalled_by_tc_frac"] = per.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0) / (cyc * 256)
        o["tcp_pending_stall_frac"] = per.get("TCP_PENDING_STALL_CYCLES_sum", 0) / (cyc * 256)
    if per.get("TCP_TOTAL_CACHE_ACCESSES_sum"):
        o["l1_miss_frac (TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)"] = per.get("TCP_TCC_READ_REQ_sum", 0) / per["TCP_TOTAL_CACHE_ACCESSES_sum"]
    if per.get("TCP_TCC_READ_REQ_sum"):
        o["l2_read_latency_cycles (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ)"] = per.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / per["TCP_TCC_READ_REQ_sum"]
    out[k] = o
    print(k)
    for kk, vv in sorted(o.items()):
        print(f"    {kk:72s} {vv:.4g}" if isinstance(vv, float) else f"    {kk:72s} {vv}")
if len(sys.argv) > 2 and sys.argv[2] != "-":
    json.dump(out, open(sys.argv[2], "w"), indent=1)
