#!/usr/bin/env python
"""gpurun_out/pmc_attn/{a,b}_counter_collection.csv (tools/pmc_attn.sh) -> per attention kernel: share of wave-cycles parked / issue-stalled /
issuing, VALU and MFMA busy fractions, instructions per wave. Prints a table and writes JSON when given a destination."""
import collections, csv, glob, json, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_attn"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(src + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "").strip()
        if "attn" not in k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
out = {}
for k, v in agg.items():
    per = {c: v[c] / max(len(n[k][c]), 1) for c in v}                 # per dispatch
    wc = per.get("SQ_WAVE_CYCLES", 0)
    cyc = per.get("GRBM_GUI_ACTIVE", 0) / 8
    o = {"dispatches": len(n[k]["SQ_WAVE_CYCLES"]),
         "parked_frac (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": per.get("SQ_WAIT_ANY", 0) / wc if wc else None,
         "issue_stall_frac (SQ_WAIT_INST_ANY)": per.get("SQ_WAIT_INST_ANY", 0) / wc if wc else None,
         "issuing_frac (SQ_ACTIVE_INST_ANY)": per.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else None,
         "valu_frac_of_wave_cycles": per.get("SQ_ACTIVE_INST_VALU", 0) / wc if wc else None,
         "lds_frac_of_wave_cycles": per.get("SQ_ACTIVE_INST_LDS", 0) / wc if wc else None,
         "mfma_busy_frac_of_gpu_cycles": per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else None,
         "insts_valu": per.get("SQ_INSTS_VALU"), "insts_mfma": per.get("SQ_INSTS_MFMA"), "insts_lds": per.get("SQ_INSTS_LDS"),
         "insts_salu": per.get("SQ_INSTS_SALU"), "insts_trans": per.get("SQ_INSTS_VALU_TRANS"),
         "wait_inst_lds": per.get("SQ_WAIT_INST_LDS"), "lds_bank_conflict": per.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active": per.get("SQ_LDS_IDX_ACTIVE")}
    if o["insts_mfma"]:
        o["valu_per_mfma"] = (o["insts_valu"] or 0) / o["insts_mfma"]
    out[k] = o
    print(k)
    for kk, vv in o.items():
        print(f"    {kk:48s} {vv}")
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
