#!/usr/bin/env python
"""gpurun_out/pmc_mfma/m_counter_collection.csv (tools/pmc_mfma.sh) -> profiles/<round>/vitl_pmc_mfma_lds.json: per kernel MFMA-busy
fraction (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)), LDS-active fraction and bank-conflict rate."""
import collections, csv, json, re, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc_mfma/m_counter_collection.csv"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01/vitl_pmc_mfma_lds.json"
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set); dur = collections.defaultdict(float)
for r in csv.DictReader(open(src)):
    n = r["Kernel_Name"]
    m = re.search(r"(gemm\w*_kernel<[^>]*>)", n)
    k = m.group(1) if m else n.split("(")[0].replace("void ", "").strip()
    if k.startswith("at::") or "rocclr" in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Dispatch_Id"] not in cnt[k]:
        cnt[k].add(r["Dispatch_Id"]); dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = {}
for k, v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    n = len(cnt[k]); cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
    out[k] = {"launches": n, "avg_us_under_profiler": dur[k] / n / 1e3,
              "mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024) if cyc else None,
              "lds_active_frac": v["SQ_LDS_IDX_ACTIVE"] / (cyc * 256) if cyc else None,
              "lds_bank_conflict_per_active_cycle": v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1)}
json.dump({"note": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA GRBM_GUI_ACTIVE "
                   "over `bench.py --steps 1 --warmup 1` (tools/pmc_mfma.sh); fractions are of the dispatch's GPU-active cycles (GRBM_GUI_ACTIVE / 8 XCDs), "
                   "whole kernel including epilogues; SQ_LDS_IDX_ACTIVE appears to count quad-cycles like the other SQ_ACTIVE_* counters (MI355X_MICROARCH.md): lds_active_frac x4 = 71-90 % for the GEMMs, which matches 256 KB of LDS traffic per K tile at 128 B/clk", "kernels": out}, open(dst, "w"), indent=1)
for k, v in list(out.items())[:12]:
    print(f"{k[:60]:60s} MFMA {v['mfma_busy_frac']*100:5.1f}%  LDS {v['lds_active_frac']*100:5.1f}%  conflicts {v['lds_bank_conflict_per_active_cycle']*100:5.2f}%")
