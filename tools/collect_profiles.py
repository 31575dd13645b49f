#!/usr/bin/env python
"""gpurun_out/{prof_<enc>_<tag>, pmc_bench_<enc>, pmc_mfma_<enc>, bench_<enc>_<btag>.json} -> profiles/<round>/ (what is committed).
usage: collect_profiles.py <round dir> <prof tag> <bench tag>      e.g.  collect_profiles.py profiles/r02 f r02c"""
import json, os, shutil, subprocess, sys
dst, ptag, btag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)
here = os.path.dirname(os.path.abspath(__file__))
for enc in ("vitl", "vits"):
    subprocess.check_call([sys.executable, os.path.join(here, "pmc_summarize.py"), f"gpurun_out/pmc_bench_{enc}", f"{dst}/{enc}_pmc_hbm_traffic.json"])
    subprocess.check_call([sys.executable, os.path.join(here, "pmc_mfma_summarize.py"), f"gpurun_out/pmc_mfma_{enc}/m_counter_collection.csv",
                           f"{dst}/{enc}_pmc_mfma_lds.json"], stdout=subprocess.DEVNULL)
    shutil.copy(f"gpurun_out/prof_{enc}_{ptag}/kernel_stats.csv", f"{dst}/{enc}_kernel_stats.csv")
    for src, name in ((f"gpurun_out/prof_{enc}_{ptag}/bench.json", "bench_under_rocprof"), (f"gpurun_out/bench_{enc}_{btag}.json", "bench_n1")):
        line = [x for x in open(src) if x.startswith("{")][-1]
        json.dump(json.loads(line), open(f"{dst}/{enc}_{name}.json", "w"), indent=1)
    d = json.load(open(f"{dst}/{enc}_bench_n1.json"))
    u = json.load(open(f"{dst}/{enc}_bench_under_rocprof.json"))
    print(enc, "value", round(d["value"], 1), "ms", round(d["ms_per_step"], 2), "dominant", d["roofline"]["kernel"], "TF", round(d["roofline"]["achieved"], 1),
          "frac", round(d["roofline"]["frac"], 3), "live us", round(d["roofline"]["avg_launch_us"], 1), "live us under rocprof", round(u["roofline"]["avg_launch_us"], 1),
          "two in flight", round(d["two_clips_in_flight"]["value"], 1), "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "model TF", round(d["model_tflops"], 1))
