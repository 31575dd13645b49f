#!/usr/bin/env python
"""In-process A/B of a launch-sequence option (vda_set_option) inside the whole forward on one box: ms per 32-frame clip,
interleaved repeats, median. usage: option_ab.py [vitl|vits] [option, default oc1_fused] [values, default 0,1]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
opt = sys.argv[2] if len(sys.argv) > 2 else "oc1_fused"
values = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "0,1").split(",")]
cfg = get_config(enc)
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
ts = {v: [] for v in values}
for rep in range(5):
    for v in values:
        m.engine.set_option(opt, v)
        m.forward(x, fp32=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.forward(x, fp32=False)
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 5)
for v in values:
    t = sorted(ts[v])[2]
    print(f"{enc} {opt}={v}: {t:.3f} ms/clip ({32e3 / t:.1f} frames/s)  all: {[round(u, 2) for u in ts[v]]}", flush=True)
