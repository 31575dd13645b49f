#!/usr/bin/env python
"""Whole forward under a vda_set_option switch, one process, interleaved: option_ab2.py [vitl|vits] option v0,v1"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
enc, opt = sys.argv[1], sys.argv[2]
vals = [int(v) for v in sys.argv[3].split(",")]
cfg = get_config(enc)
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
ts = {v: [] for v in vals}; outs = {}
side = torch.cuda.Stream() if os.environ.get("AB_STREAM") == "1" else None          # AB_STREAM=1: a non-default stream, as infer_video_depth's lanes
if side is not None:
    torch.cuda.set_stream(side)
for rep in range(7):
    for v in vals:
        m.engine.set_option(opt, v)
        d = m.forward(x, fp32=False)
        if rep == 0: outs[v] = d.clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): m.forward(x, fp32=False)
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 10)
for v in vals:
    t = sorted(ts[v])[len(ts[v]) // 2]
    print(f"{enc} {opt}={v}: {t:.3f} ms/clip ({32e3 / t:.1f} frames/s)  all: {[round(u, 2) for u in ts[v]]}", flush=True)
a, b = outs[vals[0]], outs[vals[-1]]
print(f"rel-L1 between the two: {float((a - b).abs().mean() / a.abs().mean()):.3e}")
