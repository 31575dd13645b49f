#!/usr/bin/env python
"""In-process A/B of the "ln_fold" option (LayerNorm folded into the encoder GEMMs) on one box: ms per 32-frame clip forward,
interleaved repeats, median. usage: fold_ab.py [vitl|vits]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
cfg = get_config(enc)
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
ts = {0: [], 1: []}
outs = {}
for rep in range(5):
    for opt in (0, 1):
        m.engine.set_option("ln_fold", opt)
        outs[opt] = m.forward(x, fp32=False).clone()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.forward(x, fp32=False)
        e1.record(); torch.cuda.synchronize()
        ts[opt].append(e0.elapsed_time(e1) / 5)
for opt in (0, 1):
    t = sorted(ts[opt])[2]
    print(f"{enc} ln_fold={opt}: {t:.3f} ms/clip ({32e3 / t:.1f} frames/s)  all: {[round(v, 2) for v in ts[opt]]}", flush=True)
d = (outs[0] - outs[1]).abs().mean() / outs[0].abs().mean()
print(f"rel-L1 fold vs standalone: {float(d):.3e}")
