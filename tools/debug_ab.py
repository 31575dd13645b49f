#!/usr/bin/env python
"""Whole forward under different values of one library-wide switch, one process, interleaved, results asserted bit-identical.
usage: debug_ab.py [vitl|vits] valueA,valueB,... [setter]     setter: vda_gemm_set_debug (default: diagnostic switches of the 8-phase
kernel), vda_depth_tail_set_variant (0 persistent / 1 round-1 kernel), vda_conv_up_set_variant, ..."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
flags = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,2").split(",")]
setter = getattr(_lib.lib, sys.argv[3] if len(sys.argv) > 3 else "vda_gemm_set_debug")
cfg = get_config(enc)
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
ts = {v: [] for v in flags}
ref = None
for rep in range(7):
    for v in flags:
        setter(v)
        d = m.forward(x, fp32=False)
        if rep == 0:
            ref = d.clone() if ref is None else ref
            assert torch.equal(d, ref), f"debug flags {v} change the result"
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            m.forward(x, fp32=False)
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 5)
setter(0)
for v in flags:
    t = sorted(ts[v])[len(ts[v]) // 2]
    print(f"{enc} {setter.__name__}({v}): {t:.3f} ms/clip ({32e3 / t:.1f} frames/s)  all: {[round(u, 2) for u in ts[v]]}", flush=True)
