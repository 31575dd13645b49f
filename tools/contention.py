#!/usr/bin/env python
"""How much does a communication kernel running beside the forward cost? The GEMMs are persistent, one 160-KiB-LDS workgroup per
CU: a workgroup of another kernel that holds any LDS on a CU keeps the GEMM's workgroup for that CU waiting. Emulation on one
GPU: vda_debug_occupy (N workgroups of 256 threads with 16 KiB of LDS, ~D ms) on a side stream once per forward.
usage: contention.py [vitl|vits]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
cfg = get_config(enc)
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
side = torch.cuda.Stream()
m.forward(x, fp32=False); torch.cuda.synchronize()
for wgs, ms in ((0, 0), (8, 3), (16, 3), (32, 3), (16, 6), (32, 6), (64, 6)):
    ts = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            if wgs:
                _lib.lib.vda_debug_occupy(wgs, 16384, int(ms * 2.1e6), side.cuda_stream)
            m.forward(x, fp32=False)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    print(f"{enc}: {wgs:3d} foreign workgroups for ~{ms} ms per forward: {sorted(ts)[1]:.2f} ms/clip", flush=True)
