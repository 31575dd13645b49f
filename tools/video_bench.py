#!/usr/bin/env python
"""End-to-end infer_video_depth throughput on one GPU (BASELINE.json config 4 shape: N synthetic 518x518 frames,
ViT-L): uint8 frames start in HOST memory, float32 depth ends in host memory (PCIe-inclusive), stitch included.
Reports OUTPUT frames/s; the sliding window computes 32 frames per 22 new ones (1.47x redundancy at N=1024)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.scheduler import plan_windows
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import state_dict_spec

enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cfg = get_config(enc)
g = torch.Generator().manual_seed(0)
sd = {k: (torch.randn(s, generator=g) * (0.02 if len(s) > 1 else 0.1) + (0 if len(s) > 1 else 1)) for k, s in state_dict_spec(cfg).items()}
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(sd); m = m.to("cuda")
frames = np.random.default_rng(0).integers(0, 256, (N, 518, 518, 3), dtype=np.uint8)
m.infer_video_depth(frames[:40], 24)                      # warm-up (allocations, first-touch)
torch.cuda.synchronize()
t0 = time.perf_counter()
d, _ = m.infer_video_depth(frames, 24)
dt = time.perf_counter() - t0
nw = len(plan_windows(N))
print(f"{enc} N={N} windows={nw}: {dt:.2f} s  -> {N / dt:.1f} output frames/s ({nw * 32 / dt:.1f} computed frames/s, {dt / nw * 1e3:.1f} ms/window); depth {d.shape} {d.dtype}")
