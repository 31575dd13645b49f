#!/usr/bin/env python
"""Per-shape table of every GEMM / conv launch in one clip forward: time, TFLOP/s, rounds of 256-row tiles."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import state_dict_spec
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
cfg = get_config(enc)
g = torch.Generator().manual_seed(0)
sd = {k: (torch.randn(s, generator=g) * 0.02 if len(s) > 1 else torch.ones(s)) for k, s in state_dict_spec(cfg).items()}
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(sd); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=g).cuda()
PE = m.python_engine()          # the Python launch sequence goes through ops.gemm (the handle's C++ one does not)
PE.forward(x); torch.cuda.synchronize()
rec = []
orig = ops.gemm
def wrapped(A, W, out, epi, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); orig(A, W, out, epi, **kw); e1.record()
    rec.append(((kw["M"], kw["N"], kw["K"], epi, "conv" if kw.get("conv") else "dense", _lib.lib.vda_gemm_last_kernel().decode()), e0, e1))
ops.gemm = wrapped
for _ in range(3): PE.forward(x)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, e0, e1 in rec:
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
tot = 0
for (M, N, K, epi, mode, kern), (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = ms / n * 1e3; tot += ms / 3
    print(f"{mode:5s} M={M:8d} N={N:5d} K={K:5d} epi={epi} x{n//3:3d}  {us:8.1f} us  {2.0*M*N*K/us/1e6:7.0f} TF/s  {ms/3:6.2f} ms/fwd  {kern}")
print("total gemm ms/fwd", tot)
