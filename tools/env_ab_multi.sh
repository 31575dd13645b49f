#!/bin/bash
# Same box, alternating processes, several environment settings: tools/env_ab_multi.sh <encoder> "A=1 B=2" "A=0 B=3" ...
enc=$1; shift
for rep in 1 2; do
  for cfg in "$@"; do
    env $cfg python3 - "$enc" "$cfg" <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
enc, tag = sys.argv[1], sys.argv[2]
cfg = get_config(enc)
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
for _ in range(5): m.forward(x, fp32=False)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): m.forward(x, fp32=False)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
print(f"{enc} [{tag}]: {sorted(ts)[2]:.3f} ms/clip  all {[round(t, 2) for t in ts]}", flush=True)
PY
  done
done
