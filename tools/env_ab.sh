#!/bin/bash
# Same box, alternating processes: tools/env_ab.sh VAR v0 v1 [encoder]  -> ms per clip of tools/vitl_once-style loops (for switches that are read once per process)
var=$1; a=$2; b=$3; enc=${4:-vitl}
for rep in 1 2 3; do
  for v in $a $b; do
    env $var=$v python3 - "$enc" "$var=$v" <<'PY'
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
print(f"{enc} {tag}: {sorted(ts)[2]:.3f} ms/clip  all {[round(t, 2) for t in ts]}", flush=True)
PY
  done
done
