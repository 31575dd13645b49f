"""A/B of the two residual placements against the CPU oracle (diagnostic)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vda_oracle as O
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict

def rel(a, b): return float(np.abs(a - b).mean() / np.abs(b).mean())
for name, seed, shape, xs in (("vits", 14, (1, 3, 3, 70, 84), 75), ("vits", 0, (1, 3, 3, 56, 70), 102), ("vits", 7, (1, 2, 3, 518, 518), 70), ("tiny", 1, (1, 4, 3, 42, 56), 101)):
    cfg = get_config(name); sd = synthetic_state_dict(cfg, seed=seed)
    m = VideoDepthAnything(encoder=name, features=cfg.features, out_channels=list(cfg.out_channels)); m.load_state_dict(sd); m = m.to("cuda")
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(xs))
    with torch.no_grad(): ref = O.forward(sd, cfg, x).numpy()
    a = m.forward(x.cuda(), fp32=False).cpu().numpy()
    m.engine.set_option("residual_in_ln", 0)
    b = m.forward(x.cuda(), fp32=False).cpu().numpy()
    f = m.forward(x.cuda(), fp32=True).cpu().numpy()
    print(name, seed, shape, "mean ref %.3f" % np.abs(ref).mean(), "in_ln vs oracle %.2e" % rel(a, ref), "epilogue vs oracle %.2e" % rel(b, ref), "in_ln vs epilogue %.2e" % rel(a, b), "fp32 vs oracle %.2e" % rel(f, ref), flush=True)
