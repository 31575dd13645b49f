"""Soak: N forwards of the same clip must all be bit-identical (a rare race shows up as an occasional differing run)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import state_dict_spec
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
cfg = get_config(enc)
g = torch.Generator().manual_seed(0)
sd = {k: (torch.randn(s, generator=g) * 0.02 if len(s) > 1 else torch.ones(s)) for k, s in state_dict_spec(cfg).items()}
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels)); m.load_state_dict(sd); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=g).cuda()
ref = m.forward(x, fp32=False).clone()
bad = 0
for i in range(n):
    d = m.forward(x, fp32=False)
    if not torch.equal(d, ref):
        bad += 1
        print("run", i, "differs:", int((d != ref).sum()), "elements", flush=True)
print(f"soak {enc} x{n}: differing runs = {bad}")
