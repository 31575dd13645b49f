import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import state_dict_spec
enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
if len(sys.argv) > 2:
    _lib.lib.vda_gemm_set_variant(int(sys.argv[2]))
cfg = get_config(enc)
g = torch.Generator().manual_seed(0)
sd = {k: (torch.randn(s, generator=g) * 0.02 if len(s) > 1 else torch.ones(s)) for k, s in state_dict_spec(cfg).items()}   # fast init
m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(sd); m = m.to("cuda")
x = torch.randn(1, 32, 3, 518, 518, generator=g).cuda()
d = m.forward(x, fp32=False); torch.cuda.synchronize()
print("done", float(d.mean()))
