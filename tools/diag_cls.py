"""Diagnostic: fp16-path error per stage against the engine's own fp32 path, LayerNorm fold on / off (tiny fixtures)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
def rel(a,b): return float(np.abs(a.astype(np.float64)-b).mean()/np.abs(b).mean())
for fixture, kw in (("tiny_clstoken_forward.npz", dict(use_clstoken=True)), ("tiny_forward.npz", {}), ("tiny_rope_forward.npz", dict(pe="rope"))):
    z = np.load("tests/golden/" + fixture)
    cfg = get_config("tiny", **kw)
    m = VideoDepthAnything(encoder="tiny", features=cfg.features, out_channels=list(cfg.out_channels), **kw)
    m.load_state_dict(synthetic_state_dict(cfg, seed=int(z["sd_seed"])), strict=True)
    m = m.to("cuda").eval()
    x = torch.from_numpy(z["x"]).cuda()
    names = ["tap0", "tap1", "tap2", "tap3", "layer_1", "layer_2", "layer_3", "layer_4", "path_4", "path_3", "path_2"]
    d32 = m.forward(x, fp32=True).cpu().numpy()
    s32 = {k: m.engine.stage(k)[0].float().cpu().numpy().copy() for k in names}
    print(fixture, "anchor", rel(z["depth_autocast_fp16"], z["depth"]))
    for fold in (1, 0):
        m.engine.set_option("ln_fold", fold)
        d = m.forward(x, fp32=False).cpu().numpy()
        row = " ".join(f"{k}={rel(m.engine.stage(k)[0].float().cpu().numpy(), s32[k]):.2e}" for k in names)
        print("  fold", fold, f"depth={rel(d, d32):.2e}", row)
