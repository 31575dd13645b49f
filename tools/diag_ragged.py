"""Diagnostic: the n = 1 ragged video (one 28x42 frame = a 2x3 token grid). fp16-path error against the oracle for the four equally
precise forms of the engine (LayerNorm fold on/off x fused output_conv1 on/off), the oracle's own fp16-autocast anchor, and the same
at larger frames (more tokens -> less draw noise)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vda_oracle as O
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
def rel(a, b): return float(np.abs(a.astype(np.float64) - b).mean() / np.abs(b).mean())
cfg = get_config("tiny")
sd = synthetic_state_dict(cfg, seed=3)
m = VideoDepthAnything(encoder="tiny", features=cfg.features, out_channels=list(cfg.out_channels))
m.load_state_dict(sd, strict=True); m = m.to("cuda").eval()
for (h, w, size) in ((28, 42, 28), (42, 56, 42), (70, 98, 70), (140, 196, 140)):
    for n in (1, 5):
        frames = np.random.default_rng(100 + n).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
        ref = O.infer_video_depth(sd, cfg, frames, 24, input_size=size)[0]
        with torch.no_grad(), torch.autocast("cpu", dtype=torch.float16):
            a16 = O.infer_video_depth(sd, cfg, frames, 24, input_size=size)[0]
        row = []
        for fold in (1, 0):
            for fused in (1, 0):
                m.engine.set_option("ln_fold", fold); m.engine.set_option("oc1_fused", fused)
                d, _ = m.infer_video_depth(frames, 24, input_size=size, device="cuda", fp32=False)
                row.append(rel(d, ref))
        print(f"{h}x{w} n={n}: tokens/frame {h // 14 * (w // 14)}, zeros {float((ref == 0).mean()):.2f}, anchor {rel(a16, ref):.2e}, engine (fold,fused)=(1,1),(1,0),(0,1),(0,0): " + " ".join(f"{e:.2e}" for e in row))
