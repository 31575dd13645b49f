"""vda_workspace_bytes for the BASELINE shapes (run on the GPU box: the handle needs a device)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
for enc in ("vits", "vitl"):
    cfg = get_config(enc)
    m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
    for fp32 in (False, True):
        print(enc, "fp32" if fp32 else "fp16", "1x32x518x518 workspace %.2f GB" % (m.engine.workspace_bytes(1, 32, 518, 518, fp32) / 1e9), flush=True)
