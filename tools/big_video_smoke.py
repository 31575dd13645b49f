import sys, time, numpy as np, torch
sys.path.insert(0, "/root/repo")
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
from video_depth_anything_amd.scheduler import network_size
for enc, shape in (("vits", (40, 720, 1280, 3)), ("vitl", (33, 1080, 1920, 3))):
    cfg = get_config(enc)
    m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(synthetic_state_dict(cfg, seed=1)); m = m.to("cuda")
    frames = np.random.default_rng(0).integers(0, 256, shape, dtype=np.uint8)
    t0 = time.time(); d, fps = m.infer_video_depth(frames, 24); dt = time.time() - t0
    print(enc, shape, "network", network_size(shape[1], shape[2]), "->", d.shape, d.dtype, bool(np.isfinite(d).all()), float(d.min()), round(dt, 2), "s", flush=True)
