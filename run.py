#!/usr/bin/env python
"""CLI with the reference's flags and defaults (/root/reference/run.py:24-34) over the MI355X engine.

Frame I/O is utils/dc_utils.py (same two helpers as the reference's): `--input_video` may be an .npy / .npz (key `frames`)
of uint8 [N,H,W,3] RGB frames, a directory of images, a GIF, or a video file when decord or cv2 is importable; the
visualisations are mp4 through imageio when present and animated GIFs otherwise; `--save_npz` adds <name>_depths.npz.
`--metric` selects the metric-depth variant (metric_depth/run.py: ViT-L, no scale/shift alignment).
"""
import argparse
import os

import numpy as np
import torch

from utils.dc_utils import read_video_frames, save_video
from video_depth_anything_amd.video_depth import MetricVideoDepthAnything, VideoDepthAnything


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='Video Depth Anything (MI355X)')
    parser.add_argument('--input_video', type=str, default='./assets/example_videos/davis_rollercoaster.mp4')
    parser.add_argument('--output_dir', type=str, default='./outputs')
    parser.add_argument('--input_size', type=int, default=518)
    parser.add_argument('--max_res', type=int, default=1280)
    parser.add_argument('--encoder', type=str, default='vitl', choices=['vits', 'vitl'])
    parser.add_argument('--max_len', type=int, default=-1, help='maximum length of the input video, -1 means no limit')
    parser.add_argument('--target_fps', type=int, default=-1, help='target fps of the input video, -1 means the original fps')
    parser.add_argument('--fp32', action='store_true', help='model infer with torch.float32, default is torch.float16')
    parser.add_argument('--grayscale', action='store_true', help='do not apply colorful palette')
    parser.add_argument('--save_npz', action='store_true', help='save depths as npz')
    parser.add_argument('--save_exr', action='store_true', help='save depths as exr')
    parser.add_argument('--metric', action='store_true', help='metric-depth checkpoint and stitching (metric_depth/run.py)')
    parser.add_argument('--checkpoint', type=str, default=None, help='override ./checkpoints/<name>.pth; "synthetic" = seeded random weights')
    args = parser.parse_args()

    DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'
    model_configs = {
        'vits': {'encoder': 'vits', 'features': 64, 'out_channels': [48, 96, 192, 384]},
        'vitl': {'encoder': 'vitl', 'features': 256, 'out_channels': [256, 512, 1024, 1024]},
    }
    cls = MetricVideoDepthAnything if args.metric else VideoDepthAnything
    video_depth_anything = cls(**model_configs[args.encoder])
    ckpt = args.checkpoint or (f'./checkpoints/metric_video_depth_anything_{args.encoder}.pth' if args.metric
                               else f'./checkpoints/video_depth_anything_{args.encoder}.pth')
    if ckpt == "synthetic":
        from video_depth_anything_amd.weights import synthetic_state_dict
        sd = synthetic_state_dict(video_depth_anything.cfg, seed=0)
    else:
        sd = torch.load(ckpt, map_location='cpu', weights_only=True)
    video_depth_anything.load_state_dict(sd, strict=True)
    video_depth_anything = video_depth_anything.to(DEVICE).eval()

    frames, target_fps = read_video_frames(args.input_video, args.max_len, args.target_fps, args.max_res)   # run.py:53
    depths, fps = video_depth_anything.infer_video_depth(frames, target_fps, input_size=args.input_size, device=DEVICE, fp32=args.fp32)

    video_name = os.path.basename(args.input_video)
    os.makedirs(args.output_dir, exist_ok=True)
    stem = os.path.join(args.output_dir, os.path.splitext(video_name)[0])
    # run.py:57-62: <name>_src.mp4 and <name>_vis.mp4 (GIFs when no H.264 encoder is importable)
    src_path = save_video(frames, stem + '_src.mp4', fps=fps)
    vis_path = save_video(depths, stem + '_vis.mp4', fps=fps, is_depths=True, grayscale=args.grayscale)
    if args.save_npz:
        np.savez_compressed(stem + '_depths.npz', depths=depths)
    if args.save_exr:
        import Imath
        import OpenEXR
        exr_dir = stem + '_depths_exr'
        os.makedirs(exr_dir, exist_ok=True)
        for i, depth in enumerate(depths):
            header = OpenEXR.Header(depth.shape[1], depth.shape[0])
            header["channels"] = {"Z": Imath.Channel(Imath.PixelType(Imath.PixelType.FLOAT))}
            f = OpenEXR.OutputFile(f"{exr_dir}/frame_{i:05d}.exr", header)
            f.writePixels({"Z": depth.tobytes()})
            f.close()
    print(f"{depths.shape[0]} frames -> {vis_path}" + (f", {stem}_depths.npz" if args.save_npz else ""))
