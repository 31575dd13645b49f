#!/usr/bin/env python
"""Scene-batch driver with the reference's interface (/root/reference/benchmark/infer/infer.py:12-64) over the MI355X
engine: for every scene of every dataset in the JSON manifest, stack the scene's images, run
`infer_video_depth(..., fp32=True)` and write one .npy of float32 depth per frame next to `--infer_path`, so the
reference's own `benchmark/eval` scripts can score the output.

Differences that cannot be avoided offline: images are decoded with cv2 when importable, else PIL; either way the
array handed to the model is BGR like `cv2.imread`'s (the reference feeds BGR straight in, infer.py:54), `.npy` frames are
read as-is. `--checkpoint synthetic` substitutes seeded random weights (no trained checkpoints exist offline).
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from video_depth_anything_amd.video_depth import VideoDepthAnything  # noqa: E402


def imread_bgr(path):
    if path.endswith(".npy"):
        return np.load(path)
    try:
        import cv2
        return cv2.imread(path)
    except ImportError:
        from PIL import Image
        return np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[..., ::-1])


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--infer_path', type=str, default='')
    parser.add_argument('--json_file', type=str, default='')
    parser.add_argument('--datasets', type=str, nargs='+', default=['scannet', 'nyuv2'])
    parser.add_argument('--input_size', type=int, default=518)
    parser.add_argument('--encoder', type=str, default='vitl', choices=['vits', 'vitl'])
    parser.add_argument('--checkpoint', type=str, default=None, help='override ./checkpoints/video_depth_anything_<enc>.pth; "synthetic" = seeded weights')
    args = parser.parse_args()

    DEVICE = 'cuda' if torch.cuda.is_available() else 'cpu'
    model_configs = {
        'vits': {'encoder': 'vits', 'features': 64, 'out_channels': [48, 96, 192, 384]},
        'vitl': {'encoder': 'vitl', 'features': 256, 'out_channels': [256, 512, 1024, 1024]},
    }
    video_depth_anything = VideoDepthAnything(**model_configs[args.encoder])
    ckpt = args.checkpoint or f'./checkpoints/video_depth_anything_{args.encoder}.pth'
    if ckpt == "synthetic":
        from video_depth_anything_amd.weights import synthetic_state_dict
        sd = synthetic_state_dict(video_depth_anything.cfg, seed=0)
    else:
        sd = torch.load(ckpt, map_location='cpu', weights_only=True)
    video_depth_anything.load_state_dict(sd, strict=True)
    video_depth_anything = video_depth_anything.to(DEVICE).eval()

    with open(args.json_file, 'r') as fs:
        path_json = json.load(fs)
    root_path = os.path.dirname(args.json_file)
    for dataset in args.datasets:
        for data in path_json[dataset]:
            for key in data.keys():
                infer_paths, videos = [], []
                for images in data[key]:
                    image_path = os.path.join(root_path, images['image'])
                    stem = os.path.splitext(images['image'])[0]
                    infer_paths.append(os.path.join(args.infer_path, dataset, stem + '.npy'))
                    videos.append(imread_bgr(image_path))
                videos = np.stack(videos, axis=0)
                depths, fps = video_depth_anything.infer_video_depth(videos, 1, input_size=args.input_size, device=DEVICE, fp32=True)
                for infer_path, depth in zip(infer_paths, depths):
                    os.makedirs(os.path.dirname(infer_path), exist_ok=True)
                    np.save(infer_path, depth)
                print(f"{dataset}/{key}: {len(infer_paths)} frames")
