"""Worker for test_two_ranks_share_one_gpu (launched by torch.distributed.run, backend gloo, both ranks on cuda:0):
the multi-rank branch of infer_video_depth - sharding, padded slots, gather order, device stitch - against one rank."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config  # noqa: E402
from video_depth_anything_amd.video_depth import VideoDepthAnything  # noqa: E402
from video_depth_anything_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    out = sys.argv[1]
    cfg = get_config("tiny")
    m = VideoDepthAnything(encoder="tiny", features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(synthetic_state_dict(cfg, seed=5), strict=True)
    m = m.to("cuda").eval()
    frames = np.random.default_rng(21).integers(0, 256, (60, 28, 42, 3), dtype=np.uint8)   # 3 windows: ranks get 2 + 1
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    d, _ = m.infer_video_depth(frames, 24, input_size=28)
    np.save(f"{out}_rank{rank}.npy", d)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        d1, _ = m.infer_video_depth(frames, 24, input_size=28)
        np.save(f"{out}_single.npy", d1)


if __name__ == "__main__":
    main()
