"""Worker for test_ranks_share_one_gpu (launched by torch.distributed.run, backend gloo, all ranks on cuda:0):
the multi-rank branch of infer_video_depth - round-robin shards, per-round exchange, two-slot rings, device stitch - against one rank."""
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
    frames = np.random.default_rng(21).integers(0, 256, (100, 28, 42, 3), dtype=np.uint8)   # 5 windows: ranks get 2 + 2 + 1
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d, _ = m.infer_video_depth(frames, 24, input_size=28)
    np.save(f"{out}_rank{rank}.npy", d)
    m.result_ranks = (0,)                         # only rank 0 stitches and copies back
    d0, fps = m.infer_video_depth(frames, 24, input_size=28)
    assert fps == 24 and ((d0 is None) != (rank == 0)) and (rank != 0 or np.array_equal(d0, d))
    kinds = [None] * world
    dist.all_gather_object(kinds, f"rank{rank}:{'None' if d0 is None else 'array'}")
    m.result_ranks = None
    # the key-frame exchange (scheduler.drive_windows_keys on the device): bit-equal, every rank / only rank 2 receiving
    m.exchange = "keys"
    dk, _ = m.infer_video_depth(frames, 24, input_size=28)
    assert np.array_equal(dk, d), f"rank {rank}: key-frame exchange differs from the window exchange"
    m.result_ranks = (2,)
    d2, _ = m.infer_video_depth(frames, 24, input_size=28)
    assert ((d2 is None) != (rank == 2)) and (rank != 2 or np.array_equal(d2, d))
    m.result_ranks = None
    m.exchange = "windows"
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        open(f"{out}_result_ranks.txt", "w").write(" ".join(kinds))
        d1, _ = m.infer_video_depth(frames, 24, input_size=28)
        np.save(f"{out}_single.npy", d1)


if __name__ == "__main__":
    main()
