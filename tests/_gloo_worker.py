"""Worker for test_two_ranks_over_gloo_equal_one_rank (launched by torch.distributed.run)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import vda_oracle as O  # noqa: E402
from video_depth_anything_amd import scheduler as S  # noqa: E402
from video_depth_anything_amd.config import get_config  # noqa: E402
from video_depth_anything_amd.weights import synthetic_state_dict  # noqa: E402


def main():
    out, n_frames = sys.argv[1], int(sys.argv[2])
    exchange = sys.argv[3] if len(sys.argv) > 3 else "windows"
    torch.set_num_threads(2)
    cfg = get_config("tiny")
    sd = synthetic_state_dict(cfg, seed=5)
    frames = np.random.default_rng(21).integers(0, 256, (n_frames, 28, 42, 3), dtype=np.uint8)   # 60 -> 3 windows, 100 -> 5
    calls = []

    def window_fn(win_u8):
        calls.append(1)
        x = torch.from_numpy(S.normalize_frames_host(win_u8))[None]
        with torch.no_grad():
            return O.forward(sd, cfg, x)[0].numpy()

    dist.init_process_group("gloo")
    rank = dist.get_rank()
    d = S.run_windows(frames, window_fn, exchange=exchange)
    nwin = len(S.plan_windows(n_frames))
    assert len(calls) == len(S.shard_windows(nwin, dist.get_world_size(), rank)), "each rank computes only its own windows"
    np.save(f"{out}_rank{rank}.npy", d)
    if exchange == "keys":                        # only rank 1 wants the video: the others deliver their pieces and return None
        d1 = S.run_windows(frames, window_fn, exchange="keys", result_ranks=(1,))
        assert (d1 is None) == (rank != 1) and (rank != 1 or np.array_equal(d1, d))
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        np.save(f"{out}_single.npy", S.run_windows(frames, window_fn))


if __name__ == "__main__":
    main()
