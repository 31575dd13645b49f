"""Sliding-window schedule, multi-GPU sharding and window stitching of infer_video_depth.

Host logic only (numpy); mirrors /root/reference/video_depth_anything/video_depth.py:187-254
and /root/reference/utils/util.py:40-74.

Key property: the overlap slots of window k are refilled from the previous
window's INPUT (video_depth.py:200-201), never from an output, so every window's 32 input
frames are a pure function of the source video. `plan_windows` resolves that recursion into
source-frame indices up front; windows are then independent units that can be computed in any
order / on any rank, and only the final stitch is sequential.
"""
from typing import Callable, List, Sequence, Tuple

import numpy as np

from .config import IMAGENET_MEAN, IMAGENET_STD, INFER_LEN, INTERP_LEN, KEYFRAMES, OVERLAP, PATCH


# ------------------------------------------------------------------ preprocessing geometry
def network_size(height: int, width: int, input_size: int = 518) -> Tuple[int, int]:
    """Network input (H, W) for a source frame: aspect guard (video_depth.py:167-171) then the
    keep-aspect, lower-bound, multiple-of-14 rule of Resize.get_size (util/transform.py:51-107)."""
    ratio = max(height, width) / min(height, width)
    if ratio > 1.78:
        input_size = int(input_size * 1.777 / ratio)
        input_size = round(input_size / 14) * 14

    def constrain(x, min_val):
        y = int(np.round(x / PATCH) * PATCH)
        if y < min_val:
            y = int(np.ceil(x / PATCH) * PATCH)
        return y

    scale = max(input_size / height, input_size / width)
    return constrain(scale * height, input_size), constrain(scale * width, input_size)


# ------------------------------------------------------------------ window schedule
def plan_windows(n_frames: int) -> List[List[int]]:
    """Source-frame index of every input slot of every window.

    video_depth.py:188-201: stride 22, the video is padded with copies of its last frame
    ((22 - n % 22) % 22 + 10 of them), and for k > 0 slots 0..9 are the previous window's
    slots KEYFRAMES. Padded positions map back to the last real frame."""
    if n_frames <= 0:
        raise ValueError("empty video")
    step = INFER_LEN - OVERLAP
    windows: List[List[int]] = []
    prev = None
    for start in range(0, n_frames, step):
        cur = [min(start + i, n_frames - 1) for i in range(INFER_LEN)]
        if prev is not None:
            cur[:OVERLAP] = [prev[k] for k in KEYFRAMES]
        windows.append(cur)
        prev = cur
    return windows


def shard_windows(n_windows: int, world: int, rank: int) -> range:
    """Contiguous block partition: the first (n % world) ranks take one extra window."""
    q, r = divmod(n_windows, world)
    lo = rank * q + min(rank, r)
    return range(lo, lo + q + (1 if rank < r else 0))


def gathered_order(n_windows: int, world: int) -> Tuple[int, List[int]]:
    """Layout of the all-gather: every rank sends `per` window slots (short ranks pad); returns (per, flat slot index
    r * per + j of window 0, 1, 2, ... in window order)."""
    per = (n_windows + world - 1) // world
    order = [r * per + j for r in range(world) for j in range(len(shard_windows(n_windows, world, r)))]
    return per, order


# ------------------------------------------------------------------ stitching
def compute_scale_and_shift(prediction: np.ndarray, target: np.ndarray) -> Tuple[float, float]:
    """Closed-form least squares target ~ scale*prediction + shift over all pixels
    (utils/util.py:40-62 with the all-ones mask of video_depth.py:232): fp32 sums, identity if det == 0."""
    prediction = prediction.astype(np.float32)
    target = target.astype(np.float32)
    ones = np.ones_like(target, dtype=np.float32)
    a_00 = np.sum(ones * prediction * prediction)
    a_01 = np.sum(ones * prediction)
    a_11 = np.sum(ones)
    b_0 = np.sum(ones * prediction * target)
    b_1 = np.sum(ones * target)
    det = a_00 * a_11 - a_01 * a_01
    if det != 0:
        return (a_11 * b_0 - a_01 * b_1) / det, (-a_01 * b_0 + a_00 * b_1) / det
    return 1, 0


def crossfade(pre: Sequence[np.ndarray], post: Sequence[np.ndarray]) -> List[np.ndarray]:
    """utils/util.py:65-74: weight of `post` ramps 0, 1/7, ..., 6/7, 1."""
    n = len(pre)
    assert n == len(post)
    step = 1.0 / (n - 1)
    wts = [0.0] + [i * step for i in range(1, n - 1)] + [1.0]
    return [pre[i] * (1 - wts[i]) + post[i] * wts[i] for i in range(n)]


def _clamped_affine(d, scale, shift):
    out = d * scale + shift
    out[out < 0] = 0
    return out


def stitch_windows(window_depths: Sequence[np.ndarray], n_frames: int, metric: bool = False) -> np.ndarray:
    """video_depth.py:216-254 over per-window depth [32,H0,W0] arrays (window order).
    metric=True: scale, shift = 1, 0 (metric_depth/video_depth_anything/video_depth.py:132)."""
    align_len = OVERLAP - INTERP_LEN
    kf_align = KEYFRAMES[:align_len]
    aligned: List[np.ndarray] = []
    ref_align: List[np.ndarray] = []
    for k, wd in enumerate(window_depths):
        frames = [wd[i] for i in range(INFER_LEN)]
        if k == 0:
            aligned += frames
            ref_align = [frames[kf] for kf in kf_align]
            continue
        if metric:
            scale, shift = 1.0, 0.0
        else:
            scale, shift = compute_scale_and_shift(np.concatenate(frames[:align_len]), np.concatenate(ref_align))
        post = [_clamped_affine(frames[i], scale, shift) for i in range(align_len, OVERLAP)]
        aligned[-INTERP_LEN:] = crossfade(aligned[-INTERP_LEN:], post)
        for i in range(OVERLAP, INFER_LEN):
            aligned.append(_clamped_affine(frames[i], scale, shift))
        ref_align = ref_align[:1] + [_clamped_affine(frames[kf], scale, shift) for kf in kf_align[1:]]
    return np.stack(aligned[:n_frames], axis=0)


# ------------------------------------------------------------------ driver (single or multi rank)
def run_windows(frames: np.ndarray, window_fn: Callable[[np.ndarray], np.ndarray] = None, metric: bool = False,
                group=None, batch_fn: Callable[[List[List[int]]], List[np.ndarray]] = None) -> np.ndarray:
    """Compute every window with `window_fn(frames_u8[32,H0,W0,3]) -> float32 [32,H0,W0]` (or, pipelined,
    `batch_fn(list of 32-index lists) -> list of float32 [32,H0,W0]` over this rank's windows), stitch.

    With torch.distributed initialised (one process per GPU; NCCL == RCCL over xGMI, gloo on CPU),
    windows are block-partitioned over the ranks, there is no data-path collective while they are
    computed, and ONE all-gather of the per-window depth maps precedes the (cheap, sequential)
    stitch, which every rank then runs redundantly so all ranks return the full sequence."""
    import torch
    import torch.distributed as dist

    n = frames.shape[0]
    plan = plan_windows(n)
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    mine = shard_windows(len(plan), world, rank)
    if batch_fn is not None:
        local = [np.ascontiguousarray(d, dtype=np.float32) for d in batch_fn([plan[k] for k in mine])]
    else:
        local = [np.ascontiguousarray(window_fn(frames[plan[k]]), dtype=np.float32) for k in mine]
    if world == 1:
        return stitch_windows(local, n, metric)

    H0, W0 = frames.shape[1:3]
    per, order = gathered_order(len(plan), world)               # short ranks pad to a common count
    use_cuda = dist.get_backend(group) == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if use_cuda else torch.device("cpu")
    send = torch.zeros(per, INFER_LEN, H0, W0, dtype=torch.float32, device=dev)
    if local:
        send[:len(local)] = torch.from_numpy(np.stack(local)).to(dev)
    recv = torch.empty(world * per, INFER_LEN, H0, W0, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.cpu().numpy()
    return stitch_windows([recv[i] for i in order], n, metric)


def normalize_frames_host(frames_u8: np.ndarray) -> np.ndarray:
    """Reference arithmetic of video_depth.py:198 + util/transform.py:134,147 for frames already at
    network size (host version, used by CPU-side tests of the plumbing)."""
    img = frames_u8.astype(np.float32) / 255.0
    img = (img - list(IMAGENET_MEAN)) / list(IMAGENET_STD)
    return np.ascontiguousarray(np.moveaxis(img, -1, -3)).astype(np.float32)
