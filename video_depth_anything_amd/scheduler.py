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
    """Round-robin partition: rank r computes windows r, r + world, r + 2*world, ... Round j of the job therefore holds the
    CONSECUTIVE windows j*world .. j*world + world - 1, one per rank: the all-gather of a round delivers windows in stitch
    order and the stitcher can run behind the rounds with a bounded ring of gathered windows."""
    return range(rank, n_windows, world)


def rounds(n_windows: int, world: int) -> int:
    return (n_windows + world - 1) // world


def drive_windows(n_windows: int, world: int, rank: int, send, recv, compute, gather, ready=None, release=None):
    """THE multi-rank schedule of infer_video_depth (device path and CPU rehearsal alike): yields every window's depth map,
    in window order, on every rank.

      send     ring of >= 2 buffers [32,H0,W0]: compute(k, s) fills send[s] with window k
      recv     ring of >= 2 buffers [world,32,H0,W0] (None when world == 1)
      gather   gather(s) -> handle with .wait() (or None when the call is synchronous): recv[s][r] = rank r's send[s]
      ready    ready(s): called before anything of slot s is yielded (device path: the consumer's stream waits for the slot's
               compute stream - slots are computed on their own HIP streams so that two windows are in flight)
      release  release(s): called once the consumer has taken (queued its reads of) everything yielded from slot s, i.e. before
               the slot is overwritten two rounds later (device path: records the event the slot's next compute waits for)

    Round j: this rank computes window j*world + rank (if it exists) into slot j % 2 and starts the round's all-gather; the
    PREVIOUS round's gather is then waited for and its windows are yielded, so the exchange of round j runs under the compute
    of round j + 1 and the consumer (the stitcher) works one round behind."""
    nslot = len(send)
    if world == 1:
        for k in range(n_windows):
            s = k % nslot
            compute(k, s)
            if ready is not None:
                ready(s)
            yield send[s]
            if release is not None:
                release(s)
        return
    pending = None

    def harvest(p):
        j, s, h = p
        if h is not None:
            h.wait()
        if ready is not None:
            ready(s)
        for r in range(world):
            if j * world + r < n_windows:
                yield recv[s][r]
        if release is not None:
            release(s)

    for j in range(rounds(n_windows, world)):
        s = j % nslot
        k = j * world + rank
        if k < n_windows:
            compute(k, s)
        h = gather(s)
        if pending is not None:
            yield from harvest(pending)
        pending = (j, s, h)
    yield from harvest(pending)


# ------------------------------------------------------------------ the exchange that moves only what the stitch chain needs
# SURVEY.md section 8(e), "optional refinement": the only dependence BETWEEN windows in video_depth.py:216-250 is the chain
#   (scale_k, shift_k) = lsq(window k slots [0, 1]  ->  [window 0 slot 0, aligned window k-1 slot 12])
# plus the 8-frame cross-fade of window k's slots 2..9 into window k-1's aligned slots 24..31. So instead of all-gathering whole
# windows (32 frames per rank and round) the ranks all-gather 11 KEY frames per window - slots 0, 1 (the alignment pair), 12 (the
# next reference) and 24..31 (the tail the next window's owner cross-fades into) -, every rank runs the scale/shift chain on them,
# and each rank FINALISES ITS OWN windows: window k > 0 yields the 22 output frames 22k+2 .. 22k+23, window 0 the frames 0..23,
# the last window also its tail. Only those final pieces travel to the ranks that want the video.
# Per round and rank: 11 frames all-gathered + 22..24 sent to each result rank, against 32 all-gathered; a rank that is not a
# result rank receives 7 x 11 instead of 7 x 32 frames on 8 GPUs, and the stitch work is spread over the ranks.
KEY_SLOTS = (0, 1, 12, 24, 25, 26, 27, 28, 29, 30, 31)
PIECE_FRAMES = INFER_LEN - INTERP_LEN            # 24: the largest piece (window 0); later windows fill 22 of it


def piece_position(k: int) -> Tuple[int, int]:
    """(first output frame, frame count) of the piece window k's owner finalises."""
    return (0, PIECE_FRAMES) if k == 0 else ((INFER_LEN - OVERLAP) * k + (OVERLAP - INTERP_LEN), INFER_LEN - OVERLAP)


def drive_windows_keys(n_windows: int, world: int, rank: int, ops, result_ranks=None):
    """The key-frame schedule (device path and CPU rehearsal alike): yields (first output frame, count, frames[count..]) pieces,
    every piece of the video exactly once, on the ranks in `result_ranks` (None = every rank); nothing elsewhere.

    ops.compute(k, s)          window k into slot s of a two-slot ring, its KEY_SLOTS frames into the slot's key buffer
    ops.gather_keys(s)         start the all-gather of the slot's key buffers; returns a handle with .wait() or None
    ops.chain(k, s, r)         advance the scale/shift chain by window k, whose key frames are rank r's entry of slot s's gather
                               (called on every rank for every window, in window order)
    ops.finalise(k, s)         this rank's window k (still in slot s) -> its final piece, in the slot's piece buffer
    ops.deliver(s, j)          collective: the pieces of round j's windows to the result ranks; returns [(k, frames)] there, [] elsewhere
    ops.last_tail(k, owner)    the last window's aligned tail (8 frames): the chain holds it on every rank (its key frames are slots 24..31)
    Round j's key gather runs under round j + 1's compute; the chain, the finalisation and the delivery of a round happen one
    round behind, exactly as drive_windows harvests (and are issued before the next round's gather)."""
    mine = lambda k: k % world == rank                     # noqa: E731 - round-robin shards (shard_windows)
    wanted = result_ranks is None or rank in result_ranks
    pending = None

    def harvest(p):
        j, s, h = p
        if h is not None:
            h.wait()
        for r in range(world):
            k = j * world + r
            if k < n_windows:
                ops.chain(k, s, r)
        k = j * world + rank
        if k < n_windows:
            ops.finalise(k, s)
        for k2, frames in ops.deliver(s, j):
            pos, cnt = piece_position(k2)
            yield pos, cnt, frames

    nround = rounds(n_windows, world)
    for j in range(nround):
        s = j % 2
        k = j * world + rank
        if k < n_windows:
            ops.compute(k, s)
        # Round j-1's delivery is issued BEFORE round j's key gather: on the process group's one RCCL stream a collective queues
        # behind the ones issued before it, and gather_keys(j) waits for round j's compute - behind it the pieces of round j-1
        # would reach the result ranks a whole round late. Every rank issues the collectives in this same order.
        if pending is not None:
            yield from harvest(pending)
        h = ops.gather_keys(s)
        pending = (j, s, h)
    yield from harvest(pending)
    last = n_windows - 1
    tail = ops.last_tail(last, last % world)
    if wanted and tail is not None:
        yield (INFER_LEN - OVERLAP) * last + PIECE_FRAMES, INTERP_LEN, tail


class HostKeyOps:
    """drive_windows_keys on the host: numpy arithmetic identical to stitch_windows (so the result is bit-equal to the single-rank
    stitch), torch.distributed (gloo) on CPU tensors for the exchanges. The CPU tests' stand-in for the device ops."""

    def __init__(self, frames, plan, window_fn, metric, world, rank, result_ranks, group=None):
        import torch
        self.torch, self.frames, self.plan, self.window_fn, self.metric = torch, frames, plan, window_fn, metric
        self.world, self.rank, self.result_ranks, self.group = world, rank, result_ranks, group
        H0, W0 = frames.shape[1:3]
        self.win = [None, None]
        self.keys_send = [torch.zeros(len(KEY_SLOTS), H0, W0) for _ in range(2)]
        self.keys_recv = [torch.zeros(world, len(KEY_SLOTS), H0, W0) for _ in range(2)]
        self.piece = [torch.zeros(PIECE_FRAMES, H0, W0) for _ in range(2)]
        self.anchor = self.ref1 = None
        self.scale, self.shift = 1.0, 0.0
        self.tail = None                    # aligned slots 24..31 of the window the chain saw last
        self.my = {}                        # window -> (scale, shift, previous window's aligned tail) for the windows this rank owns
        self.last_tail_frames = None

    def compute(self, k, s):
        w = np.ascontiguousarray(self.window_fn(self.frames[self.plan[k]]), dtype=np.float32)
        self.win[s] = w
        self.keys_send[s].copy_(self.torch.from_numpy(w[list(KEY_SLOTS)]))

    def gather_keys(self, s):
        import torch.distributed as dist
        if self.world == 1:
            self.keys_recv[s][0].copy_(self.keys_send[s])
            return None
        n = len(KEY_SLOTS)
        return dist.all_gather_into_tensor(self.keys_recv[s].view(self.world * n, *self.keys_send[s].shape[1:]), self.keys_send[s],
                                           group=self.group, async_op=True)

    def chain(self, k, s, r):
        keys = self.keys_recv[s][r].numpy()
        prev_tail = self.tail
        if k == 0:
            self.anchor, self.ref1 = keys[0].copy(), keys[2].copy()
            self.scale, self.shift = 1.0, 0.0
            self.tail = [keys[3 + i].copy() for i in range(INTERP_LEN)]
        else:
            if not self.metric:
                self.scale, self.shift = compute_scale_and_shift(np.concatenate([keys[0], keys[1]]), np.concatenate([self.anchor, self.ref1]))
            self.ref1 = _clamped_affine(keys[2], self.scale, self.shift)
            self.tail = [_clamped_affine(keys[3 + i], self.scale, self.shift) for i in range(INTERP_LEN)]
        if k % self.world == self.rank:
            self.my[k] = (self.scale, self.shift, prev_tail)

    def finalise(self, k, s):
        w = self.win[s]
        scale, shift, prev_tail = self.my.pop(k)
        if k == 0:
            out = [w[i] for i in range(PIECE_FRAMES)]
        else:
            post = [_clamped_affine(w[i], scale, shift) for i in range(OVERLAP - INTERP_LEN, OVERLAP)]
            out = crossfade(prev_tail, post) + [_clamped_affine(w[i], scale, shift) for i in range(OVERLAP, INFER_LEN - INTERP_LEN)]
        self.piece[s][:len(out)].copy_(self.torch.from_numpy(np.stack(out)))

    def deliver(self, s, j):
        import torch.distributed as dist
        ks = [j * self.world + r for r in range(self.world)]
        if self.world == 1:
            return [(ks[0], self.piece[s].numpy().copy())]
        got = []
        dsts = range(self.world) if self.result_ranks is None else self.result_ranks
        for dst in dsts:
            bufs = [self.torch.empty_like(self.piece[s]) for _ in range(self.world)] if self.rank == dst else None
            dist.gather(self.piece[s], bufs, dst=dst, group=self.group)
            if self.rank == dst:
                got = [(k, bufs[r].numpy()) for r, k in enumerate(ks) if k < len(self.plan)]
        return got

    def last_tail(self, k, owner):
        return np.stack(self.tail)          # every rank ran the whole chain on key frames that include slots 24..31: already here


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


# ------------------------------------------------------------------ host driver (CPU rehearsal of the multi-rank schedule)
def run_windows(frames: np.ndarray, window_fn: Callable[[np.ndarray], np.ndarray], metric: bool = False, group=None, exchange: str = "windows",
                result_ranks=None):
    """infer_video_depth's schedule on the host: `window_fn(frames_u8[32,H0,W0,3]) -> float32 [32,H0,W0]` per window, the
    same `drive_windows` rounds / ring / gather order the device path uses (gloo on CPU tensors instead of RCCL), then the
    numpy stitcher. Every rank returns the full sequence. Used by the CPU tests (world size 1, 2, 3).
    exchange="keys": the key-frame schedule (drive_windows_keys); ranks outside result_ranks return None."""
    import torch
    import torch.distributed as dist

    n = frames.shape[0]
    H0, W0 = frames.shape[1:3]
    plan = plan_windows(n)
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if exchange == "keys":
        ops = HostKeyOps(frames, plan, window_fn, metric, world, rank, result_ranks, group)
        out = np.zeros((n, H0, W0), dtype=np.float32) if (result_ranks is None or rank in result_ranks) else None
        seen = np.zeros(n, dtype=np.int32)
        for pos, cnt, piece in drive_windows_keys(len(plan), world, rank, ops, result_ranks):
            hi = min(pos + cnt, n)
            if hi > pos:
                out[pos:hi] = piece[:hi - pos]
                seen[pos:hi] += 1
        assert out is None or (seen == 1).all(), "every output frame exactly once"
        return out
    send = [torch.zeros(INFER_LEN, H0, W0, dtype=torch.float32) for _ in range(2)]
    recv = [torch.empty(world, INFER_LEN, H0, W0, dtype=torch.float32) for _ in range(2)] if world > 1 else None

    def compute(k, s):
        send[s].copy_(torch.from_numpy(np.ascontiguousarray(window_fn(frames[plan[k]]), dtype=np.float32)))

    def gather(s):
        return dist.all_gather_into_tensor(recv[s].view(world * INFER_LEN, H0, W0), send[s], group=group, async_op=True)

    wins = [w.numpy().copy() for w in drive_windows(len(plan), world, rank, send, recv, compute, gather)]
    return stitch_windows(wins, n, metric)


def normalize_frames_host(frames_u8: np.ndarray) -> np.ndarray:
    """Reference arithmetic of video_depth.py:198 + util/transform.py:134,147 for frames already at
    network size (host version, used by CPU-side tests of the plumbing)."""
    img = frames_u8.astype(np.float32) / 255.0
    img = (img - list(IMAGENET_MEAN)) / list(IMAGENET_STD)
    return np.ascontiguousarray(np.moveaxis(img, -1, -3)).astype(np.float32)
