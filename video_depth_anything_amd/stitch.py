"""Window stitcher on the device: the sequential tail of infer_video_depth
(/root/reference/video_depth_anything/video_depth.py:216-254, utils/util.py:40-74) as two HIP launches per window.

The numpy stitcher in scheduler.py restates the same arithmetic on the host (used by the CPU-side plumbing tests
and as the cross-check of this one). Here the per-window depth maps never leave HBM before they are final:

    window 0   : frames 0..23 are final, 24..31 become the tail, frames 0 and 12 the alignment references
    window k>0 : scale/shift = least squares of its frames 0..1 against the references   (vda_lsq_scale_shift_f32)
                 8-frame cross-fade into the tail, 14 final frames, new tail, new reference (vda_stitch_window_f32)

so a window contributes 22 final frames (24 for the first) and the host receives each frame exactly once.
"""
import numpy as np
import torch

from . import ops
from .config import INFER_LEN, INTERP_LEN, KEYFRAMES, OVERLAP

ALIGN_LEN = OVERLAP - INTERP_LEN
STEP = INFER_LEN - OVERLAP                     # 22 new frames per window
FIRST = INFER_LEN - INTERP_LEN                 # 24 frames final after window 0
assert (INFER_LEN, OVERLAP, INTERP_LEN, ALIGN_LEN) == (32, 10, 8, 2) and tuple(KEYFRAMES[:2]) == (0, 12), \
    "vda_stitch_window_f32 is built for the released schedule (32-frame windows, 10 overlap, 8 interpolated, key frames 0 and 12)"


def crossfade_weights():
    """utils/util.py:65-74: weight of the new window ramps 0, 1/7, ..., 6/7, 1; numpy multiplies float32 frames by the
    python floats (1 - w) and w, i.e. by their float32 roundings."""
    step = 1.0 / (INTERP_LEN - 1)
    w = [0.0] + [i * step for i in range(1, INTERP_LEN - 1)] + [1.0]
    return np.array([1.0 - x for x in w] + w, dtype=np.float32)


class DeviceStitcher:
    def __init__(self, H0, W0, device, metric=False):
        self.px = H0 * W0
        self.metric = metric
        self.k = 0
        self.ref = torch.empty(ALIGN_LEN, H0, W0, dtype=torch.float32, device=device)
        self.tail = torch.empty(INTERP_LEN, H0, W0, dtype=torch.float32, device=device)
        self.scale_shift = torch.tensor([1.0, 0.0], dtype=torch.float32, device=device)   # metric: stays (1, 0)
        self.workspace = torch.empty(4 * ops.LSQ_BLOCKS, dtype=torch.float64, device=device)
        self.wts = torch.from_numpy(crossfade_weights()).to(device)

    def push(self, win, chunk):
        """win: fp32 [32,H0,W0] (device) = the next window in order; writes its final frames to chunk[:n], returns n."""
        if self.k == 0:
            chunk[:FIRST].copy_(win[:FIRST])
            self.tail.copy_(win[FIRST:])
            self.ref[0].copy_(win[KEYFRAMES[0]])
            self.ref[1].copy_(win[KEYFRAMES[1]])
            n = FIRST
        else:
            if not self.metric:
                ops.lsq_scale_shift(win[:ALIGN_LEN], self.ref, self.workspace, self.scale_shift)
            ops.stitch_window(win, self.scale_shift, chunk, self.tail, self.ref[1], self.px, self.wts)
            n = STEP
        self.k += 1
        return n

    def first_frame_of(self, k):
        """Output position of chunk[0] of window k."""
        return 0 if k == 0 else STEP * k + ALIGN_LEN

    def tail_position(self):
        return STEP * (self.k - 1) + FIRST


def stitch_stream(windows, n_frames, H0, W0, device, metric=False):
    """Stitch an iterator of device windows (fp32 [32,H0,W0], window order) into a host float32 [n_frames,H0,W0] array.
    Window k's ONE device-to-host copy (its 22 final frames, pinned buffer, side stream) overlaps whatever the iterator
    queues for window k+1 on the current stream."""
    st = DeviceStitcher(H0, W0, device, metric)
    out = np.empty((n_frames, H0, W0), dtype=np.float32)
    compute = torch.cuda.current_stream(device)
    copy_stream = torch.cuda.Stream(device=device)
    chunk = [torch.empty(FIRST, H0, W0, dtype=torch.float32, device=device) for _ in range(2)]
    pinned = [torch.empty(FIRST, H0, W0, dtype=torch.float32, pin_memory=True) for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    pending = None                                  # (slot, first output frame, count)

    def harvest(p):
        s, lo, cnt = p
        done[s].synchronize()
        hi = min(lo + cnt, n_frames)
        if hi > lo:
            out[lo:hi] = pinned[s][:hi - lo].numpy()

    def send(src, s, lo, cnt):
        copy_stream.wait_stream(compute)
        with torch.cuda.stream(copy_stream):
            pinned[s][:cnt].copy_(src[:cnt], non_blocking=True)
            done[s].record(copy_stream)
        return (s, lo, cnt)

    k = 0
    for win in windows:
        s = k & 1
        cnt = st.push(win, chunk[s])
        nxt = send(chunk[s], s, st.first_frame_of(k), cnt)
        if pending is not None:
            harvest(pending)
        pending = nxt
        k += 1
    if pending is None:
        raise ValueError("no windows")
    harvest(pending)
    harvest(send(st.tail, 0, st.tail_position(), INTERP_LEN))   # after the last window its tail is final too
    compute.wait_stream(copy_stream)
    return out
