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


# ------------------------------------------------------------------ the key-frame exchange on the device (scheduler.drive_windows_keys)
class DeviceKeyOps:
    """Device side of the key-frame schedule: the same kernels in the same order as DeviceStitcher, so every output frame is
    bit-equal to the single-rank result. Runs on the caller's (consumer) stream; `window_depth`, `ready`, `release` and `gather`
    are infer_video_depth's closures (window compute on the slot's lane, event plumbing, the backend's all-gather)."""

    def __init__(self, send, H0, W0, device, metric, world, rank, n_windows, result_ranks, window_depth, ready, release, gather, acquire):
        from .scheduler import KEY_SLOTS, PIECE_FRAMES
        self.send, self.px, self.metric, self.world, self.rank, self.n_windows = send, H0 * W0, metric, world, rank, n_windows
        self.result_ranks, self.window_depth, self.ready, self.release, self.gather, self.acquire = result_ranks, window_depth, ready, release, gather, acquire
        f32 = dict(dtype=torch.float32, device=device)
        nk = len(KEY_SLOTS)
        self.keys_send = [torch.empty(nk, H0, W0, **f32) for _ in range(2)]
        self.keys_recv = [torch.empty(world, nk, H0, W0, **f32) for _ in range(2)]
        self.piece = [torch.empty(PIECE_FRAMES, H0, W0, **f32) for _ in range(2)]
        wanted = result_ranks is None or rank in result_ranks
        self.piece_recv = [torch.empty(world, PIECE_FRAMES, H0, W0, **f32) for _ in range(2)] if (wanted and world > 1) else None
        self.ref = torch.empty(ALIGN_LEN, H0, W0, **f32)                 # (window 0 slot 0, aligned slot 12 of the last chained window)
        self.tail = torch.empty(INTERP_LEN, H0, W0, **f32)               # aligned slots 24..31 of the last chained window
        self.my_tail = torch.empty(INTERP_LEN, H0, W0, **f32)            # ... of the window before this rank's current one
        self.my_ss = torch.tensor([1.0, 0.0], **f32)
        self.ss = torch.tensor([1.0, 0.0], **f32)
        self.scratch = torch.empty(H0, W0, **f32)
        self.workspace = torch.empty(4 * ops.LSQ_BLOCKS, dtype=torch.float64, device=device)
        self.wts = torch.from_numpy(crossfade_weights()).to(device)
        self.last_copy = None                                             # event behind the consumer's latest piece copy (collect_pieces)

    def compute(self, k, s):
        self.window_depth(k, s, self.keys_send[s])

    def gather_keys(self, s):
        return self.gather(s, self.keys_recv[s], self.keys_send[s])

    def chain(self, k, s, r):
        if r == 0:
            self.ready(s)                                                 # the consumer stream waits for the slot's lane (window + key gather)
        keys = self.keys_recv[s][r]
        mine = k % self.world == self.rank
        if mine:
            self.my_tail.copy_(self.tail)                                 # aligned tail of window k - 1 (unused for k == 0)
        if k == 0:
            self.ref[0].copy_(keys[0])
            self.ref[1].copy_(keys[2])
            self.tail.copy_(keys[3:])
        else:
            if not self.metric:
                ops.lsq_scale_shift(keys[:ALIGN_LEN], self.ref, self.workspace, self.ss)
            ops.affine_clamp(keys[2], self.ss, self.ref[1])
            ops.affine_clamp(keys[3:], self.ss, self.tail)
        if mine:
            self.my_ss.copy_(self.ss)

    def _pieces_left(self):
        """Before a piece buffer is rewritten: the consumer's copies of the pieces yielded so far are behind us."""
        if self.last_copy is not None:
            torch.cuda.current_stream().wait_event(self.last_copy)

    def finalise(self, k, s):
        win = self.send[s]
        self._pieces_left()
        if k == 0:
            self.piece[s].copy_(win[:FIRST])
        else:
            ops.stitch_window(win, self.my_ss, self.piece[s], self.my_tail, self.scratch, self.px, self.wts)

    def deliver(self, s, j):
        import torch.distributed as dist
        ks = [j * self.world + r for r in range(self.world)]
        if self.world == 1:
            self.release(s)
            return [(ks[0], self.piece[s])]
        self._pieces_left()
        got = []
        if self.result_ranks is None:
            _exchange_all(self.piece_recv[s], self.piece[s])
            got = [(k, self.piece_recv[s][r]) for r, k in enumerate(ks) if k < self.n_windows]
        else:
            for dst in self.result_ranks:
                _exchange_to(self.piece_recv[s] if self.rank == dst else None, self.piece[s], dst, self.world)
                if self.rank == dst:
                    got = [(k, self.piece_recv[s][r]) for r, k in enumerate(ks) if k < self.n_windows]
        self.release(s)                                                   # send[s], keys of the slot: consumed
        return got

    def last_tail(self, k, owner):
        return self.tail

    def copied(self, event):
        self.last_copy = event


def _exchange_all(out, inp):
    """out[r] = rank r's inp (RCCL all-gather; any other backend - the shared-GPU rehearsal over gloo - staged through the host).
    The wait is bound to the current stream here: nothing downstream depends on which stream is current later."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(out, inp, async_op=True).wait()
        return
    parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, inp.cpu())
    out.copy_(torch.stack(parts))


def _exchange_to(out, inp, dst, world):
    """rank dst: out[r] = rank r's inp; elsewhere nothing is received."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        dist.gather(inp, list(out.unbind(0)) if out is not None else None, dst=dst)
        return
    bufs = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(world)] if out is not None else None
    dist.gather(inp.cpu(), bufs, dst=dst)
    if out is not None:
        out.copy_(torch.stack(bufs))


def collect_pieces(pieces, n_frames, H0, W0, device, on_copied=None):
    """(first output frame, count, device frames) pieces -> host float32 [n_frames, H0, W0]; each piece crosses to the host once,
    through a small ring of pinned buffers on a side stream."""
    out = np.empty((n_frames, H0, W0), dtype=np.float32)
    compute = torch.cuda.current_stream(device)
    copy_stream = torch.cuda.Stream(device=device)
    NB = 4
    pinned = [torch.empty(FIRST, H0, W0, dtype=torch.float32, pin_memory=True) for _ in range(NB)]
    done = [torch.cuda.Event() for _ in range(NB)]
    inflight = [None] * NB
    seen = np.zeros(n_frames, dtype=np.int32)

    def land(b):
        if inflight[b] is not None:
            done[b].synchronize()
            lo, hi = inflight[b]
            out[lo:hi] = pinned[b][:hi - lo].numpy()
            seen[lo:hi] += 1
            inflight[b] = None

    i = 0
    for pos, cnt, frames in pieces:
        hi = min(pos + cnt, n_frames)
        if hi <= pos:
            continue
        b = i % NB
        land(b)
        copy_stream.wait_stream(compute)
        with torch.cuda.stream(copy_stream):
            pinned[b][:hi - pos].copy_(frames[:hi - pos], non_blocking=True)
            done[b].record(copy_stream)
        inflight[b] = (pos, hi)
        if on_copied is not None:
            on_copied(done[b])
        i += 1
    for b in range(NB):
        land(b)
    compute.wait_stream(copy_stream)
    assert (seen == 1).all(), "every output frame exactly once"
    return out
