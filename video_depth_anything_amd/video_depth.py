"""`VideoDepthAnything`: the drop-in class for the reference's model wrapper.

Same constructor arguments, `load_state_dict(sd, strict=True)`, `forward(x)` and
`infer_video_depth(frames, target_fps, input_size=518, device='cuda', fp32=False)` as
/root/reference/video_depth_anything/video_depth.py:37-63,89-93,161-254, so the four callers
(run.py:45-50, metric_depth/run.py:43-48, app.py:34-48, benchmark/infer/infer.py:36-58) can switch
by changing one import. All arithmetic runs in libvda_hip.so: the model is a `vda_model` handle of the C ABI
(handle.py -> csrc/host.hip), preprocessing / resize / stitch are per-kernel entry points.

Precision follows the reference: `infer_video_depth(..., fp32=False)` is its autocast path (fp16 MFMA operands, fp32
accumulation and residual streams), `fp32=True` its full-fp32 path (fp32 operands on fp32-input MFMA). A bare
`model(x)` takes the precision from the ambient `torch.autocast` state exactly as the reference's nn.Module would
(fp32 outside autocast, fp16 inside); pass `fp32=` to be explicit.
"""
import numpy as np
import torch

from .config import INFER_LEN, get_config
from .scheduler import network_size


def _all_gather(out, inp):
    """out[r] = rank r's `inp`. NCCL (= RCCL): asynchronous on the collective's own stream, returns the work handle.
    Any other backend (gloo: the ranks-share-one-GPU rehearsal of tests/test_forward_gpu.py) is staged through the host."""
    import torch.distributed as dist
    if dist.get_backend() == "nccl":
        return dist.all_gather_into_tensor(out, inp, async_op=True)
    parts = [torch.empty(inp.shape, dtype=inp.dtype) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, inp.cpu())
    out.copy_(torch.stack(parts))
    return None


class VideoDepthAnything:
    METRIC = False   # metric variant stitches with scale=1, shift=0 (metric_depth/.../video_depth.py:132)

    def __init__(self, encoder='vits', features=64, out_channels=[48, 96, 192, 384], use_bn=False, use_clstoken=False,
                 num_frames=32, pe='ape', **_unused):
        # num_block / out_channel / conv of the fork's constructor (video_depth.py:47-49) are accepted and unused, as there.
        self.encoder = encoder
        self.intermediate_layer_idx = {'vits': [2, 5, 8, 11], 'vitl': [4, 11, 17, 23]}
        # use_bn / pe='rope' (no released configuration sets them): BatchNorm folded into the fusion blocks' convs at pack time,
        # rotary embedding of q / k in the temporal attention; any other pe raises NotImplementedError as motion_module.py:226-227
        self.cfg = get_config(encoder, features, out_channels, num_frames, use_clstoken, use_bn, pe)
        self.engine = None
        self._device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self._sd = None
        # Multi-rank runs (torch.distributed initialised): ranks that should receive the stitched video. None = every rank
        # (each process returns the full sequence); e.g. (0,) lets the other ranks skip the stitch and its device-to-host
        # copies - they return (None, target_fps).
        self.result_ranks = None
        # What the ranks exchange per round: "windows" = whole windows all-gathered, every result rank stitches the video
        # (north_star's form); "keys" = 11 key frames per window all-gathered, the scale/shift chain on every rank, each rank
        # finalises its own windows and only final frames travel to the result ranks (scheduler.drive_windows_keys).
        # Both give bit-identical videos.
        self.exchange = "windows"

    # ---- nn.Module-like surface used by the callers -------------------------------------------
    def load_state_dict(self, state_dict, strict=True):
        from .weights import check_state_dict
        check_state_dict(self.cfg, state_dict, strict)
        self._sd = state_dict
        if self.engine is not None:
            self.engine.load_state_dict(state_dict, strict)
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError("video_depth_anything_amd runs on an MI355X HIP device only (got device=%r)" % (device,))
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if self.engine is not None and device != self.engine.device:
            self.engine.close()                   # the handle is bound to one device: rebuild it on the new one
            self.engine = None
        self._device = device
        self._ensure_engine()
        return self

    def cuda(self):
        return self.to('cuda')

    def eval(self):
        return self

    def _ensure_engine(self):
        if self.engine is None:
            from .handle import ModelHandle       # imports the HIP library; fails loudly if it is missing
            self.engine = ModelHandle(self.cfg, self._device)
            if self._sd is not None:
                self.engine.load_state_dict(self._sd, True)
        return self.engine

    def python_engine(self):
        """The Python launch sequence over the per-kernel ABI (engine.py): the bit-exact cross-check of the handle."""
        from .engine import Engine
        if self.cfg.use_bn or self.cfg.pe != "ape":
            raise NotImplementedError("the Python launch sequence (engine.py) covers the released configurations only; "
                                      "use_bn / pe='rope' run through the handle (vda_forward)")
        e = Engine(self.cfg, self._ensure_engine().device)
        e.load_state_dict(self._sd, True)
        return e

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, x, fp32=None):
        """x [B,T,3,H,W] (H, W multiples of 14, T <= 32) -> depth fp32 [B,T,H,W]."""
        if fp32 is None:
            fp32 = not torch.is_autocast_enabled()
        return self._ensure_engine().forward(x, fp32=bool(fp32))

    __call__ = forward

    # ---- video inference ---------------------------------------------------------------------
    def infer_video_depth(self, frames, target_fps, input_size=518, device='cuda', fp32=False):
        if torch.device(device).type != 'cuda':
            raise RuntimeError("video_depth_anything_amd runs on an MI355X HIP device only (got device=%r)" % (device,))
        eng = self._ensure_engine()
        with torch.cuda.device(eng.device):
            return self._infer_video_depth(eng, frames, target_fps, input_size, bool(fp32))

    def _infer_video_depth(self, eng, frames, target_fps, input_size, fp32):
        import torch.distributed as dist
        from . import ops
        from .scheduler import drive_windows, plan_windows, shard_windows
        from .stitch import stitch_stream
        if not isinstance(frames, np.ndarray):
            frames = np.asarray(frames)
        if frames.ndim != 4 or frames.shape[-1] != 3:
            raise ValueError("infer_video_depth: frames must be [N, H, W, 3], got shape %r" % (tuple(frames.shape),))
        if frames.dtype != np.uint8:
            # The reference computes frame.astype(float32) / 255 on whatever it is handed (video_depth.py:198). The device path
            # keeps the video as uint8 in HBM, which is the same arithmetic exactly when the values ARE 0..255 integers: arrays of
            # any dtype holding such values (wider integers, float32 frames out of a cv2 pipeline) are converted; values that are
            # not 8-bit (fractions, negatives, > 255, NaN) are refused rather than silently truncated.
            ok = True
            for i in range(0, frames.shape[0], 64):              # chunked: a memory-mapped video is not paged in at once
                c = np.asarray(frames[i:i + 64])
                ok = bool(c.size == 0 or (c.min() >= 0 and c.max() <= 255 and
                                          (np.issubdtype(c.dtype, np.integer) or np.array_equal(c, np.rint(c)))))
                if not ok:
                    break
            if not ok:
                raise TypeError("infer_video_depth: frames must hold 8-bit values (uint8, or any dtype whose values are integers "
                                "within 0..255); got dtype %s with other values" % frames.dtype)
            frames = frames.astype(np.uint8)
        H0, W0 = frames.shape[1:3]
        H, W = network_size(H0, W0, input_size)
        dev = eng.device
        n = frames.shape[0]
        plan = plan_windows(n)
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0
        mine = list(shard_windows(len(plan), world, rank))
        # multi-rank: the per-round all-gather runs beside the next windows' kernels - GEMMs that find CUs taken by it should lose
        # those CUs, not a whole shift of tiles (dynamic tile draw)
        eng.set_option("dyn_sched", 1 if world > 1 else 0)

        # The uint8 frames THIS rank's windows read (frame 0, the previous window's key frame and its own 30 frames each -
        # SURVEY.md section 8e) live in HBM in a compact buffer, and each crosses PCIe once - not all up front: what the next
        # window needs is uploaded on a side stream while the current one computes.
        # `frames` may be a memory map (utils/dc_utils.read_video_frames on a .npy): only the runs a window needs are touched
        need = sorted({f for k in mine for f in plan[k]})
        slot_of = {f: i for i, f in enumerate(need)}
        video = torch.empty((max(len(need), 1), H0, W0, 3), dtype=torch.uint8, device=dev)
        compute = torch.cuda.current_stream(dev)
        upload = torch.cuda.Stream(device=dev)
        resident = set()

        def ensure(k):
            """Queue the upload of window k's not-yet-resident frames (runs of consecutive frames = one copy each)."""
            if k is None:
                return
            todo = sorted(f for f in set(plan[k]) if f not in resident)
            with torch.cuda.stream(upload):
                i = 0
                while i < len(todo):
                    j = i
                    while j + 1 < len(todo) and todo[j + 1] == todo[j] + 1:
                        j += 1
                    s0 = slot_of[todo[i]]
                    run = frames[todo[i]:todo[j] + 1]
                    if not (run.flags.c_contiguous and run.flags.writeable):
                        run = np.array(run)                  # a memory-mapped or strided source: page this run in
                    video[s0:s0 + (j - i + 1)].copy_(torch.from_numpy(run), non_blocking=True)
                    i = j + 1
            resident.update(todo)

        # Windows are independent, so TWO are kept in flight on this GPU, each on its own HIP stream with its own input buffer,
        # workspace slot and output slot: the tail rounds and launch gaps of one window's kernels are filled by the other's
        # (measured: +6 % ViT-L, +17 % ViT-S frames/s over one window at a time, tools/two_stream.py).
        NSLOT = 2
        lanes = [torch.cuda.Stream(device=dev) for _ in range(NSLOT)]
        computed = [torch.cuda.Event() for _ in range(NSLOT)]      # slot's window is in send[s] (recorded on its lane)
        freed = [torch.cuda.Event() for _ in range(NSLOT)]         # the consumer is done with the slot (recorded on `compute`)
        xin = [torch.empty(1, INFER_LEN, 3, H, W, dtype=torch.float32, device=dev) for _ in range(NSLOT)]
        send = [torch.empty(INFER_LEN, H0, W0, dtype=torch.float32, device=dev) for _ in range(NSLOT)]
        recv = [torch.empty(world, INFER_LEN, H0, W0, dtype=torch.float32, device=dev) for _ in range(NSLOT)] if world > 1 else None
        used = [False] * NSLOT

        def acquire(s):
            """Before anything of slot s (send[s], recv[s]) is overwritten: its lane waits until the consumer has finished with
            what the slot held two rounds ago."""
            if used[s]:
                lanes[s].wait_event(freed[s])
            used[s] = True

        def window_depth(k, s, keys=None):
            """Window k on lane s: gather (+ resize to the network size) + normalise (video_depth.py:197-201,
            util/transform.py:109-147), forward, resize to the source size (video_depth.py:207-208) into send[s] [32,H0,W0]
            (keys: the window's KEY_SLOTS frames are copied there too - three contiguous runs)."""
            ensure(k)
            lane = lanes[s]
            lane.wait_stream(upload)
            acquire(s)
            with torch.cuda.stream(lane):
                idx = torch.tensor([slot_of[f] for f in plan[k]], dtype=torch.int32, device=dev)
                if (H0, W0) == (H, W):
                    ops.gather_normalize_u8(video, idx, xin[s], INFER_LEN, H0, W0)
                else:
                    # cv2.resize(INTER_CUBIC) in the reference (util/transform.py:113); cv2 is absent offline, so this leg is
                    # PARITY UNPINNED against cv2 itself: the kernel evaluates cv2's published definition (a = -0.75, half-pixel
                    # centres, clamped taps) and is tested against that definition on the CPU.
                    ops.gather_resize_normalize_u8(video, idx, xin[s], INFER_LEN, H0, W0, H, W)
                depth = eng.forward(xin[s], fp32=fp32, slot=s)                   # [1,32,H,W] fp32
                ops.bilinear_plane(depth.view(INFER_LEN, H, W), send[s], INFER_LEN, H, W, H0, W0)
                if keys is not None:
                    keys[0:2].copy_(send[s][0:2])
                    keys[2].copy_(send[s][12])
                    keys[3:].copy_(send[s][INFER_LEN - 8:])
                computed[s].record(lane)
            pos = mine.index(k)
            ensure(mine[pos + 1] if pos + 1 < len(mine) else None)              # overlaps this window's compute

        def exchange(s):
            """The one exchange of the path (RCCL all-gather over xGMI), issued on the slot's lane behind its window. A rank with
            no window in this round still takes part, so the slot is acquired here too.
            The wait for the collective is bound HERE, explicitly, to the slot's lane (Work.wait() on the NCCL backend is a
            stream-side wait of whichever stream is current - it does not block the host): computed[s] is recorded behind it, and
            the consumer's stream waits for that event in ready(s). Nothing depends on which stream happens to be current when
            drive_windows harvests the round. The lane's next window (two rounds on) queues behind the gather, which by then has
            long finished under the other lane's compute."""
            acquire(s)
            with torch.cuda.stream(lanes[s]):
                h = _all_gather(recv[s], send[s])
                if h is not None:
                    h.wait()
                computed[s].record(lanes[s])
            return None

        def ready(s):
            compute.wait_event(computed[s])

        def release(s):
            freed[s].record(compute)

        if self.exchange == "keys":
            from .scheduler import KEY_SLOTS, drive_windows_keys
            from .stitch import DeviceKeyOps, collect_pieces
            assert tuple(KEY_SLOTS) == (0, 1, 12) + tuple(range(INFER_LEN - 8, INFER_LEN))

            def gather_keys(s, out, inp):
                acquire(s)
                with torch.cuda.stream(lanes[s]):
                    if world > 1:
                        h = _all_gather(out, inp)
                        if h is not None:
                            h.wait()
                    else:
                        out[0].copy_(inp)
                    computed[s].record(lanes[s])
                return None

            kops = DeviceKeyOps(send, H0, W0, dev, self.METRIC, world, rank, len(plan), self.result_ranks, window_depth, ready, release,
                                gather_keys, acquire)
            pieces = drive_windows_keys(len(plan), world, rank, kops, self.result_ranks)
            if self.result_ranks is not None and rank not in self.result_ranks:
                for _ in pieces:
                    pass
                eng.check()
                return None, target_fps
            depths = collect_pieces(pieces, n, H0, W0, dev, on_copied=kops.copied)
            for lane in lanes:
                compute.wait_stream(lane)
            eng.check()                                   # a window whose residual stream left fp16's range is an error, not a NaN video
            return depths, target_fps

        # One process per GPU: rank r computes windows r, r + world, ... with no data-path collective; after each round the
        # finished windows are all-gathered (asynchronously, under the next round's compute) and handed to the stitcher in window
        # order, so only a two-slot ring of gathered windows ever exists. Ranks without a window in the last round contribute an
        # unused slot.
        windows = drive_windows(len(plan), world, rank, send, recv, window_depth, exchange, ready, release)
        if self.result_ranks is not None and rank not in self.result_ranks:
            for _ in windows:                                                    # compute and exchange; no stitch, no D2H
                pass
            eng.check()
            return None, target_fps
        depths = stitch_stream(windows, n, H0, W0, dev, metric=self.METRIC)
        for lane in lanes:
            compute.wait_stream(lane)
        eng.check()
        return depths, target_fps


class MetricVideoDepthAnything(VideoDepthAnything):
    """metric_depth/video_depth_anything/video_depth.py: ViT-L defaults, no scale/shift alignment."""
    METRIC = True

    def __init__(self, encoder='vitl', features=256, out_channels=[256, 512, 1024, 1024], use_bn=False, use_clstoken=False,
                 num_frames=32, pe='ape'):
        super().__init__(encoder, features, out_channels, use_bn, use_clstoken, num_frames, pe)
