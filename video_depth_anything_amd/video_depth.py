"""`VideoDepthAnything`: the drop-in class for the reference's model wrapper.

Same constructor arguments, `load_state_dict(sd, strict=True)`, `forward(x)` and
`infer_video_depth(frames, target_fps, input_size=518, device='cuda', fp32=False)` as
/root/reference/video_depth_anything/video_depth.py:37-63,89-93,161-254, so the four callers
(run.py:45-50, metric_depth/run.py:43-48, app.py:34-48, benchmark/infer/infer.py:36-58) can switch
by changing one import. All arithmetic runs in libvda_hip.so; see engine.py.
"""
import warnings

import numpy as np
import torch

from .config import INFER_LEN, get_config
from .scheduler import network_size

_FP32_WARNED = False


def _all_gather(out, inp):
    """out[r] = rank r's `inp`. NCCL (= RCCL): asynchronous on the collective's own stream, returns the work handle.
    Any other backend (gloo: the 2-ranks-on-one-GPU rehearsal of tests/test_forward_gpu.py) is staged through the host."""
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
        if use_bn or use_clstoken or pe != 'ape':
            raise NotImplementedError("only the released configuration (use_bn=False, use_clstoken=False, pe='ape') is built")
        self.encoder = encoder
        self.intermediate_layer_idx = {'vits': [2, 5, 8, 11], 'vitl': [4, 11, 17, 23]}
        self.cfg = get_config(encoder, features, out_channels, num_frames)
        self.engine = None
        self._device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self._sd = None

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
        self._device = device
        self._ensure_engine()
        return self

    def cuda(self):
        return self.to('cuda')

    def eval(self):
        return self

    def _ensure_engine(self):
        if self.engine is None:
            from .engine import Engine            # imports the HIP library; fails loudly if it is missing
            self.engine = Engine(self.cfg, self._device)
            if self._sd is not None:
                self.engine.load_state_dict(self._sd, True)
        return self.engine

    # ---- forward -----------------------------------------------------------------------------
    def forward(self, x):
        """x [B,T,3,H,W] (H, W multiples of 14, T <= 32) -> depth fp32 [B,T,H,W]."""
        return self._ensure_engine().forward(x)

    __call__ = forward

    # ---- video inference ---------------------------------------------------------------------
    def infer_video_depth(self, frames, target_fps, input_size=518, device='cuda', fp32=False):
        global _FP32_WARNED
        if torch.device(device).type != 'cuda':
            raise RuntimeError("video_depth_anything_amd runs on an MI355X HIP device only (got device=%r)" % (device,))
        if fp32 and not _FP32_WARNED:
            warnings.warn("fp32=True: this build computes with fp16 MFMA operands and fp32 accumulation/residuals; "
                          "an fp32-operand path is not built yet")
            _FP32_WARNED = True
        from . import ops
        eng = self._ensure_engine()
        frames = np.asarray(frames)
        if frames.ndim != 4 or frames.shape[-1] != 3 or frames.dtype != np.uint8:
            frames = np.ascontiguousarray(frames).astype(np.uint8)
        H0, W0 = frames.shape[1:3]
        H, W = network_size(H0, W0, input_size)

        import torch.distributed as dist
        from .scheduler import gathered_order, plan_windows, shard_windows
        from .stitch import stitch_stream
        dev = eng.device
        n = frames.shape[0]
        plan = plan_windows(n)
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        rank = dist.get_rank() if world > 1 else 0
        # The uint8 video lives in HBM and every frame crosses PCIe ONCE - but not all up front: the prefix a window needs is
        # uploaded on a side stream while the previous window computes (window k reads frames <= 22k + 31 only).
        frames = np.ascontiguousarray(frames)
        host = torch.from_numpy(frames)
        video = torch.empty((n, H0, W0, 3), dtype=torch.uint8, device=dev)
        compute = torch.cuda.current_stream(dev)
        upload = torch.cuda.Stream(device=dev)
        resident = [0]                                                          # frames [0, resident) are on the device

        def ensure(upto):
            upto = min(upto, n)
            if upto > resident[0]:
                with torch.cuda.stream(upload):
                    video[resident[0]:upto].copy_(host[resident[0]:upto], non_blocking=True)
                resident[0] = upto

        xin = torch.empty(1, INFER_LEN, 3, H0, W0, dtype=torch.float32, device=dev)

        def window_depth(idxs, out, prefetch_upto=0):
            """One window on the device: gather + normalise (video_depth.py:197-201), forward, resize to the source
            size (video_depth.py:207-208) into out [32,H0,W0]. `prefetch_upto`: frames the NEXT window will need."""
            ensure(max(idxs) + 1)
            compute.wait_stream(upload)
            idx = torch.tensor(idxs, dtype=torch.int32, device=dev)
            ops.gather_normalize_u8(video, idx, xin, INFER_LEN, H0, W0)
            x = xin
            if (H0, W0) != (H, W):
                # Reference: cv2.resize(INTER_CUBIC) BEFORE normalisation (util/transform.py:113). cv2 is absent offline,
                # so this leg is PARITY UNPINNED: device bicubic (a=-0.75, half-pixel centres); normalisation is affine,
                # so resizing after it is equivalent.
                x = torch.nn.functional.interpolate(xin[0], size=(H, W), mode='bicubic', align_corners=False)[None]
            depth = eng.forward(x)                                               # [1,32,H,W] fp32
            ops.bilinear_plane(depth.view(INFER_LEN, H, W), out, INFER_LEN, H, W, H0, W0)
            ensure(prefetch_upto)                                                # overlaps this window's compute
            return out

        def upto_of(k):
            return max(plan[k]) + 1 if k < len(plan) else 0

        if world == 1:
            wbuf = torch.empty(INFER_LEN, H0, W0, dtype=torch.float32, device=dev)
            windows = (window_depth(plan[k], wbuf, upto_of(k + 1)) for k in range(len(plan)))   # lazily: stitch k queues behind forward k
        else:
            # One process per GPU: this rank computes its block of windows with no data-path collective; each finished window
            # is all-gathered (RCCL over xGMI; asynchronously, under the next window's compute) so that every rank ends up
            # with all depth maps and stitches the whole sequence on its own GPU. Ranks with one window fewer send a zero slot.
            per, _ = gathered_order(len(plan), world)
            mine = shard_windows(len(plan), world, rank)
            send = torch.zeros(per, INFER_LEN, H0, W0, dtype=torch.float32, device=dev)
            recv = torch.empty(per, world, INFER_LEN, H0, W0, dtype=torch.float32, device=dev)
            pending = []
            for j in range(per):
                if j < len(mine):
                    k = mine[j]
                    window_depth(plan[k], send[j], upto_of(k + 1) if j + 1 < len(mine) else 0)
                pending.append(_all_gather(recv[j], send[j]))
            for h in pending:
                if h is not None:
                    h.wait()
            counts = [len(shard_windows(len(plan), world, r)) for r in range(world)]
            windows = (recv[j, r] for r in range(world) for j in range(counts[r]))     # window order = rank-major blocks
        depths = stitch_stream(windows, n, H0, W0, dev, metric=self.METRIC)
        return depths, target_fps


class MetricVideoDepthAnything(VideoDepthAnything):
    """metric_depth/video_depth_anything/video_depth.py: ViT-L defaults, no scale/shift alignment."""
    METRIC = True

    def __init__(self, encoder='vitl', features=256, out_channels=[256, 512, 1024, 1024], use_bn=False, use_clstoken=False,
                 num_frames=32, pe='ape'):
        super().__init__(encoder, features, out_channels, use_bn, use_clstoken, num_frames, pe)
