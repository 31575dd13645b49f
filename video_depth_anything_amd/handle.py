"""`ModelHandle`: the Python side of the handle API of include/vda.h (vda_create / vda_load_weight / vda_finalize_weights /
vda_forward). This is what `VideoDepthAnything` runs: weight packing and the launch sequence of a forward live in
libvda_hip.so (csrc/host.hip); torch owns the input / output tensors and the workspace block only.

The reference seam it stands for: `VideoDepthAnything(**cfg)` + `load_state_dict(strict=True)` + `forward(x)`
(/root/reference/video_depth_anything/video_depth.py:38-63,89-93,161-164; run.py:45-47).
"""
import ctypes as C
import json
import os

import torch

from . import _lib
from ._lib import lib
from .config import ModelConfig
from .weights import check_state_dict, state_dict_spec

F16, F32 = torch.float16, torch.float32


def _check(rc, what):
    if rc != 0:
        msg = lib.vda_last_error().decode()
        # state-dict problems keep torch's exception type (load_state_dict raises RuntimeError)
        raise RuntimeError(f"Error(s) in loading state_dict for VideoDepthAnything:\n\t{msg}") if what == "load" else _lib.VdaError(f"{what}: {msg} (rc={rc})")


class ModelHandle:
    def __init__(self, cfg: ModelConfig, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("video_depth_anything_amd needs an MI355X (HIP device); there is no CPU path")
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        c = _lib.Config(cfg.embed_dim, cfg.depth, cfg.num_heads, (C.c_int32 * 4)(*cfg.taps), cfg.features,
                        (C.c_int32 * 4)(*cfg.out_channels), cfg.num_frames, int(cfg.use_clstoken), int(cfg.use_bn),
                        int(cfg.pe == "rope"))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _check(lib.vda_create(C.byref(c), C.byref(h)), "vda_create")
        self._h = h
        self._ws = {}                            # workspace blocks by slot (a caller keeping two forwards in flight uses two)
        if os.environ.get("VDA_RESIDUAL_IN_LN") is not None:          # A/B switch for tools / bench runs
            _check(lib.vda_set_option(h, b"residual_in_ln", int(os.environ["VDA_RESIDUAL_IN_LN"])), "vda_set_option")
        if os.environ.get("VDA_DYN_SCHED") is not None:
            _check(lib.vda_set_option(h, b"dyn_sched", int(os.environ["VDA_DYN_SCHED"])), "vda_set_option")
        if os.environ.get("VDA_HEAD_OVERLAP") is not None:
            _check(lib.vda_set_option(h, b"head_overlap", int(os.environ["VDA_HEAD_OVERLAP"])), "vda_set_option")
        if os.environ.get("VDA_LN_FOLD") is not None:
            _check(lib.vda_set_option(h, b"ln_fold", int(os.environ["VDA_LN_FOLD"])), "vda_set_option")
        self.loaded = False
        assert lib.vda_num_weights(self._h) == len(state_dict_spec(cfg))

    def close(self):
        if getattr(self, "_h", None):
            with torch.cuda.device(self.device):
                torch.cuda.synchronize()
                lib.vda_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        check_state_dict(self.cfg, sd, strict)          # torch's wording for missing / unexpected / mis-shaped keys
        spec = state_dict_spec(self.cfg)
        with torch.cuda.device(self.device):
            for name in spec:
                if name not in sd:
                    continue                            # strict=False: vda_finalize_weights reports what is still missing
                t = sd[name].detach()
                if t.dtype != F32 or not t.is_contiguous():
                    t = t.to(F32).contiguous()
                dims = (C.c_int64 * t.dim())(*t.shape)
                _check(lib.vda_load_weight(self._h, name.encode(), C.c_void_p(t.data_ptr()), dims, t.dim(), _lib.DTYPE_F32), "load")
            _check(lib.vda_finalize_weights(self._h), "load")
        self.loaded = True

    # ------------------------------------------------------------------ forward
    def workspace_bytes(self, B, T, H, W, fp32=False):
        n = lib.vda_workspace_bytes(self._h, B, T, H, W, _lib.PREC_F32 if fp32 else _lib.PREC_F16)
        if n < 0:
            raise _lib.VdaError("vda_workspace_bytes: " + lib.vda_last_error().decode())
        return n

    @torch.no_grad()
    def forward(self, x, fp32: bool = False, slot: int = 0):
        """x: fp32 [B,T,3,H,W] -> depth fp32 [B,T,H,W] on the handle's device, enqueued on the current stream. `slot` picks the
        workspace block: forwards that may be in flight at the same time (on different streams) must use different slots."""
        if not self.loaded:
            raise RuntimeError("load_state_dict() first")
        if x.dim() != 5 or x.shape[2] != 3:
            raise ValueError(f"expected [B,T,3,H,W], got {tuple(x.shape)}")
        B, T, _, H, W = x.shape
        assert H % 14 == 0, f"Input image height {H} is not a multiple of patch height 14"
        assert W % 14 == 0, f"Input image width {W} is not a multiple of patch width: 14"
        if T > self.cfg.num_frames:
            raise ValueError(f"T={T} exceeds temporal_max_len={self.cfg.num_frames}")
        prec = _lib.PREC_F32 if fp32 else _lib.PREC_F16
        with torch.cuda.device(self.device):
            x = x.to(self.device, F32).contiguous()
            need = self.workspace_bytes(B, T, H, W, fp32)
            ws = self._ws.get(slot)
            if ws is None or ws.numel() < need:
                self._ws[slot] = ws = None               # release before growing
                self._ws[slot] = ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            _check(lib.vda_set_workspace(self._h, C.c_void_p(ws.data_ptr()), ws.numel()), "vda_set_workspace")
            out = torch.empty(B, T, H, W, dtype=F32, device=self.device)
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _check(lib.vda_forward(self._h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), B, T, H, W, prec, stream), "vda_forward")
            self._last = (B * T, H // 14, W // 14, fp32)
        return out

    def check(self, synchronize=True):
        """Deferred status of the forwards enqueued so far (vda_forward_status): raises VdaError if one of them left fp16's range in
        the split residual stream (its depth is NaN). forward() only enqueues work, so this is where such a failure surfaces;
        infer_video_depth calls it before it returns, a bare forward() reports at the next forward() at the latest."""
        with torch.cuda.device(self.device):
            if synchronize:
                torch.cuda.synchronize(self.device)
            _check(lib.vda_forward_status(self._h), "vda_forward_status")

    # ------------------------------------------------------------------ parity / measurement hooks
    def stage(self, name):
        """A named intermediate of the last forward as (tensor [rows, Cpad], h, w, Cpad): 'tap0'..'tap3', 'layer_1'..'layer_4',
        'path_4'..'path_1' (the names the golden fixtures use)."""
        BT, ph, pw, fp32 = self._last
        cfg = self.cfg
        pad = lambda c: (c + 63) // 64 * 64
        oc, Fe = [pad(c) for c in cfg.out_channels], cfg.features
        h4, w4 = (ph - 1) // 2 + 1, (pw - 1) // 2 + 1
        table = {"layer_1": ("l1", 4 * ph, 4 * pw, oc[0]), "layer_2": ("l2", 2 * ph, 2 * pw, oc[1]), "layer_3": ("l3t", ph, pw, oc[2]),
                 "layer_4": ("l4t", h4, w4, oc[3]), "path_4": ("p4t", ph, pw, Fe), "path_3": ("p3t", 2 * ph, 2 * pw, Fe),
                 "path_2": ("p2", 4 * ph, 4 * pw, Fe), "path_1": ("p1", 8 * ph, 8 * pw, Fe)}
        for i in range(4):
            table[f"tap{i}"] = (f"tap{i}", ph, pw, cfg.embed_dim)
        buf, h, w, Cp = table[name]
        t = torch.empty(BT * h * w, Cp, dtype=F32 if fp32 else F16, device=self.device)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            nbytes = t.numel() * t.element_size()
            if name == "path_1" and lib.vda_debug_copy(self._h, b"p1", C.c_void_p(t.data_ptr()), nbytes, stream) != 0:
                # default fp16 path: refinenet1's upsample is folded into output_conv1 (option "oc1_fused") and path_1 only exists
                # at half size ("p1c"); what that conv interpolates on the fly is vda_bilinear_nhwc_f16 of it
                from . import ops
                half = torch.empty(BT * (h // 2) * (w // 2), Cp, dtype=t.dtype, device=self.device)
                _check(lib.vda_debug_copy(self._h, b"p1c", C.c_void_p(half.data_ptr()), half.numel() * half.element_size(), stream), "vda_debug_copy")
                ops.bilinear_nhwc(half, t, BT, h // 2, w // 2, h, w, Cp)
            elif name != "path_1":
                _check(lib.vda_debug_copy(self._h, buf.encode(), C.c_void_p(t.data_ptr()), nbytes, stream), "vda_debug_copy")
        return t, h, w, Cp

    def _check_stage_exists(self, buf):
        """Raises unless the last forward's workspace holds a buffer of that name (tests)."""
        probe = torch.empty(16, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _check(lib.vda_debug_copy(self._h, buf.encode(), C.c_void_p(probe.data_ptr()), 16, stream), "vda_debug_copy")

    def set_option(self, name, value):
        _check(lib.vda_set_option(self._h, name.encode(), int(value)), "vda_set_option")

    def profile_start(self, every=4):
        _check(lib.vda_profile_start(self._h, every), "vda_profile_start")

    def profile_stop(self):
        buf = C.create_string_buffer(1 << 16)
        with torch.cuda.device(self.device):
            _check(lib.vda_profile_stop(self._h, buf, len(buf)), "vda_profile_stop")
        return json.loads(buf.value.decode())
