"""Host-side orchestration of one forward pass over the HIP kernels - a TEST INSTRUMENT, not the product path.

What ships is `vda_forward` (csrc/host.hip) behind `handle.ModelHandle`. This file is the same launch sequence written in Python
over the per-kernel ABI, kept as the bit-exact cross-check of the handle IN ITS `ln_fold = 0` FORM ONLY (standalone LayerNorms,
fp32 residual stream): it does not mirror the default LayerNorm-folded split stream, and it refuses `use_bn` / `pe='rope'`
(`VideoDepthAnything.python_engine`). tests/test_forward_gpu.py::test_handle_and_python_orchestration_are_bit_identical is its user.

This is the Python mirror of `VideoDepthAnything.forward`
(/root/reference/video_depth_anything/video_depth.py:89-93,161-164): DINOv2 encoder taps
(dinov2.py:271-321) -> DPTHeadTemporal (dpt_temporal.py:53-114) -> [B,T,H,W] depth.
Every arithmetic step is a call into libvda_hip.so through `ops`; torch only owns the
device buffers. There is no CPU or torch-op fallback.

Data layout in HBM (all activations token-major / NHWC so ViT tokens feed the head's 1x1
convs with no transpose):
  tok   fp32 [BT*(P+1), D]   residual stream of the encoder (as under the reference's autocast)
  xn    fp16 [BT*(P+1), D]   LayerNorm output = A operand of the next GEMM
  qkv   fp16 [BT*(P+1), 3D]  attention input, read in place with strides (no head split copy)
  taps  fp16 [BT*P, D] x4    final-norm'd patch tokens, cls dropped by the norm kernel
  head  fp16 NHWC [BT, h, w, Cpad]; channel counts padded to multiples of 64 at pack time
        (ViT-S: 48->64, 96->128) so every GEMM K is a multiple of the 64-wide K step; output_conv1 (F/2 wide) feeds only
        the depth tail, whose passes are 32 channels wide: padded to 32
  temporal residual stream fp32 [BT*hw, C]
Precision map (default, the reference's autocast path): fp16 operands, fp32 accumulate; LayerNorm/GroupNorm/softmax
statistics fp32; encoder and temporal residual streams fp32; final 32->1 projection reads fp16, writes fp32.
`forward(x, fp32=True)` (the reference's fp32=True, video_depth.py:203-205): every buffer above is fp32 and every product
runs on fp32-input MFMA (exact fp32) through the *_f32 entry points; same launch sequence, the depth tail unfused.

This Python orchestration is the cross-check of the C++ one behind `vda_forward` (csrc/host.hip), which is what
`VideoDepthAnything` runs: both issue the same launches on the same layouts, so their outputs are bit-identical
(tests/test_forward_gpu.py::test_handle_and_python_orchestration_are_bit_identical). It keeps the standalone-LayerNorm form
(`ln_fold` = 0): the handle's default path (LayerNorm folded into the GEMMs) is cross-checked against the oracle and the goldens
only - the comparison that matters. It also exposes every intermediate stage for the golden tests.

Algebraic rewrite (exact in real arithmetic): FeatureFusionBlock's `out_conv(bilinear(x))`
(util/blocks.py:156-160) runs as `bilinear(out_conv(x))`: a 1x1 conv and an align_corners
bilinear resize commute (interpolation weights sum to 1, so the bias commutes too), which
cuts the 1x1 conv's FLOPs 4x.
"""
from typing import Dict, List

import torch

from . import _lib, ops
from .config import ENC_LN_EPS, GN_EPS, GN_GROUPS, PATCH, POS_GRID, TEMPORAL_HEADS, TMP_LN_EPS, ModelConfig
from .weights import check_state_dict, temporal_channels

F16, F32 = torch.float16, torch.float32
KPATCH = 640   # 3*14*14 = 588 padded to the K step


def _pad(c):
    return ops.pad_to(c, 64)


class Engine:
    def __init__(self, cfg: ModelConfig, device="cuda"):
        if not torch.cuda.is_available():
            raise RuntimeError("video_depth_anything_amd needs an MI355X (HIP device); there is no CPU path")
        self.cfg = cfg
        self.device = torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}        # the pack of the precision in use (set by forward)
        self.act = F16                              # activation / operand dtype in use
        self._packs: Dict[torch.dtype, Dict[str, torch.Tensor]] = {}
        self._buf: Dict[tuple, torch.Tensor] = {}
        self._pos_cache: Dict[tuple, torch.Tensor] = {}
        self.loaded = False
        self.residual_in_ln = False                 # A/B option of the fp16 path (see _forward); off: measured slower end to end

    # ------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict=True):
        check_state_dict(self.cfg, sd, strict)
        self._sd = sd
        self._packs.clear()
        self._pos_cache.clear()
        with torch.cuda.device(self.device):
            self._pos_embed_dev = sd["pretrained.pos_embed"].detach().to(self.device, F32).reshape(-1, self.cfg.embed_dim).contiguous()
            self.w = self._pack(F16)
        self.loaded = True

    def _pack(self, dt):
        """Kernel layouts of every weight for operand dtype `dt` (fp16, or fp32 on the first fp32 forward)."""
        if dt in self._packs:
            return self._packs[dt]
        cfg, sd = self.cfg, self._sd
        dv = self.device
        w: Dict[str, torch.Tensor] = {}

        def f32(name):
            return sd[name].detach().to(dv, F32).contiguous()

        def lin(name, n_pad=None, k_pad=None):
            t = sd[name].detach().to(dv, F32)
            return ops.pack_linear(t.reshape(t.shape[0], -1), n_pad, k_pad, dtype=dt)

        def conv(name, cout_pad=None, cin_pad=None):
            return ops.pack_conv3x3(sd[name].detach().to(dv, F32), cout_pad, cin_pad, dtype=dt)

        def padvec(name, n):
            v = torch.zeros(n, dtype=F32, device=dv)
            src = sd[name].detach().to(dv, F32)
            v[:src.numel()] = src
            return v

        D, Fe, oc = cfg.embed_dim, cfg.features, cfg.out_channels
        ocp = [_pad(c) for c in oc]
        Fh, Fhp = Fe // 2, ops.pad_to(Fe // 2, 32)     # output_conv1's width: only the depth tail (32-channel passes) consumes it
        self.ocp, self.Fhp = ocp, Fhp
        p = "pretrained."
        w["patch.w"] = lin(p + "patch_embed.proj.weight", k_pad=KPATCH)
        w["patch.b"] = f32(p + "patch_embed.proj.bias")
        w["cls"] = f32(p + "cls_token").reshape(-1)
        for i in range(cfg.depth):
            b = f"{p}blocks.{i}."
            for n in ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.qkv.bias", "attn.proj.bias",
                      "mlp.fc1.bias", "mlp.fc2.bias", "ls1.gamma", "ls2.gamma"):
                w[f"b{i}.{n}"] = f32(b + n)
            for n in ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight"):
                w[f"b{i}.{n}"] = lin(b + n)
        w["norm.w"], w["norm.b"] = f32(p + "norm.weight"), f32(p + "norm.bias")

        h = "head."
        if cfg.use_clstoken:                                     # dpt.py:92-98: Linear(2D -> D) + GELU per tap
            for i in range(4):
                w[f"readout{i}.w"] = lin(f"{h}readout_projects.{i}.0.weight")
                w[f"readout{i}.b"] = f32(f"{h}readout_projects.{i}.0.bias")
        for i in range(4):
            w[f"proj{i}.w"] = lin(f"{h}projects.{i}.weight", n_pad=ocp[i])
            w[f"proj{i}.b"] = padvec(f"{h}projects.{i}.bias", ocp[i])
        for i in (0, 1):
            w[f"resize{i}.w"], w[f"resize{i}.b"] = ops.pack_convt(
                sd[f"{h}resize_layers.{i}.weight"].detach().to(dv, F32), sd[f"{h}resize_layers.{i}.bias"].detach().to(dv, F32), ocp[i], dtype=dt)
        w["resize3.w"] = conv(h + "resize_layers.3.weight", ocp[3], ocp[3])
        w["resize3.b"] = padvec(h + "resize_layers.3.bias", ocp[3])
        sc = h + "scratch."
        for i in range(4):
            w[f"rn{i + 1}.w"] = conv(f"{sc}layer{i + 1}_rn.weight", Fe, ocp[i])
        for i in (1, 2, 3, 4):
            r = f"{sc}refinenet{i}."
            w[f"ref{i}.out.w"], w[f"ref{i}.out.b"] = lin(r + "out_conv.weight"), f32(r + "out_conv.bias")
            for u in (1, 2):
                for c in (1, 2):
                    w[f"ref{i}.rcu{u}.c{c}.w"] = conv(f"{r}resConfUnit{u}.conv{c}.weight")
                    w[f"ref{i}.rcu{u}.c{c}.b"] = f32(f"{r}resConfUnit{u}.conv{c}.bias")
        w["oc1.w"] = conv(sc + "output_conv1.weight", Fhp, Fe)
        w["oc1.b"] = padvec(sc + "output_conv1.bias", Fhp)
        w["oc2.w"] = conv(sc + "output_conv2.0.weight", 32, Fhp)
        w["oc2.b"] = f32(sc + "output_conv2.0.bias")
        w["oc3.w"] = f32(sc + "output_conv2.2.weight").reshape(-1)
        self.oc3_bias = float(sd[sc + "output_conv2.2.bias"].reshape(-1)[0])
        for m, Cc in enumerate(temporal_channels(cfg)):
            t = f"{h}motion_modules.{m}.temporal_transformer."
            k = f"tm{m}."
            w[k + "gn.w"], w[k + "gn.b"] = f32(t + "norm.weight"), f32(t + "norm.bias")
            w[k + "in.w"], w[k + "in.b"] = lin(t + "proj_in.weight"), f32(t + "proj_in.bias")
            w[k + "out.w"], w[k + "out.b"] = lin(t + "proj_out.weight"), f32(t + "proj_out.bias")
            tb = t + "transformer_blocks.0."
            for a in (0, 1):
                ab = f"{tb}attention_blocks.{a}."
                qkv = torch.cat([sd[ab + "to_q.weight"], sd[ab + "to_k.weight"], sd[ab + "to_v.weight"]], dim=0)
                w[f"{k}a{a}.qkv.w"] = ops.pack_linear(qkv.detach().to(dv, F32), dtype=dt)
                w[f"{k}a{a}.out.w"], w[f"{k}a{a}.out.b"] = lin(ab + "to_out.0.weight"), f32(ab + "to_out.0.bias")
                w[f"{k}a{a}.pe"] = f32(ab + "pos_encoder.pe").reshape(-1, Cc)
                w[f"{k}a{a}.ln.w"], w[f"{k}a{a}.ln.b"] = f32(f"{tb}norms.{a}.weight"), f32(f"{tb}norms.{a}.bias")
            w[k + "ffln.w"], w[k + "ffln.b"] = f32(tb + "ff_norm.weight"), f32(tb + "ff_norm.bias")
            gw, gb = ops.pack_geglu(sd[tb + "ff.net.0.proj.weight"].detach().to(dv, F32), sd[tb + "ff.net.0.proj.bias"].detach().to(dv, F32), dtype=dt)
            w[k + "ff1.w"], w[k + "ff1.b"] = gw, gb
            w[k + "ff2.w"], w[k + "ff2.b"] = lin(tb + "ff.net.2.weight"), f32(tb + "ff.net.2.bias")
        self._packs[dt] = w
        return w

    # ------------------------------------------------------------------ helpers
    def buf(self, name, shape, dtype, zero=False):
        """Named workspace, allocated once per shape (no allocation in steady state)."""
        key = (name, dtype)
        t = self._buf.get(key)
        n = 1
        for s in shape:
            n *= s
        if t is None or t.dtype != dtype or t.numel() < n:
            t = (torch.zeros if zero else torch.empty)(n, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t[:n].view(*shape)

    def pos_embed(self, H, W):
        """dinov2.py:179-210. The stored grid when the input has 37 x 37 patches and is square; otherwise the grid is
        resampled on the device (vda_pos_embed_resample_f32: the reference's bicubic with its 0.1 offset), once per shape."""
        key = (H, W)
        t = self._pos_cache.get(key)
        if t is not None:
            return t
        pe = self._pos_embed_dev
        n = pe.shape[0] - 1
        ph, pw = H // PATCH, W // PATCH
        if ph * pw == n and H == W:
            t = pe
        else:
            t = torch.empty(1 + ph * pw, pe.shape[1], dtype=F32, device=self.device)
            ops.pos_embed_resample(pe, t, POS_GRID, ph, pw, pe.shape[1])
        self._pos_cache[key] = t
        return t

    def conv3x3(self, x, wname, out, B, H, W, Cin, Cout, epi, stride=1, bias=None, relu_in=False, res=None, res2=None):
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        ops.gemm(x, self.w[wname], out, epi, M=B * Ho * Wo, N=Cout, K=9 * Cin, bias=bias, res=res, res2=res2,
                 relu_in=relu_in, conv=(B, H, W, Cin, Ho, Wo, stride))
        return Ho, Wo

    # ------------------------------------------------------------------ temporal module
    def temporal(self, m, x, B, T, hw, Cc, tag):
        """motion_module.py:102-126,164-177 on NHWC fp16 x [B*T, hw, C]; returns a new fp16 tensor."""
        w, k = self.w, f"tm{m}."
        BT = B * T
        rows = BT * hw
        chunks = max(1, min(16, (hw + 31) // 32))
        part = self.buf("gn_partial", (BT * chunks * GN_GROUPS * 2,), F32)
        g = self.buf("tm_g", (rows, Cc), self.act)
        ops.groupnorm(x, g, w[k + "gn.w"], w[k + "gn.b"], GN_EPS, BT, hw, Cc, GN_GROUPS, part, chunks)
        hs = self.buf("tm_hs", (rows, Cc), F32)
        ops.gemm(g, w[k + "in.w"], hs, _lib.EPI_BIAS_F32, M=rows, N=Cc, K=Cc, bias=w[k + "in.b"])
        n = self.buf("tm_n", (rows, Cc), self.act)
        qkv = self.buf("tm_qkv", (rows, 3 * Cc), self.act)
        ao = self.buf("tm_ao", (rows, Cc), self.act)
        for a in (0, 1):
            ops.layernorm(hs, n, w[f"{k}a{a}.ln.w"], w[f"{k}a{a}.ln.b"], TMP_LN_EPS, rows, Cc,
                          pe=w[f"{k}a{a}.pe"], pe_rows_per_step=hw, pe_steps=T)
            ops.gemm(n, w[f"{k}a{a}.qkv.w"], qkv, _lib.EPI_BIAS_F16, M=rows, N=3 * Cc, K=Cc)
            for b in range(B):
                r0 = b * T * hw
                ops.temporal_attention(qkv[r0:r0 + T * hw], ao[r0:r0 + T * hw], T, hw, Cc, TEMPORAL_HEADS)
            ops.gemm(ao, w[f"{k}a{a}.out.w"], hs, _lib.EPI_SCALE_RES_F32, M=rows, N=Cc, K=Cc, bias=w[f"{k}a{a}.out.b"], res=hs)
        ops.layernorm(hs, n, w[k + "ffln.w"], w[k + "ffln.b"], TMP_LN_EPS, rows, Cc)
        gg = self.buf("tm_gg", (rows, 4 * Cc), self.act)
        ops.gemm(n, w[k + "ff1.w"], gg, _lib.EPI_GEGLU_F16, M=rows, N=8 * Cc, K=Cc, ldc=4 * Cc, bias=w[k + "ff1.b"])
        hh = self.buf("tm_hh", (rows, Cc), self.act)
        ops.gemm(gg, w[k + "ff2.w"], hh, _lib.EPI_SCALE_RES_F32_H, M=rows, N=Cc, K=4 * Cc, bias=w[k + "ff2.b"], res=hs)
        out = self.buf(tag, (rows, Cc), self.act)
        ops.gemm(hh, w[k + "out.w"], out, _lib.EPI_RES_F16, M=rows, N=Cc, K=Cc, bias=w[k + "out.b"], res=x)
        return out

    # ------------------------------------------------------------------ fusion block
    def rcu(self, i, u, x, out, B, H, W, Fe, res2=None):
        """util/blocks.py:68-91: conv2(relu(conv1(relu(x)))) + x (+ res2 fused for the block's skip add)."""
        w = self.w
        y = self.buf("rcu_y", (B * H * W, Fe), self.act)
        self.conv3x3(x, f"ref{i}.rcu{u}.c1.w", y, B, H, W, Fe, Fe, _lib.EPI_BIAS_RELU_F16, bias=w[f"ref{i}.rcu{u}.c1.b"], relu_in=True)
        self.conv3x3(y, f"ref{i}.rcu{u}.c2.w", out, B, H, W, Fe, Fe, _lib.EPI_RES_F16, bias=w[f"ref{i}.rcu{u}.c2.b"], res=x, res2=res2)

    def fusion(self, i, x0, x1, B, H, W, Ho, Wo, Fe, tag, upsample=True):
        """util/blocks.py:135-162 with out_conv moved in front of the resize (upsample=False: the caller's conv resizes itself)."""
        w = self.w
        rows = B * H * W
        s = x0
        if x1 is not None:
            s = self.buf("fus_s", (rows, Fe), self.act)
            self.rcu(i, 1, x1, s, B, H, W, Fe, res2=x0)
        r = self.buf("fus_r", (rows, Fe), self.act)
        self.rcu(i, 2, s, r, B, H, W, Fe)
        c = self.buf("fus_c" if upsample else tag, (rows, Fe), self.act)
        ops.gemm(r, w[f"ref{i}.out.w"], c, _lib.EPI_BIAS_F16, M=rows, N=Fe, K=Fe, bias=w[f"ref{i}.out.b"])
        if not upsample:
            return c
        out = self.buf(tag, (B * Ho * Wo, Fe), self.act)
        ops.bilinear_nhwc(c, out, B, H, W, Ho, Wo, Fe)
        return out

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x, taps_out: List[torch.Tensor] = None, stages: dict = None, fp32: bool = False):
        """x: fp32 cuda [B,T,3,H,W] -> depth fp32 [B,T,H,W]. fp32=True: fp32 operands and activations throughout."""
        if not self.loaded:
            raise RuntimeError("load_state_dict() first")
        with torch.cuda.device(self.device):
            return self._forward(x, taps_out, stages, fp32)

    def _forward(self, x, taps_out, stages, fp32):
        self.act = F32 if fp32 else F16
        self.w = self._pack(self.act)
        cfg, w = self.cfg, self.w
        if x.dim() != 5 or x.shape[2] != 3:
            raise ValueError(f"expected [B,T,3,H,W], got {tuple(x.shape)}")
        B, T, _, H, W = x.shape
        assert H % PATCH == 0, f"Input image height {H} is not a multiple of patch height {PATCH}"
        assert W % PATCH == 0, f"Input image width {W} is not a multiple of patch width: {PATCH}"
        if T > cfg.num_frames:
            raise ValueError(f"T={T} exceeds temporal_max_len={cfg.num_frames}")
        x = x.to(self.device, F32).contiguous()
        BT, ph, pw = B * T, H // PATCH, W // PATCH
        P, D, NH = ph * pw, cfg.embed_dim, cfg.num_heads
        Nt = P + 1
        rows = BT * Nt

        # ---- encoder (dinov2.py:212-219, block.py:105-106)
        a0 = self.buf("a0", (BT * P, KPATCH), self.act, zero=True)      # pad columns stay zero for ever
        ops.patchify(x, a0, BT, H, W, KPATCH)
        tok = self.buf("tok", (rows, D), F32)
        pos = self.pos_embed(H, W)
        ops.gemm(a0, w["patch.w"], tok, _lib.EPI_PATCH_F32, M=BT * P, N=D, K=KPATCH, bias=w["patch.b"], pos=pos, P=P)
        ops.cls_rows(tok, w["cls"], pos, BT, P, D)
        xn = self.buf("xn", (rows, D), self.act)
        qkv = self.buf("qkv", (rows, 3 * D), self.act)
        ao = self.buf("ao", (rows, D), self.act)
        hid = self.buf("hid", (rows, 4 * D), self.act)
        taps = []
        # Residual add of attn.proj / mlp.fc2: the GEMM's own fp32 in-place epilogue by default; with residual_in_ln (fp16 path,
        # A/B option) the projection stores fp16 y and x += gamma * y rides on the LayerNorm that follows
        # (vda_layernorm_residual_f32_f16) - faster GEMMs, slower LayerNorms, +1.1 ms per ViT-L clip net (csrc/host.hip).
        defer = (not fp32) and self.residual_in_ln
        yb = self.buf("ybuf", (rows, D), F16) if defer else None
        xn_ready = False
        for i in range(cfg.depth):
            k = f"b{i}."
            if not xn_ready:
                ops.layernorm(tok, xn, w[k + "norm1.weight"], w[k + "norm1.bias"], ENC_LN_EPS, rows, D)
            xn_ready = False
            ops.gemm(xn, w[k + "attn.qkv.weight"], qkv, _lib.EPI_BIAS_F16, M=rows, N=3 * D, K=D, bias=w[k + "attn.qkv.bias"])
            ops.attention(qkv, ao, BT, Nt, NH)
            if defer:
                ops.gemm(ao, w[k + "attn.proj.weight"], yb, _lib.EPI_BIAS_F16, M=rows, N=D, K=D, bias=w[k + "attn.proj.bias"])
                ops.layernorm_residual(tok, yb, w[k + "ls1.gamma"], xn, w[k + "norm2.weight"], w[k + "norm2.bias"], ENC_LN_EPS, rows, D)
            else:
                ops.gemm(ao, w[k + "attn.proj.weight"], tok, _lib.EPI_SCALE_RES_F32, M=rows, N=D, K=D, bias=w[k + "attn.proj.bias"],
                         gamma=w[k + "ls1.gamma"], res=tok)
                ops.layernorm(tok, xn, w[k + "norm2.weight"], w[k + "norm2.bias"], ENC_LN_EPS, rows, D)
            ops.gemm(xn, w[k + "mlp.fc1.weight"], hid, _lib.EPI_BIAS_GELU_F16, M=rows, N=4 * D, K=D, bias=w[k + "mlp.fc1.bias"])
            is_tap, last = i in cfg.taps, i + 1 == cfg.depth
            tp = self.buf(f"tap{len(taps)}", (BT * P, D), self.act) if is_tap else None
            if defer:
                ops.gemm(hid, w[k + "mlp.fc2.weight"], yb, _lib.EPI_BIAS_F16, M=rows, N=D, K=4 * D, bias=w[k + "mlp.fc2.bias"])
                if last and tp is not None:
                    ops.layernorm_residual(tok, yb, w[k + "ls2.gamma"], tp, w["norm.w"], w["norm.b"], ENC_LN_EPS, rows, D, group=Nt, skip=1)
                else:
                    kn = f"b{i if last else i + 1}."
                    ops.layernorm_residual(tok, yb, w[k + "ls2.gamma"], xn, w[kn + "norm1.weight"], w[kn + "norm1.bias"], ENC_LN_EPS, rows, D)
                    xn_ready = True
                    if tp is not None:
                        ops.layernorm(tok, tp, w["norm.w"], w["norm.b"], ENC_LN_EPS, rows, D, group=Nt, skip=1)
            else:
                ops.gemm(hid, w[k + "mlp.fc2.weight"], tok, _lib.EPI_SCALE_RES_F32, M=rows, N=D, K=4 * D, bias=w[k + "mlp.fc2.bias"],
                         gamma=w[k + "ls2.gamma"], res=tok)
                if tp is not None:
                    ops.layernorm(tok, tp, w["norm.w"], w["norm.b"], ENC_LN_EPS, rows, D, group=Nt, skip=1)
            if tp is not None:
                if cfg.use_clstoken:
                    # dpt_temporal.py:56-59: the tap becomes GELU(Linear([patch token, cls])); the final norm above dropped the cls
                    # row, so norm the whole token matrix again (cls kept) and gather [patch | cls] rows for one K = 2D GEMM
                    full = self.buf("rd_full", (rows, D), self.act)
                    ops.layernorm(tok, full, w["norm.w"], w["norm.b"], ENC_LN_EPS, rows, D)
                    cat = self.buf("rd_cat", (BT * P, 2 * D), self.act)
                    ops.readout_concat(full, cat, BT, P, D)
                    j = len(taps)
                    ops.gemm(cat, w[f"readout{j}.w"], tp, _lib.EPI_BIAS_GELU_F16, M=BT * P, N=D, K=2 * D, bias=w[f"readout{j}.b"])
                taps.append(tp)
        if taps_out is not None:
            taps_out.extend(taps)

        # ---- head: reassemble (dpt_temporal.py:55-69)
        ocp, Fe, Fhp = self.ocp, cfg.features, self.Fhp
        h1, w1, h2, w2 = 4 * ph, 4 * pw, 2 * ph, 2 * pw
        h4, w4 = (ph - 1) // 2 + 1, (pw - 1) // 2 + 1
        t0 = self.buf("t0", (BT * P, ocp[0]), self.act)
        ops.gemm(taps[0], w["proj0.w"], t0, _lib.EPI_BIAS_F16, M=BT * P, N=ocp[0], K=D, bias=w["proj0.b"])
        l1 = self.buf("l1", (BT * h1 * w1, ocp[0]), self.act)
        ops.gemm(t0, w["resize0.w"], l1, _lib.EPI_CONVT_F16, M=BT * P, N=16 * ocp[0], K=ocp[0], ldc=ocp[0], bias=w["resize0.b"],
                 convt=(4, ph, pw, ocp[0]))
        t1 = self.buf("t1", (BT * P, ocp[1]), self.act)
        ops.gemm(taps[1], w["proj1.w"], t1, _lib.EPI_BIAS_F16, M=BT * P, N=ocp[1], K=D, bias=w["proj1.b"])
        l2 = self.buf("l2", (BT * h2 * w2, ocp[1]), self.act)
        ops.gemm(t1, w["resize1.w"], l2, _lib.EPI_CONVT_F16, M=BT * P, N=4 * ocp[1], K=ocp[1], ldc=ocp[1], bias=w["resize1.b"],
                 convt=(2, ph, pw, ocp[1]))
        l3 = self.buf("l3", (BT * P, ocp[2]), self.act)
        ops.gemm(taps[2], w["proj2.w"], l3, _lib.EPI_BIAS_F16, M=BT * P, N=ocp[2], K=D, bias=w["proj2.b"])
        t3 = self.buf("t3", (BT * P, ocp[3]), self.act)
        ops.gemm(taps[3], w["proj3.w"], t3, _lib.EPI_BIAS_F16, M=BT * P, N=ocp[3], K=D, bias=w["proj3.b"])
        l4 = self.buf("l4", (BT * h4 * w4, ocp[3]), self.act)
        self.conv3x3(t3, "resize3.w", l4, BT, ph, pw, ocp[3], ocp[3], _lib.EPI_BIAS_F16, stride=2, bias=w["resize3.b"])

        # ---- temporal modules on layer_3 / layer_4 (dpt_temporal.py:75-76)
        l3 = self.temporal(0, l3, B, T, P, ocp[2], "l3t")
        l4 = self.temporal(1, l4, B, T, h4 * w4, ocp[3], "l4t")

        # ---- layer_rn (no bias) and the fusion pyramid (dpt_temporal.py:78-91)
        l1r = self.buf("l1r", (BT * h1 * w1, Fe), self.act)
        self.conv3x3(l1, "rn1.w", l1r, BT, h1, w1, ocp[0], Fe, _lib.EPI_BIAS_F16)
        l2r = self.buf("l2r", (BT * h2 * w2, Fe), self.act)
        self.conv3x3(l2, "rn2.w", l2r, BT, h2, w2, ocp[1], Fe, _lib.EPI_BIAS_F16)
        l3r = self.buf("l3r", (BT * P, Fe), self.act)
        self.conv3x3(l3, "rn3.w", l3r, BT, ph, pw, ocp[2], Fe, _lib.EPI_BIAS_F16)
        l4r = self.buf("l4r", (BT * h4 * w4, Fe), self.act)
        self.conv3x3(l4, "rn4.w", l4r, BT, h4, w4, ocp[3], Fe, _lib.EPI_BIAS_F16)

        p4 = self.fusion(4, l4r, None, BT, h4, w4, ph, pw, Fe, "p4")
        p4 = self.temporal(2, p4, B, T, P, Fe, "p4t")
        p3 = self.fusion(3, p4, l3r, BT, ph, pw, h2, w2, Fe, "p3")
        p3 = self.temporal(3, p3, B, T, h2 * w2, Fe, "p3t")
        p2 = self.fusion(2, p3, l2r, BT, h2, w2, h1, w1, Fe, "p2")
        # fp16 path: refinenet1's 2x upsample is folded into output_conv1 (vda_conv3x3_up2_f16), as csrc/host.hip does by default
        up_fused = not fp32 and Fhp <= 128
        p1 = self.fusion(1, p2, l1r, BT, h1, w1, 2 * h1, 2 * w1, Fe, "p1c" if up_fused else "p1", upsample=not up_fused)

        # ---- output convs (dpt.py:117-124, dpt_temporal.py:93-100)
        hh, ww = 2 * h1, 2 * w1
        o1 = self.buf("o1", (BT * hh * ww, Fhp), self.act)
        if up_fused:
            ops.conv3x3_up2(p1, w["oc1.w"], w["oc1.b"], o1, BT, h1, w1, Fe, Fhp, Fhp)
        else:
            self.conv3x3(p1, "oc1.w", o1, BT, hh, ww, Fe, Fhp, _lib.EPI_BIAS_F16, bias=w["oc1.b"])
        # bilinear to (H,W) + output_conv2 (3x3 -> ReLU -> 1x1 -> ReLU) in one kernel: the 518^2 x F/2 upsampled tensor is
        # never materialised (dpt_temporal.py:94-100)
        depth = torch.empty(B, T, H, W, dtype=F32, device=self.device)
        if not fp32:
            ops.depth_tail(o1, w["oc2.w"], w["oc2.b"], w["oc3.w"], self.oc3_bias, depth, BT, hh, ww, H, W, Fhp)
        else:
            # fp32 operands: the same three steps unfused (speed is secondary on this path)
            up = self.buf("tail_up", (BT * H * W, Fhp), F32)
            ops.bilinear_nhwc(o1, up, BT, hh, ww, H, W, Fhp)
            c2 = self.buf("tail_c2", (BT * H * W, 32), F32)
            self.conv3x3(up, "oc2.w", c2, BT, H, W, Fhp, 32, _lib.EPI_BIAS_RELU_F16, bias=w["oc2.b"])
            ops.head_out(c2, w["oc3.w"], self.oc3_bias, depth, BT * H * W, 32)
        # video_depth.py:162-163: bilinear to (H,W) is the identity here (H == 14*ph) and the ReLU is idempotent.
        if stages is not None:
            stages.update(layer_1=(l1, h1, w1, ocp[0]), layer_2=(l2, h2, w2, ocp[1]), layer_3=(l3, ph, pw, ocp[2]),
                          layer_4=(l4, h4, w4, ocp[3]), path_4=(p4, ph, pw, Fe), path_3=(p3, h2, w2, Fe),
                          path_2=(p2, h1, w1, Fe))
            # (fused path: refinenet1's output before its 2x upsample)
            stages.update(path_1_half=(p1, h1, w1, Fe)) if up_fused else stages.update(path_1=(p1, hh, ww, Fe))
        return depth
