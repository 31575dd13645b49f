"""ORACLE — TEST INFRASTRUCTURE ONLY. Not part of the product path.

A CPU fp32 restatement (torch CPU tensor ops, numpy for the stitcher) of the
reference's `VideoDepthAnything.forward` / `infer_video_depth` algorithm, written
from the reference's behaviour, operating on the flat checkpoint state dict.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module; the product (`video_depth_anything_amd`) never does and
fails loudly when its HIP library is missing.

Pinning: the reference holds no tests or golden vectors for this path
(SURVEY.md §4), so the oracle is pinned by outputs of the reference's own
modules run in the build container: `oracle/gen_golden.py` imports
/root/reference, loads the same seeded state dict, and writes the fixtures in
`tests/golden/`; `tests/test_oracle_golden.py` holds this file to them at 1e-5.
Preprocessing for frames that are NOT already at network size uses
`cv2.resize(INTER_CUBIC)` in the reference; cv2 is absent here, so that leg is
"parity unpinned" (oracle and product both restrict themselves to what is
documented at `resize_cubic`).

Every function cites the reference file:line it follows (paths relative to
/root/reference/).
"""
import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

PATCH = 14
INFER_LEN = 32
OVERLAP = 10
KEYFRAMES = [0, 12, 24, 25, 26, 27, 28, 29, 30, 31]
INTERP_LEN = 8


# --------------------------------------------------------------------------
# Encoder: DINOv2 ViT
# --------------------------------------------------------------------------
def pos_embed_for(sd: SD, h: int, w: int) -> Tensor:
    """video_depth_anything/dinov2.py:179-210 interpolate_pos_encoding.

    Note the reference names the image dims (w, h) = x.shape[2:], i.e. its `w`
    is the image HEIGHT; the stored grid is resampled with
    scale_factor = ((H/14 + 0.1)/37, (W/14 + 0.1)/37), bicubic, no antialias.
    """
    pe = sd["pretrained.pos_embed"]
    n = pe.shape[1] - 1
    ph, pw = h // PATCH, w // PATCH
    if ph * pw == n and h == w:
        return pe
    pe = pe.float()
    g = int(math.sqrt(n))
    dim = pe.shape[-1]
    sy, sx = float(ph + 0.1) / math.sqrt(n), float(pw + 0.1) / math.sqrt(n)
    grid = pe[:, 1:].reshape(1, g, g, dim).permute(0, 3, 1, 2)
    grid = F.interpolate(grid, scale_factor=(sy, sx), mode="bicubic", antialias=False)
    assert grid.shape[-2] == ph and grid.shape[-1] == pw
    grid = grid.permute(0, 2, 3, 1).reshape(1, -1, dim)
    return torch.cat((pe[:, :1], grid), dim=1)


def patch_tokens(sd: SD, x: Tensor) -> Tensor:
    """dinov2_layers/patch_embed.py:69-82 + dinov2.py:212-219: 14x14/s14 conv,
    flatten row-major, prepend cls, add pos-embed."""
    _, _, H, W = x.shape
    assert H % PATCH == 0 and W % PATCH == 0
    t = F.conv2d(x, sd["pretrained.patch_embed.proj.weight"], sd["pretrained.patch_embed.proj.bias"], stride=PATCH)
    t = t.flatten(2).transpose(1, 2)
    cls = sd["pretrained.cls_token"].expand(t.shape[0], -1, -1)
    t = torch.cat((cls, t), dim=1)
    return t + pos_embed_for(sd, H, W)


def vit_attention(sd: SD, pre: str, x: Tensor, num_heads: int) -> Tensor:
    """dinov2_layers/attention.py:49-62."""
    B, N, C = x.shape
    hd = C // num_heads
    qkv = F.linear(x, sd[pre + "qkv.weight"], sd[pre + "qkv.bias"])
    qkv = qkv.reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * hd ** -0.5, qkv[1], qkv[2]
    a = (q @ k.transpose(-2, -1)).softmax(dim=-1)
    o = (a @ v).transpose(1, 2).reshape(B, N, C)
    return F.linear(o, sd[pre + "proj.weight"], sd[pre + "proj.bias"])


def vit_block(sd: SD, i: int, x: Tensor, num_heads: int) -> Tensor:
    """dinov2_layers/block.py:105-106 (eval path), mlp.py:35-41, layer_scale.py:27-28."""
    p = f"pretrained.blocks.{i}."
    D = x.shape[-1]
    y = F.layer_norm(x, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
    x = x + vit_attention(sd, p + "attn.", y, num_heads) * sd[p + "ls1.gamma"]
    y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    y = F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"])
    y = F.gelu(y)
    y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y * sd[p + "ls2.gamma"]


def encoder_taps(sd: SD, cfg, x: Tensor) -> List[Tensor]:
    """dinov2.py:271-281,297-321 get_intermediate_layers(norm=True, return_class_token=True): final LayerNorm on every tap;
    the patch tokens, the cls token dropped (use_clstoken=False: the head ignores it, dpt_temporal.py:61) or folded in by the
    head's readout projection (use_clstoken=True, dpt_temporal.py:56-59 / dpt.py:92-98,129-132):
    x = GELU(Linear_{2D->D}(cat(patch_tokens, cls broadcast over the patches)))."""
    t = patch_tokens(sd, x)
    D = t.shape[-1]
    taps = []
    for i in range(cfg.depth):
        t = vit_block(sd, i, t, cfg.num_heads)
        if i in cfg.taps:
            n = F.layer_norm(t, (D,), sd["pretrained.norm.weight"], sd["pretrained.norm.bias"], 1e-6)
            tok, cls = n[:, 1:], n[:, 0]
            if getattr(cfg, "use_clstoken", False):
                k = f"head.readout_projects.{len(taps)}.0."
                readout = cls.unsqueeze(1).expand_as(tok)
                tok = F.gelu(F.linear(torch.cat((tok, readout), -1), sd[k + "weight"], sd[k + "bias"]))
            taps.append(tok)
    return taps


# --------------------------------------------------------------------------
# Temporal module (motion_module)
# --------------------------------------------------------------------------
def temporal_attention(sd: SD, pre: str, h: Tensor, T: int, heads: int = 8) -> Tensor:
    """motion_module/motion_module.py:230-297 + motion_module/attention.py:182-211.
    h: [(b f), d, c] normed hidden states -> same shape."""
    BT, d, C = h.shape
    b = BT // T
    x = h.reshape(b, T, d, C).permute(0, 2, 1, 3).reshape(b * d, T, C)       # (b d) f c
    rope = (pre + "pos_encoder.pe") not in sd                                 # pe='rope': no buffer in the checkpoint (motion_module.py:221-224)
    if not rope:
        x = x + sd[pre + "pos_encoder.pe"][:, :T]
    q = F.linear(x, sd[pre + "to_q.weight"])
    k = F.linear(x, sd[pre + "to_k.weight"])
    v = F.linear(x, sd[pre + "to_v.weight"])
    if rope:
        # motion_module.py:254-257 + motion_module/attention.py:403-429: channel pairs (2i, 2i+1) of q and k, over the FULL width C
        # (before the head split), rotated by frame_index * 10000^(-2i/C)
        freqs = 1.0 / (10000.0 ** (torch.arange(0, C, 2)[: C // 2].float() / C))
        ang = torch.outer(torch.arange(T, dtype=torch.float32), freqs)        # [T, C/2]
        cos, sin = torch.cos(ang), torch.sin(ang)

        def rot(t):
            a, bb = t.float().reshape(b * d, T, C // 2, 2).unbind(-1)
            return torch.stack((a * cos - bb * sin, a * sin + bb * cos), dim=-1).flatten(2).to(t.dtype)

        q, k = rot(q), rot(k)
    hd = C // heads

    def split(t):
        return t.reshape(b * d, T, heads, hd).permute(0, 2, 1, 3)

    q, k, v = split(q), split(k), split(v)
    a = (q @ k.transpose(-1, -2) * hd ** -0.5).softmax(dim=-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(b * d, T, C)
    o = F.linear(o, sd[pre + "to_out.0.weight"], sd[pre + "to_out.0.bias"])
    return o.reshape(b, d, T, C).permute(0, 2, 1, 3).reshape(BT, d, C)


def temporal_module(sd: SD, m: int, x: Tensor, T: int) -> Tensor:
    """motion_module/motion_module.py:102-126,164-177 on frame-major input
    x: [(b f), C, h, w] (the permutes at dpt_temporal.py:75 cancel)."""
    t = f"head.motion_modules.{m}.temporal_transformer."
    BT, C, hh, ww = x.shape
    g = F.group_norm(x, 32, sd[t + "norm.weight"], sd[t + "norm.bias"], 1e-6)
    hs = g.permute(0, 2, 3, 1).reshape(BT, hh * ww, C)
    hs = F.linear(hs, sd[t + "proj_in.weight"], sd[t + "proj_in.bias"])
    tb = t + "transformer_blocks.0."
    for a in (0, 1):
        n = F.layer_norm(hs, (C,), sd[f"{tb}norms.{a}.weight"], sd[f"{tb}norms.{a}.bias"], 1e-5)
        hs = temporal_attention(sd, f"{tb}attention_blocks.{a}.", n, T) + hs
    n = F.layer_norm(hs, (C,), sd[tb + "ff_norm.weight"], sd[tb + "ff_norm.bias"], 1e-5)
    p = F.linear(n, sd[tb + "ff.net.0.proj.weight"], sd[tb + "ff.net.0.proj.bias"])
    val, gate = p.chunk(2, dim=-1)                                            # attention.py:383-384
    hs = F.linear(val * F.gelu(gate), sd[tb + "ff.net.2.weight"], sd[tb + "ff.net.2.bias"]) + hs
    hs = F.linear(hs, sd[t + "proj_out.weight"], sd[t + "proj_out.bias"])
    return hs.reshape(BT, hh, ww, C).permute(0, 3, 1, 2) + x


# --------------------------------------------------------------------------
# DPT head
# --------------------------------------------------------------------------
def _conv(sd: SD, name: str, x: Tensor, stride=1, padding=1) -> Tensor:
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=stride, padding=padding)


def residual_conv_unit(sd: SD, pre: str, x: Tensor) -> Tensor:
    """util/blocks.py:68-91; activation is nn.ReLU(False) so the skip keeps the raw x."""
    def bn(y, n):                                                             # use_bn=True (util/blocks.py:80-81,85-86), eval mode
        if pre + n + ".weight" not in sd:
            return y
        return F.batch_norm(y, sd[pre + n + ".running_mean"], sd[pre + n + ".running_var"], sd[pre + n + ".weight"], sd[pre + n + ".bias"], False, 0.0, 1e-5)

    y = bn(_conv(sd, pre + "conv1", F.relu(x)), "bn1")
    y = bn(_conv(sd, pre + "conv2", F.relu(y)), "bn2")
    return y + x


def fusion_block(sd: SD, i: int, x0: Tensor, x1=None, size=None) -> Tensor:
    """util/blocks.py:135-162."""
    r = f"head.scratch.refinenet{i}."
    out = x0
    if x1 is not None:
        out = out + residual_conv_unit(sd, r + "resConfUnit1.", x1)
    out = residual_conv_unit(sd, r + "resConfUnit2.", out)
    if size is None:
        out = F.interpolate(out, scale_factor=2, mode="bilinear", align_corners=True)
    else:
        out = F.interpolate(out, size=tuple(size), mode="bilinear", align_corners=True)
    return _conv(sd, r + "out_conv", out, padding=0)


def reassemble(sd: SD, taps: Sequence[Tensor], ph: int, pw: int) -> List[Tensor]:
    """dpt_temporal.py:55-69 + dpt.py:60-90."""
    out = []
    for i, x in enumerate(taps):
        x = x.permute(0, 2, 1).reshape(x.shape[0], x.shape[-1], ph, pw)
        x = _conv(sd, f"head.projects.{i}", x, padding=0)
        if i == 0:
            x = F.conv_transpose2d(x, sd["head.resize_layers.0.weight"], sd["head.resize_layers.0.bias"], stride=4)
        elif i == 1:
            x = F.conv_transpose2d(x, sd["head.resize_layers.1.weight"], sd["head.resize_layers.1.bias"], stride=2)
        elif i == 3:
            x = _conv(sd, "head.resize_layers.3", x, stride=2, padding=1)
        out.append(x)
    return out


def head_forward(sd: SD, cfg, taps: Sequence[Tensor], ph: int, pw: int, T: int, stages: dict = None) -> Tensor:
    """dpt_temporal.py:53-114 (the micro-batch split at :88-114 is numerically a no-op)."""
    l1, l2, l3, l4 = reassemble(sd, taps, ph, pw)
    l3 = temporal_module(sd, 0, l3, T)
    l4 = temporal_module(sd, 1, l4, T)
    l1r = _conv(sd, "head.scratch.layer1_rn", l1)
    l2r = _conv(sd, "head.scratch.layer2_rn", l2)
    l3r = _conv(sd, "head.scratch.layer3_rn", l3)
    l4r = _conv(sd, "head.scratch.layer4_rn", l4)
    p4 = fusion_block(sd, 4, l4r, size=l3r.shape[2:])
    p4 = temporal_module(sd, 2, p4, T)
    p3 = fusion_block(sd, 3, p4, l3r, size=l2r.shape[2:])
    p3 = temporal_module(sd, 3, p3, T)
    p2 = fusion_block(sd, 2, p3, l2r, size=l1r.shape[2:])
    p1 = fusion_block(sd, 1, p2, l1r)
    out = _conv(sd, "head.scratch.output_conv1", p1)
    out = F.interpolate(out, (ph * PATCH, pw * PATCH), mode="bilinear", align_corners=True)
    out = F.relu(_conv(sd, "head.scratch.output_conv2.0", out))
    out = F.relu(_conv(sd, "head.scratch.output_conv2.2", out, padding=0))
    if stages is not None:
        stages.update(layer_1=l1, layer_2=l2, layer_3=l3, layer_4=l4, path_4=p4, path_3=p3, path_2=p2, path_1=p1)
    return out


def forward(sd: SD, cfg, x: Tensor, stages: dict = None) -> Tensor:
    """video_depth.py:89-93,161-164: x [B,T,3,H,W] -> depth [B,T,H,W]."""
    B, T, C, H, W = x.shape
    ph, pw = H // PATCH, W // PATCH
    taps = encoder_taps(sd, cfg, x.flatten(0, 1).float())
    if stages is not None:
        stages["taps"] = taps
    d = head_forward(sd, cfg, taps, ph, pw, T, stages)
    d = F.interpolate(d, size=(H, W), mode="bilinear", align_corners=True)
    d = F.relu(d)
    return d.squeeze(1).unflatten(0, (B, T))


# --------------------------------------------------------------------------
# Host side: preprocessing, window schedule, stitcher
# --------------------------------------------------------------------------
def constrain_to_multiple_of(x: float, min_val: int = 0, multiple: int = PATCH) -> int:
    """util/transform.py:51-60 (lower_bound branch: no max_val)."""
    y = int(np.round(x / multiple) * multiple)
    if y < min_val:
        y = int(np.ceil(x / multiple) * multiple)
    return y


def network_size(height: int, width: int, input_size: int = 518) -> Tuple[int, int, int]:
    """video_depth.py:167-171 aspect guard + util/transform.py:62-107 get_size
    (keep_aspect_ratio, lower_bound, multiple of 14). Returns (new_h, new_w, input_size)."""
    ratio = max(height, width) / min(height, width)
    if ratio > 1.78:
        input_size = int(input_size * 1.777 / ratio)
        input_size = round(input_size / 14) * 14
    scale_h, scale_w = input_size / height, input_size / width
    if scale_w > scale_h:
        scale_h = scale_w
    else:
        scale_w = scale_h
    return (constrain_to_multiple_of(scale_h * height, input_size),
            constrain_to_multiple_of(scale_w * width, input_size), input_size)


def resize_cubic(img: np.ndarray, new_h: int, new_w: int) -> np.ndarray:
    """util/transform.py:113 calls cv2.resize(INTER_CUBIC). cv2 is not available
    offline: identity sizes pass through exactly (the pinned case); any other
    size is PARITY UNPINNED and refused here rather than approximated."""
    if img.shape[0] == new_h and img.shape[1] == new_w:
        return img
    raise NotImplementedError("cv2.INTER_CUBIC resize is not reproducible offline (parity unpinned)")


def preprocess_frame(frame_u8: np.ndarray, input_size: int = 518) -> np.ndarray:
    """video_depth.py:198 + util/transform.py:109-158: /255, resize, normalise
    (numpy promotes to float64 against the python-list mean/std), CHW, fp32."""
    img = frame_u8.astype(np.float32) / 255.0
    nh, nw, _ = network_size(img.shape[0], img.shape[1], input_size)
    img = resize_cubic(img, nh, nw)
    img = (img - [0.485, 0.456, 0.406]) / [0.229, 0.224, 0.225]
    return np.ascontiguousarray(np.transpose(img, (2, 0, 1))).astype(np.float32)


def window_plan(n_frames: int) -> Tuple[int, List[int]]:
    """video_depth.py:188-195: (number of appended copies of the last frame, window starts)."""
    step = INFER_LEN - OVERLAP
    append = (step - (n_frames % step)) % step + (INFER_LEN - step)
    return append, list(range(0, n_frames, step))


def compute_scale_and_shift(prediction: np.ndarray, target: np.ndarray) -> Tuple[float, float]:
    """utils/util.py:40-62 with the all-ones mask the caller passes (video_depth.py:232)."""
    prediction = prediction.astype(np.float32)
    target = target.astype(np.float32)
    mask = np.ones_like(target, dtype=np.float32)
    a_00 = np.sum(mask * prediction * prediction)
    a_01 = np.sum(mask * prediction)
    a_11 = np.sum(mask)
    b_0 = np.sum(mask * prediction * target)
    b_1 = np.sum(mask * target)
    x_0, x_1 = 1, 0
    det = a_00 * a_11 - a_01 * a_01
    if det != 0:
        x_0 = (a_11 * b_0 - a_01 * b_1) / det
        x_1 = (-a_01 * b_0 + a_00 * b_1) / det
    return x_0, x_1


def interpolate_frames(pre: List[np.ndarray], post: List[np.ndarray]) -> List[np.ndarray]:
    """utils/util.py:65-74."""
    assert len(pre) == len(post)
    step = 1.0 / (len(pre) - 1)
    w = [0.0] + [i * step for i in range(1, len(pre) - 1)] + [1.0]
    return [pre[i] * (1 - w[i]) + post[i] * w[i] for i in range(len(pre))]


def stitch(depth_list: List[np.ndarray], n_frames: int, metric: bool = False) -> np.ndarray:
    """video_depth.py:216-254. depth_list holds 32 maps per window, in window order.
    metric=True follows metric_depth/video_depth_anything/video_depth.py:132
    (scale, shift = 1, 0; cross-fade kept)."""
    depth_list = list(depth_list)
    aligned: List[np.ndarray] = []
    ref_align: List[np.ndarray] = []
    align_len = OVERLAP - INTERP_LEN
    kf_align = KEYFRAMES[:align_len]
    for fid in range(0, len(depth_list), INFER_LEN):
        if len(aligned) == 0:
            aligned += depth_list[:INFER_LEN]
            for kf in kf_align:
                ref_align.append(depth_list[fid + kf])
        else:
            cur = [depth_list[fid + i] for i in range(len(kf_align))]
            if metric:
                scale, shift = 1.0, 0.0
            else:
                scale, shift = compute_scale_and_shift(np.concatenate(cur), np.concatenate(ref_align))
            pre = aligned[-INTERP_LEN:]
            post = depth_list[fid + align_len:fid + OVERLAP]
            for i in range(len(post)):
                post[i] = post[i] * scale + shift
                post[i][post[i] < 0] = 0
            aligned[-INTERP_LEN:] = interpolate_frames(pre, post)
            for i in range(OVERLAP, INFER_LEN):
                nd = depth_list[fid + i] * scale + shift
                nd[nd < 0] = 0
                aligned.append(nd)
            ref_align = ref_align[:1]
            for kf in kf_align[1:]:
                nd = depth_list[fid + kf] * scale + shift
                nd[nd < 0] = 0
                ref_align.append(nd)
    return np.stack(aligned[:n_frames], axis=0)


def infer_video_depth(sd: SD, cfg, frames: np.ndarray, target_fps, input_size: int = 518,
                      metric: bool = False) -> Tuple[np.ndarray, float]:
    """video_depth.py:166-254 with fp32=True semantics."""
    H0, W0 = frames[0].shape[:2]
    frame_list = [frames[i] for i in range(frames.shape[0])]
    n = len(frame_list)
    append, starts = window_plan(n)
    frame_list = frame_list + [frame_list[-1].copy()] * append
    depth_list: List[np.ndarray] = []
    pre_input = None
    for fid in starts:
        cur = torch.from_numpy(np.stack([preprocess_frame(frame_list[fid + i], input_size)
                                         for i in range(INFER_LEN)]))[None]
        if pre_input is not None:
            cur[:, :OVERLAP] = pre_input[:, KEYFRAMES]
        with torch.no_grad():
            depth = forward(sd, cfg, cur)
        depth = F.interpolate(depth.flatten(0, 1).unsqueeze(1), size=(H0, W0), mode="bilinear", align_corners=True)
        depth_list += [depth[i][0].numpy() for i in range(depth.shape[0])]
        pre_input = cur
    return stitch(depth_list, n, metric), target_fps
