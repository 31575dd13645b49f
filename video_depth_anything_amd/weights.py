"""Checkpoint contract of the hot path: key names, shapes, and a seeded synthetic
state dict.

The flat fp32 state dict is the one `run.py` hands to `load_state_dict(strict=True)`
(/root/reference/run.py:46). Key inventory follows the modules that own the
parameters:
  pretrained.*            /root/reference/video_depth_anything/dinov2.py:106-168
  head.projects/resize    /root/reference/video_depth_anything/dpt.py:60-98 (readout_projects when use_clstoken)
  head.scratch.*          /root/reference/video_depth_anything/util/blocks.py:20-32,52-58,124-129
                          /root/reference/video_depth_anything/dpt.py:117-124
  head.motion_modules.*   /root/reference/video_depth_anything/motion_module/motion_module.py:84-100,141-161,194
                          /root/reference/video_depth_anything/motion_module/attention.py:81-91,333,374

There are no trained checkpoints offline, so tests and the bench use
`synthetic_state_dict`: every tensor drawn from a seeded CPU generator with
non-trivial LayerScale / norm affine / proj_out values (the reference
zero-initialises proj_out, motion_module.py:57-58, which would turn every
temporal module into an exact no-op and hide bugs).
"""
import math
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import ModelConfig, PATCH, POS_GRID, TEMPORAL_HEADS


def state_dict_spec(cfg: ModelConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    """name -> shape, in the reference's registration order."""
    D, F_, oc = cfg.embed_dim, cfg.features, cfg.out_channels
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    p = "pretrained."
    s[p + "cls_token"] = (1, 1, D)
    s[p + "pos_embed"] = (1, POS_GRID * POS_GRID + 1, D)
    s[p + "mask_token"] = (1, D)
    s[p + "patch_embed.proj.weight"] = (D, 3, PATCH, PATCH)
    s[p + "patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        b = f"{p}blocks.{i}."
        s[b + "norm1.weight"] = (D,)
        s[b + "norm1.bias"] = (D,)
        s[b + "attn.qkv.weight"] = (3 * D, D)
        s[b + "attn.qkv.bias"] = (3 * D,)
        s[b + "attn.proj.weight"] = (D, D)
        s[b + "attn.proj.bias"] = (D,)
        s[b + "ls1.gamma"] = (D,)
        s[b + "norm2.weight"] = (D,)
        s[b + "norm2.bias"] = (D,)
        s[b + "mlp.fc1.weight"] = (cfg.mlp_ratio * D, D)
        s[b + "mlp.fc1.bias"] = (cfg.mlp_ratio * D,)
        s[b + "mlp.fc2.weight"] = (D, cfg.mlp_ratio * D)
        s[b + "mlp.fc2.bias"] = (D,)
        s[b + "ls2.gamma"] = (D,)
    s[p + "norm.weight"] = (D,)
    s[p + "norm.bias"] = (D,)

    h = "head."
    for i in range(4):
        s[f"{h}projects.{i}.weight"] = (oc[i], D, 1, 1)
        s[f"{h}projects.{i}.bias"] = (oc[i],)
    s[h + "resize_layers.0.weight"] = (oc[0], oc[0], 4, 4)   # ConvTranspose2d: [Cin, Cout, k, k]
    s[h + "resize_layers.0.bias"] = (oc[0],)
    s[h + "resize_layers.1.weight"] = (oc[1], oc[1], 2, 2)
    s[h + "resize_layers.1.bias"] = (oc[1],)
    s[h + "resize_layers.3.weight"] = (oc[3], oc[3], 3, 3)
    s[h + "resize_layers.3.bias"] = (oc[3],)
    if cfg.use_clstoken:                                     # dpt.py:92-98
        for i in range(4):
            s[f"{h}readout_projects.{i}.0.weight"] = (D, 2 * D)
            s[f"{h}readout_projects.{i}.0.bias"] = (D,)
    sc = h + "scratch."
    for i in range(4):
        s[f"{sc}layer{i + 1}_rn.weight"] = (F_, oc[i], 3, 3)
    for i in (1, 2, 3, 4):
        r = f"{sc}refinenet{i}."
        s[r + "out_conv.weight"] = (F_, F_, 1, 1)
        s[r + "out_conv.bias"] = (F_,)
        for u in (1, 2):
            for c in (1, 2):
                s[f"{r}resConfUnit{u}.conv{c}.weight"] = (F_, F_, 3, 3)
                s[f"{r}resConfUnit{u}.conv{c}.bias"] = (F_,)
            if cfg.use_bn:                                   # util/blocks.py:60-62: nn.BatchNorm2d(features) x 2, registered after the convs
                for c in (1, 2):
                    b = f"{r}resConfUnit{u}.bn{c}."
                    s[b + "weight"] = (F_,)
                    s[b + "bias"] = (F_,)
                    s[b + "running_mean"] = (F_,)
                    s[b + "running_var"] = (F_,)
                    s[b + "num_batches_tracked"] = ()
    s[sc + "output_conv1.weight"] = (F_ // 2, F_, 3, 3)
    s[sc + "output_conv1.bias"] = (F_ // 2,)
    s[sc + "output_conv2.0.weight"] = (32, F_ // 2, 3, 3)
    s[sc + "output_conv2.0.bias"] = (32,)
    s[sc + "output_conv2.2.weight"] = (1, 32, 1, 1)
    s[sc + "output_conv2.2.bias"] = (1,)
    for m, C in enumerate(temporal_channels(cfg)):
        t = f"{h}motion_modules.{m}.temporal_transformer."
        s[t + "norm.weight"] = (C,)
        s[t + "norm.bias"] = (C,)
        s[t + "proj_in.weight"] = (C, C)
        s[t + "proj_in.bias"] = (C,)
        tb = t + "transformer_blocks.0."
        for a in (0, 1):
            ab = f"{tb}attention_blocks.{a}."
            s[ab + "to_q.weight"] = (C, C)
            s[ab + "to_k.weight"] = (C, C)
            s[ab + "to_v.weight"] = (C, C)
            s[ab + "to_out.0.weight"] = (C, C)
            s[ab + "to_out.0.bias"] = (C,)
            if cfg.pe == "ape":                              # (rope: freqs_cis is a plain attribute, motion_module.py:221-224 - no key)
                s[ab + "pos_encoder.pe"] = (1, cfg.num_frames, C)
        for a in (0, 1):
            s[f"{tb}norms.{a}.weight"] = (C,)
            s[f"{tb}norms.{a}.bias"] = (C,)
        s[tb + "ff.net.0.proj.weight"] = (8 * C, C)
        s[tb + "ff.net.0.proj.bias"] = (8 * C,)
        s[tb + "ff.net.2.weight"] = (C, 4 * C)
        s[tb + "ff.net.2.bias"] = (C,)
        s[tb + "ff_norm.weight"] = (C,)
        s[tb + "ff_norm.bias"] = (C,)
        s[t + "proj_out.weight"] = (C, C)
        s[t + "proj_out.bias"] = (C,)
    return s


def temporal_channels(cfg: ModelConfig) -> Tuple[int, int, int, int]:
    """Channel width of motion_modules[0..3] (dpt_temporal.py:42-51)."""
    return (cfg.out_channels[2], cfg.out_channels[3], cfg.features, cfg.features)


def sinusoidal_pe(num_frames: int, C: int) -> torch.Tensor:
    """The `pos_encoder.pe` buffer (motion_module.py:189-194)."""
    position = torch.arange(num_frames).unsqueeze(1)
    div_term = torch.exp(torch.arange(0, C, 2) * (-math.log(10000.0) / C))
    pe = torch.zeros(1, num_frames, C)
    pe[0, :, 0::2] = torch.sin(position * div_term)
    pe[0, :, 1::2] = torch.cos(position * div_term)
    return pe


def synthetic_state_dict(cfg: ModelConfig, seed: int = 0) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic fp32 CPU state dict with the checkpoint's keys and shapes.

    Matrix weights ~ N(0, fan_in^-1/2) so activations keep O(1) scale through
    the depth of the net; affine/LayerScale terms are spread around their
    trained magnitudes instead of the constructor's constants.
    """
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def randn(shape, std=1.0):
        return torch.randn(shape, generator=g, dtype=torch.float32) * std

    for name, shape in state_dict_spec(cfg).items():
        leaf = name.rsplit(".", 1)[-1]
        if name.endswith("pos_encoder.pe"):
            t = sinusoidal_pe(shape[1], shape[2])
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "running_mean":
            t = randn(shape, 0.2)
        elif leaf == "num_batches_tracked":
            t = torch.tensor(100, dtype=torch.int64)
        elif leaf == "gamma":
            t = 0.5 + 0.5 * torch.rand(shape, generator=g)
        elif name.endswith("cls_token") or name.endswith("mask_token"):
            t = randn(shape, 0.02)
        elif name.endswith("pos_embed"):
            t = randn(shape, 0.2)
        elif name.endswith("output_conv2.2.bias"):
            t = 1.5 + randn(shape, 0.05)            # keep the final ReLU mostly open
        elif leaf == "bias":
            t = randn(shape, 0.05)
        elif len(shape) == 1:                       # norm weights
            t = 1.0 + randn(shape, 0.1)
        else:
            if "resize_layers.0" in name or "resize_layers.1" in name:
                fan_in = shape[0]                   # ConvTranspose, non-overlapping taps
            else:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
            t = randn(shape, fan_in ** -0.5)
        sd[name] = t.contiguous()
    return sd


def check_state_dict(cfg: ModelConfig, sd: Dict[str, torch.Tensor], strict: bool = True):
    """`load_state_dict(strict=True)` behaviour: report missing / unexpected keys
    and shape mismatches the way torch does (RuntimeError)."""
    spec = state_dict_spec(cfg)
    missing = [k for k in spec if k not in sd]
    unexpected = [k for k in sd if k not in spec]
    errs = []
    if strict and missing:
        errs.append("Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
    if strict and unexpected:
        errs.append("Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
    for k, shape in spec.items():
        if k in sd and tuple(sd[k].shape) != tuple(shape):
            errs.append(f"size mismatch for {k}: copying a param with shape {tuple(sd[k].shape)} "
                        f"from checkpoint, the shape in current model is {tuple(shape)}.")
    if errs:
        raise RuntimeError("Error(s) in loading state_dict for VideoDepthAnything:\n\t" + "\n\t".join(errs))
    return missing, unexpected
