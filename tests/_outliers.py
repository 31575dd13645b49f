"""Seeded state dicts whose encoder residual stream looks like a trained DINOv2's instead of a freshly initialised one
(test infrastructure; VERDICT r2 "missing" #2).

Every parity fixture otherwise uses `weights.synthetic_state_dict`: trunc-normal-like weights, a residual stream of O(1)
values with token means near zero. Trained ViTs are not like that: a handful of channels carry values hundreds of times
the typical magnitude ("massive activations") and token means need not be small against the token's standard deviation.
The reference keeps the stream fp32 under autocast (dinov2_layers/block.py:105-106); this engine's fp16 path keeps it as two
fp16 planes and feeds the rounded `hi` plane to the LayerNorm-folded GEMMs, so exactly these regimes need their own check.

kinds
  "channels": three channels of the stream at |x| ~ 300..1000 from the patch embedding on, pushed further by every block
              (patch_embed.proj.bias, attn.proj.bias, mlp.fc2.bias of those channels)
  "offset":   every channel of the stream shifted by +40 (patch_embed.proj.bias): token mean / sigma ~ 15..40
  "both":     both at once
"""
import torch
import torch.nn.functional as F

OUTLIER_CHANNELS = (7, 123, 300)
OUTLIER_VALUES = (400.0, -650.0, 900.0)
OFFSET = 40.0


def outlier_state_dict(cfg, seed, kind):
    from video_depth_anything_amd.weights import synthetic_state_dict
    assert kind in ("channels", "offset", "both")
    sd = synthetic_state_dict(cfg, seed=seed)
    D = cfg.embed_dim
    ch = [c % D for c in OUTLIER_CHANNELS]
    if kind in ("offset", "both"):
        sd["pretrained.patch_embed.proj.bias"] = sd["pretrained.patch_embed.proj.bias"] + OFFSET
        sd["pretrained.cls_token"] = sd["pretrained.cls_token"] + OFFSET            # the cls row does not see the conv bias
    if kind in ("channels", "both"):
        b = sd["pretrained.patch_embed.proj.bias"].clone()
        c = sd["pretrained.cls_token"].clone()
        for j, v in zip(ch, OUTLIER_VALUES):
            b[j] += v
            c[0, 0, j] += v
        sd["pretrained.patch_embed.proj.bias"], sd["pretrained.cls_token"] = b, c
        for i in range(cfg.depth):                                                 # every block keeps feeding them
            for name, step in ((f"pretrained.blocks.{i}.attn.proj.bias", 12.0), (f"pretrained.blocks.{i}.mlp.fc2.bias", 9.0)):
                t = sd[name].clone()
                for j, v in zip(ch, OUTLIER_VALUES):
                    t[j] += step if v > 0 else -step
                sd[name] = t
    return sd


def stream_report(sd, cfg, x):
    """What the state dict does to the ORACLE's residual stream: the largest |x| met at any LayerNorm input of the encoder and
    the median / maximum over tokens of |token mean| / token sigma there."""
    from oracle import vda_oracle as O
    t = O.patch_tokens(sd, x.flatten(0, 1))
    max_abs, ratios = 0.0, []

    def look(t):
        nonlocal max_abs
        max_abs = max(max_abs, float(t.abs().max()))
        ratios.append((t.mean(-1).abs() / t.std(-1, unbiased=False)).flatten())

    D = t.shape[-1]
    for i in range(cfg.depth):
        p = f"pretrained.blocks.{i}."
        look(t)                                                                   # input of norm1
        y = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        t = t + O.vit_attention(sd, p + "attn.", y, cfg.num_heads) * sd[p + "ls1.gamma"]
        look(t)                                                                   # input of norm2
        t = _mlp_half(sd, p, t)
    r = torch.cat(ratios)
    return {"max_abs": max_abs, "mean_over_sigma_p50": float(r.median()), "mean_over_sigma_max": float(r.max())}


def _mlp_half(sd, p, x):
    """dinov2_layers/block.py:106 (the second residual branch of a block), from the stream after the attention branch."""
    D = x.shape[-1]
    y = F.layer_norm(x, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
    y = F.gelu(F.linear(y, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
    y = F.linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return x + y * sd[p + "ls2.gamma"]
