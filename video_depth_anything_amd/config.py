"""Model and scheduler constants for the infer_video_depth hot path.

Mirrors the values the reference hard-codes:
  - encoder geometry      /root/reference/video_depth_anything/dinov2.py:339-378,398-415
  - tap layers            /root/reference/video_depth_anything/video_depth.py:53-56
  - head widths           /root/reference/run.py:40-43
  - window constants      /root/reference/video_depth_anything/video_depth.py:29-33
"""
from dataclasses import dataclass, field
from typing import List, Tuple

PATCH = 14
POS_GRID = 37            # 518 // 14: side of the stored pos_embed grid
INTERP_OFFSET = 0.1      # dinov2.py:194

# video_depth.py:29-33 ("infer settings, do not change")
INFER_LEN = 32
OVERLAP = 10
KEYFRAMES = [0, 12, 24, 25, 26, 27, 28, 29, 30, 31]
INTERP_LEN = 8

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

TEMPORAL_HEADS = 8       # dpt_temporal.py:35
GN_GROUPS = 32           # motion_module.py:39,84
GN_EPS = 1e-6            # motion_module.py:84
ENC_LN_EPS = 1e-6        # dinov2.py:95
TMP_LN_EPS = 1e-5        # nn.LayerNorm default, motion_module.py:155,161


@dataclass(frozen=True)
class ModelConfig:
    name: str
    embed_dim: int
    depth: int
    num_heads: int
    taps: Tuple[int, int, int, int]
    features: int
    out_channels: Tuple[int, int, int, int]
    num_frames: int = 32
    mlp_ratio: int = 4
    use_clstoken: bool = False        # dpt.py:92-98,129-132: readout_projects (Linear 2D -> D + GELU over [patch, cls])
    use_bn: bool = False              # util/blocks.py:60-62,80-86: BatchNorm2d after each conv of the fusion blocks' ResidualConvUnits
    pe: str = "ape"                   # motion_module.py:214-224: 'ape' (sinusoidal buffer added before q/k/v) or 'rope' (rotary on q, k)

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


_CONFIGS = {
    # run.py:41-42 + video_depth.py:53-56
    "vits": ModelConfig("vits", 384, 12, 6, (2, 5, 8, 11), 64, (48, 96, 192, 384)),
    "vitl": ModelConfig("vitl", 1024, 24, 16, (4, 11, 17, 23), 256, (256, 512, 1024, 1024)),
    # Not a reference model: a 4-block encoder with every tap and ragged head
    # widths, small enough that its weights and goldens are committed fixtures.
    "tiny": ModelConfig("tiny", 128, 4, 2, (0, 1, 2, 3), 64, (48, 96, 64, 128)),
}


def get_config(encoder: str, features: int = None, out_channels=None, num_frames: int = 32, use_clstoken: bool = False,
               use_bn: bool = False, pe: str = "ape") -> ModelConfig:
    if pe not in ("ape", "rope"):
        raise NotImplementedError(pe)              # motion_module.py:226-227
    if encoder not in _CONFIGS:
        raise KeyError(encoder)
    base = _CONFIGS[encoder]
    return ModelConfig(
        base.name, base.embed_dim, base.depth, base.num_heads, base.taps,
        base.features if features is None else int(features),
        base.out_channels if out_channels is None else tuple(int(c) for c in out_channels),
        num_frames, base.mlp_ratio, bool(use_clstoken), bool(use_bn), pe,
    )
