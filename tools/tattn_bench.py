"""Temporal attention at the ViT-L head shapes: MFMA kernel (variant 1) vs VALU kernel (variant 0), interleaved in one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
g = torch.Generator(device="cuda").manual_seed(0)
for (C, hw) in [(1024, 1369), (1024, 361), (256, 1369), (256, 5476)]:
    qkv = torch.randn(32 * hw, 3 * C, device="cuda", generator=g).half()
    o = torch.empty(32 * hw, C, dtype=torch.float16, device="cuda")
    res = {}
    for rep in range(3):
        for v in (0, 1):
            _lib.lib.vda_temporal_attention_set_variant(v)
            for _ in range(2): ops.temporal_attention(qkv, o, 32, hw, C)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.temporal_attention(qkv, o, 32, hw, C)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) / 10)
    mb = (qkv.numel() + o.numel()) * 2 / 1e6
    print(f"C={C} hw={hw}: VALU {min(res[0])*1e3:.1f} us, MFMA {min(res[1])*1e3:.1f} us ({mb / min(res[1]) / 1e3:.2f} TB/s)", flush=True)
_lib.lib.vda_temporal_attention_set_variant(1)
