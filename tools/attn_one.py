"""Spatial attention at the ViT-L clip shape (32 frames x 16 heads x 1370 tokens, d=64): timing, and the rocprofv3 --pmc target."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops
H = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(32, 1370, 3 * H * 64, device="cuda", generator=g).half()
o = torch.empty(32, 1370, H * 64, dtype=torch.float16, device="cuda")
from video_depth_anything_amd import _lib
for scale in [float(v) for v in os.environ.get('ATTN_SCALES', '0.3,1.0').split(',')]:
    q = (qkv * scale).contiguous()
    for rep in range(int(os.environ.get('ATTN_REPS', '3'))):
        for variant in [int(v) for v in os.environ.get('ATTN_VARIANTS', '1,8,9').split(',')]:
            _lib.lib.vda_attention_set_variant(variant)
            for _ in range(2):
                ops.attention(q, o, 32, 1370, H)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attention(q, o, 32, 1370, H)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 10
            print(f"attention 32x{H}x1370x64 input scale {scale} variant {variant}: {t*1e3:.1f} us  {4.0*32*H*1370*1370*64/t/1e9:.0f} TF/s", flush=True)
_lib.lib.vda_attention_set_variant(-1)
