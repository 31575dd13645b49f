"""The fused output_conv1 at the ViT-L shape under vda_conv3x3_up2_set_variant values (timing experiments), one process, interleaved.
usage: conv_up_variants.py 0,1,2"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
vs = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2").split(",")]
B, h, C, N = 32, 148, 256, 128
g = torch.Generator().manual_seed(1)
x = torch.randn(B, h, h, C, generator=g).half().cuda()
w = ops.pack_conv3x3(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda()
out = torch.empty(B, 2 * h, 2 * h, N, dtype=torch.float16, device="cuda")
ts = {v: [] for v in vs}
for rep in range(5):
    for v in vs:
        _lib.lib.vda_conv3x3_up2_set_variant(v)
        ops.conv3x3_up2(x, w, b, out, B, h, h, C, N, N); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.conv3x3_up2(x, w, b, out, B, h, h, C, N, N)
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 5)
_lib.lib.vda_conv3x3_up2_set_variant(0)
print("  ".join(f"variant {v}: {sorted(t)[2]*1e3:.0f} us" for v, t in ts.items()))
