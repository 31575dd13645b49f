"""Fused depth tail at the ViT-L shape under a list of vda_depth_tail_set_variant values (debug / timing experiments), one process, interleaved.
usage: tail_variants.py 0,1,16"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
vs = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
h, H, C = 296, 518, 128
x = torch.randn(32, h, h, C, device="cuda", generator=g).half()
w2 = (torch.randn(32, 9 * C, device="cuda", generator=g) * 0.03).half(); b2 = torch.randn(32, device="cuda", generator=g); w3 = torch.randn(32, device="cuda", generator=g)
o = torch.empty(32, H, H, dtype=torch.float32, device="cuda")
ts = {v: [] for v in vs}
for rep in range(5):
    for v in vs:
        _lib.lib.vda_depth_tail_set_variant(v)
        ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 5)
_lib.lib.vda_depth_tail_set_variant(0)
print("  ".join(f"variant {v}: {sorted(t)[2]*1e3:.0f} us" for v, t in ts.items()))
