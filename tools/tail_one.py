"""One launch set of the fused depth tail (both variants) at the ViT-L shape: what tools/pmc_tail.sh profiles."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(0)
h, H, C = 296, 518, 128
x = torch.randn(32, h, h, C, device="cuda", generator=g).half()
w2 = (torch.randn(32, 9 * C, device="cuda", generator=g) * 0.03).half(); b2 = torch.randn(32, device="cuda", generator=g); w3 = torch.randn(32, device="cuda", generator=g)
o = torch.empty(32, H, H, dtype=torch.float32, device="cuda")
for v in [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "0,1").split(",")]:
    _lib.lib.vda_depth_tail_set_variant(v)
    for _ in range(3): ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
    torch.cuda.synchronize()
_lib.lib.vda_depth_tail_set_variant(0)
