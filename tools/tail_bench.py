import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
for (h, H, C) in [(296, 518, 128), (518, 518, 128), (296, 518, 64)]:
    x = torch.randn(32, h, h, C, device="cuda", generator=g).half()
    w2 = (torch.randn(32, 9 * C, device="cuda", generator=g) * 0.03).half(); b2 = torch.randn(32, device="cuda", generator=g); w3 = torch.randn(32, device="cuda", generator=g)
    o = torch.empty(32, H, H, dtype=torch.float32, device="cuda")
    for _ in range(2): ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print(f"fused tail {h}->{H}, C={C}: {t*1e3:.0f} us  ({2*32*H*H*32*9*C/t/1e9:.0f} TFLOP/s on the conv)")
