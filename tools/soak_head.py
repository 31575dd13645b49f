"""Run-to-run determinism of the round-4 head kernels: the persistent depth tail (specialised waves, three rings behind one barrier) and the
fused output_conv1, N launches each at the benchmarked shapes, every result compared bit for bit with the first; then the whole forward."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
g = torch.Generator(device="cuda").manual_seed(0)
for (h, H, C) in [(296, 518, 128), (296, 518, 32), (74, 130, 64)]:
    x = torch.randn(32, h, h, C, device="cuda", generator=g).half()
    w2 = (torch.randn(32, 9 * C, device="cuda", generator=g) * 0.03).half(); b2 = torch.randn(32, device="cuda", generator=g); w3 = torch.randn(32, device="cuda", generator=g)
    ref = None
    for i in range(n):
        o = torch.empty(32, H, H, dtype=torch.float32, device="cuda")
        ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
        if ref is None: ref = o.clone()
        else: assert torch.equal(o, ref), f"depth tail {h}->{H} C={C}: launch {i} differs"
    print(f"depth tail {h}->{H} C={C}: {n} launches bit-identical", flush=True)
for (B, h, C, N) in [(32, 148, 256, 128), (32, 148, 64, 32)]:
    gc = torch.Generator().manual_seed(1)
    x = torch.randn(B, h, h, C, generator=gc).half().cuda()
    w = ops.pack_conv3x3(torch.randn(N, C, 3, 3, generator=gc) * (9 * C) ** -0.5).cuda(); b = torch.randn(N, generator=gc).cuda()
    ref = None
    for i in range(n):
        out = torch.empty(B, 2 * h, 2 * h, N, dtype=torch.float16, device="cuda")
        ops.conv3x3_up2(x, w, b, out, B, h, h, C, N, N)
        if ref is None: ref = out.clone()
        else: assert torch.equal(out, ref), f"conv3x3_up2 C={C} N={N}: launch {i} differs"
    print(f"conv3x3_up2 {C}->{N}: {n} launches bit-identical", flush=True)
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict
for enc in ("vits", "vitl"):
    cfg = get_config(enc)
    m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
    x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
    ref = m.forward(x, fp32=False).clone()
    for i in range(max(n // 5, 5)):
        assert torch.equal(m.forward(x, fp32=False), ref), f"{enc} forward {i} differs"
    print(f"{enc} forward: {max(n // 5, 5)} runs bit-identical", flush=True)
