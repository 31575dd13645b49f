"""The fused depth tail at the benchmarked shapes: variant 0 (persistent, round 4) against variant 1 (round 1), one process, interleaved."""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(0)
for (h, H, C) in [(296, 518, 128), (296, 518, 64), (296, 518, 32), (518, 518, 128)]:
    x = torch.randn(32, h, h, C, device="cuda", generator=g).half()
    w2 = (torch.randn(32, 9 * C, device="cuda", generator=g) * 0.03).half(); b2 = torch.randn(32, device="cuda", generator=g); w3 = torch.randn(32, device="cuda", generator=g)
    o = torch.empty(32, H, H, dtype=torch.float32, device="cuda")
    ts = {0: [], 1: []}
    for rep in range(5):
        for v in (0, 1):
            _lib.lib.vda_depth_tail_set_variant(v)
            ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): ops.depth_tail(x, w2, b2, w3, 0.1, o, 32, h, h, H, H, C)
            e1.record(); torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / 5)
    _lib.lib.vda_depth_tail_set_variant(0)
    t0, t1 = sorted(ts[0])[2], sorted(ts[1])[2]
    fl = 2 * 32 * H * H * 32 * 9 * C
    print(f"fused tail {h}->{H}, C={C}: persistent {t0*1e3:.0f} us ({fl/t0/1e9:.0f} TFLOP/s)   round-1 kernel {t1*1e3:.0f} us ({fl/t1/1e9:.0f} TFLOP/s)", flush=True)
