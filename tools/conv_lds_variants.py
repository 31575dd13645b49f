"""ViT-S 64 -> 64 conv at 148^2 under vda_conv_lds_set_variant values (0 persistent, 1 per-pass, 2 / 3 timing experiments), one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
vs = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3").split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
B, hw, C, N = 32, 148, 64, 64
x = torch.randn(B, hw, hw, C, device="cuda", generator=g).half()
w = (torch.randn(N, 9 * C, device="cuda", generator=g) * (9 * C) ** -0.5).half(); b = torch.randn(N, device="cuda", generator=g)
res = torch.randn(B, hw, hw, N, device="cuda", generator=g).half()
out = torch.empty(B, hw, hw, N, dtype=torch.float16, device="cuda")
ts = {v: [] for v in vs}
for rep in range(5):
    for v in vs:
        _lib.lib.vda_conv_lds_set_variant(v)
        f = lambda: ops.gemm(x, w, out, _lib.EPI_RES_F16, M=B * hw * hw, N=N, K=9 * C, bias=b, relu_in=True, res=res, conv=(B, hw, hw, C, hw, hw, 1))
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts[v].append(e0.elapsed_time(e1) / 10 * 1e3)
_lib.lib.vda_conv_lds_set_variant(0)
print("  ".join(f"variant {v}: {sorted(t)[2]:.1f} us" for v, t in ts.items()))
