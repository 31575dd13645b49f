"""A few launches of the ViT-S 64 -> 64 conv at 148^2 (residual epilogue), both LDS-conv variants: what tools/pmc_kernel.sh profiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(0)
B, hw, C, N = 32, 148, 64, 64
x = torch.randn(B, hw, hw, C, device="cuda", generator=g).half()
w = (torch.randn(N, 9 * C, device="cuda", generator=g) * (9 * C) ** -0.5).half(); b = torch.randn(N, device="cuda", generator=g)
res = torch.randn(B, hw, hw, N, device="cuda", generator=g).half()
out = torch.empty(B, hw, hw, N, dtype=torch.float16, device="cuda")
for v in (0, 1):
    _lib.lib.vda_conv_lds_set_variant(v)
    for _ in range(3): ops.gemm(x, w, out, _lib.EPI_RES_F16, M=B * hw * hw, N=N, K=9 * C, bias=b, relu_in=True, res=res, conv=(B, hw, hw, C, hw, hw, 1))
    torch.cuda.synchronize()
_lib.lib.vda_conv_lds_set_variant(0)
