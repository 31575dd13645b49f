"""A few launches of the fused output_conv1 (vda_conv3x3_up2_f16) at the ViT-L shape: what tools/pmc_kernel.sh profiles."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops
B, h, C, N = 32, 148, 256, 128
g = torch.Generator().manual_seed(1)
x = torch.randn(B, h, h, C, generator=g).half().cuda()
w = ops.pack_conv3x3(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5).cuda()
b = torch.randn(N, generator=g).cuda()
out = torch.empty(B, 2 * h, 2 * h, N, dtype=torch.float16, device="cuda")
for _ in range(3): ops.conv3x3_up2(x, w, b, out, B, h, h, C, N, N)
torch.cuda.synchronize()
