import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops
H = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(32, 1370, 3 * H * 64, device="cuda", generator=g).half()
o = torch.empty(32, 1370, H * 64, dtype=torch.float16, device="cuda")
for _ in range(4):
    ops.attention(qkv, o, 32, 1370, H)
torch.cuda.synchronize()
print("ok")
