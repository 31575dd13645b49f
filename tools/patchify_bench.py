"""patchify (fp32 frames -> fp16 patch rows, the patch-embed GEMM's A operand) at the ViT-L clip shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops
x = torch.randn(32, 3, 518, 518, device="cuda")
a = torch.zeros(32 * 37 * 37, 640, dtype=torch.float16, device="cuda")
for _ in range(3): ops.patchify(x, a, 32, 518, 518, 640)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.patchify(x, a, 32, 518, 518, 640)
e1.record(); torch.cuda.synchronize()
print(f"patchify 32x3x518x518 -> fp16 [43808, 640]: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us")
