"""Upper bound for intra-device overlap: two independent clips on two HIP streams (two handles) vs one clip at a time.
If the aggregate frames/s of the pair exceeds the single-stream figure, tail rounds / launch gaps of one stream's kernels
are being filled by the other's."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict

enc = sys.argv[1] if len(sys.argv) > 1 else "vitl"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg = get_config(enc)
sd = synthetic_state_dict(cfg, seed=0)
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
if len(sys.argv) > 4:                                    # cap on the persistent kernels' grids (e.g. 128: each stream takes half the chip)
    from video_depth_anything_amd import _lib
    _lib.lib.vda_set_max_wgs(int(sys.argv[4]))
ms = []
for _ in range(NS):
    m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(sd); ms.append(m.to("cuda"))
x = torch.randn(1, T, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
st = [torch.cuda.Stream() for _ in range(NS)]
for m in ms:
    for _ in range(2): m.forward(x, fp32=False)
torch.cuda.synchronize()
def run(n, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        for j in range(NS if two else 1):
            with torch.cuda.stream(st[j]):
                ms[j].forward(x, fp32=False)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    return n * (NS if two else 1) * T / dt
for rep in range(2):
    print(enc, "T", T, "one stream %.1f frames/s" % run(10, False), "%d streams %.1f frames/s" % (NS, run(10, True)), flush=True)
