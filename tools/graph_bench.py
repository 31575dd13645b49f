"""hipGraph replay of one vda_forward vs eager launches (torch.cuda.CUDAGraph around the handle call)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import synthetic_state_dict

for enc in sys.argv[1:] or ["vits", "vitl"]:
    cfg = get_config(enc)
    m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(synthetic_state_dict(cfg, seed=0)); m = m.to("cuda")
    x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
    for _ in range(3): ref = m.forward(x, fp32=False)
    ref = ref.clone(); torch.cuda.synchronize()
    def timeit(fn, n=10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    eager = timeit(lambda: m.forward(x, fp32=False))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        m.forward(x, fp32=False)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        out = m.forward(x, fp32=False)
    g.replay(); torch.cuda.synchronize()
    print(enc, "graph output equal:", torch.equal(out, ref), "eager %.2f ms" % eager, "graph %.2f ms" % timeit(g.replay), "eager again %.2f ms" % timeit(lambda: m.forward(x, fp32=False)), flush=True)
