import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from video_depth_anything_amd import ops
from video_depth_anything_amd.config import get_config
from video_depth_anything_amd.video_depth import VideoDepthAnything
from video_depth_anything_amd.weights import state_dict_spec
for enc in ("vitl", "vits"):
    cfg = get_config(enc)
    g = torch.Generator().manual_seed(0)
    sd = {k: (torch.randn(s, generator=g) * 0.02 if len(s) > 1 else torch.ones(s)) for k, s in state_dict_spec(cfg).items()}
    m = VideoDepthAnything(encoder=enc, features=cfg.features, out_channels=list(cfg.out_channels)); m.load_state_dict(sd); m = m.to("cuda")
    x = torch.randn(1, 32, 3, 518, 518, generator=g).cuda()
    for _ in range(3): m(x)
    for rep in range(2):
        for prof in (False, True):
            ops.PROFILE = ops.GemmProfile(every=4) if prof else None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): m(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            ops.PROFILE = None
            print(enc, "events" if prof else "plain ", f"{dt*1e3:.2f} ms/clip", flush=True)
    # graph
    sx = x.clone(); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        sdp = m.engine.forward(sx)
    torch.cuda.synchronize()
    ref = m(x).clone()
    gr.replay(); torch.cuda.synchronize()
    print(enc, "graph equal:", torch.equal(ref, sdp))
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): gr.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print(enc, "graph ", f"{dt*1e3:.2f} ms/clip", flush=True)
