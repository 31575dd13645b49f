"""The ViT-S head's 64-channel 3x3 convs: the persistent C = 64 kernel (variant 0) against the per-pass LDS kernel (variant 1), one process."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import ops, _lib
g = torch.Generator(device="cuda").manual_seed(0)
for (hw, N, epi) in [(148, 64, _lib.EPI_RES_F16), (148, 64, _lib.EPI_BIAS_RELU_F16), (74, 64, _lib.EPI_RES_F16), (37, 64, _lib.EPI_RES_F16), (19, 64, _lib.EPI_RES_F16), (296, 32, _lib.EPI_BIAS_F16)]:
    B, C = 32, 64
    x = torch.randn(B, hw, hw, C, device="cuda", generator=g).half()
    w = (torch.randn(N, 9 * C, device="cuda", generator=g) * (9 * C) ** -0.5).half(); b = torch.randn(N, device="cuda", generator=g)
    res = torch.randn(B, hw, hw, N, device="cuda", generator=g).half()
    out = torch.empty(B, hw, hw, N, dtype=torch.float16, device="cuda")
    kw = dict(M=B * hw * hw, N=N, K=9 * C, bias=b, relu_in=True, conv=(B, hw, hw, C, hw, hw, 1))
    if epi == _lib.EPI_RES_F16: kw.update(res=res)
    ts = {0: [], 1: []}
    for rep in range(5):
        for v in (0, 1):
            _lib.lib.vda_conv_lds_set_variant(v)
            ops.gemm(x, w, out, epi, **kw); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.gemm(x, w, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    _lib.lib.vda_conv_lds_set_variant(0)
    t0, t1 = sorted(ts[0])[2], sorted(ts[1])[2]
    fl = 2.0 * B * hw * hw * N * 9 * C
    print(f"conv 64->{N} at {hw}^2 epi {epi}: persistent {t0:.1f} us ({fl/t0/1e6:.0f} TFLOP/s)   per-pass {t1:.1f} us ({fl/t1/1e6:.0f} TFLOP/s)", flush=True)
