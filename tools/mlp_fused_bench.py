#!/usr/bin/env python
"""vda_mlp_fused_f16 against the unfused pair it replaces (fc1 + LN fold + GELU, fc2 + split-residual epilogue) at ViT-S's clip size,
one process, interleaved, median."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
M, D, H = int(sys.argv[1]) if len(sys.argv) > 1 else 43840, 384, 1536
g = torch.Generator(device="cuda").manual_seed(0)
F16 = torch.float16
hi = torch.randn(M, D, device="cuda", generator=g).half(); lo = torch.zeros_like(hi)
stats = torch.stack([torch.zeros(M), torch.ones(M)], 1).contiguous().cuda()
Wf = (torch.randn(H, D, device="cuda", generator=g) * D ** -0.5).half(); c1 = Wf.float().sum(1); c2 = torch.randn(H, device="cuda", generator=g) * 0.1
W2 = (torch.randn(D, H, device="cuda", generator=g) * H ** -0.5).half(); W2p = torch.empty_like(W2); ops.mlp_permute_w2(W2, W2p, D, H)
b2 = torch.randn(D, device="cuda", generator=g) * 0.1; gamma = torch.full((D,), 1e-3, device="cuda")
part = torch.zeros(D // 64, M, 2, device="cuda"); hid = torch.empty(M, H, dtype=F16, device="cuda")
def fused():
    ops.mlp_fused(hi, stats, Wf, c1, c2, W2p, b2, gamma, hi, lo, part, M, D, H)
def pair():
    ops.gemm(hi, Wf, hid, _lib.EPI_LN_GELU_F16, M=M, N=H, K=D, bias=c2, gamma=c1, stats=stats)
    ops.gemm(hid, W2, hi, _lib.EPI_SCALE_RES_SPLIT, M=M, N=D, K=H, bias=b2, gamma=gamma, res=hi, res2=lo, out2=lo, stats=part, pos=stats)
ts = {"fused": [], "pair": []}
for rep in range(7):
    for name, fn in (("fused", fused), ("pair", pair)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ts[name].append(e0.elapsed_time(e1) / 10 * 1e3)
fl = 4.0 * M * D * H
for k, v in ts.items():
    t = sorted(v)[len(v) // 2]
    print(f"M={M} {k}: {t:.1f} us  ({fl / t / 1e6:.0f} TFLOP/s)  all {[round(u, 1) for u in v]}", flush=True)
