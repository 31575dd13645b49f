#!/usr/bin/env python
"""A/B of GEMM variants in ONE process on one box (boxes differ by ~10 %): variants interleaved, median of repeats.
usage: gemm_ab.py v1,v2,... [epi]   (variant codes as vda_gemm_set_variant; 5 + 16*flags = 8-phase with A/B switches)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
variants = [int(v) for v in sys.argv[1].split(",")]
epi = int(sys.argv[2]) if len(sys.argv) > 2 else _lib.EPI_BIAS_F16
shapes = [(43840, 4096, 1024), (43840, 1024, 4096), (43840, 3072, 1024), (43840, 1024, 1024)]
if os.environ.get("AB_SHAPES") == "vits":
    shapes = [(43840, 1536, 384), (43840, 384, 1536), (43840, 1152, 384), (43840, 384, 384)]
conv = len(sys.argv) > 3 and sys.argv[3] == "conv"
if conv:      # head convs: 3x3, 256 -> 256 channels on 32 frames of 148^2 / 74^2 / 37^2
    shapes = [(32 * 148 * 148, 256, 2304), (32 * 74 * 74, 256, 2304), (32 * 37 * 37, 256, 2304), (32 * 296 * 296, 128, 2304)]
    if os.environ.get("AB_SHAPES") == "vits":
        shapes = [(32 * 148 * 148, 64, 576), (32 * 74 * 74, 64, 576), (32 * 296 * 296, 64, 576), (32 * 148 * 148, 64, 1152)]
    if os.environ.get("AB_SHAPES") == "small":        # the head's low-occupancy convs: 19^2 and 37^2 maps
        shapes = [(32 * 19 * 19, 256, 9216), (32 * 19 * 19, 256, 2304), (32 * 37 * 37, 256, 9216), (32 * 37 * 37, 256, 2304)]
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in shapes:
    A = torch.randn(M, K if not conv else K // 9, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    f32 = epi in (_lib.EPI_SCALE_RES_F32, _lib.EPI_BIAS_F32)
    out = torch.zeros(M, N, dtype=torch.float32 if f32 else torch.float16, device="cuda")
    kw = dict(M=M, N=N, K=K, bias=bias)
    if conv:
        hw = int(round((M // 32) ** 0.5))
        kw.update(conv=(32, hw, hw, K // 9, hw, hw, 1), relu_in=os.environ.get("AB_RELU", "1") == "1")
        if epi == _lib.EPI_RES_F16:
            kw.update(res=torch.randn(M, N, device="cuda", generator=g).half())
    if epi in (_lib.EPI_SCALE_RES_F32,):
        kw.update(res=out, gamma=torch.ones(N, device="cuda"))
    if epi == _lib.EPI_SCALE_RES_SPLIT:
        lo = torch.zeros(M, N, dtype=torch.float16, device="cuda")
        kw.update(res=out, res2=lo, out2=lo, gamma=torch.ones(N, device="cuda") * 1e-3,
                  stats=torch.zeros(N // 64, M, 2, device="cuda"))
        if os.environ.get("AB_POS", "1") == "1":          # re-centring rows, as the model passes them
            kw.update(pos=torch.zeros(M, 2, device="cuda"))
    if epi in (_lib.EPI_LN_BIAS_F16, _lib.EPI_LN_GELU_F16):
        kw.update(gamma=torch.ones(N, device="cuda"), stats=torch.ones(M, 2, device="cuda"))
    ts = {v: [] for v in variants}
    for rep in range(5):
        for v in variants:
            _lib.lib.vda_gemm_set_variant(v)
            ops.gemm(A, W, out, epi, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(A, W, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[v].append(e0.elapsed_time(e1) / 5)
    line = f"M={M} N={N} K={K} epi={epi}: "
    for v in variants:
        t = sorted(ts[v])[len(ts[v]) // 2]
        line += f" v{v}: {t*1e3:6.1f} us ({2.0*M*N*K/t/1e9:5.0f} TF/s)"
    print(line, flush=True)
_lib.lib.vda_gemm_set_variant(-1)
