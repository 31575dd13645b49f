#!/usr/bin/env python
"""A/B of vda_gemm_set_debug flag sets on the encoder / head GEMM shapes, one process, interleaved, median of repeats.
usage: gemm_dbg_ab.py flagsA,flagsB,... (e.g. 0,8: default against "drain every store at the tile start")"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
lib = _lib.lib
flags = [int(v) for v in sys.argv[1].split(",")]
g = torch.Generator(device="cuda").manual_seed(0)
M = 43840
cases = [("fc1 LN+GELU", M, 4096, 1024, _lib.EPI_LN_GELU_F16, None), ("qkv LN", M, 3072, 1024, _lib.EPI_LN_BIAS_F16, None), ("bias N=1024 K=1024", M, 1024, 1024, _lib.EPI_BIAS_F16, None),
         ("GEGLU N=8192 K=1024", 43808, 8192, 1024, _lib.EPI_GEGLU_F16, None), ("conv 256->256 148^2 relu", 32 * 148 * 148, 256, 2304, _lib.EPI_BIAS_RELU_F16, 148),
         ("conv 256->256 74^2 bias", 32 * 74 * 74, 256, 2304, _lib.EPI_BIAS_F16, 74)]
for name, Mm, N, K, epi, hw in cases:
    A = torch.randn(Mm, K if hw is None else K // 9, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(Mm, N if epi != _lib.EPI_GEGLU_F16 else N // 2, dtype=torch.float16, device="cuda")
    kw = dict(M=Mm, N=N, K=K, bias=bias)
    if epi in (_lib.EPI_LN_BIAS_F16, _lib.EPI_LN_GELU_F16):
        kw.update(gamma=torch.ones(N, device="cuda"), stats=torch.stack([torch.zeros(Mm), torch.ones(Mm)], 1).contiguous().cuda())
    if epi == _lib.EPI_GEGLU_F16:
        kw.update(ldc=N // 2)
    if hw is not None:
        kw.update(conv=(32, hw, hw, K // 9, hw, hw, 1), relu_in=True)
    ts = {f: [] for f in flags}
    ref = None
    for rep in range(7):
        for f in flags:
            lib.vda_gemm_set_debug(f)
            ops.gemm(A, W, out, epi, **kw)
            if rep == 0:
                if ref is None:
                    ref = out.clone()
                else:
                    assert torch.equal(out, ref), f"{name}: flags {f} change the result"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(A, W, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[f].append(e0.elapsed_time(e1) / 10 * 1e3)
    print(f"{name:28s}" + "".join(f"  dbg {f}: {sorted(ts[f])[len(ts[f]) // 2]:7.1f} us" for f in flags), flush=True)
lib.vda_gemm_set_debug(0)
