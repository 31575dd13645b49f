#!/usr/bin/env python
"""GEMM / conv / attention / layernorm microbenchmarks at the ViT-L clip's shapes (run on the GPU box).
Random operands (zero-filled data reads high under DVFS); interleaved rounds in one process."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops  # noqa: E402

F16, F32 = torch.float16, torch.float32


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,1")
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    M = 43840
    shapes = [("qkv", M, 3072, 1024, _lib.EPI_BIAS_F16), ("proj", M, 1024, 1024, _lib.EPI_SCALE_RES_F32),
              ("fc1", M, 4096, 1024, _lib.EPI_BIAS_GELU_F16), ("fc2", M, 1024, 4096, _lib.EPI_SCALE_RES_F32),
              ("vits_qkv", M, 1152, 384, _lib.EPI_BIAS_F16), ("vits_fc1", M, 1536, 384, _lib.EPI_BIAS_GELU_F16),
              ("vits_fc2", M, 384, 1536, _lib.EPI_SCALE_RES_F32), ("head256", 175232, 256, 256, _lib.EPI_BIAS_F16)]
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, m, n, k, epi in shapes:
        A = (torch.randn(m, k, device="cuda", generator=g)).to(F16)
        W = (torch.randn(n, k, device="cuda", generator=g) * k ** -0.5).to(F16)
        b = torch.randn(n, device="cuda", generator=g)
        f32out = epi == _lib.EPI_SCALE_RES_F32
        out = torch.zeros(m, n, dtype=F32 if f32out else F16, device="cuda")
        res = out if f32out else None
        row = [f"{name:9s} M={m} N={n} K={k}"]
        ref = None
        for v in variants:
            _lib.lib.vda_gemm_set_variant(v)
            fn = lambda: ops.gemm(A, W, out, epi, M=m, N=n, K=k, bias=b, res=res, gamma=b if f32out else None)
            med, best = timeit(fn)
            row.append(f"v{v}: {2.0 * m * n * k / med / 1e9:7.1f} TF/s ({med * 1e3:7.1f} us)")
            if args.check and not f32out:
                out.zero_()
                fn()
                cur = out.float().clone()
                if ref is None:
                    ref = cur
                else:
                    row.append(f"maxdiff vs v{variants[0]} {float((cur - ref).abs().max()):.3g}")
        print("  ".join(row), flush=True)
    _lib.lib.vda_gemm_set_variant(-1)

    # conv 3x3 at the head's heavy shapes (ViT-L: F=256)
    for name, B, H, Wd, Cin, Cout in [("rcu148", 32, 148, 148, 256, 256), ("rcu74", 32, 74, 74, 256, 256), ("oc1_296", 32, 296, 296, 256, 128),
                                      ("oc2_518", 32, 518, 518, 128, 32)]:
        x = torch.randn(B, H, Wd, Cin, device="cuda", generator=g).to(F16)
        w = (torch.randn(Cout, 9 * Cin, device="cuda", generator=g) * (9 * Cin) ** -0.5).to(F16)
        b = torch.randn(Cout, device="cuda", generator=g)
        out = torch.zeros(B * H * Wd, Cout, dtype=F16, device="cuda")
        row = [f"{name:9s} M={B * H * Wd} N={Cout} K={9 * Cin}"]
        for v in variants:
            _lib.lib.vda_gemm_set_variant(v)
            fn = lambda: ops.gemm(x, w, out, _lib.EPI_BIAS_RELU_F16, M=B * H * Wd, N=Cout, K=9 * Cin, bias=b, relu_in=True,
                                  conv=(B, H, Wd, Cin, H, Wd, 1))
            med, best = timeit(fn, iters=5)
            row.append(f"v{v}: {2.0 * B * H * Wd * Cout * 9 * Cin / med / 1e9:7.1f} TF/s ({med * 1e3:7.1f} us)")
        print("  ".join(row), flush=True)
    _lib.lib.vda_gemm_set_variant(-1)

    # attention + layernorm at the encoder shape
    for H in (16, 6):
        qkv = torch.randn(32, 1370, 3 * H * 64, device="cuda", generator=g).to(F16)
        o = torch.empty(32, 1370, H * 64, dtype=F16, device="cuda")
        med, _ = timeit(lambda: ops.attention(qkv, o, 32, 1370, H))
        fl = 4.0 * 32 * H * 1370 * 1370 * 64
        print(f"attention heads={H}: {fl / med / 1e9:7.1f} TF/s ({med * 1e3:7.1f} us)", flush=True)
    for D in (1024, 384):
        x = torch.randn(M, D, device="cuda", generator=g)
        o = torch.empty(M, D, dtype=F16, device="cuda")
        w = torch.ones(D, device="cuda")
        med, _ = timeit(lambda: ops.layernorm(x, o, w, w, 1e-6, M, D))
        print(f"layernorm D={D}: {M * D * 6 / med / 1e6:7.1f} GB/s ({med * 1e3:7.1f} us)", flush=True)


if __name__ == "__main__":
    main()
