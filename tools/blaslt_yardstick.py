#!/usr/bin/env python
"""Yardstick only (not on the product path): what the vendor GEMM reaches on the encoder's shapes."""
import torch
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K) in [(43840, 4096, 1024), (43840, 1024, 4096), (43840, 3072, 1024), (43840, 1024, 1024)]:
    A = torch.randn(M, K, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(N, device="cuda", generator=g).half()
    for name, fn in [("linear", lambda: torch.nn.functional.linear(A, W, b)), ("mm", lambda: A @ W.t())]:
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 10
        print(f"{name} M={M} N={N} K={K}: {t*1e3:.1f} us {2.0*M*N*K/t/1e9:.0f} TF/s", flush=True)
