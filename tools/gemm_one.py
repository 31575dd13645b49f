#!/usr/bin/env python
"""One GEMM shape, one kernel variant, a few launches: the target of rocprofv3 --pmc passes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops  # noqa: E402

M, N, K, variant = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 3
bcast = len(sys.argv) > 6 and sys.argv[6] == "bcast"
dbg = int(sys.argv[7]) if len(sys.argv) > 7 else 0
g = torch.Generator(device="cuda").manual_seed(0)
A = torch.randn(1 if bcast else M, K, device="cuda", generator=g).to(torch.float16)
W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).to(torch.float16)
out = torch.zeros(M, N, dtype=torch.float16, device="cuda")
_lib.lib.vda_gemm_set_variant(variant)
ts = []
for i in range(iters + 2):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(A, W, out, _lib.EPI_BIAS_F16, M=M, N=N, K=K, lda=0 if bcast else K, relu_in=dbg)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
t = sorted(ts[2:])[len(ts[2:]) // 2]
print(f"M={M} N={N} K={K} variant={variant} bcast={bcast} dbg={dbg}: {t * 1e3:.1f} us  {2.0 * M * N * K / t / 1e9:.1f} TF/s")
