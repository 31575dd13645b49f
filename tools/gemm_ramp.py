#!/usr/bin/env python
"""Where a tile's K loop spends its cycles, K tile by K tile (vda_gemm_set_debug bits 0 + 2): cycles between the tops of consecutive
K tiles of each workgroup's second tile, median over workgroups; plus prologue / K loop / epilogue cycles per tile."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
lib = _lib.lib
g = torch.Generator(device="cuda").manual_seed(0)
M = 43840
for name, N, K, epi, bits in [("fc1 + LN + GELU", 4096, 1024, _lib.EPI_LN_GELU_F16, 0), ("fc1 blocked", 4096, 1024, _lib.EPI_LN_GELU_F16, 2), ("qkv + LN", 3072, 1024, _lib.EPI_LN_BIAS_F16, 0),
                             ("plain N=1024 K=4096", 1024, 4096, _lib.EPI_BIAS_F16, 0)]:
    A = torch.randn(M, K, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    st = torch.zeros(256 * 16 * 4 * 2 + 256 * 32, dtype=torch.int64, device="cuda")
    kw = dict(M=M, N=N, K=K, bias=bias, pos=st.view(torch.float32))
    if epi in (_lib.EPI_LN_BIAS_F16, _lib.EPI_LN_GELU_F16):
        kw.update(gamma=torch.ones(N, device="cuda"), stats=torch.stack([torch.zeros(M), torch.ones(M)], 1).contiguous().cuda())
    lib.vda_gemm_set_debug(bits)
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(50):
            ops.gemm(A, W, out, epi, **kw)
        torch.cuda.synchronize()
    lib.vda_gemm_set_debug(bits | 1 | 4)
    for _ in range(10):
        ops.gemm(A, W, out, epi, **kw)
    torch.cuda.synchronize()
    raw = st.cpu().numpy()
    s = raw[:256 * 16 * 4 * 2].reshape(256, 16, 4, 2).astype(np.float64)
    kts = raw[256 * 16 * 4 * 2:].reshape(256, 32).astype(np.float64)
    nkt = min(K // 64, 32)
    d = np.diff(kts[:, :nkt], axis=1)
    ok = (kts[:, 0] > 0)
    t = s[ok, 1]                                           # second tile of each workgroup
    pro, loop, epi_c = t[:, 1, 0] - t[:, 0, 0], t[:, 2, 0] - t[:, 1, 0], t[:, 3, 0] - t[:, 2, 0]
    first = kts[ok, 0] - t[:, 1, 0]
    print(f"{name}: per tile (cycles, median over {int(ok.sum())} workgroups): prologue {np.median(pro):.0f}  K loop {np.median(loop):.0f}  epilogue {np.median(epi_c):.0f}; "
          f"K-loop start -> top of K tile 0: {np.median(first):.0f}")
    print("   cycles per K tile:", " ".join(f"{v:.0f}" for v in np.median(d[ok], axis=0)), flush=True)
lib.vda_gemm_set_debug(0)
