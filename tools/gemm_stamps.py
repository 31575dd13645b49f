"""Where a tile's time goes in the 8-phase GEMM: shader-clock stamps written by the kernel itself (prologue | K loop | epilogue)."""
import os, sys, torch, numpy as np
sys.path.insert(0, "/root/repo")
from video_depth_anything_amd import _lib, ops
g = torch.Generator(device="cuda").manual_seed(0)
for (M, N, K, epi) in [(43840, 1024, 1024, 3), (43840, 1024, 4096, 3), (43840, 4096, 1024, 1), (43840, 3072, 1024, 0)]:
    A = torch.randn(M, K, device="cuda", generator=g).half(); W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    f32 = epi == 3
    out = torch.zeros(M, N, dtype=torch.float32 if f32 else torch.float16, device="cuda")
    st = torch.zeros(256 * 16 * 4, dtype=torch.int64, device="cuda")
    kw = dict(M=M, N=N, K=K, bias=bias, pos=st.view(torch.float32))
    if f32: kw.update(res=out, gamma=torch.ones(N, device="cuda"))
    _lib.lib.vda_gemm_set_variant(5 + 16 * 2)      # A/B flag bit 1 -> stamps on
    for _ in range(3): ops.gemm(A, W, out, epi, **kw)
    torch.cuda.synchronize()
    s = st.cpu().numpy().reshape(256, 16, 4).astype(np.float64)
    rounds = int((s[:, :, 0] > 0).sum(axis=1).max())
    t0 = s[:, 0, 0].min()
    f = 100e6   # s_memtime / readcyclecounter tick = 100 MHz constant clock on gfx9
    pro = (s[:, :rounds, 1] - s[:, :rounds, 0]); loop = (s[:, :rounds, 2] - s[:, :rounds, 1]); epi_t = (s[:, :rounds, 3] - s[:, :rounds, 2])
    ok = s[:, :rounds, 0] > 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ops.gemm(A, W, out, epi, **kw)
    e1.record(); torch.cuda.synchronize()
    print(f"launch (stamps on) {e0.elapsed_time(e1)/5*1e3:.1f} us", end="  ")
    print(f"M={M} N={N} K={K} epi={epi}: rounds {rounds}; per tile (kcycles): prologue {np.mean(pro[ok])/1e3:.2f}  K-loop {np.mean(loop[ok])/1e3:.2f} ({np.mean(loop[ok])/1e3/(K//64):.3f}/K-tile)  epilogue+tile-end {np.mean(epi_t[ok])/1e3:.2f}; kernel span {(s[:,:,3].max()-t0)/1e3:.1f}", flush=True)
_lib.lib.vda_gemm_set_variant(-1)
