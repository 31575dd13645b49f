"""Position-independence screen: A = [A1; A1] (10960 rows each, so the copies sit at different tile / wave / lane offsets);
the two halves of the output must be bit-identical for every encoder GEMM shape and epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
g = torch.Generator(device="cuda").manual_seed(0)
M1 = 10960
bad = 0
for variant in (3, 5):
    _lib.lib.vda_gemm_set_variant(variant)
    for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
        for epi in (_lib.EPI_BIAS_F16, _lib.EPI_BIAS_GELU_F16, _lib.EPI_SCALE_RES_F32):
            A1 = torch.randn(M1, K, device="cuda", generator=g).half()
            A = torch.cat([A1, A1]).contiguous()
            W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
            bias = torch.randn(N, device="cuda", generator=g)
            f32 = epi == _lib.EPI_SCALE_RES_F32
            kw = dict(M=2 * M1, N=N, K=K, bias=bias)
            if f32:
                r1 = torch.randn(M1, N, device="cuda", generator=g)
                out = torch.cat([r1, r1]).contiguous()
                kw.update(res=out, gamma=torch.rand(N, device="cuda", generator=g))
            else:
                out = torch.zeros(2 * M1, N, dtype=torch.float16, device="cuda")
            for rep in range(3):
                if f32:
                    out.copy_(torch.cat([r1, r1]))
                ops.gemm(A, W, out, epi, **kw)
                torch.cuda.synchronize()
                d = (out[:M1] != out[M1:])
                nb = int(d.sum())
                if nb:
                    rows = d.any(dim=1).nonzero().flatten()
                    print(f"variant {variant} N={N} K={K} epi={epi} rep {rep}: {nb} elements differ, rows {rows[:8].tolist()}.. ({len(rows)} rows), "
                          f"max |diff| {float((out[:M1].float() - out[M1:].float()).abs().max()):.3g}")
                    bad += 1
_lib.lib.vda_gemm_set_variant(-1)
print("dup screen bad =", bad)
