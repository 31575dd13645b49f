import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
F16, F32 = torch.float16, torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
def rn(*s, scale=1.0, dtype=F16): return (torch.randn(*s, device="cuda", generator=g) * scale).to(dtype)
def repeat(name, fn, make_out, n=5):
    ref = None; nb = 0
    for it in range(n):
        out = make_out(); fn(out); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        elif not torch.equal(ref, out):
            nb += 1; d = (ref.float() - out.float()).abs(); idx = (d > 0).nonzero()
            rows = idx[:, 0]
            print(f"  {name} run {it}: {int((d > 0).sum())} differ, max {float(d.max()):.3g}; rows {int(rows.min())}..{int(rows.max())}, distinct 256-row tiles {len(torch.unique(rows // 256))}, cols {int(idx[:,1].min())}..{int(idx[:,1].max())}", flush=True)
    print(name, "bad", nb, flush=True)
B, H, Wd, Cin, Cout = 32, 148, 148, 256, 256
x = rn(B, H, Wd, Cin); w, b = rn(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5), rn(Cout, dtype=F32); r = rn(B * H * Wd, Cout); r2 = rn(B * H * Wd, Cout)
Mc = B * H * Wd
for v in (1, 2, 0):
    _lib.lib.vda_gemm_set_variant(v)
    repeat(f"v{v} conv RES_F16 res+res2", lambda o: ops.gemm(x, w, o, _lib.EPI_RES_F16, M=Mc, N=Cout, K=9 * Cin, bias=b, res=r, res2=r2, conv=(B, H, Wd, Cin, H, Wd, 1)), lambda: torch.zeros(Mc, Cout, dtype=F16, device="cuda"))
    repeat(f"v{v} conv RES_F16 res only", lambda o: ops.gemm(x, w, o, _lib.EPI_RES_F16, M=Mc, N=Cout, K=9 * Cin, bias=b, res=r, conv=(B, H, Wd, Cin, H, Wd, 1)), lambda: torch.zeros(Mc, Cout, dtype=F16, device="cuda"))
    repeat(f"v{v} conv BIAS_F16", lambda o: ops.gemm(x, w, o, _lib.EPI_BIAS_F16, M=Mc, N=Cout, K=9 * Cin, bias=b, conv=(B, H, Wd, Cin, H, Wd, 1)), lambda: torch.zeros(Mc, Cout, dtype=F16, device="cuda"))
_lib.lib.vda_gemm_set_variant(1)
M = 43840
A = rn(M, 1024); W = rn(1024, 1024, scale=1 / 32); bb = rn(1024, dtype=F32); rr = rn(M, 1024); rr2 = rn(M, 1024)
repeat("v1 dense RES_F16 multi-round", lambda o: ops.gemm(A, W, o, _lib.EPI_RES_F16, M=M, N=1024, K=1024, bias=bb, res=rr, res2=rr2), lambda: torch.zeros(M, 1024, dtype=F16, device="cuda"))
