"""Repeat every GEMM/conv flavour at the ViT-L clip's shapes (and optionally the whole forward): every repeat
must be bit-identical — there are no atomics anywhere, so any difference is a race."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
F16, F32 = torch.float16, torch.float32
g = torch.Generator(device="cuda").manual_seed(0)
bad = 0
variant = int(sys.argv[1]) if len(sys.argv) > 1 else -1
_lib.lib.vda_gemm_set_variant(variant)


def rn(*s, scale=1.0, dtype=F16):
    return (torch.randn(*s, device="cuda", generator=g) * scale).to(dtype)


def repeat(name, fn, make_out, n=6):
    global bad
    ref = None
    for it in range(n):
        out = make_out()
        fn(out)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        elif not torch.equal(ref, out):
            bad += 1
            d = (ref.float() - out.float()).abs()
            print(f"{name} run {it}: {int((d > 0).sum())} elements differ, max {float(d.max()):.4g}", flush=True)
    print(name, "done", flush=True)


M = 43840
A1, A4 = rn(M, 1024), rn(M, 4096)
W = {(n, k): rn(n, k, scale=k ** -0.5) for (n, k) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]}
b4096, b1024, b3072 = rn(4096, dtype=F32), rn(1024, dtype=F32), rn(3072, dtype=F32)
tok0 = rn(M, 1024, dtype=F32)
repeat("qkv", lambda o: ops.gemm(A1, W[(3072, 1024)], o, _lib.EPI_BIAS_F16, M=M, N=3072, K=1024, bias=b3072), lambda: torch.zeros(M, 3072, dtype=F16, device="cuda"))
repeat("proj in-place", lambda o: ops.gemm(A1, W[(1024, 1024)], o, _lib.EPI_SCALE_RES_F32, M=M, N=1024, K=1024, bias=b1024, gamma=b1024, res=o), lambda: tok0.clone())
repeat("fc1 gelu", lambda o: ops.gemm(A1, W[(4096, 1024)], o, _lib.EPI_BIAS_GELU_F16, M=M, N=4096, K=1024, bias=b4096), lambda: torch.zeros(M, 4096, dtype=F16, device="cuda"))
repeat("fc2 in-place", lambda o: ops.gemm(A4, W[(1024, 4096)], o, _lib.EPI_SCALE_RES_F32, M=M, N=1024, K=4096, bias=b1024, gamma=b1024, res=o), lambda: tok0.clone())
# LayerNorm-folded forms: split residual stream in place (hi plane is `o`; lo and the partial statistics are compared too) and the
# epilogues that consume (mean, rstd)
lo0, hi0 = rn(M, 1024, scale=1e-3), rn(M, 1024)
c1, stat = rn(4096, dtype=F32), rn(M, 2, dtype=F32)
for nm, A_, K_ in (("proj split", A1, 1024), ("fc2 split", A4, 4096)):
    keep = {}
    def run(o, A_=A_, K_=K_, keep=keep):
        lo, part = lo0.clone(), torch.zeros(16, M, 2, device="cuda")
        ops.gemm(A_, W[(1024, K_)], o, _lib.EPI_SCALE_RES_SPLIT, M=M, N=1024, K=K_, bias=b1024, gamma=b1024, res=o, res2=lo, out2=lo, stats=part)
        torch.cuda.synchronize()
        for key, t in (("lo", lo), ("part", part)):
            if key in keep and not torch.equal(keep[key], t):
                global bad
                bad += 1
                print(nm, key, "differs", flush=True)
            keep.setdefault(key, t.clone())
    repeat(nm, run, lambda: hi0.clone())
repeat("qkv ln", lambda o: ops.gemm(A1, W[(3072, 1024)], o, _lib.EPI_LN_BIAS_F16, M=M, N=3072, K=1024, bias=b3072, gamma=c1, stats=stat), lambda: torch.zeros(M, 3072, dtype=F16, device="cuda"))
repeat("fc1 ln gelu", lambda o: ops.gemm(A1, W[(4096, 1024)], o, _lib.EPI_LN_GELU_F16, M=M, N=4096, K=1024, bias=b4096, gamma=c1, stats=stat), lambda: torch.zeros(M, 4096, dtype=F16, device="cuda"))
# temporal module shapes
Mt = 11552
At = rn(Mt, 1024)
wg, bg = rn(8192, 1024, scale=1 / 32), rn(8192, dtype=F32)
repeat("geglu", lambda o: ops.gemm(At, wg, o, _lib.EPI_GEGLU_F16, M=Mt, N=8192, K=1024, ldc=4096, bias=bg), lambda: torch.zeros(Mt, 4096, dtype=F16, device="cuda"))
hs0 = rn(Mt, 1024, dtype=F32)
Agg = rn(Mt, 4096)
repeat("ff2 f32res->f16", lambda o: ops.gemm(Agg, W[(1024, 4096)], o, _lib.EPI_SCALE_RES_F32_H, M=Mt, N=1024, K=4096, bias=b1024, res=hs0), lambda: torch.zeros(Mt, 1024, dtype=F16, device="cuda"))
resh = rn(Mt, 1024)
repeat("proj_out res f16", lambda o: ops.gemm(At, W[(1024, 1024)], o, _lib.EPI_RES_F16, M=Mt, N=1024, K=1024, bias=b1024, res=resh), lambda: torch.zeros(Mt, 1024, dtype=F16, device="cuda"))
repeat("proj_in f32", lambda o: ops.gemm(At, W[(1024, 1024)], o, _lib.EPI_BIAS_F32, M=Mt, N=1024, K=1024, bias=b1024), lambda: torch.zeros(Mt, 1024, dtype=F32, device="cuda"))
# convT k=4 (layer_1) and patch embed
P = 1369
Mp = 32 * P
Ap = rn(Mp, 256)
wt, bt = rn(16 * 256, 256, scale=1 / 16), rn(16 * 256, dtype=F32)
repeat("convT k4", lambda o: ops.gemm(Ap, wt, o, _lib.EPI_CONVT_F16, M=Mp, N=4096, K=256, ldc=256, bias=bt, convt=(4, 37, 37, 256)), lambda: torch.zeros(32 * 148 * 148, 256, dtype=F16, device="cuda"))
Apt, wpt, pos = rn(Mp, 640), rn(1024, 640, scale=0.04), rn(P + 1, 1024, dtype=F32)
repeat("patch", lambda o: ops.gemm(Apt, wpt, o, _lib.EPI_PATCH_F32, M=Mp, N=1024, K=640, bias=b1024, pos=pos, P=P), lambda: torch.zeros(32 * (P + 1), 1024, dtype=F32, device="cuda"))
# convs of the head (F=256)
for name, B, H, Wd, Cin, Cout, stride, epi, relu in [("rcu c1 148", 32, 148, 148, 256, 256, 1, _lib.EPI_BIAS_RELU_F16, True), ("rcu c2 148", 32, 148, 148, 256, 256, 1, _lib.EPI_RES_F16, False),
                                                     ("rn3 37", 32, 37, 37, 1024, 256, 1, _lib.EPI_BIAS_F16, False), ("resize3 s2", 32, 37, 37, 1024, 1024, 2, _lib.EPI_BIAS_F16, False),
                                                     ("oc1 296", 32, 296, 296, 256, 128, 1, _lib.EPI_BIAS_F16, False)]:
    x = rn(B, H, Wd, Cin)
    w, b = rn(Cout, 9 * Cin, scale=(9 * Cin) ** -0.5), rn(Cout, dtype=F32)
    Ho, Wo = (H - 1) // stride + 1, (Wd - 1) // stride + 1
    r = rn(B * Ho * Wo, Cout)
    repeat(name, lambda o: ops.gemm(x, w, o, epi, M=B * Ho * Wo, N=Cout, K=9 * Cin, bias=b, res=r if epi == _lib.EPI_RES_F16 else None, res2=r if epi == _lib.EPI_RES_F16 else None,
                                    relu_in=relu, conv=(B, H, Wd, Cin, Ho, Wo, stride)), lambda: torch.zeros(B * Ho * Wo, Cout, dtype=F16, device="cuda"), n=4)
print("gemm repeats bad =", bad, flush=True)
if len(sys.argv) > 2:
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import state_dict_spec
    cfg = get_config("vitl"); gg = torch.Generator().manual_seed(0)
    sd = {k: (torch.randn(s, generator=gg) * (0.02 if len(s) > 1 else 0.1) + (0 if len(s) > 1 else 1)) for k, s in state_dict_spec(cfg).items()}
    m = VideoDepthAnything(encoder="vitl", features=cfg.features, out_channels=list(cfg.out_channels)); m.load_state_dict(sd); m = m.to("cuda")
    x = torch.randn(1, 32, 3, 518, 518, generator=gg).cuda()
    PE = m.python_engine()      # the Python launch sequence exposes every stage; bit-identical to vda_forward
    taps0, st0 = [], {}
    ref = PE.forward(x, taps_out=taps0, stages=st0).clone()
    taps0 = [t.clone() for t in taps0]; st0 = {k: v[0].clone() for k, v in st0.items()}
    for it in range(3):
        taps, st = [], {}
        d = PE.forward(x, taps_out=taps, stages=st)
        msg = []
        for i, t in enumerate(taps):
            if not torch.equal(t, taps0[i]): msg.append(f"tap{i}")
        for k, v in st.items():
            if not torch.equal(v[0], st0[k]): msg.append(k)
        if not torch.equal(ref, d):
            bad += 1
            print("forward run", it, "differs: max", float((ref - d).abs().max()), "first differing stages:", msg, flush=True)
    # the handle's launch sequence (LayerNorm folded into the GEMMs, the product path)
    href = m.forward(x, fp32=False).clone()
    for it in range(3):
        if not torch.equal(href, m.forward(x, fp32=False)):
            bad += 1
            print("vda_forward run", it, "differs", flush=True)
    print("vda_forward repeats done", flush=True)
print("TOTAL bad =", bad)
sys.exit(1 if bad else 0)
