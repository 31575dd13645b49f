#!/usr/bin/env python
"""Two BUILDS of libvda_hip.so side by side in ONE process on one device: the whole forward (and, with `gemm`, the encoder / head
GEMM shapes) timed interleaved, median of repeats. Boxes of the pool differ by several per cent, so a build-to-build comparison
across two gpurun calls says nothing; this one does.
usage: lib_ab.py [vitl|vits|gemm] [path of build A, default tools/ab/libvda_base.so] [path of build B, default the in-tree library]
(build A: `git worktree add /tmp/vda_base <commit> && (cd /tmp/vda_base && python -m video_depth_anything_amd.build)` and copy its .so)"""
import importlib.util, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "video_depth_anything_amd")
what = sys.argv[1] if len(sys.argv) > 1 else "vitl"
paths = {"A": sys.argv[2] if len(sys.argv) > 2 else os.path.join(REPO, "tools", "ab", "libvda_base.so"),
         "B": sys.argv[3] if len(sys.argv) > 3 else os.path.join(PKG, "libvda_hip.so")}


def load(alias, lib_path):
    os.environ["VDA_LIB_PATH"], os.environ["VDA_LIB_TOLERANT"] = lib_path, "1"
    spec = importlib.util.spec_from_file_location(alias, os.path.join(PKG, "__init__.py"), submodule_search_locations=[PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[alias] = mod
    spec.loader.exec_module(mod)
    importlib.import_module(alias + "._lib")          # binds THIS copy of the package to lib_path (read from the environment at import)
    return mod


pk = {k: load("vda_" + k, p) for k, p in paths.items()}
imp = lambda k, name: importlib.import_module(f"vda_{k}.{name}")      # noqa: E731
for k in pk:
    assert imp(k, "_lib").LIB_PATH == paths[k]
print({k: (paths[k], imp(k, "_lib").lib.vda_abi_version()) for k in pk}, flush=True)


def timed(fn, inner):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(inner):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / inner


if what in ("vitl", "vits"):
    x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
    models, outs = {}, {}
    for k in pk:
        cfg = imp(k, "config").get_config(what)
        m = imp(k, "video_depth").VideoDepthAnything(encoder=what, features=cfg.features, out_channels=list(cfg.out_channels))
        m.load_state_dict(imp(k, "weights").synthetic_state_dict(cfg, seed=0))
        models[k] = m.to("cuda")
        outs[k] = models[k].forward(x, fp32=False).clone()
    d = (outs["A"] - outs["B"]).abs()
    print(f"max |A - B| = {float(d.max()):.3e}, rel-L1 {float(d.mean() / outs['A'].abs().mean()):.3e}, equal: {bool(torch.equal(outs['A'], outs['B']))}")
    ts = {k: [] for k in pk}
    for rep in range(7):
        for k in pk:
            ts[k].append(timed(lambda: models[k].forward(x, fp32=False), 5))
    for k in pk:
        t = sorted(ts[k])[len(ts[k]) // 2]
        print(f"{what} build {k}: {t:.3f} ms/clip ({32e3 / t:.1f} frames/s)   all: {[round(u, 2) for u in ts[k]]}", flush=True)
else:
    M = 43840
    L = imp("A", "_lib")
    cases = [("fc1 LN+GELU", M, 4096, 1024, L.EPI_LN_GELU_F16, None), ("qkv LN", M, 3072, 1024, L.EPI_LN_BIAS_F16, None), ("proj split-res", M, 1024, 1024, L.EPI_SCALE_RES_SPLIT, None),
             ("fc2 split-res", M, 1024, 4096, L.EPI_SCALE_RES_SPLIT, None), ("bias N=1024 K=1024", M, 1024, 1024, L.EPI_BIAS_F16, None),
             ("GEGLU N=8192 K=1024", 43808, 8192, 1024, L.EPI_GEGLU_F16, None), ("conv 256->256 148^2 relu", 32 * 148 * 148, 256, 2304, L.EPI_BIAS_RELU_F16, 148),
             ("conv 256->256 148^2 res", 32 * 148 * 148, 256, 2304, L.EPI_RES_F16, 148), ("conv 256->256 74^2 bias", 32 * 74 * 74, 256, 2304, L.EPI_BIAS_F16, 74)]
    g = torch.Generator(device="cuda").manual_seed(0)
    for name, Mm, N, K, epi, hw in cases:
        A = torch.randn(Mm, K if hw is None else K // 9, device="cuda", generator=g).half()
        W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
        bias = torch.randn(N, device="cuda", generator=g)
        out = torch.zeros(Mm, N if epi != L.EPI_GEGLU_F16 else N // 2, dtype=torch.float16, device="cuda")
        kw = dict(M=Mm, N=N, K=K, bias=bias)
        if epi in (L.EPI_LN_BIAS_F16, L.EPI_LN_GELU_F16):
            kw.update(gamma=torch.ones(N, device="cuda"), stats=torch.stack([torch.zeros(Mm), torch.ones(Mm)], 1).contiguous().cuda())
        if epi == L.EPI_GEGLU_F16:
            kw.update(ldc=N // 2)
        if epi == L.EPI_SCALE_RES_SPLIT:
            lo = torch.zeros(Mm, N, dtype=torch.float16, device="cuda")
            kw.update(res=out, res2=lo, out2=lo, gamma=torch.ones(N, device="cuda") * 1e-3, stats=torch.zeros(N // 64, Mm, 2, device="cuda"), pos=torch.zeros(Mm, 2, device="cuda"))
        if hw is not None:
            kw.update(conv=(32, hw, hw, K // 9, hw, hw, 1), relu_in=True)
            if epi == L.EPI_RES_F16:
                kw.update(res=torch.randn(Mm, N, device="cuda", generator=g).half())
        ts = {k: [] for k in pk}
        for rep in range(7):
            for k in pk:
                ts[k].append(timed(lambda: imp(k, "ops").gemm(A, W, out, epi, **kw), 10) * 1e3)
        med = {k: sorted(ts[k])[len(ts[k]) // 2] for k in pk}
        print(f"{name:26s} A {med['A']:7.1f} us   B {med['B']:7.1f} us   B/A {med['B'] / med['A']:.3f}", flush=True)
