"""Randomised race / correctness screen of the large-tile GEMM kernels: random (M, N, K), every run compared with a float64
reference and with its own repeat runs bitwise (a race shows as run-to-run differences or as isolated wrong tiles)."""
import os, sys, random, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 5
nshapes = int(sys.argv[2]) if len(sys.argv) > 2 else 40
random.seed(1234)
g = torch.Generator(device="cuda").manual_seed(0)
_lib.lib.vda_gemm_set_variant(variant)
bad = 0
for it in range(nshapes):
    M = random.choice([random.randint(1, 700), random.randint(700, 9000), random.randint(9000, 60000)])
    N = 8 * random.randint(1, 512)
    K = 64 * random.choice([1, 2, 3, 4, 5, 8, 16, 24, 40, 64])
    epi = random.choice([_lib.EPI_BIAS_F16, _lib.EPI_BIAS_GELU_F16, _lib.EPI_SCALE_RES_F32, _lib.EPI_BIAS_RELU_F16])
    A = torch.randn(M, K, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    f32 = epi == _lib.EPI_SCALE_RES_F32
    res0 = torch.randn(M, N, device="cuda", generator=g) if f32 else None
    gamma = torch.rand(N, device="cuda", generator=g) if f32 else None
    outs = []
    for rep in range(3):
        if f32:
            out = res0.clone()
            ops.gemm(A, W, out, epi, M=M, N=N, K=K, bias=bias, res=out, gamma=gamma)
        else:
            out = torch.full((M, N), float("nan"), dtype=torch.float16, device="cuda")
            ops.gemm(A, W, out, epi, M=M, N=N, K=K, bias=bias)
        torch.cuda.synchronize()
        outs.append(out)
    ref = A.double() @ W.double().t() + bias.double()
    if epi == _lib.EPI_BIAS_GELU_F16: ref = torch.nn.functional.gelu(ref)
    if epi == _lib.EPI_BIAS_RELU_F16: ref = ref.clamp_min(0)
    if f32: ref = res0.double() + gamma.double() * ref
    err = (outs[0].double() - ref).abs()
    tol = 3e-3 + 3e-3 * ref.abs()
    nbad = int((err > tol).sum())
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    status = "ok" if (nbad == 0 and same) else "BAD"
    if status == "BAD": bad += 1
    print(f"{status} {_lib.lib.vda_gemm_last_kernel().decode():34s} M={M:6d} N={N:5d} K={K:5d} epi={epi} max err {float(err.max()):.2e} wrong {nbad} repeat-identical {same}", flush=True)
_lib.lib.vda_gemm_set_variant(-1)
print("screen bad =", bad)
