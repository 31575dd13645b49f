"""Per-kernel parity of the fp32-operand path (the reference's fp32=True, video_depth.py:203-205): every *_f32 C-ABI entry
point against a plain torch fp32 CPU evaluation of the same op - plus the two bicubic kernels (pos-embed grid, preprocessing
resize), which are fp32 in both paths.

Tolerances: fp32 MFMA is bit-for-bit an fmaf chain, so only summation ORDER differs from torch's CPU kernels:
max|y - ref| <= atol + rtol*|ref| with rtol = atol = 2e-5 for GEMM-like ops (K up to ~1500) and 1e-5 elsewhere."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

F16, F32 = torch.float16, torch.float32


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU; run them with -m gpu on the MI355X box only")
    from video_depth_anything_amd import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(y, ref, rtol=2e-5, atol=2e-5, what=""):
    y = y.detach().float().cpu()
    ref = ref.float()
    err = (y - ref).abs()
    bad = err > atol + rtol * ref.abs()
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} out of tolerance, max err {float(err.max()):.4g} (ref absmax {float(ref.abs().max()):.4g})"


def dev(t):
    return t.cuda()


# ---------------------------------------------------------------- GEMM epilogues (dense A)
@pytest.mark.parametrize("M,N,K", [(300, 192, 128), (128, 64, 64), (1000, 48, 192), (257, 384, 1536), (70, 32, 16), (515, 1024, 1024)])
def test_gemm_f32_bias(ops, M, N, K):
    from video_depth_anything_amd import _lib
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    out = torch.full((M, N), float("nan"), dtype=F32, device="cuda")
    ops.gemm(dev(A), dev(W), out, _lib.EPI_BIAS_F16, M=M, N=N, K=K, bias=dev(b))
    close(out, A @ W.t() + b, what=f"gemm_f32 {M}x{N}x{K}")


def test_gemm_f32_k_order_is_a_permutation_not_a_drop():
    """The kernel feeds k in a permuted order inside each 16-wide step: with an asymmetric integer-valued A and W the result
    must be EXACT (integers sum exactly in fp32), which a dropped or doubled k would not be."""
    from video_depth_anything_amd import _lib, ops
    M, N, K = 96, 64, 48
    A = (torch.arange(M * K).reshape(M, K) % 7 - 3).float()
    W = (torch.arange(N * K).reshape(N, K) % 5 - 2).float() * (1 + torch.arange(K) % 3)
    out = torch.empty(M, N, dtype=F32, device="cuda")
    ops.gemm(A.cuda(), W.cuda(), out, _lib.EPI_BIAS_F32, M=M, N=N, K=K)
    assert torch.equal(out.cpu(), A @ W.t())


def test_gemm_f32_activations(ops):
    from video_depth_anything_amd import _lib
    M, N, K = 515, 256, 320
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6)
    base = A @ W.t() + b
    for epi, fn in ((_lib.EPI_BIAS_GELU_F16, F.gelu), (_lib.EPI_BIAS_RELU_F16, F.relu), (_lib.EPI_BIAS_F32, lambda x: x)):
        out = torch.empty(M, N, dtype=F32, device="cuda")
        ops.gemm(dev(A), dev(W), out, epi, M=M, N=N, K=K, bias=dev(b))
        close(out, fn(base), what=f"epilogue {epi}")


def test_gemm_f32_scale_residual_and_two_residuals(ops):
    from video_depth_anything_amd import _lib
    M, N, K = 777, 384, 384
    A, W = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=K ** -0.5)
    b, gamma, res, res2 = rnd(N, seed=9), rnd(N, seed=10), rnd(M, N, seed=11, scale=3.0), rnd(M, N, seed=12)
    x = dev(res.clone())
    ops.gemm(dev(A), dev(W), x, _lib.EPI_SCALE_RES_F32, M=M, N=N, K=K, bias=dev(b), gamma=dev(gamma), res=x)       # in place
    close(x, res + gamma * (A @ W.t() + b), what="scale+residual")
    out = torch.empty(M, N, dtype=F32, device="cuda")
    ops.gemm(dev(A), dev(W), out, _lib.EPI_SCALE_RES_F32_H, M=M, N=N, K=K, bias=dev(b), res=dev(res))
    close(out, res + A @ W.t() + b, what="residual, separate out")
    ops.gemm(dev(A), dev(W), out, _lib.EPI_RES_F16, M=M, N=N, K=K, bias=dev(b), res=dev(res), res2=dev(res2))
    close(out, A @ W.t() + b + res + res2, what="two residuals")


def test_gemm_f32_geglu(ops):
    from video_depth_anything_amd import _lib
    M, Cc = 333, 64
    A = rnd(M, Cc, seed=17)
    w, b = rnd(8 * Cc, Cc, seed=18, scale=Cc ** -0.5), rnd(8 * Cc, seed=19)
    wi, bi = ops.pack_geglu(w, b, dtype=F32)
    out = torch.empty(M, 4 * Cc, dtype=F32, device="cuda")
    ops.gemm(dev(A), dev(wi), out, _lib.EPI_GEGLU_F16, M=M, N=8 * Cc, K=Cc, ldc=4 * Cc, bias=dev(bi))
    val, gate = (A @ w.t() + b).chunk(2, dim=-1)
    close(out, val * F.gelu(gate), what="geglu")


def test_gemm_f32_patch_embed(ops):
    from video_depth_anything_amd import _lib
    B, H, W_, D = 3, 42, 56, 128
    P = (H // 14) * (W_ // 14)
    x = rnd(B, 3, H, W_, seed=20)
    w, b = rnd(D, 3, 14, 14, seed=21, scale=588 ** -0.5), rnd(D, seed=22)
    pos, cls = rnd(P + 1, D, seed=23), rnd(D, seed=24)
    Kp = 640
    a = torch.zeros(B * P, Kp, dtype=F32, device="cuda")
    ops.patchify(dev(x), a, B, H, W_, Kp)
    tok = torch.full((B, P + 1, D), float("nan"), dtype=F32, device="cuda")
    ops.gemm(a, dev(ops.pack_linear(w.reshape(D, 588), k_pad=Kp, dtype=F32)), tok, _lib.EPI_PATCH_F32, M=B * P, N=D, K=Kp, bias=dev(b),
             pos=dev(pos), P=P)
    ops.cls_rows(tok, dev(cls), dev(pos), B, P, D)
    ref = torch.cat((cls.expand(B, 1, D), F.conv2d(x, w, b, stride=14).flatten(2).transpose(1, 2)), dim=1) + pos
    close(tok, ref, what="patch embed tokens")


@pytest.mark.parametrize("k", [2, 4])
def test_gemm_f32_convtranspose(ops, k):
    from video_depth_anything_amd import _lib
    B, h, w_, Cc, Cp = 2, 5, 7, 48, 64
    x = rnd(B, Cc, h, w_, seed=25)
    wt, b = rnd(Cc, Cc, k, k, seed=26, scale=Cc ** -0.5), rnd(Cc, seed=27)
    xin = torch.zeros(B, h, w_, Cp)
    xin[..., :Cc] = x.permute(0, 2, 3, 1)
    wp, bp = ops.pack_convt(wt, b, Cp, dtype=F32)
    out = torch.empty(B, h * k, w_ * k, Cp, dtype=F32, device="cuda")
    ops.gemm(dev(xin), dev(wp), out, _lib.EPI_CONVT_F16, M=B * h * w_, N=k * k * Cp, K=Cp, ldc=Cp, bias=dev(bp), convt=(k, h, w_, Cp))
    close(out[..., :Cc], F.conv_transpose2d(x, wt, b, stride=k).permute(0, 2, 3, 1), what=f"convT k={k}")
    assert float(out[..., Cc:].abs().max()) == 0.0, "pad channels must stay zero"


@pytest.mark.parametrize("stride,relu_in,Cin,Cout,H,W_", [(1, False, 64, 64, 9, 11), (1, True, 128, 256, 12, 7), (2, False, 64, 128, 9, 9),
                                                        (1, True, 64, 32, 20, 20), (1, False, 128, 32, 37, 23)])
def test_conv3x3_f32(ops, stride, relu_in, Cin, Cout, H, W_):
    from video_depth_anything_amd import _lib
    B = 3
    x = rnd(B, Cin, H, W_, seed=28)
    w, b = rnd(Cout, Cin, 3, 3, seed=29, scale=(9 * Cin) ** -0.5), rnd(Cout, seed=30)
    Ho, Wo = (H + 2 - 3) // stride + 1, (W_ + 2 - 3) // stride + 1
    res = rnd(B, Ho, Wo, Cout, seed=31)
    out = torch.empty(B, Ho, Wo, Cout, dtype=F32, device="cuda")
    ops.gemm(dev(x.permute(0, 2, 3, 1).contiguous()), dev(ops.pack_conv3x3(w, dtype=F32)), out, _lib.EPI_RES_F16, M=B * Ho * Wo, N=Cout,
             K=9 * Cin, bias=dev(b), res=dev(res), relu_in=relu_in, conv=(B, H, W_, Cin, Ho, Wo, stride))
    xi = F.relu(x) if relu_in else x
    close(out, F.conv2d(xi, w, b, stride=stride, padding=1).permute(0, 2, 3, 1) + res, what="conv3x3 f32")


# ---------------------------------------------------------------- norms
@pytest.mark.parametrize("D,rows", [(384, 50), (1024, 37), (128, 9), (64, 130)])
def test_layernorm_f32(ops, D, rows):
    x, w, b = rnd(rows, D, seed=32, scale=3.0) + 0.5, rnd(D, seed=33) + 1.0, rnd(D, seed=34)
    out = torch.empty(rows, D, dtype=F32, device="cuda")
    ops.layernorm(dev(x), out, dev(w), dev(b), 1e-6, rows, D)
    close(out, F.layer_norm(x, (D,), w, b, 1e-6), rtol=1e-5, atol=1e-5, what="layernorm f32")


def test_layernorm_f32_drop_cls_and_pe(ops):
    D, G, nb = 128, 13, 4
    x, w, b = rnd(nb * G, D, seed=35), rnd(D, seed=36) + 1.0, rnd(D, seed=37)
    out = torch.empty(nb * (G - 1), D, dtype=F32, device="cuda")
    ops.layernorm(dev(x), out, dev(w), dev(b), 1e-6, nb * G, D, group=G, skip=1)
    close(out, F.layer_norm(x, (D,), w, b, 1e-6).reshape(nb, G, D)[:, 1:].reshape(-1, D), rtol=1e-5, atol=1e-5, what="drop cls")
    T, hw = 4, 6
    x, pe = rnd(T * hw, D, seed=38), rnd(T, D, seed=39)
    out = torch.empty(T * hw, D, dtype=F32, device="cuda")
    ops.layernorm(dev(x), out, dev(w), dev(b), 1e-5, T * hw, D, pe=dev(pe), pe_rows_per_step=hw, pe_steps=T)
    close(out, (F.layer_norm(x, (D,), w, b, 1e-5).reshape(T, hw, D) + pe[:, None]).reshape(-1, D), rtol=1e-5, atol=1e-5, what="+ pe")


@pytest.mark.parametrize("Cc,hw,frames", [(64, 37, 3), (192, 50, 2), (1024, 19, 2), (384, 361, 2)])
def test_groupnorm_f32(ops, Cc, hw, frames):
    x = rnd(frames, hw, Cc, seed=40, scale=2.0) + 0.7
    w, b = rnd(Cc, seed=41) + 1.0, rnd(Cc, seed=42)
    chunks = min(8, hw)
    part = torch.empty(frames * chunks * 32 * 2, dtype=F32, device="cuda")
    out = torch.empty(frames, hw, Cc, dtype=F32, device="cuda")
    ops.groupnorm(dev(x), out, dev(w), dev(b), 1e-6, frames, hw, Cc, 32, part, chunks)
    close(out, F.group_norm(x.permute(0, 2, 1), 32, w, b, 1e-6).permute(0, 2, 1), rtol=2e-5, atol=2e-5, what="groupnorm f32")


# ---------------------------------------------------------------- attention
def attn_ref(qkv, B, N, H):
    q, k, v = qkv.double().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    a = ((q * 0.125) @ k.transpose(-2, -1)).softmax(dim=-1)
    return (a @ v).transpose(1, 2).reshape(B, N, H * 64).float()


@pytest.mark.parametrize("B,N,H", [(2, 13, 2), (1, 64, 1), (2, 200, 3), (1, 1370, 2)])
def test_attention_f32(ops, B, N, H):
    qkv = rnd(B, N, 3 * H * 64, seed=43, scale=1.5)
    out = torch.full((B, N, H * 64), float("nan"), dtype=F32, device="cuda")
    ops.attention(dev(qkv), out, B, N, H)
    close(out, attn_ref(qkv, B, N, H), what="attention f32")


def test_attention_f32_spiked_scores(ops):
    """Online-softmax rescale path: one key dominates late in the sequence (guide rule 26)."""
    B, N, H = 1, 300, 1
    qkv = rnd(B, N, 3 * 64, seed=44)
    qkv[0, 250, 64:128] = qkv[0, 5, 0:64] * 6.0
    out = torch.empty(B, N, 64, dtype=F32, device="cuda")
    ops.attention(dev(qkv), out, B, N, H)
    close(out, attn_ref(qkv, B, N, H), what="attention f32 spiked")


@pytest.mark.parametrize("Cc,T,hw", [(64, 32, 10), (64, 4, 12), (128, 7, 5), (192, 32, 6), (384, 32, 3), (256, 32, 4), (512, 19, 3), (1024, 32, 3)])
def test_temporal_attention_f32(ops, Cc, T, hw):
    heads, d = 8, Cc // 8
    qkv = rnd(T * hw, 3 * Cc, seed=45)
    out = torch.full((T * hw, Cc), float("nan"), dtype=F32, device="cuda")
    ops.temporal_attention(dev(qkv), out, T, hw, Cc)
    x = qkv.double().reshape(T, hw, 3, heads, d).permute(2, 1, 3, 0, 4)
    a = (x[0] @ x[1].transpose(-1, -2) * d ** -0.5).softmax(dim=-1)
    close(out, (a @ x[2]).permute(2, 0, 1, 3).reshape(T * hw, Cc).float(), what="temporal attention f32")


# ---------------------------------------------------------------- resampling / layout
@pytest.mark.parametrize("h,w_,H,W_", [(19, 19, 37, 37), (5, 7, 10, 14), (8, 6, 8, 6), (3, 4, 42, 56), (10, 14, 5, 7), (11, 9, 5, 4), (7, 8, 3, 8)])
def test_bilinear_nhwc_f32(ops, h, w_, H, W_):
    B, Cc = 2, 64
    x, add = rnd(B, Cc, h, w_, seed=46), rnd(B, H, W_, Cc, seed=47)
    out = torch.empty(B, H, W_, Cc, dtype=F32, device="cuda")
    ops.bilinear_nhwc(dev(x.permute(0, 2, 3, 1).contiguous()), out, B, h, w_, H, W_, Cc, add=dev(add))
    close(out, F.interpolate(x, size=(H, W_), mode="bilinear", align_corners=True).permute(0, 2, 3, 1) + add, rtol=1e-5, atol=1e-5, what="bilinear f32")


def test_head_out_f32(ops):
    rows, Cp = 1000, 32
    x, w = rnd(rows, Cp, seed=49), rnd(32, seed=50)
    out = torch.empty(rows, dtype=F32, device="cuda")
    ops.head_out(dev(x), dev(w), 0.3, out, rows, Cp)
    close(out, F.relu(x @ w + 0.3), rtol=1e-5, atol=1e-5, what="head out f32")


# ---------------------------------------------------------------- bicubic kernels (shared by both precisions)
@pytest.mark.parametrize("ph,pw,D", [(4, 5, 128), (3, 4, 384), (37, 66, 384), (37, 37, 64), (50, 20, 64)])
def test_pos_embed_resample_matches_the_reference_call(ops, ph, pw, D):
    """dinov2.py:185-210: F.interpolate(grid, scale_factor=((ph+0.1)/37, (pw+0.1)/37), mode='bicubic', antialias=False)."""
    g = 37
    pe = rnd(1 + g * g, D, seed=80, scale=0.2)
    out = torch.full((1 + ph * pw, D), float("nan"), dtype=F32, device="cuda")
    ops.pos_embed_resample(dev(pe), out, g, ph, pw, D)
    grid = pe[1:].reshape(1, g, g, D).permute(0, 3, 1, 2)
    sy, sx = float(ph + 0.1) / g, float(pw + 0.1) / g
    ref = F.interpolate(grid, scale_factor=(sy, sx), mode="bicubic", antialias=False)
    assert ref.shape[-2:] == (ph, pw)
    ref = torch.cat((pe[:1], ref.permute(0, 2, 3, 1).reshape(-1, D)), dim=0)
    close(out, ref, rtol=1e-5, atol=2e-6, what="pos-embed bicubic")


@pytest.mark.parametrize("H0,W0,H,W", [(36, 64, 28, 56), (45, 80, 70, 126), (50, 50, 70, 70), (90, 120, 42, 56), (28, 42, 28, 42)])
def test_gather_resize_normalize_matches_the_bicubic_definition(ops, H0, W0, H, W):
    """cv2.INTER_CUBIC's definition (a = -0.75, half-pixel centres, clamped taps, no antialias) evaluated by torch on the CPU:
    F.interpolate(frame/255, mode='bicubic', align_corners=False), then ImageNet normalisation (util/transform.py:109-147).
    cv2 itself is not installed offline: parity against cv2's own arithmetic is unpinned."""
    rng = np.random.default_rng(7)
    video = rng.integers(0, 256, (6, H0, W0, 3), dtype=np.uint8)
    idx = [0, 5, 3, 3]
    out = torch.full((4, 3, H, W), float("nan"), dtype=F32, device="cuda")
    ops.gather_resize_normalize_u8(torch.from_numpy(video).cuda(), torch.tensor(idx, dtype=torch.int32, device="cuda"), out, 4, H0, W0, H, W)
    img = torch.from_numpy(video[idx]).float().div(255.0).permute(0, 3, 1, 2)
    r = F.interpolate(img, size=(H, W), mode="bicubic", align_corners=False)
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    close(out, (r - mean) / std, rtol=1e-5, atol=1e-5, what="gather + bicubic + normalise")


def test_refusals_f32(ops):
    from video_depth_anything_amd import _lib
    A, W = torch.zeros(8, 24, dtype=F32, device="cuda"), torch.zeros(8, 24, dtype=F32, device="cuda")
    out = torch.zeros(8, 8, dtype=F32, device="cuda")
    with pytest.raises(_lib.VdaError, match="multiple of 16"):
        ops.gemm(A, W, out, _lib.EPI_BIAS_F16, M=8, N=8, K=24)
    with pytest.raises(ValueError):
        ops.gemm(A, W.half(), out, _lib.EPI_BIAS_F16, M=8, N=8, K=24)        # operand dtypes must agree
