"""Thin tensor -> pointer wrappers over the C ABI (include/vda.h).

torch is used for device memory and the current HIP stream only; every op here
is one call into libvda_hip.so and nothing else. Shapes are validated twice:
here (tensor metadata the C side cannot see) and in the library (geometry).

Two operand precisions share these wrappers, selected by the dtype of the activation tensors handed in: fp16
(the *_f16 entry points: the reference's autocast path) and fp32 (the *_f32 twins: its fp32=True path).
"""
import ctypes as C

import torch

import os
import sys

from . import _lib
from ._lib import GemmArgs, lib
from ._lib import check as _check

_TRACE = os.environ.get("VDA_TRACE_SYNC") == "1"     # debug: sync + log after every launch (finds a faulting kernel)


def check(rc, what=""):
    _check(rc, what)
    if _TRACE:
        sys.stderr.write(f"[vda] {what} ...")
        sys.stderr.flush()
        torch.cuda.synchronize()
        sys.stderr.write(" ok\n")
        sys.stderr.flush()

F16, F32 = torch.float16, torch.float32


def _stream(t=None):
    """The current stream of the device the operands live on (NOT of whichever device happens to be current)."""
    dev = t.device if t is not None else None
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _act(t, name):
    if t is None:
        return
    if not t.is_cuda or t.dtype not in (F16, F32) or not t.is_contiguous():
        raise ValueError(f"{name}: expected a contiguous cuda fp16 / fp32 activation, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t, dtype, name):
    if t is None:
        return
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise ValueError(f"{name}: expected contiguous cuda {dtype}, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")


# Optional per-launch timing (bench.py): when PROFILE is a GemmProfile, every `every`-th GEMM launch of each
# (shape, epilogue) is bracketed by two events recorded on the launch stream; all launches are counted. Bracketing
# every launch costs ~1 ms per ViT-L clip in event packets (measured), 1 in 4 keeps the timed region honest.
class GemmProfile:
    def __init__(self, every=4):
        self.every = every
        self.seen = {}          # (M, N, K, epilogue, a_mode) -> launches so far
        self.launches = {}      # kernel name -> [launches, algorithmic flops]
        self.samples = []       # (kernel name, algorithmic flops, start event, stop event)


PROFILE = None

_zero_pages = {}


def zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(256, dtype=torch.uint8, device=device)
        _zero_pages[device] = z
    return z


def gemm(A, W, out, epi, *, M, N, K, lda=None, ldc=None, bias=None, res=None, res2=None, gamma=None, pos=None,
         relu_in=False, conv=None, P=0, convt=None, out2=None, stats=None, sched=None, stats_ld=0, tile_rows=0):
    """out = epilogue(A[M,K] W[N,K]^T). conv = (B,H,W,Cin,Ho,Wo,stride) switches A to the
    implicit 3x3 window of an NHWC tensor; convt = (k, h, w, Cout) for VDA_EPI_CONVT_F16."""
    _act(A, "A"), _req(W, A.dtype, "W"), _req(bias, F32, "bias"), _req(gamma, F32, "gamma"), _req(pos, F32, "pos")
    fn, fname = (lib.vda_gemm_f32, "vda_gemm_f32") if A.dtype == F32 else (lib.vda_gemm_f16, "vda_gemm_f16")
    a = GemmArgs()
    a.A, a.W, a.bias, a.out = _p(A), _p(W), _p(bias), _p(out)
    a.res, a.res2, a.gamma, a.pos = _p(res), _p(res2), _p(gamma), _p(pos)
    _req(stats, F32, "stats")
    a.out2, a.stats, a.sched = _p(out2), _p(stats), _p(sched)
    a.zero_page = _p(zero_page(A.device))
    a.M, a.N, a.K = M, N, K
    a.lda = K if lda is None else lda
    a.ldc = N if ldc is None else ldc
    a.a_mode = _lib.A_DENSE if conv is None else _lib.A_CONV3X3
    a.epilogue = epi
    a.relu_in = int(relu_in)
    if conv is not None:
        a.cB, a.cH, a.cW, a.cCin, a.cHo, a.cWo, a.cStride = conv
    a.P = P
    a.stats_ld, a.tile_rows = stats_ld, tile_rows      # a row range of a larger GEMM (vda_gemm_row_range's rules)
    if convt is not None:
        a.tK, a.tH, a.tW, a.tCout = convt
    if W.numel() < N * K:
        raise ValueError("W smaller than N*K")
    prof = PROFILE
    if prof is None:
        check(fn(C.byref(a), _stream(A)), f"{fname} M={M} N={N} K={K} epi={epi} conv={conv} lda={a.lda} ldc={a.ldc}" if _TRACE else fname)
        return
    key = (M, N, K, epi, a.a_mode)
    n = prof.seen.get(key, 0)
    prof.seen[key] = n + 1
    timed = n % prof.every == 0
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(fn(C.byref(a), _stream(A)), fname)
    name = lib.vda_gemm_last_kernel().decode() if A.dtype == F16 else "gemm_f32_kernel"
    tot = prof.launches.setdefault(name, [0, 0.0])
    tot[0] += 1
    tot[1] += 2.0 * M * N * K
    if timed:
        e1.record()
        prof.samples.append((name, 2.0 * M * N * K, e0, e1))


def split_stats(x, hi, lo, stat, eps, rows, D, center=False):
    """fp32 rows -> the two fp16 planes of the split residual stream (x = hi + lo) and stat[r] = (mean, rstd).
    center=True: hi + lo = x - mean(x), stat[r] = (0, rstd) (the model's entry: the stream relative to each token's mean)."""
    _req(x, F32, "x"), _req(hi, F16, "hi"), _req(lo, F16, "lo"), _req(stat, F32, "stat")
    fn = lib.vda_split_center_stats_f32 if center else lib.vda_split_stats_f32
    check(fn(_p(x), _p(hi), _p(lo), _p(stat), eps, rows, D, _stream(x)), "vda_split_stats_f32")


def ln_stats_finalize(partial, stat, eps, rows, np_, overflow=None):
    """partial[np, r, 2] (sum, centred sum of squares per 64 columns) -> stat[r] = (mean, rstd)."""
    _req(partial, F32, "partial"), _req(stat, F32, "stat")
    check(lib.vda_ln_stats_finalize(_p(partial), _p(stat), eps, rows, np_, _p(overflow), _stream(partial)), "vda_ln_stats_finalize")


def layernorm_split(hi, lo, out, w, b, eps, rows, D, group=0, skip=0):
    """LayerNorm of x = hi + lo (fp32 statistics) -> fp16."""
    _req(hi, F16, "hi"), _req(lo, F16, "lo"), _req(out, F16, "out"), _req(w, F32, "w"), _req(b, F32, "b")
    check(lib.vda_layernorm_split_f16(_p(hi), _p(lo), _p(out), _p(w), _p(b), eps, rows, D, group, skip, _stream(hi)), "vda_layernorm_split_f16")


def fold_ln_weight(W, bias, ln_w, ln_b, Wf, c1, c2, N, K):
    """Wf = fp16(W * ln_w[None, :]), c1 = Wf.float().sum(1), c2 = bias + W @ ln_b (pack time)."""
    for t, n in ((W, "W"), (bias, "bias"), (ln_w, "ln_w"), (ln_b, "ln_b"), (c1, "c1"), (c2, "c2")):
        _req(t, F32, n)
    _req(Wf, F16, "Wf")
    check(lib.vda_fold_ln_weight(_p(W), _p(bias), _p(ln_w), _p(ln_b), _p(Wf), _p(c1), _p(c2), N, K, _stream(W)), "vda_fold_ln_weight")


def mlp_permute_w2(w2, w2p, D, hidden):
    """fc2's weight [D, hidden] with its hidden columns in the order the fused MLP kernel's register-resident activations have."""
    _req(w2, F16, "w2"), _req(w2p, F16, "w2p")
    check(lib.vda_mlp_permute_w2_f16(_p(w2), _p(w2p), D, hidden, _stream(w2)), "vda_mlp_permute_w2_f16")


def mlp_fused(hi_in, stats, w1, c1, c2, w2p, b2, gamma, hi, lo, part, M, D, hidden, stats_ld=None):
    """hi + lo += gamma * (fc2(GELU(fc1(LayerNorm(hi + lo)))) + b2) in one kernel (vda_mlp_fused_f16: see include/vda.h)."""
    for t, n in ((hi_in, "hi_in"), (w1, "w1"), (w2p, "w2p"), (hi, "hi"), (lo, "lo")):
        _req(t, F16, n)
    for t, n in ((stats, "stats"), (c1, "c1"), (c2, "c2"), (b2, "b2"), (gamma, "gamma"), (part, "part")):
        _req(t, F32, n)
    check(lib.vda_mlp_fused_f16(_p(hi_in), _p(stats), _p(w1), _p(c1), _p(c2), _p(w2p), _p(b2), _p(gamma), _p(hi), _p(lo), _p(part), M, D, hidden,
                                M if stats_ld is None else stats_ld, _stream(hi)), "vda_mlp_fused_f16")


def layernorm(x, out, w, b, eps, rows, D, group=0, skip=0, pe=None, pe_rows_per_step=0, pe_steps=0):
    _req(x, F32, "x"), _act(out, "out"), _req(w, F32, "w"), _req(b, F32, "b"), _req(pe, F32, "pe")
    fn = lib.vda_layernorm_f32_f32 if out.dtype == F32 else lib.vda_layernorm_f32_f16
    check(fn(_p(x), _p(out), _p(w), _p(b), eps, rows, D, group, skip, _p(pe), pe_rows_per_step, pe_steps, _stream(x)), "vda_layernorm")


def layernorm_residual(x, y, gamma, out, w, b, eps, rows, D, group=0, skip=0):
    """x += gamma * y (x fp32 in place, y fp16), out = LayerNorm(x) fp16."""
    _req(x, F32, "x"), _req(y, F16, "y"), _req(gamma, F32, "gamma"), _req(out, F16, "out"), _req(w, F32, "w"), _req(b, F32, "b")
    if x.numel() < rows * D or y.numel() < rows * D:
        raise ValueError("layernorm_residual buffers too small")
    check(lib.vda_layernorm_residual_f32_f16(_p(x), _p(y), _p(gamma), _p(out), _p(w), _p(b), eps, rows, D, group, skip, _stream(x)),
          "vda_layernorm_residual_f32_f16")


def groupnorm(x, out, w, b, eps, frames, hw, Cc, groups, partial, chunks):
    _act(x, "x"), _req(out, x.dtype, "out"), _req(w, F32, "w"), _req(b, F32, "b"), _req(partial, F32, "partial")
    if partial.numel() < frames * chunks * groups * 2:
        raise ValueError("groupnorm workspace too small")
    fn = lib.vda_groupnorm_nhwc_f32 if x.dtype == F32 else lib.vda_groupnorm_nhwc_f16
    check(fn(_p(x), _p(out), _p(w), _p(b), eps, frames, hw, Cc, groups, _p(partial), chunks, _stream(x)), "vda_groupnorm_nhwc")


def attention(qkv, out, B, N, heads):
    _act(qkv, "qkv"), _req(out, qkv.dtype, "out")
    if qkv.numel() < B * N * 3 * heads * 64 or out.numel() < B * N * heads * 64:
        raise ValueError("attention buffers too small")
    fn = lib.vda_attention_f32 if qkv.dtype == F32 else lib.vda_attention_f16
    check(fn(_p(qkv), _p(out), B, N, heads, _stream(qkv)), "vda_attention")


def temporal_attention(qkv, out, T, hw, Cc, heads=8):
    _act(qkv, "qkv"), _req(out, qkv.dtype, "out")
    if qkv.numel() < T * hw * 3 * Cc or out.numel() < T * hw * Cc:
        raise ValueError("temporal attention buffers too small")
    fn = lib.vda_temporal_attention_f32 if qkv.dtype == F32 else lib.vda_temporal_attention_f16
    check(fn(_p(qkv), _p(out), T, hw, Cc, heads, _stream(qkv)), "vda_temporal_attention")


def rope_qk(qkv, T, hw, Cc):
    """pe='rope': rotate the q and k thirds of qkv [T*hw, 3*C] in place (motion_module/attention.py:403-429)."""
    _act(qkv, "qkv")
    if qkv.numel() < T * hw * 3 * Cc:
        raise ValueError("rope_qk buffer too small")
    fn = lib.vda_rope_qk_f32 if qkv.dtype == F32 else lib.vda_rope_qk_f16
    check(fn(_p(qkv), T, hw, Cc, _stream(qkv)), "vda_rope_qk")


def bilinear_nhwc(x, out, B, h, w, H, W, Cc, add=None):
    _act(x, "x"), _req(out, x.dtype, "out"), _req(add, x.dtype, "add")
    fn = lib.vda_bilinear_nhwc_f32 if x.dtype == F32 else lib.vda_bilinear_nhwc_f16
    check(fn(_p(x), _p(out), _p(add), B, h, w, H, W, Cc, _stream(x)), "vda_bilinear_nhwc")


def bilinear_plane(x, out, B, h, w, H, W, relu=False):
    _req(x, F32, "x"), _req(out, F32, "out")
    check(lib.vda_bilinear_plane_f32(_p(x), _p(out), B, h, w, H, W, 1 if relu else 0, _stream(x)), "vda_bilinear_plane_f32")


def patchify(x, out, B, H, W, Kpad):
    _req(x, F32, "x"), _act(out, "out")
    fn = lib.vda_patchify_f32_f32 if out.dtype == F32 else lib.vda_patchify_f32_f16
    check(fn(_p(x), _p(out), B, H, W, Kpad, _stream(x)), "vda_patchify")


def pos_embed_resample(pe, out, g, ph, pw, D):
    """dinov2.py:185-210: pe fp32 [1 + g*g, D] -> out fp32 [1 + ph*pw, D]."""
    _req(pe, F32, "pe"), _req(out, F32, "out")
    if pe.numel() < (1 + g * g) * D or out.numel() < (1 + ph * pw) * D:
        raise ValueError("pos_embed_resample buffers too small")
    check(lib.vda_pos_embed_resample_f32(_p(pe), _p(out), g, ph, pw, D, _stream(pe)), "vda_pos_embed_resample_f32")


def cls_rows(tok, cls, pos, B, P, D):
    _req(tok, F32, "tok"), _req(cls, F32, "cls"), _req(pos, F32, "pos")
    check(lib.vda_cls_rows_f32(_p(tok), _p(cls), _p(pos), B, P, D, _stream(tok)), "vda_cls_rows_f32")


def readout_concat(tok, out, frames, P, D):
    """[frames*(P+1), D] (cls first) -> [frames*P, 2D] = cat(patch token, the frame's cls token)."""
    _act(tok, "tok"), _req(out, tok.dtype, "out")
    if tok.numel() < frames * (P + 1) * D or out.numel() < frames * P * 2 * D:
        raise ValueError("readout_concat buffers too small")
    fn = lib.vda_readout_concat_f32 if tok.dtype == F32 else lib.vda_readout_concat_f16
    check(fn(_p(tok), _p(out), frames, P, D, _stream(tok)), "vda_readout_concat")


def head_out(x, w, bias, out, rows, Cpad):
    _act(x, "x"), _req(w, F32, "w"), _req(out, F32, "out")
    fn = lib.vda_head_out_f32_f32 if x.dtype == F32 else lib.vda_head_out_f16_f32
    check(fn(_p(x), _p(w), float(bias), _p(out), rows, Cpad, _stream(x)), "vda_head_out")


def depth_tail(x, w2, b2, w3, b3, out, B, h, w, H, W, Cc):
    """[resize h x w -> H x W] + conv3x3(C->32)+ReLU + conv1x1(32->1)+ReLU, NHWC fp16 in, fp32 [B,H,W] out."""
    _req(x, F16, "x"), _req(w2, F16, "w2"), _req(b2, F32, "b2"), _req(w3, F32, "w3"), _req(out, F32, "out")
    if x.numel() < B * h * w * Cc or out.numel() < B * H * W or w2.numel() < 32 * 9 * Cc:
        raise ValueError("depth_tail buffers too small")
    check(lib.vda_depth_tail_f16(_p(x), _p(w2), _p(b2), _p(w3), float(b3), _p(out), _p(zero_page(x.device)), B, h, w, H, W, Cc,
                                 _stream(x)), "vda_depth_tail_f16")


def conv3x3_up2(x, w, bias, out, B, h, wd, Cc, N, ldc):
    """out [B,2h,2w,ldc] = conv3x3(bilinear 2x, align_corners, of x [B,h,wd,C]) + bias: output_conv1 over refinenet1's upsample."""
    _req(x, F16, "x"), _req(w, F16, "w"), _req(out, F16, "out")
    if bias is not None:
        _req(bias, F32, "bias")
    if x.numel() < B * h * wd * Cc or out.numel() < B * 4 * h * wd * ldc or w.numel() < N * 9 * Cc:
        raise ValueError("conv3x3_up2 buffers too small")
    check(lib.vda_conv3x3_up2_f16(_p(x), _p(w), _p(bias) if bias is not None else None, _p(out), B, h, wd, Cc, N, ldc, _stream(x)),
          "vda_conv3x3_up2_f16")


def normalize_u8(frames, out, n, H, W):
    _req(frames, torch.uint8, "frames"), _req(out, F32, "out")
    check(lib.vda_normalize_u8_f32(_p(frames), _p(out), n, H, W, _stream(frames)), "vda_normalize_u8_f32")


def gather_normalize_u8(video, idx, out, n, H, W):
    _req(video, torch.uint8, "video"), _req(idx, torch.int32, "idx"), _req(out, F32, "out")
    if idx.numel() < n or out.numel() < n * 3 * H * W:
        raise ValueError("gather_normalize buffers too small")
    check(lib.vda_gather_normalize_u8_f32(_p(video), _p(idx), _p(out), n, video.shape[0], H, W, _stream(video)), "vda_gather_normalize_u8_f32")


def gather_resize_normalize_u8(video, idx, out, n, H0, W0, H, W):
    """Window gather + bicubic resize to the network size + normalise: uint8 [N,H0,W0,3] -> fp32 [n,3,H,W]."""
    _req(video, torch.uint8, "video"), _req(idx, torch.int32, "idx"), _req(out, F32, "out")
    if idx.numel() < n or out.numel() < n * 3 * H * W or video.numel() < video.shape[0] * H0 * W0 * 3:
        raise ValueError("gather_resize_normalize buffers too small")
    check(lib.vda_gather_resize_normalize_u8_f32(_p(video), _p(idx), _p(out), n, video.shape[0], H0, W0, H, W, _stream(video)),
          "vda_gather_resize_normalize_u8_f32")


LSQ_BLOCKS = 256


def lsq_scale_shift(pred, target, workspace, scale_shift):
    """scale_shift[0..1] (device fp32) = least-squares fit target ~ scale*pred + shift over all elements."""
    _req(pred, F32, "pred"), _req(target, F32, "target"), _req(workspace, torch.float64, "workspace"), _req(scale_shift, F32, "scale_shift")
    if pred.numel() != target.numel() or workspace.numel() < 4 * LSQ_BLOCKS or scale_shift.numel() < 2:
        raise ValueError("lsq_scale_shift: mismatched sizes")
    check(lib.vda_lsq_scale_shift_f32(_p(pred), _p(target), pred.numel(), _p(workspace), LSQ_BLOCKS, _p(scale_shift), _stream(pred)),
          "vda_lsq_scale_shift_f32")


def stitch_window(win, scale_shift, chunk, tail, ref1, px, wts):
    """Affine + clamp + cross-fade + append of one window (k > 0); see include/vda.h."""
    for t, nm in ((win, "win"), (scale_shift, "scale_shift"), (chunk, "chunk"), (tail, "tail"), (ref1, "ref1"), (wts, "wts")):
        _req(t, F32, nm)
    if win.numel() < 32 * px or chunk.numel() < 22 * px or tail.numel() < 8 * px or ref1.numel() < px or wts.numel() < 16:
        raise ValueError("stitch_window buffers too small")
    check(lib.vda_stitch_window_f32(_p(win), _p(scale_shift), _p(chunk), _p(tail), _p(ref1), px, _p(wts), _stream(win)), "vda_stitch_window_f32")


def affine_clamp(x, scale_shift, out):
    """out = max(x * scale + shift, 0) elementwise (the stitch's alignment of a frame), device scale_shift[2]."""
    _req(x, F32, "x"), _req(scale_shift, F32, "scale_shift"), _req(out, F32, "out")
    if out.numel() < x.numel() or scale_shift.numel() < 2:
        raise ValueError("affine_clamp buffers too small")
    check(lib.vda_affine_clamp_f32(_p(x), _p(scale_shift), _p(out), x.numel(), _stream(x)), "vda_affine_clamp_f32")


# ---------------------------------------------------------------------------
# Weight layouts the kernels expect (done once at load time, on the host or device)
# ---------------------------------------------------------------------------
def pad_to(n, m=64):
    return (n + m - 1) // m * m


def pack_linear(w, n_pad=None, k_pad=None, dtype=F16):
    """[N,K] fp32 -> `dtype` [Npad,Kpad], zero padded."""
    N, K = w.shape
    n_pad = N if n_pad is None else n_pad
    k_pad = K if k_pad is None else k_pad
    o = torch.zeros(n_pad, k_pad, dtype=dtype, device=w.device)
    o[:N, :K] = w.to(dtype)
    return o


def pack_conv3x3(w, cout_pad=None, cin_pad=None, dtype=F16):
    """Conv2d weight [Cout,Cin,3,3] -> `dtype` [Coutpad, 9*Cinpad] with K ordered (ky,kx,ci)."""
    Co, Ci = w.shape[:2]
    cout_pad = Co if cout_pad is None else cout_pad
    cin_pad = Ci if cin_pad is None else cin_pad
    o = torch.zeros(cout_pad, 3, 3, cin_pad, dtype=dtype, device=w.device)
    o[:Co, :, :, :Ci] = w.permute(0, 2, 3, 1).to(dtype)
    return o.reshape(cout_pad, 9 * cin_pad).contiguous()


def pack_convt(w, bias, cpad, dtype=F16):
    """ConvTranspose2d (k == stride) weight [Cin,Cout,k,k] -> `dtype` [(ky,kx,co) = k*k*cpad, cpad(ci)],
    bias expanded to the same row order."""
    Ci, Co, k, _ = w.shape
    o = torch.zeros(k, k, cpad, cpad, dtype=dtype, device=w.device)
    o[:, :, :Co, :Ci] = w.permute(2, 3, 1, 0).to(dtype)
    bb = torch.zeros(k, k, cpad, dtype=F32, device=w.device)
    bb[:, :, :Co] = bias.to(F32)
    return o.reshape(k * k * cpad, cpad).contiguous(), bb.reshape(-1).contiguous()


def pack_geglu(w, bias, dtype=F16):
    """GEGLU proj [8C, C]: rows [value(4C) | gate(4C)] -> interleaved [16 value | 16 gate] per 32 rows."""
    two_n, K = w.shape
    n = two_n // 2
    assert n % 16 == 0
    wv, wg = w[:n].reshape(n // 16, 16, K), w[n:].reshape(n // 16, 16, K)
    wi = torch.stack((wv, wg), dim=1).reshape(two_n, K)
    bv, bg = bias[:n].reshape(n // 16, 16), bias[n:].reshape(n // 16, 16)
    bi = torch.stack((bv, bg), dim=1).reshape(two_n)
    return wi.to(dtype).contiguous(), bi.to(F32).contiguous()
