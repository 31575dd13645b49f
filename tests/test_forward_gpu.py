"""End-to-end parity of the HIP path on the MI355X, through the product class (VideoDepthAnything -> vda_forward).

 (a) committed golden fixtures = outputs of the reference's own modules (oracle/gen_golden.py);
 (b) the CPU oracle on seeded inputs at sizes it finishes in seconds - every BASELINE.json config's model;
 (c) BASELINE.json's full sizes through size-independent properties: bitwise run-to-run determinism, clip independence
     (B=2 == two B=1), finiteness / ReLU range, and stitched-video == per-window forward + the host stitcher.

Tolerances (relative L1 = mean|y - ref| / mean|ref| against the fp32 reference, as north_star states its bar):
  fp32=True   (fp32 operands on exact-fp32 MFMA; the reference's --fp32 path):  <= 1e-3 on every fixture, depth and stages
              (north_star's bar; measured values are ~1e-6..1e-5, written to gpurun_out/parity.json)
  fp32=False  (fp16 operands, fp32 accumulate / residual streams; the reference's autocast path): per check, <= 2x the value
              measured in round 1 (table TOL16 below; the measurement is in gpurun_out/parity.json of each run)
Next to every mean there is a tail bound (99.9th percentile of |y - ref| <= 2e-2 * (|ref| + mean|ref|)) and a border-ring
relative L1 (<= 2x the check's tolerance), so a localised error - a wrong one-pixel ring, one bad tile - cannot hide in the mean.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL32 = 1e-3
# fp16-operand path: 2x the measured value (round 1's gpurun_out/parity.json at 36eef41; the checks that are new in round 2 -
# vitl.t32 6.3e-4, vitl.metric_video 4.4e-4, resize_video 2.2e-3, tiny_cls.depth 1.84e-3 - from their first run)
TOL16 = {"tiny.tap": 1.3e-3, "tiny.stage": 2.4e-3, "tiny.depth": 1.8e-3, "vits.nonsquare.depth": 1.4e-3, "vits.518.depth_sub": 3.0e-3,
         "vits.518.row_sums": 2.5e-3, "video.relative": 3.6e-3, "video.metric": 4.3e-3, "vitl.2x518": 2.4e-3, "vits.4x518": 7e-4,
         "vitl.t32": 1.3e-3, "vitl.metric_video": 9e-4, "resize_video": 4.4e-3, "tiny_cls.depth": 3.7e-3,
         # round 3: the BENCHMARKED workloads at their own size: measured 9.2e-4 (ViT-S) / 2.0e-3 (ViT-L), bound = 2x. The reference's
         # OWN autocast-fp16 path (the oracle's torch ops on the GPU under torch.autocast, VDA_TEST_YARDSTICK=1) is 1.86e-3 / 4.5e-3
         # from the fp32 reference on the same clips: both bounds stay below it
         "vits.32x518": 1.8e-3, "vitl.32x518": 4.0e-3, "vits.bn_rope.t32": 7.6e-3, "video.ragged": 8.4e-3,
         # ... and the outlier-activation state dicts (tests/_outliers.py): see test_outlier_activations
         # (2x the larger of the two LayerNorm forms, measured: channels 4.0e-3 / 5.1e-3, offset 2.6e-3 / 3.2e-3, both 3.3e-3 / 4.1e-3 for
         # fold / standalone; the reference's OWN autocast-fp16 path on the same streams: 9.8e-3, 4.3e-2, 8.0e-3)
         "outlier.channels": 1.0e-2, "outlier.offset": 6.4e-3, "outlier.both": 8.3e-3}
_measured = {}


def rel_l1(y, ref):
    y = np.asarray(y, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.abs(y - ref).mean() / max(np.abs(ref).mean(), 1e-12))


def record(name, val):
    _measured[name] = val
    out = os.path.join(REPO, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "parity.json"), "w") as f:
        json.dump(_measured, f, indent=1, sort_keys=True)


def check_map(name, y, ref, tol, tail=True):
    """Mean relative L1, plus (for [..., H, W] maps) the tail and border-ring bounds of the module docstring."""
    y, ref = np.asarray(y, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    e = rel_l1(y, ref)
    record(name, e)
    assert e < tol, f"{name}: rel-L1 {e:.3e} >= {tol:.1e}"
    if not tail:
        return e
    scale = np.abs(ref).mean()
    q = float(np.quantile(np.abs(y - ref) / (np.abs(ref) + scale), 0.999))
    record(name + ".p999", q)
    assert q < max(2e-2, 20 * tol), f"{name}: 99.9th percentile of the relative error is {q:.3e}"
    if y.ndim >= 2 and min(y.shape[-2:]) >= 8:
        ring = np.ones(y.shape[-2:], dtype=bool)
        ring[1:-1, 1:-1] = False
        er = float(np.abs(y - ref)[..., ring].mean() / max(np.abs(ref)[..., ring].mean(), 1e-12))
        record(name + ".ring", er)
        assert er < 2 * tol, f"{name}: border-ring rel-L1 {er:.3e}"
    return e


def model_for(name, seed, cls=None):
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict
    cfg = get_config(name)
    m = (cls or VideoDepthAnything)(encoder=name, features=cfg.features, out_channels=list(cfg.out_channels))
    sd = synthetic_state_dict(cfg, seed=seed)
    m.load_state_dict(sd, strict=True)
    return m.to("cuda").eval(), cfg, sd


def nhwc_to_nchw(t, B, h, w, Cp, C):
    return t.view(B, h, w, Cp)[..., :C].permute(0, 3, 1, 2).float().cpu().numpy()


PRECISIONS = [pytest.param(False, id="fp16"), pytest.param(True, id="fp32")]

_oracle_memo = {}


def oracle_once(key, fn):
    """The CPU oracle's result for a (model, input) pair, computed once for the fp16 and fp32 parametrisations of a test."""
    if key not in _oracle_memo:
        _oracle_memo[key] = fn()
    return _oracle_memo[key]


def tol_of(key, fp32):
    return TOL32 if fp32 else TOL16[key]


# ---------------------------------------------------------------- (a) reference-generated goldens
@pytest.mark.parametrize("fp32", PRECISIONS)
def test_golden_tiny_every_stage(golden_dir, fp32):
    z = np.load(os.path.join(golden_dir, "tiny_forward.npz"))
    m, cfg, _ = model_for("tiny", int(z["sd_seed"]))
    x = torch.from_numpy(z["x"]).cuda()
    d = m.forward(x, fp32=fp32)
    BT = x.shape[0] * x.shape[1]
    tag = "tiny.f32." if fp32 else "tiny."
    for i in range(4):
        t, h, w, Cp = m.engine.stage(f"tap{i}")
        check_map(f"{tag}tap{i}", t.float().cpu().numpy().reshape(z[f"tap{i}"].shape), z[f"tap{i}"], tol_of("tiny.tap", fp32), tail=False)
    chans = {"layer_1": cfg.out_channels[0], "layer_2": cfg.out_channels[1], "layer_3": cfg.out_channels[2],
             "layer_4": cfg.out_channels[3], "path_4": cfg.features, "path_3": cfg.features, "path_2": cfg.features,
             "path_1": cfg.features}
    for k, C in chans.items():
        t, h, w, Cp = m.engine.stage(k)
        check_map(f"{tag}{k}", nhwc_to_nchw(t, BT, h, w, Cp, C), z[k], tol_of("tiny.stage", fp32), tail=False)
    check_map(f"{tag}depth", d.cpu().numpy(), z["depth"], tol_of("tiny.depth", fp32))


@pytest.mark.parametrize("fp32", PRECISIONS)
def test_golden_tiny_with_clstoken_readout(golden_dir, fp32):
    """use_clstoken=True (dpt_temporal.py:56-59; no released config sets it): reference-generated golden, both precisions, and
    the two launch sequences bit-identical on it."""
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict
    z = np.load(os.path.join(golden_dir, "tiny_clstoken_forward.npz"))
    cfg = get_config("tiny", use_clstoken=True)
    m = VideoDepthAnything(encoder="tiny", features=cfg.features, out_channels=list(cfg.out_channels), use_clstoken=True)
    assert m.cfg.use_clstoken
    m.load_state_dict(synthetic_state_dict(cfg, seed=int(z["sd_seed"])), strict=True)
    m = m.to("cuda").eval()
    x = torch.from_numpy(z["x"]).cuda()
    d = m.forward(x, fp32=fp32)
    BT = x.shape[0] * x.shape[1]
    tag = "tiny_cls.f32." if fp32 else "tiny_cls."
    for k, C in (("layer_1", cfg.out_channels[0]), ("layer_2", cfg.out_channels[1])):
        t, h, w, Cp = m.engine.stage(k)
        check_map(tag + k, nhwc_to_nchw(t, BT, h, w, Cp, C), z[k], tol_of("tiny.stage", fp32), tail=False)
    check_map(tag + "depth", d.cpu().numpy(), z["depth"], tol_of("tiny_cls.depth", fp32))
    m.engine.set_option("ln_fold", 0)                    # the Python launch sequence keeps the standalone LayerNorms
    assert torch.equal(m.forward(x, fp32=fp32), m.python_engine().forward(x, fp32=fp32))
    m.engine.set_option("ln_fold", 1)
    with pytest.raises(RuntimeError, match="Missing key"):          # the readout weights are part of the strict inventory
        m.load_state_dict(synthetic_state_dict(get_config("tiny"), seed=1), strict=True)


@pytest.mark.parametrize("fp32", PRECISIONS)
def test_golden_vits_nonsquare(golden_dir, fp32):
    z = np.load(os.path.join(golden_dir, "vits_forward.npz"))
    m, _, _ = model_for("vits", int(z["sd_seed"]))
    d = m.forward(torch.from_numpy(z["x"]).cuda(), fp32=fp32)
    check_map("vits.nonsquare.depth" + (".f32" if fp32 else ""), d.cpu().numpy(), z["depth"], tol_of("vits.nonsquare.depth", fp32))


@pytest.mark.parametrize("fp32", PRECISIONS)
def test_golden_vits_518(golden_dir, fp32):
    z = np.load(os.path.join(golden_dir, "vits_518.npz"))
    m, _, _ = model_for("vits", int(z["sd_seed"]))
    x = torch.randn(1, 1, 3, 518, 518, generator=torch.Generator().manual_seed(int(z["x_seed"])))
    d = m.forward(x.cuda(), fp32=fp32).cpu().numpy()
    sfx = ".f32" if fp32 else ""
    check_map("vits.518.depth_sub" + sfx, d[..., ::7, ::7], z["depth_sub"], tol_of("vits.518.depth_sub", fp32))
    check_map("vits.518.row_sums" + sfx, d.sum(axis=-1), z["row_sums"], tol_of("vits.518.row_sums", fp32), tail=False)


@pytest.mark.parametrize("fp32", PRECISIONS)
@pytest.mark.parametrize("name,metric", [("tiny_video.npz", False), ("tiny_metric_video.npz", True)])
def test_golden_infer_video_depth(golden_dir, name, metric, fp32):
    from video_depth_anything_amd.video_depth import MetricVideoDepthAnything, VideoDepthAnything
    z = np.load(os.path.join(golden_dir, name))
    m, _, _ = model_for("tiny", int(z["sd_seed"]), MetricVideoDepthAnything if metric else VideoDepthAnything)
    depths, fps = m.infer_video_depth(z["frames"], 24, input_size=int(z["input_size"]), device="cuda", fp32=fp32)
    assert depths.shape == z["depths"].shape and depths.dtype == np.float32 and fps == 24
    key = "video.metric" if metric else "video.relative"
    check_map(key + (".f32" if fp32 else ""), depths, z["depths"], tol_of(key, fp32))     # the scale/shift fit compounds per-window error


# ---------------------------------------------------------------- (b) the CPU oracle on seeded inputs
@pytest.mark.parametrize("fp32", PRECISIONS)
def test_oracle_vits_4frames_518(fp32):
    """ViT-S (BASELINE config 2's model), 4 frames at 518x518 (1370 tokens/frame, stored pos-embed)."""
    from oracle import vda_oracle as O
    m, cfg, sd = model_for("vits", 7)
    x = torch.randn(1, 4, 3, 518, 518, generator=torch.Generator().manual_seed(70))
    with torch.no_grad():
        ref = oracle_once("vits.4x518", lambda: O.forward(sd, cfg, x).numpy())
    d = m.forward(x.cuda(), fp32=fp32).cpu().numpy()
    check_map("vits.4x518.depth_vs_oracle" + (".f32" if fp32 else ""), d, ref, tol_of("vits.4x518", fp32))


@pytest.mark.parametrize("fp32", PRECISIONS)
def test_oracle_vitl_2frames_518(fp32):
    """The headline model (config 3): ViT-L, 2 frames at 518x518."""
    from oracle import vda_oracle as O
    m, cfg, sd = model_for("vitl", 3)
    x = torch.randn(1, 2, 3, 518, 518, generator=torch.Generator().manual_seed(72))
    with torch.no_grad():
        ref = oracle_once("vitl.2x518", lambda: O.forward(sd, cfg, x).numpy())
    d = m.forward(x.cuda(), fp32=fp32).cpu().numpy()
    check_map("vitl.2x518.depth_vs_oracle" + (".f32" if fp32 else ""), d, ref, tol_of("vitl.2x518", fp32))


@pytest.mark.parametrize("fp32", PRECISIONS)
def test_oracle_vitl_32frames_small(fp32):
    """Config 3's temporal extent: ViT-L with the full 32-frame temporal attention (d = 128 / 32 heads of the four motion
    modules, PE rows 0..31) at a small spatial size, 1x32x3x70x84, against the oracle."""
    from oracle import vda_oracle as O
    m, cfg, sd = model_for("vitl", 4)
    x = torch.randn(1, 32, 3, 70, 84, generator=torch.Generator().manual_seed(73))
    with torch.no_grad():
        ref = oracle_once("vitl.32x70x84", lambda: O.forward(sd, cfg, x).numpy())
    d = m.forward(x.cuda(), fp32=fp32).cpu().numpy()
    check_map("vitl.32x70x84.depth_vs_oracle" + (".f32" if fp32 else ""), d, ref, tol_of("vitl.t32", fp32))


@pytest.mark.parametrize("fp32", PRECISIONS)
def test_oracle_metric_vitl_two_windows(fp32):
    """Config 5: MetricVideoDepthAnything with its ViT-L defaults (metric_depth/video_depth_anything/video_depth.py:36-45),
    a 40-frame video = 2 windows at a small spatial size, against the oracle's infer_video_depth(metric=True) (:132)."""
    from oracle import vda_oracle as O
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import MetricVideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict
    m = MetricVideoDepthAnything()                       # vitl, features 256, out_channels [256, 512, 1024, 1024]
    cfg = get_config("vitl")
    assert m.cfg == cfg and m.METRIC
    sd = synthetic_state_dict(cfg, seed=9)
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    frames = np.random.default_rng(31).integers(0, 256, (40, 42, 56, 3), dtype=np.uint8)
    ref = oracle_once("vitl.metric_video", lambda: O.infer_video_depth(sd, cfg, frames, 24, input_size=42, metric=True)[0])
    d, fps = m.infer_video_depth(frames, 24, input_size=42, device="cuda", fp32=fp32)
    assert d.shape == ref.shape == (40, 42, 56) and fps == 24
    check_map("vitl.metric_video" + (".f32" if fp32 else ""), d, ref, tol_of("vitl.metric_video", fp32))


# ---- the BENCHMARKED workloads at their own size (VERDICT r2 missing #1): bench.py's x = randn(1,32,3,518,518) (seed 0) and
# synthetic_state_dict(seed 0), ViT-S (BASELINE config 2) and ViT-L (config 3, the headline), fp16 AND fp32 operand paths, against
# ONE oracle run per model (ViT-S ~15 s, ViT-L ~70 s on the box's 16 cores). Covers what no smaller case reaches: the temporal
# attention at hw = 1369 / 361 / 1369 / 5476 with T = 32, the encoder attention's 192 / 512 (frame, head) problems, every GEMM's
# partial last round and partial last row tile, 32-bit offsets at the full tensor sizes.
_full_clip_ref = {}


def full_clip_reference(name):
    if name not in _full_clip_ref:
        from oracle import vda_oracle as O
        from video_depth_anything_amd.config import get_config
        from video_depth_anything_amd.weights import synthetic_state_dict
        cfg = get_config(name)
        sd = synthetic_state_dict(cfg, seed=0)
        x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0))
        with torch.no_grad():
            _full_clip_ref[name] = (x, O.forward(sd, cfg, x).numpy())
    return _full_clip_ref[name]


@pytest.mark.parametrize("fp32", PRECISIONS)
@pytest.mark.parametrize("name", ["vits", "vitl"])
def test_oracle_full_clip_32x518(name, fp32):
    x, ref = full_clip_reference(name)
    m, _, _ = model_for(name, 0)
    d = m.forward(x.cuda(), fp32=fp32).cpu().numpy()
    assert d.shape == ref.shape == (1, 32, 518, 518)
    check_map(f"{name}.32x518.depth_vs_oracle" + (".f32" if fp32 else ""), d, ref, tol_of(f"{name}.32x518", fp32))
    # per frame too: one bad frame (a wrong tail round, a temporal-attention wave past the grid) cannot hide in the clip's mean
    worst = max(rel_l1(d[0, t], ref[0, t]) for t in range(32))
    record(f"{name}.32x518.worst_frame" + (".f32" if fp32 else ""), worst)
    assert worst < 2 * tol_of(f"{name}.32x518", fp32)
    if not fp32:
        # yardstick (recorded, see autocast_oracle_error): the reference's own autocast-fp16 path against its fp32 result, same clip
        from video_depth_anything_amd.config import get_config
        from video_depth_anything_amd.weights import synthetic_state_dict
        cfg = get_config(name)
        e_ref16 = autocast_oracle_error(synthetic_state_dict(cfg, seed=0), cfg, x, ref)
        record(f"{name}.32x518.reference_autocast_fp16", e_ref16)
        if e_ref16 is not None:
            assert rel_l1(d, ref) < 1.5 * e_ref16 + 5e-4, "the fp16 path is further from the fp32 reference than the reference's own autocast path"


@pytest.mark.parametrize("kind", ["channels", "offset", "both"])
def test_outlier_activations(kind):
    """Residual streams shaped like trained DINOv2's (VERDICT r2 missing #2): a few channels hundreds of times the
    typical magnitude ("channels"), token means tens of standard deviations from zero ("offset"), and both - produced by a
    seeded state dict with a handful of biases moved (tests/_outliers.py), ViT-S at the vits_forward fixture shape, against the
    fp32 oracle on the same weights. The fp32 path must hold north_star's 1e-3; the fp16 path is run with the LayerNorm fold on
    (default) and off and must hold its stated tolerance either way - and the fold may not be more than 2x further from the
    oracle than the standalone LayerNorm form."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from oracle import vda_oracle as O
    from _outliers import outlier_state_dict, stream_report
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    cfg = get_config("vits")
    sd = outlier_state_dict(cfg, seed=21, kind=kind)
    x = torch.randn(1, 3, 3, 56, 70, generator=torch.Generator().manual_seed(102))
    with torch.no_grad():
        ref = O.forward(sd, cfg, x).numpy()
        rep = stream_report(sd, cfg, x)                  # the oracle's own residual stream: what the state dict really produces
    for k, v in rep.items():
        record(f"outlier.{kind}.stream.{k}", v)
    if kind in ("channels", "both"):
        assert rep["max_abs"] > 300, rep
    if kind == "offset":                                 # (with the outlier channels in, they dominate sigma: "both" has |x| ~ 1e3 on top of the shifted stream)
        assert rep["mean_over_sigma_p50"] > 10, rep
    m = VideoDepthAnything(encoder="vits", features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    d32 = m.forward(x.cuda(), fp32=True).cpu().numpy()
    on = m.forward(x.cuda(), fp32=False).cpu().numpy()
    m.engine.set_option("ln_fold", 0)
    try:
        off = m.forward(x.cuda(), fp32=False).cpu().numpy()
    finally:
        m.engine.set_option("ln_fold", 1)
    e_on, e_off = rel_l1(on, ref), rel_l1(off, ref)
    record(f"outlier.{kind}.ln_fold", e_on)
    record(f"outlier.{kind}.standalone_ln", e_off)
    # Yardstick: the REFERENCE's own fp16 path on this stream = the oracle's torch ops on the GPU under torch.autocast (ATen /
    # rocBLAS / MIOpen: checker only, the product never calls them). How far autocast moves the reference from its fp32 result
    # is what "fp16 tolerance" can honestly mean here; recorded, and the engine's fp16 path may not be more than 1.5x further.
    e_ref16 = autocast_oracle_error(sd, cfg, x, ref)
    record(f"outlier.{kind}.reference_autocast_fp16", e_ref16)
    check_map(f"outlier.{kind}.f32", d32, ref, TOL32)
    assert e_on < 2 * e_off + 2e-4, f"LayerNorm fold {e_on:.3e} vs standalone {e_off:.3e} on the '{kind}' stream"
    if e_ref16 is not None:
        assert e_on < 1.5 * e_ref16 + 5e-4, f"fp16 path {e_on:.3e} vs the reference's own autocast path {e_ref16:.3e} ('{kind}')"
    check_map(f"outlier.{kind}.ln_fold", on, ref, TOL16[f"outlier.{kind}"])
    check_map(f"outlier.{kind}.standalone_ln", off, ref, TOL16[f"outlier.{kind}"])


def autocast_oracle_error(sd, cfg, x, ref):
    """rel-L1 of the oracle run on the GPU under torch.autocast(fp16) - the reference's fp16 path, video_depth.py:203-205 -
    against its fp32 CPU result `ref`. None if torch's GPU kernels cannot run the shape here, or when VDA_TEST_YARDSTICK is not
    set: MIOpen builds its convolution kernels at first use on a fresh box (minutes), so the yardstick is an opt-in run whose
    numbers are recorded in DESIGN.md section 2; the default GPU suite does not pay for it."""
    from oracle import vda_oracle as O
    if os.environ.get("VDA_TEST_YARDSTICK") != "1":
        return None
    try:
        sdc = {k: v.cuda() for k, v in sd.items()}
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            y = O.forward(sdc, cfg, x.cuda())
        return rel_l1(y.float().cpu().numpy(), ref)
    except Exception as e:                                   # noqa: BLE001 - a yardstick, not a requirement
        print(f"autocast oracle unavailable: {type(e).__name__}: {e}")
        return None


def test_handle_and_python_orchestration_are_bit_identical(golden_dir):
    """vda_forward (csrc/host.hip: C++ weight packing + launch sequence) against engine.Engine (Python packing + the same
    launches through the per-kernel ABI): bit-identical outputs on the tiny and ViT-S fixtures, both precisions."""
    # (the Python orchestration keeps the standalone LayerNorms: the handle runs with its "ln_fold" option off here)
    for name, fixture in (("tiny", "tiny_forward.npz"), ("vits", "vits_forward.npz")):
        z = np.load(os.path.join(golden_dir, fixture))
        m, _, _ = model_for(name, int(z["sd_seed"]))
        py = m.python_engine()
        x = torch.from_numpy(z["x"]).cuda()
        m.engine.set_option("ln_fold", 0)
        try:
            for fp32 in (False, True):
                a = m.forward(x, fp32=fp32).clone()
                b = py.forward(x, fp32=fp32).clone()
                assert torch.equal(a, b), f"{name} fp32={fp32}: {int((a != b).sum())} of {a.numel()} elements differ"
        finally:
            m.engine.set_option("ln_fold", 1)
    # a square 518 frame too (stored pos-embed, 1370 tokens, the 256-row GEMM kernels)
    m, _, _ = model_for("vits", 11)
    py = m.python_engine()
    x = torch.randn(1, 2, 3, 518, 518, generator=torch.Generator().manual_seed(74)).cuda()
    m.engine.set_option("ln_fold", 0)
    try:
        assert torch.equal(m.forward(x, fp32=False), py.forward(x, fp32=False))
    finally:
        m.engine.set_option("ln_fold", 1)


def test_layernorm_fold_matches_the_standalone_layernorm(golden_dir):
    """vda_set_option "ln_fold" (default on, fp16 path): LayerNorm folded into the encoder GEMMs either side of it (split fp16
    residual stream, statistics from the residual epilogue, affine folded into qkv / fc1) against the standalone LayerNorm
    launches: the two differ by operand rounding only (the stream is rounded to fp16 before instead of after the normalisation),
    each is equally far from the fp32 oracle's golden output, and the option does not touch the fp32 path."""
    z = np.load(os.path.join(golden_dir, "vits_forward.npz"))
    m, _, _ = model_for("vits", int(z["sd_seed"]))
    x = torch.from_numpy(z["x"]).cuda()
    ref = z["depth"]
    a = m.forward(x, fp32=False).clone()
    f32 = m.forward(x, fp32=True).clone()
    m.engine.set_option("ln_fold", 0)
    try:
        b = m.forward(x, fp32=False).clone()
        assert torch.equal(f32, m.forward(x, fp32=True))
    finally:
        m.engine.set_option("ln_fold", 1)
    assert torch.equal(a, m.forward(x, fp32=False))
    e = rel_l1(a.cpu().numpy(), b.cpu().numpy())
    ea, eb = rel_l1(a.cpu().numpy(), ref), rel_l1(b.cpu().numpy(), ref)
    record("vits.ln_fold_vs_standalone", e)
    record("vits.ln_fold.depth_vs_golden", ea)
    record("vits.standalone_ln.depth_vs_golden", eb)
    assert 0 < e < 1.4e-3 and ea < 1.4e-3 and eb < 1.4e-3
    # a square 518 frame: 1370 tokens per frame, the 256-row kernels' row-layout epilogues (statistics from the epilogue itself)
    m, _, _ = model_for("vits", 11)
    x = torch.randn(1, 2, 3, 518, 518, generator=torch.Generator().manual_seed(74)).cuda()
    a = m.forward(x, fp32=False).clone()
    m.engine.set_option("ln_fold", 0)
    try:
        b = m.forward(x, fp32=False).clone()
    finally:
        m.engine.set_option("ln_fold", 1)
    e = rel_l1(a.cpu().numpy(), b.cpu().numpy())
    record("vits.518.ln_fold_vs_standalone", e)
    assert 0 < e < 1.4e-3


def test_residual_in_layernorm_matches_the_epilogue_residual():
    """The two placements of the encoder's residual add (vda_set_option "residual_in_ln": the GEMM's fp32 in-place epilogue - the
    default - or fused into the next LayerNorm with the projection output stored as fp16) differ only by that fp16 rounding."""
    m, cfg, sd = model_for("vits", 0)
    x = torch.randn(1, 3, 3, 56, 70, generator=torch.Generator().manual_seed(102)).cuda()      # the vits_forward fixture's input
    m.engine.set_option("ln_fold", 0)                     # both placements keep the standalone LayerNorm
    try:
        b = m.forward(x, fp32=False).clone()
        m.engine.set_option("residual_in_ln", 1)
        a = m.forward(x, fp32=False).clone()
        m.engine.set_option("residual_in_ln", 0)
        assert torch.equal(b, m.forward(x, fp32=False))
    finally:
        m.engine.set_option("ln_fold", 1)
    e = rel_l1(a.cpu().numpy(), b.cpu().numpy())
    record("vits.residual_in_ln_vs_epilogue", e)
    assert 0 < e < 1.4e-3           # measured 6.5e-4; each form is 6.4e-4 / 6.7e-4 from the fp32 oracle on this input (tests/diag_res_ab.py)
    py = m.python_engine()
    py.residual_in_ln = True
    assert torch.equal(a, py.forward(x, fp32=False)), "both orchestrations, residual-in-LayerNorm form"


def test_dynamic_tile_schedule_does_not_change_the_forward():
    """vda_set_option "dyn_sched" (what multi-rank runs turn on: the 8-phase GEMMs draw their tiles from per-launch counters zeroed
    at the start of the forward) moves tiles between workgroups, never a bit of the result."""
    m, _, _ = model_for("vits", 11)
    x = torch.randn(1, 2, 3, 518, 518, generator=torch.Generator().manual_seed(74)).cuda()
    a = m.forward(x, fp32=False).clone()
    m.engine.set_option("dyn_sched", 1)
    try:
        b = m.forward(x, fp32=False).clone()
        g = torch.cuda.CUDAGraph()                        # the per-forward counter reset is a memset node: still capturable
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())     # (one workspace slot: the forwards must not overlap)
        with torch.cuda.stream(side):
            m.forward(x, fp32=False)
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            c = m.forward(x, fp32=False)
        g.replay()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(a, c), f"graph replay with dynamic draws: {int((a != c).sum())} elements differ"
    finally:
        m.engine.set_option("dyn_sched", 0)
    assert torch.equal(a, b), f"eager forward with dynamic draws: {int((a != b).sum())} elements differ"
    d = m.forward(x, fp32=False)
    assert torch.equal(a, d), f"static schedule again: {int((a != d).sum())} elements differ"


def test_steady_state_forward_is_graph_capturable_and_allocation_free():
    """include/vda.h: after the first forward of a (shape, precision) vda_forward only enqueues kernels on the stream it is
    given - so it can be captured into a HIP graph, and the replay reproduces the eager result bit for bit."""
    m, _, _ = model_for("vits", 15)
    x = torch.randn(1, 4, 3, 70, 84, generator=torch.Generator().manual_seed(76)).cuda()
    ref = m.forward(x, fp32=False).clone()               # first use: layout, pos-embed, workspace
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())        # (one workspace slot: the forwards must not overlap)
    with torch.cuda.stream(side):
        m.forward(x, fp32=False)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = m.forward(x, fp32=False)
    x.copy_(torch.randn(1, 4, 3, 70, 84, generator=torch.Generator().manual_seed(77)))     # new input, same buffers
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, m.forward(x, fp32=False)) and not torch.equal(out, ref)


def test_c_abi_refuses_bad_calls_with_the_reference_wording():
    import ctypes as C
    from video_depth_anything_amd._lib import lib
    m, cfg, _ = model_for("tiny", 2)
    h = m.engine._h
    x = torch.zeros(1, 2, 3, 30, 28).cuda()
    out = torch.zeros(1, 2, 30, 28).cuda()
    assert lib.vda_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), 1, 2, 30, 28, 0, None) != 0
    assert b"Input image height 30 is not a multiple of patch height 14" in lib.vda_last_error()       # patch_embed.py:73
    assert lib.vda_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), 1, 40, 28, 28, 0, None) != 0
    assert b"temporal_max_len" in lib.vda_last_error()
    assert lib.vda_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), 1, 2, 28, 28, 7, None) != 0
    assert b"precision" in lib.vda_last_error()
    assert lib.vda_workspace_bytes(h, 1, 2, 28, 28, 0) > 0
    assert lib.vda_set_workspace(h, C.c_void_p(x.data_ptr() + 8), 1 << 20) != 0 and b"256-byte aligned" in lib.vda_last_error()
    small = torch.empty(4096, dtype=torch.uint8, device="cuda")
    assert lib.vda_set_workspace(h, C.c_void_p(small.data_ptr()), small.numel()) == 0
    x = torch.zeros(1, 2, 3, 28, 28).cuda()
    assert lib.vda_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), 1, 2, 28, 28, 0, None) != 0
    assert b"vda_workspace_bytes" in lib.vda_last_error()                   # too small a block is refused, nothing is launched
    assert lib.vda_set_workspace(h, None, 0) == 0                           # back to a handle-owned block
    assert lib.vda_prepare(h, 1, 2, 28, 28, 1) == 0                         # front-loads the fp32 weight pack
    assert lib.vda_forward(h, C.c_void_p(x.data_ptr()), C.c_void_p(out.data_ptr()), 1, 2, 28, 28, 1, None) == 0
    torch.cuda.synchronize()
    assert torch.isfinite(out).all() and float(out.abs().sum()) > 0
    with pytest.raises(AssertionError, match="multiple of patch height"):
        m.forward(torch.zeros(1, 2, 3, 30, 28).cuda(), fp32=False)


def test_forward_precision_follows_autocast():
    """A bare model(x) is the reference's nn.Module call: fp32 outside torch.autocast, fp16 operands inside."""
    m, _, _ = model_for("tiny", 1)
    x = torch.randn(1, 3, 3, 42, 56, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.autocast("cuda"):
        a = m(x).clone()
    assert torch.equal(a, m.forward(x, fp32=False))
    assert torch.equal(m(x), m.forward(x, fp32=True))
    assert not torch.equal(a, m(x))


def test_handle_refuses_bad_state_dicts_like_torch():
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict
    cfg = get_config("tiny")
    sd = synthetic_state_dict(cfg, seed=0)
    m = VideoDepthAnything(encoder="tiny", features=cfg.features, out_channels=list(cfg.out_channels)).to("cuda")
    bad = dict(sd)
    del bad["head.scratch.output_conv1.bias"]
    with pytest.raises(RuntimeError, match="Missing key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd, extra=torch.zeros(1))
    with pytest.raises(RuntimeError, match="Unexpected key"):
        m.load_state_dict(bad, strict=True)
    bad = dict(sd)
    bad["pretrained.norm.weight"] = torch.zeros(7)
    with pytest.raises(RuntimeError, match="size mismatch"):
        m.load_state_dict(bad, strict=True)
    # the C side says the same when driven directly (a C host has no Python checker in front of it)
    import ctypes as C
    from video_depth_anything_amd._lib import lib
    t = torch.zeros(7)
    dims = (C.c_int64 * 1)(7)
    assert lib.vda_load_weight(m.engine._h, b"pretrained.norm.weight", C.c_void_p(t.data_ptr()), dims, 1, 0) != 0
    assert b"size mismatch for pretrained.norm.weight" in lib.vda_last_error()
    assert lib.vda_load_weight(m.engine._h, b"nope", C.c_void_p(t.data_ptr()), dims, 1, 0) != 0
    assert b"Unexpected key" in lib.vda_last_error()
    m.load_state_dict(sd, strict=True)
    assert torch.isfinite(m.forward(torch.zeros(1, 2, 3, 28, 28).cuda(), fp32=False)).all()


# ---------------------------------------------------------------- preprocessing resize (frames not at network size)
@pytest.mark.parametrize("fp32", PRECISIONS)
def test_video_with_non_network_size_frames(fp32):
    """Real-video case: 36x64 source frames, input_size 28 -> network 28x56 (aspect guard + lower-bound rule), so every window
    goes through vda_gather_resize_normalize_u8_f32. Reference path restated: bicubic (cv2.INTER_CUBIC's definition, evaluated by
    torch on the CPU - cv2 itself is absent, so parity with cv2's own arithmetic is unpinned) -> oracle forward per window ->
    bilinear back to the source size -> the oracle's stitcher."""
    import torch.nn.functional as F
    from oracle import vda_oracle as O
    from video_depth_anything_amd import scheduler as S
    m, cfg, sd = model_for("tiny", 12)
    frames = np.random.default_rng(32).integers(0, 256, (40, 36, 64, 3), dtype=np.uint8)
    H, W = S.network_size(36, 64, 28)
    assert (H, W) != (36, 64)
    d, _ = m.infer_video_depth(frames, 24, input_size=28, device="cuda", fp32=fp32)
    mean, std = torch.tensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1), torch.tensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1)
    wins = []
    for idx in S.plan_windows(40):
        img = torch.from_numpy(frames[idx]).float().div(255.0).permute(0, 3, 1, 2)
        x = (F.interpolate(img, size=(H, W), mode="bicubic", align_corners=False) - mean) / std
        with torch.no_grad():
            y = O.forward(sd, cfg, x[None])
            wins.append(F.interpolate(y.transpose(0, 1), size=(36, 64), mode="bilinear", align_corners=True)[:, 0].numpy())
    ref = S.stitch_windows(wins, 40)
    assert d.shape == ref.shape == (40, 36, 64)
    check_map("resize_video" + (".f32" if fp32 else ""), d, ref, tol_of("resize_video", fp32))


# ---------------------------------------------------------------- CLI / driver callers
def test_run_cli_synthetic(tmp_path):
    """run.py end to end with the reference's flags: frames from .npz, depths to <name>_depths.npz; --fp32 is the fp32 path."""
    import subprocess
    import sys
    frames = np.random.default_rng(9).integers(0, 256, (30, 70, 84, 3), dtype=np.uint8)
    src = tmp_path / "clip.npz"
    np.savez(src, frames=frames, fps=24)
    outs = {}
    for flag in ([], ["--fp32"]):
        out = tmp_path / ("out32" if flag else "out16")
        r = subprocess.run([sys.executable, os.path.join(REPO, "run.py"), "--input_video", str(src), "--output_dir", str(out),
                            "--encoder", "vits", "--input_size", "70", "--checkpoint", "synthetic", "--save_npz"] + flag,
                           capture_output=True, text=True, timeout=600, cwd=REPO)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        d = np.load(out / "clip_depths.npz")["depths"]
        assert d.shape == (30, 70, 84) and d.dtype == np.float32 and np.isfinite(d).all() and d.min() >= 0
        outs[bool(flag)] = d
    e = rel_l1(outs[False], outs[True])
    record("run_cli.fp16_vs_fp32", e)
    assert 0 < e < 5e-3, "--fp32 must change the arithmetic (and only slightly)"


def test_run_cli_config1_vits_fp32_full_clip(tmp_path):
    """BASELINE.json configs[0] as far as this product can run it: ViT-S, one 32x518x518 clip, fp32, through run.py. The
    reference runs that config on the CPU; this engine has no CPU path, so the same CLI call runs the fp32-operand kernels on
    the MI355X (2 windows) and is checked against the oracle's infer_video_depth on the same frames and weights."""
    import subprocess
    import sys
    from oracle import vda_oracle as O
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.weights import synthetic_state_dict
    frames = np.random.default_rng(11).integers(0, 256, (32, 518, 518, 3), dtype=np.uint8)
    src = tmp_path / "clip.npz"
    np.savez(src, frames=frames, fps=24)
    r = subprocess.run([sys.executable, os.path.join(REPO, "run.py"), "--input_video", str(src), "--output_dir", str(tmp_path / "out"),
                        "--encoder", "vits", "--fp32", "--checkpoint", "synthetic", "--save_npz"],
                       capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = np.load(tmp_path / "out" / "clip_depths.npz")["depths"]
    assert d.shape == (32, 518, 518) and d.dtype == np.float32
    # window 0 of a 32-frame video is frames 0..31 and is stitched unscaled (video_depth.py:219-224): the first 22 output
    # frames must equal the oracle's fp32 forward of those 32 frames (about 30 s of CPU time on the box)
    cfg = get_config("vits")
    sd = synthetic_state_dict(cfg, seed=0)
    x = torch.from_numpy(np.stack([O.preprocess_frame(f, 518) for f in frames]))[None]
    with torch.no_grad():
        ref = O.forward(sd, cfg, x)[0, :22].numpy()
    check_map("run_cli.config1.vits_fp32_518", d[:22], ref, TOL32)



def test_benchmark_infer_driver(tmp_path):
    """benchmark/infer/infer.py (the reference's second caller, always fp32=True): its per-frame .npy equals
    infer_video_depth(fp32=True) called directly on the same frames with the channel order PRESERVED - the reference hands
    cv2.imread's BGR arrays straight to the model (benchmark/infer/infer.py:54-58), and so does the driver here."""
    import subprocess
    import sys
    rng = np.random.default_rng(10)
    scene, imgs = [], []
    os.makedirs(tmp_path / "data" / "scene0", exist_ok=True)
    for i in range(5):
        bgr = rng.integers(0, 256, (70, 84, 3), dtype=np.uint8)
        np.save(tmp_path / "data" / "scene0" / f"{i:03d}.npy", bgr)
        imgs.append(bgr)
        scene.append({"image": f"data/scene0/{i:03d}.npy"})
    man = tmp_path / "manifest.json"
    man.write_text(json.dumps({"toy": [{"scene0": scene}]}))
    r = subprocess.run([sys.executable, os.path.join(REPO, "benchmark", "infer", "infer.py"), "--json_file", str(man), "--infer_path",
                        str(tmp_path / "pred"), "--datasets", "toy", "--encoder", "vits", "--input_size", "70", "--checkpoint", "synthetic"],
                       capture_output=True, text=True, timeout=600, cwd=REPO)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    got = np.stack([np.load(tmp_path / "pred" / "toy" / "data" / "scene0" / f"{i:03d}.npy") for i in range(5)])
    assert got.shape == (5, 70, 84) and got.dtype == np.float32
    m, _, _ = model_for("vits", 0)                       # --checkpoint synthetic = synthetic_state_dict(cfg, seed=0)
    direct, _ = m.infer_video_depth(np.stack(imgs), 1, input_size=70, device="cuda", fp32=True)
    assert np.array_equal(got, direct), "the driver must hand the frames as read (BGR) to infer_video_depth(fp32=True) and save its output unchanged"
    swapped, _ = m.infer_video_depth(np.stack([im[:, :, ::-1] for im in imgs]), 1, input_size=70, device="cuda", fp32=True)
    assert not np.array_equal(got, swapped), "channel order must matter for this check to mean anything"
    # ... and value-correct: the oracle's infer_video_depth (benchmark/infer/infer.py:54-58 hands the BGR frames to
    # infer_video_depth(..., fp32=True)) on the same frames and weights, at the fp32 path's bar
    from oracle import vda_oracle as O
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.weights import synthetic_state_dict
    cfg = get_config("vits")
    ref, _ = O.infer_video_depth(synthetic_state_dict(cfg, seed=0), cfg, np.stack(imgs), 1, input_size=70)
    check_map("benchmark_infer.vs_oracle", got, ref, TOL32)


# ---------------------------------------------------------------- (c) full sizes through properties
def test_batch_of_clips_equals_separate_clips():
    m, _, _ = model_for("vits", 8)
    x = torch.randn(2, 5, 3, 70, 84, generator=torch.Generator().manual_seed(71)).cuda()
    both = m.forward(x, fp32=False).clone()
    a = m.forward(x[:1].contiguous(), fp32=False).clone()
    b = m.forward(x[1:].contiguous(), fp32=False).clone()
    assert torch.equal(both[0], a[0]) and torch.equal(both[1], b[0])


@pytest.mark.parametrize("name", ["vits", "vitl"])
def test_full_size_properties(name):
    """BASELINE.json configs 2 and 3: ViT-S / ViT-L, 1x32x518x518, fp16 operands."""
    m, _, _ = model_for(name, 0)
    x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0)).cuda()
    d1 = m.forward(x, fp32=False).clone()
    d2 = m.forward(x, fp32=False).clone()
    assert d1.shape == (1, 32, 518, 518) and d1.dtype == torch.float32
    assert torch.isfinite(d1).all() and float(d1.min()) >= 0.0
    assert float(d1.std()) > 0, "degenerate output"
    assert torch.equal(d1, d2), "forward must be bitwise deterministic (no atomics in any reduction)"
    # clip independence at full width: a batch of the same 8-frame clip twice gives identical halves
    d3 = m.forward(torch.cat([x[:, :8], x[:, :8]], dim=0).contiguous(), fp32=False)
    assert torch.equal(d3[0], d3[1])


def test_full_size_vitl_fp32_against_fp16():
    """The fp32 path at the headline shape (8 of the 32 frames to bound the run time): finite, deterministic, and within the
    fp16 path's tolerance of it - the two paths share nothing but the launch sequence."""
    m, _, _ = model_for("vitl", 0)
    x = torch.randn(1, 8, 3, 518, 518, generator=torch.Generator().manual_seed(1)).cuda()
    a = m.forward(x, fp32=True).clone()
    b = m.forward(x, fp32=True).clone()
    assert torch.isfinite(a).all() and float(a.min()) >= 0.0 and torch.equal(a, b)
    e = rel_l1(m.forward(x, fp32=False).cpu().numpy(), a.cpu().numpy())
    record("vitl.8x518.fp16_vs_fp32", e)
    assert e < TOL16["vitl.2x518"]


def test_long_video_equals_per_window_forward_plus_host_stitcher():
    """A 230-frame 518x518 video (11 windows) through infer_video_depth (device gather, forward, device stitcher, streamed D2H)
    == forward per planned window + the numpy stitcher (scheduler.stitch_windows, bit-equal to the oracle's on the CPU)."""
    from video_depth_anything_amd import ops, scheduler as S
    m, _, _ = model_for("vits", 13)
    n = 230
    frames = np.random.default_rng(33).integers(0, 256, (n, 518, 518, 3), dtype=np.uint8)
    d, _ = m.infer_video_depth(frames, 24, input_size=518, device="cuda", fp32=False)
    assert d.shape == (n, 518, 518) and d.dtype == np.float32 and np.isfinite(d).all() and d.min() >= 0
    video = torch.from_numpy(frames).cuda()
    xin = torch.empty(1, 32, 3, 518, 518, dtype=torch.float32, device="cuda")
    wins = []
    plan = S.plan_windows(n)
    assert len(plan) == 11
    for idx in plan:
        ops.gather_normalize_u8(video, torch.tensor(idx, dtype=torch.int32, device="cuda"), xin, 32, 518, 518)
        wins.append(m.forward(xin, fp32=False)[0].cpu().numpy())
    ref = S.stitch_windows(wins, n)
    # per-window depth maps are bit-identical; the only difference is the scale/shift sums (fp64 on the device, numpy's fp32
    # closed form whose determinant cancels digits), compounded along the 10-window chain
    e = rel_l1(d, ref)
    record("long_video.device_vs_host_stitch", e)
    assert e < 5e-4
    np.testing.assert_allclose(d, ref, rtol=5e-3, atol=5e-3)
    assert np.array_equal(d[:22], wins[0][:22]), "window 0 is stitched unscaled"
    # two windows are in flight on two HIP streams with separate workspace / input / output slots: any sharing mistake between
    # the lanes would show as run-to-run differences
    d2, _ = m.infer_video_depth(frames, 24, input_size=518, device="cuda", fp32=False)
    assert np.array_equal(d, d2), "infer_video_depth must be bitwise reproducible"


def test_key_frame_exchange_equals_window_exchange_single_rank():
    """model.exchange = "keys" (key frames gathered, scale/shift chain everywhere, every rank finalises its own
    windows) on one rank: the same kernels in the same order as the window exchange + device stitcher - bit-equal, relative and
    metric, several window counts."""
    from video_depth_anything_amd.video_depth import MetricVideoDepthAnything, VideoDepthAnything
    for cls in (VideoDepthAnything, MetricVideoDepthAnything):
        m, _, _ = model_for("tiny", 6, cls)
        for n in (9, 40, 77):
            frames = np.random.default_rng(40 + n).integers(0, 256, (n, 42, 56, 3), dtype=np.uint8)
            a, _ = m.infer_video_depth(frames, 24, input_size=42, device="cuda")
            m.exchange = "keys"
            try:
                b, _ = m.infer_video_depth(frames, 24, input_size=42, device="cuda")
            finally:
                m.exchange = "windows"
            assert a.shape == b.shape == (n, 42, 56) and np.array_equal(a, b), (cls.__name__, n)


def test_memory_mapped_video_equals_in_memory_video(tmp_path):
    """utils/dc_utils.read_video_frames hands a .npy video over as a memory map; infer_video_depth pages in the runs each window
    reads and must return exactly what it returns for the same frames held in RAM."""
    from utils.dc_utils import read_video_frames
    m, _, _ = model_for("tiny", 6)
    frames = np.random.default_rng(34).integers(0, 256, (70, 42, 56, 3), dtype=np.uint8)
    np.save(tmp_path / "v.npy", frames)
    lazy, fps = read_video_frames(str(tmp_path / "v.npy"), -1)
    assert isinstance(lazy, np.memmap)
    a, _ = m.infer_video_depth(lazy, fps, input_size=42, device="cuda")
    b, _ = m.infer_video_depth(frames, fps, input_size=42, device="cuda")
    assert a.shape == (70, 42, 56) and np.array_equal(a, b)


def test_frames_that_are_not_uint8():
    """The reference divides whatever array it is handed by 255 (video_depth.py:198); the device path keeps the video as uint8,
    so arrays of any dtype holding 8-bit integer values are converted (same arithmetic) and anything else is refused, never truncated."""
    m, _, _ = model_for("tiny", 6)
    frames = np.random.default_rng(35).integers(0, 256, (12, 28, 42, 3), dtype=np.uint8)
    a, _ = m.infer_video_depth(frames, 24, input_size=28, device="cuda")
    b, _ = m.infer_video_depth(frames.astype(np.int64), 24, input_size=28, device="cuda")
    assert np.array_equal(a, b)
    c, _ = m.infer_video_depth(frames.astype(np.float32), 24, input_size=28, device="cuda")     # float frames holding 0..255 integers
    assert np.array_equal(a, c)
    with pytest.raises(TypeError, match="8-bit"):
        m.infer_video_depth(frames.astype(np.float32) / 255.0, 24, input_size=28, device="cuda")
    with pytest.raises(TypeError, match="8-bit"):
        m.infer_video_depth(frames.astype(np.float32) + 0.5, 24, input_size=28, device="cuda")
    with pytest.raises(TypeError, match="8-bit"):
        m.infer_video_depth(frames.astype(np.int32) * 2, 24, input_size=28, device="cuda")
    with pytest.raises(ValueError, match=r"\[N, H, W, 3\]"):
        m.infer_video_depth(frames[..., 0], 24, input_size=28, device="cuda")


def test_1024_frame_vitl_video_properties():
    """BASELINE.json config 4's single-GPU content: ViT-L, 1024 frames of 518x518 = 47 windows through infer_video_depth."""
    m, _, _ = model_for("vitl", 0)
    n = 1024
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (64, 518, 518, 3), dtype=np.uint8)
    frames = np.concatenate([base] * 16, axis=0)           # 1024 frames, period 64 (keeps host RAM and RNG time down)
    d, fps = m.infer_video_depth(frames, 30, input_size=518, device="cuda", fp32=False)
    assert d.shape == (n, 518, 518) and d.dtype == np.float32 and fps == 30
    assert np.isfinite(d).all() and d.min() >= 0 and d.std() > 0
    # window 0 is stitched unscaled (video_depth.py:219-224): its first frames equal a plain forward of those frames
    from video_depth_anything_amd import ops
    xin = torch.empty(1, 32, 3, 518, 518, dtype=torch.float32, device="cuda")
    ops.gather_normalize_u8(torch.from_numpy(frames[:32]).cuda(), torch.arange(32, dtype=torch.int32, device="cuda"), xin, 32, 518, 518)
    w0 = m.forward(xin, fp32=False)[0].cpu().numpy()
    assert np.array_equal(d[:22], w0[:22])


def test_ranks_share_one_gpu(tmp_path):
    """The multi-rank branch of infer_video_depth rehearsed with 3 processes on this one GPU, 5 windows (uneven shards 2+2+1;
    gloo stands in for RCCL, the exchange is staged through the host): every rank must return exactly what a single rank returns,
    and a rank outside result_ranks must return None."""
    import subprocess
    import sys
    out = tmp_path / "r"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
           "--master-port", "29741", os.path.join(REPO, "tests", "_gpu_ranks_worker.py"), str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    single = np.load(f"{out}_single.npy")
    for rank in range(3):
        d = np.load(f"{out}_rank{rank}.npy")
        assert d.shape == single.shape == (100, 28, 42)
        assert np.array_equal(d, single), f"rank {rank} differs from the single-rank result"
    assert open(f"{out}_result_ranks.txt").read() == "rank0:array rank1:None rank2:None"


def test_bench_two_ranks_rehearsal(tmp_path):
    """bench.py's multi-rank flow (rank env, barrier, per-step exchange, max-over-ranks time, one JSON line from rank 0) with 2 ranks
    sharing this GPU over gloo (VDA_BENCH_BACKEND=gloo: a rehearsal mode, never a measurement; RCCL itself only runs on the driver's
    multi-GPU node). ViT-S keeps it short."""
    import subprocess
    import sys
    env = dict(os.environ, VDA_BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29761",
           os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--encoder", "vits", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["config"]["world"] == 2 and d["config"]["backend"] == "gloo"
    assert d["scaling"] == "weak" and d["value"] > 0 and d["roofline"]["frac"] > 0 and "cpu_baseline" not in d


def test_bench_self_launch_and_video_mode(tmp_path):
    """`python bench.py --gpus 2` with NO launcher around it (the form the driver uses): the parent starts torch.distributed.run as a
    child process, relays the one JSON line and returns its exit code (VERDICT r3 missing #1). Run for the clip bench and for
    --video (BASELINE.json configs[3]: infer_video_depth, windows sharded over the ranks, result on rank 0, output frames/s) with
    2 ranks sharing this GPU over gloo - a rehearsal of the flow, never a measurement."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["VDA_BENCH_BACKEND"] = "gloo"
    for extra, want in ((["--no-inflight2"], "frames/sec at 1x32x518x518 fp16, ViT-S"),
                        (["--video", "76"], "output frames/sec of infer_video_depth on a 76-frame 518x518 video, fp16, ViT-S"),
                        (["--video", "76", "--exchange", "keys"], "output frames/sec of infer_video_depth on a 76-frame 518x518 video, fp16, ViT-S")):
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--encoder", "vits",
                            "--no-cpu-baseline"] + extra, capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        assert "[bench] --gpus 2 without a launcher" in r.stderr
        lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
        assert len(lines) == 1, "exactly one JSON line, from rank 0"
        d = json.loads(lines[0])
        assert d["metric"] == want and d["n_gpus"] == 2 and d["config"]["world"] == 2 and d["config"]["backend"] == "gloo" and d["value"] > 0
        if "--video" in extra:
            assert d["scaling"] == "strong" and d["config"]["exchange"] == (extra[-1] if "--exchange" in extra else "windows"), d
            assert d["config"]["windows"] == 4, d["config"]
    # one rank, the same entry: config 4's metric on one GPU
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--video", "54", "--steps", "1", "--warmup", "1", "--encoder", "vits"],
                       capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["config"]["windows"] == 3 and d["value"] > 0


def test_bench_rccl_calls_at_world_size_one():
    """The RCCL calls of bench.py's multi-rank flow (process-group init bound to the device, all_gather_into_tensor per step, barrier,
    all_reduce of the time) on the real backend - at world size 1, all a one-GPU box allows (VDA_BENCH_FORCE_DIST=1): the API usage
    is exercised, the inter-GPU transport is not."""
    import subprocess
    import sys
    env = dict(os.environ, VDA_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29771", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--encoder", "vits",
                        "--no-cpu-baseline", "--no-inflight2"], capture_output=True, text=True, timeout=600, cwd=REPO, env=env)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads([x for x in r.stdout.splitlines() if x.startswith("{")][-1])
    assert d["config"]["backend"] == "nccl" and d["config"]["world"] == 1 and d["n_gpus"] == 1 and d["value"] > 0


@pytest.mark.parametrize("fp32", PRECISIONS)
@pytest.mark.parametrize("name,kw", [("tiny_bn_forward.npz", {"use_bn": True}), ("tiny_rope_forward.npz", {"pe": "rope"})], ids=["use_bn", "rope"])
def test_golden_tiny_with_bn_and_with_rope(golden_dir, name, kw, fp32):
    """use_bn=True (util/blocks.py:60-62,80-86) and pe='rope' (motion_module.py:221-224,254-257): reference-generated goldens, both
    precisions, through the class and vda_forward. BatchNorm is folded into the convs when the weights are packed; the rotation
    is vda_rope_qk on the fused projection."""
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict
    z = np.load(os.path.join(golden_dir, name))
    cfg = get_config("tiny", **kw)
    m = VideoDepthAnything(encoder="tiny", features=cfg.features, out_channels=list(cfg.out_channels), **kw)
    assert m.cfg == cfg
    sd = synthetic_state_dict(cfg, seed=int(z["sd_seed"]))
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    x = torch.from_numpy(z["x"]).cuda()
    d = m.forward(x, fp32=fp32)
    BT = x.shape[0] * x.shape[1]
    tag = ("tiny_bn." if cfg.use_bn else "tiny_rope.") + ("f32." if fp32 else "")
    for k, C in (("layer_3", cfg.out_channels[2]), ("path_2", cfg.features), ("path_1", cfg.features)):
        t, h, w, Cp = m.engine.stage(k)
        check_map(tag + k, nhwc_to_nchw(t, BT, h, w, Cp, C), z[k], tol_of("tiny.stage", fp32), tail=False)
    check_map(tag + "depth", d.cpu().numpy(), z["depth"], tol_of("tiny_cls.depth", fp32))
    # packing twice (the other precision, then this one again) folds the same BatchNorm again: same bits
    m.forward(x, fp32=not fp32)
    assert torch.equal(m.forward(x, fp32=fp32), d)
    with pytest.raises(NotImplementedError):
        m.python_engine()
    # the strict inventory follows the switch
    with pytest.raises(RuntimeError, match="Missing key" if cfg.use_bn else "Unexpected key"):
        m.load_state_dict(synthetic_state_dict(get_config("tiny"), seed=1), strict=True)
    with pytest.raises(NotImplementedError):
        VideoDepthAnything(encoder="tiny", pe="alibi")


@pytest.mark.parametrize("enc,fixture", [("tiny", "tiny_forward.npz"), ("vits", "vits_forward.npz")])
def test_output_conv1_with_the_upsample_folded_in(golden_dir, enc, fixture):
    """Default fp16 path: refinenet1's 2x upsample is evaluated inside output_conv1 (vda_conv3x3_up2_f16; dpt.py:117 over
    util/blocks.py:156-160). Against the reference-generated golden with the fusion on and off; the two forms round the same
    interpolated pixels to fp16 and differ by the conv's fp32 summation order only, so their depths agree far inside the tolerance."""
    z = np.load(os.path.join(golden_dir, fixture))
    m, cfg, _ = model_for(enc, int(z["sd_seed"]))
    x = torch.from_numpy(z["x"]).cuda()
    d1 = m.forward(x, fp32=False)
    with pytest.raises(Exception):                       # path_1 does not exist at full size on this path ...
        m.engine._check_stage_exists("p1")
    t1 = m.engine.stage("path_1")[0].clone()             # ... stage() rebuilds it from the half-size buffer
    m.engine.set_option("oc1_fused", 0)
    d0 = m.forward(x, fp32=False)
    t0 = m.engine.stage("path_1")[0]
    m.engine.set_option("oc1_fused", 1)
    assert torch.equal(t0, t1), "the half-size path_1, upsampled, is the unfused path's path_1"
    key = "tiny.depth" if enc == "tiny" else "vits.nonsquare.depth"
    check_map(f"{enc}.oc1_fused.depth", d1.cpu().numpy(), z["depth"], tol_of(key, False))
    check_map(f"{enc}.oc1_unfused.depth", d0.cpu().numpy(), z["depth"], tol_of(key, False))
    rel = float((d1 - d0).abs().mean() / d0.abs().mean())
    assert rel < 1e-4, rel
    assert torch.equal(m.forward(x, fp32=False), d1)


@pytest.mark.parametrize("fp32", PRECISIONS)
def test_oracle_vits_with_bn_and_rope_32_frames(fp32):
    """use_bn=True and pe='rope' TOGETHER at the released ViT-S widths and the full temporal length (T = 32: every frame's rotation
    angle, head dims 8 / 24 / 48 of the temporal attention) on a small spatial grid, against the CPU oracle - which the
    reference-generated tiny fixtures pin for both switches (tests/test_oracle_golden.py)."""
    from oracle import vda_oracle as O
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict
    cfg = get_config("vits", use_bn=True, pe="rope")
    sd = synthetic_state_dict(cfg, seed=21)
    m = VideoDepthAnything(encoder="vits", features=cfg.features, out_channels=list(cfg.out_channels), use_bn=True, pe="rope")
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda").eval()
    x = torch.randn(1, 32, 3, 70, 98, generator=torch.Generator().manual_seed(22))
    with torch.no_grad():
        ref = oracle_once("vits.bn_rope.t32", lambda: O.forward(sd, cfg, x).numpy())
    d = m.forward(x.cuda(), fp32=fp32).cpu().numpy()
    e = check_map("vits.bn_rope.t32.depth_vs_oracle" + (".f32" if fp32 else ""), d, ref, tol_of("vits.bn_rope.t32", fp32))
    if not fp32:
        # (bound = 2x the first measurement, 3.8e-3: BatchNorm with random running statistics rescales channels by up to ~3x)
        e16 = autocast_oracle_error(sd, cfg, x, ref)          # the reference's own fp16 path, when the yardstick run is on
        if e16 is not None:
            record("vits.bn_rope.t32.autocast_oracle_vs_fp32", e16)
            assert e is None or e < 1.5 * e16 + 5e-4


@pytest.mark.parametrize("n_frames", [1, 5, 31, 32, 33, 54, 55])
def test_short_and_ragged_videos_against_the_oracle(n_frames):
    """The sliding window's edge cases (video_depth.py:166-254): a single frame, fewer frames than one window (the last frame is
    repeated to fill it), exactly one window, one frame into the second window, and the 54 / 55 boundary where a third window
    appears - infer_video_depth in the fp32 path against the oracle's, tiny model (the stitch amplifies nothing in fp32)."""
    from oracle import vda_oracle as O
    m, cfg, sd = model_for("tiny", 3)
    frames = np.random.default_rng(100 + n_frames).integers(0, 256, (n_frames, 28, 42, 3), dtype=np.uint8)
    ref = O.infer_video_depth(sd, cfg, frames, 24, input_size=28)[0]
    d, fps = m.infer_video_depth(frames, 24, input_size=28, device="cuda", fp32=True)
    assert d.shape == ref.shape == (n_frames, 28, 42) and d.dtype == np.float32 and fps == 24
    check_map(f"video.ragged.n{n_frames}.f32", d, ref, TOL32, tail=False)
    d16, _ = m.infer_video_depth(frames, 24, input_size=28, device="cuda", fp32=False)
    check_map(f"video.ragged.n{n_frames}", d16, ref, TOL16["video.ragged"], tail=False)      # 2x the largest measured (n = 1: 4.2e-3, a 2 x 3 token grid)
