"""A C++ process with no Python and no torch in it drives the handle API (examples/host_demo.cpp: vda_create ->
vda_load_weight x N -> vda_finalize_weights -> vda_forward) and must produce, bit for bit, what VideoDepthAnything.forward
produces on the same device - the C-ABI drop-in of SURVEY.md section 8(b) exercised from the language it is meant for."""
import os
import struct
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_model(path, cfg, sd):
    with open(path, "wb") as f:
        f.write(struct.pack("<16i", cfg.embed_dim, cfg.depth, cfg.num_heads, *cfg.taps, cfg.features, *cfg.out_channels, cfg.num_frames, int(cfg.use_clstoken),
                            int(cfg.use_bn), int(cfg.pe == "rope")))
        f.write(struct.pack("<i", len(sd)))
        for name, t in sd.items():
            nb = name.encode()
            a = t.detach().float().contiguous().numpy()
            f.write(struct.pack("<i", len(nb)) + nb + struct.pack("<i", a.ndim) + struct.pack(f"<{a.ndim}q", *a.shape))
            f.write(a.tobytes())


@pytest.mark.parametrize("name,shape", [("tiny", (1, 4, 3, 42, 56)), ("vits", (1, 3, 3, 56, 70))])
def test_cpp_host_matches_the_python_class(tmp_path, name, shape):
    from video_depth_anything_amd import build
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict
    exe = build.build_host_demo()
    cfg = get_config(name)
    sd = synthetic_state_dict(cfg, seed=3)
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(9))
    write_model(tmp_path / "model.bin", cfg, sd)
    with open(tmp_path / "input.bin", "wb") as f:
        f.write(struct.pack("<4i", shape[0], shape[1], shape[3], shape[4]) + x.numpy().tobytes())
    m = VideoDepthAnything(encoder=name, features=cfg.features, out_channels=list(cfg.out_channels))
    m.load_state_dict(sd, strict=True)
    m = m.to("cuda")
    for prec, fp32 in ((0, False), (1, True)):
        r = subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "input.bin"), str(tmp_path / "out.bin"), str(prec)],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-1000:]
        got = np.fromfile(tmp_path / "out.bin", dtype=np.float32).reshape(shape[0], shape[1], shape[3], shape[4])
        ref = m.forward(x.cuda(), fp32=fp32).cpu().numpy()
        assert np.array_equal(got, ref), f"{name} precision {prec}: the C++ host and the Python class differ"


def test_cpp_host_reports_state_dict_errors(tmp_path):
    """strict=True semantics from C: a missing tensor is refused by vda_finalize_weights in torch's wording."""
    from video_depth_anything_amd import build
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.weights import synthetic_state_dict
    exe = build.build_host_demo()
    cfg = get_config("tiny")
    sd = synthetic_state_dict(cfg, seed=3)
    write_model(tmp_path / "model.bin", cfg, sd)
    raw = bytearray(open(tmp_path / "model.bin", "rb").read())
    raw[64:68] = struct.pack("<i", len(sd) - 1)          # claim one tensor fewer than the handle expects
    open(tmp_path / "model.bin", "wb").write(raw)
    open(tmp_path / "input.bin", "wb").write(struct.pack("<4i", 1, 1, 14, 14) + np.zeros(3 * 14 * 14, np.float32).tobytes())
    r = subprocess.run([exe, str(tmp_path / "model.bin"), str(tmp_path / "input.bin"), str(tmp_path / "out.bin"), "0"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "tensor count differs" in r.stderr
