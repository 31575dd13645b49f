"""utils/dc_utils.py (the reference's read_video_frames / save_video surface) with the decoders this image has."""
import os

import numpy as np

from utils.dc_utils import read_video_frames, save_video


def _frames(n=7, h=20, w=30, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def test_npz_stride_and_length(tmp_path):
    f = _frames(12)
    np.savez(tmp_path / "v.npz", frames=f, fps=30)
    out, fps = read_video_frames(str(tmp_path / "v.npz"), -1)
    assert np.array_equal(out, f) and fps == 30
    out, fps = read_video_frames(str(tmp_path / "v.npz"), 9, target_fps=10)     # stride 3 over the first 9 source frames
    assert fps == 10 and np.array_equal(out, f[:9:3])


def test_directory_of_images_and_gif_round_trip(tmp_path):
    from PIL import Image
    f = _frames(5)
    d = tmp_path / "clip"
    os.makedirs(d)
    for i, im in enumerate(f):
        Image.fromarray(im).save(d / f"{i:03d}.png")
    out, fps = read_video_frames(str(d), -1)
    assert np.array_equal(out, f) and fps == 24
    path = save_video(f, str(tmp_path / "o_src.mp4"), fps=12)
    assert os.path.exists(path)
    if path.endswith(".gif"):                       # no H.264 encoder here: palette-quantised, so only geometry is checked
        back, fps2 = read_video_frames(path, -1)
        assert back.shape == f.shape and abs(fps2 - 12) < 1.0     # GIF delays are multiples of 10 ms


def test_max_res_and_depth_visualisation(tmp_path):
    f = _frames(3, 40, 64)
    np.save(tmp_path / "v.npy", f)
    out, _ = read_video_frames(str(tmp_path / "v.npy"), -1, max_res=32)
    assert out.shape == (3, 20, 32, 3) and out.dtype == np.uint8
    depth = np.linspace(0, 5, 3 * 8 * 9, dtype=np.float32).reshape(3, 8, 9)
    p = save_video(depth, str(tmp_path / "d_vis.mp4"), fps=5, is_depths=True)
    g = save_video(depth, str(tmp_path / "g_vis.mp4"), fps=5, is_depths=True, grayscale=True)
    assert os.path.exists(p) and os.path.exists(g)


def test_npy_is_memory_mapped_until_a_window_reads_it(tmp_path):
    """A .npy video stays on disk: read_video_frames returns a memory map (no stride, no down-scaling), a strided or
    down-scaled read materialises it like the reference's decoders do."""
    f = _frames(9)
    np.save(tmp_path / "v.npy", f)
    out, fps = read_video_frames(str(tmp_path / "v.npy"), -1)
    assert isinstance(out, np.memmap) and out.shape == f.shape and fps == 24.0 and np.array_equal(out, f)
    out, _ = read_video_frames(str(tmp_path / "v.npy"), 5)
    assert isinstance(out, np.memmap) and np.array_equal(out, f[:5])
    out, _ = read_video_frames(str(tmp_path / "v.npy"), -1, target_fps=12)
    assert not isinstance(out, np.memmap) and np.array_equal(out, f[::2])
    out, _ = read_video_frames(str(tmp_path / "v.npy"), -1, max_res=10)
    assert not isinstance(out, np.memmap) and max(out.shape[1:3]) == 10
