"""Frame I/O around `infer_video_depth`: the two helpers the reference's CLIs call
(/root/reference/utils/dc_utils.py:18-88 `read_video_frames`, `save_video`; same names, arguments and return values).

Not on the accelerated path. The reference decodes with decord / cv2 and encodes H.264 with
imageio-ffmpeg; none of those exist offline, so each is used when importable and otherwise replaced by what the image has:

  read_video_frames: .npy / .npz (`frames`, optional `fps`), a directory of still images, or any multi-frame file PIL opens
                     (GIF, APNG, TIFF); a video file needs decord or cv2.
  save_video       : mp4 through imageio when present, else an animated GIF (PIL) next to the requested name.

Frames larger than `max_res` are scaled like the reference's cv2 branch (round(size * max_res / max(h, w)), bilinear, half-pixel
centres, no antialiasing = cv2.INTER_LINEAR's definition); cv2 is not here to pin that bit for bit.
"""
import os

import numpy as np

IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".tif", ".tiff", ".webp")


def _resize_bilinear(frames, height, width):
    import torch
    x = torch.from_numpy(np.array(frames)).permute(0, 3, 1, 2).float()       # a copy: the source may be a read-only memory map
    y = torch.nn.functional.interpolate(x, size=(height, width), mode="bilinear", align_corners=False, antialias=False)
    return y.round().clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).contiguous().numpy()


def _decode(video_path):
    """-> (uint8 [N,H,W,3] RGB, source fps)"""
    ext = os.path.splitext(video_path)[1].lower()
    if os.path.isdir(video_path):
        from PIL import Image
        names = sorted(n for n in os.listdir(video_path) if n.lower().endswith(IMAGE_EXT + (".npy",)))
        if not names:
            raise ValueError(f"no frames in {video_path}")
        frames = [np.load(os.path.join(video_path, n)) if n.endswith(".npy") else np.asarray(Image.open(os.path.join(video_path, n)).convert("RGB"))
                  for n in names]
        return np.stack(frames), 24.0
    if ext == ".npy":
        # memory-mapped: frames are paged in when infer_video_depth uploads the windows that read them, so a long video is never
        # held in RAM as one array (the reference's decoders return the whole video, dc_utils.py:19-69)
        return np.load(video_path, mmap_mode="r"), 24.0
    if ext == ".npz":
        z = np.load(video_path)
        return z["frames"], float(z["fps"]) if "fps" in z else 24.0
    if ext in (".gif", ".png", ".apng", ".tif", ".tiff", ".webp"):
        from PIL import Image, ImageSequence
        im = Image.open(video_path)
        dur = im.info.get("duration", 0) or 0
        frames = np.stack([np.asarray(f.convert("RGB")) for f in ImageSequence.Iterator(im)])
        return frames, (1000.0 / dur if dur > 0 else 24.0)
    try:
        from decord import VideoReader, cpu
        vid = VideoReader(video_path, ctx=cpu(0))
        return vid.get_batch(list(range(len(vid)))).asnumpy(), float(vid.get_avg_fps())
    except ImportError:
        pass
    try:
        import cv2
    except ImportError as e:
        raise RuntimeError(f"cannot decode {video_path}: neither decord nor cv2 is importable here; "
                           "pass frames as .npy / .npz, a directory of images or a GIF") from e
    cap = cv2.VideoCapture(video_path)
    fps = cap.get(cv2.CAP_PROP_FPS)
    out = []
    while True:
        ok, f = cap.read()
        if not ok:
            break
        out.append(cv2.cvtColor(f, cv2.COLOR_BGR2RGB))
    cap.release()
    return np.stack(out), float(fps)


def read_video_frames(video_path, process_length, target_fps=-1, max_res=-1):
    """dc_utils.py:18-70: every `stride`-th frame (stride = max(round(src_fps / fps), 1)), at most `process_length`
    frames read, frames larger than `max_res` scaled down. Returns (uint8 [N,H,W,3], fps)."""
    frames, src_fps = _decode(video_path)
    if not isinstance(frames, np.ndarray):
        frames = np.asarray(frames)
    if frames.ndim != 4 or frames.shape[-1] != 3:
        raise ValueError(f"expected frames [N,H,W,3], got {frames.shape}")
    fps = src_fps if target_fps < 0 else target_fps
    stride = max(round(src_fps / fps), 1)
    if process_length > 0:
        frames = frames[:process_length]              # the cv2 branch counts SOURCE frames (dc_utils.py:57)
    h, w = frames.shape[1:3]
    lazy = isinstance(frames, np.memmap) and stride == 1 and frames.dtype == np.uint8 and not (max_res > 0 and max(h, w) > max_res)
    if lazy:
        return frames, fps                            # still on disk: infer_video_depth pages in what each window reads
    frames = np.ascontiguousarray(frames[::stride], dtype=np.uint8)
    if max_res > 0 and max(h, w) > max_res:
        scale = max_res / max(h, w)
        frames = _resize_bilinear(frames, round(h * scale), round(w * scale))
    return frames, fps


def _inferno(u8):
    """uint8 [..] -> uint8 [..,3]: degree-6 polynomial fit of matplotlib's 'inferno' (the reference indexes the 256-entry table,
    dc_utils.py:75-83; matplotlib is not installed here). VISUALISATION ONLY and unpinned: no fixture of the reference holds a
    colour-mapped frame; the depth values themselves (npz / exr / the returned array) never pass through this."""
    t = u8.astype(np.float32) / 255.0
    c = np.array([[0.0002189403691192265, 0.001651004631001012, -0.01948089843709184],
                  [0.1065134194856116, 0.5639564367884091, 3.932712388889277],
                  [11.60249308247187, -3.972853965665698, -15.9423941062914],
                  [-41.70399613139459, 17.43639888205313, 44.35414519872813],
                  [77.162935699427, -33.40235894210092, -81.80730925738993],
                  [-71.31942824499214, 32.62606426397723, 73.20951985803202],
                  [25.13112622477341, -12.24266895238567, -23.07032500287172]], dtype=np.float32)
    rgb = np.zeros(t.shape + (3,), dtype=np.float32)
    for k in range(6, -1, -1):
        rgb = rgb * t[..., None] + c[k]
    return (np.clip(rgb, 0, 1) * 255).astype(np.uint8)


def save_video(frames, output_video_path, fps=10, is_depths=False, grayscale=False):
    """dc_utils.py:73-88. Depth is mapped through its GLOBAL min / max to uint8 (then inferno unless `grayscale`).
    Returns the path written (the reference returns None): the .mp4 asked for, or a .gif when there is no encoder."""
    frames = np.asarray(frames)
    if is_depths:
        d_min, d_max = frames.min(), frames.max()
        norm = ((frames - d_min) / max(float(d_max - d_min), 1e-12) * 255).astype(np.uint8)
        vis = norm if grayscale else _inferno(norm)
    else:
        vis = frames
    try:
        import imageio
    except ImportError:
        imageio = None                                 # no H.264 encoder in this image: animated GIF instead (stated in the docstring)
    if imageio is not None:
        # an encoder that IS installed and fails (bad path, codec error) raises: the failure must not turn into a silent GIF
        writer = imageio.get_writer(output_video_path, fps=fps, macro_block_size=1, codec='libx264', ffmpeg_params=['-crf', '18'])
        for f in vis:
            writer.append_data(f)
        writer.close()
        return output_video_path
    else:
        from PIL import Image
        path = os.path.splitext(output_video_path)[0] + ".gif"
        ims = [Image.fromarray(f) for f in vis]
        ims[0].save(path, save_all=True, append_images=ims[1:], duration=max(int(round(1000.0 / max(fps, 1e-6))), 1), loop=0)
        return path
