"""Host logic of infer_video_depth (CPU): window plan, sharding, stitcher, and the multi-rank path
over gloo with world_size 2. The per-window compute is the CPU oracle's forward (tests may use it)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import vda_oracle as O
from video_depth_anything_amd import scheduler as S
from video_depth_anything_amd.config import INFER_LEN, KEYFRAMES, OVERLAP

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_schedule(n):
    """Simulate video_depth.py:187-201 on frame INDICES instead of pixels."""
    step = INFER_LEN - OVERLAP
    append = (step - (n % step)) % step + (INFER_LEN - step)
    lst = list(range(n)) + [n - 1] * append
    out, pre = [], None
    for fid in range(0, n, step):
        cur = [lst[fid + i] for i in range(INFER_LEN)]
        if pre is not None:
            cur[:OVERLAP] = [pre[k] for k in KEYFRAMES]
        out.append(cur)
        pre = cur
    return out


@pytest.mark.parametrize("n", [1, 5, 22, 23, 32, 33, 44, 50, 100, 1024])
def test_plan_matches_reference_recursion(n):
    assert S.plan_windows(n) == reference_schedule(n)


def test_plan_counts_and_slot_sources():
    assert len(S.plan_windows(32)) == 2 and len(S.plan_windows(1024)) == 47
    w = S.plan_windows(100)
    for k in range(1, len(w)):
        assert w[k][0] == 0, "slot 0 is always video frame 0"
        assert w[k][1] == 22 * k - 10, "slot 1 is the previous window's slot 12"
        assert w[k][2:10] == list(range(22 * k + 2, 22 * k + 10)), "slots 2..9 coincide with native frames"


def test_plan_rejects_empty():
    with pytest.raises(ValueError):
        S.plan_windows(0)


@pytest.mark.parametrize("W,G", [(47, 8), (2, 8), (5, 2), (8, 8), (1, 1), (3, 4)])
def test_shards_partition_all_windows(W, G):
    got = sorted(k for r in range(G) for k in S.shard_windows(W, G, r))
    assert got == list(range(W))
    sizes = [len(S.shard_windows(W, G, r)) for r in range(G)]
    assert max(sizes) - min(sizes) <= 1
    if (W, G) == (47, 8):
        assert sizes == [6, 6, 6, 6, 6, 6, 6, 5]


def test_network_size_matches_oracle():
    for h, w, s in [(518, 518, 518), (480, 640, 518), (720, 1280, 518), (1080, 1920, 518), (1920, 1080, 518), (42, 56, 42),
                    (300, 900, 518), (100, 100, 518)]:
        oh, ow, _ = O.network_size(h, w, s)
        assert S.network_size(h, w, s) == (oh, ow)
        assert oh % 14 == 0 and ow % 14 == 0


@pytest.mark.parametrize("metric", [False, True])
def test_stitcher_equals_oracle_stitcher(metric):
    rng = np.random.default_rng(3)
    n = 70
    nwin = len(S.plan_windows(n))
    wins = [(rng.random((INFER_LEN, 6, 7), dtype=np.float32) * (1 + 0.3 * k) + 0.1 * k) for k in range(nwin)]
    flat = [w[i] for w in wins for i in range(INFER_LEN)]
    a = S.stitch_windows(wins, n, metric)
    b = O.stitch(flat, n, metric)
    assert a.dtype == np.float32 and a.shape == (n, 6, 7)
    np.testing.assert_array_equal(a, b)


def test_stitcher_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "stitch_math.npz"))
    s, t = S.compute_scale_and_shift(z["pred"], z["targ"])
    assert s == pytest.approx(float(z["scale"]), rel=1e-6) and t == pytest.approx(float(z["shift"]), rel=1e-6)
    np.testing.assert_array_equal(np.stack(S.crossfade(list(z["pre"]), list(z["post"]))), z["mix"])
    assert S.compute_scale_and_shift(np.zeros((4, 4), np.float32), np.ones((4, 4), np.float32)) == (1, 0)   # det == 0


def test_normalize_host_matches_oracle():
    rng = np.random.default_rng(4)
    fr = rng.integers(0, 256, (3, 28, 42, 3), dtype=np.uint8)
    a = S.normalize_frames_host(fr)
    b = np.stack([O.preprocess_frame(f, 28) for f in fr])
    np.testing.assert_array_equal(a, b)


def test_run_windows_single_rank_reproduces_reference_video(golden_dir):
    """Scheduler + stitcher around the oracle forward == the reference's infer_video_depth golden."""
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.weights import synthetic_state_dict
    z = np.load(os.path.join(golden_dir, "tiny_video.npz"))
    cfg = get_config("tiny")
    sd = synthetic_state_dict(cfg, seed=int(z["sd_seed"]))

    def window_fn(win_u8):
        x = torch.from_numpy(S.normalize_frames_host(win_u8))[None]
        with torch.no_grad():
            return O.forward(sd, cfg, x)[0].numpy()

    d = S.run_windows(z["frames"], window_fn)
    err = np.abs(d - z["depths"]).max() / np.abs(z["depths"]).max()
    assert err < 5e-5


@pytest.mark.parametrize("exchange", ["windows", "keys"])
@pytest.mark.parametrize("world,n_frames", [(2, 60), (3, 100)], ids=["2ranks_3windows", "3ranks_5windows"])
def test_ranks_over_gloo_equal_one_rank(tmp_path, world, n_frames, exchange):
    """gloo, world size 2 and 3 (uneven shards: 2+1 and 2+2+1 windows): the round-robin schedule of drive_windows - the one the
    device path runs - with its per-round all-gather and two-slot rings == the single-process result, on every rank.
    exchange="keys": the key-frame schedule of SURVEY.md section 8(e) (drive_windows_keys: 11 key frames per window all-gathered, the
    scale/shift chain on every rank, each rank finalises its own windows, pieces delivered to the result ranks): bit-equal too,
    with every rank or only rank 1 as the receiver."""
    out = tmp_path / "r"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29731 + world + (10 if exchange == "keys" else 0)), os.path.join(REPO, "tests", "_gloo_worker.py"), str(out), str(n_frames),
           exchange]
    env = dict(os.environ, OMP_NUM_THREADS="2")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=REPO)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    ref = np.load(str(out) + "_single.npy")
    assert ref.shape == (n_frames, 28, 42)
    for rank in range(world):
        np.testing.assert_array_equal(np.load(f"{out}_rank{rank}.npy"), ref)


def test_key_schedule_equals_window_schedule_single_rank():
    """drive_windows_keys against drive_windows + stitch_windows on one rank: bit-equal for every window count / remainder, both
    stitch modes; every output frame is delivered exactly once (asserted inside run_windows)."""
    rng = np.random.default_rng(3)
    for n in (1, 20, 32, 33, 54, 55, 76, 131):
        frames = rng.integers(0, 256, (n, 6, 8, 3), dtype=np.uint8)

        def window_fn(w):
            x = w.astype(np.float32).mean(-1)
            return (x * (1 + 0.001 * x.mean()) + 3).astype(np.float32)

        for metric in (False, True):
            a = S.run_windows(frames, window_fn, metric=metric)
            b = S.run_windows(frames, window_fn, metric=metric, exchange="keys")
            assert np.array_equal(a, b), (n, metric)


def test_round_robin_shards_cover_every_window_once():
    for nwin, world in [(47, 8), (3, 2), (5, 3), (1, 4), (8, 8)]:
        got = sorted(k for r in range(world) for k in S.shard_windows(nwin, world, r))
        assert got == list(range(nwin))
        assert max(len(S.shard_windows(nwin, world, r)) for r in range(world)) == S.rounds(nwin, world)
