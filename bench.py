#!/usr/bin/env python
"""Headline benchmark: frames/s of VideoDepthAnything.forward on synthetic 1x32x518x518, ViT-L, fp16
operands (BASELINE.json metric / configs[2]), on N MI355X of one node.

  python bench.py [--gpus N --steps K --warmup W] [--encoder vitl|vits] [--video N_FRAMES]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W
          (a bare `python bench.py --gpus N` starts exactly that command as a CHILD process and relays its one JSON line)
  --video N_FRAMES: BASELINE.json configs[3] - infer_video_depth over N synthetic 518x518 uint8 frames (sliding windows sharded
          over the ranks, one all-gather per round to stitch, result on rank 0); `value` = OUTPUT frames/s, host uint8 in ->
          host fp32 out (the API's boundary: PCIe-inclusive, stitch included).

A step = one forward pass over one 32-frame clip (one sliding window) per rank, input resident in HBM.
Windows are independent units: ranks shard them with no data-path collective; for N > 1
each step's depth maps are all-gathered (what the stitcher needs) asynchronously under the next step's compute; the timed region
ends when every gather has completed. Rank 0 prints
ONE JSON line; `value` is whole-job frames/s. `roofline` is for the dominant kernel, from HIP events
recorded on the launch stream around each of its launches inside the timed region; `cpu_baseline` is
the CPU oracle (oracle/vda_oracle.py, the checker) timed on a bounded sample on rank 0 at N=1.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# algorithmic FLOPs per 32-frame clip (2*MAC, matmul/conv only, reference op count)
CLIP_TFLOP = {"vitl": 44.95, "vits": 3.881}
MFMA_PEAK_TFLOPS = 2500.0      # MI355X dense fp16/bf16 (MI355X_MICROARCH.md)


def physical_cores():
    """(physical cores, CPU model) of the host: torch's default thread count is the LOGICAL count, which oversubscribes
    the FMA units of an SMT machine (round 1: 128 threads ran slower than 8 threads in the build container)."""
    model, pairs, phys, core = "unknown", set(), None, None
    try:
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                core = v
            elif not k and phys is not None and core is not None:
                pairs.add((phys, core))
                phys = core = None
    except OSError:
        pass
    n = len(pairs) or max(1, (os.cpu_count() or 2) // 2)
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:                                         # a cgroup CPU quota (the GPU box gives one GPU's share of the host) caps it too
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n, model


def best_thread_count(forward, limit):
    """torch-CPU does not scale to every core of a two-socket host on these shapes (round 1: 128 threads ran slower than 8 threads
    did in the build container): time one small forward at a few thread counts and keep the fastest."""
    best, best_t = limit, None
    for n in sorted({c for c in (8, 16, 32, 64, limit) if c <= limit}):
        torch.set_num_threads(n)
        forward()                                # warm
        t0 = time.perf_counter()
        forward()
        t = time.perf_counter() - t0
        if best_t is None or t < best_t:
            best, best_t = n, t
    return best


def cpu_baseline(encoder, full=True):
    """The CPU oracle (the checker, kind = "port") on the bench's own x = randn(1,32,3,518,518) and weights, on the host's
    physical cores (thread count = the fastest of a short probe). The WHOLE 32-frame clip: ViT-S 2 reps
    (~15 s each on the box's 16 cores), ViT-L 1 rep (~70 s) after a 1-frame warm-up. --cpu-sample bounds ViT-L to the first 4
    frames of x at the full 518x518 instead (per-frame cost is what such a sample preserves: encoder and head FLOPs are per
    frame, temporal attention is 0.07 % of the clip)."""
    from oracle import vda_oracle as O
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.weights import synthetic_state_dict
    cfg = get_config(encoder)
    sd = synthetic_state_dict(cfg, seed=0)
    limit, cpu = physical_cores()
    x = torch.randn(1, 32, 3, 518, 518, generator=torch.Generator().manual_seed(0))
    frames, reps = (32, 2) if encoder == "vits" else ((32, 1) if full else (4, 1))
    with torch.no_grad():
        cores = best_thread_count(lambda: O.forward(sd, cfg, x[:, :1]), limit)        # also the warm-up
        torch.set_num_threads(cores)
        t0 = time.perf_counter()
        for _ in range(reps):
            O.forward(sd, cfg, x[:, :frames])
        dt = (time.perf_counter() - t0) / reps
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu,
            "sample": f"{encoder} fp32 torch-CPU oracle on x[:, :{frames}] of the bench's randn(1,32,3,518,518) "
                      f"({'the whole clip' if frames == 32 else f'{frames} of its 32 frames at full 518x518'}), warm-up + {reps} rep(s), "
                      f"{dt:.1f} s per rep, {cores} threads (fastest of a probe over 8..{limit}; host has {limit} usable physical cores)"}


def self_launch(n):
    """`python bench.py --gpus N` with no launcher around it: start `python -m torch.distributed.run ... bench.py <same args>` as a
    CHILD process (never an exec of this one) and relay its output; this parent makes no GPU call at all. Returns the child's
    exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:                       # a free rendezvous port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:                          # relay (rank 0's one JSON line among it)
        sys.stdout.write(line)
        sys.stdout.flush()
    return child.wait()


def video_bench(args, model, dist, backend, world, rank, dev):
    """BASELINE.json configs[3] / SURVEY.md section 8(d): infer_video_depth (video_depth.py:166-254) over N synthetic 518x518 frames.
    Windows are sharded round-robin over the ranks with no data-path collective; one all-gather per round delivers the windows to
    the stitcher (rank 0 returns the video, the other ranks return None). A step = one pass over the whole video."""
    import numpy as np
    from video_depth_anything_amd.scheduler import plan_windows
    n = args.video
    frames = np.random.default_rng(0).integers(0, 256, (n, 518, 518, 3), dtype=np.uint8)      # the same video on every rank
    model.result_ranks = (0,)
    model.exchange = args.exchange
    warm = frames[:min(n, 76)]                          # 4 windows: allocations, first-touch, both lanes
    for _ in range(max(args.warmup, 1)):
        model.infer_video_depth(warm, 24, fp32=args.fp32)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for _ in range(args.steps):
        out, _ = model.infer_video_depth(frames, 24, fp32=args.fp32)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if backend == "nccl":
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        else:
            tcpu = tmax.cpu()
            dist.all_reduce(tcpu, op=dist.ReduceOp.MAX)
            tmax = tcpu
        dt = float(tmax.item())
    if rank != 0:
        return
    assert out is not None and out.shape == (n, 518, 518) and out.dtype == np.float32 and np.isfinite(out).all() and (out >= 0).all()
    nw = len(plan_windows(n))
    prec = "fp32" if args.fp32 else "fp16"
    peak = 157.3 if args.fp32 else MFMA_PEAK_TFLOPS
    tflops = CLIP_TFLOP[args.encoder] * nw * args.steps / dt
    line = {
        "metric": f"output frames/sec of infer_video_depth on a {n}-frame 518x518 video, {prec}, {'ViT-L' if args.encoder == 'vitl' else 'ViT-S'}",
        "value": args.steps * n / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1),
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if args.fp32 else "f16", "data": "synthetic",
        "config": {"workload": f"{args.encoder} {prec} infer_video_depth, {n} uint8 frames of 518x518 (rng seed 0), {nw} sliding windows of 32 frames "
                               f"(stride 22), seeded random weights (BASELINE.json configs[3]); host uint8 in -> host fp32 out on rank 0, stitch included "
                               f"(PCIe-inclusive: the API's boundary)",
                   "windows": nw, "computed_frames_per_s": args.steps * nw * 32 / dt, "ms_per_window": dt / args.steps / nw * 1e3,
                   "windows_in_flight_per_gpu": 2, "exchange": args.exchange,
                   "parallelism": f"windows round-robin over {world} rank(s)" + (", one all-gather per round to stitch, result on rank 0" if world > 1 else ""),
                   "world": (dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else None)},
        "model_tflops": tflops, "model_mfma_frac": tflops / (peak * world),
        "roofline": None, "cpu_baseline": None,
        "note": "video mode: the per-kernel roofline and the CPU baseline are those of the clip bench (python bench.py), whose kernels these are; "
                "with two windows in flight per GPU a launch's duration includes co-running kernels and is not reported",
    }
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--encoder", default="vitl", choices=["vitl", "vits"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", action="store_true", help="time the CPU oracle on 4 of the 32 frames instead of the whole clip (ViT-L: ~10 s instead of ~70 s)")
    ap.add_argument("--fp32", action="store_true", help="bench the fp32-operand path (the reference's --fp32) instead of the headline fp16 path")
    ap.add_argument("--no-inflight2", action="store_true", help="skip the extra (untimed-for-`value`) pass with two clips in flight on two HIP streams")
    ap.add_argument("--video", type=int, default=0, metavar="N_FRAMES",
                    help="configs[3]: time infer_video_depth over N synthetic 518x518 frames (output frames/s) instead of the clip forward")
    ap.add_argument("--exchange", default="windows", choices=["windows", "keys"], help="--video, N > 1: what the ranks exchange per round")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))                 # the parent touches no GPU; the ranks are its grandchildren
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE={world}): pass the same N to both")
    # VDA_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks share devices, the exchange is
    # staged through the host): it exercises the rank / barrier / reduction logic only - never a measurement.
    backend = os.environ.get("VDA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # VDA_BENCH_FORCE_DIST=1: initialise the process group and run the exchange even at world size 1 - the only way to put the
    # RCCL calls themselves (init with device_id, all_gather_into_tensor, all_reduce, barrier) through a one-GPU box
    if world > 1 or os.environ.get("VDA_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # NCCL == RCCL over xGMI on ROCm
        else:
            dist.init_process_group(backend)

    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict

    cfg = get_config(args.encoder)
    model = VideoDepthAnything(encoder=args.encoder, features=cfg.features, out_channels=list(cfg.out_channels))
    model.load_state_dict(synthetic_state_dict(cfg, seed=0), strict=True)
    model = model.to(dev).eval()
    if args.video > 0:
        video_bench(args, model, dist, backend, world, rank, dev)
        if dist is not None:
            dist.destroy_process_group()
        return
    T, H, W = 32, 518, 518
    x = torch.randn(1, T, 3, H, W, generator=torch.Generator().manual_seed(rank)).to(dev)   # resident in HBM

    if dist is not None:
        # the per-step all-gather runs beside the next step's kernels: with dynamic tile draws a GEMM that finds part of the GPU
        # taken by the communication kernel loses those CUs, not a whole shift of tiles (tools/contention.py)
        model.engine.set_option("dyn_sched", 1)
    fwd = lambda: model.forward(x, fp32=args.fp32)          # explicit precision: a bare model(x) follows torch.autocast
    for _ in range(args.warmup):
        fwd()
    outs = torch.empty(args.steps, T, H, W, dtype=torch.float32, device=dev)
    gathered = torch.empty(args.steps, world, T, H, W, dtype=torch.float32, device=dev) if dist is not None else None

    def exchange(s):
        """The one exchange of the path: this step's depth maps to every rank (what the stitcher needs). Issued per step and
        asynchronously - RCCL runs it on its own stream behind the producing kernels, under the next step's compute."""
        if backend == "nccl":
            return dist.all_gather_into_tensor(gathered[s], outs[s], async_op=True)
        parts = [torch.empty(T, H, W, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, outs[s].cpu())
        gathered[s].copy_(torch.stack(parts))
        return None
    # 1 in 8 GEMM / conv launches of each shape with at least 40 GFLOP is bracketed by HIP events on the launch stream (an event pair
    # costs ~10 us of dispatch: every launch bracketed is ~1 ms per ViT-L clip, this is < 0.15 ms; the small launches of the head
    # cannot be the dominant kernel and are only counted)
    model.engine.set_option("profile_min_gflop", 40 if args.encoder == "vitl" else 4)
    model.engine.profile_start(every=8)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = []
    for s in range(args.steps):
        outs[s].copy_(fwd()[0])
        if dist is not None:
            pending.append(exchange(s))
    if dist is not None:
        for h in pending:
            if h is not None:
                h.wait()
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = model.engine.profile_stop()          # {kernel: launches, flops, timed, timed_ms, timed_flops}

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # Extra pass, NOT part of `value`: the same K forwards with TWO clips in flight per GPU (two HIP streams, two workspace slots) -
    # what infer_video_depth does with a video's independent windows. Reported beside the one-at-a-time headline number.
    inflight2 = None
    if world == 1 and not args.no_inflight2:
        lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
        for j in range(2):
            with torch.cuda.stream(lanes[j]):
                model.engine.forward(x, fp32=args.fp32, slot=j)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for s in range(args.steps):
            with torch.cuda.stream(lanes[s & 1]):
                outs[s].copy_(model.engine.forward(x, fp32=args.fp32, slot=s & 1)[0])
        torch.cuda.synchronize()
        dt2 = time.perf_counter() - t1
        inflight2 = {"value": args.steps * T / dt2, "unit": "frames/s", "ms_per_step": dt2 / args.steps * 1e3,
                     "note": "same K forwards, two clips in flight on two HIP streams (the video scheduler's mode); not `value`"}

    if rank == 0:
        # ---- dominant kernel: per-launch durations from the events recorded in the timed region
        # (1 launch in 8 of each large shape is bracketed; totals = sampled rate x all launches' algorithmic flops)
        agg = {k: [v["timed"], v["timed_ms"] * 1e-3, v["timed_flops"]] for k, v in prof.items() if v["timed"] > 0}
        launches = {k: (v["launches"], v["flops"]) for k, v in prof.items()}
        est = {k: launches[k][1] / (v[2] / v[1]) for k, v in agg.items()}      # estimated seconds in the timed region
        dom = max(est, key=est.get)
        sampled, secs, flops = agg[dom]
        calls = launches[dom][0]
        # HBM bytes per launch of that kernel: cannot be read live (needs rocprofv3 --pmc passes); taken from the
        # committed PMC summary of the same command (tools/pmc_bench.sh -> profiles/), null when absent.
        traffic = None
        for rnd in ("r04", "r03", "r02", "r01"):
            pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", rnd, f"{args.encoder}_pmc_hbm_traffic.json")
            if os.path.exists(pmc) and not args.fp32:
                traffic = json.load(open(pmc))["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    break
        achieved = flops / secs / 1e12
        fps = world * args.steps * T / dt
        peak = 157.3 if args.fp32 else MFMA_PEAK_TFLOPS          # fp32-input MFMA runs at 1/16 of the fp16 rate (MI355X_MICROARCH.md)
        prec = "fp32" if args.fp32 else "fp16"
        line = {
            "metric": f"frames/sec at 1x32x518x518 {prec}, ViT-L" if args.encoder == "vitl" else f"frames/sec at 1x32x518x518 {prec}, ViT-S",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.fp32 else "f16", "data": "synthetic",
            "config": {"workload": f"{args.encoder} {prec} 32-frame clip forward, x=randn(1,32,3,518,518), seeded random weights "
                                   f"(BASELINE.json configs[{2 if args.encoder == 'vitl' else 1}])",
                       "clips_per_step_per_gpu": 1, "parallelism": f"independent windows x{world}" + (", per-step all-gather of depth overlapped with compute" if world > 1 else ""),
                       "world": (dist.get_world_size() if dist is not None else 1), "backend": (dist.get_backend() if dist is not None else None)},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": traffic, "launches": calls,
                         "launches_timed": sampled, "avg_launch_us": secs / sampled * 1e6,
                         "algorithmic_gflop_per_launch": flops / sampled / 1e9, "share_of_step_time": est[dom] / dt},
            "model_tflops": CLIP_TFLOP[args.encoder] * world * args.steps / dt,
            "model_mfma_frac": CLIP_TFLOP[args.encoder] * world * args.steps / dt / (peak * world),
            "kernels": {k: {"launches": launches[k][0], "launches_timed": v[0], "ms_per_step": est[k] / args.steps * 1e3,
                            "tflops": v[2] / v[1] / 1e12} for k, v in agg.items()},
        }
        if inflight2 is not None:
            line["two_clips_in_flight"] = inflight2
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.encoder, full=not args.cpu_sample)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
