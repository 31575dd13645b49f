#!/usr/bin/env python
"""Headline benchmark: frames/s of VideoDepthAnything.forward on synthetic 1x32x518x518, ViT-L, fp16
operands (BASELINE.json metric / configs[2]), on N MI355X of one node.

  python bench.py [--gpus N --steps K --warmup W] [--encoder vitl|vits]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
              --master-port P bench.py --gpus N --steps K --warmup W

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


def cpu_baseline(encoder, frames=2):
    """Oracle on the host cores: same weights/shape per frame, `frames` of the clip's 32 frames."""
    from oracle import vda_oracle as O
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.weights import synthetic_state_dict
    cfg = get_config(encoder)
    sd = synthetic_state_dict(cfg, seed=0)
    x = torch.randn(1, frames, 3, 518, 518, generator=torch.Generator().manual_seed(0))
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    with torch.no_grad():
        O.forward(sd, cfg, x)
    dt = time.perf_counter() - t0
    return {"value": frames / dt, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{encoder} fp32 torch-CPU oracle, 1x{frames}x518x518 ({frames} of the clip's 32 frames), 1 rep, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--encoder", default="vitl", choices=["vitl", "vits"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    # VDA_BENCH_BACKEND=gloo is a REHEARSAL mode for boxes with fewer GPUs than ranks (ranks share devices, the exchange is
    # staged through the host): it exercises the rank / barrier / reduction logic only - never a measurement.
    backend = os.environ.get("VDA_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)       # NCCL == RCCL over xGMI on ROCm
        else:
            dist.init_process_group(backend)

    from video_depth_anything_amd import ops
    from video_depth_anything_amd.config import get_config
    from video_depth_anything_amd.video_depth import VideoDepthAnything
    from video_depth_anything_amd.weights import synthetic_state_dict

    cfg = get_config(args.encoder)
    model = VideoDepthAnything(encoder=args.encoder, features=cfg.features, out_channels=list(cfg.out_channels))
    model.load_state_dict(synthetic_state_dict(cfg, seed=0), strict=True)
    model = model.to(dev).eval()
    T, H, W = 32, 518, 518
    x = torch.randn(1, T, 3, H, W, generator=torch.Generator().manual_seed(rank)).to(dev)   # resident in HBM

    for _ in range(args.warmup):
        model(x)
    outs = torch.empty(args.steps, T, H, W, dtype=torch.float32, device=dev)
    gathered = torch.empty(args.steps, world, T, H, W, dtype=torch.float32, device=dev) if world > 1 else None

    def exchange(s):
        """The one exchange of the path: this step's depth maps to every rank (what the stitcher needs). Issued per step and
        asynchronously - RCCL runs it on its own stream behind the producing kernels, under the next step's compute."""
        if backend == "nccl":
            return dist.all_gather_into_tensor(gathered[s], outs[s], async_op=True)
        parts = [torch.empty(T, H, W, dtype=torch.float32) for _ in range(world)]
        dist.all_gather(parts, outs[s].cpu())
        gathered[s].copy_(torch.stack(parts))
        return None
    ops.PROFILE = ops.GemmProfile(every=4)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pending = []
    for s in range(args.steps):
        outs[s].copy_(model(x)[0])
        if dist is not None:
            pending.append(exchange(s))
    if dist is not None:
        for h in pending:
            if h is not None:
                h.wait()
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        # ---- dominant kernel: per-launch durations from the events recorded in the timed region
        # (1 launch in 4 of each shape is bracketed; totals = sampled rate x all launches' algorithmic flops)
        agg = {}
        for name, flops, e0, e1 in prof.samples:
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += 1
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += flops
        est = {k: prof.launches[k][1] / (v[2] / v[1]) for k, v in agg.items()}      # estimated seconds in the timed region
        dom = max(est, key=est.get)
        sampled, secs, flops = agg[dom]
        calls = prof.launches[dom][0]
        # HBM bytes per launch of that kernel: cannot be read live (needs rocprofv3 --pmc passes); taken from the
        # committed PMC summary of the same command (tools/pmc_bench.sh -> profiles/), null when absent.
        traffic = None
        pmc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", f"{args.encoder}_pmc_hbm_traffic.json")
        if os.path.exists(pmc):
            traffic = json.load(open(pmc))["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
        achieved = flops / secs / 1e12
        fps = world * args.steps * T / dt
        line = {
            "metric": "frames/sec at 1x32x518x518 fp16, ViT-L" if args.encoder == "vitl" else "frames/sec at 1x32x518x518 fp16, ViT-S",
            "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"{args.encoder} fp16 32-frame clip forward, x=randn(1,32,3,518,518), seeded random weights "
                                   f"(BASELINE.json configs[{2 if args.encoder == 'vitl' else 1}])",
                       "clips_per_step_per_gpu": 1, "parallelism": f"independent windows x{world}" + (", per-step all-gather of depth overlapped with compute" if world > 1 else "")},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": achieved, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / MFMA_PEAK_TFLOPS, "traffic": traffic, "launches": calls,
                         "launches_timed": sampled, "avg_launch_us": secs / sampled * 1e6,
                         "algorithmic_gflop_per_launch": flops / sampled / 1e9, "share_of_step_time": est[dom] / dt},
            "model_tflops": CLIP_TFLOP[args.encoder] * world * args.steps / dt,
            "model_mfma_frac": CLIP_TFLOP[args.encoder] * world * args.steps / dt / (MFMA_PEAK_TFLOPS * world),
            "kernels": {k: {"launches": prof.launches[k][0], "launches_timed": v[0], "ms_per_step": est[k] / args.steps * 1e3,
                            "tflops": v[2] / v[1] / 1e12} for k, v in agg.items()},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.encoder)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
