#!/usr/bin/env python
"""In-kernel clock of the encoder GEMMs (MI355X_MICROARCH.md "DVFS give-back" item 6) and the L2-blocked tile order A/B (VERDICT r3
next #1d). One process, one device, random data. Per shape and tile order: wall time per launch (HIP events, median of interleaved
repeats, no stamps), then - after >= 2 s of back-to-back launches - one stamped launch: clock = d(s_memtime) / d(s_memrealtime) x
100 MHz over the whole kernel and over the K loops alone (median over workgroups)."""
import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops
lib = _lib.lib
g = torch.Generator(device="cuda").manual_seed(0)
M = 43840
cases = [("fc1 + LN + GELU", 4096, 1024, _lib.EPI_LN_GELU_F16), ("qkv + LN", 3072, 1024, _lib.EPI_LN_BIAS_F16), ("plain bias N=1024 K=4096", 1024, 4096, _lib.EPI_BIAS_F16),
         ("plain bias N=4096 K=1024", 4096, 1024, _lib.EPI_BIAS_F16)]
out_json = {}
for name, N, K, epi in cases:
    A = torch.randn(M, K, device="cuda", generator=g).half()
    W = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g)
    out = torch.zeros(M, N, dtype=torch.float16, device="cuda")
    st = torch.zeros(256 * 16 * 4 * 2, dtype=torch.int64, device="cuda")
    kw = dict(M=M, N=N, K=K, bias=bias, pos=st.view(torch.float32))
    if epi in (_lib.EPI_LN_BIAS_F16, _lib.EPI_LN_GELU_F16):
        kw.update(gamma=torch.ones(N, device="cuda"), stats=torch.stack([torch.zeros(M), torch.ones(M)], 1).contiguous().cuda())
    modes = [("default", 0), ("blocked", 2)]
    ts = {m: [] for m, _ in modes}
    for rep in range(7):
        for m, bits in modes:
            lib.vda_gemm_set_debug(bits)
            ops.gemm(A, W, out, epi, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(A, W, out, epi, **kw)
            e1.record(); torch.cuda.synchronize()
            ts[m].append(e0.elapsed_time(e1) / 10 * 1e3)
    res = {}
    for m, bits in modes:
        lib.vda_gemm_set_debug(bits)
        t0 = time.time()
        while time.time() - t0 < 2.5:                      # >= 2 s back to back: the clock the chip HOLDS under this load
            for _ in range(50):
                ops.gemm(A, W, out, epi, **kw)
            torch.cuda.synchronize()
        lib.vda_gemm_set_debug(bits | 1)
        for _ in range(20):
            ops.gemm(A, W, out, epi, **kw)
        torch.cuda.synchronize()
        s = st.cpu().numpy().reshape(256, 16, 4, 2).astype(np.float64)
        used = s[:, :, 0, 1] > 0
        nt = used.sum(axis=1).astype(int)
        whole, loops = [], []
        for b in range(256):
            if nt[b] == 0:
                continue
            last = nt[b] - 1
            dt = s[b, last, 3, 1] - s[b, 0, 0, 1]
            if dt > 0:
                whole.append((s[b, last, 3, 0] - s[b, 0, 0, 0]) / dt * 100.0)
            lt = (s[b, :nt[b], 2, 1] - s[b, :nt[b], 1, 1]).sum()
            if lt > 0:
                loops.append((s[b, :nt[b], 2, 0] - s[b, :nt[b], 1, 0]).sum() / lt * 100.0)
        kcyc = np.mean([(s[b, :nt[b], 2, 0] - s[b, :nt[b], 1, 0]).mean() for b in range(256) if nt[b] > 0])
        ecyc = np.mean([(s[b, :nt[b], 3, 0] - s[b, :nt[b], 2, 0]).mean() for b in range(256) if nt[b] > 0])
        us = sorted(ts[m])[len(ts[m]) // 2]
        res[m] = {"us_per_launch": us, "tflops": 2.0 * M * N * K / us / 1e6, "clock_mhz_kernel": float(np.median(whole)), "clock_mhz_k_loops": float(np.median(loops)),
                  "k_loop_cycles_per_tile": float(kcyc), "k_loop_cycles_per_k_tile": float(kcyc / (K // 64)), "epilogue_cycles_per_tile": float(ecyc),
                  "tiles_per_wg_max": int(nt.max())}
        print(f"{name:28s} {m:8s}: {us:7.1f} us  {res[m]['tflops']:6.0f} TF/s  clock {res[m]['clock_mhz_kernel']:6.0f} MHz (K loops {res[m]['clock_mhz_k_loops']:6.0f})  "
              f"K loop {kcyc / 1e3:6.2f} kcyc/tile ({kcyc / (K // 64):5.0f}/K-tile)  epilogue {ecyc / 1e3:5.2f} kcyc/tile", flush=True)
    out_json[name] = res
lib.vda_gemm_set_debug(0)
if len(sys.argv) > 1:
    json.dump(out_json, open(sys.argv[1], "w"), indent=1)
