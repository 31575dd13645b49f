"""output_conv1 over refinenet1's 2x upsample: the fused kernel (vda_conv3x3_up2_f16) against the unfused pair
(vda_bilinear_nhwc_f16 -> vda_gemm_f16 conv3x3) at the benchmarked shapes, one process, interleaved.
usage: python tools/conv_up_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_depth_anything_amd import _lib, ops  # noqa: E402

F16 = torch.float16


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    for name, B, h, C, N in (("vitl", 32, 148, 256, 128), ("vits", 32, 148, 64, 32), ("vitb", 32, 148, 128, 64)):
        g = torch.Generator().manual_seed(1)
        x = torch.randn(B, h, h, C, generator=g).to(F16).cuda()
        w = ops.pack_conv3x3(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5).cuda()
        b = torch.randn(N, generator=g).cuda()
        H = 2 * h
        out = torch.empty(B, H, H, N, dtype=F16, device="cuda")
        up = torch.empty(B, H, H, C, dtype=F16, device="cuda")
        two = torch.empty(B, H, H, N, dtype=F16, device="cuda")
        fused = lambda: ops.conv3x3_up2(x, w, b, out, B, h, h, C, N, N)
        bil = lambda: ops.bilinear_nhwc(x, up, B, h, h, H, H, C)
        conv = lambda: ops.gemm(up, w, two, _lib.EPI_BIAS_F16, M=B * H * H, N=N, K=9 * C, bias=b, conv=(B, H, H, C, H, H, 1))
        res = {"fused": [], "bilinear": [], "conv": []}
        for _ in range(3):
            res["fused"].append(timed(fused, reps))
            res["bilinear"].append(timed(bil, reps))
            res["conv"].append(timed(conv, reps))
        med = {k: sorted(v)[1] for k, v in res.items()}
        flop = 2.0 * B * H * H * N * 9 * C
        d = (out.float() - two.float()).abs().max().item()
        print(f"{name}: fused {med['fused']:.0f} us ({flop / med['fused'] / 1e6:.0f} TFLOP/s)  unfused {med['bilinear']:.0f} + {med['conv']:.0f} = "
              f"{med['bilinear'] + med['conv']:.0f} us ({flop / med['conv'] / 1e6:.0f} TFLOP/s the conv alone)  max |diff| {d:.2e}", flush=True)


if __name__ == "__main__":
    main()
