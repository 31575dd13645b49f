"""The GELU the fp16 GEMM epilogues evaluate (csrc/vda_common.h: max(x, 0) - |x| * P(|x|)^-16, coefficients VDA_GELU_C0..6 from
tools/gelu_fit.py) against the exact erf form (nn.GELU() default, dinov2.py:61), evaluated in fp32 in the kernel's operation order.
CPU only: pins the constants in the header."""
import math
import os
import re

import numpy as np
from scipy.special import erf

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "video_depth_anything_amd", "csrc", "vda_common.h")


def test_header_coefficients_give_gelu_to_1e6():
    src = open(HDR).read()
    c = [np.float32(float(re.search(rf"#define VDA_GELU_C{k} ([0-9.e+-]+)f", src).group(1))) for k in range(7)]
    x = np.linspace(-12, 12, 400001).astype(np.float32)
    ax = np.abs(x)
    p = c[6] * ax + c[5]
    for k in (4, 3, 2, 1, 0):
        p = (p * ax + c[k]).astype(np.float32)
    r = (np.float32(1) / p).astype(np.float32)
    for _ in range(4):
        r = (r * r).astype(np.float32)
    y = (np.maximum(x, 0) - ax * r).astype(np.float64)
    ref = 0.5 * x.astype(np.float64) * (1 + erf(x.astype(np.float64) / math.sqrt(2)))
    err = np.abs(y - ref)
    assert err.max() < 1e-6, err.max()                                                   # measured 6.0e-7
    assert (err / np.maximum(np.abs(ref), 1e-3)).max() < 4.9e-4                          # below one fp16 ulp of the stored value (measured 2.2e-4)
