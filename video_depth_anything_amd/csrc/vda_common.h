// Shared device helpers for the gfx950 kernels. Wave = 64 lanes throughout.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../include/vda.h"

typedef _Float16 h16;
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VDA_GLOBAL_AS __attribute__((address_space(1)))
#define VDA_LDS_AS __attribute__((address_space(3)))

extern "C" void vda_set_error(const char* fmt, ...);

#define VDA_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            vda_set_error(__VA_ARGS__);        \
            return 1;                          \
        }                                      \
    } while (0)

#define VDA_LAUNCH_CHECK()                                             \
    do {                                                               \
        hipError_t e_ = hipGetLastError();                             \
        if (e_ != hipSuccess) {                                        \
            vda_set_error("launch failed: %s", hipGetErrorString(e_)); \
            return 2;                                                  \
        }                                                              \
    } while (0)

// Exact-form GELU 0.5 x (1 + erf(x / sqrt 2)) (nn.GELU() default, dinov2.py:61; F.gelu, motion_module/attention.py:378).
// erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far below the fp16 output rounding): one v_rcp, one v_exp and
// six FMAs instead of libm erff's ~30 instructions - the GELU epilogue runs 128x per lane per GEMM tile.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    p *= t;
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float r = fmaf(-p, e, 1.0f);
    return copysignf(r, x);
}
// gelu(x) = 0.5 x (1 + erf(x/sqrt2)). With erf(|z|) = 1 - p(t) e^{-z^2} (same A-S 7.1.26 polynomial, 0.5 folded into its
// coefficients, z = |x|/sqrt2 folded into t's slope and the exponent's scale):  gelu(x) = max(x, 0) - |x| * 0.5 p(t) * e^{-x^2/2}.
// No copysign / 1+erf / 0.5x: 13 VALU issues per element instead of 17 (this runs 128x per lane per fc1 tile: ~20 % of that GEMM).
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float p = fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    p = fmaf(p, t, 0.5f * 1.421413741f);
    p = fmaf(p, t, 0.5f * -0.284496736f);
    p = fmaf(p, t, 0.5f * 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(x * x * (-0.5f * 1.4426950408889634f));
    // explicit fma: the guarded and the branch-free epilogue instantiations must round identically (a row's result may not
    // depend on which tile it falls in), so nothing is left to -ffp-contract's per-instance choice
    return fmaf(-(ax * (p * t)), e, fmaxf(x, 0.f));
}

// 16-byte async global -> LDS copy. The LDS destination is the wave-uniform `lds_base`
// plus lane*16 (hardware rule); the global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const VDA_GLOBAL_AS void*)gsrc, (VDA_LDS_AS void*)lds_base, 16, 0, 0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// 8 consecutive activations as floats, for kernels instantiated for both activation types (h16: the fp16-operand path,
// float: the fp32-operand path). 16-byte accesses for h16, two for float.
__device__ __forceinline__ void load8(const h16* p, float (&v)[8]) {
    const h16x8 x = *reinterpret_cast<const h16x8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)x[e];
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        v[e] = a[e];
        v[4 + e] = b[e];
    }
}
__device__ __forceinline__ void store8(h16* p, const float (&v)[8]) {
    h16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (h16)v[e];
    *reinterpret_cast<h16x8*>(p) = o;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<f32x4*>(p) = f32x4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is PER DEVICE: a process that drives several GPUs (model.to('cuda:1'))
// must set it on each one. `done` is the caller's static per-kernel bit mask of devices already configured; returns the
// number of CUs of the current device rounded down to a multiple of 8 (persistent grids), or < 0 on error.
struct VdaKernelDeviceState {
    unsigned long long done = 0;
    int num_cu[64] = {};
};
inline int vda_prepare_kernel(const void* fn, int dyn_lds_bytes, VdaKernelDeviceState& st) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const int slot = dev & 63;
    if (!((st.done >> slot) & 1ull)) {
        if (dyn_lds_bytes > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn_lds_bytes);
            if (e != hipSuccess) {
                vda_set_error("hipFuncSetAttribute: %s", hipGetErrorString(e));
                return -1;
            }
        }
        int cu = 0;
        (void)hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (cu < 8) cu = 8;
        st.num_cu[slot] = cu & ~7;
        st.done |= 1ull << slot;
    }
    return st.num_cu[slot];
}
