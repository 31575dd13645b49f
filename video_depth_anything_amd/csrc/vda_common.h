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
// gelu(x) = 0.5 x (1 + erf(x/sqrt2)) = max(x, 0) - |x| * 0.5 erfc(|x|/sqrt2), and 0.5 erfc(|x|/sqrt2) = P(|x|)^-16 with a
// degree-6 polynomial P: the form of Abramowitz-Stegun 7.1.28, coefficients refitted here for the weight |x| (minimax over
// |x| in [0, 12], tools/gelu_fit.py: 2.2e-7 absolute in exact arithmetic, 6.0e-7 evaluated in fp32; relative error where
// |gelu| > 1e-3: 2.2e-4, half an fp16 ulp of the stored result). ONE transcendental (v_rcp_f32) per element instead of the two
// (rcp + exp) of the A-S 7.1.26 form this replaces: each costs as much as ~6 plain VALU issues in the fc1 epilogue, which runs
// this 128 times per lane and tile (measured there: 64 us of a 420 us launch, 15 us of them per transcendental).
#define VDA_GELU_C0 1.04427485e+00f
#define VDA_GELU_C1 5.20639309e-02f
#define VDA_GELU_C2 2.21137954e-02f
#define VDA_GELU_C3 3.37046981e-03f
#define VDA_GELU_C4 7.53152628e-05f
#define VDA_GELU_C5 3.96599537e-05f
#define VDA_GELU_C6 6.98591262e-06f
__device__ __forceinline__ float gelu_erf(float x) {
    const float ax = fabsf(x);
    // explicit fma everywhere: every instantiation (scalar here, packed below, guarded / branch-free epilogues) must round
    // identically - a row's result may not depend on which tile or kernel it falls in
    float p = fmaf(VDA_GELU_C6, ax, VDA_GELU_C5);
    p = fmaf(p, ax, VDA_GELU_C4);
    p = fmaf(p, ax, VDA_GELU_C3);
    p = fmaf(p, ax, VDA_GELU_C2);
    p = fmaf(p, ax, VDA_GELU_C1);
    p = fmaf(p, ax, VDA_GELU_C0);
    float r = __builtin_amdgcn_rcpf(p);
    r *= r;
    r *= r;
    r *= r;
    r *= r;
    return fmaf(-ax, r, fmaxf(x, 0.f));
}

// The same arithmetic on two elements per instruction (v_pk_fma_f32 / v_pk_mul_f32, twice the scalar rate while the matrix pipe
// is idle, as it is in an epilogue). Every operation is the IEEE operation of the scalar form in the same order: bit-identical.
typedef float vda_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ vda_f32x2 gelu_erf2(vda_f32x2 x) {
    const vda_f32x2 ax = {fabsf(x[0]), fabsf(x[1])};
#define VDA_P2(c) vda_f32x2{c, c}
    vda_f32x2 p = __builtin_elementwise_fma(VDA_P2(VDA_GELU_C6), ax, VDA_P2(VDA_GELU_C5));
    p = __builtin_elementwise_fma(p, ax, VDA_P2(VDA_GELU_C4));
    p = __builtin_elementwise_fma(p, ax, VDA_P2(VDA_GELU_C3));
    p = __builtin_elementwise_fma(p, ax, VDA_P2(VDA_GELU_C2));
    p = __builtin_elementwise_fma(p, ax, VDA_P2(VDA_GELU_C1));
    p = __builtin_elementwise_fma(p, ax, VDA_P2(VDA_GELU_C0));
#undef VDA_P2
    vda_f32x2 r = {__builtin_amdgcn_rcpf(p[0]), __builtin_amdgcn_rcpf(p[1])};
    r *= r;
    r *= r;
    r *= r;
    r *= r;
    const vda_f32x2 relu = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(-ax, r, relu);
}
template <int N>
__device__ __forceinline__ void gelu_erf_n(float (&v)[N]) {
    static_assert(N % 2 == 0, "pairs");
    // The N / 2 pairs advance in LOCKSTEP, one polynomial step at a time: a packed-fp32 result cannot feed the next packed operation
    // back to back (hipcc pads each dependent pair of v_pk_* with an s_nop), and written pair after pair the whole chain of a pair was
    // emitted serially: 34 s_nop among 127 instructions per 8 outputs in the fc1 epilogue's ISA. In lockstep a pair's next step is
    // N / 2 instructions away. Same IEEE operations in the same order per element: bit-identical to gelu_erf / gelu_erf2.
    constexpr int P = N / 2;
    vda_f32x2 ax[P], p[P], r[P];
#define VDA_P2(c) vda_f32x2{c, c}
#pragma unroll
    for (int k = 0; k < P; ++k) ax[k] = vda_f32x2{fabsf(v[2 * k]), fabsf(v[2 * k + 1])};
#pragma unroll
    for (int k = 0; k < P; ++k) p[k] = __builtin_elementwise_fma(VDA_P2(VDA_GELU_C6), ax[k], VDA_P2(VDA_GELU_C5));
#pragma unroll
    for (int k = 0; k < P; ++k) p[k] = __builtin_elementwise_fma(p[k], ax[k], VDA_P2(VDA_GELU_C4));
#pragma unroll
    for (int k = 0; k < P; ++k) p[k] = __builtin_elementwise_fma(p[k], ax[k], VDA_P2(VDA_GELU_C3));
#pragma unroll
    for (int k = 0; k < P; ++k) p[k] = __builtin_elementwise_fma(p[k], ax[k], VDA_P2(VDA_GELU_C2));
#pragma unroll
    for (int k = 0; k < P; ++k) p[k] = __builtin_elementwise_fma(p[k], ax[k], VDA_P2(VDA_GELU_C1));
#pragma unroll
    for (int k = 0; k < P; ++k) p[k] = __builtin_elementwise_fma(p[k], ax[k], VDA_P2(VDA_GELU_C0));
#undef VDA_P2
#pragma unroll
    for (int k = 0; k < P; ++k) r[k] = vda_f32x2{__builtin_amdgcn_rcpf(p[k][0]), __builtin_amdgcn_rcpf(p[k][1])};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int k = 0; k < P; ++k) r[k] *= r[k];
    }
#pragma unroll
    for (int k = 0; k < P; ++k) {
        const vda_f32x2 relu = {fmaxf(v[2 * k], 0.f), fmaxf(v[2 * k + 1], 0.f)};
        const vda_f32x2 o = __builtin_elementwise_fma(-ax[k], r[k], relu);
        v[2 * k] = o[0];
        v[2 * k + 1] = o[1];
    }
}

// 16-byte async global -> LDS copy. The LDS destination is the wave-uniform `lds_base`
// plus lane*16 (hardware rule); the global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_base) {
    __builtin_amdgcn_global_load_lds((const VDA_GLOBAL_AS void*)gsrc, (VDA_LDS_AS void*)lds_base, 16, 0, 0);
}

// LDS-DMA as opaque assembly: with the builtin (glds16) hipcc's waitcnt pass guards later ds_reads of the wave with vmcnt waits - it
// cannot tell the DMA's destination from the LDS buffers being read - which puts the DMA's whole latency in front of whatever reads
// LDS next (found in round 4: the depth tail's interpolation, and EVERY k-step of conv3x3_up2_kernel, whose fragment reads waited for
// the weights of the NEXT step to land). Callers wait for their DMA themselves (s_waitcnt vmcnt) before the barrier that publishes it.
__device__ __forceinline__ void glds16_opaque(const void* gsrc, void* lds_base) {
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds_base);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(gsrc) : "memory", "m0");
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
// Cap on the workgroups a persistent kernel launches (vda_set_max_wgs; 0 = one per CU). DIAGNOSTIC: two launch sequences that each
// take HALF the chip were measured against two full-grid sequences interleaving freely (tools/two_stream.py, round 4): the capped
// form loses (ViT-L 2 x 16 frames: 619 against 630 frames/s; two clips: 636 against 652), so nothing in the product sets it.
extern int g_vda_max_wgs;
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
    const int cap = g_vda_max_wgs;
    return (cap >= 8 && cap < st.num_cu[slot]) ? (cap & ~7) : st.num_cu[slot];
}
