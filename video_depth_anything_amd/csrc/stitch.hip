// Window stitcher of infer_video_depth on the device (video_depth.py:216-254, utils/util.py:40-74).
//
// Per window k > 0 the reference (numpy, host) does: least-squares scale/shift of the window's first two frames
// against the two reference key frames, affine + clamp of frames 2..31, an 8-frame linear cross-fade into the
// previous window's tail, and appends the rest. Here that is one deterministic reduction (fp64 partials, fixed
// combination order, no atomics) and one fused element-wise kernel; scale/shift never leave the device, so the
// stitch of window k queues behind its forward on the same stream and costs ~20 us instead of ~15 ms of numpy.
//
// Both kernels are HBM-bound streaming passes (64 MB per 518x518 window).
#include "vda_common.h"

namespace {

constexpr int LSQ_T = 256;

// partial[b] = { sum p*p, sum p, sum p*t, sum t } of block b's grid-stride slice, fp64
__global__ void __launch_bounds__(LSQ_T) lsq_partial_kernel(const float* __restrict__ pred, const float* __restrict__ target, long long n,
                                                            double* __restrict__ partial) {
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (long long i = (long long)blockIdx.x * LSQ_T + threadIdx.x; i < n; i += (long long)gridDim.x * LSQ_T) {
        const double p = (double)pred[i], t = (double)target[i];
        s[0] += p * p;
        s[1] += p;
        s[2] += p * t;
        s[3] += t;
    }
    __shared__ double red[4][LSQ_T];
#pragma unroll
    for (int j = 0; j < 4; ++j) red[j][threadIdx.x] = s[j];
    __syncthreads();
    for (int w = LSQ_T / 2; w > 0; w >>= 1) {                 // fixed tree: the same sum order every run
        if ((int)threadIdx.x < w) {
#pragma unroll
            for (int j = 0; j < 4; ++j) red[j][threadIdx.x] += red[j][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) partial[(size_t)blockIdx.x * 4 + threadIdx.x] = red[threadIdx.x][0];
}

// utils/util.py:40-62 with the all-ones mask: a_11 = n. The reference evaluates the closed form on fp32 sums, where
// det = a00*a11 - a01^2 cancels most of its digits; fp64 here, rounded to fp32 at the end.
__global__ void lsq_finish_kernel(const double* __restrict__ partial, int nblk, double n, float* __restrict__ scale_shift) {
    __shared__ double tot[4];
    if (threadIdx.x < 4) {
        double a = 0.0;
        for (int b = 0; b < nblk; ++b) a += partial[(size_t)b * 4 + threadIdx.x];
        tot[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a00 = tot[0], a01 = tot[1], b0 = tot[2], b1 = tot[3], a11 = n;
        const double det = a00 * a11 - a01 * a01;
        double sc = 1.0, sh = 0.0;
        if (det != 0.0) {
            sc = (a11 * b0 - a01 * b1) / det;
            sh = (-a01 * b0 + a00 * b1) / det;
        }
        scale_shift[0] = (float)sc;
        scale_shift[1] = (float)sh;
    }
}

// numpy evaluates a*b + c on float32 arrays with two roundings. THIS FILE IS BUILT WITH -ffp-contract=off (build.py
// PER_FILE): with -ffp-contract=fast the AMDGPU backend fuses mul+add whatever the source says (HIP's __fmul_rn is a
// plain multiply and `#pragma clang fp contract(off)` does not survive), which changes the last bit of the stitch.

// out = d * scale + shift; out[out < 0] = 0   (NaN stays NaN)
__device__ __forceinline__ float affine_clamp(float d, float sc, float sh) {
    const float m = d * sc;
    const float o = m + sh;
    return o < 0.f ? 0.f : o;
}

// One pixel per thread, every slot of the window:
//   chunk[j]     = tail[j] * (1 - w_j) + aff(win[2 + j]) * w_j      j = 0..7    cross-fade into the previous tail (util.py:65-74)
//   chunk[8 + j] = aff(win[10 + j])                                 j = 0..13   final frames
//   tail[j]      = aff(win[24 + j])                                 j = 0..7    may still be cross-faded by the next window
//   ref1         = aff(win[12])                                     second alignment key frame (video_depth.py:243-248)
__global__ void __launch_bounds__(256) stitch_window_kernel(const float* __restrict__ win, const float* __restrict__ scale_shift,
                                                            float* __restrict__ chunk, float* __restrict__ tail, float* __restrict__ ref1,
                                                            long long px, const float* __restrict__ wts) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= px) return;
    const float sc = scale_shift[0], sh = scale_shift[1];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = affine_clamp(win[(2 + j) * px + i], sc, sh);
        const float a = tail[j * px + i] * wts[j], b = v * wts[8 + j];
        chunk[j * px + i] = a + b;
    }
#pragma unroll
    for (int j = 0; j < 14; ++j) {
        const float v = affine_clamp(win[(10 + j) * px + i], sc, sh);
        chunk[(8 + j) * px + i] = v;
        if (j == 2) ref1[i] = v;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) tail[j * px + i] = affine_clamp(win[(24 + j) * px + i], sc, sh);
}

// out[i] = aff(in[i]): the aligned form of key frames other ranks computed (the key-frame exchange, scheduler.drive_windows_keys);
// the same affine_clamp as stitch_window_kernel, so a frame aligned here is bit-equal to the one its owner's stitch produces
__global__ void __launch_bounds__(256) affine_clamp_kernel(const float* __restrict__ in, const float* __restrict__ scale_shift, float* __restrict__ out,
                                                           long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = affine_clamp(in[i], scale_shift[0], scale_shift[1]);
}

}  // namespace

extern "C" int vda_affine_clamp_f32(const float* in, const float* scale_shift, float* out, long long n, vda_stream_t stream) {
    VDA_REQUIRE(in && scale_shift && out, "vda_affine_clamp_f32: null pointer");
    VDA_REQUIRE(n > 0 && n < (1ll << 31) * 256, "vda_affine_clamp_f32: bad size %lld", n);
    hipLaunchKernelGGL(affine_clamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, scale_shift, out, n);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_lsq_scale_shift_f32(const float* pred, const float* target, long long n, double* workspace, int nblk, float* scale_shift,
                                       vda_stream_t stream) {
    VDA_REQUIRE(pred && target && workspace && scale_shift, "vda_lsq_scale_shift_f32: null pointer");
    VDA_REQUIRE(n > 0 && nblk > 0 && nblk <= 4096, "vda_lsq_scale_shift_f32: bad sizes n=%lld nblk=%d", n, nblk);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lsq_partial_kernel, dim3(nblk), dim3(LSQ_T), 0, s, pred, target, n, workspace);
    hipLaunchKernelGGL(lsq_finish_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, nblk, (double)n, scale_shift);
    VDA_LAUNCH_CHECK();
    return 0;
}

extern "C" int vda_stitch_window_f32(const float* win, const float* scale_shift, float* chunk, float* tail, float* ref1, long long px,
                                     const float* wts, vda_stream_t stream) {
    VDA_REQUIRE(win && scale_shift && chunk && tail && ref1 && wts, "vda_stitch_window_f32: null pointer");
    VDA_REQUIRE(px > 0 && px < (1ll << 31) * 256, "vda_stitch_window_f32: bad frame size %lld", px);
    hipLaunchKernelGGL(stitch_window_kernel, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, (hipStream_t)stream, win, scale_shift, chunk, tail,
                       ref1, px, wts);
    VDA_LAUNCH_CHECK();
    return 0;
}
