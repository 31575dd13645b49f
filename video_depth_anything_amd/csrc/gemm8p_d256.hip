// 8-phase two-group schedule of the 256 x 256 tile, dense A (tuning variant 5).
#include "gemm8p_kernel.h"

int vda_gemm8p_dense_bn256(const vda_gemm_args& a, hipStream_t s) { return vda_gemm8p::launch_dense<256>(a, s); }

// A/B schedules of the K loop (tools/gemm_ab.py: variant 5 + 32 * sched), plain-bias / fp32-residual epilogues only:
// sched 1 = burstier 2/2/0/4 DMA issue with vmcnt(8) (SCHED 0), sched 2 = the same without s_setprio (SCHED 2)
int vda_gemm8p_dense_bn256_sched(const vda_gemm_args& a, hipStream_t s, int sched) {
    using namespace vda_gemm8p;
    if (sched == 1) {
        if (a.epilogue == VDA_EPI_BIAS_F16) return launch256<256, VDA_A_DENSE, VDA_EPI_BIAS_F16, 0>(a, s);
        if (a.epilogue == VDA_EPI_SCALE_RES_F32) return launch256<256, VDA_A_DENSE, VDA_EPI_SCALE_RES_F32, 0>(a, s);
    }
    if (sched == 2 && a.epilogue == VDA_EPI_BIAS_F16) return launch256<256, VDA_A_DENSE, VDA_EPI_BIAS_F16, 2>(a, s);
    return launch_dense<256>(a, s);
}
