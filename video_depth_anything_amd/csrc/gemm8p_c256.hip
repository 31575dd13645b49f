// 8-phase two-group schedule of the 256 x 256 tile, conv A (tuning variant 5).
#include "gemm8p_kernel.h"

int vda_gemm8p_conv_bn256(const vda_gemm_args& a, hipStream_t s) { return vda_gemm8p::launch_conv<256>(a, s); }
