// 8-phase two-group schedule of the 256 x 128 tile, conv A (tuning variant 9).
#include "gemm8p_kernel.h"

int vda_gemm8p_conv_bn128(const vda_gemm_args& a, hipStream_t s) { return vda_gemm8p::launch_conv<128>(a, s); }
