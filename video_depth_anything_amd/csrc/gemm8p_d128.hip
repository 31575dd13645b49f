// 8-phase two-group schedule of the 256 x 128 tile, dense A (tuning variant 9).
#include "gemm8p_kernel.h"

int vda_gemm8p_dense_bn128(const vda_gemm_args& a, hipStream_t s) { return vda_gemm8p::launch_dense<128>(a, s); }
