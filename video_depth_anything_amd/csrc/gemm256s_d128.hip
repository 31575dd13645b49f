// 16x16x32-MFMA variant of the large-tile GEMM: one (tile width, A-operand) family per translation unit.
#include "gemm256s_kernel.h"

int vda_gemm256s_dense_bn128(const vda_gemm_args& a, hipStream_t s) { return vda_gemm256s::launch_dense<128>(a, s); }
