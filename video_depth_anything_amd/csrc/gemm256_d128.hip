// One (tile width, A-operand) family of the large-tile GEMM per translation unit (parallel build).
#include "gemm256_kernel.h"

int vda_gemm256_dense_bn128(const vda_gemm_args& a, hipStream_t s) { return vda_gemm256::launch_dense<128>(a, s); }
