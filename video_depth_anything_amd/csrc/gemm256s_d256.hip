// 16x16x32-MFMA variant of the large-tile GEMM, dense A, BN=256 (tuning variant 3).
#include "gemm256s_kernel.h"

int vda_gemm256s_dense_bn256(const vda_gemm_args& a, hipStream_t s) { return vda_gemm256s::launch_dense<256>(a, s); }
