// 8-phase two-group schedule on 192 x 256 tiles, dense A: the remainder launch of a row-split GEMM (vda_gemm_plan_split).
#include "gemm8p_kernel.h"

int vda_gemm8p_dense_bn256_bm192(const vda_gemm_args& a, hipStream_t s) { return vda_gemm8p::launch_dense_bm192<256>(a, s); }
