// 192 x 128 tiles on six waves (gemm256s_kernel.h): for problems whose 256-row tile count sits just above a multiple of the CU count.
#include "gemm256s_kernel.h"

int vda_gemm256s_dense_bn128_bm192(const vda_gemm_args& a, hipStream_t s) { return vda_gemm256s::launch_dense_bm192<128>(a, s); }
int vda_gemm256s_dense_bn128_bm192_x2(const vda_gemm_args& a, hipStream_t s) { return vda_gemm256s::launch_dense_bm192_x2<128>(a, s); }
