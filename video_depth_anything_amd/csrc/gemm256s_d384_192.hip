// 192 x 384 tiles on twelve waves (gemm256s_kernel.h): ViT-S's embedding width in one tile.
#include "gemm256s_kernel.h"

int vda_gemm256s_dense_bn384_bm192(const vda_gemm_args& a, hipStream_t s) { return vda_gemm256s::launch_dense_bn384<384>(a, s); }
