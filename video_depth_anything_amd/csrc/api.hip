// Error reporting and version entry points of the C ABI (include/vda.h).
#include "vda_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

extern "C" void vda_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* vda_last_error(void) { return g_err; }
extern "C" int vda_abi_version(void) { return 8; }

int g_vda_max_wgs = 0;
extern "C" int vda_set_max_wgs(int n) {
    g_vda_max_wgs = n > 0 ? n : 0;
    return 0;
}

// Test utility (tools/contention.py): occupy `wgs` compute units' worth of workgroups (256 threads, `lds_bytes` of LDS each) for
// about `cycles` shader clocks - a stand-in for a communication kernel (RCCL's channel workgroups) running beside the forward.
// Every wave leaves after `cycles` clocks at the latest: the loop is bounded by the clock and by an iteration count.
__global__ void __launch_bounds__(256) occupy_kernel(long long cycles, int* sink) {
    extern __shared__ int hog_lds[];
    const long long t0 = __builtin_readcyclecounter();
    int acc = 0;
    for (long long it = 0; it < (cycles >> 6) + 1; ++it) {
        __builtin_amdgcn_s_sleep(1);
        acc += (int)it;
        if ((long long)__builtin_readcyclecounter() - t0 > cycles) break;
    }
    if (threadIdx.x == 0) hog_lds[0] = acc;
    if (acc == -12345 && sink) *sink = hog_lds[0];
}

extern "C" int vda_debug_occupy(int wgs, int lds_bytes, long long cycles, vda_stream_t stream) {
    VDA_REQUIRE(wgs > 0 && wgs <= 1024 && lds_bytes >= 0 && lds_bytes <= 48 * 1024 && cycles > 0 && cycles < (1ll << 32), "vda_debug_occupy: bad arguments");
    hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream, cycles, (int*)nullptr);
    VDA_LAUNCH_CHECK();
    return 0;
}
