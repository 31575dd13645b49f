// A C++ host of the handle API (include/vda.h) with no Python and no torch in the process: what a maintainer of a C / C++
// caller writes in place of the reference's
//     model = VideoDepthAnything(**cfg); model.load_state_dict(sd, strict=True); depth = model.forward(x)
// (video_depth.py:38-63,89-93; run.py:45-47).
//
//   host_demo <model.bin> <input.bin> <output.bin> <precision: 0 = fp16 operands, 1 = fp32 operands>
//
// model.bin : vda_config (16 x int32), int32 n, then n x { int32 name_len, name, int32 ndim, int64 dims[ndim], float data[] }
// input.bin : int32 B, T, H, W, then float x[B,T,3,H,W] (normalised frames)
// output.bin: float depth[B,T,H,W]
// tests/test_host_demo_gpu.py writes the inputs from Python, runs this program and requires its output to be bit-identical to
// VideoDepthAnything.forward on the same device.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "vda.h"

static void die(const char* what, const char* detail) {
    fprintf(stderr, "host_demo: %s: %s\n", what, detail);
    exit(1);
}
#define HIP_OK(expr)                                            \
    do {                                                        \
        hipError_t e_ = (expr);                                 \
        if (e_ != hipSuccess) die(#expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
static void rd(FILE* f, T* p, size_t n, const char* what) {
    if (fread(p, sizeof(T), n, f) != n) die("short read", what);
}

int main(int argc, char** argv) {
    if (argc != 5) die("usage", "host_demo model.bin input.bin output.bin precision");
    const int precision = atoi(argv[4]);
    HIP_OK(hipSetDevice(0));

    FILE* f = fopen(argv[1], "rb");
    if (!f) die("cannot open", argv[1]);
    vda_config cfg;
    rd(f, &cfg, 1, "config");
    vda_model* m = nullptr;
    if (vda_create(&cfg, &m)) die("vda_create", vda_last_error());
    int32_t n = 0;
    rd(f, &n, 1, "tensor count");
    if (n != vda_num_weights(m)) die("state dict", "tensor count differs from vda_num_weights()");
    std::vector<float> data;
    for (int i = 0; i < n; ++i) {
        int32_t len = 0, ndim = 0;
        rd(f, &len, 1, "name length");
        std::string name(len, '\0');
        rd(f, &name[0], len, "name");
        rd(f, &ndim, 1, "ndim");
        std::vector<int64_t> dims(ndim);
        rd(f, dims.data(), ndim, "dims");
        size_t count = 1;
        for (int64_t d : dims) count *= (size_t)d;
        data.resize(count);
        rd(f, data.data(), count, name.c_str());
        if (vda_load_weight(m, name.c_str(), data.data(), dims.data(), ndim, VDA_DTYPE_F32)) die("vda_load_weight", vda_last_error());
    }
    fclose(f);
    if (vda_finalize_weights(m)) die("vda_finalize_weights", vda_last_error());

    f = fopen(argv[2], "rb");
    if (!f) die("cannot open", argv[2]);
    int32_t shape[4];
    rd(f, shape, 4, "input shape");
    const int B = shape[0], T = shape[1], H = shape[2], W = shape[3];
    std::vector<float> x((size_t)B * T * 3 * H * W), depth((size_t)B * T * H * W);
    rd(f, x.data(), x.size(), "input");
    fclose(f);

    const int64_t ws = vda_workspace_bytes(m, B, T, H, W, precision);
    if (ws < 0) die("vda_workspace_bytes", vda_last_error());
    float *dx = nullptr, *dd = nullptr;
    HIP_OK(hipMalloc(&dx, x.size() * sizeof(float)));
    HIP_OK(hipMalloc(&dd, depth.size() * sizeof(float)));
    HIP_OK(hipMemcpy(dx, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t s;
    HIP_OK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep)                      // twice: the second forward allocates nothing
        if (vda_forward(m, dx, dd, B, T, H, W, precision, s)) die("vda_forward", vda_last_error());
    HIP_OK(hipStreamSynchronize(s));
    if (vda_forward_status(m)) die("vda_forward_status", vda_last_error());      // deferred status of the forwards above (fp16 stream overflow)
    HIP_OK(hipMemcpy(depth.data(), dd, depth.size() * sizeof(float), hipMemcpyDeviceToHost));

    f = fopen(argv[3], "wb");
    if (!f || fwrite(depth.data(), sizeof(float), depth.size(), f) != depth.size()) die("cannot write", argv[3]);
    fclose(f);
    printf("host_demo: %dx%dx%dx%d, precision %d, workspace %.1f MB, depth[0] = %g\n", B, T, H, W, precision, ws / 1e6, depth[0]);
    HIP_OK(hipFree(dx));
    HIP_OK(hipFree(dd));
    vda_destroy(m);
    return 0;
}
