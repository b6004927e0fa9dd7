#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *src, const int *rows, float *out) {
    __shared__ __attribute__((aligned(16))) float buf[8 * 64 * 4];
    const int lane = threadIdx.x;
    const int g = lane >> 4, q = lane & 15;
    for (int j = 0; j < 8; ++j) {
        const float *p = src + (size_t)rows[8 * g + j] * 64 + q * 4;
        __builtin_amdgcn_global_load_lds(p, buf + j * 256, 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) etc.
    __syncthreads();
    for (int j = 0; j < 8; ++j) {
        f4 v = *reinterpret_cast<f4 *>(buf + j * 256 + lane * 4);
        for (int e = 0; e < 4; ++e) out[(j * 64 + lane) * 4 + e] = v[e];
    }
}
int main() {
    std::vector<float> src(100 * 64); std::vector<int> rows(32); std::vector<float> out(8 * 64 * 4);
    for (size_t i = 0; i < src.size(); ++i) src[i] = (float)i;
    for (int i = 0; i < 32; ++i) rows[i] = (i * 37 + 5) % 100;
    float *ds, *dout; int *dr;
    hipMalloc(&ds, src.size() * 4); hipMalloc(&dr, 128); hipMalloc(&dout, out.size() * 4);
    hipMemcpy(ds, src.data(), src.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, rows.data(), 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, ds, dr, dout);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int j = 0; j < 8; ++j) for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
        int g = lane >> 4, q = lane & 15;
        float want = (float)(rows[8 * g + j] * 64 + q * 4 + e);
        if (out[(j * 64 + lane) * 4 + e] != want) { if (bad < 5) printf("mismatch j%d lane%d e%d got %g want %g\n", j, lane, e, out[(j*64+lane)*4+e], want); ++bad; }
    }
    printf("global_load_lds 16B per lane, lane-consecutive layout: %s\n", bad ? "WRONG" : "ok");
    return 0;
}
