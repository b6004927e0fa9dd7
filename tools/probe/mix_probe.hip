// Does v_fma_mixlo_f16 / v_fma_mixhi_f16 give the low part of the two-way fp16 split,
// l = fp16(x*S - fp16(x*S)), bit for bit as "convert h back, subtract, round"?
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mix_probe.hip -o tools/probe/mix_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f16x2 split_low_mix(f16x2 h, float x0, float x1, float scale) {
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    unsigned out;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(out) : "v"(x0), "v"(scale), "v"(hb));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "+v"(out)
        : "v"(x1), "v"(scale), "v"(hb));
    return __builtin_bit_cast(f16x2, out);
}
__global__ void k(const float *x, float scale, int n, unsigned *bad, unsigned *first) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float x0 = x[2 * i], x1 = x[2 * i + 1];
    const f32x2 xs = f32x2{x0, x1} * scale;
    const f16x2 h = __builtin_convertvector(xs, f16x2);
    const f32x2 rest = xs - __builtin_convertvector(h, f32x2);
    const f16x2 want = __builtin_convertvector(rest, f16x2);
    const f16x2 got = split_low_mix(h, x0, x1, scale);
    if (__builtin_bit_cast(unsigned, want) != __builtin_bit_cast(unsigned, got)) {
        if (atomicAdd(bad, 1u) == 0) {
            first[0] = __builtin_bit_cast(unsigned, x0);
            first[1] = __builtin_bit_cast(unsigned, x1);
            first[2] = __builtin_bit_cast(unsigned, want);
            first[3] = __builtin_bit_cast(unsigned, got);
        }
    }
}
int main() {
    const int n = 1 << 24;
    std::vector<float> x(n);
    srand(7);
    for (int i = 0; i < n; ++i) {
        // magnitudes over 30 binades below 1, both signs, a few exact zeros and tiny values
        const float m = (float)rand() / RAND_MAX;
        const int e = rand() % 30;
        x[i] = (rand() & 1 ? 1.0f : -1.0f) * ldexpf(m, -e);
        if (i % 1000 == 0) x[i] = 0.0f;
        if (i % 1000 == 1) x[i] = ldexpf(m, -60);
    }
    float *dx; unsigned *dbad, *dfirst;
    hipMalloc(&dx, n * 4); hipMalloc(&dbad, 4); hipMalloc(&dfirst, 16);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    int rc = 0;
    for (int se = 10; se <= 14; se += 2) {
        hipMemset(dbad, 0, 4);
        hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, dx, ldexpf(1.0f, se), n, dbad, dfirst);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        unsigned bad, first[4];
        hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost); hipMemcpy(first, dfirst, 16, hipMemcpyDeviceToHost);
        printf("scale 2^%d: %u of %d pairs differ", se, bad, n / 2);
        if (bad) { printf(" (first: x %08x %08x want %08x got %08x)", first[0], first[1], first[2], first[3]); rc = 1; }
        printf("\n");
    }
    return rc;
}
