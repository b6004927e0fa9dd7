// Does v_mfma_f32_16x16x32_f16 keep fp16 subnormal inputs, and is v_cvt_pk_f16_f32 round-to-nearest?
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(float a0, float b0, float x, float *out) {
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0; b[j] = (_Float16)0; }
    a[0] = (_Float16)a0;
    b[0] = (_Float16)b0;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = c[0];
        f2 v = {x, -x};
        h2 h = __builtin_convertvector(v, h2);
        f2 hf = __builtin_convertvector(h, f2);
        f2 l = v - hf;
        h2 lo = __builtin_convertvector(l, h2);
        f2 lf = __builtin_convertvector(lo, f2);
        out[1] = hf[0]; out[2] = lf[0]; out[3] = hf[1]; out[4] = lf[1];
    }
}
int main() {
    float *d, h[5];
    hipMalloc(&d, 20);
    const float sub = ldexpf(1.0f, -20);  // fp16 subnormal
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, sub, 1024.0f, 0.333333343f, d);
    hipDeviceSynchronize();
    hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
    printf("subnormal a * 1024 = %g (want %g) -> %s\n", h[0], sub * 1024.0f, h[0] == sub * 1024.0f ? "subnormal inputs kept" : "FLUSHED");
    const double x = 0.333333343f;
    printf("x=%.9g h=%.9g l=%.9g  h+l-x = %.3g (rel %.3g)\n", x, h[1], h[2], (double)h[1] + h[2] - x, ((double)h[1] + h[2] - x) / x);
    printf("-x: h=%.9g l=%.9g  sum+x = %.3g\n", h[3], h[4], (double)h[3] + h[4] + x);
    printf("|l|/|x| = %.3g (round to nearest gives <= 2^-12 = %.3g)\n", fabs(h[2]) / x, ldexp(1.0, -12));
    return 0;
}
