// v_mfma_f32_16x16x32_f16: is the result the sum over lane groups g and operand slots s of
// A(lane (g,i), slot s) * B(lane (g,j), slot s)?  (slot s of A always meets slot s of B)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/mfma32_probe.hip -o tools/probe/mfma32_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *a, const float *b, float *out) {
    const int lane = threadIdx.x;
    f16x8 x, y;
    for (int s = 0; s < 8; ++s) { x[s] = (_Float16)a[lane * 8 + s]; y[s] = (_Float16)b[lane * 8 + s]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    float a[512], b[512], out[256];
    srand(3);
    for (int i = 0; i < 512; ++i) { a[i] = (float)(rand() % 17 - 8); b[i] = (float)(rand() % 13 - 6); }
    float *da, *db, *dout;
    (void)hipMalloc(&da, sizeof a); (void)hipMalloc(&db, sizeof b); (void)hipMalloc(&dout, sizeof out);
    (void)hipMemcpy(da, a, sizeof a, hipMemcpyHostToDevice); (void)hipMemcpy(db, b, sizeof b, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dout);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    (void)hipMemcpy(out, dout, sizeof out, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int r = 0; r < 4; ++r) {
        const int g = lane >> 4, q = lane & 15, i = 4 * g + r, j = q;  // D[i][j]
        float want = 0;
        for (int gg = 0; gg < 4; ++gg) for (int s = 0; s < 8; ++s) want += a[(16 * gg + i) * 8 + s] * b[(16 * gg + j) * 8 + s];
        if (want != out[lane * 4 + r]) { if (bad < 4) printf("lane %d r %d want %g got %g\n", lane, r, want, out[lane * 4 + r]); ++bad; }
    }
    printf("16x16x32 f16: slot s of A meets slot s of B in every lane group: %s\n", bad ? "NO" : "yes");
    return bad != 0;
}
