// Which lane does each DPP control used by the library read from?  Prints, per control, the
// source lane seen by lanes 0..15 of the first row.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int CTRL>
__device__ int get(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
__global__ void k(int *out) {
    const int lane = threadIdx.x;
    out[0 * 64 + lane] = get<0x128>(lane);
    out[1 * 64 + lane] = get<0x124>(lane);
    out[2 * 64 + lane] = get<0x12C>(lane);
    out[3 * 64 + lane] = get<0x4E>(lane);
    out[4 * 64 + lane] = get<0xB1>(lane);
    out[5 * 64 + lane] = get<0x141>(lane);
    out[6 * 64 + lane] = get<0x140>(lane);
}
int main() {
    int *d, h[7 * 64];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[] = {"row_ror:8", "row_ror:4", "row_ror:12", "quad[2,3,0,1]", "quad[1,0,3,2]",
                           "row_half_mirror", "row_mirror"};
    for (int c = 0; c < 7; ++c) {
        printf("%-16s", names[c]);
        for (int l = 16; l < 32; ++l) printf(" %2d", h[c * 64 + l] - 16);
        printf("\n");
    }
    return 0;
}
