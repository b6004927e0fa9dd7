// Probe of the register layouts the block solver relies on (run on a gfx950 device):
//  - v_mfma_f32_4x4x1_16B_f32: which lane feeds which block/row/column, where D lands
//  - v_mfma_f32_16x16x4_f32 with accumulator-layout registers as both operands (X^T Y)
//  - ds_bpermute and the DPP row reductions
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float *a_in, const float *b_in, float *d_out, const float *x_in,
                      const float *y_in, float *xty_out, float *misc) {
    const int lane = threadIdx.x;
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a_in[lane], b_in[lane], c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d_out[lane * 4 + r] = c[r];
    // X, Y given in accumulator layout: lane (g,q), reg r = M[4g+r][q]
    f4 x, y, z = {0, 0, 0, 0};
    for (int r = 0; r < 4; ++r) {
        x[r] = x_in[lane * 4 + r];
        y[r] = y_in[lane * 4 + r];
    }
    for (int s = 0; s < 4; ++s) z = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], y[s], z, 0, 0, 0);
    for (int r = 0; r < 4; ++r) xty_out[lane * 4 + r] = z[r];
    // bpermute: dst[lane] = src[addr/4]
    const int src = lane * 10;
    misc[lane] = (float)__builtin_amdgcn_ds_bpermute(4 * ((lane * 7 + 3) & 63), src);
    // DPP all-reduce over the 16 lanes of a row
    float v = (float)(lane + 1);
    int t;
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false); v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false); v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false); v += __int_as_float(t);
    t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false); v += __int_as_float(t);
    misc[64 + lane] = v;
}

int main() {
    std::vector<float> a(64), b(64), d(256), x(256), y(256), z(256), misc(128);
    for (int l = 0; l < 64; ++l) { a[l] = (float)(l + 1); b[l] = (float)(100 * (l + 1)); }
    for (int i = 0; i < 256; ++i) { x[i] = (float)((i * 37) % 11 - 5); y[i] = (float)((i * 53) % 13 - 6); }
    float *da, *db, *dd, *dx, *dy, *dz, *dm;
    hipMalloc(&da, 256); hipMalloc(&db, 256); hipMalloc(&dd, 1024); hipMalloc(&dx, 1024);
    hipMalloc(&dy, 1024); hipMalloc(&dz, 1024); hipMalloc(&dm, 512);
    hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dx, x.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd, dx, dy, dz, dm);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
    hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost); hipMemcpy(z.data(), dz, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(misc.data(), dm, 512, hipMemcpyDeviceToHost);
    // hypothesis for 4x4x1: block = lane/4, A row i = lane%4, B col j = lane%4, D: lane%4 = col, reg = row
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const int blk = l / 4, j = l % 4;
            const float want = a[blk * 4 + r] * b[blk * 4 + j];
            if (d[l * 4 + r] != want) { if (bad < 8) printf("4x4x1 mismatch lane %d reg %d got %g want %g\n", l, r, d[l*4+r], want); ++bad; }
        }
    printf("4x4x1 layout hypothesis: %s\n", bad ? "WRONG" : "ok");
    // X^T Y with M[4g+r][q] layout
    bad = 0;
    auto X = [&](int i, int j) { return x[(16 * (i / 4) + j) * 4 + i % 4]; };
    auto Y = [&](int i, int j) { return y[(16 * (i / 4) + j) * 4 + i % 4]; };
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float want = 0;
            for (int k = 0; k < 16; ++k) want += X(k, i) * Y(k, j);
            const float got = z[(16 * (i / 4) + j) * 4 + i % 4];
            if (got != want) { if (bad < 8) printf("XtY mismatch (%d,%d) got %g want %g\n", i, j, got, want); ++bad; }
        }
    printf("X^T Y from accumulator-layout registers: %s\n", bad ? "WRONG" : "ok");
    bad = 0;
    for (int l = 0; l < 64; ++l) if (misc[l] != (float)(((l * 7 + 3) & 63) * 10)) ++bad;
    printf("bpermute: %s\n", bad ? "WRONG" : "ok");
    bad = 0;
    for (int l = 0; l < 64; ++l) { const int g = l / 16; const float want = 16 * (16 * g) + 136; if (misc[64 + l] != want) { if (bad < 4) printf("dpp lane %d got %g want %g\n", l, misc[64+l], want); ++bad; } }
    printf("dpp row all-reduce: %s\n", bad ? "WRONG" : "ok");
    return 0;
}
