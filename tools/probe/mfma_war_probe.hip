// Do the f32-input matrix instructions of gfx950 need software wait states that hipcc does
// not insert?  Two experiments per instruction (v_mfma_f32_16x16x4_f32, v_mfma_f32_4x4x1_16b_f32),
// each with 0..8 idle issue slots (s_nop) between the two instructions of the pair:
//   WAR: the matrix instruction reads accumulator input v[40:43]; the next vector instructions
//        overwrite v[40:43] with a marker.  A result that contains the marker means the matrix
//        instruction read its accumulator input after the overwrite.
//   RAW: a vector instruction reads the destination v[44:47] right after the matrix
//        instruction.  A stale value means the read overtook the write.
//   WAR via LDS: the same with ds_bpermute_b32 (an LDS return) as the overwriting instruction.
// Half of the waves repeat the pair `iters` times on fresh operands; the other half issue
// back-to-back matrix instructions, so the pipe the pair goes through is contended.  Prints the number of wrong results per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define STR2(x) #x
#define STR(x) STR2(x)

template <int KIND, int NOPS>
__device__ __forceinline__ void pair(float a, float b, float c0, float marker, float &d0,
                                     float &d1, float &d2, float &d3) {
    // KIND 0/1: WAR on 16x16x4 / 4x4x1;  KIND 2/3: RAW on 16x16x4 / 4x4x1
#define MFMA16 "v_mfma_f32_16x16x4_f32 v[44:47], %[a], %[b], v[40:43]\n"
#define MFMA4 "v_mfma_f32_4x4x1_16b_f32 v[44:47], %[a], %[b], v[40:43]\n"
#define LOADC                                                                      \
    "v_mov_b32 v40, %[c]\n v_mov_b32 v41, %[c]\n v_mov_b32 v42, %[c]\n v_mov_b32 v43, %[c]\n" \
    "v_mov_b32 v44, %[m]\n v_mov_b32 v45, %[m]\n v_mov_b32 v46, %[m]\n v_mov_b32 v47, %[m]\n" \
    "s_nop 7\n"
#define GAP "s_nop " STR(NOPSM1) "\n"
#define CLOBBER \
    "v_mov_b32 v40, %[m]\n v_mov_b32 v41, %[m]\n v_mov_b32 v42, %[m]\n v_mov_b32 v43, %[m]\n"
#define DRAIN "s_nop 15\n s_nop 15\n s_nop 15\n"
#define READ \
    "v_mov_b32 %[d0], v44\n v_mov_b32 %[d1], v45\n v_mov_b32 %[d2], v46\n v_mov_b32 %[d3], v47\n"
#define OPS                                                                         \
    : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)                  \
    : [a] "v"(a), [b] "v"(b), [c] "v"(c0), [m] "v"(marker)                            \
    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"
    if constexpr (KIND == 0) {
        if constexpr (NOPS == 0) asm volatile(LOADC MFMA16 CLOBBER DRAIN READ OPS);
        else asm volatile(LOADC MFMA16 "s_nop %[n]\n" CLOBBER DRAIN READ
                          : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                          : [a] "v"(a), [b] "v"(b), [c] "v"(c0), [m] "v"(marker), [n] "n"(NOPS - 1)
                          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if constexpr (KIND == 1) {
        if constexpr (NOPS == 0) asm volatile(LOADC MFMA4 CLOBBER DRAIN READ OPS);
        else asm volatile(LOADC MFMA4 "s_nop %[n]\n" CLOBBER DRAIN READ
                          : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                          : [a] "v"(a), [b] "v"(b), [c] "v"(c0), [m] "v"(marker), [n] "n"(NOPS - 1)
                          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if constexpr (KIND == 2) {
        if constexpr (NOPS == 0) asm volatile(LOADC MFMA16 READ OPS);
        else asm volatile(LOADC MFMA16 "s_nop %[n]\n" READ
                          : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                          : [a] "v"(a), [b] "v"(b), [c] "v"(c0), [m] "v"(marker), [n] "n"(NOPS - 1)
                          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    } else if constexpr (KIND == 4 || KIND == 5) {
        // WAR through an LDS return: ds_bpermute writes v40 (lane permutation of the marker)
        // while the matrix instruction may still be reading v[40:43]
#define BPERM "ds_bpermute_b32 v40, %[addr], %[m]\n ds_bpermute_b32 v41, %[addr], %[m]\n" \
              "ds_bpermute_b32 v42, %[addr], %[m]\n ds_bpermute_b32 v43, %[addr], %[m]\n"
        const int addr = (threadIdx.x & 63) * 4;
        if constexpr (KIND == 4)
            asm volatile(LOADC MFMA16 "s_nop %[n]\n" BPERM "s_waitcnt lgkmcnt(0)\n" DRAIN READ
                         : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                         : [a] "v"(a), [b] "v"(b), [c] "v"(c0), [m] "v"(marker), [n] "n"(NOPS),
                           [addr] "v"(addr)
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
        else
            asm volatile(LOADC MFMA4 "s_nop %[n]\n" BPERM "s_waitcnt lgkmcnt(0)\n" DRAIN READ
                         : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                         : [a] "v"(a), [b] "v"(b), [c] "v"(c0), [m] "v"(marker), [n] "n"(NOPS),
                           [addr] "v"(addr)
                         : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "memory");
    } else {
        if constexpr (NOPS == 0) asm volatile(LOADC MFMA4 READ OPS);
        else asm volatile(LOADC MFMA4 "s_nop %[n]\n" READ
                          : [d0] "=&v"(d0), [d1] "=&v"(d1), [d2] "=&v"(d2), [d3] "=&v"(d3)
                          : [a] "v"(a), [b] "v"(b), [c] "v"(c0), [m] "v"(marker), [n] "n"(NOPS - 1)
                          : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47");
    }
}

template <int KIND, int NOPS>
__global__ __launch_bounds__(256) void probe(int iters, unsigned long long *bad) {
    const int lane = threadIdx.x & 63;
    unsigned long long wrong = 0;
    if ((threadIdx.x >> 6) & 1) {
        // every other wave keeps the SIMD's matrix pipe busy with back-to-back instructions
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 acc[4] = {};
        for (int it = 0; it < iters * 2; ++it)
            for (int k = 0; k < 4; ++k)
                acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)lane, 1.0f, acc[k], 0, 0, 0);
        if (acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0] == 12345.0f) atomicAdd(bad, 1ull << 40);
        return;
    }
    for (int it = 0; it < iters; ++it) {
        // operands: small integers, so every expected value is exact
        const float a = (float)((lane + it) & 3), b = (float)((lane * 7 + it) & 3);
        const float c0 = (float)(1 + (it & 7));
        const float marker = 1.0e6f;
        float d[4];
        pair<KIND, NOPS>(a, b, c0, marker, d[0], d[1], d[2], d[3]);
        // any correct result is c0 + (a sum of at most 4 products <= 9 each): below 1e3
        for (int r = 0; r < 4; ++r) wrong += !(d[r] >= 1.0f && d[r] < 1.0e3f);
    }
    if (wrong) atomicAdd(bad, wrong);
}

template <int KIND, int NOPS>
static void run(const char *name, unsigned long long *dbad) {
    hipMemset(dbad, 0, 8);
    hipLaunchKernelGGL((probe<KIND, NOPS>), dim3(256 * 8), dim3(256), 0, 0, 2000, dbad);
    unsigned long long h = 0;
    hipMemcpy(&h, dbad, 8, hipMemcpyDeviceToHost);
    printf("%-28s idle slots %d: %llu wrong of %llu\n", name, NOPS, h,
           256ull * 8 * 128 * 2000 * 4);
}

int main() {
    unsigned long long *dbad;
    if (hipMalloc(&dbad, 8) != hipSuccess) return 1;
#define ALL(KIND, NAME) \
    run<KIND, 0>(NAME, dbad); run<KIND, 1>(NAME, dbad); run<KIND, 2>(NAME, dbad); \
    run<KIND, 4>(NAME, dbad); run<KIND, 8>(NAME, dbad);
    ALL(0, "WAR 16x16x4_f32 srcC")
    ALL(1, "WAR 4x4x1_16b_f32 srcC")
    ALL(2, "RAW 16x16x4_f32 vdst")
    ALL(3, "RAW 4x4x1_16b_f32 vdst")
    ALL(4, "WAR 16x16x4_f32 srcC via LDS")
    ALL(5, "WAR 4x4x1_16b_f32 srcC via LDS")
    return 0;
}
