// Shared host/device helpers for libalsmi (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/alsmi.h"

namespace alsmi {

void set_error(const char *fmt, ...);
int device_cu_count();

#define ALSMI_HIP_CHECK(expr)                                                       \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess) {                                                     \
            alsmi::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                             __FILE__, __LINE__);                                   \
            return ALSMI_EHIP;                                                      \
        }                                                                           \
    } while (0)

#define ALSMI_REQUIRE(cond, ...)           \
    do {                                   \
        if (!(cond)) {                     \
            alsmi::set_error(__VA_ARGS__); \
            return ALSMI_EINVAL;           \
        }                                  \
    } while (0)

constexpr int WAVE = 64;

// ---- MFMA 16x16x4 wrappers --------------------------------------------------
// Operand maps (one element per lane): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15].  The accumulator maps differ between f32 and f64.
template <typename T>
struct Mfma;

template <>
struct Mfma<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    // row of accumulator register r in lane group g (= lane>>4); column = lane&15
    static __device__ __forceinline__ int crow(int g, int r) { return 4 * g + r; }
};

template <>
struct Mfma<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int crow(int g, int r) { return g + 4 * r; }
};

// ---- cross-lane helpers -------------------------------------------------------
__device__ __forceinline__ float read_lane(float v, int lane) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double read_lane(double v, int lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)lo);
}
__device__ __forceinline__ int read_lane(int v, int lane) {
    return __builtin_amdgcn_readlane(v, lane);
}

// NV consecutive elements per lane, NV in {1, 2, 4, 8, 16}; p is NV*sizeof(T) aligned.
template <typename T, int NV>
__device__ __forceinline__ void load_vec(T (&v)[NV], const T *p) {
    static_assert(NV == 1 || NV == 2 || NV == 4 || NV == 8 || NV == 16, "unsupported width");
    if constexpr (NV == 1) {
        v[0] = p[0];
    } else if constexpr (NV == 16) {
        typedef T half_t __attribute__((ext_vector_type(8)));
        const half_t lo = *reinterpret_cast<const half_t *>(p);
        const half_t hi = *reinterpret_cast<const half_t *>(p + 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            v[e] = lo[e];
            v[8 + e] = hi[e];
        }
    } else {
        typedef T vec_t __attribute__((ext_vector_type(NV)));
        const vec_t x = *reinterpret_cast<const vec_t *>(p);
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] = x[e];
    }
}

// ---- fp32 products on the fp16 matrix cores: the scaled two-way split ---------------------------
// Rounded to nearest, x*S = h + l with h = fp16(x*S) and l = fp16(x*S - h) leaves an error of
// 2^-24 |x*S|, the precision of an fp32 value, provided l stays in fp16's normal range; S is a
// power of two chosen per table so that max |x| * S lies in [2^13, 2^14).  Then
// x*y*S*S' = hh' + hl' + lh' up to 2^-24: three 16-bit matrix instructions per product.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// S from the bit pattern of max |x| over the source table (0 for an all-zero table).
__device__ __forceinline__ float split_scale(unsigned max_bits) {
    int e = (int)(max_bits >> 23) - 127;  // floor(log2 max)
    if (max_bits == 0u || (max_bits >> 23) == 255u) e = 13;
    e = e < -40 ? -40 : (e > 40 ? 40 : e);
    return __builtin_bit_cast(float, (unsigned)(127 + 13 - e) << 23);
}

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

// Low part of the split of two values x0*s0, x1*s1 whose high parts are the halves of h:
// fp16(x*s - h) by v_fma_mixlo/mixhi_f16, a fused multiply-add with fp32 and fp16 operands
// rounded once to fp16: one instruction per element instead of the two of "convert h back,
// subtract, round".  With s a power of two x*s - h is exactly representable in fp32 and both
// ways give the same bits (tools/probe/mix_probe.hip).  The compiler selects these instructions
// only here and there by itself, hence the inline assembly.  The wait states the compiler puts
// between a vector instruction and a matrix instruction that reads its result do not cover
// registers written by inline assembly (without them one row in some thousands came out
// wrong): every caller runs split_fence() between its last split_low and the first matrix
// instruction that reads the results.
__device__ __forceinline__ f16x2 split_low(f16x2 h, float x0, float s0, float x1, float s1) {
    const unsigned hb = __builtin_bit_cast(unsigned, h);
    unsigned out;
    asm("v_fma_mixlo_f16 %0, %1, %2, -%5 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(out)
        : "v"(x0), "v"(s0), "v"(x1), "v"(s1), "v"(hb));
    return __builtin_bit_cast(f16x2, out);
}
__device__ __forceinline__ void split_fence() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 3");  // two wait states are required; two more for margin
    __builtin_amdgcn_sched_barrier(0);
}

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

}  // namespace alsmi
