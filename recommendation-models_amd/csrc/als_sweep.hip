// ALS half-sweep for gfx950: per-row normal equations on MFMA + in-wave Cholesky solve.
//
// Replaces the reference's per-row Python loop (src/als.py:173-300): for every row r
//   A = sum_t f_t f_t^T + alpha n_r I,   b = sum_t val_t f_t,   x_r = A^-1 b.
//
// Data layout
//   factor tables: row-major [n x KP], KP = padded rank (16/32/64/128), pad columns zero.
//   One wave owns one row at a time.  Lane (g = lane>>4, q = lane&15) of a wave loads the
//   NV = KP/16 consecutive elements q*NV .. q*NV+NV-1 of the factor row of neighbour
//   4*step+g, so 16 lanes read one whole contiguous factor row and one wave instruction
//   reads four rows.  Register e of that load is, over the 16 lanes q, a 16-column slice
//   {q*NV+e} of the factor matrix — exactly the one-element-per-lane A and B operand of
//   v_mfma_{f32,f64}_16x16x4 (k = the 4 neighbours of the step).  Tile (e,f) of the Gram
//   matrix is therefore mfma(v[e], v[f]) with no LDS staging and no transpose; only the
//   NV(NV+1)/2 tiles with e <= f are computed.
//   The accumulators go through a wave-private LDS slab once (tile layout -> one matrix row
//   per lane), the LDL^T factorisation, the forward substitution (fused) and the back
//   substitution run inside the wave with v_readlane broadcasts — no barrier, no host round
//   trip.
//   Rows longer than `chunk` entries are cut into chunks; a first kernel writes each chunk's
//   partial accumulators to a workspace slab and the solve kernel adds the slabs of a row in
//   chunk order, so results do not depend on how rows are spread over waves or GPUs.
#include <stdlib.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "alsmi_common.h"

namespace alsmi {

__host__ __device__ constexpr int tri_index(int e, int f, int nv) {
    return e * nv - (e * (e - 1)) / 2 + (f - e);
}

template <typename T, int NV>
struct Gram {
    static constexpr int NTRI = NV * (NV + 1) / 2;
    static constexpr int SLAB_SLOTS = NTRI * 4 + NV + 1;  // x 64 lanes; last = error partial
    typename Mfma<T>::acc_t acc[NTRI];
    T b[NV];

    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int t = 0; t < NTRI; ++t) acc[t] = typename Mfma<T>::acc_t{0, 0, 0, 0};
#pragma unroll
        for (int e = 0; e < NV; ++e) b[e] = T(0);
    }
    __device__ __forceinline__ void store(T *slab, int lane, T err) const {
        int s = 0;
#pragma unroll
        for (int t = 0; t < NTRI; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[(s++) * WAVE + lane] = acc[t][r];
#pragma unroll
        for (int e = 0; e < NV; ++e) slab[(s++) * WAVE + lane] = b[e];
        slab[s * WAVE + lane] = err;
    }
    __device__ __forceinline__ void add(const T *slab, int lane, T &err) {
        int s = 0;
#pragma unroll
        for (int t = 0; t < NTRI; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[t][r] += slab[(s++) * WAVE + lane];
#pragma unroll
        for (int e = 0; e < NV; ++e) b[e] += slab[(s++) * WAVE + lane];
        err += slab[s * WAVE + lane];
    }
};

// Squared error of a row's CURRENT vector u against its stored entries, from the normal
// equations the half-sweep builds anyway (src/als.py:161-165 computes the same quantity in a
// separate pass over all entries):
//     sum_t (r_t - <u, f_t>)^2 = sum_t r_t^2 - 2 <u, b> + u^T G u,   G = sum f f^T, b = sum r f.
// Only sum r^2 has to be collected per entry (ErrAcc; ON = false compiles to nothing): every
// lane squares the one rating it loads per 64-entry block.  The two products are taken once
// per row from the accumulators (row_error_part).
template <typename T, int NV, bool ON>
struct ErrAcc {
    T sum;
    __device__ __forceinline__ void init() { sum = T(0); }
    __device__ __forceinline__ void add(T rating) {
        if constexpr (ON) sum += rating * rating;
    }
};

// This lane's share of the row's squared error; the shares of the 64 lanes add up to it.
// Must run before the ridge goes into the accumulators; inv_s2 undoes the scale of the
// accumulators (1 when they are unscaled).  cur_row = the row's current vector
// (KP elements).  Tile (e,f), register r of lane (g,q) holds G[crow(g,r)*NV+e][q*NV+f]; an
// off-diagonal tile stands for its mirror image too.
template <typename T, int NV>
__device__ __forceinline__ double row_error_part(const Gram<T, NV> &G, T r2,
                                                 const T *__restrict__ cur_row, int lane,
                                                 T inv_s2) {
    asm volatile("" : "+v"(lane));  // derive the lane constants here, not outside the row loop
    const int g = lane >> 4, q = lane & 15;
    T uc[NV];
    load_vec<T, NV>(uc, cur_row + q * NV);
    T ub = T(0), quad = T(0);
#pragma unroll
    for (int e = 0; e < NV; ++e) ub += G.b[e] * uc[e];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        T ur[4];  // u at the tile rows of this lane, dimension class e
#pragma unroll
        for (int r = 0; r < 4; ++r) ur[r] = cur_row[Mfma<T>::crow(g, r) * NV + e];
#pragma unroll
        for (int f = e; f < NV; ++f) {
            T t = T(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) t += G.acc[tri_index(e, f, NV)][r] * ur[r];
            quad += (e == f ? T(1) : T(2)) * t * uc[f];
        }
    }
    return (double)r2 - 2.0 * (double)ub + (double)(quad * inv_s2);
}

// fp64 at padded rank 128 has 36 accumulator tiles of 8 registers, the accumulator file holds
// 32 of them.  The compiler gives every matrix instruction of a function an accumulator-file
// destination and then moves the four tiles that do not fit in and out of it around each
// instruction (16 moves and a wait for the result per use).  Those four tiles (the last four
// of the triangle) take the same instruction with a vector-register destination instead, which
// only inline assembly can ask for.  The compiler does not see a matrix instruction in it and
// adds none of the wait states its result needs before another instruction may touch it
// (19 for this one), so they are part of the statement; they pass while the matrix pipe works on
// the instruction (64 cycles) and cost nothing.
#ifndef F64_VDST_TILES
#define F64_VDST_TILES 4
#endif
typedef double f64x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f64x4 mfma_f64_vdst(double a, double b, f64x4 c) {
    asm("s_nop 1\n\t"  // a vector instruction may just have written an operand
        "v_mfma_f64_16x16x4_f64 %0, %1, %2, %0\n\t"
        "s_nop 15\n\t"
        "s_nop 3"
        : "+v"(c)
        : "v"(a), "v"(b));
    return c;
}

// One k=4 MFMA step: four neighbours (one per lane group), all upper tiles.
template <typename T, int NV>
__device__ __forceinline__ void gram_step(Gram<T, NV> &G, const T (&v)[NV], T rv) {
#pragma unroll
    for (int e = 0; e < NV; ++e) G.b[e] += rv * v[e];
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = e; f < NV; ++f) {
            constexpr int NTRI = Gram<T, NV>::NTRI;
            if constexpr (sizeof(T) == 8 && NV == 8) {
                if (tri_index(e, f, NV) >= NTRI - F64_VDST_TILES) {
                    G.acc[tri_index(e, f, NV)] =
                        mfma_f64_vdst(v[e], v[f], G.acc[tri_index(e, f, NV)]);
                    continue;
                }
            }
            G.acc[tri_index(e, f, NV)] = Mfma<T>::mma(v[e], v[f], G.acc[tri_index(e, f, NV)]);
        }
}

// U consecutive k=4 steps starting at step s: issue all their gathers first, then the MFMAs.
template <typename T, int NV, int U, bool ERR>
__device__ __forceinline__ void gram_steps(Gram<T, NV> &G, ErrAcc<T, NV, ERR> &E,
                                           const T *__restrict__ srcq, int idx, T rv, int s,
                                           int g) {
    constexpr int KP = NV * 16;
    T v[U][NV];
    T r[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int sl = 4 * (s + u) + g;
        const int id = __shfl(idx, sl);
        r[u] = __shfl(rv, sl);
        load_vec<T, NV>(v[u], srcq + (int64_t)id * KP);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        gram_step<T, NV>(G, v[u], r[u]);
    }
}

// Accumulate entries [beg, end) of one row.  The index / rating block of the next 64 entries
// is requested before the current block is processed.
template <typename T, int NV, bool ERR>
__device__ __forceinline__ void gram_range(Gram<T, NV> &G, ErrAcc<T, NV, ERR> &E,
                                           const int32_t *__restrict__ indices,
                                           const T *__restrict__ vals,
                                           const T *__restrict__ src, int64_t beg, int64_t end,
                                           int lane) {
    constexpr int KP = NV * 16;
    // registers: 8 steps in flight cost 8*NV operand registers per lane
#ifndef ALSMI_F64_R128_DEEP
#define ALSMI_F64_R128_DEEP 4
#endif
#ifndef ALSMI_F64_R64_DEEP
#define ALSMI_F64_R64_DEEP 4
#endif
    constexpr int DEEP = (NV * (int)sizeof(T) <= 16) ? 8 : (NV * (int)sizeof(T) == 64 ? ALSMI_F64_R128_DEEP : (sizeof(T) == 8 ? ALSMI_F64_R64_DEEP : 4));
    const int g = lane >> 4, q = lane & 15;
    const T *srcq = src + q * NV;
    int idx_n = 0;
    T rv_n = T(0);
    if (beg + lane < end) {
        idx_n = indices[beg + lane];
        rv_n = vals[beg + lane];
    }
    // fp64 from this many 16-dimension blocks on: software-pipelined in units of PU steps.  The
    // kernel then runs one wave per SIMD (288 accumulator registers at padded rank 128) and
    // nothing else hides the gather's latency: the loads of unit k+1 are issued before the
    // matrix instructions of unit k, across the 64-entry index blocks of the row.
#ifndef F64_PIPE_MIN_NV
#define F64_PIPE_MIN_NV 8
#endif
    constexpr bool PIPE = sizeof(T) == 8 && NV >= F64_PIPE_MIN_NV;
    constexpr int PU = 2;
    T pv[PIPE ? PU : 1][NV], pr[PIPE ? PU : 1];  // the unit whose matrix instructions are due
    bool pending = false;
    for (int64_t base = beg; base < end; base += WAVE) {
        const int64_t left = end - base;
        const int cnt = left < WAVE ? (int)left : WAVE;
        const int idx = idx_n;
        const T rv = rv_n;
        E.add(rv);  // lanes past the end hold a zero rating
        idx_n = 0;
        rv_n = T(0);
        if (base + WAVE + lane < end) {
            idx_n = indices[base + WAVE + lane];
            rv_n = vals[base + WAVE + lane];
        }
        const int nfull = cnt >> 2;
        int s = 0;
        if constexpr (PIPE) {
            const int nunits = nfull / PU;
            // the row's last block ends with one more turn that only drains the pipeline
            const int turns = nunits + (left <= WAVE ? 1 : 0);
#pragma unroll 1
            for (int k = 0; k < turns; ++k) {
                T v[PU][NV], r[PU];
                const bool fetch = k < nunits;
                if (fetch) {
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        const int sl = 4 * (k * PU + u) + g;
                        const int id = __shfl(idx, sl);
                        r[u] = __shfl(rv, sl);
                        load_vec<T, NV>(v[u], srcq + (int64_t)id * KP);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (pending) {
#pragma unroll
                    for (int u = 0; u < PU; ++u) gram_step<T, NV>(G, pv[u], pr[u]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (fetch) {
#pragma unroll
                    for (int u = 0; u < PU; ++u) {
                        pr[u] = r[u];
#pragma unroll
                        for (int e = 0; e < NV; ++e) pv[u][e] = v[u][e];
                    }
                }
                pending = fetch;
            }
            // what is left of the row's last block: at most PU - 1 whole steps and a partial one
#pragma unroll 1
            for (s = nunits * PU; 4 * s < cnt; ++s) {
                const int sl = 4 * s + g;
                const int id = __shfl(idx, sl);
                const T r1 = __shfl(rv, sl);
                T v[NV];
                load_vec<T, NV>(v, srcq + (int64_t)id * KP);
                const bool valid = sl < cnt;
#pragma unroll
                for (int e = 0; e < NV; ++e) v[e] = valid ? v[e] : T(0);
                gram_step<T, NV>(G, v, r1);
            }
            continue;
        }
        for (; s + DEEP <= nfull; s += DEEP)
            gram_steps<T, NV, DEEP, ERR>(G, E, srcq, idx, rv, s, g);
        if constexpr (DEEP > 4) {
            if (s + 4 <= nfull) {
                gram_steps<T, NV, 4, ERR>(G, E, srcq, idx, rv, s, g);
                s += 4;
            }
        }
        for (; s < nfull; ++s) gram_steps<T, NV, 1, ERR>(G, E, srcq, idx, rv, s, g);
        if (cnt & 3) {  // last, partial step: lanes past the end contribute zeros
            const int sl = 4 * nfull + g;
            const int id = __shfl(idx, sl);
            const T r1 = __shfl(rv, sl);
            T v[NV];
            load_vec<T, NV>(v, srcq + (int64_t)id * KP);
            const bool valid = sl < cnt;
#pragma unroll
            for (int e = 0; e < NV; ++e) v[e] = valid ? v[e] : T(0);
            gram_step<T, NV>(G, v, r1);
        }
    }
}

// ---- fp32 Gram build on the bf16 matrix cores (exact three-way split) -------------------------
// The fp32 MFMA runs at the vector-ALU rate and does not overlap with vector work; the bf16 MFMA
// is sixteen times faster per element.  An fp32 value is EXACTLY the sum of three bf16 values
// x = h + m + l (8 + 8 + 8 significant bits, obtained by truncation and exact subtraction), and
// x*y = hh' + hm' + mh' + hl' + lh' + mm' up to terms of relative size 2^-24, each product being
// exact in the fp32 accumulator.  Six v_mfma_f32_16x16x32_bf16 (16 cycles, 32 neighbours each)
// replace eight v_mfma_f32_16x16x4_f32 (32 cycles, 4 neighbours each) per tile: 96 against 256
// cycles, at fp32-equivalent accuracy.
// Operand map of 16x16x32: lane (g,q) holds A[row q][k = 8g + j], j = 0..7.  Load step j of lane
// group g therefore fetches neighbour 8g + j (still four whole rows per wave instruction), and
// the eight loaded registers of column class e ARE the operand — no transpose.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hi16_pair(unsigned lo_elem, unsigned hi_elem) {
    return (lo_elem >> 16) | (hi_elem & 0xffff0000u);  // bf16 of element 2i | element 2i+1
}

// 32 neighbours, at entries off .. off+31 of the wave's 64-entry index block; `valid` of them
// exist (FULL: all 32).
template <int NV, bool FULL, bool ERR>
__device__ __forceinline__ void gram_block32_split(Gram<float, NV> &G, ErrAcc<float, NV, ERR> &E,
                                                   const float *__restrict__ srcq, int idx,
                                                   float rv, int off, int valid, int g) {
    constexpr int KP = NV * 16;
    float v[8][NV];
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int t = 8 * g + j;
        const int id = __shfl(idx, off + t);
        r[j] = __shfl(rv, off + t);
        load_vec<float, NV>(v[j], srcq + (int64_t)id * KP);
        if constexpr (!FULL) {
            const bool ok = t < valid;
#pragma unroll
            for (int e = 0; e < NV; ++e) v[j][e] = ok ? v[j][e] : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int e = 0; e < NV; ++e) G.b[e] += r[j] * v[j][e];
    }
    bf16x8 H[NV], M[NV], L[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        unsigned hb[8], mb[8], lb[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float x = v[j][e];
            const unsigned xb = __builtin_bit_cast(unsigned, x) & 0xffff0000u;
            const float r1 = x - __builtin_bit_cast(float, xb);  // exact
            const unsigned rb = __builtin_bit_cast(unsigned, r1) & 0xffff0000u;
            const float r2 = r1 - __builtin_bit_cast(float, rb);  // exact, <= 8 significant bits
            hb[j] = xb;
            mb[j] = rb;
            lb[j] = __builtin_bit_cast(unsigned, r2);
        }
        u32x4 ph, pm, pl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ph[i] = hi16_pair(hb[2 * i], hb[2 * i + 1]);
            pm[i] = hi16_pair(mb[2 * i], mb[2 * i + 1]);
            pl[i] = hi16_pair(lb[2 * i], lb[2 * i + 1]);
        }
        H[e] = __builtin_bit_cast(bf16x8, ph);
        M[e] = __builtin_bit_cast(bf16x8, pm);
        L[e] = __builtin_bit_cast(bf16x8, pl);
    }
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = e; f < NV; ++f) {
            Mfma<float>::acc_t c = G.acc[tri_index(e, f, NV)];
            // smallest terms first
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M[e], M[f], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H[e], L[f], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L[e], H[f], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H[e], M[f], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M[e], H[f], c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H[e], H[f], c, 0, 0, 0);
            G.acc[tri_index(e, f, NV)] = c;
        }
}

// ---- fp32 Gram build on the fp16 matrix cores (two-way split of scaled values) ----------------
// Rounded to nearest, x*S = h + l with h = fp16(x*S) and l = fp16(x*S - h) leaves an error of
// 2^-24 |x*S| (|l| <= 2^-12 |x*S|, and its own rounding is relative to l), the precision of an
// fp32 value, provided l stays in fp16's normal range.  S is a power of two chosen per source
// table so that max |x| * S lies in [2^13, 2^14): every element larger than max * 2^-16 keeps full
// precision, smaller ones lose bits only below 2^-29 of the largest.  Then
// x*y*S^2 = hh' + hl' + lh' up to 2^-24: three v_mfma_f32_16x16x32_f16 per tile and 32
// neighbours instead of six bf16 ones, and the conversion is 3 vector instructions per element
// (v_pk_mul_f32, v_cvt_pk_f16_f32 twice, two v_cvt_f32_f16, v_pk_add_f32 per pair) instead of
// 5.5.  The accumulators hold S^2 * G; the ridge and the right-hand side are scaled to match
// before the solve.  Operand map as for the bf16 instruction above.
__device__ __forceinline__ float bperm(int byte_addr, float v);
// One block of 4*J neighbours (J per lane group): J = 8 feeds v_mfma_f32_16x16x32_f16, J = 4
// the half-depth v_mfma_f32_16x16x16_f16 (half the operand registers per block; what padded
// rank 128 needs to run two waves per SIMD beside its 36 accumulator tiles).
template <int J>
struct F16Op;
template <>
struct F16Op<8> {
    typedef f16x8 vec_t;
    static __device__ __forceinline__ Mfma<float>::acc_t mma(vec_t a, vec_t b,
                                                             Mfma<float>::acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <>
struct F16Op<4> {
    typedef f16x4 vec_t;
    static __device__ __forceinline__ Mfma<float>::acc_t mma(vec_t a, vec_t b,
                                                             Mfma<float>::acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
    }
};

// 4*J neighbours at entries off .. off+4J-1 of the wave's 64-entry index block; `valid` of
// them exist.  FULL = 1: all of them (known at compile time), 0: fewer, 2: decided at run time
// by a branch around the masking alone, so that the matrix instructions exist once (two copies
// of them meet in a copy of all accumulators, which two waves per SIMD have no registers for).
template <int NV, int J, int FULL, bool WIDE, bool ERR>
__device__ __forceinline__ void gram_block_f16(Gram<float, NV> &G, ErrAcc<float, NV, ERR> &E,
                                               const float *__restrict__ src, unsigned qoff,
                                               int idx, float rv, int off, int valid, int g,
                                               float scale) {
    constexpr int KP = NV * 16;
    float v[J][NV];
    float r[J];
    // ds_bpermute byte address of lane J*g; the constant part of each neighbour's lane goes
    // into the instruction's offset field, so one register serves all fetches
    const int lane_jg = 4 * J * g;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int t = J * g + j;
        const unsigned id = (unsigned)__builtin_amdgcn_ds_bpermute(lane_jg + 4 * (off + j), idx);
        r[j] = bperm(lane_jg + 4 * (off + j), rv);
        if constexpr (WIDE) {
            // 64-bit addressing: at padded rank 128 a table of 10 M rows passes 4 GiB
            load_vec<float, NV>(v[j], reinterpret_cast<const float *>(
                                          reinterpret_cast<const char *>(src) +
                                          ((uint64_t)id * (KP * 4u) + qoff)));
        } else {
            // uniform base + 32-bit byte offset (the host checks that the table is below 4 GiB)
            load_vec<float, NV>(v[j], reinterpret_cast<const float *>(
                                          reinterpret_cast<const char *>(src) +
                                          (id * (KP * 4u) + qoff)));
        }
        if constexpr (FULL == 0) {
            const bool ok = t < valid;
#pragma unroll
            for (int e = 0; e < NV; ++e) v[j][e] = ok ? v[j][e] : 0.0f;
        }
    }
    if constexpr (FULL == 2) {
        if (valid < 4 * J) {
#pragma unroll
            for (int j = 0; j < J; ++j) {
                const bool ok = J * g + j < valid;
#pragma unroll
                for (int e = 0; e < NV; ++e) v[j][e] = ok ? v[j][e] : 0.0f;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int e = 0; e < NV; ++e) G.b[e] += r[j] * v[j][e];
    }
    // The packed fp32 instructions work on the register pairs a load delivers (two dimensions
    // of one neighbour); the fp16 packs join two neighbours of one dimension, which the
    // conversion instructions take from any two registers.  Two neighbours at a time, so
    // that few temporaries are live.
    typename F16Op<J>::vec_t H[NV], L[NV];
#pragma unroll
    for (int i = 0; i < J / 2; ++i) {
        float xs[2][NV];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            if constexpr (NV >= 2) {
#pragma unroll
                for (int e = 0; e < NV; e += 2) {
                    const f32x2 x = f32x2{v[2 * i + jj][e], v[2 * i + jj][e + 1]} * scale;
                    xs[jj][e] = x[0];
                    xs[jj][e + 1] = x[1];
                }
            } else {
                xs[jj][0] = v[2 * i + jj][0] * scale;
            }
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const f16x2 h = __builtin_convertvector(f32x2{xs[0][e], xs[1][e]}, f16x2);
            H[e][2 * i] = h[0];
            H[e][2 * i + 1] = h[1];
            const f16x2 l = split_low(h, v[2 * i][e], scale, v[2 * i + 1][e], scale);
            L[e][2 * i] = l[0];
            L[e][2 * i + 1] = l[1];
        }
    }
    split_fence();
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = e; f < NV; ++f) {
            Mfma<float>::acc_t c = G.acc[tri_index(e, f, NV)];
            c = F16Op<J>::mma(H[e], L[f], c);
            c = F16Op<J>::mma(L[e], H[f], c);
            c = F16Op<J>::mma(H[e], H[f], c);
            G.acc[tri_index(e, f, NV)] = c;
        }
}

// The same block in two halves, for the software-pipelined loop (padded rank 64 and 128):
// gram_issue issues the block's loads, gram_convert turns them, once they have arrived, into
// the operand registers, gram_mma issues the block's matrix instructions.  The loop runs the
// matrix instructions of block n between the loads of block n+1 and their first use, so that a
// wave's memory latency and its matrix work overlap (the other waves of the SIMD are in the
// same loop, or in their solve).  The loop is not unrolled: at padded rank 64 that also frees
// the registers the four unrolled blocks of the plain form needed.
template <int NV, int J>
__device__ __forceinline__ void gram_mma(Gram<float, NV> &G,
                                         const typename F16Op<J>::vec_t (&H)[NV],
                                         const typename F16Op<J>::vec_t (&L)[NV]) {
    if constexpr (J == 4) {
        // The half-depth instruction runs at half the rate of the full-depth one (both take 16
        // cycles).  With a lane's eight contraction slots = its four neighbours twice, one
        // full-depth instruction takes two products of a tile at once:
        //   [H_e | L_e]^T [L_f | H_f] = H_e^T L_f + L_e^T H_f,
        //   [H_e | L_e]^T [H_f | L_f] = H_e^T H_f + L_e^T L_f
        // (the fourth product, which the three-instruction form drops, comes for free): 32
        // instead of 48 matrix-pipe cycles per tile and block.  [L_f | H_f] is put together per
        // tile column (four register moves), so that only one is live at a time.  (The second
        // product as a half-depth instruction behind the first gave garbage in some tiles: two
        // matrix instructions of different depth back to back on one accumulator.)
        f16x8 hl[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e)
            hl[e] = __builtin_shufflevector(H[e], L[e], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int f = 0; f < NV; ++f) {
            const f16x8 lh = __builtin_shufflevector(L[f], H[f], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int e = 0; e <= f; ++e) {
                Mfma<float>::acc_t c = G.acc[tri_index(e, f, NV)];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(hl[e], lh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(hl[e], hl[f], c, 0, 0, 0);
                G.acc[tri_index(e, f, NV)] = c;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < NV; ++e)
#pragma unroll
            for (int f = e; f < NV; ++f) {
                Mfma<float>::acc_t c = G.acc[tri_index(e, f, NV)];
                c = F16Op<J>::mma(H[e], L[f], c);
                c = F16Op<J>::mma(L[e], H[f], c);
                c = F16Op<J>::mma(H[e], H[f], c);
                G.acc[tri_index(e, f, NV)] = c;
            }
    }
}

template <int NV, int J, bool WIDE>
__device__ __forceinline__ void gram_issue(float (&v)[J][NV], float (&r)[J],
                                           const float *__restrict__ src, unsigned qoff, int idx,
                                           float rv, int off, int g) {
    constexpr int KP = NV * 16;
    const int lane_jg = 4 * J * g;
#pragma unroll
    for (int j = 0; j < J; ++j) {
        const int at = lane_jg + 4 * (off + j);
        const unsigned id = (unsigned)__builtin_amdgcn_ds_bpermute(at, idx);
        r[j] = bperm(at, rv);
        if constexpr (WIDE)
            load_vec<float, NV>(v[j], reinterpret_cast<const float *>(
                                          reinterpret_cast<const char *>(src) +
                                          ((uint64_t)id * (KP * 4u) + qoff)));
        else
            load_vec<float, NV>(v[j], reinterpret_cast<const float *>(
                                          reinterpret_cast<const char *>(src) +
                                          (id * (KP * 4u) + qoff)));
    }
}

template <int NV, int J>
__device__ __forceinline__ void gram_convert(Gram<float, NV> &G, float (&v)[J][NV],
                                             const float (&r)[J], int valid, int g, float scale,
                                             typename F16Op<J>::vec_t (&H)[NV],
                                             typename F16Op<J>::vec_t (&L)[NV]) {
    if (valid < 4 * J) {
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const bool ok = J * g + j < valid;
#pragma unroll
            for (int e = 0; e < NV; ++e) v[j][e] = ok ? v[j][e] : 0.0f;
        }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) {
#pragma unroll
        for (int e = 0; e < NV; ++e) G.b[e] += r[j] * v[j][e];
    }
#pragma unroll
    for (int i = 0; i < J / 2; ++i) {
        float xs[2][NV];
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < NV; e += 2) {
                const f32x2 x = f32x2{v[2 * i + jj][e], v[2 * i + jj][e + 1]} * scale;
                xs[jj][e] = x[0];
                xs[jj][e + 1] = x[1];
            }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            const f16x2 h = __builtin_convertvector(f32x2{xs[0][e], xs[1][e]}, f16x2);
            H[e][2 * i] = h[0];
            H[e][2 * i + 1] = h[1];
            const f16x2 l = split_low(h, v[2 * i][e], scale, v[2 * i + 1][e], scale);
            L[e][2 * i] = l[0];
            L[e][2 * i + 1] = l[1];
        }
    }
}

// Waves per SIMD the fp32 rank-64 kernels are compiled for (register budget 512 / waves).
#ifndef SOLVE_MIN_WAVES
#define SOLVE_MIN_WAVES 3
#endif
// Padded rank 128 (fp32): two waves per SIMD beside the 36 accumulator tiles, which the Gram
// build pays for with half-depth matrix instructions (16 neighbours per block).
#ifndef R128_MIN_WAVES
#define R128_MIN_WAVES 2
#endif

// How an instantiation builds the Gram matrix.
enum { GRAM_NATIVE = 0,  // the MFMA of the data type (fp64; fp32 at padded rank 128)
       GRAM_BF16X3 = 1,  // fp32 data, exact three-way bf16 split, unscaled accumulators
       GRAM_F16X2 = 2,   // fp32 data, scaled two-way fp16 split, accumulators hold S^2 G
       GRAM_F16X2_WIDE = 3 };  // the same with 64-bit gather addresses (tables above 4 GiB)
__host__ __device__ constexpr bool is_f16x2(int mode) {
    return mode == GRAM_F16X2 || mode == GRAM_F16X2_WIDE;
}

// Accumulate entries [beg, end) of one row on the 16-bit matrix cores (fp32 data).
template <int NV, int MODE, bool ERR>
__device__ __forceinline__ void gram_range_split(Gram<float, NV> &G, ErrAcc<float, NV, ERR> &E,
                                                 const int32_t *__restrict__ indices,
                                                 const float *__restrict__ vals,
                                                 const float *__restrict__ src, int64_t beg,
                                                 int64_t end, int lane, float scale) {
    const int g = lane >> 4, q = lane & 15;
    const float *srcq = src + q * NV;
    int idx_n = 0;
    float rv_n = 0.0f;
    if (beg + lane < end) {
        idx_n = indices[beg + lane];
        rv_n = vals[beg + lane];
    }
    // neighbours per lane group and block: 8 (the full-depth instruction) except at padded rank
    // 128 compiled for two waves per SIMD, where the operand registers of a block must halve
    constexpr int J = (NV == 8 && R128_MIN_WAVES >= 2) ? 4 : 8;
    auto block = [&](auto full, int idx, float rv, int off, int valid) {
        constexpr bool FULL = decltype(full)::value;
        if constexpr (is_f16x2(MODE))
            gram_block_f16<NV, J, FULL ? 1 : 0, MODE == GRAM_F16X2_WIDE, ERR>(
                G, E, src, (unsigned)(q * NV * 4), idx, rv, off, valid, g, scale);
        else
            gram_block32_split<NV, FULL, ERR>(G, E, srcq, idx, rv, off, valid, g);
    };
    constexpr int BLK = is_f16x2(MODE) ? 4 * J : 32;  // neighbours per block
    // software-pipelined from this many 16-dimension blocks on (measured: -2 % at padded rank
    // 64, -4 % at 128; narrower ranks keep the unrolled plain form)
#ifndef GRAM_PIPE_MIN_NV
#define GRAM_PIPE_MIN_NV 4
#endif
    constexpr bool PIPE = NV >= GRAM_PIPE_MIN_NV && is_f16x2(MODE);
    // pipelined loop: the operands of the block whose matrix instructions are still to be issued
    typename F16Op<J>::vec_t H[NV], L[NV];
    bool pending = false;
    for (int64_t base = beg; base < end; base += WAVE) {
        const int64_t left = end - base;
        const int cnt = left < WAVE ? (int)left : WAVE;
        const int idx = idx_n;
        const float rv = rv_n;
        E.add(rv);  // lanes past the end hold a zero rating
        idx_n = 0;
        rv_n = 0.0f;
        if (base + WAVE + lane < end) {
            idx_n = indices[base + WAVE + lane];
            rv_n = vals[base + WAVE + lane];
        }
        if constexpr (PIPE) {
#pragma unroll 1
            for (int off = 0; off < cnt; off += BLK) {
                float v[J][NV], r[J];
                gram_issue<NV, J, MODE == GRAM_F16X2_WIDE>(v, r, src, (unsigned)(q * NV * 4), idx,
                                                           rv, off, g);
                __builtin_amdgcn_sched_barrier(0);
                if (pending) gram_mma<NV, J>(G, H, L);  // the previous block, under the loads
                __builtin_amdgcn_sched_barrier(0);
                gram_convert<NV, J>(G, v, r, cnt - off, g, scale, H, L);
                split_fence();
                pending = true;
            }
        } else {
#pragma unroll
            for (int off = 0; off < WAVE; off += BLK) {
                if (cnt >= off + BLK)
                    block(std::true_type{}, idx, rv, off, BLK);
                else if (cnt > off)
                    block(std::false_type{}, idx, rv, off, cnt - off);
            }
        }
    }
    if constexpr (PIPE) {
        if (pending) gram_mma<NV, J>(G, H, L);
    }
}

// fp32 data goes to the 16-bit matrix cores; fp64 keeps its own MFMA.  The debug dump of the
// normal equations accumulates exactly as the half-sweep does and undoes the power-of-two
// scale afterwards, so that it shows what production solves.  ABLATE 4 / 10 force the native
// MFMA / the bf16 split (timing comparisons).
template <typename T, int NV, int ABLATE, bool DUMP>
constexpr int gram_mode() {
    if (sizeof(T) != 4 || NV > 8 || ABLATE == 4) return GRAM_NATIVE;
    if (ABLATE == 20) return GRAM_F16X2_WIDE;
    return ABLATE == 10 ? GRAM_BF16X3 : GRAM_F16X2;
}

template <typename T, int NV, int MODE, bool ERR>
__device__ __forceinline__ void gram_accumulate(Gram<T, NV> &G, ErrAcc<T, NV, ERR> &E,
                                                const int32_t *indices, const T *vals,
                                                const T *src, int64_t beg, int64_t end,
                                                int lane, float scale) {
    if constexpr (MODE != GRAM_NATIVE && sizeof(T) == 4)
        gram_range_split<NV, MODE, ERR>(G, E, indices, vals, src, beg, end, lane, scale);
    else
        gram_range<T, NV, ERR>(G, E, indices, vals, src, beg, end, lane);
}

// Largest |x| of a table as the bit pattern of a non-negative float (which orders like the
// unsigned integer), combined with atomicMax into *out, which is zero when the kernel starts:
// the plan keeps two such words and every call zeroes the one the NEXT call accumulates into
// (`next`), so no separate fill precedes the kernel.  NaN entries are skipped here and propagate
// through the arithmetic instead.
__global__ __launch_bounds__(256) void max_abs_kernel(const float *__restrict__ x, int64_t n4,
                                                      unsigned *__restrict__ out,
                                                      unsigned *__restrict__ next) {
    typedef float vec_t __attribute__((ext_vector_type(4)));
    __shared__ float part[4];
    if (blockIdx.x == 0 && threadIdx.x == 0) *next = 0u;
    auto vmax = [](vec_t a) {
        return fmaxf(fmaxf(fabsf(a[0]), fabsf(a[1])), fmaxf(fabsf(a[2]), fabsf(a[3])));
    };
    float m0 = 0.0f, m1 = 0.0f, m2 = 0.0f, m3 = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {  // four independent loads in flight
        const vec_t a = reinterpret_cast<const vec_t *>(x)[i];
        const vec_t b = reinterpret_cast<const vec_t *>(x)[i + stride];
        const vec_t c = reinterpret_cast<const vec_t *>(x)[i + 2 * stride];
        const vec_t d = reinterpret_cast<const vec_t *>(x)[i + 3 * stride];
        m0 = fmaxf(m0, vmax(a));
        m1 = fmaxf(m1, vmax(b));
        m2 = fmaxf(m2, vmax(c));
        m3 = fmaxf(m3, vmax(d));
    }
    for (; i < n4; i += stride) m0 = fmaxf(m0, vmax(reinterpret_cast<const vec_t *>(x)[i]));
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)  // one atomic per block
        atomicMax(out, __builtin_bit_cast(
                           unsigned, fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}


// ---- kernel 1: partial normal equations of the chunks of long rows ---------------------
// ERR: also the chunk's sum of squared ratings (last slab slot).
template <typename T, int NV, int MODE, bool ERR = false>
__global__ __launch_bounds__(256, (NV == 4 ? (sizeof(T) == 4 ? SOLVE_MIN_WAVES : 2) : (NV == 8 && sizeof(T) == 4 && is_f16x2(MODE) ? R128_MIN_WAVES : 1))) void gram_chunks_kernel(
    const int64_t *__restrict__ ch_beg, const int64_t *__restrict__ ch_end, int n_chunks,
    const int32_t *__restrict__ indices, const T *__restrict__ vals, const T *__restrict__ src,
    const unsigned *__restrict__ max_bits, T *__restrict__ slabs) {
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (w >= n_chunks) return;
    Gram<T, NV> G;
    G.zero();
    ErrAcc<T, NV, ERR> E;
    E.init();
    float scale = 1.0f;
    if constexpr (is_f16x2(MODE)) scale = split_scale(max_bits[0]);
    gram_accumulate<T, NV, MODE, ERR>(G, E, indices, vals, src, ch_beg[w], ch_end[w], lane, scale);
    G.store(slabs + (size_t)w * (Gram<T, NV>::SLAB_SLOTS * WAVE), lane, E.sum);
}

// Sum of the per-wave squared-error partials, in index order, by one thread block.
__global__ __launch_bounds__(256) void sum_err_kernel(const double *__restrict__ part, int n,
                                                      double *__restrict__ out) {
    __shared__ double sh[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sh[0];
}

// ---- kernel 1b: add the slabs of each long row into its first slab -----------------------
// One thread per (row, accumulator element); the chunks are added in chunk order, so the sum
// is reproducible.  Replaces a serial walk over up to dozens of slabs by the wave that solves
// the row.
template <typename T>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const int32_t *__restrict__ slab_first,
                                                        const int32_t *__restrict__ slab_count,
                                                        int n_heavy, int slab_elems,
                                                        T *__restrict__ slabs) {
    const int per_row = (slab_elems + 255) / 256;
    const int t = blockIdx.x / per_row;
    if (t >= n_heavy) return;
    const int i = (blockIdx.x % per_row) * 256 + threadIdx.x;
    if (i >= slab_elems) return;
    const int nch = slab_count[t];
    T *first = slabs + (size_t)slab_first[t] * slab_elems;
    T sum = first[i];
    for (int c = 1; c < nch; ++c) sum += first[(size_t)c * slab_elems + i];
    first[i] = sum;
}

// ---- LDS slab geometry ---------------------------------------------------------------------
template <typename T, int KP>
struct Slab {
    static constexpr int PAD = 16 / (int)sizeof(T);
    static constexpr int LD = KP + PAD;
    static constexpr int ELEMS = KP * LD + 2 * KP;  // matrix, rhs, 1/diag
    static constexpr size_t BYTES = (size_t)ELEMS * sizeof(T);
};

// Accumulator tiles -> symmetric matrix in LDS (natural column order), rhs, ridge.
template <typename T, int NV>
__device__ __forceinline__ void assemble(Gram<T, NV> &G, T *Asm, T *bsm, int lane, int rank,
                                         T ridge) {
    constexpr int KP = NV * 16;
    constexpr int LD = Slab<T, KP>::LD;
    const int g = lane >> 4, q = lane & 15;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        G.b[e] += __shfl_xor(G.b[e], 16);
        G.b[e] += __shfl_xor(G.b[e], 32);
    }
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int f = e; f < NV; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = Mfma<T>::crow(g, r) * NV + e;
                const int j = q * NV + f;
                const T val = G.acc[tri_index(e, f, NV)][r];
                Asm[i * LD + j] = val;
                if (e != f) Asm[j * LD + i] = val;
            }
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) bsm[q * NV + e] = G.b[e];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rr = 0; rr < (KP + 63) / 64; ++rr) {
        const int row = lane + 64 * rr;
        if (row < KP) Asm[row * LD + row] += (row < rank) ? ridge : T(1);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 1/d for a positive pivot.  fp32: hardware reciprocal plus one Newton step (full fp32
// accuracy, no division sequence on the critical path); fp64: a true division.
__device__ __forceinline__ float pivot_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(r, fmaf(-d, r, 1.0f), r);
}
__device__ __forceinline__ double pivot_rcp(double d) { return 1.0 / d; }

// ---- fast solve path (padded rank <= 64): registers only, LDS just for a small stage ----------
// LDS per wave: a [KP x 16] transposition stage (row stride padded) and the right-hand side.
template <typename T, int KP>
struct Stage {
    static constexpr int PAD = 16 / (int)sizeof(T);
    static constexpr int LD = 16 + PAD;
    static constexpr int ELEMS = KP * LD + KP;
    static constexpr size_t BYTES = (size_t)ELEMS * sizeof(T);
};

// alpha*n on the diagonal of the real dimensions, 1 on the pad dimensions, applied to the
// accumulators in place: the diagonal of tile (e,e) sits where the tile row equals the lane's
// column.
template <typename T, int NV>
__device__ __forceinline__ void add_ridge(Gram<T, NV> &G, int lane, int rank, T ridge) {
    const int g = lane >> 4, q = lane & 15;
#pragma unroll
    for (int e = 0; e < NV; ++e)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool diag = Mfma<T>::crow(g, r) == q;
            const T add = (q * NV + e < rank) ? ridge : T(1);
            G.acc[tri_index(e, e, NV)][r] += diag ? add : T(0);
        }
}

// The same for the neighbour-by-neighbour system of a short row (see solve_rows_dual_kernel):
// tile (a,a) holds neighbours 16a .. 16a+15, of which the first n_real of the row exist.
template <int ND>
__device__ __forceinline__ void add_ridge_dual(Gram<float, ND> &G, int lane, int n_real,
                                               float ridge) {
    const int g = lane >> 4, q = lane & 15;
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool diag = 4 * g + r == q;
            const float add = (16 * a + q < n_real) ? ridge : 1.0f;
            G.acc[tri_index(a, a, ND)][r] += diag ? add : 0.0f;
        }
}

template <int ND>
__device__ __forceinline__ void add_ridge_dual_d(Gram<double, ND> &G, int lane, int n_real,
                                                 double ridge) {
    const int g = lane >> 4, q = lane & 15;
#pragma unroll
    for (int a = 0; a < ND; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool diag = Mfma<double>::crow(g, r) == q;
            const double add = (16 * a + q < n_real) ? ridge : 1.0;
            G.acc[tri_index(a, a, ND)][r] += diag ? add : 0.0;
        }
}

// Accumulator tiles -> one matrix row per lane, NV passes of 16 columns each through the
// stage.  Pass f moves the columns {c*NV + f}: tiles (e,f) with e <= f are stored as they are,
// tiles with e > f come from tile (f,e) by symmetry.  a[] ends up in natural column order.
template <typename T, int NV>
__device__ __forceinline__ void rows_to_regs(Gram<T, NV> &G, T *stage, T *bsm, int lane,
                                             T (&a)[NV * 16], T &bb) {
    constexpr int KP = NV * 16;
    constexpr int LD = Stage<T, KP>::LD;
    constexpr int VW = 16 / (int)sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(VW)));
    const int g = lane >> 4, q = lane & 15;
    const int row = lane < KP ? lane : KP - 1;
#pragma unroll
    for (int e = 0; e < NV; ++e) {
        G.b[e] += __shfl_xor(G.b[e], 16);
        G.b[e] += __shfl_xor(G.b[e], 32);
    }
    if (g == 0) {
#pragma unroll
        for (int e = 0; e < NV; ++e) bsm[q * NV + e] = G.b[e];
    }
#pragma unroll
    for (int f = 0; f < NV; ++f) {
#pragma unroll
        for (int e = 0; e < NV; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cr = Mfma<T>::crow(g, r);
                if (e <= f)
                    stage[(cr * NV + e) * LD + q] = G.acc[tri_index(e, f, NV)][r];
                else
                    stage[(q * NV + e) * LD + cr] = G.acc[tri_index(f, e, NV)][r];
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int c = 0; c < 16; c += VW) {
            const vec_t t = *reinterpret_cast<const vec_t *>(stage + row * LD + c);
#pragma unroll
            for (int i = 0; i < VW; ++i) a[(c + i) * NV + f] = t[i];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    bb = bsm[row];
}

// One Gauss-Jordan step: eliminate column P from every other row.  With one row per lane the
// rows above the pivot are lanes that a triangular factorisation would leave idle, so clearing
// them too costs no instruction and removes the back substitution.  The trailing block stays
// the symmetric Schur complement, hence the pivot row's entries A[P][c] are read from column P
// of lane c.
template <typename T, int KP, int P>
struct GjStep {
    static __device__ __forceinline__ void run(T (&a)[KP], T &bb, T &dinv, int &bad, int lane) {
        constexpr int BATCH = 8;
        const T u = a[P];
        const T d = read_lane(u, P);
        bad |= (d > T(0)) ? 0 : 1;
        const T inv = pivot_rcp(d);
        const T zp = read_lane(bb, P);
        const bool me = lane == P;
        const T m = me ? T(0) : u * inv;  // multiplier of this row; the pivot row is kept
        dinv = me ? inv : dinv;
        bb -= m * zp;
#pragma unroll
        for (int c0 = P + 1; c0 < KP; c0 += BATCH) {
            T sc[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i)
                if (c0 + i < KP) sc[i] = read_lane(u, c0 + i);  // A[P][c] = A[c][P]
#pragma unroll
            for (int i = 0; i < BATCH; ++i)
                if (c0 + i < KP) a[c0 + i] -= m * sc[i];
        }
        GjStep<T, KP, P + 1>::run(a, bb, dinv, bad, lane);
    }
};
template <typename T, int KP>
struct GjStep<T, KP, KP> {
    static __device__ __forceinline__ void run(T (&)[KP], T &, T &, int &, int) {}
};

// Solve from the accumulators; x_out is the solution component of row `lane`.
// Returns false when a pivot is not positive.
template <typename T, int NV>
__device__ __forceinline__ bool gj_solve(Gram<T, NV> &G, T *stage, T *bsm, int lane, int rank,
                                         T ridge, T &x_out) {
    constexpr int KP = NV * 16;
    add_ridge<T, NV>(G, lane, rank, ridge);
    T a[KP];
    T bb;
    rows_to_regs<T, NV>(G, stage, bsm, lane, a, bb);
    T dinv = T(1);
    int bad = 0;  // bitwise, so that no step is skipped through an early-out branch
    GjStep<T, KP, 0>::run(a, bb, dinv, bad, lane);
    x_out = bb * dinv;
    return bad == 0;
}

// ---- fp32 solve on the matrix cores, straight from the accumulators (padded rank <= 64) -------
// The NV x NV tiles of the accumulators are the blocks of the normal matrix in a permuted index
// order (block e = the dimensions congruent to e modulo NV), each in the matrix core's result
// layout: lane (g,q), register r = M[4g+r][q].  Three facts make a block LDL^T factorisation
// possible without an LDS transposition stage:
//  * fed to v_mfma_f32_16x16x4 as BOTH operands, register s at step s, two tiles X and Y in
//    that layout give X^T Y (the contraction runs over the row index of both);
//  * the same registers are the sixteen 4x4 blocks of v_mfma_f32_4x4x1 (block = lane/4,
//    register = row, lane%4 = column), so adding a multiple of one tile row to all the others
//    (one elimination step) is one such instruction whose operands are one value per lane;
//  * row p of a tile sits in one register of the 16 lanes (p/4, .), so one ds_bpermute hands it
//    to every lane, indexed by the lane's column or by the row the lane feeds to the 4x4x1.
// Per pivot block k, sixteen elimination steps (rows below the pivot only; the multipliers come
// from row p through the symmetry of the trailing block) are applied to the whole block row
// [S_kk | S_kJ, J > k | R]: R starts as the identity and ends as inv(L_kk) (its last column,
// which no row operation changes, carries the block's right-hand side instead), S_kJ becomes
// Z_J = inv(L_kk) S_kJ.  The trailing
// blocks then take S_IJ -= Z_I^T D^-1 Z_J on the 16x16x4 instruction.  No Schur complement is
// ever inverted, so the error behaves like that of a Cholesky solve.  Vectors live in two
// layouts, "column" (lane (.,q) holds v[q]) and "row" (lane (g,.) register r holds v[4g+r]); a
// product with a tile turns one into the other.
typedef float f4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(float, t);
}
// the value of another lane of the row (DPP control CTRL)
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a row, result in each of them
__device__ __forceinline__ float row_allreduce(float v) {
    v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
    v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
    v = dpp_add<0x141>(v);  // row_half_mirror
    v = dpp_add<0x140>(v);  // row_mirror
    return v;
}
// sum over the four rows (lanes with the same q), result in each of them
__device__ __forceinline__ float col_allreduce(float v) {
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}
__device__ __forceinline__ float bperm(int byte_addr, float v) {
    return __builtin_bit_cast(float,
                              __builtin_amdgcn_ds_bpermute(byte_addr, __builtin_bit_cast(int, v)));
}
// Scheduling fence after a group of fp32 matrix instructions.  History: an early version of
// the block solver gave results that changed from run to run in about one row in 10^5, and
// this fence went in together with the rewrite that removed the symptom, on the theory that
// the compiler lets a vector instruction overwrite the accumulator INPUT of an fp32 matrix
// instruction right behind it (it does emit that pattern: tools/mfma_war_scan.py lists the
// sites, because LLVM applies its "MFMA reads SrcC, VALU writes it" wait states only to the
// instructions it classes as XDL, and the f32-input ones are not).  The theory is wrong for
// gfx950: tools/probe/mfma_war_probe.hip overwrites the accumulator input (by a vector move and
// by an LDS return) and reads the destination 0..8 issue slots after v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_4x4x1_16b_f32, with the matrix pipe kept busy by other waves, and finds no wrong
// value in 2 x 10^9 results per variant: the hardware interlocks these instructions.  A build
// without the fence (-DALSMI_NO_MFMA_FENCE) is bit-reproducible over 8 runs of 1.6 M row
// solves at ranks 10 to 128 and runs at the same speed.  The fence is kept only because it
// pins the instruction order that all parity results were obtained with; the run-to-run
// bit-equality test at full size (tests/test_gpu_fullsize.py) guards the property itself.
template <int NOPS>
__device__ __forceinline__ void mfma_f32_fence() {
#ifndef ALSMI_NO_MFMA_FENCE
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop %0" ::"n"(NOPS));
    __builtin_amdgcn_sched_barrier(0);
#endif
}
__device__ __forceinline__ f4 mfma_xty(f4 x, f4 y, f4 c) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f32_16x16x4f32(x[s], y[s], c, 0, 0, 0);
    mfma_f32_fence<7>();
    return c;
}

// ---- trailing updates of the block solver on the fp16 matrix cores -----------------------------
// S_IJ -= Z_I^T D^-1 Z_J is 60 % of the solver's matrix-pipe time on the fp32 instruction (32
// busy cycles per 16x16x4, and the vector ALU waits meanwhile).  With Zs = D^-1/2 Z it is
// Zs_I^T Zs_J, the Cholesky form, whose elements are bounded by the square root of the largest
// diagonal element of the matrix; solve_prescale() brings that into [2^27, 2^28) by a power of
// two (exact; the right-hand side takes the same factor, the solution none), so |Zs| < 2^14 and
// the two-way fp16 split h + l of the Gram build applies: hh' + hl' + lh' on three
// v_mfma_f32_16x16x16_f16 (8 busy cycles each, vector ALU free) instead of four fp32 ones per
// tile, with an error of 2^-22 of the largest diagonal element per product at worst, the same
// size as the rounding of the fp32 accumulation of a full row.  A tile in the accumulator layout
// (lane (g,q) register r = Z[4g+r][q]) IS the f16x4 operand (lane (g,q): k = 4g .. 4g+3 of
// column q) once converted, on either side, and two of them give X^T Y.
struct SplitTile {
    f16x4 h, l;
};
__device__ __forceinline__ SplitTile split_tile(f4 z, f4 s) {
    SplitTile t;
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
        const f32x2 x = f32x2{z[r], z[r + 1]} * f32x2{s[r], s[r + 1]};
        const f16x2 h = __builtin_convertvector(x, f16x2);
        const f16x2 l = split_low(h, z[r], s[r], z[r + 1], s[r + 1]);  // (z s exact in the fma)
        t.h[r] = h[0];
        t.h[r + 1] = h[1];
        t.l[r] = l[0];
        t.l[r + 1] = l[1];
    }
    return t;
}
__device__ __forceinline__ f16x4 negate(f16x4 v) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    u32x2 b = __builtin_bit_cast(u32x2, v);
    b ^= 0x80008000u;
    return __builtin_bit_cast(f16x4, b);
}
template <int CTRL>
__device__ __forceinline__ float dpp_max(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false);
    return fmaxf(v, __builtin_bit_cast(float, t));
}
// The power of two that brings the largest diagonal element (of the real unknowns; `dmax` is
// this lane's share of them) into [2^27, 2^28); 1 when there is no positive finite one.
__device__ __forceinline__ float prescale_factor(float dmax) {
    dmax = dpp_max<0xB1>(dmax);
    dmax = dpp_max<0x4E>(dmax);
    dmax = dpp_max<0x141>(dmax);
    dmax = dpp_max<0x140>(dmax);
    dmax = fmaxf(dmax, __shfl_xor(dmax, 16));
    dmax = fmaxf(dmax, __shfl_xor(dmax, 32));
    const unsigned bits = __builtin_bit_cast(unsigned, dmax);
    const int eb = (int)(bits >> 23) & 0xFF;
    int field = 127 + 27 + 127 - eb;
    field = field < 1 ? 1 : (field > 254 ? 254 : field);
    const bool usable = dmax > 0.0f && eb != 255 && eb != 0;
    return usable ? __builtin_bit_cast(float, (unsigned)field << 23) : 1.0f;
}

// Timing build (-DALSMI_PHASE_CLOCK): wave-clock cycles per phase of the fp32 row kernel, summed
// over all waves into g_phase (read back with alsmi_debug_phase): 0 row header, 1 Gram build,
// 2 error terms, 3 ridge/prescale, 4 elimination steps, 5 split + trailing updates, 6 back
// substitution + store, 7 rows.
#ifdef ALSMI_PHASE_CLOCK
__device__ unsigned long long g_phase[8];
#define PHASE_T(var) const long long var = (long long)__builtin_readcyclecounter()
#define PHASE_ADD(pc, slot, a, b) \
    do {                          \
        if (pc) (pc)[slot] += (b) - (a); \
    } while (0)
#else
#define PHASE_T(var)
#define PHASE_ADD(pc, slot, a, b)
#endif

// 1: the block row right of the pivot block is eliminated as one product with inv(L_kk) on the
// matrix core (PivotBlock below); 0: it takes the sixteen rank-one steps with the pivot block
// (the earlier form, kept for timing comparisons).
#ifndef SOLVE_ZPROD
#define SOLVE_ZPROD 1
#endif

#ifndef DUAL_TR16_MIN
#define DUAL_TR16_MIN 5
#endif

// Lane constants of the block solver.
struct SolveLane {
    int lane, g, q, irow, addr_row, addr_col;
    bool last_col;
#ifdef ALSMI_PHASE_CLOCK
    long long *pc;
#endif
};

// Forward phase of pivot block K (see above); K is a template parameter so that every tile
// index is static — a loop over K that the optimiser declines to unroll would turn the
// accumulators into scratch memory.
// TR16: trailing updates on the fp16 matrix cores (the system has been through the prescale).
template <int NV, int K, bool TR16>
struct PivotBlock {
    static __device__ __forceinline__ void run(Gram<float, NV> &G, const SolveLane &L,
                                               float (&y)[NV], f4 (&rs)[NV], f4 (&zr)[NV],
                                               int &bad, float *stash) {
        const int g = L.g, q = L.q;
        // R starts as the identity.  Its last column stays e_15 whatever the row operations, so
        // it carries the right-hand side instead: the same operations turn y into inv(L) y.
        f4 rt, dv = {1.0f, 1.0f, 1.0f, 1.0f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float yrow = bperm(4 * (4 * g + r), y[K]);  // column -> row layout
            rt[r] = L.last_col ? yrow : ((4 * g + r == q) ? 1.0f : 0.0f);
        }
        PHASE_T(tc0);
#pragma unroll
        for (int p = 0; p < 16; ++p) {
            const int gp = p >> 2, rp = p & 3;
            f4 &a = G.acc[tri_index(K, K, NV)];
            const float src = a[rp];
            const float cola = bperm(L.addr_col + 64 * gp, src);  // A[p][irow] = A[irow][p]
            const float d = read_lane(src, 16 * gp + p);
            const float inv = __builtin_amdgcn_rcpf(d);  // 1 ulp; a multiplier needs no more
            const float m = (L.irow > p) ? -cola * inv : 0.0f;
            // all row fetches of the step first, so that their latencies overlap
            constexpr int JEND = SOLVE_ZPROD ? K + 1 : NV;  // tiles that take the sixteen steps
            float rows[NV + 1];
#pragma unroll
            for (int j = K; j < JEND; ++j)
                rows[j] = bperm(L.addr_row + 64 * gp, G.acc[tri_index(K, j, NV)][rp]);
            rows[NV] = bperm(L.addr_row + 64 * gp, rt[rp]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = K; j < JEND; ++j) {
                f4 &t = G.acc[tri_index(K, j, NV)];
                t = __builtin_amdgcn_mfma_f32_4x4x1f32(m, rows[j], t, 0, 0, 0);
            }
            rt = __builtin_amdgcn_mfma_f32_4x4x1f32(m, rows[NV], rt, 0, 0, 0);
            // (the step boundary stays a scheduling barrier; the wait states a matrix result
            // needs before a cross-lane fetch reads it are the compiler's)
            __builtin_amdgcn_sched_barrier(0);
            dv[rp] = (g == gp) ? inv : dv[rp];
        }
        if constexpr (SOLVE_ZPROD && K + 1 < NV) {
            // The sixteen steps ran on [S_kk | R] alone.  R is now E = inv(L_kk) (its last column,
            // which carried the right-hand side, is e_15 in truth), and the rest of the block
            // row takes all sixteen steps at once as the product Z_J = E S_kJ: two tiles in this
            // layout give X^T Y, so E goes through the matrix core once against the identity
            // (E^T I, exact) and then (E^T)^T S_kJ — four matrix instructions per tile instead
            // of sixteen rank-one updates with a cross-lane fetch each.
            f4 et, ident;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool diag = 4 * g + r == q;
                ident[r] = diag ? 1.0f : 0.0f;
                et[r] = L.last_col ? ((4 * g + r == 15) ? 1.0f : 0.0f) : rt[r];
            }
            const f4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            et = mfma_xty(et, ident, zero);
#pragma unroll
            for (int j = K + 1; j < NV; ++j)
                G.acc[tri_index(K, j, NV)] = mfma_xty(et, G.acc[tri_index(K, j, NV)], zero);
        }
        PHASE_T(tc1);
        PHASE_ADD(L.pc, 4, tc0, tc1);
        // every pivot positive and finite <=> every reciprocal positive and finite
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= (dv[r] > 0.0f && dv[r] < INFINITY) ? 0 : 1;
        f4 dz;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zr[K][r] = bperm(4 * (16 * g + 15), rt[r]);  // z from lane 15 of each row
            dz[r] = dv[r] * zr[K][r];
            const float rtrue = L.last_col ? ((4 * g + r == 15) ? 1.0f : 0.0f) : rt[r];
            rs[K][r] = dv[r] * rtrue;
        }
        if constexpr (!TR16) {
#pragma unroll
            for (int i = K + 1; i < NV; ++i) {
                const f4 zi = G.acc[tri_index(K, i, NV)];
#pragma unroll
                for (int j = i; j < NV; ++j)
                    G.acc[tri_index(i, j, NV)] = mfma_xty(zi, -dv * G.acc[tri_index(K, j, NV)],
                                                          G.acc[tri_index(i, j, NV)]);
            }
        } else if constexpr (K + 1 < NV) {
            f4 sd;
#pragma unroll
            for (int r = 0; r < 4; ++r) sd[r] = __builtin_amdgcn_sqrtf(dv[r]);
            SplitTile zs[NV];
#pragma unroll
            for (int j = K + 1; j < NV; ++j) zs[j] = split_tile(G.acc[tri_index(K, j, NV)], sd);
            split_fence();
            // Two full-depth instructions per tile, [h|l]^T[l'|h'] and [h|l]^T[h'|l'] (all four
            // products of the split; the contraction index doubled by concatenation, as in the
            // Gram build at padded rank 128), product by product over all tiles, so that
            // consecutive matrix instructions do not wait on each other's accumulator.
            f16x8 hl[NV];
#pragma unroll
            for (int j = K + 1; j < NV; ++j)
                hl[j] = __builtin_shufflevector(zs[j].h, zs[j].l, 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int term = 0; term < 2; ++term)
#pragma unroll
                for (int i = K + 1; i < NV; ++i) {
                    const f16x8 a = __builtin_shufflevector(negate(zs[i].h), negate(zs[i].l), 0, 1,
                                                            2, 3, 4, 5, 6, 7);
#pragma unroll
                    for (int j = i; j < NV; ++j) {
                        const f16x8 b = term == 0 ? __builtin_shufflevector(zs[j].l, zs[j].h, 0, 1,
                                                                            2, 3, 4, 5, 6, 7)
                                                  : hl[j];
                        G.acc[tri_index(i, j, NV)] = __builtin_amdgcn_mfma_f32_16x16x32_f16(
                            a, b, G.acc[tri_index(i, j, NV)], 0, 0, 0);
                    }
                }
        }
#pragma unroll
        for (int i = K + 1; i < NV; ++i) {
            const f4 zi = G.acc[tri_index(K, i, NV)];
            float part = 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) part += zi[r] * dz[r];
            y[i] -= col_allreduce(part);
        }
        if (stash) {  // keep the block's back-substitution data in LDS instead of registers
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                stash[((K * 8) + r) * WAVE + L.lane] = rs[K][r];
                stash[((K * 8) + 4 + r) * WAVE + L.lane] = zr[K][r];
            }
        }
        PHASE_T(tc2);
        PHASE_ADD(L.pc, 5, tc1, tc2);
        PivotBlock<NV, K + 1, TR16>::run(G, L, y, rs, zr, bad, stash);
    }
};
template <int NV, bool TR16>
struct PivotBlock<NV, NV, TR16> {
    static __device__ __forceinline__ void run(Gram<float, NV> &, const SolveLane &, float (&)[NV],
                                               f4 (&)[NV], f4 (&)[NV], int &, float *) {}
};

// Back substitution of block K: x_K = inv(L_KK)^T D^-1 (z_K - sum_J Z_KJ x_J), last block first.
template <int NV, int K>
struct BackBlock {
    static __device__ __forceinline__ void run(const Gram<float, NV> &G, const f4 (&rs)[NV],
                                               const f4 (&zr)[NV], float (&xc)[NV], float &chk,
                                               const float *stash, int lane) {
        f4 rhs = zr[K], rsk = rs[K];
        if (stash) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                rsk[r] = stash[((K * 8) + r) * WAVE + lane];
                rhs[r] = stash[((K * 8) + 4 + r) * WAVE + lane];
            }
        }
        if constexpr (K < NV - 1) {
            f4 w = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int j = K + 1; j < NV; ++j) w += G.acc[tri_index(K, j, NV)] * xc[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) rhs[r] -= row_allreduce(w[r]);
        }
        float part = 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) part += rsk[r] * rhs[r];
        xc[K] = col_allreduce(part);
        chk += fabsf(xc[K]);
        if constexpr (K > 0) BackBlock<NV, K - 1>::run(G, rs, zr, xc, chk, stash, lane);
    }
};

// Solves the NV*16 system held in G (ridge not yet added).  xc[k] of lane (.,q) is the solution
// component of dimension q*NV + k.  Returns false when a pivot is not positive or the result is
// not finite.
// DUAL: the unknowns are the neighbours of a short row (block a = neighbours 16a .. 16a+15, the
// first `rank` of them real) instead of the factor dimensions.
template <int NV, bool DUAL = false>
__device__ __forceinline__ bool block_solve(Gram<float, NV> &G, int lane, int rank, float ridge,
                                            float (&xc)[NV], float *stash = nullptr,
                                            long long *pc = nullptr) {
    PHASE_T(ta);
    // The lane-dependent constants below are cheap to derive; hiding the lane id from the
    // optimiser keeps them from being hoisted out of the row loop, where they would occupy
    // registers (or spill slots) throughout the accumulation.
    asm volatile("" : "+v"(lane));
    SolveLane L;
    L.lane = lane;
    L.g = lane >> 4;
    L.q = lane & 15;
    L.irow = 4 * L.g + (L.q & 3);  // the tile row this lane feeds to the 4x4x1 instruction
    L.addr_row = 4 * L.q;          // + 64*gp: lane (gp, q)
    L.addr_col = 4 * L.irow;       // + 64*gp: lane (gp, irow)
    L.last_col = L.q == 15;
#ifdef ALSMI_PHASE_CLOCK
    L.pc = pc;
#endif
    if constexpr (DUAL)
        add_ridge_dual<NV>(G, lane, rank, ridge);
    else
        add_ridge<float, NV>(G, lane, rank, ridge);
    float y[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) y[e] = col_allreduce(G.b[e]);
    // The neighbour systems of short rows keep the fp32 instruction: their solve is a small part
    // of their kernels and the split's registers would cost them a wave per SIMD.
    constexpr bool TR16 = (!DUAL || NV >= DUAL_TR16_MIN) && NV > 1;
    if constexpr (TR16) {
        // largest diagonal element of the real unknowns -> [2^27, 2^28) (see split_tile above)
        float dmax = 0.0f;
#pragma unroll
        for (int e = 0; e < NV; ++e)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool real = DUAL ? (16 * e + L.q < rank) : (L.q * NV + e < rank);
                const bool diag = 4 * L.g + r == L.q;
                dmax = fmaxf(dmax, (diag && real) ? G.acc[tri_index(e, e, NV)][r] : 0.0f);
            }
        const float tau = prescale_factor(dmax);
#pragma unroll
        for (int k = 0; k < Gram<float, NV>::NTRI; ++k) G.acc[k] *= tau;
#pragma unroll
        for (int e = 0; e < NV; ++e) y[e] *= tau;
    }
    // Per pivot block, kept for the back substitution: D^-1 inv(L_kk) (rows scaled) and
    // z_k = inv(L_kk) y_k in the row layout.
    f4 rs[NV], zr[NV];
    int bad = 0;
    PHASE_T(tb);
    PHASE_ADD(pc, 3, ta, tb);
    PivotBlock<NV, 0, TR16>::run(G, L, y, rs, zr, bad, stash);
    PHASE_T(td);
    float chk = 0.0f;
    BackBlock<NV, NV - 1>::run(G, rs, zr, xc, chk, stash, lane);
    const bool all_ok = !__any(bad != 0 || !(chk < INFINITY));
    PHASE_T(te);
    PHASE_ADD(pc, 6, td, te);
    return all_ok;
}

// ---- fp64 block solver (padded rank 128) --------------------------------------------------------
// The same block LDL^T as block_solve above, on the fp64 accumulator tiles (lane (g,q), register
// r = M[g + 4r][q]).  Two differences: an elimination step is four fused multiply-adds per tile
// with the multipliers and the pivot row fetched by ds_bpermute (there is no fp64 counterpart of
// the 4x4x1 rank-one instruction), and the trailing update S_IJ -= Z_I^T D^-1 Z_J runs on
// v_mfma_f64_16x16x4, which — fed two tiles in that layout as both operands, register s at
// step s — again yields X^T Y.  It replaces an in-LDS Cholesky that kept the whole 128 x 128
// matrix in LDS (one wave per CU, about a millisecond per row).
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ double shfl_d(double v, int src_lane) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_ds_bpermute(4 * src_lane, (int)(b & 0xffffffffLL));
    const int hi = __builtin_amdgcn_ds_bpermute(4 * src_lane, (int)(b >> 32));
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_add_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, false);
    return v + __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double row_allreduce_d(double v) {
    v = dpp_add_d<0xB1>(v);
    v = dpp_add_d<0x4E>(v);
    v = dpp_add_d<0x141>(v);
    v = dpp_add_d<0x140>(v);
    return v;
}
// Cross-lane moves of the fp64 solver on the vector ALU instead of ds_bpermute: the four waves of
// a CU share one LDS pipe, and at one wave per SIMD the solver's permutes (two per double) kept
// it busy for most of the solve.
// v of lane P of this lane's row of 16 (DPP row_newbcast)
template <int P>
__device__ __forceinline__ double row_lane_d(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), 0x150 + P, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + P, 0xF, 0xF, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double row_lane_d(double v, int p) {  // p: known after unrolling
    switch (p) {
        case 0: return row_lane_d<0>(v);
        case 1: return row_lane_d<1>(v);
        case 2: return row_lane_d<2>(v);
        case 3: return row_lane_d<3>(v);
        case 4: return row_lane_d<4>(v);
        case 5: return row_lane_d<5>(v);
        case 6: return row_lane_d<6>(v);
        case 7: return row_lane_d<7>(v);
        case 8: return row_lane_d<8>(v);
        case 9: return row_lane_d<9>(v);
        case 10: return row_lane_d<10>(v);
        case 11: return row_lane_d<11>(v);
        case 12: return row_lane_d<12>(v);
        case 13: return row_lane_d<13>(v);
        case 14: return row_lane_d<14>(v);
        default: return row_lane_d<15>(v);
    }
}
// v + v of lane ^ 16 (v_permlane16_swap: row 1 of the first operand changes places with row 0 of
// the second, row 3 with row 2), then + that of lane ^ 32 (v_permlane32_swap: the upper half of
// the first operand with the lower half of the second); the sums in the order of the former
// shuffles, bit for bit
__device__ __forceinline__ double col_allreduce_d(double v) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)b, (unsigned)b, false, false);
        const u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        const double even = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
        const double odd = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
        v = even + odd;
    }
    {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)b, (unsigned)b, false, false);
        const u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)(b >> 32), (unsigned)(b >> 32), false, false);
        const double lower = __builtin_bit_cast(double, ((unsigned long long)hi[0] << 32) | lo[0]);
        const double upper = __builtin_bit_cast(double, ((unsigned long long)hi[1] << 32) | lo[1]);
        v = lower + upper;
    }
    return v;
}
// v of the four lanes with this lane's q (one per lane group), in every one of them
__device__ __forceinline__ void col_gather_d(double v, double (&out)[4]) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned w[2] = {(unsigned)b, (unsigned)(b >> 32)};
    unsigned o[4][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        // rows (0,1,0,1) and rows (2,3,2,3), then each of the four rows in all four
        const u2 s = __builtin_amdgcn_permlane32_swap(w[h], w[h], false, false);
        const u2 e = __builtin_amdgcn_permlane16_swap(s[0], s[0], false, false);
        const u2 f = __builtin_amdgcn_permlane16_swap(s[1], s[1], false, false);
        o[0][h] = e[0];
        o[1][h] = e[1];
        o[2][h] = f[0];
        o[3][h] = f[1];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k)
        out[k] = __builtin_bit_cast(double, ((unsigned long long)o[k][1] << 32) | o[k][0]);
}
// 1/d of a pivot by v_rcp_f64 and three Newton steps (an ulp or two off the rounded quotient,
// a third of the instructions of the division; zero, denormal and non-finite pivots give values
// that the solver's pivot and result checks reject)
__device__ __forceinline__ double rcp_d(double d) {
    double r = __builtin_amdgcn_rcp(d);
#pragma unroll
    for (int it = 0; it < 3; ++it) r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ d4 mfma_xty_d(d4 x, d4 y, d4 c) {
    // (scheduling fences: without them the scaled operands of all the tile updates of a block
    // are formed up front, which costs more registers than there are)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s) c = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s], y[s], c, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    return c;
}

// (the four-pivot form of the elimination, pivot4() below, from this many 16-dimension blocks
// on: below, three waves share a SIMD and hide the cross-lane latency of the pivot-by-pivot form
// for each other, and the registers of the four-pivot form would cost one of them)
#ifndef F64_PIVOT4_MIN_NV
#define F64_PIVOT4_MIN_NV 8
#endif
template <int NV, int K>
struct PivotBlockD {
    // four pivots (rows 4B .. 4B+3 of the block row) at a time, see run()
    template <int B>
    static __device__ __forceinline__ void pivot4(Gram<double, NV> &G, int g, int q, d4 &rt,
                                                  d4 &dv, int &bad) {
            d4 &a = G.acc[tri_index(K, K, NV)];
            double c[4];
            col_gather_d(a[B], c);  // c[k] = A[4b+k][q]
            const double a00 = row_lane_d<4 * B>(c[0]);
            const double a10 = row_lane_d<4 * B>(c[1]), a11 = row_lane_d<4 * B + 1>(c[1]);
            const double a20 = row_lane_d<4 * B>(c[2]), a21 = row_lane_d<4 * B + 1>(c[2]);
            const double a22 = row_lane_d<4 * B + 2>(c[2]);
            const double a30 = row_lane_d<4 * B>(c[3]), a31 = row_lane_d<4 * B + 1>(c[3]);
            const double a32 = row_lane_d<4 * B + 2>(c[3]), a33 = row_lane_d<4 * B + 3>(c[3]);
            const double i0 = rcp_d(a00);
            const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
            const double d1 = fma(-l10, a10, a11);
            const double i1 = rcp_d(d1);
            const double t21 = fma(-l20, a10, a21), t31 = fma(-l30, a10, a31);
            const double l21 = t21 * i1, l31 = t31 * i1;
            const double d2 = fma(-l21, t21, fma(-l20, a20, a22));
            const double i2 = rcp_d(d2);
            const double t32 = fma(-l31, t21, fma(-l30, a20, a32));
            const double l32 = t32 * i2;
            const double d3 = fma(-l32, t32, fma(-l31, t31, fma(-l30, a30, a33)));
            const double i3 = rcp_d(d3);
            const double dmin = fmin(fmin(a00, d1), fmin(d2, d3));
            const double dmax = fmax(fmax(a00, d1), fmax(d2, d3));
            bad |= ((dmin > 0.0) & (dmax < (double)INFINITY)) ? 0 : 1;
            asm volatile("" : "+v"(bad));  // (decided here, not from sixteen pivots kept until the end)
            // A11 x = c for this lane's column q
            const double y0 = c[0];
            const double y1 = fma(-l10, y0, c[1]);
            const double y2 = fma(-l21, y1, fma(-l20, y0, c[2]));
            const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, c[3])));
            const double x3 = y3 * i3;
            const double x2 = fma(-l32, x3, y2 * i2);
            const double x1 = fma(-l31, x3, fma(-l21, x2, y1 * i1));
            const double x0 = fma(-l30, x3, fma(-l20, x2, fma(-l10, x1, y0 * i0)));
            const double xg = g == 0 ? x0 : (g == 1 ? x1 : (g == 2 ? x2 : x3));
            // inv(L) - I, row q - 4b, column g
            const double m10 = -l10, m21 = -l21, m32 = -l32;
            const double m20 = -fma(l21, m10, l20), m31 = -fma(l32, m21, l31);
            const double m30 = -fma(l32, m20, fma(l31, m10, l30));
            double w1 = g == 0 ? m10 : 0.0;
            double w2 = g == 0 ? m20 : (g == 1 ? m21 : 0.0);
            double w3 = g == 0 ? m30 : (g == 1 ? m31 : (g == 2 ? m32 : 0.0));
            // (both kinds of multiplier exist in every lane before the choice: left to itself the
            // compiler turns the choice into branches on the lane's q around their computation)
            double xs = -xg;
            asm volatile("" : "+v"(xs), "+v"(w1), "+v"(w2), "+v"(w3));
            const int k = q - 4 * B;
            // (one choice per candidate, joined by OR: a chain of choices on k becomes a switch)
            auto pick = [](bool on, double v) {
                return on ? __builtin_bit_cast(unsigned long long, v) : 0ull;
            };
            const double w = __builtin_bit_cast(
                double, pick(k > 3, xs) | pick(k == 1, w1) | pick(k == 2, w2) | pick(k == 3, w3));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = K; j < NV; ++j) {
                d4 &t = G.acc[tri_index(K, j, NV)];
                t = __builtin_amdgcn_mfma_f64_16x16x4f64(w, t[B], t, 0, 0, 0);
            }
            rt = __builtin_amdgcn_mfma_f64_16x16x4f64(w, rt[B], rt, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            dv[B] = g == 0 ? i0 : (g == 1 ? i1 : (g == 2 ? i2 : i3));
    }
    static __device__ __forceinline__ void run(Gram<double, NV> &G, int lane, double (&y)[NV],
                                               int &bad, double *stash, long long *pc) {
        PHASE_T(tc0);
        const int g = lane >> 4, q = lane & 15;
        const bool last_col = q == 15;
        // R starts as the identity; its last column, which no row operation changes, carries the
        // block's right-hand side (see PivotBlock)
        d4 rt, dv = {1.0, 1.0, 1.0, 1.0};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double yrow = shfl_d(y[K], g + 4 * r);  // column -> row layout
            rt[r] = last_col ? yrow : ((g + 4 * r == q) ? 1.0 : 0.0);
        }
        if constexpr (NV >= F64_PIVOT4_MIN_NV) {
            // Four pivots at a time, the row operations on the matrix cores.  Rows 4b .. 4b+3 of
            // the block row (register b of lane group 0..3) are, as they stand, the B operand of
            // v_mfma_f64_16x16x4 (lane (g,q): row 4b+g, column q), so T += W R with a 16 x 4
            // matrix W of multipliers in the A operand (lane (g,q): W[q][g]) applies four
            // elimination steps to a tile in one instruction and without moving a row between
            // lanes.  With A11 = L D L^T the 4 x 4 pivot block, W is
            //   rows below it     -A21 inv(A11): row q solves A11 x = -(column q of the four pivot
            //                     rows), which by symmetry are A21's entries, via L and D
            //   rows 4b .. 4b+3   inv(L) - I (what the four steps do to the pivot rows themselves)
            //   rows above        0.
            // Every lane factors A11 for itself (ten values fetched from its row of 16 by DPP) and
            // solves for its own q; the only traffic between lane groups is the gather of the
            // diagonal tile's four pivot-row values per lane (permlane swaps), none through LDS.
            pivot4<0>(G, g, q, rt, dv, bad);
            pivot4<1>(G, g, q, rt, dv, bad);
            pivot4<2>(G, g, q, rt, dv, bad);
            pivot4<3>(G, g, q, rt, dv, bad);
        } else {
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int gp = p & 3, rp = p >> 2;  // row p: lane group gp, register rp
                d4 &a = G.acc[tri_index(K, K, NV)];
                const double d = read_lane(a[rp], 16 * gp + p);
                const double inv = 1.0 / d;
                double m[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // rows at or above the pivot (compile-time known for most r) keep a zero
                    if (4 * r + 3 <= p) {
                        m[r] = 0.0;
                    } else {
                        const double colv = row_lane_d(a[r], p);  // A[g + 4r][p]
                        m[r] = (g + 4 * r > p) ? -colv * inv : 0.0;
                    }
                }
                double rows[NV + 1];
#pragma unroll
                for (int j = K; j < NV; ++j) rows[j] = shfl_d(G.acc[tri_index(K, j, NV)][rp], 16 * gp + q);
                rows[NV] = shfl_d(rt[rp], 16 * gp + q);
#pragma unroll
                for (int j = K; j < NV; ++j) {
                    d4 &t = G.acc[tri_index(K, j, NV)];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (4 * r + 3 > p) t[r] = fma(m[r], rows[j], t[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * r + 3 > p) rt[r] = fma(m[r], rows[NV], rt[r]);
                dv[rp] = (g == gp) ? inv : dv[rp];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bad |= (dv[r] > 0.0 && dv[r] < (double)INFINITY) ? 0 : 1;
#ifdef ALSMI_PHASE_CLOCK
        asm volatile("" ::"v"(rt[0]), "v"(dv[0]), "v"(bad));
#endif
        PHASE_T(tc1);
        PHASE_ADD(pc, 4, tc0, tc1);
        d4 dz, zr, rs;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            zr[r] = row_lane_d<15>(rt[r]);  // z from lane 15 of each lane group
            dz[r] = dv[r] * zr[r];
            const double rtrue = last_col ? ((g + 4 * r == 15) ? 1.0 : 0.0) : rt[r];
            rs[r] = dv[r] * rtrue;
        }
        if constexpr (NV >= F64_PIVOT4_MIN_NV) {
            // (one wave per SIMD: registers for the scaled block row once per block, and the
            // four dependent instructions of a tile spread over all tiles of the trailing part,
            // so that none waits for its predecessor)
            d4 ys[NV];
#pragma unroll
            for (int j = K + 1; j < NV; ++j) ys[j] = -dv * G.acc[tri_index(K, j, NV)];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = K + 1; i < NV; ++i)
#pragma unroll
                    for (int j = i; j < NV; ++j)
                        G.acc[tri_index(i, j, NV)] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            G.acc[tri_index(K, i, NV)][s], ys[j][s], G.acc[tri_index(i, j, NV)], 0,
                            0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = K + 1; i < NV; ++i) {
                const d4 zi = G.acc[tri_index(K, i, NV)];
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) part += zi[r] * dz[r];
                y[i] -= col_allreduce_d(part);
            }
        } else {
#pragma unroll
            for (int i = K + 1; i < NV; ++i) {
                const d4 zi = G.acc[tri_index(K, i, NV)];
                // two tiles at a time, their instructions alternating: four in a row on one
                // accumulator each wait for the one before
#pragma unroll
                for (int j = i; j < NV; j += 2) {
                    if (j + 1 < NV) {
                        const d4 ya = -dv * G.acc[tri_index(K, j, NV)];
                        const d4 yb = -dv * G.acc[tri_index(K, j + 1, NV)];
                        d4 &ca = G.acc[tri_index(i, j, NV)];
                        d4 &cb = G.acc[tri_index(i, j + 1, NV)];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s = 0; s < 4; ++s) {
                            ca = __builtin_amdgcn_mfma_f64_16x16x4f64(zi[s], ya[s], ca, 0, 0, 0);
                            cb = __builtin_amdgcn_mfma_f64_16x16x4f64(zi[s], yb[s], cb, 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                        G.acc[tri_index(i, j, NV)] = mfma_xty_d(zi, -dv * G.acc[tri_index(K, j, NV)],
                                                                G.acc[tri_index(i, j, NV)]);
                    }
                }
                double part = 0.0;
#pragma unroll
                for (int r = 0; r < 4; ++r) part += zi[r] * dz[r];
                y[i] -= col_allreduce_d(part);
            }
        }
        // the block's back-substitution data waits in LDS
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            stash[((K * 8) + r) * WAVE + lane] = rs[r];
            stash[((K * 8) + 4 + r) * WAVE + lane] = zr[r];
        }
#ifdef ALSMI_PHASE_CLOCK
        asm volatile("" ::"v"(y[NV - 1]), "v"(G.acc[Gram<double, NV>::NTRI - 1][0]));
#endif
        PHASE_T(tc2);
        PHASE_ADD(pc, 5, tc1, tc2);
        PivotBlockD<NV, K + 1>::run(G, lane, y, bad, stash, pc);
    }
};
template <int NV>
struct PivotBlockD<NV, NV> {
    static __device__ __forceinline__ void run(Gram<double, NV> &, int, double (&)[NV], int &,
                                               double *, long long *) {}
};

template <int NV, int K>
struct BackBlockD {
    static __device__ __forceinline__ void run(const Gram<double, NV> &G, double (&xc)[NV],
                                               double &chk, const double *stash, int lane) {
        d4 rhs, rsk;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rsk[r] = stash[((K * 8) + r) * WAVE + lane];
            rhs[r] = stash[((K * 8) + 4 + r) * WAVE + lane];
        }
        if constexpr (K < NV - 1) {
            d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int j = K + 1; j < NV; ++j) w += G.acc[tri_index(K, j, NV)] * xc[j];
#pragma unroll
            for (int r = 0; r < 4; ++r) rhs[r] -= row_allreduce_d(w[r]);
        }
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) part += rsk[r] * rhs[r];
        xc[K] = col_allreduce_d(part);
        chk += fabs(xc[K]);
        if constexpr (K > 0) BackBlockD<NV, K - 1>::run(G, xc, chk, stash, lane);
    }
};

// Solves the NV*16 system held in G (ridge not yet added); xc[k] of lane (.,q) is the solution
// component of dimension q*NV + k.  False when a pivot is not positive or the result not finite.
// DUAL: the unknowns are the neighbours of a short row (block a = neighbours 16a .. 16a+15, the
// first `rank` of them real), see solve_rows_dual_d_kernel.
template <int NV, bool DUAL = false>
__device__ __forceinline__ bool block_solve_d(Gram<double, NV> &G, int lane, int rank, double ridge,
                                              double (&xc)[NV], double *stash,
                                              long long *pc = nullptr) {
    asm volatile("" : "+v"(lane));
    PHASE_T(ta);
    if constexpr (DUAL)
        add_ridge_dual_d<NV>(G, lane, rank, ridge);
    else
        add_ridge<double, NV>(G, lane, rank, ridge);
    double y[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) y[e] = col_allreduce_d(G.b[e]);
    int bad = 0;
#ifdef ALSMI_PHASE_CLOCK
    asm volatile("" ::"v"(y[NV - 1]));
#endif
    PHASE_T(tb);
    PHASE_ADD(pc, 3, ta, tb);
    PivotBlockD<NV, 0>::run(G, lane, y, bad, stash, pc);
    PHASE_T(td);
    double chk = 0.0;
    BackBlockD<NV, NV - 1>::run(G, xc, chk, stash, lane);
#ifdef ALSMI_PHASE_CLOCK
    asm volatile("" ::"v"(xc[0]), "v"(chk));
#endif
    PHASE_T(te);
    PHASE_ADD(pc, 6, td, te);
    return !__any(bad != 0 || !(chk < (double)INFINITY));
}

// Generic in-LDS Cholesky + solve for any KP (two rows per lane at KP = 128).  Works on the
// upper triangle stored row-major so that every vector access is contiguous over lanes.
template <typename T, int KP>
__device__ __forceinline__ bool chol_solve_lds(T *Asm, T *bsm, T *dsm, int lane, T (&x_out)[(KP + 63) / 64]) {
    constexpr int LD = Slab<T, KP>::LD;
    constexpr int R = (KP + 63) / 64;
    bool ok = true;
    for (int p = 0; p < KP; ++p) {
        const T d = Asm[p * LD + p];
        ok = ok && (d > T(0));
        const T rs = T(1) / sqrt(d);
        const T yp = bsm[p] * rs;
        T lp[R];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int row = lane + 64 * rr;
            lp[rr] = T(0);
            if (row >= p && row < KP) {
                lp[rr] = Asm[p * LD + row] * rs;  // L[row][p], kept at (p,row)
                Asm[p * LD + row] = lp[rr];
                if (row == p) {
                    bsm[p] = yp;
                    dsm[p] = rs;
                } else {
                    bsm[row] -= lp[rr] * yp;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int c = p + 1; c < KP; ++c) {
            const T lc = Asm[p * LD + c];
#pragma unroll
            for (int rr = 0; rr < R; ++rr) {
                const int row = lane + 64 * rr;
                if (row >= c && row < KP) Asm[c * LD + row] -= lp[rr] * lc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    for (int p = KP - 1; p >= 0; --p) {
        const T xp = bsm[p] * dsm[p];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const int row = lane + 64 * rr;
            if (row == p) bsm[p] = xp;
            if (row < p) bsm[row] -= Asm[row * LD + p] * xp;  // L[p][row] kept at (row,p)
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
        const int row = lane + 64 * rr;
        x_out[rr] = row < KP ? bsm[row] : T(0);
    }
    return ok;
}

template <typename T>
struct SolveArgs {
    const int32_t *order;       // [n_tasks] row ids, longest first
    const int32_t *slab_first;  // [n_tasks] first workspace slab of the row, -1 = short row
    const int32_t *slab_count;  // [n_tasks]
    int t_begin, t_end;  // the kernel's share of the degree-sorted task list
    const int64_t *indptr;
    const int32_t *indices;
    const T *vals;
    const T *src;
    const T *slabs;
    const unsigned *max_bits;  // bit pattern of max |src| (scaled fp16 Gram build only)
    T *dst;
    int rank;
    T alpha;
    // debug dump (normal equations instead of the solve)
    T *A_out;
    T *b_out;
    int64_t row_begin;
    int solve_prio;  // raise the wave's priority for the solve: bit 0 in the ld x ld kernel
                     // (plans of short rows), bit 1 in the short-row kernels
    // squared error of the rows' current vectors (ERR builds only)
    const T *cur;      // table the rows' current vectors are read from (may be dst)
    double *err_part;  // [waves of the grid]
};

// Rows of at most this many entries are solved in the neighbour-by-neighbour ("dual") form
// when the padded rank is larger (fp32: up to 96 entries at padded rank 128, up to 48 at 64),
// in classes of 16: class k holds the rows of at most DUAL_MAX_DEG - 16 k entries.
#ifndef DUAL_MAX_DEG_128
#define DUAL_MAX_DEG_128 96
#endif
constexpr int DUAL_MAX_DEG = DUAL_MAX_DEG_128;
constexpr int DUAL_CLASSES = DUAL_MAX_DEG / 16;

// LDS elements one wave needs: the small transposition stage on the fast path (padded rank
// <= 64, no dump), the whole matrix otherwise.
// fp64: smallest padded rank solved by the block LDL^T on the accumulator tiles (block_solve_d)
// instead of the Gauss-Jordan elimination with one matrix row per lane.
#ifndef F64_BLOCK_MIN_KP
#define F64_BLOCK_MIN_KP 64
#endif
template <typename T, int KP, bool DUMP>
struct WaveLds {
    static constexpr bool FAST = !DUMP && KP <= 128;
    // work area: block-solver paths (fp32; fp64 at padded rank 128) = the row's current vector
    // (error variant) only; fp64 Gauss-Jordan path = the transposition stage; otherwise (the
    // dump) the whole matrix
    static constexpr bool BLOCK = sizeof(T) == 4 || KP >= F64_BLOCK_MIN_KP;
    static constexpr int WORK = FAST ? (BLOCK ? KP : Stage<T, KP < 64 ? KP : 64>::ELEMS)
                                     : Slab<T, KP>::ELEMS;
    // + 64 doubles behind the work area: the lanes' squared-error partial sums
    static constexpr int ERR_AT = (WORK + 1) / 2 * 2;
    // + (padded rank 128) the back-substitution data of the eight pivot blocks
    static constexpr int STASH_AT = ERR_AT + 512 / (int)sizeof(T);
    // (fp64: from padded rank F64_BLOCK_MIN_KP on, one 8 x 64 slab per pivot block)
    static constexpr int STASH_BLOCKS =
        DUMP ? 0 : (KP == 128 ? 8 : ((sizeof(T) == 8 && KP >= F64_BLOCK_MIN_KP) ? KP / 16 : 0));
    static constexpr int ELEMS = STASH_AT + STASH_BLOCKS * 8 * WAVE;
    static constexpr size_t BYTES = (size_t)ELEMS * sizeof(T);
};

// ---- kernel 2: one wave per row — accumulate (or take the row's summed slab), solve ---------
// ABLATE selects a diagnostic build used only for timing experiments (env ALSMI_ABLATE, f32
// rank 64): 1 = accumulate but skip the solve, 2 = solve but skip the accumulation, 4 = Gram
// build on the fp32 MFMA, 10 = on the bf16 matrix cores with the exact three-way split (both
// kernels).  The production kernel is the ABLATE = 0 instantiation and contains none of that code.
// ABLATE = 20 is not a diagnostic: it selects 64-bit gather addresses (padded rank 128, source
// tables above 4 GiB).
// ERR builds also sum (rating - <current vector, neighbour>)^2 over every entry they read: the
// training error of the factors BEFORE this half-sweep, at no extra memory traffic.
template <typename T, int NV, int WPB, bool DUMP, int ABLATE = 0, bool ERR = false>
__global__ __launch_bounds__((WPB * 64), (NV == 4 && !DUMP ? (sizeof(T) == 4 ? SOLVE_MIN_WAVES : 2) : (NV == 8 && !DUMP && sizeof(T) == 4 ? R128_MIN_WAVES : 1))) void solve_rows_kernel(
    SolveArgs<T> p) {
    constexpr int KP = NV * 16;
    constexpr int LD = Slab<T, KP>::LD;
    constexpr int R = (KP + 63) / 64;
    constexpr bool FAST = WaveLds<T, KP, DUMP>::FAST;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    T *Asm = reinterpret_cast<T *>(smem) + (size_t)wl * WaveLds<T, KP, DUMP>::ELEMS;
    T *bsm = Asm + KP * (FAST ? Stage<T, KP < 64 ? KP : 64>::LD : LD);
    T *dsm = bsm + KP;
    const int W = gridDim.x * WPB;
    constexpr int MODE = gram_mode<T, NV, ABLATE, DUMP>();
    float scale = 1.0f;
    if constexpr (is_f16x2(MODE)) scale = split_scale(p.max_bits[0]);
    const T s2 = (T)(scale * scale), inv_s2 = (T)(1.0f / (scale * scale));
    // This lane's share of the squared error of the wave's rows lives in the wave's LDS area
    // (64 doubles behind the work area), not in two registers across the gather.
    double *lane_err_slot =
        reinterpret_cast<double *>(Asm + WaveLds<T, KP, DUMP>::ERR_AT) + lane;
    if constexpr (ERR) *lane_err_slot = 0.0;
#ifdef ALSMI_PHASE_CLOCK
    long long pcs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long *pc = pcs;
#else
    long long *pc = nullptr;
#endif
    for (int t = p.t_begin + blockIdx.x * WPB + wl; t < p.t_end; t += W) {
        PHASE_T(t0);
        const int row = p.order[t];
        const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
        T *out = p.dst + (int64_t)row * KP;
        if (end <= beg) {
            if constexpr (DUMP) {
                T *Ao = p.A_out + (row - p.row_begin) * (int64_t)KP * KP;
                for (int i = lane; i < KP * KP; i += WAVE) Ao[i] = T(0);
                for (int i = lane; i < KP; i += WAVE) p.b_out[(row - p.row_begin) * KP + i] = T(0);
            } else {
#pragma unroll
                for (int rr = 0; rr < R; ++rr)
                    if (lane + 64 * rr < KP) out[lane + 64 * rr] = T(0);
            }
            continue;
        }
        if constexpr (ERR) {
            // the row's current vector goes to the wave's LDS area now, so that the error
            // terms after the accumulation do not wait on memory
            const T *cur_row = p.cur + (int64_t)row * KP;
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
                if (lane + 64 * rr < KP) Asm[lane + 64 * rr] = cur_row[lane + 64 * rr];
        }
        Gram<T, NV> G;
        G.zero();
        const int sf = p.slab_first[t];
        ErrAcc<T, NV, ERR> E;
        E.init();
#ifdef ALSMI_PHASE_CLOCK
        asm volatile("" ::"s"(sf), "s"(beg), "s"(end));  // the header's loads end here
#endif
        PHASE_T(t1);
        PHASE_ADD(pc, 0, t0, t1);
        if (sf >= 0) {
            // a long row: sum_slabs_kernel left the sum of its chunks in its first slab
            G.add(p.slabs + (size_t)sf * (Gram<T, NV>::SLAB_SLOTS * WAVE), lane, E.sum);
        } else if constexpr (ABLATE != 2) {
            gram_accumulate<T, NV, MODE, ERR>(G, E, p.indices, p.vals, p.src, beg, end, lane,
                                              scale);
        }
        PHASE_T(t2);
        PHASE_ADD(pc, 1, t1, t2);
        if constexpr (ERR) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            *lane_err_slot += row_error_part<T, NV>(G, E.sum, Asm, lane, inv_s2);
        }
        PHASE_T(t3);
        PHASE_ADD(pc, 2, t2, t3);
        if constexpr (is_f16x2(MODE) && DUMP) {  // show the unscaled system (exact: S^2 = 2^k)
#pragma unroll
            for (int k = 0; k < Gram<T, NV>::NTRI; ++k) G.acc[k] *= inv_s2;
        } else if constexpr (is_f16x2(MODE)) {  // (S^2 A) x = S^2 b
#pragma unroll
            for (int e = 0; e < NV; ++e) G.b[e] *= s2;
        }
        if constexpr (ABLATE == 1) {
            T sum = G.b[0];
#pragma unroll
            for (int k = 0; k < Gram<T, NV>::NTRI; ++k)
                sum += G.acc[k][0] + G.acc[k][1] + G.acc[k][2] + G.acc[k][3];
            if (lane < KP) out[lane] = sum;
            continue;
        }
        const T ridge = p.alpha * (T)(end - beg) * (DUMP ? T(1) : s2);
        if constexpr (FAST && sizeof(T) == 4) {
            float xs[NV];
            float *stash = nullptr;
            if constexpr (KP == 128) stash = reinterpret_cast<float *>(Asm) + WaveLds<T, KP, DUMP>::STASH_AT;
            // Waves in their solve issue ahead of waves in their Gram build where rows are short
            // (p.solve_prio, set by the host from the plan's mean row length): the solve is a
            // chain of dependent steps, the Gram build waits on memory most of the time.
            // Measured at 138 493 x 26 744, rank 64: user half -4 %; on the item half, whose
            // long rows are bound by their gather, the same costs 2-3 %.
            if (p.solve_prio & 1) __builtin_amdgcn_s_setprio(1);
            const bool ok = block_solve<NV>(G, lane, p.rank, ridge, xs, stash, pc);
            if (p.solve_prio & 1) __builtin_amdgcn_s_setprio(0);
            if (lane < 16) {
                typedef float vec_t __attribute__((ext_vector_type(NV)));
                vec_t v;
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    if constexpr (NV == 1) v = ok ? xs[k] : 0.0f;
                    else v[k] = ok ? xs[k] : 0.0f;
                }
                *reinterpret_cast<vec_t *>(out + lane * NV) = v;
            }
#ifdef ALSMI_PHASE_CLOCK
            pcs[7] += 1;
#endif
            continue;
        } else if constexpr (FAST && KP >= F64_BLOCK_MIN_KP) {
            double xs[NV];
            if (p.solve_prio & 1) __builtin_amdgcn_s_setprio(1);
            const bool ok = block_solve_d<NV>(G, lane, p.rank, ridge, xs,
                                              reinterpret_cast<double *>(Asm) +
                                                  WaveLds<T, KP, DUMP>::STASH_AT,
                                              pc);
#ifdef ALSMI_PHASE_CLOCK
            pcs[7] += 1;
#endif
            if (p.solve_prio & 1) __builtin_amdgcn_s_setprio(0);
            if (lane < 16) {
#pragma unroll
                for (int k = 0; k < NV; ++k) out[lane * NV + k] = ok ? xs[k] : 0.0;
            }
            continue;
        } else if constexpr (FAST) {
            T x;
            const bool ok = gj_solve<T, NV>(G, Asm, bsm, lane, p.rank, ridge, x);
            if (lane < KP) out[lane] = ok ? x : T(0);
            continue;
        }
        assemble<T, NV>(G, Asm, bsm, lane, p.rank, ridge);
        if constexpr (DUMP) {
            T *Ao = p.A_out + (row - p.row_begin) * (int64_t)KP * KP;
            for (int i = lane; i < KP * KP; i += WAVE) Ao[i] = Asm[(i / KP) * LD + (i % KP)];
            for (int i = lane; i < KP; i += WAVE) p.b_out[(row - p.row_begin) * KP + i] = bsm[i];
        } else {
            T x[R];
            const bool ok = chol_solve_lds<T, KP>(Asm, bsm, dsm, lane, x);
#pragma unroll
            for (int rr = 0; rr < R; ++rr)
                if (lane + 64 * rr < KP) out[lane + 64 * rr] = ok ? x[rr] : T(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if constexpr (ERR) {
        const double wave_err = wave_sum(*lane_err_slot);
        if (lane == 0) p.err_part[blockIdx.x * WPB + wl] = wave_err;
    }
#ifdef ALSMI_PHASE_CLOCK
    if (lane == 0)
        for (int i = 0; i < 8; ++i) atomicAdd(&g_phase[i], (unsigned long long)pcs[i]);
#endif
}

// ---- kernel 2b: short rows in the neighbour-by-neighbour ("dual") form ---------------------
// A row with n stored entries, n well below the padded rank, has
//     x = (F^T F + c I)^-1 F^T r = F^T (F F^T + c I)^-1 r,     F = [n x KP] its neighbours' rows,
// so the n x n matrix K = F F^T replaces the KP x KP one: at padded rank 128 and n <= 64 the
// factorisation is 8 times cheaper (64 times at n <= 32) and — unlike F^T F — K is a product
// whose contraction runs along the factor dimension, the direction that is contiguous in
// memory: a lane's 32-byte load (8 dimensions of one neighbour) is one matrix-core operand
// as it stands (lane (g,q): neighbour 16a+q, dimensions 32c+8g .. +7 of chunk c), nothing is
// transposed.  The products again go through the scaled two-way fp16 split.  Afterwards
// w = (K + c I)^-1 r is formed by the same block solver (unknowns = neighbours; rows of K
// beyond n are zero, get a unit diagonal and a zero right-hand side) and x = F^T w re-reads
// the neighbours' rows, which the wave has just touched.
// ERR: the squared error of the row's current vector needs <u, f_t> per neighbour, which is
// a by-product of the second pass here (the fp32 rows are in registers and the accumulator
// tiles are dead), so it is summed directly instead of through the quadratic form the
// KP x KP kernel uses.
// Waves per SIMD the dual kernels are compiled for: all loads of a row are in flight at once,
// which at 64 neighbours needs the registers of two waves per SIMD.
#ifndef DUAL_ABLATE
#define DUAL_ABLATE 0  // timing builds only: 1 no solve, 2 no x = F^T w pass, 4 no error pass
#endif
#ifndef DUAL_KEEP_MAX
#define DUAL_KEEP_MAX 2  // largest ND whose neighbour rows stay in registers for the whole row
#endif
#ifndef DUAL64_MIN_WAVES
#define DUAL64_MIN_WAVES 2
#endif
#ifndef DUAL32_MIN_WAVES
#define DUAL32_MIN_WAVES 4
#endif
// Do the neighbours' rows of a short row stay in registers for the whole row (32 ND registers at
// padded rank 128, 16 ND at 64)?  Up to 32 entries always; 33 to 48 entries at padded rank 64,
// and at 128 in the error variant, which would otherwise read them three times (measured: with
// two reads only, the wave per SIMD this costs outweighs the second read).
constexpr bool dual_keep(int nv, int nd, bool err) {
    return nd <= DUAL_KEEP_MAX || (nd == 3 && (nv == 4 || err));
}
constexpr int dual_min_waves(int nv, int nd, bool err) {
    const bool keep = dual_keep(nv, nd, err);
    if (nd > 3) return DUAL64_MIN_WAVES;
    if (nd == 3) return keep && nv == 8 ? 2 : 3;
    if (nd == 2 && keep && err && nv == 8) return 3;
    return DUAL32_MIN_WAVES;
}
template <int NV, int ND, bool ERR>
__global__ __launch_bounds__(256, dual_min_waves(NV, ND, ERR)) void solve_rows_dual_kernel(SolveArgs<float> p) {
    constexpr int KP = NV * 16;
    constexpr int NC = KP / 32;  // 32-dimension chunks, one matrix instruction deep each
    constexpr bool KEEP = dual_keep(NV, ND, ERR);
    static_assert(KP % 32 == 0 && ND * 16 <= 2 * WAVE, "at most two entries per lane");
    constexpr bool TWO = ND > 4;  // rows of more than 64 entries: a second entry per lane
    const int lane = threadIdx.x & 63;
    const int wl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = gridDim.x * 4;
    const float scale = split_scale(p.max_bits[0]);
    const float s2 = scale * scale;
    double err_acc = 0.0;
    for (int t = p.t_begin + blockIdx.x * 4 + wl; t < p.t_end; t += W) {
        const int row = p.order[t];
        const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
        const int n = (int)(end - beg);
        float *out = p.dst + (int64_t)row * KP;
        if (n <= 0) {
#pragma unroll
            for (int rr = 0; rr < KP / 64; ++rr) out[lane + 64 * rr] = 0.0f;
            continue;
        }
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));  // lane constants derived per row, not kept across rows
        const int g = lane_v >> 4, q = lane_v & 15;
        // lanes past the row's end repeat its first entry (their rows of F are zeroed below; the
        // address stays one the row really reads)
        const int64_t at = beg + (lane_v < n ? lane_v : 0);
        const int idx = p.indices[at];
        const float rv = lane_v < n ? p.vals[at] : 0.0f;
        int idx_hi = 0;
        float rv_hi = 0.0f;
        if constexpr (TWO) {
            const int64_t at_hi = beg + (lane_v + WAVE < n ? lane_v + WAVE : 0);
            idx_hi = p.indices[at_hi];
            rv_hi = lane_v + WAVE < n ? p.vals[at_hi] : 0.0f;
        }
        int id[ND];
        bool real[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            id[a] = __builtin_amdgcn_ds_bpermute(4 * (16 * (a & 3) + q), a < 4 ? idx : idx_hi);
            real[a] = 16 * a + q < n;
        }
        const float *srcg = p.src + 8 * g;
        Gram<float, ND> G;
        G.zero();
        float pred[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) pred[a] = 0.0f;
        // KEEP: the neighbours' rows stay in registers from here to the end of the row (32 ND
        // registers), so the two later passes over them cost no loads; otherwise they are
        // re-read (cache hits) before each pass.
        float keep[KEEP ? NC : 1][KEEP ? ND : 1][8];
        if constexpr (KEEP) {
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int a = 0; a < ND; ++a)
                    load_vec<float, 8>(keep[c][a], srcg + ((int64_t)id[a] * KP + 32 * c));
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float vl[KEEP ? 1 : ND][8];
            if constexpr (!KEEP) {
#pragma unroll
                for (int a = 0; a < ND; ++a)
                    load_vec<float, 8>(vl[a], srcg + ((int64_t)id[a] * KP + 32 * c));
            }
            auto &v = [&]() -> float(&)[ND][8] {
                if constexpr (KEEP) return keep[c]; else return vl;
            }();
            if constexpr (ERR && KEEP) {
                // <current vector, neighbour> rides along (the current vector may live in the
                // table being written: it is read before the row's store)
                float uc[8];
                load_vec<float, 8>(uc, p.cur + ((int64_t)row * KP + 32 * c) + 8 * g);
#pragma unroll
                for (int a = 0; a < ND; ++a)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pred[a] += v[a][j] * uc[j];
            }
            f16x8 H[ND], L[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
                // instantiation ND only sees rows of more than 16 (ND - 1) entries (shorter ones
                // go to a smaller one), so all its tiles but the last are full
                const bool maybe_partial = a == ND - 1;
#pragma unroll
                for (int j = 0; j < 8; j += 2) {
                    f32x2 u = f32x2{v[a][j], v[a][j + 1]};
                    if (maybe_partial) u = real[a] ? u : f32x2{0.0f, 0.0f};
                    const f32x2 x = u * scale;
                    const f16x2 h = __builtin_convertvector(x, f16x2);
                    const f16x2 l = split_low(h, u[0], scale, u[1], scale);
                    H[a][j] = h[0];
                    H[a][j + 1] = h[1];
                    L[a][j] = l[0];
                    L[a][j + 1] = l[1];
                }
            }
            split_fence();
#pragma unroll
            for (int a = 0; a < ND; ++a)
#pragma unroll
                for (int b = a; b < ND; ++b) {
                    Mfma<float>::acc_t acc = G.acc[tri_index(a, b, ND)];
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(H[a], L[b], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(L[a], H[b], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(H[a], H[b], acc, 0, 0, 0);
                    G.acc[tri_index(a, b, ND)] = acc;
                }
        }
        // right-hand side S^2 r, one neighbour per column; block_solve adds the four lane groups
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            const float ra = bperm(4 * (16 * (a & 3) + q), a < 4 ? rv : rv_hi);
            G.b[a] = g == 0 ? ra * s2 : 0.0f;
        }
        float w[ND];
#if DUAL_ABLATE & 1
        const bool ok = true;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            float t = G.b[a];
#pragma unroll
            for (int b = a; b < ND; ++b)
                for (int r = 0; r < 4; ++r) t += G.acc[tri_index(a, b, ND)][r];
            w[a] = t;
        }
#else
        if (p.solve_prio & 2) __builtin_amdgcn_s_setprio(1);
        const bool ok = block_solve<ND, true>(G, lane, n, p.alpha * (float)n * s2, w);
        if (p.solve_prio & 2) __builtin_amdgcn_s_setprio(0);
#endif
        // x = F^T w: this lane's neighbours again (a lane past the row's end re-reads the
        // row's first neighbour with a zero weight), 8 dimensions of each chunk; then the sum
        // over the 16 lanes of a lane group, as a reduce-scatter: after the exchanges over
        // lane bits 3 and 2 a lane keeps one chunk, after bits 1 and 0 two dimensions of it.
        float xp[NC][8];
        // (without KEEP the optimiser must not keep the first pass's loaded registers alive
        // through the solve in order to skip these loads)
        if constexpr (!KEEP) {
#pragma unroll
            for (int a = 0; a < ND; ++a) asm volatile("" : "+v"(id[a]));
        }
#if DUAL_ABLATE & 2
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int a = 0; a < ND; ++a) acc += (float)(c + j + 1) * (ok ? w[a] : 0.0f);
                xp[c][j] = acc;
            }
#else
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float vl[KEEP ? 1 : ND][8];
            if constexpr (!KEEP) {
#pragma unroll
                for (int a = 0; a < ND; ++a)
                    load_vec<float, 8>(vl[a], srcg + ((int64_t)id[a] * KP + 32 * c));
            }
            auto &v = [&]() -> float(&)[ND][8] {
                if constexpr (KEEP) return keep[c]; else return vl;
            }();
            if constexpr (ERR && !KEEP && !(DUAL_ABLATE & 4)) {
                // <current vector, neighbour> rides along with this pass (in the first one it
                // would cost a wave per SIMD in registers, beside the accumulator tiles).  The
                // current vector may live in the table being written: the row's store comes
                // after this loop.
                float uc[8];
                load_vec<float, 8>(uc, p.cur + ((int64_t)row * KP + 32 * c) + 8 * g);
#pragma unroll
                for (int a = 0; a < ND; ++a)
#pragma unroll
                    for (int j = 0; j < 8; ++j) pred[a] += v[a][j] * uc[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float acc = 0.0f;
#pragma unroll
                for (int a = 0; a < ND; ++a) acc += v[a][j] * (ok ? w[a] : 0.0f);
                xp[c][j] = acc;
            }
        }
#endif
        static_assert(NC == 4 || NC == 2, "chunks per row");
        const bool b3 = q & 8, b2 = q & 4, b1 = q & 2, b0 = q & 1;
        float s8[8], s4[4], s2[2];
        if constexpr (NC == 4) {
            float h[2][8];  // chunks {0,1} stay with b3 = 0, chunks {2,3} with b3 = 1
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float keep = b3 ? xp[c + 2][j] : xp[c][j];
                    const float give = b3 ? xp[c][j] : xp[c + 2][j];
                    h[c][j] = keep + dpp_get<0x128>(give);  // row_ror:8 = lane q ^ 8
                }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float keep = b2 ? h[1][j] : h[0][j];
                const float give = b2 ? h[0][j] : h[1][j];
                // lane q ^ 4: rotate by 4 towards the partner, either direction by lane
                // (both fetched by every lane: a fetch inside a branch would read lanes the
                // branch has switched off)
                const float below = dpp_get<0x124>(give), above = dpp_get<0x12C>(give);
                s8[j] = keep + (b2 ? below : above);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = xp[0][j] + dpp_get<0x128>(xp[0][j]);
                const float u = xp[1][j] + dpp_get<0x128>(xp[1][j]);
                const float keep = b2 ? u : t;
                const float give = b2 ? t : u;
                // (both fetched by every lane: a fetch inside a branch would read lanes the
                // branch has switched off)
                const float below = dpp_get<0x124>(give), above = dpp_get<0x12C>(give);
                s8[j] = keep + (b2 ? below : above);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float keep = b1 ? s8[4 + j] : s8[j];
            const float give = b1 ? s8[j] : s8[4 + j];
            s4[j] = keep + dpp_get<0x4E>(give);  // quad_perm [2,3,0,1] = lane q ^ 2
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = b0 ? s4[2 + j] : s4[j];
            const float give = b0 ? s4[j] : s4[2 + j];
            s2[j] = keep + dpp_get<0xB1>(give);  // quad_perm [1,0,3,2] = lane q ^ 1
        }
        if constexpr (ERR && !(DUAL_ABLATE & 4)) {
#pragma unroll
            for (int a = 0; a < ND; ++a) {
                const float ra = bperm(4 * (16 * (a & 3) + q), a < 4 ? rv : rv_hi);
                const float pa = col_allreduce(pred[a]);
                const double e = (double)ra - (double)pa;
                err_acc += (g == 0 && real[a]) ? e * e : 0.0;
            }
        }
        // lane q now holds dimensions 32*c + 8*g + 4*b1 + 2*b0 + {0,1} of chunk
        // c = (NC == 4 ? 2*b3 + b2 : b2); with NC == 2 both b3 halves hold the same sums
        {
            const int c = NC == 4 ? ((q >> 2) & 3) : ((q >> 2) & 1);
            f32x2 o = {s2[0], s2[1]};
            float *dstp = out + 32 * c + 8 * g + 2 * (q & 3);
            if (NC == 4 || !b3) *reinterpret_cast<f32x2 *>(dstp) = o;
        }
    }
    if constexpr (ERR) {
        const double wave_err = wave_sum(err_acc);
        if (lane == 0) p.err_part[blockIdx.x * 4 + wl] = wave_err;
    }
}

// ---- kernel 2c: short rows in the neighbour-by-neighbour form, fp64 ------------------------------
// The same identity as in solve_rows_dual_kernel, on the fp64 matrix pipe: K = F F^T is formed
// by v_mfma_f64_16x16x4, whose operand is one double per lane (lane (g,q): row q, contraction
// slot g).  The contraction runs over the factor dimensions in any order, so lane group g takes
// dimensions 16c + 4g .. + 3 of chunk c, one per step: a lane's 32-byte load of one neighbour is
// four steps' operands as it stands.  w = (K + c I)^-1 r comes from the fp64 block solver, and
// x = F^T w from a second pass over the neighbours' rows (cache hits), summed over a lane
// group's 16 lanes.  At padded rank 128 the n x n system saves most where the rank x rank one
// costs most: the fp64 solve of 128 unknowns is as long as the Gram build of 120 entries.
// Waves per SIMD: the 21 tiles of 96 neighbours take the registers of a whole SIMD (and four
// such waves' back-substitution data 96 KB of LDS); 80 neighbours fit twice (-29 % measured).
#ifndef DUAL_D_ABLATE
#define DUAL_D_ABLATE 0  // timing builds only: 1 no solve, 2 no x = F^T w pass, 4 no matrix instructions
#endif
constexpr int dual_d_min_waves(int nd) { return nd >= 6 ? 1 : (nd >= 3 ? 2 : 3); }
template <int NV, int ND, bool ERR>
__global__ __launch_bounds__(256, dual_d_min_waves(ND)) void solve_rows_dual_d_kernel(
    SolveArgs<double> p) {
    constexpr int KP = NV * 16;
    constexpr int NC = KP / 16;   // 16-dimension chunks, four matrix instructions deep each
    constexpr bool TWO = ND > 4;  // rows of more than 64 entries: a second entry per lane
    static_assert(ND * 16 <= 2 * WAVE, "at most two entries per lane");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wl = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the pivot blocks' back-substitution data (block_solve_d)
    double *stash = reinterpret_cast<double *>(smem) + (size_t)wl * (ND * 8 * WAVE);
    const int W = gridDim.x * 4;
    double err_acc = 0.0;
    for (int t = p.t_begin + blockIdx.x * 4 + wl; t < p.t_end; t += W) {
        const int row = p.order[t];
        const int64_t beg = p.indptr[row], end = p.indptr[row + 1];
        const int n = (int)(end - beg);
        double *out = p.dst + (int64_t)row * KP;
        if (n <= 0) {
#pragma unroll
            for (int rr = 0; rr < KP / 64; ++rr) out[lane + 64 * rr] = 0.0;
            continue;
        }
        int lane_v = lane;
        asm volatile("" : "+v"(lane_v));  // lane constants derived per row, not kept across rows
        const int g = lane_v >> 4, q = lane_v & 15;
        // lanes past the row's end repeat its first entry (their rows of F are zeroed below)
        const int64_t at = beg + (lane_v < n ? lane_v : 0);
        const int idx = p.indices[at];
        const double rv = lane_v < n ? p.vals[at] : 0.0;
        int idx_hi = 0;
        double rv_hi = 0.0;
        if constexpr (TWO) {
            const int64_t at_hi = beg + (lane_v + WAVE < n ? lane_v + WAVE : 0);
            idx_hi = p.indices[at_hi];
            rv_hi = lane_v + WAVE < n ? p.vals[at_hi] : 0.0;
        }
        int id[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a)
            id[a] = __builtin_amdgcn_ds_bpermute(4 * (16 * (a & 3) + q), a < 4 ? idx : idx_hi);
        // (instantiation ND only sees rows of more than 16 (ND - 1) entries: all tiles but the
        // last are full)
        const bool last_real = 16 * (ND - 1) + q < n;
        const double *srcg = p.src + 4 * g;
        Gram<double, ND> G;
        G.zero();
        double pred[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) pred[a] = 0.0;
#pragma unroll 2
        for (int c = 0; c < NC; ++c) {
            double v[ND][4];
#pragma unroll
            for (int a = 0; a < ND; ++a)
                load_vec<double, 4>(v[a], srcg + ((int64_t)id[a] * KP + 16 * c));
#pragma unroll
            for (int s = 0; s < 4; ++s) v[ND - 1][s] = last_real ? v[ND - 1][s] : 0.0;
            if constexpr (ERR) {
                // <current vector, neighbour> rides along (the current vector may live in the
                // table being written: it is read before the row's stores)
                double uc[4];
                load_vec<double, 4>(uc, p.cur + ((int64_t)row * KP + 16 * c) + 4 * g);
#pragma unroll
                for (int a = 0; a < ND; ++a)
#pragma unroll
                    for (int s = 0; s < 4; ++s) pred[a] += v[a][s] * uc[s];
            }
#if DUAL_D_ABLATE & 4
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < ND; ++a)
                    G.acc[tri_index(a, a, ND)][s] += v[a][s];
#else
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int a = 0; a < ND; ++a)
#pragma unroll
                    for (int b = a; b < ND; ++b)
                        G.acc[tri_index(a, b, ND)] = __builtin_amdgcn_mfma_f64_16x16x4f64(
                            v[a][s], v[b][s], G.acc[tri_index(a, b, ND)], 0, 0, 0);
#endif
        }
        // right-hand side r, one neighbour per column; block_solve_d adds the four lane groups
        double ra[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            ra[a] = shfl_d(a < 4 ? rv : rv_hi, 16 * (a & 3) + q);
            G.b[a] = g == 0 ? ra[a] : 0.0;
        }
        double w[ND];
#if DUAL_D_ABLATE & 1
        bool ok = true;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            double tt = G.b[a];
#pragma unroll
            for (int b = a; b < ND; ++b)
                for (int r = 0; r < 4; ++r) tt += G.acc[tri_index(a, b, ND)][r];
            w[a] = tt;
        }
#else
        const bool ok = block_solve_d<ND, true>(G, lane, n, p.alpha * (double)n, w, stash);
#endif
        if constexpr (ERR) {
#pragma unroll
            for (int a = 0; a < ND; ++a) {
                const double pa = col_allreduce_d(pred[a]);
                const double e = ra[a] - pa;
                err_acc += (g == 0 && 16 * a + q < n) ? e * e : 0.0;
            }
        }
        // x = F^T w: the neighbours' rows again, this lane's four dimensions of each chunk,
        // summed over the 16 lanes of the lane group; lane q = 0 of each group stores them
#pragma unroll
        for (int a = 0; a < ND; ++a) {
            asm volatile("" : "+v"(id[a]));  // re-read, do not keep the first pass's registers
            w[a] = (ok && 16 * a + q < n) ? w[a] : 0.0;
        }
#if DUAL_D_ABLATE & 2
        if (q == 0) out[4 * g] = w[0] + w[ND - 1];
#else
#pragma unroll 2
        for (int c = 0; c < NC; ++c) {
            double v[ND][4];
#pragma unroll
            for (int a = 0; a < ND; ++a)
                load_vec<double, 4>(v[a], srcg + ((int64_t)id[a] * KP + 16 * c));
            f64x4 x;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                double acc = 0.0;
#pragma unroll
                for (int a = 0; a < ND; ++a) acc += v[a][s] * w[a];
                x[s] = row_allreduce_d(acc);
            }
            if (q == 0) *reinterpret_cast<f64x4 *>(out + 16 * c + 4 * g) = x;
        }
#endif
    }
    if constexpr (ERR) {
        const double wave_err = wave_sum(err_acc);
        if (lane == 0) p.err_part[blockIdx.x * 4 + wl] = wave_err;
    }
}

}  // namespace alsmi

// ---- plan ------------------------------------------------------------------------------------

struct alsmi_plan {
    int dtype, rank, kp, chunk, device, cu_count;
    int64_t n_rows_total, row_begin, row_end;
    int n_tasks, n_chunks, n_heavy;
    // Tasks [t_dual[k], t_dual[k+1]) have at most DUAL_MAX_DEG - 16 k entries (and more than
    // DUAL_MAX_DEG - 16 (k + 1)); the last = n_tasks, and all = n_tasks when the dual form does
    // not apply to the plan.
    int t_dual[alsmi::DUAL_CLASSES + 1];
    int solve_prio;  // bit 0: the plan's rows are short on average (see solve_rows_kernel);
                     // bit 1: the short-row kernels raise their priority too (always, -1 %)
    int32_t *order, *slab_first, *slab_count;
    int64_t *ch_beg, *ch_end;
    void *slabs;
    size_t slab_bytes;
    double *err_part;  // one squared-error partial per wave of the solve grid
    int err_part_cap;
    unsigned *max_bits;  // two words: bit pattern of the largest |source factor| of the running
                         // call in word max_slot ^ 1 (after the call's first kernel), the other
                         // one zeroed for the next call
    int max_slot;
};

namespace alsmi {

static size_t slab_slots(int kp) {
    const int nv = kp / 16;
    return (size_t)(nv * (nv + 1) / 2 * 4 + nv + 1) * WAVE;
}

// Plans whose rows hold at most this many entries on average run their solves at raised wave
// priority (ALSMI_SOLVE_PRIO=0..3 overrides: bit 0 the ld x ld kernel, bit 1 the short-row kernels).
constexpr int64_t SOLVE_PRIO_MAX_MEAN = 384;

// Diagnostic build selected by the environment (fp32, padded rank 64 — and 128 for the values
// 1 and 2 —, plain half-sweep only).
static int ablation() {
    static const int v = getenv("ALSMI_ABLATE") ? atoi(getenv("ALSMI_ABLATE")) : 0;
    return v;
}

// Switches read once from the environment (diagnostics; results stay within the same
// tolerances): ALSMI_DUAL=0 sends short rows through the KP x KP form as well.
static bool dual_enabled() {
    static const bool v = !(getenv("ALSMI_DUAL") && atoi(getenv("ALSMI_DUAL")) == 0);
    return v;
}

// ALSMI_DUAL_F64=0: the same for fp64 alone.
static bool dual_f64_enabled() {
    static const bool v = !(getenv("ALSMI_DUAL_F64") && atoi(getenv("ALSMI_DUAL_F64")) == 0);
    return v && dual_enabled();
}

// Does an fp32 source table of n_src rows need 64-bit gather addresses?  ALSMI_WIDE=1 forces
// them on (test aid).
static bool wide_table(int64_t n_src, int kp) {
    static const bool force = getenv("ALSMI_WIDE") && atoi(getenv("ALSMI_WIDE")) != 0;
    return force || n_src * (int64_t)kp * 4 > (int64_t)UINT32_MAX;
}

// Grid for a kernel whose waves take tasks round-robin: every wave resident at once.
static int resident_grid(const alsmi_plan *pl, const void *kern, int threads, size_t lds,
                         int n_tasks, int wpb, int *grid_out) {
    int bpc = 0;
    ALSMI_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, kern, threads, lds));
    if (bpc < 1) bpc = 1;
    int grid = pl->cu_count * bpc;
    const int need = (n_tasks + wpb - 1) / wpb;
    if (grid > need) grid = need;
    *grid_out = grid;
    return ALSMI_OK;
}

// Launches the KP x KP kernel on tasks [a.t_begin, a.t_end).  With want_err the kernel's
// per-wave squared-error partials go to a.err_part[0 .. *n_err).
template <typename T, int NV, int WPB>
static int launch_solve(const alsmi_plan *pl, SolveArgs<T> a, bool dump, bool want_err,
                        bool wide, int *n_err, hipStream_t st) {
    constexpr int KP = NV * 16;
    *n_err = 0;
    const int n_tasks = a.t_end - a.t_begin;
    if (n_tasks <= 0) return ALSMI_OK;
    const size_t lds =
        (dump ? WaveLds<T, KP, true>::BYTES : WaveLds<T, KP, false>::BYTES) * WPB;
    auto kern = dump ? solve_rows_kernel<T, NV, WPB, true> : solve_rows_kernel<T, NV, WPB, false>;
    if (want_err) kern = solve_rows_kernel<T, NV, WPB, false, 0, true>;
    if constexpr (sizeof(T) == 4 && NV == 8) {
        if (wide && !dump)
            kern = want_err ? solve_rows_kernel<T, NV, WPB, false, 20, true>
                            : solve_rows_kernel<T, NV, WPB, false, 20, false>;
    }
    if constexpr (sizeof(T) == 4 && NV == 8) {  // (accumulate-only / solve-only timing builds)
        const int ablate = ablation();
        if (!dump && !want_err && !wide) {
            if (ablate == 1) kern = solve_rows_kernel<T, NV, WPB, false, 1>;
            if (ablate == 2) kern = solve_rows_kernel<T, NV, WPB, false, 2>;
        }
    }
    if constexpr (sizeof(T) == 4 && NV == 4) {
        const int ablate = ablation();
        if (!dump && !want_err) {
            if (ablate == 1) kern = solve_rows_kernel<T, NV, WPB, false, 1>;
            if (ablate == 2) kern = solve_rows_kernel<T, NV, WPB, false, 2>;
            if (ablate == 4) kern = solve_rows_kernel<T, NV, WPB, false, 4>;
            if (ablate == 10) kern = solve_rows_kernel<T, NV, WPB, false, 10>;
        }
    }
    ALSMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // Rows are dealt to waves round-robin over the degree-sorted order, which balances the
    // load only if every wave of the grid is resident at once: size the grid to what the
    // registers and the LDS of this instantiation admit per CU.  (Waves never wait on each
    // other, so an over-estimate costs time, not correctness.)
    int grid = 0;
    int rc = resident_grid(pl, reinterpret_cast<const void *>(kern), WPB * 64, lds, n_tasks, WPB,
                           &grid);
    if (rc != ALSMI_OK) return rc;
    if (want_err && grid * WPB > pl->err_part_cap) grid = pl->err_part_cap / WPB;
    if (grid < 1) return ALSMI_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WPB * 64), lds, st, a);
    ALSMI_HIP_CHECK(hipGetLastError());
    if (want_err) *n_err = grid * WPB;
    return ALSMI_OK;
}

// The dual-form kernel on tasks [a.t_begin, a.t_end), all of at most 16*ND entries.
template <int NV, int ND>
static int launch_dual(const alsmi_plan *pl, SolveArgs<float> a, bool want_err, int *n_err,
                       hipStream_t st) {
    *n_err = 0;
    const int n_tasks = a.t_end - a.t_begin;
    if (n_tasks <= 0) return ALSMI_OK;
    auto kern = want_err ? solve_rows_dual_kernel<NV, ND, true>
                         : solve_rows_dual_kernel<NV, ND, false>;
    int grid = 0;
    int rc = resident_grid(pl, reinterpret_cast<const void *>(kern), 256, 0, n_tasks, 4, &grid);
    if (rc != ALSMI_OK) return rc;
    if (want_err && grid * 4 > pl->err_part_cap) grid = pl->err_part_cap / 4;
    if (grid < 1) return ALSMI_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a);
    ALSMI_HIP_CHECK(hipGetLastError());
    if (want_err) *n_err = grid * 4;
    return ALSMI_OK;
}

template <int NV, int ND>
static int launch_dual_d(const alsmi_plan *pl, SolveArgs<double> a, bool want_err, int *n_err,
                         hipStream_t st) {
    *n_err = 0;
    const int n_tasks = a.t_end - a.t_begin;
    if (n_tasks <= 0) return ALSMI_OK;
    auto kern = want_err ? solve_rows_dual_d_kernel<NV, ND, true>
                         : solve_rows_dual_d_kernel<NV, ND, false>;
    const size_t lds = (size_t)4 * ND * 8 * WAVE * sizeof(double);
    ALSMI_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int grid = 0;
    int rc = resident_grid(pl, reinterpret_cast<const void *>(kern), 256, lds, n_tasks, 4, &grid);
    if (rc != ALSMI_OK) return rc;
    if (want_err && grid * 4 > pl->err_part_cap) grid = pl->err_part_cap / 4;
    if (grid < 1) return ALSMI_OK;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, a);
    ALSMI_HIP_CHECK(hipGetLastError());
    if (want_err) *n_err = grid * 4;
    return ALSMI_OK;
}

template <typename T, int NV, int WPB>
static int run_half_sweep(alsmi_plan *pl, const int64_t *indptr, const int32_t *indices,
                          const void *vals, const void *src, int64_t n_src, void *dst,
                          double alpha, void *A_out, void *b_out, const void *cur,
                          double *err_out, hipStream_t st) {
    const bool dump = A_out != nullptr;
    int mode = GRAM_NATIVE;
    unsigned *max_word = pl->max_bits;
    if constexpr (sizeof(T) == 4 && NV <= 8) {
        mode = GRAM_F16X2;  // the dump included: it accumulates as the half-sweep does
        if constexpr (NV == 4) {
            if (!dump && !err_out && ablation() == 4) mode = GRAM_NATIVE;
            if (!dump && !err_out && ablation() == 10) mode = GRAM_BF16X3;
        }
        if (mode == GRAM_F16X2) {
            if (wide_table(n_src, NV * 16)) {
                // padded rank 128 has a build with 64-bit gather addresses (a table of 10 M
                // rows passes 4 GiB there); the smaller ranks do not
                ALSMI_REQUIRE(NV == 8,
                              "half-sweep: an fp32 source table of %lld rows at padded rank %d "
                              "exceeds 4 GiB", (long long)n_src, NV * 16);
                if (!dump)
                    mode = GRAM_F16X2_WIDE;
                else
                    ALSMI_REQUIRE(n_src * (int64_t)(NV * 16) * 4 <= (int64_t)UINT32_MAX,
                                  "alsmi_normal_equations: source table above 4 GiB");
            }
            // scale of the fp16 split: the largest magnitude of the source table, into the word
            // the previous call left zeroed
            max_word = pl->max_bits + pl->max_slot;
            const int64_t n4 = n_src * (int64_t)(NV * 16) / 4;
            int blocks = (int)std::min<int64_t>((n4 + 255) / 256, 2 * (int64_t)pl->cu_count);
            if (blocks < 1) blocks = 1;
            hipLaunchKernelGGL(max_abs_kernel, dim3(blocks), dim3(256), 0, st,
                               static_cast<const float *>(src), n4, max_word,
                               pl->max_bits + (pl->max_slot ^ 1));
            ALSMI_HIP_CHECK(hipGetLastError());
            pl->max_slot ^= 1;
        }
    }
    if (pl->n_chunks > 0) {
        const int grid = (pl->n_chunks + 3) / 4;
        auto chunks = gram_chunks_kernel<T, NV, GRAM_NATIVE, false>;
        if (err_out) chunks = gram_chunks_kernel<T, NV, GRAM_NATIVE, true>;
        if constexpr (sizeof(T) == 4 && NV <= 8) {
            if (mode == GRAM_F16X2)
                chunks = err_out ? gram_chunks_kernel<T, NV, GRAM_F16X2, true>
                                 : gram_chunks_kernel<T, NV, GRAM_F16X2, false>;
            if constexpr (NV == 8)
                if (mode == GRAM_F16X2_WIDE)
                    chunks = err_out ? gram_chunks_kernel<T, NV, GRAM_F16X2_WIDE, true>
                                     : gram_chunks_kernel<T, NV, GRAM_F16X2_WIDE, false>;
            if constexpr (NV <= 4)
                if (mode == GRAM_BF16X3) chunks = gram_chunks_kernel<T, NV, GRAM_BF16X3, false>;
        }
        hipLaunchKernelGGL(chunks, dim3(grid), dim3(256), 0, st, pl->ch_beg, pl->ch_end,
                           pl->n_chunks, indices, static_cast<const T *>(vals),
                           static_cast<const T *>(src), max_word, static_cast<T *>(pl->slabs));
        ALSMI_HIP_CHECK(hipGetLastError());
        const int slab_elems = Gram<T, NV>::SLAB_SLOTS * WAVE;
        const int per_row = (slab_elems + 255) / 256;
        hipLaunchKernelGGL((sum_slabs_kernel<T>), dim3((unsigned)pl->n_heavy * per_row),
                           dim3(256), 0, st, pl->slab_first, pl->slab_count, pl->n_heavy,
                           slab_elems, static_cast<T *>(pl->slabs));
        ALSMI_HIP_CHECK(hipGetLastError());
    }
    SolveArgs<T> a;
    a.order = pl->order;
    a.slab_first = pl->slab_first;
    a.slab_count = pl->slab_count;
    a.indptr = indptr;
    a.indices = indices;
    a.vals = static_cast<const T *>(vals);
    a.src = static_cast<const T *>(src);
    a.slabs = static_cast<const T *>(pl->slabs);
    a.max_bits = max_word;
    a.dst = static_cast<T *>(dst);
    a.rank = pl->rank;
    a.alpha = (T)alpha;
    a.A_out = static_cast<T *>(A_out);
    a.b_out = static_cast<T *>(b_out);
    a.row_begin = pl->row_begin;
    a.solve_prio = pl->solve_prio;
    a.cur = static_cast<const T *>(cur);
    a.err_part = pl->err_part;
    // Short rows go to the dual-form kernels (fp32, padded rank 128, not for the dump, whose
    // output is the KP x KP system itself).  The task list is sorted by length, so each kernel
    // takes a contiguous share; each appends its squared-error partials to the same array.
    bool dual = false;
    if constexpr (sizeof(T) == 4 && (NV == 8 || NV == 4))
        dual = !dump && is_f16x2(mode) && dual_enabled() && (NV == 8 || ablation() == 0);
    if constexpr (sizeof(T) == 8 && (NV == 8 || NV == 4)) dual = !dump && dual_f64_enabled();
    int n_err = 0, used = 0;
    a.t_begin = 0;
    a.t_end = dual ? pl->t_dual[0] : pl->n_tasks;
    int rc = launch_solve<T, NV, WPB>(pl, a, dump, err_out != nullptr, mode == GRAM_F16X2_WIDE,
                                      &used, st);
    if (rc != ALSMI_OK) return rc;
    n_err += used;
    if constexpr (sizeof(T) == 4 && (NV == 8 || NV == 4)) {
        if (dual) {
            for (int k = 0; k < DUAL_CLASSES; ++k) {
                a.err_part = pl->err_part + n_err;
                a.t_begin = pl->t_dual[k];
                a.t_end = pl->t_dual[k + 1];
                used = 0;
                // (a neighbour system of 64 or more gains nothing at padded rank 64: its classes
                // are empty there, see alsmi_plan_create)
                switch (DUAL_CLASSES - k) {
#if DUAL_MAX_DEG_128 >= 96
                    case 6:
                        if constexpr (NV == 8)
                            rc = launch_dual<NV, 6>(pl, a, err_out != nullptr, &used, st);
                        break;
#endif
#if DUAL_MAX_DEG_128 >= 80
                    case 5:
                        if constexpr (NV == 8)
                            rc = launch_dual<NV, 5>(pl, a, err_out != nullptr, &used, st);
                        break;
#endif
                    case 4:
                        if constexpr (NV == 8)
                            rc = launch_dual<NV, 4>(pl, a, err_out != nullptr, &used, st);
                        break;
                    case 3: rc = launch_dual<NV, 3>(pl, a, err_out != nullptr, &used, st); break;
                    case 2: rc = launch_dual<NV, 2>(pl, a, err_out != nullptr, &used, st); break;
                    default: rc = launch_dual<NV, 1>(pl, a, err_out != nullptr, &used, st); break;
                }
                if (rc != ALSMI_OK) return rc;
                n_err += used;
            }
        }
    }
    if constexpr (sizeof(T) == 8 && (NV == 8 || NV == 4)) {
        if (dual) {
            for (int k = 0; k < DUAL_CLASSES; ++k) {
                a.err_part = pl->err_part + n_err;
                a.t_begin = pl->t_dual[k];
                a.t_end = pl->t_dual[k + 1];
                used = 0;
                const bool e = err_out != nullptr;
                switch (DUAL_CLASSES - k) {
#if DUAL_MAX_DEG_128 >= 96
                    case 6:
                        if constexpr (NV == 8) rc = launch_dual_d<NV, 6>(pl, a, e, &used, st);
                        break;
#endif
#if DUAL_MAX_DEG_128 >= 80
                    case 5:
                        if constexpr (NV == 8) rc = launch_dual_d<NV, 5>(pl, a, e, &used, st);
                        break;
#endif
                    case 4:
                        if constexpr (NV == 8) rc = launch_dual_d<NV, 4>(pl, a, e, &used, st);
                        break;
                    case 3: rc = launch_dual_d<NV, 3>(pl, a, e, &used, st); break;
                    case 2: rc = launch_dual_d<NV, 2>(pl, a, e, &used, st); break;
                    default: rc = launch_dual_d<NV, 1>(pl, a, e, &used, st); break;
                }
                if (rc != ALSMI_OK) return rc;
                n_err += used;
            }
        }
    }
    if (err_out) {
        hipLaunchKernelGGL(sum_err_kernel, dim3(1), dim3(256), 0, st, pl->err_part, n_err,
                           err_out);
        ALSMI_HIP_CHECK(hipGetLastError());
    }
    return ALSMI_OK;
}

static int dispatch(alsmi_plan *pl, const int64_t *indptr, const int32_t *indices,
                    const void *vals, const void *src, int64_t n_src, void *dst, double alpha,
                    void *A_out, void *b_out, const void *cur, double *err_out, hipStream_t st) {
#define ALSMI_CASE(T, NV, WPB)                                                                \
    return run_half_sweep<T, NV, WPB>(pl, indptr, indices, vals, src, n_src, dst, alpha, A_out, \
                                      b_out, cur, err_out, st)
    if (pl->dtype == ALSMI_F32) {
        switch (pl->kp) {
            case 16: ALSMI_CASE(float, 1, 4);
            case 32: ALSMI_CASE(float, 2, 4);
            case 64: ALSMI_CASE(float, 4, 4);
            case 128: ALSMI_CASE(float, 8, 2);
        }
    } else {
        switch (pl->kp) {
            case 16: ALSMI_CASE(double, 1, 4);
            case 32: ALSMI_CASE(double, 2, 4);
            case 64: ALSMI_CASE(double, 4, 4);
            case 128: ALSMI_CASE(double, 8, 1);
        }
    }
#undef ALSMI_CASE
    set_error("unsupported dtype/rank combination");
    return ALSMI_EINVAL;
}

}  // namespace alsmi

extern "C" {

int alsmi_plan_create(alsmi_plan **out, const int64_t *indptr_host, int64_t n_rows_total,
                      int64_t row_begin, int64_t row_end, int rank, int dtype, int chunk) {
    using namespace alsmi;
    ALSMI_REQUIRE(out && indptr_host, "alsmi_plan_create: null pointer");
    ALSMI_REQUIRE(dtype == ALSMI_F32 || dtype == ALSMI_F64, "alsmi_plan_create: bad dtype %d",
                  dtype);
    const int kp = alsmi_padded_rank(rank);
    ALSMI_REQUIRE(kp > 0, "alsmi_plan_create: rank %d outside 1..128", rank);
    ALSMI_REQUIRE(n_rows_total >= 0 && n_rows_total < INT32_MAX, "alsmi_plan_create: bad row count");
    ALSMI_REQUIRE(0 <= row_begin && row_begin <= row_end && row_end <= n_rows_total,
                  "alsmi_plan_create: bad row range [%lld,%lld) of %lld", (long long)row_begin,
                  (long long)row_end, (long long)n_rows_total);
    const int64_t n = row_end - row_begin;
    if (chunk <= 0) {
        // measured at 138 493 x 26 744 with 20 M entries: the side with short rows (144 entries
        // on average) is fastest with 2048, the side with long rows (748) with 4096
        const int64_t entries = indptr_host[row_end] - indptr_host[row_begin];
        chunk = (n > 0 && entries / n > 512) ? 4096 : 2048;
    }
    chunk = (chunk + 63) / 64 * 64;
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (int32_t)row_begin);
    for (int64_t r = row_begin; r < row_end; ++r)
        ALSMI_REQUIRE(indptr_host[r + 1] >= indptr_host[r] &&
                          indptr_host[r + 1] - indptr_host[r] < INT32_MAX,
                      "alsmi_plan_create: indptr not monotone at row %lld", (long long)r);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        return indptr_host[a + 1] - indptr_host[a] > indptr_host[b + 1] - indptr_host[b];
    });
    std::vector<int32_t> slab_first((size_t)n, -1), slab_count((size_t)n, 0);
    std::vector<int64_t> ch_beg, ch_end;
    int n_heavy = 0;
    for (int64_t t = 0; t < n; ++t) {
        const int64_t beg = indptr_host[order[t]], end = indptr_host[order[t] + 1];
        const int64_t deg = end - beg;
        if (deg <= chunk) break;  // sorted: the rest is short
        int64_t nch = (deg + chunk - 1) / chunk;
        int64_t len = ((deg + nch - 1) / nch + 63) / 64 * 64;
        nch = (deg + len - 1) / len;
        ALSMI_REQUIRE(ch_beg.size() + (size_t)nch < (size_t)INT32_MAX, "too many chunks");
        slab_first[t] = (int32_t)ch_beg.size();
        slab_count[t] = (int32_t)nch;
        for (int64_t c = 0; c < nch; ++c) {
            ch_beg.push_back(beg + c * len);
            ch_end.push_back(std::min(end, beg + (c + 1) * len));
        }
        ++n_heavy;
    }
    alsmi_plan *pl = new (std::nothrow) alsmi_plan();
    if (!pl) {
        set_error("alsmi_plan_create: out of host memory");
        return ALSMI_ENOMEM;
    }
    memset(pl, 0, sizeof(*pl));
    pl->dtype = dtype;
    pl->rank = rank;
    pl->kp = kp;
    pl->chunk = chunk;
    pl->n_rows_total = n_rows_total;
    pl->row_begin = row_begin;
    pl->row_end = row_end;
    pl->n_tasks = (int)n;
    pl->n_chunks = (int)ch_beg.size();
    pl->n_heavy = n_heavy;
    for (int k = 0; k <= alsmi::DUAL_CLASSES; ++k) pl->t_dual[k] = (int)n;
    {
        const int64_t entries = indptr_host[row_end] - indptr_host[row_begin];
        static const int forced = getenv("ALSMI_SOLVE_PRIO") ? atoi(getenv("ALSMI_SOLVE_PRIO")) : -1;
        pl->solve_prio = forced >= 0 ? forced : ((n > 0 && entries / n <= SOLVE_PRIO_MAX_MEAN ? 1 : 0) | 2);
    }
    if (kp == 128 || kp == 64) {
        // The n x n form is used for rows of at most 48 (padded rank 64) / 96 (128) entries AND
        // at most `rank` entries: with more entries than dimensions F F^T is rank deficient, its
        // condition number is that of the ridge alone, and fp32 loses rows the rank x rank
        // form solves easily (found at rank 33 with rows of 34 to 48 entries).
        const int64_t dual_max = std::min<int64_t>(kp == 64 ? 48 : alsmi::DUAL_MAX_DEG, rank);
        for (int k = 0; k < alsmi::DUAL_CLASSES; ++k) {
            const int64_t deg = std::min<int64_t>(alsmi::DUAL_MAX_DEG - 16 * k, dual_max);
            pl->t_dual[k] = (int)(std::partition_point(order.begin(), order.end(), [&](int32_t r) {
                                      return indptr_host[r + 1] - indptr_host[r] > deg;
                                  }) - order.begin());
        }
    }
    pl->cu_count = device_cu_count();
    const size_t esz = dtype == ALSMI_F32 ? 4 : 8;
    pl->slab_bytes = (size_t)pl->n_chunks * slab_slots(kp) * esz;
    hipError_t e = hipGetDevice(&pl->device);
    auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
        *dst = nullptr;
        if (bytes == 0) return hipSuccess;
        hipError_t r = hipMalloc(dst, bytes);
        if (r != hipSuccess) return r;
        return src ? hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
    };
    if (e == hipSuccess) e = upload((void **)&pl->order, order.data(), (size_t)n * 4);
    if (e == hipSuccess) e = upload((void **)&pl->slab_first, slab_first.data(), (size_t)n * 4);
    if (e == hipSuccess) e = upload((void **)&pl->slab_count, slab_count.data(), (size_t)n * 4);
    if (e == hipSuccess) e = upload((void **)&pl->ch_beg, ch_beg.data(), ch_beg.size() * 8);
    if (e == hipSuccess) e = upload((void **)&pl->ch_end, ch_end.data(), ch_end.size() * 8);
    if (e == hipSuccess) e = upload(&pl->slabs, nullptr, pl->slab_bytes);
    if (e == hipSuccess) e = upload((void **)&pl->max_bits, nullptr, 2 * sizeof(unsigned));
    if (e == hipSuccess) e = hipMemset(pl->max_bits, 0, 2 * sizeof(unsigned));
    pl->err_part_cap = pl->cu_count * 64;
    if (e == hipSuccess)
        e = upload((void **)&pl->err_part, nullptr,
                   (alsmi::DUAL_CLASSES + 1) * (size_t)pl->err_part_cap *
                       sizeof(double));  // one share per kernel of a call
    if (e != hipSuccess) {
        set_error("alsmi_plan_create: device allocation failed: %s", hipGetErrorString(e));
        alsmi_plan_destroy(pl);
        return e == hipErrorOutOfMemory ? ALSMI_ENOMEM : ALSMI_EHIP;
    }
    *out = pl;
    return ALSMI_OK;
}

int alsmi_plan_destroy(alsmi_plan *pl) {
    if (!pl) return ALSMI_OK;
    (void)hipFree(pl->order);
    (void)hipFree(pl->slab_first);
    (void)hipFree(pl->slab_count);
    (void)hipFree(pl->ch_beg);
    (void)hipFree(pl->ch_end);
    (void)hipFree(pl->slabs);
    (void)hipFree(pl->err_part);
    (void)hipFree(pl->max_bits);
    delete pl;
    return ALSMI_OK;
}

int alsmi_plan_info(const alsmi_plan *pl, int64_t *n_rows, int64_t *n_heavy_rows,
                    int64_t *n_chunks, int64_t *workspace_bytes) {
    ALSMI_REQUIRE(pl, "alsmi_plan_info: null plan");
    if (n_rows) *n_rows = pl->n_tasks;
    if (n_heavy_rows) *n_heavy_rows = pl->n_heavy;
    if (n_chunks) *n_chunks = pl->n_chunks;
    if (workspace_bytes) *workspace_bytes = (int64_t)pl->slab_bytes;
    return ALSMI_OK;
}

static int check_sweep_args(const alsmi_plan *pl, const int64_t *indptr, const int32_t *indices,
                            const void *vals, const void *src, int64_t n_src, const char *who) {
    ALSMI_REQUIRE(pl, "%s: null plan", who);
    if (pl->n_tasks == 0) return ALSMI_OK;
    ALSMI_REQUIRE(indptr && src, "%s: null pointer", who);
    ALSMI_REQUIRE(n_src >= 1, "%s: the source factor table must have at least one row", who);
    ALSMI_REQUIRE(indices && vals, "%s: null indices/vals", who);
    return ALSMI_OK;
}

int alsmi_half_sweep(alsmi_plan *pl, const int64_t *indptr, const int32_t *indices,
                     const void *vals, const void *src_factors, int64_t n_src, void *dst_factors,
                     double alpha, void *stream) {
    int rc = check_sweep_args(pl, indptr, indices, vals, src_factors, n_src, "alsmi_half_sweep");
    if (rc != ALSMI_OK) return rc;
    if (pl->n_tasks == 0) return ALSMI_OK;
    ALSMI_REQUIRE(dst_factors, "alsmi_half_sweep: null destination");
    return alsmi::dispatch(pl, indptr, indices, vals, src_factors, n_src, dst_factors, alpha,
                           nullptr, nullptr, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

int alsmi_half_sweep_err(alsmi_plan *pl, const int64_t *indptr, const int32_t *indices,
                         const void *vals, const void *src_factors, int64_t n_src,
                         const void *cur_factors, void *dst_factors, double alpha,
                         double *sse_out, void *stream) {
    int rc =
        check_sweep_args(pl, indptr, indices, vals, src_factors, n_src, "alsmi_half_sweep_err");
    if (rc != ALSMI_OK) return rc;
    ALSMI_REQUIRE(sse_out, "alsmi_half_sweep_err: null sse_out");
    if (pl->n_tasks == 0) {
        ALSMI_HIP_CHECK(hipMemsetAsync(sse_out, 0, sizeof(double), static_cast<hipStream_t>(stream)));
        return ALSMI_OK;
    }
    ALSMI_REQUIRE(dst_factors && cur_factors, "alsmi_half_sweep_err: null factor table");
    return alsmi::dispatch(pl, indptr, indices, vals, src_factors, n_src, dst_factors, alpha,
                           nullptr, nullptr, cur_factors, sse_out, static_cast<hipStream_t>(stream));
}

int alsmi_normal_equations(alsmi_plan *pl, const int64_t *indptr, const int32_t *indices,
                           const void *vals, const void *src_factors, int64_t n_src, void *A_out,
                           void *b_out, double alpha, void *stream) {
    int rc = check_sweep_args(pl, indptr, indices, vals, src_factors, n_src,
                              "alsmi_normal_equations");
    if (rc != ALSMI_OK) return rc;
    if (pl->n_tasks == 0) return ALSMI_OK;
    ALSMI_REQUIRE(A_out && b_out, "alsmi_normal_equations: null output");
    return alsmi::dispatch(pl, indptr, indices, vals, src_factors, n_src, nullptr, alpha, A_out,
                           b_out, nullptr, nullptr, static_cast<hipStream_t>(stream));
}

}  // extern "C"

#ifdef ALSMI_PHASE_CLOCK
extern "C" int alsmi_debug_phase(unsigned long long *out8, int reset) {
    if (out8 && hipMemcpyFromSymbol(out8, HIP_SYMBOL(alsmi::g_phase), 64) != hipSuccess) return -2;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(alsmi::g_phase), z, 64) != hipSuccess) return -2;
    }
    return 0;
}
#endif
