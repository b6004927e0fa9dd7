// Top-k items per user: batched scores U[u] . V[i] over all items and a per-user selection.
//
// The reference's predict_all (src/als.py:347-364) returns one user's scores for every item
// and leaves the ranking to the caller; a recommender serves "the k best items of these
// users".  Two kernels:
//   scores: the [users x items] block as a tiled product on the matrix cores
//           (v_mfma_{f32,f64}_16x16x4), 32 or 64 users per wave against tiles of 16 items;
//           bound by the write of the score block.
//   select: one workgroup per user: 4-pass 8-bit radix select on the order-preserving integer
//           image of the scores (histograms in LDS) finds the k-th largest score, the
//           candidates at or above it are compacted in item order and the k winners are
//           sorted by (score descending, item id ascending) in LDS.  Deterministic.
// For small k and many items the score block is never written (it is 14.8 GB for 138 493 users
// x 26 744 items, and writing and re-reading it was all of the time): the two kernels first
// run on a strided SAMPLE of the item tiles only, whose k-th best score per user is a lower
// bound of the user's true k-th best; a second pass over ALL items computes the same scores
// (bit for bit: a score depends on its two vectors only) and keeps just the (item, score)
// pairs at or above that bound — a few hundred candidates per user, which contain the top k
// — and a small kernel sorts each user's candidates by (score descending, item ascending).
// Should a user collect more candidates than there are slots (many equal scores, e.g. an
// all-zero factor table), a flag on the device switches in the two-kernel form for the whole
// call; the result is the same either way.
#include <stdlib.h>

#include "alsmi_common.h"

namespace alsmi {

constexpr int TOPK_MAX_K = 1024;
constexpr int TOPK_CAND = 1024;         // candidates per user the final sort takes
constexpr int TOPK_CHUNK_CAND = 32;     // candidate slots per user and wave (item range of a wave)
constexpr int TOPK_SAMPLE_MIN = 1024;   // sampled items, at least
constexpr int64_t TOPK_FILTER_MIN_ITEMS = 8192;

// order-preserving map float/double -> unsigned (larger value, larger key)
__device__ __forceinline__ unsigned key_of(float x) {
    const unsigned b = __builtin_bit_cast(unsigned, x);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ unsigned long long key_of(double x) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

// Scores as a tiled matrix product on the matrix cores.  A wave holds UT tiles of 16 users as the
// A operands (lane (g,q): dimensions [g KS, (g+1) KS) of user q's row, KS = ld/4 — any split of
// the contraction index over lane groups and steps will do as long as both operands use the
// same) and walks ITW tiles of 16 items: one 4 KS-byte contiguous load per lane is the B operand
// of all KS steps, the 16 x 16 result tile holds scores[user crow(g,r)][item q].  The same item
// vector gives the same score bit for bit wherever it sits, which the tie rule relies on.
template <typename T, int NV>
struct ScoreTile {
    static constexpr int KS = 4 * NV;  // contraction steps = elements per lane and row
    // user tiles per wave: at most 128 registers of A operands
    static constexpr int UT = (KS * (int)sizeof(T) / 4) * 4 <= 128 ? 4 : 2;
};

template <typename T>
struct KeyT;
template <>
struct KeyT<float> {
    typedef unsigned type;
    static constexpr int BITS = 32;
};
template <>
struct KeyT<double> {
    typedef unsigned long long type;
    static constexpr int BITS = 64;
};
__device__ __forceinline__ float score_of(unsigned k) {  // inverse of key_of
    return __builtin_bit_cast(float, (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}
__device__ __forceinline__ double score_of(unsigned long long k) {
    return __builtin_bit_cast(double, (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k);
}

// Where the scores of topk_scores_kernel go.
//  MODE 0: every score to the dense block scores[user * row_len + item] (row_len = n_items).
//  MODE 1: the sampled tiles only: sampled tile j is item tile j * tile_stride (a whole tile),
//          its scores go to scores[user * row_len + 16 j + q] (row_len = 16 x sampled tiles).
//  MODE 2: all items; a score at or above the user's threshold (order of key_of) is appended
//          to the user's candidate list, the rest is dropped.
template <typename T>
struct ScoreOut {
    T *scores;
    int64_t row_len, tile_stride;
    const T *thresh;       // MODE 2: threshold of user u = thresh[u * thresh_stride]
    int64_t thresh_stride;
    // MODE 2: every wave owns TOPK_CHUNK_CAND slots per user for the item range it walks
    // (wave w of the item direction: tiles [w itw, (w+1) itw)), so no atomics are needed:
    int *count;            // [n_sel x n_wchunks] candidates found (may exceed the slots: overflow)
    int32_t *cand_item;    // [n_sel x n_wchunks x TOPK_CHUNK_CAND]
    T *cand_score;
    int64_t n_wchunks;
    const int *run_if;     // when not null: the launch does nothing unless *run_if != 0
};

// MODE 2 bookkeeping of one wave: cnt[t][r] = candidates found so far for the user of tile t,
// register r of this lane's group (the same value in the 16 lanes of the group).  A hit takes
// the next slot of its user's list for this wave: the slots of the hits of one instruction are
// numbered by lane inside the group (ballot + population counts), so nothing is atomic and the
// list is in item order.
template <typename T>
__device__ __forceinline__ void push_candidates(const ScoreOut<T> &o, bool hit, T val, int64_t item,
                                                int64_t user, int64_t wchunk, int g, int q,
                                                int &cnt) {
    const unsigned long long m = __ballot(hit);
    if (m == 0) return;  // (the same for the whole wave)
    const unsigned rowm = (unsigned)(m >> (16 * g)) & 0xffffu;
    const int slot = cnt + __popc(rowm & ((1u << q) - 1u));
    if (hit && slot < TOPK_CHUNK_CAND) {
        const int64_t at = (user * o.n_wchunks + wchunk) * TOPK_CHUNK_CAND + slot;
        o.cand_item[at] = (int32_t)item;
        o.cand_score[at] = val;
    }
    cnt += __popc(rowm);
}

template <typename T, int NV, int MODE>
__global__ __launch_bounds__(256) void topk_scores_kernel(int n_sel, const int32_t *__restrict__ sel,
                                                          const T *__restrict__ U,
                                                          const T *__restrict__ V, int64_t n_items,
                                                          ScoreOut<T> o, int itw) {
    typedef typename KeyT<T>::type key_t;
    constexpr int KP = NV * 16;
    constexpr int KS = ScoreTile<T, NV>::KS;
    constexpr int UT = ScoreTile<T, NV>::UT;
    if (o.run_if && *o.run_if == 0) return;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int wl = threadIdx.x >> 6;
    const int64_t user_base = (int64_t)blockIdx.x * (16 * UT);
    // tiles this launch walks: all item tiles, or (MODE 1) the sampled ones
    const int64_t n_tiles = MODE == 1 ? o.row_len / 16 : (n_items + 15) / 16;
    const int64_t tile_begin = ((int64_t)blockIdx.y * 4 + wl) * itw;
    if (tile_begin >= n_tiles) return;
    const int64_t tile_end = tile_begin + itw < n_tiles ? tile_begin + itw : n_tiles;
    T a[UT][KS];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
        int64_t ui = user_base + 16 * t + q;
        ui = ui < n_sel ? ui : n_sel - 1;  // (rows past the end are computed and not stored)
        const T *row = U + (int64_t)sel[ui] * KP + g * KS;
#pragma unroll
        for (int c = 0; c < KS; c += 4) load_vec<T, 4>(*reinterpret_cast<T(*)[4]>(&a[t][c]), row + c);
    }
    const int64_t wchunk = (int64_t)blockIdx.y * 4 + wl;
    key_t kthr[MODE == 2 ? UT : 1][4];
    int cnt[MODE == 2 ? UT : 1][4];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t user = user_base + 16 * t + Mfma<T>::crow(g, r);
                user = user < n_sel ? user : n_sel - 1;
                kthr[t][r] = key_of(o.thresh[user * o.thresh_stride]);
                cnt[t][r] = 0;
            }
    }
    for (int64_t it = tile_begin; it < tile_end; ++it) {
        const int64_t item = 16 * (MODE == 1 ? it * o.tile_stride : it) + q;
        const bool item_ok = item < n_items;
        const T *row = V + (item_ok ? item : n_items - 1) * KP + g * KS;
        T b[KS];
#pragma unroll
        for (int c = 0; c < KS; c += 4) load_vec<T, 4>(*reinterpret_cast<T(*)[4]>(&b[c]), row + c);
#pragma unroll
        for (int t = 0; t < UT; ++t) {
            typename Mfma<T>::acc_t acc = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < KS; ++k) acc = Mfma<T>::mma(a[t][k], b[k], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t user = user_base + 16 * t + Mfma<T>::crow(g, r);
                if (!(item_ok && user < n_sel)) continue;
                if constexpr (MODE == 0) {
                    o.scores[user * o.row_len + item] = acc[r];
                } else if constexpr (MODE == 1) {
                    o.scores[user * o.row_len + 16 * it + q] = acc[r];
                }
            }
            if constexpr (MODE == 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t user = user_base + 16 * t + Mfma<T>::crow(g, r);
                    const bool hit = item_ok && user < n_sel && key_of(acc[r]) >= kthr[t][r];
                    push_candidates<T>(o, hit, acc[r], item, user, wchunk, g, q, cnt[t][r]);
                }
            }
        }
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t user = user_base + 16 * t + Mfma<T>::crow(g, r);
                if (q == 0 && user < n_sel) o.count[user * o.n_wchunks + wchunk] = cnt[t][r];
            }
    }
}

// Largest |x| over the given rows of a table (rows[] when not null, else rows 0 .. n_rows-1) as
// the bit pattern of a non-negative float, combined into *out with atomicMax (zeroed by the
// caller): the power-of-two scale of the fp16 split is taken from it.
__global__ __launch_bounds__(256) void topk_max_abs_kernel(int64_t n_rows,
                                                           const int32_t *__restrict__ rows,
                                                           const float *__restrict__ table, int kp,
                                                           unsigned *__restrict__ out) {
    typedef float vec_t __attribute__((ext_vector_type(4)));
    const int per_row = kp / 4;
    const int64_t total = n_rows * per_row;
    float m = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row;
        const int64_t src = rows ? (int64_t)rows[r] : r;
        const vec_t v = reinterpret_cast<const vec_t *>(table + src * kp)[i - r * per_row];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0)  // one atomic per block (they all go to one address)
        atomicMax(out, __builtin_bit_cast(
                           unsigned, fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]))));
}

// fp32 scores on the fp16 matrix cores (the scaled two-way split of alsmi_common.h, as in the
// half-sweep's Gram build): the fp32 matrix instruction runs at the vector rate, 16 times below
// the 16-bit one, and the score product was all of this kernel's time.  Lane (g,q) holds
// dimensions 32 s + 8 g .. + 7 of user q (A) / item q (B) for step s: one 32-byte load each, the
// operand of v_mfma_f32_16x16x32_f16 as it stands once split; three instructions per step
// (h l' + l h' + h h').  The accumulators hold S_u S_v times the score; both scales are powers of
// two, so the product below is exact.  A score depends on its two vectors (and the two scales
// of the call) only.
template <int NV>
struct ScoreTileF16 {
    static constexpr int KP = NV * 16;
    static constexpr int NS = (KP + 31) / 32;   // steps of 32 dimensions
    static constexpr int UT = NS <= 2 ? 4 : 2;  // user tiles per wave (8 NS operand registers each)
};

template <int NS>
__device__ __forceinline__ void split_row(const float *__restrict__ row, int kp, int g, float scale,
                                          f16x8 (&h)[NS], f16x8 (&l)[NS]) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (32 * s + 8 * g < kp) load_vec<float, 8>(x, row + 32 * s + 8 * g);
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
            const f32x2 xs = f32x2{x[j], x[j + 1]} * scale;
            const f16x2 hh = __builtin_convertvector(xs, f16x2);
            const f16x2 ll = split_low(hh, x[j], scale, x[j + 1], scale);
            h[s][j] = hh[0];
            h[s][j + 1] = hh[1];
            l[s][j] = ll[0];
            l[s][j + 1] = ll[1];
        }
    }
}

template <int NV, int MODE>
__global__ __launch_bounds__(256) void topk_scores_f16_kernel(int n_sel,
                                                              const int32_t *__restrict__ sel,
                                                              const float *__restrict__ U,
                                                              const float *__restrict__ V,
                                                              int64_t n_items, ScoreOut<float> o,
                                                              int itw,
                                                              const unsigned *__restrict__ max_bits) {
    constexpr int KP = NV * 16;
    constexpr int NS = ScoreTileF16<NV>::NS;
    constexpr int UT = ScoreTileF16<NV>::UT;
    if (o.run_if && *o.run_if == 0) return;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int wl = threadIdx.x >> 6;
    const int64_t user_base = (int64_t)blockIdx.x * (16 * UT);
    const int64_t n_tiles = MODE == 1 ? o.row_len / 16 : (n_items + 15) / 16;
    const int64_t tile_begin = ((int64_t)blockIdx.y * 4 + wl) * itw;
    if (tile_begin >= n_tiles) return;
    const int64_t tile_end = tile_begin + itw < n_tiles ? tile_begin + itw : n_tiles;
    const float su = split_scale(max_bits[0]), sv = split_scale(max_bits[1]);
    const float unscale = 1.0f / (su * sv);
    f16x8 ah[UT][NS], al[UT][NS];
#pragma unroll
    for (int t = 0; t < UT; ++t) {
        int64_t ui = user_base + 16 * t + q;
        ui = ui < n_sel ? ui : n_sel - 1;  // (rows past the end are computed and not stored)
        split_row<NS>(U + (int64_t)sel[ui] * KP, KP, g, su, ah[t], al[t]);
    }
    const int64_t wchunk = (int64_t)blockIdx.y * 4 + wl;
    float thr[MODE == 2 ? UT : 1][4];
    int cnt[MODE == 2 ? UT : 1][4];
    if constexpr (MODE == 2) {
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int64_t user = user_base + 16 * t + 4 * g + r;
                user = user < n_sel ? user : n_sel - 1;
                thr[t][r] = o.thresh[user * o.thresh_stride];
                cnt[t][r] = 0;
            }
    }
    for (int64_t it = tile_begin; it < tile_end; ++it) {
        const int64_t item = 16 * (MODE == 1 ? it * o.tile_stride : it) + q;
        const bool item_ok = item < n_items;
        f16x8 bh[NS], bl[NS];
        split_row<NS>(V + (item_ok ? item : n_items - 1) * KP, KP, g, sv, bh, bl);
        split_fence();
        Mfma<float>::acc_t acc[UT];
#pragma unroll
        for (int t = 0; t < UT; ++t) acc[t] = Mfma<float>::acc_t{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int t = 0; t < UT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][s], bl[s], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < UT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t][s], bh[s], acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < UT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t][s], bh[s], acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t user = user_base + 16 * t + 4 * g + r;
                const float val = acc[t][r] * unscale;
                if constexpr (MODE == 0) {
                    if (item_ok && user < n_sel) o.scores[user * o.row_len + item] = val;
                } else if constexpr (MODE == 1) {
                    if (item_ok && user < n_sel) o.scores[user * o.row_len + 16 * it + q] = val;
                } else {
                    // "not below the threshold" in floating point keeps a superset of the
                    // scores whose key is at or above the threshold's (-0.0 beside +0.0, NaNs)
                    const bool hit = !(val < thr[t][r]) && item_ok && user < n_sel;
                    push_candidates<float>(o, hit, val, item, user, wchunk, g, q, cnt[t][r]);
                }
            }
    }
    if constexpr (MODE == 2) {
#pragma unroll
        for (int t = 0; t < UT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t user = user_base + 16 * t + 4 * g + r;
                if (q == 0 && user < n_sel) o.count[user * o.n_wchunks + wchunk] = cnt[t][r];
            }
    }
}

// Every thread's share of a row, eight independent loads at a time (one load per loop turn left
// the passes waiting for a cache round trip per element); f(key, item index) per element.
template <typename T, typename F>
__device__ __forceinline__ void scan_row(const T *__restrict__ row, int64_t n_items, int tid, F f) {
    for (int64_t base = tid; base < n_items; base += 256 * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = base + 256 * u;
            v[u] = row[i < n_items ? i : base];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = base + 256 * u;
            if (i < n_items) f(key_of(v[u]), i);
        }
    }
}

// One workgroup per selected user.  LDS: 256-bin histogram, candidate arrays of TOPK_MAX_K.
template <typename T>
__global__ __launch_bounds__(256) void topk_select_kernel(const T *__restrict__ scores,
                                                          int64_t n_items, int k,
                                                          int32_t *__restrict__ out_idx,
                                                          T *__restrict__ out_score,
                                                          const int *__restrict__ run_if) {
    typedef typename KeyT<T>::type key_t;
    if (run_if && *run_if == 0) return;  // (the same for the whole launch)
    __shared__ unsigned hist[256];
    __shared__ key_t prefix_sh;
    __shared__ int want_sh, n_eq_sh;
    __shared__ key_t ckey[TOPK_MAX_K];
    __shared__ int32_t cidx[TOPK_MAX_K];
    __shared__ int n_above_sh;
    const int tid = threadIdx.x;
    const T *row = scores + (int64_t)blockIdx.x * n_items;
    // radix select: after the passes `prefix` is the key of the k-th largest score
    key_t prefix = 0;
    int want = k;  // rank (1-based, from the top) still to be located inside the prefix class
    for (int shift = KeyT<T>::BITS - 8; shift >= 0; shift -= 8) {
        hist[tid] = 0;
        __syncthreads();
        const key_t mask_hi = (shift + 8 >= KeyT<T>::BITS) ? (key_t)0 : (~(key_t)0 << (shift + 8));
        // scores share their leading bits: a thread adds up a run of equal digits before it
        // touches the histogram (one atomic per run instead of one per element on one bin)
        int run_bin = -1;
        unsigned run = 0;
        scan_row(row, n_items, tid, [&](key_t kk, int64_t) {
            if ((kk & mask_hi) != (prefix & mask_hi)) return;
            const int bin = (int)((kk >> shift) & 255);
            if (bin != run_bin) {
                if (run) atomicAdd(&hist[run_bin], run);
                run_bin = bin;
                run = 0;
            }
            ++run;
        });
        if (run) atomicAdd(&hist[run_bin], run);
        __syncthreads();
        if (tid == 0) {
            int acc = 0, b = 255;
            for (; b > 0; --b) {
                if (acc + (int)hist[b] >= want) break;
                acc += (int)hist[b];
            }
            prefix_sh = prefix | ((key_t)b << shift);
            want_sh = want - acc;
            // the class of the chosen digit: after the last pass, the scores equal to the
            // threshold; earlier, as soon as it holds no more scores than are still wanted,
            // all of it is in and the remaining digits need not be looked at
            n_eq_sh = (int)hist[b];
        }
        __syncthreads();
        prefix = prefix_sh;
        want = want_sh;
        const bool settled = n_eq_sh == want;
        __syncthreads();
        if (settled) break;  // (prefix keeps zeros in its lower digits: "kk >= prefix" below)
    }
    // `want` = how many scores equal to the threshold belong to the top k (lowest item ids)
    const int n_eq = n_eq_sh;
    if (tid == 0) {
        n_above_sh = 0;
    }
    __syncthreads();
    const int n_above = k - want;
    if (n_eq == want) {
        // every score equal to the threshold is in: one pass collects all k in any order
        // (sorted below)
        scan_row(row, n_items, tid, [&](key_t kk, int64_t i) {
            if (kk >= prefix) {
                const int at = atomicAdd(&n_above_sh, 1);
                ckey[at] = kk;
                cidx[at] = (int32_t)i;
            }
        });
    } else {
        // ties across the boundary: strictly above in any order; equal: the first `want` in
        // item order, which needs an ordered scan — one wave stepping through the row
        scan_row(row, n_items, tid, [&](key_t kk, int64_t i) {
            if (kk > prefix) {
                const int at = atomicAdd(&n_above_sh, 1);
                ckey[at] = kk;
                cidx[at] = (int32_t)i;
            }
        });
        __syncthreads();
        if (tid < 64) {
            int taken = 0;
            for (int64_t b0 = 0; b0 < n_items && taken < want; b0 += 64) {
                const int64_t i = b0 + tid;
                const bool eq = i < n_items && key_of(row[i]) == prefix;
                const unsigned long long m = __ballot(eq);
                const int before = __popcll(m & ((1ull << tid) - 1ull));
                if (eq && taken + before < want) {
                    ckey[n_above + taken + before] = prefix;
                    cidx[n_above + taken + before] = (int32_t)i;
                }
                taken += __popcll(m);
            }
        }
    }
    __syncthreads();
    // sort the k candidates: key descending, item id ascending (bitonic over the next power
    // of two, padding with the smallest key and the largest id)
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    for (int i = tid; i < np2; i += 256)
        if (i >= k) {
            ckey[i] = 0;
            cidx[i] = INT32_MAX;
        }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < np2 / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));  // index with the `stride` bit clear
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;  // this run sorts "best first"
                const bool lo_better = ckey[lo] > ckey[hi] ||
                                       (ckey[lo] == ckey[hi] && cidx[lo] < cidx[hi]);
                if (lo_better != up) {
                    const key_t tk = ckey[lo];
                    ckey[lo] = ckey[hi];
                    ckey[hi] = tk;
                    const int32_t ti = cidx[lo];
                    cidx[lo] = cidx[hi];
                    cidx[hi] = ti;
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += 256) {
        out_idx[(int64_t)blockIdx.x * k + i] = cidx[i];
        out_score[(int64_t)blockIdx.x * k + i] = row[cidx[i]];
    }
}

// One workgroup per user: the user's candidates (the lists its waves of the score pass left,
// TOPK_CAND in all at most) sorted by (score descending, item id ascending) in LDS, the first k
// written out.  A list longer than its slots, or more candidates than the sort takes, raises
// *overflow (the caller then redoes the call in the two-kernel form); such a user writes nothing.
template <typename T>
__global__ __launch_bounds__(256) void topk_candidates_kernel(const int *__restrict__ count,
                                                              const int32_t *__restrict__ cand_item,
                                                              const T *__restrict__ cand_score,
                                                              int64_t n_wchunks, int k,
                                                              int32_t *__restrict__ out_idx,
                                                              T *__restrict__ out_score,
                                                              int *__restrict__ overflow) {
    typedef typename KeyT<T>::type key_t;
    __shared__ key_t ckey[TOPK_CAND];
    __shared__ int32_t cidx[TOPK_CAND];
    __shared__ int n_sh, bad_sh;
    const int tid = threadIdx.x;
    const int64_t u = blockIdx.x;
    if (tid == 0) {
        n_sh = 0;
        bad_sh = 0;
    }
    __syncthreads();
    for (int64_t w = tid; w < n_wchunks; w += 256) {
        const int c = count[u * n_wchunks + w];
        if (c == 0) continue;
        if (c > TOPK_CHUNK_CAND) {
            bad_sh = 1;
            continue;
        }
        const int at = atomicAdd(&n_sh, c);
        if (at + c > TOPK_CAND) {
            bad_sh = 1;
            continue;
        }
        const int64_t from = (u * n_wchunks + w) * TOPK_CHUNK_CAND;
        for (int i = 0; i < c; ++i) {
            ckey[at + i] = key_of(cand_score[from + i]);
            cidx[at + i] = cand_item[from + i];
        }
    }
    __syncthreads();
    if (bad_sh) {
        if (tid == 0) atomicOr(overflow, 1);
        return;
    }
    const int n = n_sh;
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += 256) {
        ckey[i] = (key_t)0;
        cidx[i] = INT32_MAX;
    }
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < np2 / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const bool lo_better = ckey[lo] > ckey[hi] ||
                                       (ckey[lo] == ckey[hi] && cidx[lo] < cidx[hi]);
                if (lo_better != up) {
                    const key_t tk = ckey[lo];
                    ckey[lo] = ckey[hi];
                    ckey[hi] = tk;
                    const int32_t ti = cidx[lo];
                    cidx[lo] = cidx[hi];
                    cidx[hi] = ti;
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += 256) {
        out_idx[u * k + i] = cidx[i];
        out_score[u * k + i] = score_of(ckey[i]);
    }
}


// ---- fp64 lists through an fp32 filter ---------------------------------------------------------
// Scoring all users x items in fp64 costs sixteen times the matrix-pipe time of the fp32 pass
// on the 16-bit cores, and almost all of those scores are thrown away.  With d_u a bound on
// |s32(u,i) - s64(u,i)| for every item (s32: the fp32 pass on the factors rounded to fp32),
// the k-th largest s32 of a user, t32, is at most d_u above the k-th largest s64, so every item
// of the user's fp64 top k has s32 >= t32 - 2 d_u.  The fp32 filtered form therefore runs with
// its thresholds lowered by 2 d_u, and only the few candidates at or above t32 - 2 d_u are
// scored again in fp64 (from the fp64 tables) and ranked by those scores: the lists are exactly
// those of fp64 scores of all items.  d_u = 2^-13 |u| max_i |v_i|: rounding the factors to fp32
// (2^-23 per product), the two-way fp16 split (2^-20) and 3 x 128 fp32 additions (2^-15.4, were
// every one off by a whole ulp) stay below 2^-15 sum_d |u_d v_d| <= 2^-15 |u| |v_i|.
// Non-finite or overflowing factors make every item a candidate; the lists overflow and the
// call falls back to the dense fp64 form like any other overflow.
constexpr float TOPK_F64_MARGIN = 1.0f / 8192.0f;  // 2^-13

// Rows of an fp64 table (rows[] when not null, else rows 0 .. n_rows-1) to fp32, sixteen lanes per
// row (EL = kp / 16 consecutive dimensions each), the waves striding over the rows; the row's
// Euclidean norm, rounded up (norm_out and/or its maximum in *maxnorm_bits) and the largest
// |fp32 value| (*maxabs_bits); the two maxima as bit patterns of non-negative floats, which
// order like unsigned integers, zeroed by the caller: one atomic per block.
template <int EL>
__global__ __launch_bounds__(256) void topk_to_f32_kernel(int64_t n_rows,
                                                          const int32_t *__restrict__ rows,
                                                          const double *__restrict__ table,
                                                          float *__restrict__ dst,
                                                          float *__restrict__ norm_out,
                                                          unsigned *__restrict__ maxnorm_bits,
                                                          unsigned *__restrict__ maxabs_bits) {
    constexpr int KP = 16 * EL;
    const int lane = threadIdx.x & 63;
    const int sub = lane >> 4, ql = lane & 15;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    float mx = 0.0f, mn = 0.0f;
    bool nan_norm = false;
    for (int64_t r0 = 4 * wave; r0 < n_rows; r0 += 4 * n_waves) {
        const int64_t r = r0 + sub;
        double n2 = 0.0;
        if (r < n_rows) {
            const int64_t src = rows ? (int64_t)rows[r] : r;
            double x[EL];
            load_vec<double, EL>(x, table + src * KP + ql * EL);
            float f[EL];
#pragma unroll
            for (int e = 0; e < EL; ++e) {
                f[e] = (float)x[e];
                n2 += x[e] * x[e];
                mx = fmaxf(mx, fabsf(f[e]));
            }
            typedef float vec_t __attribute__((ext_vector_type(EL)));
            vec_t v;
#pragma unroll
            for (int e = 0; e < EL; ++e) {
                if constexpr (EL == 1) v = f[e];
                else v[e] = f[e];
            }
            *reinterpret_cast<vec_t *>(dst + r * KP + ql * EL) = v;
        }
        n2 += __shfl_xor(n2, 8);
        n2 += __shfl_xor(n2, 4);
        n2 += __shfl_xor(n2, 2);
        n2 += __shfl_xor(n2, 1);
        // (NaN propagates: a NaN norm makes the user's margin NaN and every item a candidate)
        const float nf = (float)sqrt(n2) * 1.0001f;
        if (norm_out && r < n_rows && ql == 0) norm_out[r] = nf;
        nan_norm = nan_norm || nf != nf;
        mn = fmaxf(mn, nf);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, off));
        mn = fmaxf(mn, __shfl_xor(mn, off));
    }
    // one atomic per block and word (they all go to two addresses and are served one by one)
    __shared__ unsigned part[2][4];
    const bool any_nan = __any(nan_norm);
    if (lane == 0) {
        part[0][threadIdx.x >> 6] = any_nan ? 0x7fc00000u : __builtin_bit_cast(unsigned, mn);
        part[1][threadIdx.x >> 6] = __builtin_bit_cast(unsigned, mx);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const unsigned *p4 = part[threadIdx.x];
        const unsigned a = p4[0] > p4[1] ? p4[0] : p4[1], b = p4[2] > p4[3] ? p4[2] : p4[3];
        unsigned *word = threadIdx.x == 0 ? maxnorm_bits : maxabs_bits;
        if (word) atomicMax(word, a > b ? a : b);
    }
}

__global__ __launch_bounds__(256) void iota_kernel(int n, int32_t *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = i;
}

// margin2[u] = 2 d_u and the lowered threshold of the filter pass: the k-th best sampled fp32
// score of user u (sample_best[u * k + k - 1]) minus it.
__global__ __launch_bounds__(256) void topk_f64_margin_kernel(int n_sel, int k,
                                                              const float *__restrict__ sample_best,
                                                              const float *__restrict__ unorm,
                                                              const unsigned *__restrict__ vnorm_bits,
                                                              float *__restrict__ margin2,
                                                              float *__restrict__ thr_low) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= n_sel) return;
    const float vmax = __builtin_bit_cast(float, vnorm_bits[0]);
    const float m2 = 2.0f * TOPK_F64_MARGIN * unorm[u] * vmax;
    margin2[u] = m2;
    thr_low[u] = sample_best[(int64_t)u * k + k - 1] - m2;
}

// One workgroup per user: the user's fp32 candidates sorted by (fp32 score descending, item
// ascending); those at or above (k-th fp32 score - margin2) scored again in fp64 — sixteen lanes
// per candidate, each a fixed share of the dimensions, summed in a fixed order, so a score depends
// on its two vectors only —, sorted by (fp64 score descending, item ascending), the first k
// written out.  Overflowing lists raise *overflow as in topk_candidates_kernel.
template <int NV>
__global__ __launch_bounds__(256) void topk_rescore_kernel(
    const int *__restrict__ count, const int32_t *__restrict__ cand_item,
    const float *__restrict__ cand_score, int64_t n_wchunks, int k, const int32_t *__restrict__ sel,
    const double *__restrict__ U, const double *__restrict__ V, const float *__restrict__ margin2,
    int32_t *__restrict__ out_idx, double *__restrict__ out_score, int *__restrict__ overflow) {
    constexpr int KP = NV * 16;
    __shared__ unsigned ckey[TOPK_CAND];
    __shared__ int32_t cidx[TOPK_CAND];
    __shared__ unsigned long long dkey[TOPK_CAND];
    __shared__ int n_sh, bad_sh, r_sh;
    const int tid = threadIdx.x;
    const int64_t u = blockIdx.x;
    if (tid == 0) {
        n_sh = 0;
        bad_sh = 0;
        r_sh = 0;
    }
    __syncthreads();
    for (int64_t w = tid; w < n_wchunks; w += 256) {
        const int c = count[u * n_wchunks + w];
        if (c == 0) continue;
        if (c > TOPK_CHUNK_CAND) {
            bad_sh = 1;
            continue;
        }
        const int at = atomicAdd(&n_sh, c);
        if (at + c > TOPK_CAND) {
            bad_sh = 1;
            continue;
        }
        const int64_t from = (u * n_wchunks + w) * TOPK_CHUNK_CAND;
        for (int i = 0; i < c; ++i) {
            ckey[at + i] = key_of(cand_score[from + i]);
            cidx[at + i] = cand_item[from + i];
        }
    }
    __syncthreads();
    if (bad_sh || n_sh < k) {  // (fewer than k candidates cannot happen with finite scores)
        if (tid == 0) atomicOr(overflow, 1);
        return;
    }
    const int n = n_sh;
    auto sort32 = [&](int np2) {
        for (int size = 2; size <= np2; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int i = tid; i < np2 / 2; i += 256) {
                    const int lo = 2 * i - (i & (stride - 1));
                    const int hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const bool lo_better = ckey[lo] > ckey[hi] ||
                                           (ckey[lo] == ckey[hi] && cidx[lo] < cidx[hi]);
                    if (lo_better != up) {
                        const unsigned tk = ckey[lo];
                        ckey[lo] = ckey[hi];
                        ckey[hi] = tk;
                        const int32_t ti = cidx[lo];
                        cidx[lo] = cidx[hi];
                        cidx[hi] = ti;
                    }
                }
                __syncthreads();
            }
    };
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    for (int i = n + tid; i < np2; i += 256) {
        ckey[i] = 0u;
        cidx[i] = INT32_MAX;
    }
    __syncthreads();
    sort32(np2);
    // the candidates to score again: the leading ones down to the cut ("not below": NaN stays in)
    const float cut = score_of(ckey[k - 1]) - margin2[u];
    for (int i = tid; i < n; i += 256) {
        const bool in = !(score_of(ckey[i]) < cut);
        const bool next_in = i + 1 < n && !(score_of(ckey[i + 1]) < cut);
        if (in && !next_in) atomicMax(&r_sh, i + 1);
    }
    __syncthreads();
    const int R = r_sh < k ? k : r_sh;
    // fp64 scores: lane ql of a group of 16 takes dimensions ql, ql + 16, ...
    const int grp = tid >> 4, ql = tid & 15;
    const double *urow = U + (int64_t)sel[u] * KP;
    double ur[KP / 16];
#pragma unroll
    for (int j = 0; j < KP / 16; ++j) ur[j] = urow[ql + 16 * j];
    for (int c0 = 0; c0 < R; c0 += 16) {
        const int c = c0 + grp;
        const int32_t item = cidx[c < R ? c : R - 1];
        const double *vrow = V + (int64_t)item * KP;
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < KP / 16; ++j) acc = fma(ur[j], vrow[ql + 16 * j], acc);
        acc += __shfl_xor(acc, 8);
        acc += __shfl_xor(acc, 4);
        acc += __shfl_xor(acc, 2);
        acc += __shfl_xor(acc, 1);
        if (ql == 0 && c < R) dkey[c] = key_of(acc);
    }
    __syncthreads();
    int rp2 = 2;
    while (rp2 < R) rp2 <<= 1;
    for (int i = R + tid; i < rp2; i += 256) {
        dkey[i] = 0ull;
        cidx[i] = INT32_MAX;
    }
    __syncthreads();
    for (int size = 2; size <= rp2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = tid; i < rp2 / 2; i += 256) {
                const int lo = 2 * i - (i & (stride - 1));
                const int hi = lo + stride;
                const bool up = (lo & size) == 0;
                const bool lo_better = dkey[lo] > dkey[hi] ||
                                       (dkey[lo] == dkey[hi] && cidx[lo] < cidx[hi]);
                if (lo_better != up) {
                    const unsigned long long tk = dkey[lo];
                    dkey[lo] = dkey[hi];
                    dkey[hi] = tk;
                    const int32_t ti = cidx[lo];
                    cidx[lo] = cidx[hi];
                    cidx[hi] = ti;
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < k; i += 256) {
        out_idx[u * k + i] = cidx[i];
        out_score[u * k + i] = score_of(dkey[i]);
    }
}

static size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Users per wave of the score kernel of (T, NV).
template <typename T, int NV>
constexpr int score_users_per_wave() {
    if constexpr (sizeof(T) == 4) return 16 * ScoreTileF16<NV>::UT;
    else return 16 * ScoreTile<T, NV>::UT;
}

template <typename T, int NV, int MODE>
static void launch_scores(dim3 grid, hipStream_t st, int n_sel, const int32_t *sel, const T *U,
                          const T *V, int64_t n_items, const ScoreOut<T> &o, int itw,
                          const unsigned *max_bits) {
    if constexpr (sizeof(T) == 4)
        hipLaunchKernelGGL((topk_scores_f16_kernel<NV, MODE>), grid, dim3(256), 0, st, n_sel, sel, U,
                           V, n_items, o, itw, max_bits);
    else
        hipLaunchKernelGGL((topk_scores_kernel<T, NV, MODE>), grid, dim3(256), 0, st, n_sel, sel, U,
                           V, n_items, o, itw);
}

// Sampled item tiles of the filtered form for this (k, n_items), or 0 when the two-kernel form
// is used: the expected number of candidates per user is k * n_items / sampled items, held to a
// third of the slots; the sample must stay below half of the items to be worth its pass.
static int64_t topk_sample_tiles(int64_t n_items, int k) {
    if (n_items < TOPK_FILTER_MIN_ITEMS) return 0;
    int64_t sample = (3 * (int64_t)k * n_items + TOPK_CAND - 1) / TOPK_CAND;
    if (sample < TOPK_SAMPLE_MIN) sample = TOPK_SAMPLE_MIN;
    if (sample < 4 * (int64_t)k) sample = 4 * (int64_t)k;
    const int64_t tiles = (sample + 15) / 16, full_tiles = n_items / 16;
    return 2 * tiles <= full_tiles ? tiles : 0;
}

template <typename T, int NV>
static int run_topk(int n_sel, const int32_t *sel, const void *U, const void *V, int64_t n_items,
                    int k, int32_t *out_idx, void *out_score, void *workspace, hipStream_t st) {
    T *scores = static_cast<T *>(workspace);
    const T *Ut = static_cast<const T *>(U), *Vt = static_cast<const T *>(V);
    T *out_s = static_cast<T *>(out_score);
    // item tiles per wave: long walks when there are many user blocks to fill the chip with,
    // short ones for a handful of users
    constexpr int UPW = score_users_per_wave<T, NV>();
    const int64_t user_blocks = (n_sel + UPW - 1) / UPW;
    const int64_t n_tiles = (n_items + 15) / 16;
    const int itw = user_blocks >= 64 ? 32 : 4;
    const int64_t chunks = (n_tiles + 4 * itw - 1) / (4 * itw);
    ALSMI_REQUIRE(chunks <= 65535, "alsmi_topk_items: too many items (%lld)", (long long)n_items);
    const dim3 grid_all((unsigned)user_blocks, (unsigned)chunks);
    const size_t block_bytes = (size_t)n_sel * (size_t)n_items * sizeof(T);
    // behind the dense block the workspace has 256 spare bytes: the overflow flag of the
    // filtered form and (fp32) the largest |factor| of the selected users and of the items
    char *tail = static_cast<char *>(workspace) + align_up(block_bytes, 16);
    unsigned *max_bits = reinterpret_cast<unsigned *>(tail + 16);
    if constexpr (sizeof(T) == 4) {
        ALSMI_HIP_CHECK(hipMemsetAsync(max_bits, 0, 2 * sizeof(unsigned), st));
        const int kp = NV * 16;
        auto blocks_for = [&](int64_t rows) {
            const int64_t b = (rows * (kp / 4) + 255) / 256;
            return (unsigned)(b < 1 ? 1 : (b > 512 ? 512 : b));
        };
        hipLaunchKernelGGL(topk_max_abs_kernel, dim3(blocks_for(n_sel)), dim3(256), 0, st,
                           (int64_t)n_sel, sel, reinterpret_cast<const float *>(Ut), kp, max_bits);
        hipLaunchKernelGGL(topk_max_abs_kernel, dim3(blocks_for(n_items)), dim3(256), 0, st, n_items,
                           (const int32_t *)nullptr, reinterpret_cast<const float *>(Vt), kp,
                           max_bits + 1);
        ALSMI_HIP_CHECK(hipGetLastError());
    }

    // ---- filtered form (see the head of the file) -------------------------------------------
    const int *flag = nullptr;
    static const bool filter_on = !(getenv("ALSMI_TOPK_FILTER") && atoi(getenv("ALSMI_TOPK_FILTER")) == 0);
    const int64_t s_tiles = filter_on ? topk_sample_tiles(n_items, k) : 0;
    // buffers inside the dense block's memory: sampled scores | counts | candidate items | scores
    const int64_t n_wchunks = (n_tiles + itw - 1) / itw;  // waves of the item direction with work
    const size_t off_count = align_up((size_t)n_sel * s_tiles * 16 * sizeof(T), 256);
    const size_t off_item = off_count + align_up((size_t)n_sel * n_wchunks * sizeof(int), 256);
    const size_t off_score = off_item + align_up((size_t)n_sel * n_wchunks * TOPK_CHUNK_CAND *
                                                 sizeof(int32_t), 256);
    const size_t need = off_score + (size_t)n_sel * n_wchunks * TOPK_CHUNK_CAND * sizeof(T);
    bool prefiltered = false;
    if constexpr (sizeof(T) == 8) {
        // fp64 through the fp32 filter (see topk_rescore_kernel); everything it needs lies in
        // the dense block's memory, which the fallback only touches if the flag is raised
        static const bool pre_on =
            !(getenv("ALSMI_TOPK_F64_PREFILTER") && atoi(getenv("ALSMI_TOPK_F64_PREFILTER")) == 0);
        constexpr int kp = NV * 16;
        constexpr int UPW32 = score_users_per_wave<float, NV>();
        const int64_t ub32 = (n_sel + UPW32 - 1) / UPW32;
        const int itw32 = ub32 >= 64 ? 32 : 4;
        const int64_t chunks32 = (n_tiles + 4 * itw32 - 1) / (4 * itw32);
        const int64_t nw32 = (n_tiles + itw32 - 1) / itw32;
        size_t off = 0;
        auto take = [&](size_t bytes) {
            const size_t at = off;
            off += align_up(bytes, 256);
            return at;
        };
        const size_t o_u32 = take((size_t)n_sel * kp * 4), o_v32 = take((size_t)n_items * kp * 4);
        const size_t o_ident = take((size_t)n_sel * 4), o_unorm = take((size_t)n_sel * 4);
        const size_t o_margin = take((size_t)n_sel * 4), o_thr = take((size_t)n_sel * 4);
        const size_t o_bidx = take((size_t)n_sel * k * 4), o_bs = take((size_t)n_sel * k * 4);
        const size_t o_sample = take((size_t)n_sel * s_tiles * 16 * 4);
        const size_t o_count = take((size_t)n_sel * nw32 * sizeof(int));
        const size_t o_citem = take((size_t)n_sel * nw32 * TOPK_CHUNK_CAND * 4);
        const size_t o_cscore = take((size_t)n_sel * nw32 * TOPK_CHUNK_CAND * 4);
        if (pre_on && s_tiles > 0 && off <= block_bytes && chunks32 <= 65535) {
            char *base = static_cast<char *>(workspace);
            int *flag_w = reinterpret_cast<int *>(tail);
            unsigned *vnorm_bits = reinterpret_cast<unsigned *>(tail + 32);
            float *U32 = reinterpret_cast<float *>(base + o_u32);
            float *V32 = reinterpret_cast<float *>(base + o_v32);
            int32_t *ident = reinterpret_cast<int32_t *>(base + o_ident);
            float *unorm = reinterpret_cast<float *>(base + o_unorm);
            float *margin2 = reinterpret_cast<float *>(base + o_margin);
            float *thr_low = reinterpret_cast<float *>(base + o_thr);
            int32_t *best_idx = reinterpret_cast<int32_t *>(base + o_bidx);
            float *best_s = reinterpret_cast<float *>(base + o_bs);
            ALSMI_HIP_CHECK(hipMemsetAsync(tail, 0, 64, st));  // flag, the two scales, the norm
            auto conv_blocks = [](int64_t rows) {  // four waves of four rows per block and turn
                const int64_t b = (rows + 15) / 16;
                return (unsigned)(b < 1 ? 1 : (b > 256 ? 256 : b));
            };
            hipLaunchKernelGGL((topk_to_f32_kernel<NV>), dim3(conv_blocks(n_sel)), dim3(256), 0, st,
                               (int64_t)n_sel, sel, reinterpret_cast<const double *>(Ut), U32, unorm,
                               (unsigned *)nullptr, max_bits);
            hipLaunchKernelGGL((topk_to_f32_kernel<NV>), dim3(conv_blocks(n_items)), dim3(256), 0,
                               st, n_items, (const int32_t *)nullptr,
                               reinterpret_cast<const double *>(Vt), V32, (float *)nullptr,
                               vnorm_bits, max_bits + 1);
            hipLaunchKernelGGL(iota_kernel, dim3((unsigned)((n_sel + 255) / 256)), dim3(256), 0, st,
                               n_sel, ident);
            ALSMI_HIP_CHECK(hipGetLastError());
            ScoreOut<float> o = {};
            o.scores = reinterpret_cast<float *>(base + o_sample);
            o.row_len = s_tiles * 16;
            o.tile_stride = (n_items / 16) / s_tiles;
            const int64_t s_chunks = (s_tiles + 4 * itw32 - 1) / (4 * itw32);
            launch_scores<float, NV, 1>(dim3((unsigned)ub32, (unsigned)s_chunks), st, n_sel, ident,
                                        U32, V32, n_items, o, itw32, max_bits);
            ALSMI_HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL((topk_select_kernel<float>), dim3(n_sel), dim3(256), 0, st, o.scores,
                               (int64_t)(s_tiles * 16), k, best_idx, best_s, (const int *)nullptr);
            hipLaunchKernelGGL(topk_f64_margin_kernel, dim3((unsigned)((n_sel + 255) / 256)),
                               dim3(256), 0, st, n_sel, k, best_s, unorm, vnorm_bits, margin2,
                               thr_low);
            ALSMI_HIP_CHECK(hipGetLastError());
            o.thresh = thr_low;
            o.thresh_stride = 1;
            o.count = reinterpret_cast<int *>(base + o_count);
            o.cand_item = reinterpret_cast<int32_t *>(base + o_citem);
            o.cand_score = reinterpret_cast<float *>(base + o_cscore);
            o.n_wchunks = nw32;
            launch_scores<float, NV, 2>(dim3((unsigned)ub32, (unsigned)chunks32), st, n_sel, ident,
                                        U32, V32, n_items, o, itw32, max_bits);
            ALSMI_HIP_CHECK(hipGetLastError());
            hipLaunchKernelGGL((topk_rescore_kernel<NV>), dim3(n_sel), dim3(256), 0, st, o.count,
                               o.cand_item, o.cand_score, nw32, k, sel,
                               reinterpret_cast<const double *>(Ut),
                               reinterpret_cast<const double *>(Vt), margin2, out_idx,
                               reinterpret_cast<double *>(out_s), flag_w);
            ALSMI_HIP_CHECK(hipGetLastError());
            flag = flag_w;
            prefiltered = true;
        }
    }
    if (!prefiltered && s_tiles > 0 && need <= block_bytes) {
        char *base = static_cast<char *>(workspace);
        int *flag_w = reinterpret_cast<int *>(tail);
        int *count = reinterpret_cast<int *>(base + off_count);
        ALSMI_HIP_CHECK(hipMemsetAsync(flag_w, 0, sizeof(int), st));
        ScoreOut<T> o = {};
        o.scores = scores;
        o.row_len = s_tiles * 16;
        o.tile_stride = (n_items / 16) / s_tiles;
        const int64_t s_chunks = (s_tiles + 4 * itw - 1) / (4 * itw);
        launch_scores<T, NV, 1>(dim3((unsigned)user_blocks, (unsigned)s_chunks), st, n_sel, sel, Ut,
                                Vt, n_items, o, itw, max_bits);
        ALSMI_HIP_CHECK(hipGetLastError());
        // the k best of the sample, best first: out_s[u * k + k - 1] is the user's threshold
        hipLaunchKernelGGL((topk_select_kernel<T>), dim3(n_sel), dim3(256), 0, st, scores,
                           (int64_t)(s_tiles * 16), k, out_idx, out_s, (const int *)nullptr);
        ALSMI_HIP_CHECK(hipGetLastError());
        o.thresh = out_s + (k - 1);
        o.thresh_stride = k;
        o.count = count;
        o.cand_item = reinterpret_cast<int32_t *>(base + off_item);
        o.cand_score = reinterpret_cast<T *>(base + off_score);
        o.n_wchunks = n_wchunks;
        launch_scores<T, NV, 2>(grid_all, st, n_sel, sel, Ut, Vt, n_items, o, itw, max_bits);
        ALSMI_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL((topk_candidates_kernel<T>), dim3(n_sel), dim3(256), 0, st, count,
                           o.cand_item, o.cand_score, n_wchunks, k, out_idx, out_s, flag_w);
        ALSMI_HIP_CHECK(hipGetLastError());
        flag = flag_w;  // the two kernels below run only if a user overflowed its slots
    }
    ScoreOut<T> all = {};
    all.scores = scores;
    all.row_len = n_items;
    all.run_if = flag;
    launch_scores<T, NV, 0>(grid_all, st, n_sel, sel, Ut, Vt, n_items, all, itw, max_bits);
    ALSMI_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL((topk_select_kernel<T>), dim3(n_sel), dim3(256), 0, st, scores, n_items, k,
                       out_idx, out_s, flag);
    ALSMI_HIP_CHECK(hipGetLastError());
    return ALSMI_OK;
}

}  // namespace alsmi

extern "C" {

size_t alsmi_topk_workspace_bytes(int dtype, int64_t n_sel, int64_t n_items) {
    if ((dtype != ALSMI_F32 && dtype != ALSMI_F64) || n_sel < 0 || n_items < 0) return 0;
    return (size_t)n_sel * (size_t)n_items * (dtype == ALSMI_F32 ? 4 : 8) + 256;
}

int alsmi_topk_items(int dtype, int rank, int64_t n_sel, const int32_t *users,
                     const void *user_factors, const void *item_factors, int64_t n_items, int k,
                     int32_t *items_out, void *scores_out, void *workspace, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(dtype == ALSMI_F32 || dtype == ALSMI_F64, "alsmi_topk_items: bad dtype %d", dtype);
    const int kp = alsmi_padded_rank(rank);
    ALSMI_REQUIRE(kp > 0, "alsmi_topk_items: rank %d outside 1..128", rank);
    ALSMI_REQUIRE(n_sel >= 0 && n_sel < INT32_MAX, "alsmi_topk_items: bad user count");
    ALSMI_REQUIRE(k >= 1 && k <= TOPK_MAX_K && k <= n_items,
                  "alsmi_topk_items: need 1 <= k <= min(%d, n_items), got k=%d n_items=%lld",
                  TOPK_MAX_K, k, (long long)n_items);
    if (n_sel == 0) return ALSMI_OK;
    ALSMI_REQUIRE(users && user_factors && item_factors && items_out && scores_out && workspace,
                  "alsmi_topk_items: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define ALSMI_CASE(T, NV)                                                                      \
    return run_topk<T, NV>((int)n_sel, users, user_factors, item_factors, n_items, k, items_out, \
                           scores_out, workspace, st)
    if (dtype == ALSMI_F32) {
        switch (kp) {
            case 16: ALSMI_CASE(float, 1);
            case 32: ALSMI_CASE(float, 2);
            case 64: ALSMI_CASE(float, 4);
            case 128: ALSMI_CASE(float, 8);
        }
    } else {
        switch (kp) {
            case 16: ALSMI_CASE(double, 1);
            case 32: ALSMI_CASE(double, 2);
            case 64: ALSMI_CASE(double, 4);
            case 128: ALSMI_CASE(double, 8);
        }
    }
#undef ALSMI_CASE
    set_error("alsmi_topk_items: unsupported rank");
    return ALSMI_EINVAL;
}

}  // extern "C"
