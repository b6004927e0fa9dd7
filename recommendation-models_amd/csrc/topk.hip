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
#include "alsmi_common.h"

namespace alsmi {

constexpr int TOPK_MAX_K = 1024;

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

template <typename T, int NV>
__global__ __launch_bounds__(256) void topk_scores_kernel(int n_sel, const int32_t *__restrict__ sel,
                                                          const T *__restrict__ U,
                                                          const T *__restrict__ V, int64_t n_items,
                                                          T *__restrict__ scores, int itw) {
    constexpr int KP = NV * 16;
    constexpr int KS = ScoreTile<T, NV>::KS;
    constexpr int UT = ScoreTile<T, NV>::UT;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int wl = threadIdx.x >> 6;
    const int64_t user_base = (int64_t)blockIdx.x * (16 * UT);
    const int64_t n_tiles = (n_items + 15) / 16;
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
    for (int64_t it = tile_begin; it < tile_end; ++it) {
        const int64_t item = 16 * it + q;
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
                if (item_ok && user < n_sel) scores[user * n_items + item] = acc[r];
            }
        }
    }
}

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
                                                          T *__restrict__ out_score) {
    typedef typename KeyT<T>::type key_t;
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

template <typename T, int NV>
static int run_topk(int n_sel, const int32_t *sel, const void *U, const void *V, int64_t n_items,
                    int k, int32_t *out_idx, void *out_score, void *workspace, hipStream_t st) {
    T *scores = static_cast<T *>(workspace);
    // item tiles per wave: long walks when there are many user blocks to fill the chip with,
    // short ones for a handful of users
    const int64_t user_blocks = (n_sel + 16 * ScoreTile<T, NV>::UT - 1) / (16 * ScoreTile<T, NV>::UT);
    const int64_t n_tiles = (n_items + 15) / 16;
    const int itw = user_blocks >= 64 ? 32 : 4;
    const int64_t chunks = (n_tiles + 4 * itw - 1) / (4 * itw);
    ALSMI_REQUIRE(chunks <= 65535, "alsmi_topk_items: too many items (%lld)", (long long)n_items);
    hipLaunchKernelGGL((topk_scores_kernel<T, NV>), dim3((unsigned)user_blocks, (unsigned)chunks),
                       dim3(256), 0, st, n_sel, sel, static_cast<const T *>(U),
                       static_cast<const T *>(V), n_items, scores, itw);
    ALSMI_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL((topk_select_kernel<T>), dim3(n_sel), dim3(256), 0, st, scores, n_items, k,
                       out_idx, static_cast<T *>(out_score));
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
