// Top-k items per user: batched scores U[u] . V[i] over all items and a per-user selection.
//
// The reference's predict_all (src/als.py:347-364) returns one user's scores for every item
// and leaves the ranking to the caller; a recommender serves "the k best items of these
// users".  Two kernels:
//   scores: one wave per (user, block of 64 items): 16 lanes read one item row, four items per
//           step (the access pattern of the pair-prediction kernel), the user's vector is
//           held in registers.  HBM/L2-bound: n_sel * n_items * ld * s bytes of item rows.
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

template <typename T, int NV>
__global__ __launch_bounds__(256) void topk_scores_kernel(int n_sel, const int32_t *__restrict__ sel,
                                                          const T *__restrict__ U,
                                                          const T *__restrict__ V, int64_t n_items,
                                                          T *__restrict__ scores) {
    constexpr int KP = NV * 16;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int64_t blocks_per_user = (n_items + 255) / 256;
    const int64_t u_at = blockIdx.x / blocks_per_user;
    if (u_at >= n_sel) return;
    const int64_t base = (blockIdx.x % blocks_per_user) * 256 + (threadIdx.x >> 6) * 64;
    if (base >= n_items) return;
    T uv[NV];
    load_vec<T, NV>(uv, U + (int64_t)sel[u_at] * KP + q * NV);
    T *out = scores + u_at * n_items;
    const int cnt = n_items - base < 64 ? (int)(n_items - base) : 64;
    T mine = T(0);
#pragma unroll 4
    for (int s = 0; s < 16; ++s) {
        const int it = 4 * s + g;
        T d = T(0);
        if (it < cnt) {
            T v[NV];
            load_vec<T, NV>(v, V + (base + it) * KP + q * NV);
#pragma unroll
            for (int e = 0; e < NV; ++e) d += uv[e] * v[e];
        }
        d += __shfl_xor(d, 1);
        d += __shfl_xor(d, 2);
        d += __shfl_xor(d, 4);
        d += __shfl_xor(d, 8);
        // lane `it` of the wave keeps item `it`, so that the store below is one coalesced row
        const T got = __shfl(d, 16 * (lane & 3), 64);  // the sum of lane group (lane & 3)
        if ((lane >> 2) == s) mine = got;
    }
    if (lane < cnt) out[base + lane] = mine;
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

// One workgroup per selected user.  LDS: 256-bin histogram, candidate arrays of 2*TOPK_MAX_K.
template <typename T>
__global__ __launch_bounds__(256) void topk_select_kernel(const T *__restrict__ scores,
                                                          int64_t n_items, int k,
                                                          int32_t *__restrict__ out_idx,
                                                          T *__restrict__ out_score) {
    typedef typename KeyT<T>::type key_t;
    __shared__ unsigned hist[256];
    __shared__ key_t prefix_sh;
    __shared__ int want_sh;
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
        for (int64_t i = tid; i < n_items; i += 256) {
            const key_t kk = key_of(row[i]);
            if ((kk & mask_hi) == (prefix & mask_hi)) atomicAdd(&hist[(kk >> shift) & 255], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, b = 255;
            for (; b > 0; --b) {
                if (acc + (int)hist[b] >= want) break;
                acc += (int)hist[b];
            }
            prefix_sh = prefix | ((key_t)b << shift);
            want_sh = want - acc;
        }
        __syncthreads();
        prefix = prefix_sh;
        want = want_sh;
        __syncthreads();
    }
    // `want` = how many scores equal to the threshold belong to the top k (lowest item ids)
    if (tid == 0) {
        n_above_sh = 0;
    }
    __syncthreads();
    const int n_above = k - want;
    // strictly above the threshold: any order (sorted below); equal: the first `want` in item
    // order, which needs an ordered scan — done by one wave stepping through the row
    for (int64_t i = tid; i < n_items; i += 256) {
        const key_t kk = key_of(row[i]);
        if (kk > prefix) {
            const int at = atomicAdd(&n_above_sh, 1);
            ckey[at] = kk;
            cidx[at] = (int32_t)i;
        }
    }
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
    const int64_t blocks = (int64_t)n_sel * ((n_items + 255) / 256);
    ALSMI_REQUIRE(blocks < INT32_MAX, "alsmi_topk_items: too many (user, item block) pairs");
    hipLaunchKernelGGL((topk_scores_kernel<T, NV>), dim3((unsigned)blocks), dim3(256), 0, st, n_sel,
                       sel, static_cast<const T *>(U), static_cast<const T *>(V), n_items, scores);
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
