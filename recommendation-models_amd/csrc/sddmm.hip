// Pair predictions <U[u_p], V[i_p]> and their squared error against stored ratings.
//
// Replaces the reference's gather-multiply-sum over pairs (src/als.py:161-165 train RMSE,
// :316-321 _predict, :383-388 score), which materialises two [n_pairs x rank] temporaries.
// HBM-bound: per pair one row of each factor table (rank*s bytes each), two int32 indices and
// one rating.  Sixteen lanes read one contiguous factor row; a wave handles four pairs per
// step, 64 pairs per index load.  The squared error is accumulated in fp64 and reduced in a
// fixed order (per-block partial sums, then one block adds the partials), so the result is
// reproducible for a given n_pairs.
#include <mutex>

#include "alsmi_common.h"

namespace alsmi {

constexpr int SDDMM_MAX_BLOCKS = 2048;

// Per-block partial sums of alsmi_sddmm_noalloc: device memory that belongs to the library
// image instead of a caller's workspace (16 KiB on every GPU the library runs on, held until
// the process ends).  Host code never uses this name as an address; it asks the runtime for
// the array's address on the current device once (noalloc_scratch) and passes that pointer on.
static __device__ __align__(256) double scratch[SDDMM_MAX_BLOCKS];

constexpr int NOALLOC_MAX_DEVICES = 64;

template <typename T, int NV>
__device__ __forceinline__ T pair_dot(const T *__restrict__ U, const T *__restrict__ V, int uu,
                                      int ii, int q) {
    constexpr int KP = NV * 16;
    T a[NV], b[NV];
    load_vec<T, NV>(a, U + (int64_t)uu * KP + q * NV);
    load_vec<T, NV>(b, V + (int64_t)ii * KP + q * NV);
    T d = T(0);
#pragma unroll
    for (int e = 0; e < NV; ++e) d += a[e] * b[e];
    return d;
}

template <typename T>
__device__ __forceinline__ T group16_sum(T d) {
    d += __shfl_xor(d, 1);
    d += __shfl_xor(d, 2);
    d += __shfl_xor(d, 4);
    d += __shfl_xor(d, 8);
    return d;
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void sddmm_kernel(int64_t n_pairs,
                                                    const int32_t *__restrict__ users,
                                                    const int32_t *__restrict__ items,
                                                    const T *__restrict__ U,
                                                    const T *__restrict__ V,
                                                    const T *__restrict__ vals,
                                                    T *__restrict__ pred_out,
                                                    double *__restrict__ partials) {
    __shared__ double wsum[4];
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int wl = threadIdx.x >> 6;
    const int64_t wave = (int64_t)blockIdx.x * 4 + wl;
    const int64_t stride = (int64_t)gridDim.x * 4 * WAVE;
    double sse = 0.0;
    for (int64_t base = wave * WAVE; base < n_pairs; base += stride) {
        const int64_t left = n_pairs - base;
        const int cnt = left < WAVE ? (int)left : WAVE;
        int u = 0, i = 0;
        T r = T(0);
        if (lane < cnt) {
            u = users[base + lane];
            i = items[base + lane];
            if (vals) r = vals[base + lane];
        }
        const int nsteps = (cnt + 3) >> 2;
        int s = 0;
        for (; s + 4 <= nsteps; s += 4) {
            T d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sl = 4 * (s + k) + g;
                d[k] = pair_dot<T, NV>(U, V, __shfl(u, sl), __shfl(i, sl), q);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int sl = 4 * (s + k) + g;
                const T pr = group16_sum(d[k]);
                const T rr = __shfl(r, sl);
                if (q == 0 && sl < cnt) {
                    if (pred_out) pred_out[base + sl] = pr;
                    const double e = (double)rr - (double)pr;
                    sse += e * e;
                }
            }
        }
        for (; s < nsteps; ++s) {
            const int sl = 4 * s + g;
            const T pr = group16_sum(pair_dot<T, NV>(U, V, __shfl(u, sl), __shfl(i, sl), q));
            const T rr = __shfl(r, sl);
            if (q == 0 && sl < cnt) {
                if (pred_out) pred_out[base + sl] = pr;
                const double e = (double)rr - (double)pr;
                sse += e * e;
            }
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) wsum[wl] = sse;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ partials,
                                                           int n, double *__restrict__ out) {
    __shared__ double sm[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partials[i];
    sm[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sm[threadIdx.x] += sm[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = sm[0];
}

__global__ __launch_bounds__(256) void expand_rows_kernel(const int64_t *__restrict__ indptr,
                                                          int64_t row_begin, int64_t row_end,
                                                          int32_t *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    const int64_t off = indptr[row_begin];
    for (int64_t r = row_begin + wave; r < row_end; r += nw) {
        const int64_t b = indptr[r] - off, e = indptr[r + 1] - off;
        for (int64_t p = b + lane; p < e; p += WAVE) out[p] = (int32_t)r;
    }
}

template <typename T, int NV>
static int launch_sddmm(int64_t n_pairs, const int32_t *users, const int32_t *items,
                        const void *U, const void *V, const void *vals, void *pred,
                        double *sse_out, void *workspace, hipStream_t st) {
    int64_t want = (n_pairs + 4 * WAVE - 1) / (4 * WAVE);
    int grid = (int)(want < SDDMM_MAX_BLOCKS ? want : SDDMM_MAX_BLOCKS);
    double *partials = static_cast<double *>(workspace);
    if (grid > 0) {
        hipLaunchKernelGGL((sddmm_kernel<T, NV>), dim3(grid), dim3(256), 0, st, n_pairs, users,
                           items, static_cast<const T *>(U), static_cast<const T *>(V),
                           static_cast<const T *>(vals), static_cast<T *>(pred), partials);
        ALSMI_HIP_CHECK(hipGetLastError());
    }
    if (sse_out) {
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, st, partials, grid,
                           sse_out);
        ALSMI_HIP_CHECK(hipGetLastError());
    }
    return ALSMI_OK;
}

// Address of `scratch` on the current device, looked up once per device (each GPU holds its own
// copy of the array at its own address).
static int noalloc_scratch(double **out) {
    static std::mutex lock;
    static double *per_device[NOALLOC_MAX_DEVICES] = {};
    int dev = -1;
    ALSMI_HIP_CHECK(hipGetDevice(&dev));
    ALSMI_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES,
                  "alsmi_sddmm_noalloc: device ordinal %d outside 0..%d", dev,
                  NOALLOC_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> hold(lock);
    if (!per_device[dev]) {
        void *ptr = nullptr;
        ALSMI_HIP_CHECK(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        ALSMI_REQUIRE(ptr, "alsmi_sddmm_noalloc: the scratch array has no address on device %d",
                      dev);
        per_device[dev] = static_cast<double *>(ptr);
    }
    *out = per_device[dev];
    return ALSMI_OK;
}

}  // namespace alsmi

extern "C" {

size_t alsmi_sddmm_workspace_bytes(void) { return alsmi::SDDMM_MAX_BLOCKS * sizeof(double); }

int alsmi_sddmm(int dtype, int rank, int64_t n_pairs, const int32_t *users, const int32_t *items,
                const void *row_factors, const void *col_factors, const void *vals,
                void *pred_out, double *sse_out, void *workspace, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(dtype == ALSMI_F32 || dtype == ALSMI_F64, "alsmi_sddmm: bad dtype %d", dtype);
    const int kp = alsmi_padded_rank(rank);
    ALSMI_REQUIRE(kp > 0, "alsmi_sddmm: rank %d outside 1..128", rank);
    ALSMI_REQUIRE(n_pairs >= 0, "alsmi_sddmm: negative pair count");
    ALSMI_REQUIRE(workspace, "alsmi_sddmm: null workspace");
    ALSMI_REQUIRE(!sse_out || vals || n_pairs == 0, "alsmi_sddmm: sse_out needs vals");
    if (n_pairs > 0)
        ALSMI_REQUIRE(users && items && row_factors && col_factors, "alsmi_sddmm: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define ALSMI_CASE(T, NV)                                                                       \
    return launch_sddmm<T, NV>(n_pairs, users, items, row_factors, col_factors, vals, pred_out, \
                               sse_out, workspace, st)
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
    set_error("alsmi_sddmm: unsupported rank");
    return ALSMI_EINVAL;
}

int alsmi_sddmm_noalloc(int dtype, int rank, int64_t n_pairs, const int32_t *users,
                        const int32_t *items, const void *row_factors, const void *col_factors,
                        const void *vals, void *pred_out, double *sse_out, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(n_pairs <= ALSMI_SDDMM_NOALLOC_MAX_PAIRS,
                  "alsmi_sddmm_noalloc: %lld pairs exceed the limit of %lld of the entry point "
                  "that allocates nothing; use alsmi_sddmm with a workspace",
                  (long long)n_pairs, (long long)ALSMI_SDDMM_NOALLOC_MAX_PAIRS);
    double *ws = nullptr;
    const int rc = noalloc_scratch(&ws);
    if (rc != ALSMI_OK) return rc;
    // Every partial the final sum reads is written by this call's own blocks first (both run
    // in stream order), so nothing of the previous call's contents has to be cleared.
    return alsmi_sddmm(dtype, rank, n_pairs, users, items, row_factors, col_factors, vals,
                       pred_out, sse_out, ws, stream);
}

int alsmi_expand_rows(const int64_t *indptr, int64_t row_begin, int64_t row_end,
                      int32_t *rows_out, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(0 <= row_begin && row_begin <= row_end, "alsmi_expand_rows: bad row range");
    if (row_begin == row_end) return ALSMI_OK;
    ALSMI_REQUIRE(indptr && rows_out, "alsmi_expand_rows: null pointer");
    int64_t want = (row_end - row_begin + 3) / 4;
    int grid = (int)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(expand_rows_kernel, dim3(grid), dim3(256), 0,
                       static_cast<hipStream_t>(stream), indptr, row_begin, row_end, rows_out);
    ALSMI_HIP_CHECK(hipGetLastError());
    return ALSMI_OK;
}

}  // extern "C"
