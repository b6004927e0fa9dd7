// Sparse structure construction on the device: ratings triplets -> CSR, CSR -> CSC.
//
// Replaces two host-side steps of the reference whose cost grows with the number of ratings:
//   * the element-by-element fill of a lil_matrix followed by tocsr() (src/als.py:137-140),
//     whose observable result is: a position written more than once keeps the LAST value, a
//     final value of exactly 0 is not stored, column indices inside a row are sorted;
//   * scipy's csr_tocsc (src/als.py:275-279, which the reference even re-runs per item): entries
//     ordered by column, row ids ascending inside a column.
// Both are a stable sort by an integer key (hipCUB's device radix sort) plus a compaction, so the
// index arrays come out bit-identical to scipy's; the binding checks that against the oracle.
// HBM-bound integer work: no arithmetic on the values beyond the copy (and the cast to the
// arithmetic type of the sweep).
#include <hipcub/hipcub.hpp>

#include "alsmi_common.h"

namespace alsmi {

static size_t align256s(size_t x) { return (x + 255) / 256 * 256; }

static int bits_for(uint64_t max_value) {
    int b = 1;
    while (b < 64 && (max_value >> b) != 0) ++b;
    return b;
}

// workspace carve-up shared by the size query and the two entry points
struct SparseWs {
    size_t keys_in, keys_out, pos_in, pos_out, flags, offs, rows, cub, total;
};

static SparseWs sparse_layout(int64_t n_entries, size_t cub_bytes) {
    SparseWs w;
    const size_t n = (size_t)(n_entries > 0 ? n_entries : 1);
    size_t at = 0;
    auto take = [&](size_t bytes) {
        const size_t here = at;
        at += align256s(bytes);
        return here;
    };
    w.keys_in = take(n * 8);
    w.keys_out = take(n * 8);
    w.pos_in = take(n * 4);
    w.pos_out = take(n * 4);
    w.flags = take((n + 1) * 8);
    w.offs = take((n + 1) * 8);
    w.rows = take(n * 4);
    w.cub = take(cub_bytes);
    w.total = at;
    return w;
}

static size_t cub_temp_bytes(int64_t n_entries) {
    const int n = (int)(n_entries > 0 ? n_entries : 1);
    size_t sort_bytes = 0, scan_bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const uint64_t *)nullptr,
                                             (uint64_t *)nullptr, (const int32_t *)nullptr,
                                             (int32_t *)nullptr, n, 0, 64, (hipStream_t) nullptr);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, (const int64_t *)nullptr,
                                           (int64_t *)nullptr, n + 1, (hipStream_t) nullptr);
    return sort_bytes > scan_bytes ? sort_bytes : scan_bytes;
}

__global__ __launch_bounds__(256) void triplet_keys_kernel(int64_t n_in,
                                                           const int32_t *__restrict__ users,
                                                           const int32_t *__restrict__ items,
                                                           int64_t m, uint64_t *__restrict__ keys,
                                                           int32_t *__restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in) return;
    keys[i] = (uint64_t)users[i] * (uint64_t)m + (uint64_t)items[i];
    pos[i] = (int32_t)i;
}

// keep[i] = 1 when sorted entry i is the last of its run of equal keys and its value is non-zero
__global__ __launch_bounds__(256) void keep_flags_kernel(int64_t n_in,
                                                         const uint64_t *__restrict__ keys,
                                                         const int32_t *__restrict__ pos,
                                                         const double *__restrict__ vals,
                                                         int64_t *__restrict__ keep) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i > n_in) return;
    if (i == n_in) {  // one extra slot so that the exclusive scan yields the total
        keep[i] = 0;
        return;
    }
    const bool last = (i + 1 == n_in) || (keys[i + 1] != keys[i]);
    keep[i] = (last && vals[pos[i]] != 0.0) ? 1 : 0;
}

template <typename T>
__global__ __launch_bounds__(256) void compact_csr_kernel(
    int64_t n_in, int64_t m, const uint64_t *__restrict__ keys, const int32_t *__restrict__ pos,
    const double *__restrict__ vals, const int64_t *__restrict__ keep,
    const int64_t *__restrict__ offs, int32_t *__restrict__ rows_out,
    int32_t *__restrict__ indices_out, T *__restrict__ vals_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_in || !keep[i]) return;
    const int64_t o = offs[i];
    const uint64_t k = keys[i];
    rows_out[o] = (int32_t)(k / (uint64_t)m);
    indices_out[o] = (int32_t)(k % (uint64_t)m);
    vals_out[o] = (T)vals[pos[i]];
}

// ptr[r] = number of entries whose (sorted, ascending) id is < r, for r in [0, n_ids]
__global__ __launch_bounds__(256) void lower_bound_ptr_kernel(int64_t n_ids,
                                                              const int32_t *__restrict__ sorted,
                                                              const int64_t *__restrict__ count_ptr,
                                                              int64_t count_host,
                                                              int64_t *__restrict__ ptr) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r > n_ids) return;
    const int64_t count = count_ptr ? count_ptr[0] : count_host;
    int64_t lo = 0, hi = count;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)sorted[mid] < r)
            lo = mid + 1;
        else
            hi = mid;
    }
    ptr[r] = lo;
}

__global__ __launch_bounds__(256) void column_keys_kernel(int64_t nnz,
                                                          const int32_t *__restrict__ indices,
                                                          uint64_t *__restrict__ keys,
                                                          int32_t *__restrict__ pos) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    keys[i] = (uint64_t)(uint32_t)indices[i];
    pos[i] = (int32_t)i;
}

template <typename T>
__global__ __launch_bounds__(256) void gather_csc_kernel(int64_t nnz,
                                                         const uint64_t *__restrict__ keys,
                                                         const int32_t *__restrict__ pos,
                                                         const int32_t *__restrict__ rows,
                                                         const T *__restrict__ vals,
                                                         int32_t *__restrict__ sorted_cols,
                                                         int32_t *__restrict__ cidx,
                                                         T *__restrict__ cvals) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nnz) return;
    const int32_t p = pos[i];
    sorted_cols[i] = (int32_t)keys[i];
    cidx[i] = rows[p];
    cvals[i] = vals[p];
}

__global__ __launch_bounds__(256) void expand_rows_plain_kernel(int64_t n_rows,
                                                                const int64_t *__restrict__ indptr,
                                                                int32_t *__restrict__ rows) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    for (int64_t r = wave; r < n_rows; r += nw)
        for (int64_t p = indptr[r] + lane; p < indptr[r + 1]; p += WAVE) rows[p] = (int32_t)r;
}

static unsigned blocks_for(int64_t n) { return (unsigned)((n + 255) / 256); }

template <typename T>
static int run_build_csr(int64_t n, int64_t m, int64_t n_in, const int32_t *users,
                         const int32_t *items, const double *vals, int64_t *indptr,
                         int32_t *indices, T *vals_out, int64_t *nnz_out, char *ws_base,
                         const SparseWs &w, size_t cub_bytes, hipStream_t st) {
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws_base + w.keys_in);
    uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws_base + w.keys_out);
    int32_t *pos_in = reinterpret_cast<int32_t *>(ws_base + w.pos_in);
    int32_t *pos_out = reinterpret_cast<int32_t *>(ws_base + w.pos_out);
    int64_t *keep = reinterpret_cast<int64_t *>(ws_base + w.flags);
    int64_t *offs = reinterpret_cast<int64_t *>(ws_base + w.offs);
    int32_t *rows = reinterpret_cast<int32_t *>(ws_base + w.rows);
    void *cub = ws_base + w.cub;
    hipLaunchKernelGGL(triplet_keys_kernel, dim3(blocks_for(n_in)), dim3(256), 0, st, n_in, users,
                       items, m, keys_in, pos_in);
    ALSMI_HIP_CHECK(hipGetLastError());
    const int end_bit = bits_for((uint64_t)n * (uint64_t)m);
    size_t tb = cub_bytes;
    ALSMI_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(cub, tb, keys_in, keys_out, pos_in, pos_out,
                                                       (int)n_in, 0, end_bit, st));
    hipLaunchKernelGGL(keep_flags_kernel, dim3(blocks_for(n_in + 1)), dim3(256), 0, st, n_in,
                       keys_out, pos_out, vals, keep);
    ALSMI_HIP_CHECK(hipGetLastError());
    tb = cub_bytes;
    ALSMI_HIP_CHECK(
        hipcub::DeviceScan::ExclusiveSum(cub, tb, keep, offs, (int)(n_in + 1), st));
    hipLaunchKernelGGL((compact_csr_kernel<T>), dim3(blocks_for(n_in)), dim3(256), 0, st, n_in, m,
                       keys_out, pos_out, vals, keep, offs, rows, indices, vals_out);
    ALSMI_HIP_CHECK(hipGetLastError());
    // offs[n_in] is the number of stored entries
    ALSMI_HIP_CHECK(hipMemcpyAsync(nnz_out, offs + n_in, sizeof(int64_t),
                                   hipMemcpyDeviceToDevice, st));
    hipLaunchKernelGGL(lower_bound_ptr_kernel, dim3(blocks_for(n + 1)), dim3(256), 0, st, n, rows,
                       offs + n_in, (int64_t)0, indptr);
    ALSMI_HIP_CHECK(hipGetLastError());
    return ALSMI_OK;
}

template <typename T>
static int run_csr_to_csc(int64_t n, int64_t m, int64_t nnz, const int64_t *indptr,
                          const int32_t *indices, const T *vals, int64_t *cptr, int32_t *cidx,
                          T *cvals, char *ws_base, const SparseWs &w, size_t cub_bytes,
                          hipStream_t st) {
    uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws_base + w.keys_in);
    uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws_base + w.keys_out);
    int32_t *pos_in = reinterpret_cast<int32_t *>(ws_base + w.pos_in);
    int32_t *pos_out = reinterpret_cast<int32_t *>(ws_base + w.pos_out);
    int32_t *rows = reinterpret_cast<int32_t *>(ws_base + w.rows);
    int32_t *sorted_cols = reinterpret_cast<int32_t *>(ws_base + w.flags);
    void *cub = ws_base + w.cub;
    int64_t want = (n + 3) / 4;
    hipLaunchKernelGGL(expand_rows_plain_kernel, dim3((unsigned)(want < 4096 ? want : 4096)),
                       dim3(256), 0, st, n, indptr, rows);
    ALSMI_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(column_keys_kernel, dim3(blocks_for(nnz)), dim3(256), 0, st, nnz, indices,
                       keys_in, pos_in);
    ALSMI_HIP_CHECK(hipGetLastError());
    size_t tb = cub_bytes;
    ALSMI_HIP_CHECK(hipcub::DeviceRadixSort::SortPairs(cub, tb, keys_in, keys_out, pos_in, pos_out,
                                                       (int)nnz, 0, bits_for((uint64_t)m), st));
    hipLaunchKernelGGL((gather_csc_kernel<T>), dim3(blocks_for(nnz)), dim3(256), 0, st, nnz,
                       keys_out, pos_out, rows, vals, sorted_cols, cidx, cvals);
    ALSMI_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(lower_bound_ptr_kernel, dim3(blocks_for(m + 1)), dim3(256), 0, st, m,
                       sorted_cols, (const int64_t *)nullptr, nnz, cptr);
    ALSMI_HIP_CHECK(hipGetLastError());
    return ALSMI_OK;
}

}  // namespace alsmi

extern "C" {

size_t alsmi_sparse_workspace_bytes(int64_t n_entries) {
    if (n_entries < 0 || n_entries >= INT32_MAX) return 0;
    const size_t cub_bytes = alsmi::cub_temp_bytes(n_entries);
    if (cub_bytes == 0) return 0;  // the size query needs a HIP device
    return alsmi::sparse_layout(n_entries, cub_bytes).total;
}

int alsmi_build_csr(int dtype, int64_t n_rows, int64_t n_cols, int64_t n_in,
                    const int32_t *users, const int32_t *items, const double *vals,
                    int64_t *indptr, int32_t *indices, void *vals_out, int64_t *nnz_out,
                    void *workspace, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(dtype == ALSMI_F32 || dtype == ALSMI_F64, "alsmi_build_csr: bad dtype %d", dtype);
    ALSMI_REQUIRE(n_rows >= 0 && n_cols >= 0 && n_rows < INT32_MAX && n_cols < INT32_MAX,
                  "alsmi_build_csr: bad shape");
    ALSMI_REQUIRE(n_in >= 0 && n_in < INT32_MAX, "alsmi_build_csr: bad entry count");
    ALSMI_REQUIRE(indptr && nnz_out && workspace, "alsmi_build_csr: null pointer");
    ALSMI_REQUIRE(n_in == 0 || (users && items && vals && indices && vals_out),
                  "alsmi_build_csr: null pointer");
    ALSMI_REQUIRE(n_in == 0 || (n_rows > 0 && n_cols > 0), "alsmi_build_csr: entries in an empty shape");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const size_t cub_bytes = cub_temp_bytes(n_in);
    const SparseWs w = sparse_layout(n_in, cub_bytes);
    if (n_in == 0) {
        ALSMI_HIP_CHECK(hipMemsetAsync(indptr, 0, (size_t)(n_rows + 1) * sizeof(int64_t), st));
        ALSMI_HIP_CHECK(hipMemsetAsync(nnz_out, 0, sizeof(int64_t), st));
        return ALSMI_OK;
    }
    char *base = static_cast<char *>(workspace);
    if (dtype == ALSMI_F32)
        return run_build_csr<float>(n_rows, n_cols, n_in, users, items, vals, indptr, indices,
                                    static_cast<float *>(vals_out), nnz_out, base, w, cub_bytes,
                                    st);
    return run_build_csr<double>(n_rows, n_cols, n_in, users, items, vals, indptr, indices,
                                 static_cast<double *>(vals_out), nnz_out, base, w, cub_bytes, st);
}

int alsmi_csr_to_csc(int dtype, int64_t n_rows, int64_t n_cols, int64_t nnz,
                     const int64_t *indptr, const int32_t *indices, const void *vals,
                     int64_t *cptr, int32_t *cidx, void *cvals, void *workspace, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(dtype == ALSMI_F32 || dtype == ALSMI_F64, "alsmi_csr_to_csc: bad dtype %d", dtype);
    ALSMI_REQUIRE(n_rows >= 0 && n_cols >= 0 && n_rows < INT32_MAX && n_cols < INT32_MAX,
                  "alsmi_csr_to_csc: bad shape");
    ALSMI_REQUIRE(nnz >= 0 && nnz < INT32_MAX, "alsmi_csr_to_csc: bad entry count");
    ALSMI_REQUIRE(cptr && workspace, "alsmi_csr_to_csc: null pointer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nnz == 0) {
        ALSMI_HIP_CHECK(hipMemsetAsync(cptr, 0, (size_t)(n_cols + 1) * sizeof(int64_t), st));
        return ALSMI_OK;
    }
    ALSMI_REQUIRE(indptr && indices && vals && cidx && cvals, "alsmi_csr_to_csc: null pointer");
    const size_t cub_bytes = cub_temp_bytes(nnz);
    const SparseWs w = sparse_layout(nnz, cub_bytes);
    char *base = static_cast<char *>(workspace);
    if (dtype == ALSMI_F32)
        return run_csr_to_csc<float>(n_rows, n_cols, nnz, indptr, indices,
                                     static_cast<const float *>(vals), cptr, cidx,
                                     static_cast<float *>(cvals), base, w, cub_bytes, st);
    return run_csr_to_csc<double>(n_rows, n_cols, nnz, indptr, indices,
                                  static_cast<const double *>(vals), cptr, cidx,
                                  static_cast<double *>(cvals), base, w, cub_bytes, st);
}

}  // extern "C"
