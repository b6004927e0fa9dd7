// IMC objective pieces for gfx950: masked residual as an SDDMM over the stored entries, the
// side-feature products as an SpMM fused into the same pass plus two dense MFMA GEMMs.
//
// Replaces the dense per-sample arithmetic of the reference's _cost, _cost_prime and
// _cost_hess (src/imc.py:43-48, :65-71, :91-96).  The reference materialises one feature row
// per SAMPLE (n_s x p and n_s x q); here the attribute matrices stay per user / per item and
// the samples are (user, item) index pairs grouped by user:
//     A   = X M                         dense GEMM   [n_users x q]
//     d_e = <A[u_e], Y[i_e]>            SDDMM over the stored entries
//     W_u = sum_{e in u} w_e Y[i_e]     SpMM, same pass (w = d - r for the gradient, d for Hv)
//     out = X^T W                       dense GEMM, reduction over users split in slices
// The regulariser terms of the reference (lam*||Z||^2 per sample, lam*Z, the scalar lam of the
// Hessian-vector product) are O(p q) and are added by the caller.
//
// Layout: Y and the two workspace tables A, W are row-major with the row stride padded to
// ld = the next of 16/32/64/128/256 that is >= q (pad columns zero) so that 16 lanes read one
// contiguous row, four rows per wave instruction, exactly like the ALS factor gather.  X [n_users x p] and M [p x q] are
// plain row-major.
#include "alsmi_common.h"

namespace alsmi {

constexpr int IMC_MAX_Q = 256;
constexpr int IMC_KSPLIT = 128;
// Slices of the reduction over users in out = X^T W: 128 at 128 x 256 attributes (16 output
// tile blocks per slice); more where the output is smaller, so that the launch still has some
// 2 000 waves (at 32 x 64 attributes one tile block per slice left 128 waves on 1 024 SIMDs) —
// the partial results take the same memory either way.
static int imc_ksplit_max(int p, int q) {
    int64_t s = (int64_t)IMC_KSPLIT * (128 * 256) / ((int64_t)p * q);
    if (s < IMC_KSPLIT) s = IMC_KSPLIT;
    if (s > 2048) s = 2048;
    return (int)s;
}

// ---- C[M x N] = op(A) B on v_mfma_*_16x16x4, a block of TM x TN 16x16 tiles per wave ----------
// TRANS_A = false: A is [M x K] row-major (lda).  TRANS_A = true: A is [K x M] row-major.
// B is [K x NB] row-major (ldb) with NB <= N real columns; output columns NB..N-1 are written
// as zero (that is how a table with a padded row stride gets its zero pad columns).
// Out-of-range operand elements read as zero, so any M, N, K work.  The K range is cut into
// gridDim.z slices; slice z writes its tiles to C + z * slice_stride (partial results, summed
// in slice order by reduce_partials_kernel).  A wave loads TM + TN operand values per TM * TN
// matrix instructions; every tile accumulates over k in a fixed order that does not depend on
// TM, TN or the grid.
template <typename T, bool TRANS_A, int TM, int TN>
__global__ __launch_bounds__(256) void gemm_tiles_kernel(int M, int N, int NB, int64_t K,
                                                         const T *__restrict__ A, int64_t lda,
                                                         const T *__restrict__ B, int64_t ldb,
                                                         T *__restrict__ C, int64_t ldc,
                                                         int64_t slice_stride) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int blocks_n = (N + 16 * TN - 1) / (16 * TN);
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_blocks = (int64_t)((M + 16 * TM - 1) / (16 * TM)) * blocks_n;
    if (blk >= n_blocks) return;
    const int i0 = (int)(blk / blocks_n) * 16 * TM, j0 = (int)(blk % blocks_n) * 16 * TN;
    const int64_t kslice = (K + gridDim.z - 1) / gridDim.z;
    const int64_t kb = (int64_t)blockIdx.z * kslice;
    const int64_t ke = kb + kslice < K ? kb + kslice : K;
    typename Mfma<T>::acc_t acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = typename Mfma<T>::acc_t{0, 0, 0, 0};
    bool row_ok[TM], bcol_ok[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) row_ok[tm] = i0 + 16 * tm + q < M;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bcol_ok[tn] = j0 + 16 * tn + q < NB;
    // Four steps of four k at a time: all their operand loads first, then the matrix
    // instructions, so that a wave has 4 (TM + TN) values in flight instead of TM + TN.
    // Within such a group of 16 k, step u of lane group g takes k = k0 + 4 g + u (any fixed
    // assignment of the k to steps and lane groups will do, the same on both operands): a row
    // of a row-major A then gives a lane its four values as ONE contiguous load, and the four
    // lane groups together read 16 contiguous values of the row.
    constexpr int GEMM_U = 4;
    const bool a_vec = !TRANS_A && (lda % 4) == 0 && (reinterpret_cast<uintptr_t>(A) % (4 * sizeof(T))) == 0;
    for (int64_t k0 = kb; k0 < ke; k0 += 4 * GEMM_U) {
        T a[GEMM_U][TM], b[GEMM_U][TN];
        const int64_t kg = k0 + 4 * g;  // this lane group's four k
        if (a_vec && k0 + 4 * GEMM_U <= ke) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                T v4[4] = {T(0), T(0), T(0), T(0)};
                if (row_ok[tm]) load_vec<T, 4>(v4, A + (int64_t)(i0 + 16 * tm + q) * lda + kg);
#pragma unroll
                for (int u = 0; u < GEMM_U; ++u) a[u][tm] = v4[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < GEMM_U; ++u)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    a[u][tm] = T(0);
                    if (kg + u < ke && row_ok[tm])
                        a[u][tm] = TRANS_A ? A[(kg + u) * lda + i0 + 16 * tm + q]
                                           : A[(int64_t)(i0 + 16 * tm + q) * lda + kg + u];
                }
        }
#pragma unroll
        for (int u = 0; u < GEMM_U; ++u)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                b[u][tn] = T(0);
                if (kg + u < ke && bcol_ok[tn]) b[u][tn] = B[(kg + u) * ldb + j0 + 16 * tn + q];
            }
#pragma unroll
        for (int u = 0; u < GEMM_U; ++u)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = Mfma<T>::mma(a[u][tm], b[u][tn], acc[tm][tn]);
    }
    T *Cz = C + (int64_t)blockIdx.z * slice_stride;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i = i0 + 16 * tm + Mfma<T>::crow(g, r);
                const int jj = j0 + 16 * tn + q;
                if (i < M && jj < N) Cz[(int64_t)i * ldc + jj] = acc[tm][tn][r];
            }
}

constexpr int GEMM_U = 4;        // k steps per operand group (16 k: step u of lane group g = k0 + 4g + u)
constexpr int GEMM_GK = 4 * GEMM_U;

// Operand values of one whole group of 16 k (all of them inside the range): no bounds checks,
// the row / column indices are already clamped to valid ones by the caller.
template <typename T, bool TRANS_A, int TM, int TN>
__device__ __forceinline__ void gemm_load_full(T (&a)[GEMM_U][TM], T (&b)[GEMM_U][TN],
                                               const T *__restrict__ A, int64_t lda,
                                               const T *__restrict__ B, int64_t ldb,
                                               const int (&arow)[TM], const int (&bcol)[TN],
                                               int64_t kg, bool a_vec) {
    if (!TRANS_A && a_vec) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            T v4[4];
            load_vec<T, 4>(v4, A + (int64_t)arow[tm] * lda + kg);
#pragma unroll
            for (int u = 0; u < GEMM_U; ++u) a[u][tm] = v4[u];
        }
    } else {
#pragma unroll
        for (int u = 0; u < GEMM_U; ++u)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                a[u][tm] = TRANS_A ? A[(kg + u) * lda + arow[tm]]
                                   : A[(int64_t)arow[tm] * lda + kg + u];
    }
#pragma unroll
    for (int u = 0; u < GEMM_U; ++u)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[u][tn] = B[(kg + u) * ldb + bcol[tn]];
}

template <typename T, int TM, int TN>
__device__ __forceinline__ void gemm_mma(typename Mfma<T>::acc_t (&acc)[TM][TN],
                                         const T (&a)[GEMM_U][TM], const T (&b)[GEMM_U][TN]) {
#pragma unroll
    for (int u = 0; u < GEMM_U; ++u)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = Mfma<T>::mma(a[u][tm], b[u][tn], acc[tm][tn]);
}

// The same product with the operands buffered twice (fp64: the kernel runs two waves per SIMD
// beside 64 accumulator registers, too few to hide the loads of a group behind the other
// waves' matrix instructions; in fp32 five waves per SIMD do that better than a second buffer:
// 145 against 196 us at 138 493 x 128 x 256).  Same accumulation order as gemm_tiles_kernel.
template <typename T, bool TRANS_A, int TM, int TN>
__global__ __launch_bounds__(256, 2) void gemm_tiles_pipe_kernel(
    int M, int N, int NB, int64_t K, const T *__restrict__ A, int64_t lda,
    const T *__restrict__ B, int64_t ldb, T *__restrict__ C, int64_t ldc, int64_t slice_stride) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int blocks_n = (N + 16 * TN - 1) / (16 * TN);
    const int64_t blk = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_blocks = (int64_t)((M + 16 * TM - 1) / (16 * TM)) * blocks_n;
    if (blk >= n_blocks) return;
    const int i0 = (int)(blk / blocks_n) * 16 * TM, j0 = (int)(blk % blocks_n) * 16 * TN;
    const int64_t kslice = (K + gridDim.z - 1) / gridDim.z;
    const int64_t kb = (int64_t)blockIdx.z * kslice;
    const int64_t ke = kb + kslice < K ? kb + kslice : K;
    typename Mfma<T>::acc_t acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = typename Mfma<T>::acc_t{0, 0, 0, 0};
    // Rows of op(A) beyond M and columns of B beyond NB are read from the last valid row /
    // column instead (an output element depends on its own row and column only; those rows are
    // not stored and those columns are stored as zero), so the loop has no bounds checks.
    int arow[TM], bcol[TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) arow[tm] = min(i0 + 16 * tm + q, M - 1);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) bcol[tn] = min(j0 + 16 * tn + q, NB - 1);
    const bool a_vec = !TRANS_A && (lda % 4) == 0 && (reinterpret_cast<uintptr_t>(A) % (4 * sizeof(T))) == 0;
    // Whole groups of 16 k with two operand buffers: the loads of the next group are issued
    // before the matrix instructions of the current one, so a wave's memory latency runs under
    // its own matrix work.  The groups are accumulated in k order whatever the buffering.
    const int64_t n_full = ke > kb ? (ke - kb) / GEMM_GK : 0;
    const int64_t kg0 = kb + 4 * g;
    T a0[GEMM_U][TM], b0[GEMM_U][TN];
    T a1[GEMM_U][TM], b1[GEMM_U][TN];
    if (n_full > 0)
        gemm_load_full<T, TRANS_A, TM, TN>(a0, b0, A, lda, B, ldb, arow, bcol, kg0, a_vec);
    int64_t i = 0;
#pragma unroll 1
    for (; i + 2 <= n_full; i += 2) {
        gemm_load_full<T, TRANS_A, TM, TN>(a1, b1, A, lda, B, ldb, arow, bcol,
                                           kg0 + (i + 1) * GEMM_GK, a_vec);
        __builtin_amdgcn_sched_barrier(0);
        gemm_mma<T, TM, TN>(acc, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        // (the group after the last whole one is read from the last whole one's place
        // again and never multiplied)
        const int64_t nxt = i + 2 < n_full ? i + 2 : n_full - 1;
        gemm_load_full<T, TRANS_A, TM, TN>(a0, b0, A, lda, B, ldb, arow, bcol,
                                           kg0 + nxt * GEMM_GK, a_vec);
        __builtin_amdgcn_sched_barrier(0);
        gemm_mma<T, TM, TN>(acc, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
    }
    if (i < n_full) gemm_mma<T, TM, TN>(acc, a0, b0);  // an odd number of whole groups
    // the slice's last, partial group: values beyond the slice's end are zero
    const int64_t kt = kb + n_full * GEMM_GK;
    if (kt < ke) {
        const int64_t kg = kt + 4 * g;
#pragma unroll
        for (int u = 0; u < GEMM_U; ++u) {
            const bool k_ok = kg + u < ke;
            const int64_t kk = k_ok ? kg + u : ke - 1;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const T v = TRANS_A ? A[kk * lda + arow[tm]] : A[(int64_t)arow[tm] * lda + kk];
                a0[u][tm] = k_ok ? v : T(0);
            }
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b0[u][tn] = B[kk * ldb + bcol[tn]];
        }
        gemm_mma<T, TM, TN>(acc, a0, b0);
    }
    T *Cz = C + (int64_t)blockIdx.z * slice_stride;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int i_out = i0 + 16 * tm + Mfma<T>::crow(g, r);
                const int jj = j0 + 16 * tn + q;
                if (i_out < M && jj < N)
                    Cz[(int64_t)i_out * ldc + jj] = jj < NB ? acc[tm][tn][r] : T(0);
            }
}

// Launches gemm_tiles_kernel with the tile block that fits the shape (wide outputs 2 x 4 tiles
// per wave, narrow ones fewer, so that no wave multiplies padding).
template <typename T, bool TRANS_A>
static void launch_gemm(int M, int N, int NB, int64_t K, const T *A, int64_t lda, const T *B,
                        int64_t ldb, T *C, int64_t ldc, int ksplit, int64_t slice_stride,
                        hipStream_t st) {
#define ALSMI_GEMM(TM, TN)                                                                       \
    do {                                                                                         \
        const int64_t blocks = (int64_t)((M + 16 * TM - 1) / (16 * TM)) *                        \
                               ((N + 16 * TN - 1) / (16 * TN));                                  \
        auto kern = sizeof(T) == 8 ? gemm_tiles_pipe_kernel<T, TRANS_A, TM, TN>                  \
                                   : gemm_tiles_kernel<T, TRANS_A, TM, TN>;                      \
        hipLaunchKernelGGL(kern, dim3((unsigned)((blocks + 3) / 4), 1, (unsigned)ksplit),        \
                           dim3(256), 0, st, M, N, NB, K, A, lda, B, ldb, C, ldc, slice_stride); \
    } while (0)
    const bool tall = M > 16;
    if (N > 32) {
        if (tall) ALSMI_GEMM(2, 4); else ALSMI_GEMM(1, 4);
    } else if (N > 16) {
        if (tall) ALSMI_GEMM(2, 2); else ALSMI_GEMM(1, 2);
    } else {
        if (tall) ALSMI_GEMM(2, 1); else ALSMI_GEMM(1, 1);
    }
#undef ALSMI_GEMM
}

// out[i] = sum_z partials[z * count + i] in a fixed order: a workgroup takes 64 consecutive
// outputs, its four waves a quarter of the slices each (four running sums per thread, so that
// four loads are in flight), and the sixteen part sums of an output are added in index order.
template <typename T>
__global__ __launch_bounds__(256) void reduce_partials_kernel(const T *__restrict__ partials,
                                                              int nz, int64_t count,
                                                              T *__restrict__ out) {
    __shared__ T part[4][64];
    const int e = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + e;
    const int per = (nz + 3) / 4;
    const int z0 = w * per, z1 = z0 + per < nz ? z0 + per : nz;
    T s[4] = {T(0), T(0), T(0), T(0)};
    if (i < count) {
        int z = z0;
        for (; z + 4 <= z1; z += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s[u] += partials[(int64_t)(z + u) * count + i];
        }
        for (int u = 0; z < z1; ++z, ++u) s[u] += partials[(int64_t)z * count + i];
    }
    part[w][e] = (s[0] + s[1]) + (s[2] + s[3]);
    __syncthreads();
    if (w == 0 && i < count) out[i] = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
}

// ---- fused SDDMM + SpMM over the stored entries ------------------------------------------------
// The entries (grouped by user through indptr) are cut into CHUNKS OF ENTRIES, not into users:
// wave c takes entries [c*CH, (c+1)*CH), finds the user its first entry belongs to by a binary
// search in indptr and walks the users of its range.  A user whose entries lie inside the chunk
// gets its W row written directly; the part of a user that is cut by a chunk border goes to one
// of the chunk's two border slots (head: the user began in an earlier chunk; tail: it goes on
// in the next) and imc_borders_kernel adds the parts of such a user in chunk order.  The work
// per wave is therefore the same whatever the degree distribution — an item with 60 000
// ratings (the item-major pass of the rank-constrained form) no longer serialises on one wave —
// and every sum is formed in a fixed order.  mode 0: w = d - r and the squared error is
// accumulated; mode 1: w = d.  W must be zero on entry (users without entries keep that).
// Entries per chunk: 2048 where users have many entries; fewer where they have few, because a
// wave walks the users of its chunk one after the other (each a chain of dependent loads) and
// only more waves hide that: with one entry per user — the reference's own per-sample form,
// where every sample is its own user — 2048-entry chunks left one wave per SIMD waiting on
// memory (3.5 ms per evaluation at 2 M samples; 128-entry chunks: see the design notes).
constexpr int IMC_CHUNK = 2048;
static int imc_chunk(int64_t n_users, int64_t n_samples) {
    const int64_t mean = n_samples / (n_users > 0 ? n_users : 1);
    return mean < 8 ? 128 : (mean < 64 ? 512 : IMC_CHUNK);
}

template <typename T, int NV>
__global__ __launch_bounds__(256) void imc_rows_kernel(int mode, int64_t n_users, int64_t n_chunks,
                                                       int chunk,
                                                       const int64_t *__restrict__ indptr,
                                                       const int32_t *__restrict__ items,
                                                       const T *__restrict__ ratings,
                                                       const T *__restrict__ Atab,
                                                       const T *__restrict__ Y,
                                                       T *__restrict__ W,
                                                       T *__restrict__ pred_out,
                                                       T *__restrict__ border,
                                                       int64_t *__restrict__ border_user,
                                                       double *__restrict__ partials) {
    constexpr int LD = NV * 16;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int64_t nnz = indptr[n_users];
    const int64_t c0 = c * chunk;
    const int64_t c1 = c0 + chunk < nnz ? c0 + chunk : nnz;
    // the user of entry c0: the last u with indptr[u] <= c0
    int64_t lo = 0, hi = n_users;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= c0) lo = mid; else hi = mid;
    }
    double sse = 0.0;
    // border slots: [c][0] head (user began before c0), [c][1] tail (user goes on after c1)
    if (lane == 0) {
        border_user[2 * c] = -1;
        border_user[2 * c + 1] = -1;
    }
    for (int64_t u = lo; u < n_users; ++u) {
        const int64_t rb = indptr[u], re = indptr[u + 1];
        if (rb >= c1) break;
        const int64_t beg = rb > c0 ? rb : c0, end = re < c1 ? re : c1;
        if (end <= beg) continue;
        T a[NV], wacc[NV];
        load_vec<T, NV>(a, Atab + u * LD + q * NV);
#pragma unroll
        for (int e = 0; e < NV; ++e) wacc[e] = T(0);
        for (int64_t base = beg; base < end; base += WAVE) {
            const int64_t left = end - base;
            const int cnt = left < WAVE ? (int)left : WAVE;
            int it = 0;
            T r = T(0);
            if (lane < cnt) {
                it = items[base + lane];
                if (mode == 0) r = ratings[base + lane];
            }
            const int nsteps = (cnt + 3) >> 2;
            for (int s = 0; s < nsteps; ++s) {
                const int sl = 4 * s + g;
                const int id = __shfl(it, sl);
                const T rr = __shfl(r, sl);
                T y[NV];
                load_vec<T, NV>(y, Y + (int64_t)id * LD + q * NV);
                T d = T(0);
#pragma unroll
                for (int e = 0; e < NV; ++e) d += a[e] * y[e];
                d += __shfl_xor(d, 1);
                d += __shfl_xor(d, 2);
                d += __shfl_xor(d, 4);
                d += __shfl_xor(d, 8);
                T w = T(0);
                if (sl < cnt) {
                    w = mode == 0 ? d - rr : d;
                    if (q == 0) {
                        if (pred_out) pred_out[base + sl] = d;
                        if (mode == 0) {
                            const double err = (double)rr - (double)d;
                            sse += err * err;
                        }
                    }
                }
#pragma unroll
                for (int e = 0; e < NV; ++e) wacc[e] += w * y[e];
            }
        }
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            wacc[e] += __shfl_xor(wacc[e], 16);
            wacc[e] += __shfl_xor(wacc[e], 32);
        }
        const bool head = rb < c0, tail = re > c1;
        T *dst = W + u * LD;
        if (head || tail) {
            const int slot = head ? 0 : 1;  // (a user that covers the whole chunk is its head)
            dst = border + (2 * c + slot) * LD;
            if (lane == 0) border_user[2 * c + slot] = u;
        }
        if (g == 0) {
#pragma unroll
            for (int e = 0; e < NV; ++e) dst[q * NV + e] = wacc[e];
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) partials[c] = sse;
}

// The same pass for users with few entries (chunks of at most 128 entries): FOUR users at a
// time, one per lane group, each walking its own entries.  With one wave per user the wave's 64
// lanes served one or two entries per step and the three dependent loads of a user (extent,
// item index, item row) were waited for one user after the other.  Each user's sums are formed
// in entry order by its own 16 lanes; the border rules are those of imc_rows_kernel.
template <typename T, int NV>
__global__ __launch_bounds__(256) void imc_rows4_kernel(int mode, int64_t n_users, int64_t n_chunks,
                                                        int chunk,
                                                        const int64_t *__restrict__ indptr,
                                                        const int32_t *__restrict__ items,
                                                        const T *__restrict__ ratings,
                                                        const T *__restrict__ Atab,
                                                        const T *__restrict__ Y,
                                                        T *__restrict__ W,
                                                        T *__restrict__ pred_out,
                                                        T *__restrict__ border,
                                                        int64_t *__restrict__ border_user,
                                                        double *__restrict__ partials) {
    constexpr int LD = NV * 16;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, q = lane & 15;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int64_t nnz = indptr[n_users];
    const int64_t c0 = c * chunk;
    const int64_t c1 = c0 + chunk < nnz ? c0 + chunk : nnz;
    int64_t lo = 0, hi = n_users;
    while (hi - lo > 1) {
        const int64_t mid = (lo + hi) >> 1;
        if (indptr[mid] <= c0) lo = mid; else hi = mid;
    }
    double sse = 0.0;
    if (lane == 0) {
        border_user[2 * c] = -1;
        border_user[2 * c + 1] = -1;
    }
    for (int64_t ub = lo; ub < n_users; ub += 4) {
        if (indptr[ub] >= c1) break;  // (the same for the whole wave)
        const int64_t u = ub + g;
        int64_t rb = 0, re = 0;
        if (u < n_users) {
            rb = indptr[u];
            re = indptr[u + 1];
        }
        const int64_t beg = rb > c0 ? rb : c0, end = re < c1 ? re : c1;
        const int cnt = end > beg ? (int)(end - beg) : 0;  // this lane group's entries
        int most = cnt;
        most = max(most, __shfl_xor(most, 16));
        most = max(most, __shfl_xor(most, 32));
        T a[NV], wacc[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) {
            a[e] = T(0);
            wacc[e] = T(0);
        }
        if (cnt > 0) load_vec<T, NV>(a, Atab + u * LD + q * NV);
        for (int t = 0; t < most; ++t) {
            const bool on = t < cnt;
            int it = 0;
            T rr = T(0);
            if (on) {
                it = items[beg + t];
                if (mode == 0) rr = ratings[beg + t];
            }
            T y[NV];
#pragma unroll
            for (int e = 0; e < NV; ++e) y[e] = T(0);
            if (on) load_vec<T, NV>(y, Y + (int64_t)it * LD + q * NV);
            T d = T(0);
#pragma unroll
            for (int e = 0; e < NV; ++e) d += a[e] * y[e];
            d += __shfl_xor(d, 1);
            d += __shfl_xor(d, 2);
            d += __shfl_xor(d, 4);
            d += __shfl_xor(d, 8);
            T w = T(0);
            if (on) {
                w = mode == 0 ? d - rr : d;
                if (q == 0) {
                    if (pred_out) pred_out[beg + t] = d;
                    if (mode == 0) {
                        const double err = (double)rr - (double)d;
                        sse += err * err;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < NV; ++e) wacc[e] += w * y[e];
        }
        if (cnt > 0) {
            const bool head = rb < c0, tail = re > c1;
            T *dst = W + u * LD;
            if (head || tail) {
                const int slot = head ? 0 : 1;
                dst = border + (2 * c + slot) * LD;
                if (q == 0) border_user[2 * c + slot] = u;
            }
#pragma unroll
            for (int e = 0; e < NV; ++e) dst[q * NV + e] = wacc[e];
        }
    }
    sse = wave_sum(sse);
    if (lane == 0) partials[c] = sse;
}

// Adds the border parts of every user that a chunk border cut, in chunk order: the wave of the
// chunk where the user's first entry lies (its tail slot, or its head slot for chunk 0 never —
// a head is by definition not the first part) walks the following chunks' head slots while they
// belong to the same user.
template <typename T>
__global__ __launch_bounds__(256) void imc_borders_kernel(int64_t n_chunks, int ld,
                                                          const T *__restrict__ border,
                                                          const int64_t *__restrict__ border_user,
                                                          T *__restrict__ W) {
    const int lane = threadIdx.x & 63;
    const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= n_chunks) return;
    const int64_t u = border_user[2 * c + 1];
    if (u < 0) return;  // no user leaves this chunk unfinished
    for (int col = lane; col < ld; col += WAVE) {
        T sum = border[(2 * c + 1) * ld + col];
        for (int64_t k = c + 1; k < n_chunks && border_user[2 * k] == u; ++k)
            sum += border[(2 * k) * ld + col];
        W[u * ld + col] = sum;
    }
}

__global__ __launch_bounds__(256) void imc_sum_sse_kernel(const double *__restrict__ partials,
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

// Row stride of Y and of the A / W workspace tables: the next of 16/32/64/128/256 >= q, so
// that each of the 16 lanes of a row group owns a power-of-two number of elements.
static int imc_ld(int q) {
    int ld = 16;
    while (ld < q) ld *= 2;
    return ld;
}

static size_t align256(size_t x) { return (x + 255) / 256 * 256; }

struct ImcWorkspace {
    size_t a_off, w_off, part_off, sse_off, border_off, border_user_off, total;
    int64_t n_chunks;
    int chunk;
};

static ImcWorkspace imc_layout(size_t esz, int64_t n_users, int64_t n_samples, int p, int q) {
    ImcWorkspace ws;
    const size_t table = align256((size_t)n_users * imc_ld(q) * esz);
    ws.chunk = imc_chunk(n_users, n_samples);
    ws.n_chunks = (n_samples + ws.chunk - 1) / ws.chunk;
    if (ws.n_chunks < 1) ws.n_chunks = 1;
    ws.a_off = 0;
    ws.w_off = table;
    ws.part_off = 2 * table;
    ws.sse_off = ws.part_off + align256((size_t)imc_ksplit_max(p, q) * p * q * esz);
    ws.border_off = ws.sse_off + align256((size_t)ws.n_chunks * sizeof(double));
    ws.border_user_off = ws.border_off + align256((size_t)2 * ws.n_chunks * imc_ld(q) * esz);
    ws.total = ws.border_user_off + align256((size_t)2 * ws.n_chunks * sizeof(int64_t));
    return ws;
}

template <typename T, int NV>
static void launch_rows(int64_t n_chunks, int chunk, int mode, int64_t n_users,
                        const int64_t *indptr,
                        const int32_t *items, const void *ratings, const T *Atab, const void *Y,
                        T *Wtab, void *pred_out, T *border, int64_t *border_user,
                        double *sse_part, hipStream_t st) {
    // users with few entries (short chunks): four users per wave
    auto kern = chunk <= 128 ? imc_rows4_kernel<T, NV> : imc_rows_kernel<T, NV>;
    hipLaunchKernelGGL(kern, dim3((unsigned)((n_chunks + 3) / 4)), dim3(256), 0,
                       st, mode, n_users, n_chunks, chunk, indptr, items,
                       static_cast<const T *>(ratings),
                       Atab, static_cast<const T *>(Y), Wtab, static_cast<T *>(pred_out), border,
                       border_user, sse_part);
}

template <typename T>
static int run_imc_eval(int mode, int64_t n_users, int64_t n_samples, int p, int q,
                        const int64_t *indptr, const int32_t *items, const void *ratings,
                        const void *X, const void *Y, const void *Mmat, void *out, void *pred_out,
                        double *sse_out, void *workspace, hipStream_t st) {
    const int ld = imc_ld(q);
    const ImcWorkspace ws = imc_layout(sizeof(T), n_users, n_samples, p, q);
    char *base = static_cast<char *>(workspace);
    T *Atab = reinterpret_cast<T *>(base + ws.a_off);
    T *Wtab = reinterpret_cast<T *>(base + ws.w_off);
    T *part = reinterpret_cast<T *>(base + ws.part_off);
    double *sse_part = reinterpret_cast<double *>(base + ws.sse_off);
    T *border = reinterpret_cast<T *>(base + ws.border_off);
    int64_t *border_user = reinterpret_cast<int64_t *>(base + ws.border_user_off);

    // A = X M with the padded row stride; M has q real columns, the pad columns come out zero
    launch_gemm<T, false>((int)n_users, ld, q, (int64_t)p, static_cast<const T *>(X), (int64_t)p,
                          static_cast<const T *>(Mmat), (int64_t)q, Atab, (int64_t)ld, 1,
                          (int64_t)0, st);
    ALSMI_HIP_CHECK(hipGetLastError());

    // d_e, squared error, W = (weights) x Y in one pass over the stored entries
    ALSMI_HIP_CHECK(hipMemsetAsync(Wtab, 0, (size_t)n_users * ld * sizeof(T), st));
    const int64_t nc = n_samples > 0 ? ws.n_chunks : 0;
    if (nc > 0) {
        switch (ld / 16) {
#define ALSMI_ROWS(NV)                                                                          \
    case NV:                                                                                    \
        launch_rows<T, NV>(nc, ws.chunk, mode, n_users, indptr, items, ratings, Atab, Y, Wtab,  \
                           pred_out,                                                            \
                           border, border_user, sse_part, st);                                  \
        break;
            ALSMI_ROWS(1) ALSMI_ROWS(2) ALSMI_ROWS(4) ALSMI_ROWS(8) ALSMI_ROWS(16)
#undef ALSMI_ROWS
            default:
                set_error("alsmi_imc_eval: unsupported padded width %d", ld);
                return ALSMI_EINVAL;
        }
        ALSMI_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL((imc_borders_kernel<T>), dim3((unsigned)((nc + 3) / 4)), dim3(256), 0, st,
                           nc, ld, border, border_user, Wtab);
        ALSMI_HIP_CHECK(hipGetLastError());
    }
    if (sse_out) {
        hipLaunchKernelGGL(imc_sum_sse_kernel, dim3(1), dim3(256), 0, st, sse_part, (int)nc,
                           sse_out);
        ALSMI_HIP_CHECK(hipGetLastError());
    }

    // out = X^T W: the reduction over users is cut into slices whose partial p x q products
    // are added in slice order (reproducible for a given n_users)
    if (out) {
        int ksplit = (int)((n_users + 255) / 256);
        if (ksplit > imc_ksplit_max(p, q)) ksplit = imc_ksplit_max(p, q);
        if (ksplit < 1) ksplit = 1;
        const int64_t count = (int64_t)p * q;
        launch_gemm<T, true>(p, q, q, n_users, static_cast<const T *>(X), (int64_t)p, Wtab,
                             (int64_t)ld, part, (int64_t)q, ksplit, count, st);
        ALSMI_HIP_CHECK(hipGetLastError());
        hipLaunchKernelGGL((reduce_partials_kernel<T>), dim3((unsigned)((count + 63) / 64)),
                           dim3(256), 0, st, part, ksplit, count, static_cast<T *>(out));
        ALSMI_HIP_CHECK(hipGetLastError());
    }
    return ALSMI_OK;
}

}  // namespace alsmi

extern "C" {

int alsmi_imc_padded_width(int q) {
    if (q < 1 || q > alsmi::IMC_MAX_Q) return 0;
    return alsmi::imc_ld(q);
}

size_t alsmi_imc_workspace_bytes(int dtype, int64_t n_users, int64_t n_samples, int p, int q) {
    if ((dtype != ALSMI_F32 && dtype != ALSMI_F64) || n_users < 0 || n_samples < 0 || p < 1 ||
        q < 1 || q > alsmi::IMC_MAX_Q)
        return 0;
    return alsmi::imc_layout(dtype == ALSMI_F32 ? 4 : 8, n_users, n_samples, p, q).total;
}

int alsmi_imc_project(int dtype, int64_t n_rows, int p, int q, const void *X, const void *M,
                      void *table_out, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(dtype == ALSMI_F32 || dtype == ALSMI_F64, "alsmi_imc_project: bad dtype %d",
                  dtype);
    ALSMI_REQUIRE(p >= 1 && q >= 1 && q <= IMC_MAX_Q,
                  "alsmi_imc_project: need p >= 1 and 1 <= q <= %d (got p=%d q=%d)", IMC_MAX_Q, p, q);
    ALSMI_REQUIRE(n_rows >= 0 && n_rows < INT32_MAX, "alsmi_imc_project: bad row count");
    if (n_rows == 0) return ALSMI_OK;
    ALSMI_REQUIRE(X && M && table_out, "alsmi_imc_project: null pointer");
    const int ld = imc_ld(q);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALSMI_F32)
        launch_gemm<float, false>((int)n_rows, ld, q, (int64_t)p, static_cast<const float *>(X),
                                  (int64_t)p, static_cast<const float *>(M), (int64_t)q,
                                  static_cast<float *>(table_out), (int64_t)ld, 1, (int64_t)0, st);
    else
        launch_gemm<double, false>((int)n_rows, ld, q, (int64_t)p, static_cast<const double *>(X),
                                   (int64_t)p, static_cast<const double *>(M), (int64_t)q,
                                   static_cast<double *>(table_out), (int64_t)ld, 1, (int64_t)0,
                                   st);
    ALSMI_HIP_CHECK(hipGetLastError());
    return ALSMI_OK;
}

int alsmi_imc_eval(int dtype, int mode, int64_t n_users, int64_t n_samples, int p, int q,
                   const int64_t *indptr, const int32_t *items, const void *ratings, const void *X,
                   const void *Y, const void *M, void *out, void *pred_out, double *sse_out,
                   void *workspace, void *stream) {
    using namespace alsmi;
    ALSMI_REQUIRE(dtype == ALSMI_F32 || dtype == ALSMI_F64, "alsmi_imc_eval: bad dtype %d", dtype);
    ALSMI_REQUIRE(mode == 0 || mode == 1, "alsmi_imc_eval: mode must be 0 or 1, got %d", mode);
    ALSMI_REQUIRE(p >= 1 && q >= 1 && q <= IMC_MAX_Q,
                  "alsmi_imc_eval: need p >= 1 and 1 <= q <= %d (got p=%d q=%d)", IMC_MAX_Q, p, q);
    ALSMI_REQUIRE(n_users >= 0 && n_users < INT32_MAX, "alsmi_imc_eval: bad user count");
    ALSMI_REQUIRE(n_samples >= 0, "alsmi_imc_eval: bad sample count");
    ALSMI_REQUIRE(mode == 0 || !sse_out, "alsmi_imc_eval: sse_out only exists in mode 0");
    if (n_users == 0) {
        ALSMI_REQUIRE(!out && !sse_out && !pred_out, "alsmi_imc_eval: no users, nothing to write");
        return ALSMI_OK;
    }
    ALSMI_REQUIRE(indptr && X && Y && M && workspace, "alsmi_imc_eval: null pointer");
    ALSMI_REQUIRE(mode == 1 || ratings, "alsmi_imc_eval: mode 0 needs ratings");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == ALSMI_F32)
        return run_imc_eval<float>(mode, n_users, n_samples, p, q, indptr, items, ratings, X, Y, M,
                                   out, pred_out, sse_out, workspace, st);
    return run_imc_eval<double>(mode, n_users, n_samples, p, q, indptr, items, ratings, X, Y, M,
                                out, pred_out, sse_out, workspace, st);
}

}  // extern "C"
