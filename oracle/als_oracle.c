/*
 * CPU restatement (plain C, fp64, optional OpenMP over rows) of the reference's ALS
 * hot path.  TEST INFRASTRUCTURE: the checker for large cases and the timed
 * `cpu_baseline` of bench.py.  Nothing in the product package links or loads it.
 *
 * Parity is pinned through tests/test_oracle_c.py, which checks this file against
 * oracle/als_oracle.py (itself pinned to vectors produced by the upstream reference)
 * and directly against the golden fixtures.
 *
 * Factor tables here are row-major [n x k] (one contiguous vector per user/item); the
 * reference keeps (k, n) arrays (src/als.py:149-151) — the arithmetic is the same.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* inv(A) by Gauss-Jordan elimination with partial pivoting, the role LAPACK's
 * getrf/getri play behind np.linalg.inv (src/als.py:247).  Returns 0 on an exactly
 * zero pivot, which is when numpy raises LinAlgError (src/als.py:248-249). */
static int invert(double *a, double *inv, int k) {
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) inv[i * k + j] = (i == j) ? 1.0 : 0.0;
    for (int c = 0; c < k; ++c) {
        int piv = c;
        double best = fabs(a[c * k + c]);
        for (int r = c + 1; r < k; ++r)
            if (fabs(a[r * k + c]) > best) {
                best = fabs(a[r * k + c]);
                piv = r;
            }
        if (best == 0.0 || best != best) return 0;
        if (piv != c)
            for (int j = 0; j < k; ++j) {
                double t = a[c * k + j];
                a[c * k + j] = a[piv * k + j];
                a[piv * k + j] = t;
                t = inv[c * k + j];
                inv[c * k + j] = inv[piv * k + j];
                inv[piv * k + j] = t;
            }
        const double d = 1.0 / a[c * k + c];
        for (int j = 0; j < k; ++j) {
            a[c * k + j] *= d;
            inv[c * k + j] *= d;
        }
        for (int r = 0; r < k; ++r) {
            if (r == c) continue;
            const double f = a[r * k + c];
            if (f == 0.0) continue;
            for (int j = 0; j < k; ++j) {
                a[r * k + j] -= f * a[c * k + j];
                inv[r * k + j] -= f * inv[c * k + j];
            }
        }
    }
    return 1;
}

/* src/als.py:218-250 for rows [row_begin, row_end): A = S S^T + alpha n I, b = S r,
 * x = inv(A) b, zeros when A is singular.  dst is [(row_end-row_begin) x k]. */
void oracle_half_sweep(const int64_t *indptr, const int32_t *indices, const double *vals,
                       const double *src, int k, double alpha, int64_t row_begin,
                       int64_t row_end, double *dst, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        double *A = (double *)malloc(sizeof(double) * k * k);
        double *Ai = (double *)malloc(sizeof(double) * k * k);
        double *b = (double *)malloc(sizeof(double) * k);
#pragma omp for schedule(dynamic, 16)
        for (int64_t r = row_begin; r < row_end; ++r) {
            const int64_t lo = indptr[r], hi = indptr[r + 1];
            double *x = dst + (r - row_begin) * k;
            memset(A, 0, sizeof(double) * k * k);
            memset(b, 0, sizeof(double) * k);
            for (int64_t t = lo; t < hi; ++t) {
                const double *f = src + (int64_t)indices[t] * k;
                const double rv = vals[t];
                for (int i = 0; i < k; ++i) {
                    const double fi = f[i];
                    b[i] += fi * rv;
                    double *Ar = A + i * k;
                    for (int j = 0; j < k; ++j) Ar[j] += fi * f[j];
                }
            }
            const double ridge = alpha * (double)(hi - lo);
            for (int i = 0; i < k; ++i) A[i * k + i] += ridge;
            if (!invert(A, Ai, k)) {
                memset(x, 0, sizeof(double) * k);
                continue;
            }
            for (int i = 0; i < k; ++i) {
                double s = 0.0;
                for (int j = 0; j < k; ++j) s += Ai[i * k + j] * b[j];
                x[i] = s;
            }
        }
        free(A);
        free(Ai);
        free(b);
    }
}

/* sum over stored entries of rows [row_begin,row_end) of (r - <U_u, V_i>)^2
 * (src/als.py:161-165 before mean and sqrt). */
double oracle_sse(const int64_t *indptr, const int32_t *indices, const double *vals,
                  const double *U, const double *V, int k, int64_t row_begin, int64_t row_end,
                  int threads) {
    double total = 0.0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(dynamic, 64) reduction(+ : total)
    for (int64_t r = row_begin; r < row_end; ++r) {
        const double *u = U + r * k;
        double acc = 0.0;
        for (int64_t t = indptr[r]; t < indptr[r + 1]; ++t) {
            const double *v = V + (int64_t)indices[t] * k;
            double p = 0.0;
            for (int i = 0; i < k; ++i) p += u[i] * v[i];
            const double e = vals[t] - p;
            acc += e * e;
        }
        total += acc;
    }
    return total;
}

/* IMC, index form (src/imc.py:43-48, :65-71, :91-96): with A = X M,
 *   d_e = <A[u_e], Y[i_e]>;   W[u] = sum_e w_e Y[i_e];   out = X^T W
 * mode 0: w = d - r (gradient part), also returns sum (r-d)^2; mode 1: w = d
 * (Hessian-vector part).  Samples are grouped by user through indptr.  out is p x q. */
double oracle_imc_eval(int mode, int64_t n_users, int p, int q, const int64_t *indptr,
                       const int32_t *items, const double *ratings, const double *X,
                       const double *Y, const double *M, double *out, int threads) {
    double sse = 0.0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
    const int nt = omp_get_max_threads();
#else
    const int nt = 1;
#endif
    double *partial = (double *)calloc((size_t)nt * p * q, sizeof(double));
#pragma omp parallel reduction(+ : sse)
    {
#ifdef _OPENMP
        double *mine = partial + (size_t)omp_get_thread_num() * p * q;
#else
        double *mine = partial;
#endif
        double *a = (double *)malloc(sizeof(double) * q);
        double *w = (double *)malloc(sizeof(double) * q);
#pragma omp for schedule(dynamic, 64)
        for (int64_t u = 0; u < n_users; ++u) {
            if (indptr[u + 1] == indptr[u]) continue;
            const double *x = X + u * p;
            for (int j = 0; j < q; ++j) a[j] = 0.0;
            for (int i = 0; i < p; ++i) {
                const double xi = x[i];
                const double *Mr = M + (size_t)i * q;
                for (int j = 0; j < q; ++j) a[j] += xi * Mr[j];
            }
            for (int j = 0; j < q; ++j) w[j] = 0.0;
            for (int64_t e = indptr[u]; e < indptr[u + 1]; ++e) {
                const double *y = Y + (int64_t)items[e] * q;
                double d = 0.0;
                for (int j = 0; j < q; ++j) d += a[j] * y[j];
                double we = d;
                if (mode == 0) {
                    const double err = ratings[e] - d;
                    sse += err * err;
                    we = d - ratings[e];
                }
                for (int j = 0; j < q; ++j) w[j] += we * y[j];
            }
            for (int i = 0; i < p; ++i) {
                const double xi = x[i];
                double *o = mine + (size_t)i * q;
                for (int j = 0; j < q; ++j) o[j] += xi * w[j];
            }
        }
        free(a);
        free(w);
    }
    for (int i = 0; i < p * q; ++i) {
        double s = 0.0;
        for (int t = 0; t < nt; ++t) s += partial[(size_t)t * p * q + i];
        out[i] = s;
    }
    free(partial);
    return sse;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
