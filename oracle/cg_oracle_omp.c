/*
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY -- not part of the shipped library.
 *
 * All-cores (OpenMP) variant of oracle_cg_steps for bench.py's second CPU figure ("additionally an
 * OpenMP all-cores variant, clearly labelled 'not the reference's behaviour'").  The reference's CPU loops are serial
 * (Parallel.For is commented out at Mgcg/cuBlas/Mgcg/SparseMatrix.cs:71 and LongVector.cs:21,44); this file only answers
 * "what would the host cores deliver on the same arithmetic".  Same operation sequence as ConjugateGradientCpu.cs:45-98;
 * the dot products are tree-reduced across threads, so results differ from the serial oracle in the last bits and no
 * parity test uses this file.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <omp.h>

int oracle_omp_threads(void) { return omp_get_max_threads(); }
void oracle_omp_set_threads(int n) { if (n > 0) omp_set_num_threads(n); }

static void spmv(const double *e, const int *c, const int *ro, int64_t n, double *y, const double *x)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) {
        double acc = 0.0;
        for (int k = ro[i]; k < ro[i + 1]; k++) acc += e[k] * x[c[k]];
        y[i] = acc;
    }
}

static double dot(const double *a, const double *b, int64_t n)
{
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (int64_t i = 0; i < n; i++) s += a[i] * b[i];
    return s;
}

/* ans = left + a * right */
static void set_added(double *ans, const double *left, const double *right, double a, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) ans[i] = left[i] + a * right[i];
}

void oracle_cg_steps_omp(const double *elements, const int *columnIndeces, const int *rowOffsets,
                         int64_t count, double *x, const double *b, int steps, double *residual, double *work)
{
    double *r = work, *p = work + count, *Ap = work + 2 * count;
    spmv(elements, columnIndeces, rowOffsets, count, Ap, x);
    set_added(r, b, Ap, -1, count);
    memcpy(p, r, sizeof(double) * (size_t)count);
    double rr = dot(r, r, count);
    for (int it = 0; it < steps; it++) {
        spmv(elements, columnIndeces, rowOffsets, count, Ap, p);
        const double alpha = rr / dot(p, Ap, count);
        set_added(x, x, p, alpha, count);
        set_added(r, r, Ap, -alpha, count);
        const double rrNew = dot(r, r, count);
        *residual = sqrt(rrNew);
        set_added(p, r, p, rrNew / rr, count);
        rr = rrNew;
    }
}
