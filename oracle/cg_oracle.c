/*
 * oracle/cg_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU restatement (plain C, serial, no FMA contraction) of the reference's
 * conjugate-gradient path.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this; the product (libMgcgGpu.so)
 * never links or calls it.
 *
 * PARITY STATUS: "parity unpinned" against recorded reference outputs -- the
 * reference (aokomoriuta/ConjugateGradient) ships no tests, no golden vectors
 * and no recorded output, and none of it can be built in this image (C#,
 * CUDA, Boost, R are absent).  The oracle is anchored
 * instead on the known-answer systems the reference hard-codes (KA-1
 * tridiagonal, KA-2 R/CG.R dense N=21, KA-3 MgcgMain banded |sin(i+j)|),
 * each checked against an independent dense/scipy solve in
 * tests/test_oracle.py with fixtures under tests/golden/.
 *
 * Every function cites the reference file:line whose semantics it follows
 * (paths relative to /root/reference).  Summation order is the reference's:
 * strictly left to right, product rounded before the add (the .NET x64 JIT
 * emits mulsd+addsd, never an FMA), so build with -ffp-contract=off.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- BLAS-1 */

/* Mgcg/cuBlas/Mgcg/SparseMatrix.cs:68-88  SparseMatrix.Multiply
 * answer[i] = 0; for k in [RowOffsets[i], RowOffsets[i+1]): answer[i] += E[k]*v[C[k]]
 * (stored order, columns need not be sorted).                                  */
void oracle_spmv(const double *elements, const int *columnIndeces,
                 const int *rowOffsets, int64_t rowCount,
                 double *answer, const double *vector)
{
    for (int64_t i = 0; i < rowCount; i++) {
        double acc = 0;
        for (int64_t k = rowOffsets[i]; k < rowOffsets[i + 1]; k++) {
            int j = columnIndeces[k];
            double prod = elements[k] * vector[j];
            acc += prod;
        }
        answer[i] = acc;
    }
}

/* Mgcg/cuBlas/Mgcg/LongVector.cs:15-31  Dot: serial left-to-right sum.
 *
 * Mode 1 (oracle_set_dot_mode, tests only) sums the SAME rounded products with Neumaier's compensation, i.e. to the last bit or two of
 * their exact sum.  It is not the reference's arithmetic; it exists to MEASURE the reference order's own rounding error, which is no
 * longer negligible at BASELINE sizes: a serial sum of 1.3e8 nearly equal terms rounds every add to the ulp of a running sum 1e8
 * times larger than the addend, in the same direction for long stretches -- relative error of order 1e-9 at 512^3
 * (tests/test_gpu_fullsize.py), where the device's two-stage tree sums are good to 1e-15.  A HIP result that differs from the
 * reference-order oracle by that much and from the compensated oracle by 1e-13 differs by the reference's rounding, not its own. */
static int g_dot_mode = 0;
void oracle_set_dot_mode(int mode) { g_dot_mode = mode; }
int oracle_get_dot_mode(void) { return g_dot_mode; }

double oracle_dot(const double *left, const double *right, int64_t n)
{
    if (g_dot_mode == 1) {
        double sum = 0, comp = 0;
        for (int64_t i = 0; i < n; i++) {
            double prod = left[i] * right[i];
            double t = sum + prod;
            if (fabs(sum) >= fabs(prod)) comp += (sum - t) + prod;
            else comp += (prod - t) + sum;
            sum = t;
        }
        return sum + comp;
    }
    double answer = 0;
    for (int64_t i = 0; i < n; i++) {
        double prod = left[i] * right[i];
        answer += prod;
    }
    return answer;
}

/* Mgcg/cuBlas/Mgcg/LongVector.cs:41-51  SetAdded: answer = left + a*right
 * (answer may alias left or right, as the callers do).                     */
void oracle_set_added(double *answer, const double *left, const double *right,
                      double a, int64_t n)
{
    for (int64_t i = 0; i < n; i++) {
        double prod = a * right[i];
        answer[i] = left[i] + prod;
    }
}

/* Mgcg/cuBlas/Mgcg/LongVector.cs:58-72  MaxAbsolute. */
double oracle_max_absolute(const double *v, int64_t n)
{
    double m = fabs(v[0]);
    for (int64_t i = 1; i < n; i++) {
        double a = fabs(v[i]);
        m = a > m ? a : m;
    }
    return m;
}

/* cublasDscal as used at Mgcg/cuBlas/MgcgGpu/Mgcg.cu:45 (x *= alpha). */
void oracle_scal(double *x, double alpha, int64_t n)
{
    for (int64_t i = 0; i < n; i++) x[i] = alpha * x[i];
}

/* ------------------------------------------------------- convergence rules */

enum {
    /* Mgcg/cuBlas/Mgcg/ConjugateGradient.cs:56-79 (C# Cpu and ParallelGpu):
     * it < Min -> no; it > Max -> ApplicationException; else res < tol.     */
    ORACLE_RULE_CSHARP = 0,
    /* Mgcg/cuBlas/MgcgGpu/Mgcg.cu:251-252 native Solve:
     * (min <= it) && res < tol; maxIteration never consulted.               */
    ORACLE_RULE_NATIVE = 1,
    /* SimpleConjugateGradient/SimpleConjugateGradient/SimpleConjugateGradient.cu:53,106-107:
     * x zero-filled first; (min < it) && res < tol.                         */
    ORACLE_RULE_SIMPLE = 2,
    /* Mgcg/HandmadeCL/MgcgCL/ConjugateGradientCpu.cs:45-97: residual is the
     * max-norm of r, rr is recomputed at the loop top, C# IsConverged rule. */
    ORACLE_RULE_HANDMADECL = 3,
    /* Mgcg/ViennaCL/Mgcg/ComputerCpu.cpp:42-98: relative test
     * (min < it) && rrNew/rr0 < tol*tol.                                    */
    ORACLE_RULE_VIENNACL = 4
};

/* status codes */
enum { ORACLE_OK = 0, ORACLE_MAXIT_EXCEEDED = 1, ORACLE_HARDCAP = 2, ORACLE_NONFINITE = 3 };

/*
 * The reference CG.  Op order follows
 *   Mgcg/cuBlas/Mgcg/ConjugateGradientCpu.cs:45-98   (primary)
 *   Mgcg/cuBlas/MgcgGpu/Mgcg.cu:217-269               (same sequence, native)
 *   SimpleConjugateGradient/.../SimpleConjugateGradient.cu:68-125
 * x is both initial guess and result.  *iteration receives the zero-based
 * index of the last executed loop body (what C# exposes as Iteration; the
 * native Solve returns that +1, ConjugateGradientSingleGpu.cs:168 subtracts it).
 * trace (optional, length >= traceCap) receives Residual per iteration.
 * hardCap bounds the loop for rules that have no max test of their own.
 */
int oracle_cg(const double *elements, const int *columnIndeces, const int *rowOffsets,
              int64_t count, double *x, const double *b,
              int rule, double allowableResidual, int minIteration, int maxIteration,
              int64_t hardCap,
              int *iteration, double *residual,
              double *trace, int64_t traceCap,
              double *work /* 3*count doubles: r, p, Ap; may be NULL */)
{
    double *own = NULL;
    if (!work) { own = (double *)malloc(sizeof(double) * 3 * (size_t)count); work = own; }
    double *r = work, *p = work + count, *Ap = work + 2 * count;
    int status = ORACLE_OK;

    if (rule == ORACLE_RULE_SIMPLE) memset(x, 0, sizeof(double) * (size_t)count);

    /* (Ap)_0 = A x ; r_0 = b - Ap ; p_0 = r_0 ; rr_0 = r_0.r_0   (ConjugateGradientCpu.cs:54-57) */
    oracle_spmv(elements, columnIndeces, rowOffsets, count, Ap, x);
    oracle_set_added(r, b, Ap, -1, count);
    memcpy(p, r, sizeof(double) * (size_t)count);
    double rr = oracle_dot(r, r, count);
    const double rr0 = rr;
    double res = 0;
    int it;
    for (it = 0;; it++) {
        if (rule == ORACLE_RULE_HANDMADECL) rr = oracle_dot(r, r, count); /* HandmadeCL ConjugateGradientCpu.cs:68 */
        oracle_spmv(elements, columnIndeces, rowOffsets, count, Ap, p);    /* :70 */
        double alpha = rr / oracle_dot(p, Ap, count);                      /* :71 */
        oracle_set_added(x, x, p, alpha, count);                           /* :72 */
        oracle_set_added(r, r, Ap, -alpha, count);                         /* :73 */
        double rrNew = 0;
        if (rule == ORACLE_RULE_HANDMADECL) {
            res = oracle_max_absolute(r, count);                           /* HandmadeCL :75 */
        } else {
            rrNew = oracle_dot(r, r, count);                               /* :74 */
            res = sqrt(rrNew);                                             /* :77 */
        }
        if (trace && it < traceCap) trace[it] = res;

        int converged;
        switch (rule) {
        case ORACLE_RULE_NATIVE:
            converged = (minIteration <= it) && (res < allowableResidual);
            break;
        case ORACLE_RULE_SIMPLE:
            converged = (minIteration < it) && (res < allowableResidual);
            break;
        case ORACLE_RULE_VIENNACL:
            if (trace && it < traceCap) trace[it] = sqrt(rrNew / rr0);
            converged = (minIteration < it) && (rrNew / rr0 < allowableResidual * allowableResidual);
            break;
        default: /* CSHARP, HANDMADECL: ConjugateGradient.cs:56-79 */
            if (it < minIteration) converged = 0;
            else if (it > maxIteration) { status = ORACLE_MAXIT_EXCEEDED; converged = 1; }
            else converged = (res < allowableResidual);
        }
        if (converged) break;
        if (!(res == res) || isinf(res)) { status = ORACLE_NONFINITE; break; }
        if (it + 1 >= hardCap) { status = ORACLE_HARDCAP; break; }

        if (rule == ORACLE_RULE_HANDMADECL) rrNew = oracle_dot(r, r, count); /* HandmadeCL :92 */
        double beta = rrNew / rr;                                          /* :93 */
        oracle_set_added(p, r, p, beta, count);                            /* :94  p = r + beta p */
        rr = rrNew;                                                        /* :95 */
    }
    *iteration = it;
    *residual = res;
    free(own);
    return status;
}

/*
 * Fixed number of CG iterations with no stop test (bench.py cpu_baseline leg:
 * "steps" of the hot path).  Same op sequence as oracle_cg.
 */
void oracle_cg_steps(const double *elements, const int *columnIndeces, const int *rowOffsets,
                     int64_t count, double *x, const double *b, int steps,
                     double *residual, double *work)
{
    double *r = work, *p = work + count, *Ap = work + 2 * count;
    oracle_spmv(elements, columnIndeces, rowOffsets, count, Ap, x);
    oracle_set_added(r, b, Ap, -1, count);
    memcpy(p, r, sizeof(double) * (size_t)count);
    double rr = oracle_dot(r, r, count);
    for (int it = 0; it < steps; it++) {
        oracle_spmv(elements, columnIndeces, rowOffsets, count, Ap, p);
        double alpha = rr / oracle_dot(p, Ap, count);
        oracle_set_added(x, x, p, alpha, count);
        oracle_set_added(r, r, Ap, -alpha, count);
        double rrNew = oracle_dot(r, r, count);
        *residual = sqrt(rrNew);
        double beta = rrNew / rr;
        oracle_set_added(p, r, p, beta, count);
        rr = rrNew;
    }
}

/* --------------------------------------------------- row-range partitioning */

/* Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:271-277: floor(N/ndev) rows
 * per device, the last device takes the remainder.  offsets has ndev+1 entries. */
void oracle_partition(int64_t count, int deviceCount, int64_t *offsets)
{
    offsets[0] = 0;
    for (int i = 1; i < deviceCount; i++)
        offsets[i] = offsets[i - 1] + (int64_t)floor((double)count / deviceCount);
    offsets[deviceCount] = count;
}

/* Mgcg/cuBlas/MgcgGpu/Mgcg.cu:83-84: min / max column id over a row slice. */
void oracle_minmax_column(const int *columnIndeces, const int *rowOffsets,
                          int64_t rowBegin, int64_t rowEnd, int *minJ, int *maxJ)
{
    int lo = INT32_MAX, hi = INT32_MIN;
    for (int64_t k = rowOffsets[rowBegin]; k < rowOffsets[rowEnd]; k++) {
        int j = columnIndeces[k];
        if (j < lo) lo = j;
        if (j > hi) hi = j;
    }
    *minJ = lo; *maxJ = hi;
}

/*
 * Multi-device CG exactly as the host drives it in
 * Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:424-565 with the phase
 * functions Mgcg/cuBlas/MgcgGpu/Mgcg.cu:116-198: every device owns rows
 * [offsets[d], offsets[d+1]), dot products are per-device partial sums added in
 * device-id order (resultsDot.Sum(), :463,499,525).  The halo staging
 * (SyncP/P2Host/P2Device) only moves values of p, so a shared full-length p is
 * numerically identical; what differs from oracle_cg is the association of the
 * dot-product sums.  Stop rule: ConjugateGradient.cs:56-79.
 */
int oracle_cg_parallel_offsets(const double *elements, const int *columnIndeces, const int *rowOffsets,
                               int64_t count, int deviceCount, const int64_t *offsets, double *x, const double *b,
                               double allowableResidual, int minIteration, int maxIteration,
                               int *iteration, double *residual, double *trace, int64_t traceCap);

int oracle_cg_parallel(const double *elements, const int *columnIndeces, const int *rowOffsets,
                       int64_t count, int deviceCount, double *x, const double *b,
                       double allowableResidual, int minIteration, int maxIteration,
                       int *iteration, double *residual, double *trace, int64_t traceCap)
{
    int64_t *off = (int64_t *)malloc(sizeof(int64_t) * (size_t)(deviceCount + 1));
    oracle_partition(count, deviceCount, off);                   /* the reference's rule: floor(count / devices) rows each */
    int st = oracle_cg_parallel_offsets(elements, columnIndeces, rowOffsets, count, deviceCount, off, x, b,
                                        allowableResidual, minIteration, maxIteration, iteration, residual, trace, traceCap);
    free(off);
    return st;
}

/* The same loop over ANY row-range partition offsets[0..deviceCount] (the reference only has the one above; the product also offers
 * slabs of equal nonzero count, which changes nothing but where the partial dot-product sums are cut). */
int oracle_cg_parallel_offsets(const double *elements, const int *columnIndeces, const int *rowOffsets,
                               int64_t count, int deviceCount, const int64_t *off, double *x, const double *b,
                               double allowableResidual, int minIteration, int maxIteration,
                               int *iteration, double *residual, double *trace, int64_t traceCap)
{
    double *r = (double *)malloc(sizeof(double) * 3 * (size_t)count);
    double *p = r + count, *Ap = r + 2 * count;
    int status = ORACLE_OK;

    /* Initialize seeds p[offset..] = x (Mgcg.cu:80); Solve0: Ap = A p; r = b - Ap; p = r; r.r (Mgcg.cu:138-141) */
    memcpy(p, x, sizeof(double) * (size_t)count);
    double rr = 0;
    for (int d = 0; d < deviceCount; d++) {
        int64_t o = off[d], n = off[d + 1] - off[d];
        const int *ro = rowOffsets + o;
        /* local SpMV over the slice's rows with global column ids */
        for (int64_t i = 0; i < n; i++) {
            double acc = 0;
            for (int64_t k = ro[i]; k < ro[i + 1]; k++) { double prod = elements[k] * p[columnIndeces[k]]; acc += prod; }
            Ap[o + i] = acc;
        }
    }
    for (int d = 0; d < deviceCount; d++) {
        int64_t o = off[d], n = off[d + 1] - off[d];
        oracle_set_added(r + o, b + o, Ap + o, -1, n);
    }
    memcpy(p, r, sizeof(double) * (size_t)count);
    for (int d = 0; d < deviceCount; d++) rr += oracle_dot(r + off[d], r + off[d], off[d + 1] - off[d]);

    double res = 0;
    int it;
    for (it = 0;; it++) {
        double pAp = 0;
        for (int d = 0; d < deviceCount; d++) {          /* Solve1, Mgcg.cu:161-162 */
            int64_t o = off[d], n = off[d + 1] - off[d];
            const int *ro = rowOffsets + o;
            for (int64_t i = 0; i < n; i++) {
                double acc = 0;
                for (int64_t k = ro[i]; k < ro[i + 1]; k++) { double prod = elements[k] * p[columnIndeces[k]]; acc += prod; }
                Ap[o + i] = acc;
            }
        }
        for (int d = 0; d < deviceCount; d++) pAp += oracle_dot(p + off[d], Ap + off[d], off[d + 1] - off[d]);
        double alpha = rr / pAp;                          /* :499 */
        double rrNew = 0;
        for (int d = 0; d < deviceCount; d++) {          /* Solve2, Mgcg.cu:181-183 */
            int64_t o = off[d], n = off[d + 1] - off[d];
            oracle_set_added(x + o, x + o, p + o, alpha, n);
            oracle_set_added(r + o, r + o, Ap + o, -alpha, n);
        }
        for (int d = 0; d < deviceCount; d++) rrNew += oracle_dot(r + off[d], r + off[d], off[d + 1] - off[d]);
        res = sqrt(rrNew);                                /* :528 */
        if (trace && it < traceCap) trace[it] = res;
        int converged;
        if (it < minIteration) converged = 0;
        else if (it > maxIteration) { status = ORACLE_MAXIT_EXCEEDED; converged = 1; }
        else converged = (res < allowableResidual);
        if (converged) break;
        if (!(res == res) || isinf(res)) { status = ORACLE_NONFINITE; break; }
        double beta = rrNew / rr;                         /* :544 */
        /* Solve3, Mgcg.cu:197: Scal(p, beta) then Axpy(p += r): p = beta*p + r */
        for (int64_t i = 0; i < count; i++) { double s = beta * p[i]; p[i] = s + r[i]; }
        rr = rrNew;
    }
    *iteration = it;
    *residual = res;
    free(r);
    return status;
}

/* ------------------------------------------------------ problem generators */
/* These fill caller-allocated CSR arrays; they exist so that the cpu_baseline
 * leg can build the 512^3 matrix in seconds.  The product has its own
 * generators (conjugategradient_amd/problems.py and the device generator);
 * tests check the two agree.                                                 */

/* 7-point (nz>1) or 5-point (nz==1) Poisson, diag = 2*dim, off = -1, Dirichlet,
 * lexicographic x-fastest, columns ascending. */
int64_t oracle_poisson_nnz(int nx, int ny, int nz)
{
    int64_t N = (int64_t)nx * ny * nz;
    int64_t nnz = N;
    nnz += 2 * ((int64_t)(nx - 1) * ny * nz);
    nnz += 2 * ((int64_t)nx * (ny - 1) * nz);
    if (nz > 1) nnz += 2 * ((int64_t)nx * ny * (nz - 1));
    return nnz;
}

void oracle_poisson_fill(int nx, int ny, int nz, double *elements, int *columnIndeces, int *rowOffsets)
{
    const double diag = (nz > 1) ? 6.0 : 4.0;
    const int64_t sxy = (int64_t)nx * ny;
    int64_t k = 0, i = 0;
    rowOffsets[0] = 0;
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x = 0; x < nx; x++, i++) {
                if (z > 0)      { elements[k] = -1; columnIndeces[k++] = (int)(i - sxy); }
                if (y > 0)      { elements[k] = -1; columnIndeces[k++] = (int)(i - nx); }
                if (x > 0)      { elements[k] = -1; columnIndeces[k++] = (int)(i - 1); }
                elements[k] = diag; columnIndeces[k++] = (int)i;
                if (x < nx - 1) { elements[k] = -1; columnIndeces[k++] = (int)(i + 1); }
                if (y < ny - 1) { elements[k] = -1; columnIndeces[k++] = (int)(i + nx); }
                if (z < nz - 1) { elements[k] = -1; columnIndeces[k++] = (int)(i + sxy); }
                rowOffsets[i + 1] = (int)k;
            }
}

/* Mgcg/cuBlas/Mgcg/MgcgMain.cs:51-84: banded a_ij = |sin(i+j)|, diagonal =
 * row sum, diagonal stored FIRST in each row, then j ascending (j != i) over
 * [max(0,i-band/2+1), min(count,i+band/2)).  Returns nnz.                   */
int64_t oracle_mgcgmain_fill(int count, int maxNonzero, double *elements, int *columnIndeces, int *rowOffsets)
{
    rowOffsets[0] = 0;
    for (int i = 0; i < count; i++) {
        int rowOffset = rowOffsets[i];
        elements[rowOffset] = 0;
        columnIndeces[rowOffset] = i;
        int nonzeroCount = 1;
        int jlo = i - maxNonzero / 2 + 1; if (jlo < 0) jlo = 0;
        int jhi = i + maxNonzero / 2;     if (jhi > count) jhi = count;
        for (int j = jlo; j < jhi; j++) {
            if (i != j) {
                double a_ij = fabs(sin((double)(i + j)));
                elements[rowOffset + nonzeroCount] = a_ij;
                columnIndeces[rowOffset + nonzeroCount] = j;
                nonzeroCount++;
                elements[rowOffset] += a_ij;
            }
        }
        rowOffsets[i + 1] = rowOffsets[i] + nonzeroCount;
    }
    return rowOffsets[count];
}
