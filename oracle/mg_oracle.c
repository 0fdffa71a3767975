/*
 * oracle/mg_oracle.c -- TEST INFRASTRUCTURE ONLY.  Not part of the product.
 *
 * CPU statement of the multigrid-preconditioned CG ("MGCG") that the north
 * star asks for.  PARITY UNPINNED: the reference never implemented its
 * multigrid -- the only traces are the class doc-comment
 * Mgcg/cuBlas/Mgcg/MgcgMain.cs:8, the stale comments "p_0 = (LDLr)_0" /
 * "beta = r'r'/rLDLr" (Mgcg/HandmadeCL/MgcgCL/ConjugateGradientSingleGpu.cs:244,281,
 * Mgcg/ViennaCL/Mgcg/ComputerGpu.cpp:47,85) and a commented-out Jacobi
 * preconditioner call (Mgcg/ViennaCL/Mgcg/ComputerGpu.cpp:96-101).  The
 * algorithm below is therefore DEFINED by this build;
 * tests/test_mg_oracle.py checks it against explicit scipy.sparse R, P, A_c
 * matrices and against multigrid theory (symmetry, contraction), and the HIP
 * path is compared with this file.
 *
 * The CG shell around the preconditioner keeps the reference's op order and
 * stop rule (Mgcg/cuBlas/Mgcg/ConjugateGradientCpu.cs:45-98,
 * Mgcg/cuBlas/Mgcg/ConjugateGradient.cs:56-79).
 *
 * Definition (cell-centred geometric multigrid on an nx*ny*nz lexicographic
 * grid, x fastest; a dimension of extent 1 is not coarsened):
 *   P  : piecewise-constant prolongation, child (x,y,z) <- parent (x/2,y/2,z/2)
 *   R  : P^T (sum over the 8 (4 in 2-D) children)
 *   A_c: sigma * P^T A P, sigma = 1/2 by default (the classic over-correction
 *        fix for piecewise-constant aggregation; for the 7-point Laplacian
 *        (6,-1) it reproduces the rediscretised operator 2*(6,-1))
 *   S  : weighted Jacobi  x <- x + omega * (D^-1 * (b - A x))
 *   V(nu,nu) with zero initial guess on every level, nuC Jacobi sweeps on the
 *   coarsest level.  nu pre == nu post and R = P^T make M^-1 symmetric.
 *   Optional (oracle_mg_set_interpolation(H, 1)): P = cell-centred (tri)linear
 *   interpolation -- per coarsened dimension child i takes 3/4 of its parent
 *   i/2 and 1/4 of the parent's neighbour on the child's side (i even: i/2-1,
 *   i odd: i/2+1; a neighbour outside the grid contributes nothing) -- and
 *   R = P^T; the coarse operators stay the ones above.
 * Arithmetic order is fixed (stated at each function) so that the HIP kernels,
 * built with -ffp-contract=off, reproduce M^-1 r bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void   oracle_spmv(const double *, const int *, const int *, int64_t, double *, const double *);
double oracle_dot(const double *, const double *, int64_t);
void   oracle_set_added(double *, const double *, const double *, double, int64_t);

typedef struct {
    int nx, ny, nz;
    int64_t n, nnz;
    double *elements; int *columnIndeces; int *rowOffsets;   /* owned unless level 0 */
    double *dinv;                                            /* 1 / a_ii */
    double *x, *b, *r, *t;                                   /* work vectors (x, rhs, residual, jacobi temp) */
    int owns_matrix;
} mg_level;

typedef struct {
    int levels;
    double omega;
    int nu, nuCoarse;
    int interp;                /* 0: piecewise constant, 1: cell-centred linear */
    mg_level *lv;
} mg_hierarchy;

/* D^-1: first stored entry of row i whose column is i (rows may be unsorted,
 * e.g. diagonal-first as in Mgcg/cuBlas/Mgcg/MgcgMain.cs:59-60). */
static void extract_dinv(const mg_level *L, double *dinv)
{
    for (int64_t i = 0; i < L->n; i++) {
        double d = 0;
        for (int64_t k = L->rowOffsets[i]; k < L->rowOffsets[i + 1]; k++)
            if (L->columnIndeces[k] == i) { d = L->elements[k]; break; }
        dinv[i] = 1.0 / d;
    }
}

/*
 * Galerkin coarse operator sigma * P^T A P on the 27-slot neighbourhood of
 * each coarse cell.  Order of accumulation per coarse row: children in
 * lexicographic order (z, y, x), each child's entries in stored order; the
 * scale is applied once at the end (value = sigma * sum).  Emitted columns:
 * every touched slot, ascending.  Returns 0, or -1 if some fine column's
 * parent lies outside the 3x3x3 neighbourhood (structure not supported).
 * Pass elementsC == NULL to count only (rowOffsetsC gets filled).
 */
int oracle_mg_galerkin(int nx, int ny, int nz,
                       const double *elements, const int *columnIndeces, const int *rowOffsets,
                       double sigma,
                       double *elementsC, int *columnIndecesC, int *rowOffsetsC)
{
    const int cx = nx > 1 ? 2 : 1, cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const int NX = nx / cx, NY = ny / cy, NZ = nz / cz;
    int64_t I = 0, kout = 0;
    rowOffsetsC[0] = 0;
    for (int Z = 0; Z < NZ; Z++)
        for (int Y = 0; Y < NY; Y++)
            for (int X = 0; X < NX; X++, I++) {
                double acc[27]; int touched[27];
                for (int s = 0; s < 27; s++) { acc[s] = 0; touched[s] = 0; }
                for (int dz = 0; dz < cz; dz++)
                    for (int dy = 0; dy < cy; dy++)
                        for (int dx = 0; dx < cx; dx++) {
                            int64_t i = ((int64_t)(Z * cz + dz) * ny + (Y * cy + dy)) * nx + (X * cx + dx);
                            for (int64_t k = rowOffsets[i]; k < rowOffsets[i + 1]; k++) {
                                int64_t j = columnIndeces[k];
                                int jx = (int)(j % nx), jy = (int)((j / nx) % ny), jz = (int)(j / ((int64_t)nx * ny));
                                int ox = jx / cx - X, oy = jy / cy - Y, oz = jz / cz - Z;
                                if (ox < -1 || ox > 1 || oy < -1 || oy > 1 || oz < -1 || oz > 1) return -1;
                                int s = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1);
                                acc[s] += elements[k];
                                touched[s] = 1;
                            }
                        }
                for (int s = 0; s < 27; s++) {
                    if (!touched[s]) continue;
                    if (elementsC) {
                        int ox = s % 3 - 1, oy = (s / 3) % 3 - 1, oz = s / 9 - 1;
                        int64_t J = ((int64_t)(Z + oz) * NY + (Y + oy)) * NX + (X + ox);
                        elementsC[kout] = sigma * acc[s];
                        columnIndecesC[kout] = (int)J;
                    }
                    kout++;
                }
                rowOffsetsC[I + 1] = (int)kout;
            }
    return 0;
}

/* b_c[I] = sum of r over the children of I, children in (z,y,x) order, left to right. */
void oracle_mg_restrict(int nx, int ny, int nz, const double *r, double *bc)
{
    const int cx = nx > 1 ? 2 : 1, cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const int NX = nx / cx, NY = ny / cy, NZ = nz / cz;
    int64_t I = 0;
    for (int Z = 0; Z < NZ; Z++)
        for (int Y = 0; Y < NY; Y++)
            for (int X = 0; X < NX; X++, I++) {
                double s = 0;
                for (int dz = 0; dz < cz; dz++)
                    for (int dy = 0; dy < cy; dy++)
                        for (int dx = 0; dx < cx; dx++)
                            s += r[((int64_t)(Z * cz + dz) * ny + (Y * cy + dy)) * nx + (X * cx + dx)];
                bc[I] = s;
            }
}

/* x[i] += e[parent(i)] */
void oracle_mg_prolong_add(int nx, int ny, int nz, double *x, const double *e)
{
    const int cx = nx > 1 ? 2 : 1, cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const int NX = nx / cx, NY = ny / cy;
    int64_t i = 0;
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x_ = 0; x_ < nx; x_++, i++)
                x[i] += e[((int64_t)(z / cz) * NY + (y / cy)) * NX + (x_ / cx)];
}

/*
 * Cell-centred linear transfer, one dimension: fine index i of a coarsened
 * dimension (fine extent n, coarse extent n/2) touches coarse cells
 *   i/2 with weight 3/4   and   i/2 - 1 (i even) or i/2 + 1 (i odd) with 1/4
 * (dropped outside [0, n/2)); a dimension of extent 1 is carried over with
 * weight 1.  All weights and their products (27/64 ... 1/64) are exact in
 * binary, so w = (wz * wy) * wx carries no rounding.
 *
 * Prolongation x[i] += sum_k w_k * e[c_k]: terms in the order z (parent,
 * then neighbour), y (parent, neighbour), x (parent, neighbour); each term is
 * the rounded product w * e, added to a running sum that starts at 0; the
 * sum is then added to x[i].
 */
static int lin_terms(int i, int n, int idx[2], double w[2])
{
    if (n == 1) { idx[0] = 0; w[0] = 1.0; return 1; }
    const int I = i / 2, J = (i % 2 == 0) ? I - 1 : I + 1;
    idx[0] = I; w[0] = 0.75;
    if (J < 0 || J >= n / 2) return 1;
    idx[1] = J; w[1] = 0.25;
    return 2;
}

void oracle_mg_prolong_add_linear(int nx, int ny, int nz, double *x, const double *e)
{
    const int NX = nx > 1 ? nx / 2 : 1, NY = ny > 1 ? ny / 2 : 1;
    int64_t i = 0;
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            for (int x_ = 0; x_ < nx; x_++, i++) {
                int iz[2], iy[2], ix[2]; double wz[2], wy[2], wx[2];
                const int kz = lin_terms(z, nz, iz, wz), ky = lin_terms(y, ny, iy, wy), kx = lin_terms(x_, nx, ix, wx);
                double s = 0;
                for (int a = 0; a < kz; a++)
                    for (int b = 0; b < ky; b++)
                        for (int c = 0; c < kx; c++) {
                            const double w = (wz[a] * wy[b]) * wx[c];
                            const double t = w * e[((int64_t)iz[a] * NY + iy[b]) * NX + ix[c]];
                            s += t;
                        }
                x[i] += s;
            }
}

/*
 * Restriction b_c = P^T r with the P above: coarse I of a coarsened dimension
 * gathers the fine cells 2I-1, 2I, 2I+1, 2I+2 with weights 1/4, 3/4, 3/4, 1/4
 * (those inside the grid), a dimension of extent 1 its single cell with
 * weight 1.  Terms in ascending (z, y, x) order of the fine cell, each the
 * rounded product w * r with w = (wz * wy) * wx, summed left to right from 0.
 */
static int lin_gather(int I, int n, int idx[4], double w[4])
{
    if (n == 1) { idx[0] = 0; w[0] = 1.0; return 1; }
    int k = 0;
    for (int d = -1; d <= 2; d++) {
        const int i = 2 * I + d;
        if (i < 0 || i >= n) continue;
        idx[k] = i; w[k] = (d == -1 || d == 2) ? 0.25 : 0.75; k++;
    }
    return k;
}

void oracle_mg_restrict_linear(int nx, int ny, int nz, const double *r, double *bc)
{
    const int NX = nx > 1 ? nx / 2 : 1, NY = ny > 1 ? ny / 2 : 1, NZ = nz > 1 ? nz / 2 : 1;
    int64_t I = 0;
    for (int Z = 0; Z < NZ; Z++)
        for (int Y = 0; Y < NY; Y++)
            for (int X = 0; X < NX; X++, I++) {
                int iz[4], iy[4], ix[4]; double wz[4], wy[4], wx[4];
                const int kz = lin_gather(Z, nz, iz, wz), ky = lin_gather(Y, ny, iy, wy), kx = lin_gather(X, nx, ix, wx);
                double s = 0;
                for (int a = 0; a < kz; a++)
                    for (int b = 0; b < ky; b++)
                        for (int c = 0; c < kx; c++) {
                            const double w = (wz[a] * wy[b]) * wx[c];
                            const double t = w * r[((int64_t)iz[a] * ny + iy[b]) * nx + ix[c]];
                            s += t;
                        }
                bc[I] = s;
            }
}

/* x = omega * (dinv * b)   (Jacobi sweep from a zero guess) */
void oracle_mg_jacobi_first(int64_t n, double omega, const double *dinv, const double *b, double *x)
{
    for (int64_t i = 0; i < n; i++) { double t = dinv[i] * b[i]; x[i] = omega * t; }
}

/* xnew = x + omega * (dinv * (b - A x)); A x summed in stored order (as oracle_spmv). */
void oracle_mg_jacobi(const double *elements, const int *columnIndeces, const int *rowOffsets, int64_t n,
                      double omega, const double *dinv, const double *b, const double *x, double *xnew)
{
    for (int64_t i = 0; i < n; i++) {
        double ax = 0;
        for (int64_t k = rowOffsets[i]; k < rowOffsets[i + 1]; k++) { double prod = elements[k] * x[columnIndeces[k]]; ax += prod; }
        double res = b[i] - ax;
        double t = dinv[i] * res;
        double s = omega * t;
        xnew[i] = x[i] + s;
    }
}

/* r = b - A x */
void oracle_mg_residual(const double *elements, const int *columnIndeces, const int *rowOffsets, int64_t n,
                        const double *b, const double *x, double *r)
{
    for (int64_t i = 0; i < n; i++) {
        double ax = 0;
        for (int64_t k = rowOffsets[i]; k < rowOffsets[i + 1]; k++) { double prod = elements[k] * x[columnIndeces[k]]; ax += prod; }
        r[i] = b[i] - ax;
    }
}

mg_hierarchy *oracle_mg_setup(int nx, int ny, int nz, int levels,
                              const double *elements, const int *columnIndeces, const int *rowOffsets,
                              double omega, int nu, int nuCoarse, double sigma)
{
    mg_hierarchy *H = (mg_hierarchy *)calloc(1, sizeof(mg_hierarchy));
    H->levels = levels; H->omega = omega; H->nu = nu; H->nuCoarse = nuCoarse;
    H->lv = (mg_level *)calloc((size_t)levels, sizeof(mg_level));
    for (int l = 0; l < levels; l++) {
        mg_level *L = &H->lv[l];
        if (l == 0) {
            L->nx = nx; L->ny = ny; L->nz = nz;
            L->n = (int64_t)nx * ny * nz;
            L->elements = (double *)elements; L->columnIndeces = (int *)columnIndeces; L->rowOffsets = (int *)rowOffsets;
            L->nnz = rowOffsets[L->n];
            L->owns_matrix = 0;
        } else {
            const mg_level *F = &H->lv[l - 1];
            if ((F->nx > 1 && F->nx % 2) || (F->ny > 1 && F->ny % 2) || (F->nz > 1 && F->nz % 2) ||     /* odd extent */
                (F->nx == 1 && F->ny == 1 && F->nz == 1)) {                                            /* a single cell: nothing left to coarsen */
                H->levels = l; break;
            }
            L->nx = F->nx > 1 ? F->nx / 2 : 1; L->ny = F->ny > 1 ? F->ny / 2 : 1; L->nz = F->nz > 1 ? F->nz / 2 : 1;
            L->n = (int64_t)L->nx * L->ny * L->nz;
            L->rowOffsets = (int *)malloc(sizeof(int) * (size_t)(L->n + 1));
            if (oracle_mg_galerkin(F->nx, F->ny, F->nz, F->elements, F->columnIndeces, F->rowOffsets, sigma, NULL, NULL, L->rowOffsets)) {
                free(L->rowOffsets); H->levels = l; break;
            }
            L->nnz = L->rowOffsets[L->n];
            L->elements = (double *)malloc(sizeof(double) * (size_t)L->nnz);
            L->columnIndeces = (int *)malloc(sizeof(int) * (size_t)L->nnz);
            oracle_mg_galerkin(F->nx, F->ny, F->nz, F->elements, F->columnIndeces, F->rowOffsets, sigma, L->elements, L->columnIndeces, L->rowOffsets);
            L->owns_matrix = 1;
        }
        L->dinv = (double *)malloc(sizeof(double) * (size_t)L->n);
        extract_dinv(L, L->dinv);
        L->x = (double *)calloc((size_t)L->n, sizeof(double));
        L->b = (double *)calloc((size_t)L->n, sizeof(double));
        L->r = (double *)calloc((size_t)L->n, sizeof(double));
        L->t = (double *)calloc((size_t)L->n, sizeof(double));
    }
    return H;
}

void oracle_mg_free(mg_hierarchy *H)
{
    if (!H) return;
    for (int l = 0; l < H->levels; l++) {
        mg_level *L = &H->lv[l];
        if (L->owns_matrix) { free(L->elements); free(L->columnIndeces); free(L->rowOffsets); }
        free(L->dinv); free(L->x); free(L->b); free(L->r); free(L->t);
    }
    free(H->lv); free(H);
}

void oracle_mg_set_interpolation(mg_hierarchy *H, int mode) { H->interp = (mode == 1) ? 1 : 0; }
int oracle_mg_levels(const mg_hierarchy *H) { return H->levels; }
int64_t oracle_mg_level_rows(const mg_hierarchy *H, int l) { return H->lv[l].n; }
int64_t oracle_mg_level_nnz(const mg_hierarchy *H, int l) { return H->lv[l].nnz; }
void oracle_mg_level_dims(const mg_hierarchy *H, int l, int *dims) { dims[0] = H->lv[l].nx; dims[1] = H->lv[l].ny; dims[2] = H->lv[l].nz; }
void oracle_mg_level_csr(const mg_hierarchy *H, int l, double *elements, int *columnIndeces, int *rowOffsets)
{
    const mg_level *L = &H->lv[l];
    memcpy(elements, L->elements, sizeof(double) * (size_t)L->nnz);
    memcpy(columnIndeces, L->columnIndeces, sizeof(int) * (size_t)L->nnz);
    memcpy(rowOffsets, L->rowOffsets, sizeof(int) * (size_t)(L->n + 1));
}
void oracle_mg_level_dinv(const mg_hierarchy *H, int l, double *dinv) { memcpy(dinv, H->lv[l].dinv, sizeof(double) * (size_t)H->lv[l].n); }

/* jacobi sweeps on level L: `first` => the first sweep starts from a zero guess. */
static void smooth(mg_hierarchy *H, mg_level *L, int sweeps, int first)
{
    for (int s = 0; s < sweeps; s++) {
        if (first && s == 0) {
            oracle_mg_jacobi_first(L->n, H->omega, L->dinv, L->b, L->x);
        } else {
            oracle_mg_jacobi(L->elements, L->columnIndeces, L->rowOffsets, L->n, H->omega, L->dinv, L->b, L->x, L->t);
            double *tmp = L->x; L->x = L->t; L->t = tmp;
        }
    }
}

static void vcycle(mg_hierarchy *H, int l)
{
    mg_level *L = &H->lv[l];
    if (l == H->levels - 1) { smooth(H, L, H->nuCoarse, 1); return; }
    mg_level *C = &H->lv[l + 1];
    smooth(H, L, H->nu, 1);
    oracle_mg_residual(L->elements, L->columnIndeces, L->rowOffsets, L->n, L->b, L->x, L->r);
    if (H->interp == 1) oracle_mg_restrict_linear(L->nx, L->ny, L->nz, L->r, C->b);
    else oracle_mg_restrict(L->nx, L->ny, L->nz, L->r, C->b);
    vcycle(H, l + 1);
    if (H->interp == 1) oracle_mg_prolong_add_linear(L->nx, L->ny, L->nz, L->x, C->x);
    else oracle_mg_prolong_add(L->nx, L->ny, L->nz, L->x, C->x);
    smooth(H, L, H->nu, 0);
}

/* z = M^-1 r : one V(nu,nu) cycle from a zero guess. */
void oracle_mg_apply(mg_hierarchy *H, const double *r, double *z)
{
    mg_level *L = &H->lv[0];
    memcpy(L->b, r, sizeof(double) * (size_t)L->n);
    vcycle(H, 0);
    memcpy(z, L->x, sizeof(double) * (size_t)L->n);
}

/*
 * Preconditioned CG with the reference's shell (ConjugateGradientCpu.cs:45-98):
 *   Ap = A x; r = b - Ap; z = M^-1 r; p = z; rz = r.z
 *   loop: Ap = A p; alpha = rz / p.Ap; x += alpha p; r -= alpha Ap;
 *         Residual = sqrt(r.r); IsConverged? ; z = M^-1 r; rzNew = r.z;
 *         beta = rzNew / rz; p = z + beta p; rz = rzNew
 * rule: 0 = ConjugateGradient.cs:56-79, 1 = Mgcg.cu:252 (min <= it && res < tol, plus maxIteration cap).
 */
/* The dot products of the row-partitioned loop: every device sums its own rows left to right, the host adds the per-device sums in
 * device order starting from 0 (resultsDot.Sum(), Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:463,499,525).  One device: oracle_dot. */
static double dot_parts(const double *l, const double *r, int deviceCount, const int64_t *off)
{
    double sum = 0;
    for (int d = 0; d < deviceCount; d++) sum += oracle_dot(l + off[d], r + off[d], off[d + 1] - off[d]);
    return sum;
}

int oracle_pcg_parts(mg_hierarchy *H, double *x, const double *b,
                     int rule, double allowableResidual, int minIteration, int maxIteration,
                     int deviceCount, const int64_t *off,
                     int *iteration, double *residual, double *trace, int64_t traceCap);

int oracle_pcg(mg_hierarchy *H, double *x, const double *b,
               int rule, double allowableResidual, int minIteration, int maxIteration,
               int *iteration, double *residual, double *trace, int64_t traceCap)
{
    const int64_t off[2] = { 0, H->lv[0].n };
    return oracle_pcg_parts(H, x, b, rule, allowableResidual, minIteration, maxIteration, 1, off, iteration, residual, trace, traceCap);
}

/* The same loop with its dot products cut at the row ranges off[0..deviceCount] of a row partition (the preconditioner itself does not
 * depend on the partition: the V-cycle has no sums across rows other than the SpMV rows themselves). */
int oracle_pcg_parts(mg_hierarchy *H, double *x, const double *b,
                     int rule, double allowableResidual, int minIteration, int maxIteration,
                     int deviceCount, const int64_t *off,
                     int *iteration, double *residual, double *trace, int64_t traceCap)
{
    mg_level *L = &H->lv[0];
    const int64_t n = L->n;
    double *r = (double *)malloc(sizeof(double) * 4 * (size_t)n);
    double *p = r + n, *Ap = r + 2 * n, *z = r + 3 * n;
    int status = 0;
    oracle_spmv(L->elements, L->columnIndeces, L->rowOffsets, n, Ap, x);
    oracle_set_added(r, b, Ap, -1, n);
    oracle_mg_apply(H, r, z);
    memcpy(p, z, sizeof(double) * (size_t)n);
    double rz = dot_parts(r, z, deviceCount, off);
    double res = 0;
    int it;
    for (it = 0;; it++) {
        oracle_spmv(L->elements, L->columnIndeces, L->rowOffsets, n, Ap, p);
        double alpha = rz / dot_parts(p, Ap, deviceCount, off);
        oracle_set_added(x, x, p, alpha, n);
        oracle_set_added(r, r, Ap, -alpha, n);
        double rrNew = dot_parts(r, r, deviceCount, off);
        res = sqrt(rrNew);
        if (trace && it < traceCap) trace[it] = res;
        int converged;
        if (rule == 1) {
            converged = (minIteration <= it) && (res < allowableResidual);
            if (!converged && it >= maxIteration) { status = 1; converged = 1; }
        } else {
            if (it < minIteration) converged = 0;
            else if (it > maxIteration) { status = 1; converged = 1; }
            else converged = (res < allowableResidual);
        }
        if (converged) break;
        if (!(res == res) || isinf(res)) { status = 3; break; }
        oracle_mg_apply(H, r, z);
        double rzNew = dot_parts(r, z, deviceCount, off);
        double beta = rzNew / rz;
        oracle_set_added(p, z, p, beta, n);
        rz = rzNew;
    }
    *iteration = it; *residual = res;
    free(r);
    return status;
}
