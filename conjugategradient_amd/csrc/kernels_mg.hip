// Multigrid transfer / set-up kernels and the device problem generator (gfx950).
// The reference's "Mgcg" never implemented its multigrid; the algorithm is
// defined in DESIGN.md section 6 and stated on the CPU in oracle/mg_oracle.c, whose arithmetic
// order every kernel here reproduces (restriction adds the children in (z,y,x) order, etc.).
// The smoother and residual passes are SpMV epilogues (kernels_spmv.hip: EPI_JACOBI, EPI_RESIDUAL).
#include "common.hpp"

namespace mgcg {

static inline int grid1(long long n)
{
    long long b = (n + kBlock - 1) / kBlock;
    if (b > kMaxGrid) b = kMaxGrid;
    if (b < 1) b = 1;
    return (int)b;
}

// x = omega * (dinv * b): a Jacobi sweep from a zero guess needs no SpMV.
typedef double d2mg __attribute__((ext_vector_type(2)));

// UNIFORM: every dinv[i] equals dscalar, the array is not read (16 instead of 24 bytes per row).  Pairs of rows per lane,
// one contiguous chunk per workgroup (the shape that streams fastest, profiles/r1/vec_probe_update_kernels.log).
template <bool UNIFORM>
__global__ __launch_bounds__(kBlock) void jacobi_first_kernel(long long n, double omega, const double* __restrict__ dinv, double dscalar,
                                                              const double* __restrict__ b, double* __restrict__ x, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const bool wide = ((((uintptr_t)b) | ((uintptr_t)x) | ((uintptr_t)dinv)) & 15) == 0;
    if (wide) {
        const long long n2 = n >> 1;
        const long long per = ((n2 + gridDim.x - 1) / gridDim.x + (kBlock - 1)) & ~(long long)(kBlock - 1);
        long long end = per * (blockIdx.x + 1);
        end = end < n2 ? end : n2;
        for (long long i = per * blockIdx.x + threadIdx.x; i < end; i += kBlock) {
            const d2mg bv = ((const d2mg*)b)[i];
            d2mg dv; if (UNIFORM) { dv.x = dscalar; dv.y = dscalar; } else dv = ((const d2mg*)dinv)[i];
            d2mg xv; double t0 = dv.x * bv.x; double t1 = dv.y * bv.y; xv.x = omega * t0; xv.y = omega * t1;
            ((d2mg*)x)[i] = xv;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { const double t = (UNIFORM ? dscalar : dinv[n - 1]) * b[n - 1]; x[n - 1] = omega * t; }
        return;
    }
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double t = (UNIFORM ? dscalar : dinv[i]) * b[i];
        x[i] = omega * t;
    }
}
void launch_jacobi_first(hipStream_t s, long long n, double omega, const double* dinv, int dinvUniform, double dinvScalar, const double* b, double* x, const int* done)
{
    if (n <= 0) return;
    long long blocks = (n / 2 + kBlock - 1) / kBlock;
    const int grid = (int)(blocks < 1 ? 1 : (blocks > kMaxGrid ? kMaxGrid : blocks));
    if (dinvUniform) hipLaunchKernelGGL(jacobi_first_kernel<true>, dim3(grid), dim3(kBlock), 0, s, n, omega, (const double*)nullptr, dinvScalar, b, x, done);
    else hipLaunchKernelGGL(jacobi_first_kernel<false>, dim3(grid), dim3(kBlock), 0, s, n, omega, dinv, 0.0, b, x, done);
}

__global__ __launch_bounds__(kBlock) void uniform_check_kernel(const double* __restrict__ v, long long n, int* flag)
{
    const long long first = __double_as_longlong(v[0]);
    bool differs = false;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) differs = differs || __double_as_longlong(v[i]) != first;
    if (differs) atomicExch(flag, 1);
}
void launch_uniform_check(hipStream_t s, const double* v, long long n, int* flag)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(uniform_check_kernel, dim3(grid1(n)), dim3(kBlock), 0, s, v, n, flag);
}

// bc[I] = sum of r over the children of I in (z,y,x) order.  One lane per coarse cell; the two
// x-children are adjacent, so a wavefront reads whole 1 KiB spans of each fine line.
__global__ __launch_bounds__(kBlock) void restrict_kernel(int nx, int ny, int nz, const double* __restrict__ r, double* __restrict__ bc, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const int cx = nx > 1 ? 2 : 1, cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const int NX = nx / cx, NY = ny / cy, NZ = nz / cz;
    const long long NC = (long long)NX * NY * NZ;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long I = (long long)blockIdx.x * kBlock + threadIdx.x; I < NC; I += stride) {
        const int X = (int)(I % NX), Y = (int)((I / NX) % NY), Z = (int)(I / ((long long)NX * NY));
        double sum = 0.0;
        for (int dz = 0; dz < cz; ++dz)
            for (int dy = 0; dy < cy; ++dy) {
                const long long base = ((long long)(Z * cz + dz) * ny + (Y * cy + dy)) * nx + (long long)X * cx;
                for (int dx = 0; dx < cx; ++dx) sum += r[base + dx];
            }
        bc[I] = sum;
    }
}
void launch_restrict(hipStream_t s, int nx, int ny, int nz, const double* r, double* bc, const int* done)
{
    const long long NC = (long long)(nx > 1 ? nx / 2 : 1) * (ny > 1 ? ny / 2 : 1) * (nz > 1 ? nz / 2 : 1);
    hipLaunchKernelGGL(restrict_kernel, dim3(grid1(NC)), dim3(kBlock), 0, s, nx, ny, nz, r, bc, done);
}

// x[i] += e[parent(i)]
__global__ __launch_bounds__(kBlock) void prolong_add_kernel(int nx, int ny, int nz, double* __restrict__ x, const double* __restrict__ e, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const int cx = nx > 1 ? 2 : 1, cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const int NX = nx / cx, NY = ny / cy;
    const long long N = (long long)nx * ny * nz;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
        const int xx = (int)(i % nx), yy = (int)((i / nx) % ny), zz = (int)(i / ((long long)nx * ny));
        x[i] += e[((long long)(zz / cz) * NY + (yy / cy)) * NX + (xx / cx)];
    }
}
// The same two updates with a lane owning the two x-children of one coarse cell: one 16-byte access of x (or b) and one read of e
// per pair, one contiguous chunk of pairs per workgroup, 32-bit index arithmetic (shifts when the extents are powers of two).
// Round 2's one-lane-per-cell forms above ran at 4.2 TB/s on the 512^3 level (0.55 ms for 17 N bytes), 64-bit divisions per cell
// included; they remain for odd nx, unaligned vectors and grids beyond 2^31 pairs.
template <bool SCALED, bool POW2>
__global__ __launch_bounds__(kBlock) void prolong_pairs_kernel(unsigned half, unsigned ny, int lgHalf, int lgNy, int cy, int cz, unsigned NX, unsigned NY, unsigned pairs,
                                                               double* __restrict__ x, const double* __restrict__ b, double inner, double outer,
                                                               const double* __restrict__ e, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const unsigned per = ((pairs + gridDim.x - 1) / gridDim.x + (kBlock - 1)) & ~(unsigned)(kBlock - 1);
    const unsigned long long endL = (unsigned long long)per * (blockIdx.x + 1);
    const unsigned end = endL < pairs ? (unsigned)endL : pairs;
    for (unsigned long long qL = (unsigned long long)per * blockIdx.x + threadIdx.x; qL < end; qL += kBlock) {
        const unsigned q = (unsigned)qL;
        unsigned row, X, yy, zz;
        if (POW2) { row = q >> lgHalf; X = q & (half - 1); yy = row & (ny - 1); zz = row >> lgNy; }
        else { row = q / half; X = q - row * half; zz = row / ny; yy = row - zz * ny; }
        const unsigned Y = cy == 2 ? yy >> 1 : yy, Z = cz == 2 ? zz >> 1 : zz;
        const double ev = e[((unsigned long long)Z * NY + Y) * NX + X];
        d2mg v = SCALED ? ((const d2mg*)b)[q] : ((const d2mg*)x)[q];
        if (SCALED) { const double t0 = inner * v.x, t1 = inner * v.y; v.x = outer * t0; v.y = outer * t1; }
        v.x = v.x + ev; v.y = v.y + ev;
        ((d2mg*)x)[q] = v;
    }
}
static bool launch_prolong_pairs(hipStream_t s, bool scaled, int nx, int ny, int nz, double* x, const double* b, double inner, double outer, const double* e, const int* done)
{
    if (nx < 2 || (nx & 1) || ((((uintptr_t)x) | ((uintptr_t)b)) & 15) != 0) return false;
    const long long pairsL = (long long)(nx / 2) * ny * nz;
    if (pairsL <= 0 || pairsL >= 0x7fffffffLL) return false;
    const unsigned half = (unsigned)nx / 2, pairs = (unsigned)pairsL;
    auto lg = [](unsigned v) { int l = 0; while ((1u << l) < v) ++l; return ((1u << l) == v) ? l : -1; };
    const int lgHalf = lg(half), lgNy = lg((unsigned)ny);
    const bool pow2 = lgHalf >= 0 && lgNy >= 0;
    const int cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const unsigned NX = half, NY = (unsigned)(ny / cy);
    const int grid = grid1((long long)pairs);
#define GO(S, P) hipLaunchKernelGGL((prolong_pairs_kernel<S, P>), dim3(grid), dim3(kBlock), 0, s, half, (unsigned)ny, lgHalf, lgNy, cy, cz, NX, NY, pairs, x, b, inner, outer, e, done)
    if (scaled) { if (pow2) GO(true, true); else GO(true, false); }
    else { if (pow2) GO(false, true); else GO(false, false); }
#undef GO
    return true;
}

void launch_prolong_add(hipStream_t s, int nx, int ny, int nz, double* x, const double* e, const int* done)
{
    if (launch_prolong_pairs(s, false, nx, ny, nz, x, x, 0.0, 0.0, e, done)) return;
    hipLaunchKernelGGL(prolong_add_kernel, dim3(grid1((long long)nx * ny * nz)), dim3(kBlock), 0, s, nx, ny, nz, x, e, done);
}

// x[i] = outer * (inner * b[i]) + e[parent(i)]: the first sweep's iterate is formed here instead of being read back
__global__ __launch_bounds__(kBlock) void prolong_scaled_kernel(int nx, int ny, int nz, double* __restrict__ x, const double* __restrict__ b,
                                                                double inner, double outer, const double* __restrict__ e, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const int cx = nx > 1 ? 2 : 1, cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const int NX = nx / cx, NY = ny / cy;
    const long long N = (long long)nx * ny * nz;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
        const int xx = (int)(i % nx), yy = (int)((i / nx) % ny), zz = (int)(i / ((long long)nx * ny));
        const double t = inner * b[i];
        const double x1 = outer * t;
        x[i] = x1 + e[((long long)(zz / cz) * NY + (yy / cy)) * NX + (xx / cx)];
    }
}
void launch_prolong_scaled(hipStream_t s, int nx, int ny, int nz, double* x, const double* b, double inner, double outer, const double* e, const int* done)
{
    if (launch_prolong_pairs(s, true, nx, ny, nz, x, b, inner, outer, e, done)) return;
    hipLaunchKernelGGL(prolong_scaled_kernel, dim3(grid1((long long)nx * ny * nz)), dim3(kBlock), 0, s, nx, ny, nz, x, b, inner, outer, e, done);
}

// ---- cell-centred linear transfer (MgSetInterpolation(mg, 1)); arithmetic order of oracle_mg_prolong_add_linear /
// oracle_mg_restrict_linear (oracle/mg_oracle.c): per coarsened dimension child i takes 3/4 of parent i/2 and 1/4 of the
// parent's neighbour on the child's side; weights and their products are exact, each term is one rounded product, the terms
// are summed from zero in (z, y, x) order.  Both kernels address the OTHER grid globally (the caller has exchanged the
// neighbours' plane of it) and their own grid by the rank's planes [z0, z1).
__device__ __forceinline__ int lin_terms(int i, int n, int idx[2], double w[2])
{
    if (n == 1) { idx[0] = 0; w[0] = 1.0; return 1; }
    const int I = i >> 1, J = (i & 1) ? I + 1 : I - 1;
    idx[0] = I; w[0] = 0.75;
    if (J < 0 || J >= (n >> 1)) return 1;
    idx[1] = J; w[1] = 0.25;
    return 2;
}

// x[i] += sum_k w_k e[c_k] for the fine cells of planes [z0, z1); x is slab-local, eFull the whole coarse vector
__global__ __launch_bounds__(kBlock) void prolong_linear_add_kernel(int nx, int ny, int nz, int z0, int z1, double* __restrict__ x,
                                                                    const double* __restrict__ eFull, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const int NX = nx > 1 ? nx / 2 : 1, NY = ny > 1 ? ny / 2 : 1;
    const long long N = (long long)nx * ny * (z1 - z0);
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < N; i += stride) {
        const int xx = (int)(i % nx), yy = (int)((i / nx) % ny), zz = z0 + (int)(i / ((long long)nx * ny));
        int iz[2], iy[2], ix[2]; double wz[2], wy[2], wx[2];
        const int kz = lin_terms(zz, nz, iz, wz), ky = lin_terms(yy, ny, iy, wy), kx = lin_terms(xx, nx, ix, wx);
        double sum = 0.0;
        for (int a = 0; a < kz; ++a)
            for (int b = 0; b < ky; ++b) {
                const double wzy = wz[a] * wy[b];
                const long long base = ((long long)iz[a] * NY + iy[b]) * NX;
                for (int c = 0; c < kx; ++c) { const double w = wzy * wx[c]; const double t = w * eFull[base + ix[c]]; sum += t; }
            }
        x[i] = x[i] + sum;
    }
}
// The same for an even nx with 16-byte aligned x: a workgroup is TX lanes along x by kBlock/TX fine rows; a lane owns the two
// children of coarse cell X (one 16-byte load and store of x, three coarse values per (z, y) term instead of four) and no
// index is divided.  Per-cell arithmetic is that of the kernel above.
__global__ __launch_bounds__(kBlock) void prolong_linear_add_rows_kernel(int nx, int ny, int nz, int z0, int z1, int tx, double* __restrict__ x,
                                                                         const double* __restrict__ eFull, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const int NX = nx >> 1, NY = ny > 1 ? ny / 2 : 1;
    const int ty = kBlock / tx;
    const int lx = threadIdx.x % tx, ly = threadIdx.x / tx;
    const long long rows = (long long)ny * (z1 - z0);
    // workgroup b runs on XCD b % 8 (each with its own L2): XCD k sweeps the k-th eighth of the rows, so that the coarse rows a fine
    // row shares with its neighbours are fetched into one L2 instead of eight
    const long long band = (rows + 7) / 8;
    const long long bandBegin = (blockIdx.x & 7) * band, bandEnd = bandBegin + band < rows ? bandBegin + band : rows;
    for (long long row = bandBegin + (long long)(blockIdx.x >> 3) * ty + ly; row < bandEnd; row += (long long)(gridDim.x >> 3) * ty) {
        const unsigned r32 = (unsigned)row;                       // rows < 2^31 (the vector has fewer entries than that): 32-bit division
        const int yy = (int)(r32 % (unsigned)ny), zz = z0 + (int)(r32 / (unsigned)ny);
        int iz[2], iy[2]; double wz[2], wy[2];
        const int kz = lin_terms(zz, nz, iz, wz), ky = lin_terms(yy, ny, iy, wy);
        d2mg* xr = (d2mg*)(x + row * nx);
        if (kz == 1) { iz[1] = iz[0]; wz[1] = 0.0; }                 // absent terms: a valid address, and (below) +0.0 instead of their product
        if (ky == 1) { iy[1] = iy[0]; wy[1] = 0.0; }
        for (int X = lx; X < NX; X += tx) {
            const bool hasL = X > 0, hasR = X + 1 < NX;
            const int XL = hasL ? X - 1 : X, XR = hasR ? X + 1 : X;
            // all twelve coarse operands are loaded before the first is used (four dependent round trips otherwise); a term that does
            // not exist contributes +0.0, which leaves a sum that started at +0.0 bit for bit as it was
            double eC[4], eL[4], eR[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double* er = eFull + ((long long)iz[t >> 1] * NY + iy[t & 1]) * NX;
                eC[t] = er[X]; eL[t] = er[XL]; eR[t] = er[XR];
            }
            double sumL = 0.0, sumR = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool term = (t >> 1) < kz && (t & 1) < ky;
                const double wzy = wz[t >> 1] * wy[t & 1];
                const double w3 = wzy * 0.75, w1 = wzy * 0.25;
                const double pC = w3 * eC[t], pL = w1 * eL[t], pR = w1 * eR[t];
                const double tC = term ? pC : 0.0;
                sumL += tC; sumR += tC;
                sumL += (term && hasL) ? pL : 0.0;
                sumR += (term && hasR) ? pR : 0.0;
            }
            d2mg v = xr[X];
            v.x = v.x + sumL; v.y = v.y + sumR;
            xr[X] = v;
        }
    }
}
static int rows_tx(int pairs) { int t = 1; while (t < pairs && t < kBlock) t <<= 1; return t; }
void launch_prolong_linear_add(hipStream_t s, int nx, int ny, int nz, int z0, int z1, double* x, const double* eFull, const int* done)
{
    if (nx >= 2 && (nx & 1) == 0 && (((uintptr_t)x) & 15) == 0) {
        const int tx = rows_tx(nx / 2), ty = kBlock / tx;
        const long long rows = (long long)ny * (z1 - z0);
        long long g = ((rows + 7) / 8 + ty - 1) / ty * 8; if (g > kMaxGrid) g = kMaxGrid; if (g < 8) g = 8;      // a multiple of 8: whole XCD rounds
        hipLaunchKernelGGL(prolong_linear_add_rows_kernel, dim3((int)g), dim3(kBlock), 0, s, nx, ny, nz, z0, z1, tx, x, eFull, done);
        return;
    }
    hipLaunchKernelGGL(prolong_linear_add_kernel, dim3(grid1((long long)nx * ny * (z1 - z0))), dim3(kBlock), 0, s, nx, ny, nz, z0, z1, x, eFull, done);
}

// bc[I] = sum over the (up to) 4x4x4 fine cells around coarse cell I of w * r, for the coarse planes of the rank's fine
// planes [z0, z1); rFull is the whole fine vector, bc slab-local
__global__ __launch_bounds__(kBlock) void restrict_linear_kernel(int nx, int ny, int nz, int z0, int z1, const double* __restrict__ rFull,
                                                                 double* __restrict__ bc, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const int cz = nz > 1 ? 2 : 1;
    const int NX = nx > 1 ? nx / 2 : 1, NY = ny > 1 ? ny / 2 : 1;
    const int Z0 = z0 / cz, ZL = (z1 - z0) / cz;
    const long long NC = (long long)NX * NY * ZL;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long I = (long long)blockIdx.x * kBlock + threadIdx.x; I < NC; I += stride) {
        const int X = (int)(I % NX), Y = (int)((I / NX) % NY), Z = Z0 + (int)(I / ((long long)NX * NY));
        // per dimension: first fine index, number of terms, and whether the first term is the quarter-weight one
        auto span = [](int C, int n, int& first, int& count, int& quarterFirst) {
            if (n == 1) { first = 0; count = 1; quarterFirst = -1; return; }
            const int lo = 2 * C - 1, hi = 2 * C + 2;
            first = lo < 0 ? 0 : lo; count = (hi >= n ? n - 1 : hi) - first + 1; quarterFirst = lo < 0 ? 0 : 1;
        };
        int fz, nzT, qz, fy, nyT, qy, fx, nxT, qx;
        span(Z, nz, fz, nzT, qz); span(Y, ny, fy, nyT, qy); span(X, nx, fx, nxT, qx);
        // weight of term k of a dimension: 1 for an uncoarsened one; else the fine index relative to 2C-1 is k + (1 - q): 0 and 3 -> 1/4
        auto weight = [](int k, int q) { if (q < 0) return 1.0; const int rel = k + (1 - q); return (rel == 0 || rel == 3) ? 0.25 : 0.75; };
        double sum = 0.0;
        for (int a = 0; a < nzT; ++a) {
            const double wza = weight(a, qz);
            for (int b = 0; b < nyT; ++b) {
                const double wzy = wza * weight(b, qy);
                const long long base = ((long long)(fz + a) * ny + (fy + b)) * nx + fx;
                for (int c = 0; c < nxT; ++c) { const double w = wzy * weight(c, qx); const double t = w * rFull[base + c]; sum += t; }
            }
        }
        bc[I] = sum;
    }
}
// The same for an even nx with 16-byte aligned r: TX lanes along the coarse x by kBlock/TX coarse rows; a lane reads its two
// children as one 16-byte load and the two quarter-weight cells beside them; no index is divided.
__global__ __launch_bounds__(kBlock) void restrict_linear_rows_kernel(int nx, int ny, int nz, int z0, int z1, int tx, const double* __restrict__ rFull,
                                                                      double* __restrict__ bc, const int* done)
{
    if (done != nullptr && *done != 0) return;
    const int cz = nz > 1 ? 2 : 1;
    const int NX = nx >> 1, NY = ny > 1 ? ny / 2 : 1;
    const int Z0 = z0 / cz, ZL = (z1 - z0) / cz;
    const int ty = kBlock / tx;
    const int lx = threadIdx.x % tx, ly = threadIdx.x / tx;
    const long long rows = (long long)NY * ZL;
    auto span = [](int C, int n, int& first, int& count, int& quarterFirst) {
        if (n == 1) { first = 0; count = 1; quarterFirst = -1; return; }
        const int lo = 2 * C - 1, hi = 2 * C + 2;
        first = lo < 0 ? 0 : lo; count = (hi >= n ? n - 1 : hi) - first + 1; quarterFirst = lo < 0 ? 0 : 1;
    };
    auto weight = [](int k, int q) { if (q < 0) return 1.0; const int rel = k + (1 - q); return (rel == 0 || rel == 3) ? 0.25 : 0.75; };
    // XCD k sweeps the k-th eighth of the coarse rows (see the prolongation): the fine rows two coarse rows share stay in one L2
    const long long band = (rows + 7) / 8;
    const long long bandBegin = (blockIdx.x & 7) * band, bandEnd = bandBegin + band < rows ? bandBegin + band : rows;
    for (long long row = bandBegin + (long long)(blockIdx.x >> 3) * ty + ly; row < bandEnd; row += (long long)(gridDim.x >> 3) * ty) {
        const unsigned r32 = (unsigned)row;
        const int Y = (int)(r32 % (unsigned)NY), Z = Z0 + (int)(r32 / (unsigned)NY);
        int fz, nzT, qz, fy, nyT, qy;
        span(Z, nz, fz, nzT, qz); span(Y, ny, fy, nyT, qy);
        for (int X = lx; X < NX; X += tx) {
            const bool hasL = X > 0, hasR = 2 * X + 2 < nx;
            // the eight (z, y) rows of two z terms are loaded before the first is used; a term that does not exist reads a valid
            // address and contributes +0.0 (bitwise neutral for a sum that started at +0.0)
            const int oL = hasL ? -1 : 0, oR = hasR ? 2 : 0;
            double sum = 0.0;
#pragma unroll
            for (int a0 = 0; a0 < 4; a0 += 2) {
                d2mg mid[8]; double left[8], right[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int a = a0 + (t >> 2), b = t & 3;
                    const int za = a < nzT ? a : 0, yb = b < nyT ? b : 0;
                    const double* rr = rFull + ((long long)(fz + za) * ny + (fy + yb)) * nx + 2 * X;
                    mid[t] = *(const d2mg*)rr; left[t] = rr[oL]; right[t] = rr[oR];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    const int a = a0 + (t >> 2), b = t & 3;
                    const bool term = a < nzT && b < nyT;
                    const double wzy = weight(a, qz) * weight(b, qy);
                    const double w3 = wzy * 0.75, w1 = wzy * 0.25;
                    const double pl = w1 * left[t], p0 = w3 * mid[t].x, p1 = w3 * mid[t].y, pr = w1 * right[t];
                    sum += (term && hasL) ? pl : 0.0;
                    sum += term ? p0 : 0.0;
                    sum += term ? p1 : 0.0;
                    sum += (term && hasR) ? pr : 0.0;
                }
            }
            bc[row * NX + X] = sum;
        }
    }
}
void launch_restrict_linear(hipStream_t s, int nx, int ny, int nz, int z0, int z1, const double* rFull, double* bc, const int* done)
{
    if (nx >= 2 && (nx & 1) == 0 && (((uintptr_t)rFull) & 15) == 0) {
        const int tx = rows_tx(nx / 2), ty = kBlock / tx;
        const long long rows = (long long)(ny > 1 ? ny / 2 : 1) * ((z1 - z0) / (nz > 1 ? 2 : 1));
        long long g = ((rows + 7) / 8 + ty - 1) / ty * 8; if (g > kMaxGrid) g = kMaxGrid; if (g < 8) g = 8;
        hipLaunchKernelGGL(restrict_linear_rows_kernel, dim3((int)g), dim3(kBlock), 0, s, nx, ny, nz, z0, z1, tx, rFull, bc, done);
        return;
    }
    const long long NC = (long long)(nx > 1 ? nx / 2 : 1) * (ny > 1 ? ny / 2 : 1) * ((z1 - z0) / (nz > 1 ? 2 : 1));
    hipLaunchKernelGGL(restrict_linear_kernel, dim3(grid1(NC)), dim3(kBlock), 0, s, nx, ny, nz, z0, z1, rFull, bc, done);
}

// dinv[i] = 1 / (first stored entry of local row i whose column is rowBase + i)
__global__ __launch_bounds__(kBlock) void extract_dinv_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets,
                                                              const int* __restrict__ columnIndeces, long long n, long long rowBase, double* __restrict__ dinv)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        double d = 0.0;
        for (int k = rowOffsets[i]; k < rowOffsets[i + 1]; ++k)
            if (columnIndeces[k] == rowBase + i) { d = elements[k]; break; }
        dinv[i] = 1.0 / d;
    }
}
void launch_extract_dinv(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                         long long n, long long rowBase, double* dinv)
{
    hipLaunchKernelGGL(extract_dinv_kernel, dim3(grid1(n)), dim3(kBlock), 0, s, elements, rowOffsets, columnIndeces, n, rowBase, dinv);
}

// sigma * P^T A P on the 27-slot neighbourhood, one lane per LOCAL coarse row (the rank's z-slab [zBegin, zEnd)
// of the fine grid; fine rows are slab-local, column ids global), same accumulation order as oracle_mg_galerkin.
// elementsC == nullptr: count pass (countsC[I] = touched slots).
__global__ __launch_bounds__(kBlock) void galerkin_kernel(int nx, int ny, int nz, int zBegin, int zEnd,
                                                          const double* __restrict__ elements, const int* __restrict__ rowOffsets,
                                                          const int* __restrict__ columnIndeces, double sigma, const int* __restrict__ rowOffsetsC,
                                                          int* __restrict__ countsC, double* __restrict__ elementsC, int* __restrict__ columnIndecesC, int* errFlag)
{
    const int cx = nx > 1 ? 2 : 1, cy = ny > 1 ? 2 : 1, cz = nz > 1 ? 2 : 1;
    const int NX = nx / cx, NY = ny / cy;
    const int ZL = (zEnd - zBegin) / cz;              // local coarse planes
    const int Z0 = zBegin / cz;
    const long long NC = (long long)NX * NY * ZL;
    const long long sxy = (long long)nx * ny;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long I = (long long)blockIdx.x * kBlock + threadIdx.x; I < NC; I += stride) {
        const int X = (int)(I % NX), Y = (int)((I / NX) % NY), Zl = (int)(I / ((long long)NX * NY));
        const int Z = Z0 + Zl;
        double acc[27];
        unsigned touched = 0u;
#pragma unroll
        for (int q = 0; q < 27; ++q) acc[q] = 0.0;
        for (int dz = 0; dz < cz; ++dz)
            for (int dy = 0; dy < cy; ++dy)
                for (int dx = 0; dx < cx; ++dx) {
                    const long long i = ((long long)(Zl * cz + dz) * ny + (Y * cy + dy)) * nx + (X * cx + dx);   // slab-local fine row
                    for (int k = rowOffsets[i]; k < rowOffsets[i + 1]; ++k) {
                        const long long j = columnIndeces[k];
                        const int jx = (int)(j % nx), jy = (int)((j / nx) % ny), jz = (int)(j / sxy);
                        const int ox = jx / cx - X, oy = jy / cy - Y, oz = jz / cz - Z;
                        if (ox < -1 || ox > 1 || oy < -1 || oy > 1 || oz < -1 || oz > 1) { *errFlag = 1; continue; }
                        const int q = (oz + 1) * 9 + (oy + 1) * 3 + (ox + 1);
                        const double v = elements[k];
                        // static indexing keeps acc[] in registers
#pragma unroll
                        for (int t = 0; t < 27; ++t) if (t == q) acc[t] += v;
                        touched |= 1u << q;
                    }
                }
        if (elementsC == nullptr) { countsC[I] = __popc(touched); continue; }
        int kout = rowOffsetsC[I];
#pragma unroll
        for (int q = 0; q < 27; ++q) {
            if (!(touched & (1u << q))) continue;
            const int ox = q % 3 - 1, oy = (q / 3) % 3 - 1, oz = q / 9 - 1;
            elementsC[kout] = sigma * acc[q];
            columnIndecesC[kout] = (int)(((long long)(Z + oz) * NY + (Y + oy)) * NX + (X + ox));   // global coarse column
            ++kout;
        }
    }
}
void launch_galerkin(hipStream_t s, int nx, int ny, int nz, int zBegin, int zEnd, const double* elements, const int* rowOffsets, const int* columnIndeces,
                     double sigma, const int* rowOffsetsC, int* countsC, double* elementsC, int* columnIndecesC, int* errFlag)
{
    const long long NC = (long long)(nx > 1 ? nx / 2 : 1) * (ny > 1 ? ny / 2 : 1) * (nz > 1 ? (zEnd - zBegin) / 2 : 1);
    hipLaunchKernelGGL(galerkin_kernel, dim3(grid1(NC)), dim3(kBlock), 0, s, nx, ny, nz, zBegin, zEnd, elements, rowOffsets, columnIndeces,
                       sigma, rowOffsetsC, countsC, elementsC, columnIndecesC, errFlag);
}

// ---------------------------------------------------------------- device problem generator
// Entries stored before global row i = (x,y,z) of the 5/7-point Dirichlet Poisson matrix
// (closed form: 7i minus the neighbours that fall off the grid in rows < i).
__device__ __host__ inline long long poisson_row_offset(long long i, int nx, int ny, int nz)
{
    const long long sxy = (long long)nx * ny;
    const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / sxy);
    const long long L = i / nx;                                   // complete x-lines before this row
    long long missing = 0;
    missing += L + (x > 0 ? 1 : 0);                               // rows with x == 0      (no -x neighbour)
    missing += L;                                                 // rows with x == nx-1   (no +x neighbour)
    missing += (long long)z * nx + (y > 0 ? nx : x);              // rows with y == 0
    missing += (long long)z * nx + (y == ny - 1 ? x : 0);         // rows with y == ny-1
    if (nz > 1) {
        missing += (z > 0) ? sxy : ((long long)y * nx + x);       // rows with z == 0
        missing += (z == nz - 1) ? ((long long)y * nx + x) : 0;   // rows with z == nz-1
        return 7 * i - missing;
    }
    return 5 * i - missing;
}

__global__ __launch_bounds__(kBlock) void poisson_kernel(int nx, int ny, int nz, int zBegin, int zEnd,
                                                         double* __restrict__ elements, int* __restrict__ rowOffsets, int* __restrict__ columnIndeces)
{
    const long long sxy = (long long)nx * ny;
    const long long rowBegin = (long long)zBegin * sxy, rowEnd = (long long)zEnd * sxy;
    const long long nLocal = rowEnd - rowBegin;
    const long long base = poisson_row_offset(rowBegin, nx, ny, nz);
    const double diag = nz > 1 ? 6.0 : 4.0;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long li = (long long)blockIdx.x * kBlock + threadIdx.x; li <= nLocal; li += stride) {
        const long long i = rowBegin + li;
        if (li == nLocal) {
            const long long total = (long long)nx * ny * nz;
            const long long endOff = (i == total) ? ((nz > 1 ? 7 : 5) * total - 2 * ((long long)ny * nz + (long long)nx * nz + (nz > 1 ? sxy : 0)))
                                                  : poisson_row_offset(i, nx, ny, nz);
            rowOffsets[li] = (int)(endOff - base);
            continue;
        }
        long long k = poisson_row_offset(i, nx, ny, nz) - base;
        rowOffsets[li] = (int)k;
        const int x = (int)(i % nx), y = (int)((i / nx) % ny), z = (int)(i / sxy);
        if (nz > 1 && z > 0)  { elements[k] = -1.0; columnIndeces[k++] = (int)(i - sxy); }
        if (y > 0)            { elements[k] = -1.0; columnIndeces[k++] = (int)(i - nx); }
        if (x > 0)            { elements[k] = -1.0; columnIndeces[k++] = (int)(i - 1); }
        elements[k] = diag;   columnIndeces[k++] = (int)i;
        if (x < nx - 1)       { elements[k] = -1.0; columnIndeces[k++] = (int)(i + 1); }
        if (y < ny - 1)       { elements[k] = -1.0; columnIndeces[k++] = (int)(i + nx); }
        if (nz > 1 && z < nz - 1) { elements[k] = -1.0; columnIndeces[k++] = (int)(i + sxy); }
    }
}
void launch_poisson(hipStream_t s, int nx, int ny, int nz, int zBegin, int zEnd, double* elements, int* rowOffsets, int* columnIndeces)
{
    const long long nLocal = (long long)(zEnd - zBegin) * nx * ny;
    hipLaunchKernelGGL(poisson_kernel, dim3(grid1(nLocal + 1)), dim3(kBlock), 0, s, nx, ny, nz, zBegin, zEnd, elements, rowOffsets, columnIndeces);
}

long long poisson_nnz_host(int nx, int ny, int nz, int zBegin, int zEnd)
{
    const long long sxy = (long long)nx * ny;
    const long long total = sxy * nz;
    auto off = [&](long long i) -> long long {
        if (i == total) return (nz > 1 ? 7 : 5) * total - 2 * ((long long)ny * nz + (long long)nx * nz + (nz > 1 ? sxy : 0));
        return poisson_row_offset(i, nx, ny, nz);
    };
    return off((long long)zEnd * sxy) - off((long long)zBegin * sxy);
}

__global__ __launch_bounds__(kBlock) void rebase_kernel(int* __restrict__ rowOffsets, long long n, int base)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) rowOffsets[i] -= base;
}
void launch_rebase(hipStream_t s, int* rowOffsets, long long n, int base)
{
    if (n <= 0 || base == 0) return;
    hipLaunchKernelGGL(rebase_kernel, dim3(grid1(n)), dim3(kBlock), 0, s, rowOffsets, n, base);
}

// out2[0] = min, out2[1] = max (out2 pre-set to INT_MAX / INT_MIN by the caller)
__global__ __launch_bounds__(kBlock) void minmax_kernel(const int* __restrict__ v, long long n, int* out2)
{
    int lo = 0x7fffffff, hi = (int)0x80000000;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) { int a = v[i]; lo = a < lo ? a : lo; hi = a > hi ? a : hi; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int l2 = __shfl_down(lo, off, 64), h2 = __shfl_down(hi, off, 64);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) { atomicMin(&out2[0], lo); atomicMax(&out2[1], hi); }
}
void launch_minmax_int(hipStream_t s, const int* v, long long n, int* out2)
{
    hipLaunchKernelGGL(minmax_kernel, dim3(grid1(n)), dim3(kBlock), 0, s, v, n, out2);
}

// Which local rows touch columns outside [offset, offset+n)?  out2[0] = 1 + last row that references a column below the
// slice (0 if none), out2[1] = first row that references a column above it (n if none); pre-set by the caller to {0, n}.
__global__ __launch_bounds__(kBlock) void halo_rows_kernel(const int* __restrict__ rowOffsets, const int* __restrict__ columnIndeces,
                                                           long long n, long long offset, int* out2)
{
    int lowNeed = 0, highNeed = 0x7fffffff;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        bool below = false, above = false;
        for (int k = rowOffsets[i]; k < rowOffsets[i + 1]; ++k) {
            const long long c = columnIndeces[k];
            below = below || c < offset;
            above = above || c >= offset + n;
        }
        if (below) lowNeed = (int)i + 1;                 // i grows along the loop
        if (above && (int)i < highNeed) highNeed = (int)i;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int l2 = __shfl_down(lowNeed, off, 64), h2 = __shfl_down(highNeed, off, 64);
        lowNeed = l2 > lowNeed ? l2 : lowNeed; highNeed = h2 < highNeed ? h2 : highNeed;
    }
    if ((threadIdx.x & 63) == 0) { atomicMax(&out2[0], lowNeed); atomicMin(&out2[1], highNeed); }
}
void launch_halo_rows(hipStream_t s, const int* rowOffsets, const int* columnIndeces, long long n, long long offset, int* out2)
{
    hipLaunchKernelGGL(halo_rows_kernel, dim3(grid1(n)), dim3(kBlock), 0, s, rowOffsets, columnIndeces, n, offset, out2);
}

void preload_kernels_mg() { preload_code_object(reinterpret_cast<const void*>(&uniform_check_kernel)); }

} // namespace mgcg
