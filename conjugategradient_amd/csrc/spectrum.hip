// Extreme eigenvalues of A (or of the Jacobi-scaled D^-1/2 A D^-1/2, which has the spectrum of D^-1 A) by a short
// Lanczos recurrence that reuses the SpMV and the dot-product kernels of the CG path.
// The reference's only spectral tool is the dense Jacobi-rotation routine GetEigenValues
// (Mgcg/HandmadeCL/MgcgCL/SparseMatrix.cs:234-372, O(n^2) memory, diagnostic); this is its scalable counterpart
// (SURVEY.md section 8 row f4): a condition-number estimate and the largest eigenvalue of D^-1 A from which the
// damping of the Jacobi smoother is chosen (multigrid.py: jacobi_omega).
#include "common.hpp"
#include <cmath>

namespace mgcg {

// deterministic start vector in (-1, 1): a 64-bit mix of (seed, i)
__global__ __launch_bounds__(kBlock) void lanczos_start_kernel(double* __restrict__ v, long long n, unsigned long long seed)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        unsigned long long z = (unsigned long long)i + seed * 0x9E3779B97F4A7C15ull + 0x632BE59BD9B4E019ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        v[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
    }
}

// y = a .* x   (mode 0)   |   a = sqrt(a) in place (mode 1; a holds 1/d, d > 0)
__global__ __launch_bounds__(kBlock) void lanczos_scale_kernel(double* y, double* a, const double* x, long long n, int mode)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        if (mode == 0) y[i] = a[i] * x[i];
        else a[i] = sqrt(a[i]);
    }
}

static int grid_of(long long n)
{
    long long b = (n + kBlock - 1) / kBlock;
    return (int)(b < 1 ? 1 : (b > kMaxGrid ? kMaxGrid : b));
}

// Number of eigenvalues of the symmetric tridiagonal (a[0..k), b[1..k)) that are < x (Sturm sequence).
static int sturm_count(const std::vector<double>& a, const std::vector<double>& b, int k, double x)
{
    int count = 0;
    double q = 1.0;
    for (int i = 0; i < k; ++i) {
        const double off = (i == 0) ? 0.0 : b[(size_t)i] * b[(size_t)i];
        q = a[(size_t)i] - x - (i == 0 ? 0.0 : off / q);
        if (q == 0.0) q = 1e-300;
        if (q < 0.0) ++count;
    }
    return count;
}

// index-th smallest eigenvalue (0-based) of the k x k tridiagonal, by bisection inside its Gershgorin interval
static double tridiagonal_eigenvalue(const std::vector<double>& a, const std::vector<double>& b, int k, int index)
{
    double lo = a[0], hi = a[0];
    for (int i = 0; i < k; ++i) {
        const double r = (i > 0 ? std::fabs(b[(size_t)i]) : 0.0) + (i + 1 < k ? std::fabs(b[(size_t)i + 1]) : 0.0);
        lo = std::fmin(lo, a[(size_t)i] - r); hi = std::fmax(hi, a[(size_t)i] + r);
    }
    for (int it = 0; it < 200 && hi - lo > 4e-16 * (std::fabs(lo) + std::fabs(hi)); ++it) {
        const double mid = 0.5 * (lo + hi);
        if (sturm_count(a, b, k, mid) > index) hi = mid; else lo = mid;
    }
    return 0.5 * (lo + hi);
}

void preload_spectrum() { preload_code_object(reinterpret_cast<const void*>(&lanczos_start_kernel)); }

} // namespace mgcg

using namespace mgcg;

extern "C" int MgcgEstimateSpectrum(MgcgBlas* cublas, MgcgSparse* cusparse,
                                    Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                                    int elementsCount, int count, int jacobiScaled, int steps, unsigned seed,
                                    double* lambdaMin, double* lambdaMax, double ritz[], int* stepsDone)
{
    if (lambdaMin) *lambdaMin = NAN;
    if (lambdaMax) *lambdaMax = NAN;
    if (stepsDone) *stepsDone = 0;
    DeviceState* d = device_state();
    if (!d) return MGCG_ERROR;
    if (!cublas || !cusparse || !elementsVector || !rowOffsetsVector || !columnIndecesVector) { set_error("MgcgEstimateSpectrum: null argument"); return MGCG_ERROR; }
    if (count < 1 || steps < 1 || elementsCount < 0 || elementsVector->size < elementsCount || columnIndecesVector->size < elementsCount || rowOffsetsVector->size < (long long)count + 1) {
        set_error("MgcgEstimateSpectrum: bad sizes"); return MGCG_ERROR;
    }
    if (steps > count) steps = count;
    Workspace& ws = cublas->ws;
    hipStream_t s = ws.stream;
    const long long n = count;
    double *vPrev = nullptr, *v = nullptr, *w = nullptr, *u = nullptr, *sc = nullptr;
    bool ok = MGCG_HIP(hipMalloc((void**)&vPrev, sizeof(double) * (size_t)n)) && MGCG_HIP(hipMalloc((void**)&v, sizeof(double) * (size_t)n)) &&
              MGCG_HIP(hipMalloc((void**)&w, sizeof(double) * (size_t)n));
    if (ok && jacobiScaled) {
        ok = MGCG_HIP(hipMalloc((void**)&u, sizeof(double) * (size_t)n)) && MGCG_HIP(hipMalloc((void**)&sc, sizeof(double) * (size_t)n));
        if (ok) {
            launch_extract_dinv(s, elementsVector->data, rowOffsetsVector->data, columnIndecesVector->data, n, 0, sc);     // 1/d
            hipLaunchKernelGGL(lanczos_scale_kernel, dim3(grid_of(n)), dim3(kBlock), 0, s, (double*)nullptr, sc, (const double*)nullptr, n, 1);   // d^-1/2
        }
    }
    auto dot = [&](const double* a, const double* b) -> double {
        const int np = launch_dot_partials(s, a, b, n, ws.partials);
        launch_reduce(s, ws.partials, np, ws.hostScalar, 0);
        if (!MGCG_HIP(hipStreamSynchronize(s))) { ok = false; return NAN; }
        return ws.hostScalar[0];
    };
    const DcsrMatrix* dc = dcsr_lookup(cusparse, elementsVector->data, rowOffsetsVector->data, columnIndecesVector->data, n, elementsCount, 0, n);
    SpmvConfig cfg; cfg.kernel = cusparse->kernel; cfg.rowsPerBlock = cusparse->rowsPerBlock; cfg.flags = cusparse->flags & ~6; cfg.gridBlocks = cusparse->gridBlocks;

    std::vector<double> alpha, beta(1, 0.0);               // T = tridiag(beta[1..], alpha[0..], beta[1..])
    if (ok) {
        hipLaunchKernelGGL(lanczos_start_kernel, dim3(grid_of(n)), dim3(kBlock), 0, s, v, n, (unsigned long long)seed);
        launch_fill(s, vPrev, 0.0, n);
        const double nrm = std::sqrt(dot(v, v));
        if (ok && nrm > 0.0) launch_scal(s, v, 1.0 / nrm, n); else ok = false;
    }
    double betaJ = 0.0, scale = 0.0;
    for (int j = 0; ok && j < steps; ++j) {
        SpmvArgs a{};
        a.elements = elementsVector->data; a.rowOffsets = rowOffsetsVector->data; a.columnIndeces = columnIndecesVector->data;
        a.elementsCount = elementsCount; a.rowCount = count; a.columnCount = count; a.alpha = 1.0; a.beta = 0.0; a.y = w;
        if (jacobiScaled) {
            hipLaunchKernelGGL(lanczos_scale_kernel, dim3(grid_of(n)), dim3(kBlock), 0, s, u, sc, (const double*)v, n, 0);     // u = S v
            a.x = u;
            launch_spmv_auto(s, EPI_AXPBY, a, cfg, dc);                                                                        // w = A u
            hipLaunchKernelGGL(lanczos_scale_kernel, dim3(grid_of(n)), dim3(kBlock), 0, s, w, sc, (const double*)w, n, 0);     // w = S w
        } else {
            a.x = v;
            launch_spmv_auto(s, EPI_AXPBY, a, cfg, dc);
        }
        const double aj = dot(w, v);
        if (!ok) break;
        alpha.push_back(aj);
        if (stepsDone) *stepsDone = (int)alpha.size();
        scale = std::fmax(scale, std::fabs(aj) + betaJ);
        launch_axpy(s, w, v, n, -aj);
        if (j > 0) launch_axpy(s, w, vPrev, n, -betaJ);
        const double bn = std::sqrt(dot(w, w));
        if (!ok) break;
        if (!(bn > 1e-13 * scale)) break;                  // invariant subspace found: T holds exact eigenvalues
        beta.push_back(bn);
        betaJ = bn;
        launch_scal(s, w, 1.0 / bn, n);
        double* t = vPrev; vPrev = v; v = w; w = t;
    }
    ok = MGCG_HIP(hipStreamSynchronize(s)) && ok;
    for (double* p : { vPrev, v, w, u, sc }) if (p) (void)hipFree(p);
    if (!ok || alpha.empty()) { if (ok) set_error("MgcgEstimateSpectrum: the start vector is zero"); return MGCG_ERROR; }
    const int k = (int)alpha.size();
    if (lambdaMin) *lambdaMin = tridiagonal_eigenvalue(alpha, beta, k, 0);
    if (lambdaMax) *lambdaMax = tridiagonal_eigenvalue(alpha, beta, k, k - 1);
    if (ritz) for (int i = 0; i < k; ++i) ritz[i] = tridiagonal_eigenvalue(alpha, beta, k, i);
    return MGCG_OK;
}
