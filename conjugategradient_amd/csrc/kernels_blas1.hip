// BLAS-1 and fused CG vector updates for gfx950.
// Replaces cublasD{axpy,dot,scal,copy}_v2 (Mgcg/cuBlas/MgcgGpu/Mgcg.cu:22-54) and the OpenCL
// AddVectorVector / MultiplyVectorVector / ReductionSum / ReductionMaxAbsolute kernels
// (Mgcg/HandmadeCL/MgcgCL/Mgcg.cl:15-159).
//
// All kernels are grid-stride over a fixed grid (<= 8 workgroups of 256 threads per CU) with
// 16-byte loads when the operands allow it.  Element-wise arithmetic keeps the reference CPU
// twin's operation order (Mgcg/cuBlas/Mgcg/LongVector.cs:41-51: answer = left + a*right, product
// rounded first; the library is built with -ffp-contract=off), so x, r and p updates are
// bit-identical to it.  Reductions: per-lane partial -> __shfl_down over the 64-lane wavefront ->
// LDS across the 4 waves -> one partial per workgroup -> a single-workgroup second stage that adds
// the partials in a fixed order.  No atomics: results are run-to-run reproducible.
#include "common.hpp"

namespace mgcg {

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double wave_sum_b(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max_b(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { double o = __shfl_down(v, off, 64); v = o > v ? o : v; }
    return v;
}
__device__ __forceinline__ double block_sum(double v, double* s_red)
{
    v = wave_sum_b(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ double block_max(double v, double* s_red)
{
    v = wave_max_b(v);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    double a = s_red[0] > s_red[1] ? s_red[0] : s_red[1];
    double b = s_red[2] > s_red[3] ? s_red[2] : s_red[3];
    return a > b ? a : b;
}

static inline int grid_for(long long n, int perThread)
{
    const int cap = kMaxGrid;
    long long blocks = (n + (long long)kBlock * perThread - 1) / ((long long)kBlock * perThread);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}
static inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

// Element-wise grid-stride driver.  V2: f2(i) handles elements i, i+1 with 16-byte accesses and the
// odd tail goes to f1 on one thread; otherwise f1(i) per element.
// Streaming hint of the CG vector passes (non-temporal loads and stores).  Measured per size, alternating inside one process
// (profiles/r2/vec_nt_ab.log; the A/B script is in the history): with the hint the CG iteration is 2-11 % faster from 4 M rows up (11 % at the 16.8 M rows
// of one rank's slab of an 8-GPU run, 2 % at 134 M) and 2.5 % slower at 2 M rows and below, where every vector stays in the caches anyway.
template <bool NTV, typename T> __device__ __forceinline__ T ldv(const T* p) { if constexpr (NTV) return __builtin_nontemporal_load(p); else return *p; }
template <bool NTV, typename T> __device__ __forceinline__ void stv(const T& v, T* p) { if constexpr (NTV) __builtin_nontemporal_store(v, p); else *p = v; }
static bool vec_nt(long long n) { return n > 3000000; }

template <bool V2, typename F2, typename F1>
__device__ __forceinline__ void grid_stride(long long n, F2 f2, F1 f1)
{
    const long long stride = (long long)gridDim.x * kBlock;
    if constexpr (V2) {
        const long long n2 = n >> 1;
        for (long long i2 = (long long)blockIdx.x * kBlock + threadIdx.x; i2 < n2; i2 += stride) f2(i2 * 2);
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) f1(n - 1);
    } else {
        for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) f1(i);
    }
}

// Contiguous-chunk driver for the loop's own vector updates: workgroup b walks its own run of element pairs, two
// 16-byte accesses per array in flight per lane.  Measured 8-15 % faster than the grid-stride form on the 1 GiB vectors
// of the 512^3 system (profiles/r1/vec_probe_update_kernels.log); same arithmetic per element.
template <typename F2>
__device__ __forceinline__ void chunk_pairs(long long n2, F2 f2x2)
{
    const long long per = ((n2 + gridDim.x - 1) / gridDim.x + (kBlock - 1)) & ~(long long)(kBlock - 1);
    long long i = per * blockIdx.x + threadIdx.x;
    long long end = per * (blockIdx.x + 1);
    end = end < n2 ? end : n2;
    for (; i < end; i += 2 * kBlock) f2x2(i, i + kBlock < end);    // pairs i and i + kBlock
}

// ------------------------------------------------------------------ axpy: y = y + alpha*x
template <bool V2>
__global__ __launch_bounds__(kBlock) void axpy_kernel(double* __restrict__ y, const double* __restrict__ x, long long n, double alpha)
{
    grid_stride<V2>(n,
        [&](long long i) { d2 xv = *(const d2*)(x + i); d2 yv = *(d2*)(y + i); d2 t; t.x = alpha * xv.x; t.y = alpha * xv.y; yv.x = yv.x + t.x; yv.y = yv.y + t.y; *(d2*)(y + i) = yv; },
        [&](long long i) { double t = alpha * x[i]; y[i] = y[i] + t; });
}
void launch_axpy(hipStream_t s, double* y, const double* x, long long n, double alpha)
{
    if (n <= 0) return;
    if (al16(y) && al16(x)) hipLaunchKernelGGL(axpy_kernel<true>, dim3(grid_for(n, 2)), dim3(kBlock), 0, s, y, x, n, alpha);
    else hipLaunchKernelGGL(axpy_kernel<false>, dim3(grid_for(n, 1)), dim3(kBlock), 0, s, y, x, n, alpha);
}

// ------------------------------------------------------------------ scal: x = alpha*x
template <bool V2>
__global__ __launch_bounds__(kBlock) void scal_kernel(double* __restrict__ x, long long n, double alpha)
{
    grid_stride<V2>(n,
        [&](long long i) { d2 v = *(d2*)(x + i); v.x = alpha * v.x; v.y = alpha * v.y; *(d2*)(x + i) = v; },
        [&](long long i) { x[i] = alpha * x[i]; });
}
void launch_scal(hipStream_t s, double* x, double alpha, long long n)
{
    if (n <= 0) return;
    if (al16(x)) hipLaunchKernelGGL(scal_kernel<true>, dim3(grid_for(n, 2)), dim3(kBlock), 0, s, x, n, alpha);
    else hipLaunchKernelGGL(scal_kernel<false>, dim3(grid_for(n, 1)), dim3(kBlock), 0, s, x, n, alpha);
}

// ------------------------------------------------------------------ xpay: y = x + beta*y   (Scal+Axpy of Mgcg.cu:197,265 fused)
template <bool V2>
__global__ __launch_bounds__(kBlock) void xpay_kernel(double* __restrict__ y, const double* __restrict__ x, long long n, double beta)
{
    grid_stride<V2>(n,
        [&](long long i) { d2 xv = *(const d2*)(x + i); d2 yv = *(d2*)(y + i); d2 t; t.x = beta * yv.x; t.y = beta * yv.y; yv.x = xv.x + t.x; yv.y = xv.y + t.y; *(d2*)(y + i) = yv; },
        [&](long long i) { double t = beta * y[i]; y[i] = x[i] + t; });
}
void launch_xpay(hipStream_t s, double* y, const double* x, long long n, double beta)
{
    if (n <= 0) return;
    if (al16(y) && al16(x)) hipLaunchKernelGGL(xpay_kernel<true>, dim3(grid_for(n, 2)), dim3(kBlock), 0, s, y, x, n, beta);
    else hipLaunchKernelGGL(xpay_kernel<false>, dim3(grid_for(n, 1)), dim3(kBlock), 0, s, y, x, n, beta);
}

// ------------------------------------------------------------------ copy / fill
__global__ __launch_bounds__(kBlock) void fill_kernel(double* __restrict__ y, double v, long long n)
{
    grid_stride<false>(n, [&](long long) {}, [&](long long i) { y[i] = v; });
}
void launch_copy(hipStream_t s, double* y, const double* x, long long n)
{
    if (n <= 0) return;
    (void)hipMemcpyAsync(y, x, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s);
}
void launch_fill(hipStream_t s, double* y, double v, long long n)
{
    if (n <= 0) return;
    if (v == 0.0) { (void)hipMemsetAsync(y, 0, (size_t)n * sizeof(double), s); return; }
    hipLaunchKernelGGL(fill_kernel, dim3(grid_for(n, 1)), dim3(kBlock), 0, s, y, v, n);
}

// ------------------------------------------------------------------ dot partials
// NT: streaming loads for vectors far larger than the caches (a pure read stream gains 4-10 % from them on this chip,
// profiles/r2/bw_probe.log; small vectors that the next kernel reads again keep the plain loads)
template <bool V2, bool NT = false>
__global__ __launch_bounds__(kBlock) void dot_kernel(const double* __restrict__ x, const double* __restrict__ y, long long n, double* __restrict__ partials)
{
    __shared__ double s_red[4];
    double acc = 0.0;
    grid_stride<V2>(n,
        [&](long long i) {
            d2 xv, yv;
            if constexpr (NT) { xv = __builtin_nontemporal_load((const d2*)(x + i)); yv = __builtin_nontemporal_load((const d2*)(y + i)); }
            else { xv = *(const d2*)(x + i); yv = *(const d2*)(y + i); }
            double t0 = xv.x * yv.x; double t1 = xv.y * yv.y; acc += t0; acc += t1; },
        [&](long long i) { double t = x[i] * y[i]; acc += t; });
    const double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
// ------------------------------------------------------------------ reference-order dot (validation mode: knob dot_order)
// The reference's CPU twin adds the rounded products strictly left to right (Mgcg/cuBlas/Mgcg/LongVector.cs:15-31) and the host adds
// the per-device sums in device order (ConjugateGradientParallelGpu.cs:463,499,525).  Everything element-wise in this library is
// already bit-identical to that twin; with dot_order = 1 every dot product of Dot / CsrMVDot / Solve / Solve0..2 / SolveParallel /
// SolveMg* takes this kernel instead of the tree sums, so whole residual traces and iterates become EQUAL to the oracle's, at any size
// and rank count.  One workgroup: waves 1-3 stage the rounded products of the next batch in LDS (coalesced) while lane 0 of wave 0
// adds the current batch in index order -- one dependent fp64 add per element, ~0.5 s per 1.3e8 entries.  A test instrument, never
// on a timed path.  (Padding a batch with +0.0 changes nothing: a running sum that started at +0.0 is never -0.0.)
constexpr int kSerialBatch = 2048;
__global__ __launch_bounds__(kBlock) void dot_serial_kernel(const double* __restrict__ x, const double* __restrict__ y, long long n,
                                                            double* __restrict__ out, const int* done)
{
    __shared__ double s_prod[2][kSerialBatch];
    if (done != nullptr && *done != 0) return;
    const int tid = threadIdx.x;
    const long long nBatches = (n + kSerialBatch - 1) / kSerialBatch;
    auto fill = [&](int buf, long long b) {
        const long long base = b * kSerialBatch;
        for (int k = tid - kWave; k < kSerialBatch; k += kBlock - kWave) {
            const long long i = base + k;
            double t = 0.0;
            if (i < n) t = x[i] * y[i];
            s_prod[buf][k] = t;
        }
    };
    if (tid >= kWave) fill(0, 0);
    __syncthreads();
    double acc = 0.0;
    for (long long b = 0; b < nBatches; ++b) {
        if (tid >= kWave) { if (b + 1 < nBatches) fill((int)((b + 1) & 1), b + 1); }
        else if (tid == 0) {
            const double* q = s_prod[b & 1];
#pragma unroll 16
            for (int k = 0; k < kSerialBatch; ++k) acc += q[k];
        }
        __syncthreads();
    }
    if (tid == 0) out[0] = acc;
}
bool dot_reference_order() { return tuning().dotOrder.load(std::memory_order_relaxed) != 0; }
void launch_dot_serial(hipStream_t s, const double* x, const double* y, long long n, double* out, const int* done)
{
    hipLaunchKernelGGL(dot_serial_kernel, dim3(1), dim3(kBlock), 0, s, x, y, n < 0 ? 0 : n, out, done);
}

int launch_dot_partials(hipStream_t s, const double* x, const double* y, long long n, double* partials)
{
    if (dot_reference_order()) { launch_dot_serial(s, x, y, n, partials, nullptr); return 1; }
    const bool v2 = al16(x) && al16(y);
    const int grid = grid_for(n, v2 ? 4 : 2);
    if (v2 && n >= (8LL << 20)) hipLaunchKernelGGL((dot_kernel<true, true>), dim3(grid), dim3(kBlock), 0, s, x, y, n, partials);
    else if (v2) hipLaunchKernelGGL((dot_kernel<true, false>), dim3(grid), dim3(kBlock), 0, s, x, y, n, partials);
    else hipLaunchKernelGGL((dot_kernel<false, false>), dim3(grid), dim3(kBlock), 0, s, x, y, n, partials);
    return grid;
}

__global__ __launch_bounds__(kBlock) void nrminf_kernel(const double* __restrict__ x, long long n, double* __restrict__ partials)
{
    __shared__ double s_red[4];
    double m = 0.0;
    grid_stride<false>(n, [&](long long) {}, [&](long long i) { double a = fabs(x[i]); m = a > m ? a : m; });
    const double t = block_max(m, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
int launch_nrminf_partials(hipStream_t s, const double* x, long long n, double* partials)
{
    const int grid = grid_for(n, 2);
    hipLaunchKernelGGL(nrminf_kernel, dim3(grid), dim3(kBlock), 0, s, x, n, partials);
    return grid;
}

// ------------------------------------------------------------------ second stage (one workgroup)
// Fixed-order sum (or max) of n partials; result valid in thread 0.
__device__ __forceinline__ double reduce_partials_block(const double* __restrict__ partials, int n, double* s_red, int mode)
{
    double acc = 0.0;
    if (mode == 0) { for (int i = threadIdx.x; i < n; i += kBlock) acc += partials[i]; return block_sum(acc, s_red); }
    for (int i = threadIdx.x; i < n; i += kBlock) { double a = partials[i]; acc = a > acc ? a : acc; }
    return block_max(acc, s_red);
}

__global__ __launch_bounds__(kBlock) void reduce_kernel(const double* __restrict__ partials, int n, double* __restrict__ out, int mode, const int* done)
{
    __shared__ double s_red[4];
    if (done != nullptr && *done != 0) return;
    const double t = reduce_partials_block(partials, n, s_red, mode);
    if (threadIdx.x == 0) out[0] = t;
}
// partials[0] = sum / max of partials[0..n), in place (every read happens before the barrier inside the block reduction)
__global__ __launch_bounds__(kBlock) void reduce_inplace_kernel(double* partials, int n, int mode, const int* done)
{
    __shared__ double s_red[4];
    if (done != nullptr && *done != 0) return;
    const double t = reduce_partials_block(partials, n, s_red, mode);
    if (threadIdx.x == 0) partials[0] = t;
}
void launch_reduce(hipStream_t s, const double* partials, int n, double* out, int mode)
{
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(kBlock), 0, s, partials, n, out, mode, (const int*)nullptr);
}
void launch_reduce_to(hipStream_t s, const double* partials, int n, double* dst, const int* done)
{
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(kBlock), 0, s, partials, n, dst, 0, done);
}
// two sums in one launch (the preconditioned loop of several ranks: r.r and r.z in front of their one all-reduce), each in reduce_kernel's order
__global__ __launch_bounds__(kBlock) void reduce2_kernel(const double* __restrict__ pA, int nA, double* __restrict__ dstA,
                                                         const double* __restrict__ pB, int nB, double* __restrict__ dstB, const int* done)
{
    __shared__ double s_red[4];
    __shared__ double s_red2[4];
    if (done != nullptr && *done != 0) return;
    const double ta = reduce_partials_block(pA, nA, s_red, 0);
    const double tb = reduce_partials_block(pB, nB, s_red2, 0);
    if (threadIdx.x == 0) { dstA[0] = ta; dstB[0] = tb; }
}
void launch_reduce2_to(hipStream_t s, const double* pA, int nA, double* dstA, const double* pB, int nB, double* dstB, const int* done)
{
    hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(kBlock), 0, s, pA, nA, dstA, pB, nB, dstB, done);
}

// ------------------------------------------------------------------ fused CG pieces
// p = r ; partial r.r
template <bool V2>
__global__ __launch_bounds__(kBlock) void copy_dot_kernel(double* __restrict__ p, const double* __restrict__ r, long long n, double* __restrict__ partials, const int* done)
{
    __shared__ double s_red[4];
    if (done != nullptr && *done != 0) return;
    double acc = 0.0;
    grid_stride<V2>(n,
        [&](long long i) { d2 rv = *(const d2*)(r + i); *(d2*)(p + i) = rv; double t0 = rv.x * rv.x; double t1 = rv.y * rv.y; acc += t0; acc += t1; },
        [&](long long i) { double rv = r[i]; p[i] = rv; double t = rv * rv; acc += t; });
    const double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
int launch_copy_dot(hipStream_t s, double* p, const double* r, long long n, double* partials, const int* done)
{
    const bool v2 = al16(p) && al16(r);
    const int grid = grid_for(n, v2 ? 4 : 2);
    if (v2) hipLaunchKernelGGL(copy_dot_kernel<true>, dim3(grid), dim3(kBlock), 0, s, p, r, n, partials, done);
    else hipLaunchKernelGGL(copy_dot_kernel<false>, dim3(grid), dim3(kBlock), 0, s, p, r, n, partials, done);
    if (dot_reference_order()) { launch_dot_serial(s, r, r, n, partials, done); return 1; }      // the one sum in the reference's order replaces the partial sums
    return grid;
}

// alpha = rr / pAp ; x = x + alpha*p ; r = r + (-alpha)*Ap ; partial r.r [, partial max|r|]
// (ConjugateGradientCpu.cs:71-74 in one pass over x, p, r, Ap)
template <bool V2, bool INF>
__global__ __launch_bounds__(kBlock) void update_xr_kernel(const CgScalars* __restrict__ sc, double* __restrict__ x, double* __restrict__ r,
                                                           const double* __restrict__ p, const double* __restrict__ Ap, long long n,
                                                           double* __restrict__ partials, double* __restrict__ partialsInf)
{
    __shared__ double s_red[4];
    __shared__ double s_red2[4];
    if (sc->done != 0) return;
    const double alpha = sc->rr / sc->pAp;
    const double malpha = -alpha;
    double acc = 0.0, mx = 0.0;
    grid_stride<V2>(n,
        [&](long long i) {
            d2 pv = *(const d2*)(p + i); d2 xv = *(d2*)(x + i); d2 av = *(const d2*)(Ap + i); d2 rv = *(d2*)(r + i);
            double t0 = alpha * pv.x; double t1 = alpha * pv.y; xv.x = xv.x + t0; xv.y = xv.y + t1; *(d2*)(x + i) = xv;
            double u0 = malpha * av.x; double u1 = malpha * av.y; rv.x = rv.x + u0; rv.y = rv.y + u1; *(d2*)(r + i) = rv;
            double q0 = rv.x * rv.x; double q1 = rv.y * rv.y; acc += q0; acc += q1;
            if (INF) { double a0 = fabs(rv.x); double a1 = fabs(rv.y); mx = a0 > mx ? a0 : mx; mx = a1 > mx ? a1 : mx; }
        },
        [&](long long i) {
            double t = alpha * p[i]; x[i] = x[i] + t;
            double u = malpha * Ap[i]; double rv = r[i] + u; r[i] = rv;
            double q = rv * rv; acc += q;
            if (INF) { double a0 = fabs(rv); mx = a0 > mx ? a0 : mx; }
        });
    const double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    if (INF) {
        const double m = block_max(mx, s_red2);
        if (threadIdx.x == 0) partialsInf[blockIdx.x] = m;
    }
}
int launch_update_xr(hipStream_t s, const CgScalars* sc, double* x, double* r, const double* p, const double* Ap,
                     long long n, double* partials, double* partialsInf)
{
    const bool v2 = al16(x) && al16(r) && al16(p) && al16(Ap);
    const int grid = grid_for(n, v2 ? 4 : 2);
    const bool inf = partialsInf != nullptr;
#define GO(V, I) hipLaunchKernelGGL((update_xr_kernel<V, I>), dim3(grid), dim3(kBlock), 0, s, sc, x, r, p, Ap, n, partials, partialsInf)
    if (v2) { if (inf) GO(true, true); else GO(true, false); }
    else { if (inf) GO(false, true); else GO(false, false); }
#undef GO
    if (dot_reference_order()) {
        launch_dot_serial(s, r, r, n, partials, nullptr);
        if (inf) hipLaunchKernelGGL(reduce_inplace_kernel, dim3(1), dim3(kBlock), 0, s, partialsInf, grid, 1, (const int*)nullptr);   // max of the partial maxima (order-free)
        return 1;
    }
    return grid;
}

// The device-resident loop splits the reference's update (ConjugateGradientCpu.cs:72-74,94) differently from the
// reference's phase functions: r first (its norm decides the stop test), then x and p together, so that p is read
// once for both x += alpha p and p = z + beta p  (64N bytes per iteration instead of 72N).
// alpha = rr / pAp ; r = r + (-alpha)*Ap ; partial r.r [, partial max|r|]
template <bool V2, bool INF, bool NTV>
__global__ __launch_bounds__(kBlock) void update_r_kernel(CgScalars* __restrict__ sc, double* __restrict__ r, const double* __restrict__ Ap, long long n,
                                                          double* __restrict__ partials, double* __restrict__ partialsInf,
                                                          const double* __restrict__ pApPartials, int nPAp, int freeze)
{
    __shared__ double s_red[4];
    __shared__ double s_red2[4];
    __shared__ double s_pAp;
    if (sc->done != 0) { if (freeze && blockIdx.x == 0 && threadIdx.x == 0) sc->fDone = 1; return; }   // tells the folded x/p update that no iteration ran
    double pAp;
    if (pApPartials != nullptr) {                                     // single-rank loop: no separate reduction launch
        const double t = reduce_partials_block(pApPartials, nPAp, s_red, 0);
        if (threadIdx.x == 0) s_pAp = t;
        __syncthreads();
        pAp = s_pAp;
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->pAp = pAp;
    } else {
        pAp = sc->pAp;
    }
    const double alpha = sc->rr / pAp;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc->alpha = alpha;                                            // for update_xp of this iteration
        if (freeze) { sc->fRr = sc->rr; sc->fRr0 = sc->rr0; sc->fAlpha = alpha; sc->fIteration = sc->iteration; sc->fDone = 0; }
    }
    const double malpha = -alpha;
    double acc = 0.0, mx = 0.0;
    auto one = [&](long long i) {
        double u = malpha * Ap[i]; double rv = r[i] + u; r[i] = rv;
        double q = rv * rv; acc += q;
        if (INF) { double a0 = fabs(rv); mx = a0 > mx ? a0 : mx; }
    };
    if constexpr (V2) {
        d2* r2 = (d2*)r; const d2* a2 = (const d2*)Ap;
        auto fin = [&](d2& rv, const d2& av) {
            double u0 = malpha * av.x; double u1 = malpha * av.y; rv.x = rv.x + u0; rv.y = rv.y + u1;
            double q0 = rv.x * rv.x; double q1 = rv.y * rv.y; acc += q0; acc += q1;
            if (INF) { double a0 = fabs(rv.x); double a1 = fabs(rv.y); mx = a0 > mx ? a0 : mx; mx = a1 > mx ? a1 : mx; }
        };
        chunk_pairs(n >> 1, [&](long long i, bool two) {
            const long long j = two ? i + kBlock : i;
            // streaming hints as in update_xp (measured there: 5.2 -> 6.3 TB/s): Ap is not read again, r not before 2 GB of other traffic
            d2 av0 = ldv<NTV>(a2 + i), rv0 = ldv<NTV>(r2 + i), av1 = ldv<NTV>(a2 + j), rv1 = ldv<NTV>(r2 + j);
            fin(rv0, av0); stv<NTV>(rv0, r2 + i);
            if (two) { fin(rv1, av1); stv<NTV>(rv1, r2 + j); }
        });
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) one(n - 1);
    } else {
        grid_stride<false>(n, [&](long long) {}, one);
    }
    const double t = block_sum(acc, s_red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
    if (INF) {
        const double m = block_max(mx, s_red2);
        if (threadIdx.x == 0) partialsInf[blockIdx.x] = m;
    }
}
int launch_update_r(hipStream_t s, CgScalars* sc, double* r, const double* Ap, long long n, double* partials, double* partialsInf,
                    const double* pApPartials, int nPAp, bool freeze)
{
    const bool v2 = al16(r) && al16(Ap);
    int grid = grid_for(n, v2 ? 4 : 2);
    // two workgroups per CU measured best for this 2-reads-1-write pass (0.565 ms against 0.59-0.61 for 768 / 1024 / 2048 and 0.72 for 256
    // workgroups at 512^3; the 3-reads-2-writes x/p pass keeps 2048)
    DeviceState* d = device_state();
    const int want = 2 * (d ? d->numCu : kNumCu);
    if (grid > want) grid = want;
    const bool inf = partialsInf != nullptr;
    const bool nt = vec_nt(n);
#define GO(V, I) do { if (nt) hipLaunchKernelGGL((update_r_kernel<V, I, true>), dim3(grid), dim3(kBlock), 0, s, sc, r, Ap, n, partials, partialsInf, pApPartials, nPAp, freeze ? 1 : 0); \
                      else hipLaunchKernelGGL((update_r_kernel<V, I, false>), dim3(grid), dim3(kBlock), 0, s, sc, r, Ap, n, partials, partialsInf, pApPartials, nPAp, freeze ? 1 : 0); } while (0)
    if (v2) { if (inf) GO(true, true); else GO(true, false); }
    else { if (inf) GO(false, true); else GO(false, false); }
#undef GO
    if (dot_reference_order()) {
        launch_dot_serial(s, r, r, n, partials, &sc->done);
        if (inf) hipLaunchKernelGGL(reduce_inplace_kernel, dim3(1), dim3(kBlock), 0, s, partialsInf, grid, 1, (const int*)&sc->done);
        return 1;
    }
    return grid;
}

// x = x + alpha*p (whenever the iteration ran: sc->pad is the finalisation kernel's "x pending" mark) and, unless the
// stop test fired, p = z + beta*p.
template <bool V2, bool NTV>
__global__ __launch_bounds__(kBlock) void update_xp_kernel(const CgScalars* __restrict__ sc, double* __restrict__ x, double* __restrict__ p,
                                                           const double* __restrict__ z, long long n)
{
    if (sc->pad == 0) return;                   // this iteration did not run (the loop had already stopped)
    const double alpha = sc->alpha, beta = sc->beta;
    if (sc->done != 0) {                        // the iteration that stopped the loop: x only
        grid_stride<V2>(n,
            [&](long long i) { d2 pv = *(const d2*)(p + i); d2 xv = *(d2*)(x + i); double t0 = alpha * pv.x; double t1 = alpha * pv.y; xv.x = xv.x + t0; xv.y = xv.y + t1; *(d2*)(x + i) = xv; },
            [&](long long i) { double t = alpha * p[i]; x[i] = x[i] + t; });
        return;
    }
    auto one = [&](long long i) { const double pv = p[i]; double t = alpha * pv; x[i] = x[i] + t; double u = beta * pv; p[i] = z[i] + u; };
    if constexpr (V2) {
        // streaming hints on both sides: none of x, p, z is touched again before ~2 GB of other traffic
        d2* x2 = (d2*)x; d2* p2 = (d2*)p; const d2* z2 = (const d2*)z;
        auto fin = [&](d2& xv, d2& pv, const d2& zv) {
            double t0 = alpha * pv.x; double t1 = alpha * pv.y; xv.x = xv.x + t0; xv.y = xv.y + t1;
            double u0 = beta * pv.x; double u1 = beta * pv.y; pv.x = zv.x + u0; pv.y = zv.y + u1;
        };
        chunk_pairs(n >> 1, [&](long long i, bool two) {
            const long long j = two ? i + kBlock : i;
            d2 pv0 = ldv<NTV>(p2 + i), xv0 = ldv<NTV>(x2 + i), zv0 = ldv<NTV>(z2 + i);
            d2 pv1 = ldv<NTV>(p2 + j), xv1 = ldv<NTV>(x2 + j), zv1 = ldv<NTV>(z2 + j);
            fin(xv0, pv0, zv0);
            stv<NTV>(xv0, x2 + i); stv<NTV>(pv0, p2 + i);
            if (two) { fin(xv1, pv1, zv1); stv<NTV>(xv1, x2 + j); stv<NTV>(pv1, p2 + j); }
        });
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) one(n - 1);
    } else {
        grid_stride<false>(n, [&](long long) {}, one);
    }
}
void launch_update_xp(hipStream_t s, const CgScalars* sc, double* x, double* p, const double* z, long long n)
{
    if (n <= 0) return;
    const bool v2 = al16(x) && al16(p) && al16(z);
    const bool nt = vec_nt(n);
    if (v2 && nt) hipLaunchKernelGGL((update_xp_kernel<true, true>), dim3(grid_for(n, 2)), dim3(kBlock), 0, s, sc, x, p, z, n);
    else if (v2) hipLaunchKernelGGL((update_xp_kernel<true, false>), dim3(grid_for(n, 2)), dim3(kBlock), 0, s, sc, x, p, z, n);
    else hipLaunchKernelGGL((update_xp_kernel<false, false>), dim3(grid_for(n, 1)), dim3(kBlock), 0, s, sc, x, p, z, n);
}

// The stop decision of one iteration (the five rules of SURVEY.md 3.5): residual to show, stop or not, status.
struct StopDecision { double res, shown; bool stop; int status; };
__device__ __forceinline__ StopDecision decide_stop(const FinalizeArgs& f, double rrNew, double inf, double rr0, int it)
{
    StopDecision d;
    d.res = sqrt(rrNew);
    if (f.rule == MGCG_RULE_HANDMADECL) d.res = inf;
    d.shown = d.res;
    bool converged;
    switch (f.rule) {
    case MGCG_RULE_NATIVE:   converged = (f.minIt <= it) && (d.res < f.tol); break;
    case MGCG_RULE_SIMPLE:   converged = (f.minIt < it) && (d.res < f.tol); break;
    case MGCG_RULE_VIENNACL: d.shown = sqrt(rrNew / rr0); converged = (f.minIt < it) && (rrNew / rr0 < f.tol * f.tol); break;
    default:                 converged = (it >= f.minIt) && (it <= f.maxIt) && (d.res < f.tol); break;  // ConjugateGradient.cs:56-79
    }
    d.status = MGCG_OK;
    d.stop = converged;
    if (!d.stop && it >= f.minIt && it > f.maxIt) { d.stop = true; d.status = MGCG_MAXIT_EXCEEDED; }
    if (!d.stop && !(d.res == d.res && fabs(d.res) <= 1.79e308)) { d.stop = true; d.status = MGCG_NONFINITE; }
    return d;
}

// The x/p update with the iteration's finalisation folded in (single rank, no preconditioner: z = r).  Every workgroup reduces the
// r.r (and max|r|) partial sums of update_r in the order finalize_kernel uses and takes the same decision from the values update_r
// froze (fRr, fRr0, fAlpha, fIteration, fDone); workgroup 0 alone rewrites the live scalars, the host mirror and the trace, which
// nothing in this kernel reads.  The arithmetic of x and p is update_xp_kernel's.
template <bool V2, bool NTV>
__global__ __launch_bounds__(kBlock) void update_xp_final_kernel(FinalizeArgs f, const double* __restrict__ partials, const double* __restrict__ partialsInf,
                                                                 int nPartials, double* __restrict__ x, double* __restrict__ p, const double* __restrict__ z, long long n)
{
    __shared__ double s_red[4];
    __shared__ double s_red2[4];
    __shared__ double s_beta;
    __shared__ int s_stop;
    CgScalars* sc = f.sc;
    if (sc->fDone != 0) return;                                       // the loop had stopped before this iteration: nothing ran, nothing is pending
    // nPartials == 0: several ranks -- r.r has been reduced and all-reduced into sc->rrNew before this launch (workgroup 0 writes the
    // same bits back below)
    const double rrNew = nPartials > 0 ? reduce_partials_block(partials, nPartials, s_red, 0) : sc->rrNew;
    double inf = 0.0;
    if (partialsInf != nullptr && nPartials > 0) inf = reduce_partials_block(partialsInf, nPartials, s_red2, 1);
    const double alpha = sc->fAlpha;
    if (threadIdx.x == 0) {
        const int it = sc->fIteration;
        const StopDecision d = decide_stop(f, rrNew, inf, sc->fRr0, it);
        const double beta = rrNew / sc->fRr;
        s_stop = d.stop ? 1 : 0; s_beta = beta;
        if (blockIdx.x == 0) {                                        // publish (what finalize_kernel writes)
            if (f.trace != nullptr && it < f.traceCap) f.trace[it] = d.shown;
            sc->rrNew = rrNew; sc->residual = d.res; sc->nrmInf = inf; sc->pad = 0;
            if (d.stop) {
                sc->done = 1; sc->status = d.status;
                f.mirror->residual = d.res; f.mirror->iteration = it; f.mirror->status = d.status;
                __threadfence_system();
                f.mirror->done = 1;
            } else {
                sc->beta = beta; sc->rr = rrNew; sc->iteration = it + 1;
                f.mirror->residual = d.res; f.mirror->iteration = it + 1;
            }
        }
    }
    __syncthreads();
    const double beta = s_beta;
    if (s_stop) {                                                     // the iteration that stops the loop: x only
        grid_stride<V2>(n,
            [&](long long i) { d2 pv = *(const d2*)(p + i); d2 xv = *(d2*)(x + i); double t0 = alpha * pv.x; double t1 = alpha * pv.y; xv.x = xv.x + t0; xv.y = xv.y + t1; *(d2*)(x + i) = xv; },
            [&](long long i) { double t = alpha * p[i]; x[i] = x[i] + t; });
        return;
    }
    auto one = [&](long long i) { const double pv = p[i]; double t = alpha * pv; x[i] = x[i] + t; double u = beta * pv; p[i] = z[i] + u; };
    if constexpr (V2) {
        d2* x2 = (d2*)x; d2* p2 = (d2*)p; const d2* z2 = (const d2*)z;
        auto fin = [&](d2& xv, d2& pv, const d2& zv) {
            double t0 = alpha * pv.x; double t1 = alpha * pv.y; xv.x = xv.x + t0; xv.y = xv.y + t1;
            double u0 = beta * pv.x; double u1 = beta * pv.y; pv.x = zv.x + u0; pv.y = zv.y + u1;
        };
        chunk_pairs(n >> 1, [&](long long i, bool two) {
            const long long j = two ? i + kBlock : i;
            d2 pv0 = ldv<NTV>(p2 + i), xv0 = ldv<NTV>(x2 + i), zv0 = ldv<NTV>(z2 + i);
            d2 pv1 = ldv<NTV>(p2 + j), xv1 = ldv<NTV>(x2 + j), zv1 = ldv<NTV>(z2 + j);
            fin(xv0, pv0, zv0);
            stv<NTV>(xv0, x2 + i); stv<NTV>(pv0, p2 + i);
            if (two) { fin(xv1, pv1, zv1); stv<NTV>(xv1, x2 + j); stv<NTV>(pv1, p2 + j); }
        });
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) one(n - 1);
    } else {
        grid_stride<false>(n, [&](long long) {}, one);
    }
}
void launch_update_xp_final(hipStream_t s, const FinalizeArgs& f, const double* partials, const double* partialsInf, int nPartials,
                            double* x, double* p, const double* z, long long n)
{
    if (n <= 0) return;
    const bool v2 = al16(x) && al16(p) && al16(z);
    const bool nt = vec_nt(n);
    const int g2 = grid_for(n, 2);
    if (v2 && nt) hipLaunchKernelGGL((update_xp_final_kernel<true, true>), dim3(g2), dim3(kBlock), 0, s, f, partials, partialsInf, nPartials, x, p, z, n);
    else if (v2) hipLaunchKernelGGL((update_xp_final_kernel<true, false>), dim3(g2), dim3(kBlock), 0, s, f, partials, partialsInf, nPartials, x, p, z, n);
    else hipLaunchKernelGGL((update_xp_final_kernel<false, false>), dim3(grid_for(n, 1)), dim3(kBlock), 0, s, f, partials, partialsInf, nPartials, x, p, z, n);
}

// p = z + beta*p   (ConjugateGradientCpu.cs:94 with z = r; the preconditioned loop passes z = M^-1 r)
template <bool V2>
__global__ __launch_bounds__(kBlock) void update_p_kernel(const CgScalars* __restrict__ sc, double* __restrict__ p, const double* __restrict__ z, long long n)
{
    if (sc->done != 0) return;
    const double beta = sc->beta;
    grid_stride<V2>(n,
        [&](long long i) { d2 zv = *(const d2*)(z + i); d2 pv = *(d2*)(p + i); double t0 = beta * pv.x; double t1 = beta * pv.y; pv.x = zv.x + t0; pv.y = zv.y + t1; *(d2*)(p + i) = pv; },
        [&](long long i) { double t = beta * p[i]; p[i] = z[i] + t; });
}
void launch_update_p(hipStream_t s, const CgScalars* sc, double* p, const double* z, long long n)
{
    if (n <= 0) return;
    const bool v2 = al16(p) && al16(z);
    if (v2) hipLaunchKernelGGL(update_p_kernel<true>, dim3(grid_for(n, 2)), dim3(kBlock), 0, s, sc, p, z, n);
    else hipLaunchKernelGGL(update_p_kernel<false>, dim3(grid_for(n, 1)), dim3(kBlock), 0, s, sc, p, z, n);
}

// ------------------------------------------------------------------ scalar bookkeeping (one workgroup)
__global__ __launch_bounds__(kBlock) void init_scalars_kernel(const double* __restrict__ partials, int n, int reduceFirst,
                                                              CgScalars* sc, HostMirror* mirror, int rule)
{
    __shared__ double s_red[4];
    double rr = 0.0;
    if (reduceFirst) rr = reduce_partials_block(partials, n, s_red, 0);
    if (threadIdx.x == 0) {
        if (!reduceFirst) rr = sc->rr;
        sc->rr = rr; sc->rr0 = rr; sc->pAp = 0; sc->rrNew = 0; sc->rzNew = 0; sc->residual = 0; sc->nrmInf = 0;
        sc->beta = 0; sc->alpha = 0; sc->iteration = 0; sc->done = 0; sc->status = 0; sc->pad = 0;
        sc->fRr = rr; sc->fRr0 = rr; sc->fAlpha = 0; sc->fIteration = 0; sc->fDone = 0;
        (void)rule;
        mirror->residual = 0; mirror->iteration = 0; mirror->status = 0; mirror->done = 0;
    }
}
void launch_init_scalars(hipStream_t s, const double* partials, int n, bool reduceFirst, CgScalars* sc, HostMirror* mirror, int rule)
{
    hipLaunchKernelGGL(init_scalars_kernel, dim3(1), dim3(kBlock), 0, s, partials, n, reduceFirst ? 1 : 0, sc, mirror, rule);
}

// Residual, stop test, beta and the rr hand-over.
__global__ __launch_bounds__(kBlock) void finalize_kernel(const double* __restrict__ partials, const double* __restrict__ partialsInf, int n,
                                                          int reduceFirst, FinalizeArgs f)
{
    __shared__ double s_red[4];
    __shared__ double s_red2[4];
    CgScalars* sc = f.sc;
    if (sc->done != 0) { if (threadIdx.x == 0) sc->pad = 0; return; }     // no iteration ran: nothing pending for update_xp
    double rrNew = 0.0, inf = 0.0;
    if (reduceFirst) {
        rrNew = reduce_partials_block(partials, n, s_red, 0);
        if (partialsInf != nullptr) inf = reduce_partials_block(partialsInf, n, s_red2, 1);
    }
    if (threadIdx.x != 0) return;
    if (!reduceFirst) { rrNew = sc->rrNew; inf = sc->nrmInf; }
    const int it = sc->iteration;
    const StopDecision d = decide_stop(f, rrNew, inf, sc->rr0, it);
    const double res = d.res, shown = d.shown;
    const bool stop = d.stop;
    const int status = d.status;
    if (f.trace != nullptr && it < f.traceCap) f.trace[it] = shown;
    sc->rrNew = rrNew; sc->residual = res; sc->nrmInf = inf;
    sc->pad = 1;                                 // this iteration's x += alpha p is still to be done (update_xp)
    if (stop) {
        sc->done = 1; sc->status = status;
        f.mirror->residual = res; f.mirror->iteration = it; f.mirror->status = status;
        __threadfence_system();
        f.mirror->done = 1;
    } else {
        if (!f.preconditioned) { sc->beta = rrNew / sc->rr; sc->rr = rrNew; }
        else if (f.preconditioned == 2) { const double rz = sc->rzNew; sc->beta = rz / sc->rr; sc->rr = rz; }   // (all-reduced r.z already in place)
        sc->iteration = it + 1;
        f.mirror->residual = res; f.mirror->iteration = it + 1;
    }
}
void launch_finalize(hipStream_t s, const double* partials, const double* partialsInf, int n, bool reduceFirst, const FinalizeArgs& f)
{
    hipLaunchKernelGGL(finalize_kernel, dim3(1), dim3(kBlock), 0, s, partials, partialsInf, n, reduceFirst ? 1 : 0, f);
}

// Preconditioned loop: rzNew = r.z ; beta = rzNew / rz ; rz = rzNew   (rz lives in sc->rr)
__global__ __launch_bounds__(kBlock) void finalize_precond_kernel(const double* __restrict__ partials, int n, int reduceFirst, CgScalars* sc)
{
    __shared__ double s_red[4];
    if (sc->done != 0) return;
    double rz = 0.0;
    if (reduceFirst) rz = reduce_partials_block(partials, n, s_red, 0);
    if (threadIdx.x != 0) return;
    if (!reduceFirst) rz = sc->rzNew;
    sc->rzNew = rz;
    sc->beta = rz / sc->rr;
    sc->rr = rz;
}
void launch_finalize_precond(hipStream_t s, const double* partials, int n, bool reduceFirst, CgScalars* sc)
{
    hipLaunchKernelGGL(finalize_precond_kernel, dim3(1), dim3(kBlock), 0, s, partials, n, reduceFirst ? 1 : 0, sc);
}

void preload_kernels_blas1() { preload_code_object(reinterpret_cast<const void*>(&fill_kernel)); }

} // namespace mgcg
