// CSR SpMV for gfx950 (replaces cusparseDcsrmv_v2 behind CsrMV, Mgcg/cuBlas/MgcgGpu/Mgcg.cu:10-19,
// and the OpenCL Matrix_x_Vector of Mgcg/HandmadeCL/MgcgCL/Mgcg.cl:171-216).
//
// Two kernels:
//  * spmv_stream_kernel -- for short rows (5/7-point stencils: 7 nnz/row is far below a 64-lane
//    wavefront).  A workgroup owns R consecutive rows; their nonzeros are ONE contiguous span of the
//    CSR arrays, so all 256 lanes stream it with 16-byte loads (4 column ids / 2+2 values per lane),
//    gather x, and park the rounded products in LDS.  Then one lane per row adds its products in
//    stored order.  HBM sees only wide coalesced loads; the row structure is resolved in LDS.
//    Because each product is rounded before it is added and the adds run left to right, the result
//    is bit-identical to the reference CPU loop (Mgcg/cuBlas/Mgcg/SparseMatrix.cs:68-88).
//  * spmv_vector_kernel -- for long rows (MgcgMain's 159/row): LANES lanes per row stride through
//    the row, __shfl_down tree inside the lane group.
// Both are grid-stride over a fixed grid (<= 8 workgroups per CU), which makes the fused dot-product
// partials (one per workgroup, fixed order) run-to-run reproducible.
#include "common.hpp"

namespace mgcg {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int    i4 __attribute__((ext_vector_type(4)));

template <bool NT, typename T>
__device__ __forceinline__ T ld_stream(const T* p)
{
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over the workgroup (256 threads = 4 waves) in a fixed order; result valid in thread 0.
__device__ __forceinline__ double block_sum_256(double v, double* s_red)
{
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

template <int EPI>
__device__ __forceinline__ void spmv_epilogue(const SpmvArgs& a, long long row, double acc, double& dotacc)
{
    if constexpr (EPI == EPI_AXPBY) {
        double v = a.alpha * acc;
        if (a.beta != 0.0) { double t = a.beta * a.y[row]; v = v + t; }
        a.y[row] = v;
    } else if constexpr (EPI == EPI_DOT) {
        a.y[row] = acc;
        double t = a.w[row] * acc;
        dotacc += t;
    } else if constexpr (EPI == EPI_RESIDUAL) {
        a.y[row] = a.b[row] - acc;
    } else if constexpr (EPI == EPI_RESIDUAL_DOT) {
        double r = a.b[row] - acc;
        a.y[row] = r;
        double t = r * r;
        dotacc += t;
    } else if constexpr (EPI == EPI_JACOBI) {
        double res = a.b[row] - acc;
        double t = a.dinv[row] * res;
        double s = a.omega * t;
        a.y[row] = a.w[row] + s;
    }
}

// R rows per workgroup, CH 16-byte chunks per lane per pass (pass capacity CAP = 1024*CH nonzeros).
template <int EPI, int R, int CH, bool NT, bool XCD, bool ALIGNED>
__global__ __launch_bounds__(kBlock) void spmv_stream_kernel(SpmvArgs a, int nRowBlocks)
{
    constexpr int CAP = kBlock * 4 * CH;
    __shared__ double s_prod[CAP];
    __shared__ int s_ro[R + 1];
    __shared__ double s_red[4];

    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;

    const int tid = threadIdx.x;
    long long rbBegin, rbEnd, rbStep;
    if constexpr (XCD) {
        // Workgroups b and b+8 share an XCD (round-robin dispatch): give each XCD one contiguous
        // eighth of the row blocks so neighbouring rows' x windows meet in that XCD's L2.
        const int xcd = blockIdx.x & (kNumXcd - 1);
        const int local = blockIdx.x >> 3;
        const int perXcd = gridDim.x >> 3;
        rbBegin = (long long)nRowBlocks * xcd / kNumXcd + local;
        rbEnd = (long long)nRowBlocks * (xcd + 1) / kNumXcd;
        rbStep = perXcd;
    } else {
        rbBegin = blockIdx.x; rbEnd = nRowBlocks; rbStep = gridDim.x;
    }

    double dotacc = 0.0;
    for (long long rb = rbBegin; rb < rbEnd; rb += rbStep) {
        const long long r0 = rb * R;
        const int nr = (int)(((long long)a.rowCount - r0) < R ? ((long long)a.rowCount - r0) : R);
        __syncthreads();                       // previous row block is done with s_ro / s_prod
        for (int t = tid; t <= nr; t += kBlock) s_ro[t] = a.rowOffsets[r0 + t];
        __syncthreads();
        const int s = s_ro[0], e = s_ro[nr];
        int my_s = e, my_e = e;
        if (tid < nr) { my_s = s_ro[tid]; my_e = s_ro[tid + 1]; }
        double acc = 0.0;
        const int tb0 = s & ~3;
        for (int tb = tb0; tb < e; tb += CAP) {
            if (tb != tb0) __syncthreads();    // the previous pass has been consumed
            // ---- load phase: every lane issues all its wide loads, then gathers ----
            i4 col[CH]; d2 va[CH], vb[CH]; bool wide[CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int k = tb + c * (kBlock * 4) + 4 * tid;
                wide[c] = ALIGNED && (k < e) && (k + 4 <= a.elementsCount);
                if (wide[c]) {
                    col[c] = ld_stream<NT>((const i4*)(a.columnIndeces + k));
                    va[c] = ld_stream<NT>((const d2*)(a.elements + k));
                    vb[c] = ld_stream<NT>((const d2*)(a.elements + k + 2));
                }
            }
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const int o = c * (kBlock * 4) + 4 * tid;
                if (wide[c]) {
                    const double x0 = a.x[col[c].x], x1 = a.x[col[c].y], x2 = a.x[col[c].z], x3 = a.x[col[c].w];
                    d2 p0, p1;
                    p0.x = va[c].x * x0; p0.y = va[c].y * x1;
                    p1.x = vb[c].x * x2; p1.y = vb[c].y * x3;
                    *(d2*)(s_prod + o) = p0;
                    *(d2*)(s_prod + o + 2) = p1;
                } else if constexpr (ALIGNED) {
                    // last few nonzeros of the arrays (k+4 would overrun elementsCount): guarded scalars
                    const int k = tb + o;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (k + j < e) s_prod[o + j] = a.elements[k + j] * a.x[a.columnIndeces[k + j]];
                } else {
                    // base pointers not 16-byte aligned (a caller-offset sub-array): lane-contiguous scalars
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int oo = c * (kBlock * 4) + j * kBlock + tid;
                        const int kk = tb + oo;
                        if (kk >= s && kk < e) s_prod[oo] = a.elements[kk] * a.x[a.columnIndeces[kk]];
                    }
                }
            }
            __syncthreads();
            // ---- reduce phase: one lane per row, stored order ----
            const int lo = my_s > tb ? my_s : tb;
            const int hi = my_e < tb + CAP ? my_e : tb + CAP;
            for (int j = lo; j < hi; ++j) acc += s_prod[j - tb];
        }
        if (tid < nr) spmv_epilogue<EPI>(a, r0 + tid, acc, dotacc);
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_RESIDUAL_DOT) {
        __syncthreads();
        const double t = block_sum_256(dotacc, s_red);
        if (tid == 0) a.partials[blockIdx.x] = t;
    }
}

// LANES lanes per row (power of two, 2..64).
template <int EPI, int LANES>
__global__ __launch_bounds__(kBlock) void spmv_vector_kernel(SpmvArgs a)
{
    __shared__ double s_red[4];
    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;
    constexpr int ROWS = kBlock / LANES;
    const int tid = threadIdx.x;
    const int sub = tid % LANES;
    double dotacc = 0.0;
    for (long long base = (long long)blockIdx.x * ROWS; base < a.rowCount; base += (long long)gridDim.x * ROWS) {
        const long long row = base + tid / LANES;
        double acc = 0.0;
        if (row < a.rowCount) {
            const int s = a.rowOffsets[row], e = a.rowOffsets[row + 1];
            for (int k = s + sub; k < e; k += LANES) {
                double prod = a.elements[k] * a.x[a.columnIndeces[k]];
                acc += prod;
            }
        }
#pragma unroll
        for (int off = LANES / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LANES);
        if (sub == 0 && row < a.rowCount) spmv_epilogue<EPI>(a, row, acc, dotacc);
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_RESIDUAL_DOT) {
        const double t = block_sum_256(dotacc, s_red);
        if (tid == 0) a.partials[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ launch plumbing

template <int EPI, int R, int CH>
static void launch_stream_rc(hipStream_t s, const SpmvArgs& a, int flags, int grid, int nRowBlocks, bool aligned)
{
    const bool nt = flags & 1, xcd = flags & 2;
#define MGCG_GO(NT_, XCD_, AL_) hipLaunchKernelGGL((spmv_stream_kernel<EPI, R, CH, NT_, XCD_, AL_>), dim3(grid), dim3(kBlock), 0, s, a, nRowBlocks)
    if (!aligned) { if (xcd) MGCG_GO(false, true, false); else MGCG_GO(false, false, false); }
    else if (nt) { if (xcd) MGCG_GO(true, true, true); else MGCG_GO(true, false, true); }
    else { if (xcd) MGCG_GO(false, true, true); else MGCG_GO(false, false, true); }
#undef MGCG_GO
}

template <int EPI>
static int launch_stream(hipStream_t s, const SpmvArgs& a, const SpmvConfig& cfg)
{
    int R = cfg.rowsPerBlock;
    if (R != 64 && R != 128 && R != 256) R = 256;
    const int nRowBlocks = (int)(((long long)a.rowCount + R - 1) / R);
    int flags = cfg.flags;
    int grid = cfg.gridBlocks > 0 ? cfg.gridBlocks : kMaxGrid;
    if (grid > nRowBlocks) grid = nRowBlocks;
    if (grid < 1) grid = 1;
    if (flags & 2) {                       // XCD mapping needs a multiple of 8 workgroups and enough row blocks
        if (nRowBlocks < 8 * kNumXcd) flags &= ~2;
        else grid = (grid / kNumXcd) * kNumXcd;
    }
    const bool aligned = (((uintptr_t)a.elements & 15) == 0) && (((uintptr_t)a.columnIndeces & 15) == 0);
    // pass capacity: 2 chunks (2048 nnz) covers R=256 rows of a 7-point stencil in one pass
    switch (R) {
    case 64:  launch_stream_rc<EPI, 64, 1>(s, a, flags, grid, nRowBlocks, aligned); break;
    case 128: launch_stream_rc<EPI, 128, 1>(s, a, flags, grid, nRowBlocks, aligned); break;
    default:  launch_stream_rc<EPI, 256, 2>(s, a, flags, grid, nRowBlocks, aligned); break;
    }
    return grid;
}

template <int EPI, int LANES>
static int launch_vector_l(hipStream_t s, const SpmvArgs& a, const SpmvConfig& cfg)
{
    constexpr int ROWS = kBlock / LANES;
    long long blocks = ((long long)a.rowCount + ROWS - 1) / ROWS;
    int grid = cfg.gridBlocks > 0 ? cfg.gridBlocks : kMaxGrid;
    if (grid > blocks) grid = (int)blocks;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL((spmv_vector_kernel<EPI, LANES>), dim3(grid), dim3(kBlock), 0, s, a);
    return grid;
}

template <int EPI>
static int launch_spmv_epi(hipStream_t s, const SpmvArgs& a, const SpmvConfig& cfg)
{
    int kernel = cfg.kernel;
    if (kernel == 0) {
        const double avg = a.rowCount > 0 ? (double)a.elementsCount / (double)a.rowCount : 0.0;
        if (avg <= 24.0) kernel = 1;
        else if (avg <= 48.0) kernel = 6;   // 16 lanes per row
        else if (avg <= 96.0) kernel = 7;   // 32 lanes per row
        else kernel = 8;                    // one wavefront per row
    }
    switch (kernel) {
    case 1: return launch_stream<EPI>(s, a, cfg);
    case 3: return launch_vector_l<EPI, 2>(s, a, cfg);
    case 4: return launch_vector_l<EPI, 4>(s, a, cfg);
    case 5: return launch_vector_l<EPI, 8>(s, a, cfg);
    case 6: return launch_vector_l<EPI, 16>(s, a, cfg);
    case 7: return launch_vector_l<EPI, 32>(s, a, cfg);
    default: return launch_vector_l<EPI, 64>(s, a, cfg);
    }
}

int launch_spmv(hipStream_t s, int epilogue, const SpmvArgs& a, const SpmvConfig& cfg)
{
    if (a.rowCount <= 0) return 0;
    switch (epilogue) {
    case EPI_AXPBY:        return launch_spmv_epi<EPI_AXPBY>(s, a, cfg);
    case EPI_DOT:          return launch_spmv_epi<EPI_DOT>(s, a, cfg);
    case EPI_RESIDUAL:     return launch_spmv_epi<EPI_RESIDUAL>(s, a, cfg);
    case EPI_RESIDUAL_DOT: return launch_spmv_epi<EPI_RESIDUAL_DOT>(s, a, cfg);
    case EPI_JACOBI:       return launch_spmv_epi<EPI_JACOBI>(s, a, cfg);
    }
    return 0;
}

} // namespace mgcg
