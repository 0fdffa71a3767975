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
typedef int    i2 __attribute__((ext_vector_type(2)));

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

// Epilogue operands of one row, loaded early (with the gathers) so the epilogue itself issues no load.
struct EpiOperands { double w, b, dinv, yold; };

template <int EPI>
__device__ __forceinline__ EpiOperands epi_prefetch(const SpmvArgs& a, long long row)
{
    EpiOperands o; o.w = 0.0; o.b = 0.0; o.dinv = 0.0; o.yold = 0.0;
    if constexpr (EPI == EPI_AXPBY_BETA) o.yold = a.y[row];
    if constexpr (EPI == EPI_DOT) o.w = a.w[row];
    if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_DOT) o.b = a.b[row];
    if constexpr (EPI == EPI_JACOBI || EPI == EPI_JACOBI_DOT) { o.b = a.b[row]; o.dinv = a.dinvUniform ? a.dinvScalar : a.dinv[row]; o.w = a.w[row]; }
    return o;
}

// The value the epilogue stores to y[row] (and the row's contribution to the fused dot product).
template <int EPI>
__device__ __forceinline__ double spmv_epilogue_value(const SpmvArgs& a, double acc, const EpiOperands& o, double& dotacc)
{
    if constexpr (EPI == EPI_AXPBY) {
        return a.alpha * acc;
    } else if constexpr (EPI == EPI_AXPBY_BETA) {
        double v = a.alpha * acc;
        double t = a.beta * o.yold;
        return v + t;
    } else if constexpr (EPI == EPI_DOT) {
        double t = o.w * acc;
        dotacc += t;
        return acc;
    } else if constexpr (EPI == EPI_RESIDUAL) {
        return o.b - acc;
    } else if constexpr (EPI == EPI_RESIDUAL_DOT) {
        double r = o.b - acc;
        double t = r * r;
        dotacc += t;
        return r;
    } else {   // EPI_JACOBI, EPI_JACOBI_DOT
        double res = o.b - acc;
        double t = o.dinv * res;
        double s = a.omega * t;
        const double v = o.w + s;
        if constexpr (EPI == EPI_JACOBI_DOT) { double q = o.b * v; dotacc += q; }
        return v;
    }
}

template <int EPI>
__device__ __forceinline__ void spmv_epilogue(const SpmvArgs& a, long long row, double acc, const EpiOperands& o, double& dotacc)
{
    a.y[row] = spmv_epilogue_value<EPI>(a, acc, o, dotacc);
}

// One pipeline stage: the first-pass loads of a row block (CH 16-byte chunks of column ids and values
// per lane), the lane's own row bounds and the block's nonzero span.
// Layout of a chunk (1024 nonzeros): two halves of 512; in half h lane t owns nonzeros 512*h + 2*t, +1.
// Every wave-instruction then reads whole, distinct 128-byte lines (64 lanes x 16 B of values = 8 lines,
// x 8 B of column ids = 4 lines), which is what lets the stream use non-temporal loads: no line is
// touched by two instructions.
template <int CH>
struct StreamStage {
    i2 col[CH][2];
    d2 val[CH][2];
    int my_s, my_e;     // nonzero range of this lane's row
    int s, e;           // nonzero span of the whole row block (wave-uniform)
};
constexpr int kSpanAlign = 32;   // row-block spans are read from a 32-nonzero boundary (128 B of column ids, 256 B of values)

// In-order sum of s_prod[lo-tb .. hi-tb): LDS reads are issued eight at a time, the adds stay strictly
// left to right (adding +0.0 for the masked tail is exact: the accumulator is never -0.0).
template <int CAP>
__device__ __forceinline__ double reduce_row(const double* s_prod, int lo, int hi, int tb, double acc)
{
    for (int j = lo; j < hi; j += 8) {
        double v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            int idx = j + q - tb;
            idx = idx < CAP - 1 ? idx : CAP - 1;
            v[q] = s_prod[idx];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) acc += (j + q < hi) ? v[q] : 0.0;
    }
    return acc;
}

// BLOCK threads per workgroup (256, or 64: a single wavefront, whose barriers cost nothing), R rows per
// workgroup, CH chunks of 4 nonzeros per lane per pass (pass capacity CAP = 4*BLOCK*CH nonzeros).
// Software pipeline per workgroup, one row block per trip:
//   gathers of x + epilogue operands of block i  ->  wide loads of block i+1  ->  products of i to LDS
//   ->  barrier  ->  one lane per row adds its products in stored order  ->  store  ->  barrier
// The matrix stream of the next block is in flight while this block is multiplied and reduced.  vmcnt
// counts in issue order, so the gathers are issued BEFORE the prefetch and every load in the trip's
// head is unconditional (clamped addresses instead of branches): the compiler can then wait for the
// gathers with a counted vmcnt that leaves the prefetch in flight.  Row-block spans are prefetched two
// trips ahead with scalar loads.
// MAP selects the row-block schedule of the grid-stride loop:
//   0  plain: workgroup b takes row blocks b, b+grid, ...
//   1  each XCD (workgroups b, b+8, ... share one) takes one contiguous eighth of the matrix
//   2  banded: the matrix couples row i with rows i +- P (P = periodRb row blocks: a grid plane of a 3-D
//      stencil).  Inside every window of P row blocks XCD k owns the contiguous eighth k; its resident
//      workgroups cover tileRb neighbouring row blocks of zPar consecutive windows at a time, so the
//      +-1 / +-line neighbours of x and the +-plane neighbours are all in flight in the same XCD's L2
//      together (x is fetched ~(zPar+2)/zPar * (lines+2)/lines times instead of 5+ times).
template <int EPI, int BLOCK, int R, int CH, bool NT, int MAP, bool ALIGNED>
__global__ __launch_bounds__(BLOCK) void spmv_stream_kernel(SpmvArgs a, int nRowBlocks, int periodRb, int tileRb, int zPar)
{
    constexpr int CAP = BLOCK * 4 * CH;
    static_assert(R <= BLOCK, "one lane per row in the reduce phase");
    __shared__ double s_prod[CAP];
    __shared__ double s_red[BLOCK / 64];

    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;

    const int tid = threadIdx.x;
    // trip t of this workgroup handles row block rbBase + (t / tripsInner) * outerStep + (t % tripsInner) * innerStep
    long long rbBase, innerStep, outerStep;
    int tripsInner, nTrips;
    if constexpr (MAP == 2) {
        // XCD k owns the contiguous eighth k of every window (grid plane).  Its workgroups form a tile of
        // tileRb neighbouring row blocks x zPar consecutive windows that advances zPar windows per trip;
        // when the sweep over the windows ends the tile moves on inside the XCD's eighth.
        // Host guarantees: gridDim.x == 8*tileRb*zPar, (periodRb/8) % tileRb == 0, windows % zPar == 0.
        const int xcd = blockIdx.x & (kNumXcd - 1);
        const int loc = blockIdx.x >> 3;
        const int eighth = periodRb >> 3;
        const int windows = nRowBlocks / periodRb;
        const int l = loc % tileRb, q = loc / tileRb;
        rbBase = (long long)q * periodRb + (long long)xcd * eighth + l;
        innerStep = (long long)zPar * periodRb; outerStep = tileRb;
        tripsInner = windows / zPar;
        nTrips = (eighth / tileRb) * tripsInner;
    } else if constexpr (MAP == 1) {
        const int xcd = blockIdx.x & (kNumXcd - 1);
        const int local = blockIdx.x >> 3;
        const int perXcd = gridDim.x >> 3;
        const long long b = (long long)nRowBlocks * xcd / kNumXcd + local;
        const long long e = (long long)nRowBlocks * (xcd + 1) / kNumXcd;
        rbBase = b; innerStep = perXcd; outerStep = 0;
        nTrips = (e > b) ? (int)((e - b + perXcd - 1) / perXcd) : 0;
        tripsInner = nTrips > 0 ? nTrips : 1;
    } else {
        rbBase = blockIdx.x; innerStep = gridDim.x; outerStep = 0;
        nTrips = ((long long)nRowBlocks > (long long)blockIdx.x) ? (int)(((long long)nRowBlocks - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
        tripsInner = nTrips > 0 ? nTrips : 1;
    }
    auto rb_of = [&](int t) -> long long {
        return rbBase + (long long)(t / tripsInner) * outerStep + (long long)(t % tripsInner) * innerStep;
    };
    const long long lastRow = (long long)a.rowCount - 1;
    // last aligned pair that lies wholly inside the arrays (host guarantees elementsCount >= 2 when ALIGNED)
    const int kMaxWide = ALIGNED ? ((a.elementsCount - 2) & ~1) : 0;

    auto span_of = [&](long long rb, int& s, int& e) {
        const long long r0 = rb * R;
        const long long r1 = (r0 + R < (long long)a.rowCount) ? r0 + R : (long long)a.rowCount;
        s = a.rowOffsets[r0];
        e = a.rowOffsets[r1];
    };
    // Issue the first-pass loads of row block rb (st.s / st.e known).  Branch-free: out-of-range lanes
    // load a clamped, valid address and ignore the result.
    auto issue = [&](StreamStage<CH>& st, long long rb) {
        const long long r0 = rb * R;
        long long row = r0 + tid;
        const bool live = (tid < R) && (row <= lastRow);
        row = row <= lastRow ? row : lastRow;
        const int ms = a.rowOffsets[row], me = a.rowOffsets[row + 1];
        st.my_s = live ? ms : st.e;
        st.my_e = live ? me : st.e;
        if constexpr (ALIGNED) {
            const int tb0 = st.s & ~(kSpanAlign - 1);
#pragma unroll
            for (int c = 0; c < CH; ++c) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    int k = tb0 + c * (BLOCK * 4) + h * (BLOCK * 2) + 2 * tid;
                    k = k < kMaxWide ? k : kMaxWide;
                    st.col[c][h] = ld_stream<NT>((const i2*)(a.columnIndeces + k));
                    st.val[c][h] = ld_stream<NT>((const d2*)(a.elements + k));
                }
            }
        }
    };

    double dotacc = 0.0;
    if (nTrips > 0) {
        StreamStage<CH> cur, nxt;
        span_of(rb_of(0), cur.s, cur.e);
        issue(cur, rb_of(0));
        nxt.s = cur.s; nxt.e = cur.e;
        if (nTrips > 1) span_of(rb_of(1), nxt.s, nxt.e);

        // The y store of a trip is issued in the NEXT trip, behind that trip's loads: vmcnt retires in issue
        // order and counts stores, so a store issued before the loads would have to be acknowledged by
        // memory before the gathers behind it can be waited for.
        double pendVal = 0.0;
        long long pendRow = -1;
        for (int t = 0; t < nTrips; ++t) {
            const long long rb = rb_of(t);
            const long long r0 = rb * R;
            const long long left = (long long)a.rowCount - r0;
            const int nr = (int)(left < R ? left : R);
            int s2 = nxt.s, e2 = nxt.e;
            if (t + 2 < nTrips) span_of(rb_of(t + 2), s2, e2);       // scalar prefetch, two trips ahead

            const int s = cur.s, e = cur.e;
            const int tb0 = ALIGNED ? (s & ~(kSpanAlign - 1)) : s;
            // ---- head of the trip, all loads unconditional ----
            // (1) gathers for the current block; clamped chunks hold real matrix entries, so the column ids are valid
            double xg[CH][2][2];
            if constexpr (ALIGNED) {
#pragma unroll
                for (int c = 0; c < CH; ++c) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        int c0 = cur.col[c][h].x, c1 = cur.col[c][h].y;
                        if (MGCG_ABLATE(a, 2)) { c0 &= 1023; c1 &= 1023; }      // lab builds: gathers served from L1
                        xg[c][h][0] = a.x[c0]; xg[c][h][1] = a.x[c1];
                    }
                }
            }
            // (2) epilogue operands of this lane's row
            long long myRow = r0 + tid;
            myRow = myRow <= lastRow ? myRow : lastRow;
            const EpiOperands eo = epi_prefetch<EPI>(a, myRow);
            // (3) the next block's matrix stream goes in flight behind them (the last trip re-reads its own block)
            issue(nxt, t + 1 < nTrips ? rb_of(t + 1) : rb);
            // (4) the previous trip's result
            if (pendRow >= 0 && !MGCG_ABLATE(a, 1)) a.y[MGCG_ABLATE(a, 8) ? (pendRow & 0xFFFF) : pendRow] = pendVal;   // (lab builds, bit3: stores stay in L2)
            // ---- products of the current block to LDS ----
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (ALIGNED) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int o = c * (BLOCK * 4) + h * (BLOCK * 2) + 2 * tid;
                        const int k = tb0 + o;
                        if (k <= kMaxWide) {
                            d2 p;
                            p.x = cur.val[c][h].x * xg[c][h][0]; p.y = cur.val[c][h].y * xg[c][h][1];
                            *(d2*)(s_prod + o) = p;
                        } else {
                            // last nonzero of an odd-length array: guarded scalars
#pragma unroll
                            for (int j = 0; j < 2; ++j)
                                if (k + j < e) s_prod[o + j] = a.elements[k + j] * a.x[a.columnIndeces[k + j]];
                        }
                    }
                } else {
                    // base pointers not 16-byte aligned (a caller-offset sub-array): lane-contiguous scalars
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int oo = c * (BLOCK * 4) + j * BLOCK + tid;
                        const int kk = tb0 + oo;
                        if (kk < e) s_prod[oo] = a.elements[kk] * a.x[a.columnIndeces[kk]];
                    }
                }
            }
            __syncthreads();
            // ---- reduce: one lane per row, stored order ----
            double acc = 0.0;
            {
                const int lo = cur.my_s > tb0 ? cur.my_s : tb0;
                const int hi = cur.my_e < tb0 + CAP ? cur.my_e : tb0 + CAP;
                acc = reduce_row<CAP>(s_prod, lo, hi, tb0, acc);
            }
            // ---- further passes for row blocks longer than CAP (long rows): not pipelined ----
            for (int tb = tb0 + CAP; tb < e; tb += CAP) {
                __syncthreads();
#pragma unroll
                for (int c = 0; c < CH; ++c) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int oo = c * (BLOCK * 4) + j * BLOCK + tid;
                        const int kk = tb + oo;
                        if (kk < e) s_prod[oo] = a.elements[kk] * a.x[a.columnIndeces[kk]];
                    }
                }
                __syncthreads();
                const int lo = cur.my_s > tb ? cur.my_s : tb;
                const int hi = cur.my_e < tb + CAP ? cur.my_e : tb + CAP;
                acc = reduce_row<CAP>(s_prod, lo, hi, tb, acc);
            }
            pendRow = -1;
            if (tid < nr) { pendVal = spmv_epilogue_value<EPI>(a, acc, eo, dotacc); pendRow = r0 + tid; }
            if (MGCG_ABLATE(a, 1) && acc == 1.2345e300) pendRow = 0;     // lab builds (no y store): keep acc alive
            __syncthreads();                                          // LDS is free for the next trip
            cur = nxt;
            nxt.s = s2; nxt.e = e2;
        }
        if (pendRow >= 0 && !MGCG_ABLATE(a, 1)) a.y[pendRow] = pendVal;
    }
    if constexpr (epi_has_dot(EPI)) {
        double t = wave_sum(dotacc);
        if constexpr (BLOCK > 64) {
            if ((tid & 63) == 0) s_red[tid >> 6] = t;
            __syncthreads();
            t = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        }
        if (tid == 0) a.partials[blockIdx.x] = t;
    }
}

// LANES lanes per row (power of two, 2..64).  VNT: the values are read with the non-temporal hint and the column ids without it -- for a matrix
// whose 12 bytes per nonzero exceed the 256 MB Infinity Cache while its column ids alone fit (the reference drivers' 200-350 k rows x 159:
// 130-220 MB of column ids): the ids then stay cache-resident from one product of a solve to the next and only the values come from HBM.
template <int EPI, int LANES, bool VNT = false>
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
            int k = s + sub;
            // four strides of the row in flight per lane (values, column ids, then the four gathers), added in the same order
            // as the rolled loop
            for (; k + 3 * LANES < e; k += 4 * LANES) {
                double v[4]; int c[4]; double xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v[u] = ld_stream<VNT>(a.elements + k + u * LANES); c[u] = a.columnIndeces[k + u * LANES]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) xv[u] = a.x[c[u]];
#pragma unroll
                for (int u = 0; u < 4; ++u) { const double prod = v[u] * xv[u]; acc += prod; }
            }
            for (; k < e; k += LANES) {
                double prod = ld_stream<VNT>(a.elements + k) * a.x[a.columnIndeces[k]];
                acc += prod;
            }
        }
#pragma unroll
        for (int off = LANES / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, LANES);
        if (sub == 0 && row < a.rowCount) { const EpiOperands eo = epi_prefetch<EPI>(a, row); spmv_epilogue<EPI>(a, row, acc, eo, dotacc); }
    }
    if constexpr (epi_has_dot(EPI)) {
        const double t = block_sum_256(dotacc, s_red);
        if (tid == 0) a.partials[blockIdx.x] = t;
    }
}

// ------------------------------------------------------------------ launch plumbing

// Resident workgroups per CU for a kernel (occupancy API, cached per instantiation); the grid of a
// grid-stride kernel is sized to exactly fill the chip so no workgroup waits for a slot.
template <typename K>
static int resident_blocks_per_cu(K kernel, int block)
{
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, block, 0) != hipSuccess || n < 1) n = 4;
    // 32 waves per CU at most; single-wavefront workgroups measured fastest at 16 per CU on the 7-point
    // 512^3 matrix (more resident waves only add cache pressure: tools/spmv_sweep.py, profiles/)
    const int cap = block == 64 ? 16 : 2048 / block;
    if (n > cap) n = cap;
    return n;
}

static int device_cu_count()
{
    DeviceState* d = device_state();
    return d ? d->numCu : kNumCu;
}

template <int EPI, int BLOCK, int R, int CH, bool NT, int MAP, bool AL>
static int launch_stream_inst(hipStream_t s, const SpmvArgs& a, const SpmvConfig& cfg, int nRowBlocks, int periodRb)
{
    static const int perCu = resident_blocks_per_cu(spmv_stream_kernel<EPI, BLOCK, R, CH, NT, MAP, AL>, BLOCK);
    int grid = cfg.gridBlocks > 0 ? cfg.gridBlocks : perCu * device_cu_count();
    if (grid > kMaxPartials) grid = kMaxPartials;
    int tileRb = 0, zPar = 0;
    if (MAP == 2) {
        // choose the tile: tileRb row blocks (a power-of-two fraction of the XCD's eighth) x zPar windows = perXcd workgroups
        const int eighth = periodRb / kNumXcd, windows = nRowBlocks / periodRb;
        int perXcd = grid / kNumXcd;
        if (perXcd < 1) perXcd = 1;
        tileRb = cfg.tileRows > 0 ? cfg.tileRows / R : 0;
        if (tileRb < 1 || eighth % tileRb != 0 || tileRb > perXcd) {
            tileRb = eighth;
            while (tileRb > perXcd && tileRb % 2 == 0) tileRb /= 2;    // default: the widest tile, one window at a time
        }
        zPar = cfg.tilePlanes > 0 ? cfg.tilePlanes : perXcd / (tileRb > 0 ? tileRb : 1);
        if (zPar < 1) zPar = 1;
        if (zPar > windows) zPar = windows;
        while (zPar > 1 && (windows % zPar != 0 || tileRb * zPar > perXcd)) --zPar;
        if (tileRb > perXcd) { tileRb = 0; }                           // cannot tile: handled by the caller's fallback
        grid = kNumXcd * tileRb * zPar;
    } else {
        if (grid > nRowBlocks) grid = nRowBlocks;
        if (MAP == 1) grid = (grid / kNumXcd) * kNumXcd;
    }
    if (grid < 1) return -1;
    hipLaunchKernelGGL((spmv_stream_kernel<EPI, BLOCK, R, CH, NT, MAP, AL>), dim3(grid), dim3(BLOCK), 0, s, a, nRowBlocks, periodRb, tileRb, zPar);
    return grid;
}

template <int EPI, int BLOCK, int R, int CH>
static int launch_stream_rc(hipStream_t s, const SpmvArgs& a, int flags, const SpmvConfig& cfg, int nRowBlocks, int periodRb, bool aligned)
{
    const bool nt = flags & 1;
    const int map = (flags & 4) ? 2 : ((flags & 2) ? 1 : 0);
    if (map == 2) {     // banded schedule; if the tile cannot be formed fall through to the plain schedule
        int g = -1;
        if (!aligned) g = launch_stream_inst<EPI, BLOCK, R, CH, false, 2, false>(s, a, cfg, nRowBlocks, periodRb);
        else if (nt) g = launch_stream_inst<EPI, BLOCK, R, CH, true, 2, true>(s, a, cfg, nRowBlocks, periodRb);
        else g = launch_stream_inst<EPI, BLOCK, R, CH, false, 2, true>(s, a, cfg, nRowBlocks, periodRb);
        if (g > 0) return g;
    }
#define MGCG_GO(NT_, MAP_, AL_) return launch_stream_inst<EPI, BLOCK, R, CH, NT_, MAP_, AL_>(s, a, cfg, nRowBlocks, periodRb)
    if (!aligned) { if (map == 1) MGCG_GO(false, 1, false); MGCG_GO(false, 0, false); }
    if (nt) { if (map == 1) MGCG_GO(true, 1, true); MGCG_GO(true, 0, true); }
    if (map == 1) MGCG_GO(false, 1, true);
    MGCG_GO(false, 0, true);
#undef MGCG_GO
}

// rowsPerBlock selects the workgroup shape: 64 -> one wavefront per 64 rows (no barriers), 128 -> 256 threads
// per 128 rows, 256 -> 256 threads per 256 rows, 32 -> one wavefront per 32 rows.
template <int EPI>
static int launch_stream(hipStream_t s, const SpmvArgs& a, const SpmvConfig& cfg)
{
    int R = cfg.rowsPerBlock;
    if (R != 32 && R != 64 && R != 128 && R != 256) R = 64;
    const int nRowBlocks = (int)(((long long)a.rowCount + R - 1) / R);
    int flags = cfg.flags;
    if ((flags & 2) && nRowBlocks < 8 * kNumXcd) flags &= ~2;    // XCD mapping needs enough row blocks
    // banded schedule: the period must be whole row blocks, split evenly over the 8 XCDs and tile the matrix
    int periodRb = 0;
    if ((flags & 4) && cfg.periodRows > 0 && cfg.periodRows % (R * kNumXcd) == 0 && a.rowCount % R == 0) {
        periodRb = cfg.periodRows / R;
        if (nRowBlocks % periodRb != 0 || nRowBlocks / periodRb < 2) periodRb = 0;
    }
    if (periodRb == 0) flags &= ~4;
    // The wide path reads pairs from a 32-nonzero boundary of the span: needs 8-byte aligned column ids and
    // 16-byte aligned values at even nonzero indices.
    const bool aligned = (((uintptr_t)a.elements & 15) == 0) && (((uintptr_t)a.columnIndeces & 7) == 0) && a.elementsCount >= 2;
    // pass capacity 4*BLOCK*CH must cover R rows of a 7-point stencil plus <=31 nonzeros of alignment in one pass
    switch (R) {
    case 32:  return launch_stream_rc<EPI, 64, 32, 1>(s, a, flags, cfg, nRowBlocks, periodRb, aligned);     // 224+31 <= 256
    case 128: return launch_stream_rc<EPI, 256, 128, 1>(s, a, flags, cfg, nRowBlocks, periodRb, aligned);   // 896+31 <= 1024
    case 256: return launch_stream_rc<EPI, 256, 256, 2>(s, a, flags, cfg, nRowBlocks, periodRb, aligned);   // 1792+31 <= 2048
    default:  return launch_stream_rc<EPI, 64, 64, 2>(s, a, flags, cfg, nRowBlocks, periodRb, aligned);     // 448+31 <= 512
    }
}

template <int EPI, int LANES>
static int launch_vector_l(hipStream_t s, const SpmvArgs& a, const SpmvConfig& cfg)
{
    constexpr int ROWS = kBlock / LANES;
    long long blocks = ((long long)a.rowCount + ROWS - 1) / ROWS;
    int grid = cfg.gridBlocks > 0 ? cfg.gridBlocks : kMaxGrid;
    if (grid > blocks) grid = (int)blocks;
    if (grid < 1) grid = 1;
    // values past the Infinity Cache, column ids inside it (see the kernel)
    const long long idBytes = 4LL * a.elementsCount;
    const bool vnt = 3 * idBytes > (256LL << 20) && idBytes <= (224LL << 20);
    if (vnt) hipLaunchKernelGGL((spmv_vector_kernel<EPI, LANES, true>), dim3(grid), dim3(kBlock), 0, s, a);
    else hipLaunchKernelGGL((spmv_vector_kernel<EPI, LANES>), dim3(grid), dim3(kBlock), 0, s, a);
    return grid;
}

template <int EPI>
static int launch_spmv_epi(hipStream_t s, const SpmvArgs& a, const SpmvConfig& cfg)
{
    int kernel = cfg.kernel;
    if (kernel == 0) {
        const double avg = a.rowCount > 0 ? (double)a.elementsCount / (double)a.rowCount : 0.0;
        kernel = spmv_auto_kernel(avg);
        // gathers without locality (flag 16: the sampled entries lie beyond an L2's reach of the diagonal, x far larger than an L2): the
        // product is bound by the gathers, not by the row shape, and the row-block stream form -- bit-identical to the oracle, unlike the
        // lanes-per-row forms -- is the faster CSR kernel there (10 M-row random matrix: 5.77 ms against 6.08-6.31, profiles/r3/config5_*)
        if ((cfg.flags & 16) != 0 && kernel >= 5 && kernel <= 7 && avg <= 64.0) kernel = 1;
    }
    if (kernel == 10) {     // row-tile kernel (four wavefronts per tile of 256 rows); 16-byte aligned arrays
        const bool ok = (((uintptr_t)a.elements & 15) == 0) && (((uintptr_t)a.columnIndeces & 15) == 0) && a.elementsCount >= 8;
        if (ok) return launch_spmv_rowtile(s, EPI == EPI_AXPBY_BETA ? (int)EPI_AXPBY : (int)EPI, a, cfg.periodRows, cfg.gridBlocks, cfg.maxRow, (cfg.flags & 8) != 0);
        kernel = 9;
    }
    if (kernel == 9) {      // row-block kernel, "stage raw, multiply by row" form; same alignment needs as the stream kernel's wide path
        const bool ok = (((uintptr_t)a.elements & 15) == 0) && (((uintptr_t)a.columnIndeces & 7) == 0) && a.elementsCount >= 8;
        if (ok) return launch_spmv_rows(s, EPI == EPI_AXPBY_BETA ? (int)EPI_AXPBY : (int)EPI, a, nullptr, cfg.gridBlocks);
        kernel = 1;
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
    case EPI_AXPBY:        return a.beta != 0.0 ? launch_spmv_epi<EPI_AXPBY_BETA>(s, a, cfg) : launch_spmv_epi<EPI_AXPBY>(s, a, cfg);
    case EPI_DOT:          return launch_spmv_epi<EPI_DOT>(s, a, cfg);
    case EPI_RESIDUAL:     return launch_spmv_epi<EPI_RESIDUAL>(s, a, cfg);
    case EPI_RESIDUAL_DOT: return launch_spmv_epi<EPI_RESIDUAL_DOT>(s, a, cfg);
    case EPI_JACOBI:       return launch_spmv_epi<EPI_JACOBI>(s, a, cfg);
    case EPI_JACOBI_DOT:   return launch_spmv_epi<EPI_JACOBI_DOT>(s, a, cfg);
    }
    return 0;
}

void preload_kernels_spmv() { preload_code_object(reinterpret_cast<const void*>(&(spmv_vector_kernel<EPI_DOT, 32>))); }

} // namespace mgcg
