// Column-tiled form for matrices whose gathers have no locality (class 4 of the opt-in analysis; BASELINE config 5:
// random SPD, 10 M rows, ~31 nonzeros per row spread over the whole column range).
//
// On such a matrix the CSR kernels are bound by the gathers: x (80 MB) lives beyond the 4 MB L2 of an XCD, every 8-byte
// gather pulls a 128-byte line through the fabric (40 GB per product instead of 3.9 GB of algorithmic bytes,
// profiles/r1/config5_random_spd_10M.log).  Here the nonzeros are re-laid out ONCE by column tile (tile = a window of
// x that stays in L2), row-major inside a tile, each entry with its row id, and the product runs tile by tile: pass t
// multiplies tile t's entries ENTRY-parallel (coalesced streams of values, column ids and row ids; gathers only from the
// tile's x window), the first entry of every (row, tile) segment then adds the segment's products to y[row] one after the
// other.  For rows stored with ascending column ids -- the analysis requires it -- the tiles of a row are visited in
// stored order and the running sum continues where the previous tile stopped, so every row is summed in exactly the
// order of the CSR kernels: bit-identical results.  A last pass over the rows applies the epilogue.
// Cost: one more copy of the matrix (16 B/nnz) and y is read and written once per (row, tile) segment.
#include "common.hpp"
#include <algorithm>

namespace mgcg {

constexpr int kScanBlock = 256;
constexpr int kScanPer = 8;                    // elements per thread in the scan kernels

// ---------------------------------------------------------------- analysis
// One pass over the rows: are the column ids of every row ascending?  how far from the diagonal is the typical entry?
// and counts[t * rows + i] = entries of row i in column tile t.
__global__ __launch_bounds__(kBlock) void tiled_count_kernel(const int* __restrict__ rowOffsets, const int* __restrict__ columnIndeces,
                                                             long long rows, long long rowBase, int tileWidth, int nTiles,
                                                             int* __restrict__ counts, unsigned long long* stats /* [0] unsorted rows, [1] sum |col-row| >> 10 */)
{
    const long long stride = (long long)gridDim.x * kBlock;
    unsigned long long far = 0, unsorted = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        const int s = rowOffsets[i], e = rowOffsets[i + 1];
        int prev = -1, tile = -1, run = 0;
        for (int k = s; k < e; ++k) {
            const int c = columnIndeces[k];
            if (c <= prev) unsorted = 1;
            prev = c;
            const long long d = (long long)c - (rowBase + i);
            far += (unsigned long long)((d < 0 ? -d : d) >> 10);
            const int t = c / tileWidth;
            if (t != tile) { if (tile >= 0 && tile < nTiles) counts[(long long)tile * rows + i] = run; tile = t; run = 0; }
            ++run;
        }
        if (tile >= 0 && tile < nTiles) counts[(long long)tile * rows + i] = run;
    }
    if (far) atomicAdd(&stats[1], far);
    if (unsorted) atomicAdd(&stats[0], 1ull);
}

// exclusive scan of n ints in three steps (block-local scan, scan of the block totals, add-back); totals fit int (nnz < 2^31)
__global__ __launch_bounds__(kScanBlock) void scan_local_kernel(int* __restrict__ data, long long n, int* __restrict__ blockTotals)
{
    __shared__ int s_sum[kScanBlock];
    const long long base = ((long long)blockIdx.x * kScanBlock + threadIdx.x) * kScanPer;
    int v[kScanPer], total = 0;
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { v[j] = (base + j < n) ? data[base + j] : 0; total += v[j]; }
    s_sum[threadIdx.x] = total;
    __syncthreads();
    for (int off = 1; off < kScanBlock; off <<= 1) {                // Hillis-Steele over the thread totals
        const int add = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0;
        __syncthreads();
        s_sum[threadIdx.x] += add;
        __syncthreads();
    }
    int run = s_sum[threadIdx.x] - total;                            // exclusive prefix of this thread inside the block
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) { if (base + j < n) data[base + j] = run; run += v[j]; }
    if (threadIdx.x == kScanBlock - 1) blockTotals[blockIdx.x] = s_sum[kScanBlock - 1];
}
__global__ __launch_bounds__(kScanBlock) void scan_totals_kernel(int* __restrict__ blockTotals, int nBlocks)
{
    __shared__ int s_sum[kScanBlock];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < nBlocks; base += kScanBlock) {         // one workgroup walks the totals
        const int i = base + threadIdx.x;
        const int v = i < nBlocks ? blockTotals[i] : 0;
        s_sum[threadIdx.x] = v;
        __syncthreads();
        for (int off = 1; off < kScanBlock; off <<= 1) {
            const int add = threadIdx.x >= off ? s_sum[threadIdx.x - off] : 0;
            __syncthreads();
            s_sum[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nBlocks) blockTotals[i] = s_carry + s_sum[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == kScanBlock - 1) s_carry += s_sum[kScanBlock - 1];
        __syncthreads();
    }
}
__global__ __launch_bounds__(kScanBlock) void scan_add_kernel(int* __restrict__ data, long long n, const int* __restrict__ blockTotals)
{
    const long long base = ((long long)blockIdx.x * kScanBlock + threadIdx.x) * kScanPer;
    const int add = blockTotals[blockIdx.x];
#pragma unroll
    for (int j = 0; j < kScanPer; ++j) if (base + j < n) data[base + j] += add;
}

__global__ __launch_bounds__(kBlock) void tiled_zero_kernel(double* __restrict__ y, long long n, const int* __restrict__ doneFlag)
{
    if (doneFlag != nullptr && *doneFlag != 0) return;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) y[i] = 0.0;
}
static int grid_rows(long long n)
{
    const long long blocks = (n + kBlock - 1) / kBlock;
    return (int)(blocks < 1 ? 1 : (blocks > kMaxGrid ? kMaxGrid : blocks));
}

// entries of row i go to [ptr[t * rows + i], ...) of tile t in stored (= ascending column) order, each with its row id
__global__ __launch_bounds__(kBlock) void tiled_scatter_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets,
                                                               const int* __restrict__ columnIndeces, long long rows, int tileWidth,
                                                               const int* __restrict__ ptr, double* __restrict__ tVals, int* __restrict__ tCols, int* __restrict__ tRows)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        const int s = rowOffsets[i], e = rowOffsets[i + 1];
        int tile = -1, pos = 0;
        for (int k = s; k < e; ++k) {
            const int c = columnIndeces[k];
            const int t = c / tileWidth;
            if (t != tile) { tile = t; pos = ptr[(long long)t * rows + i]; }
            tVals[pos] = elements[k]; tCols[pos] = c; tRows[pos] = (int)i;
            ++pos;
        }
    }
}

// ---------------------------------------------------------------- SpMV: one launch per column tile, then the epilogue
// Entries [kBegin, kEnd) of one tile, kTileE entries per thread (entry e of thread t: blockBase + e * kBlock + t, so every
// load is coalesced and kTileE gathers per lane are in flight).  y holds the running row sums (zeroed before the first tile).
constexpr int kTileE = 4;

template <bool NT>
__global__ __launch_bounds__(kBlock) void spmv_tile_pass_kernel(const double* __restrict__ x, double* __restrict__ y,
                                                                const double* __restrict__ tVals, const int* __restrict__ tCols, const int* __restrict__ tRows,
                                                                int kBegin, int kEnd, const int* __restrict__ doneFlag)
{
    __shared__ double s_p[kBlock * kTileE];
    __shared__ int s_r[kBlock * kTileE];
    if (doneFlag != nullptr && *doneFlag != 0) return;
    const int blockBase = kBegin + (int)blockIdx.x * (kBlock * kTileE);
    const int blockCount = (kEnd - blockBase) < kBlock * kTileE ? (kEnd - blockBase) : kBlock * kTileE;
    double v[kTileE]; int c[kTileE], r[kTileE];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        const int k = blockBase + (j < blockCount ? j : 0);          // (blockCount >= 1: the grid covers the tile exactly)
        if (NT) { v[e] = __builtin_nontemporal_load(tVals + k); c[e] = __builtin_nontemporal_load(tCols + k); r[e] = __builtin_nontemporal_load(tRows + k); }
        else { v[e] = tVals[k]; c[e] = tCols[k]; r[e] = tRows[k]; }
    }
    double xv[kTileE];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) xv[e] = x[c[e]];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        if (j < blockCount) { s_p[j] = v[e] * xv[e]; s_r[j] = r[e]; }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j0 = e * kBlock + (int)threadIdx.x;
        if (j0 >= blockCount) continue;
        const int row = r[e];
        const int prevRow = j0 > 0 ? s_r[j0 - 1] : (blockBase > kBegin ? tRows[blockBase - 1] : -1);
        if (prevRow == row) continue;                               // not the first entry of its (row, tile) segment
        double acc = y[row];
        int j = j0;
        while (j < blockCount && s_r[j] == row) { acc += s_p[j]; ++j; }
        if (j == blockCount) {                                      // the segment runs on into the next workgroup's entries
            for (int kk = blockBase + blockCount; kk < kEnd && tRows[kk] == row; ++kk) { const double q = tVals[kk] * x[tCols[kk]]; acc += q; }
        }
        y[row] = acc;
    }
}

// The same pass on 12-byte entries (value + one packed word: column offset inside the tile | row - base row of the entry's block of
// kTilePack entries; hdr[b] = { base row, row of the entry in front of the block or -1 }).  What conjugategradient_amd/tools/tile_lab.hip
// measured on a 10 M-row matrix of config 5's shape (profiles/r3/tile_lab_*.log): the three kinds of traffic of the 16-byte pass do not
// overlap -- entry streams 0.94 ms + gathers beyond L1 1.6 ms + y read-modify-write 0.6 ms = the 3.2 ms of the whole -- so the bytes of
// the entry stream are time: 12-byte entries with non-temporal loads 2.57 ms (16-byte: 3.23; plain loads 2.71; 8 / 16 entries per thread
// 2.66 / 3.29; y requested before the gathers 2.70; a persistent loop with the next block prefetched 3.09; tiles of 2^18 / 2^20 columns
// 2.66 / 3.59).  Bit-identical to the 16-byte pass and to the CSR kernels.
constexpr int kTilePack = kBlock * kTileE;     // entries per block header

__global__ __launch_bounds__(kBlock) void spmv_tile_pass_packed_kernel(const double* __restrict__ x, double* __restrict__ y,
                                                                       const double* __restrict__ tVals, const unsigned* __restrict__ tPacked, const int2* __restrict__ hdr,
                                                                       int hdrBase, int kBegin, int kEnd, int tileCol0, int shift, const int* __restrict__ doneFlag)
{
    __shared__ double s_p[kTilePack];
    __shared__ int s_r[kTilePack + 1];                              // s_r[j + 1] = row of entry j of the block, s_r[0] = row of the entry in front of it
    if (doneFlag != nullptr && *doneFlag != 0) return;
    const int blockBase = kBegin + (int)blockIdx.x * kTilePack;
    const int blockCount = (kEnd - blockBase) < kTilePack ? (kEnd - blockBase) : kTilePack;
    const int2 h = hdr[hdrBase + blockIdx.x];
    const unsigned colMask = (1u << shift) - 1u;
    double v[kTileE]; unsigned pk[kTileE]; int r[kTileE];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        const int k = blockBase + (j < blockCount ? j : 0);
        v[e] = __builtin_nontemporal_load(tVals + k); pk[e] = __builtin_nontemporal_load(tPacked + k);
    }
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        r[e] = h.x + (int)(pk[e] >> shift);
        s_r[j + 1] = j < blockCount ? r[e] : -2;
    }
    if (threadIdx.x == 0) s_r[0] = h.y;
    __syncthreads();                                                // (the segment leaders are found BEFORE the gathers: 2.59 against 2.68 ms per product with
    bool lead[kTileE];                                              //  one barrier and the test behind the products -- tools/tile_lab.hip, V1 / V1b)
#pragma unroll
    for (int e = 0; e < kTileE; ++e) { const int j = e * kBlock + (int)threadIdx.x; lead[e] = j < blockCount && s_r[j] != r[e]; }
    double xv[kTileE];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) xv[e] = x[tileCol0 + (int)(pk[e] & colMask)];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) { const int j = e * kBlock + (int)threadIdx.x; s_p[j] = v[e] * xv[e]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j0 = e * kBlock + (int)threadIdx.x;
        if (!lead[e]) continue;                                     // not the first entry of its (row, tile) segment
        const int row = r[e];
        double acc = y[row];
        int j = j0;
        while (j < blockCount && s_r[j + 1] == row) { acc += s_p[j]; ++j; }
        if (j == blockCount) {                                      // the segment runs on into the next block's entries
            for (int kk = blockBase + blockCount; kk < kEnd; ++kk) {
                const int2 hn = hdr[hdrBase + (kk - kBegin) / kTilePack];
                const unsigned p = tPacked[kk];
                if (hn.x + (int)(p >> shift) != row) break;
                const double q = tVals[kk] * x[tileCol0 + (int)(p & colMask)]; acc += q;
            }
        }
        y[row] = acc;
    }
}

// hdr and packed words of one tile's entries [kBegin, kEnd); *overflow is raised when a row id does not fit the packed word
__global__ __launch_bounds__(kBlock) void tiled_pack_kernel(const int* __restrict__ tCols, const int* __restrict__ tRows, int kBegin, int kEnd, int shift, int tileCol0,
                                                            unsigned* __restrict__ tPacked, int2* __restrict__ hdr, int hdrBase, int* __restrict__ overflow)
{
    const int nBlocks = (kEnd - kBegin + kTilePack - 1) / kTilePack;
    const unsigned colMask = (1u << shift) - 1u;
    const int rowLimit = (shift >= 31) ? 1 : (int)((1u << (32 - shift)) - 1u);
    for (int b = blockIdx.x; b < nBlocks; b += gridDim.x) {
        const int base = kBegin + b * kTilePack;
        const int end = base + kTilePack < kEnd ? base + kTilePack : kEnd;
        const int baseRow = tRows[base];
        if (threadIdx.x == 0) { int2 h; h.x = baseRow; h.y = base > kBegin ? tRows[base - 1] : -1; hdr[hdrBase + b] = h; }
        for (int k = base + (int)threadIdx.x; k < end; k += kBlock) {
            const int lr = tRows[k] - baseRow;
            if (lr < 0 || lr > rowLimit) { *overflow = 1; continue; }
            tPacked[k] = ((unsigned)lr << shift) | ((unsigned)(tCols[k] - tileCol0) & colMask);
        }
    }
}

// y[i] = epilogue(row sum y[i])  (+ partial sums of the fused dot product)
template <int EPI>
__global__ __launch_bounds__(kBlock) void tiled_epilogue_kernel(SpmvArgs a)
{
    __shared__ double s_red[4];
    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;
    double dotacc = 0.0;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < a.rowCount; i += stride) {
        const double acc = a.y[i];
        double val;
        if constexpr (EPI == EPI_AXPBY) val = a.alpha * acc;
        else if constexpr (EPI == EPI_DOT) { const double t = a.w[i] * acc; dotacc += t; val = acc; }
        else if constexpr (EPI == EPI_RESIDUAL) val = a.b[i] - acc;
        else if constexpr (EPI == EPI_RESIDUAL_DOT) { val = a.b[i] - acc; const double t = val * val; dotacc += t; }
        else { const double res = a.b[i] - acc; const double t = (a.dinvUniform ? a.dinvScalar : a.dinv[i]) * res; const double sft = a.omega * t; val = a.w[i] + sft;
               if constexpr (EPI == EPI_JACOBI_DOT) { const double q = a.b[i] * val; dotacc += q; } }
        if constexpr (EPI != EPI_DOT) a.y[i] = val;                  // (EPI_DOT leaves y = A x as it is)
    }
    if constexpr (epi_has_dot(EPI)) {
        double v = dotacc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
        __syncthreads();
        if (threadIdx.x == 0) a.partials[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    }
}

template <int EPI>
static int launch_tiled_epi(hipStream_t s, const SpmvArgs& a, const DcsrView& m)
{
    // the running sums start at +0.0 (0.0 + p == p exactly, the CSR kernels' start); skipped if the solve has stopped
    hipLaunchKernelGGL(tiled_zero_kernel, dim3(grid_rows(a.rowCount)), dim3(kBlock), 0, s, a.y, (long long)a.rowCount, a.doneFlag);
    for (int t = 0; t < m.nTiles; ++t) {
        const int kb = m.tileStartHost[t], ke = m.tileStartHost[t + 1];
        if (ke <= kb) continue;
        const dim3 g((ke - kb + kBlock * kTileE - 1) / (kBlock * kTileE));
        if (m.tilePacked != nullptr && m.tileHdrBaseHost[t] >= 0) {
            hipLaunchKernelGGL(spmv_tile_pass_packed_kernel, g, dim3(kBlock), 0, s, a.x, a.y, m.tileVals, m.tilePacked, m.tileHdr, m.tileHdrBaseHost[t], kb, ke,
                               (int)((long long)t * m.tileWidth), m.tileShift, a.doneFlag);
            continue;
        }
        hipLaunchKernelGGL(spmv_tile_pass_kernel<false>, g, dim3(kBlock), 0, s, a.x, a.y, m.tileVals, m.tileCols, m.tileRowIds, kb, ke, a.doneFlag);
    }
    if (EPI == EPI_AXPBY && a.alpha == 1.0) return 0;                // y = A x is already in place
    const int grid = grid_rows(a.rowCount);
    hipLaunchKernelGGL((tiled_epilogue_kernel<EPI>), dim3(grid), dim3(kBlock), 0, s, a);
    return grid;
}

// y must not be an input of the epilogue (EPI_AXPBY with beta != 0 is served by the CSR kernels)
int launch_spmv_tiled(hipStream_t s, int epilogue, const SpmvArgs& a, const DcsrView& m)
{
    if (a.rowCount <= 0) return 0;
    switch (epilogue) {
    case EPI_AXPBY:        return launch_tiled_epi<EPI_AXPBY>(s, a, m);
    case EPI_DOT:          return launch_tiled_epi<EPI_DOT>(s, a, m);
    case EPI_RESIDUAL:     return launch_tiled_epi<EPI_RESIDUAL>(s, a, m);
    case EPI_RESIDUAL_DOT: return launch_tiled_epi<EPI_RESIDUAL_DOT>(s, a, m);
    case EPI_JACOBI:       return launch_tiled_epi<EPI_JACOBI>(s, a, m);
    case EPI_JACOBI_DOT:   return launch_tiled_epi<EPI_JACOBI_DOT>(s, a, m);
    }
    return 0;
}

// ---------------------------------------------------------------- build
static bool exclusive_scan(hipStream_t s, int* data, long long n)
{
    const long long perBlock = (long long)kScanBlock * kScanPer;
    const long long nBlocks = (n + perBlock - 1) / perBlock;
    if (nBlocks > 0x7fffffffLL) { set_error("tiled analysis: too many scan blocks"); return false; }
    int* totals = nullptr;
    if (!MGCG_HIP(hipMalloc((void**)&totals, sizeof(int) * (size_t)nBlocks))) return false;
    hipLaunchKernelGGL(scan_local_kernel, dim3((unsigned)nBlocks), dim3(kScanBlock), 0, s, data, n, totals);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(kScanBlock), 0, s, totals, (int)nBlocks);
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nBlocks), dim3(kScanBlock), 0, s, data, n, totals);
    const bool ok = MGCG_HIP(hipGetLastError()) && MGCG_HIP(hipStreamSynchronize(s));
    (void)hipFree(totals);
    return ok;
}

// On success out->tileVals != nullptr says whether the tiled form exists (sorted rows, far-from-diagonal entries, an x
// that does not fit the L2).  columns = length of x.  Leaves the other fields of *out alone.
bool tiled_build(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                 long long rows, long long nnz, long long rowBase, long long columns, DcsrMatrix* out)
{
    // Width of a tile's x window.  Round 3 used 2^19 columns = 4 MiB, the whole of an XCD's 4 MiB L2: the y lines and the entry streams of a
    // pass then push window lines out again and every XCD fetches its window about nine times per pass (PMC, tools/tile_width_sweep.sh,
    // profiles/r4/tile_width/pmc_T20_T32_T40.json, pmc_T24_T27.json: 0.55 GB of L2-miss reads per pass where entries + y + window need 0.27; 12.6 GB per product = 1.8 x the
    // form's own bytes).  At 2.85 MiB the re-fetches all but vanish (9.7 GB per product = 1.2 x its own bytes, 27 passes instead of 20 on the
    // 10 M-column matrix) for the same time per product (2.68 against 2.66 ms: each pass costs its y sweep, ~21 us, whatever its width); below
    // that the additional passes cost more than the misses (2.77 / 2.95 / 3.36 ms at 2.4 / 1.9 / 1.4 MiB).  MGCG_TILE_SHIFT = s asks for 2^s columns.
    int tileShift = 19;                                            // bits of the packed word that hold the column offset inside a tile
    long long tileCols = 365LL * 1024;                             // 2.85 MiB of x
    { const int v = tuning().tileShift.load(std::memory_order_relaxed); if (v >= 8 && v <= 26) { tileShift = v; tileCols = 1LL << v; } }
    const int nTiles = (int)((columns + tileCols - 1) / tileCols);
    // equal-width tiles (not the last one narrow): tile t = columns [t * tileWidth, (t + 1) * tileWidth), tileWidth <= 2^tileShift -- every tile
    // then has the same share of a uniformly spread matrix, and the packed entries (below) fit in all of them
    const int tileWidth = nTiles > 0 ? (int)((columns + nTiles - 1) / nTiles) : (int)tileCols;
    if (rows <= 0 || nnz <= 0 || nTiles < 4 || nTiles > 256) return true;            // x fits a few L2s, or absurdly many passes
    const long long cells = (long long)nTiles * rows;
    {   // the analysis needs 4 B per (tile, row) cell and the tiled copy 16 B per nonzero: decline rather than exhaust the device
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return true; }
        const double need = 4.0 * (double)(cells + 1) + 16.0 * (double)nnz;
        if (cells + 1 > 0x7fffffffLL * 4LL || need > 0.5 * (double)freeB) return true;
    }
    int* counts = nullptr; unsigned long long* stats = nullptr;
    double* tv = nullptr; int* tc = nullptr; int* tr = nullptr;
    bool ok = MGCG_HIP(hipMalloc((void**)&counts, sizeof(int) * (size_t)(cells + 1))) && MGCG_HIP(hipMalloc((void**)&stats, 2 * sizeof(unsigned long long)));
    ok = ok && MGCG_HIP(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)(cells + 1), s)) && MGCG_HIP(hipMemsetAsync(stats, 0, 2 * sizeof(unsigned long long), s));
    auto fail = [&](bool hard) { if (counts) (void)hipFree(counts); if (stats) (void)hipFree(stats); if (tv) (void)hipFree(tv); if (tc) (void)hipFree(tc); if (tr) (void)hipFree(tr); return !hard; };
    if (!ok) return fail(true);
    long long blocks = (rows + kBlock - 1) / kBlock;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    hipLaunchKernelGGL(tiled_count_kernel, dim3((int)blocks), dim3(kBlock), 0, s, rowOffsets, columnIndeces, rows, rowBase, tileWidth, nTiles, counts, stats);
    unsigned long long hs[2] = { 0, 0 };
    ok = MGCG_HIP(hipMemcpyAsync(hs, stats, sizeof(hs), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
    if (!ok) return fail(true);
    const double meanDistance = (double)hs[1] * 1024.0 / (double)nnz;
    if (hs[0] != 0 || meanDistance < (double)tileWidth) return fail(false);            // unsorted rows, or gathers that are local anyway
    if (!exclusive_scan(s, counts, cells + 1)) return fail(true);
    ok = MGCG_HIP(hipMalloc((void**)&tv, sizeof(double) * (size_t)nnz)) && MGCG_HIP(hipMalloc((void**)&tc, sizeof(int) * (size_t)nnz)) &&
         MGCG_HIP(hipMalloc((void**)&tr, sizeof(int) * (size_t)nnz));
    if (!ok) return fail(true);
    hipLaunchKernelGGL(tiled_scatter_kernel, dim3((int)blocks), dim3(kBlock), 0, s, elements, rowOffsets, columnIndeces, rows, tileWidth, counts, tv, tc, tr);
    std::vector<int> starts((size_t)nTiles + 1, 0);                  // first entry of every tile = the flat scan at (tile, row 0)
    ok = MGCG_HIP(hipGetLastError());
    for (int t = 0; ok && t <= nTiles; ++t)
        ok = MGCG_HIP(hipMemcpyAsync(&starts[(size_t)t], counts + (long long)t * rows, sizeof(int), hipMemcpyDeviceToHost, s));
    ok = ok && MGCG_HIP(hipStreamSynchronize(s));
    if (!ok) return fail(true);
    if ((long long)starts[(size_t)nTiles] != nnz) { set_error("tiled analysis: %d of %lld nonzeros placed", starts[(size_t)nTiles], nnz); return fail(true); }
    (void)hipFree(stats); (void)hipFree(counts);
    out->tileVals = tv; out->tileCols = tc; out->tileRowIds = tr; out->nTiles = nTiles; out->tileRows = rows; out->tileStart = starts; out->tileShift = tileShift; out->tileWidth = tileWidth;
    // 12-byte entries: one packed word instead of column id + row id, when every row id fits next to the column offset (else the
    // 16-byte form above stays); MGCG_TILE_PACK=0 keeps the 16-byte form (A/B)
    if (tuning().tilePack.load(std::memory_order_relaxed) != 0 && tileShift <= 24) {
        std::vector<int> hdrBase((size_t)nTiles, 0);
        long long nHdr = 0;
        for (int t = 0; t < nTiles; ++t) { hdrBase[(size_t)t] = (int)nHdr; nHdr += ((long long)starts[(size_t)t + 1] - starts[(size_t)t] + kTilePack - 1) / kTilePack; }
        unsigned* tp = nullptr; int2* th = nullptr; int* ovf = nullptr;
        std::vector<int> hovf((size_t)nTiles, 0);
        bool pok = MGCG_HIP(hipMalloc((void**)&tp, sizeof(unsigned) * (size_t)nnz)) && MGCG_HIP(hipMalloc((void**)&th, sizeof(int2) * (size_t)(nHdr + 1))) &&
                   MGCG_HIP(hipMalloc((void**)&ovf, sizeof(int) * (size_t)nTiles)) && MGCG_HIP(hipMemsetAsync(ovf, 0, sizeof(int) * (size_t)nTiles, s));
        for (int t = 0; pok && t < nTiles; ++t) {
            const int kb = starts[(size_t)t], ke = starts[(size_t)t + 1];
            if (ke <= kb) continue;
            const long long nb = ((long long)ke - kb + kTilePack - 1) / kTilePack;
            hipLaunchKernelGGL(tiled_pack_kernel, dim3((unsigned)(nb > kMaxGrid ? kMaxGrid : nb)), dim3(kBlock), 0, s, tc, tr, kb, ke, tileShift, (int)((long long)t * tileWidth), tp, th, hdrBase[(size_t)t], ovf + t);
        }
        pok = pok && MGCG_HIP(hipGetLastError()) && MGCG_HIP(hipMemcpyAsync(hovf.data(), ovf, sizeof(int) * (size_t)nTiles, hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
        if (ovf) (void)hipFree(ovf);
        if (pok) {
            // a tile whose entries are so sparse that 1024 of them span more rows than the packed word can name (the narrow last tile of a
            // column range that is not a multiple of the tile width) keeps its 16-byte entries; the arrays they live in stay in that case
            bool all = true;
            for (int t = 0; t < nTiles; ++t) if (hovf[(size_t)t] != 0) { hdrBase[(size_t)t] = -1; all = false; }
            out->tilePacked = tp; out->tileHdr = th; out->tileHdrBase = hdrBase;
            if (all) { (void)hipFree(tc); (void)hipFree(tr); out->tileCols = nullptr; out->tileRowIds = nullptr; }     // the packed words carry both
        } else {
            if (tp) (void)hipFree(tp);
            if (th) (void)hipFree(th);
            (void)hipGetLastError();
        }
    }
    return true;
}

// 64-bit checksum of a CSR matrix (position-weighted sums of the bit patterns of the three arrays; the order of the additions does not
// matter): forms the library built without being asked are re-verified with it before every solve.
__global__ __launch_bounds__(kBlock) void csr_checksum_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets, const int* __restrict__ columnIndeces,
                                                              long long rows, long long nnz, unsigned long long* __restrict__ out)
{
    unsigned long long acc = 0;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride) {
        const unsigned long long w = 2ull * (unsigned long long)k + 1ull;
        acc += (unsigned long long)__double_as_longlong(__builtin_nontemporal_load(elements + k)) * w;
        acc += ((unsigned long long)(unsigned)__builtin_nontemporal_load(columnIndeces + k) + 0x9E3779B97F4A7C15ull) * (w * 0xBF58476D1CE4E5B9ull);
    }
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i <= rows; i += stride)
        acc += ((unsigned long long)(unsigned)rowOffsets[i] + 0x94D049BB133111EBull) * (2ull * (unsigned long long)i + 0x632BE59BD9B4E019ull);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}
unsigned long long csr_checksum(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces, long long rows, long long nnz, unsigned long long* scratch)
{
    unsigned long long v = 0;
    if (!MGCG_HIP(hipMemsetAsync(scratch, 0, sizeof(unsigned long long), s))) return 0;
    long long blocks = (nnz + kBlock - 1) / kBlock;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(csr_checksum_kernel, dim3((int)blocks), dim3(kBlock), 0, s, elements, rowOffsets, columnIndeces, rows, nnz, scratch);
    if (!MGCG_HIP(hipGetLastError()) || !MGCG_HIP(hipMemcpyAsync(&v, scratch, sizeof(v), hipMemcpyDeviceToHost, s)) || !MGCG_HIP(hipStreamSynchronize(s))) return 0;
    return v | 1ull;                                                // never 0: 0 says "no checksum"
}

void preload_kernels_tiled() { preload_code_object(reinterpret_cast<const void*>(&scan_totals_kernel)); }

} // namespace mgcg
