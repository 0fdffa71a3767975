// Row-tile SpMV for short rows (gfx950): the default CSR kernel when the average row has <= 8 nonzeros.
// Replaces cusparseDcsrmv of Mgcg/cuBlas/MgcgGpu/Mgcg.cu:10-19 on plain CSR arrays (nothing is re-encoded or cached).
//
// A workgroup of four wavefronts owns one tile of 256 consecutive rows per trip; inside it every wavefront is on its own:
// 64 rows, lane = row, a private 6 KB slice of LDS.  Per trip a wavefront
//   1. parks the raw column ids / values of its 64 rows (one contiguous span of the CSR arrays, fetched one trip earlier
//      with 16-byte loads) in LDS,
//   2. reads its own row's entries back (lane = row), issues the x gathers, the epilogue operands, the y store of the
//      PREVIOUS trip, the raw loads of the NEXT trip and the row offsets of the trip after that -- in this order, all
//      unconditional, so that every wait is a counted vmcnt that leaves the younger prefetches in flight,
//   3. forms acc += value * x[col] in stored order (product rounded, then added: bit-identical to
//      Mgcg/cuBlas/Mgcg/SparseMatrix.cs:68-88) and the epilogue value.
// What the measurements behind this shape say (conjugategradient_amd/tools/spmv_lab.hip, profiles/r2/spmv_lab_*.log):
//   * four wavefronts that issue the raw loads of four adjacent 64-row blocks together (one barrier per trip keeps them
//     in step) stream 7-10 % faster than four independent wavefronts: the HBM rows of a 26 KB span are opened once;
//   * 8 wavefronts per CU (grid = 2 workgroups per CU) is the sweet spot; 10-14 and 20 per CU lose 15-25 %;
//   * the y store costs 0.3-0.5 ms per GB whatever its shape (write-to-read turnarounds of the HBM); a non-temporal store
//     issued between the gathers and the next raw loads hides its latency and is the cheapest form measured;
//   * with a known plane stride (7-point stencils) the tiles are walked plane by plane inside one eighth of the
//     xy-plane per XCD ("z sweep"): the +-plane and +-line neighbours of a row are then multiplied on the same XCD one
//     trip apart and x is fetched from beyond L2 1.1 times instead of 4.7 times (PMC: 12.97 GB read for 12.87 GB
//     algorithmic at 512^3).
// Blocks the fast path cannot take (a span of more than 512 nonzeros, a row of more than 8, the ragged end of the arrays)
// are multiplied row by row from global memory by the same wavefront (slow, rare for the matrices this kernel is chosen for).
#include "common.hpp"
#include "spmv_epilogue.hpp"
#include <type_traits>

namespace mgcg {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));

constexpr int kTW = 4;              // wavefronts per workgroup
constexpr int kTRows = 64 * kTW;    // rows per tile
static_assert(kTRows == kRowTileRows, "common.hpp names the tile height for the callers that cut row ranges");
constexpr int kTCap = 512;          // nonzeros a wavefront stages per trip

// Order in which the tiles are walked ("z sweep" when the plane stride of a stencil matrix is known).
//   mode 0: tile = workgroup + trip * workgroups (memory order).
//   mode 1 (a plane has `per` = tilesPerPlane / workgroups slices of `workgroups` tiles): trip t covers slice t / nPlanes of
//           plane t % nPlanes -- one CONTIGUOUS run of `workgroups` tiles, of which XCD k = workgroup % 8 takes the k-th eighth;
//           every slice is swept through all planes before the next slice starts.
//   mode 2 (a plane is smaller than the grid: `per` = workgroups / tilesPerPlane whole planes per trip, again one contiguous run):
//           XCD k takes the k-th eighth of each of them.
// Either way the +-line and +-plane neighbours of a row are multiplied on the same XCD in the same or the adjacent trip (x is
// fetched from beyond L2 ~1.1 times instead of 4.7), and what the whole chip streams during a trip is contiguous in memory.
// (A variant in which the eight XCDs read eight separate runs per trip measured 6 % slower: profiles/r2/spmv_sweep_rows_vs_rowtile_disjoint_runs.log.)
// (struct TileMap: common.hpp; tile_map_trips / tile_map_tile: spmv_epilogue.hpp; make_tile_map below)

template <int EPI, int XS>
__device__ __forceinline__ RowsEpi tile_epi_prefetch(const SpmvArgs& a, int row)
{
    // as rows_epi_prefetch, but without a conditional load (a branch around a load would make the waits uncounted)
    RowsEpi o; o.w = 0.0; o.b = 0.0; o.dinv = 0.0; o.yold = 0.0;
    if constexpr (XS == 2) {       // Jacobi sweep on x1 + P e formed on the fly (uniform diagonal): w is finished by tile_epi_own below
        o.b = a.b[row]; o.w = a.xCoarse[coarse_of(a, a.cRowBase + row)]; o.dinv = a.dinvScalar;
        return o;
    }
    if constexpr (EPI == EPI_AXPBY_BETA) o.yold = a.y[row];
    if constexpr (EPI == EPI_DOT) o.w = a.w[row];
    if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_DOT) o.b = a.b[row];
    if constexpr (EPI == EPI_JACOBI || EPI == EPI_JACOBI_DOT) {
        o.b = a.b[row]; o.w = a.w[row];
        const double* dp = a.dinvUniform ? a.b : a.dinv;      // uniform diagonal: the array is not read (a.b stands in, same line as o.b)
        const double dl = dp[row];
        o.dinv = a.dinvUniform ? a.dinvScalar : dl;
    }
    return o;
}

// xScaled == 2: the sweep's own iterate x1[row] + e[parent(row)], x1 = xOuter * (xInner * b[row]) (o.w holds e[parent] so far)
template <int XS>
__device__ __forceinline__ void tile_epi_own(const SpmvArgs& a, RowsEpi& o)
{
    if constexpr (XS == 2) { const double t = a.xInner * o.b; const double x1 = a.xOuter * t; o.w = x1 + o.w; }
}

// NG = gathers issued per row in the fast path: 8, or 7 when no row of the matrix is longer (a 7-point stencil: one LDS read
// pair, one gather and one product fewer per row, 1-2.5 % -- profiles/r2/spmv_lab_lab15_micro.log, spmv_lab_lab17_records.log).
// XS: the multiplied vector is xOuter * (xInner * x[col]), formed per gather with the same two rounded products a stored first Jacobi
// sweep from zero would have used (the V(1,*) fold of the multigrid's residual pass, solver.hip: one 16 N pass less per level).
// XS == 2: ... + xCoarse[parent(col)] -- the prolongation of the coarse correction folded into the last sweep of a V(1,1) cycle as well
// (the prolongation kernel and the stored x1 + P e go: 33 N bytes less per level); a second gather per entry, from a vector an eighth the size.
template <int EPI, int NG, bool NT = false, int XS = 0>
__global__ __launch_bounds__(64 * kTW) void spmv_rowtile_kernel(SpmvArgs a, TileMap tm, int nTiles)
{
    __shared__ __attribute__((aligned(16))) int s_colAll[kTCap * kTW];
    __shared__ __attribute__((aligned(16))) double s_valAll[kTCap * kTW];
    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;
    const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* s_col = s_colAll + wv * kTCap;
    double* s_val = s_valAll + wv * kTCap;
    const int wg = blockIdx.x, nWG = gridDim.x;
    const int lastRow = a.rowCount - 1;                // (the loop runs over full tiles only: every row index below is <= lastRow)
    const int kMax4 = (a.elementsCount - 4) & ~3;     // last aligned quad of column ids that lies inside the arrays
    const int kMax2 = (a.elementsCount - 2) & ~1;

    // ---- the trips of this workgroup (workgroup-uniform, scalar)
    const int nTrips = tile_map_trips(tm, wg, nWG, nTiles);
    double dot = 0.0;
    auto finish = [&]() {
        // rows behind the last full tile: workgroup 0, one row per thread, straight from global memory
        const int tailRow = (nTiles + tm.gapSkip) * kTRows + (int)threadIdx.x;
        if (wg == 0 && tailRow <= lastRow) {
            double acc = 0.0;
            for (int k = a.rowOffsets[tailRow]; k < a.rowOffsets[tailRow + 1]; ++k) {
                const int col = a.columnIndeces[k];
                double xv = a.x[col];
                if constexpr (XS != 0) { const double t = a.xInner * xv; xv = a.xOuter * t; }
                if constexpr (XS == 2) xv = xv + a.xCoarse[coarse_of(a, col)];
                const double p = a.elements[k] * xv; acc += p;
            }
            RowsEpi eo;
            if constexpr (XS == 2) { eo = tile_epi_prefetch<EPI, XS>(a, tailRow); tile_epi_own<XS>(a, eo); }
            else eo = rows_epi_prefetch<EPI>(a, tailRow);
            a.y[tailRow] = rows_epilogue_value<EPI>(a, acc, eo, dot);
        }
        if constexpr (epi_has_dot(EPI)) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
            if (tid == 0) a.partials[blockIdx.x * kTW + wv] = dot;
        }
    };
    if (nTrips <= 0) { finish(); return; }                         // (workgroup-uniform)
    auto tile_of = [&](int t) -> int {
        t = t < nTrips ? t : nTrips - 1;                           // past the end: the last tile again (loads only, results unused)
        const int tile = tile_map_tile(tm, t, wg, nWG);
        return tile >= tm.gapAt ? tile + tm.gapSkip : tile;        // (gapSkip = 0 unless two row ranges share the launch)
    };

    // ---- per-trip pieces
    int roA_s, roA_e, roB_s, roB_e;                                // row offsets of my row: this trip / next trip
    i4 c0, c1; d2 v0, v1, v2, v3;                                  // raw span of this wavefront's 64 rows
    auto load_ro = [&](int tile, int& rs, int& re) {
        const int r = tile * kTRows + wv * 64 + tid;
        rs = a.rowOffsets[r]; re = a.rowOffsets[r + 1];
    };
    auto load_raw = [&](int s) {
        const int tb = s & ~3;
        int k0 = tb + 4 * tid, k1 = k0 + 256;
        k0 = k0 < kMax4 ? k0 : kMax4; k1 = k1 < kMax4 ? k1 : kMax4;
        if constexpr (NT) { c0 = __builtin_nontemporal_load((const i4*)(a.columnIndeces + k0)); c1 = __builtin_nontemporal_load((const i4*)(a.columnIndeces + k1)); }
        else { c0 = *(const i4*)(a.columnIndeces + k0); c1 = *(const i4*)(a.columnIndeces + k1); }
        int j0 = tb + 2 * tid, j1 = j0 + 128, j2 = j0 + 256, j3 = j0 + 384;
        j0 = j0 < kMax2 ? j0 : kMax2; j1 = j1 < kMax2 ? j1 : kMax2; j2 = j2 < kMax2 ? j2 : kMax2; j3 = j3 < kMax2 ? j3 : kMax2;
        if constexpr (NT) {
            v0 = __builtin_nontemporal_load((const d2*)(a.elements + j0)); v1 = __builtin_nontemporal_load((const d2*)(a.elements + j1));
            v2 = __builtin_nontemporal_load((const d2*)(a.elements + j2)); v3 = __builtin_nontemporal_load((const d2*)(a.elements + j3));
        } else { v0 = *(const d2*)(a.elements + j0); v1 = *(const d2*)(a.elements + j1); v2 = *(const d2*)(a.elements + j2); v3 = *(const d2*)(a.elements + j3); }
    };

    // prologue in the loop's own issue order: ro(0); raw(0); ro(1)
    int tileCur = tile_of(0), tileNext = tile_of(1);
    load_ro(tileCur, roA_s, roA_e);
    load_raw(__builtin_amdgcn_readfirstlane(roA_s));
    load_ro(tileNext, roB_s, roB_e);

    double pend = 0.0;
    int pendRow = tileCur * kTRows + wv * 64 + tid;
    // (the first store of the loop writes 0.0 to the first row, which the same lane overwrites one trip later; for epilogues
    //  that read y or an operand aliased with y the read of a row is always issued before any store to it)
    for (int t = 0; t < nTrips; ++t) {
        const int s = __builtin_amdgcn_readfirstlane(roA_s);
        const int e = __builtin_amdgcn_readlane(roA_e, 63);
        const int tb = s & ~3;
        const int my_s = roA_s, cnt = roA_e - roA_s;
        const int row = tileCur * kTRows + wv * 64 + tid;
        // the fast path holds the whole span in one pass and every row in one batch of NG gathers
        const bool spanOk = (e - tb <= kTCap) && (e <= kMax4 + 4);
        const bool short8 = __ballot(cnt > NG) == 0ull;
        const bool fast = spanOk && short8;                        // wavefront-uniform

        *(i4*)(s_col + 4 * tid) = c0; *(i4*)(s_col + 256 + 4 * tid) = c1;
        *(d2*)(s_val + 2 * tid) = v0; *(d2*)(s_val + 128 + 2 * tid) = v1; *(d2*)(s_val + 256 + 2 * tid) = v2; *(d2*)(s_val + 384 + 2 * tid) = v3;
        __syncthreads();
        int cc[NG]; double vv[NG], xg[NG];
        double accSlow = 0.0;
        if (fast) {
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                int idx = my_s - tb + j;
                idx = j < cnt ? idx : 0;                           // masked slots read entry 0 of the span (a valid column)
                cc[j] = s_col[idx]; vv[j] = s_val[idx];
            }
        } else {
#pragma unroll
            for (int j = 0; j < NG; ++j) { cc[j] = 0; vv[j] = 0.0; }
            for (int k = my_s; k < roA_e; ++k) {
                const int col = a.columnIndeces[k];
                double xv = a.x[col];
                if constexpr (XS != 0) { const double t = a.xInner * xv; xv = a.xOuter * t; }
                if constexpr (XS == 2) xv = xv + a.xCoarse[coarse_of(a, col)];
                const double p = a.elements[k] * xv; accSlow += p;
            }
        }
        __builtin_amdgcn_sched_barrier(0);                         // all LDS reads in flight before the first gather waits for its column id
#pragma unroll
        for (int j = 0; j < NG; ++j) xg[j] = a.x[cc[j]];
        double eg[XS == 2 ? NG : 1];
        if constexpr (XS == 2) {
#pragma unroll
            for (int j = 0; j < NG; ++j) eg[j] = a.xCoarse[coarse_of(a, cc[j])];
        }
        RowsEpi eo = tile_epi_prefetch<EPI, XS>(a, row);
        // the previous trip's result, then the next trip's raw stream and the row offsets of the trip after it
        __builtin_nontemporal_store(pend, a.y + pendRow);
        load_raw(__builtin_amdgcn_readfirstlane(roB_s));
        roA_s = roB_s; roA_e = roB_e;
        const int tileAfter = tile_of(t + 2);
        load_ro(tileAfter, roB_s, roB_e);
        __builtin_amdgcn_sched_barrier(0);                         // no product in front of the prefetch: its wait would hold the raw loads back
        double acc = 0.0;
        if constexpr (XS != 0) {
#pragma unroll
            for (int j = 0; j < NG; ++j) { const double t = a.xInner * xg[j]; xg[j] = a.xOuter * t; }
        }
        if constexpr (XS == 2) {
#pragma unroll
            for (int j = 0; j < NG; ++j) xg[j] = xg[j] + eg[j];
            tile_epi_own<XS>(a, eo);
        }
#pragma unroll
        for (int j = 0; j < NG; ++j) { const double p = vv[j] * xg[j]; acc += (j < cnt) ? p : 0.0; }
        acc = fast ? acc : accSlow;
        pend = rows_epilogue_value<EPI>(a, acc, eo, dot);
        pendRow = row;
        __syncthreads();
        tileCur = tileNext; tileNext = tileAfter;
    }
    __builtin_nontemporal_store(pend, a.y + pendRow);
    finish();
}

// out[0] = distance of the farthest entry of row `row` from the diagonal, out[1] = longest row of the matrix, out[2] / out[3] = sum of
// |col - row| >> 16 and number of entries over a sample of ~8192 rows (how far from the diagonal the gathers reach)  (out zeroed by the caller)
__global__ __launch_bounds__(kBlock) void matrix_shape_kernel(const int* __restrict__ rowOffsets, const int* __restrict__ columnIndeces, long long rows, long long row, long long rowBase, int* out)
{
    const long long sampleStride = rows > 8192 ? rows / 8192 : 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        long long far = 0;
        for (int k = rowOffsets[row]; k < rowOffsets[row + 1]; ++k) {
            long long d = (long long)columnIndeces[k] - (rowBase + row);
            d = d < 0 ? -d : d;
            far = d > far ? d : far;
        }
        out[0] = far > 0x7fffffffLL ? 0 : (int)far;
    }
    int longest = 0;
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        const int len = rowOffsets[i + 1] - rowOffsets[i]; longest = len > longest ? len : longest;
        if (i % sampleStride == 0 && len > 0 && len <= 4096) {
            long long far = 0;
            for (int k = rowOffsets[i]; k < rowOffsets[i + 1]; ++k) { long long d = (long long)columnIndeces[k] - (rowBase + i); far += (d < 0 ? -d : d) >> 16; }
            atomicAdd(&out[2], (int)(far > 0x3fffffLL ? 0x3fffffLL : far)); atomicAdd(&out[3], len);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_down(longest, off, 64); longest = o > longest ? o : longest; }
    __shared__ int s_max[kBlock / 64];
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = longest;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kBlock / 64; ++w) longest = s_max[w] > longest ? s_max[w] : longest;
        if (longest > 0) atomicMax(&out[1], longest);          // one atomic per workgroup, device memory
    }
}
void launch_matrix_shape(hipStream_t s, const int* rowOffsets, const int* columnIndeces, long long rows, long long row, long long rowBase, int* out2)
{
    long long blocks = (rows + kBlock - 1) / kBlock;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(matrix_shape_kernel, dim3((int)blocks), dim3(kBlock), 0, s, rowOffsets, columnIndeces, rows, row, rowBase, out2);
}

// Tile order for a matrix of `rows` rows whose far band lies periodRows rows from the diagonal (0: unknown).
TileMap make_tile_map(long long rows, int periodRows, int nWG, int tileRows)
{
    TileMap tm{};
    tm.mode = 0;
    if (periodRows <= 0 || (nWG & 7) != 0 || nWG < 8) return tm;
    if (tileRows <= 0 || periodRows % (tileRows * 8) != 0 || rows % periodRows != 0) return tm;
    const int tilesPerPlane = periodRows / tileRows;
    const long long nPlanes = rows / periodRows;
    if (nPlanes < 3 || nPlanes > 65535) return tm;
    tm.tilesPerPlane = tilesPerPlane; tm.nPlanes = (int)nPlanes;
    if (tilesPerPlane >= nWG && tilesPerPlane % nWG == 0) { tm.mode = 1; tm.per = tilesPerPlane / nWG; }
    else if (tilesPerPlane < nWG && nWG % tilesPerPlane == 0) { tm.mode = 2; tm.per = nWG / tilesPerPlane; }
    return tm;
}

template <int EPI>
static int launch_rowtile_epi(hipStream_t s, const SpmvArgs& a, int periodRows, int gridReq, int maxRow, bool ntWindow, int gapAt, int gapSkip)
{
    const int nTiles = a.rowCount / kTRows - gapSkip;             // full tiles to do; the kernel's workgroup 0 takes the rows behind the last one
    DeviceState* d = device_state();
    const int numCu = d ? d->numCu : kNumCu;
    int nWG = gridReq > 0 ? gridReq / kTW : 2 * numCu;            // 8 wavefronts per CU
    if (nWG * kTW > kMaxPartials) nWG = kMaxPartials / kTW;
    if (nWG > nTiles) nWG = nTiles;
    if (nWG < 1) nWG = 1;
    TileMap tm = make_tile_map(a.rowCount, gapSkip > 0 ? 0 : periodRows, nWG, kTRows);
    tm.gapAt = gapAt; tm.gapSkip = gapSkip;
    // Matrix streams with the non-temporal hint when the vectors of the system are small enough to live in the 256 MB Infinity Cache
    // between the kernels of an iteration (the slab of one rank of an 8-GPU run, the coarse levels of a hierarchy): the matrix, read once
    // per product, then does not push them out.  CG iteration, alternating inside one process (profiles/r2, the A/B script is in the history): -1 % at 8.4 M rows, -3.5 % at 16.8 M,
    // -1.5 % at 33.5 M, +0.2 % at 42 M, +1.5 % at 134 M (there the hint only costs) and +10 % at 2 M (there the matrix itself would have stayed in the cache).  Only the plain CG loop asks for it (SpmvConfig::flags & 8): inside the V-cycle
    // the same hint made the slab's MGCG iteration 1-3 % slower.
    const bool nt = ntWindow && a.rowCount >= 8000000 && a.rowCount <= 36000000;
    const bool seven = maxRow > 0 && maxRow <= 7;
    if (a.xScaled == 2) {                                         // (the multigrid's last sweep of a V(1,1) cycle: x1 + P e per gather)
        if constexpr (EPI == EPI_JACOBI || EPI == EPI_JACOBI_DOT) {
            if (a.xCoarse == nullptr || !a.dinvUniform) { set_error("row-tile SpMV: the folded prolongation needs the coarse vector and a uniform diagonal"); return 0; }
            if (seven) hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 7, false, 2>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
            else hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 8, false, 2>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
            return nWG * kTW;
        } else { set_error("row-tile SpMV: the folded prolongation is implemented for the Jacobi epilogues only"); return 0; }
    }
    if (a.xScaled) {                                              // (only the multigrid's residual pass asks; never with the streaming hint)
        if constexpr (EPI == EPI_RESIDUAL) {
            if (seven) hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 7, false, 1>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
            else hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 8, false, 1>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
            return nWG * kTW;
        } else { set_error("row-tile SpMV: a scaled multiplicand is implemented for the residual epilogue only"); return 0; }
    }
    if (nt) {
        if (seven) hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 7, true>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
        else hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 8, true>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
    }
    else if (seven) hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 7>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
    else hipLaunchKernelGGL((spmv_rowtile_kernel<EPI, 8>), dim3(nWG), dim3(64 * kTW), 0, s, a, tm, nTiles);
    return nWG * kTW;
}

// Requires 16-byte aligned elements and columnIndeces and elementsCount >= 8 (checked by the caller).
int launch_spmv_rowtile(hipStream_t s, int epilogue, const SpmvArgs& a, int periodRows, int gridReq, int maxRow, bool ntWindow, int gapAt, int gapSkip)
{
    if (a.rowCount <= 0) return 0;
    if (gapSkip <= 0) { gapAt = 0; gapSkip = 0; }
    switch (epilogue) {
    case EPI_AXPBY:        return a.beta != 0.0 ? launch_rowtile_epi<EPI_AXPBY_BETA>(s, a, periodRows, gridReq, maxRow, ntWindow, gapAt, gapSkip) : launch_rowtile_epi<EPI_AXPBY>(s, a, periodRows, gridReq, maxRow, ntWindow, gapAt, gapSkip);
    case EPI_DOT:          return launch_rowtile_epi<EPI_DOT>(s, a, periodRows, gridReq, maxRow, ntWindow, gapAt, gapSkip);
    case EPI_RESIDUAL:     return launch_rowtile_epi<EPI_RESIDUAL>(s, a, periodRows, gridReq, maxRow, ntWindow, gapAt, gapSkip);
    case EPI_RESIDUAL_DOT: return launch_rowtile_epi<EPI_RESIDUAL_DOT>(s, a, periodRows, gridReq, maxRow, ntWindow, gapAt, gapSkip);
    case EPI_JACOBI:       return launch_rowtile_epi<EPI_JACOBI>(s, a, periodRows, gridReq, maxRow, ntWindow, gapAt, gapSkip);
    case EPI_JACOBI_DOT:   return launch_rowtile_epi<EPI_JACOBI_DOT>(s, a, periodRows, gridReq, maxRow, ntWindow, gapAt, gapSkip);
    }
    return 0;
}

void preload_kernels_rowtile() { preload_code_object(reinterpret_cast<const void*>(&matrix_shape_kernel)); }

} // namespace mgcg

// Host-side view of the tile order (no device needed): the tile every workgroup takes in every trip, for tests of the enumeration.
// tiles[wg * maxTrips + t] = tile index or -1; returns the TileMap mode (0 memory order, 1 / 2 z sweep), or -1 on bad arguments.
extern "C" int MgcgDebugTileOrder(long long rows, int periodRows, int workgroups, int tileRows, int* tiles, int maxTrips)
{
    if (rows < 0 || workgroups < 1 || tileRows < 1 || maxTrips < 1 || !tiles) return -1;
    const mgcg::TileMap tm = mgcg::make_tile_map(rows, periodRows, workgroups, tileRows);
    const int nTiles = (int)(rows / tileRows);
    for (int wg = 0; wg < workgroups; ++wg) {
        const int n = mgcg::tile_map_trips(tm, wg, workgroups, nTiles);
        if (n > maxTrips) return -1;
        for (int t = 0; t < maxTrips; ++t) tiles[(long long)wg * maxTrips + t] = t < n ? mgcg::tile_map_tile(tm, t, wg, workgroups) : -1;
    }
    return tm.mode;
}
