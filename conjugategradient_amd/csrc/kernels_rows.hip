// Row-block SpMV, "stage raw, multiply by row" form (gfx950): one wavefront owns 64 consecutive rows per trip.
//   1. all 64 lanes stream the rows' contiguous nonzero span with whole-line loads (CSR: column ids + values,
//      12 B/nnz; DCSR: one or two code bytes per nonzero, see kernels_dcsr.hip) and park the RAW entries in LDS;
//   2. lane r then walks row r: reads its entries from LDS in stored order, gathers x and accumulates
//      acc += value * x[col] (product rounded, then added: bit-identical to Mgcg/cuBlas/Mgcg/SparseMatrix.cs:68-88).
// Compared with spmv_stream_kernel (products parked in LDS, then reduced) the gathers are issued per diagonal:
// in gather j the 64 lanes address x[row + offset_j], i.e. 64 nearly consecutive doubles (4-5 cache lines) instead
// of a mix of all diagonals (10+ lines), and the compressed formats need no row search (lane == row).
// Same software pipeline as the stream kernel: gathers(t) -> epilogue operands(t) -> raw loads(t+1) -> y store(t-1),
// so every wait is a counted vmcnt that leaves the prefetch and the store in flight.
#include "common.hpp"
#include "spmv_epilogue.hpp"

namespace mgcg {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

enum { FMT_CSR = 0, FMT_DCSR8 = 1, FMT_DCSR64 = 2 };   // DCSR8: offset + value codes; DCSR64: offset codes + fp64 values

constexpr int kRR = 64;            // rows per trip (one per lane)
constexpr int kRCap = 512;         // nonzeros staged per pass (8 per lane)
constexpr int kRAlign = 32;        // spans are read from a 32-nonzero boundary (128 B of column ids / 32 B of codes)
constexpr int kRDict = 256;

// Raw data of one pass held in registers between the prefetch and the LDS staging.
template <int FMT>
struct RowsStage {
    i2 col[4];          // FMT_CSR: 8 column ids (4 x int2, one per 128-nonzero quarter)
    d2 val[4];          // FMT_CSR / FMT_DCSR64: 8 values
    u2 cc, vc;          // DCSR: 8 offset codes, 8 value codes
    int my_s, my_e;     // nonzero range of this lane's row
    int s, e;           // nonzero span of the row block (wave-uniform)
};

// Lane layout of a pass (512 nonzeros from tb): CSR and DCSR64 values: quarter q (128 nonzeros) lane t owns 128q+2t, +1
// (whole 1 KiB / 512 B lines per wave-instruction); codes: lane t owns 8t..8t+7 (one 8-byte load).
template <int EPI, int FMT>
__global__ __launch_bounds__(64) void spmv_rows_kernel(SpmvArgs a, DcsrView m, int nRowBlocks)
{
    __shared__ __attribute__((aligned(16))) double s_val[(FMT == FMT_DCSR8) ? 2 : kRCap];
    __shared__ __attribute__((aligned(16))) int s_col[(FMT == FMT_CSR) ? kRCap : 4];
    __shared__ __attribute__((aligned(16))) unsigned char s_cc[(FMT == FMT_CSR) ? 16 : kRCap];
    __shared__ __attribute__((aligned(16))) unsigned char s_vc[(FMT == FMT_DCSR8) ? kRCap : 16];
    __shared__ double s_vD[(FMT == FMT_DCSR8) ? kRDict : 1];
    __shared__ int s_dD[(FMT == FMT_CSR) ? 1 : kRDict];

    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;
    const int tid = threadIdx.x;
    if constexpr (FMT != FMT_CSR) {
        for (int i = tid; i < kRDict; i += 64) {
            s_dD[i] = i < m.nDelta ? m.deltaDict[i] : 0;
            if constexpr (FMT == FMT_DCSR8) s_vD[i] = i < m.nValue ? m.valueDict[i] : 0.0;
        }
        __syncthreads();
    }

    const long long lastRow = (long long)a.rowCount - 1;
    const int nTrips = ((long long)nRowBlocks > (long long)blockIdx.x) ? (int)(((long long)nRowBlocks - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    auto rb_of = [&](int t) -> long long { return (long long)blockIdx.x + (long long)t * gridDim.x; };
    auto span_of = [&](long long rb, int& s, int& e) {
        const long long r0 = rb * kRR;
        const long long r1 = (r0 + kRR < (long long)a.rowCount) ? r0 + kRR : (long long)a.rowCount;
        s = a.rowOffsets[r0]; e = a.rowOffsets[r1];
    };
    // clamps for the unconditional wide loads (host guarantees elementsCount >= 8)
    const int kMaxPair = (a.elementsCount - 2) & ~1;
    const int kMaxOct = (a.elementsCount - 8) & ~7;

    // issue the raw loads of the pass that starts at nonzero tb (a multiple of 8)
    auto load_pass = [&](RowsStage<FMT>& st, int tb) {
        if constexpr (FMT == FMT_CSR || FMT == FMT_DCSR64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int k = tb + q * 128 + 2 * tid;
                k = k < kMaxPair ? k : kMaxPair;
                if constexpr (FMT == FMT_CSR) st.col[q] = *(const i2*)(a.columnIndeces + k);
                st.val[q] = *(const d2*)(a.elements + k);
            }
        }
        if constexpr (FMT != FMT_CSR) {
            int k = tb + 8 * tid;
            k = k < kMaxOct ? k : kMaxOct;
            st.cc = *(const u2*)(m.colCode + k);
            if constexpr (FMT == FMT_DCSR8) st.vc = *(const u2*)(m.valCode + k);
        }
    };
    auto issue = [&](RowsStage<FMT>& st, long long rb) {
        const long long r0 = rb * kRR;
        long long row = r0 + tid;
        const bool live = row <= lastRow;
        row = live ? row : lastRow;
        const int ms = a.rowOffsets[row], me = a.rowOffsets[row + 1];
        st.my_s = live ? ms : st.e;
        st.my_e = live ? me : st.e;
        load_pass(st, st.s & ~(kRAlign - 1));
    };
    // park the raw pass in LDS; entries past the arrays' end are re-read with guards (they were clamped in load_pass)
    auto stage_pass = [&](const RowsStage<FMT>& st, int tb, int e) {
        if constexpr (FMT == FMT_CSR || FMT == FMT_DCSR64) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = q * 128 + 2 * tid;
                const int k = tb + o;
                if (k <= kMaxPair) {
                    if constexpr (FMT == FMT_CSR) *(i2*)(s_col + o) = st.col[q];
                    *(d2*)(s_val + o) = st.val[q];
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        if (k + j < e) { if constexpr (FMT == FMT_CSR) s_col[o + j] = a.columnIndeces[k + j]; s_val[o + j] = a.elements[k + j]; }
                }
            }
        }
        if constexpr (FMT != FMT_CSR) {
            const int o = 8 * tid;
            const int k = tb + o;
            if (k <= kMaxOct) {
                *(u2*)(s_cc + o) = st.cc;
                if constexpr (FMT == FMT_DCSR8) *(u2*)(s_vc + o) = st.vc;
            } else {
                for (int j = 0; j < 8; ++j)
                    if (k + j < e) { s_cc[o + j] = m.colCode[k + j]; if constexpr (FMT == FMT_DCSR8) s_vc[o + j] = m.valCode[k + j]; }
            }
        }
    };

    double dotacc = 0.0;
    if (nTrips > 0) {
        RowsStage<FMT> cur, nxt;
        span_of(rb_of(0), cur.s, cur.e);
        issue(cur, rb_of(0));
        nxt.s = cur.s; nxt.e = cur.e;
        if (nTrips > 1) span_of(rb_of(1), nxt.s, nxt.e);
        double pendVal = 0.0;
        long long pendRow = -1;
        for (int t = 0; t < nTrips; ++t) {
            const long long rb = rb_of(t);
            const long long r0 = rb * kRR;
            const long long left = (long long)a.rowCount - r0;
            const int nr = (int)(left < kRR ? left : kRR);
            int s2 = nxt.s, e2 = nxt.e;
            if (t + 2 < nTrips) span_of(rb_of(t + 2), s2, e2);
            const int e = cur.e;
            const int tb0 = cur.s & ~(kRAlign - 1);
            const long long myRowGlobal = m.rowBase + r0 + tid;       // DCSR: col = global row + offset

            // ---- first pass: stage, then 8 entries of my row ----
            stage_pass(cur, tb0, e);
            __syncthreads();
            double acc = 0.0;
            int k = cur.my_s;                                          // next entry of my row
            const int passEnd0 = tb0 + kRCap;
            double xg[8], vv[8];
            int cnt = cur.my_e - k; cnt = cnt < 8 ? cnt : 8;
            { const int room = passEnd0 - k; cnt = cnt < room ? cnt : room; cnt = cnt > 0 ? cnt : 0; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                int idx = k + j - tb0;
                idx = (j < cnt) ? idx : 0;                            // masked lanes read slot 0 (a real entry of the block)
                long long col;
                if constexpr (FMT == FMT_CSR) { col = s_col[idx]; vv[j] = s_val[idx]; }
                else {
                    col = myRowGlobal + s_dD[s_cc[idx]];
                    if constexpr (FMT == FMT_DCSR8) vv[j] = s_vD[s_vc[idx]]; else vv[j] = s_val[idx];
                }
                if (j >= cnt) col = 0;                                // masked: any valid column
                if (MGCG_ABLATE(a, 2)) col &= 1023;
                xg[j] = a.x[col];
            }
            long long myRow = r0 + tid;
            myRow = myRow <= lastRow ? myRow : lastRow;
            const RowsEpi eo = rows_epi_prefetch<EPI>(a, myRow);
            // the next block's raw stream goes in flight behind the gathers, then the previous trip's result
            issue(nxt, t + 1 < nTrips ? rb_of(t + 1) : rb);
            if (pendRow >= 0 && !MGCG_ABLATE(a, 1)) a.y[pendRow] = pendVal;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const double p = vv[j] * xg[j]; acc += (j < cnt) ? p : 0.0; }
            k += cnt;
            // ---- the rest of a long row inside this pass, then further passes (not pipelined) ----
            int tb = tb0;
            while (true) {
                const int passEnd = tb + kRCap;
                const int stop = cur.my_e < passEnd ? cur.my_e : passEnd;
                for (; k < stop; ++k) {
                    const int idx = k - tb;
                    long long col; double v;
                    if constexpr (FMT == FMT_CSR) { col = s_col[idx]; v = s_val[idx]; }
                    else { col = myRowGlobal + s_dD[s_cc[idx]]; if constexpr (FMT == FMT_DCSR8) v = s_vD[s_vc[idx]]; else v = s_val[idx]; }
                    const double p = v * a.x[col];
                    acc += p;
                }
                tb += kRCap;
                if (tb >= e) break;                                   // wave-uniform
                __syncthreads();
                RowsStage<FMT> more;
                more.s = cur.s; more.e = cur.e; more.my_s = cur.my_s; more.my_e = cur.my_e;
                load_pass(more, tb);
                stage_pass(more, tb, e);
                __syncthreads();
            }
            pendRow = -1;
            if (tid < nr) { pendVal = rows_epilogue_value<EPI>(a, acc, eo, dotacc); pendRow = r0 + tid; }
            if (MGCG_ABLATE(a, 1) && acc == 1.2345e300) pendRow = 0;
            __syncthreads();                                          // LDS is free for the next trip
            cur = nxt;
            nxt.s = s2; nxt.e = e2;
        }
        if (pendRow >= 0 && !MGCG_ABLATE(a, 1)) a.y[pendRow] = pendVal;
    }
    if constexpr (epi_has_dot(EPI)) {
        double v = dotacc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (tid == 0) a.partials[blockIdx.x] = v;
    }
}

template <int EPI>
static int launch_rows_epi(hipStream_t s, const SpmvArgs& a, const DcsrView* m, int gridReq)
{
    const int nRowBlocks = (int)(((long long)a.rowCount + kRR - 1) / kRR);
    DeviceState* d = device_state();
    // resident wavefronts per CU measured best on the 7-point 512^3 matrix: 8 for CSR (HBM-bound), 16 for the compressed
    // forms (latency-bound decode); more only adds cache pressure (profiles/r1/spmv_sweep_rows_dcsr_512.log)
    int grid = gridReq > 0 ? gridReq : (m == nullptr ? 8 : 16) * (d ? d->numCu : kNumCu);
    if (grid > kMaxPartials) grid = kMaxPartials;
    if (grid > nRowBlocks) grid = nRowBlocks;
    if (grid < 1) grid = 1;
    DcsrView none{};
    if (m == nullptr) hipLaunchKernelGGL((spmv_rows_kernel<EPI, FMT_CSR>), dim3(grid), dim3(64), 0, s, a, none, nRowBlocks);
    else if (m->valCode != nullptr) hipLaunchKernelGGL((spmv_rows_kernel<EPI, FMT_DCSR8>), dim3(grid), dim3(64), 0, s, a, *m, nRowBlocks);
    else hipLaunchKernelGGL((spmv_rows_kernel<EPI, FMT_DCSR64>), dim3(grid), dim3(64), 0, s, a, *m, nRowBlocks);
    return grid;
}

// ---------------------------------------------------------------- row-pattern form (class 3, kernels_dcsr.hip)
// One byte per row names the row's whole sequence of (col - row, value) pairs; the table of sequences sits in LDS.
// Lane = row, RPL rows per lane per trip (rows r, r+64, ...): per wave-trip RPL coalesced 64-byte id loads, then per
// entry slot j one LDS read of (offset, value) and one gather -- no row offsets, no column ids, no values from HBM.
// Eight gathers per row are issued back to back before the first product is needed; products are added in stored order
// (masked slots add +0.0, which leaves the sum's bits alone), so the result equals the CSR kernels' bit for bit.
template <int EPI, int CH>
__global__ __launch_bounds__(64) void spmv_pattern_kernel(SpmvArgs a, DcsrView m, int nRowBlocks, int groupBlocks, TileMap tm)
{
    constexpr int RPL = 2;                                        // rows per lane per trip (rows r and r + 64)
    extern __shared__ __attribute__((aligned(16))) unsigned char s_raw[];
    const int W = m.patWidth, nP = m.nPattern;
    double* s_val = (double*)s_raw;                               // [nP * W]
    int* s_delta = (int*)(s_val + nP * W);                        // [nP * W]
    int* s_cnt = s_delta + nP * W;                                // [nP]
    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < nP * W; i += 64) { s_val[i] = m.patValue[i]; s_delta[i] = m.patDelta[i]; }
    for (int i = tid; i < nP; i += 64) s_cnt[i] = m.patCount[i];
    __syncthreads();

    // Trip -> row block.  Workgroup k runs on XCD k % 8 (round-robin dispatch); with groupBlocks = G > 0 every XCD owns
    // every 8th run of G consecutive row blocks, so the +-1 grid-line (and, for the usual extents, +-1 plane) neighbours
    // of a row are multiplied on the same XCD and their x lines are shared through its L2.  G = 0 (default): plain
    // grid-stride -- measured equal or faster at 512^3 (profiles/r1/spmv_sweep_pattern_512.log); the launcher passes G = 0 since round 5.
    const bool grouped = groupBlocks > 0 && (gridDim.x & 7) == 0;
    const long long G = grouped ? groupBlocks : 1;
    const int xcd = grouped ? (int)(blockIdx.x & 7) : 0;
    const long long first = grouped ? (long long)(blockIdx.x >> 3) : (long long)blockIdx.x;
    const long long step = grouped ? (long long)(gridDim.x >> 3) : (long long)gridDim.x;
    const long long count = grouped ? (((long long)nRowBlocks + 8 * G - 1) / (8 * G)) * G : (long long)nRowBlocks;
    auto block_of = [&](long long L) -> long long { return grouped ? ((L / G) * 8 + xcd) * G + (L % G) : L; };

    const long long lastRow = (long long)a.rowCount - 1;
    auto load_ids = [&](long long rb, int* pid) {
        const long long base = rb * (64 * RPL);
#pragma unroll
        for (int u = 0; u < RPL; ++u) { long long r = base + u * 64 + tid; r = r <= lastRow ? r : lastRow; r = r >= 0 ? r : 0; pid[u] = m.patternId[r]; }
    };
    // tm.mode != 0: the z sweep of the row-tile kernel (kernels_rowtile.hip) with tiles of 128 rows -- one contiguous run of row blocks
    // per trip for the whole chip, XCD k its k-th eighth, plane after plane: x comes from beyond L2 ~1.1 times instead of 2.5
    const bool sweep = tm.mode != 0;
    const long long nTrips = sweep ? tile_map_trips(tm, (int)blockIdx.x, (int)gridDim.x, nRowBlocks) : (count > first ? (count - first + step - 1) / step : 0);
    auto trip_block = [&](long long t) -> long long { return sweep ? (long long)tile_map_tile(tm, (int)t, (int)blockIdx.x, (int)gridDim.x) : block_of(first + t * step); };
    double dotacc = 0.0;
    int pidNext[RPL];
    if (nTrips > 0) load_ids(trip_block(0) < nRowBlocks ? trip_block(0) : 0, pidNext);
    for (long long t = 0; t < nTrips; ++t) {
        const long long rb = trip_block(t);
        int pid[RPL];
#pragma unroll
        for (int u = 0; u < RPL; ++u) pid[u] = pidNext[u];
        if (t + 1 < nTrips) { const long long nb = trip_block(t + 1); load_ids(nb < nRowBlocks ? nb : 0, pidNext); }   // ids of the next trip, in flight behind the gathers
        if (rb >= nRowBlocks) continue;                            // wave-uniform (tail of the grouped enumeration)
        const long long base = rb * (64 * RPL);
        long long row[RPL]; bool live[RPL]; int tb[RPL], cnt[RPL]; double acc[RPL]; RowsEpi eo[RPL];
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
            row[u] = base + u * 64 + tid;
            live[u] = row[u] <= lastRow;
            row[u] = live[u] ? row[u] : lastRow;
        }
#pragma unroll
        for (int u = 0; u < RPL; ++u) { eo[u] = rows_epi_prefetch<EPI>(a, row[u]); tb[u] = pid[u] * W; cnt[u] = s_cnt[pid[u]]; acc[u] = 0.0; }
        int longest = 0, shortest = 0x7fffffff;
#pragma unroll
        for (int u = 0; u < RPL; ++u) { longest = cnt[u] > longest ? cnt[u] : longest; shortest = cnt[u] < shortest ? cnt[u] : shortest; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int o = __shfl_xor(longest, off, 64), q = __shfl_xor(shortest, off, 64);
            longest = o > longest ? o : longest; shortest = q < shortest ? q : shortest;
        }
        if (shortest == CH && longest == CH) {
            // every row of the block is a full-length row (the interior of a stencil): CH gathers per row, no masks
            double xg[RPL][CH], vv[RPL][CH];
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const long long g = m.rowBase + row[u];
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    vv[u][j] = s_val[tb[u] + j];
                    long long col = g + s_delta[tb[u] + j];
                    if (MGCG_ABLATE(a, 2)) col &= 1023;           // lab builds: gathers served from L1
                    xg[u][j] = a.x[col];
                }
            }
            if (a.xScaled) {                                       // wave-uniform: x[col] stands for xOuter * (xInner * x[col])
#pragma unroll
                for (int u = 0; u < RPL; ++u)
#pragma unroll
                    for (int j = 0; j < CH; ++j) { const double t = a.xInner * xg[u][j]; xg[u][j] = a.xOuter * t; }
            }
#pragma unroll
            for (int u = 0; u < RPL; ++u)
#pragma unroll
                for (int j = 0; j < CH; ++j) { const double p = vv[u][j] * xg[u][j]; acc[u] += p; }
        } else {
            for (int j0 = 0; j0 < longest; j0 += CH) {             // wave-uniform trip count
                double xg[RPL][CH], vv[RPL][CH];
#pragma unroll
                for (int u = 0; u < RPL; ++u) {
                    const long long g = m.rowBase + row[u];
#pragma unroll
                    for (int j = 0; j < CH; ++j) {
                        const bool on = j0 + j < cnt[u];
                        const int t = tb[u] + (on ? j0 + j : 0);   // masked slots re-read the row's first entry
                        vv[u][j] = s_val[t];
                        xg[u][j] = a.x[g + s_delta[t]];
                    }
                }
                if (a.xScaled) {
#pragma unroll
                    for (int u = 0; u < RPL; ++u)
#pragma unroll
                        for (int j = 0; j < CH; ++j) { const double t = a.xInner * xg[u][j]; xg[u][j] = a.xOuter * t; }
                }
#pragma unroll
                for (int u = 0; u < RPL; ++u)
#pragma unroll
                    for (int j = 0; j < CH; ++j) { const double p = vv[u][j] * xg[u][j]; acc[u] += (j0 + j < cnt[u]) ? p : 0.0; }
            }
        }
#pragma unroll
        for (int u = 0; u < RPL; ++u)
            if (live[u]) {
                const double v = rows_epilogue_value<EPI>(a, acc[u], eo[u], dotacc);
                if (!MGCG_ABLATE(a, 1) || v == 1.2345e300) a.y[row[u]] = v;       // (lab builds, bit0: no y store)
            }
    }
    if constexpr (epi_has_dot(EPI)) {
        double v = dotacc;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (tid == 0) a.partials[blockIdx.x] = v;
    }
}

constexpr int kPatternWavesPerCu = 16;     // (A/B of rounds 2-3: 8 / 16 / 32 wavefronts per CU and grouped row blocks; 16 ungrouped stayed)

template <int EPI>
static int launch_pattern_epi(hipStream_t s, const SpmvArgs& a, const DcsrView& m, int gridReq, int periodRows)
{
    const int nRowBlocks = (int)(((long long)a.rowCount + 127) / 128);
    DeviceState* d = device_state();
    int grid = gridReq > 0 ? gridReq : kPatternWavesPerCu * (d ? d->numCu : kNumCu);
    if (grid > kMaxPartials) grid = kMaxPartials;
    if (grid > nRowBlocks) grid = nRowBlocks;
    if (grid < 1) grid = 1;
    const size_t lds = (size_t)m.nPattern * m.patWidth * 12 + (size_t)m.nPattern * 4;
    const int group = 0;
    TileMap tm = make_tile_map(a.rowCount, (a.rowCount % 128 != 0) ? 0 : periodRows, grid, 128);
    // slots per pass = the longest row when it is 5 or 7 (the 2-D / 3-D stencils), else 8
    if (m.patWidth == 7) hipLaunchKernelGGL((spmv_pattern_kernel<EPI, 7>), dim3(grid), dim3(64), lds, s, a, m, nRowBlocks, group, tm);
    else if (m.patWidth == 5) hipLaunchKernelGGL((spmv_pattern_kernel<EPI, 5>), dim3(grid), dim3(64), lds, s, a, m, nRowBlocks, group, tm);
    else hipLaunchKernelGGL((spmv_pattern_kernel<EPI, 8>), dim3(grid), dim3(64), lds, s, a, m, nRowBlocks, group, tm);
    return grid;
}

static int launch_spmv_pattern(hipStream_t s, int epilogue, const SpmvArgs& a, const DcsrView& m, int gridReq, int periodRows)
{
    switch (epilogue) {
    case EPI_AXPBY:        return a.beta != 0.0 ? launch_pattern_epi<EPI_AXPBY_BETA>(s, a, m, gridReq, periodRows) : launch_pattern_epi<EPI_AXPBY>(s, a, m, gridReq, periodRows);
    case EPI_DOT:          return launch_pattern_epi<EPI_DOT>(s, a, m, gridReq, periodRows);
    case EPI_RESIDUAL:     return launch_pattern_epi<EPI_RESIDUAL>(s, a, m, gridReq, periodRows);
    case EPI_RESIDUAL_DOT: return launch_pattern_epi<EPI_RESIDUAL_DOT>(s, a, m, gridReq, periodRows);
    case EPI_JACOBI:       return launch_pattern_epi<EPI_JACOBI>(s, a, m, gridReq, periodRows);
    case EPI_JACOBI_DOT:   return launch_pattern_epi<EPI_JACOBI_DOT>(s, a, m, gridReq, periodRows);
    }
    return 0;
}

// m == nullptr: plain CSR.  Requires 16-byte aligned elements, 8-byte aligned columnIndeces and elementsCount >= 8.
int launch_spmv_rows(hipStream_t s, int epilogue, const SpmvArgs& a, const DcsrView* m, int gridReq, int periodRows)
{
    if (a.rowCount <= 0) return 0;
    if (m != nullptr && m->patternId != nullptr) return launch_spmv_pattern(s, epilogue, a, *m, gridReq, periodRows);
    switch (epilogue) {
    case EPI_AXPBY:        return a.beta != 0.0 ? launch_rows_epi<EPI_AXPBY_BETA>(s, a, m, gridReq) : launch_rows_epi<EPI_AXPBY>(s, a, m, gridReq);
    case EPI_DOT:          return launch_rows_epi<EPI_DOT>(s, a, m, gridReq);
    case EPI_RESIDUAL:     return launch_rows_epi<EPI_RESIDUAL>(s, a, m, gridReq);
    case EPI_RESIDUAL_DOT: return launch_rows_epi<EPI_RESIDUAL_DOT>(s, a, m, gridReq);
    case EPI_JACOBI:       return launch_rows_epi<EPI_JACOBI>(s, a, m, gridReq);
    case EPI_JACOBI_DOT:   return launch_rows_epi<EPI_JACOBI_DOT>(s, a, m, gridReq);
    }
    return 0;
}

void preload_kernels_rows() { preload_code_object(reinterpret_cast<const void*>(&(spmv_rows_kernel<EPI_DOT, FMT_CSR>))); }

} // namespace mgcg
