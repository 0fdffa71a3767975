// tile_lab: the column-tile pass of the opt-in class-4 form (csrc/kernels_tiled.hip) taken apart on a synthetic matrix of BASELINE
// config 5's shape (10 M rows, ~31 entries per row, columns uniform over the whole range): which part of the 3.3 ms per product is the
// gathers, which the y read-modify-write, which the streams -- and what a 12-byte entry (value + one packed word) buys.  Measurement
// tool, not part of the library.  Every variant is checked bit for bit against the production-shaped pass (V0) and V0 against the host.
//   tile_lab [rows=10000000] [tiles=20] [tileShift=19] [meanPerCell=1.55] [tileWidth=2^tileShift columns; any width <= 2^tileShift]
//   TILE_LAB_QUICK=1: only the 12-byte production pass (for PMC runs); TILE_LAB_V3=1 / TILE_LAB_V4=1: the row-block-persistent variants;
//   TILE_LAB_V3_ONE=1: the one V3 variant that did best (for PMC runs)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

#ifndef KBLOCK
#define KBLOCK 256
#endif
#ifndef KTILEE
#define KTILEE 4
#endif
constexpr int kBlock = KBLOCK;
constexpr int kTileE = KTILEE;
constexpr int kEnt = kBlock * kTileE;      // entries per workgroup

static inline unsigned long long mix(unsigned long long h) { h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 27; h *= 0x94D049BB133111EBull; h ^= h >> 31; return h; }

// ---------------------------------------------------------------- V0: the production pass (kernels_tiled.hip), verbatim, with ablation bits
// ABL bit0: no y read / write, bit1: gathers from an 8 KB window (L1), bit2: no gathers at all (x = 1)
template <int ABL>
__global__ __launch_bounds__(kBlock) void pass_v0(const double* __restrict__ x, double* __restrict__ y,
                                                  const double* __restrict__ tVals, const int* __restrict__ tCols, const int* __restrict__ tRows, int kBegin, int kEnd)
{
    __shared__ double s_p[kEnt];
    __shared__ int s_r[kEnt];
    const int blockBase = kBegin + (int)blockIdx.x * kEnt;
    const int blockCount = (kEnd - blockBase) < kEnt ? (kEnd - blockBase) : kEnt;
    double v[kTileE]; int c[kTileE], r[kTileE];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        const int k = blockBase + (j < blockCount ? j : 0);
        v[e] = tVals[k]; c[e] = tCols[k]; r[e] = tRows[k];
    }
    double xv[kTileE];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) xv[e] = (ABL & 4) ? 1.0 : x[(ABL & 2) ? (c[e] & 1023) : c[e]];
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
        if (prevRow == row) continue;
        double acc = (ABL & 1) ? 0.0 : y[row];
        int j = j0;
        while (j < blockCount && s_r[j] == row) { acc += s_p[j]; ++j; }
        if (j == blockCount) { for (int kk = blockBase + blockCount; kk < kEnd && tRows[kk] == row; ++kk) { const double q = tVals[kk] * x[tCols[kk]]; acc += q; } }
        if (!(ABL & 1) || acc == 1.2345e300) y[row] = acc;
    }
}

// ---------------------------------------------------------------- V1: 12-byte entries
// packed = column offset inside the tile (low `shift` bits) | row - baseRow of the workgroup's block (high bits); hdr[b] = { baseRow, row
// of the entry in front of the block or -1 }.  The first entry of a segment is found with a lane shuffle (no LDS round trip), y of the
// segment leaders is requested before the gathers.
// YEARLY: 1 = request y before the gathers
template <int YEARLY, int E = kTileE, bool NT = true, int YMODE = 0>
__global__ __launch_bounds__(kBlock) void pass_v1(const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ tVals, const unsigned* __restrict__ tPacked,
                                                  const int2* __restrict__ hdr, int hdrBase, int kBegin, int kEnd, int tileCol0, int shift)
{
    constexpr int ENT = kBlock * E;
    __shared__ double s_p[ENT];
    __shared__ int s_r[ENT + 1];
    const int blockBase = kBegin + (int)blockIdx.x * ENT;
    const int blockCount = (kEnd - blockBase) < ENT ? (kEnd - blockBase) : ENT;
    const int hb = hdrBase + (int)blockIdx.x * (ENT / kEnt);       // headers are per kEnt = 1024 entries
    const unsigned colMask = (1u << shift) - 1u;
    double v[E]; unsigned pk[E]; int r[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        const int k = blockBase + (j < blockCount ? j : 0);
        if (NT) { v[e] = __builtin_nontemporal_load(tVals + k); pk[e] = __builtin_nontemporal_load(tPacked + k); }
        else { v[e] = tVals[k]; pk[e] = tPacked[k]; }
    }
    bool lead[E]; double yv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        r[e] = hdr[hb + (j < blockCount ? j : 0) / kEnt].x + (int)(pk[e] >> shift);
        s_r[j + 1] = j < blockCount ? r[e] : -2;
    }
    if (threadIdx.x == 0) s_r[0] = hdr[hb].y;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const int j = e * kBlock + (int)threadIdx.x;
        lead[e] = j < blockCount && s_r[j] != r[e];
        yv[e] = 0.0;
        if (YEARLY == 1 && lead[e]) yv[e] = y[r[e]];
    }
    double xv[E];
#pragma unroll
    for (int e = 0; e < E; ++e) xv[e] = x[tileCol0 + (int)(pk[e] & colMask)];
    if (YEARLY == 2) {                                             // y requested BEHIND the gathers: its latency overlaps the products and the barrier
#pragma unroll
        for (int e = 0; e < E; ++e) if (lead[e]) yv[e] = y[r[e]];
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; s_p[j] = v[e] * xv[e]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) {
        if (!lead[e]) continue;
        const int j0 = e * kBlock + (int)threadIdx.x;
        const int row = r[e];
        double acc = YEARLY ? yv[e] : ((YMODE & 2) ? __builtin_nontemporal_load(y + row) : y[row]);
        int j = j0;
        while (j < blockCount && s_r[j + 1] == row) { acc += s_p[j]; ++j; }
        if (j == blockCount) {
            for (int kk = blockBase + blockCount; kk < kEnd; ++kk) {         // the segment runs on into the next block (rare)
                const int2 hn = hdr[hdrBase + (kk - kBegin) / kEnt];
                const unsigned p = tPacked[kk];
                if (hn.x + (int)(p >> shift) != row) break;
                const double q = tVals[kk] * x[tileCol0 + (int)(p & colMask)]; acc += q;
            }
        }
        if (YMODE & 1) __builtin_nontemporal_store(acc, y + row); else y[row] = acc;
    }
}

// ---------------------------------------------------------------- V1b: the production ordering (kernels_tiled.hip): one barrier, leaders found after it
__global__ __launch_bounds__(kBlock) void pass_v1b(const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ tVals, const unsigned* __restrict__ tPacked,
                                                   const int2* __restrict__ hdr, int hdrBase, int kBegin, int kEnd, int tileCol0, int shift, const int* __restrict__ doneFlag)
{
    __shared__ double s_p[kEnt];
    __shared__ int s_r[kEnt + 1];
    if (doneFlag != nullptr && *doneFlag != 0) return;
    const int blockBase = kBegin + (int)blockIdx.x * kEnt;
    const int blockCount = (kEnd - blockBase) < kEnt ? (kEnd - blockBase) : kEnt;
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
    double xv[kTileE];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) xv[e] = x[tileCol0 + (int)(pk[e] & colMask)];
#pragma unroll
    for (int e = 0; e < kTileE; ++e) { const int j = e * kBlock + (int)threadIdx.x; s_p[j] = v[e] * xv[e]; }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < kTileE; ++e) {
        const int j0 = e * kBlock + (int)threadIdx.x;
        if (j0 >= blockCount || s_r[j0] == r[e]) continue;
        const int row = r[e];
        double acc = y[row];
        int j = j0;
        while (j < blockCount && s_r[j + 1] == row) { acc += s_p[j]; ++j; }
        if (j == blockCount) {
            for (int kk = blockBase + blockCount; kk < kEnd; ++kk) {
                const int2 hn = hdr[hdrBase + (kk - kBegin) / kEnt];
                const unsigned p = tPacked[kk];
                if (hn.x + (int)(p >> shift) != row) break;
                const double q = tVals[kk] * x[tileCol0 + (int)(p & colMask)]; acc += q;
            }
        }
        y[row] = acc;
    }
}

// ---------------------------------------------------------------- V2: V1 as a persistent loop with the next block's entries prefetched
// Every workgroup walks blocks b = blockIdx.x, + gridDim.x, ...: the entry loads of the NEXT block are issued behind the gathers of the
// current one (three kinds of traffic of one workgroup in flight together: entry streams from HBM, gathers from L2, y).
template <int E>
__global__ __launch_bounds__(kBlock) void pass_v2(const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ tVals, const unsigned* __restrict__ tPacked,
                                                  const int2* __restrict__ hdr, int hdrBase, int kBegin, int kEnd, int tileCol0, int shift, int nBlocks)
{
    constexpr int ENT = kBlock * E;
    __shared__ double s_p[ENT];
    __shared__ int s_r[ENT + 1];
    const unsigned colMask = (1u << shift) - 1u;
    const int kLast = kEnd - 1;
    double v[E], vn[E]; unsigned pk[E], pkn[E]; int r[E];
    auto load = [&](int b, double* vv, unsigned* pp) {
        const int base = kBegin + b * ENT;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            int k = base + e * kBlock + (int)threadIdx.x;
            k = k < kLast ? k : kLast;                               // clamped: unconditional loads
            vv[e] = __builtin_nontemporal_load(tVals + k); pp[e] = __builtin_nontemporal_load(tPacked + k);
        }
    };
    int b = blockIdx.x;
    if (b >= nBlocks) return;
    load(b, v, pk);
    for (; b < nBlocks; b += gridDim.x) {
        const int blockBase = kBegin + b * ENT;
        const int blockCount = (kEnd - blockBase) < ENT ? (kEnd - blockBase) : ENT;
        const int2 h = hdr[hdrBase + b];
        bool lead[E];
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int j = e * kBlock + (int)threadIdx.x;
            r[e] = h.x + (int)(pk[e] >> shift);
            s_r[j + 1] = j < blockCount ? r[e] : -2;
        }
        if (threadIdx.x == 0) s_r[0] = h.y;
        double xv[E];
#pragma unroll
        for (int e = 0; e < E; ++e) xv[e] = x[tileCol0 + (int)(pk[e] & colMask)];
        const int bn = b + (int)gridDim.x < nBlocks ? b + (int)gridDim.x : b;
        load(bn, vn, pkn);                                           // prefetch behind the gathers
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; lead[e] = j < blockCount && s_r[j] != r[e]; }
#pragma unroll
        for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; s_p[j] = v[e] * xv[e]; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (!lead[e]) continue;
            const int j0 = e * kBlock + (int)threadIdx.x;
            const int row = r[e];
            double acc = y[row];
            int j = j0;
            while (j < blockCount && s_r[j + 1] == row) { acc += s_p[j]; ++j; }
            if (j == blockCount) {
                for (int kk = blockBase + blockCount; kk < kEnd; ++kk) {
                    const int2 hn = hdr[hdrBase + (kk - kBegin) / ENT];
                    const unsigned p = tPacked[kk];
                    if (hn.x + (int)(p >> shift) != row) break;
                    const double q = tVals[kk] * x[tileCol0 + (int)(p & colMask)]; acc += q;
                }
            }
            y[row] = acc;
        }
        __syncthreads();                                             // s_r / s_p are rewritten by the next trip
#pragma unroll
        for (int e = 0; e < E; ++e) { v[e] = vn[e]; pk[e] = pkn[e]; }
    }
}

// ---------------------------------------------------------------- V3: row-block-persistent accumulators
// The layout is turned around: row BLOCK major (RB rows), column tile major inside a block, row major inside a (block, tile) run.  A
// workgroup owns a row block for the whole product: its running row sums live in LDS while it walks the block's runs tile by tile -- every
// workgroup of the chip starts with tile 0, so an XCD's workgroups read the same x window at about the same time -- and y is written ONCE at
// the end (no read-modify-write of y per (row, tile) segment: 4.3 GB of the 9.7 GB a product moves in the tile-major form; the epilogue of
// the solver's fused forms could ride on that store).  Rows are still summed tile by tile, columns ascending inside a tile: stored order.
// packed = local row inside the block (bits >= shift) | column offset inside the tile; seg[b * (T + 1) + t] = first entry of run (b, t).
// ABL bit1: gathers from an 8 KB window (L1), bit2: no gathers (x = 1)   (timing ablations: wrong results by design)
template <int RB, int E, int ABL = 0>
__global__ __launch_bounds__(kBlock) void pass_v3(const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ tVals, const unsigned* __restrict__ tPacked,
                                                  const int* __restrict__ seg, int nBlocks, int T, int width, int shift, long long rows)
{
    constexpr int CH = kBlock * E;                                   // entries per chunk
    __shared__ double s_acc[RB];
    __shared__ double s_p[CH];
    __shared__ int s_r[CH + 1];
    const unsigned colMask = (1u << shift) - 1u;
    for (int b = blockIdx.x; b < nBlocks; b += gridDim.x) {
        for (int i = threadIdx.x; i < RB; i += kBlock) s_acc[i] = 0.0;
        const int* sg = seg + (long long)b * (T + 1);
        for (int t = 0; t < T; ++t) {
            const int kb = sg[t], ke = sg[t + 1];
            const int col0 = t * width;
            for (int base = kb; base < ke; base += CH) {
                const int cnt = (ke - base) < CH ? (ke - base) : CH;
                double v[E]; unsigned pk[E]; int r[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int j = e * kBlock + (int)threadIdx.x;
                    const int k = base + (j < cnt ? j : 0);
                    v[e] = __builtin_nontemporal_load(tVals + k); pk[e] = __builtin_nontemporal_load(tPacked + k);
                }
                __syncthreads();                                     // (the leaders of the previous chunk are done with s_p / s_r; s_acc zeroed)
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; r[e] = (int)(pk[e] >> shift); s_r[j + 1] = j < cnt ? r[e] : -2; }
                if (threadIdx.x == 0) s_r[0] = -1;                   // the first entry of a chunk always leads: the row's sum so far is in s_acc
                __syncthreads();
                bool lead[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; lead[e] = j < cnt && s_r[j] != r[e]; }
                double xv[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (ABL & 4) xv[e] = 1.0;
                    else if (ABL & 2) xv[e] = x[(pk[e] & colMask) & 1023u];
                    else xv[e] = x[col0 + (int)(pk[e] & colMask)];
                }
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; s_p[j] = v[e] * xv[e]; }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (!lead[e]) continue;
                    const int row = r[e];
                    double acc = s_acc[row];
                    int j = e * kBlock + (int)threadIdx.x;
                    while (j < cnt && s_r[j + 1] == row) { acc += s_p[j]; ++j; }
                    s_acc[row] = acc;
                }
            }
        }
        __syncthreads();
        const long long row0 = (long long)b * RB;
        for (int i = threadIdx.x; i < RB; i += kBlock) if (row0 + i < rows) __builtin_nontemporal_store(s_acc[i], y + row0 + i);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- V5: V3 with a SOFT lockstep of an XCD's workgroups over the tiles
// V3 loses because nothing keeps the workgroups of an XCD on one x window (counters: L2 hit rate 51 %).  Here a workgroup that is about to start
// step k = (round, tile) first looks at how many workgroups of ITS XCD (blockIdx.x & 7: round-robin dispatch) have finished step k - 1 - SLACK
// and sleeps while fewer than `need` have -- at most kMaxNaps short naps, then it goes on regardless: a best-effort pacing that can never
// deadlock (no workgroup ever depends on another one being resident).  progress[xcd * steps + k] counts finished workgroups (zeroed per launch).
constexpr int kMaxNaps = 4000;
template <int RB, int E, int SLACK>
__global__ __launch_bounds__(kBlock) void pass_v5(const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ tVals, const unsigned* __restrict__ tPacked,
                                                  const int* __restrict__ seg, int nBlocks, int T, int width, int shift, long long rows, unsigned* __restrict__ progress, int needPerMille)
{
    constexpr int CH = kBlock * E;
    __shared__ double s_acc[RB];
    __shared__ double s_p[CH];
    __shared__ int s_r[CH + 1];
    const unsigned colMask = (1u << shift) - 1u;
    const int xcd = blockIdx.x & 7;
    const int rounds = (nBlocks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int steps = rounds * T;
    unsigned* prog = progress + (long long)xcd * steps;
    int round = 0;
    for (int b = blockIdx.x; b < nBlocks; b += gridDim.x, ++round) {
        for (int i = threadIdx.x; i < RB; i += kBlock) s_acc[i] = 0.0;
        const int* sg = seg + (long long)b * (T + 1);
        // workgroups of my XCD that take part in this round
        const int lastFull = nBlocks - round * (int)gridDim.x;                         // blocks left at the start of this round
        const int active = lastFull >= (int)gridDim.x ? (int)gridDim.x / 8 : (lastFull - xcd + 7) / 8;
        const unsigned need = (unsigned)((long long)active * needPerMille / 1000);
        for (int t = 0; t < T; ++t) {
            const int k = round * T + t;
            if (k - 1 - SLACK >= 0 && threadIdx.x == 0) {
                int naps = 0;
                while (__hip_atomic_load(prog + (k - 1 - SLACK), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need && naps < kMaxNaps) { __builtin_amdgcn_s_sleep(8); ++naps; }
            }
            __syncthreads();
            const int kb = sg[t], ke = sg[t + 1];
            const int col0 = t * width;
            for (int base = kb; base < ke; base += CH) {
                const int cnt = (ke - base) < CH ? (ke - base) : CH;
                double v[E]; unsigned pk[E]; int r[E];
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    const int j = e * kBlock + (int)threadIdx.x;
                    const int kk = base + (j < cnt ? j : 0);
                    v[e] = __builtin_nontemporal_load(tVals + kk); pk[e] = __builtin_nontemporal_load(tPacked + kk);
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; r[e] = (int)(pk[e] >> shift); s_r[j + 1] = j < cnt ? r[e] : -2; }
                if (threadIdx.x == 0) s_r[0] = -1;
                __syncthreads();
                bool lead[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; lead[e] = j < cnt && s_r[j] != r[e]; }
                double xv[E];
#pragma unroll
                for (int e = 0; e < E; ++e) xv[e] = x[col0 + (int)(pk[e] & colMask)];
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = e * kBlock + (int)threadIdx.x; s_p[j] = v[e] * xv[e]; }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    if (!lead[e]) continue;
                    const int row = r[e];
                    double acc = s_acc[row];
                    int j = e * kBlock + (int)threadIdx.x;
                    while (j < cnt && s_r[j + 1] == row) { acc += s_p[j]; ++j; }
                    s_acc[row] = acc;
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) atomicAdd(prog + k, 1u);                             // (a pacing hint, not a synchronisation: no data depends on it)
        }
        const long long row0 = (long long)b * RB;
        for (int i = threadIdx.x; i < RB; i += kBlock) if (row0 + i < rows) __builtin_nontemporal_store(s_acc[i], y + row0 + i);
        __syncthreads();
    }
}

// ---------------------------------------------------------------- V4: V3 with TH threads per workgroup and the next chunk's entries prefetched
// (the loads of chunk n + 1 are issued before the gathers of chunk n: entry streams from HBM and gathers from L2 in flight together)
template <int RB, int E, int TH, int ABL = 0>
__global__ __launch_bounds__(TH) void pass_v4(const double* __restrict__ x, double* __restrict__ y, const double* __restrict__ tVals, const unsigned* __restrict__ tPacked,
                                              const int* __restrict__ seg, int nBlocks, int T, int width, int shift, long long rows)
{
    constexpr int CH = TH * E;
    __shared__ double s_acc[RB];
    __shared__ double s_p[CH];
    __shared__ int s_r[CH + 1];
    const unsigned colMask = (1u << shift) - 1u;
    const int tid = (int)threadIdx.x;
    for (int b = blockIdx.x; b < nBlocks; b += gridDim.x) {
        for (int i = tid; i < RB; i += TH) s_acc[i] = 0.0;
        const int* sg = seg + (long long)b * (T + 1);
        // chunk cursor: tile t, entries [base, base + cnt) of run (b, t); advance() steps to the next non-empty chunk
        int t = 0, base = sg[0], end = sg[1];
        auto settle = [&]() { while (base >= end && t + 1 < T) { ++t; base = sg[t]; end = sg[t + 1]; } };
        settle();
        double v[E]; unsigned pk[E];
        int cnt = (end - base) < CH ? (end - base) : CH;
        int col0 = t * width;
        bool have = base < end;
        if (have) {
#pragma unroll
            for (int e = 0; e < E; ++e) { const int j = e * TH + tid; const int k = base + (j < cnt ? j : 0); v[e] = __builtin_nontemporal_load(tVals + k); pk[e] = __builtin_nontemporal_load(tPacked + k); }
        }
        while (have) {
            // next chunk
            int t2 = t, base2 = base + cnt, end2 = end;
            while (base2 >= end2 && t2 + 1 < T) { ++t2; base2 = sg[t2]; end2 = sg[t2 + 1]; }
            const bool have2 = base2 < end2;
            const int cnt2 = have2 ? ((end2 - base2) < CH ? (end2 - base2) : CH) : 0;
            double v2[E]; unsigned pk2[E];
            if (have2) {
#pragma unroll
                for (int e = 0; e < E; ++e) { const int j = e * TH + tid; const int k = base2 + (j < cnt2 ? j : 0); v2[e] = __builtin_nontemporal_load(tVals + k); pk2[e] = __builtin_nontemporal_load(tPacked + k); }
            }
            int r[E];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) { const int j = e * TH + tid; r[e] = (int)(pk[e] >> shift); s_r[j + 1] = j < cnt ? r[e] : -2; }
            if (tid == 0) s_r[0] = -1;
            __syncthreads();
            bool lead[E];
#pragma unroll
            for (int e = 0; e < E; ++e) { const int j = e * TH + tid; lead[e] = j < cnt && s_r[j] != r[e]; }
            double xv[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (ABL & 4) xv[e] = 1.0;
                else if (ABL & 2) xv[e] = x[(pk[e] & colMask) & 1023u];
                else xv[e] = x[col0 + (int)(pk[e] & colMask)];
            }
#pragma unroll
            for (int e = 0; e < E; ++e) { const int j = e * TH + tid; s_p[j] = v[e] * xv[e]; }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (!lead[e]) continue;
                const int row = r[e];
                double acc = s_acc[row];
                int j = e * TH + tid;
                while (j < cnt && s_r[j + 1] == row) { acc += s_p[j]; ++j; }
                s_acc[row] = acc;
            }
            have = have2; t = t2; base = base2; end = end2; cnt = cnt2; col0 = t * width;
#pragma unroll
            for (int e = 0; e < E; ++e) { v[e] = v2[e]; pk[e] = pk2[e]; }
        }
        __syncthreads();
        const long long row0 = (long long)b * RB;
        for (int i = tid; i < RB; i += TH) if (row0 + i < rows) __builtin_nontemporal_store(s_acc[i], y + row0 + i);
        __syncthreads();
    }
}

__global__ void zero_kernel(double* y, long long n) { for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = 0.0; }

template <typename F> static double time_ms(F f, int reps = 5)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    const long long rows = argc > 1 ? atoll(argv[1]) : 10000000LL;
    const int T = argc > 2 ? atoi(argv[2]) : 20;
    const int shift = argc > 3 ? atoi(argv[3]) : 19;
    const double mean = argc > 4 ? atof(argv[4]) : 1.55;
    const int width = argc > 5 ? atoi(argv[5]) : (1 << shift);          // columns per tile (<= 2^shift: the packed word keeps `shift` bits for the offset)
    if (width < 1 || width > (1 << shift)) { printf("tileWidth must be in [1, 2^tileShift]\n"); return 1; }
    const bool quick = getenv("TILE_LAB_QUICK") != nullptr;
    const long long cols = (long long)T * width;
    printf("tile_lab: %lld rows, %d tiles of %d columns = %.2f MiB of x each (x = %.1f MB), mean %.2f entries per (row, tile)\n", rows, T, width, width * 8 / 1048576.0, cols * 8 / 1048576.0, mean);
    // ---- host generation, tile-major, row-major inside a tile, ascending columns inside a (row, tile) segment
    std::vector<double> hv; std::vector<int> hc, hr; std::vector<int> tileStart((size_t)T + 1, 0);
    hv.reserve((size_t)(rows * T * mean * 1.05)); hc.reserve(hv.capacity()); hr.reserve(hv.capacity());
    const double pz = exp(-mean);
    for (int t = 0; t < T; ++t) {
        tileStart[(size_t)t] = (int)hv.size();
        for (long long i = 0; i < rows; ++i) {
            unsigned long long h = mix((unsigned long long)t * 0x9E3779B97F4A7C15ull + (unsigned long long)i);
            double u = (double)(h >> 11) * (1.0 / 9007199254740992.0), p = pz, cum = pz;
            int cnt = 0;
            while (u > cum && cnt < 12) { ++cnt; p *= mean / cnt; cum += p; }   // Poisson(mean)
            int cs[12];
            for (int q = 0; q < cnt; ++q) { h = mix(h + 0x632BE59BD9B4E019ull); cs[q] = (int)(h % (unsigned long long)width); }
            std::sort(cs, cs + cnt);
            for (int q = 0; q < cnt; ++q) {
                if (q > 0 && cs[q] == cs[q - 1]) continue;
                h = mix(h + 1);
                hv.push_back(-((double)(h >> 11) * (1.0 / 9007199254740992.0))); hc.push_back(t * width + cs[q]); hr.push_back((int)i);
            }
        }
    }
    tileStart[(size_t)T] = (int)hv.size();
    const long long nnz = (long long)hv.size();
    printf("nnz %lld (%.2f per row)\n", nnz, (double)nnz / rows);
    std::vector<double> hx((size_t)cols);
    for (long long i = 0; i < cols; ++i) hx[(size_t)i] = cos((double)i * 0.01);
    // packed form: fixed blocks of kEnt entries per tile
    std::vector<unsigned> hp((size_t)nnz); std::vector<int2> hh; std::vector<int> hdrBase((size_t)T, 0);
    long long overflow = 0;
    for (int t = 0; t < T; ++t) {
        hdrBase[(size_t)t] = (int)hh.size();
        for (int kb = tileStart[(size_t)t]; kb < tileStart[(size_t)t + 1]; kb += kEnt) {
            const int ke = std::min(kb + kEnt, tileStart[(size_t)t + 1]);
            const int tileCol0 = t * width;
            int2 h; h.x = hr[(size_t)kb]; h.y = kb > tileStart[(size_t)t] ? hr[(size_t)kb - 1] : -1;
            hh.push_back(h);
            for (int k = kb; k < ke; ++k) {
                const int lr = hr[(size_t)k] - h.x;
                if (lr >= (1 << (32 - shift))) ++overflow;
                hp[(size_t)k] = ((unsigned)lr << shift) | (unsigned)(hc[(size_t)k] - tileCol0);
            }
        }
    }
    printf("blocks %zu, local-row overflows %lld\n", hh.size(), overflow);
    // host reference (row sums in tile order = stored order)
    std::vector<double> yref((size_t)rows, 0.0);
    for (long long k = 0; k < nnz; ++k) { const double p = hv[(size_t)k] * hx[(size_t)hc[(size_t)k]]; yref[(size_t)hr[(size_t)k]] += p; }

    double *dv, *dx, *dy; int *dc, *dr; unsigned* dp; int2* dh;
    CK(hipMalloc(&dv, nnz * 8)); CK(hipMalloc(&dc, nnz * 4)); CK(hipMalloc(&dr, nnz * 4)); CK(hipMalloc(&dp, nnz * 4)); CK(hipMalloc(&dh, hh.size() * sizeof(int2)));
    CK(hipMalloc(&dx, cols * 8)); CK(hipMalloc(&dy, rows * 8));
    CK(hipMemcpy(dv, hv.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, hc.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dr, hr.data(), nnz * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dp, hp.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dh, hh.data(), hh.size() * sizeof(int2), hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, hx.data(), cols * 8, hipMemcpyHostToDevice));
    std::vector<double> got((size_t)rows);
    auto check = [&](const char* name, bool expectExact) {
        CK(hipMemcpy(got.data(), dy, rows * 8, hipMemcpyDeviceToHost));
        long long bad = 0;
        for (long long i = 0; i < rows; ++i) if (memcmp(&got[(size_t)i], &yref[(size_t)i], 8) != 0) ++bad;
        printf("   %-46s %s (%lld rows differ)\n", name, bad == 0 ? "bit-identical to the host sums" : (expectExact ? "MISMATCH" : "differs (ablation: expected)"), bad);
    };
#define PASSES(...) [&] { hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, 0, dy, rows); \
        for (int t = 0; t < T; ++t) { const int kb = tileStart[(size_t)t], ke = tileStart[(size_t)t + 1]; if (ke <= kb) continue; const dim3 g((ke - kb + kEnt - 1) / kEnt); __VA_ARGS__; } }
#define V0(ABL) PASSES(pass_v0<ABL><<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dc, dr, kb, ke))
#define V1(YE) PASSES(pass_v1<YE><<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift))
    struct Row { const char* name; double ms; };
    std::vector<Row> res;
    auto run = [&](const char* name, auto f, bool exact) { const double ms = time_ms(f); res.push_back({ name, ms }); printf("%-48s %7.3f ms per product\n", name, ms); check(name, exact); fflush(stdout); };
    if (quick) {
        run("V1b production ordering (12 B entries)", PASSES(pass_v1b<<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift, (const int*)nullptr)), true);
        return 0;
    }
    // ---- V3 data: block-major, tile-major inside a block (regenerated from the same hashes in that order)
    auto build_v3 = [&](int RB, std::vector<double>& v3v, std::vector<unsigned>& v3p, std::vector<int>& v3s) {
        const int nB = (int)((rows + RB - 1) / RB);
        v3v.clear(); v3p.clear(); v3s.assign((size_t)nB * (T + 1), 0);
        v3v.reserve((size_t)nnz); v3p.reserve((size_t)nnz);
        std::vector<int> cursor((size_t)T);                          // position inside tile t of the tile-major arrays (rows ascend there)
        for (int t = 0; t < T; ++t) cursor[(size_t)t] = tileStart[(size_t)t];
        for (int b = 0; b < nB; ++b) {
            const long long r0 = (long long)b * RB, r1 = std::min(rows, r0 + RB);
            for (int t = 0; t < T; ++t) {
                v3s[(size_t)b * (T + 1) + t] = (int)v3v.size();
                int& k = cursor[(size_t)t];
                while (k < tileStart[(size_t)t + 1] && hr[(size_t)k] < r1) {
                    v3v.push_back(hv[(size_t)k]);
                    v3p.push_back(((unsigned)(hr[(size_t)k] - r0) << shift) | (unsigned)(hc[(size_t)k] - t * width));
                    ++k;
                }
            }
            v3s[(size_t)b * (T + 1) + T] = (int)v3v.size();
        }
        return nB;
    };
    auto run_v3 = [&](auto kernel, const char* name, int RB, int wgs, bool exact) {
        if ((long long)RB << shift > (1LL << 32)) { printf("%s: RB does not fit the packed word\n", name); return; }
        std::vector<double> v3v; std::vector<unsigned> v3p; std::vector<int> v3s;
        const int nB = build_v3(RB, v3v, v3p, v3s);
        double* d3v; unsigned* d3p; int* d3s;
        CK(hipMalloc(&d3v, nnz * 8)); CK(hipMalloc(&d3p, nnz * 4)); CK(hipMalloc(&d3s, v3s.size() * 4));
        CK(hipMemcpy(d3v, v3v.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d3p, v3p.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d3s, v3s.data(), v3s.size() * 4, hipMemcpyHostToDevice));
        const int g = wgs < nB ? wgs : nB;
        run(name, [&] { kernel<<<dim3(g), dim3(kBlock), 0, 0>>>(dx, dy, d3v, d3p, d3s, nB, T, width, shift, rows); }, exact);
        CK(hipFree(d3v)); CK(hipFree(d3p)); CK(hipFree(d3s));
    };
    auto run_v4 = [&](auto kernel, const char* name, int RB, int TH, int wgs, bool exact) {
        std::vector<double> v3v; std::vector<unsigned> v3p; std::vector<int> v3s;
        const int nB = build_v3(RB, v3v, v3p, v3s);
        double* d3v; unsigned* d3p; int* d3s;
        CK(hipMalloc(&d3v, nnz * 8)); CK(hipMalloc(&d3p, nnz * 4)); CK(hipMalloc(&d3s, v3s.size() * 4));
        CK(hipMemcpy(d3v, v3v.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d3p, v3p.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d3s, v3s.data(), v3s.size() * 4, hipMemcpyHostToDevice));
        const int g = wgs < nB ? wgs : nB;
        run(name, [&] { kernel<<<dim3(g), dim3(TH), 0, 0>>>(dx, dy, d3v, d3p, d3s, nB, T, width, shift, rows); }, exact);
        CK(hipFree(d3v)); CK(hipFree(d3p)); CK(hipFree(d3s));
    };
    if (getenv("TILE_LAB_V5")) {       // V3 with the soft lockstep of an XCD's workgroups
        run("V1b production ordering (12 B entries)", PASSES(pass_v1b<<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift, (const int*)nullptr)), true);
        auto run_v5 = [&](auto kernel, const char* name, int RB, int wgs, int perMille) {
            std::vector<double> v3v; std::vector<unsigned> v3p; std::vector<int> v3s;
            const int nB = build_v3(RB, v3v, v3p, v3s);
            double* d3v; unsigned* d3p; int* d3s; unsigned* dprog;
            const int g = (wgs < nB ? wgs : nB) & ~7;
            const int rounds = (nB + g - 1) / g;
            const size_t progN = (size_t)8 * rounds * T;
            CK(hipMalloc(&d3v, nnz * 8)); CK(hipMalloc(&d3p, nnz * 4)); CK(hipMalloc(&d3s, v3s.size() * 4)); CK(hipMalloc(&dprog, progN * 4));
            CK(hipMemcpy(d3v, v3v.data(), nnz * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d3p, v3p.data(), nnz * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d3s, v3s.data(), v3s.size() * 4, hipMemcpyHostToDevice));
            run(name, [&] { CK(hipMemsetAsync(dprog, 0, progN * 4, 0)); kernel<<<dim3(g), dim3(kBlock), 0, 0>>>(dx, dy, d3v, d3p, d3s, nB, T, width, shift, rows, dprog, perMille); }, true);
            CK(hipFree(d3v)); CK(hipFree(d3p)); CK(hipFree(d3s)); CK(hipFree(dprog));
        };
        run_v5(pass_v5<4096, 8, 0>, "V5 RB 4096 / 2048, 512 wgs, same tile, all of the XCD", 4096, 512, 1000);
        run_v5(pass_v5<4096, 8, 0>, "V5 RB 4096 / 2048, 512 wgs, same tile, 3/4 of the XCD", 4096, 512, 750);
        run_v5(pass_v5<4096, 8, 1>, "V5 RB 4096 / 2048, 512 wgs, one tile of slack, all", 4096, 512, 1000);
        run_v5(pass_v5<4096, 8, 0>, "V5 RB 4096 / 2048, 768 wgs, same tile, all", 4096, 768, 1000);
        run_v5(pass_v5<2048, 4, 0>, "V5 RB 2048 / 1024, 1280 wgs, same tile, all", 2048, 1280, 1000);
        run_v5(pass_v5<8192, 4, 0>, "V5 RB 8192 / 1024, 512 wgs, same tile, all", 8192, 512, 1000);
        return 0;
    }
    if (getenv("TILE_LAB_V3_ONE")) {   // the one row-block-persistent variant, for counter passes
        run_v3(pass_v3<4096, 8>, "V3 row blocks of 4096, 2048-entry chunks, 512 wgs", 4096, 512, true);
        return 0;
    }
    if (getenv("TILE_LAB_V4")) {
        run("V1b production ordering (12 B entries)", PASSES(pass_v1b<<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift, (const int*)nullptr)), true);
        run_v4(pass_v4<4096, 8, 256>, "V4 RB 4096, 256 threads x 8, 512 wgs", 4096, 256, 512, true);
        run_v4(pass_v4<4096, 4, 512>, "V4 RB 4096, 512 threads x 4, 512 wgs", 4096, 512, 512, true);
        run_v4(pass_v4<8192, 4, 512>, "V4 RB 8192, 512 threads x 4, 256 wgs", 8192, 512, 256, true);
        run_v4(pass_v4<8192, 4, 1024>, "V4 RB 8192, 1024 threads x 4, 256 wgs", 8192, 1024, 256, true);
        run_v4(pass_v4<8192, 2, 1024>, "V4 RB 8192, 1024 threads x 2, 256 wgs", 8192, 1024, 256, true);
        run_v4(pass_v4<4096, 4, 1024>, "V4 RB 4096, 1024 threads x 4, 512 wgs", 4096, 1024, 512, true);
        run_v4(pass_v4<4096, 8, 256, 4>, "V4 RB 4096, 256 x 8: no gathers (x = 1)", 4096, 256, 512, false);
        run_v4(pass_v4<8192, 4, 1024, 4>, "V4 RB 8192, 1024 x 4: no gathers (x = 1)", 8192, 1024, 256, false);
        return 0;
    }
    if (getenv("TILE_LAB_V3")) {       // only the row-block-persistent variants (and the production pass next to them)
        run("V1b production ordering (12 B entries)", PASSES(pass_v1b<<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift, (const int*)nullptr)), true);
        run_v3(pass_v3<2048, 4>, "V3 row blocks of 2048, 1024-entry chunks, 1280 wgs", 2048, 1280, true);
        run_v3(pass_v3<2048, 4>, "V3 row blocks of 2048, 1024-entry chunks, 2560 wgs", 2048, 2560, true);
        run_v3(pass_v3<4096, 4>, "V3 row blocks of 4096, 1024-entry chunks, 768 wgs", 4096, 768, true);
        run_v3(pass_v3<4096, 8>, "V3 row blocks of 4096, 2048-entry chunks, 512 wgs", 4096, 512, true);
        run_v3(pass_v3<2048, 8>, "V3 row blocks of 2048, 2048-entry chunks, 1024 wgs", 2048, 1024, true);
        run_v3(pass_v3<8192, 4>, "V3 row blocks of 8192, 1024-entry chunks, 512 wgs", 8192, 512, true);
        run_v3(pass_v3<2048, 4, 2>, "V3 2048 / 1024: gathers from an 8 KB window (L1)", 2048, 1280, false);
        run_v3(pass_v3<2048, 4, 4>, "V3 2048 / 1024: no gathers (x = 1)", 2048, 1280, false);
        return 0;
    }
    run("V0 production pass (16 B entries)", V0(0), true);
    run("V0 without the y read-modify-write", V0(1), false);
    run("V0 gathers from an 8 KB window (L1)", V0(2), false);
    run("V0 no gathers (x = 1)", V0(4), false);
    run("V0 no gathers, no y", V0(5), false);
    run("V1 12 B entries, y after the gathers", V1(0), true);
    run("V1 12 B entries, y requested before the gathers", V1(1), true);
#define V2(E, G) [&] { hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, 0, dy, rows); \
        for (int t = 0; t < T; ++t) { const int kb = tileStart[(size_t)t], ke = tileStart[(size_t)t + 1]; if (ke <= kb) continue; const int nb = (ke - kb + kBlock * E - 1) / (kBlock * E); \
            pass_v2<E><<<dim3(nb < (G) ? nb : (G)), dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift, nb); } }
    run("V1 12 B entries, y requested behind the gathers", V1(2), true);
    run("V1b production ordering (one barrier)", PASSES(pass_v1b<<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift, (const int*)nullptr)), true);
    {   // the same with 8 GB of other allocations alive (the product holds the CSR arrays next to the tiled copy)
        void* extra = nullptr; CK(hipMalloc(&extra, 8ull << 30)); CK(hipMemset(extra, 1, 8ull << 30));
        run("V1b with 8 GB of other allocations alive", PASSES(pass_v1b<<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift, (const int*)nullptr)), true);
        CK(hipFree(extra));
    }
    run("V2 persistent, next block prefetched, 2048 wgs", V2(4, 2048), true);
#define V1E(E, NT) [&] { hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, 0, dy, rows); \
        for (int t = 0; t < T; ++t) { const int kb = tileStart[(size_t)t], ke = tileStart[(size_t)t + 1]; if (ke <= kb) continue; const dim3 g((ke - kb + kBlock * E - 1) / (kBlock * E)); \
            pass_v1<0, E, NT><<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift); } }
#define V1Y(YM) [&] { hipLaunchKernelGGL(zero_kernel, dim3(2048), dim3(256), 0, 0, dy, rows); \
        for (int t = 0; t < T; ++t) { const int kb = tileStart[(size_t)t], ke = tileStart[(size_t)t + 1]; if (ke <= kb) continue; const dim3 g((ke - kb + kEnt - 1) / kEnt); \
            pass_v1<0, 4, true, YM><<<g, dim3(kBlock), 0, 0>>>(dx, dy, dv, dp, dh, hdrBase[(size_t)t], kb, ke, t * width, shift); } }
    run("V1 12 B, y stored non-temporally", V1Y(1), true);
    run("V1 12 B, y loaded non-temporally", V1Y(2), true);
    run("V1 12 B, y loaded and stored non-temporally", V1Y(3), true);
    run("V1 12 B, 4 entries per thread, plain loads", V1E(4, false), true);
    run("V1 12 B, 8 entries per thread", V1E(8, true), true);
    run("V1 12 B, 8 entries per thread, plain loads", V1E(8, false), true);
    run("V1 12 B, 16 entries per thread", V1E(16, true), true);
    printf("streams: V0 %.2f GB, V1 %.2f GB of entries; y traffic per product ~ %.2f GB; gathers %lld\n", nnz * 16 / 1e9, nnz * 12 / 1e9, 0.0, nnz);
    return 0;
}
