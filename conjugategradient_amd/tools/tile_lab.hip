// tile_lab: the column-tile pass of the opt-in class-4 form (csrc/kernels_tiled.hip) taken apart on a synthetic matrix of BASELINE
// config 5's shape (10 M rows, ~31 entries per row, columns uniform over the whole range): which part of the 3.3 ms per product is the
// gathers, which the y read-modify-write, which the streams -- and what a 12-byte entry (value + one packed word) buys.  Measurement
// tool, not part of the library.  Every variant is checked bit for bit against the production-shaped pass (V0) and V0 against the host.
//   tile_lab [rows=10000000] [tiles=20] [tileShift=19] [meanPerCell=1.55] [tileWidth=2^tileShift columns; any width <= 2^tileShift]
//   TILE_LAB_QUICK=1: only the 12-byte production pass (for PMC runs)
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
