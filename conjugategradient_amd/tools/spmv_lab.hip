// spmv_lab: design-space probe for the short-row CSR SpMV on gfx950 (measurement tool, not product).
// Builds the 7-point Poisson n^3 CSR matrix on the device, checks every variant bit for bit against a
// thread-per-row reference and times it.  Build: make -C conjugategradient_amd/csrc spmvlab
//   spmv_lab [n=512] [reps=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------ matrix
__global__ void fill_poisson(int n, const int* __restrict__ ro, int* __restrict__ col, double* __restrict__ val)
{
    const long long N = (long long)n * n * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % n), y = (int)((i / n) % n), z = (int)(i / ((long long)n * n));
        int k = ro[i];
        if (z > 0) { col[k] = (int)(i - (long long)n * n); val[k++] = -1.0; }
        if (y > 0) { col[k] = (int)(i - n); val[k++] = -1.0; }
        if (x > 0) { col[k] = (int)(i - 1); val[k++] = -1.0; }
        col[k] = (int)i; val[k++] = 6.0;
        if (x < n - 1) { col[k] = (int)(i + 1); val[k++] = -1.0; }
        if (y < n - 1) { col[k] = (int)(i + n); val[k++] = -1.0; }
        if (z < n - 1) { col[k] = (int)(i + (long long)n * n); val[k++] = -1.0; }
    }
}
__global__ void fill_x(double* x, long long N)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        x[i] = (double)(h & 0xFFFFF) / 1048576.0 - 0.5;
    }
}
__global__ void ref_spmv(const int* __restrict__ ro, const int* __restrict__ col, const double* __restrict__ val, const double* __restrict__ x, double* __restrict__ y, long long N)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x) {
        double acc = 0.0;
        for (int k = ro[i]; k < ro[i + 1]; ++k) { const double p = val[k] * x[col[k]]; acc += p; }
        y[i] = acc;
    }
}
__global__ void cmp_kernel(const double* a, const double* b, long long N, unsigned long long* bad)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (long long)gridDim.x * blockDim.x)
        if (__double_as_longlong(a[i]) != __double_as_longlong(b[i])) atomicAdd(bad, 1ull);
}

struct Args {
    const int* ro; const int* col; const double* val; const double* x; const double* w; double* y; double* partials;
    int rows; int nnz; int nBlocks; int gmask;
};

// ------------------------------------------------------------------ V: lane = row, one wave per workgroup
// Per trip (64 rows): [top] raw(t) registers -> LDS; LDS reads of my row; 8 gathers; w; raw(t+1); ro(t+2); wait gathers;
// products; y store.  Every VMEM op of the loop body is unconditional, so every wait is a counted vmcnt.
//   NT    : non-temporal loads of the matrix streams
//   STORE : 0 none, 1 plain after the products (youngest op of the trip), 2 non-temporal, 3 plain but deferred one trip and
//           issued in front of the gathers, 4 deferred one trip, unconditional, issued between the gathers and raw(t+1)
//   GATH  : 0 normal, 1 gathers masked into 8 KiB (L1), diagnostic
template <bool NT, int STORE, int GATH>
__global__ __launch_bounds__(64) void v_rows(Args a)
{
    __shared__ __attribute__((aligned(16))) int s_col[512];
    __shared__ __attribute__((aligned(16))) double s_val[512];
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int nB = a.nBlocks;
    const int lastRow = a.rows - 1;
    const int kMax4 = (a.nnz - 4) & ~3;     // last aligned quad of column ids that is fully inside the array
    const int kMax2 = (a.nnz - 2) & ~1;

    auto ld_i = [&](const int* p) -> int { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto ld_i4 = [&](const int* p) -> i4 { return NT ? __builtin_nontemporal_load((const i4*)p) : *(const i4*)p; };
    auto ld_d2 = [&](const double* p) -> d2 { return NT ? __builtin_nontemporal_load((const d2*)p) : *(const d2*)p; };

    int rb = blockIdx.x;
    if (rb >= nB) return;
    // ro(0); raw(0); ro(1) -- the same order of outstanding loads as the loop's back edge
    int roA_s, roA_e, roB_s, roB_e;
    {
        int r = rb * 64 + tid; r = r < lastRow ? r : lastRow;
        roA_s = ld_i(a.ro + r); roA_e = ld_i(a.ro + r + 1);
    }
    i4 c0, c1; d2 v0, v1, v2, v3;
    auto raw = [&](int s) {
        const int tb = s & ~3;
        int k0 = tb + 4 * tid, k1 = k0 + 256;
        k0 = k0 < kMax4 ? k0 : kMax4; k1 = k1 < kMax4 ? k1 : kMax4;
        c0 = ld_i4(a.col + k0); c1 = ld_i4(a.col + k1);
        int j0 = tb + 2 * tid, j1 = j0 + 128, j2 = j0 + 256, j3 = j0 + 384;
        j0 = j0 < kMax2 ? j0 : kMax2; j1 = j1 < kMax2 ? j1 : kMax2; j2 = j2 < kMax2 ? j2 : kMax2; j3 = j3 < kMax2 ? j3 : kMax2;
        v0 = ld_d2(a.val + j0); v1 = ld_d2(a.val + j1); v2 = ld_d2(a.val + j2); v3 = ld_d2(a.val + j3);
    };
    raw(__builtin_amdgcn_readfirstlane(roA_s));
    {
        int rbn = rb + G; rbn = rbn < nB ? rbn : rb;
        int r2 = rbn * 64 + tid; r2 = r2 < lastRow ? r2 : lastRow;
        roB_s = ld_i(a.ro + r2); roB_e = ld_i(a.ro + r2 + 1);
    }
    double dot = 0.0;
    double pend = 0.0; int pendRow = -1;
    if (STORE == 4) { pendRow = rb * 64 + tid; pendRow = pendRow < lastRow ? pendRow : lastRow; }   // harmless first store, overwritten by the same lane
    for (; rb < nB; rb += G) {
        const int s = __builtin_amdgcn_readfirstlane(roA_s);
        const int tb = s & ~3;
        const int my_s = roA_s, my_e = roA_e;
        // stage
        *(i4*)(s_col + 4 * tid) = c0; *(i4*)(s_col + 256 + 4 * tid) = c1;
        *(d2*)(s_val + 2 * tid) = v0; *(d2*)(s_val + 128 + 2 * tid) = v1; *(d2*)(s_val + 256 + 2 * tid) = v2; *(d2*)(s_val + 384 + 2 * tid) = v3;
        __syncthreads();
        int row = rb * 64 + tid;
        const bool live = row <= lastRow;
        row = live ? row : lastRow;
        const int cnt = my_e - my_s;
        double xg[8], vv[8];
        if (STORE == 3) { if (pendRow >= 0) a.y[pendRow] = pend; }
        int cc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int idx = my_s - tb + j;
            idx = j < cnt ? idx : 0;
            cc[j] = s_col[idx & 511]; vv[j] = s_val[idx & 511];
        }
        __builtin_amdgcn_sched_barrier(0);          // all LDS reads in flight before the first gather waits for its column id
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int c = cc[j];
            if (GATH == 1) c &= 1023;
            xg[j] = a.x[c];
        }
        const double w = a.w[row];
        if (STORE == 4) a.y[pendRow] = pend;
        // raw(t+1) behind the gathers, ro(t+2) behind that
        const int sN = __builtin_amdgcn_readfirstlane(roB_s);
        raw(sN);
        roA_s = roB_s; roA_e = roB_e;
        {
            int rb2 = rb + 2 * G; rb2 = rb2 < nB ? rb2 : rb;
            int r2 = rb2 * 64 + tid; r2 = r2 < lastRow ? r2 : lastRow;
            roB_s = ld_i(a.ro + r2); roB_e = ld_i(a.ro + r2 + 1);
        }
        __builtin_amdgcn_sched_barrier(0);          // no product may move in front of the prefetch: its wait would hold the raw loads back
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const double p = vv[j] * xg[j]; acc += (j < cnt) ? p : 0.0; }
        const double t = w * acc; dot += live ? t : 0.0;
        if (STORE == 1) a.y[row] = acc;
        else if (STORE == 2) __builtin_nontemporal_store(acc, a.y + row);
        else if (STORE == 3 || STORE == 4) { pend = acc; pendRow = row; }
        else if (acc == 1.2345e300) a.y[row] = acc;
        __syncthreads();
    }
    if ((STORE == 3 || STORE == 4) && pendRow >= 0) a.y[pendRow] = pend;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
    if (tid == 0) a.partials[blockIdx.x] = dot;
}

// ------------------------------------------------------------------ S: the same streams without the product (ceiling of the access pattern)
//   W: 0 no y store, 1 y store; XR: also read x and w coalesced (8 B per row each)
template <bool NT, int W, int XR>
__global__ __launch_bounds__(64) void s_streams(Args a)
{
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int nB = a.nBlocks;
    const int kMax4 = (a.nnz - 4) & ~3, kMax2 = (a.nnz - 2) & ~1;
    auto ld_i = [&](const int* p) -> int { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto ld_i4 = [&](const int* p) -> i4 { return NT ? __builtin_nontemporal_load((const i4*)p) : *(const i4*)p; };
    auto ld_d2 = [&](const double* p) -> d2 { return NT ? __builtin_nontemporal_load((const d2*)p) : *(const d2*)p; };
    double acc = 0.0;
    for (int rb = blockIdx.x; rb < nB; rb += G) {
        const int r = rb * 64 + tid;
        const int rs = ld_i(a.ro + r), re = ld_i(a.ro + r + 1);
        const int tb = (rb * 448) & ~3;      // interior estimate of the span start: no dependent load (ceiling only)
        int k0 = tb + 4 * tid, k1 = k0 + 256;
        k0 = k0 < kMax4 ? k0 : kMax4; k1 = k1 < kMax4 ? k1 : kMax4;
        const i4 c0 = ld_i4(a.col + k0), c1 = ld_i4(a.col + k1);
        int j0 = tb + 2 * tid, j1 = j0 + 128, j2 = j0 + 256, j3 = j0 + 384;
        j0 = j0 < kMax2 ? j0 : kMax2; j1 = j1 < kMax2 ? j1 : kMax2; j2 = j2 < kMax2 ? j2 : kMax2; j3 = j3 < kMax2 ? j3 : kMax2;
        const d2 v0 = ld_d2(a.val + j0), v1 = ld_d2(a.val + j1), v2 = ld_d2(a.val + j2), v3 = ld_d2(a.val + j3);
        double t = v0.x + v1.y + v2.x + v3.y + (double)(c0.x + c1.w + rs + re);
        if (XR) t += a.x[r] + a.w[r];
        acc += t;
        if (W) a.y[r] = t;
    }
    if (acc == 1.2345e300) a.partials[blockIdx.x] = acc;
}


// ------------------------------------------------------------------ W: store-shape study on the bare streams
// Wave g handles K consecutive 64-row blocks per super-trip ((g + G*T)*K + k).  MODE: 0 no store, 1 store 512 B per block as it is
// produced, 2 the same confined to 512 KB, 3 nt store, 4 the K results kept in registers and stored back to back at the end of
// the super-trip, 5 as 4 with sc0 sc1 (write-through) stores, 6 as 4 nt
template <bool NT, int K, int MODE>
__global__ __launch_bounds__(64) void w_streams(Args a)
{
    const int tid = threadIdx.x;
    const int G = gridDim.x;
    const int nB = a.nBlocks;
    const int kMax4 = (a.nnz - 4) & ~3, kMax2 = (a.nnz - 2) & ~1;
    auto ld_i = [&](const int* p) -> int { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto ld_i4 = [&](const int* p) -> i4 { return NT ? __builtin_nontemporal_load((const i4*)p) : *(const i4*)p; };
    auto ld_d2 = [&](const double* p) -> d2 { return NT ? __builtin_nontemporal_load((const d2*)p) : *(const d2*)p; };
    double acc = 0.0;
    for (int sb = blockIdx.x * K; sb < nB; sb += G * K) {
        double res[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            int rb = (MODE >= 10) ? (sb / K + k * G) + (sb / (G * K)) * (G * (K - 1)) : sb + k;   // MODE >= 10: K blocks a grid-stride apart (same bytes in flight per wave, not contiguous)
            rb = rb < nB ? rb : nB - 1;
            const int r = rb * 64 + tid;
            const int rs = ld_i(a.ro + r), re = ld_i(a.ro + r + 1);
            const int tb = (rb * 448) & ~3;
            int k0 = tb + 4 * tid, k1 = k0 + 256;
            k0 = k0 < kMax4 ? k0 : kMax4; k1 = k1 < kMax4 ? k1 : kMax4;
            const i4 c0 = ld_i4(a.col + k0), c1 = ld_i4(a.col + k1);
            int j0 = tb + 2 * tid, j1 = j0 + 128, j2 = j0 + 256, j3 = j0 + 384;
            j0 = j0 < kMax2 ? j0 : kMax2; j1 = j1 < kMax2 ? j1 : kMax2; j2 = j2 < kMax2 ? j2 : kMax2; j3 = j3 < kMax2 ? j3 : kMax2;
            const d2 v0 = ld_d2(a.val + j0), v1 = ld_d2(a.val + j1), v2 = ld_d2(a.val + j2), v3 = ld_d2(a.val + j3);
            const double t = v0.x + v1.y + v2.x + v3.y + (double)(c0.x + c1.w + rs + re);
            acc += t; res[k] = t;
            if (MODE == 1) a.y[r] = t;
            else if (MODE == 2) a.y[r & 65535] = t;
            else if (MODE == 3) __builtin_nontemporal_store(t, a.y + r);
        }
        if (MODE == 7 || MODE == 8) {          // the K * 64 results as 16-byte-per-lane stores: K / 2 instructions of 1 KiB each
#pragma unroll
            for (int k = 0; k < K; k += 2) {
                d2 v2; v2.x = res[k]; v2.y = res[k + 1];
                d2* q = (d2*)(a.y + (long long)(sb + k) * 64) + tid;
                if (MODE == 7) *q = v2; else __builtin_nontemporal_store(v2, q);
            }
        } else if (MODE >= 4) {
#pragma unroll
            for (int k = 0; k < K; ++k) {
                int rb = sb + k; rb = rb < nB ? rb : nB - 1;
                double* q = a.y + rb * 64 + tid;
                if (MODE == 4) *q = res[k];
                else if (MODE == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" :: "v"(q), "v"(res[k]) : "memory");
                else __builtin_nontemporal_store(res[k], q);
            }
        }
    }
    if (acc == 1.2345e300) a.partials[blockIdx.x] = acc;
}


// ------------------------------------------------------------------ V2: lane = row, RPT row blocks (64 rows each) per trip, mini-chunks
// Wave g walks "super blocks" of KCH consecutive row blocks; a trip covers RPT consecutive blocks of the super block (rows
// l, l + 64, ... per lane).  map 0: super block = g + G*q.  map 1: XCD k = g % 8 owns a contiguous run of G/8 super blocks per
// super trip.  map 2: plane-aware (planeBlocks row blocks per grid plane): XCD k sweeps the k-th eighth of every plane.
// STORE: 0 none, 1 plain right after the products, 2 nt, 4 deferred one trip (between gathers and raw(t+1)), 5 deferred nt
// WPB waves per workgroup: every wave runs the same single-wave pipeline on its own LDS slice; the barriers only keep the waves
// of a workgroup (which own adjacent super blocks) issuing their raw loads together.
template <bool NT, int STORE, int RPT, int KCH, int WPB, int NG = 8>
__global__ __launch_bounds__(64 * WPB) void v2_rows(Args a, int map, int planeBlocks)
{
    constexpr int CAP = 512 * RPT;
    constexpr int TPS = KCH / RPT;                 // trips per super block
    __shared__ __attribute__((aligned(16))) int s_colAll[CAP * WPB];
    __shared__ __attribute__((aligned(16))) double s_valAll[CAP * WPB];
    const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* s_col = s_colAll + wv * CAP; double* s_val = s_valAll + wv * CAP;
    const int G = gridDim.x * WPB, g = blockIdx.x * WPB + wv;
    const int nB = a.nBlocks;
    const int lastRow = a.rows - 1;
    const int kMax4 = (a.nnz - 4) & ~3;
    const int kMax2 = (a.nnz - 2) & ~1;
    auto ld_i = [&](const int* p) -> int { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto ld_i4 = [&](const int* p) -> i4 { return NT ? __builtin_nontemporal_load((const i4*)p) : *(const i4*)p; };
    auto ld_d2 = [&](const double* p) -> d2 { return NT ? __builtin_nontemporal_load((const d2*)p) : *(const d2*)p; };

    const int nSuper = (nB + KCH - 1) / KCH;
    // number of super blocks this wave owns, and the i-th of them
    int nMine;
    const int xcd = g & 7, wx = g >> 3, GX = G >> 3;
    const int eighth = planeBlocks / 8 / KCH;      // super blocks per plane-eighth (map 2)
    if (map == 0 || map == 3 || (map & 255) == 5 || (map & 255) == 6 || map >= 16) nMine = (nSuper + G - 1) / G;   // (lab: nSuper is a multiple of G)
    if (map == 0) nMine = nSuper > g ? (nSuper - g + G - 1) / G : 0;
    else if (map == 1 || map == 3 || (map & 255) == 5 || (map & 255) == 6 || map >= 16) { nMine = (nSuper + G - 1) / G; }
    else { const int per = nSuper / 8; nMine = per > wx ? (per - wx + GX - 1) / GX : 0; }
    auto super_of = [&](int q) -> int {
        if ((map & 255) == 5) {      // sweep z for one y-slice of the planes at a time: trip q = h * nz + p covers tiles [p*T + h*WG, +WG) of plane p, XCD k a contiguous eighth of them
            const int wg = blockIdx.x, WG = gridDim.x, T = planeBlocks / WPB, nzp = nB / planeBlocks;
            const int h = q / nzp, pz = q - h * nzp;
            const int tile = pz * T + h * WG + (wg & 7) * (WG >> 3) + (wg >> 3);
            return tile * WPB + wv;
        }
        if ((map & 255) == 6) { const int wg = blockIdx.x, WG = gridDim.x; return (q * WG + (wg & 7) * (WG >> 3) + (wg >> 3)) * WPB + wv; }   // memory order, XCD k a contiguous eighth of every trip
        if (map >= 16 && map < 256) { const int CH = map >> 4; const int wg = blockIdx.x, WG = gridDim.x; return ((wg + WG * (q / CH)) * CH + q % CH) * WPB + wv; }   // a workgroup walks CH * WPB consecutive blocks
        if (map == 0) return g + G * q;
        if (map == 1) return q * G + xcd * GX + wx;
        if (map == 3) { const int wg = blockIdx.x, WG = gridDim.x; const int k8 = wg & 7, j = wg >> 3; return q * G + (j * WPB + wv) * 8 + k8; }   // workgroup wg (XCD wg % 8): blocks = k8 mod 8, WPB of them 8 apart
        const int m = q * GX + wx;                 // m-th super block of this XCD's list: plane p = m / eighth, j = m % eighth
        const int p = m / eighth, j = m - p * eighth;
        return p * (planeBlocks / KCH) + xcd * eighth + j;
    };
    const int myTrips = nMine * TPS;
    int nTrips = myTrips;
    if (WPB > 1) { __shared__ int s_trips; if (threadIdx.x == 0) s_trips = myTrips; __syncthreads(); nTrips = s_trips; }   // wave 0 owns the most
    auto block_of = [&](int t) -> int {
        int tt = t < myTrips ? t : myTrips - 1; tt = tt > 0 ? tt : 0;
        const int q = tt / TPS, sub = tt - q * TPS;
        int b = super_of(q) * KCH + sub * RPT;
        return b < nB ? b : nB - RPT;              // (lab: nB is a multiple of KCH; clamp only guards map 1's ragged end)
    };
    if (nTrips <= 0) return;                       // workgroup-uniform

    int ro_s[RPT], ro_e[RPT], rn_s[RPT], rn_e[RPT];
    auto load_ro = [&](int b, int* rs, int* re) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) { int r = (b + u) * 64 + tid; r = r < lastRow ? r : lastRow; rs[u] = ld_i(a.ro + r); re[u] = ld_i(a.ro + r + 1); }
    };
    i4 c[2 * RPT]; d2 v[4 * RPT];
    auto raw = [&](int s) {
        const int tb = s & ~3;
#pragma unroll
        for (int q = 0; q < 2 * RPT; ++q) { int k = tb + 4 * tid + 256 * q; k = k < kMax4 ? k : kMax4; c[q] = ld_i4(a.col + k); }
#pragma unroll
        for (int q = 0; q < 4 * RPT; ++q) { int k = tb + 2 * tid + 128 * q; k = k < kMax2 ? k : kMax2; v[q] = ld_d2(a.val + k); }
    };
    int bCur = block_of(0);
    load_ro(bCur, ro_s, ro_e);
    raw(__builtin_amdgcn_readfirstlane(ro_s[0]));
    int bNext = block_of(1);
    load_ro(bNext, rn_s, rn_e);
    double dot = 0.0;
    __amdgpu_buffer_rsrc_t yres = __builtin_amdgcn_make_buffer_rsrc((void*)a.y, 0, -1, 0x00020000);
    double pend[RPT]; int pendRow[RPT];
#pragma unroll
    for (int u = 0; u < RPT; ++u) { pend[u] = 0.0; int r = (bCur + u) * 64 + tid; pendRow[u] = r < lastRow ? r : lastRow; }
    for (int t = 0; t < nTrips; ++t) {
        const int tb = __builtin_amdgcn_readfirstlane(ro_s[0]) & ~3;
#pragma unroll
        for (int q = 0; q < 2 * RPT; ++q) *(i4*)(s_col + 4 * tid + 256 * q) = c[q];
#pragma unroll
        for (int q = 0; q < 4 * RPT; ++q) *(d2*)(s_val + 2 * tid + 128 * q) = v[q];
        __syncthreads();
        int row[RPT], cnt[RPT]; bool live[RPT];
        int cc[RPT][NG]; double vv[RPT][NG], xg[RPT][NG];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            row[u] = (bCur + u) * 64 + tid; live[u] = row[u] <= lastRow && t < myTrips; row[u] = live[u] ? row[u] : lastRow;
            cnt[u] = ro_e[u] - ro_s[u];
#pragma unroll
            for (int j = 0; j < NG; ++j) {
                int idx = ro_s[u] - tb + j;
                idx = j < cnt[u] ? idx : 0;
                cc[u][j] = s_col[idx & (CAP - 1)]; vv[u][j] = s_val[idx & (CAP - 1)];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < RPT; ++u)
#pragma unroll
            for (int j = 0; j < NG; ++j) xg[u][j] = a.x[cc[u][j] & a.gmask];
        double w[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) w[u] = a.w[row[u]];
        if (STORE == 4 || STORE == 5) {
#pragma unroll
            for (int u = 0; u < RPT; ++u) { if (STORE == 4) a.y[pendRow[u]] = pend[u]; else __builtin_nontemporal_store(pend[u], a.y + pendRow[u]); }
        }
        if (STORE >= 100) {
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                u2v bits; { const unsigned long long q = (unsigned long long)__double_as_longlong(pend[u]); bits.x = (unsigned)q; bits.y = (unsigned)(q >> 32); }
                __builtin_amdgcn_raw_buffer_store_b64(bits, yres, pendRow[u] * 8, 0, STORE - 100);
            }
        }
        raw(__builtin_amdgcn_readfirstlane(rn_s[0]));
#pragma unroll
        for (int u = 0; u < RPT; ++u) { ro_s[u] = rn_s[u]; ro_e[u] = rn_e[u]; }
        const int bAfter = block_of(t + 2);
        load_ro(bAfter, rn_s, rn_e);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NG; ++j) { const double p = vv[u][j] * xg[u][j]; acc += (j < cnt[u]) ? p : 0.0; }
            const double tt = w[u] * acc; dot += live[u] ? tt : 0.0;
            if (STORE == 1) a.y[row[u]] = acc;
            else if (STORE == 2) __builtin_nontemporal_store(acc, a.y + row[u]);
            else if (STORE >= 4) { pend[u] = acc; pendRow[u] = row[u]; }
            else if (acc == 1.2345e300) a.y[row[u]] = acc;
        }
        if (!(map & 256)) __syncthreads();           // (map bit 8: one barrier per trip -- LDS slices are private to a wavefront)
        bCur = bNext; bNext = bAfter;
    }
    if (STORE >= 4) {
#pragma unroll
        for (int u = 0; u < RPT; ++u) a.y[pendRow[u]] = pend[u];
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
    if (tid == 0) a.partials[blockIdx.x * WPB + wv] = dot;
}

// ------------------------------------------------------------------ V4: as V2 (RPT 1, KCH 1) with the y stores of B consecutive trips kept in
// registers and flushed together at trip numbers = 0 mod B -- the same trip numbers in every workgroup, so that (while the workgroups
// stay roughly in step) the whole chip writes in bursts and reads in between
template <int WPB, int NG, int B>
__global__ __launch_bounds__(64 * WPB) void v4_rows(Args a, int map, int planeBlocks)
{
    __shared__ double s_pendAll[B * 64 * WPB];
    __shared__ int s_prowAll[B * WPB];
    constexpr int CAP = 512;
    __shared__ __attribute__((aligned(16))) int s_colAll[CAP * WPB];
    __shared__ __attribute__((aligned(16))) double s_valAll[CAP * WPB];
    const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* s_col = s_colAll + wv * CAP; double* s_val = s_valAll + wv * CAP;
    double* s_pend = s_pendAll + wv * B * 64; int* s_prow = s_prowAll + wv * B;
    const int nB = a.nBlocks;
    const int kMax4 = (a.nnz - 4) & ~3, kMax2 = (a.nnz - 2) & ~1;
    const int wg = blockIdx.x, WG = gridDim.x;
    const int nTrips = nB / (WG * WPB);             // lab: exact
    auto block_of = [&](int t) -> int {
        t = t < nTrips ? t : nTrips - 1;
        if (map == 5) {
            const int T = planeBlocks / WPB, nzp = nB / planeBlocks;
            const int h = t / nzp, pz = t - h * nzp;
            return (pz * T + h * WG + (wg & 7) * (WG >> 3) + (wg >> 3)) * WPB + wv;
        }
        return (wg + WG * t) * WPB + wv;
    };
    int roA_s, roA_e, roB_s, roB_e;
    i4 c0, c1; d2 v0, v1, v2, v3;
    auto load_ro = [&](int b, int& rs, int& re) { const int r = b * 64 + tid; rs = a.ro[r]; re = a.ro[r + 1]; };
    auto raw = [&](int s) {
        const int tb = s & ~3;
        int k0 = tb + 4 * tid, k1 = k0 + 256; k0 = k0 < kMax4 ? k0 : kMax4; k1 = k1 < kMax4 ? k1 : kMax4;
        c0 = *(const i4*)(a.col + k0); c1 = *(const i4*)(a.col + k1);
        int j0 = tb + 2 * tid, j1 = j0 + 128, j2 = j0 + 256, j3 = j0 + 384;
        j0 = j0 < kMax2 ? j0 : kMax2; j1 = j1 < kMax2 ? j1 : kMax2; j2 = j2 < kMax2 ? j2 : kMax2; j3 = j3 < kMax2 ? j3 : kMax2;
        v0 = *(const d2*)(a.val + j0); v1 = *(const d2*)(a.val + j1); v2 = *(const d2*)(a.val + j2); v3 = *(const d2*)(a.val + j3);
    };
    int bCur = block_of(0), bNext = block_of(1);
    load_ro(bCur, roA_s, roA_e);
    raw(__builtin_amdgcn_readfirstlane(roA_s));
    load_ro(bNext, roB_s, roB_e);
    double dot = 0.0;
    int nPend = 0;
    for (int t = 0; t < nTrips; ++t) {
        {
            const int b = t % B;
            const int tb = __builtin_amdgcn_readfirstlane(roA_s) & ~3;
            const int my_s = roA_s, cnt = roA_e - roA_s;
            *(i4*)(s_col + 4 * tid) = c0; *(i4*)(s_col + 256 + 4 * tid) = c1;
            *(d2*)(s_val + 2 * tid) = v0; *(d2*)(s_val + 128 + 2 * tid) = v1; *(d2*)(s_val + 256 + 2 * tid) = v2; *(d2*)(s_val + 384 + 2 * tid) = v3;
            __syncthreads();
            const int row = bCur * 64 + tid;
            int cc[NG]; double vv[NG], xg[NG];
#pragma unroll
            for (int j = 0; j < NG; ++j) { int idx = my_s - tb + j; idx = j < cnt ? idx : 0; cc[j] = s_col[idx & 511]; vv[j] = s_val[idx & 511]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NG; ++j) xg[j] = a.x[cc[j]];
            const double w = a.w[row];
            if (b == 0 && nPend > 0) {               // wavefront-uniform: flush the B results of the previous batch
                for (int q = 0; q < B; ++q) __builtin_nontemporal_store(s_pend[q * 64 + tid], a.y + s_prow[q] + tid);
            }
            raw(__builtin_amdgcn_readfirstlane(roB_s));
            roA_s = roB_s; roA_e = roB_e;
            const int bAfter = block_of(t + 2);
            load_ro(bAfter, roB_s, roB_e);
            __builtin_amdgcn_sched_barrier(0);
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < NG; ++j) { const double p = vv[j] * xg[j]; acc += (j < cnt) ? p : 0.0; }
            dot += w * acc;
            s_pend[b * 64 + tid] = acc; if (tid == 0) s_prow[b] = bCur * 64; nPend = 1;
            __syncthreads();
            bCur = bNext; bNext = bAfter;
        }
    }
    __syncthreads();
    for (int q = 0; q < B; ++q) __builtin_nontemporal_store(s_pend[q * 64 + tid], a.y + s_prow[q] + tid);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
    if (tid == 0) a.partials[blockIdx.x * WPB + wv] = dot;
}

// ------------------------------------------------------------------ I: one interleaved stream.  Per 256-row tile a record of 22528 B = [256 row
// offsets... (1 KiB)][cols 7 KiB][vals 14 KiB] (interior tile of the 7-point matrix): the same bytes as the three CSR streams,
// contiguous.  Bare streaming (no product), WPB4-style: 4 waves x 5632 B per trip, + optional y store and x,w reads.
template <int W, int XR>
__global__ __launch_bounds__(256) void i_stream(const d2* __restrict__ rec, long long recBytes, int nTiles, Args a)
{
    const int tid = threadIdx.x;
    double acc = 0.0;
    for (int tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        const d2* base = (const d2*)((const char*)rec + (long long)tile * 22528);
        d2 v[6];
#pragma unroll
        for (int q = 0; q < 5; ++q) v[q] = base[tid + 256 * q];          // 5 x 4096 B
        v[5] = tid < 128 ? base[tid + 1280] : v[4];                       // + 2048 B = 22528
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 6; ++q) t += v[q].x + v[q].y;
        const int r = tile * 256 + tid;
        if (XR) t += a.x[r] + a.w[r];
        acc += t;
        if (W) __builtin_nontemporal_store(t, a.y + r);
    }
    if (acc == 1.2345e300) a.partials[blockIdx.x] = acc;
}
// the same bytes from the three CSR arrays, same workgroup shape
template <int W, int XR>
__global__ __launch_bounds__(256) void c_stream(Args a, int nTiles)
{
    const int tid = threadIdx.x;
    const int kMax4 = (a.nnz - 4) & ~3, kMax2 = (a.nnz - 2) & ~1;
    double acc = 0.0;
    for (int tile = blockIdx.x; tile < nTiles; tile += gridDim.x) {
        const int r = tile * 256 + tid;
        const int rs = a.ro[r];
        const int tb = (tile * 1792) & ~3;
        int k0 = tb + 4 * tid, k1 = k0 + 1024; k0 = k0 < kMax4 ? k0 : kMax4; k1 = k1 < kMax4 ? k1 : kMax4;
        const i4 c0 = *(const i4*)(a.col + k0), c1 = *(const i4*)(a.col + k1);
        d2 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { int j = tb + 2 * tid + 512 * q; j = j < kMax2 ? j : kMax2; v[q] = *(const d2*)(a.val + j); }
        double t = (double)(rs + c0.x + c0.y + c0.z + c0.w + c1.x + c1.y + c1.z + c1.w);
#pragma unroll
        for (int q = 0; q < 4; ++q) t += v[q].x + v[q].y;
        if (XR) t += a.x[r] + a.w[r];
        acc += t;
        if (W) __builtin_nontemporal_store(t, a.y + r);
    }
    if (acc == 1.2345e300) a.partials[blockIdx.x] = acc;
}

// ------------------------------------------------------------------ V5: block records.  Per 64-row block one contiguous, 16-byte aligned record
//   [65 int32: start of every row relative to the block, then the block's nnz][pad to 272 B][cols, padded to a multiple of 4][vals]
// = the CSR numbers of the block in one piece (12 B per nonzero + 4.25 B per row); recOff[b] = byte offset of block b's record / 16.
__global__ void rec_size_kernel(const int* __restrict__ ro, int nBlocks, int rows, unsigned* __restrict__ units)
{
    for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < nBlocks; b += gridDim.x * blockDim.x) {
        const int r0 = b * 64, r1 = (r0 + 64 < rows) ? r0 + 64 : rows;
        const int nnzB = ro[r1] - ro[r0];
        const int P = (nnzB + 3) & ~3;
        units[b] = (272 + 12 * P) / 16;
    }
}
__global__ __launch_bounds__(64) void rec_fill_kernel(const int* __restrict__ ro, const int* __restrict__ col, const double* __restrict__ val, int nBlocks, int rows,
                                                      const long long* __restrict__ recOff, char* __restrict__ rec)
{
    for (int b = blockIdx.x; b < nBlocks; b += gridDim.x) {
        const int r0 = b * 64, r1 = (r0 + 64 < rows) ? r0 + 64 : rows;
        const int s = ro[r0], nnzB = ro[r1] - s, P = (nnzB + 3) & ~3;
        char* base = rec + recOff[b] * 16;
        int* hdr = (int*)base;
        const int l = threadIdx.x;
        { const int r = r0 + l < r1 ? r0 + l : r1; hdr[l] = ro[r] - s; }
        if (l == 0) { hdr[64] = nnzB; hdr[65] = 0; hdr[66] = 0; hdr[67] = 0; }
        int* c = (int*)(base + 272); double* v = (double*)(base + 272 + 4 * P);
        for (int k = l; k < P; k += 64) { c[k] = k < nnzB ? col[s + k] : 0; v[k] = k < nnzB ? val[s + k] : 0.0; }
    }
}
template <int NG>
__global__ __launch_bounds__(256) void v5_rows(Args a, const long long* __restrict__ recOff, const char* __restrict__ rec, int map, int planeBlocks)
{
    constexpr int WPB = 4;
    __shared__ __attribute__((aligned(16))) char s_recAll[6144 * WPB];
    const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    char* s_rec = s_recAll + wv * 6144;
    const int* s_hdr = (const int*)s_rec; const int* s_col = (const int*)(s_rec + 272);
    const int nB = a.nBlocks;
    const int wg = blockIdx.x, WG = gridDim.x;
    const int nTrips = nB / (WG * WPB);
    auto block_of = [&](int t) -> int {
        t = t < nTrips ? t : nTrips - 1;
        if (map == 5) {
            const int T = planeBlocks / WPB, nzp = nB / planeBlocks;
            const int h = t / nzp, pz = t - h * nzp;
            return (pz * T + h * WG + (wg & 7) * (WG >> 3) + (wg >> 3)) * WPB + wv;
        }
        return (wg + WG * t) * WPB + wv;
    };
    typedef int i4v __attribute__((ext_vector_type(4)));
    i4v raw[6];
    auto load_raw = [&](long long off16, long long end16) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { long long u = off16 + tid + 64 * q; u = u < end16 ? u : end16 - 1; raw[q] = *(const i4v*)(rec + u * 16); }
    };
    int bCur = block_of(0), bNext = block_of(1);
    long long oA = recOff[bCur], eA = recOff[bCur + 1];
    load_raw(oA, eA);
    long long oB = recOff[bNext], eB = recOff[bNext + 1];
    double dot = 0.0, pend = 0.0;
    int pendRow = bCur * 64 + tid;
    for (int t = 0; t < nTrips; ++t) {
#pragma unroll
        for (int q = 0; q < 6; ++q) *(i4v*)(s_rec + 16 * (tid + 64 * q)) = raw[q];
        __syncthreads();
        const int my_s = s_hdr[tid], my_e = s_hdr[tid + 1], nnzB = s_hdr[64];
        const int cnt = my_e - my_s;
        const int P = (nnzB + 3) & ~3;
        const double* s_val = (const double*)(s_rec + 272 + 4 * P);
        const int row = bCur * 64 + tid;
        int cc[NG]; double vv[NG], xg[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) { int idx = my_s + j; idx = j < cnt ? idx : 0; cc[j] = s_col[idx]; vv[j] = s_val[idx]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NG; ++j) xg[j] = a.x[cc[j]];
        const double w = a.w[row];
        __builtin_nontemporal_store(pend, a.y + pendRow);
        load_raw(oB, eB);
        const int bAfter = block_of(t + 2);
        oB = recOff[bAfter]; eB = recOff[bAfter + 1];
        __builtin_amdgcn_sched_barrier(0);
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NG; ++j) { const double p = vv[j] * xg[j]; acc += (j < cnt) ? p : 0.0; }
        dot += w * acc;
        pend = acc; pendRow = row;
        __syncthreads();
        bCur = bNext; bNext = bAfter;
    }
    __builtin_nontemporal_store(pend, a.y + pendRow);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
    if (tid == 0) a.partials[blockIdx.x * WPB + wv] = dot;
}

// ------------------------------------------------------------------ V6: as V4 (results of up to B trips parked in LDS) but the flush is keyed to the
// chip-wide 100 MHz clock (s_memrealtime): every workgroup flushes at its first trip boundary after the clock crosses a multiple of
// `period` ticks -- write bursts of the whole chip without any communication.  A flush always stores all B slots (static count:
// counted waits); slots not refilled since the last flush are stored again with the same values.
template <int NG, int B>
__global__ __launch_bounds__(256) void v6_rows(Args a, int map, int planeBlocks, int period)
{
    constexpr int WPB = 4, CAP = 512;
    __shared__ __attribute__((aligned(16))) int s_colAll[CAP * WPB];
    __shared__ __attribute__((aligned(16))) double s_valAll[CAP * WPB];
    __shared__ double s_pendAll[B * 64 * WPB];
    __shared__ int s_prowAll[B * WPB];
    const int tid = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int* s_col = s_colAll + wv * CAP; double* s_val = s_valAll + wv * CAP;
    double* s_pend = s_pendAll + wv * B * 64; int* s_prow = s_prowAll + wv * B;
    const int nB = a.nBlocks;
    const int kMax4 = (a.nnz - 4) & ~3, kMax2 = (a.nnz - 2) & ~1;
    const int wg = blockIdx.x, WG = gridDim.x;
    const int nTrips = nB / (WG * WPB);
    auto block_of = [&](int t) -> int {
        t = t < nTrips ? t : nTrips - 1;
        if (map == 5) { const int T = planeBlocks / WPB, nzp = nB / planeBlocks; const int h = t / nzp, pz = t - h * nzp; return (pz * T + h * WG + (wg & 7) * (WG >> 3) + (wg >> 3)) * WPB + wv; }
        return (wg + WG * t) * WPB + wv;
    };
    int roA_s, roA_e, roB_s, roB_e;
    i4 c0, c1; d2 v0, v1, v2, v3;
    auto load_ro = [&](int b, int& rs, int& re) { const int r = b * 64 + tid; rs = a.ro[r]; re = a.ro[r + 1]; };
    auto raw = [&](int s) {
        const int tb = s & ~3;
        int k0 = tb + 4 * tid, k1 = k0 + 256; k0 = k0 < kMax4 ? k0 : kMax4; k1 = k1 < kMax4 ? k1 : kMax4;
        c0 = *(const i4*)(a.col + k0); c1 = *(const i4*)(a.col + k1);
        int j0 = tb + 2 * tid, j1 = j0 + 128, j2 = j0 + 256, j3 = j0 + 384;
        j0 = j0 < kMax2 ? j0 : kMax2; j1 = j1 < kMax2 ? j1 : kMax2; j2 = j2 < kMax2 ? j2 : kMax2; j3 = j3 < kMax2 ? j3 : kMax2;
        v0 = *(const d2*)(a.val + j0); v1 = *(const d2*)(a.val + j1); v2 = *(const d2*)(a.val + j2); v3 = *(const d2*)(a.val + j3);
    };
    int bCur = block_of(0), bNext = block_of(1);
    load_ro(bCur, roA_s, roA_e);
    raw(__builtin_amdgcn_readfirstlane(roA_s));
    load_ro(bNext, roB_s, roB_e);
    // every slot starts as "row of my first block, value 0": a harmless store until the slot is filled
    for (int q = 0; q < B; ++q) { s_pend[q * 64 + tid] = 0.0; if (tid == 0) s_prow[q] = bCur * 64; }
    double dot = 0.0;
    int nbuf = 0;
    unsigned long long epoch = __builtin_amdgcn_s_memrealtime() / (unsigned)period;
    for (int t = 0; t < nTrips; ++t) {
        const int tb = __builtin_amdgcn_readfirstlane(roA_s) & ~3;
        const int my_s = roA_s, cnt = roA_e - roA_s;
        *(i4*)(s_col + 4 * tid) = c0; *(i4*)(s_col + 256 + 4 * tid) = c1;
        *(d2*)(s_val + 2 * tid) = v0; *(d2*)(s_val + 128 + 2 * tid) = v1; *(d2*)(s_val + 256 + 2 * tid) = v2; *(d2*)(s_val + 384 + 2 * tid) = v3;
        __syncthreads();
        const int row = bCur * 64 + tid;
        int cc[NG]; double vv[NG], xg[NG];
#pragma unroll
        for (int j = 0; j < NG; ++j) { int idx = my_s - tb + j; idx = j < cnt ? idx : 0; cc[j] = s_col[idx & 511]; vv[j] = s_val[idx & 511]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NG; ++j) xg[j] = a.x[cc[j]];
        const double w = a.w[row];
        const unsigned long long now = __builtin_amdgcn_s_memrealtime() / (unsigned)period;
        const bool flush = (nbuf == B) || (now != epoch && nbuf >= B / 2);        // wavefront-uniform (workgroup-nearly-uniform: each wave decides alone)
        if (flush) {
#pragma unroll
            for (int q = 0; q < B; ++q) __builtin_nontemporal_store(s_pend[q * 64 + tid], a.y + s_prow[q] + tid);
            nbuf = 0; epoch = now;
        }
        raw(__builtin_amdgcn_readfirstlane(roB_s));
        roA_s = roB_s; roA_e = roB_e;
        const int bAfter = block_of(t + 2);
        load_ro(bAfter, roB_s, roB_e);
        __builtin_amdgcn_sched_barrier(0);
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < NG; ++j) { const double p = vv[j] * xg[j]; acc += (j < cnt) ? p : 0.0; }
        dot += w * acc;
        s_pend[nbuf * 64 + tid] = acc; if (tid == 0) s_prow[nbuf] = bCur * 64;
        ++nbuf;
        __syncthreads();
        bCur = bNext; bNext = bAfter;
    }
    for (int q = 0; q < B; ++q) __builtin_nontemporal_store(s_pend[q * 64 + tid], a.y + s_prow[q] + tid);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_down(dot, off, 64);
    if (tid == 0) a.partials[blockIdx.x * WPB + wv] = dot;
}

template <typename F>
static double time_ms(F f, int reps)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f(); CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int r = 0; r < reps; ++r) { CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms); }
    std::sort(t.begin(), t.end());
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const long long N = (long long)n * n * n;
    const long long nnz = 7 * N - 6LL * n * n;
    printf("grid %d^3 rows %lld nnz %lld\n", n, N, nnz);
    std::vector<int> ro(N + 1);
    {
        long long k = 0;
        for (int z = 0; z < n; ++z) for (int y = 0; y < n; ++y) for (int x = 0; x < n; ++x) {
            ro[((long long)z * n + y) * n + x] = (int)k;
            k += 1 + (z > 0) + (y > 0) + (x > 0) + (x < n - 1) + (y < n - 1) + (z < n - 1);
        }
        ro[N] = (int)k;
        if (k != nnz) { printf("nnz mismatch\n"); return 1; }
    }
    int *d_ro, *d_col; double *d_val, *d_x, *d_y, *d_ref, *d_part; unsigned long long* d_bad;
    CK(hipMalloc(&d_ro, (N + 1) * 4)); CK(hipMalloc(&d_col, nnz * 4)); CK(hipMalloc(&d_val, nnz * 8));
    CK(hipMalloc(&d_x, N * 8)); CK(hipMalloc(&d_y, N * 8)); CK(hipMalloc(&d_ref, N * 8)); CK(hipMalloc(&d_part, 65536 * 8)); CK(hipMalloc(&d_bad, 8));
    CK(hipMemcpy(d_ro, ro.data(), (N + 1) * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill_poisson, dim3(8192), dim3(256), 0, 0, n, d_ro, d_col, d_val);
    hipLaunchKernelGGL(fill_x, dim3(8192), dim3(256), 0, 0, d_x, N);
    hipLaunchKernelGGL(ref_spmv, dim3(8192), dim3(256), 0, 0, d_ro, d_col, d_val, d_x, d_ref, N);
    CK(hipDeviceSynchronize());
    const double algo = 12.0 * nnz + 4.0 * (N + 1) + 16.0 * N;
    Args a{ d_ro, d_col, d_val, d_x, d_x, d_y, d_part, (int)N, (int)nnz, (int)((N + 63) / 64), -1 };

    auto check = [&](const char* name) {
        CK(hipMemset(d_bad, 0, 8));
        hipLaunchKernelGGL(cmp_kernel, dim3(4096), dim3(256), 0, 0, d_y, d_ref, N, d_bad);
        unsigned long long bad; CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
        if (bad) printf("    !! %s: %llu rows differ from the reference\n", name, bad);
        return bad == 0;
    };
    auto report = [&](const char* name, int wpc, double ms, bool ok) {
        printf("  %-34s waves/CU %2d  %7.3f ms  %7.1f GB/s  %5.1f%% of 8 TB/s %s\n", name, wpc, ms, algo / ms / 1e6, algo / ms / 1e6 / 80.0, ok ? "" : "(unchecked/BAD)");
        fflush(stdout);
    };
#define RUNV(NTv, ST, GA, label, verify) \
    for (int wpc : wpcs) { CK(hipMemset(d_y, 0xff, N * 8)); \
        double ms = time_ms([&] { hipLaunchKernelGGL((v_rows<NTv, ST, GA>), dim3(wpc * 256), dim3(64), 0, 0, a); }, reps); \
        bool ok = verify ? check(label) : true; report(label, wpc, ms, ok); }
#define RUNS(NTv, Wv, XRv, label) \
    for (int wpc : wpcs) { double ms = time_ms([&] { hipLaunchKernelGGL((s_streams<NTv, Wv, XRv>), dim3(wpc * 256), dim3(64), 0, 0, a); }, reps); report(label, wpc, ms, true); }

#define RUNW(NTv, Kv, Mv, label) \
    for (int wpc : wpcs) { double ms = time_ms([&] { hipLaunchKernelGGL((w_streams<NTv, Kv, Mv>), dim3(wpc * 256), dim3(64), 0, 0, a); }, reps); report(label, wpc, ms, true); }
    std::vector<int> wpcs = { 8, 16 };
    if (argc > 3 && !strcmp(argv[3], "stores")) {
        printf("-- store shapes on the bare streams (K blocks per wave per super-trip)\n");
        RUNW(false, 1, 0, "W plain K1 no store")
        RUNW(false, 1, 1, "W plain K1 store")
        RUNW(false, 1, 2, "W plain K1 store confined 512KB")
        RUNW(false, 1, 3, "W plain K1 nt store")
        RUNW(false, 1, 5, "W plain K1 sc0sc1 store")
        RUNW(true, 1, 0, "W nt K1 no store")
        RUNW(true, 1, 1, "W nt K1 store")
        RUNW(true, 1, 2, "W nt K1 store confined 512KB")
        RUNW(false, 4, 0, "W plain K4 no store")
        RUNW(false, 4, 1, "W plain K4 store as produced")
        RUNW(false, 4, 4, "W plain K4 store at end")
        RUNW(false, 4, 6, "W plain K4 nt store at end")
        RUNW(false, 4, 5, "W plain K4 sc0sc1 store at end")
        RUNW(false, 8, 0, "W plain K8 no store")
        RUNW(false, 8, 1, "W plain K8 store as produced")
        RUNW(false, 8, 4, "W plain K8 store at end")
        RUNW(false, 8, 6, "W plain K8 nt store at end")
        RUNW(true, 8, 4, "W nt K8 store at end")
        RUNW(false, 16, 0, "W plain K16 no store")
        RUNW(false, 16, 4, "W plain K16 store at end")
        RUNW(false, 16, 6, "W plain K16 nt store at end")
        return 0;
    }
#define RUNV2(NTv, ST, RPTv, KCHv, mapv, label, verify) \
    for (int wpc : wpcs) { CK(hipMemset(d_y, 0xff, N * 8)); \
        double ms = time_ms([&] { hipLaunchKernelGGL((v2_rows<NTv, ST, RPTv, KCHv, 1>), dim3(wpc * 256), dim3(64), 0, 0, a, mapv, n * n / 64); }, reps); \
        bool ok = verify ? check(label) : true; report(label, wpc, ms, ok); }
#define RUNV3(NTv, ST, RPTv, KCHv, WPBv, mapv, label, verify) \
    for (int wpc : wpcs) { CK(hipMemset(d_y, 0xff, N * 8)); \
        double ms = time_ms([&] { hipLaunchKernelGGL((v2_rows<NTv, ST, RPTv, KCHv, WPBv>), dim3(wpc * 256 / WPBv), dim3(64 * WPBv), 0, 0, a, mapv, n * n / 64); }, reps); \
        bool ok = verify ? check(label) : true; report(label, wpc, ms, ok); }
    if (argc > 3 && !strcmp(argv[3], "scan")) {
        struct Cfg { const char* name; void (*fn)(Args, int, int); int wpb; };
        std::vector<Cfg> cfgs;
#define ADD(ST, R, K, W, NGv) cfgs.push_back({ "st" #ST " R" #R " K" #K " WPB" #W " NG" #NGv, v2_rows<false, ST, R, K, W, NGv>, W });
#define ADDW(ST, R, K, NGv) ADD(ST, R, K, 1, NGv) ADD(ST, R, K, 2, NGv) ADD(ST, R, K, 4, NGv) ADD(ST, R, K, 8, NGv)
        ADDW(5, 1, 1, 8) ADDW(5, 1, 2, 8) ADDW(5, 1, 4, 8) ADDW(5, 1, 8, 8)
        ADDW(5, 1, 1, 7) ADDW(5, 1, 4, 7)
        ADDW(4, 1, 1, 7)
        ADD(5, 2, 2, 1, 8) ADD(5, 2, 2, 2, 8) ADD(5, 2, 2, 4, 8) ADD(5, 2, 4, 2, 8) ADD(5, 2, 8, 2, 8) ADD(5, 2, 2, 2, 7) ADD(5, 2, 2, 4, 7)
        std::vector<int> gs = { 1024, 1280, 1536, 1792, 2048, 2304, 2560, 3072, 3584, 4096 };
        if (argc > 4) { gs.clear(); char* tok = strtok(argv[4], ","); while (tok) { gs.push_back(atoi(tok)); tok = strtok(nullptr, ","); } }
        const bool brief = argc > 5;
        if (brief) { std::vector<Cfg> keep; for (auto& c : cfgs) if (strstr(c.name, "st5 R1 K1 ") || strstr(c.name, "st5 R1 K4 ")) if (strstr(c.name, "NG7")) keep.push_back(c); cfgs = keep; }
        printf("%-26s", "config \\ waves");
        for (int G : gs) printf(" %7d", G);
        printf("\n");
        double best = 1e9; std::string bestName;
        for (auto& c : cfgs) {
            printf("%-26s", c.name);
            for (int G : gs) {
                if (G % c.wpb) { printf("       -"); continue; }
                CK(hipMemset(d_y, 0xff, N * 8));
                double ms = time_ms([&] { hipLaunchKernelGGL(c.fn, dim3(G / c.wpb), dim3(64 * c.wpb), 0, 0, a, 0, n * n / 64); }, reps);
                bool ok = check(c.name);
                printf(" %7.3f%s", ms, ok ? "" : "!");
                if (ok && ms < best) { best = ms; bestName = std::string(c.name) + " G" + std::to_string(G); }
            }
            printf("\n"); fflush(stdout);
        }
        printf("best %.3f ms (%.1f%% of 8 TB/s): %s\n", best, algo / best / 1e6 / 80.0, bestName.c_str());
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "pmc")) {
        // a few single launches for rocprofv3 --pmc (told apart by their grid size)
        for (int G : { 2048, 3072, 4096, 2560 }) {
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((v2_rows<false, 5, 1, 1, 4, 7>), dim3(G / 4), dim3(256), 0, 0, a, 0, n * n / 64);
            CK(hipDeviceSynchronize());
        }
        for (int G : { 2048, 3072 }) {
            for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((v2_rows<false, 5, 1, 1, 1, 7>), dim3(G), dim3(64), 0, 0, a, 0, n * n / 64);
            CK(hipDeviceSynchronize());
        }
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((v2_rows<false, 4, 1, 1, 4, 7>), dim3(512), dim3(256), 0, 0, a, 5, n * n / 64);     // map 5 (st4 to tell it apart)
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((v2_rows<false, 4, 1, 1, 4, 8>), dim3(256), dim3(256), 0, 0, a, 5, n * n / 64);     // map 5, 4 waves per CU
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((v2_rows<true, 5, 1, 1, 4, 7>), dim3(512), dim3(256), 0, 0, a, 0, n * n / 64);     // nt streams
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((v2_rows<false, 5, 1, 1, 4, 8>), dim3(512), dim3(256), 0, 0, a, 3, n * n / 64);    // map 3 (NG 8 to tell it apart)
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((v2_rows<false, 5, 1, 1, 8, 8>), dim3(256), dim3(512), 0, 0, a, 3, n * n / 64);    // map 3, 8 waves
        CK(hipDeviceSynchronize());
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "wide")) {
        wpcs = { 8, 16 };
        for (int rep = 0; rep < 2; ++rep) {
            RUNW(false, 2, 0, "W K2 no store")
            RUNW(false, 2, 4, "W K2 2 x 512 B at end")
            RUNW(false, 2, 7, "W K2 1 x 1 KiB (16 B/lane)")
            RUNW(false, 2, 8, "W K2 1 x 1 KiB nt")
            RUNW(false, 4, 0, "W K4 no store")
            RUNW(false, 4, 4, "W K4 4 x 512 B at end")
            RUNW(false, 4, 7, "W K4 2 x 1 KiB (16 B/lane)")
            RUNW(false, 4, 8, "W K4 2 x 1 KiB nt")
            RUNW(false, 8, 0, "W K8 no store")
            RUNW(false, 8, 7, "W K8 4 x 1 KiB (16 B/lane)")
            RUNW(false, 8, 8, "W K8 4 x 1 KiB nt")
        }
        return 0;
    }
#define RUNV4(NGv, Bv, mapv, label) \
    { CK(hipMemset(d_y, 0xff, N * 8)); \
        double ms = time_ms([&] { hipLaunchKernelGGL((v4_rows<4, NGv, Bv>), dim3(512), dim3(256), 0, 0, a, mapv, n * n / 64); }, reps); \
        bool ok = check(label); report(label, 8, ms, ok); }
    if (argc > 3 && !strcmp(argv[3], "interleave")) {
        const int nTiles = (int)(N / 256);
        const long long recBytes = (long long)nTiles * 22528;
        d2* rec; CK(hipMalloc(&rec, recBytes + 4096)); CK(hipMemset(rec, 0x11, recBytes + 4096));
        printf("one interleaved stream (%.2f GB) vs the three CSR arrays (%.2f GB), bare streaming, 512 workgroups of 256\n", recBytes / 1e9, (12.0 * nnz + 4.0 * N) / 1e9);
        for (int rep = 0; rep < 3; ++rep) {
            for (int g : { 512, 1024 }) {
                double a0 = time_ms([&] { hipLaunchKernelGGL((i_stream<0, 0>), dim3(g), dim3(256), 0, 0, rec, recBytes, nTiles, a); }, reps);
                double a1 = time_ms([&] { hipLaunchKernelGGL((i_stream<1, 1>), dim3(g), dim3(256), 0, 0, rec, recBytes, nTiles, a); }, reps);
                double b0 = time_ms([&] { hipLaunchKernelGGL((c_stream<0, 0>), dim3(g), dim3(256), 0, 0, a, nTiles); }, reps);
                double b1 = time_ms([&] { hipLaunchKernelGGL((c_stream<1, 1>), dim3(g), dim3(256), 0, 0, a, nTiles); }, reps);
                printf("  grid %4d: interleaved %.3f ms, + y store + x,w %.3f ms | three arrays %.3f ms, + y store + x,w %.3f ms\n", g, a0, a1, b0, b1);
            }
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "records")) {
        const int nBlk = (int)(N / 64);
        unsigned* d_units; long long* d_off; CK(hipMalloc(&d_units, nBlk * 4)); CK(hipMalloc(&d_off, (nBlk + 1) * 8));
        hipLaunchKernelGGL(rec_size_kernel, dim3(2048), dim3(256), 0, 0, d_ro, nBlk, (int)N, d_units);
        std::vector<unsigned> units(nBlk); CK(hipMemcpy(units.data(), d_units, nBlk * 4, hipMemcpyDeviceToHost));
        std::vector<long long> off(nBlk + 1); off[0] = 0; for (int b = 0; b < nBlk; ++b) off[b + 1] = off[b] + units[b];
        CK(hipMemcpy(d_off, off.data(), (nBlk + 1) * 8, hipMemcpyHostToDevice));
        char* d_rec; CK(hipMalloc(&d_rec, off[nBlk] * 16 + 4096));
        double tb = time_ms([&] { hipLaunchKernelGGL(rec_fill_kernel, dim3(16384), dim3(64), 0, 0, d_ro, d_col, d_val, nBlk, (int)N, d_off, d_rec); }, 3);
        printf("block records: %.3f GB (CSR arrays %.3f GB), built in %.2f ms\n", off[nBlk] * 16 / 1e9, (12.0 * nnz + 4.0 * (N + 1)) / 1e9, tb);
        for (int rep = 0; rep < 3; ++rep) {
            wpcs = { 8 };
            RUNV3(false, 5, 1, 1, 4, 5, "CSR arrays, NG8, z sweep", true)
            { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v5_rows<8>), dim3(512), dim3(256), 0, 0, a, d_off, d_rec, 5, n * n / 64); }, reps); bool ok = check("v5"); report("block records, NG8, z sweep", 8, ms, ok); }
            { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v5_rows<7>), dim3(512), dim3(256), 0, 0, a, d_off, d_rec, 5, n * n / 64); }, reps); bool ok = check("v5"); report("block records, NG7, z sweep", 8, ms, ok); }
            { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v5_rows<8>), dim3(512), dim3(256), 0, 0, a, d_off, d_rec, 0, n * n / 64); }, reps); bool ok = check("v5"); report("block records, NG8, grid-stride", 8, ms, ok); }
            { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v5_rows<8>), dim3(1024), dim3(256), 0, 0, a, d_off, d_rec, 0, n * n / 64); }, reps); bool ok = check("v5"); report("block records, NG8, grid-stride", 16, ms, ok); }
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "clock")) {
        wpcs = { 8 };
        for (int rep = 0; rep < 2; ++rep) {
            RUNV3(false, 5, 1, 1, 4, 5, "baseline (store every trip), NG8", true)
            { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v2_rows<false, 5, 1, 1, 4, 7>), dim3(512), dim3(256), 0, 0, a, 5, n * n / 64); }, reps); bool ok = check("b7"); report("baseline (store every trip), NG7", 8, ms, ok); }
            for (int period : { 500, 1000, 2000, 4000, 100000000 }) {
                char label[96];
                { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v6_rows<7, 8>), dim3(512), dim3(256), 0, 0, a, 5, n * n / 64, period); }, reps); bool ok = check("v6"); snprintf(label, sizeof label, "clocked flush B8, period %d ticks", period); report(label, 8, ms, ok); }
                { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v6_rows<7, 16>), dim3(512), dim3(256), 0, 0, a, 5, n * n / 64, period); }, reps); bool ok = check("v6"); snprintf(label, sizeof label, "clocked flush B16, period %d ticks", period); report(label, 8, ms, ok); }
            }
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "order")) {
        wpcs = { 8 };
        for (int rep = 0; rep < 3; ++rep) {
            RUNV3(false, 5, 1, 1, 4, 0, "grid-stride (tile = wg + WG t)", true)
            RUNV3(false, 5, 1, 1, 4, 5, "z sweep (plane by plane per half)", true)
            RUNV3(false, 5, 1, 1, 4, 6, "memory order, XCD-contiguous eighths", true)
        }
        wpcs = { 16 };
        RUNV3(false, 5, 1, 1, 4, 0, "grid-stride", true)
        RUNV3(false, 5, 1, 1, 4, 5, "z sweep (H = 1: whole planes)", true)
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "micro")) {
        wpcs = { 8 };
        for (int rep = 0; rep < 3; ++rep) {
            RUNV3(false, 5, 1, 1, 4, 5, "NG8 two barriers", true)
            RUNV3(false, 5, 1, 1, 4, 5 + 256, "NG8 one barrier", true)
            { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v2_rows<false, 5, 1, 1, 4, 7>), dim3(512), dim3(256), 0, 0, a, 5, n * n / 64); }, reps); bool ok = check("ng7"); report("NG7 two barriers", 8, ms, ok); }
            { CK(hipMemset(d_y, 0xff, N * 8)); double ms = time_ms([&] { hipLaunchKernelGGL((v2_rows<false, 5, 1, 1, 4, 7>), dim3(512), dim3(256), 0, 0, a, 5 + 256, n * n / 64); }, reps); bool ok = check("ng7"); report("NG7 one barrier", 8, ms, ok); }
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "batch")) {
        for (int rep = 0; rep < 2; ++rep) {
            RUNV4(7, 1, 5, "V4 NG7 B1 sweep")
            RUNV4(7, 2, 5, "V4 NG7 B2 sweep")
            RUNV4(7, 4, 5, "V4 NG7 B4 sweep")
            RUNV4(7, 8, 5, "V4 NG7 B8 sweep")
            RUNV4(7, 16, 5, "V4 NG7 B16 sweep")
            RUNV4(7, 32, 5, "V4 NG7 B32 sweep")
            RUNV4(7, 64, 5, "V4 NG7 B64 sweep")
            RUNV4(8, 1, 5, "V4 NG8 B1 sweep")
            RUNV4(7, 1, 0, "V4 NG7 B1 map0")
            RUNV4(7, 8, 0, "V4 NG7 B8 map0")
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "aux")) {
        wpcs = { 8 };
        for (int rep = 0; rep < 2; ++rep) {
            RUNV3(false, 5, 1, 1, 4, 5, "builtin nt store", true)
            RUNV3(false, 100, 1, 1, 4, 5, "buffer store aux 0", true)
            RUNV3(false, 101, 1, 1, 4, 5, "buffer store sc0", true)
            RUNV3(false, 102, 1, 1, 4, 5, "buffer store nt", true)
            RUNV3(false, 103, 1, 1, 4, 5, "buffer store sc0 nt", true)
            RUNV3(false, 116, 1, 1, 4, 5, "buffer store sc1", true)
            RUNV3(false, 117, 1, 1, 4, 5, "buffer store sc0 sc1", true)
            RUNV3(false, 118, 1, 1, 4, 5, "buffer store sc1 nt", true)
            RUNV3(false, 119, 1, 1, 4, 5, "buffer store sc0 sc1 nt", true)
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "sweep")) {
        for (int rep = 0; rep < 2; ++rep) {
            wpcs = { 8 };
            RUNV3(false, 5, 1, 1, 4, 0, "st5 K1 WPB4 map0", true)
            RUNV3(false, 5, 1, 1, 4, 5, "st5 K1 WPB4 map5 (z sweep, H=2)", true)
            RUNV3(false, 4, 1, 1, 4, 5, "st4 K1 WPB4 map5 (z sweep, H=2)", true)
            RUNV3(true, 5, 1, 1, 4, 5, "nt st5 K1 WPB4 map5", true)
            RUNV3(false, 0, 1, 1, 4, 5, "no store K1 WPB4 map5", false)
            wpcs = { 4, 16 };
            RUNV3(false, 5, 1, 1, 4, 5, "st5 K1 WPB4 map5 (H=4 / H=1)", true)
            wpcs = { 8, 16 };
            RUNV3(false, 5, 1, 1, 8, 5, "st5 K1 WPB8 map5", true)
            RUNV3(false, 5, 1, 1, 2, 5, "st5 K1 WPB2 map5", true)
            RUNV3(false, 5, 1, 1, 1, 5, "st5 K1 WPB1 map5", true)
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "xcost")) {
        wpcs = { 8, 16 };
        for (int rep = 0; rep < 2; ++rep) {
            a.gmask = -1;
            RUNV3(false, 5, 1, 1, 4, 0, "st5 K1 WPB4", true)
            RUNV3(false, 0, 1, 1, 4, 0, "no store K1 WPB4", false)
            a.gmask = 1023;
            RUNV3(false, 5, 1, 1, 4, 0, "st5 K1 WPB4 | gathers from L1", false)
            RUNV3(false, 0, 1, 1, 4, 0, "no store K1 WPB4 | gathers from L1", false)
            a.gmask = 0xFFFFF;   // 1 M doubles = 8 MB: gathers from L2 / MALL, no HBM
            RUNV3(false, 5, 1, 1, 4, 0, "st5 K1 WPB4 | gathers in 8 MB", false)
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "chunk")) {
        wpcs = { 8, 16 };
        RUNV3(false, 5, 1, 1, 4, 0, "st5 K1 WPB4 map0", true)
        RUNV3(false, 5, 1, 4, 4, 0, "st5 K4 WPB4 map0 (per-wave chunks)", true)
        RUNV3(false, 5, 1, 1, 4, 32, "st5 K1 WPB4 wg-chunk 2", true)
        RUNV3(false, 5, 1, 1, 4, 64, "st5 K1 WPB4 wg-chunk 4", true)
        RUNV3(false, 5, 1, 1, 4, 128, "st5 K1 WPB4 wg-chunk 8", true)
        RUNV3(false, 5, 1, 1, 4, 256, "st5 K1 WPB4 wg-chunk 16", true)
        RUNV3(false, 5, 1, 1, 4, 1024, "st5 K1 WPB4 wg-chunk 64", true)
        RUNV3(false, 5, 1, 1, 8, 32, "st5 K1 WPB8 wg-chunk 2", true)
        RUNV3(false, 5, 1, 1, 8, 64, "st5 K1 WPB8 wg-chunk 4", true)
        RUNV3(false, 5, 1, 1, 8, 128, "st5 K1 WPB8 wg-chunk 8", true)
        RUNV3(false, 5, 1, 1, 2, 64, "st5 K1 WPB2 wg-chunk 4", true)
        RUNV3(false, 5, 1, 1, 2, 128, "st5 K1 WPB2 wg-chunk 8", true)
        RUNV3(false, 5, 1, 1, 1, 64, "st5 K1 WPB1 wg-chunk 4", true)
        RUNV3(false, 5, 1, 1, 1, 128, "st5 K1 WPB1 wg-chunk 8", true)
        RUNV3(false, 4, 1, 1, 4, 64, "st4 K1 WPB4 wg-chunk 4", true)
        RUNV3(false, 0, 1, 1, 4, 64, "no store K1 WPB4 wg-chunk 4", false)
        RUNV3(false, 0, 1, 1, 4, 0, "no store K1 WPB4 map0", false)
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "map")) {
        wpcs = { 8, 16 };
        RUNV3(false, 5, 1, 1, 4, 0, "st5 R1 K1 WPB4 map0", true)
        RUNV3(false, 5, 1, 1, 4, 3, "st5 R1 K1 WPB4 map3", true)
        RUNV3(false, 5, 1, 1, 8, 3, "st5 R1 K1 WPB8 map3", true)
        RUNV3(false, 5, 1, 1, 2, 3, "st5 R1 K1 WPB2 map3", true)
        RUNV3(true, 5, 1, 1, 4, 3, "nt st5 R1 K1 WPB4 map3", true)
        RUNV3(true, 5, 1, 1, 8, 3, "nt st5 R1 K1 WPB8 map3", true)
        RUNV3(true, 5, 1, 1, 4, 0, "nt st5 R1 K1 WPB4 map0", true)
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "occ")) {
        // occupancy pinned by LDS padding: exactly wgPerCu workgroups fit a CU, grid = wgPerCu * 256 -> every CU holds the same number
        struct Cfg { const char* name; void (*fn)(Args, int, int); int wpb; int staticLds; };
        std::vector<Cfg> cfgs;
#define ADDO(ST, R, K, W, NGv) cfgs.push_back({ "st" #ST " R" #R " K" #K " WPB" #W " NG" #NGv, v2_rows<false, ST, R, K, W, NGv>, W, 6144 * R * W + 16 });
        ADDO(5, 1, 1, 1, 7) ADDO(5, 1, 1, 2, 7) ADDO(5, 1, 1, 4, 7) ADDO(5, 1, 1, 8, 7) ADDO(5, 1, 4, 4, 7) ADDO(5, 1, 4, 8, 7) ADDO(5, 1, 8, 4, 7) ADDO(5, 2, 2, 2, 7) ADDO(5, 2, 2, 4, 7) ADDO(5, 2, 8, 2, 7)
        const int wpcs2[] = { 4, 6, 8, 10, 12, 16, 20 };
        printf("%-26s", "config \\ waves/CU (pinned)");
        for (int w : wpcs2) printf(" %7d", w);
        printf("\n");
        for (auto& c : cfgs) {
            printf("%-26s", c.name);
            for (int w : wpcs2) {
                if (w % c.wpb) { printf("       -"); continue; }
                const int wgPerCu = w / c.wpb;
                int per = (160 * 1024) / wgPerCu; per = (per / 1280) * 1280;      // allocation granule
                const int dyn = per - c.staticLds - 256;
                if (dyn < 0) { printf("       -"); continue; }
                CK(hipFuncSetAttribute((const void*)c.fn, hipFuncAttributeMaxDynamicSharedMemorySize, dyn));
                CK(hipMemset(d_y, 0xff, N * 8));
                double ms = time_ms([&] { hipLaunchKernelGGL(c.fn, dim3(wgPerCu * 256), dim3(64 * c.wpb), dyn, 0, a, 0, n * n / 64); }, reps);
                bool ok = check(c.name);
                printf(" %7.3f%s", ms, ok ? "" : "!");
            }
            printf("\n"); fflush(stdout);
        }
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "v3")) {
        wpcs = { 4, 8, 12, 16 };
        printf("-- lane = row; WPB waves per workgroup issue together\n");
        RUNV3(false, 4, 1, 1, 1, 0, "V3 st4 R1 K1 WPB1", true)
        RUNV3(false, 4, 1, 1, 2, 0, "V3 st4 R1 K1 WPB2", true)
        RUNV3(false, 4, 1, 1, 4, 0, "V3 st4 R1 K1 WPB4", true)
        RUNV3(false, 4, 1, 1, 8, 0, "V3 st4 R1 K1 WPB8", true)
        RUNV3(false, 5, 1, 1, 4, 0, "V3 st5 R1 K1 WPB4", true)
        RUNV3(false, 5, 1, 1, 8, 0, "V3 st5 R1 K1 WPB8", true)
        RUNV3(false, 4, 1, 4, 4, 0, "V3 st4 R1 K4 WPB4", true)
        RUNV3(false, 4, 1, 4, 8, 0, "V3 st4 R1 K4 WPB8", true)
        RUNV3(false, 4, 2, 2, 2, 0, "V3 st4 R2 K2 WPB2", true)
        RUNV3(false, 4, 2, 2, 4, 0, "V3 st4 R2 K2 WPB4", true)
        RUNV3(false, 5, 2, 2, 4, 0, "V3 st5 R2 K2 WPB4", true)
        RUNV3(false, 0, 1, 1, 4, 0, "V3 no store R1 K1 WPB4", false)
        RUNV3(false, 0, 1, 1, 8, 0, "V3 no store R1 K1 WPB8", false)
        RUNV3(true, 4, 1, 1, 4, 0, "V3 nt st4 R1 K1 WPB4", true)
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "v2")) {
        wpcs = { 4, 6, 8, 12, 16 };
        printf("-- bare streams: K blocks in flight per wave, contiguous or a grid-stride apart\n");
        RUNW(false, 4, 0, "W plain K4 contiguous no store")
        RUNW(false, 4, 10, "W plain K4 strided no store")
        RUNW(false, 8, 0, "W plain K8 contiguous no store")
        RUNW(false, 8, 10, "W plain K8 strided no store")
        printf("-- lane = row, RPT blocks per trip, KCH blocks per super block, map\n");
        RUNV2(false, 1, 1, 1, 0, "V2 plain st1 R1 K1 map0", true)
        RUNV2(false, 4, 1, 1, 0, "V2 plain st4 R1 K1 map0", true)
        RUNV2(false, 4, 1, 4, 0, "V2 plain st4 R1 K4 map0", true)
        RUNV2(false, 4, 1, 8, 0, "V2 plain st4 R1 K8 map0", true)
        RUNV2(false, 4, 1, 4, 1, "V2 plain st4 R1 K4 map1", true)
        RUNV2(false, 4, 1, 4, 2, "V2 plain st4 R1 K4 map2", true)
        RUNV2(false, 4, 1, 8, 2, "V2 plain st4 R1 K8 map2", true)
        RUNV2(false, 5, 1, 4, 0, "V2 plain st5(nt) R1 K4 map0", true)
        RUNV2(true, 4, 1, 4, 0, "V2 nt st4 R1 K4 map0", true)
        RUNV2(true, 4, 1, 4, 2, "V2 nt st4 R1 K4 map2", true)
        RUNV2(false, 4, 2, 2, 0, "V2 plain st4 R2 K2 map0", true)
        RUNV2(false, 4, 2, 4, 0, "V2 plain st4 R2 K4 map0", true)
        RUNV2(false, 4, 2, 8, 0, "V2 plain st4 R2 K8 map0", true)
        RUNV2(false, 4, 2, 8, 2, "V2 plain st4 R2 K8 map2", true)
        RUNV2(false, 5, 2, 8, 0, "V2 plain st5(nt) R2 K8 map0", true)
        RUNV2(true, 4, 2, 8, 0, "V2 nt st4 R2 K8 map0", true)
        RUNV2(false, 0, 2, 8, 0, "V2 plain no store R2 K8 map0", false)
        RUNV2(false, 0, 1, 4, 0, "V2 plain no store R1 K4 map0", false)
        return 0;
    }
    wpcs = { 8, 12, 16, 24 };
    printf("-- streams only (ceiling of the access pattern)\n");
    RUNS(false, 0, 0, "S  matrix streams")
    RUNS(true, 0, 0, "S  matrix streams nt")
    RUNS(true, 1, 0, "S  nt + y store")
    RUNS(true, 1, 1, "S  nt + y store + x,w reads")
    RUNS(false, 1, 1, "S  plain + y store + x,w reads")
    printf("-- lane = row kernels\n");
    RUNV(false, 1, 0, "V  plain, store after products", true)
    RUNV(true, 1, 0, "V  nt, store after products", true)
    RUNV(true, 2, 0, "V  nt, nt store", true)
    RUNV(true, 3, 0, "V  nt, deferred store first", true)
    RUNV(true, 4, 0, "V  nt, deferred store mid", true)
    RUNV(false, 4, 0, "V  plain, deferred store mid", true)
    RUNV(true, 0, 0, "V  nt, no store", false)
    RUNV(true, 1, 1, "V  nt, gathers from L1", false)
    RUNV(true, 0, 1, "V  nt, no store, L1 gathers", false)
    return 0;
}
