// vec_probe: which launch shape streams the CG vector updates fastest on this MI355X?  (measurement tool, not product)
//   xp: x += a p ; p = z + b p   (reads x,p,z  writes x,p : 40 B/elt)      r: r -= a Ap, sum r.r  (24 B/elt)
// Variants: grid size, 16-byte accesses per thread per trip (U), non-temporal stores / loads, contiguous chunk per workgroup.
// Build: make -C conjugategradient_amd/csrc vecprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <string>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int U, bool NTS, bool NTL, bool CHUNK>
__global__ __launch_bounds__(256) void xp_kernel(d2* __restrict__ x, d2* __restrict__ p, const d2* __restrict__ z, long long n2, double a, double b)
{
    long long i, end, step;
    if (CHUNK) {
        const long long per = ((n2 + gridDim.x - 1) / gridDim.x + 255) & ~255LL;
        i = per * blockIdx.x + threadIdx.x; end = per * (blockIdx.x + 1); if (end > n2) end = n2; step = 256;
    } else { i = (long long)blockIdx.x * 256 + threadIdx.x; end = n2; step = (long long)gridDim.x * 256; }
    for (; i + (U - 1) * step < end; i += U * step) {
        d2 pv[U], xv[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pv[u] = NTL ? __builtin_nontemporal_load(p + i + u * step) : p[i + u * step];
            xv[u] = NTL ? __builtin_nontemporal_load(x + i + u * step) : x[i + u * step];
            zv[u] = NTL ? __builtin_nontemporal_load(z + i + u * step) : z[i + u * step];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u].x += a * pv[u].x; xv[u].y += a * pv[u].y;
            pv[u].x = zv[u].x + b * pv[u].x; pv[u].y = zv[u].y + b * pv[u].y;
            if (NTS) { __builtin_nontemporal_store(xv[u], x + i + u * step); __builtin_nontemporal_store(pv[u], p + i + u * step); }
            else { x[i + u * step] = xv[u]; p[i + u * step] = pv[u]; }
        }
    }
    for (; i < end; i += step) { d2 pv = p[i], xv = x[i], zv = z[i]; xv.x += a * pv.x; xv.y += a * pv.y; pv.x = zv.x + b * pv.x; pv.y = zv.y + b * pv.y; x[i] = xv; p[i] = pv; }
}

template <int U, bool NTS, bool CHUNK>
__global__ __launch_bounds__(256) void r_kernel(d2* __restrict__ r, const d2* __restrict__ Ap, long long n2, double a, double* partials)
{
    __shared__ double red[4];
    long long i, end, step;
    if (CHUNK) {
        const long long per = ((n2 + gridDim.x - 1) / gridDim.x + 255) & ~255LL;
        i = per * blockIdx.x + threadIdx.x; end = per * (blockIdx.x + 1); if (end > n2) end = n2; step = 256;
    } else { i = (long long)blockIdx.x * 256 + threadIdx.x; end = n2; step = (long long)gridDim.x * 256; }
    double acc = 0;
    for (; i + (U - 1) * step < end; i += U * step) {
        d2 rv[U], av[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { rv[u] = r[i + u * step]; av[u] = Ap[i + u * step]; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            rv[u].x -= a * av[u].x; rv[u].y -= a * av[u].y; acc += rv[u].x * rv[u].x + rv[u].y * rv[u].y;
            if (NTS) __builtin_nontemporal_store(rv[u], r + i + u * step); else r[i + u * step] = rv[u];
        }
    }
    for (; i < end; i += step) { d2 rv = r[i], av = Ap[i]; rv.x -= a * av.x; rv.y -= a * av.y; acc += rv.x * rv.x + rv.y * rv.y; r[i] = rv; }
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// chunk mapping with a compile-time workgroup size (the library uses 256)
template <int BLOCK, int U, bool NT>
__global__ __launch_bounds__(BLOCK) void xp_chunk_kernel(d2* __restrict__ x, d2* __restrict__ p, const d2* __restrict__ z, long long n2, double a, double b)
{
    const long long per = ((n2 + gridDim.x - 1) / gridDim.x + (BLOCK - 1)) & ~(long long)(BLOCK - 1);
    long long i = per * blockIdx.x + threadIdx.x, end = per * (blockIdx.x + 1);
    if (end > n2) end = n2;
    for (; i + (U - 1) * BLOCK < end; i += U * BLOCK) {
        d2 pv[U], xv[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            pv[u] = NT ? __builtin_nontemporal_load(p + i + u * BLOCK) : p[i + u * BLOCK];
            xv[u] = NT ? __builtin_nontemporal_load(x + i + u * BLOCK) : x[i + u * BLOCK];
            zv[u] = NT ? __builtin_nontemporal_load(z + i + u * BLOCK) : z[i + u * BLOCK];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            xv[u].x += a * pv[u].x; xv[u].y += a * pv[u].y;
            pv[u].x = zv[u].x + b * pv[u].x; pv[u].y = zv[u].y + b * pv[u].y;
            if (NT) { __builtin_nontemporal_store(xv[u], x + i + u * BLOCK); __builtin_nontemporal_store(pv[u], p + i + u * BLOCK); }
            else { x[i + u * BLOCK] = xv[u]; p[i + u * BLOCK] = pv[u]; }
        }
    }
    for (; i < end; i += BLOCK) { d2 pv = p[i], xv = x[i], zv = z[i]; xv.x += a * pv.x; xv.y += a * pv.y; pv.x = zv.x + b * pv.x; pv.y = zv.y + b * pv.y; x[i] = xv; p[i] = pv; }
}
template <int BLOCK, int U, bool NT>
static void go_xpc(int grid, d2* x, d2* p, d2* z, long long n2, double*) { hipLaunchKernelGGL((xp_chunk_kernel<BLOCK, U, NT>), dim3(grid), dim3(BLOCK), 0, 0, x, p, z, n2, 1e-9, 0.999999); }

struct Variant { std::string name; int kind; int grid; void (*launch)(int, d2*, d2*, d2*, long long, double*); };

template <int U, bool NTS, bool NTL, bool CHUNK>
static void go_xp(int grid, d2* x, d2* p, d2* z, long long n2, double*) { hipLaunchKernelGGL((xp_kernel<U, NTS, NTL, CHUNK>), dim3(grid), dim3(256), 0, 0, x, p, z, n2, 1e-9, 0.999999); }
template <int U, bool NTS, bool CHUNK>
static void go_r(int grid, d2* x, d2* p, d2*, long long n2, double* part) { hipLaunchKernelGGL((r_kernel<U, NTS, CHUNK>), dim3(grid), dim3(256), 0, 0, x, p, n2, 1e-9, part); }

int main(int argc, char** argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 134217728LL;
    const int rounds = argc > 2 ? atoi(argv[2]) : 7;
    const long long n2 = n / 2;
    const long long skewBytes = argc > 3 ? atoll(argv[3]) : 0;      // p and z start skewBytes, 2*skewBytes past their allocation
    char *xb, *pb, *zb; double* part;
    CK(hipMalloc(&xb, n * 8 + 3 * skewBytes)); CK(hipMalloc(&pb, n * 8 + 3 * skewBytes)); CK(hipMalloc(&zb, n * 8 + 3 * skewBytes)); CK(hipMalloc(&part, 65536 * 8));
    CK(hipMemset(xb, 0, n * 8)); CK(hipMemset(pb, 0, n * 8 + 3 * skewBytes)); CK(hipMemset(zb, 0, n * 8 + 3 * skewBytes));
    d2* x = (d2*)xb; d2* p = (d2*)(pb + skewBytes); d2* z = (d2*)(zb + 2 * skewBytes);
    printf("x %p  p %p  z %p  (skew %lld B)\n", (void*)x, (void*)p, (void*)z, skewBytes);
    std::vector<Variant> vs;
    for (int g : { 1024, 2048, 4096, 8192 }) {
        vs.push_back({ "xpc B128 U2 nt g" + std::to_string(g), 0, g, go_xpc<128, 2, true> });
        vs.push_back({ "xpc B128 U4 nt g" + std::to_string(g), 0, g, go_xpc<128, 4, true> });
        vs.push_back({ "xpc B256 U2 nt g" + std::to_string(g), 0, g, go_xpc<256, 2, true> });
        vs.push_back({ "xpc B256 U4 nt g" + std::to_string(g), 0, g, go_xpc<256, 4, true> });
        vs.push_back({ "xpc B512 U2 nt g" + std::to_string(g), 0, g, go_xpc<512, 2, true> });
        vs.push_back({ "xpc B512 U1 nt g" + std::to_string(g), 0, g, go_xpc<512, 1, true> });
        vs.push_back({ "xpc B256 U2 g" + std::to_string(g), 0, g, go_xpc<256, 2, false> });
    }
    for (int g : { 2048 }) {
        vs.push_back({ "xp U1 chunk g" + std::to_string(g), 0, g, go_xp<1, false, false, true> });
        vs.push_back({ "xp U4 chunk g" + std::to_string(g), 0, g, go_xp<4, false, false, true> });
        vs.push_back({ "xp U2 chunk nts ntl g" + std::to_string(g), 0, g, go_xp<2, true, true, true> });
        vs.push_back({ "xp U2 chunk nts g" + std::to_string(g), 0, g, go_xp<2, true, false, true> });
        vs.push_back({ "r U1 chunk g" + std::to_string(g), 1, g, go_r<1, false, true> });
        vs.push_back({ "r U4 chunk g" + std::to_string(g), 1, g, go_r<4, false, true> });
        vs.push_back({ "r U2 chunk nts g" + std::to_string(g), 1, g, go_r<2, true, true> });
        vs.push_back({ "xp U1 g" + std::to_string(g), 0, g, go_xp<1, false, false, false> });
        vs.push_back({ "xp U2 g" + std::to_string(g), 0, g, go_xp<2, false, false, false> });
        vs.push_back({ "xp U4 g" + std::to_string(g), 0, g, go_xp<4, false, false, false> });
        vs.push_back({ "xp U2 nts g" + std::to_string(g), 0, g, go_xp<2, true, false, false> });
        vs.push_back({ "xp U2 nts ntl g" + std::to_string(g), 0, g, go_xp<2, true, true, false> });
        vs.push_back({ "xp U2 chunk g" + std::to_string(g), 0, g, go_xp<2, false, false, true> });
        vs.push_back({ "r U1 g" + std::to_string(g), 1, g, go_r<1, false, false> });
        vs.push_back({ "r U2 g" + std::to_string(g), 1, g, go_r<2, false, false> });
        vs.push_back({ "r U4 g" + std::to_string(g), 1, g, go_r<4, false, false> });
        vs.push_back({ "r U2 nts g" + std::to_string(g), 1, g, go_r<2, true, false> });
        vs.push_back({ "r U2 chunk g" + std::to_string(g), 1, g, go_r<2, false, true> });
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<std::vector<float>> ms(vs.size());
    for (int r = -1; r < rounds; ++r)
        for (size_t k = 0; k < vs.size(); ++k) {
            CK(hipEventRecord(e0, 0));
            for (int rep = 0; rep < 3; ++rep) vs[k].launch(vs[k].grid, x, p, z, n2, part);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float t; CK(hipEventElapsedTime(&t, e0, e1));
            if (r >= 0) ms[k].push_back(t / 3);
        }
    CK(hipGetLastError());
    printf("n = %lld doubles per vector\n", n);
    for (size_t k = 0; k < vs.size(); ++k) {
        std::sort(ms[k].begin(), ms[k].end());
        const float med = ms[k][ms[k].size() / 2];
        const double bytes = (vs[k].kind == 0 ? 40.0 : 24.0) * n;
        printf("  %-26s median %7.3f ms  min %7.3f ms  %7.1f GB/s\n", vs[k].name.c_str(), med, ms[k][0], bytes / med / 1e6);
    }
    return 0;
}
