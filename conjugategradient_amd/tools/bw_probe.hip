// bw_probe: what streaming bandwidth does this MI355X actually deliver?  (measurement tool, not product)
// Read-only sums and copies over a buffer far larger than the 256 MiB Infinity Cache, for several
// grid sizes / loads-in-flight per lane / cache policies.  Build: make -C conjugategradient_amd/csrc bwprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

template <int U, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const d2* __restrict__ p, long long n2, double* out)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    double acc = 0;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * stride) : p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    for (; i < n2; i += stride) { d2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[0] = acc;
}

// contiguous-chunk variant: each workgroup walks its own contiguous region (like a row-block kernel)
template <int U>
__global__ __launch_bounds__(256) void read_chunk_kernel(const d2* __restrict__ p, long long n2, double* out)
{
    const long long per = (n2 + gridDim.x - 1) / gridDim.x;
    const long long b = per * blockIdx.x, e = (b + per < n2) ? b + per : n2;
    double acc = 0;
    long long i = b + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y;
    }
    for (; i < e; i += 256) { d2 v = p[i]; acc += v.x + v.y; }
    if (acc == 1.2345e300) out[0] = acc;
}

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_kernel(const d2* __restrict__ p, d2* __restrict__ q, long long n2)
{
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n2; i += U * stride) {
        d2 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], q + i + u * stride); else q[i + u * stride] = v[u]; }
    }
    for (; i < n2; i += stride) q[i] = p[i];
}

// SpMV-like mix: stream a (4 B/elem) and b (8 B/elem) together, optionally write 8 B per 8 elements
template <bool WRITE, bool NT>
__global__ __launch_bounds__(256) void mix_kernel(const i2v* __restrict__ a, const d2* __restrict__ b, double* __restrict__ w, long long npairs, double* out)
{
    const long long stride = (long long)gridDim.x * 256;
    double acc = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npairs; i += stride) {
        i2v c = NT ? __builtin_nontemporal_load(a + i) : a[i];
        d2 v = NT ? __builtin_nontemporal_load(b + i) : b[i];
        acc += v.x * c.x + v.y * c.y;
        if (WRITE && (i & 3) == 0) w[i >> 2] = acc;
    }
    if (acc == 1.2345e300) out[0] = acc;
}

// as mix_kernel<true>, but every lane keeps B results in registers and stores them in one burst
template <int B>
__global__ __launch_bounds__(256) void mix_burst_kernel(const i2v* __restrict__ a, const d2* __restrict__ b, double* __restrict__ w, long long npairs, double* out)
{
    const long long stride = (long long)gridDim.x * 256;
    double acc = 0;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    while (i < npairs) {
        double buf[B];
        long long at[B];
#pragma unroll
        for (int k = 0; k < B; ++k) {
            buf[k] = 0; at[k] = -1;
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // 4 pairs read per 8-byte result
                if (i < npairs) { i2v c = a[i]; d2 v = b[i]; acc += v.x * c.x + v.y * c.y; at[k] = i; i += stride; }
            }
            buf[k] = acc;
        }
#pragma unroll
        for (int k = 0; k < B; ++k) if (at[k] >= 0) w[at[k] >> 2] = buf[k];
    }
    if (acc == 1.2345e300) out[0] = acc;
}

template <typename F>
static double time_ms(F f, int reps = 5)
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
    const long long bytes = (argc > 1 ? atoll(argv[1]) : 8LL) << 30;
    const long long n2 = bytes / 16;
    d2 *p, *q; double* out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&q, bytes)); CK(hipMalloc(&out, 8));
    CK(hipMemset(p, 0x11, bytes)); CK(hipMemset(q, 0, bytes));
    printf("buffer %lld GiB\n", bytes >> 30);
    const int grids[] = { 512, 1024, 2048, 4096, 8192, 16384, 65536 };
    for (int g : grids) {
        double m1 = time_ms([&] { hipLaunchKernelGGL((read_kernel<1, false>), dim3(g), dim3(256), 0, 0, p, n2, out); });
        double m2 = time_ms([&] { hipLaunchKernelGGL((read_kernel<2, false>), dim3(g), dim3(256), 0, 0, p, n2, out); });
        double m4 = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, false>), dim3(g), dim3(256), 0, 0, p, n2, out); });
        double m8 = time_ms([&] { hipLaunchKernelGGL((read_kernel<8, false>), dim3(g), dim3(256), 0, 0, p, n2, out); });
        double n4 = time_ms([&] { hipLaunchKernelGGL((read_kernel<4, true>), dim3(g), dim3(256), 0, 0, p, n2, out); });
        double c4 = time_ms([&] { hipLaunchKernelGGL((read_chunk_kernel<4>), dim3(g), dim3(256), 0, 0, p, n2, out); });
        printf("read  grid %6d: U1 %7.1f  U2 %7.1f  U4 %7.1f  U8 %7.1f  U4nt %7.1f  chunkU4 %7.1f GB/s\n", g,
               bytes / m1 / 1e6, bytes / m2 / 1e6, bytes / m4 / 1e6, bytes / m8 / 1e6, bytes / n4 / 1e6, bytes / c4 / 1e6);
    }
    for (int g : grids) {
        double m1 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<1, false>), dim3(g), dim3(256), 0, 0, p, q, n2); });
        double m4 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, false>), dim3(g), dim3(256), 0, 0, p, q, n2); });
        double n4 = time_ms([&] { hipLaunchKernelGGL((copy_kernel<4, true>), dim3(g), dim3(256), 0, 0, p, q, n2); });
        printf("copy  grid %6d: U1 %7.1f  U4 %7.1f  U4nt %7.1f GB/s (read+write bytes)\n", g, 2.0 * bytes / m1 / 1e6, 2.0 * bytes / m4 / 1e6, 2.0 * bytes / n4 / 1e6);
    }
    {
        // a: first third of p as int2 (8 B per pair), b: q as d2 (16 B per pair); npairs such that b spans `bytes`
        const long long npairs = bytes / 16;
        double* w = (double*)p + (bytes / 8) * 3 / 4;      // write target inside p, away from a
        for (int g : { 2048, 8192 }) {
            double r0 = time_ms([&] { hipLaunchKernelGGL((mix_kernel<false, false>), dim3(g), dim3(256), 0, 0, (const i2v*)p, q, w, npairs, out); });
            double r1 = time_ms([&] { hipLaunchKernelGGL((mix_kernel<false, true>), dim3(g), dim3(256), 0, 0, (const i2v*)p, q, w, npairs, out); });
            double w0 = time_ms([&] { hipLaunchKernelGGL((mix_kernel<true, false>), dim3(g), dim3(256), 0, 0, (const i2v*)p, q, w, npairs, out); });
            double w1 = time_ms([&] { hipLaunchKernelGGL((mix_kernel<true, true>), dim3(g), dim3(256), 0, 0, (const i2v*)p, q, w, npairs, out); });
            double b4 = time_ms([&] { hipLaunchKernelGGL((mix_burst_kernel<4>), dim3(g), dim3(256), 0, 0, (const i2v*)p, q, w, npairs, out); });
            double b16 = time_ms([&] { hipLaunchKernelGGL((mix_burst_kernel<16>), dim3(g), dim3(256), 0, 0, (const i2v*)p, q, w, npairs, out); });
            printf("burst grid %6d: 4 results/burst %7.1f  16 results/burst %7.1f GB/s\n", g, 26.0 * npairs / b4 / 1e6, 26.0 * npairs / b16 / 1e6);
            printf("mix   grid %6d: 2 read streams %7.1f  nt %7.1f | + 8%% writes %7.1f  nt %7.1f GB/s\n", g,
                   24.0 * npairs / r0 / 1e6, 24.0 * npairs / r1 / 1e6, 26.0 * npairs / w0 / 1e6, 26.0 * npairs / w1 / 1e6);
        }
    }
    double mm = time_ms([&] { CK(hipMemcpyAsync(q, p, bytes, hipMemcpyDeviceToDevice, 0)); });
    printf("hipMemcpy D2D: %7.1f GB/s (read+write bytes)\n", 2.0 * bytes / mm / 1e6);
    return 0;
}
