// gather_lab: how fast can 8-byte random gathers from an L2-sized window be served, by load flavour?  (measurement tool)
//   gather_lab [windowDoubles=524288] [n=268435456]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ void fill_idx(int* idx, long long n, int window)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        unsigned long long h = (unsigned long long)i * 0x9E3779B97F4A7C15ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        idx[i] = (int)(h % (unsigned)window);
    }
}
// MODE 0 plain, 1 nt, 2 buffer aux sc0, 3 buffer aux sc1, 4 buffer aux sc0|sc1, 5 buffer aux nt, 6 buffer aux 0, 7 float gathers (4 B)
template <int MODE, int U>
__global__ __launch_bounds__(256) void gather_kernel(const double* __restrict__ x, const int* __restrict__ idx, long long n, double* __restrict__ out, int window)
{
    __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, window * 8, 0x00020000);
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
        int c[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = __builtin_nontemporal_load(idx + i + u * stride);
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) v[u] = x[c[u]];
            else if (MODE == 1) v[u] = __builtin_nontemporal_load(x + c[u]);
            else if (MODE == 7) v[u] = (double)((const float*)x)[c[u]];
            else {
                constexpr int aux = MODE == 2 ? 1 : MODE == 3 ? 16 : MODE == 4 ? 17 : MODE == 5 ? 2 : 0;
                typedef unsigned u2v __attribute__((ext_vector_type(2)));
                const u2v b = __builtin_amdgcn_raw_buffer_load_b64(xr, c[u] * 8, 0, aux);
                v[u] = __longlong_as_double(((long long)b.y << 32) | b.x);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc == 1.2345e300) out[0] = acc;
}
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
    const long long n = argc > 2 ? atoll(argv[2]) : 268435456LL;
    int* idx; double *x, *out;
    CK(hipMalloc(&idx, n * 4)); CK(hipMalloc(&x, 16LL << 20 << 3)); CK(hipMalloc(&out, 8));
    CK(hipMemset(x, 0, 16LL << 20 << 3));
    for (int window : { 1 << 12, 1 << 16, 1 << 19, 1 << 21, 1 << 23 }) {
        if (argc > 1 && atoi(argv[1]) > 0) window = atoi(argv[1]);
        hipLaunchKernelGGL(fill_idx, dim3(4096), dim3(256), 0, 0, idx, n, window);
        CK(hipDeviceSynchronize());
        printf("window %8d doubles (%6.1f MB): ", window, window * 8 / 1048576.0);
#define RUN(M, U, name) { double ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<M, U>), dim3(8192), dim3(256), 0, 0, x, idx, n, out, window); }); printf(" %s %6.1f G/s", name, n / ms / 1e6); }
        RUN(0, 4, "plain") RUN(0, 8, "plainU8") RUN(1, 4, "nt") RUN(6, 4, "buf") RUN(2, 4, "buf.sc0") RUN(3, 4, "buf.sc1") RUN(4, 4, "buf.sc0sc1") RUN(5, 4, "buf.nt") RUN(7, 4, "f32")
        printf("\n"); fflush(stdout);
        if (argc > 1 && atoi(argv[1]) > 0) break;
    }
    return 0;
}
