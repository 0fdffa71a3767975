// gridsync_probe -- what does a grid-wide barrier cost on this chip?  (a persistent kernel for small systems would
// trade four dependent launches per CG iteration, ~18 us, for three grid barriers.)  cooperative_groups grid.sync() in a loop,
// 256-thread workgroups, grids of 1 / 2 / 4 / 8 workgroups per CU; also with each thread writing and then reading a neighbour's
// double across the barrier (the traffic a CG phase boundary really has).
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
#include <cstdlib>
namespace cg = cooperative_groups;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void sync_loop(int iters, double* buf, int withData, double* out)
{
    cg::grid_group grid = cg::this_grid();
    const long long n = (long long)gridDim.x * blockDim.x;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (int k = 0; k < iters; ++k) {
        if (withData) buf[i] = (double)(k + i);
        grid.sync();
        if (withData) acc += buf[(i + 4099) % n];          // another workgroup's (usually another XCD's) value
    }
    if (withData) out[i] = acc;
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double *buf, *out;
    CK(hipMalloc(&buf, sizeof(double) * 8 * cus * 256)); CK(hipMalloc(&out, sizeof(double) * 8 * cus * 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int withData = 0; withData < 2; ++withData)
        for (int perCu : { 1, 2, 4, 8 }) {
            int iters = 2000;
            int grid = cus * perCu;
            void* args[] = { &iters, &buf, &withData, &out };
            CK(hipLaunchCooperativeKernel((const void*)sync_loop, dim3(grid), dim3(256), args, 0, nullptr));   // warm-up
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((const void*)sync_loop, dim3(grid), dim3(256), args, 0, nullptr));
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s grid %5d (%d per CU): %.2f us per grid.sync\n", withData ? "write+sync+read" : "sync only      ", grid, perCu, 1e3 * ms / iters);
        }
    return 0;
}
