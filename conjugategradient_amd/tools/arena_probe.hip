// arena_probe -- does ONE physical allocation (hipMemCreate + hipMemMap) holding all five arrays of the CSR SpMV make its time
// reproducible from process to process, and which relative offsets of x and y are fast?  (placement_probe.py: with one hipMalloc
// per array the same CsrMV reads 2.24-2.55 ms depending on where x landed.)
//   arena_probe [n=512] [mode]      mode 0: x offset sweep, 1: y offset sweep, 2: both fixed, repeated
// Links libMgcgGpu.so for the matrix generator and the CsrMV export (raw device pointers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../../include/MgcgGpu.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 512;
    const int mode = argc > 2 ? atoi(argv[2]) : 0;
    const long long N = (long long)n * n * n;
    SetDevice(0);
    MgcgBlas* blas = CreateBlas(); MgcgSparse* sparse = CreateSparse(); MgcgMatDescr* descr = CreateMatDescr();
    (void)blas;
    const long long nnz = MgcgPoissonNnz(n, n, n, 0, n);
    // the matrix is generated into library vectors, then copied into the arena
    Vector* ve = Create_Double((int)nnz); VectorInt* vc = Create_Int((int)nnz); VectorInt* vr = Create_Int((int)N + 1);
    if (MgcgGeneratePoisson(ve, vr, vc, n, n, n, 0, n) != 0) { printf("generate failed: %s\n", MgcgGetLastError()); return 1; }
    MgcgDeviceSynchronize();

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    const size_t MiB = 1u << 20;
    auto up = [&](size_t b) { return (b + gran - 1) / gran * gran; };
    const size_t szE = up(8 * (size_t)nnz), szC = up(4 * (size_t)nnz), szR = up(4 * (size_t)(N + 1)), szV = up(8 * (size_t)N);
    const size_t slack = 256 * MiB;
    const size_t total = szE + szC + szR + 2 * (szV + slack) + gran;
    void* base = nullptr;
    const int how = argc > 3 ? atoi(argv[3]) : 0;
    if (how == 4 || how == 5) { CK(hipMalloc(&base, 4096)); }
    else      // 0: hipMemCreate + hipMemMap, 1: one hipMalloc, 2: VMM with 2 MiB granularity and 1 GiB aligned reserve
    if (how == 1) {
        CK(hipMalloc(&base, total));
    } else if (how == 3) {
        CK(hipExtMallocWithFlags(&base, total, hipDeviceMallocContiguous));       // physically contiguous VRAM
    } else {
        if (how == 2) gran = 2 * MiB;
        const size_t tot2 = (total + gran - 1) / gran * gran;
        hipMemGenericAllocationHandle_t h;
        CK(hipMemCreate(&h, tot2, &prop, 0));
        CK(hipMemAddressReserve(&base, tot2, how == 2 ? (size_t)1 << 30 : gran, nullptr, 0));
        CK(hipMemMap(base, tot2, 0, h, 0));
        hipMemAccessDesc acc = {};
        acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(base, tot2, &acc, 1));
    }
    char* p = (char*)base;
    double* e = (double*)p; p += szE;
    int* c = (int*)p; p += szC;
    int* r = (int*)p; p += szR;
    char* yRegion = p; p += szV + slack;
    char* xRegion = p;
    if (how == 4 || how == 5) {                        // one allocation per array: 4 physically contiguous, 5 default (the library's way)
        const unsigned fl = how == 4 ? hipDeviceMallocContiguous : hipDeviceMallocDefault;
        void* q = nullptr;
        CK(hipExtMallocWithFlags(&q, szE, fl)); e = (double*)q;
        CK(hipExtMallocWithFlags(&q, szC, fl)); c = (int*)q;
        CK(hipExtMallocWithFlags(&q, szR, fl)); r = (int*)q;
        CK(hipExtMallocWithFlags(&q, szV + slack, fl)); yRegion = (char*)q;
        CK(hipExtMallocWithFlags(&q, szV + slack, fl)); xRegion = (char*)q;
    }
    printf("granularity %zu KiB, arena %.2f GiB at %p: e +0, c +%zu MiB, r +%zu MiB, y +%zu MiB, x +%zu MiB\n", gran >> 10, total / 1073741824.0, base,
           (size_t)((char*)c - (char*)base) / MiB, (size_t)((char*)r - (char*)base) / MiB, (size_t)(yRegion - (char*)base) / MiB, (size_t)(xRegion - (char*)base) / MiB);
    CK(hipMemcpy(e, ToRawPtr_Double(ve), 8 * (size_t)nnz, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(c, ToRawPtr_Int(vc), 4 * (size_t)nnz, hipMemcpyDeviceToDevice));
    CK(hipMemcpy(r, ToRawPtr_Int(vr), 4 * (size_t)(N + 1), hipMemcpyDeviceToDevice));
    Delete_Double(ve); Delete_Int(vc); Delete_Int(vr);
    {   // x = 1 everywhere in its region
        std::vector<double> ones(1 << 20, 1.0);
        for (size_t off = 0; off < szV + slack; off += ones.size() * 8) {
            const size_t len = std::min(ones.size() * 8, szV + slack - off);
            CK(hipMemcpy(xRegion + off, ones.data(), len, hipMemcpyHostToDevice));
        }
    }
    void* ev0 = MgcgEventCreate(); void* ev1 = MgcgEventCreate();
    auto timed = [&](size_t xs, size_t ys) {
        double* x = (double*)(xRegion + xs); double* y = (double*)(yRegion + ys);
        for (int k = 0; k < 2; ++k) CsrMV(sparse, descr, y, e, r, c, x, (int)nnz, (int)N, (int)N, 1.0, 0.0);
        double t[3];
        for (int rep = 0; rep < 3; ++rep) {
            MgcgEventRecord(ev0);
            for (int k = 0; k < 6; ++k) CsrMV(sparse, descr, y, e, r, c, x, (int)nnz, (int)N, (int)N, 1.0, 0.0);
            MgcgEventRecord(ev1);
            t[rep] = MgcgEventElapsedMs(ev0, ev1) / 6;
        }
        std::sort(t, t + 3);
        return t[1];
    };
    if (mode == 2) {
        for (int k = 0; k < 4; ++k) printf("x +0 y +0: %.3f ms\n", timed(0, 0));
    } else {
        const size_t shifts[] = { 0, 1, 2, 3, 4, 6, 8, 10, 12, 14, 16, 20, 24, 32, 48, 64, 96, 128, 192, 255 };
        for (size_t s : shifts) {
            const double ms = mode == 0 ? timed(s * MiB, 0) : timed(0, s * MiB);
            printf("%s +%3zu MiB: %.3f ms\n", mode == 0 ? "x" : "y", s, ms);
        }
    }
    const char* err = MgcgGetLastError();
    if (err && *err) printf("library error: %s\n", err);
    return 0;
}
