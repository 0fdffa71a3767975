// Dictionary-compressed CSR ("DCSR"): an opt-in, lossless re-encoding that the Solve-family builds once per
// matrix (the role of cuSPARSE's csrmv analysis in the reference's stack).  Many matrices on this path have
// few distinct column offsets col-row (every stencil / banded matrix) and few distinct values (constant-
// coefficient stencils: the 7-point Poisson matrix has 7 offsets and 2 values).  If there are <= 256 of each,
// a nonzero is stored as two bytes (code of the offset, code of the value) instead of 4 + 8 bytes; if only the
// offsets qualify, as one byte + the fp64 value.  The products are formed from exactly the same doubles and
// summed in the same order as in the CSR kernel, so results are bit-identical (tests/test_gpu_dcsr.py).
//
// This file holds the analysis (distinct-offset / distinct-value sets, code assignment, encoding); the SpMV
// kernel that consumes the codes is spmv_rows_kernel<FMT_DCSR8 / FMT_DCSR64> in kernels_rows.hip.
#include "common.hpp"
#include <algorithm>
#include <climits>

namespace mgcg {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

constexpr int kHashCap = 1024;                 // open-addressing sets used by the analysis (power of two)
constexpr int kDictMax = 256;
constexpr int kEmptyDelta = INT_MIN;
constexpr unsigned long long kEmptyValue = 0xFFF8DEADBEEF0001ull;   // a NaN payload no finite matrix entry has

// ---------------------------------------------------------------- analysis
__device__ inline void set_insert_delta(int* set, int* count, int key)
{
    unsigned h = ((unsigned)key * 2654435761u) & (kHashCap - 1);
    for (int probe = 0; probe < kHashCap; ++probe) {
        const int cur = set[h];
        if (cur == key) return;
        if (cur == kEmptyDelta) {
            const int prev = atomicCAS(&set[h], kEmptyDelta, key);
            if (prev == kEmptyDelta) { atomicAdd(count, 1); return; }
            if (prev == key) return;
        }
        h = (h + 1) & (kHashCap - 1);
    }
    atomicAdd(count, kHashCap);               // table full: far more than 256 distinct keys
}
__device__ inline void set_insert_value(unsigned long long* set, int* count, unsigned long long key)
{
    unsigned h = (unsigned)((key * 0x9E3779B97F4A7C15ull) >> 40) & (kHashCap - 1);
    for (int probe = 0; probe < kHashCap; ++probe) {
        const unsigned long long cur = set[h];
        if (cur == key) return;
        if (cur == kEmptyValue) {
            const unsigned long long prev = atomicCAS(&set[h], kEmptyValue, key);
            if (prev == kEmptyValue) { atomicAdd(count, 1); return; }
            if (prev == key) return;
        }
        h = (h + 1) & (kHashCap - 1);
    }
    atomicAdd(count, kHashCap);
}

// counts[0] = distinct offsets, counts[1] = distinct values (stops inserting once a set is hopeless)
__global__ __launch_bounds__(kBlock) void dcsr_collect_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets,
                                                              const int* __restrict__ columnIndeces, long long rows, long long rowBase,
                                                              int* deltaSet, unsigned long long* valueSet, int* counts)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        const bool doDelta = counts[0] <= kDictMax, doValue = counts[1] <= kDictMax;
        if (!doDelta && !doValue) return;
        for (int k = rowOffsets[i]; k < rowOffsets[i + 1]; ++k) {
            if (doDelta) set_insert_delta(deltaSet, &counts[0], (int)((long long)columnIndeces[k] - (rowBase + i)));
            if (doValue) set_insert_value(valueSet, &counts[1], (unsigned long long)__double_as_longlong(elements[k]));
        }
    }
}

__device__ inline int find_sorted(const int* dict, int n, int key)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (dict[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}
__device__ inline int find_sorted(const unsigned long long* dict, int n, unsigned long long key)
{
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (dict[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

__global__ __launch_bounds__(kBlock) void dcsr_encode_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets,
                                                             const int* __restrict__ columnIndeces, long long rows, long long rowBase,
                                                             const int* __restrict__ deltaDict, int nDelta,
                                                             const unsigned long long* __restrict__ valueDict, int nValue,
                                                             unsigned char* __restrict__ colCode, unsigned char* __restrict__ valCode)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        for (int k = rowOffsets[i]; k < rowOffsets[i + 1]; ++k) {
            colCode[k] = (unsigned char)find_sorted(deltaDict, nDelta, (int)((long long)columnIndeces[k] - (rowBase + i)));
            if (valCode != nullptr) valCode[k] = (unsigned char)find_sorted(valueDict, nValue, (unsigned long long)__double_as_longlong(elements[k]));
        }
    }
}

// ---------------------------------------------------------------- build
void DcsrMatrix::release()
{
    if (colCode) (void)hipFree(colCode);
    if (valCode) (void)hipFree(valCode);
    if (deltaDict) (void)hipFree(deltaDict);
    if (valueDict) (void)hipFree(valueDict);
    colCode = valCode = nullptr; deltaDict = nullptr; valueDict = nullptr; nDelta = nValue = 0; usable = false;
}

// Analyse the CSR slice (rows `rows`, first global row `rowBase`); on success out->usable says whether a
// compressed form exists (it needs <= 256 distinct offsets; values are compressed too when <= 256 are distinct).
bool dcsr_build(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                long long rows, long long nnz, long long rowBase, DcsrMatrix* out)
{
    out->release();
    out->elements = elements; out->rowOffsets = rowOffsets; out->columnIndeces = columnIndeces; out->rows = rows; out->nnz = nnz; out->rowBase = rowBase;
    if (rows <= 0 || nnz < 8) return true;            // nothing worth encoding
    int* dSet = nullptr; unsigned long long* vSet = nullptr; int* counts = nullptr;
    bool ok = MGCG_HIP(hipMalloc((void**)&dSet, sizeof(int) * kHashCap)) && MGCG_HIP(hipMalloc((void**)&vSet, sizeof(unsigned long long) * kHashCap)) &&
              MGCG_HIP(hipMalloc((void**)&counts, 2 * sizeof(int)));
    std::vector<int> hD(kHashCap, kEmptyDelta);
    std::vector<unsigned long long> hV(kHashCap, kEmptyValue);
    int hC[2] = { 0, 0 };
    ok = ok && MGCG_HIP(hipMemcpyAsync(dSet, hD.data(), sizeof(int) * kHashCap, hipMemcpyHostToDevice, s));
    ok = ok && MGCG_HIP(hipMemcpyAsync(vSet, hV.data(), sizeof(unsigned long long) * kHashCap, hipMemcpyHostToDevice, s));
    ok = ok && MGCG_HIP(hipMemcpyAsync(counts, hC, sizeof(hC), hipMemcpyHostToDevice, s));
    if (ok) {
        long long blocks = (rows + kBlock - 1) / kBlock;
        if (blocks > kMaxGrid) blocks = kMaxGrid;
        hipLaunchKernelGGL(dcsr_collect_kernel, dim3((int)blocks), dim3(kBlock), 0, s, elements, rowOffsets, columnIndeces, rows, rowBase, dSet, vSet, counts);
        ok = MGCG_HIP(hipMemcpyAsync(hC, counts, sizeof(hC), hipMemcpyDeviceToHost, s));
        ok = ok && MGCG_HIP(hipMemcpyAsync(hD.data(), dSet, sizeof(int) * kHashCap, hipMemcpyDeviceToHost, s));
        ok = ok && MGCG_HIP(hipMemcpyAsync(hV.data(), vSet, sizeof(unsigned long long) * kHashCap, hipMemcpyDeviceToHost, s));
        ok = ok && MGCG_HIP(hipStreamSynchronize(s));
    }
    if (dSet) (void)hipFree(dSet);
    if (vSet) (void)hipFree(vSet);
    if (counts) (void)hipFree(counts);
    if (!ok) return false;
    if (hC[0] > kDictMax) return true;                 // too many distinct offsets: stay with plain CSR
    std::vector<int> deltas;
    for (int v : hD) if (v != kEmptyDelta) deltas.push_back(v);
    std::sort(deltas.begin(), deltas.end());
    std::vector<unsigned long long> values;
    const bool val8 = hC[1] <= kDictMax;
    if (val8) { for (unsigned long long v : hV) if (v != kEmptyValue) values.push_back(v); std::sort(values.begin(), values.end()); }
    if ((int)deltas.size() != hC[0] || (val8 && (int)values.size() != hC[1])) { set_error("dcsr_build: inconsistent dictionary"); return false; }

    unsigned long long* dValBits = nullptr;
    ok = MGCG_HIP(hipMalloc((void**)&out->deltaDict, sizeof(int) * kDictMax)) && MGCG_HIP(hipMalloc((void**)&out->colCode, (size_t)nnz + 16));
    ok = ok && MGCG_HIP(hipMemcpyAsync(out->deltaDict, deltas.data(), sizeof(int) * deltas.size(), hipMemcpyHostToDevice, s));
    if (val8) {
        ok = ok && MGCG_HIP(hipMalloc((void**)&out->valueDict, sizeof(double) * kDictMax)) && MGCG_HIP(hipMalloc((void**)&out->valCode, (size_t)nnz + 16));
        ok = ok && MGCG_HIP(hipMalloc((void**)&dValBits, sizeof(unsigned long long) * kDictMax));
        ok = ok && MGCG_HIP(hipMemcpyAsync(dValBits, values.data(), sizeof(unsigned long long) * values.size(), hipMemcpyHostToDevice, s));
        ok = ok && MGCG_HIP(hipMemcpyAsync(out->valueDict, values.data(), sizeof(double) * values.size(), hipMemcpyHostToDevice, s));   // same bits
    }
    if (ok) {
        long long blocks = (rows + kBlock - 1) / kBlock;
        if (blocks > kMaxGrid) blocks = kMaxGrid;
        hipLaunchKernelGGL(dcsr_encode_kernel, dim3((int)blocks), dim3(kBlock), 0, s, elements, rowOffsets, columnIndeces, rows, rowBase,
                           out->deltaDict, (int)deltas.size(), dValBits, (int)values.size(), out->colCode, val8 ? out->valCode : nullptr);
        ok = MGCG_HIP(hipGetLastError()) && MGCG_HIP(hipStreamSynchronize(s));
    }
    if (dValBits) (void)hipFree(dValBits);
    if (!ok) { out->release(); return false; }
    out->nDelta = (int)deltas.size(); out->nValue = (int)values.size(); out->usable = true;
    return true;
}

} // namespace mgcg
