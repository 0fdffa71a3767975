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

// ---------------------------------------------------------------- row-pattern analysis
// A constant-coefficient stencil matrix has very few distinct ROWS when a row is read as the sequence of its
// (col - row, value) pairs in stored order: 27 for the 7-point Laplacian on a box (interior + the boundary combinations).
// With <= 256 of them one byte per row replaces the row's offsets, column ids and values altogether; the SpMV kernel
// (spmv_pattern_kernel, kernels_rows.hip) keeps the table of sequences in LDS.  Rows are matched by a 64-bit hash and
// then VERIFIED entry by entry against the table, so a hash collision can only make the analysis decline.
constexpr int kPatHashCap = 4096;
constexpr unsigned long long kPatEmpty = 0ull;

__device__ inline unsigned long long pat_mix(unsigned long long h, unsigned long long v)
{
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xFF51AFD7ED558CCDull;
    h ^= h >> 33;
    return h;
}

// rowHash[i] = hash of row i's sequence; distinct hashes are collected in `set`; stats[0] = distinct count (stops
// growing meaningfully past kPatMax), stats[1] = longest row
__global__ __launch_bounds__(kBlock) void pattern_hash_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets,
                                                              const int* __restrict__ columnIndeces, long long rows, long long rowBase,
                                                              unsigned long long* __restrict__ rowHash, unsigned long long* set, int* stats)
{
    const long long stride = (long long)gridDim.x * kBlock;
    int longest = 0;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        if (stats[0] > kPatMax) break;                             // hopeless: stop early (the analysis declines)
        const int s = rowOffsets[i], e = rowOffsets[i + 1];
        unsigned long long h = pat_mix(0x243F6A8885A308D3ull, (unsigned long long)(e - s));
        for (int k = s; k < e; ++k) {
            h = pat_mix(h, (unsigned long long)((long long)columnIndeces[k] - (rowBase + i)));
            h = pat_mix(h, (unsigned long long)__double_as_longlong(elements[k]));
        }
        if (h == kPatEmpty) h = 1ull;
        rowHash[i] = h;
        longest = (e - s) > longest ? (e - s) : longest;
        unsigned slot = (unsigned)(h >> 20) & (kPatHashCap - 1);
        for (int probe = 0; probe < kPatHashCap; ++probe) {
            const unsigned long long cur = set[slot];
            if (cur == h) break;
            if (cur == kPatEmpty) {
                const unsigned long long prev = atomicCAS(&set[slot], kPatEmpty, h);
                if (prev == kPatEmpty) { atomicAdd(&stats[0], 1); break; }
                if (prev == h) break;
            }
            slot = (slot + 1) & (kPatHashCap - 1);
        }
    }
    atomicMax(&stats[1], longest);
}

// patternId[i] = rank of the row's hash in the sorted list; rep[id] = smallest row with that id
__global__ __launch_bounds__(kBlock) void pattern_assign_kernel(const unsigned long long* __restrict__ rowHash, long long rows,
                                                                const unsigned long long* __restrict__ sorted, int n,
                                                                unsigned char* __restrict__ patternId, long long* rep)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        const int id = find_sorted(sorted, n, rowHash[i]);
        patternId[i] = (unsigned char)id;
        if (rep[id] > i) atomicMin((unsigned long long*)&rep[id], (unsigned long long)i);
    }
}

__global__ void pattern_table_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets, const int* __restrict__ columnIndeces,
                                     long long rowBase, const long long* __restrict__ rep, int n, int width,
                                     int* __restrict__ patCount, int* __restrict__ patDelta, double* __restrict__ patValue)
{
    const int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    const long long i = rep[id];
    const int s = rowOffsets[i], e = rowOffsets[i + 1];
    patCount[id] = e - s;
    for (int j = 0; j < width; ++j) {
        const bool in = s + j < e;
        patDelta[id * width + j] = in ? (int)((long long)columnIndeces[s + j] - (rowBase + i)) : 0;
        patValue[id * width + j] = in ? elements[s + j] : 0.0;
    }
}

// every row must equal its pattern entry by entry (guards against hash collisions); *err != 0 otherwise
__global__ __launch_bounds__(kBlock) void pattern_verify_kernel(const double* __restrict__ elements, const int* __restrict__ rowOffsets,
                                                                const int* __restrict__ columnIndeces, long long rows, long long rowBase,
                                                                const unsigned char* __restrict__ patternId, int width,
                                                                const int* __restrict__ patCount, const int* __restrict__ patDelta,
                                                                const double* __restrict__ patValue, int* err)
{
    const long long stride = (long long)gridDim.x * kBlock;
    bool bad = false;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < rows; i += stride) {
        const int id = patternId[i];
        const int s = rowOffsets[i], e = rowOffsets[i + 1];
        if (e - s != patCount[id]) { bad = true; continue; }
        for (int k = s; k < e; ++k) {
            const int d = (int)((long long)columnIndeces[k] - (rowBase + i));
            bad = bad || d != patDelta[id * width + (k - s)] ||
                  __double_as_longlong(elements[k]) != __double_as_longlong(patValue[id * width + (k - s)]);
        }
    }
    if (bad) atomicExch(err, 1);
}

// ---------------------------------------------------------------- build
void DcsrMatrix::release()
{
    if (colCode) (void)hipFree(colCode);
    if (valCode) (void)hipFree(valCode);
    if (deltaDict) (void)hipFree(deltaDict);
    if (valueDict) (void)hipFree(valueDict);
    if (patternId) (void)hipFree(patternId);
    if (patCount) (void)hipFree(patCount);
    if (patDelta) (void)hipFree(patDelta);
    if (patValue) (void)hipFree(patValue);
    if (tileVals) (void)hipFree(tileVals);
    if (tileCols) (void)hipFree(tileCols);
    if (tileRowIds) (void)hipFree(tileRowIds);
    if (tilePacked) (void)hipFree(tilePacked);
    if (tileHdr) (void)hipFree(tileHdr);
    tileVals = nullptr; tileCols = nullptr; tileRowIds = nullptr; nTiles = 0; tileRows = 0; tileStart.clear();
    tilePacked = nullptr; tileHdr = nullptr; tileShift = 0; tileWidth = 0; tileHdrBase.clear(); checksum = 0;
    colCode = valCode = nullptr; deltaDict = nullptr; valueDict = nullptr; nDelta = nValue = 0;
    patternId = nullptr; patCount = nullptr; patDelta = nullptr; patValue = nullptr; nPattern = patWidth = 0;
    usable = false;
}

// On success out->patternId != nullptr says whether the row-pattern form exists.  Leaves the other fields of *out alone.
bool pattern_build(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                   long long rows, long long nnz, long long rowBase, DcsrMatrix* out)
{
    if (rows <= 0 || nnz <= 0) return true;
    unsigned long long *rowHash = nullptr, *set = nullptr, *dSorted = nullptr;
    long long* rep = nullptr;
    int* stats = nullptr;
    unsigned char* ids = nullptr; int* pc = nullptr; int* pd = nullptr; double* pv = nullptr;
    auto cleanup = [&](bool keep) {
        for (void* p : { (void*)rowHash, (void*)set, (void*)dSorted, (void*)rep, (void*)stats }) if (p) (void)hipFree(p);
        if (!keep) for (void* p : { (void*)ids, (void*)pc, (void*)pd, (void*)pv }) if (p) (void)hipFree(p);
    };
    bool ok = MGCG_HIP(hipMalloc((void**)&rowHash, sizeof(unsigned long long) * (size_t)rows)) &&
              MGCG_HIP(hipMalloc((void**)&set, sizeof(unsigned long long) * kPatHashCap)) && MGCG_HIP(hipMalloc((void**)&stats, 3 * sizeof(int)));
    ok = ok && MGCG_HIP(hipMemsetAsync(set, 0, sizeof(unsigned long long) * kPatHashCap, s)) && MGCG_HIP(hipMemsetAsync(stats, 0, 3 * sizeof(int), s));
    if (!ok) { cleanup(false); return false; }
    long long blocks = (rows + kBlock - 1) / kBlock;
    if (blocks > kMaxGrid) blocks = kMaxGrid;
    hipLaunchKernelGGL(pattern_hash_kernel, dim3((int)blocks), dim3(kBlock), 0, s, elements, rowOffsets, columnIndeces, rows, rowBase, rowHash, set, stats);
    int hS[3] = { 0, 0, 0 };
    std::vector<unsigned long long> hSet(kPatHashCap);
    ok = MGCG_HIP(hipMemcpyAsync(hS, stats, sizeof(hS), hipMemcpyDeviceToHost, s)) &&
         MGCG_HIP(hipMemcpyAsync(hSet.data(), set, sizeof(unsigned long long) * kPatHashCap, hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
    if (!ok) { cleanup(false); return false; }
    const int n = hS[0], width = hS[1];
    if (n < 1 || n > kPatMax || width < 1 || (long long)n * width > kPatEntries) { cleanup(false); return true; }   // not a pattern matrix
    std::vector<unsigned long long> sorted;
    for (unsigned long long v : hSet) if (v != kPatEmpty) sorted.push_back(v);
    std::sort(sorted.begin(), sorted.end());
    if ((int)sorted.size() != n) { cleanup(false); return true; }
    std::vector<long long> hRep((size_t)n, 0x7fffffffffffffffLL);
    ok = MGCG_HIP(hipMalloc((void**)&dSorted, sizeof(unsigned long long) * (size_t)n)) && MGCG_HIP(hipMalloc((void**)&rep, sizeof(long long) * (size_t)n)) &&
         MGCG_HIP(hipMalloc((void**)&ids, (size_t)rows + 16)) && MGCG_HIP(hipMalloc((void**)&pc, sizeof(int) * (size_t)n)) &&
         MGCG_HIP(hipMalloc((void**)&pd, sizeof(int) * (size_t)n * width)) && MGCG_HIP(hipMalloc((void**)&pv, sizeof(double) * (size_t)n * width));
    ok = ok && MGCG_HIP(hipMemcpyAsync(dSorted, sorted.data(), sizeof(unsigned long long) * (size_t)n, hipMemcpyHostToDevice, s)) &&
         MGCG_HIP(hipMemcpyAsync(rep, hRep.data(), sizeof(long long) * (size_t)n, hipMemcpyHostToDevice, s));
    if (!ok) { cleanup(false); return false; }
    hipLaunchKernelGGL(pattern_assign_kernel, dim3((int)blocks), dim3(kBlock), 0, s, rowHash, rows, dSorted, n, ids, rep);
    hipLaunchKernelGGL(pattern_table_kernel, dim3((n + 63) / 64), dim3(64), 0, s, elements, rowOffsets, columnIndeces, rowBase, rep, n, width, pc, pd, pv);
    hipLaunchKernelGGL(pattern_verify_kernel, dim3((int)blocks), dim3(kBlock), 0, s, elements, rowOffsets, columnIndeces, rows, rowBase, ids, width, pc, pd, pv, &stats[2]);
    ok = MGCG_HIP(hipGetLastError()) && MGCG_HIP(hipMemcpyAsync(hS, stats, sizeof(hS), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
    if (!ok) { cleanup(false); return false; }
    if (hS[2] != 0) { cleanup(false); return true; }                 // two different rows shared a hash: decline
    out->patternId = ids; out->patCount = pc; out->patDelta = pd; out->patValue = pv; out->nPattern = n; out->patWidth = width;
    cleanup(true);
    return true;
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

void preload_kernels_dcsr() { preload_code_object(reinterpret_cast<const void*>(&dcsr_collect_kernel)); }

} // namespace mgcg
