// Dictionary-compressed CSR ("DCSR"): an opt-in, lossless re-encoding that the Solve-family builds once per
// matrix (the role of cuSPARSE's csrmv analysis in the reference's stack).  Many matrices on this path have
// few distinct column offsets col-row (every stencil / banded matrix) and few distinct values (constant-
// coefficient stencils: the 7-point Poisson matrix has 7 offsets and 2 values).  If there are <= 256 of each,
// a nonzero is stored as two bytes (code of the offset, code of the value) instead of 4 + 8 bytes; if only the
// offsets qualify, as one byte + the fp64 value.  The products are formed from exactly the same doubles and
// summed in the same order as in the CSR kernel, so results are bit-identical (tests/test_gpu_dcsr.py).
//
// SpMV kernel = the row-block stream kernel of kernels_spmv.hip (one wavefront per 64 rows, products parked in
// LDS, one lane per row adds in stored order, same software pipeline and epilogues) with the wide loads
// replaced by 8 codes per lane and an LDS dictionary look-up; the row of a nonzero (needed for col = row +
// offset) comes from a binary search of the block's row offsets in LDS.
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

// ---------------------------------------------------------------- SpMV
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

struct DcsrEpiOperands { double w, b, dinv, yold; };

template <int EPI>
__device__ __forceinline__ DcsrEpiOperands dcsr_epi_prefetch(const SpmvArgs& a, long long row)
{
    DcsrEpiOperands o; o.w = 0.0; o.b = 0.0; o.dinv = 0.0; o.yold = 0.0;
    if constexpr (EPI == EPI_AXPBY_BETA) o.yold = a.y[row];
    if constexpr (EPI == EPI_DOT) o.w = a.w[row];
    if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_DOT) o.b = a.b[row];
    if constexpr (EPI == EPI_JACOBI) { o.b = a.b[row]; o.dinv = a.dinv[row]; o.w = a.w[row]; }
    return o;
}

template <int EPI>
__device__ __forceinline__ double dcsr_epilogue_value(const SpmvArgs& a, double acc, const DcsrEpiOperands& o, double& dotacc)
{
    if constexpr (EPI == EPI_AXPBY) return a.alpha * acc;
    else if constexpr (EPI == EPI_AXPBY_BETA) { double v = a.alpha * acc; double t = a.beta * o.yold; return v + t; }
    else if constexpr (EPI == EPI_DOT) { double t = o.w * acc; dotacc += t; return acc; }
    else if constexpr (EPI == EPI_RESIDUAL) return o.b - acc;
    else if constexpr (EPI == EPI_RESIDUAL_DOT) { double r = o.b - acc; double t = r * r; dotacc += t; return r; }
    else { double res = o.b - acc; double t = o.dinv * res; double s = a.omega * t; return o.w + s; }
}

constexpr int kDR = 64;          // rows per wavefront trip
constexpr int kDCap = 512;       // nonzeros per pass: 8 per lane
constexpr int kDAlign = 8;       // spans are read from an 8-nonzero boundary (8-byte code loads)

// One wavefront per workgroup.  VAL8: values are dictionary codes too (2 B/nnz), otherwise fp64 values + offset codes.
template <int EPI, bool VAL8>
__global__ __launch_bounds__(64) void spmv_dcsr_kernel(SpmvArgs a, DcsrView m, int nRowBlocks)
{
    __shared__ double s_prod[kDCap];
    __shared__ double s_vD[kDictMax];
    __shared__ int s_dD[kDictMax];
    __shared__ int s_ro[kDR + 1];

    if (a.doneFlag != nullptr && *a.doneFlag != 0) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < kDictMax; i += 64) {
        s_dD[i] = i < m.nDelta ? m.deltaDict[i] : 0;
        if (VAL8) s_vD[i] = i < m.nValue ? m.valueDict[i] : 0.0;
    }
    __syncthreads();

    const long long lastRow = (long long)a.rowCount - 1;
    const int nTrips = ((long long)nRowBlocks > (long long)blockIdx.x) ? (int)(((long long)nRowBlocks - blockIdx.x + gridDim.x - 1) / gridDim.x) : 0;
    auto rb_of = [&](int t) -> long long { return (long long)blockIdx.x + (long long)t * gridDim.x; };
    auto span_of = [&](long long rb, int& s, int& e) {
        const long long r0 = rb * kDR;
        const long long r1 = (r0 + kDR < (long long)a.rowCount) ? r0 + kDR : (long long)a.rowCount;
        s = a.rowOffsets[r0]; e = a.rowOffsets[r1];
    };
    // last aligned 8-code group wholly inside the arrays (host guarantees elementsCount >= 8)
    const int kMaxWide = (a.elementsCount - 8) & ~7;

    struct Stage { u2 cc, vc; d2 v[4]; int my_s, my_e, s, e; };
    auto issue = [&](Stage& st, long long rb) {
        const long long r0 = rb * kDR;
        long long row = r0 + tid;
        const bool live = row <= lastRow;
        row = live ? row : lastRow;
        const int ms = a.rowOffsets[row], me = a.rowOffsets[row + 1];
        st.my_s = live ? ms : st.e;
        st.my_e = live ? me : st.e;
        int k = (st.s & ~(kDAlign - 1)) + 8 * tid;
        k = k < kMaxWide ? k : kMaxWide;
        st.cc = *(const u2*)(m.colCode + k);
        if constexpr (VAL8) st.vc = *(const u2*)(m.valCode + k);
        else {
#pragma unroll
            for (int q = 0; q < 4; ++q) st.v[q] = *(const d2*)(a.elements + k + 2 * q);
        }
    };

    double dotacc = 0.0;
    if (nTrips > 0) {
        Stage cur, nxt;
        span_of(rb_of(0), cur.s, cur.e);
        issue(cur, rb_of(0));
        nxt.s = cur.s; nxt.e = cur.e;
        if (nTrips > 1) span_of(rb_of(1), nxt.s, nxt.e);
        double pendVal = 0.0;
        long long pendRow = -1;
        for (int t = 0; t < nTrips; ++t) {
            const long long rb = rb_of(t);
            const long long r0 = rb * kDR;
            const long long left = (long long)a.rowCount - r0;
            const int nr = (int)(left < kDR ? left : kDR);
            int s2 = nxt.s, e2 = nxt.e;
            if (t + 2 < nTrips) span_of(rb_of(t + 2), s2, e2);
            const int s = cur.s, e = cur.e;
            const int tb0 = s & ~(kDAlign - 1);

            // row offsets of the block to LDS (for the row look-up of every nonzero)
            s_ro[tid] = cur.my_s;
            if (tid == nr - 1) s_ro[nr] = cur.my_e;
            __syncthreads();
            // row of this lane's first nonzero: largest j < nr with s_ro[j] <= k0 (k0 < s happens only in the alignment head)
            const int k0 = tb0 + 8 * tid;
            int row = 0;
            {
                int lo = 0, hi = nr - 1;
                while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_ro[mid] <= k0) lo = mid; else hi = mid - 1; }
                row = lo;
            }
            int nextStart = s_ro[row + 1];
            // ---- decode + gathers of the current block ----
            double xg[8], vv[8];
            const long long rowBase = m.rowBase + r0;
            const bool wideOk = (k0 <= kMaxWide);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                while (row < nr - 1 && k >= nextStart) { ++row; nextStart = s_ro[row + 1]; }
                const unsigned cword = j < 4 ? cur.cc.x : cur.cc.y;
                const unsigned code = (cword >> (8 * (j & 3))) & 255u;
                const bool valid = wideOk && k >= s && k < e;
                long long col = valid ? (rowBase + row + s_dD[code]) : rowBase;
                if (a.ablate & 2) col &= 1023;
                xg[j] = a.x[col];
                if constexpr (VAL8) {
                    const unsigned vword = j < 4 ? cur.vc.x : cur.vc.y;
                    vv[j] = s_vD[(vword >> (8 * (j & 3))) & 255u];
                } else {
                    vv[j] = (j & 1) ? cur.v[j >> 1].y : cur.v[j >> 1].x;
                }
            }
            long long myRow = r0 + tid;
            myRow = myRow <= lastRow ? myRow : lastRow;
            const DcsrEpiOperands eo = dcsr_epi_prefetch<EPI>(a, myRow);
            issue(nxt, t + 1 < nTrips ? rb_of(t + 1) : rb);
            if (pendRow >= 0 && !(a.ablate & 1)) a.y[pendRow] = pendVal;
            // ---- products to LDS ----
            if (wideOk) {
#pragma unroll
                for (int j = 0; j < 8; j += 2) { d2 p; p.x = vv[j] * xg[j]; p.y = vv[j + 1] * xg[j + 1]; *(d2*)(s_prod + 8 * tid + j) = p; }
            } else {
                // array tail (fewer than 8 codes left): guarded scalars straight from the CSR arrays
                for (int j = 0; j < 8; ++j) { const int k = k0 + j; if (k >= s && k < e) s_prod[8 * tid + j] = a.elements[k] * a.x[a.columnIndeces[k]]; }
            }
            __syncthreads();
            // ---- reduce: one lane per row, stored order ----
            double acc = 0.0;
            {
                const int lo = cur.my_s > tb0 ? cur.my_s : tb0;
                const int hi = cur.my_e < tb0 + kDCap ? cur.my_e : tb0 + kDCap;
                for (int j = lo; j < hi; j += 8) {
                    double v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) { int idx = j + q - tb0; idx = idx < kDCap - 1 ? idx : kDCap - 1; v[q] = s_prod[idx]; }
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc += (j + q < hi) ? v[q] : 0.0;
                }
            }
            // ---- further passes (rows longer than the pass): from the CSR arrays, not pipelined ----
            for (int tb = tb0 + kDCap; tb < e; tb += kDCap) {
                __syncthreads();
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int oo = j * 64 + tid; const int kk = tb + oo; if (kk < e) s_prod[oo] = a.elements[kk] * a.x[a.columnIndeces[kk]]; }
                __syncthreads();
                const int lo = cur.my_s > tb ? cur.my_s : tb;
                const int hi = cur.my_e < tb + kDCap ? cur.my_e : tb + kDCap;
                for (int j = lo; j < hi; ++j) acc += s_prod[j - tb];
            }
            pendRow = -1;
            if (tid < nr) { pendVal = dcsr_epilogue_value<EPI>(a, acc, eo, dotacc); pendRow = r0 + tid; }
            if ((a.ablate & 1) && acc == 1.2345e300) pendRow = 0;
            __syncthreads();
            cur = nxt;
            nxt.s = s2; nxt.e = e2;
        }
        if (pendRow >= 0 && !(a.ablate & 1)) a.y[pendRow] = pendVal;
    }
    if constexpr (EPI == EPI_DOT || EPI == EPI_RESIDUAL_DOT) {
        const double tsum = wave_sum_d(dotacc);
        if (tid == 0) a.partials[blockIdx.x] = tsum;
    }
}

template <int EPI>
static int launch_dcsr_epi(hipStream_t s, const SpmvArgs& a, const DcsrView& m, int gridReq)
{
    const int nRowBlocks = (int)(((long long)a.rowCount + kDR - 1) / kDR);
    DeviceState* d = device_state();
    int grid = gridReq > 0 ? gridReq : 16 * (d ? d->numCu : kNumCu);
    if (grid > kMaxPartials) grid = kMaxPartials;
    if (grid > nRowBlocks) grid = nRowBlocks;
    if (grid < 1) grid = 1;
    if (m.valCode != nullptr) hipLaunchKernelGGL((spmv_dcsr_kernel<EPI, true>), dim3(grid), dim3(64), 0, s, a, m, nRowBlocks);
    else hipLaunchKernelGGL((spmv_dcsr_kernel<EPI, false>), dim3(grid), dim3(64), 0, s, a, m, nRowBlocks);
    return grid;
}

int launch_spmv_dcsr(hipStream_t s, int epilogue, const SpmvArgs& a, const DcsrView& m, int gridReq)
{
    if (a.rowCount <= 0) return 0;
    switch (epilogue) {
    case EPI_AXPBY:        return a.beta != 0.0 ? launch_dcsr_epi<EPI_AXPBY_BETA>(s, a, m, gridReq) : launch_dcsr_epi<EPI_AXPBY>(s, a, m, gridReq);
    case EPI_DOT:          return launch_dcsr_epi<EPI_DOT>(s, a, m, gridReq);
    case EPI_RESIDUAL:     return launch_dcsr_epi<EPI_RESIDUAL>(s, a, m, gridReq);
    case EPI_RESIDUAL_DOT: return launch_dcsr_epi<EPI_RESIDUAL_DOT>(s, a, m, gridReq);
    case EPI_JACOBI:       return launch_dcsr_epi<EPI_JACOBI>(s, a, m, gridReq);
    }
    return 0;
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
