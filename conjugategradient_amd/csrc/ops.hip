// Operator half of the C ABI: CsrMV / Axpy / Dot / Scal / Copy, the multi-device set-up and halo
// staging (Initialize, P2Host, P2Device) and the phase functions Solve0..Solve3.
// Replaces Mgcg/cuBlas/MgcgGpu/Mgcg.cu:10-198.
#include "common.hpp"
#include <climits>

namespace mgcg {
long long poisson_nnz_host(int nx, int ny, int nz, int zBegin, int zEnd);

__global__ void set_alpha_kernel(CgScalars* sc, double alpha, double beta)
{
    sc->rr = alpha; sc->pAp = 1.0; sc->beta = beta; sc->done = 0;
}

static SpmvConfig cfg_of(const MgcgSparse* h)
{
    SpmvConfig c; c.kernel = h->kernel; c.rowsPerBlock = h->rowsPerBlock; c.flags = h->flags; c.gridBlocks = h->gridBlocks; c.periodRows = h->periodRows; c.tileRows = h->tileRows; c.tilePlanes = h->tilePlanes;
    return c;
}
static SpmvConfig cfg_for(MgcgSparse* h, const SpmvArgs& a, long long rowBase)
{
    SpmvConfig c = cfg_of(h);
    long long meanDistance = 0;
    if (a.elementsCount >= 8) c.periodRows = spmv_period(h, a.rowOffsets, a.columnIndeces, a.rowCount, rowBase, &c.maxRow, &meanDistance);
    if (meanDistance >= (1LL << 19) && a.columnCount >= (8LL << 19)) c.flags |= 16;     // gathers without locality (kernels_spmv.hip: launch_spmv_epi)
    return c;
}

// sum of `n` partials -> host double (blocking)
static double finish_reduction(Workspace& ws, int n, int mode)
{
    launch_reduce(ws.stream, ws.partials, n, ws.hostScalar, mode);
    if (!MGCG_HIP(hipGetLastError())) return NAN;
    if (!MGCG_HIP(hipStreamSynchronize(ws.stream))) return NAN;
    return ws.hostScalar[0];
}
void preload_ops() { preload_code_object(reinterpret_cast<const void*>(&set_alpha_kernel)); }

} // namespace mgcg

using namespace mgcg;

#define NEED_DEVICE(ret) do { if (!device_state()) return ret; } while (0)

extern "C" {

void CsrMV(MgcgSparse* cusparse, MgcgMatDescr* matDescr, double* y,
           const double* elements, const int* rowOffsets, const int* columnIndeces, const double* x,
           int elementsCount, int rowCount, int columnCount, double alpha, double beta)
{
    (void)matDescr;
    NEED_DEVICE();
    if (!cusparse || !y || !rowOffsets || !x || (elementsCount > 0 && (!elements || !columnIndeces))) { set_error("CsrMV: null argument"); return; }
    if (rowCount < 0 || elementsCount < 0 || columnCount < 0) { set_error("CsrMV: negative size"); return; }
    SpmvArgs a{};
    a.elements = elements; a.rowOffsets = rowOffsets; a.columnIndeces = columnIndeces; a.x = x; a.y = y;
    a.elementsCount = elementsCount; a.rowCount = rowCount; a.columnCount = columnCount; a.alpha = alpha; a.beta = beta;
#ifdef MGCG_LAB
    if (const char* ab = getenv("MGCG_SPMV_ABLATE")) a.ablate = atoi(ab);     // lab builds only: timing ablations with WRONG results
#endif
    const SpmvConfig cfg = cfg_for(cusparse, a, 0);
    const DcsrMatrix* dc = dcsr_lookup_op(cusparse, a, 0);
    analysis_note_write(y, sizeof(double) * (size_t)rowCount);
    launch_spmv_auto(cusparse->ws.stream, EPI_AXPBY, a, cfg, dc);
    (void)MGCG_HIP(hipGetLastError());
}

double CsrMVDot(MgcgBlas* cublas, MgcgSparse* cusparse, double* y,
                const double* elements, const int* rowOffsets, const int* columnIndeces,
                const double* x, const double* w, int elementsCount, int rowCount, int columnCount)
{
    NEED_DEVICE(NAN);
    if (!cublas || !cusparse || !y || !rowOffsets || !x || !w) { set_error("CsrMVDot: null argument"); return NAN; }
    if (rowCount <= 0) return 0.0;
    SpmvArgs a{};
#ifdef MGCG_LAB
    if (const char* ab = getenv("MGCG_SPMV_ABLATE")) a.ablate = atoi(ab);
#endif
    a.elements = elements; a.rowOffsets = rowOffsets; a.columnIndeces = columnIndeces; a.x = x; a.y = y;
    a.elementsCount = elementsCount; a.rowCount = rowCount; a.columnCount = columnCount; a.w = w; a.partials = cublas->ws.partials;
    const SpmvConfig cfg = cfg_for(cusparse, a, 0);
    const DcsrMatrix* dc = dcsr_lookup_op(cusparse, a, 0);
    int n = launch_spmv_auto(cublas->ws.stream, EPI_DOT, a, cfg, dc);
    if (dot_reference_order()) { launch_dot_serial(cublas->ws.stream, w, y, rowCount, cublas->ws.partials, nullptr); n = 1; }
    return finish_reduction(cublas->ws, n, 0);
}

void Axpy(MgcgBlas* cublas, double* y, const double* x, int count, double alpha)
{
    NEED_DEVICE();
    if (!cublas || !y || !x) { set_error("Axpy: null argument"); return; }
    if (count > 0) analysis_note_write(y, sizeof(double) * (size_t)count);
    launch_axpy(cublas->ws.stream, y, x, count, alpha);
    (void)MGCG_HIP(hipGetLastError());
}

double Dot(MgcgBlas* cublas, double* y, const double* x, int count)
{
    NEED_DEVICE(NAN);
    if (!cublas || !y || !x) { set_error("Dot: null argument"); return NAN; }
    if (count <= 0) return 0.0;
    const int n = launch_dot_partials(cublas->ws.stream, x, y, count, cublas->ws.partials);
    return finish_reduction(cublas->ws, n, 0);
}

double NrmInf(MgcgBlas* cublas, const double* x, int count)
{
    NEED_DEVICE(NAN);
    if (!cublas || !x) { set_error("NrmInf: null argument"); return NAN; }
    if (count <= 0) return 0.0;
    const int n = launch_nrminf_partials(cublas->ws.stream, x, count, cublas->ws.partials);
    return finish_reduction(cublas->ws, n, 1);
}

void Scal(MgcgBlas* cublas, double* x, double alpha, int count)
{
    NEED_DEVICE();
    if (!cublas || !x) { set_error("Scal: null argument"); return; }
    if (count > 0) analysis_note_write(x, sizeof(double) * (size_t)count);
    launch_scal(cublas->ws.stream, x, alpha, count);
    (void)MGCG_HIP(hipGetLastError());
}

void Xpay(MgcgBlas* cublas, double* y, const double* x, int count, double beta)
{
    NEED_DEVICE();
    if (!cublas || !y || !x) { set_error("Xpay: null argument"); return; }
    if (count > 0) analysis_note_write(y, sizeof(double) * (size_t)count);
    launch_xpay(cublas->ws.stream, y, x, count, beta);
    (void)MGCG_HIP(hipGetLastError());
}

void Copy(MgcgBlas* cublas, double* y, const double* x, int count, int yOffset, int xOffset)
{
    NEED_DEVICE();
    if (!cublas || !y || !x) { set_error("Copy: null argument"); return; }
    if (count > 0) analysis_note_write(y + yOffset, sizeof(double) * (size_t)count);
    launch_copy(cublas->ws.stream, y + yOffset, x + xOffset, count);
}

// ------------------------------------------------------------------ partition upload (Mgcg.cu:57-85)
void Initialize(const double elements[], const int rowOffsets[], const int columnIndeces[],
                const double x[], const double b[],
                Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                Vector* xVector, Vector* bVector, Vector* pVector,
                int* minJ, int* maxJ, int count,
                int countForDevice, int offsetForDevice, int elementCountForDevice, int elementOffsetForDevice)
{
    DeviceState* d = device_state();
    if (!d) return;
    if (!elements || !rowOffsets || !columnIndeces || !x || !b || !elementsVector || !rowOffsetsVector || !columnIndecesVector ||
        !xVector || !bVector || !pVector || !minJ || !maxJ) { set_error("Initialize: null argument"); return; }
    if (countForDevice < 0 || offsetForDevice < 0 || (long long)offsetForDevice + countForDevice > count ||
        elementCountForDevice < 0 || elementOffsetForDevice < 0) { set_error("Initialize: bad partition"); return; }
    if (elementsVector->size < elementCountForDevice || columnIndecesVector->size < elementCountForDevice ||
        rowOffsetsVector->size < (long long)countForDevice + 1 || xVector->size < countForDevice || bVector->size < countForDevice ||
        pVector->size < count) { set_error("Initialize: a device vector is too small for the partition"); return; }
    hipStream_t s = d->stream;
    bool ok = true;
    // the reference re-uploads A into the same device vectors on every Initialize(): analyses made from them are void
    analysis_note_write(elementsVector->data, sizeof(double) * (size_t)elementCountForDevice);
    analysis_note_write(columnIndecesVector->data, sizeof(int) * (size_t)elementCountForDevice);
    analysis_note_write(rowOffsetsVector->data, sizeof(int) * ((size_t)countForDevice + 1));
    if (elementCountForDevice > 0) {
        ok = ok && MGCG_HIP(hipMemcpyAsync(elementsVector->data, elements + elementOffsetForDevice, sizeof(double) * (size_t)elementCountForDevice, hipMemcpyHostToDevice, s));
        ok = ok && MGCG_HIP(hipMemcpyAsync(columnIndecesVector->data, columnIndeces + elementOffsetForDevice, sizeof(int) * (size_t)elementCountForDevice, hipMemcpyHostToDevice, s));
    }
    ok = ok && MGCG_HIP(hipMemcpyAsync(rowOffsetsVector->data, rowOffsets + offsetForDevice, sizeof(int) * (size_t)(countForDevice + 1), hipMemcpyHostToDevice, s));
    if (!ok) return;
    launch_rebase(s, rowOffsetsVector->data, (long long)countForDevice + 1, elementOffsetForDevice);   // Mgcg.cu:73
    if (countForDevice > 0) {
        ok = ok && MGCG_HIP(hipMemcpyAsync(xVector->data, x + offsetForDevice, sizeof(double) * (size_t)countForDevice, hipMemcpyHostToDevice, s));
        ok = ok && MGCG_HIP(hipMemcpyAsync(bVector->data, b + offsetForDevice, sizeof(double) * (size_t)countForDevice, hipMemcpyHostToDevice, s));
        ok = ok && MGCG_HIP(hipMemcpyAsync(pVector->data + offsetForDevice, xVector->data, sizeof(double) * (size_t)countForDevice, hipMemcpyDeviceToDevice, s));  // Mgcg.cu:80
    }
    if (!ok) return;
    // column range of the slice (Mgcg.cu:83-84), reduced on the device like the reference
    int* dmm = nullptr;
    if (!MGCG_HIP(hipMalloc((void**)&dmm, 2 * sizeof(int)))) return;
    int init[2] = { INT_MAX, INT_MIN };
    ok = MGCG_HIP(hipMemcpyAsync(dmm, init, sizeof(init), hipMemcpyHostToDevice, s));
    if (ok && elementCountForDevice > 0) launch_minmax_int(s, columnIndecesVector->data, elementCountForDevice, dmm);
    int out[2] = { 0, 0 };
    ok = ok && MGCG_HIP(hipMemcpyAsync(out, dmm, sizeof(out), hipMemcpyDeviceToHost, s));
    ok = ok && MGCG_HIP(hipStreamSynchronize(s));
    (void)hipFree(dmm);
    if (!ok) return;
    *minJ = out[0]; *maxJ = out[1];
}

int MgcgMinMaxColumn(VectorInt* columnIndecesVector, int elementCount, int* minJ, int* maxJ)
{
    DeviceState* d = device_state();
    if (!d) return -1;
    if (!columnIndecesVector || !minJ || !maxJ || elementCount < 0 || elementCount > columnIndecesVector->size) { set_error("MgcgMinMaxColumn: bad argument"); return -1; }
    int* dmm = nullptr;
    if (!MGCG_HIP(hipMalloc((void**)&dmm, 2 * sizeof(int)))) return -1;
    int init[2] = { INT_MAX, INT_MIN };
    bool ok = MGCG_HIP(hipMemcpyAsync(dmm, init, sizeof(init), hipMemcpyHostToDevice, d->stream));
    if (ok && elementCount > 0) launch_minmax_int(d->stream, columnIndecesVector->data, elementCount, dmm);
    int out[2] = { 0, 0 };
    ok = ok && MGCG_HIP(hipMemcpyAsync(out, dmm, sizeof(out), hipMemcpyDeviceToHost, d->stream));
    ok = ok && MGCG_HIP(hipStreamSynchronize(d->stream));
    (void)hipFree(dmm);
    if (!ok) return -1;
    *minJ = out[0]; *maxJ = out[1];
    return 0;
}

// ------------------------------------------------------------------ host-staged halo (Mgcg.cu:88-113)
void P2Host(Vector* pVector, double p[], int thisCount, int thisOffset, int lastCount, int nextCount)
{
    DeviceState* d = device_state();
    if (!d) return;
    if (!pVector || !p) { set_error("P2Host: null argument"); return; }
    if (lastCount < 0 || nextCount < 0 || lastCount > thisCount || nextCount > thisCount || thisOffset < 0 ||
        (long long)thisOffset + thisCount > pVector->size) { set_error("P2Host: halo widths outside this device's slice"); return; }
    bool ok = true;
    if (lastCount > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(p + thisOffset, pVector->data + thisOffset, sizeof(double) * (size_t)lastCount, hipMemcpyDeviceToHost, d->stream));
    if (nextCount > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(p + thisOffset + thisCount - nextCount, pVector->data + thisOffset + thisCount - nextCount,
                                                          sizeof(double) * (size_t)nextCount, hipMemcpyDeviceToHost, d->stream));
    if (ok) (void)MGCG_HIP(hipStreamSynchronize(d->stream));
}

void P2Device(Vector* pVector, double p[], int thisCount, int thisOffset, int lastCount, int nextCount)
{
    DeviceState* d = device_state();
    if (!d) return;
    if (!pVector || !p) { set_error("P2Device: null argument"); return; }
    if (lastCount < 0 || nextCount < 0 || thisOffset - lastCount < 0 ||
        (long long)thisOffset + thisCount + nextCount > pVector->size) { set_error("P2Device: halo outside the vector"); return; }
    bool ok = true;
    if (lastCount > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(pVector->data + thisOffset - lastCount, p + thisOffset - lastCount, sizeof(double) * (size_t)lastCount, hipMemcpyHostToDevice, d->stream));
    if (nextCount > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(pVector->data + thisOffset + thisCount, p + thisOffset + thisCount, sizeof(double) * (size_t)nextCount, hipMemcpyHostToDevice, d->stream));
    if (ok) (void)MGCG_HIP(hipStreamSynchronize(d->stream));
}

// ------------------------------------------------------------------ phase functions (Mgcg.cu:116-198)
double Solve0(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
              Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
              Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector,
              int count, int countForDevice, int offsetForDevice, int elementsCountForDevice)
{
    (void)matDescr; (void)xVector;
    NEED_DEVICE(NAN);
    if (!cublas || !cusparse || !elementsVector || !rowOffsetsVector || !columnIndecesVector || !bVector || !ApVector || !pVector || !rVector) { set_error("Solve0: null argument"); return NAN; }
    hipStream_t s = cublas->ws.stream;
    SpmvArgs a{};
    a.elements = elementsVector->data; a.rowOffsets = rowOffsetsVector->data; a.columnIndeces = columnIndecesVector->data;
    a.x = pVector->data; a.y = ApVector->data; a.elementsCount = elementsCountForDevice; a.rowCount = countForDevice; a.columnCount = count;
    a.alpha = 1.0; a.beta = 0.0;
    {
        const SpmvConfig cfg = cfg_for(cusparse, a, offsetForDevice);
        launch_spmv_auto(s, EPI_AXPBY, a, cfg, dcsr_lookup_op(cusparse, a, offsetForDevice));   // Ap = A p            (:138)
    }
    launch_copy(s, rVector->data, bVector->data, countForDevice);                     // r = b               (:139)
    launch_axpy(s, rVector->data, ApVector->data, countForDevice, -1.0);              // r -= Ap
    launch_copy(s, pVector->data + offsetForDevice, rVector->data, countForDevice);   // p[offset..] = r     (:140)
    if (countForDevice <= 0) return 0.0;
    const int n = launch_dot_partials(s, rVector->data, rVector->data, countForDevice, cublas->ws.partials);
    return finish_reduction(cublas->ws, n, 0);                                        // r.r                 (:141)
}

double Solve1(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
              Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
              Vector* ApVector, Vector* pVector,
              int count, int countForDevice, int offsetForDevice, int elementsCountForDevice)
{
    (void)matDescr;
    NEED_DEVICE(NAN);
    if (!cublas || !cusparse || !elementsVector || !rowOffsetsVector || !columnIndecesVector || !ApVector || !pVector) { set_error("Solve1: null argument"); return NAN; }
    if (countForDevice <= 0) return 0.0;
    SpmvArgs a{};
    a.elements = elementsVector->data; a.rowOffsets = rowOffsetsVector->data; a.columnIndeces = columnIndecesVector->data;
    a.x = pVector->data; a.y = ApVector->data; a.elementsCount = elementsCountForDevice; a.rowCount = countForDevice; a.columnCount = count;
    a.w = pVector->data + offsetForDevice; a.partials = cublas->ws.partials;
    const SpmvConfig cfg = cfg_for(cusparse, a, offsetForDevice);
    int n = launch_spmv_auto(cublas->ws.stream, EPI_DOT, a, cfg, dcsr_lookup_op(cusparse, a, offsetForDevice));   // Ap = A p ; p_loc.Ap  (:161-162) in one pass
    if (dot_reference_order()) { launch_dot_serial(cublas->ws.stream, a.w, a.y, countForDevice, cublas->ws.partials, nullptr); n = 1; }
    return finish_reduction(cublas->ws, n, 0);
}

double Solve2(MgcgBlas* cublas, double alpha, Vector* xVector, Vector* ApVector, Vector* pVector, Vector* rVector,
              int countForDevice, int offsetForDevice)
{
    NEED_DEVICE(NAN);
    if (!cublas || !xVector || !ApVector || !pVector || !rVector) { set_error("Solve2: null argument"); return NAN; }
    if (countForDevice <= 0) return 0.0;
    hipStream_t s = cublas->ws.stream;
    hipLaunchKernelGGL(set_alpha_kernel, dim3(1), dim3(1), 0, s, cublas->ws.scalars, alpha, 0.0);
    const int n = launch_update_xr(s, cublas->ws.scalars, xVector->data, rVector->data, pVector->data + offsetForDevice, ApVector->data,
                                   countForDevice, cublas->ws.partials, nullptr);     // x += a p ; r -= a Ap ; r.r  (:181-183)
    return finish_reduction(cublas->ws, n, 0);
}

void Solve3(MgcgBlas* cublas, double beta, Vector* pVector, Vector* rVector, int countForDevice, int offsetForDevice)
{
    NEED_DEVICE();
    if (!cublas || !pVector || !rVector) { set_error("Solve3: null argument"); return; }
    launch_xpay(cublas->ws.stream, pVector->data + offsetForDevice, rVector->data, countForDevice, beta);   // p = r + beta p  (:197)
    (void)MGCG_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ device problem generator
long long MgcgPoissonNnz(int nx, int ny, int nz, int zBegin, int zEnd)
{
    if (nx < 1 || ny < 1 || nz < 1 || zBegin < 0 || zEnd > nz || zBegin > zEnd) return -1;
    return poisson_nnz_host(nx, ny, nz, zBegin, zEnd);
}

int MgcgGeneratePoisson(Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                        int nx, int ny, int nz, int zBegin, int zEnd)
{
    DeviceState* d = device_state();
    if (!d) return -1;
    const long long nnz = MgcgPoissonNnz(nx, ny, nz, zBegin, zEnd);
    if (nnz < 0) { set_error("MgcgGeneratePoisson: bad grid"); return -1; }
    const long long rows = (long long)(zEnd - zBegin) * nx * ny;
    if ((long long)nx * ny * nz > INT_MAX || nnz > INT_MAX) { set_error("MgcgGeneratePoisson: grid too large for int32 indices"); return -1; }
    if (!elementsVector || !rowOffsetsVector || !columnIndecesVector || elementsVector->size < nnz || columnIndecesVector->size < nnz || rowOffsetsVector->size < rows + 1) {
        set_error("MgcgGeneratePoisson: vectors too small (need %lld nnz, %lld rows)", nnz, rows); return -1;
    }
    analysis_note_write(elementsVector->data, sizeof(double) * (size_t)nnz);
    analysis_note_write(columnIndecesVector->data, sizeof(int) * (size_t)nnz);
    analysis_note_write(rowOffsetsVector->data, sizeof(int) * ((size_t)rows + 1));
    launch_poisson(d->stream, nx, ny, nz, zBegin, zEnd, elementsVector->data, rowOffsetsVector->data, columnIndecesVector->data);
    return MGCG_HIP(hipGetLastError()) ? 0 : -1;
}

} // extern "C"
