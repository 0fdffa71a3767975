// Runtime half of the C ABI: devices, handles, device vectors, errors.
// Replaces Mgcg/cuBlas/MgcgGpu/Runtime.cu, Vector_Double.cu and Vector_Int.cu.
#include "common.hpp"
#include <map>

namespace mgcg {

// ---------------------------------------------------------------- errors (thread-local, never abort)
static thread_local std::string t_lastError;

void set_error(const char* fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    t_lastError = buf;
    if (tuning().verbose.load(std::memory_order_relaxed)) fprintf(stderr, "[MgcgGpu] %s\n", buf);
}

// ---------------------------------------------------------------- tuning knobs (common.hpp)
namespace {
struct Knob { const char* env; const char* name; std::atomic<int> Tuning::*field; int unset; bool flag; };   // flag: the variable's presence means 1
const Knob kKnobs[] = {
    { "MGCG_OVERLAP", "overlap", &Tuning::overlap, 1, false },
    { "MGCG_NO_FOLD", "no_fold", &Tuning::noFold, 0, true },
    { "MGCG_FOLD_UP", "fold_up", &Tuning::foldUp, -1, false },
    { "MGCG_CHECK_EVERY", "check_every", &Tuning::checkEvery, 4, false },
    { "MGCG_TILE_SHIFT", "tile_shift", &Tuning::tileShift, 0, false },
    { "MGCG_TILE_PACK", "tile_pack", &Tuning::tilePack, 1, false },
    { "MGCG_AUTO_TILES", "auto_tiles", &Tuning::autoTiles, 1, false },
    { "MGCG_VERBOSE", "verbose", &Tuning::verbose, 0, false },
    { "MGCG_VIRTUAL_DEVICES", "virtual_devices", &Tuning::virtualDevices, 0, false },
    { "MGCG_HALO_STREAM", "halo_stream", &Tuning::haloStream, 0, false },
    { "MGCG_FORCE_MULTIRANK", "force_multirank", &Tuning::forceMultiRank, 0, false },
    { "MGCG_FAIL_COMM_INIT", "fail_comm_init", &Tuning::failCommInit, 0, false },
    { "MGCG_DOT_ORDER", "dot_order", &Tuning::dotOrder, 0, false },
    { "MGCG_DEEP_HALO", "deep_halo", &Tuning::deepHalo, 1, false },
    { "MGCG_PLACEMENT", "placement", &Tuning::placement, 3, false },
};
Tuning g_tuning;
std::once_flag g_tuningOnce;
void tuning_read_environment()
{
    for (const Knob& k : kKnobs) {
        const char* e = getenv(k.env);
        (g_tuning.*(k.field)).store(e ? (k.flag ? 1 : atoi(e)) : k.unset, std::memory_order_relaxed);
    }
}
} // namespace
Tuning& tuning() { std::call_once(g_tuningOnce, tuning_read_environment); return g_tuning; }
void tuning_reload() { (void)tuning(); tuning_read_environment(); }

bool hip_ok(hipError_t e, const char* what, const char* file, int line)
{
    if (e == hipSuccess) return true;
    set_error("%s failed: %s (%s:%d)", what, hipGetErrorString(e), file, line);
    return false;
}

// ---------------------------------------------------------------- devices
// MGCG_VIRTUAL_DEVICES=n makes GetDeviceCount() report n devices mapped round-robin onto the
// physical ones, so the reference's multi-device host logic (ConjugateGradientParallelGpu) can be
// exercised on a one-GPU box.
static int physical_count()
{
    static int n = -2;
    if (n == -2) { int c = 0; if (hipGetDeviceCount(&c) != hipSuccess) c = 0; n = c; }
    return n;
}
static int virtual_count()
{
    const int v = tuning().virtualDevices.load(std::memory_order_relaxed);
    if (v > 0 && physical_count() > 0) return v;
    return physical_count();
}

static thread_local int t_device = 0;   // virtual id
static std::mutex g_devMutex;
static DeviceState g_dev[kMaxDevices];

int current_device() { return t_device; }

// hipSetDevice for the calling thread's (virtual) device and nothing else: no stream, no allocation
bool select_device_only()
{
    const int phys = physical_count();
    if (phys <= 0) { set_error("no HIP device available (hipGetDeviceCount = %d): the HIP path cannot run", phys); return false; }
    return MGCG_HIP(hipSetDevice(t_device % phys));
}

DeviceState* device_state()
{
    const int phys = physical_count();
    if (phys <= 0) { set_error("no HIP device available (hipGetDeviceCount = %d): the HIP path cannot run", phys); return nullptr; }
    const int vd = t_device;
    if (vd < 0 || vd >= kMaxDevices) { set_error("device id %d out of range", vd); return nullptr; }
    const int pd = vd % phys;
    if (!MGCG_HIP(hipSetDevice(pd))) return nullptr;
    std::lock_guard<std::mutex> lock(g_devMutex);
    DeviceState* d = &g_dev[pd];                       // virtual devices on one physical GPU share its stream
    if (d->stream == nullptr) {
        d->device = pd;
        if (!MGCG_HIP(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking))) return nullptr;
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, pd) == hipSuccess) d->numCu = prop.multiProcessorCount;
        preload_kernels_spmv(); preload_kernels_rowtile(); preload_kernels_blas1(); preload_solver(); preload_ops();
        preload_kernels_rows(); preload_kernels_mg(); preload_kernels_dcsr(); preload_kernels_tiled(); preload_comm(); preload_spectrum();
    }
    return d;
}

// ---------------------------------------------------------------- workspace
bool Workspace::init()
{
    DeviceState* d = device_state();
    if (!d) return false;
    device = current_device();
    stream = d->stream;
    if (!MGCG_HIP(hipMalloc((void**)&partials, sizeof(double) * kMaxPartials * 3))) return false;
    if (!MGCG_HIP(hipMalloc((void**)&scalars, sizeof(CgScalars)))) return false;
    if (!MGCG_HIP(hipMemset(scalars, 0, sizeof(CgScalars)))) return false;
    if (!MGCG_HIP(hipHostMalloc((void**)&mirror, sizeof(HostMirror), hipHostMallocMapped))) return false;
    memset((void*)mirror, 0, sizeof(HostMirror));
    if (!MGCG_HIP(hipHostMalloc((void**)&hostScalar, sizeof(double) * 4, hipHostMallocMapped))) return false;
    if (!MGCG_HIP(hipMalloc((void**)&devInts, sizeof(int) * 8))) return false;
    return true;
}
void Workspace::destroy()
{
    if (stream) (void)hipStreamSynchronize(stream);     // nothing enqueued may still use what is freed below
    if (partials) (void)hipFree(partials);
    if (scalars) (void)hipFree(scalars);
    if (mirror) (void)hipHostFree((void*)mirror);
    if (hostScalar) (void)hipHostFree(hostScalar);
    if (devInts) (void)hipFree(devInts);
    if (trace) (void)hipFree(trace);
    partials = nullptr; scalars = nullptr; mirror = nullptr; hostScalar = nullptr; devInts = nullptr; trace = nullptr; traceCap = 0;
}
bool Workspace::ensure_trace(int cap)
{
    if (cap <= traceCap) return true;
    if (trace) (void)hipFree(trace);
    trace = nullptr; traceCap = 0;
    if (!MGCG_HIP(hipMalloc((void**)&trace, sizeof(double) * (size_t)cap))) return false;
    traceCap = cap;
    return true;
}

// ---------------------------------------------------------------- analysed matrices: registry and invalidation
// All analysed entries of all handles, so that a write through ANY handle or export can find the analyses it invalidates.
static std::mutex g_analysedMutex;
static std::vector<DcsrMatrix*> g_analysed;
static std::atomic<int> g_analysedCount{0};

static void registry_add(DcsrMatrix* m)
{
    std::lock_guard<std::mutex> lock(g_analysedMutex);
    g_analysed.push_back(m);
    g_analysedCount.store((int)g_analysed.size(), std::memory_order_relaxed);
}
void registry_remove(DcsrMatrix* m)
{
    std::lock_guard<std::mutex> lock(g_analysedMutex);
    for (size_t i = 0; i < g_analysed.size(); ++i) if (g_analysed[i] == m) { g_analysed.erase(g_analysed.begin() + (long)i); break; }
    g_analysedCount.store((int)g_analysed.size(), std::memory_order_relaxed);
}
void analysis_note_write(const void* p, size_t bytes)
{
    if (g_analysedCount.load(std::memory_order_relaxed) == 0 || p == nullptr || bytes == 0) return;
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    auto hits = [&](const void* base, size_t n) { const uintptr_t b = (uintptr_t)base; return base != nullptr && b < hi && lo < b + n; };
    std::lock_guard<std::mutex> lock(g_analysedMutex);
    for (DcsrMatrix* m : g_analysed)
        if (hits(m->elements, sizeof(double) * (size_t)m->nnz) || hits(m->columnIndeces, sizeof(int) * (size_t)m->nnz) ||
            hits(m->rowOffsets, sizeof(int) * (size_t)(m->rows + 1)))
            m->stale.store(true, std::memory_order_release);
}

// ---------------------------------------------------------------- device vectors the library allocated
static std::mutex g_allocMutex;
static std::map<uintptr_t, size_t> g_allocs;
void vector_registry_add(const void* p, size_t bytes) { if (p && bytes) { std::lock_guard<std::mutex> lock(g_allocMutex); g_allocs[(uintptr_t)p] = bytes; } }
void vector_registry_remove(const void* p) { if (p) { std::lock_guard<std::mutex> lock(g_allocMutex); g_allocs.erase((uintptr_t)p); } }
bool vector_owned(const void* p, size_t bytes)
{
    if (!p) return false;
    std::lock_guard<std::mutex> lock(g_allocMutex);
    auto it = g_allocs.upper_bound((uintptr_t)p);
    if (it == g_allocs.begin()) return false;
    --it;
    return (uintptr_t)p + bytes <= it->first + it->second;
}

const DcsrMatrix* dcsr_lookup_op(MgcgSparse* h, const SpmvArgs& a, long long rowBase)
{
    if (!h) return nullptr;
    if (h->compression != 0) return dcsr_lookup(h, a.elements, a.rowOffsets, a.columnIndeces, a.rowCount, a.elementsCount, rowBase, a.columnCount);
    if (a.elementsCount < (4 << 20) || tuning().autoTiles.load(std::memory_order_relaxed) == 0) return nullptr;
    MgcgSparse::PeriodEntry* pe = nullptr;             // (filled by spmv_period, which every per-op call makes first: cfg_for)
    for (auto& e : h->periods)
        if (e.rowOffsets == a.rowOffsets && e.columnIndeces == a.columnIndeces && e.rows == a.rowCount && e.rowBase == rowBase) { pe = &e; break; }
    if (!pe || pe->meanDistance < (1LL << 19)) return nullptr;
    // a form that exists and has seen no write since: use it
    for (DcsrMatrix* q : h->analysed)
        if (q->automatic && q->elements == a.elements && q->rowOffsets == a.rowOffsets && q->columnIndeces == a.columnIndeces && q->rows == a.rowCount &&
            q->nnz == a.elementsCount && q->rowBase == rowBase && !q->stale.load(std::memory_order_acquire)) {
            if (!q->usable) return nullptr;
            if (++q->trustedUses < 16) return q;
            // every 16th product: is the form still the matrix?  (writes through ToRawPtr_* pointers by the caller's own kernels or copies
            // are invisible to the write registry)
            q->trustedUses = 0;
            if (csr_checksum(h->ws.stream, a.elements, a.rowOffsets, a.columnIndeces, a.rowCount, a.elementsCount, (unsigned long long*)(h->ws.devInts + 4)) == q->checksum) return q;
            q->stale.store(true, std::memory_order_release);
            break;                                     // changed behind the library's back: the CSR kernels now, a rebuild after `threshold` products
        }
    // none, or written to since: the CSR kernels serve the next `threshold` products, then the tiles are (re)built -- a matrix that is rewritten
    // between its products pays for ever fewer builds (the threshold doubles with every rebuild)
    if (++pe->products < pe->threshold) return nullptr;
    if (!vector_owned(a.elements, sizeof(double) * (size_t)a.elementsCount) || !vector_owned(a.columnIndeces, sizeof(int) * (size_t)a.elementsCount) ||
        !vector_owned(a.rowOffsets, sizeof(int) * ((size_t)a.rowCount + 1))) { pe->products = 0; pe->threshold = 1 << 30; return nullptr; }   // not ours: never
    const DcsrMatrix* m = dcsr_lookup(h, a.elements, a.rowOffsets, a.columnIndeces, a.rowCount, a.elementsCount, rowBase, a.columnCount, pe->meanDistance, true);
    for (auto& e : h->periods)                         // (dcsr_lookup does not touch h->periods, but stay safe against reallocation)
        if (e.rowOffsets == a.rowOffsets && e.columnIndeces == a.columnIndeces && e.rows == a.rowCount && e.rowBase == rowBase) { e.products = 0; if (e.threshold < (1 << 20)) e.threshold *= 2; break; }
    return m;
}

const DcsrMatrix* dcsr_lookup(MgcgSparse* h, const double* elements, const int* rowOffsets, const int* columnIndeces,
                              long long rows, long long nnz, long long rowBase, long long columns, long long autoMeanDistance, bool trustRegistry)
{
    if (!h) return nullptr;
    // The library's own choice (compression off, a Solve-family call): column tiles for a large matrix whose gathers have no locality --
    // x far beyond one L2 and the sampled entries a tile width or more from the diagonal (BASELINE config 5: 2.6 ms per product instead of
    // 5.7-6.1).  The form is lossless and bit-identical; because the caller did not ask for a cache keyed by pointers, the CSR arrays are
    // re-verified by checksum at every solve (one streaming read: 0.8 ms for 310 M nonzeros) and the form is rebuilt when they changed.
    const bool automatic = h->compression == 0;
    if (automatic) {
        const long long tileCols = 1LL << 19;
        const bool take = !(autoMeanDistance < tileCols || columns < 8 * tileCols || nnz < (4LL << 20) || rows <= 0 || nnz > 64 * rows ||
                            tuning().autoTiles.load(std::memory_order_relaxed) == 0 || tuning().tileShift.load(std::memory_order_relaxed) != 0);
        if (tuning().verbose.load(std::memory_order_relaxed) >= 2 && autoMeanDistance >= 0)
            fprintf(stderr, "[MgcgGpu] column tiles by the library's choice: %s (rows %lld, columns %lld, nnz %lld, sampled distance %lld)\n",
                    take ? "considered" : "no", rows, columns, nnz, autoMeanDistance);
        if (!take) return nullptr;
    }
    DcsrMatrix* m = nullptr;
    for (DcsrMatrix* q : h->analysed)
        if (q->elements == elements && q->rowOffsets == rowOffsets && q->columnIndeces == columnIndeces && q->rows == rows && q->nnz == nnz && q->rowBase == rowBase &&
            q->automatic == automatic) {                    // (forms the caller asked for and the library's own choice are separate entries)
            if (!q->stale.load(std::memory_order_acquire)) {
                if (!q->automatic) return q->usable ? q : nullptr;
                if (!q->usable) return nullptr;
                if (trustRegistry) return q;
                if (csr_checksum(h->ws.stream, elements, rowOffsets, columnIndeces, rows, nnz, (unsigned long long*)(h->ws.devInts + 4)) == q->checksum) return q;
            }
            m = q;                                          // written to since: same slot, new analysis
            break;
        }
    if (m == nullptr && (rows <= 0 || nnz < 8)) return nullptr;
    const double avg = rows > 0 ? (double)nnz / (double)rows : 0.0;
    if (m == nullptr && avg > 64.0) return nullptr;
    // The identity fields are what analysis_note_write (any thread, any device) compares a written range with: they change only under
    // the registry's mutex.  stale is cleared BEFORE the arrays are read again, so a write that arrives while the new analysis is being
    // built marks it stale once more instead of being lost.
    auto identify = [&] { m->elements = elements; m->rowOffsets = rowOffsets; m->columnIndeces = columnIndeces; m->rows = rows; m->nnz = nnz; m->rowBase = rowBase; };
    if (m != nullptr) {
        (void)hipStreamSynchronize(h->ws.stream);           // kernels that still read the old form
        std::lock_guard<std::mutex> lock(g_analysedMutex);
        m->release();
        m->stale.store(false, std::memory_order_release);
        identify();
    } else {
        m = new DcsrMatrix();
        identify();
        h->analysed.push_back(m);
        registry_add(m);
    }
    m->automatic = automatic;
    if (automatic) {
        m->checksum = csr_checksum(h->ws.stream, elements, rowOffsets, columnIndeces, rows, nnz, (unsigned long long*)(h->ws.devInts + 4));
        if (tiled_build(h->ws.stream, elements, rowOffsets, columnIndeces, rows, nnz, rowBase, columns, m) && m->tileVals != nullptr) m->usable = true;
        return m->usable ? m : nullptr;
    }
    // 1. one byte per row (few distinct rows-as-sequences), 2. one or two bytes per nonzero (few distinct offsets / values;
    //    the wide loads of that kernel need 16-byte aligned values and short average rows: one pass per 64-row block)
    // 3. a column-tiled copy for matrices whose gathers have no locality (sorted rows, entries far from the diagonal)
    if (h->compression == 1 && avg <= 32.0 && pattern_build(h->ws.stream, elements, rowOffsets, columnIndeces, rows, nnz, rowBase, m) && m->patternId != nullptr) {
        m->usable = true;
    } else if ((((uintptr_t)elements) & 15) == 0 && avg <= 7.75) {
        if (!dcsr_build(h->ws.stream, elements, rowOffsets, columnIndeces, rows, nnz, rowBase, m)) { std::lock_guard<std::mutex> lock(g_analysedMutex); m->release(); identify(); }
    }
    if (!m->usable && h->compression == 1 && columns > 0) {
        if (tiled_build(h->ws.stream, elements, rowOffsets, columnIndeces, rows, nnz, rowBase, columns, m) && m->tileVals != nullptr) m->usable = true;
    }
    h->analysedMode = h->compression;                  // (also when the mode came from MGCG_COMPRESSION, not from the setter)
    return m->usable ? m : nullptr;
}

int spmv_period(MgcgSparse* h, const int* rowOffsets, const int* columnIndeces, long long rows, long long rowBase, int* maxRow, long long* meanDistance)
{
    if (maxRow) *maxRow = 0;
    if (meanDistance) *meanDistance = 0;
    if (!h || rows < 4096) return h ? h->periodRows : 0;
    for (const auto& e : h->periods)
        if (e.rowOffsets == rowOffsets && e.columnIndeces == columnIndeces && e.rows == rows && e.rowBase == rowBase) {
            if (maxRow) *maxRow = e.maxRow;
            if (meanDistance) *meanDistance = e.meanDistance;
            return h->periodRows > 0 ? h->periodRows : e.period;
        }
    int period = 0, longest = 0;
    long long mean = 0;
    int got[4] = { 0, 0, 0, 0 };
    bool ok = MGCG_HIP(hipMemsetAsync(h->ws.devInts, 0, 4 * sizeof(int), h->ws.stream));
    if (ok) launch_matrix_shape(h->ws.stream, rowOffsets, columnIndeces, rows, rows / 2, rowBase, h->ws.devInts);
    ok = ok && MGCG_HIP(hipGetLastError()) && MGCG_HIP(hipMemcpyAsync(got, h->ws.devInts, sizeof(got), hipMemcpyDeviceToHost, h->ws.stream)) && MGCG_HIP(hipStreamSynchronize(h->ws.stream));
    if (ok) { period = got[0]; longest = got[1]; mean = got[3] > 0 ? (((long long)got[2] << 16) / got[3]) : 0; }
    if (h->periods.size() >= 64) h->periods.clear();
    h->periods.push_back({ rowOffsets, columnIndeces, rows, rowBase, period, longest, mean });
    if (maxRow) *maxRow = longest;
    if (meanDistance) *meanDistance = mean;
    return h->periodRows > 0 ? h->periodRows : period;
}

int launch_spmv_auto(hipStream_t s, int epilogue, const SpmvArgs& a, const SpmvConfig& cfg, const DcsrMatrix* dc)
{
    if (dc != nullptr && dc->usable && a.elementsCount >= 8) {
        const DcsrView v = dc->view();
        if (v.tileVals == nullptr) return launch_spmv_rows(s, epilogue, a, &v, cfg.gridBlocks, cfg.periodRows);
        if (!(epilogue == EPI_AXPBY && a.beta != 0.0)) return launch_spmv_tiled(s, epilogue, a, v);   // (beta != 0 reads y: CSR kernels)
    }
    return launch_spmv(s, epilogue, a, cfg);
}

int launch_spmv_range(hipStream_t s, int epilogue, const SpmvArgs& whole, const SpmvConfig& cfg, const DcsrMatrix* dc,
                      long long r0, long long r1, double* partials, int maxGrid)
{
    if (r1 <= r0) return 0;
    SpmvArgs a = whole;
    a.rowOffsets += r0; a.y += r0; a.rowCount = (int)(r1 - r0); a.partials = partials;   // offsets stay absolute into elements / columnIndeces
    a.cRowBase += (int)r0;
    if (a.w) a.w += r0;
    if (a.b) a.b += r0;
    if (a.dinv) a.dinv += r0;
    const bool compressed = dc != nullptr && dc->usable && whole.elementsCount >= 8;
    SpmvConfig c = cfg;
    c.flags &= ~6;                                       // the XCD / banded maps assume the whole matrix
    if (maxGrid > 0) {
        const int numCu = device_state() ? device_state()->numCu : kNumCu;
        int g = c.gridBlocks > 0 ? c.gridBlocks : (compressed ? 16 : 8) * numCu;
        c.gridBlocks = g < maxGrid ? g : maxGrid;
    }
    if (compressed) {
        DcsrView v = dc->view();
        v.rowBase += r0;
        if (v.patternId) v.patternId += r0;
        if (v.tileVals != nullptr) {                                            // the tiled passes cover whole matrices only: CSR kernels for a row range
            if (c.kernel == 0) c.kernel = spmv_auto_kernel(whole.rowCount > 0 ? (double)whole.elementsCount / (double)whole.rowCount : 0.0);
            return launch_spmv(s, epilogue, a, c);
        }
        return launch_spmv_rows(s, epilogue, a, &v, c.gridBlocks, c.periodRows);
    }
    if (c.kernel == 0) {                                 // the kernel choice follows the whole matrix, not the slice
        c.kernel = spmv_auto_kernel(whole.rowCount > 0 ? (double)whole.elementsCount / (double)whole.rowCount : 0.0);
    }
    return launch_spmv(s, epilogue, a, c);
}

int launch_spmv_two_ranges(hipStream_t s, int epilogue, const SpmvArgs& whole, const SpmvConfig& cfg, const DcsrMatrix* dc,
                           long long i0, long long i1, double* partials, int maxGrid)
{
    const long long n = whole.rowCount;
    if (i0 < 0) i0 = 0;
    if (i1 > n) i1 = n;
    if (i0 >= i1) return launch_spmv_range(s, epilogue, whole, cfg, dc, 0, n, partials, maxGrid);
    const bool compressed = dc != nullptr && dc->usable && whole.elementsCount >= 8;
    int kernel = cfg.kernel;
    if (kernel == 0) kernel = spmv_auto_kernel(n > 0 ? (double)whole.elementsCount / (double)n : 0.0);
    const bool aligned = (((uintptr_t)whole.elements & 15) == 0) && (((uintptr_t)whole.columnIndeces & 15) == 0) && whole.elementsCount >= 8;
    if (!compressed && kernel == 10 && aligned && i0 % kRowTileRows == 0 && i1 % kRowTileRows == 0 && (i0 > 0 || i1 < n)) {
        SpmvArgs a = whole;
        a.partials = partials;
        int grid = cfg.gridBlocks;
        if (maxGrid > 0) {
            const int numCu = device_state() ? device_state()->numCu : kNumCu;
            const int g = cfg.gridBlocks > 0 ? cfg.gridBlocks : 8 * numCu;
            grid = g < maxGrid ? g : maxGrid;
        }
        return launch_spmv_rowtile(s, epilogue, a, 0, grid, cfg.maxRow, (cfg.flags & 8) != 0, (int)(i0 / kRowTileRows), (int)((i1 - i0) / kRowTileRows));
    }
    int k = launch_spmv_range(s, epilogue, whole, cfg, dc, 0, i0, partials, maxGrid > 0 ? maxGrid / 2 : 0);
    k += launch_spmv_range(s, epilogue, whole, cfg, dc, i1, n, partials ? partials + k : nullptr, maxGrid > 0 ? maxGrid / 2 : 0);
    return k;
}

} // namespace mgcg

using namespace mgcg;

// ---------------------------------------------------------------- device vectors
template <typename V, typename T>
static V* create_vec(long long size)
{
    DeviceState* d = device_state();
    if (!d) return nullptr;
    if (size < 0) { set_error("negative vector size %lld", size); return nullptr; }
    V* v = new V();
    v->size = size; v->device = current_device();
    if (size > 0) {
        if (!MGCG_HIP(hipMalloc((void**)&v->data, sizeof(T) * (size_t)size))) { delete v; return nullptr; }
        // thrust::device_vector<T>(size) value-initialises (Vector_Double.cu:9)
        if (!MGCG_HIP(hipMemsetAsync(v->data, 0, sizeof(T) * (size_t)size, d->stream))) { (void)hipFree(v->data); delete v; return nullptr; }
        mgcg::vector_registry_add(v->data, sizeof(T) * (size_t)size);
    }
    return v;
}

template <typename V, typename T>
static void copy_to_array(const V* src, T* dst, int count, int srcOff, int dstOff, const char* name)
{
    DeviceState* d = device_state();
    if (!d) return;
    if (!src || !dst) { set_error("%s: null argument", name); return; }
    if (count < 0 || srcOff < 0 || (long long)srcOff + count > src->size) { set_error("%s: range [%d,+%d) outside vector of %lld", name, srcOff, count, src->size); return; }
    if (count == 0) return;
    if (!MGCG_HIP(hipMemcpyAsync(dst + dstOff, src->data + srcOff, sizeof(T) * (size_t)count, hipMemcpyDeviceToHost, d->stream))) return;
    (void)MGCG_HIP(hipStreamSynchronize(d->stream));
}

template <typename V, typename T>
static void copy_from_array(V* dst, const T* src, int count, int srcOff, int dstOff, const char* name)
{
    DeviceState* d = device_state();
    if (!d) return;
    if (!src || !dst) { set_error("%s: null argument", name); return; }
    if (count < 0 || dstOff < 0 || (long long)dstOff + count > dst->size) { set_error("%s: range [%d,+%d) outside vector of %lld", name, dstOff, count, dst->size); return; }
    if (count == 0) return;
    analysis_note_write(dst->data + dstOff, sizeof(T) * (size_t)count);
    if (!MGCG_HIP(hipMemcpyAsync(dst->data + dstOff, src + srcOff, sizeof(T) * (size_t)count, hipMemcpyHostToDevice, d->stream))) return;
    (void)MGCG_HIP(hipStreamSynchronize(d->stream));   // pageable source must stay valid: complete before returning
}

extern "C" {

const char* MgcgGetLastError(void) { return mgcg::t_lastError.c_str(); }
void MgcgClearLastError(void) { mgcg::t_lastError.clear(); }
int MgcgAbiVersion(void) { return 3; }   // 3: round 5 -- dot_order, 14 knobs retired, Vector / DcsrMatrix grew

void MgcgReloadEnvironment(void) { mgcg::tuning_reload(); }
int MgcgSetTuning(const char* name, int value)
{
    if (name) for (const mgcg::Knob& k : mgcg::kKnobs)
        if (strcmp(name, k.name) == 0 || strcmp(name, k.env) == 0) { (mgcg::tuning().*(k.field)).store(value, std::memory_order_relaxed); return 0; }
    mgcg::set_error("MgcgSetTuning: unknown knob '%s'", name ? name : "(null)");
    return -1;
}
int MgcgGetTuning(const char* name, int* value)
{
    if (name && value) for (const mgcg::Knob& k : mgcg::kKnobs)
        if (strcmp(name, k.name) == 0 || strcmp(name, k.env) == 0) { *value = (mgcg::tuning().*(k.field)).load(std::memory_order_relaxed); return 0; }
    mgcg::set_error("MgcgGetTuning: unknown knob '%s'", name ? name : "(null)");
    return -1;
}

int GetDeviceCount(void) { return virtual_count(); }

void SetDevice(int deviceID)
{
    const int n = virtual_count();
    if (deviceID < 0 || deviceID >= (n > 0 ? n : 1)) { set_error("SetDevice(%d): only %d device(s)", deviceID, n); return; }
    t_device = deviceID;
    const int phys = physical_count();
    if (phys > 0) (void)MGCG_HIP(hipSetDevice(deviceID % phys));
}

MgcgBlas* CreateBlas(void)
{
    MgcgBlas* h = new MgcgBlas();
    if (!h->ws.init()) { h->ws.destroy(); delete h; return nullptr; }
    return h;
}
void DestroyBlas(MgcgBlas* h) { if (!h) return; h->ws.destroy(); delete h; }

MgcgSparse* CreateSparse(void)
{
    MgcgSparse* h = new MgcgSparse();
    if (!h->ws.init()) { h->ws.destroy(); delete h; return nullptr; }
    const char* k = getenv("MGCG_SPMV_KERNEL");       if (k) h->kernel = atoi(k);
    const char* r = getenv("MGCG_SPMV_ROWS");         if (r) h->rowsPerBlock = atoi(r);
    const char* f = getenv("MGCG_SPMV_FLAGS");        if (f) h->flags = atoi(f);
    const char* g = getenv("MGCG_SPMV_GRID");         if (g) h->gridBlocks = atoi(g);
    const char* tr = getenv("MGCG_SPMV_TILE_ROWS");   if (tr) h->tileRows = atoi(tr);
    const char* tp = getenv("MGCG_SPMV_TILE_PLANES"); if (tp) h->tilePlanes = atoi(tp);
    const char* cm = getenv("MGCG_COMPRESSION");      if (cm) { const int v = atoi(cm); const int mode = v < 0 ? 0 : (v > 2 ? 1 : v); h->compression = mode; if (mode != 0) h->analysedMode = mode; }   // as MgcgSetMatrixCompression
    const char* p = getenv("MGCG_SPMV_PERIOD");       if (p) { h->periodRows = atoi(p); if (h->periodRows > 0) h->flags |= 4; }
    return h;
}
void DestroySparse(MgcgSparse* h)
{
    if (!h) return;
    if (h->ws.stream) (void)hipStreamSynchronize(h->ws.stream);      // kernels that still read the analysed forms
    for (auto* m : h->analysed) { registry_remove(m); m->release(); delete m; }
    for (hipEvent_t e : h->prof.start) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->prof.stop) (void)hipEventDestroy(e);
    h->ws.destroy();
    delete h;
}

void MgcgAnalysisClear(MgcgSparse* h);
void MgcgSetMatrixCompression(MgcgSparse* h, int enable)
{
    if (!h) return;
    const int mode = enable < 0 ? 0 : (enable > 2 ? 1 : enable);
    if (mode != 0 && h->analysedMode != 0 && mode != h->analysedMode && !h->analysed.empty()) MgcgAnalysisClear(h);   // another form was asked for: analyse again
    if (mode != 0) h->analysedMode = mode;
    h->compression = mode;
}
void MgcgAnalysisClear(MgcgSparse* h)
{
    if (!h) return;
    DeviceState* d = device_state();
    if (d) (void)hipStreamSynchronize(d->stream);
    for (auto* m : h->analysed) { registry_remove(m); m->release(); delete m; }
    h->analysed.clear();
}
int MgcgAnalysisInfo(MgcgSparse* h, int index, int* distinctOffsets, int* distinctValues, long long* rows, long long* nnz)
{
    if (!h || index < 0 || index >= (int)h->analysed.size()) return -1;
    const mgcg::DcsrMatrix* m = h->analysed[(size_t)index];
    const bool pat = m->patternId != nullptr;
    if (distinctOffsets) *distinctOffsets = pat ? m->nPattern : m->nDelta;     // class 3: distinct rows-as-sequences
    if (distinctValues) *distinctValues = pat ? m->patWidth : m->nValue;       // class 3: longest row
    if (rows) *rows = m->rows;
    if (nnz) *nnz = m->nnz;
    if (!m->usable) return 0;
    if (m->tileVals != nullptr) { if (distinctOffsets) *distinctOffsets = m->nTiles; if (distinctValues) *distinctValues = 0; return 4; }   // class 4: column tiles
    return pat ? 3 : (m->valCode ? 2 : 1);   // 3 = one byte per row, 2 = offset + value code per nonzero, 1 = offset code per nonzero, 0 = plain CSR
}

MgcgMatDescr* CreateMatDescr(void) { return new MgcgMatDescr(); }
void DestroyMatDescr(MgcgMatDescr* d) { delete d; }

void MgcgSetSpmvKernel(MgcgSparse* h, int kernel) { if (h) h->kernel = kernel; }
void MgcgSetSpmvTuning(MgcgSparse* h, int rowsPerBlock, int flags, int gridBlocks)
{
    if (!h) return;
    h->rowsPerBlock = rowsPerBlock; h->flags = flags; h->gridBlocks = gridBlocks;
}
void MgcgSetSpmvPeriod(MgcgSparse* h, int periodRows)
{
    if (!h) return;
    h->periodRows = periodRows;
    if (periodRows > 0) h->flags |= 4; else h->flags &= ~4;
}
void MgcgSetSpmvTile(MgcgSparse* h, int tileRows, int tilePlanes)
{
    if (!h) return;
    h->tileRows = tileRows; h->tilePlanes = tilePlanes;
}

int MgcgDeviceSynchronize(void)
{
    DeviceState* d = device_state();
    if (!d) return -1;
    return MGCG_HIP(hipStreamSynchronize(d->stream)) ? 0 : -1;
}

void* MgcgEventCreate(void)
{
    if (!device_state()) return nullptr;
    hipEvent_t e;
    if (!MGCG_HIP(hipEventCreate(&e))) return nullptr;
    return (void*)e;
}
void MgcgEventRecord(void* ev)
{
    DeviceState* d = device_state();
    if (!d || !ev) return;
    (void)MGCG_HIP(hipEventRecord((hipEvent_t)ev, d->stream));
}
float MgcgEventElapsedMs(void* start, void* stop)
{
    float ms = -1.0f;
    if (!start || !stop) return ms;
    if (!MGCG_HIP(hipEventSynchronize((hipEvent_t)stop))) return -1.0f;
    if (!MGCG_HIP(hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop))) return -1.0f;
    return ms;
}
void MgcgEventDestroy(void* ev) { if (ev) (void)hipEventDestroy((hipEvent_t)ev); }

int MgcgMemGetInfo(long long* freeBytes, long long* totalBytes)
{
    if (!device_state()) return -1;
    size_t f = 0, t = 0;
    if (!MGCG_HIP(hipMemGetInfo(&f, &t))) return -1;
    if (freeBytes) *freeBytes = (long long)f;
    if (totalBytes) *totalBytes = (long long)t;
    return 0;
}

Vector* Create_Double(int size) { return create_vec<Vector, double>(size); }
Vector* MgcgCreateDouble64(long long size) { return create_vec<Vector, double>(size); }
VectorInt* Create_Int(int size) { return create_vec<VectorInt, int>(size); }
VectorInt* MgcgCreateInt64(long long size) { return create_vec<VectorInt, int>(size); }
long long MgcgVectorSize(const Vector* v) { return v ? v->size : 0; }

void CopyToArray_Double(const Vector* s, double d[], int count, int so, int dofs) { copy_to_array<Vector, double>(s, d, count, so, dofs, "CopyToArray_Double"); }
void CopyFromArray_Double(Vector* d, const double s[], int count, int so, int dofs) { copy_from_array<Vector, double>(d, s, count, so, dofs, "CopyFromArray_Double"); }
void CopyToArray_Int(const VectorInt* s, int d[], int count, int so, int dofs) { copy_to_array<VectorInt, int>(s, d, count, so, dofs, "CopyToArray_Int"); }
void CopyFromArray_Int(VectorInt* d, int s[], int count, int so, int dofs) { copy_from_array<VectorInt, int>(d, s, count, so, dofs, "CopyFromArray_Int"); }

void Delete_Double(Vector* v)
{
    if (!v) return;
    if (v->data) { analysis_note_write(v->data, sizeof(double) * (size_t)v->size); vector_registry_remove(v->data); DeviceState* d = device_state(); if (d) (void)hipStreamSynchronize(d->stream); (void)hipFree(v->data); }
    delete v;
}
void Delete_Int(VectorInt* v)
{
    if (!v) return;
    if (v->data) { analysis_note_write(v->data, sizeof(int) * (size_t)v->size); vector_registry_remove(v->data); DeviceState* d = device_state(); if (d) (void)hipStreamSynchronize(d->stream); (void)hipFree(v->data); }
    delete v;
}
double* ToRawPtr_Double(Vector* v) { if (v) v->rawExported = true; return v ? v->data : nullptr; }
int* ToRawPtr_Int(VectorInt* v) { return v ? v->data : nullptr; }

void CopyFromDevice_Double(const double* source, double* destination, int count, int sourceOffset, int destinationOffset)
{
    DeviceState* d = device_state();
    if (!d) return;
    if (!source || !destination || count < 0) { set_error("CopyFromDevice_Double: bad argument"); return; }
    if (count == 0) return;
    // device-to-device, possibly across devices (hipMemcpyDefault resolves the peers); count in ELEMENTS
    analysis_note_write(destination + destinationOffset, sizeof(double) * (size_t)count);
    if (!MGCG_HIP(hipMemcpyAsync(destination + destinationOffset, source + sourceOffset, sizeof(double) * (size_t)count, hipMemcpyDefault, d->stream))) return;
    (void)MGCG_HIP(hipStreamSynchronize(d->stream));
}

void MgcgFill(Vector* v, double value)
{
    DeviceState* d = device_state();
    if (!d || !v) return;
    analysis_note_write(v->data, sizeof(double) * (size_t)v->size);
    launch_fill(d->stream, v->data, value, v->size);
}

} // extern "C"
