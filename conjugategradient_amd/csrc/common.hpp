// Internal declarations shared by the translation units of libMgcgGpu.so.
// gfx950 (MI355X, CDNA4) only: 64-wide wavefronts, 256 CUs in 8 XCDs.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdarg>
#include <cstring>
#include <cmath>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

// Only the C ABI is exported from the shared object (-fvisibility=hidden for everything else).
#pragma GCC visibility push(default)
#include "../../include/MgcgGpu.h"
#pragma GCC visibility pop

namespace mgcg {

constexpr int kWave = 64;
constexpr int kBlock = 256;          // threads per workgroup for every streaming kernel
constexpr int kNumXcd = 8;
constexpr int kNumCu = 256;
constexpr int kMaxGrid = kNumCu * 8; // 8 resident 256-thread workgroups per CU (32 waves/CU)
constexpr int kMaxPartials = kNumCu * 32; // single-wavefront workgroups: up to 32 per CU, one partial sum each
constexpr int kMaxDevices = 64;
constexpr int kPatMax = 256;         // row-pattern form: distinct rows-as-sequences a matrix may have ...
constexpr int kPatEntries = 2048;    // ... and nPattern * longest row (the table lives in LDS: 12 bytes per entry)

// ---------------------------------------------------------------- errors
void set_error(const char* fmt, ...);
bool hip_ok(hipError_t e, const char* what, const char* file, int line);
#define MGCG_HIP(call) ::mgcg::hip_ok((call), #call, __FILE__, __LINE__)

// ---------------------------------------------------------------- tuning knobs
// Every MGCG_* environment variable the library honours is read ONCE (first use) into this block of atomics; launches read
// the atomics, never the environment (getenv racing a setenv of another thread is undefined, and ranks that read different
// values would take different collective paths).  MgcgSetTuning(name, value) changes one knob for the in-process A/B tools,
// MgcgReloadEnvironment() reads the environment again.  Nothing here changes results: every knob picks between bit-identical
// schedules (include/MgcgGpu.h lists them).
struct Tuning {
    std::atomic<int> overlap{1};             // MGCG_OVERLAP            0 off, 1 where the measured exchange costs more than the hops that hide it, 2 whenever an interior exists
    std::atomic<int> noFold{0};              // MGCG_NO_FOLD            V(1,*): store the first sweep instead of forming it per gather
    std::atomic<int> foldUp{-1};             // MGCG_FOLD_UP            V(1,1): x1 + P e formed per gather of the last sweep instead of a prolongation kernel + stored iterate:
                                             //                         -1 by level size (it pays up to a few ten million rows), 0 never, 1 wherever possible
    std::atomic<int> checkEvery{4};          // MGCG_CHECK_EVERY        iterations the host enqueues ahead of the stop flag
    std::atomic<int> tileShift{0};           // MGCG_TILE_SHIFT         column tiles of 2^shift columns (0: equal-width tiles of ~2.85 MiB of x, the default)
    std::atomic<int> tilePack{1};            // MGCG_TILE_PACK          column tiles with 12-byte entries (0: the 16-byte form)
    std::atomic<int> autoTiles{1};           // MGCG_AUTO_TILES         Solve-family calls build the column tiles themselves for matrices without locality
    std::atomic<int> verbose{0};             // MGCG_VERBOSE            errors also go to stderr
    std::atomic<int> virtualDevices{0};      // MGCG_VIRTUAL_DEVICES    one physical GPU shown as n devices (tests)
    std::atomic<int> haloStream{0};          // MGCG_HALO_STREAM        overlap schedule: 0 (default) the interior rows on the side stream, every RCCL call on the main stream;
                                             //                         1 the halo exchange on the side stream, all rows on the main stream (measured faster on one GPU, solver.hip;
                                             //                         opt-in until RCCL on two streams of one communicator has run on real multi-GPU hardware)
    std::atomic<int> forceMultiRank{0};      // MGCG_FORCE_MULTIRANK    a one-rank communicator takes the several-ranks code path (measurement)
    std::atomic<int> placement{3};           // MGCG_PLACEMENT          Solve-family calls on >= 32 M-entry p: time the SpMV on this many EXTRA allocations of p and keep the fastest (0: off)
    std::atomic<int> deepHalo{1};            // MGCG_DEEP_HALO          row-partitioned V(1,1): one exchange of a few planes of the right-hand side per coarse level and the sweeps
                                             //                         recomputed on those planes (4 exchanges per MGCG iteration), 0: one exchange per sweep (8)
    std::atomic<int> dotOrder{0};            // MGCG_DOT_ORDER          validation only: 1 = every dot product adds its rounded products strictly left to right and the ranks'
                                             //                         sums in rank order, as the reference's CPU twin does (LongVector.cs:15-31, resultsDot.Sum()) -- traces and
                                             //                         iterates then EQUAL the oracle's; ~0.5 s per 1.3e8-entry dot, never on a timed path
    std::atomic<int> failCommInit{0};        // MGCG_FAIL_COMM_INIT     tests only: MgcgCommInitAll / MgcgCommInitRank report failure (what a host without a working RCCL does)
};
Tuning& tuning();
void tuning_reload();

// ---------------------------------------------------------------- per-device state
// One stream per device, shared by every handle created on that device, so calls made through
// different handles stay ordered exactly as they are on the reference's default stream.
struct DeviceState {
    int device = -1;
    hipStream_t stream = nullptr;
    int numCu = kNumCu;
};
DeviceState* device_state();   // for the calling thread's current device (lazily created)
int current_device();
bool select_device_only();       // hipSetDevice for the calling thread's device; creates nothing

// The HIP runtime loads a translation unit's code object on the first launch of any of its kernels (1-9 ms each here: 12 ms
// in all, which would land inside the first timed Solve -- a third of the reference driver's 200 iterations).  Asking for a
// kernel's attributes forces the load; device_state() does that once per device for every translation unit of the library.
inline void preload_code_object(const void* kernel) { hipFuncAttributes at; (void)hipFuncGetAttributes(&at, kernel); }
void preload_ops(); void preload_solver(); void preload_comm(); void preload_kernels_spmv(); void preload_kernels_rows();
void preload_kernels_rowtile(); void preload_kernels_dcsr(); void preload_kernels_tiled(); void preload_kernels_blas1();
void preload_kernels_mg(); void preload_spectrum();

// Device scalars of one CG run (lives in the handle's workspace).
struct CgScalars {
    double rr;        // r.r  (or r.z for the preconditioned loop) of the previous iteration
    double pAp;
    double rrNew;     // r.r of this iteration (stop test)
    double rzNew;     // r.z of this iteration (preconditioned loop)
    double residual;
    double rr0;
    double nrmInf;
    double beta;
    double alpha;
    int iteration;    // index of the iteration being executed
    int done;         // set by the finalisation kernel; every later kernel exits at once
    int status;
    int pad;          // "x pending": set by the finalisation kernel for update_xp, cleared when no iteration ran
    // Copy of {rr, rr0, alpha, iteration, done} taken by update_r for the x/p update that also finalises the iteration
    // (single rank, no preconditioner): that kernel's workgroups read these while its first workgroup rewrites the live fields
    double fRr, fRr0, fAlpha;
    int fIteration, fDone;
};

static_assert(offsetof(CgScalars, rzNew) == offsetof(CgScalars, rrNew) + sizeof(double), "{rrNew, rzNew} are all-reduced as one pair");

// What the host polls (pinned, device-written).
struct HostMirror {
    volatile double residual;
    volatile int iteration;
    volatile int done;
    volatile int status;
    volatile int pad;
};

// ---------------------------------------------------------------- handles
struct Workspace {
    int device = -1;
    hipStream_t stream = nullptr;
    double* partials = nullptr;      // 3 * kMaxPartials doubles: per-workgroup partial sums (SpMV dot | max-norm | r.r of the fused single-rank path)
    CgScalars* scalars = nullptr;    // device
    HostMirror* mirror = nullptr;    // pinned host, device-visible
    double* hostScalar = nullptr;    // pinned host, 4 doubles (Dot results)
    int* devInts = nullptr;          // device, 8 ints (small integer results: far band, longest row, sampled distance; a 64-bit checksum in [4..5])
    double* trace = nullptr;         // device residual trace
    int traceCap = 0;
    bool init();
    void destroy();
    bool ensure_trace(int cap);
};

} // namespace mgcg

struct MgcgBlas   { mgcg::Workspace ws; };
namespace mgcg {
// Dictionary-compressed CSR (kernels_dcsr.hip): what the kernel sees ...
struct DcsrView {
    const unsigned char* colCode;    // code of (col - row) per nonzero
    const unsigned char* valCode;    // code of the value per nonzero, or nullptr (values stay fp64 in `elements`)
    const int* deltaDict;
    const double* valueDict;
    int nDelta, nValue;
    long long rowBase;               // global index of local row 0
    // row-pattern form (class 3): one byte per ROW names the row's whole (offsets, values) sequence
    const unsigned char* patternId;  // nullptr: not in pattern form
    const int* patCount;             // [nPattern] entries per pattern
    const int* patDelta;             // [nPattern * patWidth]
    const double* patValue;          // [nPattern * patWidth]
    int nPattern, patWidth;
    // column-tiled form (class 4): nonzeros re-laid out by column tile, row-major inside a tile
    const double* tileVals;          // nullptr: not tiled; else the nonzeros in tile-major, row-major order ...
    const int* tileCols;             // ... their column ids ...
    const int* tileRowIds;           // ... and (local) row ids
    const int* tileStartHost;        // HOST array [nTiles + 1]: first entry of every tile
    // 12-byte entries (the default when the row ids fit): tilePacked[k] = column offset inside the tile (low tileShift bits) | row - base row
    // of the entry's block of 1024 (high bits); tileHdr[b] = { base row, row of the entry in front of the block or -1 }, blocks numbered
    // tile by tile from tileHdrBaseHost[t]; tileCols / tileRowIds are then not kept
    const unsigned* tilePacked;
    const int2* tileHdr;
    const int* tileHdrBaseHost;      // HOST array [nTiles]: first header of the tile, or -1 for a tile that keeps 16-byte entries (its row ids do not fit)
    int tileShift;
    int tileWidth;                   // columns per tile (equal-width tiles, <= 2^tileShift)
    int nTiles;
    long long tileRows;              // rows of the analysed matrix
};
// ... and what the handle caches per analysed matrix (keyed by the CSR pointers and sizes).
struct DcsrMatrix {
    const double* elements = nullptr; const int* rowOffsets = nullptr; const int* columnIndeces = nullptr;
    long long rows = 0, nnz = 0, rowBase = 0;
    unsigned char* colCode = nullptr; unsigned char* valCode = nullptr;
    int* deltaDict = nullptr; double* valueDict = nullptr;
    int nDelta = 0, nValue = 0;
    unsigned char* patternId = nullptr; int* patCount = nullptr; int* patDelta = nullptr; double* patValue = nullptr;
    int nPattern = 0, patWidth = 0;
    double* tileVals = nullptr; int* tileCols = nullptr; int* tileRowIds = nullptr; int nTiles = 0; long long tileRows = 0;
    unsigned* tilePacked = nullptr; int2* tileHdr = nullptr; int tileShift = 0, tileWidth = 0;
    std::vector<int> tileStart, tileHdrBase;
    unsigned long long checksum = 0;  // of the CSR arrays the analysis was made from (forms chosen without the caller asking are re-verified per solve)
    int trustedUses = 0;              // per-op products served from this form since its checksum was last verified (dcsr_lookup_op: re-verified every 16th)
    bool automatic = false;           // built by the library's own choice (column tiles for a matrix without locality), not by MgcgSetMatrixCompression
    bool usable = false;
    std::atomic<bool> stale{false};   // a write through the library touched the arrays the analysis was made from: analyse again at the next use
    void release();
    DcsrView view() const
    {
        DcsrView v; v.colCode = colCode; v.valCode = valCode; v.deltaDict = deltaDict; v.valueDict = valueDict; v.nDelta = nDelta; v.nValue = nValue; v.rowBase = rowBase;
        v.patternId = patternId; v.patCount = patCount; v.patDelta = patDelta; v.patValue = patValue; v.nPattern = nPattern; v.patWidth = patWidth;
        v.tileVals = tileVals; v.tileCols = tileCols; v.tileRowIds = tileRowIds; v.tileStartHost = tileStart.data(); v.nTiles = nTiles; v.tileRows = tileRows;
        v.tilePacked = tilePacked; v.tileHdr = tileHdr; v.tileHdrBaseHost = tileHdrBase.data(); v.tileShift = tileShift; v.tileWidth = tileWidth;
        return v;
    }
};
// Optional per-launch timing of the SpMV inside the CG loop (bench.py's roofline figure).
struct SpmvProfile {
    bool enabled = false;
    std::vector<hipEvent_t> start, stop;
    int used = 0;
};
}
struct MgcgSparse {
    mgcg::Workspace ws;
    mgcg::SpmvProfile prof;
    int kernel = 0;          // 0 auto
    int rowsPerBlock = 64;
    int flags = 0;           // bit0 nt loads, bit1 xcd-contiguous mapping, bit2 banded schedule (periodRows)
    int gridBlocks = 0;
    int periodRows = 0;      // rows between strongly coupled windows (a grid plane); 0 = unknown
    int tileRows = 0, tilePlanes = 0;   // banded schedule tile (0 = default)
    int analysedMode = 0;               // the compression mode the cached analyses were built under
    int compression = 0;                // opt-in: 0 off, 1 best lossless compact form (row patterns, else per-nonzero codes), 2 per-nonzero codes only
    std::vector<mgcg::DcsrMatrix*> analysed;
    // far-band distance found per matrix (the tile order of the row-tile kernel; any value gives the same results, so a stale
    // entry can only cost locality): keyed by the array pointers and the row count
    struct PeriodEntry { const int* rowOffsets; const int* columnIndeces; long long rows, rowBase; int period; int maxRow; long long meanDistance;
                         int products = 0, threshold = 8; };   // per-op calls on this matrix since the column tiles were last (re)built / products before the next build
    std::vector<PeriodEntry> periods;
};
struct MgcgMatDescr { int type = 0; int base = 0; };
struct Vector    { double* data = nullptr; long long size = 0; int device = -1;
                   bool rawExported = false;   // ToRawPtr_Double handed the address out: the library must not move the data any more
                   bool placed = false;        // the placement draw (solver.hip) has looked at this vector
                   int drawCount = 0, drawChosen = -1; float drawMs[16] = {}; };   // ... and what it measured (MgcgLastPlacement)
struct VectorInt { int* data = nullptr;    long long size = 0; int device = -1; };

namespace mgcg {

// ---------------------------------------------------------------- SpMV
enum SpmvEpilogue {
    EPI_AXPBY = 0,   // y = alpha*(A x) + beta*y
    EPI_DOT = 1,     // y = A x ; partial += w_i * y_i
    EPI_RESIDUAL = 2,// y = b - A x
    EPI_JACOBI = 3,  // y = xo + omega*(dinv*(b - A x))
    EPI_RESIDUAL_DOT = 4, // y = b - A x ; partial += y_i*y_i
    EPI_AXPBY_BETA = 5,  // internal: EPI_AXPBY with beta != 0 (reads y)
    EPI_JACOBI_DOT = 6   // EPI_JACOBI ; partial += b_i * y_i   (last sweep of the V-cycle: r.z of the PCG loop rides along)
};
// Timing ablations that produce WRONG results exist only in lab builds of the library (make lab: -DMGCG_LAB).
#ifdef MGCG_LAB
#define MGCG_ABLATE(a, bits) (((a).ablate & (bits)) != 0)
#else
#define MGCG_ABLATE(a, bits) false
#endif
constexpr bool epi_has_dot(int e) { return e == EPI_DOT || e == EPI_RESIDUAL_DOT || e == EPI_JACOBI_DOT; }

struct SpmvArgs {
    const double* elements;
    const int* rowOffsets;
    const int* columnIndeces;
    const double* x;
    double* y;
    int elementsCount;
    int rowCount;
    int columnCount;
    double alpha, beta;      // EPI_AXPBY
    const double* w;         // EPI_DOT: weights (own slice of x); EPI_JACOBI: xo (own slice of x)
    const double* b;         // EPI_RESIDUAL / EPI_JACOBI
    const double* dinv;      // EPI_JACOBI
    double omega;            // EPI_JACOBI
    int dinvUniform;         // EPI_JACOBI: every diagonal is the same; dinvScalar is used and the dinv array is not read
    double dinvScalar;
    int xScaled;             // row-pattern and row-tile kernels: the multiplied vector is xOuter * (xInner * x[col]), formed per gather
    double xInner, xOuter;   //   (a first Jacobi sweep from zero folded into the residual pass of the V-cycle)
    // xScaled == 2 (row-tile kernel, Jacobi epilogues): ... + xCoarse[parent(col)] on top -- the piecewise-constant prolongation of the coarse
    // correction folded into the last sweep of a V(1,1) cycle; the sweep's own iterate (w) is formed the same way from b.  The grid has
    // power-of-two nx, ny (lx, ly bits), y / z halved when sy / sz = 1: parent(c) = ((c >> 1) & cM0) | ((c >> cS1) & cM1) | ((c >> cS2) & cM2) with
    // cM0 = nx/2 - 1, cS1 = 1 + sy, cM1 = ((ny >> sy) - 1) << (lx - 1), cS2 = 1 + sy + sz, cM2 = every bit from lx - 1 + ly - sy up
    const double* xCoarse;
    int cM0, cM1, cM2, cS1, cS2;
    int cRowBase;            //   global cell of the launch's row 0 (several ranks / a row range: the sweep's own parent is parent(cRowBase + row))
    double* partials;        // EPI_DOT / EPI_RESIDUAL_DOT / EPI_JACOBI_DOT: one double per workgroup
    const int* doneFlag;     // optional: exit at once when *doneFlag != 0
    int ablate;              // lab builds only (-DMGCG_LAB, tools/spmv_sweep.py --ablate): bit0 skip the y store, bit1 gathers from L1; the product never reads it
};

// Kernel family by average row length (measured on banded and random matrices, profiles/r1/rowlen_sweep.log):
// 10 row-tile "lane = row" form (kernels_rowtile.hip; 9 = its single-wavefront predecessor, kernels_rows.hip),
// 1 row-block stream form, 5 / 6 / 7 = 8 / 16 / 32 lanes per row.
// Validation mode (knob dot_order): every sum in the reference's order -- the lanes-per-row forms add a row's products lane by lane and
// then across lanes (1e-13 from the stored order of SparseMatrix.cs:68-88); the row-block stream form adds them in stored order.
bool dot_reference_order();
inline int spmv_auto_kernel(double avgRow)
{
    if (avgRow <= 8.0) return 10;
    if (avgRow <= 20.0 || dot_reference_order()) return 1;
    if (avgRow <= 28.0) return 5;
    if (avgRow <= 128.0) return 6;
    return 7;
}

// Order in which a lane = row kernel walks its tiles of `tileRows` rows (kernels_rowtile.hip explains the modes).
// gapAt / gapSkip: tiles [gapAt, gapAt + gapSkip) are left out (the enumeration runs over the remaining tiles): the two boundary row
// ranges of a rank's slice in ONE launch while the interior rows between them are multiplied on another stream.
struct TileMap { int mode; int tilesPerPlane; int nPlanes; int per; int gapAt; int gapSkip; };
constexpr int kRowTileRows = 256;    // rows per tile of the row-tile SpMV kernel (kernels_rowtile.hip)
TileMap make_tile_map(long long rows, int periodRows, int nWG, int tileRows);

struct SpmvConfig { int kernel = 0; int rowsPerBlock = 64; int flags = 0; int gridBlocks = 0; int periodRows = 0; int tileRows = 0; int tilePlanes = 0;
                    int maxRow = 0; /* longest row if known (row-tile kernel: 7 gathers per row when <= 7), 0 = unknown */ };

// Launches the SpMV; returns the number of partials written (grid size) for dot epilogues.
int launch_spmv(hipStream_t s, int epilogue, const SpmvArgs& a, const SpmvConfig& cfg);
// Will launch_spmv hand this plain-CSR product to the row-tile kernel?  (the only CSR kernel that can scale x per gather: SpmvArgs::xScaled)
inline bool spmv_takes_rowtile(const SpmvArgs& a, const SpmvConfig& cfg)
{
    int kernel = cfg.kernel;
    if (kernel == 0) kernel = spmv_auto_kernel(a.rowCount > 0 ? (double)a.elementsCount / (double)a.rowCount : 0.0);
    return kernel == 10 && (((uintptr_t)a.elements & 15) == 0) && (((uintptr_t)a.columnIndeces & 15) == 0) && a.elementsCount >= 8 && a.rowCount > 0;
}
// Row-tile kernel for short rows on plain CSR (kernels_rowtile.hip); periodRows = distance of the far band in rows (0: unknown),
// gridReq = wavefronts (0: 8 per CU).  Needs 16-byte aligned elements / columnIndeces and elementsCount >= 8.
int launch_spmv_rowtile(hipStream_t s, int epilogue, const SpmvArgs& a, int periodRows, int gridReq, int maxRow = 0, bool ntWindow = false,
                        int gapAtTile = 0, int gapSkipTiles = 0);
// Distance (in rows) of the farthest band of the matrix, read off one row in the middle of the slice, and the longest row (one pass
// over the row offsets; both remembered per handle -- stale values can only cost speed: any period gives the same results, and rows
// longer than the remembered maximum take the kernel's slow path); the caller's hint (MgcgSetSpmvPeriod) wins for the period.
int spmv_period(MgcgSparse* h, const int* rowOffsets, const int* columnIndeces, long long rows, long long rowBase, int* maxRow = nullptr, long long* meanDistance = nullptr);
void launch_matrix_shape(hipStream_t s, const int* rowOffsets, const int* columnIndeces, long long rows, long long row, long long rowBase, int* out2);
// The same on the dictionary-compressed form of the matrix (a.elements / a.columnIndeces still serve array tails and long rows).
int launch_spmv_rows(hipStream_t s, int epilogue, const SpmvArgs& a, const DcsrView* m, int gridReq, int periodRows = 0);   // m == nullptr: plain CSR; periodRows: z sweep of the row-pattern kernel
bool dcsr_build(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                long long rows, long long nnz, long long rowBase, DcsrMatrix* out);
// Row-pattern form: usable (out->patternId != nullptr) when the matrix has <= 256 distinct rows-as-sequences.
bool pattern_build(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                   long long rows, long long nnz, long long rowBase, DcsrMatrix* out);
// Column-tiled form: usable (out->tileVals != nullptr) for sorted rows whose entries lie far from the diagonal when x
// (columns doubles) does not fit a few L2s.
bool tiled_build(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                 long long rows, long long nnz, long long rowBase, long long columns, DcsrMatrix* out);
int launch_spmv_tiled(hipStream_t s, int epilogue, const SpmvArgs& a, const DcsrView& m);
// Cached analysis of a matrix on a handle (nullptr: compression off, not applicable, or the build failed).
const DcsrMatrix* dcsr_lookup(MgcgSparse* h, const double* elements, const int* rowOffsets, const int* columnIndeces,
                              long long rows, long long nnz, long long rowBase, long long columns = 0,    // columns: length of x (0: unknown, no column tiling)
                              long long autoMeanDistance = -1,    // >= 0: a Solve-family call; with compression off the library may still build column tiles for a matrix whose entries lie this far (on average) from the diagonal
                              bool trustRegistry = false);        // the three arrays are library-owned vectors: every write the library can see marks the form stale, so no per-call checksum
// Per-op calls (CsrMV, CsrMVDot, Solve0, Solve1: the reference's own phase driver multiplies through these, Mgcg.cu:10-19,116-163): the form
// the caller asked for (MgcgSetMatrixCompression), or -- compression off -- the library's own column tiles for a matrix without locality
// once it has been multiplied a few times through this handle, provided the three arrays are library-owned vectors (Create_*): writes
// through any export mark the form stale (analysis_note_write), so it is trusted without a checksum per call.  Writes by the caller's
// own kernels through ToRawPtr_* pointers are outside the library's view: every 16th product served from the form re-verifies the 64-bit
// checksum of the CSR arrays (0.8 ms per 310 M nonzeros: 2 % of the 16 products), so such a write is noticed within 16 products at the
// latest; a caller that rewrites matrices that way says so with MgcgAnalysisClear, or switches the feature off (MGCG_AUTO_TILES=0).
const DcsrMatrix* dcsr_lookup_op(MgcgSparse* h, const SpmvArgs& a, long long rowBase);
bool vector_owned(const void* p, size_t bytes);   // [p, p + bytes) lies inside a device vector the library allocated
void vector_registry_add(const void* p, size_t bytes);
void vector_registry_remove(const void* p);
unsigned long long csr_checksum(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces, long long rows, long long nnz, unsigned long long* scratch);
// Every export that writes device memory (or frees it) names the range here: analyses made from arrays it overlaps go stale
// and are rebuilt at their next use (the reference re-uploads A into the same vectors on every Initialize():
// Mgcg/cuBlas/Mgcg/ConjugateGradientSingleGpu.cs:134-147).  Costs one relaxed atomic load when nothing is analysed.
void analysis_note_write(const void* p, size_t bytes);
// CSR or compressed, whichever the handle has for this matrix.
int launch_spmv_auto(hipStream_t s, int epilogue, const SpmvArgs& a, const SpmvConfig& cfg, const DcsrMatrix* dc);
// Rows [r0, r1) of the same matrix (a describes the WHOLE local matrix; y, w, b, dinv are shifted here); partials for
// dot epilogues go to `partials`, at most maxGrid of them.
int launch_spmv_range(hipStream_t s, int epilogue, const SpmvArgs& a, const SpmvConfig& cfg, const DcsrMatrix* dc,
                      long long r0, long long r1, double* partials, int maxGrid);

// Rows [0, i0) and [i1, rowCount) -- the boundary rows either side of an interior range -- in ONE launch when the row-tile kernel serves
// the matrix and both cuts fall on its tile boundaries, else in two; partials as launch_spmv_range.
int launch_spmv_two_ranges(hipStream_t s, int epilogue, const SpmvArgs& a, const SpmvConfig& cfg, const DcsrMatrix* dc,
                           long long i0, long long i1, double* partials, int maxGrid);

// ---------------------------------------------------------------- BLAS-1 and fused CG updates
void launch_axpy(hipStream_t s, double* y, const double* x, long long n, double alpha);
void launch_scal(hipStream_t s, double* x, double alpha, long long n);
void launch_xpay(hipStream_t s, double* y, const double* x, long long n, double beta);
void launch_copy(hipStream_t s, double* y, const double* x, long long n);
void launch_fill(hipStream_t s, double* y, double v, long long n);
// knob dot_order: every dot product in the reference's order (one serial sum; the launch_* functions below then return 1 partial)
void launch_dot_serial(hipStream_t s, const double* x, const double* y, long long n, double* out, const int* done);   // out[0] = ((x0 y0 + x1 y1) + x2 y2) + ...
// partials[0..grid) = per-workgroup partial of sum x_i*y_i ; returns grid
int  launch_dot_partials(hipStream_t s, const double* x, const double* y, long long n, double* partials);
int  launch_nrminf_partials(hipStream_t s, const double* x, long long n, double* partials);
// out[0] = sum(partials[0..n)) in a fixed order (mode 0) or max (mode 1)
void launch_reduce(hipStream_t s, const double* partials, int n, double* out, int mode);

// p = r ; partial r.r      (CG init)
int  launch_copy_dot(hipStream_t s, double* p, const double* r, long long n, double* partials, const int* done);
// alpha = sc->rr / sc->pAp ; x += alpha p ; r -= alpha Ap ; partial r.r (and max|r| in partials2 when wantInf)
int  launch_update_xr(hipStream_t s, const CgScalars* sc, double* x, double* r, const double* p, const double* Ap,
                      long long n, double* partials, double* partialsInf);
// p = z + beta p with beta = sc->beta ; skipped when sc->done
void launch_update_p(hipStream_t s, const CgScalars* sc, double* p, const double* z, long long n);
// the loop's own split: r -= alpha Ap (+ r.r), then x += alpha p and p = z + beta p in one pass over p
// pApPartials != nullptr: p.Ap is still in nPAp per-workgroup partial sums; every workgroup adds them up itself (same fixed
// order everywhere) instead of a separate reduction launch.  partials (output) must not overlap pApPartials.
int  launch_update_r(hipStream_t s, CgScalars* sc, double* r, const double* Ap, long long n, double* partials, double* partialsInf,
                     const double* pApPartials = nullptr, int nPAp = 0, bool freeze = false);
void launch_update_xp(hipStream_t s, const CgScalars* sc, double* x, double* p, const double* z, long long n);
// x/p update that first does what finalize_kernel does (every workgroup reduces the r.r partial sums itself and takes the same stop
// decision; workgroup 0 publishes it): one launch fewer per iteration.  Needs update_r launched with freeze = true.
struct FinalizeArgs;
void launch_update_xp_final(hipStream_t s, const FinalizeArgs& f, const double* partials, const double* partialsInf, int nPartials,
                            double* x, double* p, const double* z, long long n);

struct FinalizeArgs {
    CgScalars* sc;
    HostMirror* mirror;
    double* trace; int traceCap;
    double tol; int minIt; int maxIt; int rule;
    int preconditioned;     // 1: beta = rzNew/rr(rz) computed by finalize_precond instead; 2: by this kernel from the all-reduced sc->rzNew
};
// Single-workgroup kernels that turn partial sums into the loop's scalars.
void launch_reduce_to(hipStream_t s, const double* partials, int n, double* dst, const int* done);          // dst = sum
void launch_reduce2_to(hipStream_t s, const double* pA, int nA, double* dstA, const double* pB, int nB, double* dstB, const int* done);   // two sums, one launch
void launch_finalize(hipStream_t s, const double* partials, const double* partialsInf, int n, bool reduceFirst, const FinalizeArgs& f);
void launch_finalize_precond(hipStream_t s, const double* partials, int n, bool reduceFirst, CgScalars* sc);  // rzNew -> beta, rr
void launch_init_scalars(hipStream_t s, const double* partials, int n, bool reduceFirst, CgScalars* sc, HostMirror* mirror, int rule);

// ---------------------------------------------------------------- multigrid kernels
// dinvUniform: dinv[i] == dinvScalar for every i (the array is then not read)
void launch_jacobi_first(hipStream_t s, long long n, double omega, const double* dinv, int dinvUniform, double dinvScalar, const double* b, double* x, const int* done);
// *flag = 1 if some v[i] differs (bitwise) from v[0]; flag pre-set to 0 by the caller
void launch_uniform_check(hipStream_t s, const double* v, long long n, int* flag);
void launch_restrict(hipStream_t s, int nx, int ny, int nz, const double* r, double* bc, const int* done);
void launch_prolong_add(hipStream_t s, int nx, int ny, int nz, double* x, const double* e, const int* done);
// x[i] = outer * (inner * b[i]) + e[parent(i)]   (prolong_add onto a first Jacobi sweep that was never stored)
void launch_prolong_linear_add(hipStream_t s, int nx, int ny, int nz, int z0, int z1, double* x, const double* eFull, const int* done);
void launch_restrict_linear(hipStream_t s, int nx, int ny, int nz, int z0, int z1, const double* rFull, double* bc, const int* done);
void launch_prolong_scaled(hipStream_t s, int nx, int ny, int nz, double* x, const double* b, double inner, double outer, const double* e, const int* done);
void launch_extract_dinv(hipStream_t s, const double* elements, const int* rowOffsets, const int* columnIndeces,
                         long long n, long long rowBase, double* dinv);
// Galerkin: count pass (elementsC == nullptr) writes per-row counts to countsC[I]; fill pass writes entries.
void launch_galerkin(hipStream_t s, int nx, int ny, int nz, int zBegin, int zEnd, const double* elements, const int* rowOffsets, const int* columnIndeces,
                     double sigma, const int* rowOffsetsC, int* countsC, double* elementsC, int* columnIndecesC, int* errFlag);
void launch_poisson(hipStream_t s, int nx, int ny, int nz, int zBegin, int zEnd, double* elements, int* rowOffsets, int* columnIndeces);
void launch_rebase(hipStream_t s, int* rowOffsets, long long n, int base);
long long poisson_nnz_host(int nx, int ny, int nz, int zBegin, int zEnd);
void launch_minmax_int(hipStream_t s, const int* v, long long n, int* out2 /* device int[2] */);
void launch_halo_rows(hipStream_t s, const int* rowOffsets, const int* columnIndeces, long long n, long long offset, int* out2 /* {0, n} */);

// ---------------------------------------------------------------- RCCL (dlopen'ed)
struct CommImpl;
bool comm_allreduce_sum(MgcgComm* c, double* devPtr, int count, hipStream_t s);
bool comm_multi(const MgcgComm* c);   // several ranks, or one rank forced onto the several-ranks path (force_multirank knob)
struct HaloPlan;  // per-peer contiguous send/recv ranges of p
// columnIndeces / nnz (device; may be null): when the slice is unstructured (the contiguous ranges come to a quarter of the vector or
// more) the plan is rebuilt from the column ids actually referenced -- per-peer index lists, packed and unpacked around the exchange.
// reuse: the plan of a solve's own vector p -- kept on the communicator and handed out again while every rank calls with the same partition
// (one 8-byte all-reduce decides); such a plan is not freed by halo_plan_destroy.  Plans of multigrid levels are never shared.
HaloPlan* halo_plan_create(MgcgComm* c, long long count, long long offset, long long countLocal, int minJ, int maxJ,
                           const int* columnIndeces = nullptr, long long nnz = 0, bool reuse = false, bool localOk = true);
// localOk (reuse plans only, several ranks): this rank's verdict on its own arguments, folded into the plan's one all-reduce -- when any
// rank says false, every rank gets nullptr and nobody enters the solve's collectives.
bool comm_agree(MgcgComm* c, bool localOk, const char* who);
bool comm_all(MgcgComm* c, bool mine, bool* all);                 // do all ranks say yes?  (a "no" is not an error)
// set-up only (collective): host vectors to and from ranks rank - 1 and rank + 1
bool comm_neighbour_exchange_host(MgcgComm* c, const std::vector<double>& toLower, const std::vector<double>& toUpper,
                                  std::vector<double>& fromLower, std::vector<double>& fromUpper, bool localOk = true);   // the same agreement as a call of its own (set-up paths)
void halo_last(long long out[3]);   // calling thread's last exchange: {index lists used, entries received, entries the contiguous plan receives}
void halo_plan_destroy(HaloPlan* h);
bool halo_exchange(MgcgComm* c, HaloPlan* h, double* p, hipStream_t s);
// interior rows on a side stream while the halo travels on the main stream (all RCCL calls stay on the main stream)
bool halo_overlap_available(MgcgComm* c);
hipStream_t halo_overlap_fork(MgcgComm* c, hipStream_t mainStream);   // side stream, ordered after everything enqueued on mainStream so far
bool halo_overlap_join(MgcgComm* c, hipStream_t mainStream);          // mainStream waits for the side stream
// the measured overlap rule (collective; once per plan): *pays = this plan's exchange in line costs more than the hops that would hide it
bool halo_overlap_pays(MgcgComm* c, HaloPlan* h, double* vec, hipStream_t s, bool* pays);
void halo_overlap_clear_times();
void halo_overlap_last_times(double out[3]);   // calling thread's last decision: {measured?, exchange in line (us), fork + launch + join (us)}

} // namespace mgcg
