// Solver half of the C ABI: the whole CG / preconditioned-CG loop, resident on the device.
// Replaces Mgcg/cuBlas/MgcgGpu/Mgcg.cu:201-270 (Solve) and the host-driven multi-device loop of
// Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:424-565.
//
// Per iteration the reference issues 7 library calls and 2 blocking 8-byte device->host reads.
// Here an iteration is 3 streaming kernels + 2 single-workgroup scalar kernels, alpha / beta /
// the stop test live in device memory, and the host only enqueues: it runs `checkEvery` iterations
// ahead and looks at a pinned flag the device wrote.  Once the stop test fires on the device every
// later kernel exits at its first instruction, so x, r, p, Iteration and Residual are exactly those
// of the iteration the reference would have stopped at.
#include "common.hpp"
#include <chrono>

namespace mgcg {

// ---------------------------------------------------------------- multigrid hierarchy
// Every level lives on the rank's z-slab [z0, z1) of that level's grid: local CSR rows with GLOBAL column ids,
// local rhs / residual / D^-1, and two FULL-length iterate buffers (like p in the CG loop) whose halo planes are
// exchanged before every SpMV-shaped pass.  A single rank is the special case z0 = 0, z1 = nz.
struct MgLevel {
    int nx = 0, ny = 0, nz = 0;            // global grid of the level
    int z0 = 0, z1 = 0;                    // this rank's planes
    long long n = 0, nnz = 0;              // local rows / nonzeros
    long long nGlobal = 0, offset = 0;     // global rows, first local row
    double* elements = nullptr; int* rowOffsets = nullptr; int* columnIndeces = nullptr;
    bool ownsMatrix = false;
    double* dinv = nullptr;
    bool dinvUniform = false; double dinvScalar = 0.0;   // every diagonal equal: the kernels use the scalar and skip the array
    double *xa = nullptr, *xb = nullptr;   // iterate ping-pong, full length (Jacobi is not in place)
    double *b = nullptr, *r = nullptr;     // right-hand side (restricted residual), residual; local
    SpmvConfig cfg;                        // per level: kernel picked from its nnz/row, banded period = nx*ny
    const DcsrMatrix* dcsr = nullptr;      // compressed form (owned by the MgcgSparse handle's cache)
    HaloPlan* halo = nullptr;              // multi-rank: planes of the iterate owned by the neighbours
    HaloPlan* transferHalo = nullptr;      // multi-rank, linear transfer: exactly one grid plane from each z-neighbour
    int minJ = 0, maxJ = -1;
    bool overlap = false;                  // interior rows [interior0, interior1) are multiplied while the halo travels
    long long interior0 = 0, interior1 = 0;
    // Deep-halo cycle (several ranks, levels >= 1; mg_deep_level below): the right-hand side lives in a FULL-length buffer whose `deep`
    // planes either side of the slab arrive in ONE exchange per cycle (bHalo), and the level's matrix rows, right-hand side and D^-1 for
    // those planes are here too (copied from the neighbours at set-up), so that the sweeps can be recomputed on them instead of exchanged.
    int deep = 0;                          // planes either side (0: the level takes one exchange per sweep)
    int extZ0 = 0, extZ1 = 0;              // planes [extZ0, extZ1) of the extended slab (clipped to the grid)
    double* bFull = nullptr;               // b = bFull + offset
    HaloPlan* bHalo = nullptr;
    double* extElements = nullptr; int* extRowOffsets = nullptr; int* extColumnIndeces = nullptr; double* extDinv = nullptr;
    long long extRows = 0, extNnz = 0, extBase = 0;   // rows of the extended slab, its nonzeros, global index of its first row
};

} // namespace mgcg

struct MgcgMg {
    int levels = 0;
    double omega = 0; int nu = 1, nuCoarse = 4; double sigma = 0.5;
    int interp = 0;                        // 0: piecewise-constant P (slab-local), 1: cell-centred linear P (MgSetInterpolation)
    std::vector<mgcg::MgLevel> lv;
    mgcg::SpmvConfig cfg;
    hipStream_t stream = nullptr;
    MgcgComm* comm = nullptr;              // not owned
    int nranks = 1;
    bool multi = false;                    // several ranks (or one rank forced onto that path, comm_multi): full-length iterates, halo exchanges, all-reduces
    bool haloOnSide = false;               // overlap schedule (tuning knob halo_stream, resolved at set-up)
    // r.z of the PCG loop rides on the V-cycle's last sweep (single rank): partial sums go here, fusedDotCount of them
    double* fuseDotPartials = nullptr;
    int fusedDotCount = 0;
    // several ranks: the last sweep of the cycle writes the rank's rows of z = M^-1 r straight into the caller's vector (nobody needs the
    // halo of the final iterate) instead of into a full-length buffer that is then copied out
    double* finalOut = nullptr;
    bool finalWritten = false;
    bool deep = false;                     // every level >= 1 has its deep halo (decided at set-up, the same on every rank)
    bool deep0 = false;                    // ... and the finest level may form its first iterate per gather from the EXCHANGED right-hand side: every rank's
                                           // rows have one and the same diagonal and take the row-tile kernel (agreed at set-up: it decides WHAT the level exchanges)
    // deep-halo cycle, finest level: the PCG loop keeps its residual r (the cycle's right-hand side) in this buffer -- the slab's rows with
    // room for one grid plane either side -- so that ONE exchange brings r's halo planes and every sweep of the level can form its iterate
    // per gather, exactly as the single-rank cycle does (no stored first sweep, no boundary zones, no separate boundary launches)
    double* rExt = nullptr;                // plane + n + plane doubles; the loop's r = rExt + plane
    bool deepFolds = false;                // levels >= 1 form x_1 and x_1 + P e per gather too (uniform diagonals, power-of-two nx and ny)
    bool skipHalo = false;                 // the next SpMV-shaped pass finds its halo planes already in place (deep-halo cycle: formed locally)
};

namespace mgcg {

static SpmvConfig cfg_of(const MgcgSparse* h)
{
    SpmvConfig c; c.kernel = h->kernel; c.rowsPerBlock = h->rowsPerBlock; c.flags = h->flags; c.gridBlocks = h->gridBlocks; c.periodRows = h->periodRows; c.tileRows = h->tileRows; c.tilePlanes = h->tilePlanes;
    return c;
}

// The interior rows of a row slice -- rows [*i0, *i1) reference local columns only -- and whether its halo exchange should hide behind them
// (*active).  The range is found for every slice of several ranks (the multigrid's folded residual pass uses it with or without overlap);
// MGCG_OVERLAP decides *active: 0 never; 2 whenever an interior exists (tests); 1 (default) BY MEASUREMENT -- on slices of at least
// kOverlapMinRows rows per rank on average the plan's own exchange is timed in line against the fork / launch / join round trip of the
// overlap schedule on this communicator (halo_overlap_pays, comm.hip: once per plan, the same answer on every rank), and the exchange is
// hidden only where it costs more than the hops that hide it.  (Round 3 decided by size alone, >= 3 M rows: on the one-rank RCCL
// communicator of a one-GPU box that rule picked the slowest schedule measured, profiles/r3/slab_latency_box_e_final_code.json.)
// halo / vec: the plan of the slice and the full-length vector it exchanges.  nGlobal: rows of all ranks (the same number everywhere,
// so all ranks enter the measurement's collective together or not at all).
// d2: two device ints of the caller's workspace (Workspace::devInts + 6)
constexpr long long kOverlapMinRows = 1000000;
static bool plan_overlap(hipStream_t s, MgcgComm* comm, bool multi, const int* rowOffsets, const int* columnIndeces,
                         long long n, long long offset, bool* active, long long* i0, long long* i1, int* d2,
                         HaloPlan* halo, double* vec, long long nGlobal)
{
    *active = false; *i0 = 0; *i1 = 0;
    halo_overlap_clear_times();                     // MgcgLastOverlapTimes speaks of the last plan: nothing measured yet
    if (!multi) return true;
    const int mode = tuning().overlap.load(std::memory_order_relaxed);
    bool wanted = mode == 2;
    if (mode == 1 && nGlobal >= kOverlapMinRows * (long long)MgcgCommSize(comm)) {
        if (!halo_overlap_pays(comm, halo, vec, s, &wanted)) return false;       // collective
    }
    if (n <= 0) return true;
    if (MgcgCommSize(comm) == 1) {
        // one rank forced onto the several-ranks path (measurement): an artificial split -- the first and last force_multirank rows
        // (rounded to SpMV tiles) play the boundary
        long long w = tuning().forceMultiRank.load(std::memory_order_relaxed);
        w = (w + 255) & ~255LL;
        if (n >= 4 * w && w > 0) { *i0 = w; *i1 = n - w; *active = wanted && halo_overlap_available(comm); }
        return true;
    }
    int h2[2] = { 0, (int)n };
    bool ok = MGCG_HIP(hipMemcpyAsync(d2, h2, sizeof(h2), hipMemcpyHostToDevice, s));
    if (ok) launch_halo_rows(s, rowOffsets, columnIndeces, n, offset, d2);
    ok = ok && MGCG_HIP(hipMemcpyAsync(h2, d2, sizeof(h2), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
    if (!ok) return false;
    const long long lo = h2[0], hi = h2[1] < n ? h2[1] : n;
    if (hi > lo) {
        *i0 = lo; *i1 = hi;
        *active = wanted && (mode == 2 || 2 * (hi - lo) >= n) && halo_overlap_available(comm);
    }
    return true;
}

static bool mg_halo(MgcgMg* mg, MgLevel& L, double* xfull)
{
    if (!mg->multi || mg->skipHalo) return true;
    return halo_exchange(mg->comm, L.halo, xfull, mg->stream);
}

// Halo of the full-length iterate xfull, then the SpMV-shaped pass; interior rows first when the level overlaps.
// partials / nPartials (dot epilogues): where the per-workgroup partial sums go and how many were written.
// aInt describes the pass for the interior rows (they reference local columns only), aBnd for the boundary rows and for a level that
// does not overlap: the same arguments except, for the folded residual pass of several ranks (mg_vcycle), the multiplied vector.
static bool mg_spmv2(MgcgMg* mg, MgLevel& L, int epilogue, const SpmvArgs& aInt, const SpmvArgs& aBnd, double* xfull, double* partials = nullptr, int* nPartials = nullptr,
                     bool splitInLine = false)
{
    hipStream_t s = mg->stream;
    int n = 0;
    if (nPartials) *nPartials = 0;
    const int half = partials ? kMaxPartials / 2 : 0;
    if (!L.overlap || mg->skipHalo) {                  // (nothing travels: nothing to hide)
        if (!mg_halo(mg, L, xfull)) return false;
        if (splitInLine) {                           // two kinds of rows, one stream: interior rows with aInt, the rows either side with aBnd
            n = launch_spmv_range(s, epilogue, aInt, L.cfg, L.dcsr, L.interior0, L.interior1, partials, half);
            n += launch_spmv_two_ranges(s, epilogue, aBnd, L.cfg, L.dcsr, L.interior0, L.interior1, partials ? partials + n : nullptr, half);
        } else {
            SpmvArgs b = aBnd; b.partials = partials;
            n = launch_spmv_auto(s, epilogue, b, L.cfg, L.dcsr);
        }
        if (nPartials) *nPartials = n;
        return true;
    }
    hipStream_t side = halo_overlap_fork(mg->comm, s);
    if (!side) return false;
    if (mg->haloOnSide) {
        // the exchange on the side stream, all rows on the main stream: both cross-stream hops hide behind the interior rows
        if (!halo_exchange(mg->comm, L.halo, xfull, side)) return false;
        n = launch_spmv_range(s, epilogue, aInt, L.cfg, L.dcsr, L.interior0, L.interior1, partials, half);
        if (!halo_overlap_join(mg->comm, s)) return false;
        n += launch_spmv_two_ranges(s, epilogue, aBnd, L.cfg, L.dcsr, L.interior0, L.interior1, partials ? partials + n : nullptr, half);   // the boundary rows either side of the interior
        if (nPartials) *nPartials = n;
        return true;
    }
    n = launch_spmv_range(side, epilogue, aInt, L.cfg, L.dcsr, L.interior0, L.interior1, partials, half);
    if (!mg_halo(mg, L, xfull)) return false;
    n += launch_spmv_two_ranges(s, epilogue, aBnd, L.cfg, L.dcsr, L.interior0, L.interior1, partials ? partials + n : nullptr, half);
    if (nPartials) *nPartials = n;
    return halo_overlap_join(mg->comm, s);
}
static bool mg_spmv(MgcgMg* mg, MgLevel& L, int epilogue, const SpmvArgs& a, double* xfull, double* partials = nullptr, int* nPartials = nullptr)
{
    return mg_spmv2(mg, L, epilogue, a, a, xfull, partials, nPartials);
}

// xout_loc = xin_loc + omega * (dinv * (b - A xin)); xin / xout are full-length buffers
constexpr long long kFoldUpMaxRows = 100000000;     // (bench.py FOLD_UP_MAX_ROWS counts the bytes with the same limit)
static thread_local int t_lastFolds = 0;        // MgcgLastVcycleFolds: bit 0 first sweep folded, bit 1 prolongation folded (last V-cycle of this thread)

static int log2_exact(long long v) { int l = 0; while ((1LL << l) < v) ++l; return ((1LL << l) == v) ? l : -1; }

// coarse != nullptr (one rank, V(1,1), uniform diagonal, row-tile kernel on plain CSR, power-of-two nx and ny): the sweep runs on
// x1 + P e without that iterate ever being stored -- x1 = omega (d0 b) and the parent's correction are formed per gather and for the
// row itself (SpmvArgs::xScaled == 2); xin is not read.
// Several ranks (coarse = the FULL-length coarse iterate, interiorOnly): only the interior rows form it on the fly (their columns are
// local: (b - offset)[col]); the boundary rows multiply the stored iterate, which the caller has written within two planes of the
// rank's boundaries and mg_spmv2 exchanges.
// globalB (deep-halo cycle, b has its halo planes): ALL rows of the rank form the iterate per gather -- b is addressed by global column ids
// ((b - offset)[col]) and coarse is the full-length coarse result; one launch, as on a single rank.
static bool mg_jacobi(MgcgMg* mg, MgLevel& L, const double* b, double* xin, double* xout, const int* done, bool withDot = false, const double* coarse = nullptr,
                      bool interiorOnly = false, bool globalB = false)
{
    SpmvArgs a{};
    a.elements = L.elements; a.rowOffsets = L.rowOffsets; a.columnIndeces = L.columnIndeces; a.x = xin; a.y = xout + L.offset;
    a.elementsCount = (int)L.nnz; a.rowCount = (int)L.n; a.columnCount = (int)L.nGlobal;
    a.w = xin + L.offset; a.b = b; a.dinv = L.dinv; a.dinvUniform = L.dinvUniform ? 1 : 0; a.dinvScalar = L.dinvScalar; a.omega = mg->omega; a.doneFlag = done;
    SpmvArgs stored = a;                            // (the sweep on a stored iterate: what the boundary rows of several ranks run)
    if (coarse != nullptr) {
        const int lx = log2_exact(L.nx), ly = log2_exact(L.ny);
        a.x = b - ((interiorOnly || globalB) ? L.offset : 0); a.w = nullptr; a.xScaled = 2; a.xInner = L.dinvScalar; a.xOuter = mg->omega; a.xCoarse = coarse;
        const int sy = L.ny > 1 ? 1 : 0, sz = L.nz > 1 ? 1 : 0;
        a.cM0 = L.nx / 2 - 1; a.cS1 = 1 + sy; a.cM1 = ((L.ny >> sy) - 1) << (lx - 1);
        a.cS2 = 1 + sy + sz; a.cM2 = (int)(~0u << (lx - 1 + ly - sy));
        a.cRowBase = (interiorOnly || globalB) ? (int)L.offset : 0;
    }
    if (interiorOnly) {
        if (withDot && mg->finalOut != nullptr) { a.y = mg->finalOut; stored.y = mg->finalOut; mg->finalWritten = true; }
        const bool dot = withDot && mg->fuseDotPartials != nullptr;
        return mg_spmv2(mg, L, dot ? EPI_JACOBI_DOT : EPI_JACOBI, a, stored, xin, dot ? mg->fuseDotPartials : nullptr, dot ? &mg->fusedDotCount : nullptr, true);
    }
    // withDot marks the LAST sweep of the cycle on the finest level: r.z of the PCG loop rides on it (partial sums of b . xout), and
    // with several ranks it writes the rank's rows of the result where the caller wants them
    if (withDot && mg->multi && mg->finalOut != nullptr) { a.y = mg->finalOut; mg->finalWritten = true; }
    if (withDot && mg->fuseDotPartials != nullptr) {
        if (!mg->multi) {
            a.partials = mg->fuseDotPartials;
            mg->fusedDotCount = launch_spmv_auto(mg->stream, EPI_JACOBI_DOT, a, L.cfg, L.dcsr);
            return true;
        }
        return mg_spmv(mg, L, EPI_JACOBI_DOT, a, xin, mg->fuseDotPartials, &mg->fusedDotCount);
    }
    return mg_spmv(mg, L, EPI_JACOBI, a, xin);
}

// `sweeps` Jacobi sweeps on level L for right-hand side b.  first: the first sweep starts from zero.
// cur is the buffer holding the iterate (ignored when first); *result receives the buffer holding the result.
static bool mg_smooth(MgcgMg* mg, MgLevel& L, const double* b, double* cur, double* other, int sweeps, bool first, const int* done, double** result,
                      bool dotOnLast = false)
{
    for (int sIdx = 0; sIdx < sweeps; ++sIdx) {
        if (first && sIdx == 0) {
            launch_jacobi_first(mg->stream, L.n, mg->omega, L.dinv, L.dinvUniform ? 1 : 0, L.dinvScalar, b, cur + L.offset, done);
        } else {
            if (!mg_jacobi(mg, L, b, cur, other, done, dotOnLast && sIdx == sweeps - 1)) return false;
            double* t = cur; cur = other; other = t;
        }
    }
    *result = cur;
    return true;
}

// ---------------------------------------------------------------- deep-halo cycle (several ranks, V(1,1), piecewise-constant transfer)
// The per-sweep schedule above exchanges a halo plane before EVERY SpMV-shaped pass: on 3 levels with 4 coarse sweeps that is seven exchanges
// per cycle (two on the finest level, two on the middle one, three on the coarsest), each a latency-bound grouped send/recv -- where the
// reference's loop has ONE exchange per iteration (SyncP, ConjugateGradientParallelGpu.cs:384-419,469).  Here every level >= 1 takes ONE
// exchange per cycle: `deep` planes of its RIGHT-HAND SIDE either side of the slab, after which everything the level computes within reach
// of its boundaries is recomputed locally on those planes -- the same operations on the same values in the same order, so every entry is the
// bit the owning rank computes (tests: z = M^-1 r equal to the single-domain oracle bit for bit).  Counting planes of the level:
//   coarsest level, nu_c sweeps, result wanted one plane beyond the slab (the finer level prolongs it onto ITS halo planes):
//     x_1 = omega D^-1 b on +-nu_c planes, sweep k on +-(nu_c + 1 - k), x_{nu_c} on +-1          -> deep = nu_c
//   a middle level: x_1 = omega D^-1 b on +-2, residual on the slab, y = x_1 + P e on +-2 (e from below on +-1 of ITS planes = +-2 of these),
//     post-smoothing sweep on +-1                                                              -> deep = 2
//   the finest level has one exchange too: of its right-hand side when that is the loop's r in MgcgMg::rExt and every rank holds the same
//     uniform diagonal (MgcgMg::deep0, agreed at set-up: x_1 and x_1 + P e are then formed per gather on all rows), else of the stored x_1
//     before the residual pass -- the post-smoothing sweep then finds x_1 in the halo planes and P e is added there (mg_deep_prolong_halo);
//     either way                                                                                       -> no second exchange.
// Per MGCG iteration: SyncP of p + 3 exchanges in the cycle instead of SyncP + 7.  The redundant rows are deep planes on slabs of 32 / 16
// planes of levels that hold 1/8 and 1/64 of the work.
static bool mg_deep_level(MgcgMg* mg, int l, const int* done, double** result)
{
    MgLevel& L = mg->lv[l];
    hipStream_t s = mg->stream;
    const long long plane = (long long)L.nx * L.ny;
    const bool coarsest = l == mg->levels - 1;
    auto zlo = [&](int d) { const int z = L.z0 - d; return z < L.extZ0 ? L.extZ0 : z; };
    auto zhi = [&](int d) { const int z = L.z1 + d; return z > L.extZ1 ? L.extZ1 : z; };
    // rows of planes [za, zb) of the extended slab
    auto first = [&](int d, double* x) {            // x = omega D^-1 b on +-d planes
        const long long r0 = (long long)zlo(d) * plane, r1 = (long long)zhi(d) * plane;
        launch_jacobi_first(s, r1 - r0, mg->omega, L.extDinv + (r0 - L.extBase), 0, 0.0, L.bFull + r0, x + r0, done);
    };
    auto sweep = [&](int d, double* xin, double* xout) {   // xout = xin + omega D^-1 (b - A xin) on +-d planes
        const long long r0 = (long long)zlo(d) * plane - L.extBase, r1 = (long long)zhi(d) * plane - L.extBase;
        SpmvArgs a{};
        a.elements = L.extElements; a.rowOffsets = L.extRowOffsets; a.columnIndeces = L.extColumnIndeces; a.x = xin; a.y = xout + L.extBase;
        a.elementsCount = (int)L.extNnz; a.rowCount = (int)L.extRows; a.columnCount = (int)L.nGlobal;
        a.w = xin + L.extBase; a.b = L.bFull + L.extBase; a.dinv = L.extDinv; a.dinvUniform = 0; a.omega = mg->omega; a.doneFlag = done;
        (void)launch_spmv_range(s, EPI_JACOBI, a, L.cfg, nullptr, r0, r1, nullptr, 0);
    };
    if (!halo_exchange(mg->comm, L.bHalo, L.bFull, s)) return false;                  // the level's ONE exchange
    if (coarsest) {
        double* cur = L.xa; double* other = L.xb;
        first(mg->nuCoarse, cur);
        for (int k = 2; k <= mg->nuCoarse; ++k) { sweep(mg->nuCoarse + 1 - k, cur, other); double* t = cur; cur = other; other = t; }
        *result = cur;                                                               // valid on the slab and one plane either side
        return true;
    }
    MgLevel& C = mg->lv[l + 1];
    const bool folded = mg->deepFolds && (L.dcsr == nullptr || !L.dcsr->usable) &&   // (a compact form of the level's own matrix keeps the stored iterates)
                        tuning().noFold.load(std::memory_order_relaxed) == 0 && tuning().foldUp.load(std::memory_order_relaxed) != 0;
    if (!folded) first(2, L.xa);
    SpmvArgs a{};                                                                    // r = b - A x_1 on the slab (the level's own matrix)
    a.elements = L.elements; a.rowOffsets = L.rowOffsets; a.columnIndeces = L.columnIndeces; a.x = L.xa; a.y = L.r;
    a.elementsCount = (int)L.nnz; a.rowCount = (int)L.n; a.columnCount = (int)L.nGlobal; a.b = L.b; a.doneFlag = done;
    if (folded) { a.x = L.bFull; a.xScaled = 1; a.xInner = L.dinvScalar; a.xOuter = mg->omega; }   // x_1[col] = omega (d0 b[col]) per gather: b is full length
    (void)launch_spmv_auto(s, EPI_RESIDUAL, a, L.cfg, L.dcsr);
    launch_restrict(s, L.nx, L.ny, L.z1 - L.z0, L.r, C.b, done);                      // b_c = P^T r (slab-local)
    double* e = nullptr;
    if (!mg_deep_level(mg, l + 1, done, &e)) return false;
    if (folded) {
        // post-smoothing sweep on +-1 plane with x_1 + P e formed per gather (and for the row itself): b on +-2 and e on +-1 coarse plane are here
        const long long r0 = (long long)zlo(1) * plane - L.extBase, r1 = (long long)zhi(1) * plane - L.extBase;
        SpmvArgs f{};
        f.elements = L.extElements; f.rowOffsets = L.extRowOffsets; f.columnIndeces = L.extColumnIndeces; f.x = L.bFull; f.y = L.xb + L.extBase;
        f.elementsCount = (int)L.extNnz; f.rowCount = (int)L.extRows; f.columnCount = (int)L.nGlobal;
        f.w = nullptr; f.b = L.bFull + L.extBase; f.dinv = L.extDinv; f.dinvUniform = 1; f.dinvScalar = L.dinvScalar; f.omega = mg->omega; f.doneFlag = done;
        f.xScaled = 2; f.xInner = L.dinvScalar; f.xOuter = mg->omega; f.xCoarse = e;
        const int lx = log2_exact(L.nx), ly = log2_exact(L.ny), sy = L.ny > 1 ? 1 : 0, sz = L.nz > 1 ? 1 : 0;
        f.cM0 = L.nx / 2 - 1; f.cS1 = 1 + sy; f.cM1 = ((L.ny >> sy) - 1) << (lx - 1);
        f.cS2 = 1 + sy + sz; f.cM2 = (int)(~0u << (lx - 1 + ly - sy));
        f.cRowBase = (int)L.extBase;
        (void)launch_spmv_range(s, EPI_JACOBI, f, L.cfg, nullptr, r0, r1, nullptr, 0);
        *result = L.xb;
        return true;
    }
    const int za = zlo(2), zb = zhi(2);                                              // (even planes: the slab is aligned and 2 is even)
    launch_prolong_add(s, L.nx, L.ny, zb - za, L.xa + (long long)za * plane, e + (long long)(za / 2) * C.nx * C.ny, done);   // y = x_1 + P e on +-2
    sweep(1, L.xa, L.xb);
    *result = L.xb;                                                                  // valid on the slab and one plane either side
    return true;
}

// the finest level's halo planes of `cur` hold x_1 (from the exchange before the residual pass): add P e there, so that the post-smoothing
// sweep needs no exchange.  e: the coarse result, valid one coarse plane beyond the slab.
static void mg_deep_prolong_halo(MgcgMg* mg, MgLevel& L, MgLevel& C, double* cur, const double* e, const int* done)
{
    const long long plane = (long long)L.nx * L.ny, cplane = (long long)C.nx * C.ny;
    if (L.z0 > 0) launch_prolong_add(mg->stream, L.nx, L.ny, 1, cur + (long long)(L.z0 - 1) * plane, e + (long long)((L.z0 - 1) / 2) * cplane, done);
    if (L.z1 < L.nz) launch_prolong_add(mg->stream, L.nx, L.ny, 1, cur + (long long)L.z1 * plane, e + (long long)(L.z1 / 2) * cplane, done);
}

// One V(nu,nu) cycle on level l for right-hand side b (local); x0/x1 are that level's two full-length iterate
// buffers.  *result receives the buffer that holds the answer.  Mirrors vcycle() of oracle/mg_oracle.c.
static bool mg_vcycle(MgcgMg* mg, int l, const double* b, double* x0, double* x1, const int* done, double** result)
{
    MgLevel& L = mg->lv[l];
    if (l == mg->levels - 1) return mg_smooth(mg, L, b, x0, x1, mg->nuCoarse, true, done, result, l == 0);
    MgLevel& C = mg->lv[l + 1];
    double* cur = nullptr;
    // V(1,*) on one rank with a uniform diagonal: the first sweep x1 = omega (d0 b) is not stored; the residual pass forms x1[col] per
    // gather (row-pattern form, or the row-tile kernel on plain CSR) and the prolongation forms x1[i] again when it adds the correction
    const bool linear = mg->interp == 1;
    bool canScale = false;
    if (L.dcsr != nullptr && L.dcsr->usable) canScale = L.dcsr->patternId != nullptr;
    else {
        SpmvArgs probe{}; probe.elements = L.elements; probe.columnIndeces = L.columnIndeces; probe.elementsCount = (int)L.nnz; probe.rowCount = (int)L.n;
        canScale = spmv_takes_rowtile(probe, L.cfg);
    }
    const bool mayFold = mg->nu == 1 && !linear && L.dinvUniform && canScale && tuning().noFold.load(std::memory_order_relaxed) == 0;
    const bool fold = mayFold && !mg->multi;
    // Deep-halo cycle with the right-hand side in the loop's extended buffer (b = rExt + plane: one grid plane of room either side): ONE
    // exchange brings b's halo planes, after which every row of the rank -- boundary rows included -- forms x_1 per gather of the residual
    // pass and x_1 + P e per gather of the post-smoothing sweep: the single-rank cycle's kernels, one launch each, no zones.
    if (l == 0 && mg->deep && mg->deep0 && mg->rExt != nullptr && b == mg->rExt + (long long)L.nx * L.ny && mayFold && (L.dcsr == nullptr || !L.dcsr->usable) &&
        L.nx >= 2 && log2_exact(L.nx) >= 1 && log2_exact(L.ny) >= 0 && L.nGlobal < 0x7fffffffLL && tuning().foldUp.load(std::memory_order_relaxed) != 0) {
        SpmvArgs a{};
        a.elements = L.elements; a.rowOffsets = L.rowOffsets; a.columnIndeces = L.columnIndeces; a.x = b - L.offset; a.y = L.r;
        a.elementsCount = (int)L.nnz; a.rowCount = (int)L.n; a.columnCount = (int)L.nGlobal; a.b = b; a.doneFlag = done;
        a.xScaled = 1; a.xInner = L.dinvScalar; a.xOuter = mg->omega;
        if (!mg_spmv(mg, L, EPI_RESIDUAL, a, const_cast<double*>(b) - L.offset)) return false;     // the level's one exchange (b's halo planes), then r = b - A x_1
        launch_restrict(mg->stream, L.nx, L.ny, L.z1 - L.z0, L.r, C.b, done);
        double* e = nullptr;
        if (!mg_deep_level(mg, 1, done, &e)) return false;
        t_lastFolds |= 1 | 2 | 4 | 8;
        struct SkipHalo { MgcgMg* m; ~SkipHalo() { m->skipHalo = false; } } skip{ mg };
        mg->skipHalo = true;
        if (!mg_jacobi(mg, L, b, x0, x1, done, true, e, false, true)) return false;              // the last sweep (+ r.z), straight into the caller's z
        *result = x1;
        return true;
    }
    // Several ranks: the same fold for the INTERIOR rows (they reference local columns only: x1[col] formed per gather
    // from the local right-hand side), while the boundary rows multiply the stored iterate, which therefore exists only where they reach --
    // the local rows within two grid planes of a boundary (every level is a 27-point-neighbourhood operator: MgSetup's Galerkin pass has
    // checked it) -- and, after the exchange, in the halo planes.
    const long long plane = (long long)L.nx * L.ny;
    const long long zoneLo = L.interior0 > 0 ? L.interior0 + 2 * plane : 0;                 // x1 stored for local rows [0, zoneLo) ...
    const long long zoneHi = L.interior1 < L.n ? L.interior1 - 2 * plane : L.n;             // ... and [zoneHi, n)
    // (worth two more launches only on a large level: 16 bytes per row saved against ~15 us of launches and a boundary pass of its own)
    const bool foldInterior = mayFold && mg->multi && (L.dcsr == nullptr || !L.dcsr->usable) && zoneLo <= zoneHi && 2 * (L.interior1 - L.interior0) >= L.n &&
                              (L.n >= 3000000 || tuning().overlap.load(std::memory_order_relaxed) == 2);
    if (fold || foldInterior) t_lastFolds |= 1;
    if (fold) cur = x0;
    else if (foldInterior) {
        cur = x0;
        if (zoneLo > 0) launch_jacobi_first(mg->stream, zoneLo, mg->omega, L.dinv, 1, L.dinvScalar, b, cur + L.offset, done);
        if (zoneHi < L.n) launch_jacobi_first(mg->stream, L.n - zoneHi, mg->omega, L.dinv, 1, L.dinvScalar, b + zoneHi, cur + L.offset + zoneHi, done);
    }
    else if (!mg_smooth(mg, L, b, x0, x1, mg->nu, true, done, &cur)) return false;
    double* other = (cur == x0) ? x1 : x0;
    SpmvArgs a{};
    // linear transfer on several ranks: the restriction reads one plane of r from each z-neighbour, so r goes to the spare
    // full-length iterate buffer (dead until the post-smoothing writes it) and that plane is exchanged
    const bool rFullLength = linear && mg->multi;
    a.elements = L.elements; a.rowOffsets = L.rowOffsets; a.columnIndeces = L.columnIndeces; a.x = fold ? b : cur; a.y = rFullLength ? other + L.offset : L.r;
    a.elementsCount = (int)L.nnz; a.rowCount = (int)L.n; a.columnCount = (int)L.nGlobal; a.b = b; a.doneFlag = done;
    if (fold) { a.xScaled = 1; a.xInner = L.dinvScalar; a.xOuter = mg->omega; }
    if (foldInterior) {
        SpmvArgs ai = a;
        ai.x = b - L.offset;                         // global column ids of the interior rows are local: (b - offset)[col] = b[col - offset]
        ai.xScaled = 1; ai.xInner = L.dinvScalar; ai.xOuter = mg->omega;
        if (!mg_spmv2(mg, L, EPI_RESIDUAL, ai, a, cur, nullptr, nullptr, true)) return false;   // r = b - A x1, interior rows from b, boundary rows from the exchanged iterate
    }
    else if (!mg_spmv(mg, L, EPI_RESIDUAL, a, cur)) return false;                             // r = b - A x
    if (linear) {
        if (rFullLength && !halo_exchange(mg->comm, L.transferHalo, other, mg->stream)) return false;
        launch_restrict_linear(mg->stream, L.nx, L.ny, L.nz, L.z0, L.z1, rFullLength ? other : L.r, C.b, done);   // b_c = P^T r
    } else {
        launch_restrict(mg->stream, L.nx, L.ny, L.z1 - L.z0, L.r, C.b, done);                 // b_c = P^T r (slab-local)
    }
    double* e = nullptr;
    const bool deep = mg->deep && !linear && l == 0;                                      // (levels >= 1 of a deep-halo cycle never come through here)
    if (deep) { if (!mg_deep_level(mg, 1, done, &e)) return false; t_lastFolds |= 4; }
    else if (!mg_vcycle(mg, l + 1, C.b, C.xa, C.xb, done, &e)) return false;
    // deep-halo cycle: the post-smoothing sweep's halo planes are formed here (x_1 is in them since the residual pass's exchange, P e is
    // added by mg_deep_prolong_halo), so the sweep itself exchanges nothing
    struct SkipHalo { MgcgMg* m; bool on; ~SkipHalo() { if (on) m->skipHalo = false; } } skip{ mg, deep };
    if (linear) {
        if (mg->multi && !halo_exchange(mg->comm, C.transferHalo, e, mg->stream)) return false;
        launch_prolong_linear_add(mg->stream, L.nx, L.ny, L.nz, L.z0, L.z1, cur + L.offset, e, done);               // x += P e
    }
    else if (fold && (L.dcsr == nullptr || !L.dcsr->usable) && L.nx >= 2 && log2_exact(L.nx) >= 1 && log2_exact(L.ny) >= 0 && L.nGlobal < 0x7fffffffLL &&
             (tuning().foldUp.load(std::memory_order_relaxed) < 0 ? L.n <= kFoldUpMaxRows : tuning().foldUp.load(std::memory_order_relaxed) != 0)) {
        // V(1,1) on plain CSR: the prolongation is folded into the one post-smoothing sweep as well -- x1 + P e is formed per gather there
        // (no prolongation kernel, the iterate buffer `cur` is never written: 33 N bytes less on this level).  The second gather per entry
        // costs the sweep more than the bytes save on the largest levels (profiles/r3/foldup_ab.log, the A/B script is in the history: 256^3 as the
        // finest level -4.5 % per MGCG iteration, 512 x 512 x 256 -1.6 %, as level 1 of 512^3 -56 us; on the 512^3 level itself +0.13 ms): by level size.
        if (!mg_jacobi(mg, L, b, cur, other, done, l == 0, e + C.offset)) return false;
        t_lastFolds |= 2;
        *result = other;
        return true;
    }
    else if (foldInterior && L.nx >= 2 && log2_exact(L.nx) >= 1 && log2_exact(L.ny) >= 0 && L.nGlobal < 0x7fffffffLL && L.nz > 1 && (L.z0 & 1) == 0 && ((L.z1 - L.z0) & 1) == 0 &&
             (tuning().foldUp.load(std::memory_order_relaxed) < 0 ? L.n <= kFoldUpMaxRows : tuning().foldUp.load(std::memory_order_relaxed) != 0)) {
        // several ranks: the same for the interior rows; x1 + P e is stored only where the boundary rows reach (the zones of the first fold, the
        // upper one begun on an even plane so that its parents start on a whole coarse plane), and the halo planes arrive by the exchange
        const long long planes = L.z1 - L.z0, coarsePlane = (long long)C.nx * C.ny;
        const long long loPlanes = zoneLo / plane;
        long long hiStart = zoneHi / plane;
        if (hiStart & 1) --hiStart;
        if (hiStart < loPlanes) {                                                   // the zones meet: the whole slab in one piece
            launch_prolong_scaled(mg->stream, L.nx, L.ny, (int)planes, cur + L.offset, b, L.dinvScalar, mg->omega, e + C.offset, done);
        } else {
            if (loPlanes > 0) launch_prolong_scaled(mg->stream, L.nx, L.ny, (int)loPlanes, cur + L.offset, b, L.dinvScalar, mg->omega, e + C.offset, done);
            if (zoneHi < L.n) {
                const long long r0 = hiStart * plane;
                launch_prolong_scaled(mg->stream, L.nx, L.ny, (int)(planes - hiStart), cur + L.offset + r0, b + r0, L.dinvScalar, mg->omega,
                                      e + C.offset + (hiStart / 2) * coarsePlane, done);
            }
        }
        if (deep) { mg_deep_prolong_halo(mg, L, C, cur, e, done); mg->skipHalo = true; }
        if (!mg_jacobi(mg, L, b, cur, other, done, l == 0, e, true)) return false;
        t_lastFolds |= 2;
        *result = other;
        return true;
    }
    else if (fold || foldInterior) launch_prolong_scaled(mg->stream, L.nx, L.ny, L.z1 - L.z0, cur + L.offset, b, L.dinvScalar, mg->omega, e + C.offset, done);
    else launch_prolong_add(mg->stream, L.nx, L.ny, L.z1 - L.z0, cur + L.offset, e + C.offset, done);   // x += P e (slab-local)
    if (deep) { mg_deep_prolong_halo(mg, L, C, cur, e, done); mg->skipHalo = true; }
    return mg_smooth(mg, L, b, cur, other, mg->nu, false, done, result, l == 0);
}

// z = M^-1 r (both local).  A single rank ping-pongs level 0 between z itself and lv[0].xa so that the result
// lands in z; with several ranks the iterates must be full length, so the result is copied out of lv[0].xa/xb.
// dotPartials (optional, single rank): the last sweep also leaves *nDot partial sums of r . z there (0: not fused,
// the caller runs the dot product itself).
static bool mg_apply(MgcgMg* mg, const double* r, double* z, const int* done, double* dotPartials = nullptr, int* nDot = nullptr)
{
    mg->fuseDotPartials = dotPartials; mg->fusedDotCount = 0;
    t_lastFolds = 0;
    struct Reset { MgcgMg* m; int* n; ~Reset() { if (n) *n = m->fusedDotCount; m->fuseDotPartials = nullptr; } } reset{ mg, nDot };
    MgLevel& L0 = mg->lv[0];
    double* res = nullptr;
    if (!mg->multi) {
        // swaps on level 0: (nu-1) pre + nu post, or (nuCoarse-1) when there is a single level
        const int swaps = (mg->levels == 1) ? (mg->nuCoarse - 1) : (2 * mg->nu - 1);
        double* start = (swaps % 2 == 0) ? z : L0.xa;
        double* other = (start == z) ? L0.xa : z;
        if (!mg_vcycle(mg, 0, r, start, other, done, &res)) return false;
        if (res != z) launch_copy(mg->stream, z, res, L0.n);       // not reached for the buffer choice above
        return true;
    }
    mg->finalOut = z; mg->finalWritten = false;
    const bool ok = mg_vcycle(mg, 0, r, L0.xa, L0.xb, done, &res);
    mg->finalOut = nullptr;
    if (!ok) return false;
    if (!mg->finalWritten) launch_copy(mg->stream, z, res + L0.offset, L0.n);   // (a cycle whose last step is not a Jacobi sweep: one level, one sweep)
    return true;
}

// ---------------------------------------------------------------- the CG loop
struct CgRun {
    Workspace* ws = nullptr;
    SpmvConfig cfg;
    MgcgComm* comm = nullptr;
    HaloPlan* halo = nullptr;
    MgcgMg* mg = nullptr;
    const double* elements = nullptr; const int* rowOffsets = nullptr; const int* columnIndeces = nullptr;
    int elementsCount = 0;
    double *x = nullptr, *b = nullptr, *Ap = nullptr, *p = nullptr, *r = nullptr, *z = nullptr;  // p is FULL length (count), the rest local
    long long count = 0, nLocal = 0, offset = 0;
    int nranks = 1;
    bool multi = false;                    // comm_multi(comm)
    double tol = 0; int minIt = 0, maxIt = 0, rule = 0;
    bool wantInf = false;
    SpmvProfile* prof = nullptr;
    const DcsrMatrix* dcsr = nullptr;      // compressed form of the matrix if the handle has one
    MgcgSparse* cusparse = nullptr;
    // rows [interior0, interior1) reference local columns only: they are multiplied (side stream) while the halo of p is in flight
    bool overlap = false;
    long long interior0 = 0, interior1 = 0;
    bool haloOnSide = false;               // tuning knob, resolved once per solve (every iteration of every rank takes the same path)
    Vector* pVec = nullptr;                // the handles behind p and Ap (the placement draw may move their data)
    Vector* ApVec = nullptr;
};

static thread_local long long t_lastOverlap[3] = { 0, 0, 0 };

static bool cg_plan_overlap(CgRun& R)
{
    t_lastOverlap[0] = 0; t_lastOverlap[1] = 0; t_lastOverlap[2] = 0;
    if (!plan_overlap(R.ws->stream, R.comm, R.multi, R.rowOffsets, R.columnIndeces, R.nLocal, R.offset, &R.overlap, &R.interior0, &R.interior1, R.ws->devInts + 6, R.halo, R.p, R.count)) return false;
    if (R.overlap) { t_lastOverlap[0] = 1; t_lastOverlap[1] = R.interior0; t_lastOverlap[2] = R.interior1; }
    return true;
}

static void prof_mark(CgRun& R, bool begin)
{
    SpmvProfile* p = R.prof;
    if (!p || !p->enabled) return;
    if (begin) {
        if (p->used >= (int)p->start.size()) {
            hipEvent_t a = nullptr, b = nullptr;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
            p->start.push_back(a); p->stop.push_back(b);
        }
        (void)hipEventRecord(p->start[p->used], R.ws->stream);
    } else if (p->used < (int)p->start.size()) {
        (void)hipEventRecord(p->stop[p->used], R.ws->stream);
        p->used++;
    }
}

__global__ void snapshot_kernel(const CgScalars* sc, HostMirror* m, volatile int* slot)
{
    *slot = sc->done;
    m->residual = sc->residual;
}

__global__ void clear_done_kernel(CgScalars* sc) { sc->done = 0; sc->status = 0; }

// ---------------------------------------------------------------- placement draw for Ap and p
// The same SpMV binary on the same matrix runs up to 17 % apart depending on WHERE the runtime placed its two vectors (identical HBM
// traffic and L2 hit rates: the timing of the same requests -- the channel / bank hash of the physical pages).  Round 4's probe
// (tools/placement_probe2.py, profiles/r4/placement_probe2.log: every pair of 4 x and 4 y allocations in one process) shows the WRITTEN
// vector dominating -- one y allocation is good or bad with every x (2.23-2.28 against 2.48-2.62 ms) -- and the gathered one adding a
// few per cent.  A kernel cannot steer that, but the library owns every vector (Create_Double): at the first solve on vectors of at
// least kPlacementMinEntries entries (256 MB: beyond the Infinity Cache) it allocates `placement` more buffers for Ap, times the loop's
// own SpMV (fused with p.Ap) on each and keeps the fastest, then does the same for p: one draw becomes the best of k, twice.
// One-off cost at 512^3: 2 x 4 candidates x 7 launches x 2.4 ms + 3 GiB of copies ~ 0.14 s, inside the first solve (the role cuSPARSE's
// csrmv analysis plays in the reference's stack).  Nothing numerical changes: the same doubles at another address.  A vector whose
// address the caller has seen (ToRawPtr_Double) is never moved.
constexpr long long kPlacementMinEntries = 32LL << 20;
static thread_local double t_placementMs[2][16];
static thread_local int t_placementInfo[2][2] = { { 0, -1 }, { 0, -1 } };     // per stage (0: Ap, 1: p): candidates timed, chosen

static void placement_report(int stage, const Vector* v)
{
    t_placementInfo[stage][0] = v ? v->drawCount : 0; t_placementInfo[stage][1] = v ? v->drawChosen : -1;
    for (int i = 0; v && i < v->drawCount && i < 16; ++i) t_placementMs[stage][i] = v->drawMs[i];
}

// one stage: candidates for vector v (stage 0: the SpMV's output Ap, whose contents do not matter at this point; stage 1: its input p)
static void placement_stage(CgRun& R, int stage, Vector* v)
{
    placement_report(stage, v);                    // MgcgLastPlacement speaks of THIS solve's vectors: the record of their one draw, or nothing
    const int extra = tuning().placement.load(std::memory_order_relaxed);
    double*& mine = stage == 0 ? R.Ap : R.p;
    if (!v || v->placed || v->rawExported || extra <= 0 || v->size < kPlacementMinEntries || v->data != mine || R.nLocal < 4096 || R.elementsCount < 8) return;
    v->placed = true;                              // one draw per vector, whatever comes of it
    hipStream_t s = R.ws->stream;
    const size_t bytes = sizeof(double) * (size_t)v->size;
    size_t freeB = 0, totalB = 0;
    if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { (void)hipGetLastError(); return; }
    double* cand[16];
    int n = 1;
    cand[0] = v->data;
    const int want = extra > 15 ? 15 : extra;
    for (int i = 0; i < want && freeB > (size_t)(i + 1) * bytes + (2ULL << 30); ++i) {
        double* q = nullptr;
        if (hipMalloc((void**)&q, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        cand[n++] = q;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool ok = n > 1 && MGCG_HIP(hipEventCreate(&e0)) && MGCG_HIP(hipEventCreate(&e1));
    for (int i = 1; ok && i < n; ++i) ok = MGCG_HIP(hipMemcpyAsync(cand[i], v->data, bytes, hipMemcpyDeviceToDevice, s));   // the same contents everywhere
    int best = 0;
    constexpr int kReps = 6;
    double ms[16];
    for (int i = 0; ok && i < n; ++i) {
        SpmvArgs a{};
        a.elements = R.elements; a.rowOffsets = R.rowOffsets; a.columnIndeces = R.columnIndeces;
        a.x = stage == 1 ? cand[i] : R.p; a.y = stage == 0 ? cand[i] : R.Ap;
        a.elementsCount = R.elementsCount; a.rowCount = (int)R.nLocal; a.columnCount = (int)R.count;
        a.w = a.x + R.offset; a.partials = R.ws->partials; a.doneFlag = nullptr;
        (void)launch_spmv_auto(s, EPI_DOT, a, R.cfg, R.dcsr);                              // warm-up (Ap and the partial sums are rewritten by the solve)
        ok = MGCG_HIP(hipEventRecord(e0, s));
        for (int k = 0; k < kReps; ++k) (void)launch_spmv_auto(s, EPI_DOT, a, R.cfg, R.dcsr);
        ok = ok && MGCG_HIP(hipEventRecord(e1, s)) && MGCG_HIP(hipEventSynchronize(e1));
        float t = 0.0f;
        ok = ok && MGCG_HIP(hipEventElapsedTime(&t, e0, e1));
        ms[i] = (double)t / kReps;
        if (ok && ms[i] < ms[best]) best = i;
    }
    (void)hipStreamSynchronize(s);
    if (!ok) { best = 0; (void)hipGetLastError(); }
    for (int i = 0; i < n; ++i) if (i != best) { analysis_note_write(cand[i], bytes); if (i == 0) vector_registry_remove(cand[0]); (void)hipFree(cand[i]); }   // (freed addresses may be handed out again)
    if (best != 0) { v->data = cand[best]; mine = v->data; vector_registry_add(v->data, bytes); }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (ok) { v->drawCount = n; v->drawChosen = best; for (int i = 0; i < n; ++i) v->drawMs[i] = (float)ms[i]; placement_report(stage, v); }
    if (ok && tuning().verbose.load(std::memory_order_relaxed) >= 1) {
        fprintf(stderr, "[MgcgGpu] placement draw for %s (%lld entries): SpMV", stage == 0 ? "Ap" : "p", v->size);
        for (int i = 0; i < n; ++i) fprintf(stderr, " %.3f", ms[i]);
        fprintf(stderr, " ms -> candidate %d\n", best);
    }
}

// Is the draw worth its price for THIS call?  It costs about 2 stages x 3 candidates x 7 launches = 42 SpMV times (+ 3 GiB of copies at 512^3:
// 0.14 s in all) and wins about 4-5 % of one SpMV per iteration (profiles/r4/placement_ab_*.log): it pays from about a thousand iterations.
// The preconditioned loop converges in a few hundred (config 3: 157, where the draw recovered 28 ms of its 140: ADVICE r4) -- no draw there;
// the plain loop draws when its iteration cap leaves room for that many (the 512^3 system needs 1225), and CgSteps -- the fixed-length
// form a caller uses to time the steady state -- always does (steps = 0 in R.maxIt).
constexpr int kPlacementMinIterations = 1000;
static void placement_draw(CgRun& R, bool fixedSteps)
{
    if (R.mg != nullptr || (!fixedSteps && R.maxIt < kPlacementMinIterations)) { placement_report(0, nullptr); placement_report(1, nullptr); return; }
    placement_stage(R, 0, R.ApVec);                // the written vector first: it decides the most
    placement_stage(R, 1, R.pVec);
}

static bool cg_enqueue_init(CgRun& R, bool fixedSteps = false)
{
    hipStream_t s = R.ws->stream;
    long long meanDistance = 0;
    // MGCG_VERBOSE=2: what a solve pays before its first iteration is enqueued (the reference's driver times a cold Solve(): MgcgMain.cs:121-126)
    const bool report = tuning().verbose.load(std::memory_order_relaxed) >= 2;
    const auto t0 = std::chrono::steady_clock::now();
    auto since = [&](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - a).count(); };
    if (R.cusparse && R.elementsCount >= 8) R.cfg.periodRows = spmv_period(R.cusparse, R.rowOffsets, R.columnIndeces, R.nLocal, R.offset, &R.cfg.maxRow, &meanDistance);
    const double usShape = since(t0);
    const auto t1 = std::chrono::steady_clock::now();
    if (R.cusparse) R.dcsr = dcsr_lookup(R.cusparse, R.elements, R.rowOffsets, R.columnIndeces, R.nLocal, R.elementsCount, R.offset, R.count, R.mg ? -1 : meanDistance);
    const double usForm = since(t1);
    if (meanDistance >= (1LL << 19) && R.count >= (8LL << 19)) R.cfg.flags |= 16;      // gathers without locality: the stream form among the CSR kernels
    if (!R.mg) R.cfg.flags |= 8;                 // plain CG loop: the row-tile kernel may read the matrix with the non-temporal hint (kernels_rowtile.hip)
    const auto t2 = std::chrono::steady_clock::now();
    placement_draw(R, fixedSteps);               // (before p is touched: may move p's data once per vector)
    if (report) fprintf(stderr, "[MgcgGpu] solve set-up: matrix shape %.0f us, matrix form %.0f us, placement draw %.0f us\n", usShape, usForm, since(t2));
    CgScalars* sc = R.ws->scalars;
    double* pLoc = R.p + R.offset;
    if (R.rule == MGCG_RULE_SIMPLE) launch_fill(s, R.x, 0.0, R.nLocal);             // SimpleConjugateGradient.cu:53
    launch_copy(s, pLoc, R.x, R.nLocal);                                             // p_loc = x  (Mgcg.cu:80)
    if (!halo_exchange(R.comm, R.halo, R.p, s)) return false;                        // SyncP  (ConjugateGradientParallelGpu.cs:427)
    SpmvArgs a{};
    a.elements = R.elements; a.rowOffsets = R.rowOffsets; a.columnIndeces = R.columnIndeces; a.x = R.p; a.y = R.r;
    a.elementsCount = R.elementsCount; a.rowCount = (int)R.nLocal; a.columnCount = (int)R.count; a.b = R.b;
    launch_spmv_auto(s, EPI_RESIDUAL, a, R.cfg, R.dcsr);                             // r = b - A x   (Mgcg.cu:225-226)
    int n;
    if (R.mg) {
        int nz = 0;
        if (!mg_apply(R.mg, R.r, R.z, nullptr, dot_reference_order() ? nullptr : R.ws->partials, &nz)) return false;   // z = M^-1 r (+ r.z)
        launch_copy(s, pLoc, R.z, R.nLocal);                                         // p = z
        n = nz > 0 ? nz : launch_dot_partials(s, R.r, R.z, R.nLocal, R.ws->partials);   // rz = r.z
    } else {
        n = launch_copy_dot(s, pLoc, R.r, R.nLocal, R.ws->partials, nullptr);        // p = r ; rr = r.r  (Mgcg.cu:227-228)
    }
    if (R.multi) {
        launch_reduce_to(s, R.ws->partials, n, &sc->rr, nullptr);
        if (!comm_allreduce_sum(R.comm, &sc->rr, 1, s)) return false;                // resultsDot.Sum()  (:463)
        launch_init_scalars(s, R.ws->partials, n, false, sc, R.ws->mirror, R.rule);
    } else {
        launch_init_scalars(s, R.ws->partials, n, true, sc, R.ws->mirror, R.rule);
    }
    return MGCG_HIP(hipGetLastError());
}

static bool cg_enqueue_iteration(CgRun& R, bool withStopTest)
{
    hipStream_t s = R.ws->stream;
    CgScalars* sc = R.ws->scalars;
    double* pLoc = R.p + R.offset;
    const int* done = &sc->done;
    SpmvArgs a{};
    a.elements = R.elements; a.rowOffsets = R.rowOffsets; a.columnIndeces = R.columnIndeces; a.x = R.p; a.y = R.Ap;
    a.elementsCount = R.elementsCount; a.rowCount = (int)R.nLocal; a.columnCount = (int)R.count;
    a.w = pLoc; a.partials = R.ws->partials; a.doneFlag = done;
    int n;
    if (R.overlap) {
        // interior rows on the side stream while SyncP (:469) travels on the main stream, then the boundary rows
        prof_mark(R, true);
        hipStream_t side = halo_overlap_fork(R.comm, s);
        if (!side) return false;
        if (R.haloOnSide) {
            // SyncP (:469) on the side stream, all rows on the main stream: the two cross-stream hops (10 us each, measured) and the wire
            // time of the planes hide behind the interior rows; the join finds the exchange long finished
            if (!halo_exchange(R.comm, R.halo, R.p, side)) return false;
            n = launch_spmv_range(s, EPI_DOT, a, R.cfg, R.dcsr, R.interior0, R.interior1, R.ws->partials, kMaxPartials / 2);
            if (!halo_overlap_join(R.comm, s)) return false;
            n += launch_spmv_two_ranges(s, EPI_DOT, a, R.cfg, R.dcsr, R.interior0, R.interior1, R.ws->partials + n, kMaxPartials / 2);   // boundary rows: one launch when the cuts fall on tiles
        } else {
            n = launch_spmv_range(side, EPI_DOT, a, R.cfg, R.dcsr, R.interior0, R.interior1, R.ws->partials, kMaxPartials / 2);
            if (!halo_exchange(R.comm, R.halo, R.p, s)) return false;
            n += launch_spmv_two_ranges(s, EPI_DOT, a, R.cfg, R.dcsr, R.interior0, R.interior1, R.ws->partials + n, kMaxPartials / 2);
            if (!halo_overlap_join(R.comm, s)) return false;
        }
        prof_mark(R, false);
    } else {
        if (!halo_exchange(R.comm, R.halo, R.p, s)) return false;                    // SyncP  (:469)
        prof_mark(R, true);
        n = launch_spmv_auto(s, EPI_DOT, a, R.cfg, R.dcsr);                          // Ap = A p ; p.Ap   (Mgcg.cu:244-245)
        prof_mark(R, false);
    }
    const bool refDots = dot_reference_order();                                      // validation mode: p.Ap once more, in the reference's order
    if (refDots) { launch_dot_serial(s, pLoc, R.Ap, R.nLocal, R.ws->partials, done); n = 1; }
    double* pInf = R.wantInf ? R.ws->partials + kMaxPartials : nullptr;
    double* rrPartials = R.ws->partials;
    // one rank, no preconditioner: the x/p update finalises the iteration itself (one launch fewer)
    const bool fold = !R.multi && !R.mg && R.nLocal > 0;
    const bool foldRanks = R.multi && !R.mg && R.nLocal > 0;                    // several ranks: the same fold behind the all-reduce of r.r
    if (R.multi) {
        launch_reduce_to(s, R.ws->partials, n, &sc->pAp, done);
        if (!comm_allreduce_sum(R.comm, &sc->pAp, 1, s)) return false;               // (:499)
        if (R.mg) rrPartials = R.ws->partials + 2 * kMaxPartials;                    // (they must outlive the V-cycle, whose r.z partial sums take the first region)
        n = launch_update_r(s, sc, R.r, R.Ap, R.nLocal, rrPartials, pInf, nullptr, 0, foldRanks);   // r -= a Ap ; r.r  (:247-248); x += a p rides with the p update below
    } else {
        // one rank: the workgroups of the r update add the p.Ap partial sums themselves (one launch fewer per iteration);
        // their own r.r partial sums go to the third region of the buffer
        rrPartials = R.ws->partials + 2 * kMaxPartials;
        n = launch_update_r(s, sc, R.r, R.Ap, R.nLocal, rrPartials, pInf, R.ws->partials, n, fold);
    }
    FinalizeArgs f{};
    f.sc = sc; f.mirror = R.ws->mirror; f.trace = R.ws->trace; f.traceCap = R.ws->traceCap;
    f.tol = R.tol; f.minIt = R.minIt; f.maxIt = R.maxIt; f.rule = R.rule; f.preconditioned = R.mg ? 1 : 0;
    if (!withStopTest) { f.tol = -1.0; f.minIt = 0; f.maxIt = 0x7fffffff; f.rule = MGCG_RULE_NATIVE; }   // never converges
    if (fold) {
        launch_update_xp_final(s, f, rrPartials, pInf, n, R.x, pLoc, R.r, R.nLocal);     // residual, stop test, beta (:251-266) ; x += a p (:246) ; p = r + beta p (:265)
        return MGCG_HIP(hipGetLastError());
    }
    if (R.multi && R.mg) {
        // Preconditioned, several ranks: r.r (stop test) and r.z (beta) travel in ONE all-reduce of two doubles behind the
        // V-cycle ("[r.z, r.r] batched"); the stop decision of an iteration is taken one V-cycle later,
        // which costs one wasted V-cycle at the very end and saves a collective per iteration.
        const int nrr = n;
        int nz = 0;
        if (!mg_apply(R.mg, R.r, R.z, done, refDots ? nullptr : R.ws->partials, &nz)) return false;      // z = M^-1 r (+ partial sums of r.z on the last sweep)
        n = nz > 0 ? nz : launch_dot_partials(s, R.r, R.z, R.nLocal, R.ws->partials);
        launch_reduce2_to(s, rrPartials, nrr, &sc->rrNew, R.ws->partials, n, &sc->rzNew, done);   // local r.r and r.z, one launch
        if (!comm_allreduce_sum(R.comm, &sc->rrNew, 2, s)) return false;             // {rrNew, rzNew} are adjacent in CgScalars  (:525 and the PCG's r.z)
        f.preconditioned = 2;                                                        // finalize also does beta = rzNew / rz, rz = rzNew
        launch_finalize(s, rrPartials, pInf, n, false, f);
        launch_update_xp(s, sc, R.x, pLoc, R.z, R.nLocal);                           // x += a p (:246) ; p = z + beta p
        return MGCG_HIP(hipGetLastError());
    }
    if (R.multi) {
        launch_reduce_to(s, rrPartials, n, &sc->rrNew, done);
        if (!comm_allreduce_sum(R.comm, &sc->rrNew, 1, s)) return false;             // (:525)
        if (foldRanks) {
            launch_update_xp_final(s, f, nullptr, nullptr, 0, R.x, pLoc, R.r, R.nLocal);     // stop test, beta, x += a p, p = r + beta p in one launch
            return MGCG_HIP(hipGetLastError());
        }
        launch_finalize(s, rrPartials, pInf, n, false, f);
    } else {
        launch_finalize(s, rrPartials, pInf, n, true, f);                            // residual, stop test, beta  (:251-266)
    }
    if (R.mg) {
        int nz = 0;
        if (!mg_apply(R.mg, R.r, R.z, done, (!R.multi && !refDots) ? R.ws->partials : nullptr, &nz)) return false;   // z = M^-1 r (+ r.z on the last sweep)
        n = nz > 0 ? nz : launch_dot_partials(s, R.r, R.z, R.nLocal, R.ws->partials);
        if (R.multi) {
            launch_reduce_to(s, R.ws->partials, n, &sc->rzNew, done);
            if (!comm_allreduce_sum(R.comm, &sc->rzNew, 1, s)) return false;
            launch_finalize_precond(s, R.ws->partials, n, false, sc);
        } else {
            launch_finalize_precond(s, R.ws->partials, n, true, sc);                 // beta = rzNew / rz
        }
        launch_update_xp(s, sc, R.x, pLoc, R.z, R.nLocal);                           // x += a p (:246) ; p = z + beta p
    } else {
        launch_update_xp(s, sc, R.x, pLoc, R.r, R.nLocal);                           // x += a p (:246) ; p = r + beta p   (:265)
    }
    return MGCG_HIP(hipGetLastError());
}

static int cg_solve(CgRun& R, int* iteration, double* residual, double* residualTrace, int traceCapacity)
{
    hipStream_t s = R.ws->stream;
    HostMirror* m = R.ws->mirror;
    if (R.rule < MGCG_RULE_NATIVE || R.rule > MGCG_RULE_VIENNACL) { set_error("unknown stop rule %d", R.rule); return MGCG_ERROR; }
    R.wantInf = (R.rule == MGCG_RULE_HANDMADECL);
    if (R.wantInf && R.nranks > 1) { set_error("the max-norm rule is single-rank only"); return MGCG_ERROR; }
    if (residualTrace && traceCapacity > 0) { if (!R.ws->ensure_trace(traceCapacity)) return MGCG_ERROR; }
    const int devTraceCap = (residualTrace && traceCapacity > 0) ? traceCapacity : 0;
    double* savedTrace = R.ws->trace; const int savedCap = R.ws->traceCap;
    if (!devTraceCap) { R.ws->trace = nullptr; R.ws->traceCap = 0; } else R.ws->traceCap = devTraceCap;

    R.haloOnSide = tuning().haloStream.load(std::memory_order_relaxed) != 0;
    int status = MGCG_ERROR;
    int checkEvery = 4;
    { const int v = tuning().checkEvery.load(std::memory_order_relaxed); if (v >= 1) checkEvery = v; }
    const long long hostCap = (long long)(R.maxIt > R.minIt ? R.maxIt : R.minIt) + 4;
    hipEvent_t ev[2] = { nullptr, nullptr };
    volatile int* slots = (volatile int*)&R.ws->hostScalar[2];   // two ints per double: slots[0..3]
    bool ok = MGCG_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming)) && MGCG_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    for (int i = 0; i < 4; ++i) slots[i] = 0;
    const bool report = tuning().verbose.load(std::memory_order_relaxed) >= 2;
    const auto hostT0 = std::chrono::steady_clock::now();
    ok = ok && cg_enqueue_init(R);
    if (report) {
        const double enq = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hostT0).count();
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "[MgcgGpu] solve: initial phase enqueued after %.0f us, finished on the device after %.0f us\n", enq,
                std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hostT0).count());
    }
    long long enqueued = 0;
    int chunk = 0;
    bool finished = false;
    while (ok && !finished) {
        for (int k = 0; ok && k < checkEvery; ++k) ok = cg_enqueue_iteration(R, true);
        enqueued += checkEvery;
        hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(1), 0, s, R.ws->scalars, m, &slots[chunk & 1]);
        ok = ok && MGCG_HIP(hipEventRecord(ev[chunk & 1], s));
        if (chunk > 0) {                                  // look at the chunk BEFORE the one just enqueued
            ok = ok && MGCG_HIP(hipEventSynchronize(ev[(chunk - 1) & 1]));
            if (ok && slots[(chunk - 1) & 1] != 0) finished = true;
        }
        if (!finished && enqueued > hostCap + 2LL * checkEvery) {
            ok = ok && MGCG_HIP(hipStreamSynchronize(s));
            if (ok && slots[chunk & 1] != 0) finished = true;
            else { set_error("CG: the device never raised its stop flag after %lld iterations", enqueued); ok = false; }
        }
        ++chunk;
    }
    ok = MGCG_HIP(hipStreamSynchronize(s)) && ok;
    if (report) fprintf(stderr, "[MgcgGpu] solve: %lld iterations enqueued, loop drained %.0f us after the call began\n", enqueued,
                        std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - hostT0).count());
    if (ok) {
        status = m->status;
        if (iteration) *iteration = m->iteration;
        if (residual) *residual = m->residual;
        if (devTraceCap) {
            int nTrace = m->iteration + 1; if (nTrace > devTraceCap) nTrace = devTraceCap;
            ok = MGCG_HIP(hipMemcpy(residualTrace, R.ws->trace, sizeof(double) * (size_t)nTrace, hipMemcpyDeviceToHost));
        }
        if (status == MGCG_MAXIT_EXCEEDED) set_error("CG did not converge: iteration %d exceeded maxIteration %d (residual %g)", m->iteration, R.maxIt, m->residual);
        if (status == MGCG_NONFINITE) set_error("CG stopped: residual is not finite at iteration %d", m->iteration);
    }
    if (ev[0]) (void)hipEventDestroy(ev[0]);
    if (ev[1]) (void)hipEventDestroy(ev[1]);
    R.ws->trace = savedTrace; R.ws->traceCap = savedCap;
    return ok ? status : MGCG_ERROR;
}

static bool check_vectors(const char* who, Vector* e, VectorInt* ro, VectorInt* ci, Vector* x, Vector* b, Vector* Ap, Vector* p, Vector* r,
                          long long elementsCount, long long nLocal, long long count)
{
    if (!e || !ro || !ci || !x || !b || !Ap || !p || !r) { set_error("%s: null vector handle", who); return false; }
    if (elementsCount < 0 || nLocal < 0 || count < nLocal) { set_error("%s: bad sizes", who); return false; }
    if (e->size < elementsCount || ci->size < elementsCount || ro->size < nLocal + 1 || x->size < nLocal || b->size < nLocal ||
        Ap->size < nLocal || r->size < nLocal || p->size < count) { set_error("%s: a device vector is smaller than the problem", who); return false; }
    return true;
}

void preload_solver() { preload_code_object(reinterpret_cast<const void*>(&snapshot_kernel)); }

} // namespace mgcg

using namespace mgcg;

extern "C" {

int SolveEx(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
            Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
            Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector,
            int elementsCount, int count,
            double allowableResidual, int minIteration, int maxIteration, int rule,
            int* iteration, double* residual, double* residualTrace, int traceCapacity)
{
    (void)matDescr;
    if (!device_state()) return MGCG_ERROR;
    if (!cublas || !cusparse) { set_error("SolveEx: null handle"); return MGCG_ERROR; }
    if (!check_vectors("SolveEx", elementsVector, rowOffsetsVector, columnIndecesVector, xVector, bVector, ApVector, pVector, rVector, elementsCount, count, count)) return MGCG_ERROR;
    CgRun R;
    R.ws = &cublas->ws; R.cfg = cfg_of(cusparse); R.prof = &cusparse->prof; R.cusparse = cusparse;
    R.elements = elementsVector->data; R.rowOffsets = rowOffsetsVector->data; R.columnIndeces = columnIndecesVector->data; R.elementsCount = elementsCount;
    R.x = xVector->data; R.b = bVector->data; R.Ap = ApVector->data; R.p = pVector->data; R.r = rVector->data; R.pVec = pVector; R.ApVec = ApVector;
    R.count = count; R.nLocal = count; R.offset = 0;
    R.tol = allowableResidual; R.minIt = minIteration; R.maxIt = maxIteration; R.rule = rule;
    return cg_solve(R, iteration, residual, residualTrace, traceCapacity);
}

void Solve(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
           Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
           Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector,
           int elementsCount, int count,
           double allowableResidual, int minIteration, int maxIteration,
           int* iteration, double* residual)
{
    int it = 0; double res = NAN;
    const int st = SolveEx(cublas, cusparse, matDescr, elementsVector, rowOffsetsVector, columnIndecesVector, xVector, bVector,
                           ApVector, pVector, rVector, elementsCount, count, allowableResidual, minIteration, maxIteration,
                           MGCG_RULE_NATIVE, &it, &res, nullptr, 0);
    (void)st;
    if (iteration) *iteration = it + 1;     // the reference returns its post-incremented loop counter (Mgcg.cu:234)
    if (residual) *residual = res;
}

int SolveParallel(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
                  Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                  Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector,
                  int count, int countForDevice, int offsetForDevice, int elementsCountForDevice,
                  int minJ, int maxJ,
                  double allowableResidual, int minIteration, int maxIteration, int rule,
                  int* iteration, double* residual, double* residualTrace, int traceCapacity)
{
    (void)matDescr;
    if (!device_state()) return MGCG_ERROR;
    // A rank whose own arguments are unusable must not simply return: its peers would block in the solve's first collective.  The verdict
    // travels in the halo plan's one all-reduce (halo_plan_create, localOk) and every rank leaves with MGCG_ERROR.
    bool pre = true;
    if (!cublas || !cusparse) { set_error("SolveParallel: null handle"); pre = false; }
    pre = pre && check_vectors("SolveParallel", elementsVector, rowOffsetsVector, columnIndecesVector, xVector, bVector, ApVector, pVector, rVector,
                               elementsCountForDevice, countForDevice, count);
    if (pre && (offsetForDevice < 0 || (long long)offsetForDevice + countForDevice > count)) { set_error("SolveParallel: bad partition"); pre = false; }
    if (!pre) {
        if (MgcgCommSize(comm) > 1) (void)halo_plan_create(comm, count, offsetForDevice, countForDevice, minJ, maxJ, nullptr, 0, true, false);
        return MGCG_ERROR;
    }
    CgRun R;
    R.ws = &cublas->ws; R.cfg = cfg_of(cusparse); R.prof = &cusparse->prof; R.cusparse = cusparse; R.comm = comm; R.nranks = MgcgCommSize(comm); R.multi = comm_multi(comm);
    R.elements = elementsVector->data; R.rowOffsets = rowOffsetsVector->data; R.columnIndeces = columnIndecesVector->data; R.elementsCount = elementsCountForDevice;
    R.x = xVector->data; R.b = bVector->data; R.Ap = ApVector->data; R.p = pVector->data; R.r = rVector->data; R.pVec = pVector; R.ApVec = ApVector;
    R.count = count; R.nLocal = countForDevice; R.offset = offsetForDevice;
    R.tol = allowableResidual; R.minIt = minIteration; R.maxIt = maxIteration; R.rule = rule;
    if (R.multi) {
        R.halo = halo_plan_create(comm, count, offsetForDevice, countForDevice, minJ, maxJ, R.columnIndeces, R.elementsCount, true);
        if (!R.halo) return MGCG_ERROR;
        if (!cg_plan_overlap(R)) { halo_plan_destroy(R.halo); return MGCG_ERROR; }
    }
    const int st = cg_solve(R, iteration, residual, residualTrace, traceCapacity);
    if (R.halo) halo_plan_destroy(R.halo);
    return st;
}

double CgSteps(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse,
               Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
               Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector,
               int count, int countForDevice, int offsetForDevice, int elementsCountForDevice,
               int minJ, int maxJ, int steps, int restart)
{
    if (!device_state()) return NAN;
    bool pre = true;
    if (!cublas || !cusparse) { set_error("CgSteps: null handle"); pre = false; }
    pre = pre && check_vectors("CgSteps", elementsVector, rowOffsetsVector, columnIndecesVector, xVector, bVector, ApVector, pVector, rVector,
                               elementsCountForDevice, countForDevice, count);
    if (pre && (offsetForDevice < 0 || (long long)offsetForDevice + countForDevice > count)) { set_error("CgSteps: bad partition"); pre = false; }
    if (!pre) {       // leave together with the peers (see SolveParallel)
        if (MgcgCommSize(comm) > 1) (void)halo_plan_create(comm, count, offsetForDevice, countForDevice, minJ, maxJ, nullptr, 0, true, false);
        return NAN;
    }
    CgRun R;
    R.ws = &cublas->ws; R.cfg = cfg_of(cusparse); R.prof = &cusparse->prof; R.cusparse = cusparse; R.comm = comm; R.nranks = MgcgCommSize(comm); R.multi = comm_multi(comm);
    R.elements = elementsVector->data; R.rowOffsets = rowOffsetsVector->data; R.columnIndeces = columnIndecesVector->data; R.elementsCount = elementsCountForDevice;
    R.x = xVector->data; R.b = bVector->data; R.Ap = ApVector->data; R.p = pVector->data; R.r = rVector->data; R.pVec = pVector; R.ApVec = ApVector;
    R.count = count; R.nLocal = countForDevice; R.offset = offsetForDevice; R.rule = MGCG_RULE_NATIVE;
    double* savedTrace = R.ws->trace; const int savedCap = R.ws->traceCap;
    R.ws->trace = nullptr; R.ws->traceCap = 0;
    bool ok = true;
    if (R.multi) { R.halo = halo_plan_create(comm, count, offsetForDevice, countForDevice, minJ, maxJ, R.columnIndeces, R.elementsCount, true); ok = R.halo != nullptr && cg_plan_overlap(R); }
    long long meanDistance = 0;
    if (R.elementsCount >= 8) R.cfg.periodRows = spmv_period(cusparse, R.rowOffsets, R.columnIndeces, R.nLocal, R.offset, &R.cfg.maxRow, &meanDistance);
    R.dcsr = dcsr_lookup(cusparse, R.elements, R.rowOffsets, R.columnIndeces, R.nLocal, R.elementsCount, R.offset, R.count, meanDistance);
    if (meanDistance >= (1LL << 19) && R.count >= (8LL << 19)) R.cfg.flags |= 16;
    R.cfg.flags |= 8;
    R.haloOnSide = tuning().haloStream.load(std::memory_order_relaxed) != 0;
    if (ok && restart) ok = cg_enqueue_init(R, true);
    else if (ok) hipLaunchKernelGGL(clear_done_kernel, dim3(1), dim3(1), 0, R.ws->stream, R.ws->scalars);
    const bool report = tuning().verbose.load(std::memory_order_relaxed) >= 2;      // MGCG_VERBOSE=2: is the host or the device the limit?
    const auto h0 = std::chrono::steady_clock::now();
    for (int k = 0; ok && k < steps; ++k) ok = cg_enqueue_iteration(R, false);
    if (ok) hipLaunchKernelGGL(snapshot_kernel, dim3(1), dim3(1), 0, R.ws->stream, R.ws->scalars, R.ws->mirror, (volatile int*)&R.ws->hostScalar[2]);
    const auto h1 = std::chrono::steady_clock::now();
    ok = MGCG_HIP(hipStreamSynchronize(R.ws->stream)) && ok;
    if (report && steps > 0) {
        const double enq = std::chrono::duration<double, std::micro>(h1 - h0).count() / steps, all = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count() / steps;
        fprintf(stderr, "[MgcgGpu] CgSteps: host enqueue %.1f us per iteration, enqueue + drain %.1f us per iteration (%d steps)\n", enq, all, steps);
    }
    if (R.halo) halo_plan_destroy(R.halo);
    R.ws->trace = savedTrace; R.ws->traceCap = savedCap;
    return ok ? (double)R.ws->mirror->residual : NAN;
}

int MgcgLastHalo(long long volume[2])
{
    long long v[3]; halo_last(v);
    if (volume) { volume[0] = v[1]; volume[1] = v[2]; }
    return (int)v[0];
}

int MgcgLastVcycleFolds(void) { return t_lastFolds; }

int MgcgLastPlacement(int which, double milliseconds[], int capacity, int* chosen)
{
    if (which < 0 || which > 1) { if (chosen) *chosen = -1; return 0; }
    const int n = t_placementInfo[which][0];
    for (int i = 0; milliseconds && i < n && i < capacity; ++i) milliseconds[i] = t_placementMs[which][i];
    if (chosen) *chosen = t_placementInfo[which][1];
    return n;
}

int MgcgLastOverlapTimes(double microseconds[2])
{
    double t[3]; halo_overlap_last_times(t);
    if (microseconds) { microseconds[0] = t[1]; microseconds[1] = t[2]; }
    return (int)t[0];
}

int MgcgLastOverlap(long long interior[2])
{
    if (interior) { interior[0] = t_lastOverlap[1]; interior[1] = t_lastOverlap[2]; }
    return (int)t_lastOverlap[0];
}

void MgcgProfileSpmv(MgcgSparse* h, int enable)
{
    if (!h) return;
    h->prof.enabled = enable != 0;
    h->prof.used = 0;
}

double MgcgProfileSpmvMs(MgcgSparse* h, int* launches)
{
    if (launches) *launches = 0;
    if (!h) return 0.0;
    std::vector<float> ms((size_t)h->prof.used, 0.0f);
    float longest = 0.0f;
    for (int i = 0; i < h->prof.used; ++i) {
        if (hipEventSynchronize(h->prof.stop[i]) != hipSuccess) continue;
        if (hipEventElapsedTime(&ms[(size_t)i], h->prof.start[i], h->prof.stop[i]) != hipSuccess) ms[(size_t)i] = 0.0f;
        if (ms[(size_t)i] > longest) longest = ms[(size_t)i];
    }
    // launches enqueued after the device raised its stop flag return at their first instruction: not SpMV work
    double total = 0.0;
    for (float t : ms) if (t > 0.05f * longest) { total += t; if (launches) (*launches)++; }
    return total;
}

// ---------------------------------------------------------------- multigrid
MgcgMg* MgSetupParallel(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse,
                        Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                        int elementsCount, int nx, int ny, int nz, int zBegin, int zEnd,
                        int levels, double omega, int nu, int nuCoarse, double sigma)
{
    DeviceState* d = device_state();
    if (!d) return nullptr;
    const long long nGlobal0 = (long long)nx * ny * nz;
    const int nranks = MgcgCommSize(comm);
    const long long nLocal0 = (long long)nx * ny * (zEnd - zBegin);
    bool pre = true;
    if (!cublas || !cusparse || !elementsVector || !rowOffsetsVector || !columnIndecesVector) { set_error("MgSetup: null argument"); pre = false; }
    if (pre && (nx < 1 || ny < 1 || nz < 1 || levels < 1 || nu < 1 || nuCoarse < 1 || nGlobal0 > 0x7fffffffLL || zBegin < 0 || zEnd > nz || zBegin >= zEnd)) { set_error("MgSetup: bad parameters"); pre = false; }
    if (pre && (rowOffsetsVector->size < nLocal0 + 1 || elementsVector->size < elementsCount || columnIndecesVector->size < elementsCount)) { set_error("MgSetup: matrix vectors too small"); pre = false; }
    if (pre && nranks > 1 && (nz % nranks != 0 || (zEnd - zBegin) != nz / nranks || zBegin != MgcgCommRank(comm) * (nz / nranks))) {
        set_error("MgSetup: with several ranks the grid must be split into equal z-slabs in rank order"); pre = false;
    }
    if (!comm_agree(comm, pre, "MgSetup")) return nullptr;       // collective: a rank with unusable arguments leaves together with its peers
    hipStream_t s = d->stream;
    MgcgMg* mg = new MgcgMg();
    mg->omega = omega; mg->nu = nu; mg->nuCoarse = nuCoarse; mg->sigma = sigma; mg->stream = s; mg->cfg = cfg_of(cusparse);
    mg->cfg.kernel = 0;   // every level picks its kernel from its own nnz/row
    mg->comm = comm; mg->nranks = nranks; mg->multi = comm_multi(comm);
    mg->haloOnSide = tuning().haloStream.load(std::memory_order_relaxed) != 0;
    const bool multi = mg->multi;
    int* dErr = nullptr;
    int* dmm = nullptr;
    bool ok = MGCG_HIP(hipMalloc((void**)&dErr, sizeof(int))) && MGCG_HIP(hipMemsetAsync(dErr, 0, sizeof(int), s)) && MGCG_HIP(hipMalloc((void**)&dmm, 2 * sizeof(int)));
    // Galerkin matrix of a level from its finer one (two passes: row lengths, then entries).  Whatever it allocated stays in L for MgDestroy.
    auto coarse_matrix = [&](MgLevel& L, const MgLevel& F) -> bool {
        L.ownsMatrix = true;
        int* counts = nullptr;
        bool good = MGCG_HIP(hipMalloc((void**)&L.rowOffsets, sizeof(int) * (size_t)(L.n + 1))) && MGCG_HIP(hipMalloc((void**)&counts, sizeof(int) * (size_t)(L.n > 0 ? L.n : 1)));
        std::vector<int> h((size_t)L.n + 1);
        int err = 0;
        if (good) {
            launch_galerkin(s, F.nx, F.ny, F.nz, F.z0, F.z1, F.elements, F.rowOffsets, F.columnIndeces, sigma, nullptr, counts, nullptr, nullptr, dErr);
            good = MGCG_HIP(hipMemcpyAsync(h.data() + 1, counts, sizeof(int) * (size_t)L.n, hipMemcpyDeviceToHost, s)) &&
                   MGCG_HIP(hipMemcpyAsync(&err, dErr, sizeof(int), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
        }
        if (counts) (void)hipFree(counts);
        if (!good) return false;
        if (err) { set_error("MgSetup: matrix is not a 27-point-neighbourhood operator on the %dx%dx%d grid", F.nx, F.ny, F.nz); return false; }
        h[0] = 0;
        long long run = 0;
        for (long long i = 1; i <= L.n; ++i) { run += h[(size_t)i]; h[(size_t)i] = (int)run; }   // exclusive scan on the host (set-up only)
        L.nnz = run;
        good = MGCG_HIP(hipMemcpyAsync(L.rowOffsets, h.data(), sizeof(int) * (size_t)(L.n + 1), hipMemcpyHostToDevice, s)) &&
               MGCG_HIP(hipMalloc((void**)&L.elements, sizeof(double) * (size_t)(L.nnz > 0 ? L.nnz : 1))) &&
               MGCG_HIP(hipMalloc((void**)&L.columnIndeces, sizeof(int) * (size_t)(L.nnz > 0 ? L.nnz : 1)));
        if (good) launch_galerkin(s, F.nx, F.ny, F.nz, F.z0, F.z1, F.elements, F.rowOffsets, F.columnIndeces, sigma, L.rowOffsets, nullptr, L.elements, L.columnIndeces, dErr);
        return MGCG_HIP(hipStreamSynchronize(s)) && good;   // (h must outlive the copy)
    };
    bool agreedToLeave = false;
    for (int l = 0; ok && l < levels; ++l) {
        MgLevel L;
        if (l == 0) {
            L.nx = nx; L.ny = ny; L.nz = nz; L.z0 = zBegin; L.z1 = zEnd; L.nnz = elementsCount;
            L.elements = elementsVector->data; L.rowOffsets = rowOffsetsVector->data; L.columnIndeces = columnIndecesVector->data;
        } else {
            const MgLevel& F = mg->lv[l - 1];
            if ((F.nx > 1 && F.nx % 2) || (F.ny > 1 && F.ny % 2) || (F.nz > 1 && F.nz % 2)) break;   // cannot coarsen an odd extent
            if (F.nx == 1 && F.ny == 1 && F.nz == 1) break;
            // the slab must stay aligned: every rank sees the same answer because the slabs are equal
            if (F.nz > 1 && ((F.z0 % 2) || (F.z1 % 2))) break;
            if (nranks > 1 && F.nz == 1) break;
            L.nx = F.nx > 1 ? F.nx / 2 : 1; L.ny = F.ny > 1 ? F.ny / 2 : 1; L.nz = F.nz > 1 ? F.nz / 2 : 1;
            L.z0 = F.nz > 1 ? F.z0 / 2 : 0; L.z1 = F.nz > 1 ? F.z1 / 2 : 1;
        }
        L.nGlobal = (long long)L.nx * L.ny * L.nz;
        L.n = (long long)L.nx * L.ny * (L.z1 - L.z0);
        L.offset = (long long)L.nx * L.ny * L.z0;
        if (l > 0) ok = ok && coarse_matrix(L, mg->lv[l - 1]);
        ok = ok && MGCG_HIP(hipMalloc((void**)&L.dinv, sizeof(double) * (size_t)L.n));
        ok = ok && MGCG_HIP(hipMalloc((void**)&L.xa, sizeof(double) * (size_t)L.nGlobal));
        ok = ok && MGCG_HIP(hipMalloc((void**)&L.r, sizeof(double) * (size_t)L.n));
        if (l > 0 || multi) ok = ok && MGCG_HIP(hipMalloc((void**)&L.xb, sizeof(double) * (size_t)L.nGlobal));
        if (l > 0) ok = ok && MGCG_HIP(hipMalloc((void**)&L.b, sizeof(double) * (size_t)L.n));
        if (ok && multi) {   // halo entries that no neighbour owns (outside the global range) are never read; the rest must start defined
            ok = ok && MGCG_HIP(hipMemsetAsync(L.xa, 0, sizeof(double) * (size_t)L.nGlobal, s)) && MGCG_HIP(hipMemsetAsync(L.xb, 0, sizeof(double) * (size_t)L.nGlobal, s));
        }
        if (ok) launch_extract_dinv(s, L.elements, L.rowOffsets, L.columnIndeces, L.n, L.offset, L.dinv);
        if (ok && L.n > 0) {                              // constant-coefficient operators: one diagonal for all rows
            int differs = 1;
            ok = MGCG_HIP(hipMemsetAsync(dmm, 0, sizeof(int), s));      // (dmm is re-initialised before its min/max use below)
            if (ok) launch_uniform_check(s, L.dinv, L.n, dmm);
            ok = ok && MGCG_HIP(hipMemcpyAsync(&differs, dmm, sizeof(int), hipMemcpyDeviceToHost, s)) &&
                 MGCG_HIP(hipMemcpyAsync(&L.dinvScalar, L.dinv, sizeof(double), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
            L.dinvUniform = ok && differs == 0;
        }
        if (ok) L.dcsr = dcsr_lookup(cusparse, L.elements, L.rowOffsets, L.columnIndeces, L.n, L.nnz, L.offset);
        L.cfg = mg->cfg;
        if (ok && L.nnz >= 8) (void)spmv_period(cusparse, L.rowOffsets, L.columnIndeces, L.n, L.offset, &L.cfg.maxRow);   // longest row of the level (7 on every Galerkin level of a 7-point operator)
        L.cfg.periodRows = (L.nz > 1) ? L.nx * L.ny : 0;       // far band of a 3-D stencil = one grid plane (used only if the caller switched the banded schedule on)
        if (ok && multi) {
            int init[2] = { 0x7fffffff, (int)0x80000000 }, out[2] = { 0, -1 };
            ok = ok && MGCG_HIP(hipMemcpyAsync(dmm, init, sizeof(init), hipMemcpyHostToDevice, s));
            if (ok && L.nnz > 0) launch_minmax_int(s, L.columnIndeces, L.nnz, dmm);
            ok = ok && MGCG_HIP(hipMemcpyAsync(out, dmm, sizeof(out), hipMemcpyDeviceToHost, s));
            ok = ok && MGCG_HIP(hipStreamSynchronize(s));
            L.minJ = out[0]; L.maxJ = out[1];
        }
        if (multi) {
            // the plan and the overlap rule are collective: a rank whose level failed (its slab's matrix is not a stencil operator, an allocation)
            // says so first, and every rank leaves the set-up in the same place
            ok = comm_agree(comm, ok, "MgSetup");
            if (ok) { L.halo = halo_plan_create(comm, L.nGlobal, L.offset, L.n, L.minJ, L.maxJ); ok = comm_agree(comm, L.halo != nullptr, "MgSetup"); }
            agreedToLeave = !ok;                                // (every rank holds the same verdict: no second agreement behind the loop)
            ok = ok && plan_overlap(s, comm, mg->multi, L.rowOffsets, L.columnIndeces, L.n, L.offset, &L.overlap, &L.interior0, &L.interior1, cublas->ws.devInts + 6, L.halo, L.xa, L.nGlobal);
        }
        mg->lv.push_back(L);                                    // (also a level that failed: MgDestroy frees what it had allocated)
        if (!ok) break;
        mg->levels = (int)mg->lv.size();
    }
    ok = ok && MGCG_HIP(hipStreamSynchronize(s));
    if (dErr) (void)hipFree(dErr);
    if (dmm) (void)hipFree(dmm);
    // Deep-halo cycle (mg_deep_level): every level >= 1 gets a full-length right-hand side, the plan that brings `deep` planes of it from
    // the neighbours, and the neighbours' matrix rows for those planes.  Taken when every such level's slab is at least `deep` planes thick
    // (then only the two adjacent ranks are involved); the slabs are equal, so every rank decides the same.  Collective from here on: a
    // local failure is agreed on (comm_agree) before any rank walks into the exchanges.
    bool wantDeep = ok && multi && nu == 1 && mg->levels >= 2 && tuning().deepHalo.load(std::memory_order_relaxed) != 0;
    bool foldsEverywhere = true;
    for (int l = 1; wantDeep && l < mg->levels; ++l) {
        const MgLevel& L = mg->lv[(size_t)l];
        const int d = (l == mg->levels - 1) ? nuCoarse : 2;
        if (L.nz <= 1 || (L.z1 - L.z0) < d || (L.z0 & 1) || ((L.z1 - L.z0) & 1)) wantDeep = false;
    }
    // (a rank that failed outside the agreements above -- before the loop, or in the local tail of a level -- says so here; its peers hear it in
    //  their next agreement, which is this one or the next level's first)
    if (multi && (agreedToLeave || !comm_agree(comm, ok, "MgSetup"))) { MgDestroy(mg); return nullptr; }
    for (int l = 1; ok && wantDeep && l < mg->levels; ++l) {
        MgLevel& L = mg->lv[(size_t)l];
        const long long plane = (long long)L.nx * L.ny;
        const int d = (l == mg->levels - 1) ? nuCoarse : 2;
        L.deep = d;
        L.extZ0 = L.z0 - d < 0 ? 0 : L.z0 - d;
        L.extZ1 = L.z1 + d > L.nz ? L.nz : L.z1 + d;
        L.extBase = (long long)L.extZ0 * plane; L.extRows = (long long)(L.extZ1 - L.extZ0) * plane;
        const long long edge = (long long)d * plane;                                  // rows of d planes
        // my first / last d planes, packed: { rows, nnz, row lengths, values, column ids }
        std::vector<int> ro((size_t)L.n + 1);
        bool lok = MGCG_HIP(hipMemcpyAsync(ro.data(), L.rowOffsets, sizeof(int) * (size_t)(L.n + 1), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
        auto pack = [&](long long r0, long long r1, std::vector<double>& out) {
            out.clear();
            if (!lok || r1 <= r0) return;
            const long long k0 = ro[(size_t)r0], k1 = ro[(size_t)r1], K = k1 - k0, Rn = r1 - r0;
            std::vector<double> v((size_t)K); std::vector<int> c((size_t)K);
            lok = lok && (K == 0 || (MGCG_HIP(hipMemcpyAsync(v.data(), L.elements + k0, sizeof(double) * (size_t)K, hipMemcpyDeviceToHost, s)) &&
                                     MGCG_HIP(hipMemcpyAsync(c.data(), L.columnIndeces + k0, sizeof(int) * (size_t)K, hipMemcpyDeviceToHost, s)))) && MGCG_HIP(hipStreamSynchronize(s));
            out.reserve((size_t)(2 + Rn + 2 * K));
            out.push_back((double)Rn); out.push_back((double)K);
            for (long long i = r0; i < r1; ++i) out.push_back((double)(ro[(size_t)i + 1] - ro[(size_t)i]));
            out.insert(out.end(), v.begin(), v.end());
            for (int j : c) out.push_back((double)j);
        };
        std::vector<double> toLower, toUpper, fromLower, fromUpper;
        if (L.z0 > 0) pack(0, edge, toLower);                                         // becomes the lower neighbour's upper planes
        if (L.z1 < L.nz) pack(L.n - edge, L.n, toUpper);
        ok = comm_neighbour_exchange_host(comm, toLower, toUpper, fromLower, fromUpper, lok) && lok;
        // the extended slab: [rows from below | my rows | rows from above]
        const long long rowsLo = L.offset - L.extBase, rowsHi = L.extRows - rowsLo - L.n;
        auto header = [&](const std::vector<double>& v, long long rows, long long& K) {
            K = 0;
            if (rows == 0) return v.empty() || nranks == 1;
            if (v.size() < 2 || (long long)v[0] != rows) return false;
            K = (long long)v[1];
            return K >= 0 && (long long)v.size() == 2 + rows + 2 * K;
        };
        long long kLo = 0, kHi = 0;
        if (ok && nranks > 1 && (!header(fromLower, rowsLo, kLo) || !header(fromUpper, rowsHi, kHi))) { set_error("MgSetup: a neighbour sent %zu / %zu values for the halo rows of level %d, not what %lld / %lld rows need", fromLower.size(), fromUpper.size(), l, rowsLo, rowsHi); ok = false; }
        if (ok && nranks == 1 && (rowsLo != 0 || rowsHi != 0)) ok = false;             // (one rank owns every plane: nothing beyond the slab)
        L.extNnz = kLo + L.nnz + kHi;
        if (ok && L.extNnz >= 0x7fffffffLL) { set_error("MgSetup: extended slab of level %d exceeds int32 offsets", l); ok = false; }
        std::vector<int> ero;
        std::vector<int> cLo, cHi;
        if (ok) {
            ero.resize((size_t)L.extRows + 1);
            long long run = 0; size_t at = 0;
            ero[at++] = 0;
            for (long long i = 0; i < rowsLo; ++i) { run += (long long)fromLower[2 + (size_t)i]; ero[at++] = (int)run; }
            for (long long i = 0; i < L.n; ++i) { run += ro[(size_t)i + 1] - ro[(size_t)i]; ero[at++] = (int)run; }
            for (long long i = 0; i < rowsHi; ++i) { run += (long long)fromUpper[2 + (size_t)i]; ero[at++] = (int)run; }
            if (run != L.extNnz) { set_error("MgSetup: halo rows of level %d are inconsistent", l); ok = false; }
            cLo.resize((size_t)kLo); cHi.resize((size_t)kHi);
            for (long long k = 0; k < kLo; ++k) cLo[(size_t)k] = (int)fromLower[2 + (size_t)rowsLo + (size_t)kLo + (size_t)k];
            for (long long k = 0; k < kHi; ++k) cHi[(size_t)k] = (int)fromUpper[2 + (size_t)rowsHi + (size_t)kHi + (size_t)k];
        }
        const size_t nzAlloc = (size_t)(L.extNnz > 0 ? L.extNnz : 1);
        ok = ok && MGCG_HIP(hipMalloc((void**)&L.extElements, sizeof(double) * nzAlloc)) && MGCG_HIP(hipMalloc((void**)&L.extColumnIndeces, sizeof(int) * nzAlloc)) &&
             MGCG_HIP(hipMalloc((void**)&L.extRowOffsets, sizeof(int) * ((size_t)L.extRows + 1))) && MGCG_HIP(hipMalloc((void**)&L.extDinv, sizeof(double) * (size_t)(L.extRows > 0 ? L.extRows : 1))) &&
             MGCG_HIP(hipMalloc((void**)&L.bFull, sizeof(double) * (size_t)L.nGlobal)) && MGCG_HIP(hipMemsetAsync(L.bFull, 0, sizeof(double) * (size_t)L.nGlobal, s));
        if (ok) {
            ok = MGCG_HIP(hipMemcpyAsync(L.extRowOffsets, ero.data(), sizeof(int) * ero.size(), hipMemcpyHostToDevice, s));
            if (kLo > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(L.extElements, fromLower.data() + 2 + rowsLo, sizeof(double) * (size_t)kLo, hipMemcpyHostToDevice, s)) &&
                                    MGCG_HIP(hipMemcpyAsync(L.extColumnIndeces, cLo.data(), sizeof(int) * (size_t)kLo, hipMemcpyHostToDevice, s));
            if (L.nnz > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(L.extElements + kLo, L.elements, sizeof(double) * (size_t)L.nnz, hipMemcpyDeviceToDevice, s)) &&
                                      MGCG_HIP(hipMemcpyAsync(L.extColumnIndeces + kLo, L.columnIndeces, sizeof(int) * (size_t)L.nnz, hipMemcpyDeviceToDevice, s));
            if (kHi > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(L.extElements + kLo + L.nnz, fromUpper.data() + 2 + rowsHi, sizeof(double) * (size_t)kHi, hipMemcpyHostToDevice, s)) &&
                                    MGCG_HIP(hipMemcpyAsync(L.extColumnIndeces + kLo + L.nnz, cHi.data(), sizeof(int) * (size_t)kHi, hipMemcpyHostToDevice, s));
            if (ok && L.extRows > 0) launch_extract_dinv(s, L.extElements, L.extRowOffsets, L.extColumnIndeces, L.extRows, L.extBase, L.extDinv);
            ok = ok && MGCG_HIP(hipStreamSynchronize(s));                               // (the host staging vectors go out of scope)
        }
        if (ok) {                                                                       // the right-hand side moves into the full-length buffer
            if (L.b) (void)hipFree(L.b);
            L.b = L.bFull + L.offset;
        }
        // may the level form its iterates per gather (mg_deep_level)?  One diagonal for every row of the extended slab, power-of-two nx and
        // ny, the row-tile kernel on both matrices.  (A rank-local decision: the results are the same bits either way.)
        if (ok && l < mg->levels - 1) {
            int differs = 1; double d0 = 0.0;
            int* dflag = nullptr;
            bool u = MGCG_HIP(hipMalloc((void**)&dflag, sizeof(int))) && MGCG_HIP(hipMemsetAsync(dflag, 0, sizeof(int), s));
            if (u) launch_uniform_check(s, L.extDinv, L.extRows, dflag);
            u = u && MGCG_HIP(hipMemcpyAsync(&differs, dflag, sizeof(int), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipMemcpyAsync(&d0, L.extDinv, sizeof(double), hipMemcpyDeviceToHost, s)) &&
                MGCG_HIP(hipStreamSynchronize(s));
            if (dflag) (void)hipFree(dflag);
            SpmvArgs own{}; own.elements = L.elements; own.columnIndeces = L.columnIndeces; own.elementsCount = (int)L.nnz; own.rowCount = (int)L.n;
            SpmvArgs ext{}; ext.elements = L.extElements; ext.columnIndeces = L.extColumnIndeces; ext.elementsCount = (int)L.extNnz; ext.rowCount = (int)L.extRows;
            const bool can = u && differs == 0 && L.dinvUniform && d0 == L.dinvScalar && L.nx >= 2 && log2_exact(L.nx) >= 1 && log2_exact(L.ny) >= 0 && L.nGlobal < 0x7fffffffLL &&
                             spmv_takes_rowtile(own, L.cfg) && spmv_takes_rowtile(ext, L.cfg);
            foldsEverywhere = foldsEverywhere && can;
        }
        // the plan that brings the `deep` planes of b (collective; every rank reaches it with the same verdict)
        if (!comm_agree(comm, ok, "MgSetup")) { ok = false; break; }
        L.bHalo = halo_plan_create(comm, L.nGlobal, L.offset, L.n, (int)L.extBase, (int)(L.extBase + L.extRows - 1));
        if (!comm_agree(comm, L.bHalo != nullptr, "MgSetup")) { ok = false; break; }
    }
    mg->deep = ok && wantDeep;
    mg->deepFolds = mg->deep && foldsEverywhere;
    if (mg->deep) {       // the loop's residual with room for one grid plane either side (SolveMgParallel keeps r there)
        const MgLevel& L0 = mg->lv[0];
        const size_t len = (size_t)(L0.n + 2LL * L0.nx * L0.ny);
        ok = MGCG_HIP(hipMalloc((void**)&mg->rExt, sizeof(double) * len)) && MGCG_HIP(hipMemsetAsync(mg->rExt, 0, sizeof(double) * len, s)) && MGCG_HIP(hipStreamSynchronize(s));
    }
    // (whether the loop keeps r there decides which exchanges the cycle makes: every rank has the buffer, or the set-up fails on every rank)
    if (multi && wantDeep && !comm_agree(comm, ok, "MgSetup")) ok = false;
    // The finest level of the cycle exchanges its right-hand side and forms x_1 = omega d b per gather -- for the neighbours' columns with the
    // rank's OWN d -- or it exchanges the stored x_1: a decision every rank must take alike, and only when all of them hold the same diagonal.
    if (ok && multi && wantDeep) {
        const MgLevel& L0 = mg->lv[0];
        SpmvArgs probe{}; probe.elements = L0.elements; probe.columnIndeces = L0.columnIndeces; probe.elementsCount = (int)L0.nnz; probe.rowCount = (int)L0.n;
        bool mine = L0.dinvUniform && (L0.dcsr == nullptr || !L0.dcsr->usable) && spmv_takes_rowtile(probe, L0.cfg);
        const std::vector<double> say{ mine ? 1.0 : 0.0, L0.dinvScalar };
        std::vector<double> fromLower, fromUpper;
        ok = comm_neighbour_exchange_host(comm, say, say, fromLower, fromUpper, true);
        for (const std::vector<double>* v : { &fromLower, &fromUpper })
            if (!v->empty()) mine = mine && v->size() == 2 && (*v)[0] == 1.0 && (*v)[1] == L0.dinvScalar;
        bool all = false;
        ok = ok && comm_all(comm, mine, &all);
        mg->deep0 = ok && all;
    }
    if (!ok || mg->levels == 0) { MgDestroy(mg); return nullptr; }
    return mg;
}

MgcgMg* MgSetup(MgcgBlas* cublas, MgcgSparse* cusparse,
                Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                int elementsCount, int nx, int ny, int nz,
                int levels, double omega, int nu, int nuCoarse, double sigma)
{
    return MgSetupParallel(nullptr, cublas, cusparse, elementsVector, rowOffsetsVector, columnIndecesVector, elementsCount,
                           nx, ny, nz, 0, nz, levels, omega, nu, nuCoarse, sigma);
}

void MgDestroy(MgcgMg* mg)
{
    if (!mg) return;
    if (mg->stream) (void)hipStreamSynchronize(mg->stream);
    if (mg->rExt) (void)hipFree(mg->rExt);
    for (auto& L : mg->lv) {
        if (L.ownsMatrix) {       // freed addresses may be handed out again: analyses keyed by them are void
            if (L.elements) { analysis_note_write(L.elements, sizeof(double) * (size_t)(L.nnz > 0 ? L.nnz : 1)); (void)hipFree(L.elements); }
            if (L.rowOffsets) { analysis_note_write(L.rowOffsets, sizeof(int) * (size_t)(L.n + 1)); (void)hipFree(L.rowOffsets); }
            if (L.columnIndeces) { analysis_note_write(L.columnIndeces, sizeof(int) * (size_t)(L.nnz > 0 ? L.nnz : 1)); (void)hipFree(L.columnIndeces); }
        }
        if (L.dinv) (void)hipFree(L.dinv);
        if (L.xa) (void)hipFree(L.xa);
        if (L.xb) (void)hipFree(L.xb);
        if (L.bFull) { if (L.b == L.bFull + L.offset) L.b = nullptr; (void)hipFree(L.bFull); }      // (b pointed into it -- unless the set-up failed in between)
        if (L.extElements) (void)hipFree(L.extElements);
        if (L.extColumnIndeces) (void)hipFree(L.extColumnIndeces);
        if (L.extRowOffsets) (void)hipFree(L.extRowOffsets);
        if (L.extDinv) (void)hipFree(L.extDinv);
        if (L.bHalo) halo_plan_destroy(L.bHalo);
        if (L.b) (void)hipFree(L.b);
        if (L.r) (void)hipFree(L.r);
        if (L.halo) halo_plan_destroy(L.halo);
        if (L.transferHalo) halo_plan_destroy(L.transferHalo);
    }
    delete mg;
}

int MgSetInterpolation(MgcgMg* mg, int mode)
{
    if (!mg || (mode != 0 && mode != 1)) { set_error("MgSetInterpolation: mode must be 0 (piecewise constant) or 1 (cell-centred linear)"); return -1; }
    if (!device_state()) return -1;
    if (mode == 1 && mg->multi) {
        // collective: every rank builds, level by level, the plan that brings one grid plane from each z-neighbour
        for (MgLevel& L : mg->lv) {
            if (L.transferHalo) continue;
            const long long plane = (long long)L.nx * L.ny;
            long long lo = L.offset - plane, hi = L.offset + L.n + plane - 1;
            if (lo < 0) lo = 0;
            if (hi > L.nGlobal - 1) hi = L.nGlobal - 1;
            L.transferHalo = halo_plan_create(mg->comm, L.nGlobal, L.offset, L.n, (int)lo, (int)hi);
            if (!L.transferHalo) return -1;
        }
    }
    mg->interp = mode;
    return 0;
}

int MgLevels(const MgcgMg* mg) { return mg ? mg->levels : 0; }
long long MgLevelRows(const MgcgMg* mg, int l) { return (mg && l >= 0 && l < mg->levels) ? mg->lv[l].n : -1; }
long long MgLevelNnz(const MgcgMg* mg, int l) { return (mg && l >= 0 && l < mg->levels) ? mg->lv[l].nnz : -1; }

void MgLevelCopyCsr(const MgcgMg* mg, int l, double elements[], int columnIndeces[], int rowOffsets[])
{
    if (!mg || l < 0 || l >= mg->levels) { set_error("MgLevelCopyCsr: bad level"); return; }
    const MgLevel& L = mg->lv[l];
    (void)hipStreamSynchronize(mg->stream);
    (void)MGCG_HIP(hipMemcpy(elements, L.elements, sizeof(double) * (size_t)L.nnz, hipMemcpyDeviceToHost));
    (void)MGCG_HIP(hipMemcpy(columnIndeces, L.columnIndeces, sizeof(int) * (size_t)L.nnz, hipMemcpyDeviceToHost));
    (void)MGCG_HIP(hipMemcpy(rowOffsets, L.rowOffsets, sizeof(int) * (size_t)(L.n + 1), hipMemcpyDeviceToHost));
}
void MgLevelCopyDinv(const MgcgMg* mg, int l, double dinv[])
{
    if (!mg || l < 0 || l >= mg->levels) { set_error("MgLevelCopyDinv: bad level"); return; }
    (void)hipStreamSynchronize(mg->stream);
    (void)MGCG_HIP(hipMemcpy(dinv, mg->lv[l].dinv, sizeof(double) * (size_t)mg->lv[l].n, hipMemcpyDeviceToHost));
}

void MgApply(MgcgMg* mg, const double* r, double* z)
{
    if (!device_state()) return;
    if (!mg || !r || !z) { set_error("MgApply: null argument"); return; }
    (void)mg_apply(mg, r, z, nullptr);
    (void)MGCG_HIP(hipGetLastError());
}

int SolveMgParallel(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr, MgcgMg* mg,
                    Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                    Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector, Vector* zVector,
                    int count, int countForDevice, int offsetForDevice, int elementsCountForDevice, int minJ, int maxJ,
                    double allowableResidual, int minIteration, int maxIteration, int rule,
                    int* iteration, double* residual, double* residualTrace, int traceCapacity)
{
    (void)matDescr;
    if (!device_state()) return MGCG_ERROR;
    bool pre = true;
    if (!cublas || !cusparse || !mg || !zVector) { set_error("SolveMg: null handle"); pre = false; }
    pre = pre && check_vectors("SolveMg", elementsVector, rowOffsetsVector, columnIndecesVector, xVector, bVector, ApVector, pVector, rVector,
                               elementsCountForDevice, countForDevice, count);
    if (pre && (zVector->size < countForDevice || mg->lv[0].n != countForDevice || mg->lv[0].nGlobal != count || mg->lv[0].offset != offsetForDevice)) {
        set_error("SolveMg: z vector or hierarchy does not match the problem"); pre = false;
    }
    if (pre && (mg->nranks != MgcgCommSize(comm) || mg->multi != comm_multi(comm))) { set_error("SolveMg: the hierarchy was built for %d rank(s)%s", mg->nranks, mg->multi ? " on the several-ranks path" : ""); pre = false; }
    if (pre && (rule == MGCG_RULE_HANDMADECL || rule == MGCG_RULE_VIENNACL)) { set_error("SolveMg supports the 2-norm absolute rules only"); pre = false; }
    if (!pre) {       // leave together with the peers (see SolveParallel)
        if (MgcgCommSize(comm) > 1) (void)halo_plan_create(comm, count, offsetForDevice, countForDevice, minJ, maxJ, nullptr, 0, true, false);
        return MGCG_ERROR;
    }
    CgRun R;
    R.ws = &cublas->ws; R.cfg = cfg_of(cusparse); R.prof = &cusparse->prof; R.cusparse = cusparse; R.mg = mg; R.comm = comm; R.nranks = MgcgCommSize(comm); R.multi = comm_multi(comm);
    R.elements = elementsVector->data; R.rowOffsets = rowOffsetsVector->data; R.columnIndeces = columnIndecesVector->data; R.elementsCount = elementsCountForDevice;
    R.x = xVector->data; R.b = bVector->data; R.Ap = ApVector->data; R.p = pVector->data; R.r = rVector->data; R.z = zVector->data; R.pVec = pVector; R.ApVec = ApVector;
    R.count = count; R.nLocal = countForDevice; R.offset = offsetForDevice;
    R.tol = allowableResidual; R.minIt = minIteration; R.maxIt = maxIteration; R.rule = rule;
    if (R.multi) {
        R.halo = halo_plan_create(comm, count, offsetForDevice, countForDevice, minJ, maxJ, R.columnIndeces, R.elementsCount, true);
        if (!R.halo) return MGCG_ERROR;
        if (!cg_plan_overlap(R)) { halo_plan_destroy(R.halo); return MGCG_ERROR; }
    }
    // deep-halo cycle: the loop keeps r in the hierarchy's extended buffer (room for the halo planes the cycle's one exchange brings) and the
    // caller's vector receives it when the solve is over
    const bool rExtended = R.multi && mg->deep && mg->rExt != nullptr;
    if (rExtended) R.r = mg->rExt + (long long)mg->lv[0].nx * mg->lv[0].ny;
    int st = cg_solve(R, iteration, residual, residualTrace, traceCapacity);
    if (rExtended && countForDevice > 0) {
        analysis_note_write(rVector->data, sizeof(double) * (size_t)countForDevice);
        if (!MGCG_HIP(hipMemcpyAsync(rVector->data, R.r, sizeof(double) * (size_t)countForDevice, hipMemcpyDeviceToDevice, R.ws->stream)) || !MGCG_HIP(hipStreamSynchronize(R.ws->stream))) st = MGCG_ERROR;
    }
    if (R.halo) halo_plan_destroy(R.halo);
    return st;
}

int SolveMg(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr, MgcgMg* mg,
            Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
            Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector, Vector* zVector,
            int elementsCount, int count,
            double allowableResidual, int minIteration, int maxIteration, int rule,
            int* iteration, double* residual, double* residualTrace, int traceCapacity)
{
    return SolveMgParallel(nullptr, cublas, cusparse, matDescr, mg, elementsVector, rowOffsetsVector, columnIndecesVector,
                           xVector, bVector, ApVector, pVector, rVector, zVector, count, count, 0, elementsCount, 0, count - 1,
                           allowableResidual, minIteration, maxIteration, rule, iteration, residual, residualTrace, traceCapacity);
}

} // extern "C"
