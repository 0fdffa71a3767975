// One process per GPU: RCCL over xGMI, bound at run time (dlopen) so that libMgcgGpu.so loads on
// hosts without RCCL and a single-GPU caller never touches it.
// Replaces the reference's host-staged "collectives": resultsDot.Sum() and SyncP/P2Host/P2Device
// (Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:384-419,463,499,525).
#include "common.hpp"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <condition_variable>
#include <memory>

namespace mgcg {

// Types and enums come from rccl.h; the functions are bound at run time.
typedef ncclUniqueId NcclUniqueId;
typedef ncclComm_t NcclComm;
static const ncclDataType_t NCCL_DOUBLE = ncclDouble;
static const ncclDataType_t NCCL_INT32 = ncclInt32;
static const ncclRedOp_t NCCL_SUM = ncclSum;

struct HaloPlan;
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

static Rccl* rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        // Prefer a copy already in the process (e.g. the one torch loaded) so there is one RCCL per process.
        const char* names[] = { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
        for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break; }
        if (!r.lib) for (const char* n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
        if (!r.lib) return;
#define SYM(field, name) r.field = (decltype(r.field))dlsym(r.lib, name)
        SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
        SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
        SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    });
    if (!r.lib || !r.GetUniqueId || !r.CommInitRank || !r.AllReduce || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd || !r.AllGather) {
        const char* why = r.lib ? nullptr : dlerror();        // dlerror() clears itself: read it once
        set_error("RCCL is not available (%s)", r.lib ? "librccl lacks a needed symbol" : (why ? why : "dlopen librccl.so.1 failed"));
        return nullptr;
    }
    return &r;
}

static bool nccl_ok(ncclResult_t rc, const char* what)
{
    if (rc == ncclSuccess) return true;
    Rccl* r = rccl();
    set_error("%s failed: %s", what, (r && r->GetErrorString) ? r->GetErrorString(rc) : "rccl error");
    return false;
}

} // namespace mgcg

// In-process loopback group: N ranks of ONE process (host threads, e.g. on MGCG_VIRTUAL_DEVICES of a single
// GPU) exchange through host memory behind a barrier.  It exists so that the multi-rank logic of
// SolveParallel / the distributed V-cycle can be tested on a one-GPU box where RCCL cannot form a
// communicator (one device per rank); production ranks use RCCL.
struct MgcgLoopback {
    int nranks = 1;
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0;
    long long generation = 0;
    std::vector<double> slots;                       // nranks * 8 doubles (all-reduce staging)
    std::vector<long long> meta;                     // nranks * 4 (halo-plan all-gather)
    std::vector<std::vector<double>> mailbox;        // [src * nranks + dst]
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const long long gen = generation;
        if (++arrived == nranks) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen; });
    }
};

struct MgcgComm {
    mgcg::NcclComm comm = nullptr;   // RCCL transport
    MgcgLoopback* loop = nullptr;    // loopback transport
    std::shared_ptr<MgcgLoopback> ownedLoop;   // MgcgCommInitAll on virtual devices: the group lives as long as any of its communicators
    int device = -1;                 // (virtual) device the communicator was made on
    int nranks = 1, rank = 0;
    hipStream_t stream = nullptr;
    double* scratch = nullptr;       // device, 8 doubles
    bool poisoned = false;           // host-staged transports: a local copy failed inside an exchange the peers had already entered -- the next all-reduce carries NaN
    double* gather = nullptr;        // device, nranks * 8 doubles (knob dot_order: the ranks' values side by side, added in rank order); allocated at first use
    // callback transport (host-staged; MgcgCommInitCallbacks)
    MgcgAllGatherFn cbAllGather = nullptr; MgcgAllReduceFn cbAllReduce = nullptr; MgcgExchangeFn cbExchange = nullptr; void* cbUser = nullptr;
    std::vector<std::vector<double>> cbSend, cbRecv;
    hipStream_t haloStream = nullptr;                 // side stream of the overlap schedule
    hipEvent_t evReady = nullptr, evHalo = nullptr;
    // The contiguous-range plan of the last SolveParallel / CgSteps on this communicator, reused when EVERY rank calls again with the same
    // partition (agreed in one 8-byte all-reduce): rebuilding it costs an all-gather with two stream synchronisations, inside the timed
    // solve.  Index-list plans depend on the column ids themselves and are rebuilt every time.
    mgcg::HaloPlan* cachedPlan = nullptr;
    long long cachedKey[5] = { -1, -1, -1, -1, -1 };  // count, offset, countLocal, minJ, maxJ
};

namespace mgcg {

// Several ranks -- or ONE rank with a real RCCL communicator and the force_multirank knob set (MGCG_FORCE_MULTIRANK = entries of an
// artificial halo): the several-ranks code path (reduction launches, ncclAllReduce on the stream, fork / join, interior and boundary
// row ranges, a self send/recv of that many entries) then runs on a one-GPU box, where its device-side cost can be timed.
bool comm_multi(const MgcgComm* c)
{
    if (!c) return false;
    return c->nranks > 1 || (c->comm != nullptr && tuning().forceMultiRank.load(std::memory_order_relaxed) > 0);
}

// out[i] = ((0 + v[0][i]) + v[1][i]) + ... : the order of resultsDot.Sum() (ConjugateGradientParallelGpu.cs:463,499,525)
__global__ void rank_order_sum_kernel(const double* __restrict__ all, int nranks, int count, double* __restrict__ out)
{
    const int i = threadIdx.x;
    if (i >= count) return;
    double a = 0.0;
    for (int q = 0; q < nranks; ++q) a += all[(size_t)q * count + i];
    out[i] = a;
}

bool comm_allreduce_sum(MgcgComm* c, double* devPtr, int count, hipStream_t s)
{
    if (!c) return true;
    const bool rankOrder = tuning().dotOrder.load(std::memory_order_relaxed) != 0;     // validation mode (the loopback transport adds in rank order anyway)
    if (rankOrder && c->cbAllGather && c->cbAllReduce && c->nranks > 1) {
        // callbacks: the caller's all-reduce adds in an order of its own; its all-gather (4 int64 per rank) carries the bit patterns instead
        if (count > 8) { set_error("callback all-reduce: at most 8 values"); return false; }
        double vals[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        bool ok = MGCG_HIP(hipMemcpyAsync(vals, devPtr, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
        if (c->poisoned) { ok = false; c->poisoned = false; }
        if (!ok) for (int i = 0; i < count; ++i) vals[i] = NAN;
        std::vector<long long> all(4 * (size_t)c->nranks);
        double sum[8];
        for (int at = 0; at < count; at += 4) {
            long long mine[4];
            memcpy(mine, vals + at, sizeof(mine));
            c->cbAllGather(mine, all.data(), c->cbUser);
            for (int i = at; i < count && i < at + 4; ++i) {
                double a = 0.0;
                for (int q = 0; q < c->nranks; ++q) { double v; memcpy(&v, &all[4 * (size_t)q + (size_t)(i - at)], sizeof(v)); a += v; }
                sum[i] = a;
            }
        }
        return MGCG_HIP(hipMemcpyAsync(devPtr, sum, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s)) && MGCG_HIP(hipStreamSynchronize(s)) && ok;
    }
    if (rankOrder && c->comm != nullptr && !c->loop && !c->cbAllReduce && c->nranks > 1) {
        Rccl* r = rccl();
        if (!r) return false;
        if (count > 8) { set_error("rank-ordered all-reduce: at most 8 values"); return false; }
        if (!c->gather && !MGCG_HIP(hipMalloc((void**)&c->gather, sizeof(double) * 8 * (size_t)c->nranks))) return false;
        if (!nccl_ok(r->AllGather(devPtr, c->gather, (size_t)count, NCCL_DOUBLE, c->comm, s), "ncclAllGather")) return false;
        hipLaunchKernelGGL(rank_order_sum_kernel, dim3(1), dim3(8), 0, s, (const double*)c->gather, c->nranks, count, devPtr);
        return MGCG_HIP(hipGetLastError());
    }
    if (c->loop) {
        if (count > 8) { set_error("loopback all-reduce: at most 8 values"); return false; }
        MgcgLoopback* g = c->loop;
        double mine[8];
        bool ok = MGCG_HIP(hipMemcpyAsync(mine, devPtr, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
        ok = ok && MGCG_HIP(hipStreamSynchronize(s));
        if (c->poisoned) { ok = false; c->poisoned = false; }      // (an exchange before this sum failed locally: every rank must stop in THIS iteration)
        for (int i = 0; i < count; ++i) g->slots[(size_t)c->rank * 8 + i] = ok ? mine[i] : NAN;
        g->barrier();
        double sum[8];
        for (int i = 0; i < count; ++i) { double a = 0; for (int q = 0; q < g->nranks; ++q) a += g->slots[(size_t)q * 8 + i]; sum[i] = a; }   // rank order: same bits on every rank
        g->barrier();
        ok = ok && MGCG_HIP(hipMemcpyAsync(devPtr, sum, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s));
        ok = ok && MGCG_HIP(hipStreamSynchronize(s));
        return ok;
    }
    if (c->cbAllReduce) {
        if (count > 8) { set_error("callback all-reduce: at most 8 values"); return false; }
        double vals[8];
        bool ok = MGCG_HIP(hipMemcpyAsync(vals, devPtr, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
        if (c->poisoned) { ok = false; c->poisoned = false; }
        if (!ok) for (int i = 0; i < count; ++i) vals[i] = NAN;     // the peers are inside the callback: join them, with a value that poisons the sum
        c->cbAllReduce(vals, count, c->cbUser);
        return MGCG_HIP(hipMemcpyAsync(devPtr, vals, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s)) && MGCG_HIP(hipStreamSynchronize(s)) && ok;
    }
    if (c->comm == nullptr) return true;             // single rank without a communicator: the local sum is the sum
    Rccl* r = rccl();
    if (!r) return false;
    return nccl_ok(r->AllReduce(devPtr, devPtr, (size_t)count, NCCL_DOUBLE, NCCL_SUM, c->comm, s), "ncclAllReduce");
}

// For every peer: the contiguous range of p this rank must send (its own rows that the peer's
// matrix slice references) and receive (the peer's rows this rank's slice references).
struct HaloPlan {
    int nranks = 1;
    std::vector<long long> sendBegin, sendCount, recvBegin, recvCount;
    // Index-list form (unstructured slices): per peer, the entries of p this rank sends / receives are lists of global ids
    // instead of one contiguous range.  The lists of all peers are concatenated (peer q's part starts at sendAt[q] / recvAt[q]);
    // pack gathers p[sendIdx] into sendBuf, unpack scatters recvBuf into p[recvIdx].
    bool indexed = false;
    bool cached = false;             // owned by the communicator's cache: halo_plan_destroy leaves it alone
    std::vector<long long> sendAt, recvAt;
    int* sendIdx = nullptr; int* recvIdx = nullptr;
    double* sendBuf = nullptr; double* recvBuf = nullptr;
    long long sendTotal = 0, recvTotal = 0, contiguousRecv = 0;
    // force_multirank on a one-rank communicator: selfCount entries of p travel rank 0 -> rank 0 into selfBuf (results are not used)
    long long selfBegin = 0, selfCount = 0;
    double* selfBuf = nullptr;
    // the measured overlap rule (halo_overlap_pays): -1 not measured yet, 0 exchange in line, 1 hide it behind the interior rows
    int overlapPays = -1;
    double exchangeUs = 0.0, forkJoinUs = 0.0;
};

// flags has `count` bytes followed by one aligned int: set when a column id lies outside [0, count) (nothing is written for it)
__global__ __launch_bounds__(kBlock) void halo_mark_kernel(const int* __restrict__ columnIndeces, long long nnz, long long ownBegin, long long ownEnd, long long count,
                                                           unsigned char* __restrict__ flags, int* __restrict__ outOfRange)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long k = (long long)blockIdx.x * kBlock + threadIdx.x; k < nnz; k += stride) {
        const int col = columnIndeces[k];
        if ((unsigned long long)(long long)col >= (unsigned long long)count) { *outOfRange = 1; continue; }
        if (col < ownBegin || col >= ownEnd) flags[col] = 1;       // (benign race: every writer stores the same byte)
    }
}
__global__ __launch_bounds__(kBlock) void halo_pack_kernel(const double* __restrict__ p, const int* __restrict__ idx, long long n, double* __restrict__ buf)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) buf[i] = p[idx[i]];
}
__global__ __launch_bounds__(kBlock) void halo_unpack_kernel(double* __restrict__ p, const int* __restrict__ idx, long long n, const double* __restrict__ buf)
{
    const long long stride = (long long)gridDim.x * kBlock;
    for (long long i = (long long)blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) p[idx[i]] = buf[i];
}
static int halo_grid(long long n) { const long long b = (n + kBlock - 1) / kBlock; return (int)(b < 1 ? 1 : (b > kMaxGrid ? kMaxGrid : b)); }

static thread_local long long t_lastHalo[3] = { 0, 0, 0 };       // indexed?, entries received per exchange, entries of the contiguous plan

// One exchange of host vectors of doubles between all ranks (plan set-up only): recv[q] must be sized to what q sends.
static bool exchange_host(MgcgComm* c, const std::vector<std::vector<double>>& send, std::vector<std::vector<double>>& recv)
{
    const int n = c->nranks;
    if (c->loop) {
        MgcgLoopback* g = c->loop;
        for (int q = 0; q < n; ++q) g->mailbox[(size_t)c->rank * n + q] = send[(size_t)q];
        g->barrier();
        bool ok = true;
        for (int q = 0; q < n; ++q) {
            const std::vector<double>& box = g->mailbox[(size_t)q * n + c->rank];
            if (box.size() != recv[(size_t)q].size()) { set_error("loopback exchange: size mismatch with rank %d", q); ok = false; continue; }
            recv[(size_t)q] = box;
        }
        g->barrier();
        return ok;
    }
    if (c->cbExchange) {
        std::vector<const double*> sp((size_t)n, nullptr); std::vector<double*> rp((size_t)n, nullptr);
        std::vector<long long> sc((size_t)n, 0), rc((size_t)n, 0);
        for (int q = 0; q < n; ++q) { sp[(size_t)q] = send[(size_t)q].data(); rp[(size_t)q] = recv[(size_t)q].data(); sc[(size_t)q] = (long long)send[(size_t)q].size(); rc[(size_t)q] = (long long)recv[(size_t)q].size(); }
        c->cbExchange(n, sp.data(), sc.data(), rp.data(), rc.data(), c->cbUser);
        return true;
    }
    Rccl* r = rccl();
    if (!r) return false;
    size_t sTot = 0, rTot = 0;
    for (int q = 0; q < n; ++q) { sTot += send[(size_t)q].size(); rTot += recv[(size_t)q].size(); }
    double *dS = nullptr, *dR = nullptr;
    bool ok = MGCG_HIP(hipMalloc((void**)&dS, sizeof(double) * (sTot + 1))) && MGCG_HIP(hipMalloc((void**)&dR, sizeof(double) * (rTot + 1)));
    size_t at = 0;
    for (int q = 0; ok && q < n; ++q) { if (!send[(size_t)q].empty()) ok = MGCG_HIP(hipMemcpyAsync(dS + at, send[(size_t)q].data(), sizeof(double) * send[(size_t)q].size(), hipMemcpyHostToDevice, c->stream)); at += send[(size_t)q].size(); }
    ok = ok && nccl_ok(r->GroupStart(), "ncclGroupStart");
    size_t sa = 0, ra = 0;
    for (int q = 0; ok && q < n; ++q) {
        if (!send[(size_t)q].empty()) ok = ok && nccl_ok(r->Send(dS + sa, send[(size_t)q].size(), NCCL_DOUBLE, q, c->comm, c->stream), "ncclSend");
        if (!recv[(size_t)q].empty()) ok = ok && nccl_ok(r->Recv(dR + ra, recv[(size_t)q].size(), NCCL_DOUBLE, q, c->comm, c->stream), "ncclRecv");
        sa += send[(size_t)q].size(); ra += recv[(size_t)q].size();
    }
    ok = nccl_ok(r->GroupEnd(), "ncclGroupEnd") && ok;
    ra = 0;
    for (int q = 0; ok && q < n; ++q) { if (!recv[(size_t)q].empty()) ok = MGCG_HIP(hipMemcpyAsync(recv[(size_t)q].data(), dR + ra, sizeof(double) * recv[(size_t)q].size(), hipMemcpyDeviceToHost, c->stream)); ra += recv[(size_t)q].size(); }
    ok = MGCG_HIP(hipStreamSynchronize(c->stream)) && ok;
    if (dS) (void)hipFree(dS);
    if (dR) (void)hipFree(dR);
    return ok;
}

static bool halo_plan_index(MgcgComm* c, HaloPlan* h, const std::vector<long long>& all, long long count, long long offset, long long countLocal,
                            const int* columnIndeces, long long nnz);

// Set-up exchange with the two neighbours in rank order (collective: every rank calls, the end ranks with nothing for the side they lack):
// toLower goes to rank - 1 (and arrives there as fromUpper), toUpper to rank + 1.  Sizes travel first, then the payloads; a local failure
// is carried through both rounds so that no rank leaves early.
bool comm_neighbour_exchange_host(MgcgComm* c, const std::vector<double>& toLower, const std::vector<double>& toUpper,
                                  std::vector<double>& fromLower, std::vector<double>& fromUpper, bool localOk)
{
    fromLower.clear(); fromUpper.clear();
    if (!c || c->nranks <= 1) return localOk;
    const int n = c->nranks, lo = c->rank - 1, hi = c->rank + 1;
    std::vector<std::vector<double>> send((size_t)n), recv((size_t)n);
    // round 1: sizes (-1: this rank has failed; its neighbours then fail with it)
    if (lo >= 0) { send[(size_t)lo] = { localOk ? (double)toLower.size() : -1.0 }; recv[(size_t)lo].assign(1, 0.0); }
    if (hi < n)  { send[(size_t)hi] = { localOk ? (double)toUpper.size() : -1.0 }; recv[(size_t)hi].assign(1, 0.0); }
    bool ok = exchange_host(c, send, recv) && localOk;
    long long nLo = lo >= 0 ? (long long)recv[(size_t)lo][0] : 0, nHi = hi < n ? (long long)recv[(size_t)hi][0] : 0;
    if (nLo < 0 || nHi < 0) { if (ok) set_error("a neighbouring rank failed while the halo rows of a multigrid level were being exchanged"); ok = false; nLo = nLo < 0 ? 0 : nLo; nHi = nHi < 0 ? 0 : nHi; }
    // round 2: payloads (a failed rank sends what it announced: nothing)
    for (auto& v : send) v.clear();
    for (auto& v : recv) v.clear();
    if (lo >= 0) { if (localOk) send[(size_t)lo] = toLower; recv[(size_t)lo].assign((size_t)nLo, 0.0); }
    if (hi < n)  { if (localOk) send[(size_t)hi] = toUpper; recv[(size_t)hi].assign((size_t)nHi, 0.0); }
    ok = exchange_host(c, send, recv) && ok;
    if (lo >= 0) fromLower.swap(recv[(size_t)lo]);
    if (hi < n) fromUpper.swap(recv[(size_t)hi]);
    return ok;
}

// Agreement on a local precondition among the ranks of a communicator (one all-reduce of a flag): false on EVERY rank when any rank
// says false, so that no rank walks into the collectives of a call that another rank has already left.
bool comm_agree(MgcgComm* c, bool localOk, const char* who)
{
    if (!c || c->nranks <= 1) return localOk;
    double failed = localOk ? 0.0 : 1.0;
    bool ok = MGCG_HIP(hipMemcpyAsync(c->scratch, &failed, sizeof(double), hipMemcpyHostToDevice, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
    ok = comm_allreduce_sum(c, c->scratch, 1, c->stream) && ok;
    ok = ok && MGCG_HIP(hipMemcpyAsync(&failed, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
    if (!ok) return false;
    if (!(failed == 0.0)) { if (localOk) set_error("%s: another rank failed its preconditions", who); return false; }
    return true;
}

// Do all ranks say yes?  One all-reduce like comm_agree, but a "no" is an answer, not a failure: nobody's error text is touched.
// false: the all-reduce itself failed (*all is then false).
bool comm_all(MgcgComm* c, bool mine, bool* all)
{
    *all = mine;
    if (!c || c->nranks <= 1) return true;
    *all = false;
    double no = mine ? 0.0 : 1.0;
    bool ok = MGCG_HIP(hipMemcpyAsync(c->scratch, &no, sizeof(double), hipMemcpyHostToDevice, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
    ok = comm_allreduce_sum(c, c->scratch, 1, c->stream) && ok;
    ok = ok && MGCG_HIP(hipMemcpyAsync(&no, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
    if (!ok) return false;
    *all = (no == 0.0);
    return true;
}

HaloPlan* halo_plan_create(MgcgComm* c, long long count, long long offset, long long countLocal, int minJ, int maxJ,
                           const int* columnIndeces, long long nnz, bool reuse, bool localOk)
{
    if (!localOk && !(c && c->nranks > 1 && reuse)) return nullptr;      // nobody to tell
    HaloPlan* h = new HaloPlan();
    if (c && c->nranks == 1 && comm_multi(c) && countLocal > 0) {
        long long w = tuning().forceMultiRank.load(std::memory_order_relaxed);
        if (w > countLocal) w = countLocal;
        h->selfBegin = offset; h->selfCount = w;
        if (!MGCG_HIP(hipMalloc((void**)&h->selfBuf, sizeof(double) * (size_t)w))) { delete h; return nullptr; }
        return h;
    }
    if (!c || c->nranks == 1) return h;
    const int n = c->nranks;
    h->nranks = n;
    if (reuse) {   // the same partition as last time on every rank?  (collective: one all-reduce of "mine changed")
        const long long key[5] = { count, offset, countLocal, (long long)minJ, (long long)maxJ };
        const bool same = c->cachedPlan != nullptr && memcmp(key, c->cachedKey, sizeof(key)) == 0;
        // {my partition changed, my preconditions failed}: the second flag makes a rank whose arguments are unusable leave TOGETHER with
        // its peers here, instead of returning early and leaving them blocked in the solve's first collective
        double flags[2] = { same ? 0.0 : 1.0, localOk ? 0.0 : 1.0 };
        bool ok = MGCG_HIP(hipMemcpyAsync(c->scratch, flags, sizeof(flags), hipMemcpyHostToDevice, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
        ok = comm_allreduce_sum(c, c->scratch, 2, c->stream) && ok;
        ok = ok && MGCG_HIP(hipMemcpyAsync(flags, c->scratch, sizeof(flags), hipMemcpyDeviceToHost, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
        if (!ok) { delete h; return nullptr; }
        if (!(flags[1] == 0.0)) { if (localOk) set_error("another rank failed its preconditions: the solve was not started on any rank"); delete h; return nullptr; }
        const double changed = flags[0];
        if (changed == 0.0) { delete h; return c->cachedPlan; }
        if (c->cachedPlan) { c->cachedPlan->cached = false; halo_plan_destroy(c->cachedPlan); c->cachedPlan = nullptr; }
        memcpy(c->cachedKey, key, sizeof(key));
    }
    // all-gather (offset, count, minJ, maxJ) of every rank
    std::vector<long long> all(4 * (size_t)n);
    if (c->loop) {
        MgcgLoopback* g = c->loop;
        g->meta[4 * (size_t)c->rank + 0] = offset; g->meta[4 * (size_t)c->rank + 1] = countLocal;
        g->meta[4 * (size_t)c->rank + 2] = minJ;   g->meta[4 * (size_t)c->rank + 3] = maxJ;
        g->barrier();
        all = g->meta;
        g->barrier();
    } else if (c->cbAllGather) {
        const long long mine[4] = { offset, countLocal, (long long)minJ, (long long)maxJ };
        c->cbAllGather(mine, all.data(), c->cbUser);
    } else {
        Rccl* r = rccl();
        if (!r) { delete h; return nullptr; }
        long long mine[4] = { offset, countLocal, (long long)minJ, (long long)maxJ };
        long long* dAll = nullptr;
        if (!MGCG_HIP(hipMalloc((void**)&dAll, sizeof(long long) * 4 * (size_t)n))) { delete h; return nullptr; }
        bool ok = MGCG_HIP(hipMemcpyAsync(dAll + 4 * c->rank, mine, sizeof(mine), hipMemcpyHostToDevice, c->stream));
        ok = ok && MGCG_HIP(hipStreamSynchronize(c->stream));
        // 4 int64 = 8 int32 per rank, in place
        ok = ok && nccl_ok(r->AllGather(dAll + 4 * c->rank, dAll, 8, NCCL_INT32, c->comm, c->stream), "ncclAllGather");
        ok = ok && MGCG_HIP(hipMemcpyAsync(all.data(), dAll, sizeof(long long) * 4 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        ok = ok && MGCG_HIP(hipStreamSynchronize(c->stream));
        (void)hipFree(dAll);
        if (!ok) { delete h; return nullptr; }
    }
    h->sendBegin.assign(n, 0); h->sendCount.assign(n, 0); h->recvBegin.assign(n, 0); h->recvCount.assign(n, 0);
    auto clip = [](long long lo, long long hi, long long a, long long b, long long& begin, long long& cnt) {
        const long long s = lo > a ? lo : a, e = hi < b ? hi : b;
        begin = s; cnt = e > s ? e - s : 0;
    };
    for (int q = 0; q < n; ++q) {
        if (q == c->rank) continue;
        const long long qOff = all[4 * q], qCnt = all[4 * q + 1], qMin = all[4 * q + 2], qMax = all[4 * q + 3];
        // what I need from q: q's rows inside [minJ, maxJ]
        if (countLocal > 0 && maxJ >= minJ) clip(minJ, (long long)maxJ + 1, qOff, qOff + qCnt, h->recvBegin[q], h->recvCount[q]);
        // what q needs from me: my rows inside [qMin, qMax]
        if (qCnt > 0 && qMax >= qMin) clip(qMin, qMax + 1, offset, offset + countLocal, h->sendBegin[q], h->sendCount[q]);
    }
    for (int q = 0; q < n; ++q) h->contiguousRecv += h->recvCount[q];
    // Unstructured slices: the contiguous ranges degenerate to (nearly) the whole vector.  Whether index lists pay is a
    // collective decision (every rank must take the same path): all ranks look at the same table.
    bool wide = false;
    for (int q = 0; q < n; ++q) {
        const long long qCnt = all[4 * q + 1], qMin = all[4 * q + 2], qMax = all[4 * q + 3];
        if (qCnt > 0 && qMax >= qMin && (qMax - qMin + 1 - qCnt) * 4 >= count) wide = true;     // some rank asks for >= a quarter of the vector from others
    }
    if (wide && columnIndeces != nullptr && count < 0x7fffffffLL) {
        if (!halo_plan_index(c, h, all, count, offset, countLocal, columnIndeces, nnz)) { halo_plan_destroy(h); return nullptr; }
    }
    // contiguous ranges are a function of the table every rank has just agreed on: keep them; `wide` is the same on every rank (same
    // table), so every rank caches or none does
    if (reuse && !wide) { h->cached = true; c->cachedPlan = h; }
    return h;
}

// Replace the contiguous ranges by index lists if (collectively) that at least halves the entries moved.
// Every rank walks through the same sequence of collectives whatever happens locally: a local failure is carried in `ok`, folded into
// the all-reduce that takes the decision (and into a second one behind the device allocations), and all ranks leave together --
// a rank that returned early would leave its peers blocked inside the next exchange.
static bool halo_plan_index(MgcgComm* c, HaloPlan* h, const std::vector<long long>& all, long long count, long long offset, long long countLocal,
                            const int* columnIndeces, long long nnz)
{
    const int n = c->nranks;
    bool ok = true;
    // 1. which columns outside my rows does my slice reference?  (a column id outside [0, count) is an error, not a write)
    std::vector<unsigned char> flags((size_t)count, 0);
    {
        const size_t flagAt = ((size_t)count + 7) & ~(size_t)7;
        unsigned char* dFlags = nullptr;
        int bad = 0;
        ok = MGCG_HIP(hipMalloc((void**)&dFlags, flagAt + 8)) && MGCG_HIP(hipMemsetAsync(dFlags, 0, flagAt + 8, c->stream));
        if (ok && nnz > 0) hipLaunchKernelGGL(halo_mark_kernel, dim3(halo_grid(nnz)), dim3(kBlock), 0, c->stream, columnIndeces, nnz, offset, offset + countLocal, count, dFlags, (int*)(dFlags + flagAt));
        ok = ok && MGCG_HIP(hipMemcpyAsync(flags.data(), dFlags, (size_t)count, hipMemcpyDeviceToHost, c->stream)) &&
             MGCG_HIP(hipMemcpyAsync(&bad, dFlags + flagAt, sizeof(int), hipMemcpyDeviceToHost, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
        if (dFlags) (void)hipFree(dFlags);
        if (ok && bad) { set_error("indexed halo: a column index of the slice lies outside [0, %lld)", count); ok = false; }
        if (!ok) flags.assign((size_t)count, 0);                  // take part in the exchanges below with empty lists
    }
    // 2. per owner, the sorted list of what I need
    std::vector<std::vector<double>> need((size_t)n), give((size_t)n), cnt((size_t)n, std::vector<double>(1, 0.0)), cntIn((size_t)n, std::vector<double>(1, 0.0));
    for (int q = 0; q < n; ++q) {
        if (q == c->rank) continue;
        const long long qOff = all[4 * q], qCnt = all[4 * q + 1];
        for (long long j = qOff; j < qOff + qCnt; ++j) if (flags[(size_t)j]) need[(size_t)q].push_back((double)j);
        cnt[(size_t)q][0] = (double)need[(size_t)q].size();
    }
    cnt[(size_t)c->rank].clear(); cntIn[(size_t)c->rank].clear();
    // 3. tell every owner how many and which (two rounds over the transport)
    ok = exchange_host(c, cnt, cntIn) && ok;
    for (int q = 0; q < n; ++q) if (q != c->rank) give[(size_t)q].resize((size_t)(cntIn[(size_t)q][0] > 0 ? cntIn[(size_t)q][0] : 0));
    ok = exchange_host(c, need, give) && ok;
    for (int q = 0; q < n && ok; ++q)
        for (double v : give[(size_t)q]) {
            const long long j = (long long)v;
            if (j < offset || j >= offset + countLocal) { set_error("indexed halo: rank %d asked rank %d for entry %lld it does not own", q, c->rank, j); ok = false; break; }
        }
    // 4. collective decision: total entries moved with lists vs with ranges -- and whether any rank failed so far
    double mine[3] = { 0.0, (double)h->contiguousRecv, ok ? 0.0 : 1.0 };
    for (int q = 0; q < n; ++q) mine[0] += (double)need[(size_t)q].size();
    auto agree = [&](double* v, int k) {                         // sum over ranks through the communicator's own scratch (allocated with it)
        bool g = MGCG_HIP(hipMemcpyAsync(c->scratch, v, sizeof(double) * (size_t)k, hipMemcpyHostToDevice, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
        g = comm_allreduce_sum(c, c->scratch, k, c->stream) && g;
        g = g && MGCG_HIP(hipMemcpyAsync(v, c->scratch, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost, c->stream)) && MGCG_HIP(hipStreamSynchronize(c->stream));
        return g;
    };
    if (!agree(mine, 3)) return false;
    if (!(mine[2] == 0.0)) { if (ok) set_error("indexed halo: another rank failed to build its lists"); return false; }
    if (!(mine[0] * 2.0 <= mine[1])) return true;                 // lists do not halve the volume: keep the ranges
    // 5. device lists and staging buffers
    h->sendAt.assign((size_t)n + 1, 0); h->recvAt.assign((size_t)n + 1, 0);
    std::vector<int> sIdx, rIdx;
    for (int q = 0; q < n; ++q) {
        h->sendAt[(size_t)q] = (long long)sIdx.size(); h->recvAt[(size_t)q] = (long long)rIdx.size();
        for (double v : give[(size_t)q]) sIdx.push_back((int)(long long)v);
        for (double v : need[(size_t)q]) rIdx.push_back((int)(long long)v);
    }
    h->sendAt[(size_t)n] = (long long)sIdx.size(); h->recvAt[(size_t)n] = (long long)rIdx.size();
    h->sendTotal = (long long)sIdx.size(); h->recvTotal = (long long)rIdx.size();
    ok = MGCG_HIP(hipMalloc((void**)&h->sendIdx, sizeof(int) * (sIdx.size() + 1))) && MGCG_HIP(hipMalloc((void**)&h->recvIdx, sizeof(int) * (rIdx.size() + 1))) &&
         MGCG_HIP(hipMalloc((void**)&h->sendBuf, sizeof(double) * (sIdx.size() + 1))) && MGCG_HIP(hipMalloc((void**)&h->recvBuf, sizeof(double) * (rIdx.size() + 1)));
    if (ok && !sIdx.empty()) ok = MGCG_HIP(hipMemcpyAsync(h->sendIdx, sIdx.data(), sizeof(int) * sIdx.size(), hipMemcpyHostToDevice, c->stream));
    if (ok && !rIdx.empty()) ok = MGCG_HIP(hipMemcpyAsync(h->recvIdx, rIdx.data(), sizeof(int) * rIdx.size(), hipMemcpyHostToDevice, c->stream));
    ok = MGCG_HIP(hipStreamSynchronize(c->stream)) && ok;
    double failed[1] = { ok ? 0.0 : 1.0 };
    if (!agree(failed, 1)) return false;
    if (!(failed[0] == 0.0)) { if (ok) set_error("indexed halo: another rank could not allocate its lists"); return false; }
    for (int q = 0; q < n; ++q) { h->sendBegin[q] = h->sendAt[(size_t)q]; h->sendCount[q] = h->sendAt[(size_t)q + 1] - h->sendAt[(size_t)q]; h->recvBegin[q] = h->recvAt[(size_t)q]; h->recvCount[q] = h->recvAt[(size_t)q + 1] - h->recvAt[(size_t)q]; }
    h->indexed = true;                                            // from here on send/recv Begin/Count address sendBuf / recvBuf
    return true;
}

void halo_plan_destroy(HaloPlan* h)
{
    if (!h || h->cached) return;
    if (h->sendIdx) (void)hipFree(h->sendIdx);
    if (h->recvIdx) (void)hipFree(h->recvIdx);
    if (h->sendBuf) (void)hipFree(h->sendBuf);
    if (h->recvBuf) (void)hipFree(h->recvBuf);
    if (h->selfBuf) (void)hipFree(h->selfBuf);
    delete h;
}
void halo_last(long long out[3]) { out[0] = t_lastHalo[0]; out[1] = t_lastHalo[1]; out[2] = t_lastHalo[2]; }

static bool halo_exchange_ranges(MgcgComm* c, HaloPlan* h, const double* src, double* dst, hipStream_t s);

bool halo_exchange(MgcgComm* c, HaloPlan* h, double* p, hipStream_t s)
{
    if (c && h && h->selfCount > 0 && c->comm != nullptr) {      // one rank made to take the several-ranks path: a grouped send/recv to itself
        Rccl* r = rccl();
        if (!r) return false;
        bool ok = nccl_ok(r->GroupStart(), "ncclGroupStart");
        ok = ok && nccl_ok(r->Send(p + h->selfBegin, (size_t)h->selfCount, NCCL_DOUBLE, 0, c->comm, s), "ncclSend");
        ok = ok && nccl_ok(r->Recv(h->selfBuf, (size_t)h->selfCount, NCCL_DOUBLE, 0, c->comm, s), "ncclRecv");
        return nccl_ok(r->GroupEnd(), "ncclGroupEnd") && ok;
    }
    if (!c || c->nranks == 1 || !h) return true;
    t_lastHalo[0] = h->indexed ? 1 : 0; t_lastHalo[1] = h->indexed ? h->recvTotal : h->contiguousRecv; t_lastHalo[2] = h->contiguousRecv;
    if (!h->indexed) return halo_exchange_ranges(c, h, p, p, s);
    if (h->sendTotal > 0) hipLaunchKernelGGL(halo_pack_kernel, dim3(halo_grid(h->sendTotal)), dim3(kBlock), 0, s, (const double*)p, (const int*)h->sendIdx, h->sendTotal, h->sendBuf);
    if (!halo_exchange_ranges(c, h, h->sendBuf, h->recvBuf, s)) return false;
    if (h->recvTotal > 0) hipLaunchKernelGGL(halo_unpack_kernel, dim3(halo_grid(h->recvTotal)), dim3(kBlock), 0, s, p, (const int*)h->recvIdx, h->recvTotal, (const double*)h->recvBuf);
    return MGCG_HIP(hipGetLastError());
}

// per peer: src[sendBegin, +sendCount) goes out, dst[recvBegin, +recvCount) comes in
static bool halo_exchange_ranges(MgcgComm* c, HaloPlan* h, const double* src, double* dst, hipStream_t s)
{
    const double* p = src;
    if (c->loop) {
        MgcgLoopback* g = c->loop;
        bool ok = true;
        for (int q = 0; q < h->nranks; ++q) {
            std::vector<double>& box = g->mailbox[(size_t)c->rank * g->nranks + q];
            box.resize((size_t)h->sendCount[q]);
            if (h->sendCount[q] > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(box.data(), p + h->sendBegin[q], sizeof(double) * (size_t)h->sendCount[q], hipMemcpyDeviceToHost, s));
        }
        ok = MGCG_HIP(hipStreamSynchronize(s)) && ok;
        g->barrier();
        for (int q = 0; q < h->nranks; ++q) {
            const std::vector<double>& box = g->mailbox[(size_t)q * g->nranks + c->rank];
            if (h->recvCount[q] > 0) {
                if ((long long)box.size() != h->recvCount[q]) { set_error("loopback halo: rank %d sent %zu values, rank %d expected %lld", q, box.size(), c->rank, h->recvCount[q]); ok = false; continue; }
                ok = ok && MGCG_HIP(hipMemcpyAsync(dst + h->recvBegin[q], box.data(), sizeof(double) * box.size(), hipMemcpyHostToDevice, s));
            }
        }
        ok = MGCG_HIP(hipStreamSynchronize(s)) && ok;
        g->barrier();
        // a local copy error: the peers have their halo and carry on -- this rank carries on WITH them and poisons the iteration's next
        // all-reduce (comm_allreduce_sum), so that every rank stops in the same iteration instead of blocking in a collective this rank left
        if (!ok) c->poisoned = true;
        return true;
    }
    if (c->cbExchange) {
        const int n = h->nranks;
        c->cbSend.resize((size_t)n); c->cbRecv.resize((size_t)n);
        std::vector<const double*> sp((size_t)n, nullptr); std::vector<double*> rp((size_t)n, nullptr);
        bool ok = true;
        for (int q = 0; q < n; ++q) {
            c->cbSend[(size_t)q].resize((size_t)h->sendCount[q]); c->cbRecv[(size_t)q].resize((size_t)h->recvCount[q]);
            sp[(size_t)q] = c->cbSend[(size_t)q].data(); rp[(size_t)q] = c->cbRecv[(size_t)q].data();
            if (h->sendCount[q] > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(c->cbSend[(size_t)q].data(), p + h->sendBegin[q], sizeof(double) * (size_t)h->sendCount[q], hipMemcpyDeviceToHost, s));
        }
        ok = MGCG_HIP(hipStreamSynchronize(s)) && ok;
        c->cbExchange(n, sp.data(), h->sendCount.data(), rp.data(), h->recvCount.data(), c->cbUser);     // (also after a local copy error: the peers are waiting in it)
        for (int q = 0; ok && q < n; ++q)
            if (h->recvCount[q] > 0) ok = ok && MGCG_HIP(hipMemcpyAsync(dst + h->recvBegin[q], c->cbRecv[(size_t)q].data(), sizeof(double) * (size_t)h->recvCount[q], hipMemcpyHostToDevice, s));
        ok = MGCG_HIP(hipStreamSynchronize(s)) && ok;
        // after a local copy error the peers hold garbage of this rank and carry on: stay with them, and let the iteration's next all-reduce
        // carry NaN (comm_allreduce_sum), so that every rank stops in the same iteration
        if (!ok) c->poisoned = true;
        return true;
    }
    Rccl* r = rccl();
    if (!r) return false;
    bool ok = nccl_ok(r->GroupStart(), "ncclGroupStart");
    for (int q = 0; ok && q < h->nranks; ++q) {
        if (h->sendCount[q] > 0) ok = ok && nccl_ok(r->Send(p + h->sendBegin[q], (size_t)h->sendCount[q], NCCL_DOUBLE, q, c->comm, s), "ncclSend");
        if (h->recvCount[q] > 0) ok = ok && nccl_ok(r->Recv(dst + h->recvBegin[q], (size_t)h->recvCount[q], NCCL_DOUBLE, q, c->comm, s), "ncclRecv");
    }
    ok = nccl_ok(r->GroupEnd(), "ncclGroupEnd") && ok;
    return ok;
}

// Overlapped form.  Every RCCL call stays on the rank's main stream (one stream per communicator, the pattern RCCL is
// used with everywhere); what moves is the interior rows: fork() makes a side stream wait for the point where p is
// final and returns it, the caller launches the interior rows there, exchanges the halo and multiplies the boundary
// rows on the main stream, and join() makes the main stream wait for the side stream.
bool halo_overlap_available(MgcgComm* c)
{
    if (!comm_multi(c)) return false;
    if (c->haloStream) return true;
    if (!MGCG_HIP(hipStreamCreateWithFlags(&c->haloStream, hipStreamNonBlocking))) { c->haloStream = nullptr; return false; }
    if (!MGCG_HIP(hipEventCreateWithFlags(&c->evReady, hipEventDisableTiming)) || !MGCG_HIP(hipEventCreateWithFlags(&c->evHalo, hipEventDisableTiming))) {
        (void)hipStreamDestroy(c->haloStream); c->haloStream = nullptr; return false;
    }
    return true;
}
hipStream_t halo_overlap_fork(MgcgComm* c, hipStream_t mainStream)
{
    if (!MGCG_HIP(hipEventRecord(c->evReady, mainStream))) return nullptr;
    if (!MGCG_HIP(hipStreamWaitEvent(c->haloStream, c->evReady, 0))) return nullptr;
    return c->haloStream;
}
bool halo_overlap_join(MgcgComm* c, hipStream_t mainStream)
{
    return MGCG_HIP(hipEventRecord(c->evHalo, c->haloStream)) && MGCG_HIP(hipStreamWaitEvent(mainStream, c->evHalo, 0));
}

__global__ void overlap_probe_kernel(int* p) { if (p) *p = 0; }

static thread_local double t_lastOverlapTimes[3] = { 0.0, 0.0, 0.0 };   // measured?, exchange in line (us), fork + launch + join (us)
void halo_overlap_clear_times() { t_lastOverlapTimes[0] = 0.0; t_lastOverlapTimes[1] = 0.0; t_lastOverlapTimes[2] = 0.0; }
void halo_overlap_last_times(double out[3]) { out[0] = t_lastOverlapTimes[0]; out[1] = t_lastOverlapTimes[1]; out[2] = t_lastOverlapTimes[2]; }

// Does hiding THIS plan's exchange behind the interior rows pay on THIS communicator?  Measured once per plan, on the stream the loop
// runs on: kReps exchanges of `vec` in line (idempotent: every halo entry is overwritten with its owner's value) against kReps
// fork / empty launch / join round trips of the overlap schedule, both with HIP events; the overlap schedule also splits the boundary
// rows off as a launch of their own (kBoundaryLaunchUs, profiles/r3/forced_path_timeline_*.log).  The two times are averaged over the
// ranks by one all-reduce, so every rank takes the same decision from the same bits (a collective: all ranks call together, as they
// do for the plan itself).  A local failure travels in the all-reduce as NaN and every rank answers "in line".
constexpr double kBoundaryLaunchUs = 13.0;
bool halo_overlap_pays(MgcgComm* c, HaloPlan* h, double* vec, hipStream_t s, bool* pays)
{
    *pays = false;
    t_lastOverlapTimes[0] = 0.0; t_lastOverlapTimes[1] = 0.0; t_lastOverlapTimes[2] = 0.0;
    if (!c || !h || !vec) return true;
    if (h->overlapPays < 0) {
        constexpr int kReps = 5;
        hipEvent_t e[4] = { nullptr, nullptr, nullptr, nullptr };
        bool ok = halo_overlap_available(c);
        for (int i = 0; i < 4; ++i) ok = MGCG_HIP(hipEventCreate(&e[i])) && ok;
        // every rank walks through the same exchanges whatever happened locally (a rank that left early would block its peers);
        // ex says whether they all went out, ok whether the local timing can be trusted
        bool ex = true;
        for (int i = 0; i < 2; ++i) ex = halo_exchange(c, h, vec, s) && ex;                  // warm-up: RCCL builds its channels at the first use
        ok = ok && MGCG_HIP(hipEventRecord(e[0], s));
        for (int i = 0; i < kReps; ++i) ex = halo_exchange(c, h, vec, s) && ex;
        ok = ok && MGCG_HIP(hipEventRecord(e[1], s));
        for (int i = 0; ok && i < kReps + 1; ++i) {                                          // local only: fork / empty launch / join
            if (i == 1) ok = MGCG_HIP(hipEventRecord(e[2], s));                              // (the first round trip is the warm-up)
            hipStream_t side = ok ? halo_overlap_fork(c, s) : nullptr;
            ok = ok && side != nullptr;
            if (ok) hipLaunchKernelGGL(overlap_probe_kernel, dim3(1), dim3(1), 0, side, (int*)nullptr);
            ok = ok && halo_overlap_join(c, s);
        }
        ok = ok && MGCG_HIP(hipEventRecord(e[3], s)) && MGCG_HIP(hipEventSynchronize(e[3]));
        double us[2] = { NAN, NAN };
        float msEx = 0.0f, msFj = 0.0f;
        if (ok && ex && MGCG_HIP(hipEventElapsedTime(&msEx, e[0], e[1])) && MGCG_HIP(hipEventElapsedTime(&msFj, e[2], e[3]))) {
            us[0] = 1e3 * (double)msEx / kReps; us[1] = 1e3 * (double)msFj / kReps;
        }
        (void)hipStreamSynchronize(s);
        for (int i = 0; i < 4; ++i) if (e[i]) (void)hipEventDestroy(e[i]);
        bool g = MGCG_HIP(hipMemcpyAsync(c->scratch, us, sizeof(us), hipMemcpyHostToDevice, s)) && MGCG_HIP(hipStreamSynchronize(s));
        g = comm_allreduce_sum(c, c->scratch, 2, s) && g;
        g = g && MGCG_HIP(hipMemcpyAsync(us, c->scratch, sizeof(us), hipMemcpyDeviceToHost, s)) && MGCG_HIP(hipStreamSynchronize(s));
        if (!g) return false;
        const int n = c->nranks > 0 ? c->nranks : 1;
        h->exchangeUs = us[0] / n; h->forkJoinUs = us[1] / n;
        // (a NaN -- some rank could not measure -- compares false: in line)
        h->overlapPays = (h->exchangeUs > h->forkJoinUs + kBoundaryLaunchUs) ? 1 : 0;
        if (tuning().verbose.load(std::memory_order_relaxed) >= 1)
            fprintf(stderr, "[MgcgGpu] halo of %lld entries: exchange in line %.1f us, fork + launch + join %.1f us (+ %.0f us for the boundary rows' own launch) -> %s\n",
                    h->indexed ? h->recvTotal : (h->selfCount > 0 ? h->selfCount : h->contiguousRecv), h->exchangeUs, h->forkJoinUs, kBoundaryLaunchUs, h->overlapPays ? "overlap" : "in line");
    }
    *pays = h->overlapPays == 1;
    t_lastOverlapTimes[0] = 1.0; t_lastOverlapTimes[1] = h->exchangeUs; t_lastOverlapTimes[2] = h->forkJoinUs;
    return true;
}

void preload_comm() { preload_code_object(reinterpret_cast<const void*>(&halo_pack_kernel)); }

} // namespace mgcg

using namespace mgcg;

extern "C" {

int MgcgRcclAvailable(void) { return mgcg::rccl() != nullptr ? 1 : 0; }

int MgcgCommGetUniqueId(void* id128)
{
    Rccl* r = rccl();
    if (!r || !id128) return -1;
    NcclUniqueId id;
    if (!nccl_ok(r->GetUniqueId(&id), "ncclGetUniqueId")) return -1;
    memcpy(id128, &id, sizeof(id));
    return 0;
}

MgcgComm* MgcgCommInitRank(const void* id128, int nranks, int rank)
{
    if (nranks < 1 || rank < 0 || rank >= nranks) { set_error("MgcgCommInitRank: bad rank %d of %d", rank, nranks); return nullptr; }
    if (tuning().failCommInit.load(std::memory_order_relaxed)) { set_error("MgcgCommInitRank: MGCG_FAIL_COMM_INIT is set (test hook)"); return nullptr; }
    // ncclCommInitRank comes first: before this rank creates its stream or allocates anything (only the device is selected)
    if (!select_device_only()) return nullptr;
    MgcgComm* c = new MgcgComm();
    c->nranks = nranks; c->rank = rank;
    if (nranks > 1 || id128 != nullptr) {          // a unique id with nranks == 1 builds a real one-rank communicator
        Rccl* r = rccl();
        if (!r || !id128) { if (r) set_error("MgcgCommInitRank: null unique id"); delete c; return nullptr; }
        NcclUniqueId id;
        memcpy(&id, id128, sizeof(id));
        if (!nccl_ok(r->CommInitRank(&c->comm, nranks, id, rank), "ncclCommInitRank")) { delete c; return nullptr; }
    }
    DeviceState* d = device_state();
    if (!d || !MGCG_HIP(hipMalloc((void**)&c->scratch, 8 * sizeof(double)))) {
        if (c->comm) { Rccl* r = rccl(); if (r && r->CommDestroy) (void)r->CommDestroy(c->comm); }
        delete c; return nullptr;
    }
    c->stream = d->stream;
    return c;
}

int MgcgCommInitAll(MgcgComm* comms[], int ndev)
{
    if (!comms || ndev < 1 || ndev > kMaxDevices) { set_error("MgcgCommInitAll: bad argument"); return -1; }
    for (int d = 0; d < ndev; ++d) comms[d] = nullptr;            // (the caller's array has ndev entries: that is the contract)
    if (tuning().failCommInit.load(std::memory_order_relaxed)) { set_error("MgcgCommInitAll: MGCG_FAIL_COMM_INIT is set (test hook)"); return -1; }
    int phys = 0;
    if (hipGetDeviceCount(&phys) != hipSuccess || phys <= 0) { set_error("no HIP device available (hipGetDeviceCount = %d): the HIP path cannot run", phys); return -1; }
    if (ndev > GetDeviceCount()) { set_error("MgcgCommInitAll: %d communicators asked for, %d device(s)", ndev, GetDeviceCount()); return -1; }
    const int saved = current_device();
    bool ok = true;
    std::shared_ptr<MgcgLoopback> group;
    if (ndev > 1 && ndev <= phys) {
        // one physical device per rank: RCCL.  All ncclCommInitRank calls of the process are made by this thread inside one group,
        // before any of the ranks creates its stream or allocates (as MgcgCommInitRank does for one rank per process).
        Rccl* r = rccl();
        if (!r) return -1;
        NcclUniqueId id;
        ok = nccl_ok(r->GetUniqueId(&id), "ncclGetUniqueId") && nccl_ok(r->GroupStart(), "ncclGroupStart");
        if (!ok) return -1;
        for (int d = 0; d < ndev && ok; ++d) {
            SetDevice(d);
            ok = select_device_only();
            if (!ok) break;
            comms[d] = new MgcgComm();
            comms[d]->nranks = ndev; comms[d]->rank = d; comms[d]->device = d;
            ok = nccl_ok(r->CommInitRank(&comms[d]->comm, ndev, id, d), "ncclCommInitRank");
        }
        ok = nccl_ok(r->GroupEnd(), "ncclGroupEnd") && ok;
    } else if (ndev > 1) {
        // fewer physical devices than ranks (MGCG_VIRTUAL_DEVICES, tests): the in-process loopback group, owned by its communicators
        group.reset(MgcgLoopbackCreate(ndev), [](MgcgLoopback* g) { MgcgLoopbackDestroy(g); });
        ok = group != nullptr;
        for (int d = 0; d < ndev && ok; ++d) { comms[d] = new MgcgComm(); comms[d]->nranks = ndev; comms[d]->rank = d; comms[d]->device = d; comms[d]->loop = group.get(); comms[d]->ownedLoop = group; }
    } else {
        comms[0] = new MgcgComm(); comms[0]->device = saved;
    }
    for (int d = 0; d < ndev && ok; ++d) {
        if (ndev > 1) SetDevice(d);
        DeviceState* st = device_state();
        ok = st != nullptr && MGCG_HIP(hipMalloc((void**)&comms[d]->scratch, 8 * sizeof(double)));
        if (ok) comms[d]->stream = st->stream;
    }
    if (ndev > 1) SetDevice(saved);
    if (!ok) { for (int d = 0; d < ndev; ++d) { if (comms[d]) { if (ndev > 1) SetDevice(d); MgcgCommDestroy(comms[d]); comms[d] = nullptr; } } if (ndev > 1) SetDevice(saved); return -1; }
    return 0;
}

const char* MgcgCommTransport(const MgcgComm* c)
{
    if (!c) return "none";
    if (c->loop) return "loopback";
    if (c->cbAllReduce) return "callbacks";
    if (c->comm) return "rccl";
    return "single";
}

__global__ void probe_empty_kernel(int* p) { if (p) *p = 0; }

double MgcgCommProbe(MgcgComm* c, int what, int count, int reps)
{
    DeviceState* d = device_state();
    if (!d || !c || reps < 1 || count < 0 || what < 0 || what > 4) { if (d) set_error("MgcgCommProbe: bad argument"); return NAN; }
    hipStream_t s = c->stream ? c->stream : d->stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    double* buf = nullptr;
    const size_t n = (size_t)(count > 8 ? count : 8);
    bool ok = MGCG_HIP(hipEventCreate(&e0)) && MGCG_HIP(hipEventCreate(&e1)) && MGCG_HIP(hipMalloc((void**)&buf, sizeof(double) * 4 * n)) &&
              MGCG_HIP(hipMemsetAsync(buf, 0, sizeof(double) * 4 * n, s));
    Rccl* r = (c->comm != nullptr) ? rccl() : nullptr;
    if ((what == 1 || what == 4) && !r) {                          // host-staged transports: the planes travel through the launcher, nothing to time on the device
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        if (buf) (void)hipFree(buf);
        return NAN;                                                // (not an error: no message is set)
    }
    auto once = [&]() -> bool {
        switch (what) {
        case 0: return comm_allreduce_sum(c, buf, count > 8 ? 8 : (count < 1 ? 1 : count), s);
        case 1: {
            if (!r) return true;                                   // host-staged transports: nothing to time on the device
            bool g = nccl_ok(r->GroupStart(), "ncclGroupStart");
            for (int q = 0; g && q < c->nranks; ++q) {
                if (q == c->rank && c->nranks > 1) continue;
                g = g && nccl_ok(r->Send(buf, (size_t)count, NCCL_DOUBLE, q, c->comm, s), "ncclSend");
                g = g && nccl_ok(r->Recv(buf + n, (size_t)count, NCCL_DOUBLE, q, c->comm, s), "ncclRecv");
            }
            return nccl_ok(r->GroupEnd(), "ncclGroupEnd") && g;
        }
        case 4: {                                                  // the stencil's exchange: `count` doubles to and from ranks rank - 1 and rank + 1 only
            if (!r) return true;
            bool g = nccl_ok(r->GroupStart(), "ncclGroupStart");
            for (int dq = -1; g && dq <= 1; dq += 2) {
                const int q = c->nranks > 1 ? c->rank + dq : 0;
                if (q < 0 || q >= c->nranks || (c->nranks == 1 && dq > 0)) continue;
                g = g && nccl_ok(r->Send(buf + (dq > 0 ? count : 0), (size_t)count, NCCL_DOUBLE, q, c->comm, s), "ncclSend");
                g = g && nccl_ok(r->Recv(buf + 2 * n + (dq > 0 ? count : 0), (size_t)count, NCCL_DOUBLE, q, c->comm, s), "ncclRecv");
            }
            return nccl_ok(r->GroupEnd(), "ncclGroupEnd") && g;
        }
        case 2: {
            if (!c->haloStream && (!MGCG_HIP(hipStreamCreateWithFlags(&c->haloStream, hipStreamNonBlocking)) ||
                                   !MGCG_HIP(hipEventCreateWithFlags(&c->evReady, hipEventDisableTiming)) || !MGCG_HIP(hipEventCreateWithFlags(&c->evHalo, hipEventDisableTiming)))) return false;
            hipStream_t side = halo_overlap_fork(c, s);
            if (!side) return false;
            hipLaunchKernelGGL(probe_empty_kernel, dim3(1), dim3(1), 0, side, (int*)nullptr);
            return halo_overlap_join(c, s);
        }
        default: hipLaunchKernelGGL(probe_empty_kernel, dim3(1), dim3(1), 0, s, (int*)nullptr); return true;
        }
    };
    for (int i = 0; ok && i < 3; ++i) ok = once();                // warm-up (RCCL builds its channels at the first use)
    ok = ok && MGCG_HIP(hipEventRecord(e0, s));
    for (int i = 0; ok && i < reps; ++i) ok = once();
    ok = ok && MGCG_HIP(hipEventRecord(e1, s)) && MGCG_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    ok = ok && MGCG_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipStreamSynchronize(s);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (buf) (void)hipFree(buf);
    return ok ? 1e3 * (double)ms / reps : NAN;
}

MgcgLoopback* MgcgLoopbackCreate(int nranks)
{
    if (nranks < 1 || nranks > 64) { set_error("MgcgLoopbackCreate: bad rank count %d", nranks); return nullptr; }
    MgcgLoopback* g = new MgcgLoopback();
    g->nranks = nranks;
    g->slots.assign((size_t)nranks * 8, 0.0);
    g->meta.assign((size_t)nranks * 4, 0);
    g->mailbox.resize((size_t)nranks * nranks);
    return g;
}

void MgcgLoopbackDestroy(MgcgLoopback* g) { delete g; }

MgcgComm* MgcgCommInitLoopback(MgcgLoopback* group, int rank)
{
    DeviceState* d = device_state();
    if (!d) return nullptr;
    if (!group || rank < 0 || rank >= group->nranks) { set_error("MgcgCommInitLoopback: bad argument"); return nullptr; }
    MgcgComm* c = new MgcgComm();
    c->nranks = group->nranks; c->rank = rank; c->stream = d->stream; c->loop = group;
    if (!MGCG_HIP(hipMalloc((void**)&c->scratch, 8 * sizeof(double)))) { delete c; return nullptr; }
    return c;
}

MgcgComm* MgcgCommInitCallbacks(int nranks, int rank, MgcgAllGatherFn allGather, MgcgAllReduceFn allReduce, MgcgExchangeFn exchange, void* user)
{
    DeviceState* d = device_state();
    if (!d) return nullptr;
    if (nranks < 1 || rank < 0 || rank >= nranks || !allGather || !allReduce || !exchange) { set_error("MgcgCommInitCallbacks: bad argument"); return nullptr; }
    MgcgComm* c = new MgcgComm();
    c->nranks = nranks; c->rank = rank; c->stream = d->stream;
    c->cbAllGather = allGather; c->cbAllReduce = allReduce; c->cbExchange = exchange; c->cbUser = user;
    if (!MGCG_HIP(hipMalloc((void**)&c->scratch, 8 * sizeof(double)))) { delete c; return nullptr; }
    return c;
}

void MgcgCommDestroy(MgcgComm* c)
{
    if (!c) return;
    if (c->cachedPlan) { c->cachedPlan->cached = false; halo_plan_destroy(c->cachedPlan); c->cachedPlan = nullptr; }
    if (c->comm) { Rccl* r = rccl(); if (r && r->CommDestroy) (void)r->CommDestroy(c->comm); }
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->gather) (void)hipFree(c->gather);
    if (c->evReady) (void)hipEventDestroy(c->evReady);
    if (c->evHalo) (void)hipEventDestroy(c->evHalo);
    if (c->haloStream) (void)hipStreamDestroy(c->haloStream);
    delete c;
}

int MgcgCommRank(const MgcgComm* c) { return c ? c->rank : 0; }
int MgcgCommSize(const MgcgComm* c) { return c ? c->nranks : 1; }

double MgcgCommAllReduceSum(MgcgComm* c, double value)
{
    if (!c) return value;
    if (!device_state()) return NAN;
    bool ok = MGCG_HIP(hipMemcpyAsync(c->scratch, &value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    ok = ok && MGCG_HIP(hipStreamSynchronize(c->stream));
    ok = ok && comm_allreduce_sum(c, c->scratch, 1, c->stream);
    double out = NAN;
    ok = ok && MGCG_HIP(hipMemcpyAsync(&out, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    ok = ok && MGCG_HIP(hipStreamSynchronize(c->stream));
    return ok ? out : NAN;
}

} // extern "C"
