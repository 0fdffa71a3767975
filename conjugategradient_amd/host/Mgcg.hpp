// Mgcg.hpp -- C++ twin of the reference's C# solver classes (namespace LWisteria.Mgcg,
// Mgcg/cuBlas/Mgcg/*.cs), calling libMgcgGpu.so through the C ABI of include/MgcgGpu.h exactly where
// the C# classes P/Invoke MgcgGpu.dll.  Host code only: no HIP types appear here.
//
//   LinerEquations           LinerEquations.cs:6-47
//   ConjugateGradient        ConjugateGradient.cs:6-84        (IsConverged :56-79)
//   ConjugateGradientGpu     ConjugateGradientGpu.cs:10-90
//   ConjugateGradientSingleGpu   ConjugateGradientSingleGpu.cs:9-179
//   ConjugateGradientParallelGpu ConjugateGradientParallelGpu.cs:11-595  (Solve() = the native multi-device loop, SolveParallel over
//                                RCCL, one host thread per device; the reference's host-driven phases stay behind UsePhases)
//   SparseMatrix             SparseMatrix.cs:8-101
//   VectorDouble / VectorInt VectorDouble.cs:8-113 / VectorInt.cs:8-106
// The CPU solver of the reference (ConjugateGradientCpu.cs) is NOT here: its restatement is the test
// oracle (oracle/cg_oracle.c), and the product has no CPU compute path.
#pragma once
#include <atomic>
#include <chrono>
#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "../../include/MgcgGpu.h"

namespace LWisteria { namespace Mgcg {

struct ApplicationException : std::runtime_error { using std::runtime_error::runtime_error; };   // ConjugateGradient.cs:73
struct MgcgError : std::runtime_error { using std::runtime_error::runtime_error; };

inline void Check(const char* where)
{
    const char* e = MgcgGetLastError();
    if (e && *e) { std::string m = std::string(where) + ": " + e; MgcgClearLastError(); throw MgcgError(m); }
}

class VectorDouble {
public:
    Vector* Ptr;
    explicit VectorDouble(int size) : Ptr(Create_Double(size)) { Check("Create_Double"); if (!Ptr) throw MgcgError("Create_Double returned NULL"); }
    ~VectorDouble() { Dispose(); }
    VectorDouble(const VectorDouble&) = delete;
    VectorDouble& operator=(const VectorDouble&) = delete;
    void Dispose() { if (Ptr) { Delete_Double(Ptr); Ptr = nullptr; } }
    void CopyFrom(const double* array, int count, int arrayOffset = 0, int vectorOffset = 0) { CopyFromArray_Double(Ptr, array, count, arrayOffset, vectorOffset); Check("CopyFromArray_Double"); }
    void CopyTo(double* array, int count, int arrayOffset = 0, int vectorOffset = 0) const { CopyToArray_Double(Ptr, array, count, vectorOffset, arrayOffset); Check("CopyToArray_Double"); }
    double* ToRawPtr() { return ToRawPtr_Double(Ptr); }
};

class VectorInt {
public:
    ::VectorInt* Ptr;
    explicit VectorInt(int size) : Ptr(Create_Int(size)) { Check("Create_Int"); if (!Ptr) throw MgcgError("Create_Int returned NULL"); }
    ~VectorInt() { Dispose(); }
    VectorInt(const VectorInt&) = delete;
    VectorInt& operator=(const VectorInt&) = delete;
    void Dispose() { if (Ptr) { Delete_Int(Ptr); Ptr = nullptr; } }
    void CopyFrom(int* array, int count, int arrayOffset = 0, int vectorOffset = 0) { CopyFromArray_Int(Ptr, array, count, arrayOffset, vectorOffset); Check("CopyFromArray_Int"); }
    void CopyTo(int* array, int count, int arrayOffset = 0, int vectorOffset = 0) const { CopyToArray_Int(Ptr, array, count, vectorOffset, arrayOffset); Check("CopyToArray_Int"); }
    int* ToRawPtr() { return ToRawPtr_Int(Ptr); }
};

// Parallel.For of the reference (ConjugateGradientParallelGpu.cs:427-530: five of them per iteration) runs its bodies on the
// .NET thread pool.  Here: one long-lived worker per device, woken per phase -- a phase of the 207 402-row driver is 5-80 us of
// GPU work, and creating and joining a std::thread per phase cost 45 us each (five per iteration: 2/3 of the iteration).  With
// one device the body runs on the calling thread, as Parallel.For does for a single index.
class DeviceWorkers {
    const int n;
    std::vector<std::thread> threads;
    std::vector<std::string> errors;
    std::mutex m;
    std::condition_variable go, finished;
    std::atomic<unsigned long> generation{0};
    std::atomic<int> pending{0};
    std::atomic<bool> quit{false};
    const std::function<void(int)>* job = nullptr;

    template <typename P> static bool SpinUntil(P ready)
    {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(100);
        for (int i = 0;; ++i) {
            if (ready()) return true;
            if ((i & 63) == 63 && std::chrono::steady_clock::now() > until) return false;
        }
    }
    void Body(int d)
    {
        try { (*job)(d); } catch (std::exception& e) { errors[(size_t)d] = e.what(); }
    }
    void Loop(int d)
    {
        unsigned long seen = 0;
        for (;;) {
            auto ready = [&] { return generation.load(std::memory_order_acquire) != seen || quit.load(std::memory_order_acquire); };
            if (!SpinUntil(ready)) { std::unique_lock<std::mutex> lock(m); go.wait(lock, ready); }
            if (quit.load(std::memory_order_acquire)) return;
            seen = generation.load(std::memory_order_acquire);
            Body(d);
            if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lock(m); finished.notify_one(); }
        }
    }
public:
    explicit DeviceWorkers(int count) : n(count), errors((size_t)count)
    {
        if (n > 1) for (int d = 0; d < n; d++) threads.emplace_back([this, d] { Loop(d); });
    }
    ~DeviceWorkers()
    {
        { std::lock_guard<std::mutex> lock(m); quit.store(true, std::memory_order_release); }
        go.notify_all();
        for (auto& t : threads) t.join();
    }
    DeviceWorkers(const DeviceWorkers&) = delete;
    DeviceWorkers& operator=(const DeviceWorkers&) = delete;
    void Run(const std::function<void(int)>& f)
    {
        job = &f;
        if (n == 1) Body(0);
        else {
            { std::lock_guard<std::mutex> lock(m); pending.store(n, std::memory_order_release); generation.fetch_add(1, std::memory_order_acq_rel); }
            go.notify_all();
            auto done = [&] { return pending.load(std::memory_order_acquire) == 0; };
            if (!SpinUntil(done)) { std::unique_lock<std::mutex> lock(m); finished.wait(lock, done); }
        }
        job = nullptr;
        for (auto& e : errors) if (!e.empty()) { std::string msg; msg.swap(e); for (auto& r : errors) r.clear(); throw MgcgError(msg); }
    }
};

class SparseMatrix {
public:
    std::vector<double> Elements;
    std::vector<int> ColumnIndeces;
    std::vector<int> RowOffsets;
    SparseMatrix(int rowCount, int maxNonzeroCountPerRow)
        : Elements((size_t)rowCount * maxNonzeroCountPerRow), ColumnIndeces((size_t)rowCount * maxNonzeroCountPerRow), RowOffsets((size_t)rowCount + 1) { Clear(); }
    void Clear()
    {
        for (auto& o : RowOffsets) o = 0;
        for (size_t i = 0; i < Elements.size(); i++) { Elements[i] = 0; ColumnIndeces[i] = -1; }
    }
    int RowCount() const { return (int)RowOffsets.size(); }    // sic: RowOffsets.Length (SparseMatrix.cs:93-100)
};

class LinerEquations {
public:
    SparseMatrix* A = nullptr;
    std::vector<double> x, b;
    LinerEquations(int count, int /*maxNonZeroCount*/) : x((size_t)count), b((size_t)count) {}
    virtual ~LinerEquations() {}
    int Count() const { return (int)x.size(); }
};

class ConjugateGradient : public LinerEquations {
public:
    const int MinIteration, MaxIteration;
    const double AllowableResidual;
    int Iteration = 0;
    double Residual = 0;
    ConjugateGradient(int count, int maxNonZeroCount, int minIteration, int maxIteration, double allowableResidual)
        : LinerEquations(count, maxNonZeroCount), MinIteration(minIteration), MaxIteration(maxIteration), AllowableResidual(allowableResidual) {}
    virtual void Solve() = 0;
protected:
    bool IsConverged() const
    {
        if (Iteration < MinIteration) return false;
        else if (Iteration > MaxIteration) throw ApplicationException("the pressure equation did not converge");
        return Residual < AllowableResidual;
    }
};

class ConjugateGradientGpu : public ConjugateGradient {
public:
    using ConjugateGradient::ConjugateGradient;
    virtual void Initialize() = 0;
    virtual void Read() = 0;
};

class ConjugateGradientSingleGpu : public ConjugateGradientGpu {
    MgcgBlas* cublas; MgcgSparse* cusparse; MgcgMatDescr* matDescr;
    VectorDouble vectorA; VectorInt vectorColumnIndeces, vectorRowOffsets;
    VectorDouble vectorB, vectorX, vectorAp, vectorP, vectorR;
public:
    ConjugateGradientSingleGpu(int count, int maxNonZeroCount, int minIteration, int maxIteration, double allowableResidual)
        : ConjugateGradientGpu(count, maxNonZeroCount, minIteration, maxIteration, allowableResidual),
          cublas(CreateBlas()), cusparse(CreateSparse()), matDescr(CreateMatDescr()),
          vectorA(count * maxNonZeroCount), vectorColumnIndeces(count * maxNonZeroCount), vectorRowOffsets(count + 1),
          vectorB(count), vectorX(count), vectorAp(count), vectorP(count), vectorR(count)
    { Check("ConjugateGradientSingleGpu"); if (!cublas || !cusparse) throw MgcgError("no GPU handle"); }
    ~ConjugateGradientSingleGpu() override { DestroyBlas(cublas); DestroySparse(cusparse); DestroyMatDescr(matDescr); }
    void Initialize() override
    {
        const int nonzeroCount = A->RowOffsets[Count()];
        vectorA.CopyFrom(A->Elements.data(), nonzeroCount);
        vectorColumnIndeces.CopyFrom(A->ColumnIndeces.data(), nonzeroCount);
        vectorRowOffsets.CopyFrom(A->RowOffsets.data(), Count() + 1);
        vectorB.CopyFrom(b.data(), Count());
        vectorX.CopyFrom(x.data(), Count());
    }
    void Solve() override
    {
        const int nonzeroCount = A->RowOffsets[Count()];
        int iteration = 0; double residual = 0;
        ::Solve(cublas, cusparse, matDescr, vectorA.Ptr, vectorRowOffsets.Ptr, vectorColumnIndeces.Ptr,
                vectorX.Ptr, vectorB.Ptr, vectorAp.Ptr, vectorP.Ptr, vectorR.Ptr,
                nonzeroCount, Count(), AllowableResidual, MinIteration, MaxIteration, &iteration, &residual);
        Iteration = iteration - 1;      // ConjugateGradientSingleGpu.cs:168
        Residual = residual;
        const char* e = MgcgGetLastError();
        if (e && *e) { std::string m(e); MgcgClearLastError(); if (m.find("did not converge") != std::string::npos) throw ApplicationException(m); throw MgcgError(m); }
    }
    void Read() override { vectorX.CopyTo(x.data(), Count()); }
};

// Every device of this process, one host thread per device (the reference's Parallel.For).  Solve() hands each device's thread
// the WHOLE loop (SolveParallel: halo of p by grouped ncclSend/ncclRecv, the three sums of :463,499,525 by ncclAllReduce, all on
// the device's stream -- no host round trip inside an iteration); the communicators of all devices are formed in the
// constructor (MgcgCommInitAll).  UsePhases = true brings back the reference's own structure (:424-565): five Parallel.For
// per iteration over Solve0..3, host-staged halo (SyncP) and host sums of the per-device dot products in device order.
class ConjugateGradientParallelGpu : public ConjugateGradientGpu {
public:
    bool UsePhases = false;
    bool BalanceNonzeros = false;        // not in the reference: row ranges of equal NONZERO count instead of floor(count / devices) rows, chosen at
                                         // Initialize() from A's row offsets (a matrix with uneven rows: the slowest device sets the iteration time)
    std::string LastPath = "none";       // "native loop (SolveParallel over <transport>)" or "host-driven phases (Solve0..3)"
    std::string CommFailure;             // why the communicators of the native loop could not be formed ("" when they were); UsePhases is then true
private:
    int deviceCount;
    std::vector<MgcgComm*> comms;
    std::vector<int> offsetsForDevice, minJ, maxJ;
    std::vector<MgcgBlas*> cublas; std::vector<MgcgSparse*> cusparse; std::vector<MgcgMatDescr*> matDescr;
    std::vector<VectorDouble*> vectorElements, vectorX, vectorB, vectorAp, vectorP, vectorR;
    std::vector<VectorInt*> vectorColumnIndeces, vectorRowOffsets;
    std::vector<double> bufferHost, resultsDot;

    std::unique_ptr<DeviceWorkers> workers;
    template <typename F> void ParallelFor(F f)
    {
        if (!workers) workers.reset(new DeviceWorkers(deviceCount));
        workers->Run([&](int d) { SetDevice(d); f(d); Check("device phase"); });
    }
    int CountForDevice(int d) const { return (0 <= d && d < deviceCount) ? offsetsForDevice[(size_t)d + 1] - offsetsForDevice[(size_t)d] : 0; }
    int ElementCount(int d) const { return A->RowOffsets[(size_t)offsetsForDevice[(size_t)d + 1]] - A->RowOffsets[(size_t)offsetsForDevice[(size_t)d]]; }
    double Sum() const { double s = 0; for (double v : resultsDot) s += v; return s; }   // resultsDot.Sum(): device-id order
    void SyncP()
    {
        auto widths = [&](int d, int& lastCount, int& nextCount) {
            lastCount = (d > 0) ? offsetsForDevice[(size_t)d] - minJ[(size_t)d] : 0;
            nextCount = (d < deviceCount - 1) ? maxJ[(size_t)d] - CountForDevice(d) - offsetsForDevice[(size_t)d] + 1 : 0;
        };
        ParallelFor([&](int d) { int l, n; widths(d, l, n); P2Host(vectorP[(size_t)d]->Ptr, bufferHost.data(), CountForDevice(d), offsetsForDevice[(size_t)d], l, n); });
        ParallelFor([&](int d) { int l, n; widths(d, l, n); P2Device(vectorP[(size_t)d]->Ptr, bufferHost.data(), CountForDevice(d), offsetsForDevice[(size_t)d], l, n); });
    }
public:
    ConjugateGradientParallelGpu(int count, int maxNonZeroCount, int minIteration, int maxIteration, double allowableResidual)
        : ConjugateGradientGpu(count, maxNonZeroCount, minIteration, maxIteration, allowableResidual), deviceCount(GetDeviceCount())
    {
        if (deviceCount < 1) throw MgcgError("no GPU");
        const size_t n = (size_t)deviceCount;
        comms.assign(n, nullptr);
        // The communicators of the native loop come first (before any handle or vector exists).  If they cannot be formed -- RCCL missing or
        // refusing on this host -- the class still works: it keeps the reference's own host-driven phases, which need no communicator, and
        // CommFailure / LastPath say why.
        if (MgcgCommInitAll(comms.data(), deviceCount) != 0) {
            const char* e = MgcgGetLastError();
            CommFailure = (e && *e) ? e : "MgcgCommInitAll failed";
            MgcgClearLastError();
            comms.assign(n, nullptr);
            UsePhases = true;
        }
        offsetsForDevice.assign(n + 1, 0);
        for (int i = 1; i < deviceCount; i++) offsetsForDevice[(size_t)i] = offsetsForDevice[(size_t)i - 1] + (int)std::floor((double)Count() / deviceCount);
        offsetsForDevice[n] = Count();
        resultsDot.assign(n, 0); bufferHost.assign((size_t)Count(), 0); minJ.assign(n, 0); maxJ.assign(n, 0);
        cublas.assign(n, nullptr); cusparse.assign(n, nullptr); matDescr.assign(n, nullptr);
        vectorElements.assign(n, nullptr); vectorX.assign(n, nullptr); vectorB.assign(n, nullptr); vectorAp.assign(n, nullptr); vectorP.assign(n, nullptr); vectorR.assign(n, nullptr);
        vectorColumnIndeces.assign(n, nullptr); vectorRowOffsets.assign(n, nullptr);
        try {
            ParallelFor([&](int d) {
                const size_t i = (size_t)d; const int c = CountForDevice(d);
                cublas[i] = CreateBlas(); cusparse[i] = CreateSparse(); matDescr[i] = CreateMatDescr();
                vectorElements[i] = new VectorDouble(c * maxNonZeroCount); vectorColumnIndeces[i] = new VectorInt(c * maxNonZeroCount); vectorRowOffsets[i] = new VectorInt(c + 1);
                vectorX[i] = new VectorDouble(c); vectorB[i] = new VectorDouble(c); vectorAp[i] = new VectorDouble(c); vectorP[i] = new VectorDouble(count); vectorR[i] = new VectorDouble(c);
            });
        } catch (...) { Release(); throw; }      // (a constructor that throws runs no destructor: nothing may be left on the devices)
    }
private:
    void Release()
    {
        for (int d = 0; d < deviceCount; d++) {
            const size_t i = (size_t)d; SetDevice(d);
            delete vectorElements[i]; delete vectorColumnIndeces[i]; delete vectorRowOffsets[i];
            delete vectorX[i]; delete vectorB[i]; delete vectorAp[i]; delete vectorP[i]; delete vectorR[i];
            vectorElements[i] = vectorX[i] = vectorB[i] = vectorAp[i] = vectorP[i] = vectorR[i] = nullptr; vectorColumnIndeces[i] = vectorRowOffsets[i] = nullptr;
            if (cublas[i]) DestroyBlas(cublas[i]);
            if (cusparse[i]) DestroySparse(cusparse[i]);
            if (matDescr[i]) DestroyMatDescr(matDescr[i]);
            cublas[i] = nullptr; cusparse[i] = nullptr; matDescr[i] = nullptr;
            if (comms[i]) MgcgCommDestroy(comms[i]);
            comms[i] = nullptr;
        }
    }
public:
    ~ConjugateGradientParallelGpu() override { Release(); }
    int DeviceCount() const { return deviceCount; }
    int OffsetForDevice(int d) const { return offsetsForDevice[(size_t)d]; }
    void Initialize() override
    {
        if (BalanceNonzeros) {
            const int n = Count();
            std::vector<int> off((size_t)deviceCount + 1, 0);
            const long long first = A->RowOffsets[0], nnz = (long long)A->RowOffsets[(size_t)n] - first;
            for (int r = 1; r < deviceCount; r++) {           // the first row whose offset reaches r / devices of the nonzeros
                const int target = (int)(first + (long long)r * nnz / deviceCount);
                const int cut = (int)(std::lower_bound(A->RowOffsets.begin(), A->RowOffsets.begin() + n + 1, target) - A->RowOffsets.begin());
                off[(size_t)r] = std::max(std::min(cut, n), off[(size_t)r - 1]);
            }
            off[(size_t)deviceCount] = n;
            const std::vector<int> old = offsetsForDevice;
            offsetsForDevice = off;
            ParallelFor([&](int d) {
                const size_t i = (size_t)d; const int c = CountForDevice(d);
                if (c == old[i + 1] - old[i]) return;
                delete vectorElements[i]; delete vectorColumnIndeces[i]; delete vectorRowOffsets[i]; delete vectorX[i]; delete vectorB[i]; delete vectorAp[i]; delete vectorR[i];
                const int e = std::max(ElementCount(d), 1);
                vectorElements[i] = new VectorDouble(e); vectorColumnIndeces[i] = new VectorInt(e); vectorRowOffsets[i] = new VectorInt(c + 1);
                vectorX[i] = new VectorDouble(c); vectorB[i] = new VectorDouble(c); vectorAp[i] = new VectorDouble(c); vectorR[i] = new VectorDouble(c);
            });
        }
        ParallelFor([&](int d) {
            const size_t i = (size_t)d;
            ::Initialize(A->Elements.data(), A->RowOffsets.data(), A->ColumnIndeces.data(), x.data(), b.data(),
                         vectorElements[i]->Ptr, vectorRowOffsets[i]->Ptr, vectorColumnIndeces[i]->Ptr, vectorX[i]->Ptr, vectorB[i]->Ptr, vectorP[i]->Ptr,
                         &minJ[i], &maxJ[i], Count(), CountForDevice(d), offsetsForDevice[i], ElementCount(d), A->RowOffsets[(size_t)offsetsForDevice[i]]);
        });
    }
    void Solve() override
    {
        if (UsePhases || !comms[0]) {
            LastPath = std::string("host-driven phases (Solve0..3)") + (CommFailure.empty() ? "" : " -- no communicators: " + CommFailure);
            SolvePhases(); return;
        }
        LastPath = std::string("native loop (SolveParallel over ") + MgcgCommTransport(comms[0]) + ")";
        std::vector<int> iteration((size_t)deviceCount, 0), status((size_t)deviceCount, MGCG_ERROR);
        std::vector<double> residual((size_t)deviceCount, 0.0);
        ParallelFor([&](int d) { const size_t i = (size_t)d;
            status[i] = SolveParallel(comms[i], cublas[i], cusparse[i], matDescr[i], vectorElements[i]->Ptr, vectorRowOffsets[i]->Ptr, vectorColumnIndeces[i]->Ptr,
                                      vectorX[i]->Ptr, vectorB[i]->Ptr, vectorAp[i]->Ptr, vectorP[i]->Ptr, vectorR[i]->Ptr,
                                      Count(), CountForDevice(d), offsetsForDevice[i], ElementCount(d), minJ[i], maxJ[i],
                                      AllowableResidual, MinIteration, MaxIteration, MGCG_RULE_CSHARP, &iteration[i], &residual[i], nullptr, 0);
            if (status[i] == MGCG_MAXIT_EXCEEDED) MgcgClearLastError();               // reported below as the reference does (ConjugateGradient.cs:70-74)
        });
        Iteration = iteration[0]; Residual = residual[0];                             // every rank holds the same all-reduced values
        for (int d = 0; d < deviceCount; d++) {
            if (status[(size_t)d] == MGCG_MAXIT_EXCEEDED) throw ApplicationException("the pressure equation did not converge");
            if (status[(size_t)d] != MGCG_OK) throw MgcgError("SolveParallel failed on device " + std::to_string(d));
        }
    }
    // the reference's own structure, phase by phase (:424-565)
    void SolvePhases()
    {
        SyncP();
        ParallelFor([&](int d) { const size_t i = (size_t)d;
            resultsDot[i] = Solve0(cublas[i], cusparse[i], matDescr[i], vectorElements[i]->Ptr, vectorRowOffsets[i]->Ptr, vectorColumnIndeces[i]->Ptr,
                                   vectorX[i]->Ptr, vectorB[i]->Ptr, vectorAp[i]->Ptr, vectorP[i]->Ptr, vectorR[i]->Ptr, Count(), CountForDevice(d), offsetsForDevice[i], ElementCount(d)); });
        double rr = Sum();
        for (Iteration = 0;; Iteration++) {
            SyncP();
            ParallelFor([&](int d) { const size_t i = (size_t)d;
                resultsDot[i] = Solve1(cublas[i], cusparse[i], matDescr[i], vectorElements[i]->Ptr, vectorRowOffsets[i]->Ptr, vectorColumnIndeces[i]->Ptr,
                                       vectorAp[i]->Ptr, vectorP[i]->Ptr, Count(), CountForDevice(d), offsetsForDevice[i], ElementCount(d)); });
            const double alpha = rr / Sum();
            ParallelFor([&](int d) { const size_t i = (size_t)d;
                resultsDot[i] = Solve2(cublas[i], alpha, vectorX[i]->Ptr, vectorAp[i]->Ptr, vectorP[i]->Ptr, vectorR[i]->Ptr, CountForDevice(d), offsetsForDevice[i]); });
            const double rrNew = Sum();
            Residual = std::sqrt(rrNew);
            if (IsConverged()) break;
            const double beta = rrNew / rr;
            ParallelFor([&](int d) { const size_t i = (size_t)d; Solve3(cublas[i], beta, vectorP[i]->Ptr, vectorR[i]->Ptr, CountForDevice(d), offsetsForDevice[i]); });
            rr = rrNew;
        }
    }
    void Read() override
    {
        ParallelFor([&](int d) { const size_t i = (size_t)d; vectorX[i]->CopyTo(x.data(), CountForDevice(d), offsetsForDevice[i]); });
    }
};

}} // namespace LWisteria::Mgcg
