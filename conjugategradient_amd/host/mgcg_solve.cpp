// mgcg_solve -- command-line driver over the C ABI (SURVEY.md section 5 "config / flags": the reference fixes its
// problem in compile-time constants of each Main; here they are flags).  Builds a 5/7-point Poisson system in HBM,
// runs CG or MGCG on one device or on R devices of this process and prints one JSON line.
//   mgcg_solve [--nx N] [--ny N] [--nz N] [--mgcg] [--levels L] [--nu K] [--nu-coarse K] [--omega W] [--linear-transfer] [--tol T] [--rel-tol T]
//              [--min-it I] [--max-it I] [--rule native|csharp|simple|viennacl|handmadecl] [--compression 0|1|2] [--b V] [--x0 V] [--ranks R]
//              [--write-x FILE]     (the solution as raw little-endian doubles, for element-by-element comparison: MgcgMain.cs:129-162 compares so)
// --ranks R > 1: the grid is split into R equal z-slabs, one per device of this process and one host thread per device (the shape of the
// reference's ConjugateGradientParallelGpu), communicators from MgcgCommInitAll, SolveParallel / MgSetupParallel + SolveMgParallel per rank.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <string>
#include <vector>

#include "Mgcg.hpp"

using namespace LWisteria::Mgcg;

static int rule_of(const std::string& s)
{
    if (s == "native") return MGCG_RULE_NATIVE;
    if (s == "csharp") return MGCG_RULE_CSHARP;
    if (s == "simple") return MGCG_RULE_SIMPLE;
    if (s == "viennacl") return MGCG_RULE_VIENNACL;
    if (s == "handmadecl") return MGCG_RULE_HANDMADECL;
    throw MgcgError("unknown --rule " + s);
}

int main(int argc, char** argv)
{
    int nx = 64, ny = 64, nz = 64, levels = 3, nu = 1, nuCoarse = 4, minIt = 0, maxIt = -1, compression = 1, ranks = 1;
    double omega = 0, tol = 1e-8, relTol = 0, bValue = 1.0, x0Value = 0.0;
    bool mgcg = false, linearTransfer = false;
    std::string rule = "csharp", writeX;
    try {
        for (int i = 1; i < argc; ++i) {
            const std::string a = argv[i];
            auto val = [&]() -> const char* { if (i + 1 >= argc) throw MgcgError("missing value after " + a); return argv[++i]; };
            if (a == "--nx") nx = std::atoi(val()); else if (a == "--ny") ny = std::atoi(val()); else if (a == "--nz") nz = std::atoi(val());
            else if (a == "--n") { nx = ny = nz = std::atoi(val()); }
            else if (a == "--mgcg") mgcg = true;
            else if (a == "--linear-transfer") linearTransfer = true;
            else if (a == "--levels") levels = std::atoi(val()); else if (a == "--nu") nu = std::atoi(val()); else if (a == "--nu-coarse") nuCoarse = std::atoi(val());
            else if (a == "--omega") omega = std::atof(val()); else if (a == "--tol") tol = std::atof(val()); else if (a == "--rel-tol") relTol = std::atof(val());
            else if (a == "--min-it") minIt = std::atoi(val()); else if (a == "--max-it") maxIt = std::atoi(val());
            else if (a == "--rule") rule = val(); else if (a == "--compression") compression = std::atoi(val());
            else if (a == "--ranks") ranks = std::atoi(val());
            else if (a == "--write-x") writeX = val();
            else if (a == "--b") bValue = std::atof(val()); else if (a == "--x0") x0Value = std::atof(val());
            else throw MgcgError("unknown flag " + a);
        }
        const long long count = (long long)nx * ny * nz;
        if (nx < 1 || ny < 1 || nz < 1 || count > 0x7fffffffLL) throw MgcgError("bad grid");
        if (maxIt < 0) maxIt = (int)count;
        if (omega == 0) omega = nz > 1 ? 6.0 / 7.0 : 4.0 / 5.0;                // the smoothing optimum of the 3-D / 2-D Laplacian
        if (relTol > 0) tol = relTol * std::fabs(bValue) * std::sqrt((double)count);   // ||b||_2 for constant b
        if (GetDeviceCount() < 1) { Check("GetDeviceCount"); throw MgcgError("no HIP device"); }
        if (ranks < 1 || ranks > GetDeviceCount()) throw MgcgError("--ranks must be between 1 and the number of devices (" + std::to_string(GetDeviceCount()) + ")");
        if (ranks > 1 && nz % ranks != 0) throw MgcgError("--ranks must divide nz (equal z-slabs, floor(N / ranks) rows each)");
        // One rank per device, one host thread per rank (DeviceWorkers), the communicators of all ranks formed first -- the shape of
        // ConjugateGradientParallelGpu (ConjugateGradientParallelGpu.cs:264-324); with one rank the communicator is NULL.
        std::vector<MgcgComm*> comms((size_t)ranks, nullptr);
        if (ranks > 1 && MgcgCommInitAll(comms.data(), ranks) != 0) { Check("MgcgCommInitAll"); throw MgcgError("MgcgCommInitAll failed"); }
        struct Rank {
            MgcgBlas* blas = nullptr; MgcgSparse* sparse = nullptr; MgcgMatDescr* descr = nullptr; MgcgMg* mg = nullptr;
            std::unique_ptr<LWisteria::Mgcg::VectorDouble> e, x, b, Ap, p, r, z;
            std::unique_ptr<LWisteria::Mgcg::VectorInt> c, ro;
            long long nnz = 0; int z0 = 0, z1 = 0, rows = 0, offset = 0, minJ = 0, maxJ = -1, levels = 0;
            int status = MGCG_ERROR, iteration = 0; double residual = 0;
        };
        std::vector<Rank> R((size_t)ranks);
        std::vector<double> hx((size_t)count);
        DeviceWorkers workers(ranks);
        auto each = [&](const std::function<void(int, Rank&)>& f) { workers.Run([&](int d) { SetDevice(ranks > 1 ? d : 0); f(d, R[(size_t)d]); Check("rank phase"); }); };
        const std::string transport = ranks > 1 ? MgcgCommTransport(comms[0]) : "single";
        const auto tSetup = std::chrono::steady_clock::now();
        each([&](int d, Rank& k) {
            k.z0 = d * (nz / ranks); k.z1 = (d + 1) * (nz / ranks); k.rows = nx * ny * (k.z1 - k.z0); k.offset = nx * ny * k.z0;
            k.blas = CreateBlas(); k.sparse = CreateSparse(); k.descr = CreateMatDescr();
            Check("handles");
            MgcgSetMatrixCompression(k.sparse, compression);
            k.nnz = MgcgPoissonNnz(nx, ny, nz, k.z0, k.z1);
            using VD = LWisteria::Mgcg::VectorDouble; using VI = LWisteria::Mgcg::VectorInt;
            k.e.reset(new VD((int)k.nnz)); k.c.reset(new VI((int)k.nnz)); k.ro.reset(new VI(k.rows + 1));
            k.x.reset(new VD(k.rows)); k.b.reset(new VD(k.rows)); k.Ap.reset(new VD(k.rows)); k.r.reset(new VD(k.rows)); k.z.reset(new VD(k.rows));
            k.p.reset(new VD((int)count));                              // full length, as in the reference (:317)
            if (MgcgGeneratePoisson(k.e->Ptr, k.ro->Ptr, k.c->Ptr, nx, ny, nz, k.z0, k.z1) != 0) Check("MgcgGeneratePoisson");
            if (MgcgMinMaxColumn(k.c->Ptr, (int)k.nnz, &k.minJ, &k.maxJ) != 0) Check("MgcgMinMaxColumn");
            MgcgFill(k.b->Ptr, bValue); MgcgFill(k.x->Ptr, x0Value);
            MgcgDeviceSynchronize();
        });
        double setupS = 0;
        if (mgcg) {
            const auto t0 = std::chrono::steady_clock::now();
            each([&](int d, Rank& k) {
                k.mg = MgSetupParallel(comms[(size_t)d], k.blas, k.sparse, k.e->Ptr, k.ro->Ptr, k.c->Ptr, (int)k.nnz, nx, ny, nz, k.z0, k.z1, levels, omega, nu, nuCoarse, 0.5);
                if (!k.mg) { Check("MgSetup"); throw MgcgError("MgSetup failed"); }
                if (linearTransfer && MgSetInterpolation(k.mg, 1) != 0) { Check("MgSetInterpolation"); throw MgcgError("MgSetInterpolation failed"); }
                MgcgDeviceSynchronize();
                k.levels = MgLevels(k.mg);
            });
            setupS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        }
        (void)tSetup;
        const int ruleId = rule_of(rule);
        std::vector<std::string> errors((size_t)ranks);
        const auto t0 = std::chrono::steady_clock::now();               // Solve() only, like MgcgMain.cs:121-126
        each([&](int d, Rank& k) {
            k.status = mgcg
                ? SolveMgParallel(comms[(size_t)d], k.blas, k.sparse, k.descr, k.mg, k.e->Ptr, k.ro->Ptr, k.c->Ptr, k.x->Ptr, k.b->Ptr, k.Ap->Ptr, k.p->Ptr, k.r->Ptr, k.z->Ptr,
                                  (int)count, k.rows, k.offset, (int)k.nnz, k.minJ, k.maxJ, tol, minIt, maxIt, ruleId, &k.iteration, &k.residual, nullptr, 0)
                : SolveParallel(comms[(size_t)d], k.blas, k.sparse, k.descr, k.e->Ptr, k.ro->Ptr, k.c->Ptr, k.x->Ptr, k.b->Ptr, k.Ap->Ptr, k.p->Ptr, k.r->Ptr,
                                (int)count, k.rows, k.offset, (int)k.nnz, k.minJ, k.maxJ, tol, minIt, maxIt, ruleId, &k.iteration, &k.residual, nullptr, 0);
            const char* e = MgcgGetLastError();
            if (k.status != MGCG_OK && e) errors[(size_t)d] = e;
            MgcgClearLastError();                                       // (non-convergence is reported through the status below)
        });
        const double solveS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        each([&](int, Rank& k) { k.x->CopyTo(hx.data() + k.offset, k.rows); });
        if (!writeX.empty()) {
            FILE* f = std::fopen(writeX.c_str(), "wb");
            if (!f || std::fwrite(hx.data(), sizeof(double), hx.size(), f) != hx.size()) { if (f) std::fclose(f); throw MgcgError("cannot write " + writeX); }
            std::fclose(f);
        }
        int st = MGCG_OK; std::string err;
        for (int d = 0; d < ranks; ++d) if (R[(size_t)d].status != MGCG_OK) { st = R[(size_t)d].status; err = errors[(size_t)d]; break; }
        const int iteration = R[0].iteration; const double residual = R[0].residual;
        long long nnz = 0;
        for (const Rank& k : R) nnz += k.nnz;
        double sum = 0, mx = 0;
        for (double v : hx) { sum += v; mx = std::fmax(mx, std::fabs(v)); }
        std::printf("{\"solver\": \"%s\", \"grid\": [%d, %d, %d], \"rows\": %lld, \"nnz\": %lld, \"levels\": %d, \"status\": %d, \"iteration\": %d, "
                    "\"residual\": %.17g, \"tolerance\": %.17g, \"rule\": \"%s\", \"compression\": %d, \"setup_s\": %.6f, \"solve_s\": %.6f, "
                    "\"ms_per_iteration\": %.6f, \"sum_x\": %.17g, \"max_abs_x\": %.17g, \"ranks\": %d, \"transport\": \"%s\", \"error\": \"%s\"}\n",
                    mgcg ? "mgcg" : "cg", nx, ny, nz, count, nnz, R[0].levels, st, iteration, residual, tol, rule.c_str(), compression,
                    setupS, solveS, 1e3 * solveS / (iteration + 1), sum, mx, ranks, transport.c_str(), err.c_str());
        each([&](int d, Rank& k) {
            if (k.mg) MgDestroy(k.mg);
            k.e.reset(); k.c.reset(); k.ro.reset(); k.x.reset(); k.b.reset(); k.Ap.reset(); k.p.reset(); k.r.reset(); k.z.reset();
            DestroyBlas(k.blas); DestroySparse(k.sparse); DestroyMatDescr(k.descr);
            if (comms[(size_t)d]) MgcgCommDestroy(comms[(size_t)d]);
        });
        return st == MGCG_OK ? 0 : 2;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "mgcg_solve: %s\n", ex.what());
        return 1;
    }
}
