// mgcg_solve -- command-line driver over the C ABI (SURVEY.md section 5 "config / flags": the reference fixes its
// problem in compile-time constants of each Main; here they are flags).  Builds a 5/7-point Poisson system in HBM,
// runs CG or MGCG on device 0 and prints one JSON line.
//   mgcg_solve [--nx N] [--ny N] [--nz N] [--mgcg] [--levels L] [--nu K] [--nu-coarse K] [--omega W] [--linear-transfer] [--tol T] [--rel-tol T]
//              [--min-it I] [--max-it I] [--rule native|csharp|simple|viennacl|handmadecl] [--compression 0|1|2] [--b V] [--x0 V]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

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
    int nx = 64, ny = 64, nz = 64, levels = 3, nu = 1, nuCoarse = 4, minIt = 0, maxIt = -1, compression = 1;
    double omega = 0, tol = 1e-8, relTol = 0, bValue = 1.0, x0Value = 0.0;
    bool mgcg = false, linearTransfer = false;
    std::string rule = "csharp";
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
            else if (a == "--b") bValue = std::atof(val()); else if (a == "--x0") x0Value = std::atof(val());
            else throw MgcgError("unknown flag " + a);
        }
        const long long count = (long long)nx * ny * nz;
        if (nx < 1 || ny < 1 || nz < 1 || count > 0x7fffffffLL) throw MgcgError("bad grid");
        if (maxIt < 0) maxIt = (int)count;
        if (omega == 0) omega = nz > 1 ? 6.0 / 7.0 : 4.0 / 5.0;                // the smoothing optimum of the 3-D / 2-D Laplacian
        if (relTol > 0) tol = relTol * std::fabs(bValue) * std::sqrt((double)count);   // ||b||_2 for constant b
        if (GetDeviceCount() < 1) { Check("GetDeviceCount"); throw MgcgError("no HIP device"); }
        SetDevice(0);
        MgcgBlas* blas = CreateBlas(); MgcgSparse* sparse = CreateSparse(); MgcgMatDescr* descr = CreateMatDescr();
        Check("handles");
        MgcgSetMatrixCompression(sparse, compression);
        const long long nnz = MgcgPoissonNnz(nx, ny, nz, 0, nz);
        LWisteria::Mgcg::VectorDouble e((int)nnz), x((int)count), b((int)count), Ap((int)count), p((int)count), r((int)count), z((int)count);
        LWisteria::Mgcg::VectorInt c((int)nnz), ro((int)count + 1);
        if (MgcgGeneratePoisson(e.Ptr, ro.Ptr, c.Ptr, nx, ny, nz, 0, nz) != 0) Check("MgcgGeneratePoisson");
        MgcgFill(b.Ptr, bValue); MgcgFill(x.Ptr, x0Value);
        MgcgDeviceSynchronize();
        Check("setup");
        MgcgMg* mg = nullptr;
        double setupS = 0;
        if (mgcg) {
            const auto t0 = std::chrono::steady_clock::now();
            mg = MgSetup(blas, sparse, e.Ptr, ro.Ptr, c.Ptr, (int)nnz, nx, ny, nz, levels, omega, nu, nuCoarse, 0.5);
            MgcgDeviceSynchronize();
            setupS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (!mg) { Check("MgSetup"); throw MgcgError("MgSetup failed"); }
            if (linearTransfer && MgSetInterpolation(mg, 1) != 0) { Check("MgSetInterpolation"); throw MgcgError("MgSetInterpolation failed"); }
        }
        int iteration = 0; double residual = 0;
        const auto t0 = std::chrono::steady_clock::now();               // Solve() only, like MgcgMain.cs:121-126
        const int st = mgcg
            ? SolveMg(blas, sparse, descr, mg, e.Ptr, ro.Ptr, c.Ptr, x.Ptr, b.Ptr, Ap.Ptr, p.Ptr, r.Ptr, z.Ptr, (int)nnz, (int)count, tol, minIt, maxIt, rule_of(rule), &iteration, &residual, nullptr, 0)
            : SolveEx(blas, sparse, descr, e.Ptr, ro.Ptr, c.Ptr, x.Ptr, b.Ptr, Ap.Ptr, p.Ptr, r.Ptr, (int)nnz, (int)count, tol, minIt, maxIt, rule_of(rule), &iteration, &residual, nullptr, 0);
        const double solveS = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const char* err = MgcgGetLastError();
        std::vector<double> hx((size_t)count);
        x.CopyTo(hx.data(), (int)count);
        double sum = 0, mx = 0;
        for (double v : hx) { sum += v; mx = std::fmax(mx, std::fabs(v)); }
        std::printf("{\"solver\": \"%s\", \"grid\": [%d, %d, %d], \"rows\": %lld, \"nnz\": %lld, \"levels\": %d, \"status\": %d, \"iteration\": %d, "
                    "\"residual\": %.17g, \"tolerance\": %.17g, \"rule\": \"%s\", \"compression\": %d, \"setup_s\": %.6f, \"solve_s\": %.6f, "
                    "\"ms_per_iteration\": %.6f, \"sum_x\": %.17g, \"max_abs_x\": %.17g, \"error\": \"%s\"}\n",
                    mgcg ? "mgcg" : "cg", nx, ny, nz, count, nnz, mg ? MgLevels(mg) : 0, st, iteration, residual, tol, rule.c_str(), compression,
                    setupS, solveS, 1e3 * solveS / (iteration + 1), sum, mx, (err && st != MGCG_OK) ? err : "");
        if (mg) MgDestroy(mg);
        DestroyBlas(blas); DestroySparse(sparse); DestroyMatDescr(descr);
        return st == MGCG_OK ? 0 : 2;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "mgcg_solve: %s\n", ex.what());
        return 1;
    }
}
