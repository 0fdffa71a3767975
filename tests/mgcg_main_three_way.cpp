// TEST HARNESS (lives under tests/ because it links the CPU oracle, which only tests may do): the reference's own driver program
// Mgcg/cuBlas/Mgcg/MgcgMain.cs:41-178 in full -- CPU solver, single GPU, every GPU of the process -- with its element-by-element
// comparison (:129-162: every element of the two GPU answers against the CPU answer, 1 % relative) and its three "ticks per iteration" lines
// (:165-167).  The product's own twin, conjugategradient_amd/host/MgcgMain.cpp, is two-way by design: the library has no CPU compute path.
// Here the CPU leg is the class ConjugateGradientCpu of the reference (ConjugateGradientCpu.cs:9-100) restated on the test oracle.
//
//   mgcg_main_three_way [COUNT] [MIN_ITERATION] [write=PREFIX]
// Prints the reference's lines and one JSON line; PREFIX.cpu.f64 / .single.f64 / .parallel.f64 receive the three solutions as raw doubles.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>

#include "../conjugategradient_amd/host/Mgcg.hpp"

extern "C" int oracle_cg(const double* elements, const int* columnIndeces, const int* rowOffsets, int64_t count, double* x, const double* b,
                         int rule, double allowableResidual, int minIteration, int maxIteration, int64_t hardCap,
                         int* iteration, double* residual, double* trace, int64_t traceCap, double* work);

using namespace LWisteria::Mgcg;

// ConjugateGradientCpu.cs:9-100: same constructor, members and Solve(); the loop itself is oracle_cg (rule 0 = ConjugateGradient.cs:56-79)
class ConjugateGradientCpu : public ConjugateGradient {
public:
    using ConjugateGradient::ConjugateGradient;
    void Solve() override
    {
        int iteration = 0; double residual = 0;
        const int st = oracle_cg(A->Elements.data(), A->ColumnIndeces.data(), A->RowOffsets.data(), Count(), x.data(), b.data(),
                                 0, AllowableResidual, MinIteration, MaxIteration, (int64_t)MaxIteration + 2, &iteration, &residual, nullptr, 0, nullptr);
        Iteration = iteration; Residual = residual;
        if (st == 1) throw ApplicationException("the pressure equation did not converge");
    }
};

int main(int argc, char** argv)
{
    const int COUNT = argc > 1 ? atoi(argv[1]) : 34567 * 6;            // MgcgMain.cs:15
    const int MAX_NONZERO_COUNT = 160;                                  // :20
    const int MIN_ITERATION = argc > 2 ? atoi(argv[2]) : 200;           // :25
    const int MAX_ITERATION = COUNT;                                    // :30
    const double ALLOWABLE_RESIDUAL = 1e-8;                             // :35
    std::string writePrefix;
    for (int i = 3; i < argc; i++) if (std::string(argv[i]).rfind("write=", 0) == 0) writePrefix = std::string(argv[i]).substr(6);
    auto writeX = [&](const char* which, const std::vector<double>& x) {
        if (writePrefix.empty()) return;
        const std::string path = writePrefix + "." + which + ".f64";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f || fwrite(x.data(), sizeof(double), x.size(), f) != x.size()) { if (f) fclose(f); throw MgcgError("cannot write " + path); }
        fclose(f);
    };
    printf("N=%d\n", COUNT);
    try {
        ConjugateGradientCpu cgCpu(COUNT, MAX_NONZERO_COUNT, MIN_ITERATION, MAX_ITERATION, ALLOWABLE_RESIDUAL);                   // :44
        ConjugateGradientSingleGpu cgGpuSingle(COUNT, MAX_NONZERO_COUNT, MIN_ITERATION, MAX_ITERATION, ALLOWABLE_RESIDUAL);       // :45
        ConjugateGradientParallelGpu cgGpuParallel(COUNT, MAX_NONZERO_COUNT, MIN_ITERATION, MAX_ITERATION, ALLOWABLE_RESIDUAL);   // :46
        SparseMatrix A(COUNT, MAX_NONZERO_COUNT);
        A.RowOffsets[0] = 0;
        for (int i = 0; i < COUNT; i++) {                               // :53-84
            const int rowOffset = A.RowOffsets[(size_t)i];
            A.Elements[(size_t)rowOffset] = 0;
            A.ColumnIndeces[(size_t)rowOffset] = i;
            int nonzeroCount = 1;
            for (int j = std::max(0, i - MAX_NONZERO_COUNT / 2 + 1); j < std::min(COUNT, i + MAX_NONZERO_COUNT / 2); j++) {
                if (i != j) {
                    const double a_ij = std::fabs(std::sin((double)(i + j)));
                    A.Elements[(size_t)(rowOffset + nonzeroCount)] = a_ij;
                    A.ColumnIndeces[(size_t)(rowOffset + nonzeroCount)] = j;
                    nonzeroCount++;
                    A.Elements[(size_t)rowOffset] += a_ij;
                }
            }
            A.RowOffsets[(size_t)i + 1] = A.RowOffsets[(size_t)i] + nonzeroCount;
        }
        cgCpu.A = &A; cgGpuSingle.A = &A; cgGpuParallel.A = &A;
        for (int i = 0; i < COUNT; i++) {                               // :91-104
            const double b_i = std::cos((double)i) * 10, x_i = (double)i / 100;
            cgCpu.b[(size_t)i] = b_i; cgGpuSingle.b[(size_t)i] = b_i; cgGpuParallel.b[(size_t)i] = b_i;
            cgCpu.x[(size_t)i] = x_i; cgGpuSingle.x[(size_t)i] = x_i; cgGpuParallel.x[(size_t)i] = x_i;
        }
        printf("start\n");
        using clk = std::chrono::steady_clock;
        auto seconds = [](clk::time_point t0) { return std::chrono::duration<double>(clk::now() - t0).count(); };

        auto t0 = clk::now();                                           // :113-119
        cgCpu.Solve();
        const double cpuSec = seconds(t0);

        cgGpuSingle.Initialize();                                       // :121-127 (Solve() alone is timed)
        t0 = clk::now();
        cgGpuSingle.Solve();
        const double singleSec = seconds(t0);
        cgGpuSingle.Read();

        int mismatchSingle = 0, mismatchParallel = 0;
        double maxRelSingle = 0, maxRelParallel = 0;
        for (int i = 0; i < COUNT; i++) {                               // :129-140
            const double residual = std::fabs(cgCpu.x[(size_t)i] - cgGpuSingle.x[(size_t)i]);
            const double rel = residual / cgCpu.x[(size_t)i];
            if (std::fabs(cgCpu.x[(size_t)i]) > 0) maxRelSingle = std::max(maxRelSingle, std::fabs(rel));
            if (rel > 0.01) { if (mismatchSingle < 10) printf("Single %4d: %e (%e vs %e)\n", i, residual, cgCpu.x[(size_t)i], cgGpuSingle.x[(size_t)i]); mismatchSingle++; }
        }

        cgGpuParallel.Initialize();                                     // :143-149
        t0 = clk::now();
        cgGpuParallel.Solve();
        const double parallelSec = seconds(t0);
        cgGpuParallel.Read();
        for (int i = 0; i < COUNT; i++) {                               // :151-162
            const double residual = std::fabs(cgCpu.x[(size_t)i] - cgGpuParallel.x[(size_t)i]);
            const double rel = residual / cgCpu.x[(size_t)i];
            if (std::fabs(cgCpu.x[(size_t)i]) > 0) maxRelParallel = std::max(maxRelParallel, std::fabs(rel));
            if (rel > 0.01) { if (mismatchParallel < 10) printf("Parallel %4d: %e (%e vs %e)\n", i, residual, cgCpu.x[(size_t)i], cgGpuParallel.x[(size_t)i]); mismatchParallel++; }
        }
        writeX("cpu", cgCpu.x); writeX("single", cgGpuSingle.x); writeX("parallel", cgGpuParallel.x);
        // :165-167 (the reference prints ticks / iteration; here microseconds per iteration)
        printf("CPU         : %12.6f s / %d = %12.3f us per iteration\n", cpuSec, cgCpu.Iteration, 1e6 * cpuSec / std::max(1, cgCpu.Iteration));
        printf("single GPU  : %12.6f s / %d = %12.3f us per iteration\n", singleSec, cgGpuSingle.Iteration, 1e6 * singleSec / std::max(1, cgGpuSingle.Iteration));
        printf("parallel GPU: %12.6f s / %d = %12.3f us per iteration (%d devices) -- %s\n", parallelSec, cgGpuParallel.Iteration,
               1e6 * parallelSec / std::max(1, cgGpuParallel.Iteration), cgGpuParallel.DeviceCount(), cgGpuParallel.LastPath.c_str());
        printf("{\"count\": %d, \"devices\": %d, \"iteration_cpu\": %d, \"iteration_single\": %d, \"iteration_parallel\": %d, \"residual_cpu\": %.17g, \"residual_single\": %.17g, "
               "\"residual_parallel\": %.17g, \"mismatches_single\": %d, \"mismatches_parallel\": %d, \"max_rel_cpu_vs_single\": %.3e, \"max_rel_cpu_vs_parallel\": %.3e, "
               "\"us_per_iteration_cpu\": %.3f, \"us_per_iteration_single\": %.3f, \"us_per_iteration_parallel\": %.3f, \"parallel_path\": \"%s\"}\n",
               COUNT, cgGpuParallel.DeviceCount(), cgCpu.Iteration, cgGpuSingle.Iteration, cgGpuParallel.Iteration, cgCpu.Residual, cgGpuSingle.Residual, cgGpuParallel.Residual,
               mismatchSingle, mismatchParallel, maxRelSingle, maxRelParallel, 1e6 * cpuSec / std::max(1, cgCpu.Iteration), 1e6 * singleSec / std::max(1, cgGpuSingle.Iteration),
               1e6 * parallelSec / std::max(1, cgGpuParallel.Iteration), cgGpuParallel.LastPath.c_str());
        return (mismatchSingle == 0 && mismatchParallel == 0) ? 0 : 1;
    } catch (std::exception& e) {
        printf("!!!!%s\n", e.what());
        return 2;
    }
}
