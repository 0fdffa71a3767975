// MgcgMain -- C++ twin of the reference driver Mgcg/cuBlas/Mgcg/MgcgMain.cs:41-178: builds the banded
// |sin(i+j)| system (:51-104), solves it on one GPU and on every GPU of the process, and reports
// ticks per iteration (:165-167).  The reference also runs its CPU solver and prints every element that
// differs by more than 1 % (:129-162); the product has no CPU compute path, so the two GPU paths are
// compared with each other here and tests/test_gpu_host_cpp.py compares the printed solution
// checksum with the CPU oracle.
//
//   MgcgMain [COUNT] [MIN_ITERATION] [phases] [balance] [write=PREFIX]      (defaults: 34567*6 and 200, the reference's constants)
// write=PREFIX: the three solutions as raw doubles in PREFIX.single.f64 / PREFIX.phases.f64 / PREFIX.parallel.f64, so that a test can do what
// the reference's driver does with its CPU leg (:129-162): compare element by element.
// The multi-device solver runs twice: on the reference's host-driven phases (Solve0..3, P2Host / P2Device) and on the native
// loop (SolveParallel on every device's own thread, collectives on the device streams); which path produced the kept answer
// is printed ("phases" as third argument keeps the phase structure for it).
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <string>
#include <vector>

#include "Mgcg.hpp"

using namespace LWisteria::Mgcg;

int main(int argc, char** argv)
{
    const int COUNT = argc > 1 ? atoi(argv[1]) : 34567 * 6;            // MgcgMain.cs:15
    const int MAX_NONZERO_COUNT = 160;                                  // :20
    const int MIN_ITERATION = argc > 2 ? atoi(argv[2]) : 200;           // :25
    const int MAX_ITERATION = COUNT;                                    // :30
    const double ALLOWABLE_RESIDUAL = 1e-8;                             // :35
    bool usePhasesOnly = false, balance = false;
    std::string writePrefix;
    auto writeX = [&](const char* which, const std::vector<double>& x) {
        if (writePrefix.empty()) return;
        const std::string path = writePrefix + "." + which + ".f64";
        FILE* f = fopen(path.c_str(), "wb");
        if (!f || fwrite(x.data(), sizeof(double), x.size(), f) != x.size()) { if (f) fclose(f); throw MgcgError("cannot write " + path); }
        fclose(f);
    };
    for (int i = 3; i < argc; i++) {
        if (std::string(argv[i]).rfind("write=", 0) == 0) writePrefix = std::string(argv[i]).substr(6);
        if (std::string(argv[i]) == "phases") usePhasesOnly = true;           // keep the reference's phase structure for the kept answer too
        if (std::string(argv[i]) == "balance") balance = true;                // row ranges of equal nonzero count (BalanceNonzeros) instead of equal row count
    }
    printf("N=%d\n", COUNT);
    try {
        ConjugateGradientSingleGpu cgGpuSingle(COUNT, MAX_NONZERO_COUNT, MIN_ITERATION, MAX_ITERATION, ALLOWABLE_RESIDUAL);
        ConjugateGradientParallelGpu cgGpuParallel(COUNT, MAX_NONZERO_COUNT, MIN_ITERATION, MAX_ITERATION, ALLOWABLE_RESIDUAL);

        SparseMatrix A(COUNT, MAX_NONZERO_COUNT);
        A.RowOffsets[0] = 0;
        for (int i = 0; i < COUNT; i++) {                               // :53-84
            const int rowOffset = A.RowOffsets[(size_t)i];
            A.Elements[(size_t)rowOffset] = 0;
            A.ColumnIndeces[(size_t)rowOffset] = i;                     // diagonal first
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
        cgGpuSingle.A = &A;
        cgGpuParallel.A = &A;
        cgGpuParallel.BalanceNonzeros = balance;
        for (int i = 0; i < COUNT; i++) {                               // :91-104
            const double b_i = std::cos((double)i) * 10;
            const double x_i = (double)i / 100;
            cgGpuSingle.b[(size_t)i] = b_i; cgGpuParallel.b[(size_t)i] = b_i;
            cgGpuSingle.x[(size_t)i] = x_i; cgGpuParallel.x[(size_t)i] = x_i;
        }
        printf("start\n");
        using clk = std::chrono::steady_clock;

        cgGpuSingle.Initialize();
        auto t0 = clk::now();
        cgGpuSingle.Solve();
        const double singleSec = std::chrono::duration<double>(clk::now() - t0).count();
        cgGpuSingle.Read();
        const std::vector<double> xSingle = cgGpuSingle.x;
        const int singleIteration = cgGpuSingle.Iteration;
        // the reference times ONE cold Solve() (:121-126); the same solve once more in this process says what of that was a one-off
        for (int i = 0; i < COUNT; i++) cgGpuSingle.x[(size_t)i] = (double)i / 100;
        cgGpuSingle.Initialize();
        t0 = clk::now();
        cgGpuSingle.Solve();
        const double singleSecondSec = std::chrono::duration<double>(clk::now() - t0).count();
        cgGpuSingle.Read();
        if (cgGpuSingle.Iteration != singleIteration || cgGpuSingle.x != xSingle) { printf("!!!!the second single-GPU solve differs from the first\n"); return 2; }

        // the reference's own phase structure first (Solve0..3 + host-staged SyncP), then -- the answer that is kept -- the native loop
        cgGpuParallel.UsePhases = true;
        cgGpuParallel.Initialize();
        t0 = clk::now();
        cgGpuParallel.Solve();
        const double phasesSec = std::chrono::duration<double>(clk::now() - t0).count();
        const int phasesIteration = cgGpuParallel.Iteration;
        cgGpuParallel.Read();
        const std::vector<double> xPhases = cgGpuParallel.x;
        for (int i = 0; i < COUNT; i++) cgGpuParallel.x[(size_t)i] = (double)i / 100;
        cgGpuParallel.UsePhases = usePhasesOnly;
        cgGpuParallel.Initialize();
        t0 = clk::now();
        cgGpuParallel.Solve();
        const double parallelSec = std::chrono::duration<double>(clk::now() - t0).count();
        cgGpuParallel.Read();
        double maxRelPhases = 0;
        for (int i = 0; i < COUNT; i++) if (std::fabs(xPhases[(size_t)i]) > 0) maxRelPhases = std::max(maxRelPhases, std::fabs(xPhases[(size_t)i] - cgGpuParallel.x[(size_t)i]) / std::fabs(xPhases[(size_t)i]));

        writeX("single", cgGpuSingle.x); writeX("phases", xPhases); writeX("parallel", cgGpuParallel.x);
        int mismatches = 0;
        double checksum = 0, maxRel = 0;
        for (int i = 0; i < COUNT; i++) {                               // :151-162 with the single-GPU result as the baseline
            const double ref = cgGpuSingle.x[(size_t)i];
            const double residual = std::fabs(ref - cgGpuParallel.x[(size_t)i]);
            if (std::fabs(ref) > 0) maxRel = std::max(maxRel, residual / std::fabs(ref));
            if (residual / ref > 0.01) { if (mismatches < 10) printf("Parallel %4d: %e (%e vs %e)\n", i, residual, ref, cgGpuParallel.x[(size_t)i]); mismatches++; }
            checksum += ref * (double)((i % 7) + 1);
        }
        printf("single GPU  : %12.6f s / %d = %12.3f us per iteration (the first Solve() of the process; the same solve again: %.3f us per iteration)\n", singleSec, cgGpuSingle.Iteration,
               1e6 * singleSec / std::max(1, cgGpuSingle.Iteration), 1e6 * singleSecondSec / std::max(1, cgGpuSingle.Iteration));
        printf("parallel GPU: %12.6f s / %d = %12.3f us per iteration (%d devices) -- %s\n", parallelSec, cgGpuParallel.Iteration,
               1e6 * parallelSec / std::max(1, cgGpuParallel.Iteration), cgGpuParallel.DeviceCount(), cgGpuParallel.LastPath.c_str());
        printf("   (phases) : %12.6f s / %d = %12.3f us per iteration -- host-driven phases (Solve0..3), the reference's structure\n", phasesSec, phasesIteration,
               1e6 * phasesSec / std::max(1, phasesIteration));
        std::string offsets = "[";
        for (int d = 0; d <= cgGpuParallel.DeviceCount(); d++) offsets += (d ? ", " : "") + std::to_string(cgGpuParallel.OffsetForDevice(d));
        offsets += "]";
        printf("{\"offsets\": %s, \"count\": %d, \"devices\": %d, \"iteration_single\": %d, \"iteration_parallel\": %d, \"iteration_phases\": %d, \"residual_single\": %.17g, "
               "\"residual_parallel\": %.17g, \"mismatches\": %d, \"max_rel_single_vs_parallel\": %.3e, \"max_rel_phases_vs_parallel\": %.3e, \"checksum\": %.17g, \"x0\": %.17g, \"xlast\": %.17g, "
               "\"parallel_path\": \"%s\", \"us_per_iteration_single\": %.3f, \"us_per_iteration_single_second_solve\": %.3f, \"us_per_iteration_parallel\": %.3f, \"us_per_iteration_phases\": %.3f}\n",
               offsets.c_str(), COUNT, cgGpuParallel.DeviceCount(), cgGpuSingle.Iteration, cgGpuParallel.Iteration, phasesIteration, cgGpuSingle.Residual, cgGpuParallel.Residual,
               mismatches, maxRel, maxRelPhases, checksum, cgGpuSingle.x[0], cgGpuSingle.x[(size_t)COUNT - 1],
               cgGpuParallel.LastPath.c_str(), 1e6 * singleSec / std::max(1, cgGpuSingle.Iteration), 1e6 * singleSecondSec / std::max(1, cgGpuSingle.Iteration),
               1e6 * parallelSec / std::max(1, cgGpuParallel.Iteration),
               1e6 * phasesSec / std::max(1, phasesIteration));
        return mismatches == 0 ? 0 : 1;
    } catch (std::exception& e) {
        printf("!!!!%s\n", e.what());
        return 2;
    }
}
