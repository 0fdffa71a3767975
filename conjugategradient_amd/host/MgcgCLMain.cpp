// Twin of the reference's two other driver programs on the C++ front-ends of MgcgFrontends.hpp:
//   Mgcg/HandmadeCL/MgcgCL/MgcgCLMain.cs:15-150  (ELL builder, max-norm rule, MIN_ITERATION = 50, 1e-4)
//   Mgcg/ViennaCL/MgcgCL/MgcgCL.cs:14-120        (dictionary builder, relative rule, 1e-4)
// Usage: MgcgCLMain [COUNT] [PREFIX]  -- prints "family iteration residual checksum" lines that tests/test_gpu_host_cpp.py parses; with PREFIX
// the two solutions also go to PREFIX.handmadecl.f64 / PREFIX.viennacl.f64 as raw doubles (element-by-element comparison, MgcgCLMain.cs:120-130).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "MgcgFrontends.hpp"

using namespace LWisteria;

int main(int argc, char** argv)
{
    const int COUNT = argc > 1 ? std::atoi(argv[1]) : 3456;
    const int MAX_NONZERO_COUNT = 160;
    const std::string prefix = argc > 2 ? argv[2] : "";
    auto writeX = [&](const char* which, const std::vector<double>& x) {
        if (prefix.empty()) return;
        const std::string path = prefix + "." + which + ".f64";
        FILE* f = std::fopen(path.c_str(), "wb");
        if (!f || std::fwrite(x.data(), sizeof(double), x.size(), f) != x.size()) { if (f) std::fclose(f); throw std::runtime_error("cannot write " + path); }
        std::fclose(f);
    };
    try {
        {   // HandmadeCL family
            MgcgCL::ConjugateGradientSingleGpu cg(COUNT, MAX_NONZERO_COUNT, 50, COUNT, 1e-4);
            for (int i = 0; i < COUNT; ++i) {
                cg.A.Set(i, i, 0);
                const int lo = std::max(0, i - MAX_NONZERO_COUNT / 2 + 1), hi = std::min(COUNT, i + MAX_NONZERO_COUNT / 2);
                for (int j = lo; j < hi; ++j)
                    if (i != j) { const double a = std::fabs(std::sin((double)(i + j))); cg.A.Set(i, j, a); cg.A.Add(i, i, a); }
                cg.b[(size_t)i] = std::cos((double)i) * 10;
                cg.x[(size_t)i] = (double)i / 100;
            }
            cg.Initialize();
            cg.Solve();
            cg.Read();
            double sum = 0;
            for (double v : cg.x) sum += v;
            std::printf("handmadecl %d %.17g %.17g\n", cg.Iteration, cg.Residual, sum);
            writeX("handmadecl", cg.x);
        }
        {   // ViennaCL family
            const int N = COUNT, BAND_WIDTH = 160;
            ViennaCL::CompressedMatrix A;
            std::vector<double> x((size_t)N, 0.0), b((size_t)N);
            for (int i = 0; i < N; ++i) {
                A.Set(i, i, i);
                for (int j = std::max(0, i - BAND_WIDTH / 2); j <= std::min(N - 1, i + BAND_WIDTH / 2); ++j)
                    if (i != j) { const double a = std::fabs(std::sin((double)(i + j))); A.Set(i, j, a); A.Add(i, i, a); }
                b[(size_t)i] = std::asin((double)i / N);
            }
            std::vector<double> elements; std::vector<unsigned> rowOffsets, columnIndeces;
            A.ToCsr(N, elements, rowOffsets, columnIndeces);
            ViennaCL::ComputerGpu gpu(N);
            gpu.Write(elements.data(), rowOffsets.data(), columnIndeces.data(), x.data(), b.data());
            gpu.Solve(1e-4, 0, N);
            gpu.Read(x.data());
            double sum = 0;
            for (double v : x) sum += v;
            std::printf("viennacl %d %.17g %.17g\n", gpu.Iteration(), 0.0, sum);
            writeX("viennacl", x);
        }
    } catch (const std::exception& e) {
        std::fprintf(stderr, "MgcgCLMain: %s\n", e.what());
        return 1;
    }
    return 0;
}
