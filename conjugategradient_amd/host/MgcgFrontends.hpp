// C++ host twins of the reference's two other solver families on the same C ABI (no HIP headers needed):
//   MgcgCL::SparseMatrix / ConjugateGradientSingleGpu  -- Mgcg/HandmadeCL/MgcgCL/SparseMatrix.cs:8-230,
//                                                         ConjugateGradientSingleGpu.cs (max-norm residual, :268)
//   ViennaCL::CompressedMatrix / ComputerGpu            -- Mgcg/ViennaCL/MgcgCL/CompressedMatrix.cs:8-70,
//                                                         Mgcg/ViennaCL/Mgcg/ComputerGpu.hpp:68-100, ComputerGpu.cpp:18-95
// Same member names, argument meaning and error behaviour; the arithmetic is SolveEx with the matching stop rule.
#ifndef MGCG_FRONTENDS_HPP
#define MGCG_FRONTENDS_HPP
#include "Mgcg.hpp"
#include <map>
#include <unordered_map>

namespace LWisteria {

namespace MgcgCL {

// Slot-0-diagonal ELL builder (SparseMatrix.cs): MaxNonzeroCountPerRow slots per row, slot 0 is the diagonal and always
// present, further entries are appended in the order of their first assignment.
class SparseMatrix {
public:
    std::vector<double> Elements;
    std::vector<int> ColumnIndeces;
    std::vector<int> NonzeroCounts;
    const int MaxNonzeroCountPerRow;

    SparseMatrix(int rowCount, int maxNonzeroCountPerRow)
        : Elements((size_t)rowCount * maxNonzeroCountPerRow), ColumnIndeces((size_t)rowCount * maxNonzeroCountPerRow),
          NonzeroCounts((size_t)rowCount), MaxNonzeroCountPerRow(maxNonzeroCountPerRow) { Clear(); }

    void Clear()                                                   // SparseMatrix.cs:52-63
    {
        for (size_t i = 0; i < NonzeroCounts.size(); ++i) { NonzeroCounts[i] = 1; Set(i, i, 0.0); }
    }
    int RowCount() const { return (int)NonzeroCounts.size(); }

    double Get(size_t i, size_t j) const                           // this[i, j] get (:96-118)
    {
        if (i == j) return Elements[i * MaxNonzeroCountPerRow];
        const int k = LocalIndex(i, j);
        return k >= 0 ? Elements[i * MaxNonzeroCountPerRow + k] : 0.0;
    }
    void Set(size_t i, size_t j, double value)                     // this[i, j] set (:119-152)
    {
        const size_t first = i * MaxNonzeroCountPerRow;
        if (i == j) { Elements[first] = value; ColumnIndeces[first] = (int)i; return; }
        int k = LocalIndex(i, j);
        if (k < 0) {
            if (NonzeroCounts[i] == MaxNonzeroCountPerRow) throw std::out_of_range("row " + std::to_string(i) + " is full");   // IndexOutOfRangeException (:123)
            k = NonzeroCounts[i]++;
            ColumnIndeces[first + k] = (int)j;
        }
        Elements[first + k] = value;
    }
    void Add(size_t i, size_t j, double value) { Set(i, j, Get(i, j) + value); }   // A[i, j] += v

    // rows packed in stored order (diagonal first): what the library's CSR entry points take
    void ToCsr(std::vector<double>& elements, std::vector<int>& columnIndeces, std::vector<int>& rowOffsets) const
    {
        const size_t n = NonzeroCounts.size();
        rowOffsets.assign(n + 1, 0);
        for (size_t i = 0; i < n; ++i) rowOffsets[i + 1] = rowOffsets[i] + NonzeroCounts[i];
        elements.resize((size_t)rowOffsets[n]); columnIndeces.resize((size_t)rowOffsets[n]);
        for (size_t i = 0; i < n; ++i)
            for (int k = 0; k < NonzeroCounts[i]; ++k) {
                elements[(size_t)rowOffsets[i] + k] = Elements[i * MaxNonzeroCountPerRow + k];
                columnIndeces[(size_t)rowOffsets[i] + k] = ColumnIndeces[i * MaxNonzeroCountPerRow + k];
            }
    }

private:
    int LocalIndex(size_t i, size_t j) const                       // GetLocalIndex (:163-181): slot 0 is never matched
    {
        const size_t first = i * MaxNonzeroCountPerRow;
        for (int k = 1; k < NonzeroCounts[i]; ++k) if (ColumnIndeces[first + k] == (int)j) return k;
        return -1;
    }
};

// ConjugateGradientSingleGpu of the HandmadeCL family: Residual is max|r_i|, IsConverged as in ConjugateGradient.cs:56-79.
class ConjugateGradientSingleGpu {
public:
    SparseMatrix A;
    std::vector<double> x, b;
    int MinIteration, MaxIteration;
    double AllowableResidual;
    int Iteration = 0;
    double Residual = 0;

    ConjugateGradientSingleGpu(int count, int maxNonZeroCount, int minIteration, int maxIteration, double allowableResidual)
        : A(count, maxNonZeroCount), x((size_t)count), b((size_t)count), MinIteration(minIteration), MaxIteration(maxIteration),
          AllowableResidual(allowableResidual), cublas(CreateBlas()), cusparse(CreateSparse()), matDescr(CreateMatDescr()),
          vectorX(count), vectorB(count), vectorAp(count), vectorP(count), vectorR(count)
    {
        Mgcg::Check("ConjugateGradientSingleGpu");
    }
    ~ConjugateGradientSingleGpu() { DestroyBlas(cublas); DestroySparse(cusparse); DestroyMatDescr(matDescr); }
    int Count() const { return (int)x.size(); }

    void Initialize()
    {
        std::vector<double> e; std::vector<int> c, ro;
        A.ToCsr(e, c, ro);
        nnz = ro[(size_t)Count()];
        vectorA.reset(new Mgcg::VectorDouble(nnz > 0 ? nnz : 1));
        vectorColumnIndeces.reset(new Mgcg::VectorInt(nnz > 0 ? nnz : 1));
        vectorRowOffsets.reset(new Mgcg::VectorInt(Count() + 1));
        vectorA->CopyFrom(e.data(), nnz);
        vectorColumnIndeces->CopyFrom(c.data(), nnz);
        vectorRowOffsets->CopyFrom(ro.data(), Count() + 1);
        vectorX.CopyFrom(x.data(), Count());
        vectorB.CopyFrom(b.data(), Count());
    }
    void Solve()
    {
        const int st = SolveEx(cublas, cusparse, matDescr, vectorA->Ptr, vectorRowOffsets->Ptr, vectorColumnIndeces->Ptr,
                               vectorX.Ptr, vectorB.Ptr, vectorAp.Ptr, vectorP.Ptr, vectorR.Ptr, nnz, Count(),
                               AllowableResidual, MinIteration, MaxIteration, MGCG_RULE_HANDMADECL, &Iteration, &Residual, nullptr, 0);
        if (st == MGCG_MAXIT_EXCEEDED) { MgcgClearLastError(); throw Mgcg::ApplicationException("the pressure equation did not converge"); }
        if (st != MGCG_OK) Mgcg::Check("SolveEx");
    }
    void Read() { vectorX.CopyTo(x.data(), Count()); }

private:
    MgcgBlas* cublas; MgcgSparse* cusparse; MgcgMatDescr* matDescr;
    Mgcg::VectorDouble vectorX, vectorB, vectorAp, vectorP, vectorR;
    std::unique_ptr<Mgcg::VectorDouble> vectorA;
    std::unique_ptr<Mgcg::VectorInt> vectorColumnIndeces, vectorRowOffsets;
    int nnz = 0;
};

} // namespace MgcgCL

namespace ViennaCL {

// The driver's dictionary-of-rows matrix (CompressedMatrix.cs): rows appear on first assignment, entries keep their
// insertion order (what C#'s Dictionary enumerates when nothing is removed), keys are (uint) column ids.
class CompressedMatrix {
public:
    struct Row { std::vector<unsigned> Keys; std::vector<double> Values; std::unordered_map<unsigned, size_t> Slot; };
    std::vector<Row> Elements;

    double Get(int i, int j) const
    {
        if ((size_t)i >= Elements.size()) return 0.0;
        auto it = Elements[(size_t)i].Slot.find((unsigned)j);
        return it == Elements[(size_t)i].Slot.end() ? 0.0 : Elements[(size_t)i].Values[it->second];
    }
    void Set(int i, int j, double value)
    {
        while ((size_t)i >= Elements.size()) Elements.emplace_back();
        Row& r = Elements[(size_t)i];
        auto it = r.Slot.find((unsigned)j);
        if (it != r.Slot.end()) { r.Values[it->second] = value; return; }
        r.Slot.emplace((unsigned)j, r.Keys.size()); r.Keys.push_back((unsigned)j); r.Values.push_back(value);
    }
    void Add(int i, int j, double value) { Set(i, j, Get(i, j) + value); }

    // MgcgCL.cs:85-97: keys and values of every row in dictionary order, unsigned offsets and column ids
    void ToCsr(int n, std::vector<double>& elements, std::vector<unsigned>& rowOffsets, std::vector<unsigned>& columnIndeces) const
    {
        rowOffsets.assign((size_t)n + 1, 0u); elements.clear(); columnIndeces.clear();
        for (int i = 0; i < n; ++i) {
            if ((size_t)i < Elements.size()) {
                elements.insert(elements.end(), Elements[(size_t)i].Values.begin(), Elements[(size_t)i].Values.end());
                columnIndeces.insert(columnIndeces.end(), Elements[(size_t)i].Keys.begin(), Elements[(size_t)i].Keys.end());
            }
            rowOffsets[(size_t)i + 1] = (unsigned)elements.size();
        }
    }
};

// ComputerGpu.hpp:68-100: Write / Solve(residual, min, max) / Read / Iteration; stop rule
// minIteration < it && rrNew/rr0 < residual^2 (ComputerGpu.cpp:78).
class ComputerGpu {
public:
    explicit ComputerGpu(int n)
        : count(n), cublas(CreateBlas()), cusparse(CreateSparse()), matDescr(CreateMatDescr()), vx(n), vb(n), vAp(n), vp(n), vr(n)
    {
        Mgcg::Check("ComputerGpu");
    }
    ~ComputerGpu() { DestroyBlas(cublas); DestroySparse(cusparse); DestroyMatDescr(matDescr); }

    void Write(const double elements[], const unsigned rowOffsets[], const unsigned columnIndeces[], const double x[], const double b[])
    {
        nnz = (int)rowOffsets[count];
        std::vector<int> ro((size_t)count + 1), ci((size_t)nnz);
        for (int i = 0; i <= count; ++i) ro[(size_t)i] = (int)rowOffsets[i];
        for (int k = 0; k < nnz; ++k) ci[(size_t)k] = (int)columnIndeces[k];
        vE.reset(new Mgcg::VectorDouble(nnz > 0 ? nnz : 1)); vC.reset(new Mgcg::VectorInt(nnz > 0 ? nnz : 1)); vRO.reset(new Mgcg::VectorInt(count + 1));
        vE->CopyFrom(elements, nnz); vC->CopyFrom(ci.data(), nnz); vRO->CopyFrom(ro.data(), count + 1);
        vx.CopyFrom(x, count); vb.CopyFrom(b, count);
    }
    void Solve(double residual, int minIteration, int maxIteration)
    {
        int it = 0; double rel = 0;
        const int st = SolveEx(cublas, cusparse, matDescr, vE->Ptr, vRO->Ptr, vC->Ptr, vx.Ptr, vb.Ptr, vAp.Ptr, vp.Ptr, vr.Ptr, nnz, count,
                               residual, minIteration, maxIteration, MGCG_RULE_VIENNACL, &it, &rel, nullptr, 0);
        iteration = it + 1;                                        // the post-incremented loop counter (ComputerGpu.cpp:66)
        if (st != MGCG_OK) Mgcg::Check("SolveEx");
    }
    void Read(double xOut[]) { vx.CopyTo(xOut, count); }
    int Iteration() const { return iteration; }

private:
    int count, nnz = 0, iteration = 0;
    MgcgBlas* cublas; MgcgSparse* cusparse; MgcgMatDescr* matDescr;
    Mgcg::VectorDouble vx, vb, vAp, vp, vr;
    std::unique_ptr<Mgcg::VectorDouble> vE;
    std::unique_ptr<Mgcg::VectorInt> vC, vRO;
};

} // namespace ViennaCL
} // namespace LWisteria
#endif
