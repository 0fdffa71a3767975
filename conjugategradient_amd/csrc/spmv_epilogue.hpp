// Epilogue pieces shared by the lane = row SpMV kernels (kernels_rows.hip, kernels_rowtile.hip): the operands a row's
// epilogue needs are loaded early (behind the gathers), the value is formed after the row sum.
#pragma once
#include "common.hpp"

namespace mgcg {

struct RowsEpi { double w, b, dinv, yold; };

template <int EPI>
__device__ __forceinline__ RowsEpi rows_epi_prefetch(const SpmvArgs& a, long long row)
{
    RowsEpi o; o.w = 0.0; o.b = 0.0; o.dinv = 0.0; o.yold = 0.0;
    if constexpr (EPI == EPI_AXPBY_BETA) o.yold = a.y[row];
    if constexpr (EPI == EPI_DOT) o.w = a.w[row];
    if constexpr (EPI == EPI_RESIDUAL || EPI == EPI_RESIDUAL_DOT) o.b = a.b[row];
    if constexpr (EPI == EPI_JACOBI || EPI == EPI_JACOBI_DOT) { o.b = a.b[row]; o.dinv = a.dinvUniform ? a.dinvScalar : a.dinv[row]; o.w = a.w[row]; }
    return o;
}

template <int EPI>
__device__ __forceinline__ double rows_epilogue_value(const SpmvArgs& a, double acc, const RowsEpi& o, double& dotacc)
{
    if constexpr (EPI == EPI_AXPBY) return a.alpha * acc;
    else if constexpr (EPI == EPI_AXPBY_BETA) { double v = a.alpha * acc; double t = a.beta * o.yold; return v + t; }
    else if constexpr (EPI == EPI_DOT) { double t = o.w * acc; dotacc += t; return acc; }
    else if constexpr (EPI == EPI_RESIDUAL) return o.b - acc;
    else if constexpr (EPI == EPI_RESIDUAL_DOT) { double r = o.b - acc; double t = r * r; dotacc += t; return r; }
    else {
        double res = o.b - acc; double t = o.dinv * res; double s = a.omega * t; const double v = o.w + s;
        if constexpr (EPI == EPI_JACOBI_DOT) { double q = o.b * v; dotacc += q; }
        return v;
    }
}

// parent cell of fine cell c on a grid with power-of-two nx, ny (SpmvArgs::xScaled == 2)
// (c = [iz | iy | ix] as bit fields; the parent drops the lowest bit of every coarsened field: three shifts, one and, two and-or)
__device__ __forceinline__ int coarse_of(const SpmvArgs& a, int c)
{
    const unsigned u = (unsigned)c;
    return (int)(((u >> 1) & (unsigned)a.cM0) | ((u >> a.cS1) & (unsigned)a.cM1) | ((u >> a.cS2) & (unsigned)a.cM2));
}

// ---- tile order (TileMap, common.hpp): trips of workgroup `wg` of `nWG`, and the tile of its t-th trip
__host__ __device__ __forceinline__ int tile_map_trips(const TileMap& tm, int wg, int nWG, int nTiles)
{
    if (tm.mode == 0) return nTiles > wg ? (nTiles - wg + nWG - 1) / nWG : 0;
    if (tm.mode == 1) return tm.per * tm.nPlanes;
    const int eighth = tm.tilesPerPlane >> 3, sub = (wg >> 3) / eighth;
    return tm.nPlanes > sub ? (tm.nPlanes - sub + tm.per - 1) / tm.per : 0;
}
__host__ __device__ __forceinline__ int tile_map_tile(const TileMap& tm, int t, int wg, int nWG)
{
    if (tm.mode == 0) return wg + nWG * t;
    const int xcd = wg & 7, slot = wg >> 3;
    if (tm.mode == 1) { const int h = t / tm.nPlanes, p = t - h * tm.nPlanes; return p * tm.tilesPerPlane + h * nWG + xcd * (nWG >> 3) + slot; }
    const int eighth = tm.tilesPerPlane >> 3, sub = slot / eighth, j = slot - sub * eighth;
    return (t * tm.per + sub) * tm.tilesPerPlane + xcd * eighth + j;
}

} // namespace mgcg
