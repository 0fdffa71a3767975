/*
 * MgcgGpu.h -- C ABI of libMgcgGpu.so, the MI355X (gfx950) replacement for the
 * reference's native library MgcgGpu.dll (the .cu files under Mgcg/cuBlas/MgcgGpu/), which the
 * C# solver classes bind with [DllImport(MgcgGpu.DLL_NAME, EntryPoint = "...")]
 * (Mgcg/cuBlas/Mgcg/MgcgGpu.cs:11).
 *
 * Part 1 declares the reference's 32 exports with identical names, argument
 * order and meaning (each cites the reference definition it replaces).  The
 * reference's `int&`/`double&` out-parameters are pointers here: identical at
 * the binary level on x86-64, which is what P/Invoke `out int` relies on.
 * `_stdcall` is a no-op on x64.
 *
 * Part 2 is additive (no reference analogue): error reporting, fused ops, the
 * device problem generator, the multigrid preconditioner and the
 * one-process-per-GPU RCCL solver.
 *
 * Conventions: plain pointers and sizes only; every function is safe to call
 * concurrently from different host threads working on different devices; the
 * current device is per host thread (SetDevice); every call is complete (its
 * results visible to the host / to later calls) when it returns a value, and
 * stream-ordered on the device's single stream otherwise (same contract as
 * cuBLAS/cuSPARSE on the default stream).  Errors never abort: the call
 * returns 0 / NaN / NULL, and MgcgGetLastError() holds a message.
 */
#ifndef MGCG_GPU_H
#define MGCG_GPU_H

#ifdef __cplusplus
extern "C" {
#endif

/* Opaque handles.  In the reference these are heap pointers to the vendor
 * handles (cublasHandle_t*, cusparseHandle_t*, cusparseMatDescr_t*,
 * thrust::device_vector<T>*); here they are heap pointers to library-owned
 * structs.  Caller owns them and frees them with the matching Destroy/Delete. */
typedef struct MgcgBlas     MgcgBlas;      /* stream + reduction workspace           */
typedef struct MgcgSparse   MgcgSparse;    /* stream + SpMV launch state             */
typedef struct MgcgMatDescr MgcgMatDescr;  /* general matrix, index base 0           */
typedef struct Vector       Vector;        /* device double[]  (zero-initialised)    */
typedef struct VectorInt    VectorInt;     /* device int[]     (zero-initialised)    */

/* ===================================================================== */
/* Part 1: the reference's exports                                        */
/* ===================================================================== */

/* ---- Runtime.cu ---- */
int           GetDeviceCount(void);                                  /* Runtime.cu:7  */
void          SetDevice(int deviceID);                               /* Runtime.cu:15 */
MgcgBlas*     CreateBlas(void);                                      /* Runtime.cu:20 */
void          DestroyBlas(MgcgBlas* cublas);                         /* Runtime.cu:28 */
MgcgSparse*   CreateSparse(void);                                    /* Runtime.cu:34 */
void          DestroySparse(MgcgSparse* cusparse);                   /* Runtime.cu:42 */
MgcgMatDescr* CreateMatDescr(void);                                  /* Runtime.cu:48 */
void          DestroyMatDescr(MgcgMatDescr* matDescr);               /* Runtime.cu:58 */

/* ---- Vector_Double.cu ---- (counts and offsets in ELEMENTS) */
Vector* Create_Double(int size);                                                       /* :7  */
void    CopyToArray_Double(const Vector* source, double destination[],
                           int count, int sourceOffset, int destinationOffset);        /* :15 */
void    CopyFromArray_Double(Vector* destination, const double source[],
                             int count, int sourceOffset, int destinationOffset);      /* :22 */
void    Delete_Double(Vector* vec);                                                    /* :29 */
double* ToRawPtr_Double(Vector* vec);                                                  /* :35 */
/* Vector_Double.cu:41.  The reference passes `count` as the BYTE count to
 * cudaMemcpy (missing *sizeof(double)); here count is in elements, as the
 * name and every other export imply. */
void    CopyFromDevice_Double(const double* source, double* destination,
                              int count, int sourceOffset, int destinationOffset);

/* ---- Vector_Int.cu ---- */
VectorInt* Create_Int(int size);                                                       /* :9  */
void       CopyToArray_Int(const VectorInt* source, int destination[],
                           int count, int sourceOffset, int destinationOffset);        /* :17 */
void       CopyFromArray_Int(VectorInt* destination, int source[],
                             int count, int sourceOffset, int destinationOffset);      /* :24 */
void       Delete_Int(VectorInt* vec);                                                 /* :31 */
int*       ToRawPtr_Int(VectorInt* vec);                                               /* :37 */

/* ---- Mgcg.cu: ops on RAW DEVICE POINTERS ---- */
/* y = alpha*A*x + beta*y, A general CSR, 0-based int32 (Mgcg.cu:10-19, was cusparseDcsrmv_v2).
 * Columns may be unsorted within a row.  beta == 0 never reads y. */
void   CsrMV(MgcgSparse* cusparse, MgcgMatDescr* matDescr,
             double* y,
             const double* elements, const int* rowOffsets, const int* columnIndeces,
             const double* x,
             int elementsCount, int rowCount, int columnCount,
             double alpha, double beta);
void   Axpy(MgcgBlas* cublas, double* y, const double* x, int count, double alpha);    /* y += alpha x   Mgcg.cu:22 */
double Dot(MgcgBlas* cublas, double* y, const double* x, int count);                   /* sum x_i y_i    Mgcg.cu:30 */
void   Scal(MgcgBlas* cublas, double* x, double alpha, int count);                     /* x *= alpha     Mgcg.cu:41 */
void   Copy(MgcgBlas* cublas, double* y, const double* x,
            int count, int yOffset, int xOffset);                                      /* y[yOff..] = x[xOff..]  Mgcg.cu:49 */

/* ---- Mgcg.cu: multi-device partition set-up and halo staging (HOST arrays in) ---- */
/* Mgcg.cu:57-85: upload rows [offsetForDevice, +countForDevice) of A, rebase the
 * row offsets by elementOffsetForDevice, upload the x and b slices, seed
 * p[offsetForDevice..] = x slice, return min/max column id of the slice. */
void Initialize(const double elements[], const int rowOffsets[], const int columnIndeces[],
                const double x[], const double b[],
                Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                Vector* xVector, Vector* bVector,
                Vector* pVector,
                int* minJ, int* maxJ,
                int count,
                int countForDevice, int offsetForDevice, int elementCountForDevice, int elementOffsetForDevice);
/* Mgcg.cu:88-99: publish the first lastCount and last nextCount entries of this
 * device's slice of p into the shared host array p[]. */
void P2Host(Vector* pVector, double p[], int thisCount, int thisOffset, int lastCount, int nextCount);
/* Mgcg.cu:102-113: pull the lastCount entries before and nextCount entries after
 * this device's slice from the shared host array into pVector. */
void P2Device(Vector* pVector, double p[], int thisCount, int thisOffset, int lastCount, int nextCount);

/* ---- Mgcg.cu: the CG loop split at its three global-scalar dependencies ---- */
/* Mgcg.cu:116-142: Ap = A_loc p; r = b - Ap; p[offset..] = r; returns r.r (local) */
double Solve0(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
              Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
              Vector* xVector, Vector* bVector,
              Vector* ApVector, Vector* pVector, Vector* rVector,
              int count,
              int countForDevice, int offsetForDevice, int elementsCountForDevice);
/* Mgcg.cu:145-163: Ap = A_loc p; returns p_loc.Ap (local) */
double Solve1(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
              Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
              Vector* ApVector, Vector* pVector,
              int count,
              int countForDevice, int offsetForDevice, int elementsCountForDevice);
/* Mgcg.cu:166-184: x += alpha p_loc; r -= alpha Ap; returns r.r (local) */
double Solve2(MgcgBlas* cublas, double alpha,
              Vector* xVector,
              Vector* ApVector, Vector* pVector, Vector* rVector,
              int countForDevice, int offsetForDevice);
/* Mgcg.cu:187-198: p_loc = beta p_loc + r */
void   Solve3(MgcgBlas* cublas, double beta,
              Vector* pVector, Vector* rVector,
              int countForDevice, int offsetForDevice);

/* Mgcg.cu:201-270: the whole single-device CG.  x is initial guess and result.
 * *iteration = number of loop bodies executed (last index + 1, as the
 * reference's post-incremented counter); *residual = sqrt(r.r) of the
 * recurrence residual.  Stop rule (minIteration <= it) && (residual <
 * allowableResidual).  Unlike the reference (which never reads maxIteration and
 * spins forever on NaN), the loop also ends after index maxIteration or on a
 * non-finite residual; MgcgGetLastError() then says so. */
void Solve(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
           Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
           Vector* xVector, Vector* bVector,
           Vector* ApVector, Vector* pVector, Vector* rVector,
           int elementsCount, int count,
           double allowableResidual, int minIteration, int maxIteration,
           int* iteration, double* residual);

/* ===================================================================== */
/* Part 2: additive exports (no reference analogue)                       */
/* ===================================================================== */

/* Thread-local message of the last failed call ("" if none); cleared by MgcgClearLastError. */
const char* MgcgGetLastError(void);
void        MgcgClearLastError(void);
/* Library ABI revision (3 since round 5: exports added, Vector grew, 14 tuning knobs retired, dot_order added). */
int         MgcgAbiVersion(void);
/* Tuning knobs (15).  Every MGCG_* environment variable the library honours is read once, at first use; launches never read the
 * environment.  No knob changes an element-wise result (SpMV rows, vector updates, the V-cycle: the same doubles under every
 * setting).  What a SCHEDULE knob may change is how a dot product's partial sums are grouped: "overlap" reduces p.Ap from the
 * interior rows' launch and the boundary rows' launch instead of one launch, so with the default overlap = 1 -- a decision taken
 * from a timing -- two runs of a several-ranks solve on >= 1 M rows per rank can differ in the last bits of alpha / beta (1e-16
 * relative; near the tolerance, by one iteration).  For reproducible bits set overlap to 0 or 2, or dot_order to 1 (which makes
 * every sum independent of every schedule).  bench.py's N > 1 line names the schedule that ran.
 *   overlap (MGCG_OVERLAP: 0 halo exchange in line; 1 [default] hidden behind the interior rows where that pays BY MEASUREMENT -- for
 *     slices of >= 1 M rows per rank the plan's own exchange is timed in line against the fork / launch / join round trip of the overlap
 *     schedule on the live communicator, once per plan, and every rank takes the same decision from the all-reduced times; 2 whenever an
 *     interior exists);
 *   halo_stream (overlap schedule: 0 [default] the interior rows run on the communicator's side stream and every RCCL call on the
 *     main stream; 1 the halo exchange runs on the side stream and all rows on the main stream -- opt-in until RCCL on two streams of
 *     one communicator has been run on real multi-GPU hardware);
 *   deep_halo (MGCG_DEEP_HALO, default 1: the row-partitioned V(1,1) cycle takes ONE exchange per level -- a few planes of the level's
 *     right-hand side, after which everything within reach of the slab's boundaries is recomputed locally, bit for bit what the owner
 *     computes -- 4 exchanges per MGCG iteration; 0: one exchange per SpMV-shaped pass, 8 per iteration.  Needs slabs at least as thick
 *     as the halo on every coarse level (2 planes; nu_c on the coarsest), else the library takes the per-pass schedule by itself);
 *   no_fold (MGCG_NO_FOLD: the first Jacobi sweep of a V(1,.) cycle is stored instead of being formed per gather of the residual pass);
 *   fold_up (MGCG_FOLD_UP: -1 [default] the prolongation of a V(1,1) cycle is formed per gather of the post-smoothing sweep on levels of
 *     up to 100 M rows, 0 never, 1 on every level);
 *   check_every (iterations enqueued ahead of the stop flag, default 4);
 *   auto_tiles (MGCG_AUTO_TILES, default 1: Solve-family and per-op calls build column tiles themselves for matrices without locality);
 *   tile_shift (MGCG_TILE_SHIFT: column tiles of 2^shift columns; 0 [default] equal-width tiles of about 2.85 MiB of x),
 *     tile_pack (MGCG_TILE_PACK, default 1: 12-byte tile entries; 0 the 16-byte form);
 *   placement (MGCG_PLACEMENT, default 3: the first CG Solve-family call on vectors of >= 32 M entries times the loop's SpMV on that many
 *     EXTRA allocations of Ap, keeps the fastest, and then does the same for p -- where the runtime places the two vectors moves the
 *     SpMV by up to 17 %, the written one most; 0: off; a vector whose address ToRawPtr_Double has handed out is never moved; the
 *     preconditioned loop draws only when maxIteration leaves room to win the draw's cost back);
 *   dot_order (MGCG_DOT_ORDER, default 0; VALIDATION ONLY: 1 = every dot product adds its rounded products strictly left to right, the
 *     ranks' sums are added in rank order and SpMV rows are always summed in stored order -- the reference CPU twin's arithmetic,
 *     LongVector.cs:15-31, SparseMatrix.cs:68-88, resultsDot.Sum() -- so residual traces and iterates EQUAL the oracle's bit for bit at
 *     every size and rank count; one serial sum of 1.3e8 terms takes ~0.5 s: never on a timed path);
 *   verbose (MGCG_VERBOSE: errors and decisions also go to stderr);
 *   virtual_devices (MGCG_VIRTUAL_DEVICES: one physical GPU shown as n devices, tests only);
 *   force_multirank (MGCG_FORCE_MULTIRANK = w > 0: a one-rank RCCL communicator takes the several-ranks code path with an
 *     artificial halo of w entries, for measuring that path's device-side cost on a one-GPU box);
 *   fail_comm_init (MGCG_FAIL_COMM_INIT, tests only: MgcgCommInitAll / MgcgCommInitRank report failure, as on a host whose RCCL
 *     cannot form a communicator -- callers must then fall back or fail loudly).
 * (MGCG_COMPRESSION and the MGCG_SPMV_* variables are per-handle defaults of MgcgSetMatrixCompression / MgcgSetSpmv*, read by
 * CreateSparse.)  Knobs whose A/B was settled in rounds 2-4 have been removed with their settled value compiled in.
 * MgcgSetTuning / MgcgGetTuning take the knob's name or its environment variable; they return 0, or -1 for an unknown
 * name (MgcgGetLastError).  MgcgReloadEnvironment reads all variables again.  Change knobs only while no solve is running. */
int         MgcgSetTuning(const char* name, int value);
int         MgcgGetTuning(const char* name, int* value);
void        MgcgReloadEnvironment(void);
/* Wait for the current device's stream.  Returns 0 on success. */
int         MgcgDeviceSynchronize(void);
/* Elapsed-time helpers on the current device's stream (HIP events). */
void*       MgcgEventCreate(void);
void        MgcgEventRecord(void* ev);
float       MgcgEventElapsedMs(void* start, void* stop);   /* synchronises on stop */
void        MgcgEventDestroy(void* ev);
/* Free / total bytes of HBM on the current device. */
int         MgcgMemGetInfo(long long* freeBytes, long long* totalBytes);
/* 64-bit-size vector creation (Create_Double takes int). */
Vector*     MgcgCreateDouble64(long long size);
VectorInt*  MgcgCreateInt64(long long size);
long long   MgcgVectorSize(const Vector* v);

/* ---- extra BLAS-1 / fused ops on raw device pointers ---- */
/* y = x + beta*y  (the reference's Scal+Axpy pair Mgcg.cu:197,265 in one pass) */
void   Xpay(MgcgBlas* cublas, double* y, const double* x, int count, double beta);
/* max_i |x_i|  (HandmadeCL residual norm, Mgcg/HandmadeCL/MgcgCL/Mgcg.cl:110-159) */
double NrmInf(MgcgBlas* cublas, const double* x, int count);
/* y = A x and returns sum_i w_i y_i in the same pass (w = the rows' own slice of x in CG) */
double CsrMVDot(MgcgBlas* cublas, MgcgSparse* cusparse,
                double* y, const double* elements, const int* rowOffsets, const int* columnIndeces,
                const double* x, const double* w,
                int elementsCount, int rowCount, int columnCount);
/* SpMV kernel selection for CsrMV-family calls on this handle:
 * 0 auto, 1 row-block LDS stream kernel, 2..8 = 2^(k-2) lanes per row (k=8: one wavefront per row). */
void   MgcgSetSpmvKernel(MgcgSparse* cusparse, int kernel);
/* Tuning knobs of the stream kernel: rowsPerBlock in {64,128,256}, flags bit0 = non-temporal matrix
 * loads, bit1 = XCD-contiguous row-block mapping, bit2 = banded schedule; gridBlocks (0 = chip-filling default). */
void   MgcgSetSpmvTuning(MgcgSparse* cusparse, int rowsPerBlock, int flags, int gridBlocks);
/* Banded schedule hint for the stream kernel: periodRows = distance (in rows) of the far band of the
 * matrix, e.g. nx*ny for a 3-D stencil (sets flag bit2).  The Solve-family exports detect it themselves;
 * 0 switches the schedule off.  A period the kernel cannot use (not a multiple of 8 row blocks, not
 * tiling the matrix) silently falls back to the plain schedule.  Results are identical either way. */
void   MgcgSetSpmvPeriod(MgcgSparse* cusparse, int periodRows);
/* Tile of the banded schedule: tileRows neighbouring rows x tilePlanes consecutive windows are in flight
 * per XCD at a time (0, 0 = the library's default).  Pure scheduling: results do not change. */
void   MgcgSetSpmvTile(MgcgSparse* cusparse, int tileRows, int tilePlanes);

/* Opt-in analysis (the role cuSPARSE's csrmv analysis plays in the reference's stack): with compression enabled the
 * Solve-family, MgSetup and CsrMV/CsrMVDot on this handle re-encode a matrix ONCE into a lossless compact form and use
 * it for every later SpMV on the same arrays.  Results are bit-identical to the CSR kernels (same doubles, same order).
 *   enable = 1: the best form the matrix admits --
 *       class 3  one byte per ROW when the matrix has <= 256 distinct rows read as sequences of (col-row, value)
 *                pairs (constant-coefficient stencils: 27 for the 7-point Laplacian on a box; every Galerkin level),
 *       class 2  two bytes per nonzero when it has <= 256 distinct offsets col-row and <= 256 distinct values,
 *       class 1  one byte per nonzero + the fp64 value when only the offsets qualify,
 *       class 4  a column-tiled copy (16 bytes per nonzero MORE, not fewer) for sorted rows whose entries are spread over
 *                an x far larger than the L2: the gathers of a pass stay inside one 4 MiB window of x,
 *       class 0  otherwise (plain CSR kernels);
 *   enable = 2: per-nonzero codes only (classes 2 / 1 / 0);   enable = 0: off.
 * The cache is keyed by the array pointers and sizes: a caller that rewrites a matrix in place must call
 * MgcgAnalysisClear.  MgcgAnalysisInfo(index) reports a cached analysis: returns the class (-1 past the end);
 * distinctOffsets / distinctValues receive, for class 3, the number of distinct rows and the longest row; for class 4
 * the number of column tiles and 0. */
void   MgcgSetMatrixCompression(MgcgSparse* cusparse, int enable);
void   MgcgAnalysisClear(MgcgSparse* cusparse);
int    MgcgAnalysisInfo(MgcgSparse* cusparse, int index, int* distinctOffsets, int* distinctValues, long long* rows, long long* nnz);

/* Per-launch HIP-event timing of the SpMV kernel inside the Solve.. / CgSteps calls on this handle's stream:
 * enable, run, then read the summed milliseconds and the number of launches timed. */
void   MgcgProfileSpmv(MgcgSparse* cusparse, int enable);
double MgcgProfileSpmvMs(MgcgSparse* cusparse, int* launches);

/* ---- synthetic structured problems generated directly in HBM ---- */
/* nnz of rows z in [zBegin, zEnd) of the 5/7-point Poisson matrix on nx*ny*nz. */
long long MgcgPoissonNnz(int nx, int ny, int nz, int zBegin, int zEnd);
/* Fill the CSR slice for rows with z in [zBegin, zEnd): diagonal 2*dim, off-diagonals -1,
 * Dirichlet, lexicographic x-fastest, ascending GLOBAL column ids, row offsets rebased to the
 * slice (as Initialize does).  Vectors must hold MgcgPoissonNnz / rows+1 entries.  Returns 0 on success. */
int MgcgGeneratePoisson(Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                        int nx, int ny, int nz, int zBegin, int zEnd);
/* min / max column id over the first elementCount entries (what Initialize returns as minJ, maxJ;
 * Mgcg.cu:83-84) for matrices that were generated on the device.  Returns 0 on success. */
int MgcgMinMaxColumn(VectorInt* columnIndecesVector, int elementCount, int* minJ, int* maxJ);
/* Fill a device vector with a constant. */
void MgcgFill(Vector* v, double value);

/* ---- solver with selectable stop rule ---- */
enum {
    MGCG_RULE_NATIVE     = 0,  /* Mgcg.cu:251-252   (min <= it) && res < tol                      */
    MGCG_RULE_CSHARP     = 1,  /* ConjugateGradient.cs:56-79  it<min: no; it>max: error; res<tol  */
    MGCG_RULE_SIMPLE     = 2,  /* SimpleConjugateGradient.cu:53,107  x:=0; (min < it) && res<tol  */
    MGCG_RULE_HANDMADECL = 3,  /* HandmadeCL ConjugateGradientCpu.cs:68-95  max-norm, C# rule     */
    MGCG_RULE_VIENNACL   = 4   /* ViennaCL ComputerGpu.cpp:78  (min < it) && rrNew/rr0 < tol^2    */
};
enum { MGCG_OK = 0, MGCG_MAXIT_EXCEEDED = 1, MGCG_NONFINITE = 3, MGCG_ERROR = -1 };
/* As Solve, plus: rule (above); residualTrace (host, may be NULL) receives the residual of each
 * iteration up to traceCapacity; returns a status code.  *iteration is the zero-based index of the
 * last executed loop body (what the C# classes expose as Iteration). */
int SolveEx(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
            Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
            Vector* xVector, Vector* bVector,
            Vector* ApVector, Vector* pVector, Vector* rVector,
            int elementsCount, int count,
            double allowableResidual, int minIteration, int maxIteration, int rule,
            int* iteration, double* residual,
            double* residualTrace, int traceCapacity);

/* ---- multigrid preconditioner (defined by this build; the reference's "Mgcg" never implemented it) ---- */
typedef struct MgcgMg MgcgMg;
/* Geometric cell-centred hierarchy on an nx*ny*nz lexicographic grid for the CSR matrix in the
 * vectors (count = nx*ny*nz rows): piecewise-constant P, R = P^T, A_c = sigma*P^T A P, weighted-Jacobi
 * V(nu,nu), nuCoarse sweeps on the last level.  levels is clipped to what the grid allows. */
MgcgMg* MgSetup(MgcgBlas* cublas, MgcgSparse* cusparse,
                Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                int elementsCount, int nx, int ny, int nz,
                int levels, double omega, int nu, int nuCoarse, double sigma);
/* The same on this rank's z-slab [zBegin, zEnd) of the grid (equal slabs in rank order; the vectors hold the local
 * rows with global column ids, as Initialize lays them out).  Every level keeps the slab aligned; per-level halo
 * planes travel over `comm` before each smoothing / residual pass.  comm == NULL: one rank. */
typedef struct MgcgComm MgcgComm;
MgcgMg* MgSetupParallel(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse,
                        Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                        int elementsCount, int nx, int ny, int nz, int zBegin, int zEnd,
                        int levels, double omega, int nu, int nuCoarse, double sigma);
void    MgDestroy(MgcgMg* mg);
/* Transfer operators of the V-cycle: 0 (default) piecewise-constant P, 1 cell-centred linear P (per coarsened dimension
 * a child takes 3/4 of its parent and 1/4 of the parent's neighbour on the child's side), R = P^T in both; the coarse
 * operators are the same.  With several ranks the call is collective (it plans one extra plane exchange per level and
 * transfer).  Returns 0, or -1 with MgcgGetLastError(). */
int     MgSetInterpolation(MgcgMg* mg, int mode);
int     MgLevels(const MgcgMg* mg);
/* rows / nnz / grid of level l; copy level l's CSR and D^-1 to host arrays (for tests). */
long long MgLevelRows(const MgcgMg* mg, int level);
long long MgLevelNnz(const MgcgMg* mg, int level);
void    MgLevelCopyCsr(const MgcgMg* mg, int level, double elements[], int columnIndeces[], int rowOffsets[]);
void    MgLevelCopyDinv(const MgcgMg* mg, int level, double dinv[]);
/* z = M^-1 r : one V-cycle from a zero guess (raw device pointers, count = level-0 rows). */
void    MgApply(MgcgMg* mg, const double* r, double* z);
/* Preconditioned CG with the reference's shell and stop rules; zVector is one more work vector. */
int     SolveMg(MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr, MgcgMg* mg,
                Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                Vector* xVector, Vector* bVector,
                Vector* ApVector, Vector* pVector, Vector* rVector, Vector* zVector,
                int elementsCount, int count,
                double allowableResidual, int minIteration, int maxIteration, int rule,
                int* iteration, double* residual,
                double* residualTrace, int traceCapacity);

/* ---- one process per GPU: RCCL over xGMI ---- */
/* 128-byte RCCL unique id created on one rank and handed to every rank by the launcher
 * (torch.distributed store / MPI / a file). Returns 0 on success. */
int       MgcgCommGetUniqueId(void* id128);
/* 1 if librccl resolved in this process with every entry point the library binds, else 0 (MgcgGetLastError says why).
 * ncclCommInitRank is collective: launchers agree on this (and on one device per local rank) over their own channel
 * BEFORE any rank calls MgcgCommInitRank, so that a rank that cannot enter it never leaves the others blocked inside. */
int       MgcgRcclAvailable(void);
MgcgComm* MgcgCommInitRank(const void* id128, int nranks, int rank);
void      MgcgCommDestroy(MgcgComm* comm);
/* ONE process driving ndev devices, one host thread per device -- the shape of the reference's ConjugateGradientParallelGpu
 * (Mgcg/cuBlas/Mgcg/ConjugateGradientParallelGpu.cs:264-324: every device of the process; :424-565: a Parallel.For per phase).
 * Forms the ndev communicators of devices 0 .. ndev-1 at once, from the calling thread (ncclGroupStart / ndev x
 * ncclCommInitRank / ncclGroupEnd): comms[d] is rank d of ndev and belongs to device d.  Call it before the devices'
 * handles and vectors are created.  Afterwards thread d calls SetDevice(d) and SolveParallel / MgSetupParallel /
 * SolveMgParallel(comms[d], ...); all ndev calls must be in flight together (they meet in collectives).
 * With fewer physical devices than ndev (MGCG_VIRTUAL_DEVICES, tests) the communicators share an in-process loopback group;
 * ndev == 1 gives a single-rank communicator.  Returns 0, or -1 with MgcgGetLastError() and every comms[d] NULL. */
int       MgcgCommInitAll(MgcgComm* comms[], int ndev);
/* "rccl", "loopback", "callbacks", "single" (one rank, no transport) or "none" (NULL). */
const char* MgcgCommTransport(const MgcgComm* comm);
/* Device-side cost of one collective step of the multi-rank loop on this communicator's stream: `reps` back-to-back
 *   what = 0  all-reduces (sum) of `count` doubles (count <= 8),
 *   what = 1  halo exchanges: one grouped send/recv of `count` doubles with every other rank (with itself on one rank),
 *   what = 2  fork / join pairs of the overlap schedule (event record + stream wait on the side stream and back),
 *   what = 3  empty single-workgroup kernel launches (the price of a kernel boundary on this stream),
 *   what = 4  the stencil's halo exchange: one grouped send/recv of `count` doubles with ranks rank - 1 and rank + 1 only,
 * timed with HIP events around the batch; returns microseconds per repetition (NaN on error -- and, without an error message, for
 * what = 1 / 4 on a transport without RCCL: host-staged planes travel through the launcher, there is nothing to time on the device).
 * Collective: every rank of the communicator must make the same call. */
double    MgcgCommProbe(MgcgComm* comm, int what, int count, int reps);
int       MgcgCommRank(const MgcgComm* comm);
int       MgcgCommSize(const MgcgComm* comm);
/* In-process loopback transport: N ranks of ONE process (host threads, e.g. MGCG_VIRTUAL_DEVICES on one GPU)
 * exchange through host memory behind a barrier.  For testing the multi-rank logic where RCCL cannot form a
 * communicator (it needs one device per rank); every rank must make the same sequence of solver calls. */
typedef struct MgcgLoopback MgcgLoopback;
MgcgLoopback* MgcgLoopbackCreate(int nranks);
void          MgcgLoopbackDestroy(MgcgLoopback* group);
MgcgComm*     MgcgCommInitLoopback(MgcgLoopback* group, int rank);
/* Host-staged transport through callbacks of the launcher (e.g. torch.distributed / MPI on host memory): a fallback for
 * hosts where RCCL cannot form a communicator, and a way to drive the multi-rank loop from any message layer.
 *   allGather(mine, all, user):   all[4*q .. 4*q+3] = the four int64 of rank q  (every rank's `mine`, rank order)
 *   allReduce(values, count, user): values[i] = sum over ranks, in place, the SAME bits on every rank
 *   exchange(nranks, sendBufs, sendCounts, recvBufs, recvCounts, user): for every peer q with sendCounts[q] > 0 send
 *       sendBufs[q][0 .. sendCounts[q]) to q, with recvCounts[q] > 0 receive into recvBufs[q]; returns when all arrived.
 * All buffers are host memory owned by the library; the callbacks are invoked from the thread that calls the solver. */
typedef void (*MgcgAllGatherFn)(const long long mine[4], long long all[], void* user);
typedef void (*MgcgAllReduceFn)(double values[], int count, void* user);
typedef void (*MgcgExchangeFn)(int nranks, const double* const sendBufs[], const long long sendCounts[],
                               double* const recvBufs[], const long long recvCounts[], void* user);
MgcgComm*     MgcgCommInitCallbacks(int nranks, int rank, MgcgAllGatherFn allGather, MgcgAllReduceFn allReduce, MgcgExchangeFn exchange, void* user);
/* sum of one double over all ranks (test / bootstrap helper; blocking). */
double    MgcgCommAllReduceSum(MgcgComm* comm, double value);
/* The whole multi-rank CG of ConjugateGradientParallelGpu.Solve (ConjugateGradientParallelGpu.cs:424-565)
 * run natively: this rank owns rows [offsetForDevice, +countForDevice) set up by Initialize; pVector is
 * full length (count).  Halo = the entries of p in [minJ, offset) and [offset+count, maxJ] (exchanged with
 * grouped ncclSend/ncclRecv between the ranks that own them), dot products = local partial + ncclAllReduce.
 * comm == NULL means a single rank.  Same outputs and rules as SolveEx. */
int SolveParallel(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr,
                  Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                  Vector* xVector, Vector* bVector,
                  Vector* ApVector, Vector* pVector, Vector* rVector,
                  int count, int countForDevice, int offsetForDevice, int elementsCountForDevice,
                  int minJ, int maxJ,
                  double allowableResidual, int minIteration, int maxIteration, int rule,
                  int* iteration, double* residual,
                  double* residualTrace, int traceCapacity);

/* Multi-rank preconditioned CG: SolveParallel with the V-cycle of an MgSetupParallel hierarchy (zVector: local work vector). */
int SolveMgParallel(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse, MgcgMatDescr* matDescr, MgcgMg* mg,
                    Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                    Vector* xVector, Vector* bVector, Vector* ApVector, Vector* pVector, Vector* rVector, Vector* zVector,
                    int count, int countForDevice, int offsetForDevice, int elementsCountForDevice, int minJ, int maxJ,
                    double allowableResidual, int minIteration, int maxIteration, int rule,
                    int* iteration, double* residual, double* residualTrace, int traceCapacity);

/* Fixed number of CG iterations with no stop test and no host synchronisation inside (bench.py's
 * "steps"): runs `steps` more iterations of the recurrence held in x,r,p (call with restart != 0 first
 * to compute r = b - A x, p = r, rr).  comm may be NULL.  Returns the residual after the last step. */
double CgSteps(MgcgComm* comm, MgcgBlas* cublas, MgcgSparse* cusparse,
               Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
               Vector* xVector, Vector* bVector,
               Vector* ApVector, Vector* pVector, Vector* rVector,
               int count, int countForDevice, int offsetForDevice, int elementsCountForDevice,
               int minJ, int maxJ, int steps, int restart);

/* Extreme eigenvalues by `steps` Lanczos steps on the device (SpMV + dot kernels of the CG path): of A, or with
 * jacobiScaled != 0 of D^-1/2 A D^-1/2 (the spectrum of D^-1 A, what the Jacobi smoother's damping depends on).
 * Scalable counterpart of the reference's dense Jacobi-rotation GetEigenValues
 * (Mgcg/HandmadeCL/MgcgCL/SparseMatrix.cs:234-372).  Ritz values lie inside the spectrum: lambdaMax is approached
 * from below, lambdaMin from above.  ritz (optional) receives the *stepsDone Ritz values in ascending order (room for
 * `steps`).  The start vector is a fixed function of `seed`.  Returns MGCG_OK or MGCG_ERROR. */
int MgcgEstimateSpectrum(MgcgBlas* cublas, MgcgSparse* cusparse,
                         Vector* elementsVector, VectorInt* rowOffsetsVector, VectorInt* columnIndecesVector,
                         int elementsCount, int count, int jacobiScaled, int steps, unsigned seed,
                         double* lambdaMin, double* lambdaMax, double ritz[], int* stepsDone);

/* How the last multi-rank solve on the calling thread scheduled its halo: returns 1 and the local row range
 * [interior[0], interior[1]) that was multiplied while the halo of p travelled on the communicator's own stream
 * (rows outside it wait for the halo), or 0 when the exchange ran in line (single rank, MGCG_OVERLAP=0, or the
 * slice has too few rows that reference local columns only).  interior may be NULL. */
int MgcgLastOverlap(long long interior[2]);
/* The calling thread's last placement draw (tuning knob `placement`): which = 0 the SpMV's output vector Ap, 1 its gathered input p.
 * Returns the number of candidate allocations timed (0: no draw happened -- small vector, knob off, address already exported),
 * milliseconds[i] = the loop's SpMV on candidate i (candidate 0 is the allocation the vector came with), *chosen = the one kept.
 * milliseconds / chosen may be NULL. */
int MgcgLastPlacement(int which, double milliseconds[], int capacity, int* chosen);
/* The measurement behind that choice (overlap = 1): returns 1 and microseconds[0] = one halo exchange in line, microseconds[1] = one
 * fork / empty launch / join round trip, both averaged over the ranks, if the calling thread's last plan was decided by measurement;
 * 0 if it was decided by the knob or the slice's size alone.  microseconds may be NULL. */
int MgcgLastOverlapTimes(double microseconds[2]);
/* Which folds the calling thread's LAST V-cycle took (Apply / SolveMg / SolveMgParallel): bit 0 = on some level the first sweep from
 * zero was formed per gather of the residual pass instead of being stored, bit 1 = on some level the prolongation was formed per gather
 * of the post-smoothing sweep as well (V(1,1), plain CSR, uniform diagonal, power-of-two nx and ny), bit 2 (4) = several ranks: the
 * deep-halo cycle ran (one exchange per coarse level instead of one per pass; knob deep_halo), bit 3 (8) = ... and the finest level's
 * right-hand side carried its halo planes, so every level ran the single-rank folded kernels (SolveMgParallel keeps r in the hierarchy's
 * buffer; MgApply on a caller's vector cannot; taken only when ALL ranks agreed at set-up that their rows hold one and the same uniform
 * diagonal -- the form decides what the level exchanges, r or the stored first sweep).  Schedules only: the results are bit-identical either way (MGCG_NO_FOLD=1 /
 * MGCG_FOLD_UP=0 switch the folds off; MGCG_FOLD_UP=1 takes the second one on levels of any size, by default it is taken up to 100 M rows). */
int MgcgLastVcycleFolds(void);
/* The calling thread's last halo exchange: returns 1 if it moved per-peer index lists (unstructured slices: only the entries
 * of p the slice's column ids reference -- plan built once from them), 0 if contiguous ranges (banded / stencil slices,
 * the reference's [minJ, offset) and [offset + count, maxJ]: Mgcg.cu:83-84, ConjugateGradientParallelGpu.cs:397-398);
 * volume[0] = entries this rank received per exchange, volume[1] = entries the contiguous ranges would have received. */
int MgcgLastHalo(long long volume[2]);
/* Test hook (no device needed): the order in which the lane = row SpMV kernels walk their tiles of `tileRows` rows with
 * `workgroups` workgroups when the far band of the matrix lies `periodRows` rows from the diagonal (0: unknown).
 * tiles[wg * maxTrips + t] = tile of workgroup wg in trip t, or -1; returns 0 (memory order), 1 / 2 (z sweep), -1 (bad arguments). */
int MgcgDebugTileOrder(long long rows, int periodRows, int workgroups, int tileRows, int* tiles, int maxTrips);

#ifdef __cplusplus
}
#endif
#endif /* MGCG_GPU_H */
