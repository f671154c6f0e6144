// mg_kernels.hpp - hand-written gfx950 (CDNA4, wave64) kernels of the multigrid cycle.
//
// Every SpMV/SpMM on the reference's hot path is a row-parallel CSR gather product
// (reference src/Multigrid/SpMatMul.jl:4-26; the arrays of the transposed CSC are the CSR arrays of A,
// MGdef.jl:75-77).  All kernels here are HBM-bandwidth bound (0.13-0.65 flop/B): no MFMA.
//
// CSR-stream layout of one launch:
//   * the host cuts the rows into "row blocks": consecutive rows whose non-zeros fit one LDS chunk
//     (MG_CHUNK entries) - so a workgroup streams ONE contiguous nnz segment of val/colidx with
//     16 B / 8 B per lane fully coalesced loads, independent of the row lengths;
//   * products val*x[col] are staged in LDS; each row is then reduced from LDS (segmented reduction:
//     1..64 lanes per row depending on how many rows the block holds) and the epilogue
//     (axpby / residual / damped-Jacobi update) is fused, so the vectors are touched once;
//   * blockIdx is remapped so that each XCD (private 4 MiB L2) walks one contiguous band of rows:
//     the x gather window (3 grid planes for a 7-point stencil) then lives in that XCD's L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mgk {

#ifndef MG_BLK
#define MG_BLK 256
#endif
#ifndef MG_ITEMS
#define MG_ITEMS 8
#endif
constexpr int BLK = MG_BLK;         // threads per workgroup (4 waves)
constexpr int ITEMS = MG_ITEMS;     // non-zeros per thread per chunk
constexpr int CHUNK = BLK * ITEMS;  // 2048 non-zeros staged per workgroup (16 KiB of products)
#ifndef MG_MAXROWS
#define MG_MAXROWS MG_BLK
#endif
constexpr int MAXROWS = MG_MAXROWS;        // rows per row block (one LDS row-pointer slot per thread)
constexpr int PAIRS = ITEMS / 2;
// The block-RHS kernels use their own, smaller row blocks: half the LDS per workgroup lets the wave limit (8
// workgroups per CU) instead of LDS set the occupancy - measured on C5: fused sweep 2.51 -> 2.18 ms
// (profiles/r01_nt_ab.md).
constexpr int MM_ITEMS = 4;
constexpr int MM_CHUNK = BLK * MM_ITEMS;  // 1024 non-zeros (values + column indices: 12 KiB of LDS)
constexpr int MM_MAXROWS = 128;
constexpr int MM_PAIRS = MM_ITEMS / 2;

enum Mode { AXPBY = 0, RESID = 1, SMOOTH = 2 };

typedef double d2_t __attribute__((ext_vector_type(2)));
typedef int i2_t __attribute__((ext_vector_type(2)));
// The matrix stream (values, column indices) is read exactly once per launch.  NT = non-temporal loads:
// the stream then does not evict the gather window of the source vector from the XCD's L2.  Measured on
// C2 (profiles/r01_nt_ab.md): +11..17 % on the transfer operators, -2 % on the 7-point A -> chosen
// per operator by the host (Csr::nt).
template <bool NT>
__device__ __forceinline__ d2_t load_stream(const double* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(p));
  return *reinterpret_cast<const d2_t*>(p);
}
template <bool NT>
__device__ __forceinline__ i2_t load_stream(const int* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const i2_t*>(p));
  return *reinterpret_cast<const i2_t*>(p);
}

// PTR: the type of the row pointers - int (12 B per non-zero + 4 B per row, SURVEY 8d's device widths) for operators of fewer than
// 2^31 non-zeros, long long beyond (round 6: the reference is Int64 throughout, Multigrid.jl:19; the streaming kernels csr_stream_spmv and
// csr_longrow_spmv are instantiated for both, every other format is built for int operators only)
template <typename PTR>
struct CsrDevT {
  typedef PTR ptr_t;
  const PTR* rowptr;   // n_rows+1, 0-based
  const int* colidx;   // nnz (+pad), 0-based
  const double* val;   // nnz (+pad)
  const int* blk_row;  // nblocks+1 row-block boundaries
  const int* sched;    // optional processing order of the row blocks (nullptr: natural order)
  int nblocks;
  int n_rows;
  int n_cols;
};
typedef CsrDevT<int> CsrDev;
typedef CsrDevT<long long> CsrDev64;
__device__ __forceinline__ int uniform_first(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uniform_first(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v & 0xffffffffull));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | (unsigned long long)lo);
}

struct VecArgs {
  double* sumsq;    // optional: per-row-block sum of squares of the output (fused ||r||^2, nrhs == 1)
  const double* x;  // gathered vector           [n_cols][nrhs]
  const double* xs; // SMOOTH: the row's own x   [n_rows][nrhs] (== x unless the operator holds a row sub-range)
  double* y;        // output                    [n_rows][nrhs]
  const double* b;  // RESID / SMOOTH            [n_rows][nrhs]
  const double* d;  // SMOOTH: relaxPrec          [n_rows]; nullptr in the row-class kernels = read it from the class dictionary
  const double* d_full;  // SMOOTH: always the relaxPrec vector (exception rows of a row-class operator)
  double* y2;       // RESID, csr_rowclass_tile_spmv only: optional second output x + d.*(b - A x), i.e. the first
                    // damped-Jacobi update of the NEXT cycle, written while r and x are at hand (solve loop)
  double alpha;     // AXPBY
  double beta;      // AXPBY
  int nrhs;
  int dotx;         // csr_rowclass_march_spmv only: the partials in `sumsq` are of x[row]*out[row] (p'Ap of CG) instead of out^2
};

// Each XCD gets a contiguous band of logical blocks (workgroups are dealt round-robin over the 8 XCDs:
// MI355X_MICROARCH.md "Workgroup dispatch").  Bijective for every nb.  Placement only affects speed.
__device__ __forceinline__ int xcd_band(int bid, int nb) {
  const int q = nb >> 3, rem = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return xcd * q + (xcd < rem ? xcd : rem) + idx;
}

template <int MODE>
__device__ __forceinline__ double epilogue(const VecArgs& v, int row, double acc, double pb, double pd,
                                           double px) {
  if (MODE == AXPBY) return v.alpha * acc + pb;  // pb = beta*y[row] (0 when beta == 0)
  if (MODE == RESID) return pb - acc;            // pb = b[row]
  return px + pd * (pb - acc);                   // SMOOTH: x + d*(b - A x)
}

// ------------------------------------------------------------------------------------------------
// CSR-stream SpMV, one right-hand side.
// ------------------------------------------------------------------------------------------------
template <int MODE, bool NT, typename AD = CsrDev>
__global__ __launch_bounds__(BLK) void csr_stream_spmv(AD A, VecArgs v) {
  typedef typename AD::ptr_t K;      // position in the non-zero stream: int, or long long for operators of >= 2^31 non-zeros
  __shared__ double prod[CHUNK];
  __shared__ int srow[MAXROWS + 1];
  __shared__ double red[BLK / 64];

  const int tid = threadIdx.x;
  int bid = xcd_band(blockIdx.x, A.nblocks);
  if (A.sched) bid = A.sched[bid];
  const int r0 = A.blk_row[bid];
  const int r1 = A.blk_row[bid + 1];
  const int nrows = r1 - r0;
  const K k0 = A.rowptr[r0];
  const K k1 = A.rowptr[r1];

  if (nrows == 1 && (k1 - k0) > CHUNK - 2) {
    // One row longer than a chunk: the whole workgroup strides over it.
    double acc = 0.0;
    for (K k = k0 + tid; k < k1; k += BLK) acc += A.val[k] * v.x[A.colidx[k]];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < BLK / 64; ++w) s += red[w];
      double pb = 0.0, pd = 0.0, px = 0.0;
      if (MODE == AXPBY) pb = (v.beta != 0.0) ? v.beta * v.y[r0] : 0.0;
      else pb = v.b[r0];
      if (MODE == SMOOTH) { pd = v.d[r0]; px = v.xs[r0]; }
      const double o = epilogue<MODE>(v, r0, s, pb, pd, px);
      v.y[r0] = o;
      if (v.sumsq) v.sumsq[bid] = o * o;
    }
    return;
  }

  // lanes per row in the reduction phase: as many as the block's row count leaves room for
  int sh = 0;
  while (sh < 6 && (2 << sh) * nrows <= BLK) ++sh;
  const int tpr = 1 << sh;
  const int lrow = tid >> sh;
  const int sub = tid & (tpr - 1);
  const bool owner = (lrow < nrows) && (sub == 0);

  // ---- issue every global load up front: matrix stream, row pointers, epilogue operands -------
  const K base = k0 & ~(K)1;  // 16-B aligned start of the value stream
  d2_t va[PAIRS];
  i2_t ca[PAIRS];
#pragma unroll
  for (int it = 0; it < PAIRS; ++it) {
    const K idx = base + it * (2 * BLK) + 2 * tid;
    if (idx < k1) {
      va[it] = load_stream<NT>(A.val + idx);
      ca[it] = load_stream<NT>(A.colidx + idx);
    } else {
      va[it] = d2_t{0.0, 0.0};
      ca[it] = i2_t{0, 0};
    }
  }
  if (tid <= nrows) srow[tid] = (int)(A.rowptr[r0 + tid] - base);
  if (tid == 0 && nrows == MAXROWS) srow[MAXROWS] = (int)(k1 - base);
  double pb = 0.0, pd = 0.0, px = 0.0;
  if (owner) {
    const int row = r0 + lrow;
    if (MODE == AXPBY) { if (v.beta != 0.0) pb = v.beta * v.y[row]; }
    else pb = v.b[row];
    if (MODE == SMOOTH) { pd = v.d[row]; px = v.xs[row]; }
  }
  // ---- gather x and stage the products --------------------------------------------------------
#pragma unroll
  for (int it = 0; it < PAIRS; ++it) {
    const K idx = base + it * (2 * BLK) + 2 * tid;
    if (idx < k1) {
      d2_t p;
      p.x = va[it].x * v.x[ca[it].x];
      p.y = va[it].y * v.x[ca[it].y];
      *reinterpret_cast<d2_t*>(&prod[it * (2 * BLK) + 2 * tid]) = p;
    }
  }
  __syncthreads();
  // ---- segmented reduction from LDS + fused epilogue -------------------------------------------
  double acc = 0.0;
  if (lrow < nrows) {
    const int s = srow[lrow], e = srow[lrow + 1];
    for (int k = s + sub; k < e; k += tpr) acc += prod[k];
  }
  for (int o = tpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  double outv = 0.0;
  if (owner) {
    outv = epilogue<MODE>(v, r0 + lrow, acc, pb, pd, px);
    v.y[r0 + lrow] = outv;
    // AXPBY with v.y2: also d[row]*out - the restriction hands the coarse level its first update x = d.*bc (as the lane kernel does)
    if (MODE == AXPBY && v.y2) v.y2[r0 + lrow] = v.d_full[r0 + lrow] * outv;
  }
  if (v.sumsq) {  // fused Frobenius norm (SolveFuncs.jl:30): deterministic per-block partial, summed by sum_final
    double sq = outv * outv;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < BLK / 64; ++w) t += red[w];
      v.sumsq[bid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Pattern-coded CSR-stream SpMV.
// Same CSR value stream (row-major, staged through LDS with the same coalesced 16-byte loads), but the
// 4-byte column index per non-zero is replaced by TWO numbers per row: the row's first column and the id
// of its offset pattern (cols - cols[0]) in a small dictionary.  Operators that come from grids - the
// 7-point fine operator, the 27-point Galerkin operators, full-weighting P and R - have a handful to a
// few hundred distinct patterns, so the index traffic drops from 4 B/nnz to 6 B/row (fine A: 88 -> 66 B per
// row in total) and, just as important, the gather addresses no longer depend on an index load: a lane
// computes first + off[k] and issues its gathers back to back; consecutive rows with the same pattern
// gather consecutive addresses (fully coalesced).  The host falls back to plain CSR (csr_stream_spmv) when
// the dictionary would not be small (unstructured matrices).  Products are summed in stored order when one
// lane owns a row - exactly the order of a sequential CPU row loop.
// ------------------------------------------------------------------------------------------------
constexpr int DICT_LDS = 1024;    // offset dictionaries up to this many entries are staged in LDS
constexpr int MAXRUNS = 32;       // row blocks with at most this many descriptor runs use the run-length form
struct PatDev {
  // Run-length form of the row descriptors: inside a row block, consecutive rows that share a pattern and
  // whose first column advances by a constant stride form one run {local row0, pattern id, first column of
  // row0, stride, value offset of row0} (5 ints).  A grid line is 1-3 runs, so the per-row descriptor
  // traffic (row pointer 4 B + first column 4 B + pattern id 2 B) shrinks to a few dozen bytes per block.
  const int* run_ptr;             // nblocks+1 (nullptr: no run form); run_ptr[b+1]-run_ptr[b] == 0: per-row form
  const int* runs;                // 5 ints per run
  const int* firstcol;            // n_rows: first column index of the row
  const unsigned short* pat;      // n_rows: pattern id
  const int* pat_ptr;             // npat+1
  const int* pat_off;             // concatenated offset lists (off[0] == 0)
  int dict_entries;
  int npat;
};

template <int MODE, bool NT, bool DLDS>
__global__ __launch_bounds__(BLK) void csr_pattern_spmv(CsrDev A, PatDev P, VecArgs v) {
  __shared__ double sval[CHUNK];
  __shared__ int srow[MAXROWS + 1];
  __shared__ double red[BLK / 64];
  __shared__ int soff[DLDS ? DICT_LDS : 1];
  __shared__ int sptr[DLDS ? DICT_LDS : 1];   // pattern start offsets (npat <= dict_entries <= DICT_LDS)

  const int tid = threadIdx.x;
  int bid = xcd_band(blockIdx.x, A.nblocks);
  if (A.sched) bid = A.sched[bid];
  const int r0 = A.blk_row[bid];
  const int r1 = A.blk_row[bid + 1];
  const int nrows = r1 - r0;
  const int k0 = A.rowptr[r0];
  const int k1 = A.rowptr[r1];

  if (nrows == 1 && (k1 - k0) > CHUNK - 2) {  // one row longer than a chunk
    const int first = P.firstcol[r0];
    const int po = P.pat_ptr[P.pat[r0]];
    double acc = 0.0;
    for (int k = k0 + tid; k < k1; k += BLK) acc += A.val[k] * v.x[first + P.pat_off[po + (k - k0)]];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((tid & 63) == 0) red[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < BLK / 64; ++w) s += red[w];
      double pb = 0.0, pd = 0.0, px = 0.0;
      if (MODE == AXPBY) pb = (v.beta != 0.0) ? v.beta * v.y[r0] : 0.0;
      else pb = v.b[r0];
      if (MODE == SMOOTH) { pd = v.d[r0]; px = v.xs[r0]; }
      const double o = epilogue<MODE>(v, r0, s, pb, pd, px);
      v.y[r0] = o;
      if (v.sumsq) v.sumsq[bid] = o * o;
    }
    return;
  }

  int sh = 0;
  while (sh < 6 && (2 << sh) * nrows <= BLK) ++sh;
  const int tpr = 1 << sh;
  const int lrow = tid >> sh;
  const int sub = tid & (tpr - 1);
  const bool owner = (lrow < nrows) && (sub == 0);

  // ---- all global loads up front: value stream, row pointers, row descriptors, epilogue operands ----
  const int base = k0 & ~1;
  d2_t va[PAIRS];
#pragma unroll
  for (int it = 0; it < PAIRS; ++it) {
    const int idx = base + it * (2 * BLK) + 2 * tid;
    va[it] = (idx < k1) ? load_stream<NT>(A.val + idx) : d2_t{0.0, 0.0};
  }
  __shared__ int srun[5 * MAXRUNS];
  int nruns = 0, run0 = 0;
  if (P.run_ptr) {
    run0 = P.run_ptr[bid];
    nruns = P.run_ptr[bid + 1] - run0;
  }
  if (nruns > 0) {
    if (tid < 5 * nruns) srun[tid] = P.runs[5 * run0 + tid];
  } else {
    if (tid <= nrows) srow[tid] = A.rowptr[r0 + tid] - base;
    if (tid == 0 && nrows == MAXROWS) srow[MAXROWS] = k1 - base;
  }
  int first = 0, po = 0;
  if (nruns == 0 && lrow < nrows) {
    first = P.firstcol[r0 + lrow];
    po = P.pat[r0 + lrow];                      // pattern id; resolved to its dictionary offset below
    if (!DLDS) po = P.pat_ptr[po];
  }
  double pb = 0.0, pd = 0.0, px = 0.0;
  if (owner) {
    const int row = r0 + lrow;
    if (MODE == AXPBY) { if (v.beta != 0.0) pb = v.beta * v.y[row]; }
    else pb = v.b[row];
    if (MODE == SMOOTH) { pd = v.d[row]; px = v.xs[row]; }
  }
  if (DLDS) {   // the pattern dictionary: loads behind everything above, all issued before the first LDS write
    constexpr int ND = (DICT_LDS + BLK - 1) / BLK;
    int so[ND], sp[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      so[u] = i < P.dict_entries ? P.pat_off[i] : 0;
      sp[u] = i <= P.npat ? P.pat_ptr[i] : 0;
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      if (i < DICT_LDS) {
        if (i < P.dict_entries) soff[i] = so[u];
        if (i <= P.npat) sptr[i] = sp[u];
      }
    }
  }
#pragma unroll
  for (int it = 0; it < PAIRS; ++it) {
    const int idx = base + it * (2 * BLK) + 2 * tid;
    if (idx < k1) *reinterpret_cast<d2_t*>(&sval[it * (2 * BLK) + 2 * tid]) = va[it];
  }
  __syncthreads();
  // ---- row phase: gather addresses come from the pattern, not from a loaded index -----------------
  double acc = 0.0;
  if (lrow < nrows) {
    int s, e;
    if (nruns > 0) {  // locate the row's run (1-3 runs per grid line) and derive its descriptor
      int ri = 0;
      while (ri + 1 < nruns && srun[5 * (ri + 1)] <= lrow) ++ri;
      const int dr = lrow - srun[5 * ri];
      po = srun[5 * ri + 1];
      first = srun[5 * ri + 2] + dr * srun[5 * ri + 3];
      const int p0 = DLDS ? sptr[po] : P.pat_ptr[po];
      const int len = (DLDS ? sptr[po + 1] : P.pat_ptr[po + 1]) - p0;
      s = srun[5 * ri + 4] + dr * len - base;
      e = s + len;
      po = p0;
    } else {
      s = srow[lrow];
      e = srow[lrow + 1];
      if (DLDS) po = sptr[po];
    }
    const int* off = (DLDS ? soff : P.pat_off) + po - s;  // off[k] for k in [s, e)
    // first 8 entries of the lane: all gathers issued back to back, then summed in stored order
    double xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = s + sub + j * tpr;
      xv[j] = (kk < e) ? v.x[first + off[kk]] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int kk = s + sub + j * tpr;
      if (kk < e) acc += sval[kk] * xv[j];
    }
    int k = s + sub + 8 * tpr;
    for (; k + 3 * tpr < e; k += 4 * tpr) {
      const double x0 = v.x[first + off[k]];
      const double x1 = v.x[first + off[k + tpr]];
      const double x2 = v.x[first + off[k + 2 * tpr]];
      const double x3 = v.x[first + off[k + 3 * tpr]];
      acc += sval[k] * x0;
      acc += sval[k + tpr] * x1;
      acc += sval[k + 2 * tpr] * x2;
      acc += sval[k + 3 * tpr] * x3;
    }
    for (; k < e; k += tpr) acc += sval[k] * v.x[first + off[k]];
  }
  for (int o = tpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  double outv = 0.0;
  if (owner) {
    outv = epilogue<MODE>(v, r0 + lrow, acc, pb, pd, px);
    v.y[r0 + lrow] = outv;
    // AXPBY with v.y2: also d[row]*out - the restriction hands the coarse level its first update x = d.*bc (as the lane kernel does)
    if (MODE == AXPBY && v.y2) v.y2[r0 + lrow] = v.d_full[r0 + lrow] * outv;
  }
  if (v.sumsq) {
    double sq = outv * outv;
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < BLK / 64; ++w) t += red[w];
      v.sumsq[bid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-class SpMV (value-indexed, pattern-coded CSR; one right-hand side).
// Operators of constant-coefficient stencils - and their Galerkin coarse operators, and the full-weighting P / R -
// consist of a handful of distinct rows: the same column offsets (relative to the row's first column) carrying the
// same values.  The host finds these classes at upload (bit-exact comparison of offsets and values, mgvcycle.hip
// build_rowclasses) and, when the dictionary is small, the device keeps per row only {first column int32, class
// uint16}; offsets and values come from the dictionary.  The matrix stream (12 B per non-zero) disappears: a launch
// moves 6 B per row plus the vectors.  LOSSLESS: the products and their summation order (ascending k, as in the
// reference's CSR loop, SpMatMul.jl:4-13 / parRelax.cpp) are those of the CSR arrays, which stay resident for the
// block-RHS kernels, the numeric Galerkin product and mg_get_values.  Operators without this redundancy (variable
// coefficients, SA-AMG levels) never get here and use the kernels above.
// One thread per row: consecutive lanes gather consecutive x entries (coalesced) and the dictionary loads of a wave
// whose 64 rows share one class are wave-uniform (scalar loads).
// ------------------------------------------------------------------------------------------------
struct RowClassDev {
  const int* firstcol;          // n_rows; nullptr: first column = row + cls_delta[class] (square grid operators)
  const unsigned short* cls;    // n_rows
  const int* cls_ptr;           // ncls + 1
  const int* cls_off;           // dictionary: column offset from the row's first column
  const double* cls_val;        // dictionary: value
  const int* cls_delta;         // ncls: first column minus row index (only when firstcol == nullptr)
  const double* cls_d;          // ncls: the level's relaxPrec of every row of the class (SMOOTH with v.d == nullptr)
  // a handful of exception rows (class id 0xFFFF) are computed by the last workgroup of the row-class kernel itself,
  // from the CSR arrays; longer lists get their own launch (csr_rows_spmv)
  const int* exc_rows;
  int nexc_inline;              // <= BLK, 0 = none handled in-kernel
  const int* rowptr;
  const int* colidx;
  const double* val;
  int nblocks;                  // ceil(n_rows / RC_ROWS)
  int n_rows;
};

template <int MODE>
__device__ __forceinline__ double rowclass_exception_rows(const RowClassDev& C, const VecArgs& v, int tid) {
  if (tid >= C.nexc_inline) return 0.0;
  const int row = C.exc_rows[tid];
  double pb = 0.0, pd = 0.0, px = 0.0;
  if (MODE == AXPBY) {
    if (v.beta != 0.0) pb = v.beta * v.y[row];
  } else {
    pb = v.b[row];
    if (MODE == SMOOTH) {
      pd = v.d_full[row];
      px = v.xs[row];
    }
  }
  double acc = 0.0;
  for (int k = C.rowptr[row]; k < C.rowptr[row + 1]; ++k) acc += C.val[k] * v.x[C.colidx[k]];
  const double outv = epilogue<MODE>(v, row, acc, pb, pd, px);
  v.y[row] = outv;
  return outv * outv;
}

// rows per lane of the plain row-class kernel: 1 measured best once the staged variants serve the square operators
// (profiles/r01_nt_ab.md: transfer operators -4 %, sharded path -2 %)
#ifndef MG_RC_RPT
#define MG_RC_RPT 1
#endif
constexpr int RC_RPT = MG_RC_RPT;       // rows per lane: RC_RPT independent gather chains in flight
constexpr int RC_ROWS = BLK * RC_RPT;   // rows per workgroup (lane t holds rows t, t + BLK, ...)

// PAIR = false: lane t holds row t of its workgroup's 256 (RPT = 1).  PAIR = true: lane t holds rows 2t and 2t+1 of
// 512 - for operators whose classes alternate from row to row (a prolongation: even / odd fine nodes) each class then
// fills a whole waterfall pass instead of half the lanes of two (chosen per operator by the host, build_rowclasses).
template <int MODE, bool EXC, bool PAIR>
__global__ __launch_bounds__(BLK) void csr_rowclass_spmv(RowClassDev C, VecArgs v) {
  constexpr int RPT = PAIR ? 2 : RC_RPT;
  constexpr int RSTRIDE = PAIR ? 1 : BLK;
  __shared__ double red[BLK / 64];
  const int tid = threadIdx.x;
  const int bid = xcd_band(blockIdx.x, C.nblocks);
  const int base = PAIR ? bid * (BLK * 2) + 2 * tid : bid * RC_ROWS + tid;
  int first[RPT], cls[RPT];
  double pb[RPT], pd[RPT], px[RPT], acc[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int row = base + j * RSTRIDE;
    const int rr = row < C.n_rows ? row : C.n_rows - 1;   // dead lanes repeat the last row (never stored)
    first[j] = C.firstcol ? C.firstcol[rr] : rr;            // implicit form: the class delta is added in its pass
    cls[j] = C.cls[rr];
    pb[j] = pd[j] = px[j] = 0.0;
    acc[j] = 0.0;
    if (MODE == AXPBY) {
      if (v.beta != 0.0) pb[j] = v.beta * v.y[rr];
    } else {
      pb[j] = v.b[rr];
      if (MODE == SMOOTH) {
        if (v.d) pd[j] = v.d[rr];                           // else: class-constant, set in the class's pass
        px[j] = v.xs[rr];
      }
    }
  }
  // Waterfall over the distinct classes held by the wave (one for a grid interior; a line end or a coarse/fine
  // parity adds a pass).  A pass serves every lane x slot of class cc with wave-uniform dictionary accesses (scalar
  // loads): k outermost, the gathers of all slots issued back to back.  Slots of another class gather at the
  // leader's base instead (a valid, discarded broadcast read), so nothing is masked.  The membership predicates come
  // from ballot masks, not from `cls == cc`: the compiler would otherwise substitute the per-lane value for the
  // uniform cc inside the branch and fall back to per-lane (vector) dictionary loads.
  const unsigned long long lanebit = 1ull << (tid & 63);
  unsigned long long todo[RPT];
#pragma unroll
  for (int j = 0; j < RPT; ++j) todo[j] = __ballot(cls[j] != 0xFFFF);   // 0xFFFF: exception row (csr_rows_spmv)
  for (;;) {
    int cc = 0, lead = 0;
    bool any = false;
#pragma unroll
    for (int j = RPT - 1; j >= 0; --j)
      if (todo[j]) {   // wave-uniform
        const int l = __builtin_ctzll(todo[j]);
        cc = __builtin_amdgcn_readlane(cls[j], l);
        lead = __builtin_amdgcn_readlane(first[j], l);
        any = true;
      }
    if (!any) break;
    const int delta = C.firstcol ? 0 : C.cls_delta[cc];
    lead += delta;
    double dcc = 0.0;
    if (MODE == SMOOTH && !v.d) dcc = C.cls_d[cc];
    bool in[RPT];
    const double* xb[RPT];
    double a[RPT];
#pragma unroll
    for (int j = 0; j < RPT; ++j) {
      const unsigned long long m = __ballot(cls[j] == cc) & todo[j];
      todo[j] &= ~m;
      in[j] = (m & lanebit) != 0;
      xb[j] = v.x + (in[j] ? first[j] + delta : lead);
      if (MODE == SMOOTH && !v.d && in[j]) pd[j] = dcc;
      a[j] = 0.0;
    }
    const int s = C.cls_ptr[cc], e = C.cls_ptr[cc + 1];
    int k = s;
    for (; k + 3 < e; k += 4) {
      const int o0 = C.cls_off[k], o1 = C.cls_off[k + 1], o2 = C.cls_off[k + 2], o3 = C.cls_off[k + 3];
      const double a0 = C.cls_val[k], a1 = C.cls_val[k + 1], a2 = C.cls_val[k + 2], a3 = C.cls_val[k + 3];
      double x0[RPT], x1[RPT], x2[RPT], x3[RPT];
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        x0[j] = xb[j][o0];
        x1[j] = xb[j][o1];
        x2[j] = xb[j][o2];
        x3[j] = xb[j][o3];
      }
#pragma unroll
      for (int j = 0; j < RPT; ++j) {
        a[j] += a0 * x0[j];
        a[j] += a1 * x1[j];
        a[j] += a2 * x2[j];
        a[j] += a3 * x3[j];
      }
    }
    for (; k < e; ++k) {
      const int o0 = C.cls_off[k];
      const double a0 = C.cls_val[k];
#pragma unroll
      for (int j = 0; j < RPT; ++j) a[j] += a0 * xb[j][o0];
    }
#pragma unroll
    for (int j = 0; j < RPT; ++j)
      if (in[j]) acc[j] = a[j];
  }
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const int row = base + j * RSTRIDE;
    if (row < C.n_rows && cls[j] != 0xFFFF) {
      const double outv = epilogue<MODE>(v, row, acc[j], pb[j], pd[j], px[j]);
      v.y[row] = outv;
      sq += outv * outv;
    }
  }
  if (EXC && blockIdx.x == gridDim.x - 1) sq += rowclass_exception_rows<MODE>(C, v, tid);   // EXC: C.nexc_inline > 0
  if (v.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < BLK / 64; ++w) t += red[w];
      v.sumsq[bid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-class SpMV, every lane walking its own row's class (any operator in row-class form whose dictionary fits LDS:
// at most RL_DCAP entries / RL_NCLS classes).  The waterfall kernel above serves one class per pass with a chain of
// scalar dictionary loads; measured on C2 (profiles/r02_march_ab.md) that chain, not memory, bounds it wherever a
// wavefront holds more than one class (a prolongation alternates classes from row to row, a restriction has 27 entries
// per row).  Here the dictionary lives in LDS as 16-byte records {value, column offset}, a lane reads the records of its
// own class (ascending k: the summation order of the CSR row) four at a time and keeps the four gathers of each of its
// two rows in flight.  No passes, no scalar chain; rows of different classes in one wavefront cost nothing extra.
// ------------------------------------------------------------------------------------------------
constexpr int RL_DCAP = 512;
constexpr int RL_NCLS = 128;
constexpr int RL_ROWS = 2 * BLK;   // rows per workgroup (lane t: rows t and t + BLK)
static_assert(BLK > RL_NCLS && RL_DCAP % BLK == 0, "the lane kernels stage their dictionary with one pass over the classes");
struct LaneDev {
  int ncls, nent, maxlen, nblocks;
};
struct LaneEnt {
  double val;
  int off;
  int pad;
};

// PAIR: lane t holds the ADJACENT rows 2t, 2t+1 of its workgroup's 512 (16-byte loads / stores of the row operands;
// chosen for operators whose classes alternate row by row, i.e. prolongations); else rows t and t + 256.
template <int MODE, bool EXC, bool PAIR>
__global__ __launch_bounds__(BLK) void csr_rowclass_lane_spmv(RowClassDev C, VecArgs v, LaneDev T) {
  __shared__ LaneEnt ent[RL_DCAP];
  __shared__ int ptr[RL_NCLS + 1];
  __shared__ int delta[RL_NCLS];
  __shared__ double dd[RL_NCLS];
  __shared__ double red[BLK / 64];
  const int tid = threadIdx.x;
  const int bid = xcd_band(blockIdx.x, T.nblocks);
  const bool class_d = (MODE == SMOOTH) && !v.d;
  int row[2], s[2], len[2];
  const double* xb[2];
  double pb[2], pd[2], px[2], acc[2];
  bool live[2];
  const int r0 = bid * RL_ROWS + 2 * tid;
  const bool both = PAIR && (r0 + 1 < C.n_rows);
  if (both) {   // 16-byte / 8-byte / 4-byte loads of the two adjacent rows' operands (r0 is even)
    const unsigned int cc = *reinterpret_cast<const unsigned int*>(C.cls + r0);
    const int c0 = (int)(cc & 0xFFFFu), c1 = (int)(cc >> 16);
    int f0 = r0, f1 = r0 + 1;
    if (C.firstcol) {
      const i2_t f = *reinterpret_cast<const i2_t*>(C.firstcol + r0);
      f0 = f.x;
      f1 = f.y;
    }
    row[0] = r0;
    row[1] = r0 + 1;
    live[0] = c0 != 0xFFFF;
    live[1] = c1 != 0xFFFF;
    s[0] = live[0] ? c0 : 0;
    s[1] = live[1] ? c1 : 0;
    xb[0] = v.x + f0;
    xb[1] = v.x + f1;
    pb[0] = pb[1] = pd[0] = pd[1] = px[0] = px[1] = 0.0;
    acc[0] = acc[1] = 0.0;
    if (MODE == AXPBY) {
      if (v.beta != 0.0) {
        const d2_t yy = *reinterpret_cast<const d2_t*>(v.y + r0);
        pb[0] = v.beta * yy.x;
        pb[1] = v.beta * yy.y;
      }
    } else {
      const d2_t bb = *reinterpret_cast<const d2_t*>(v.b + r0);
      pb[0] = bb.x;
      pb[1] = bb.y;
      if (MODE == SMOOTH) {
        if (v.d) {
          const d2_t t = *reinterpret_cast<const d2_t*>(v.d + r0);
          pd[0] = t.x;
          pd[1] = t.y;
        }
        const d2_t xx = *reinterpret_cast<const d2_t*>(v.xs + r0);
        px[0] = xx.x;
        px[1] = xx.y;
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      row[j] = PAIR ? r0 + j : bid * RL_ROWS + j * BLK + tid;
      const bool in = row[j] < C.n_rows;
      const int rr = in ? row[j] : C.n_rows - 1;
      const int cls = C.cls[rr];
      live[j] = in && cls != 0xFFFF;                          // 0xFFFF: exception row (csr_rows_spmv)
      const int first = C.firstcol ? C.firstcol[rr] : rr;
      pb[j] = pd[j] = px[j] = 0.0;
      acc[j] = 0.0;
      if (MODE == AXPBY) {
        if (v.beta != 0.0) pb[j] = v.beta * v.y[rr];
      } else {
        pb[j] = v.b[rr];
        if (MODE == SMOOTH) {
          if (v.d) pd[j] = v.d[rr];
          px[j] = v.xs[rr];
        }
      }
      s[j] = live[j] ? cls : 0;                               // (class id for now; resolved after the barrier)
      xb[j] = v.x + first;
    }
  }
  // the dictionary into LDS - BEHIND the loads of the row operands above (all in flight together) and with every load issued
  // before the first LDS write (round 3; as three load-store loops in front of the row loads this was a chain of 4-5 round trips)
  {
    constexpr int ND = RL_DCAP / BLK;                          // nent <= RL_DCAP, ncls <= RL_NCLS < BLK
    LaneEnt e[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      const int ii = i < T.nent ? i : 0;
      e[u].val = C.cls_val[ii];
      e[u].off = C.cls_off[ii];
      e[u].pad = 0;
    }
    const int pi = tid <= T.ncls ? tid : 0, ci = tid < T.ncls ? tid : 0;
    const int pv = C.cls_ptr[pi];
    const int dv = C.firstcol ? 0 : C.cls_delta[ci];
    const double ddv = class_d ? C.cls_d[ci] : 0.0;
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      if (i < T.nent) ent[i] = e[u];
    }
    if (tid <= T.ncls) ptr[tid] = pv;
    if (tid < T.ncls) {
      delta[tid] = dv;
      dd[tid] = ddv;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cq = s[j];
    s[j] = ptr[cq];
    len[j] = live[j] ? ptr[cq + 1] - s[j] : 0;
    xb[j] += delta[cq];
    if (class_d) pd[j] = dd[cq];
  }
  // gathers per row in flight.  A restriction row has 27 entries: with 4 at a time a lane waits out 7 round trips to a source
  // that the preceding pass has just pushed out of the caches; 9 at a time (3 round trips, 40 more registers) takes the fine
  // restriction from 73.7 to 66.0 us and the step from 0.6274 to 0.6170 ms (profiles/r03_dead_ends.md, the one positive entry);
  // 14 at a time is slower again (occupancy)
#ifndef MG_LANE_GROUP
#define MG_LANE_GROUP 9
#endif
#ifndef MG_LANE_GROUP_A
#define MG_LANE_GROUP_A 4
#endif
  constexpr int LG = (MODE == AXPBY && !PAIR) ? MG_LANE_GROUP : MG_LANE_GROUP_A;
  for (int k = 0; k < T.maxlen; k += LG) {
    double g[2][LG];
    int id[2][LG];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < LG; ++u) {
        id[j][u] = s[j] + min(k + u, len[j] > 0 ? len[j] - 1 : 0);
        g[j][u] = (k + u < len[j]) ? xb[j][ent[id[j][u]].off] : 0.0;
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < LG; ++u) {
        const double t = acc[j] + ent[id[j][u]].val * g[j][u];
        acc[j] = (k + u < len[j]) ? t : acc[j];
      }
  }
  // AXPBY with v.y2: also write d[row]*out - the restriction bc = R r hands the coarse level its first update x = d.*bc
  // (relax from x = 0: MGcycle.jl:134 with r = b), so that level needs no dscale launch
  const bool scale2 = (MODE == AXPBY) && v.y2 != nullptr;
  double sq = 0.0;
  if (both && live[0] && live[1]) {
    d2_t o;
    o.x = epilogue<MODE>(v, row[0], acc[0], pb[0], pd[0], px[0]);
    o.y = epilogue<MODE>(v, row[1], acc[1], pb[1], pd[1], px[1]);
    *reinterpret_cast<d2_t*>(v.y + r0) = o;
    if (scale2) {
      v.y2[r0] = v.d_full[r0] * o.x;
      v.y2[r0 + 1] = v.d_full[r0 + 1] * o.y;
    }
    sq = o.x * o.x + o.y * o.y;
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (live[j]) {
        const double outv = epilogue<MODE>(v, row[j], acc[j], pb[j], pd[j], px[j]);
        v.y[row[j]] = outv;
        if (scale2) v.y2[row[j]] = v.d_full[row[j]] * outv;
        sq += outv * outv;
      }
  }
  if (EXC && blockIdx.x == gridDim.x - 1) sq += rowclass_exception_rows<MODE>(C, v, tid);   // EXC: C.nexc_inline > 0
  if (v.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < BLK / 64; ++w) t += red[w];
      v.sumsq[bid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Prolongation-shaped row-class product with the SOURCE staged in LDS (round 2): y = alpha*M*x + beta*y for an operator
// whose rows are a fine grid (PF rows per plane) and whose columns are a coarse grid (PC columns per plane), every row
// gathering from at most two consecutive coarse planes inside a narrow in-plane window - a prolongation: x += P*xc
// (MGcycle.jl:90).  csr_rowclass_lane_spmv spends its time waiting for 1..8 gathers of xc per fine row (L2 / Infinity
// Cache).  Here a workgroup owns WP_ROWS consecutive rows of ONE fine plane, stages the two coarse-plane windows its rows
// read (2*W doubles, coalesced loads) in LDS, and the per-lane class walk reads LDS; the per-row first column is replaced
// by its 16-bit index inside the window (wf).  All tables are derived from the DATA by the host (build_winp), which also
// checks that every column of every row falls inside the staged windows; nothing here knows what a prolongation is.
// Same products in the same order and the same epilogue as the lane kernel.
// ------------------------------------------------------------------------------------------------
#ifndef MG_WP_T
#define MG_WP_T 512
#endif
constexpr int WP_T = MG_WP_T;        // threads per workgroup
constexpr int WP_ROWS = 4 * WP_T;   // rows of one fine plane per workgroup (lane t: rows t, t + WP_T, t + 2 WP_T, t + 3 WP_T)
struct WinPDev {
  const unsigned short* wf;   // per row: index of the row's first column inside the first staged window
  const int* cz0;             // per fine plane: the coarse plane of its rows' first columns, bit 30 set when the plane's rows read ONLY that coarse plane
  const int* wlo;             // per chunk: in-plane index of the first staged coarse entry
  const int* code;            // per dictionary entry: dzc*W + rest (column offset = dzc*PC + rest)
  int PF, nplanes, PC, W, chunks, nblocks, n_cols, ncls, nent, maxlen;
};

template <int DUMMY>
__global__ __launch_bounds__(WP_T) void csr_rowclass_winp_spmv(RowClassDev C, VecArgs v, WinPDev T) {
  extern __shared__ double win[];                                  // [2*W] coarse windows | dictionary
  LaneEnt* ent = reinterpret_cast<LaneEnt*>(win + 2 * T.W);        // [ncls][maxlen] {value or 0, BYTE code}: padded dictionary
  const int tid = threadIdx.x;
  const int bid = xcd_band(blockIdx.x, T.nblocks);
  const int z = bid / T.chunks, c = bid - z * T.chunks;
  int row[4], cq[4], wfi[4];
  bool in[4];
  double pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = c * WP_ROWS + j * WP_T + tid;
    in[j] = p < T.PF;
    row[j] = z * T.PF + p;
    const int rr = in[j] ? row[j] : C.n_rows - 1;
    cq[j] = C.cls[rr];
    wfi[j] = T.wf[rr];
    pb[j] = (v.beta != 0.0) ? v.beta * v.y[rr] : 0.0;
    if (cq[j] == 0xFFFF) {   // an exception row (reads halo columns): csr_rows_spmv computes it
      in[j] = false;
      cq[j] = 0;
    }
  }
  // Both staging loops issue ALL their loads before the first LDS write (round 3): as load-store loops every load was waited
  // for before its ds_write - a chain of up to 8 + 2 x 2 round trips per workgroup.
  {
    const int czz = T.cz0[z];
    const long long base0 = (long long)(czz & 0x3FFFFFFF) * T.PC + T.wlo[c];
    const int nstage = (czz & 0x40000000) ? T.W : 2 * T.W;   // (a plane whose rows read one coarse plane: one window)
    constexpr int NU = (4096 + WP_T - 1) / WP_T;               // (W <= 2048: build_winp)
    double st[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = tid + u * WP_T;
      const bool second = i >= T.W;
      long long col = base0 + (second ? (long long)T.PC + (i - T.W) : (long long)i);
      col = col < 0 ? 0 : (col > T.n_cols - 1 ? T.n_cols - 1 : col);
      st[u] = (i < nstage) ? v.x[col] : 0.0;
    }
    // records beyond a class's length: value 0 at the class's first entry (they add +-0: no clamps, no predicated additions)
    constexpr int ND = (1024 + WP_T - 1) / WP_T;               // (ncls * maxlen <= 1024: build_winp)
    int s0[ND], ln[ND], kk[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * WP_T;
      const bool on = i < T.ncls * T.maxlen;
      const int cc = on ? i / T.maxlen : 0;
      kk[u] = on ? i - cc * T.maxlen : 0;
      s0[u] = C.cls_ptr[cc];
      ln[u] = C.cls_ptr[cc + 1] - s0[u];
    }
    LaneEnt e[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      e[u].val = kk[u] < ln[u] ? C.cls_val[s0[u] + kk[u]] : 0.0;
      e[u].off = T.code[s0[u] + (kk[u] < ln[u] ? kk[u] : 0)] * 8;
      e[u].pad = ln[u];   // (the class's length rides in every record)
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int i = tid + u * WP_T;
      if (i < nstage) win[i] = st[u];
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * WP_T;
      if (i < T.ncls * T.maxlen) ent[i] = e[u];
    }
  }
  __syncthreads();
  const LaneEnt* rp[4];
  const char* wb[4];
  double acc[4];
  int lenmax = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    rp[j] = ent + cq[j] * T.maxlen;
    wb[j] = reinterpret_cast<const char*>(win) + wfi[j] * 8;
    acc[j] = 0.0;
    if (in[j]) lenmax = max(lenmax, rp[j][0].pad);
  }
  for (int k = 0; k < T.maxlen; k += 4) {
    if (k > 0 && __ballot(lenmax > k) == 0ull) break;   // no row of this wavefront is longer (padded records would add 0)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      LaneEnt e[4];
      double g[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) e[u] = rp[j][min(k + u, T.maxlen - 1)];
#pragma unroll
      for (int u = 0; u < 4; ++u) g[u] = *reinterpret_cast<const double*>(wb[j] + e[u].off);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (k + u < T.maxlen) acc[j] = acc[j] + e[u].val * g[u];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (in[j]) v.y[row[j]] = epilogue<AXPBY>(v, row[j], acc[j], pb[j], 0.0, 0.0);
}

// ------------------------------------------------------------------------------------------------
// Row-class SpMV with LDS windows (square operators in the implicit-first form).
// The plain row-class kernel above is bound by L2->L1 line traffic: every dictionary entry is its own gather and
// the y+-1 / z+-1 neighbours of a grid row arrive as separate, unaligned 512-byte requests (~25 cache lines of x
// per 64 rows; profiles/r01_nt_ab.md).  Here a workgroup owns RW_ROWS consecutive rows and, for the class of its
// middle row (the "staged class": a grid interior), loads the UNION of what the member rows gather -
// entry k of the class reads x[row + delta + off_k], i.e. the window [rmin + delta + off_k, rmax + delta + off_k];
// overlapping windows (x+-1 and y+-1 neighbours of a few consecutive grid lines) merge - into LDS once, with
// contiguous coalesced loads (~14 lines of x per 64 rows), and the member rows then read LDS.  Rows of any other
// class (grid-line ends, boundary planes) take the waterfall path of the kernel above.  Purely a data-movement
// change: products and their order are unchanged.  Nothing here knows about grids: the staged class is the operator's
// most frequent one and the kernel is used only when its windows fit (RW_CAP doubles, RW_MAXINT windows, RW_MAXLEN
// entries; decided by the host at upload).
// ------------------------------------------------------------------------------------------------
#ifndef MG_RW_RPT
#define MG_RW_RPT 2
#endif
constexpr int RW_RPT = MG_RW_RPT;
constexpr int RW_ROWS = BLK * RW_RPT;   // 1024 rows per workgroup
constexpr int RW_CAP = 6144;            // doubles of x staged per workgroup (48 KiB)
constexpr int RW_MAXLEN = 32;           // dictionary entries of the staged class
constexpr int RW_MAXINT = 8;            // disjoint windows
// Window layout, computed once on the host (build_rowclasses) from the operator's most frequent class for
// W = RW_ROWS:  meta[0] that class, [1] number of windows, [2] LDS index of shift 0 (the row's own x) or -1,
// [3] doubles staged, [4] unused, then per window {first shift, length, LDS base}.
// A dictionary entry of ANY class with shift sh = delta_c + off reads x[row + sh]; when sh is one of the staged
// shifts its LDS index is cls_lb[entry] (x[row + sh] <-> win[cls_lb + (row - r0)], r0 = first row of the
// workgroup), else cls_lb = -1 and the entry gathers from global memory.  Boundary classes of a grid operator are
// subsets of the interior class's shifts, so in practice every gather is served from LDS.
constexpr int RW_META_HDR = 5;

template <int MODE, bool EXC>
__global__ __launch_bounds__(BLK) void csr_rowclass_window_spmv(RowClassDev C, VecArgs v, const int* __restrict__ meta,
                                                                const int* __restrict__ cls_lb, int nblocks_w,
                                                                int n_cols) {
  extern __shared__ double win[];
  __shared__ double red[BLK / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bid = xcd_band(blockIdx.x, nblocks_w);
  const int r0 = bid * RW_ROWS;
  const int cd = meta[0], nwin = meta[1], lb0 = meta[2];
  const int* __restrict__ ivs = meta + RW_META_HDR;
  int row[RW_RPT], cls[RW_RPT], idx[RW_RPT];
  double pb[RW_RPT], pd[RW_RPT], px[RW_RPT], acc[RW_RPT];
  bool live[RW_RPT];
#pragma unroll
  for (int j = 0; j < RW_RPT; ++j) {
    row[j] = r0 + j * BLK + tid;
    live[j] = row[j] < C.n_rows;
    const int rr = live[j] ? row[j] : C.n_rows - 1;
    idx[j] = rr - r0;
    cls[j] = C.cls[rr];
    live[j] = live[j] && cls[j] != 0xFFFF;   // exception rows are computed by csr_rows_spmv
    pb[j] = pd[j] = px[j] = 0.0;
    acc[j] = 0.0;
    if (MODE == AXPBY) {
      if (v.beta != 0.0) pb[j] = v.beta * v.y[rr];
    } else {
      pb[j] = v.b[rr];
      if (MODE == SMOOTH && v.d) pd[j] = v.d[rr];
    }
  }
  // stage the windows (clamped at the ends of x: no row reads a clamped slot through a staged shift)
  const int delta_cd = C.cls_delta[cd];
  for (int m = 0; m < nwin; ++m) {
    const int g0 = r0 + delta_cd + ivs[3 * m], L = ivs[3 * m + 1], B = ivs[3 * m + 2];
    for (int i = tid; i < L; i += BLK) win[B + i] = v.x[min(max(g0 + i, 0), n_cols - 1)];
  }
  __syncthreads();
  if (MODE == SMOOTH) {
    const bool own = (lb0 >= 0) && (v.xs == v.x);
#pragma unroll
    for (int j = 0; j < RW_RPT; ++j)
      if (live[j]) px[j] = own ? win[lb0 + idx[j]] : v.xs[row[j]];
  }
  // waterfall over the classes the wave holds, as in csr_rowclass_spmv, with the gathers served from LDS
  const unsigned long long lanebit = 1ull << lane;
  unsigned long long todo[RW_RPT];
#pragma unroll
  for (int j = 0; j < RW_RPT; ++j) todo[j] = __ballot(live[j]);
  for (;;) {
    int cc = 0, lead = 0;
    bool any = false;
#pragma unroll
    for (int j = RW_RPT - 1; j >= 0; --j)
      if (todo[j]) {
        const int l = __builtin_ctzll(todo[j]);
        cc = __builtin_amdgcn_readlane(cls[j], l);
        lead = __builtin_amdgcn_readlane(row[j], l);
        any = true;
      }
    if (!any) break;
    const int delta = C.cls_delta[cc];
    lead += delta;
    double dcc = 0.0;
    if (MODE == SMOOTH && !v.d) dcc = C.cls_d[cc];
    bool in[RW_RPT];
    double a[RW_RPT];
#pragma unroll
    for (int j = 0; j < RW_RPT; ++j) {
      const unsigned long long m = __ballot(cls[j] == cc) & todo[j];
      todo[j] &= ~m;
      in[j] = (m & lanebit) != 0;
      if (MODE == SMOOTH && !v.d && in[j]) pd[j] = dcc;
      a[j] = 0.0;
    }
    const int s = C.cls_ptr[cc], e = C.cls_ptr[cc + 1];
    for (int k = s; k < e; ++k) {
      const int lb = cls_lb[k];
      const double a0 = C.cls_val[k];
      if (lb >= 0) {   // wave-uniform
#pragma unroll
        for (int j = 0; j < RW_RPT; ++j) a[j] += a0 * win[lb + idx[j]];   // rows of another class: a valid slot, unused
      } else {
        const int o0 = C.cls_off[k];
#pragma unroll
        for (int j = 0; j < RW_RPT; ++j) a[j] += a0 * v.x[(in[j] ? row[j] + delta : lead) + o0];
      }
    }
#pragma unroll
    for (int j = 0; j < RW_RPT; ++j)
      if (in[j]) acc[j] = a[j];
  }
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < RW_RPT; ++j) {
    if (live[j]) {
      const double outv = epilogue<MODE>(v, row[j], acc[j], pb[j], pd[j], px[j]);
      v.y[row[j]] = outv;
      sq += outv * outv;
    }
  }
  if (EXC && blockIdx.x == gridDim.x - 1) sq += rowclass_exception_rows<MODE>(C, v, tid);   // EXC: C.nexc_inline > 0
  if (v.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < BLK / 64; ++w) t += red[w];
      v.sumsq[bid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-class SpMV on plane tiles (square operators in the implicit-first form whose rows the caller declared to be an
// n1 x n2 x n3 grid, mg_set_grid_hint; P = n1*n2 rows per plane).
// Measured (profiles/r01_nt_ab.md): the row-class sweeps run at ~5.5 TB/s of L1 FILLS, HBM and L2 hits alike - each
// CU keeps a bounded number of cache-line fills in flight - so what is left to win is the fill volume.  In row order a
// row's z+-1 neighbours are P rows away: every x line is filled three times per sweep (once as the row's own
// plane, twice as a neighbour plane).  Here a 1024-thread workgroup owns the SAME RT_CR-row chunk of RT_NP consecutive
// planes (slot j of a lane = plane j) and stages RT_NP + 2 slabs of x (chunk +- HALO rows, planes -1 .. RT_NP) in LDS:
// 6 slabs serve 4 planes, i.e. ~2.3 fills of x per row instead of 4.  Every dictionary entry of every class whose
// shift (delta + offset) is dz*P + rest with |dz| <= 1, |rest| <= HALO reads LDS (index precomputed by the host in
// tile_lb), anything else gathers from global memory, so correctness never depends on the hint being "true": it is
// a row partition plus a cache.  Products and summation order are unchanged.
// ------------------------------------------------------------------------------------------------
#ifndef MG_RT_CR
#define MG_RT_CR 1024
#endif
#ifndef MG_RT_NP
#define MG_RT_NP 4
#endif
constexpr int RT_NP = MG_RT_NP;   // planes per workgroup = rows per lane
constexpr int RT_CR = MG_RT_CR;   // rows of a plane per workgroup = threads per workgroup

struct TileDev {
  const int* tile_lb;   // per dictionary entry: LDS index of (slot 0, lane 0) or -1
  int P;                // rows per plane
  int nplanes;          // n_rows == nplanes * P
  int halo;             // slab = CR + 2*halo entries of x
  int chunks;           // ceil(P / CR)
  int nblocks;          // ceil(nplanes / RT_NP) * chunks
  int n_cols;
  // per-lane walk (round 2): every lane walks the records of its own rows' class once for its RT_NP rows - the planes of
  // a lane share the in-plane position and, away from the first and last plane, the class - from a dictionary padded to
  // maxlen records per class in LDS (value 0 beyond a class's length: no predication).  Set by the host when every
  // dictionary entry is a staged shift and ncls * maxlen <= RT_LCAP.
  int lane;
  int ncls, maxlen;
};
constexpr int RT_LCAP = 1024;   // records of the padded dictionary (16 B each)
struct TileRec {
  double val;
  int off8;   // byte offset of the gathered entry relative to (slot 0, lane 0)
  int pad;
};

// CR: rows of a plane per workgroup = threads per workgroup (1024; 256 for levels whose 1024-row tiles would not fill the chip)
template <int MODE, bool EXC, int CR>
__global__ __launch_bounds__(CR, CR >= 1024 ? 8 : 4) void csr_rowclass_tile_spmv(RowClassDev C, VecArgs v, TileDev T) {
  extern __shared__ double win[];
  __shared__ double red[CR / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int bid = xcd_band(blockIdx.x, T.nblocks);
  const int g = bid / T.chunks, c = bid - g * T.chunks;
  const int SL = CR + 2 * T.halo;
  const int pl0 = g * RT_NP;                           // first plane of the tile
  const int inplane = c * CR + tid;                 // position of the lane's rows inside their planes
  int row[RT_NP], cls[RT_NP];
  double pb[RT_NP], pd[RT_NP], acc[RT_NP];   // (the row's own x is read from LDS in the epilogue: registers are tight)
  bool live[RT_NP];
#pragma unroll
  for (int j = 0; j < RT_NP; ++j) {
    live[j] = (pl0 + j < T.nplanes) && (inplane < T.P);
    row[j] = (pl0 + j) * T.P + inplane;
    const int rr = live[j] ? row[j] : C.n_rows - 1;
    cls[j] = C.cls[rr];
    live[j] = live[j] && cls[j] != 0xFFFF;   // exception rows are computed by csr_rows_spmv
    pb[j] = pd[j] = 0.0;
    acc[j] = 0.0;
    if (MODE == AXPBY) {
      if (v.beta != 0.0) pb[j] = v.beta * v.y[rr];
    } else {
      pb[j] = v.b[rr];
      if ((MODE == SMOOTH || (MODE == RESID && v.y2)) && v.d) pd[j] = v.d[rr];
    }
  }
  const bool class_dl = (MODE == SMOOTH || (MODE == RESID && v.y2)) && !v.d;
  // stage slabs q = 0 .. RT_NP+1 <-> planes pl0-1 .. pl0+RT_NP, rows [c*CR - halo, c*CR + CR + halo).  ALL loads of a lane
  // are issued before the first LDS write (round 3): written as one load-store loop the compiler waits for every load before
  // its ds_write - up to 12 round trips in a row per workgroup, which is what the level-2 / level-3 launches spent their
  // time on (SQ counters: 66 % of the wave cycles waiting)
  {
    constexpr int NQ = RT_NP + 2, NU = 2;        // (halo <= CR/2: two entries per lane and slab; a longer slab takes the loop below)
    double st[NQ][NU];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const long long g0 = (long long)(pl0 + q - 1) * T.P + c * CR - T.halo;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int i = tid + u * CR;
        long long gi = g0 + i;
        gi = gi < 0 ? 0 : (gi > T.n_cols - 1 ? T.n_cols - 1 : gi);
        st[q][u] = (i < SL) ? v.x[gi] : 0.0;
      }
    }
    // the padded dictionary (T.lane): its loads go out behind the slab loads, before any LDS write
    TileRec* drec = reinterpret_cast<TileRec*>(win + (RT_NP + 2) * SL);     // [ncls][maxlen] (T.lane only)
    double* ddl = reinterpret_cast<double*>(drec + (T.lane ? T.ncls * T.maxlen : 0));
    constexpr int ND = (RT_LCAP + CR - 1) / CR;                            // ncls * maxlen <= RT_LCAP
    TileRec dr[ND];
    double ddv = 0.0;
    if (T.lane) {
      int ds[ND], dl[ND], dk[ND];
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int i = tid + u * CR;
        const bool on = i < T.ncls * T.maxlen;
        const int cc = on ? i / T.maxlen : 0;
        dk[u] = on ? i - cc * T.maxlen : 0;
        ds[u] = C.cls_ptr[cc];
        dl[u] = C.cls_ptr[cc + 1] - ds[u];
      }
      if (class_dl && tid < T.ncls) ddv = C.cls_d[tid];
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        dr[u].val = dk[u] < dl[u] ? C.cls_val[ds[u] + dk[u]] : 0.0;
        dr[u].off8 = T.tile_lb[ds[u] + (dk[u] < dl[u] ? dk[u] : 0)] * 8;
        dr[u].pad = 0;
      }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int i = tid + u * CR;
        if (i < SL) win[q * SL + i] = st[q][u];
      }
    for (int q = 0; q < NQ; ++q) {
      const long long g0 = (long long)(pl0 + q - 1) * T.P + c * CR - T.halo;
      for (int i = tid + NU * CR; i < SL; i += CR) {
        long long gi = g0 + i;
        gi = gi < 0 ? 0 : (gi > T.n_cols - 1 ? T.n_cols - 1 : gi);
        win[q * SL + i] = v.x[gi];
      }
    }
    if (T.lane) {
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int i = tid + u * CR;
        if (i < T.ncls * T.maxlen) drec[i] = dr[u];
      }
      for (int i = tid; i < T.ncls; i += CR) ddl[i] = (i == tid) ? ddv : (class_dl ? C.cls_d[i] : 0.0);
    }
  }
  TileRec* drec = reinterpret_cast<TileRec*>(win + (RT_NP + 2) * SL);     // [ncls][maxlen] (T.lane only)
  double* ddl = reinterpret_cast<double*>(drec + (T.lane ? T.ncls * T.maxlen : 0));
  __syncthreads();
  const unsigned long long lanebit = 1ull << lane;
  unsigned long long todo[RT_NP];
#pragma unroll
  for (int j = 0; j < RT_NP; ++j) todo[j] = __ballot(live[j]);
  if (T.lane) {
    // the class of the lane's first live row; do all its live rows share it?
    int cref = 0;
    bool anylive = false, mixed = false;
#pragma unroll
    for (int j = RT_NP - 1; j >= 0; --j)
      if (live[j]) {
        cref = cls[j];
        anylive = true;
      }
#pragma unroll
    for (int j = 0; j < RT_NP; ++j) mixed = mixed || (live[j] && cls[j] != cref);
    const char* wb = reinterpret_cast<const char*>(win) + tid * 8;
    const int SL8 = SL * 8;
    if (__ballot(mixed) == 0ull) {
      if (__ballot(anylive) != 0ull) {
        const TileRec* rp = drec + cref * T.maxlen;
#pragma unroll 3
        for (int k = 0; k < T.maxlen; ++k) {
          const TileRec r = rp[k];
          const char* a0 = wb + r.off8;
#pragma unroll
          for (int j = 0; j < RT_NP; ++j) acc[j] = acc[j] + r.val * *reinterpret_cast<const double*>(a0 + j * SL8);
        }
      }
    } else {   // (a tile that holds the first or the last plane: the rows of a lane differ in class)
#pragma unroll
      for (int j = 0; j < RT_NP; ++j) {
        const TileRec* rp = drec + (live[j] ? cls[j] : 0) * T.maxlen;
        const char* a0 = wb + j * SL8;
        for (int k = 0; k < T.maxlen; ++k) {
          const TileRec r = rp[k];
          acc[j] = acc[j] + r.val * *reinterpret_cast<const double*>(a0 + r.off8);
        }
      }
    }
    if (class_dl) {
#pragma unroll
      for (int j = 0; j < RT_NP; ++j)
        if (live[j]) pd[j] = ddl[cls[j]];
    }
#pragma unroll
    for (int j = 0; j < RT_NP; ++j) todo[j] = 0ull;
  }
  for (;;) {
    int cc = 0, lead = 0;
    bool any = false;
#pragma unroll
    for (int j = RT_NP - 1; j >= 0; --j)
      if (todo[j]) {
        const int l = __builtin_ctzll(todo[j]);
        cc = __builtin_amdgcn_readlane(cls[j], l);
        lead = __builtin_amdgcn_readlane(row[j], l);
        any = true;
      }
    if (!any) break;
    const int delta = C.cls_delta[cc];
    lead += delta;
    double dcc = 0.0;
    const bool class_d = (MODE == SMOOTH || (MODE == RESID && v.y2)) && !v.d;
    if (class_d) dcc = C.cls_d[cc];
    bool in[RT_NP];
    double a[RT_NP];
#pragma unroll
    for (int j = 0; j < RT_NP; ++j) {
      const unsigned long long m = __ballot(cls[j] == cc) & todo[j];
      todo[j] &= ~m;
      in[j] = (m & lanebit) != 0;
      if (class_d && in[j]) pd[j] = dcc;
      a[j] = 0.0;
    }
    const int s = C.cls_ptr[cc], e = C.cls_ptr[cc + 1];
    for (int k = s; k < e; ++k) {
      const int lb = T.tile_lb[k];
      const double a0 = C.cls_val[k];
      if (lb >= 0) {   // wave-uniform
#pragma unroll
        for (int j = 0; j < RT_NP; ++j) a[j] += a0 * win[lb + j * SL + tid];   // other classes: a valid slot, unused
      } else {
        const int o0 = C.cls_off[k];
#pragma unroll
        for (int j = 0; j < RT_NP; ++j) a[j] += a0 * v.x[(in[j] ? row[j] + delta : lead) + o0];
      }
    }
#pragma unroll
    for (int j = 0; j < RT_NP; ++j)
      if (in[j]) acc[j] = a[j];
  }
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < RT_NP; ++j) {
    if (live[j]) {
      double pxj = 0.0;
      if (MODE == SMOOTH) pxj = (v.xs == v.x) ? win[(j + 1) * SL + T.halo + tid] : v.xs[row[j]];
      const double outv = epilogue<MODE>(v, row[j], acc[j], pb[j], pd[j], pxj);
      if (MODE != RESID || v.y) v.y[row[j]] = outv;   // (the solve loop needs only ||r|| and x + d.*r: y may be null)
      if (MODE == RESID && v.y2) v.y2[row[j]] = win[(j + 1) * SL + T.halo + tid] + pd[j] * outv;   // x + d.*r
      sq += outv * outv;
    }
  }
  if (EXC && blockIdx.x == gridDim.x - 1) sq += rowclass_exception_rows<MODE>(C, v, tid);   // EXC: C.nexc_inline > 0
  if (v.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < CR / 64; ++w) t += red[w];
      v.sumsq[bid] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-class SpMV marching along z (square operators in the implicit-first form with a grid hint; P = n1*n2 rows per
// plane).  Measured on C2's fine level (profiles/r02_march_ab.md): the plane-tile kernel above spends ~105 us in its
// load phases (8-byte loads, six slabs per four planes) and ~90 us in the class passes (a chain of scalar dictionary
// loads -> LDS read -> FMA per entry), overlapping only across the two workgroups of a CU.  Here
//  * a workgroup owns ONE in-plane chunk of RM_C rows and walks a run of consecutive planes with a ring of four slabs
//    in LDS: every plane of x is staged once per chunk (in-plane halo rows come from the neighbour chunk's lines in
//    L2), with 16-byte loads (slabs start at an even global index; the 0/1 entry shift is a per-plane scalar);
//  * the loads of plane z+3 are in flight in registers while plane z is computed; everything loaded is consumed at the
//    top of the next iteration and the stores of a plane are issued one iteration late, so the one vector-memory wait
//    per iteration never stalls on anything younger than a plane's worth of work; one barrier per plane;
//  * the class dictionary lives in LDS as 16-byte records {value, LDS index code} and every LANE walks the entries of
//    its own row's class (ascending k: the summation order of the CSR row): no waterfall passes, no scalar chain -
//    boundary rows cost nothing extra, and the compiler can keep several LDS reads in flight;
//  * the (chunk, plane) items are dealt to the workgroups as equal contiguous ranges of the chunk-major list, so a
//    launch is exactly one balanced round of the resident workgroups (a range that runs off the end of a chunk
//    continues at plane 0 of the next one).
// (Rounds 2-5 carried an optional prologue that staged x + P*xc instead of x - the coarse-grid correction fused into the first
// post-smoothing sweep; measured slower than the prolongation kernel + sweep in every round, off by default since round 2, retired in round 6.)
// The host selects this kernel only when EVERY dictionary entry of every class is a staged shift dz*P + rest with
// |dz| <= 1, |rest| <= halo (true for grid operators; anything else keeps the plane-tile kernel and its global gathers).
// ------------------------------------------------------------------------------------------------
constexpr int RM_C = 1024;    // rows of a plane per workgroup = threads per workgroup
constexpr int RM_RING = 4;    // slabs in LDS: planes z-1, z, z+1 in use, z+2 being written
constexpr int RM_DCAP = 512;  // dictionary entries of the operator
constexpr int RM_NCLS = 128;  // classes of the operator
constexpr int RM_DICT_BYTES = 16 * RM_DCAP + 8 * RM_NCLS + 4 * (RM_NCLS + 4);

struct MarchDev {
  const int* lb;     // per dictionary entry: ((rest + halo) << 2) | (dz + 1)   (all >= 0: host check)
  int P;             // rows per plane
  int nplanes;       // n_rows == nplanes * P
  int halo;          // slab = RM_C + 2*halo entries of x (+ alignment pad)
  int chunks;        // ceil(P / RM_C)
  int nblocks;       // workgroups; each gets chunks*nplanes/nblocks consecutive (chunk, plane) items
  int n_cols;
  int ncls, nent;    // classes / dictionary entries (<= RM_NCLS / RM_DCAP)
  int maxlen;        // longest class
};

struct MarchEnt {   // one dictionary record in LDS
  double val;
  int code;         // LDS index code ((rest + halo) << 2 | dz + 1)
  int pad;
};

// one 16-byte pair of the slab: global entries e0, e0 + 1 (e0 even); clamped scalar loads at the ends of the vector
// (no live row reads a clamped slot through a staged shift)
__device__ __forceinline__ d2_t march_load_pair(const double* __restrict__ x, long long e0, bool act, int n_cols) {
  d2_t r = d2_t{0.0, 0.0};
  if (act) {
    if (e0 >= 0 && e0 + 1 < n_cols) {
      r = *reinterpret_cast<const d2_t*>(x + e0);
    } else {
      const long long n1 = n_cols - 1;
      const long long c0 = e0 < 0 ? 0 : (e0 > n1 ? n1 : e0), c1 = e0 + 1 < 0 ? 0 : (e0 + 1 > n1 ? n1 : e0 + 1);
      r.x = x[c0];
      r.y = x[c1];
    }
  }
  return r;
}

// The same pair with ONE 16-byte load in every lane, no branch and NO fix-up behind the load (a select on the loaded value
// would be a wait right after the issue): the number of vector-memory instructions a lane issues is static (the loop-top
// wait of the marching kernels counts them) and no load waits for an earlier one's destination registers - the two-path
// form above makes the compiler put `s_waitcnt vmcnt(1)` in front of the second path's load, i.e. every iteration waits for
// the stores of the previous one.  e0 even.  Returns the RAW pair at a clamped address: entries outside [0, n_cols) and
// inactive pairs come back as some finite entries of x (their consumers multiply them by 0 or drop them); when only entry
// e0 exists (n_cols odd, e0 = n_cols - 1) the pair one entry earlier is read and march_pair_fix, applied where the pair is
// CONSUMED, moves it into place.
__device__ __forceinline__ bool march_pair_tail(long long e0, bool act, int n_cols) { return act && e0 == (long long)n_cols - 1; }
__device__ __forceinline__ d2_t march_load_pair_raw(const double* __restrict__ x, long long e0, bool act, int n_cols) {
  const bool in = act && e0 >= 0 && e0 <= (long long)n_cols - 2;
  const long long ec = in ? e0 : (march_pair_tail(e0, act, n_cols) ? e0 - 1 : 0);
  return *reinterpret_cast<const d2_t*>(x + ec);           // (tail: a 16-byte load at an 8-byte aligned address)
}
__device__ __forceinline__ void march_pair_fix(d2_t& v, long long e0, bool act, int n_cols) {
  if (march_pair_tail(e0, act, n_cols)) v.x = v.y;
}
template <int MODE, bool EXC>
__global__ __launch_bounds__(RM_C, 8) void csr_rowclass_march_spmv(RowClassDev C, VecArgs v, MarchDev T) {
  extern __shared__ double win[];
  __shared__ double red[RM_C / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = xcd_band(blockIdx.x, T.nblocks);
  const int SL = RM_C + 2 * T.halo;
  const int SLP = (SL + 3) & ~1;                       // even, >= SL + 1 (the alignment shift)
  const int npair = (SL + 2) >> 1;                     // 16-byte pairs per slab (<= RM_C: host check)
  // ---- LDS: ring of slabs | dictionaries ----------------------------------------------------------------------
  MarchEnt* dent = reinterpret_cast<MarchEnt*>(win + RM_RING * SLP);      // [RM_DCAP]
  double* dd = reinterpret_cast<double*>(dent + RM_DCAP);                 // [RM_NCLS] class relaxPrec
  int* dptr = reinterpret_cast<int*>(dd + RM_NCLS);                       // [RM_NCLS + 1] (+3 pad)
  const bool class_d = (MODE == SMOOTH || (MODE == RESID && v.y2)) && !v.d;
  for (int i = tid; i < T.nent; i += RM_C) {
    MarchEnt e;
    e.val = C.cls_val[i];
    e.code = T.lb[i];
    e.pad = 0;
    dent[i] = e;
  }
  for (int i = tid; i <= T.ncls; i += RM_C) dptr[i] = C.cls_ptr[i];
  for (int i = tid; i < T.ncls; i += RM_C) dd[i] = class_d ? C.cls_d[i] : 0.0;
  const long long tot = (long long)T.chunks * T.nplanes;
  long long it = tot * w / T.nblocks;
  const long long it_end = tot * (w + 1) / T.nblocks;
  const bool pact = tid < npair;
  double sq = 0.0;

  // slab of plane p of chunk c: global entries [a0, a0 + 2*npair), a0 = even floor of p*P + c*RM_C - halo
#define MARCH_G0(c, p) ((long long)(p) * T.P + (long long)(c) * RM_C - T.halo)
#define MARCH_E0(c, p) ((MARCH_G0(c, p) & ~1LL) + 2 * tid)
  // pair -> LDS
#define MARCH_STAGE(slot, val)                                                                                         \
  do {                                                                                                                 \
    const d2_t sv_ = (val);                                                                                            \
    if (pact) {                                                                                                        \
      win[(slot) * SLP + 2 * tid] = sv_.x;                                                                             \
      win[(slot) * SLP + 2 * tid + 1] = sv_.y;                                                                         \
    }                                                                                                                  \
  } while (0)

  __syncthreads();   // dictionaries in place
  while (it < it_end) {
    const int c = (int)(it / T.nplanes);
    const int z0 = (int)(it - (long long)c * T.nplanes);
    const int z1 = (int)((it_end - it) < (long long)(T.nplanes - z0) ? z0 + (it_end - it) : T.nplanes);
    it += z1 - z0;
    const int inplane = c * RM_C + tid;
    const bool rowlive = inplane < T.P;
    // ---- fill the ring: planes z0-1, z0, z0+1; plane z0+2 goes into registers -------------------------------------
    d2_t pre;
    {
      const d2_t q0 = march_load_pair(v.x, MARCH_E0(c, z0 - 1), pact, T.n_cols);
      const d2_t q1 = march_load_pair(v.x, MARCH_E0(c, z0), pact, T.n_cols);
      const d2_t q2 = march_load_pair(v.x, MARCH_E0(c, z0 + 1), pact, T.n_cols);
      MARCH_STAGE(0, q0);
      MARCH_STAGE(1, q1);
      MARCH_STAGE(2, q2);
    }
    pre = march_load_pair(v.x, MARCH_E0(c, z0 + 2), pact, T.n_cols);
    int ncls;
    double npb = 0.0, npd = 0.0;
    {
      const int row = z0 * T.P + inplane;
      const int rr = rowlive ? row : C.n_rows - 1;
      ncls = C.cls[rr];
      if (MODE == AXPBY) {
        if (v.beta != 0.0) npb = v.beta * v.y[rr];
      } else {
        npb = v.b[rr];
        if ((MODE == SMOOTH || (MODE == RESID && v.y2)) && v.d) npd = v.d[rr];
      }
    }
    __syncthreads();
    // Every vector-memory result is consumed at the TOP of an iteration, one full iteration after it was issued, and
    // the stores of a plane are issued at the top of the NEXT iteration: the single wait per iteration then never
    // stalls on an operation younger than one plane's worth of work.
    double st_out = 0.0, st_out2 = 0.0;
    int st_row = -1;
    for (int z = z0; z < z1; ++z) {
      const int q = z - z0 + 1;                        // ring index of plane z (plane z0-1 is 0)
      d2_t cur = pre;
      int cls = ncls;
      double pb = npb, pd = npd;
      asm volatile("" : "+v"(cur.x), "+v"(cur.y), "+v"(cls), "+v"(pb), "+v"(pd));   // the wait of this iteration
      // ---- plane z+2 into its slot (that of plane z-2, last read before the previous barrier) ----------------------
      if (z + 2 <= z1) MARCH_STAGE((q + 2) & 3, cur);   // (planes beyond z1 are not needed by this run)
      // ---- stores of plane z-1, then the loads of plane z+3 and of the row operands of plane z+1 ---------------------
      if (st_row >= 0) {
        if (MODE != RESID || v.y) v.y[st_row] = st_out;   // (the solve loop needs only ||r|| and x + d.*r: y may be null)
        if (MODE == RESID && v.y2) v.y2[st_row] = st_out2;
      }
      if (z + 3 <= z1) {
        pre = march_load_pair(v.x, MARCH_E0(c, z + 3), pact, T.n_cols);
      }
      if (z + 1 < z1) {
        const int row = (z + 1) * T.P + inplane;
        const int rr = rowlive ? row : C.n_rows - 1;
        ncls = C.cls[rr];
        if (MODE == AXPBY) {
          if (v.beta != 0.0) npb = v.beta * v.y[rr];
        } else {
          npb = v.b[rr];
          if ((MODE == SMOOTH || (MODE == RESID && v.y2)) && v.d) npd = v.d[rr];
        }
      }
      // ---- compute plane z from the ring: every lane walks its own row's class ------------------------------------
      const int row = z * T.P + inplane;
      const bool live = rowlive && cls != 0xFFFF;      // exception rows are computed by csr_rows_spmv
      const int sb0 = ((q - 1) & 3) * SLP + (int)(MARCH_G0(c, z - 1) & 1LL) + tid;
      const int sb1 = (q & 3) * SLP + (int)(MARCH_G0(c, z) & 1LL) + tid;
      const int sb2 = ((q + 1) & 3) * SLP + (int)(MARCH_G0(c, z + 1) & 1LL) + tid;
      const int cq = live ? cls : 0;
      const int s = dptr[cq], len = live ? dptr[cq + 1] - s : 0;
      if (class_d) pd = dd[cq];
      double acc = 0.0;
      // four entries per trip: four dictionary reads, then four x reads in flight; entries beyond the lane's class
      // re-read its last one and are not added (the order of the additions is the stored order of the row)
      for (int k = 0; k < T.maxlen; k += 4) {
        MarchEnt e[4];
        double xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = dent[s + min(k + u, len - 1 < 0 ? 0 : len - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int dz1 = e[u].code & 3;
          xv[u] = win[(dz1 == 0 ? sb0 : (dz1 == 1 ? sb1 : sb2)) + (e[u].code >> 2)];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double t = acc + e[u].val * xv[u];
          acc = (k + u < len) ? t : acc;
        }
      }
      st_row = -1;
      if (live) {
        const double own = win[sb1 + T.halo];          // the row's own x (SMOOTH: v.xs == v.x is a launch condition)
        const double outv = epilogue<MODE>(v, row, acc, pb, pd, own);
        st_row = row;
        st_out = outv;
        if (MODE == RESID && v.y2) st_out2 = own + pd * outv;   // x + d.*r
        sq += (MODE == AXPBY && v.dotx) ? own * outv : outv * outv;
      }
      __syncthreads();
    }
    if (st_row >= 0) {   // the last plane of the run
      if (MODE != RESID || v.y) v.y[st_row] = st_out;
      if (MODE == RESID && v.y2) v.y2[st_row] = st_out2;
    }
  }
  if (EXC && blockIdx.x == gridDim.x - 1) sq += rowclass_exception_rows<MODE>(C, v, tid);   // EXC: C.nexc_inline > 0
  if (v.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w2 = 0; w2 < RM_C / 64; ++w2) t += red[w2];
      v.sumsq[w] = t;
    }
  }
#undef MARCH_G0
#undef MARCH_E0
#undef MARCH_STAGE
}

// ------------------------------------------------------------------------------------------------
// Two stages per marching pass (temporal blocking): one damped-Jacobi sweep and the residual of its result,
//   t = x + d.*(b - A x)  (MGcycle.jl:129-131)   and   r = b - A t  (MGcycle.jl:58-60 / SolveFuncs.jl:26-27),
// in ONE walk along z - the last pre-smoothing sweep with the residual the restriction needs, and the last
// post-smoothing sweep with the solve loop's residual (||r||^2 partials and the next cycle's first update t + d.*r).
// Two launches of csr_rowclass_march_spmv move 2 x 26 B per row for that; this one moves 34 (x, b, class id in; t and r
// - or t and t + d.*r - out).  The price is the halo: stage 2 of a chunk needs t on the chunk's in-plane halo, so stage
// 1 is evaluated on RM_C + 2*halo rows per plane (x staged on RM_C + 4*halo), and on one extra plane at each end of a
// run.  Ring of 4 slabs of x and 4 slabs of t in LDS, one barrier per plane, stage 2 two planes behind stage 1:
//   iteration z:  x plane z+2 -> ring | stage 1 on plane z (x planes z-1..z+1 -> t plane z) | stage 2 on plane z-2
//                 (t planes z-3..z-1) | barrier
// Same dictionary walk, same products in the same order and the same epilogue expressions as the single-stage
// kernels: t, r and t + d.*r are bit-identical to theirs.  Operators without exception rows whose relaxPrec is
// constant per class (host: march2_ok); one workgroup per CU (125 KB of LDS), 128 VGPRs.
// ------------------------------------------------------------------------------------------------
struct March2Args {
  const double* x;   // the iterate before the sweep                    [n_rows]
  const double* b;   //                                                 [n_rows]
  double* t;         // out (optional): x + d.*(b - A x)                [n_rows]
  double* r;         // out (optional): b - A t                         [n_rows]
  double* xn;        // out (optional): t + d.*r                        [n_rows]
  double* sumsq;     // out (optional): per-workgroup sums of r.^2      [nblocks]
  const double* d;   // march3 on a box operator (exception rows), ZERO: the full relaxPrec vector (rows without a class)
  double* sink;      // march3: scratch [12 * nblocks * threads] that takes the stores of lanes / planes with nothing to store
};

constexpr int RM2_CLEN = 7;   // longest class whose records a lane keeps in registers

// class ids of the rows e0, e0 + 1 (e0 even: one aligned 32-bit word; the class array is padded) packed lo | hi << 16;
// rows outside the operator get class 0
__device__ __forceinline__ unsigned int march_load_clspair(const unsigned short* __restrict__ cls, long long e0, bool act, int n_rows) {
  unsigned int r = 0u;
  if (act) {
    if (e0 >= 0 && e0 + 1 < n_rows) {
      r = *reinterpret_cast<const unsigned int*>(cls + e0);
    } else {
      if (e0 >= 0 && e0 < n_rows) r = cls[e0];
      if (e0 + 1 >= 0 && e0 + 1 < n_rows) r |= (unsigned int)cls[e0 + 1] << 16;
    }
  }
  return r;
}

// ZERO: the sweep starts from x = 0, i.e. its input is x1 = d.*b (relax's first update, MGcycle.jl:134 with r = b): the
// staged vector is computed from b and the class ids while the slab passes through registers, and the dscale launch with
// its write and re-read of x1 disappears (a.x is not read).  The product is dscale_kernel's (one multiply).
template <bool ZERO>
__global__ __launch_bounds__(RM_C, 4) void csr_rowclass_march2_spmv(RowClassDev C, March2Args a, MarchDev T) {
  extern __shared__ double win[];
  __shared__ double red[RM_C / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = xcd_band(blockIdx.x, T.nblocks);
  const int H = T.halo;
  const int SLX = RM_C + 4 * H;
  const int SLXP = SLX + 2;                            // entry k of a slab = global entry g0 + k; the pair loads overhang by <= 1
  const int SLTP = RM_C + 2 * H;
  const int npair = (SLX + 2) >> 1;                    // 16-byte pairs per x slab (<= 2*RM_C: host check)
  double* xw = win;                                    // [4][SLXP]  x planes, entry 0 = in-plane index c0 - 2H
  double* tw = win + 4 * SLXP;                         // [4][SLTP]  t planes, entry 0 = in-plane index c0 - H
  MarchEnt* dent = reinterpret_cast<MarchEnt*>(tw + 4 * SLTP);            // [RM_DCAP]
  double* dd = reinterpret_cast<double*>(dent + RM_DCAP);                 // [RM_NCLS] class relaxPrec
  int* dptr = reinterpret_cast<int*>(dd + RM_NCLS);                       // [RM_NCLS + 1]
  for (int i = tid; i < T.nent; i += RM_C) {
    MarchEnt e;
    e.val = C.cls_val[i];
    e.code = T.lb[i];
    e.pad = 0;
    dent[i] = e;
  }
  for (int i = tid; i <= T.ncls; i += RM_C) dptr[i] = C.cls_ptr[i];
  for (int i = tid; i < T.ncls; i += RM_C) dd[i] = C.cls_d[i];
  const long long tot = (long long)T.chunks * T.nplanes;
  long long it = tot * w / T.nblocks;
  const long long it_end = tot * (w + 1) / T.nblocks;
  const bool pact0 = tid < npair, pact1 = tid + RM_C < npair;
  const bool hact = tid < 2 * H;                       // lanes that also own a halo row of stage 1
  const int hj = tid < H ? tid : (hact ? RM_C + tid : 0);   // that row's offset from c0 - H (core rows: tid + H)
  const bool regs = T.maxlen <= RM2_CLEN;              // class records in registers (else: per-lane walk of the LDS dictionary)
  const int SLX8 = SLXP * 8, SLT8 = SLTP * 8, TW8 = 4 * SLXP * 8;
  double sq = 0.0;

  // x slab of plane p of chunk c: global entries g0 = p*P + c*RM_C - 2H onwards; loaded as 16-byte pairs from the even
  // floor of g0, stored so that entry k of the slab is global entry g0 + k (a leading entry of an odd g0 is dropped)
#define M2_G0(c, p) ((long long)(p) * T.P + (long long)(c) * RM_C - 2 * H)
#define M2_E0(c, p, k) ((M2_G0(c, p) & ~1LL) + 2 * (tid + (k) * RM_C))
#define M2_STAGE(slot, par, v0, v1)                                                                                    \
  do {                                                                                                                 \
    if (pact0) {                                                                                                       \
      const int i_ = (slot) * SLXP + 2 * tid - (par);                                                                  \
      if (i_ >= (slot) * SLXP) xw[i_] = (v0).x;                                                                        \
      xw[i_ + 1] = (v0).y;                                                                                             \
    }                                                                                                                  \
    if (pact1) {                                                                                                       \
      const int i_ = (slot) * SLXP + 2 * (tid + RM_C) - (par);                                                         \
      xw[i_] = (v1).x;                                                                                                 \
      xw[i_ + 1] = (v1).y;                                                                                             \
    }                                                                                                                  \
  } while (0)
  // per-lane walk of the LDS dictionary (classes longer than RM2_CLEN): acc = sum over the entries of class cq (ascending k)
  // of value * ring[base(dz) + code >> 2], additions predicated on k < len
#define M2_WALK(acc, ring, bm, bz, bp, cq, livev)                                                                      \
  do {                                                                                                                 \
    const int s_ = dptr[cq], len_ = (livev) ? dptr[(cq) + 1] - s_ : 0;                                                 \
    for (int k_ = 0; k_ < T.maxlen; k_ += 4) {                                                                         \
      MarchEnt e_[4];                                                                                                  \
      double xv_[4];                                                                                                   \
      _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) e_[u_] = dent[s_ + min(k_ + u_, len_ - 1 < 0 ? 0 : len_ - 1)];  \
      _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                               \
        const int dz1_ = e_[u_].code & 3;                                                                              \
        xv_[u_] = (ring)[(dz1_ == 0 ? (bm) : (dz1_ == 1 ? (bz) : (bp))) + (e_[u_].code >> 2)];                         \
      }                                                                                                                \
      _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                               \
        const double t_ = (acc) + e_[u_].val * xv_[u_];                                                                \
        (acc) = (k_ + u_ < len_) ? t_ : (acc);                                                                         \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
  // Class records in registers.  A lane's row keeps its in-plane position from plane to plane and with it - away from
  // the first and the last plane - its class: the records are re-read from the LDS dictionary only when the class id
  // changes.  Per record: the value and one int = ((column offset + the lane's slab offset) * 8) | (dz + 1), i.e. the
  // BYTE offset of the gathered entry inside a slab with the plane shift in the two spare low bits; the slab of plane
  // z + dz is slot (q + dz) & 3 of the ring.  Records beyond the class's length carry the value 0 at the class's first
  // offset: they add +-0 (no predication).  A walk then costs per record 4 integer operations, one 8-byte LDS read and
  // one FMA (the per-lane walk of the LDS dictionary: ~12 operations and 24 bytes - the walks are VALU-issue bound).
#define M2_LOADRECS(vals, offs, cq, jx)                                                                                \
  do {                                                                                                                 \
    const int s_ = dptr[cq], len_ = dptr[(cq) + 1] - s_;                                                               \
    _Pragma("unroll") for (int u_ = 0; u_ < RM2_CLEN; ++u_) {                                                          \
      const MarchEnt e_ = dent[s_ + (u_ < len_ ? u_ : 0)];                                                             \
      (vals)[u_] = (u_ < len_) ? e_.val : 0.0;                                                                         \
      (offs)[u_] = ((((e_.code >> 2) + (jx)) << 3)) | (e_.code & 3);                                                   \
    }                                                                                                                  \
  } while (0)
  // qs: ring slot of plane z - 1 (so that the slot of plane z + dz is (qs + dz + 1) & 3); base8: byte offset of the ring; extra8: added
  // to every record's byte offset (the t ring is walked with the x ring's records: other lane offset, other slab length)
#define M2_REGWALK(acc, vals, offs, qs, sl8, base8)                                                                    \
  do {                                                                                                                 \
    double xv_[RM2_CLEN];                                                                                              \
    _Pragma("unroll") for (int u_ = 0; u_ < RM2_CLEN; ++u_) {                                                          \
      const int o_ = (offs)[u_];                                                                                       \
      const int slot_ = (o_ + (qs)) & 3;                                                                               \
      xv_[u_] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(win) + (slot_ * (sl8) + (o_ & ~7) + (base8))); \
    }                                                                                                                  \
    _Pragma("unroll") for (int u_ = 0; u_ < RM2_CLEN; ++u_) (acc) = (acc) + (vals)[u_] * xv_[u_];                      \
  } while (0)

  // the source of the slabs: x, or (ZERO) b with the class ids of the same entries
  const double* src = ZERO ? a.b : a.x;
#define M2_X1(v, cc)                                                                                                   \
  do {                                                                                                                 \
    if (ZERO) {                                                                                                        \
      (v).x = dd[(cc) & 0xFFFFu] * (v).x;                                                                              \
      (v).y = dd[(cc) >> 16] * (v).y;                                                                                  \
    }                                                                                                                  \
  } while (0)
  __syncthreads();   // dictionaries in place
  while (it < it_end) {
    const int c = (int)(it / T.nplanes);
    const int z0 = (int)(it - (long long)c * T.nplanes);
    const int z1 = (int)((it_end - it) < (long long)(T.nplanes - z0) ? z0 + (it_end - it) : T.nplanes);
    it += z1 - z0;
    const int c0 = c * RM_C;
    const int ipc = c0 + tid;                          // the lane's core row (in-plane index)
    const bool livec = ipc < T.P;
    const int iph = c0 - H + hj;                       // the lane's halo row of stage 1
    const bool liveh = hact && iph >= 0 && iph < T.P;
    // ---- fill the x ring: planes z0-2, z0-1, z0 (slots 0..2); plane z0+1 goes into registers -------------------------
#pragma unroll 1
    for (int pp = 0; pp < 3; ++pp) {
      d2_t q0 = march_load_pair(src, M2_E0(c, z0 - 2 + pp, 0), pact0, T.n_cols);
      d2_t q1 = march_load_pair(src, M2_E0(c, z0 - 2 + pp, 1), pact1, T.n_cols);
      if (ZERO) {
        const unsigned int k0 = march_load_clspair(C.cls, M2_E0(c, z0 - 2 + pp, 0), pact0, C.n_rows);
        const unsigned int k1 = march_load_clspair(C.cls, M2_E0(c, z0 - 2 + pp, 1), pact1, C.n_rows);
        M2_X1(q0, k0);
        M2_X1(q1, k1);
      }
      M2_STAGE(pp, (int)(M2_G0(c, z0 - 2 + pp) & 1LL), q0, q1);
    }
    d2_t pre0 = march_load_pair(src, M2_E0(c, z0 + 1, 0), pact0, T.n_cols);
    d2_t pre1 = march_load_pair(src, M2_E0(c, z0 + 1, 1), pact1, T.n_cols);
    unsigned int cpre0 = 0u, cpre1 = 0u;
    if (ZERO) {
      cpre0 = march_load_clspair(C.cls, M2_E0(c, z0 + 1, 0), pact0, C.n_rows);
      cpre1 = march_load_clspair(C.cls, M2_E0(c, z0 + 1, 1), pact1, C.n_rows);
    }
    // row operands of stage 1, plane zz: class id and b of the core row and of the halo row (a safe row when not live)
    int nclsc, nclsh;
    double nbc, nbh;
#define M2_OPERANDS(zz)                                                                                                \
  do {                                                                                                                 \
    const bool pv_ = (zz) >= 0 && (zz) < T.nplanes;                                                                    \
    const int rc_ = (pv_ && livec) ? (zz) * T.P + ipc : C.n_rows - 1;                                                  \
    const int rh_ = (pv_ && liveh) ? (zz) * T.P + iph : C.n_rows - 1;                                                  \
    nclsc = C.cls[rc_];                                                                                                \
    nbc = a.b[rc_];                                                                                                    \
    nclsh = C.cls[rh_];                                                                                                \
    nbh = a.b[rh_];                                                                                                    \
  } while (0)
    M2_OPERANDS(z0 - 1);
    __syncthreads();
    // pending stores (issued at the top of the next iteration) and the operand pipeline of the core row for stage 2
    double st_t = 0.0, st_r = 0.0, st_xn = 0.0;
    int st_trow = -1, st_rrow = -1;
    int cls1 = 0, cls2 = 0;        // class of the core row in planes z-1, z-2
    double b1 = 0.0, b2 = 0.0;
    // class records in registers: core row (x ring, lane offset tid + H; the t ring is walked with the same records
    // shifted by the constant (tid - (tid + H)) * 8) and halo row (x ring, lane offset hj)
    double cv[RM2_CLEN], hv[RM2_CLEN];
    int co[RM2_CLEN], ho[RM2_CLEN];
    int ccls = -1, hcls = -1;
    for (int z = z0 - 1; z <= z1 + 1; ++z) {
      const int q = z - z0 + 2;                        // ring index of plane z (plane z0-2 is 0)
      d2_t cur0 = pre0, cur1 = pre1;
      unsigned int ccur0 = cpre0, ccur1 = cpre1;
      int clsc = nclsc, clsh = nclsh;
      double bc = nbc, bh = nbh;
      asm volatile("" : "+v"(cur0.x), "+v"(cur0.y), "+v"(cur1.x), "+v"(cur1.y), "+v"(clsc), "+v"(bc), "+v"(clsh), "+v"(bh));
      if (ZERO) {
        asm volatile("" : "+v"(ccur0), "+v"(ccur1));
        M2_X1(cur0, ccur0);
        M2_X1(cur1, ccur1);
      }
      // ---- x plane z+2 into its slot (that of plane z-2, last read before the previous barrier) -----------------------
      if (z + 2 <= z1 + 1) M2_STAGE((q + 2) & 3, (int)(M2_G0(c, z + 2) & 1LL), cur0, cur1);
      // ---- stores of the previous iteration, then the loads of x plane z+3 and of the operands of plane z+1 ------------
      if (st_trow >= 0 && a.t) a.t[st_trow] = st_t;
      if (st_rrow >= 0) {
        if (a.r) a.r[st_rrow] = st_r;
        if (a.xn) a.xn[st_rrow] = st_xn;
      }
      if (z + 3 <= z1 + 1) {
        pre0 = march_load_pair(src, M2_E0(c, z + 3, 0), pact0, T.n_cols);
        pre1 = march_load_pair(src, M2_E0(c, z + 3, 1), pact1, T.n_cols);
        if (ZERO) {
          cpre0 = march_load_clspair(C.cls, M2_E0(c, z + 3, 0), pact0, C.n_rows);
          cpre1 = march_load_clspair(C.cls, M2_E0(c, z + 3, 1), pact1, C.n_rows);
        }
      }
      if (z + 1 <= z1) M2_OPERANDS(z + 1);
      // ---- stage 1 on plane z: t = x + d.*(b - A x) on the core row and on the halo row ---------------------------------
      st_trow = -1;
      if (z >= 0 && z < T.nplanes && z <= z1) {        // (uniform)
        const int xbm = ((q - 1) & 3) * SLXP, xbz = (q & 3) * SLXP, xbp = ((q + 1) & 3) * SLXP;
        {
          const int cq = livec ? clsc : 0;
          const int j = tid + H;
          double acc = 0.0;
          if (regs) {
            if (cq != ccls) {
              M2_LOADRECS(cv, co, cq, j);
              ccls = cq;
            }
            M2_REGWALK(acc, cv, co, q - 1, SLX8, 0);
          } else {
            int w0 = xbm + j, w1 = xbz + j, w2 = xbp + j;
            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));   // (opaque: else the compiler branches per entry)
            M2_WALK(acc, xw, w0, w1, w2, cq, livec);
          }
          if (livec) {
            const double own = xw[xbz + j + H];
            const double tv = own + dd[cq] * (bc - acc);
            tw[(q & 3) * SLTP + j] = tv;
            if (z >= z0 && z < z1) {
              st_trow = z * T.P + ipc;
              st_t = tv;
            }
          }
        }
        if (hact) {
          const int cq = liveh ? clsh : 0;
          double acc = 0.0;
          if (regs) {
            if (cq != hcls) {
              M2_LOADRECS(hv, ho, cq, hj);
              hcls = cq;
            }
            M2_REGWALK(acc, hv, ho, q - 1, SLX8, 0);
          } else {
            int w0 = xbm + hj, w1 = xbz + hj, w2 = xbp + hj;
            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
            M2_WALK(acc, xw, w0, w1, w2, cq, liveh);
          }
          if (liveh) {
            const double own = xw[xbz + hj + H];
            tw[(q & 3) * SLTP + hj] = own + dd[cq] * (bh - acc);
          }
        }
      }
      // ---- stage 2 on plane z-2: r = b - A t from the t ring (planes z-3, z-2, z-1: written before the last barrier) ----
      st_rrow = -1;
      if (z - 2 >= z0 && z - 2 < z1) {                 // (uniform)
        const int tbz = ((q - 2) & 3) * SLTP + tid;    // entry of column ip + rest: (tid + H) + rest = tid + (code >> 2)
        const int cq = livec ? cls2 : 0;
        double acc = 0.0;
        if (regs) {
          if (cq != ccls) {
            M2_LOADRECS(cv, co, cq, tid + H);
            ccls = cq;
          }
          M2_REGWALK(acc, cv, co, q - 3, SLT8, TW8 - 8 * H);   // (the records hold (off + tid + H) * 8: the t slab wants off + tid)
        } else {
          int w0 = ((q - 3) & 3) * SLTP + tid, w1 = tbz, w2 = ((q - 1) & 3) * SLTP + tid;
          asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
          M2_WALK(acc, tw, w0, w1, w2, cq, livec);
        }
        if (livec) {
          const double own = tw[tbz + H];
          const double rv = b2 - acc;
          st_rrow = (z - 2) * T.P + ipc;
          st_r = rv;
          st_xn = own + dd[cq] * rv;
          sq += rv * rv;
        }
      }
      cls2 = cls1;
      b2 = b1;
      cls1 = clsc;
      b1 = bc;
      __syncthreads();
    }
    if (st_trow >= 0 && a.t) a.t[st_trow] = st_t;      // (not reached with the loop's last iteration being stage 2 only)
    if (st_rrow >= 0) {
      if (a.r) a.r[st_rrow] = st_r;
      if (a.xn) a.xn[st_rrow] = st_xn;
    }
  }
  if (a.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w2 = 0; w2 < RM_C / 64; ++w2) t += red[w2];
      a.sumsq[w] = t;
    }
  }
#undef M2_G0
#undef M2_E0
#undef M2_STAGE
#undef M2_X1
#undef M2_WALK
#undef M2_LOADRECS
#undef M2_REGWALK
#undef M2_OPERANDS
}

// ------------------------------------------------------------------------------------------------
// Two stages per pass on 2-D IN-PLANE TILES (round 3): the same pair as csr_rowclass_march2_spmv,
//   t = x + d.*(b - A x)  (MGcycle.jl:129-131)   and   r = b - A t  (MGcycle.jl:58-60 / SolveFuncs.jl:26-27),
// for operators whose classes are "z-stars": at most one entry in plane z-1 and one in plane z+1, both at the row's own
// in-plane position, and up to RM3_NIP in-plane entries (dy, dx) with |dy|, |dx| <= 1 - the 7-point operator and its
// boundary classes.  What changes against the 1-D chunks of march2:
//  * a workgroup owns a TX x TY tile of the plane (x fastest) and walks a run of planes; stage 1 is evaluated on the tile
//    + one ring of rows, from x staged on the tile + two rings: (TX+4)(TY+4) / (TX TY) of x instead of (C + 4 n1) / C -
//    the halo no longer grows with the line length (a 513-node line is served like a 257-node one);
//  * the entries in planes z-1 / z+1 read the row's OWN position only, so they come from registers: a lane keeps x of its
//    rows for planes z-1, z (carried) and reads z+1 once; t of planes z-2, z-1, z likewise.  LDS then holds 3 slabs of x
//    (z: in-plane reads, z+1: own read, z+2: being written) and 2 of t instead of 4 + 4, which is what lets the tile grow,
//    and stage 2 runs ONE plane behind stage 1:
//      iteration z:  x plane z+2 -> ring | stage 1 on plane z | stage 2 on plane z-1 | barrier
//  * lane (xx, j) of the (TX+2)-wide stage-1 region owns rows (xx, j + s*SY), s < K1: its rows share the x position, hence
//    (away from the first/last line of the grid and the first/last plane) the class - ONE set of class records per lane, in
//    registers: {value of the z-1 entry, value of the z+1 entry, RM3_NIP x (value, byte offset inside a slab)}; classes
//    with fewer entries are padded with value 0 at their first in-plane offset (adds +-0);
//  * NO class-id stream: the host has verified that cls(x, y, z) = tab[cz[z]][cy[y]][cx[x]] for small index maps cx, cy,
//    cz (true for every grid operator: the class says which neighbours exist) - the maps live in LDS and the 2 bytes per
//    row (3.7 with the halo and the partial cache lines of a tile's short segments) are not read at all;
//  * schedule: either equal contiguous ranges of the (tile, plane) list (as march), or LOCKSTEP - every workgroup one
//    segment of one tile, all tiles of a segment side by side on one XCD, so that the halo lines two neighbouring tiles
//    both stage are fetched from HBM once and hit that XCD's L2 the second time.
// Same products in the same order (z-1 entry, in-plane entries in stored order, z+1 entry = ascending columns), same
// epilogue expressions as march2 / the single-stage kernels.  Loads are consumed one iteration after their issue, stores
// not waited for (every lane issues every store instruction, so the wait counts them).  OUT: bit 0 = r is written,
// bit 1 = t + d.*r, bit 2 = t.
// ------------------------------------------------------------------------------------------------
constexpr int RM3_NIP = 5;      // in-plane entries of a class
constexpr int RM3_NCLS = 128;   // classes of the operator
constexpr int RM3_TAB = 1024;   // entries of the class table tab[cz][cy][cx]
struct M3Class {                // 80 bytes per class; built on the host (build_march3), copied to LDS by every workgroup
  double v_lo, v_hi;            // value of the entry in plane z-1 / z+1 (0: the class has none)
  double v[RM3_NIP];            // in-plane values in stored order, 0 beyond the class's length
  int off[RM3_NIP];             // byte offset (dy*pitch + dx)*8 of each; padding repeats the first one
  int flags;                    // bit 0: stage 2 of such a row is not computed by this kernel
};
static_assert(sizeof(M3Class) == 80, "M3Class is read with 16-byte LDS loads");
struct March3Dev {
  const M3Class* cls;           // [ncls]
  const unsigned short* cmap;   // cx[n1] | cy[n2] | cz[nplanes] | tab[ncz*ncy*ncx]  (class = tab[(cz*ncy + cy)*ncx + cx])
  int ncx, ncy, ntab;
  int n1, n2, nplanes, P;       // grid (x fastest), P = n1*n2
  int TX, TY;                   // core tile
  int tiles_x, tiles_y;
  int WX, SY;                   // width of the stage-1 region (TX + 2); lines of it per slot pass (NT / WX)
  int pitch;                    // doubles per slab line (2*NPL, even)
  int LY, NPL;                  // lines of an x slab (TY + 4); 16-byte pairs per line
  int nblocks;
  int segs, seglen;             // lockstep schedule: segs > 0: workgroup w = segment (w / tiles) of tile (w % tiles)
  int n_cols, ncls;
  const double* vband;          // VAR (coefficients differ from row to row): 7 planar arrays of n_rows values, slot k at
  long long vstride;            // vband + k*vstride: k = 0 the z-1 entry, 1..5 the in-plane entries in stored order, 6 the z+1 entry
  long long vsrc[RM3_NIP + 2];  // where slot k of row r is READ: vband[vsrc[k] + r].  k*vstride; for a symmetric operator in canonical slots
                                // (build_band) the three lower entries come from the upper ones of the neighbour - slot 0 of row r = slot 6
                                // of row r - P, slot 1 = slot 5 of row r - n1, slot 2 = slot 4 of row r - 1 (zeros in front of every slot) -
                                // so that 4 of the 7 planes are streamed from HBM, the other three reads hit lines just fetched
  int has_exc;                  // box operator of a sharded level: the table holds 0xFFFF for rows that read the halo (neither
                                // stage is computed here) and cmap continues with sx[n1] | sy[n2] | sz[nplanes]: stage 2 of
                                // row (x, y, z) is left out where sx[x] | sy[y] | sz[z] (a neighbour is such a row)
};

constexpr int RM3_PD = 1;   // x planes in flight in registers (2 was measured slower: profiles/r03_march3_ab.md section 7)
// (slab pairs: ONE 16-byte load in every lane at a clamped address, no branch - march_load_pair_raw; the two-path form put
// `s_waitcnt vmcnt(1)` in front of its second path: every iteration waited for the stores of the previous one)
// VAR (round 3): the same pass for grid operators whose COEFFICIENTS differ from row to row (div sigma grad - what jInv feeds
// this package; MGsetup.jl:226-270 exists because sigma changes every outer iteration).  The classes then describe the
// STRUCTURE only (which neighbours a row has: offsets, no values), still as a verified product map; the values are streamed
// from 7 planar arrays (T.vband: coalesced 8-byte loads along x, 0 where a row has no such entry) and relaxPrec from the
// level's vector, both one plane ahead like b.  Stage 2 of plane z-1 reuses the values stage 1 of that plane loaded one
// iteration earlier (registers): the pair streams the matrix ONCE - 56 + 8 B/row instead of 2 x (56 + 4 + 8) through the
// pattern-coded CSR kernels.
template <bool ZERO, int OUT, int NT, int K1, int NPM, bool VAR = false>
__global__ __launch_bounds__(NT) void csr_rowclass_march3_spmv(RowClassDev C, March2Args a, March3Dev T) {
  extern __shared__ double win[];
  __shared__ double red[NT / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = xcd_band(blockIdx.x, T.nblocks);
  const int pitch = T.pitch;
  const int XS = T.LY * pitch;                  // doubles per x slab
  const int TS = (T.TY + 2) * pitch;            // doubles per t slab (same pitch: one set of record offsets serves both)
  double* xw = win;                             // [3][XS]
  double* tw = win + 3 * XS;                    // [2][TS]
  M3Class* dcl = reinterpret_cast<M3Class*>(tw + 2 * TS);    // [ncls]
  double* dd = reinterpret_cast<double*>(dcl + T.ncls);      // [ncls] class relaxPrec
  unsigned short* cxL = reinterpret_cast<unsigned short*>(dd + T.ncls);   // cx | cy | cz | tab
  unsigned short* cyL = cxL + T.n1;
  unsigned short* czL = cyL + T.n2;
  unsigned short* tabL = czL + T.nplanes;
  unsigned short* sxL = tabL + T.ntab;          // (has_exc only)
  unsigned short* syL = sxL + T.n1;
  unsigned short* szL = syL + T.n2;
  {
    const int nw = T.ncls * (int)(sizeof(M3Class) / 8);
    const double* srcd = reinterpret_cast<const double*>(T.cls);
    double* dstd = reinterpret_cast<double*>(dcl);
    for (int i = tid; i < nw; i += NT) dstd[i] = srcd[i];
    if (!VAR)
      for (int i = tid; i < T.ncls; i += NT) dd[i] = C.cls_d[i];
    const int nm = (T.n1 + T.n2 + T.nplanes) * (T.has_exc ? 2 : 1) + T.ntab;
    for (int i = tid; i < nm; i += NT) cxL[i] = T.cmap[i];
    // band form in canonical slots: a row without some neighbour still reads that neighbour's place (times the 0 in its slot) - the
    // t slabs hold computed rows only, so every entry starts finite
    if (VAR)
      for (int i = tid; i < 2 * TS; i += NT) tw[i] = 0.0;
  }
  const int zstride = T.ncy * T.ncx;            // class = tabL[cz[z]*zstride + (cy[y]*ncx + cx[x])]
  // ---- the lane's place: column xx of the stage-1 region, lines j + s*SY ---------------------------------------------
  const int xx = tid % T.WX, j = tid / T.WX;
  const bool lane_ok = j < T.SY;
  const int own8 = ((j + 1) * pitch + xx + 1) * 8;      // byte offset of slot 0's own entry inside an x slab
  const int tdelta8 = -(pitch + 1) * 8;                 // ... and of the same row inside a t slab, relative to it
  const int sstride8 = T.SY * pitch * 8;                // from slot s to slot s + 1
  // ---- the lane's 16-byte pairs of a slab: pair pid = tid + m*NT is pair i of line l ---------------------------------
  int pofs[NPM];        // element index of the pair inside a slab (line*pitch + 2*i)
  int pline[NPM];       // its line
  unsigned pflag = 0u;  // per m: bit 4m = the pair exists, bit 4m+1 = first pair of its line, bit 4m+2 = its line is inside the grid
#pragma unroll
  for (int m = 0; m < NPM; ++m) {
    const int pid = tid + m * NT;
    const int l = pid / T.NPL, i = pid - l * T.NPL;
    pline[m] = l;
    pofs[m] = l * pitch + 2 * i;
    if (pid < T.LY * T.NPL) pflag |= 1u << (4 * m);
    if (i == 0) pflag |= 2u << (4 * m);
  }
  const int ntiles = T.tiles_x * T.tiles_y;
  long long it, it_end;
  if (T.segs > 0) {     // lockstep: one segment of one tile
    const int seg = w / ntiles, c = w - seg * ntiles;
    const int zs = seg * T.seglen, ze = zs + T.seglen < T.nplanes ? zs + T.seglen : T.nplanes;
    it = (long long)c * T.nplanes + zs;
    it_end = (long long)c * T.nplanes + (ze > zs ? ze : zs);
  } else {
    const long long tot = (long long)ntiles * T.nplanes;
    it = tot * w / T.nblocks;
    it_end = tot * (w + 1) / T.nblocks;
  }
  const double* src = ZERO ? a.b : a.x;
  // this lane's slot of the sink: 32 workgroup-sized slabs shared by all workgroups (what lands there is never read) - small
  // enough to stay in L2: one slab per workgroup (12 doubles x 250 000 lanes = 19 MB) was written back to HBM once or twice per
  // launch (PMC: +30 MB of writes on the 272 MB of t and r)
  double* sk = a.sink + ((size_t)(w & 31) * NT + tid);
  double sq = 0.0;
  // class records of the lane (registers)
  double rlo = 0.0, rhi = 0.0, rv[RM3_NIP];
  int ro[RM3_NIP], rcls = -1;
  [[maybe_unused]] int rflag = 0;
#pragma unroll
  for (int u = 0; u < RM3_NIP; ++u) {
    rv[u] = 0.0;
    ro[u] = 0;
  }
#define M3_LOADRECS(cq)                                                                                                \
  do {                                                                                                                 \
    const M3Class* q_ = dcl + (cq);                                                                                    \
    rlo = q_->v_lo;                                                                                                    \
    rhi = q_->v_hi;                                                                                                    \
    _Pragma("unroll") for (int u_ = 0; u_ < RM3_NIP; ++u_) {                                                           \
      rv[u_] = q_->v[u_];                                                                                              \
      ro[u_] = q_->off[u_];                                                                                            \
    }                                                                                                                  \
    rflag = q_->flags;                                                                                                 \
    rcls = (cq);                                                                                                       \
  } while (0)
  // acc = (z-1 entry) + in-plane entries in stored order + (z+1 entry); base8: byte address of the row's own entry in the slab
#define M3_WALK(acc, lo_, hi_, slab, base8)                                                                            \
  do {                                                                                                                 \
    double xv_[RM3_NIP];                                                                                               \
    _Pragma("unroll") for (int u_ = 0; u_ < RM3_NIP; ++u_)                                                             \
      xv_[u_] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(slab) + ((base8) + ro[u_]));            \
    (acc) = (acc) + rlo * (lo_);                                                                                       \
    _Pragma("unroll") for (int u_ = 0; u_ < RM3_NIP; ++u_) (acc) = (acc) + rv[u_] * xv_[u_];                           \
    (acc) = (acc) + rhi * (hi_);                                                                                       \
  } while (0)
  // VAR: the same walk with the row's own values vv_[0..6] (z-1, in-plane x RM3_NIP, z+1); the record gives the offsets
#define M3_WALKV(acc, lo_, hi_, slab, base8, vv_)                                                                      \
  do {                                                                                                                 \
    double xv_[RM3_NIP];                                                                                               \
    _Pragma("unroll") for (int u_ = 0; u_ < RM3_NIP; ++u_)                                                             \
      xv_[u_] = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(slab) + ((base8) + ro[u_]));            \
    (acc) = (acc) + (vv_)[0] * (lo_);                                                                                  \
    _Pragma("unroll") for (int u_ = 0; u_ < RM3_NIP; ++u_) (acc) = (acc) + (vv_)[1 + u_] * xv_[u_];                    \
    (acc) = (acc) + (vv_)[RM3_NIP + 1] * (hi_);                                                                        \
  } while (0)
  constexpr int NV = VAR ? RM3_NIP + 2 : 1;   // values per row kept in registers
  constexpr int KV = VAR ? K1 : 1;
  __syncthreads();   // dictionaries in place
  while (it < it_end) {
    const int c = (int)(it / T.nplanes);
    const int z0 = (int)(it - (long long)c * T.nplanes);
    const int z1 = (int)((it_end - it) < (long long)(T.nplanes - z0) ? z0 + (it_end - it) : T.nplanes);
    it += z1 - z0;
    const int ty = c / T.tiles_x, tx = c - ty * T.tiles_x;
    const int x0 = tx * T.TX, y0 = ty * T.TY;
    // ---- per-tile lane state -------------------------------------------------------------------------------------------
    int pg[NPM];          // in-plane index of the pair's first element (before the even floor; may be negative)
    int pr0[NPM], pr1[NPM];   // ZERO: cy*ncx + cx of the pair's two elements as they land in the slab (clamped into the grid)
    pflag &= ~0x4444u;
#pragma unroll
    for (int m = 0; m < NPM; ++m) {
      const int yl = y0 - 2 + pline[m];
      const int i2 = pofs[m] - pline[m] * pitch;        // 2*i
      if (yl >= 0 && yl < T.n2) pflag |= 4u << (4 * m);
      pg[m] = yl * T.n1 + x0 - 2 + i2;
      pr0[m] = pr1[m] = 0;
    }
    const int gx = x0 - 1 + xx;
    const bool xin = lane_ok && gx >= 0 && gx < T.n1;
    const bool xcore = xx >= 1 && xx <= T.TX;
    const int ip0 = (y0 - 1 + j) * T.n1 + gx;           // in-plane index of slot 0's row; slot s: + s*SY*n1
    const int ipstride = T.SY * T.n1;
    unsigned live1 = 0u, core = 0u;                     // per slot: stage 1 is computed / the row belongs to the core tile
    unsigned skip2 = 0u;                                // per slot: stage 2 is left out (box operators: next to a row that reads the halo)
    int rp[K1];                                         // cy*ncx + cx of the slot's row (class = tab[cz*zstride + rp])
    const int cxo = xin ? (int)cxL[gx] : 0;
#pragma unroll
    for (int s = 0; s < K1; ++s) {
      const int yy = j + s * T.SY, gy = y0 - 1 + yy;
      const bool l1 = xin && yy < T.TY + 2 && gy >= 0 && gy < T.n2;
      live1 |= (l1 ? 1u : 0u) << s;
      core |= ((l1 && xcore && yy >= 1 && yy <= T.TY) ? 1u : 0u) << s;
      rp[s] = l1 ? (int)cyL[gy] * T.ncx + cxo : 0;
      if (T.has_exc && l1 && (sxL[gx] | syL[gy])) skip2 |= 1u << s;
    }
#define M3_PAR(p, m) ((int)(((long long)(p) * T.P + pg[m]) & 1LL))
#define M3_LOADPAIR(dst, p, m)                                                                                         \
  do {                                                                                                                 \
    const bool act_ = ((pflag >> (4 * (m))) & 5u) == 5u && (p) >= 0 && (p) < T.nplanes;                                \
    const long long e0_ = ((long long)(p) * T.P + pg[m]) & ~1LL;                                                       \
    (dst) = march_load_pair_raw(src, e0_, act_, T.n_cols);                                                             \
  } while (0)
  // (raw loads: the one pair whose second entry does not exist was read one entry earlier - moved into place where consumed)
#define M3_FIXPAIR(v, p, m)                                                                                            \
  do {                                                                                                                 \
    if ((p) == T.nplanes - 1) {                                                                           /* (uniform) */   \
      const bool act_ = ((pflag >> (4 * (m))) & 5u) == 5u;                                                             \
      const long long e0_ = ((long long)(p) * T.P + pg[m]) & ~1LL;                                                     \
      march_pair_fix((v), e0_, act_, T.n_cols);                                                                        \
    }                                                                                                                  \
  } while (0)
  // entry k of a slab line = in-plane index (line start) + k: a leading entry of an odd line start is dropped
#define M3_STAGE(slot, p, m, v)                                                                                        \
  do {                                                                                                                 \
    if ((pflag >> (4 * (m))) & 1u) {                                                                                   \
      const int par_ = M3_PAR(p, m);                                                                                   \
      double* q_ = xw + ((slot) * XS + pofs[m] - par_);                                                                \
      if (!(par_ && ((pflag >> (4 * (m))) & 2u))) q_[0] = (v).x;                                                       \
      q_[1] = (v).y;                                                                                                   \
    }                                                                                                                  \
  } while (0)
    if (ZERO) {   // x1 = d.*b at the slab positions: the classes of the two elements a pair stores (positions, not loads)
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        const int yl = y0 - 2 + pline[m];
        const int i2 = pofs[m] - pline[m] * pitch;
        const int yc = yl < 0 ? 0 : (yl >= T.n2 ? T.n2 - 1 : yl);
        const int base = (int)cyL[yc] * T.ncx;
        pr0[m] = base;    // completed per plane: the parity shifts which two columns the pair holds
        pr1[m] = x0 - 2 + i2;
      }
    }
    // relaxPrec of a row without a class (box operators: a row that reads the halo), from the level's vector
#define M3_DROW(e) (((e) >= 0 && (e) < (long long)T.nplanes * T.P) ? a.d[(e)] : 0.0)
    // class of the slab element at column col (clamped into the line) of pair m's line
#define M3_PAIRCLS(m, col, zb) ((int)tabL[(zb) + pr0[m] + (int)cxL[(col) < 0 ? 0 : ((col) >= T.n1 ? T.n1 - 1 : (col))]])
#define M3_X1(v, p, m)                                                                                                 \
  do {                                                                                                                 \
    if (ZERO) {                                                                                                        \
      const int zc_ = (p) < 0 ? 0 : ((p) >= T.nplanes ? T.nplanes - 1 : (p));                                          \
      const int zb_ = (int)czL[zc_] * zstride;                                                                         \
      const int c0_ = pr1[m] - M3_PAR(p, m);                                                                           \
      const int k0_ = M3_PAIRCLS(m, c0_, zb_), k1_ = M3_PAIRCLS(m, c0_ + 1, zb_);                                      \
      const long long e_ = (((long long)(p) * T.P + pg[m]) & ~1LL);                                                    \
      (v).x = ((VAR || k0_ == 0xFFFF) ? M3_DROW(e_) : dd[k0_]) * (v).x;                                                \
      (v).y = ((VAR || k1_ == 0xFFFF) ? M3_DROW(e_ + 1) : dd[k1_]) * (v).y;                                            \
    }                                                                                                                  \
  } while (0)
    // ---- fill the ring: planes z0-1 (slot 0) and z0 (slot 1); plane z0+1 goes into registers ----------------------------
#pragma unroll 1
    for (int pp = 0; pp < 2; ++pp) {
      d2_t q[NPM];
#pragma unroll
      for (int m = 0; m < NPM; ++m) M3_LOADPAIR(q[m], z0 - 1 + pp, m);
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        M3_FIXPAIR(q[m], z0 - 1 + pp, m);
        M3_X1(q[m], z0 - 1 + pp, m);
        M3_STAGE(pp, z0 - 1 + pp, m, q[m]);
      }
    }
    // Loads in flight.  RM3_PD planes ahead: with 2, iteration z consumes what iteration z-2 asked for and refills the SAME
    // register buffer, so nothing is copied between buffers (a copy of a register whose load is in flight is a wait): the
    // loop below is unrolled by two and each half uses its own buffer (PB = half).
    constexpr int NBUF = RM3_PD == 2 ? 2 : 1;
    d2_t preb[NBUF][NPM];         // x planes z+2 [, z+3]
#pragma unroll
    for (int q = 0; q < NBUF; ++q)
#pragma unroll
      for (int m = 0; m < NPM; ++m) M3_LOADPAIR(preb[q][m], z0 + 1 + q, m);
    // b of plane zz for every slot's row (a safe row where the slot is not live)
    double nbb[NBUF][K1];
    double nvb[NBUF][KV][NV], ndb[NBUF][KV];   // VAR: the rows' values and relaxPrec, in flight with b
#define M3_OPERANDS(zz, PB_)                                                                                           \
  do {                                                                                                                 \
    const bool pv_ = (zz) >= 0 && (zz) < T.nplanes;                                                                    \
    _Pragma("unroll") for (int s_ = 0; s_ < K1; ++s_) {                                                                \
      const int r_ = (pv_ && ((live1 >> s_) & 1u)) ? (zz) * T.P + ip0 + s_ * ipstride : C.n_rows - 1;                  \
      nbb[PB_][s_] = a.b[r_];                                                                                          \
      if (VAR) {                                                                                                       \
        _Pragma("unroll") for (int k_ = 0; k_ < NV; ++k_)                                                              \
          nvb[PB_][s_ % KV][k_] = T.vband[T.vsrc[k_] + (long long)r_];                                                 \
        ndb[PB_][s_ % KV] = a.d[r_];                                                                                   \
      }                                                                                                                \
    }                                                                                                                  \
  } while (0)
    M3_OPERANDS(z0 - 1, 0);
    if (NBUF == 2) M3_OPERANDS(z0, NBUF - 1);
    // own x of plane z0-2 (the z-1 entry of stage 1 on plane z0-1), straight from global memory
    double xm[K1], xc[K1];
#pragma unroll
    for (int s = 0; s < K1; ++s) {
      xm[s] = 0.0;
      if (z0 - 2 >= 0 && ((live1 >> s) & 1u)) {
        const int r_ = (z0 - 2) * T.P + ip0 + s * ipstride;
        if (ZERO) {
          const int k_ = (int)tabL[(int)czL[z0 - 2] * zstride + rp[s]];
          xm[s] = ((VAR || k_ == 0xFFFF) ? a.d[r_] : dd[k_]) * a.b[r_];
        } else {
          xm[s] = a.x[r_];
        }
      }
    }
    // As many stores as an iteration of the loop below issues, BEHIND the loads above: the compiler's wait for those loads
    // at the top of the loop is then s_waitcnt vmcnt(number of stores) on the entry path as well as on the back edge
    {
#pragma unroll
      for (int i = 0; i < K1 * (((OUT >> 2) & 1) + ((OUT >> 1) & 1) + (OUT & 1)); ++i) sk[(size_t)i * 32 * NT] = 0.0;   // (distinct slots: distinct instructions)
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < K1; ++s)   // own x of plane z0-1 from its slab (slot 0)
      xc[s] = ((live1 >> s) & 1u) ? *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xw) + (own8 + s * sstride8)) : 0.0;
    double t1[K1], t2[K1];         // own t of planes z-1, z-2
    double b1[K1];                 // b of plane z-1 (stage 2)
    double v1[KV][NV], d1[KV];     // VAR: values and relaxPrec of the rows of plane z-1 (stage 2)
#pragma unroll
    for (int s = 0; s < K1; ++s) t1[s] = t2[s] = b1[s] = 0.0;
#pragma unroll
    for (int s = 0; s < KV; ++s) {
      d1[s] = 0.0;
#pragma unroll
      for (int k = 0; k < NV; ++k) v1[s][k] = 0.0;
    }
    int qz = 0;                    // ring slot of plane z (plane z0-1 is slot 0)
    for (int zz = z0 - 1; zz <= z1; zz += NBUF) {
#pragma unroll
     for (int half = 0; half < NBUF; ++half) {
      const int z = zz + half;
      if (z > z1) break;                                       // (uniform)
      const int PB = half;                                     // (a constant after the unroll: the buffers stay in registers)
      d2_t cur[NPM];
      double b0[K1];
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        cur[m] = preb[PB][m];
        asm volatile("" : "+v"(cur[m].x), "+v"(cur[m].y));     // the wait of this iteration: the loads, not the stores behind them
      }
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        b0[s] = nbb[PB][s];
        asm volatile("" : "+v"(b0[s]));
      }
      double v0[KV][NV], d0[KV];
      if (VAR) {
#pragma unroll
        for (int s = 0; s < KV; ++s) {
          d0[s] = ndb[PB][s];
          asm volatile("" : "+v"(d0[s]));
#pragma unroll
          for (int k = 0; k < NV; ++k) {
            v0[s][k] = nvb[PB][s][k];
            asm volatile("" : "+v"(v0[s][k]));
          }
        }
      }
      const int q1 = qz == 2 ? 0 : qz + 1, q2 = q1 == 2 ? 0 : q1 + 1;   // slots of planes z+1, z+2
      // ---- x plane z+2 into its slot (that of plane z-1, last read before the previous barrier) ------------------------
      if (z + 2 <= z1 + 1) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) {
          M3_FIXPAIR(cur[m], z + 2, m);
          M3_X1(cur[m], z + 2, m);
          M3_STAGE(q2, z + 2, m, cur[m]);
        }
      }
      // ---- the loads of x plane z+2+NBUF and of b of plane z+NBUF into the buffer just consumed, in flight while this plane
      // (and with two buffers the next one) is computed
      if (z + 2 + NBUF <= z1 + 1) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) M3_LOADPAIR(preb[PB][m], z + 2 + NBUF, m);
      }
      if (z + NBUF <= z1) M3_OPERANDS(z + NBUF, PB);
      // ---- stage 1 on plane z: t = x + d.*(b - A x) on every live row of the lane ----------------------------------------
      const bool s1 = z >= 0 && z < T.nplanes;          // (uniform)
      const int zb0 = s1 ? (int)czL[z] * zstride : 0;   // class table rows of planes z and z-1
      const int zb1 = (z - 1 >= 0 && z - 1 < T.nplanes) ? (int)czL[z - 1] * zstride : 0;
      double tc[K1];
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        tc[s] = 0.0;
        if ((live1 >> s) & 1u) {
          const int o8 = own8 + s * sstride8;
          const double xp = *reinterpret_cast<const double*>(reinterpret_cast<const char*>(xw) + (q1 * XS * 8 + o8));
          const int cq = s1 ? (int)tabL[zb0 + rp[s]] : 0xFFFF;
          if (cq != 0xFFFF) {                           // (0xFFFF: a row that reads the halo - computed after the exchange)
            if (cq != rcls) M3_LOADRECS(cq);
            double acc = 0.0;
            if (VAR) {
              M3_WALKV(acc, xm[s], xp, xw, qz * XS * 8 + o8, v0[s % KV]);
            } else {
              M3_WALK(acc, xm[s], xp, xw, qz * XS * 8 + o8);
            }
            const double tv = xc[s] + (VAR ? d0[s % KV] : dd[cq]) * (b0[s] - acc);
            *reinterpret_cast<double*>(reinterpret_cast<char*>(tw) + ((z & 1) * TS * 8 + o8 + tdelta8)) = tv;
            tc[s] = tv;
          }
          xm[s] = xc[s];
          xc[s] = xp;
        }
      }
      // ---- stage 2 on plane z-1: r = b - A t (in-plane neighbours from the t slab written before the last barrier) -----------
      double st_r[K1], st_x[K1];
      unsigned done2 = 0u;
      const bool s2 = z - 1 >= z0 && z - 1 < z1;        // (uniform)
      const unsigned skz = (T.has_exc && s2 && szL[z - 1]) ? ~0u : skip2;
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        st_r[s] = st_x[s] = 0.0;
        if (s2 && ((core >> s) & 1u) && !((skz >> s) & 1u)) {
          const int cq = (int)tabL[zb1 + rp[s]];
          if (cq != 0xFFFF) {
            if (cq != rcls) M3_LOADRECS(cq);
            const int o8 = own8 + s * sstride8 + tdelta8;
            double acc = 0.0;
            if (VAR) {
              M3_WALKV(acc, t2[s], tc[s], tw, ((z - 1) & 1) * TS * 8 + o8, v1[s % KV]);
            } else {
              M3_WALK(acc, t2[s], tc[s], tw, ((z - 1) & 1) * TS * 8 + o8);
            }
            const double rr = b1[s] - acc;
            st_r[s] = rr;
            st_x[s] = t1[s] + (VAR ? d1[s % KV] : dd[cq]) * rr;
            sq += rr * rr;
            done2 |= 1u << s;
          }
        }
      }
      // ---- the stores of this iteration: EVERY lane issues every store instruction (lanes / planes with nothing to store
      // write to their slot of the sink), so their number is fixed and the wait at the top of the next iteration can be
      // for the loads alone (s_waitcnt vmcnt(number of stores)): a store's acknowledgement is not on the critical path ------
      {
        const bool wt = s1 && z >= z0 && z < z1;        // (uniform) plane z belongs to this run: its t is stored
#pragma unroll
        for (int s = 0; s < K1; ++s) {
          const int rowt = z * T.P + ip0 + s * ipstride, rowr = rowt - T.P;
          if (OUT & 4) {
            double* q_ = (wt && ((core >> s) & 1u)) ? a.t + rowt : sk;
            *q_ = tc[s];
          }
          if (OUT & 1) {
            double* q_ = ((done2 >> s) & 1u) ? a.r + rowr : sk;
            *q_ = st_r[s];
          }
          if (OUT & 2) {
            double* q_ = ((done2 >> s) & 1u) ? a.xn + rowr : sk;
            *q_ = st_x[s];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        t2[s] = t1[s];
        t1[s] = tc[s];
        b1[s] = b0[s];
      }
      if (VAR) {
#pragma unroll
        for (int s = 0; s < KV; ++s) {
          d1[s] = d0[s];
#pragma unroll
          for (int k = 0; k < NV; ++k) v1[s][k] = v0[s][k];
        }
      }
      qz = q1;
      __syncthreads();
     }
    }
  }
  if (a.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w2 = 0; w2 < NT / 64; ++w2) t += red[w2];
      a.sumsq[w] = t;
    }
  }
#undef M3_LOADRECS
#undef M3_WALK
#undef M3_WALKV
#undef M3_X1
#undef M3_PAR
#undef M3_PAIRCLS
#undef M3_DROW
#undef M3_LOADPAIR
#undef M3_FIXPAIR
#undef M3_STAGE
#undef M3_OPERANDS
}

// ------------------------------------------------------------------------------------------------
// FOUR stages in one walk along z (round 4): the solve loop's two fine-level passes are back to back across the stopping
// test - the last post-smoothing sweep + residual (+ ||r||^2, + the next cycle's first update) of step k writes the vector
// the second pre-smoothing sweep + residual of step k+1 reads (SolveFuncs.jl:24-37 around MGcycle.jl:26-31,54-60,122-136):
//     t  = x  + d.*(b - A x)        last post-smoothing sweep of step k        (the iterate of step k)
//     r  = b - A t ;  ||r||^2       the stopping test's residual               (SolveFuncs.jl:26-30)
//     xn = t  + d.*r                first pre-smoothing update of step k+1     (MGcycle.jl:134 with the r just computed)
//     t' = xn + d.*(b - A xn)       second pre-smoothing sweep                 (MGcycle.jl:128-131)
//     r' = b - A t'                 the residual the restriction needs         (MGcycle.jl:58-60)
// x and b go in ONCE, t' and r' come out once: 32 B/row where the two passes moved 24 + 32 and ran the tile twice.  The
// two stages across the stopping test are speculative: if the test ends the loop, the caller re-creates the iterate t with
// one single-stage sweep of x (still in its buffer) and drops t', r'.
// Layout: the two-stage tile kernel's (z-star classes from a verified product map, z+-1 entries from the lane's own
// registers, one barrier per plane) with three more rings and the in-plane star in CANONICAL form:
//   * the tile core TX x TY carries a halo of 3 rows for stage 1 (x staged with 4), 2 for stage 2, 1 for stage 3; 3 slabs
//     of x and 2 each of t, xn, t', all with the same pitch, origin and line count; stage k runs k-1 planes behind stage 1:
//         iteration z:  x plane z+2 -> ring | t(z) | r, xn (z-1) | t' (z-2) | r' (z-3) | barrier
//   * the in-plane entries of every class are a subset of {-y, -x, 0, +x, +y} (checked on the host): a class record is
//     {value of the z-1 entry, of the z+1 entry, 5 in-plane values in that - the stored, ascending-column - order, 0 where
//     the class has none, relaxPrec} with NO offsets; a missing entry multiplies a finite number by 0: the LDS is cleared
//     once and only finite values are ever written - rows (outside the grid, beyond a stage's rings) and planes a stage does
//     not serve are computed like any other from staged x / finite slab entries and a valid b: bounded garbage that no
//     served row reads with a non-zero weight (a served row of stage k reads rows within one ring of itself = rows stage
//     k-1 serves; only ||r||^2 and the global stores select);
//   * a lane owns K1 VERTICALLY ADJACENT rows of one column: a row's own entry and its neighbours inside the lane's strip
//     come from registers, so a row-stage reads 2 (left, right) + 2/K1 (above the strip, below it) values from LDS
//     instead of 5, with immediate offsets;
//   * a lane holds ONE class record for its strip, re-read per stage only where the plane's z-class changes: the K1
//     accumulation chains of a stage carry no branch and interleave.  The strips of a tile row are shifted (ysh) so that
//     none holds rows of different classes - the first line of the grid ends a strip, the last one starts one (host check;
//     an operator whose lines differ in class beyond that keeps the two passes).
// Same products in the same order and the same epilogue expressions as march3 (a product with a 0 value adds +-0): t', r'
// are bit-identical to the two passes' (a zero may change its sign).  Stores: t'(z-2) and r'(z-3) by every lane (sink for
// lanes / planes with nothing to store).
// ------------------------------------------------------------------------------------------------
struct M4Class {                // 64 bytes per class; built on the host (build_march4), copied to LDS by every workgroup
  double v_lo, v_hi;            // value of the entry in plane z-1 / z+1 (0: the class has none)
  double v[5];                  // in-plane values: (dy,dx) = (-1,0), (0,-1), (0,0), (0,+1), (+1,0); 0 where the class has none
  double d;                     // relaxPrec of the class
};
static_assert(sizeof(M4Class) == 64, "M4Class is read with 16-byte LDS loads");
struct March4Dev {
  const M4Class* cls;           // [ncls]
  const unsigned short* cmap;   // cx[n1] | cy[n2] | cz[nplanes] | tab  (shared with march3)
  int ncx, ncy, ncz, ntab;
  int n1, n2, nplanes, P;
  int TX, TY;                   // core tile
  int tiles_x, tiles_y;
  int WX, SY;                   // width of the stage-1 region (TX + 6); lanes per column = strips of K1 lines (K1*SY >= TY + 6 + K1 - 1)
  int pitch;                    // doubles per slab line (2*NPL)
  int LY, NPL;                  // staged lines of an x slab (TY + 8 + K1 - 1); 16-byte pairs per line
  int LYA;                      // allocated lines of every slab (K1*SY + 2 >= LY)
  int nblocks;
  int segs, seglen;             // workgroup w = segment (w / tiles) of tile (w % tiles)
  int n_cols, ncls;
  const int* ysh;               // [tiles_y] lines the strips of a tile row are shifted down by (0..K1-1) so that no strip holds
                                // rows of different classes (the first / last line of the grid ends / starts a strip)
  // rows whose r^2 counts towards ||r||^2: [oxl, oxh) x [oyl, oyh) x [ozl, ozh) - the whole grid, or the OWNED box of a rank
  // whose grid is its box extended by ghost layers (mg_ghost_*: the ghost rows are some other rank's to count)
  int oxl, oxh, oyl, oyh, ozl, ozh;
};
constexpr int RM4_G = 4;        // halo of the staged x (stage 1 runs on core + 3 rings)

template <int NT, int K1, int NPM, int PITCH>
__global__ __launch_bounds__(NT) void csr_rowclass_march4_spmv(RowClassDev C, March2Args a, March4Dev T) {
  extern __shared__ double win[];
  __shared__ double red[NT / 64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int w = xcd_band(blockIdx.x, T.nblocks);
  constexpr int pitch = PITCH;   // (compile-time: the line offsets of every LDS access fold into the instruction's immediate offset)
  constexpr int P8 = pitch * 8;
  const int XS = T.LYA * pitch;                                   // doubles per slab (every ring)
  const int XS8 = XS * 8;
  // rings (byte offsets from win): x 3 slabs, t 2, xn 2, t' 2; slab line ly of every ring = in-plane line y0 - 4 + ly
  const int tB = 3 * XS8, nB = 5 * XS8, pB = 7 * XS8;
  double* xw = win;
  char* winb = reinterpret_cast<char*>(win);
  M4Class* dcl = reinterpret_cast<M4Class*>(win + 9 * XS);                                 // [ncls]
  // (cx | cy are read once per lane, from global memory; the LDS copy holds cz | tab: 2*(n1 + n2) bytes more for the slabs)
  const unsigned short* cxG = T.cmap;
  const unsigned short* cyG = cxG + T.n1;
  unsigned short* czL = reinterpret_cast<unsigned short*>(dcl + T.ncls);                   // cz | tab
  unsigned short* tabL = czL + T.nplanes;
  {
    const int nw = T.ncls * (int)(sizeof(M4Class) / 8);
    const double* srcd = reinterpret_cast<const double*>(T.cls);
    double* dstd = reinterpret_cast<double*>(dcl);
    for (int i = tid; i < nw; i += NT) dstd[i] = srcd[i];
    const int nm = T.nplanes + T.ntab;
    const unsigned short* czG = cyG + T.n2;
    for (int i = tid; i < nm; i += NT) czL[i] = czG[i];
    for (int i = tid; i < 9 * XS; i += NT) win[i] = 0.0;          // every slab entry finite from the start
  }
  const int zstride = T.ncy * T.ncx;
  // ---- the lane's place: column xx of the stage-1 region, lines K1*j .. K1*j + K1-1 ------------------------------------
  // Lanes beyond the WX*SY the region needs are TWINS of the lanes of the last strip line: same reads, same values, nothing
  // written to the slabs (a twin's write could overtake the original's read of the entry it replaces), nothing stored to
  // global memory, nothing added to ||r||^2 - so that no branch guards the arithmetic.
  const int xx = tid % T.WX, jt = tid / T.WX;
  const bool twin = jt >= T.SY;
  const int j = twin ? T.SY - 1 : jt;
  const int own8 = ((K1 * j + 1) * pitch + xx + 1) * 8;   // byte offset of row 0's own entry inside a slab (every ring); row s: + s*P8
#if defined(MG_M4_DPP)
  const bool edge_l = xx == 0 || lane == 0, edge_r = xx == T.WX - 1 || lane == 63;   // lanes whose neighbour lane is not their neighbour column
#endif
  // ---- the lane's 16-byte pairs of an x slab ---------------------------------------------------------------------------
  int pofs[NPM], pline[NPM];
  unsigned pflag = 0u;  // per m: bit 4m = the pair exists, bit 4m+1 = first pair of its line, bit 4m+2 = its line is inside the grid, bit 4m+3 = pg[m] is odd
#pragma unroll
  for (int m = 0; m < NPM; ++m) {
    const int pid = tid + m * NT;
    const int l = pid / T.NPL, i = pid - l * T.NPL;
    pline[m] = l;
    pofs[m] = l * pitch + 2 * i;
    if (pid < T.LY * T.NPL) pflag |= 1u << (4 * m);
    if (i == 0) pflag |= 2u << (4 * m);
  }
  const int ntiles = T.tiles_x * T.tiles_y;
  const int seg = w / ntiles, c = w - seg * ntiles;
  const int zs = seg * T.seglen, ze = zs + T.seglen < T.nplanes ? zs + T.seglen : T.nplanes;
  double* sk = a.sink + ((size_t)(w & 31) * NT + tid);
  double sq = 0.0;
  // class record of the lane (registers) - of the first of the four stages that needed another one
  double rlo = 0.0, rhi = 0.0, rd = 0.0, rv0 = 0.0, rv1 = 0.0, rv2 = 0.0, rv3 = 0.0, rv4 = 0.0;
  int rcls = -1;
#define M4_LOADRECS(cq)                                                                                                \
  do {                                                                                                                 \
    const M4Class* q_ = dcl + (cq);                                                                                    \
    rlo = q_->v_lo; rhi = q_->v_hi;                                                                                    \
    rv0 = q_->v[0]; rv1 = q_->v[1]; rv2 = q_->v[2]; rv3 = q_->v[3]; rv4 = q_->v[4];                                    \
    rd = q_->d;                                                                                                        \
    rcls = (cq);                                                                                                       \
  } while (0)
#define M4_LDS(off8) (*reinterpret_cast<const double*>(winb + (off8)))
#define M4_LDSW(off8) (*reinterpret_cast<double*>(winb + (off8)))
  __syncthreads();   // dictionaries in place, slabs cleared
  if (ze > zs) {
    const int ty = c / T.tiles_x, tx = c - ty * T.tiles_x;
    const int ysh = T.ysh[ty];
    const int x0 = tx * T.TX, y0 = ty * T.TY - ysh;     // slab line ly = in-plane line y0 - 4 + ly; the core starts at line y0 + ysh
    int pg[NPM];
#pragma unroll
    for (int m = 0; m < NPM; ++m) {
      const int yl = y0 - RM4_G + pline[m];
      const int i2 = pofs[m] - pline[m] * pitch;        // 2*i
      if (yl >= 0 && yl < T.n2) pflag |= 4u << (4 * m);
      pg[m] = yl * T.n1 + x0 - RM4_G + i2;
      if (pg[m] & 1) pflag |= 8u << (4 * m);
    }
    const int gx = x0 - 3 + xx;
    const bool xin = gx >= 0 && gx < T.n1;
    const int dxo = xx < 3 ? 3 - xx : (xx > T.TX + 2 ? xx - (T.TX + 2) : 0);    // rings between the column and the core
    const int ip0 = (y0 - 3 + K1 * j) * T.n1 + gx;      // in-plane index of row 0; row s: + s*n1
    unsigned lv = 0u;                                   // per row s: bit s = stage 1 is computed on it, bit 8+s stage 2, 16+s stage 3, 24+s stage 4 (core),
                                                        // bit 4+s = a core row whose r^2 counts (inside the owned box)
    const bool xown = gx >= T.oxl && gx < T.oxh;
    unsigned pk[K1];                                    // class of the row per z-class: byte zc = tab[zc][cy][cx]
    const int cxo = xin ? (int)cxG[gx] : 0;
    unsigned pk0 = 0u;                                  // class word of the lane's first live row
    bool have0 = false;
#pragma unroll
    for (int s = 0; s < K1; ++s) {
      const int yy = K1 * j + s, gy = y0 - 3 + yy;
      const int yc = yy - ysh;                          // line of the stage-1 region (0 .. TY + 5)
      const bool l1 = xin && yc >= 0 && yc < T.TY + 6 && gy >= 0 && gy < T.n2;
      const int dyo = yc < 3 ? 3 - yc : (yc > T.TY + 2 ? yc - (T.TY + 2) : 0);
      const int dist = dxo > dyo ? dxo : dyo;
      lv |= (l1 ? 1u : 0u) << s;
      lv |= ((l1 && dist <= 2) ? 1u : 0u) << (8 + s);
      lv |= ((l1 && dist <= 1) ? 1u : 0u) << (16 + s);
      lv |= ((l1 && dist == 0 && !twin) ? 1u : 0u) << (24 + s);
      lv |= ((l1 && dist == 0 && !twin && xown && gy >= T.oyl && gy < T.oyh) ? 1u : 0u) << (4 + s);
      unsigned v = 0u;
      if (l1) {
        const int rp = (int)cyG[gy] * T.ncx + cxo;
        for (int zc = 0; zc < T.ncz; ++zc) v |= ((unsigned)tabL[zc * zstride + rp] & 255u) << (8 * zc);
        if (!have0) { pk0 = v; have0 = true; }
      }
      pk[s] = v;
    }
    // Pair m of plane p: global entries e0, e0 + 1 with e0 = (p*P + pg[m]) rounded down to even.  The load is ONE 16-byte
    // access in every lane through a scalar base (x + pc*P - 4, pc = p clamped into the grid) and a 32-bit lane offset, no
    // branch, no fix-up behind it: pairs of lines / planes outside the grid and offsets that would leave [0, n_cols - 2] are
    // clamped to some valid pair of x (finite entries nobody reads with a non-zero weight).
#define M4_PAR(p, m) ((int)((((unsigned)(p) & (unsigned)T.P) ^ (pflag >> (4 * (m) + 3))) & 1u))
#define M4_LOADPAIR(dst, p, m)                                                                                         \
  do {                                                                                                                 \
    const int pc_ = (p) < 0 ? 0 : ((p) >= T.nplanes ? T.nplanes - 1 : (p));                          /* (uniform) */   \
    const double* base_ = a.x + ((long long)pc_ * T.P - 4);                                          /* (uniform) */   \
    const int lim_ = (int)((long long)T.n_cols - 2 - (long long)pc_ * T.P) + 4;                      /* (uniform) */   \
    int off_ = pg[m] - M4_PAR(pc_, m) + 4;                                                                             \
    off_ = (((pflag >> (4 * (m))) & 5u) == 5u) ? off_ : 4;                                                             \
    const int lo_ = pc_ == 0 ? 4 : 0;                                     /* (uniform: nothing in front of x) */       \
    off_ = off_ < lo_ ? lo_ : (off_ > lim_ ? lim_ : off_);                                                             \
    (dst) = *reinterpret_cast<const d2_t*>(base_ + (unsigned)off_);                                               \
  } while (0)
    // (n_cols odd: the last entry of x is the first of a pair whose second does not exist - the clamp above read the pair
    // one entry earlier; move it into place.  Only in the last plane.)
#define M4_FIXPAIR(v, p, m)                                                                                            \
  do {                                                                                                                 \
    if ((p) == T.nplanes - 1 && (T.n_cols & 1)) {                                                    /* (uniform) */   \
      const int lim_ = T.n_cols - 2 - (p) * T.P + 4;                                                                   \
      if ((((pflag >> (4 * (m))) & 5u) == 5u) && pg[m] - M4_PAR(p, m) + 4 == lim_ + 1) (v).x = (v).y;                  \
    }                                                                                                                  \
  } while (0)
#define M4_STAGE(slot, p, m, v)                                                                                        \
  do {                                                                                                                 \
    if ((pflag >> (4 * (m))) & 1u) {                                                                                   \
      const int par_ = M4_PAR(p, m);                                                                                   \
      double* q_ = xw + ((slot) * XS + pofs[m] - par_);                                                                \
      if (!(par_ && ((pflag >> (4 * (m))) & 2u))) q_[0] = (v).x;                                                       \
      q_[1] = (v).y;                                                                                                   \
    }                                                                                                                  \
  } while (0)
    const int zA = zs - 3 > 0 ? zs - 3 : 0;             // first plane of stage 1
    const int zE = ze + 2;                              // last iteration (stage 4 on plane ze - 1)
    // ---- fill the ring: planes zA (slot 0) and zA+1 (slot 1); plane zA+2 goes into registers ----------------------------
#pragma unroll 1
    for (int pp = 0; pp < 2; ++pp) {
      d2_t q[NPM];
#pragma unroll
      for (int m = 0; m < NPM; ++m) M4_LOADPAIR(q[m], zA + pp, m);
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        M4_FIXPAIR(q[m], zA + pp, m);
        M4_STAGE(pp, zA + pp, m, q[m]);
      }
    }
    d2_t preb[NPM];
#pragma unroll
    for (int m = 0; m < NPM; ++m) M4_LOADPAIR(preb[m], zA + 2, m);
    double nbb[K1];
#define M4_OPERANDS(zz)                                                                                                \
  do {                                                                                                                 \
    const int pc_ = (zz) < 0 ? 0 : ((zz) >= T.nplanes ? T.nplanes - 1 : (zz));                       /* (uniform) */   \
    const double* base_ = a.b + (long long)pc_ * T.P;                                                /* (uniform) */   \
    _Pragma("unroll") for (int s_ = 0; s_ < K1; ++s_) {                                                                \
      const int r_ = ((lv >> s_) & 1u) ? ip0 + s_ * T.n1 : 0;                                                          \
      nbb[s_] = base_[(unsigned)r_];                                                            \
    }                                                                                                                  \
  } while (0)
    M4_OPERANDS(zA);
    double xm[K1], xc[K1];
#pragma unroll
    for (int s = 0; s < K1; ++s) {
      xm[s] = 0.0;
      if (zA - 1 >= 0 && ((lv >> s) & 1u)) xm[s] = a.x[(zA - 1) * T.P + ip0 + s * T.n1];
    }
#pragma unroll
    for (int i = 0; i < 2 * K1; ++i) sk[(size_t)i * 32 * NT] = 0.0;   // as many stores as an iteration issues, behind the loads (see march3)
    __syncthreads();
#pragma unroll
    for (int s = 0; s < K1; ++s)   // own x of plane zA from its slab (slot 0) - every row of the strip: a row beyond the region is its neighbour's neighbour
      xc[s] = M4_LDS(own8 + s * P8);
    double t1[K1], n1[K1], p1[K1], b1[K1], b2[K1], b3[K1];
#pragma unroll
    for (int s = 0; s < K1; ++s) t1[s] = n1[s] = p1[s] = b1[s] = b2[s] = b3[s] = 0.0;
    int qz = 0;                    // ring slot of x plane z
    int zc1 = 0, zc2 = 0, zc3 = 0; // z-classes of planes z-1, z-2, z-3
    // acc = (z-1 entry) + in-plane entries in canonical (= stored) order + (z+1 entry)
#define M4_ACC(acc, lo_, up_, le_, ow_, ri_, dn_, hi_)                                                                 \
  do {                                                                                                                 \
    (acc) = 0.0;                                                                                                       \
    (acc) = (acc) + rlo * (lo_);                                                                                       \
    (acc) = (acc) + rv0 * (up_);                                                                                       \
    (acc) = (acc) + rv1 * (le_);                                                                                       \
    (acc) = (acc) + rv2 * (ow_);                                                                                       \
    (acc) = (acc) + rv3 * (ri_);                                                                                       \
    (acc) = (acc) + rv4 * (dn_);                                                                                       \
    (acc) = (acc) + rhi * (hi_);                                                                                       \
  } while (0)
    // the record of this pass's class in the z-class of the stage's plane (re-read only when it differs from the one held:
    // at the first / last planes, and in wavefronts that run more than one pass)
#define M4_REC(zc_)                                                                                                    \
  do {                                                                                                                 \
    const int cq_ = (int)((pwp >> (8 * (zc_))) & 255u);                                                                \
    if (cq_ != rcls) M4_LOADRECS(cq_);                                                                                 \
  } while (0)
    // one row-stage: left / right from the slab, above / below from the strip's registers or (first / last row) the slab
#if defined(MG_M4_DPP)
    // Variant (make variant NAME=dpp DEFS=-DMG_M4_DPP; profiles/r06_march4_variants.md): the left / right neighbours of a row are the SAME
    // row's values in the neighbour lanes (lane = column): taken by a wavefront shift (DPP wave_shr / wave_shl, no LDS bank touched);
    // only the lanes at the ends of a line or of the wavefront read theirs from the slab (the same values: every lane's row values
    // are what it wrote there).  Halves the slab reads of a row-stage.
#define M4_DPP_F64(dst, src, ctrl)                                                                                     \
  do {                                                                                                                 \
    const unsigned long long u_ = (unsigned long long)__double_as_longlong(src);                                       \
    const int lo32_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u_ & 0xffffffffull), (ctrl), 0xf, 0xf, false);    \
    const int hi32_ = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u_ >> 32), (ctrl), 0xf, 0xf, false);              \
    (dst) = __longlong_as_double((long long)(((unsigned long long)(unsigned)hi32_ << 32) | (unsigned long long)(unsigned)lo32_)); \
  } while (0)
#define M4_ROWSTAGE(rB, val, lo_, hi_, acc)                                                                            \
  do {                                                                                                                 \
    const int o8_ = (rB) + own8 + s * P8;                                                                              \
    double le_, ri_;                                                                                                   \
    M4_DPP_F64(le_, (val)[s], 0x138);   /* wave_shr:1 - lane i takes lane i-1's */                                     \
    M4_DPP_F64(ri_, (val)[s], 0x130);   /* wave_shl:1 - lane i takes lane i+1's */                                     \
    if (edge_l) le_ = M4_LDS(o8_ - 8);                                                                                 \
    if (edge_r) ri_ = M4_LDS(o8_ + 8);                                                                                 \
    const double up_ = s == 0 ? M4_LDS(o8_ - P8) : (val)[s > 0 ? s - 1 : 0];                                           \
    const double dn_ = s == K1 - 1 ? M4_LDS(o8_ + P8) : (val)[s < K1 - 1 ? s + 1 : s];                                 \
    M4_ACC(acc, (lo_)[s], up_, le_, (val)[s], ri_, dn_, (hi_)[s]);                                                     \
  } while (0)
#else
#define M4_ROWSTAGE(rB, val, lo_, hi_, acc)                                                                            \
  do {                                                                                                                 \
    const int o8_ = (rB) + own8 + s * P8;                                                                              \
    const double le_ = M4_LDS(o8_ - 8), ri_ = M4_LDS(o8_ + 8);                                                         \
    const double up_ = s == 0 ? M4_LDS(o8_ - P8) : (val)[s > 0 ? s - 1 : 0];                                           \
    const double dn_ = s == K1 - 1 ? M4_LDS(o8_ + P8) : (val)[s < K1 - 1 ? s + 1 : s];                                 \
    M4_ACC(acc, (lo_)[s], up_, le_, (val)[s], ri_, dn_, (hi_)[s]);                                                     \
  } while (0)
#endif
// (the stages of an iteration are not interleaved by the scheduler: live ranges stay within 128 / 168 registers)
#if defined(MG_M4_NO_SCHED)      /* (make variant NAME=nosched DEFS=-DMG_M4_NO_SCHED: profiles/r06_march4_variants.md) */
#define M4_SCHED
#else
#define M4_SCHED __builtin_amdgcn_sched_barrier(0)   /* (free scheduling of the 768- / 512-thread variants changes nothing: profiles/r05_march4_notes.md) */
#endif
#if defined(MG_M4_UNROLL)     /* (make variant NAME=unrollN DEFS=-DMG_M4_UNROLL=N: the plane loop unrolled by the compiler - the register rotations at its
                                 end become renamings; profiles/r06_march4_variants.md) */
#define M4_STR2(x) #x
#define M4_STR(x) M4_STR2(x)
    _Pragma(M4_STR(unroll MG_M4_UNROLL))
#endif
    for (int z = zA; z <= zE; ++z) {
      d2_t cur[NPM];
      double b0[K1];
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        cur[m] = preb[m];
        asm volatile("" : "+v"(cur[m].x), "+v"(cur[m].y));     // the wait of this iteration: the loads, not the stores behind them
      }
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        b0[s] = nbb[s];
        asm volatile("" : "+v"(b0[s]));
      }
      const int q1 = qz == 2 ? 0 : qz + 1, q2 = q1 == 2 ? 0 : q1 + 1;   // slots of planes z+1, z+2
      if (z + 2 <= zE + 1) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) {
          M4_FIXPAIR(cur[m], z + 2, m);
          M4_STAGE(q2, z + 2, m, cur[m]);
        }
      }
      if (z + 3 <= zE + 1) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) M4_LOADPAIR(preb[m], z + 3, m);
      }
      if (z + 1 <= zE) M4_OPERANDS(z + 1);
      const bool s1 = z < T.nplanes;                                            // (uniform; z >= zA >= 0)
      const bool s2 = z - 1 >= 0 && z - 1 < T.nplanes && z >= zs - 1;           // plane z-1 in [zs-2, ze+1]
      const bool s3 = z - 2 >= 0 && z - 2 < T.nplanes && z >= zs + 1;   // plane z-2 in [zs-1, ze]
      const bool s4 = z >= zs + 3;                                      // plane z-3 in [zs, ze)
      const int zc0 = s1 ? __builtin_amdgcn_readfirstlane((int)czL[z]) : zc1;   // (uniform)
      const bool cnt2 = z - 1 >= zs && z - 1 < ze && z - 1 >= T.ozl && z - 1 < T.ozh;   // (uniform) plane z-1 belongs to this segment (and to the owned box): its ||r||^2 counts
      const int xq8 = qz * XS8, xq18 = q1 * XS8;
      const int tW8 = tB + (z & 1) * XS8, tR8 = tB + ((z - 1) & 1) * XS8;
      const int nW8 = nB + ((z - 1) & 1) * XS8, nR8 = nB + (z & 1) * XS8;       // xn(z-1) written, xn(z-2) read
      const int pW8 = pB + (z & 1) * XS8, pR8 = pB + ((z - 1) & 1) * XS8;       // t'(z-2) written, t'(z-3) read
      double tc[K1], nc[K1], pc[K1], r4[K1];
      {
        double xp[K1], t2[K1], n2[K1], p2[K1];
#pragma unroll
        for (int s = 0; s < K1; ++s) xp[s] = M4_LDS(xq18 + own8 + s * P8);      // own x of plane z+1
        // the z-1 entries of stages 2-4: the row's own entry of the slab this iteration overwrites (t(z-2), xn(z-3), t'(z-4))
#pragma unroll
        for (int s = 0; s < K1; ++s) {
          t2[s] = M4_LDS(tW8 + own8 + s * P8);
          n2[s] = M4_LDS(nW8 + own8 + s * P8);
          p2[s] = M4_LDS(pW8 + own8 + s * P8);
        }
        {
          const unsigned pwp = pk0;                             // (the strip's rows share their class: host check)
          // ---- stage 1 on plane z: t = x + d.*(b - A x) --------------------------------------------------------------------
          M4_REC(zc0);
#pragma unroll
          for (int s = 0; s < K1; ++s) {
            double acc;
            M4_ROWSTAGE(xq8, xc, xm, xp, acc);
            const double tv = xc[s] + rd * (b0[s] - acc);
            tc[s] = tv;   // (rows / planes the stage does not serve hold bounded garbage: read with weight 0 or never - header)
          }
#pragma unroll
          for (int s = 0; s < K1; ++s)
            if (!twin) M4_LDSW(tW8 + own8 + s * P8) = tc[s];
          M4_SCHED;
          // ---- stage 2 on plane z-1: r = b - A t, xn = t + d.*r, ||r||^2 on the core ---------------------------------------
          M4_REC(zc1);
#pragma unroll
          for (int s = 0; s < K1; ++s) {
            double acc;
            M4_ROWSTAGE(tR8, t1, t2, tc, acc);
            const double rr = b1[s] - acc;
            const double xv = t1[s] + rd * rr;
            nc[s] = xv;
            sq += (cnt2 && s2 && ((lv >> (4 + s)) & 1u)) ? rr * rr : 0.0;
          }
#pragma unroll
          for (int s = 0; s < K1; ++s)
            if (!twin) M4_LDSW(nW8 + own8 + s * P8) = nc[s];
          M4_SCHED;
          // ---- stage 3 on plane z-2: t' = xn + d.*(b - A xn) ------------------------------------------------------------------
          M4_REC(zc2);
#pragma unroll
          for (int s = 0; s < K1; ++s) {
            double acc;
            M4_ROWSTAGE(nR8, n1, n2, nc, acc);
            const double tv = n1[s] + rd * (b2[s] - acc);
            pc[s] = tv;
          }
#pragma unroll
          for (int s = 0; s < K1; ++s)
            if (!twin) M4_LDSW(pW8 + own8 + s * P8) = pc[s];
          M4_SCHED;
          // ---- stage 4 on plane z-3: r' = b - A t' --------------------------------------------------------------------------
          M4_REC(zc3);
#pragma unroll
          for (int s = 0; s < K1; ++s) {
            double acc;
            M4_ROWSTAGE(pR8, p1, p2, pc, acc);
            r4[s] = b3[s] - acc;
          }
        }
#pragma unroll
        for (int s = 0; s < K1; ++s) {
          xm[s] = xc[s];
          xc[s] = xp[s];
        }
      }
      // ---- the stores of this iteration (every lane issues every store instruction) ---------------------------------------
      {
        const bool w3 = s3 && z - 2 >= zs && z - 2 < ze;   // (uniform) t' of plane z-2 belongs to this segment
        double* bt_ = a.t + (long long)(z - 2) * T.P;      // (uniform bases; dereferenced for planes of the segment only)
        double* br_ = a.r + (long long)(z - 3) * T.P;
#pragma unroll
        for (int s = 0; s < K1; ++s) {
          const unsigned row_ = (unsigned)(ip0 + s * T.n1);     // (a core row's in-plane index; anything for the others: sink)
          double* qt_ = (w3 && ((lv >> (24 + s)) & 1u)) ? bt_ + row_ : sk;
          *qt_ = pc[s];
          double* qr_ = (s4 && ((lv >> (24 + s)) & 1u)) ? br_ + row_ : sk;
          *qr_ = r4[s];
        }
      }
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        t1[s] = tc[s];
        n1[s] = nc[s];
        p1[s] = pc[s];
        b3[s] = b2[s];
        b2[s] = b1[s];
        b1[s] = b0[s];
      }
      zc3 = zc2;
      zc2 = zc1;
      zc1 = zc0;
      qz = q1;
      __syncthreads();
    }
  }
  if (a.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) red[wave] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w2 = 0; w2 < NT / 64; ++w2) t += red[w2];
      a.sumsq[w] = t;
    }
  }
#undef M4_LOADRECS
#undef M4_LDS
#undef M4_LDSW
#undef M4_PAR
#undef M4_LOADPAIR
#undef M4_FIXPAIR
#undef M4_STAGE
#undef M4_OPERANDS
#undef M4_ACC
#undef M4_REC
#undef M4_ROWSTAGE
#if defined(MG_M4_DPP)
#undef M4_DPP_F64
#endif
#undef M4_SCHED
}

// Band form: the values of a CSR operator re-laid as planar slots (build_band).  slot[c*NS + e] = the planar array entry e
// of a row of structure class c goes to; slots a class does not fill keep the 0 they were initialised with.
__global__ __launch_bounds__(BLK) void band_fill(const int* __restrict__ rowptr, const double* __restrict__ val,
                                                 const unsigned short* __restrict__ cls, const int* __restrict__ slot, int ns,
                                                 double* __restrict__ vband, long long vstride, int n) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const int c = cls[i], k0 = rowptr[i], len = rowptr[i + 1] - k0;
  for (int e = 0; e < len && e < ns; ++e) vband[(size_t)slot[c * ns + e] * (size_t)vstride + (size_t)i] = val[k0 + e];
}

// Is a band in canonical slots (0: -z, 1: -y, 2: -x, 3: diagonal, 4: +x, 5: +y, 6: +z) symmetric entry by entry - A[r, r-s] == A[r-s, r]
// for s = P, n1, 1?  bad += rows where it is not (the entries in front of a slot are zeros: a row without the neighbour must hold 0)
__global__ __launch_bounds__(BLK) void band_sym_check(const double* __restrict__ vband, long long vstride, int n, int n1, int P, int* bad) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n) return;
  const double* s0 = vband;
  const bool ok = s0[i] == s0[6 * vstride + i - P] && s0[vstride + i] == s0[5 * vstride + i - n1] && s0[2 * vstride + i] == s0[4 * vstride + i - 1];
  if (!ok) atomicAdd(bad, 1);
}

// ------------------------------------------------------------------------------------------------
// Block of right-hand sides [n][k] (row-major: a row's k entries side by side) <-> k columns of stride ns (column c at
// out + c*ns): the column-wise solve of a block (solve_dev_columns) runs the single-vector kernels on each column.
// 256 rows per workgroup through LDS: both sides coalesced.
// ------------------------------------------------------------------------------------------------
template <bool TO_COLUMNS>
__global__ __launch_bounds__(256) void block_columns_transpose(const double* __restrict__ in, double* __restrict__ out, long long n, int k, long long ns) {
  extern __shared__ double tile[];          // [k][257]
  const long long r0 = (long long)blockIdx.x * 256;
  const int rows = (int)((n - r0) < 256 ? (n - r0) : 256);
  const int tid = threadIdx.x;
  if (TO_COLUMNS) {
    for (int idx = tid; idx < rows * k; idx += 256) {
      const int row = idx / k, col = idx - row * k;
      tile[col * 257 + row] = in[r0 * k + idx];
    }
    __syncthreads();
    for (int idx = tid; idx < 256 * k; idx += 256) {
      const int col = idx >> 8, row = idx & 255;
      if (row < rows) out[(long long)col * ns + r0 + row] = tile[col * 257 + row];
    }
  } else {
    for (int idx = tid; idx < 256 * k; idx += 256) {
      const int col = idx >> 8, row = idx & 255;
      if (row < rows) tile[col * 257 + row] = in[(long long)col * ns + r0 + row];
    }
    __syncthreads();
    for (int idx = tid; idx < rows * k; idx += 256) {
      const int row = idx / k, col = idx - row * k;
      out[r0 * k + idx] = tile[col * 257 + row];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Exception rows of a row-class operator (rows whose class was too rare for the dictionary: a few per cent next to
// sub-domain faces or irregular boundaries): one lane per listed row, straight from the CSR arrays, same epilogues.
// ------------------------------------------------------------------------------------------------
// (round 3) EIGHT lanes per listed row: they load eight consecutive entries of the row at once (one or two cache lines of the
// CSR arrays per row instead of one scattered 8/4-byte access per lane and entry - the face layers of a sharded box are
// hundreds of thousands of rows, and with one lane per row the uncoalesced walk took ~1 us per 1000 rows), then every lane of
// the group adds the products in stored order through shuffles: the same fused multiply-adds in the same order as before.
template <int MODE>
__global__ __launch_bounds__(BLK) void csr_rows_spmv(CsrDev A, const int* __restrict__ rows, int nrows, VecArgs v,
                                                     int sumsq_off) {
  __shared__ double red[BLK / 64];
  const int tid = threadIdx.x, g = tid & 7, grp = tid >> 3;
  double sq = 0.0;
#pragma unroll 1
  for (int pass = 0; pass < 8; ++pass) {                 // BLK rows per workgroup (one ||.||^2 partial each), 32 at a time
    const int i = blockIdx.x * BLK + pass * (BLK / 8) + grp;
    if (i < nrows) {
      const int row = rows[i];
      const int s = A.rowptr[row], e = A.rowptr[row + 1];
      double acc = 0.0;
      for (int base = s; base < e; base += 8) {
        const int k = base + g;
        double val = 0.0, xv = 0.0;
        if (k < e) {
          val = A.val[k];
          xv = v.x[A.colidx[k]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double vu = __shfl(val, u, 8), xu = __shfl(xv, u, 8);
          if (base + u < e) acc += vu * xu;
        }
      }
      if (g == 0) {
        double pb = 0.0, pd = 0.0, px = 0.0;
        if (MODE == AXPBY) {
          if (v.beta != 0.0) pb = v.beta * v.y[row];
        } else {
          pb = v.b[row];
          if (MODE == SMOOTH || (MODE == RESID && v.y2)) {
            pd = v.d[row];
            px = v.xs[row];
          }
        }
        const double outv = epilogue<MODE>(v, row, acc, pb, pd, px);
        if (MODE != RESID || v.y) v.y[row] = outv;
        if (MODE == RESID && v.y2) v.y2[row] = px + pd * outv;   // x + d.*r: the next cycle's first damped-Jacobi update
        sq += outv * outv;
      }
    }
  }
  if (v.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < BLK / 64; ++w) t += red[w];
      v.sumsq[sumsq_off + blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// CSR-stream SpMM, nrhs > 1, vectors row-major [n][nrhs].
// The nnz segment (values AND column indices) is staged in LDS with coalesced loads; then G lanes
// (G = pow2 >= nrhs, <= 64) own one row x one RHS column each and walk the row from LDS (broadcast
// reads); every x gather is one contiguous nrhs*8-byte segment.
// ------------------------------------------------------------------------------------------------
template <int MODE, bool NT>
__global__ __launch_bounds__(BLK) void csr_stream_spmm(CsrDev A, VecArgs v, int G) {
  __shared__ double sval[MM_CHUNK];
  __shared__ int scol[MM_CHUNK];
  __shared__ int srow[MM_MAXROWS + 1];

  const int tid = threadIdx.x;
  int bid = xcd_band(blockIdx.x, A.nblocks);
  if (A.sched) bid = A.sched[bid];
  const int r0 = A.blk_row[bid];
  const int r1 = A.blk_row[bid + 1];
  const int nrows = r1 - r0;
  const int k0 = A.rowptr[r0];
  const int k1 = A.rowptr[r1];
  const int nrhs = v.nrhs;
  const int grp = tid / G;
  const int c = tid - grp * G;
  const int ngrp = BLK / G;

  const bool longrow = (nrows == 1 && (k1 - k0) > MM_CHUNK - 2);
  if (!longrow) {
    const int base = k0 & ~1;
#pragma unroll
    for (int it = 0; it < MM_PAIRS; ++it) {
      const int idx = base + it * (2 * BLK) + 2 * tid;
      if (idx < k1) {
        *reinterpret_cast<d2_t*>(&sval[it * (2 * BLK) + 2 * tid]) = load_stream<NT>(A.val + idx);
        *reinterpret_cast<i2_t*>(&scol[it * (2 * BLK) + 2 * tid]) = load_stream<NT>(A.colidx + idx);
      }
    }
    if (tid <= nrows) srow[tid] = A.rowptr[r0 + tid] - base;
    if (tid == 0 && nrows == MM_MAXROWS) srow[MM_MAXROWS] = k1 - base;
    __syncthreads();
  }
  for (int c0 = 0; c0 < nrhs; c0 += G) {
    const int col = c0 + c;
    const bool cact = col < nrhs;
    for (int lr = grp; lr < nrows; lr += ngrp) {
      const int row = r0 + lr;
      double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
      if (cact) {
        if (!longrow) {
          const int s = srow[lr], e = srow[lr + 1];
          int k = s;
          for (; k + 4 <= e; k += 4) {
            const double x0 = v.x[(size_t)scol[k] * nrhs + col];
            const double x1 = v.x[(size_t)scol[k + 1] * nrhs + col];
            const double x2 = v.x[(size_t)scol[k + 2] * nrhs + col];
            const double x3 = v.x[(size_t)scol[k + 3] * nrhs + col];
            acc0 += sval[k] * x0;
            acc1 += sval[k + 1] * x1;
            acc2 += sval[k + 2] * x2;
            acc3 += sval[k + 3] * x3;
          }
          for (; k < e; ++k) acc0 += sval[k] * v.x[(size_t)scol[k] * nrhs + col];
        } else {
          for (int k = k0; k < k1; ++k) acc0 += A.val[k] * v.x[(size_t)A.colidx[k] * nrhs + col];
        }
        const double acc = (acc0 + acc1) + (acc2 + acc3);
        const size_t o = (size_t)row * nrhs + col;
        double pb = 0.0, pd = 0.0, px = 0.0;
        if (MODE == AXPBY) { if (v.beta != 0.0) pb = v.beta * v.y[o]; }
        else pb = v.b[o];
        if (MODE == SMOOTH) { pd = v.d[row]; px = v.xs[o]; }
        v.y[o] = epilogue<MODE>(v, row, acc, pb, pd, px);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Row-class SpMM (nrhs > 1, vectors row-major [n][nrhs]) with a per-lane class walk: G = pow2 >= nrhs lanes own one
// row x one right-hand-side column each (a wavefront holds 64/G rows, every lane two of them), the class dictionary
// lives in LDS as {value, column offset} records, and a gather of x is one contiguous 8*nrhs-byte segment per row.
// No matrix stream at all (csr_stream_spmm stages 12 B per non-zero through LDS per launch): for C5 that is 1.4 of
// 8.1 GB per fused sweep.  Rows in stored order of their class: the summation order of the CSR row.
// The workgroups walk the rows in the L2-tiled order of the optional schedule (first row of each 2*BLK/G-row block).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(BLK) void csr_rowclass_lane_spmm(RowClassDev C, VecArgs v, LaneDev T, int G,
                                                              const int* __restrict__ sched) {
  __shared__ LaneEnt ent[RL_DCAP];
  __shared__ int ptr[RL_NCLS + 1];
  __shared__ int delta[RL_NCLS];
  const int tid = threadIdx.x;
  const int nrhs = v.nrhs;
  const int rows_wg = 2 * (BLK / G);
  int bid = xcd_band(blockIdx.x, T.nblocks);
  if (sched) bid = sched[bid];
  const int grp = tid / G, c = tid - grp * G;
  const bool cact = c < nrhs;
  int row[2], s[2], len[2];
  const double* xb[2];
  double pb[2], pd[2], px[2], acc[2];
  bool live[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    row[j] = bid * rows_wg + j * (BLK / G) + grp;
    const bool in = row[j] < C.n_rows && cact;
    const int rr = row[j] < C.n_rows ? row[j] : C.n_rows - 1;
    const int cls = C.cls[rr];
    live[j] = in && cls != 0xFFFF;
    const int first = C.firstcol ? C.firstcol[rr] : rr;
    const size_t o = (size_t)rr * nrhs + (cact ? c : 0);
    pb[j] = pd[j] = px[j] = 0.0;
    acc[j] = 0.0;
    if (MODE == AXPBY) {
      if (v.beta != 0.0) pb[j] = v.beta * v.y[o];
    } else {
      pb[j] = v.b[o];
      if (MODE == SMOOTH) {
        pd[j] = v.d[rr];
        px[j] = v.xs[o];
      }
    }
    s[j] = live[j] ? cls : 0;
    xb[j] = v.x + (size_t)first * nrhs + (cact ? c : 0);
  }
  // the dictionary into LDS behind the row operands' loads, every load issued before the first LDS write (see csr_rowclass_lane_spmv)
  {
    constexpr int ND = RL_DCAP / BLK;
    LaneEnt e[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      const int ii = i < T.nent ? i : 0;
      e[u].val = C.cls_val[ii];
      e[u].off = C.cls_off[ii];
      e[u].pad = 0;
    }
    const int pv = C.cls_ptr[tid <= T.ncls ? tid : 0];
    const int dv = C.firstcol ? 0 : C.cls_delta[tid < T.ncls ? tid : 0];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      if (i < T.nent) ent[i] = e[u];
    }
    if (tid <= T.ncls) ptr[tid] = pv;
    if (tid < T.ncls) delta[tid] = dv;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int cq = s[j];
    s[j] = ptr[cq];
    len[j] = live[j] ? ptr[cq + 1] - s[j] : 0;
    xb[j] += (long long)delta[cq] * nrhs;
  }
  for (int k = 0; k < T.maxlen; k += 4) {
    double g[2][4];
    int id[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        id[j][u] = s[j] + min(k + u, len[j] > 0 ? len[j] - 1 : 0);
        g[j][u] = (k + u < len[j]) ? xb[j][(long long)ent[id[j][u]].off * nrhs] : 0.0;
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double t = acc[j] + ent[id[j][u]].val * g[j][u];
        acc[j] = (k + u < len[j]) ? t : acc[j];
      }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (live[j]) v.y[(size_t)row[j] * nrhs + c] = epilogue<MODE>(v, row[j], acc[j], pb[j], pd[j], px[j]);
}

// The same with TWO right-hand-side columns per lane (nrhs even, 16-byte aligned vectors): every access to x, b, y is a
// 16-byte load/store, a wavefront covers twice the rows, half the vector-memory instructions for the same bytes.
// Measured on C5 (16 columns): the one-column form moves 1.02 x the compulsory bytes yet runs at 3.2 TB/s - it is bound
// by the number of 8-byte-per-lane requests in flight, not by traffic.  G = pow2 >= nrhs/2 lanes per row.
// RPL rows per lane: 3 for square operators (the sweeps and residuals of A: fused sweep 1.57 -> 1.46 ms on C5), 2 for the
// transfer operators (3 would slow the prolongation from 1.01 to 1.25 ms) - chosen per operator by the host (lane_mm_rpl).
template <int MODE, int RL2_RPL>
__global__ __launch_bounds__(BLK) void csr_rowclass_lane_spmm2(RowClassDev C, VecArgs v, LaneDev T, int G,
                                                               const int* __restrict__ sched) {
  __shared__ LaneEnt ent[RL_DCAP];
  __shared__ int ptr[RL_NCLS + 1];
  __shared__ int delta[RL_NCLS];
  __shared__ double red[BLK / 64];
  const int tid = threadIdx.x;
  const int nrhs = v.nrhs;
  const int rows_wg = RL2_RPL * (BLK / G);
  int bid = xcd_band(blockIdx.x, T.nblocks);
  if (sched) bid = sched[bid];
  const int grp = tid / G, c = tid - grp * G;      // c: column PAIR
  const bool cact = 2 * c < nrhs;
  int row[RL2_RPL], s[RL2_RPL], len[RL2_RPL];
  const double* xb[RL2_RPL];
  double2 pb[RL2_RPL], px[RL2_RPL], acc[RL2_RPL];
  double pd[RL2_RPL];
  bool live[RL2_RPL];
#pragma unroll
  for (int j = 0; j < RL2_RPL; ++j) {
    row[j] = bid * rows_wg + j * (BLK / G) + grp;
    const bool in = row[j] < C.n_rows && cact;
    const int rr = row[j] < C.n_rows ? row[j] : C.n_rows - 1;
    const int cls = C.cls[rr];
    live[j] = in && cls != 0xFFFF;
    const int first = C.firstcol ? C.firstcol[rr] : rr;
    const size_t o = (size_t)rr * nrhs + (cact ? 2 * c : 0);
    pb[j] = px[j] = acc[j] = make_double2(0.0, 0.0);
    pd[j] = 0.0;
    if (MODE == AXPBY) {
      if (v.beta != 0.0) {
        const double2 yy = *reinterpret_cast<const double2*>(v.y + o);
        pb[j] = make_double2(v.beta * yy.x, v.beta * yy.y);
      }
    } else {
      pb[j] = *reinterpret_cast<const double2*>(v.b + o);
      if (MODE == SMOOTH || (MODE == RESID && v.y2)) {
        pd[j] = v.d[rr];
        px[j] = *reinterpret_cast<const double2*>(v.xs + o);
      }
    }
    s[j] = live[j] ? cls : 0;
    xb[j] = v.x + (size_t)first * nrhs + (cact ? 2 * c : 0);
  }
  // the dictionary into LDS behind the row operands' loads, every load issued before the first LDS write (see csr_rowclass_lane_spmv)
  {
    constexpr int ND = RL_DCAP / BLK;
    LaneEnt e[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      const int ii = i < T.nent ? i : 0;
      e[u].val = C.cls_val[ii];
      e[u].off = C.cls_off[ii];
      e[u].pad = 0;
    }
    const int pv = C.cls_ptr[tid <= T.ncls ? tid : 0];
    const int dv = C.firstcol ? 0 : C.cls_delta[tid < T.ncls ? tid : 0];
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int i = tid + u * BLK;
      if (i < T.nent) ent[i] = e[u];
    }
    if (tid <= T.ncls) ptr[tid] = pv;
    if (tid < T.ncls) delta[tid] = dv;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < RL2_RPL; ++j) {
    const int cq = s[j];
    s[j] = ptr[cq];
    len[j] = live[j] ? ptr[cq + 1] - s[j] : 0;
    xb[j] += (long long)delta[cq] * nrhs;
  }
  for (int k = 0; k < T.maxlen; k += 4) {
    double2 g[RL2_RPL][4];
    int id[RL2_RPL][4];
#pragma unroll
    for (int j = 0; j < RL2_RPL; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        id[j][u] = s[j] + min(k + u, len[j] > 0 ? len[j] - 1 : 0);
        g[j][u] = (k + u < len[j]) ? *reinterpret_cast<const double2*>(xb[j] + (long long)ent[id[j][u]].off * nrhs)
                                   : make_double2(0.0, 0.0);
      }
#pragma unroll
    for (int j = 0; j < RL2_RPL; ++j)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double a = ent[id[j][u]].val;
        const double t0 = acc[j].x + a * g[j][u].x;
        const double t1 = acc[j].y + a * g[j][u].y;
        acc[j].x = (k + u < len[j]) ? t0 : acc[j].x;
        acc[j].y = (k + u < len[j]) ? t1 : acc[j].y;
      }
  }
  double sq = 0.0;
#pragma unroll
  for (int j = 0; j < RL2_RPL; ++j)
    if (live[j]) {
      double2 out;
      out.x = epilogue<MODE>(v, row[j], acc[j].x, pb[j].x, pd[j], px[j].x);
      out.y = epilogue<MODE>(v, row[j], acc[j].y, pb[j].y, pd[j], px[j].y);
      const size_t o = (size_t)row[j] * nrhs + 2 * c;
      if (MODE != RESID || v.y) *reinterpret_cast<double2*>(v.y + o) = out;   // (the solve loop may need only ||r|| and x + d.*r)
      if (MODE == RESID && v.y2) {                                            // x + d.*r: the next cycle's first update
        double2 nx;
        nx.x = px[j].x + pd[j] * out.x;
        nx.y = px[j].y + pd[j] * out.y;
        *reinterpret_cast<double2*>(v.y2 + o) = nx;
      }
      sq += out.x * out.x + out.y * out.y;
    }
  if (v.sumsq) {
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o);
    if ((tid & 63) == 0) red[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < BLK / 64; ++w) t += red[w];
      v.sumsq[blockIdx.x] = t;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Element-wise helpers (grid-stride, 16 B per lane where the length allows).
// ------------------------------------------------------------------------------------------------
// x[i][c] = d[i] * b[i][c] : first damped-Jacobi sweep from x = 0 (MGcycle.jl:134 with r = b).
__global__ __launch_bounds__(BLK) void dscale_kernel(const double* __restrict__ d,
                                                     const double* __restrict__ b,
                                                     double* __restrict__ x, long long n, int nrhs) {
  const long long total = n * nrhs;
  const long long stride = (long long)gridDim.x * BLK;
  if (nrhs == 1) {
    const long long n2 = total >> 1;
    for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n2; i += stride) {
      const double2 dd = reinterpret_cast<const double2*>(d)[i];
      const double2 bb = reinterpret_cast<const double2*>(b)[i];
      reinterpret_cast<double2*>(x)[i] = make_double2(dd.x * bb.x, dd.y * bb.y);
    }
    if ((total & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[total - 1] = d[total - 1] * b[total - 1];
  } else if ((nrhs & 1) == 0 && total < (1LL << 31) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(b)) & 15u) == 0) {
    // even nrhs: 16 bytes per lane, 32-bit row arithmetic
    const unsigned n2 = (unsigned)(total >> 1), h = (unsigned)nrhs >> 1;
    for (unsigned i = blockIdx.x * BLK + threadIdx.x; i < n2; i += (unsigned)stride) {
      const double dd = d[i / h];
      const double2 bb = reinterpret_cast<const double2*>(b)[i];
      reinterpret_cast<double2*>(x)[i] = make_double2(dd * bb.x, dd * bb.y);
    }
  } else {
    for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < total; i += stride)
      x[i] = d[i / nrhs] * b[i];
  }
}

// xout[i][c] = x[i][c] + d[i] * r[i][c] : a damped-Jacobi update from an already available residual
// (MGcycle.jl:129/134 `x .+= d.*r`), used when r = b - A x is still valid from the previous step.
__global__ __launch_bounds__(BLK) void xpdr_kernel(const double* __restrict__ x,
                                                   const double* __restrict__ d,
                                                   const double* __restrict__ r,
                                                   double* __restrict__ xout, long long n, int nrhs) {
  const long long total = n * nrhs;
  const long long stride = (long long)gridDim.x * BLK;
  if (nrhs == 1) {
    const long long n2 = total >> 1;
    for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n2; i += stride) {
      const double2 xx = reinterpret_cast<const double2*>(x)[i];
      const double2 dd = reinterpret_cast<const double2*>(d)[i];
      const double2 rr = reinterpret_cast<const double2*>(r)[i];
      reinterpret_cast<double2*>(xout)[i] = make_double2(xx.x + dd.x * rr.x, xx.y + dd.y * rr.y);
    }
    if ((total & 1) && blockIdx.x == 0 && threadIdx.x == 0)
      xout[total - 1] = x[total - 1] + d[total - 1] * r[total - 1];
  } else if ((nrhs & 1) == 0 && total < (1LL << 31) && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(r) |
                                                        reinterpret_cast<uintptr_t>(xout)) & 15u) == 0) {
    // even nrhs: 16 bytes per lane, 32-bit row arithmetic
    const unsigned n2 = (unsigned)(total >> 1), h = (unsigned)nrhs >> 1;
    for (unsigned i = blockIdx.x * BLK + threadIdx.x; i < n2; i += (unsigned)stride) {
      const double dd = d[i / h];
      const double2 xx = reinterpret_cast<const double2*>(x)[i];
      const double2 rr = reinterpret_cast<const double2*>(r)[i];
      reinterpret_cast<double2*>(xout)[i] = make_double2(xx.x + dd * rr.x, xx.y + dd * rr.y);
    }
  } else {
    for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < total; i += stride)
      xout[i] = x[i] + d[i / nrhs] * r[i];
  }
}

// xout = x + d.*r with the class-constant relaxPrec of a row-class operator: 2 B/row of class ids instead of 8 B/row
__global__ __launch_bounds__(BLK) void xpdr_cls_kernel(const double* __restrict__ x,
                                                       const unsigned short* __restrict__ cls,
                                                       const double* __restrict__ dcls,
                                                       const double* __restrict__ d,
                                                       const double* __restrict__ r,
                                                       double* __restrict__ xout, long long n) {
  const long long stride = (long long)gridDim.x * BLK;
  const long long n2 = n >> 1;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n2; i += stride) {
    const double2 xx = reinterpret_cast<const double2*>(x)[i];
    const ushort2 cc = reinterpret_cast<const ushort2*>(cls)[i];
    const double2 rr = reinterpret_cast<const double2*>(r)[i];
    const double d0 = cc.x != 0xFFFF ? dcls[cc.x] : d[2 * i];       // exception rows: d from memory
    const double d1 = cc.y != 0xFFFF ? dcls[cc.y] : d[2 * i + 1];
    reinterpret_cast<double2*>(xout)[i] = make_double2(xx.x + d0 * rr.x, xx.y + d1 * rr.y);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const unsigned short c = cls[n - 1];
    xout[n - 1] = x[n - 1] + (c != 0xFFFF ? dcls[c] : d[n - 1]) * r[n - 1];
  }
}

__global__ __launch_bounds__(BLK) void fill_kernel(double* __restrict__ x, long long n, double val) {
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) x[i] = val;
}

// out[i*nrhs + c] = in[c*n + i]   (column-major host block -> row-major device block) and back.
__global__ __launch_bounds__(BLK) void colmajor_to_rowmajor(const double* __restrict__ in,
                                                            double* __restrict__ out, long long n,
                                                            int nrhs) {
  const long long total = n * nrhs;
  const long long stride = (long long)gridDim.x * BLK;
  for (long long o = (long long)blockIdx.x * BLK + threadIdx.x; o < total; o += stride) {
    const long long i = o / nrhs;
    const int c = (int)(o - i * nrhs);
    out[o] = in[(long long)c * n + i];
  }
}
__global__ __launch_bounds__(BLK) void rowmajor_to_colmajor(const double* __restrict__ in,
                                                            double* __restrict__ out, long long n,
                                                            int nrhs) {
  const long long total = n * nrhs;
  const long long stride = (long long)gridDim.x * BLK;
  for (long long o = (long long)blockIdx.x * BLK + threadIdx.x; o < total; o += stride) {
    const int c = (int)(o / n);
    const long long i = o - (long long)c * n;
    out[o] = in[i * nrhs + c];
  }
}

// ------------------------------------------------------------------------------------------------
// Sum of squares, deterministic two-stage reduction (SolveFuncs.jl:15,20,30: Frobenius norm).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double block_sum(double acc, double* red) {
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < BLK / 64; ++w) s += red[w];
  return s;  // valid in thread 0
}

__global__ __launch_bounds__(BLK) void sumsq_partial(const double* __restrict__ x, long long n,
                                                     double* __restrict__ partial) {
  __shared__ double red[BLK / 64];
  const long long stride = (long long)gridDim.x * BLK;
  double acc = 0.0;
  const long long n2 = n >> 1;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n2; i += stride) {
    const double2 t = reinterpret_cast<const double2*>(x)[i];
    acc += t.x * t.x + t.y * t.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += x[n - 1] * x[n - 1];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// sum of squares over a sub-box [lo, hi) of an x-fastest n1 x n2 x n3 grid vector (the owned box inside a rank's extended box:
// the norms of the sharded solve count every global row once).  One (y, z) line of the sub-box per wavefront and trip.
struct BoxDev { int n1, n2, n3, lo[3], hi[3]; };
__global__ __launch_bounds__(BLK) void sumsq_box_partial(const double* __restrict__ x, BoxDev B, double* __restrict__ partial) {
  __shared__ double red[BLK / 64];
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * BLK) >> 6;
  const int ly = B.hi[1] - B.lo[1], lz = B.hi[2] - B.lo[2];
  double acc = 0.0;
  for (long long q = wave; q < (long long)ly * lz; q += nwaves) {
    const int z = (int)(q / ly) + B.lo[2], y = (int)(q % ly) + B.lo[1];
    const double* line = x + ((long long)z * B.n2 + y) * B.n1;
    for (int i = B.lo[0] + lane; i < B.hi[0]; i += 64) acc += line[i] * line[i];
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// x'y over the same sub-box (the dots of the sharded Krylov drivers count every global row once)
__global__ __launch_bounds__(BLK) void dot_box_partial(const double* __restrict__ x, const double* __restrict__ y, BoxDev B, double* __restrict__ partial) {
  __shared__ double red[BLK / 64];
  const int lane = threadIdx.x & 63;
  const long long wave = ((long long)blockIdx.x * BLK + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * BLK) >> 6;
  const int ly = B.hi[1] - B.lo[1], lz = B.hi[2] - B.lo[2];
  double acc = 0.0;
  for (long long q = wave; q < (long long)ly * lz; q += nwaves) {
    const int z = (int)(q / ly) + B.lo[2], yy = (int)(q % ly) + B.lo[1];
    const long long base = ((long long)z * B.n2 + yy) * B.n1;
    for (int i = B.lo[0] + lane; i < B.hi[0]; i += 64) acc += x[base + i] * y[base + i];
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// ghost layers of a sharded level: dst[idx[i]] = src[i] (the received values into their places in the extended box)
__global__ __launch_bounds__(BLK) void ghost_unpack(const double* __restrict__ src, const int* __restrict__ idx, double* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  if (i < n) dst[idx[i]] = src[i];
}
__global__ __launch_bounds__(BLK) void ghost_pack(const double* __restrict__ src, const int* __restrict__ idx, double* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}
// two vectors of one level in one exchange (the restricted right-hand side and the first update d.*bc the restriction wrote beside it)
__global__ __launch_bounds__(BLK) void ghost_pack2(const double* __restrict__ s1, const double* __restrict__ s2, const int* __restrict__ idx,
                                                   double* __restrict__ d1, double* __restrict__ d2, long long n) {
  const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  if (i < n) {
    const int j = idx[i];
    d1[i] = s1[j];
    d2[i] = s2[j];
  }
}
__global__ __launch_bounds__(BLK) void ghost_unpack2(const double* __restrict__ s1, const double* __restrict__ s2, const int* __restrict__ idx,
                                                     double* __restrict__ d1, double* __restrict__ d2, long long n) {
  const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  if (i < n) {
    const int j = idx[i];
    d1[j] = s1[i];
    d2[j] = s2[i];
  }
}

__global__ __launch_bounds__(BLK) void sum_partial(const double* __restrict__ x, long long n,
                                                   double* __restrict__ partial) {
  __shared__ double red[BLK / 64];
  const long long stride = (long long)gridDim.x * BLK;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) acc += x[i];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(BLK) void sum_final(const double* __restrict__ partial, int np,
                                                 double* __restrict__ out) {
  __shared__ double red[BLK / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += BLK) acc += partial[i];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = s;
}

// the same, with the sum also stored straight into pinned host memory: the solve loop's stopping test reads it there once the
// stream has drained - no 8-byte copy (a blit kernel of its own, ~4 us on the critical path of every step)
__global__ __launch_bounds__(BLK) void sum_final_mirror(const double* __restrict__ partial, int np, double* __restrict__ out,
                                                        double* __restrict__ host_mirror) {
  __shared__ double red[BLK / 64];
  double acc = 0.0;
  for (int i = threadIdx.x; i < np; i += BLK) acc += partial[i];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) {
    out[0] = s;
    host_mirror[0] = s;
  }
}

// dot(x, y), first stage (second stage: sum_final)
__global__ __launch_bounds__(BLK) void dot_partial(const double* __restrict__ x, const double* __restrict__ y,
                                                   long long n, double* __restrict__ partial) {
  __shared__ double red[BLK / 64];
  const long long stride = (long long)gridDim.x * BLK;
  double acc = 0.0;
  const long long n2 = n >> 1;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n2; i += stride) {
    const double2 a = reinterpret_cast<const double2*>(x)[i];
    const double2 b = reinterpret_cast<const double2*>(y)[i];
    acc += a.x * b.x + a.y * b.y;
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += x[n - 1] * y[n - 1];
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// CG updates: x += alpha*p ; r -= alpha*Ap   (KrylovMethods cg: two BLAS.axpy!)
__global__ __launch_bounds__(BLK) void cg_update_xr(double alpha, const double* __restrict__ p,
                                                    const double* __restrict__ Ap, double* __restrict__ x,
                                                    double* __restrict__ r, long long n) {
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
    x[i] += alpha * p[i];
    r[i] -= alpha * Ap[i];
  }
}

// the same with the per-workgroup sums of squares of the new r (||r|| of the stopping test without another pass)
__global__ __launch_bounds__(BLK) void cg_update_xr_norm(double alpha, const double* __restrict__ p,
                                                         const double* __restrict__ Ap, double* __restrict__ x,
                                                         double* __restrict__ r, long long n, double* __restrict__ partial) {
  __shared__ double red[BLK / 64];
  const long long stride = (long long)gridDim.x * BLK;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
    x[i] += alpha * p[i];
    const double rn = r[i] - alpha * Ap[i];
    r[i] = rn;
    acc += rn * rn;
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// One step of modified Gram-Schmidt with the coefficient on the device: w -= (*hk) * v, and the per-workgroup partial sums
// of the NEXT dot in the same pass - w_new . u (u = the next basis vector) or, with u == nullptr, w_new . w_new (the norm
// that ends the orthogonalisation).  The chain dot -> update -> dot never leaves the device (FGMRES Arnoldi loop).
__global__ __launch_bounds__(BLK) void mgs_step(const double* __restrict__ hk, const double* __restrict__ v,
                                                double* __restrict__ w, const double* __restrict__ u, long long n,
                                                double* __restrict__ partial) {
  __shared__ double red[BLK / 64];
  const double a = -hk[0];
  const long long stride = (long long)gridDim.x * BLK;
  double acc = 0.0;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
    const double wn = a * v[i] + 1.0 * w[i];
    w[i] = wn;
    acc += wn * (u ? u[i] : wn);
  }
  const double s = block_sum(acc, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// w *= 1/sqrt(*wn2) unless *wn2 == 0 (the new basis vector of the Arnoldi loop)
__global__ __launch_bounds__(BLK) void scale_rsqrt(const double* __restrict__ wn2, double* __restrict__ w, long long n) {
  const double nrm = sqrt(wn2[0]);
  if (nrm == 0.0) return;
  const double a = 1.0 / nrm;
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) w[i] = a * w[i];
}

// y = a*x + b*y
__global__ __launch_bounds__(BLK) void axpby_kernel(double a, const double* __restrict__ x, double b,
                                                    double* __restrict__ y, long long n) {
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride)
    y[i] = (b == 0.0) ? a * x[i] : a * x[i] + b * y[i];
}

// p = z + beta*p   (scal! + axpy!)
__global__ __launch_bounds__(BLK) void cg_update_p(double beta, const double* __restrict__ z,
                                                   double* __restrict__ p, long long n) {
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) p[i] = beta * p[i] + z[i];
}

// ------------------------------------------------------------------------------------------------
// Coarsest solve x = Ainv * b with the explicit inverse (row-major n x n), MGcycle.jl:177.
// One wavefront per (row, rhs column): coalesced sweep of the row of Ainv, shuffle reduction.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void dense_apply(const double* __restrict__ Ainv,
                                                   const double* __restrict__ b,
                                                   double* __restrict__ x, int n, int nrhs) {
  const int wave = (int)(((long long)blockIdx.x * BLK + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (wave >= n * nrhs) return;  // wave-uniform
  const int row = wave / nrhs;
  const int c = wave - row * nrhs;
  const double* a = Ainv + (size_t)row * n;
  double acc = 0.0;
  // (unrolled: the loads of eight steps go out together, the products are still added in the order j = lane, lane + 64, ...;
  // as a plain loop every step waited for its own two loads - twelve round trips for the 729-row inverse of C2)
#pragma unroll 8
  for (int j = lane; j < n; j += 64) acc += a[j] * b[(size_t)j * nrhs + c];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) x[(size_t)row * nrhs + c] = acc;
}

// ------------------------------------------------------------------------------------------------
// Coarsest solve with sparse triangular factors, x[q] = U \ (L \ b[p]) - the reference's own native scheme for
// applying Julia's (UMFPACK) factors: deps/src/parLU.cpp:120-190, CSR factors with L's diagonal LAST and U's
// diagonal FIRST in every row.  Used when the coarsest level is too large for the explicit inverse.
// One 1024-thread workgroup walks the dependency LEVELS of L and then of U (rows of one level are independent;
// level sets are computed on the host at setup), one wavefront per row, a workgroup barrier between levels:
// no inter-workgroup waiting, so nothing can hang.  Latency-bound by design (a coarse level).
// ------------------------------------------------------------------------------------------------
struct LuDev {
  int n;
  const int* Lptr; const int* Lcol; const double* Lval;   // CSR, diagonal last
  const int* Uptr; const int* Ucol; const double* Uval;   // CSR, diagonal first
  const int* p; const int* q;                             // 0-based permutations
  const int* Lorder; const int* Llvl; int nLlvl;          // rows sorted by level, level pointers
  const int* Uorder; const int* Ulvl; int nUlvl;
};

// One triangular sweep over the dependency levels.  A level with >= 16 rows gives every wavefront its own rows; a
// level with fewer rows (the dense trailing supernodes of a factor are chains of single-row levels) splits each row
// over g = 16/rows wavefronts and combines their partial sums through LDS.  Up to 4 right-hand-side columns travel
// together (y is row-interleaved like every other device vector).
template <bool LOWER>
__device__ __forceinline__ void sptrsv_sweep(const int* __restrict__ ptr, const int* __restrict__ col,
                                             const double* __restrict__ val, const int* __restrict__ order,
                                             const int* __restrict__ lvl, int nlvl, const int* __restrict__ perm,
                                             const double* __restrict__ b, double* y, int nrhs, int c0, int nc,
                                             double (*sred)[4]) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int l = 0; l < nlvl; ++l) {
    const int t0 = lvl[l], t1 = lvl[l + 1], cnt = t1 - t0;
    int g = 1;
    while (g * 2 * cnt <= nw) g *= 2;
    const int part = wave & (g - 1), stride = nw / g;
    for (int t = t0 + wave / g; t < t1; t += stride) {
      const int row = order[t];
      const int s = LOWER ? ptr[row] : ptr[row] + 1;
      const int e = LOWER ? ptr[row + 1] - 1 : ptr[row + 1];
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = s + part * 64 + lane; k < e; k += 64 * g) {
        const double v = val[k];
        const double* yy = y + (size_t)col[k] * nrhs + c0;
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (u < nc) acc[u] += v * yy[u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        for (int o = 32; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o);
      if (g == 1) {
        if (lane < nc) {
          const double dg = val[LOWER ? e : s - 1];
          const double rhs = LOWER ? b[(size_t)perm[row] * nrhs + c0 + lane] : y[(size_t)row * nrhs + c0 + lane];
          const double a = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
          y[(size_t)row * nrhs + c0 + lane] = (rhs - a) / dg;
        }
      } else if (lane < 4) {
        sred[wave][lane] = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
      }
    }
    if (g > 1) {
      __syncthreads();
      const int t = t0 + wave / g;
      if (part == 0 && t < t1 && lane < nc) {
        const int row = order[t];
        double a = 0.0;
        for (int w = 0; w < g; ++w) a += sred[wave + w][lane];
        const double dg = val[LOWER ? ptr[row + 1] - 1 : ptr[row]];
        const double rhs = LOWER ? b[(size_t)perm[row] * nrhs + c0 + lane] : y[(size_t)row * nrhs + c0 + lane];
        y[(size_t)row * nrhs + c0 + lane] = (rhs - a) / dg;
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(1024) void sptrsv_lu(LuDev F, const double* __restrict__ b, double* __restrict__ x,
                                                  double* work, int nrhs) {
  __shared__ double sred[16][4];
  for (int c0 = 0; c0 < nrhs; c0 += 4) {
    const int nc = min(4, nrhs - c0);
    sptrsv_sweep<true>(F.Lptr, F.Lcol, F.Lval, F.Lorder, F.Llvl, F.nLlvl, F.p, b, work, nrhs, c0, nc, sred);   // y = L \\ b[p]
    sptrsv_sweep<false>(F.Uptr, F.Ucol, F.Uval, F.Uorder, F.Ulvl, F.nUlvl, F.p, b, work, nrhs, c0, nc, sred);  // y = U \\ y
    for (int i = threadIdx.x; i < F.n * nc; i += blockDim.x) {
      const int r = i / nc, u = i - r * nc;
      x[(size_t)F.q[r] * nrhs + c0 + u] = work[(size_t)r * nrhs + c0 + u];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------
// The same solve spread over the whole chip (factors of more than a few thousand rows).  The dependency levels of
// a fill-reducing factorisation fall into two regimes: wide levels (thousands of independent rows) at the start of
// the elimination and, at its end, the dense trailing supernode - a CHAIN of single-row levels (2732 of the 4620
// levels of L for a 33^3 Poisson level).  The first regime gets one launch per level, one wavefront per row over as
// many workgroups as the level has rows/4; the chain is not substituted at all: its dense triangular block D is
// inverted once at setup (tri_inverse) and applied as a dense product (dense_apply), so the 2732 dependent steps
// become one bandwidth-bound kernel.  No inter-workgroup waiting anywhere.
//   sptrsv_level<LOWER>  rows order[t0..t1) of one level
//   sptrsv_tail_rhs      t = b[p] - L21*y1 for the rows of the trailing block of L
//   sptrsv_scatter       x[q[r]] = y[r]
// ------------------------------------------------------------------------------------------------
template <bool LOWER>
__global__ __launch_bounds__(BLK) void sptrsv_level(LuDev F, const int4* __restrict__ slots, int t0, int t1,
                                                    const double* __restrict__ b, double* y, int nrhs) {
  const int t = t0 + (int)(((long long)blockIdx.x * BLK + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (t >= t1) return;  // wave-uniform
  const int* __restrict__ col = LOWER ? F.Lcol : F.Ucol;
  const double* __restrict__ val = LOWER ? F.Lval : F.Uval;
  const int4 sl = slots[t];                       // one load instead of order[] -> ptr[]: the level is latency-bound
  const int row = sl.x, s = sl.y, e = sl.z;
  const double dg = val[sl.w];
  for (int c0 = 0; c0 < nrhs; c0 += 4) {
    const int nc = min(4, nrhs - c0);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = s + lane; k < e; k += 64) {
      const double v = val[k];
      const double* yy = y + (size_t)col[k] * nrhs + c0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < nc) acc[u] += v * yy[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      for (int o = 32; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o);
    if (lane < nc) {
      const double rhs = LOWER ? b[(size_t)F.p[row] * nrhs + c0 + lane] : y[(size_t)row * nrhs + c0 + lane];
      const double a = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
      y[(size_t)row * nrhs + c0 + lane] = (rhs - a) / dg;
    }
  }
}

__global__ __launch_bounds__(BLK) void sptrsv_tail_rhs(LuDev F, int n0, const double* __restrict__ b,
                                                       const double* __restrict__ y, double* __restrict__ t, int nrhs) {
  const int i = (int)(((long long)blockIdx.x * BLK + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  const int row = n0 + i;
  if (row >= F.n) return;  // wave-uniform
  const int s = F.Lptr[row], e = F.Lptr[row + 1] - 1;
  for (int c0 = 0; c0 < nrhs; c0 += 4) {
    const int nc = min(4, nrhs - c0);
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int k = s + lane; k < e; k += 64) {
      const int c = F.Lcol[k];
      if (c >= n0) continue;                       // the trailing block itself is applied through its inverse
      const double v = F.Lval[k];
      const double* yy = y + (size_t)c * nrhs + c0;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (u < nc) acc[u] += v * yy[u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      for (int o = 32; o > 0; o >>= 1) acc[u] += __shfl_xor(acc[u], o);
    if (lane < nc) {
      const double a = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
      t[(size_t)i * nrhs + c0 + lane] = b[(size_t)F.p[row] * nrhs + c0 + lane] - a;
    }
  }
}

__global__ __launch_bounds__(BLK) void sptrsv_scatter(const int* __restrict__ q, const double* __restrict__ y,
                                                      double* __restrict__ x, int n, int nrhs) {
  const long long i = (long long)blockIdx.x * BLK + threadIdx.x;
  if (i >= (long long)n * nrhs) return;
  const int r = (int)(i / nrhs), u = (int)(i - (long long)r * nrhs);
  x[(size_t)q[r] * nrhs + u] = y[i];
}

// Dense block (row-major, leading dimension ld = M rounded up to 64, zero outside the triangle, unit diagonal in the
// padding) of the rows/columns n0.. of a factor in CSR.
__global__ __launch_bounds__(BLK) void tri_gather_block(const int* __restrict__ ptr, const int* __restrict__ col,
                                                        const double* __restrict__ val, int n0, int M, int ld,
                                                        double* __restrict__ D) {
  const int i = (int)(((long long)blockIdx.x * BLK + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (i >= ld) return;
  if (i >= M) {
    if (lane == 0) D[(size_t)i * ld + i] = 1.0;
    return;
  }
  for (int k = ptr[n0 + i] + lane; k < ptr[n0 + i + 1]; k += 64) {
    const int c = col[k] - n0;
    if (c >= 0) D[(size_t)i * ld + c] = val[k];
  }
}

// X = inv(D) for a dense triangular block (ld x ld, ld a multiple of 64), blocked: one 256-thread workgroup per block
// of 64 columns of X walks the 64-row panels in substitution order; for each panel the contribution of the rows
// already known is a 64 x K x 64 product through LDS tiles (each thread a 4 x 4 register tile), the 64 x 64 diagonal
// block is then substituted in LDS by one wavefront (a column per lane).  Setup-time kernel, ~M^3/3 flops.
template <bool LOWER>
__global__ __launch_bounds__(256) void tri_inverse(const double* __restrict__ D, double* X, int ld) {
  __shared__ double As[64][17];
  __shared__ double Bs[16][64];
  __shared__ double Ts[64][65];
  __shared__ double Ds[64][65];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int C0 = blockIdx.x * 64, nP = ld / 64, cb = blockIdx.x;
  // rows outside the triangle of this column block
  for (int i = LOWER ? 0 : C0 + 64; i < (LOWER ? C0 : ld); i += 4) {
    const int r = i + (tid >> 6);
    if (r < (LOWER ? C0 : ld)) X[(size_t)r * ld + C0 + (tid & 63)] = 0.0;
  }
  for (int pp = 0; pp < (LOWER ? nP - cb : cb + 1); ++pp) {
    const int p = LOWER ? cb + pp : cb - pp;
    const int I0 = p * 64;
    const int K0 = LOWER ? C0 : I0 + 64, K1 = LOWER ? I0 : C0 + 64;
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
    for (int k0 = K0; k0 < K1; k0 += 16) {
      // D[I0 + r][k0 + kk]: 64 x 16; X[k0 + kk][C0 + c]: 16 x 64 - four elements per thread each
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = tid + 256 * u;
        As[e >> 4][e & 15] = D[(size_t)(I0 + (e >> 4)) * ld + k0 + (e & 15)];
        Bs[e >> 6][e & 63] = X[(size_t)(k0 + (e >> 6)) * ld + C0 + (e & 63)];
      }
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        double a[4], bb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = As[ty * 4 + i][kk];
#pragma unroll
        for (int j = 0; j < 4; ++j) bb[j] = Bs[kk][tx * 4 + j];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * bb[j];
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = ty * 4 + i, c = tx * 4 + j;
        Ts[r][c] = ((I0 + r == C0 + c) ? 1.0 : 0.0) - acc[i][j];
      }
    for (int e = tid; e < 64 * 64; e += 256) Ds[e >> 6][e & 63] = D[(size_t)(I0 + (e >> 6)) * ld + I0 + (e & 63)];
    __syncthreads();
    if (tid < 64) {                                 // one wavefront, a column per lane: no barriers needed inside
      const int c = tid;
      if (LOWER) {
        for (int r = 0; r < 64; ++r) {
          double v = Ts[r][c];
          for (int k = 0; k < r; ++k) v -= Ds[r][k] * Ts[k][c];
          Ts[r][c] = v / Ds[r][r];
        }
      } else {
        for (int r = 63; r >= 0; --r) {
          double v = Ts[r][c];
          for (int k = r + 1; k < 64; ++k) v -= Ds[r][k] * Ts[k][c];
          Ts[r][c] = v / Ds[r][r];
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < 64 * 64; e += 256) X[(size_t)(I0 + (e >> 6)) * ld + C0 + (e & 63)] = Ts[e >> 6][e & 63];
    __threadfence();                                // the next panel of this workgroup reads these rows back
    __syncthreads();
  }
}

// x = T * b for a dense TRIANGULAR M x M block (row-major, leading dimension ld): dense_apply that only reads the triangle.
template <bool LOWER>
__global__ __launch_bounds__(BLK) void tri_apply(const double* __restrict__ T, int ld, const double* __restrict__ b,
                                                 double* __restrict__ x, int M, int nrhs) {
  const int wave = (int)(((long long)blockIdx.x * BLK + threadIdx.x) >> 6);
  const int lane = threadIdx.x & 63;
  if (wave >= M * nrhs) return;  // wave-uniform
  const int row = wave / nrhs, c = wave - row * nrhs;
  const double* __restrict__ a = T + (size_t)row * ld;
  const int j0 = LOWER ? 0 : (row & ~63), j1 = LOWER ? row + 1 : M;
  double acc = 0.0;
  for (int j = j0 + lane; j < j1; j += 64) acc += a[j] * b[(size_t)j * nrhs + c];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0) x[(size_t)row * nrhs + c] = acc;
}

// ------------------------------------------------------------------------------------------------
// Numeric Galerkin product on a FIXED sparsity: C = R*(A*P) (replaceMatrixInHierarchy, MGsetup.jl:226-270:
// `Act = Ps[l]*AT*Rs[l]` with unchanged P, R; the pattern of C is the one the host setup produced).
// One wavefront (a 64-thread workgroup) per coarse row i.  DETERMINISTIC: the entries (i,k) of R's row and (k,j) of
// A's rows are walked one after the other in stored order; only the entries (j,c) of ONE row of P - distinct target
// columns - are spread over the lanes, each lane finds its column in C's (sorted) row by binary search in LDS and adds
// there without atomics.  Every entry of C is therefore the same sum in the same order on every run.
// ------------------------------------------------------------------------------------------------
constexpr int RAP_CAP = 2048;
__global__ __launch_bounds__(64) void rap_numeric(CsrDev R, CsrDev A, CsrDev P, const int* __restrict__ Crowptr,
                                                  const int* __restrict__ Ccol, double* __restrict__ Cval, int chunk) {
  __shared__ int scol[RAP_CAP];
  __shared__ double sacc[RAP_CAP];
  const int i = blockIdx.x;
  const int lane = threadIdx.x;
  const int c0 = Crowptr[i];
  const int len = Crowptr[i + 1] - c0;
  // Rows of C longer than RAP_CAP (SA-AMG middle levels: thousands of entries per row) are accumulated RAP_CAP target
  // columns at a time: the same walk per chunk, contributions to columns outside the chunk skipped - every entry of C is
  // still the same sum in the same order, at (number of chunks) times the walk.
  for (int t0 = 0; t0 < len; t0 += chunk) {          // chunk <= RAP_CAP (host)
    const int clen = len - t0 < chunk ? len - t0 : chunk;
    __syncthreads();
    for (int t = lane; t < clen; t += 64) {
      scol[t] = Ccol[c0 + t0 + t];
      sacc[t] = 0.0;
    }
    __syncthreads();
    const int cmin = scol[0], cmax = scol[clen - 1];
    for (int kk = R.rowptr[i]; kk < R.rowptr[i + 1]; ++kk) {        // wave-uniform, stored order
      const int k = R.colidx[kk];
      const double rv = R.val[kk];
      for (int jj = A.rowptr[k]; jj < A.rowptr[k + 1]; ++jj) {      // wave-uniform, stored order
        const int j = A.colidx[jj];
        const double ra = rv * A.val[jj];
        for (int pp = P.rowptr[j] + lane; pp < P.rowptr[j + 1]; pp += 64) {   // distinct columns: one lane each
          const int c = P.colidx[pp];
          if (c < cmin || c > cmax) continue;                       // (another chunk's column)
          int lo = 0, hi = clen - 1;
          while (lo < hi) {  // the pattern of C contains every reachable column by construction
            const int mid = (lo + hi) >> 1;
            if (scol[mid] < c) lo = mid + 1;
            else hi = mid;
          }
          if (scol[lo] == c) sacc[lo] += ra * P.val[pp];
        }
        __builtin_amdgcn_wave_barrier();   // one wavefront: LDS accesses of the next row of P follow in program order
      }
    }
    __syncthreads();
    for (int t = lane; t < clen; t += 64) Cval[c0 + t0 + t] = sacc[t];
  }
}

// d[i] = omega / a_ii  (getRelaxPrec "Jac", MGsetup.jl:145-147); s[j] += a_ij^2 (getSPAIprec, MGsetup.jl:359-362)
__global__ __launch_bounds__(BLK) void relax_jacobi(CsrDev A, double omega, double* __restrict__ d) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= A.n_rows) return;
  double diag = 0.0;
  for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
    if (A.colidx[k] == i) diag = A.val[k];
  d[i] = omega / diag;
}
// s[j] = sum over the entries of COLUMN j of a_ij^2 in ascending row order - the order in which Julia's
// sum(AT.^2, dims=2) (MGsetup.jl:360) and a row-major pass over A accumulate them: no atomics, the same bits on every run.
// tptr / tperm: the transposed pattern (entries of column j are tperm[tptr[j] .. tptr[j+1]), ascending rows).
__global__ __launch_bounds__(BLK) void colsumsq_kernel(const double* __restrict__ val, const int* __restrict__ tptr,
                                                       const int* __restrict__ tperm, int n_cols, double* __restrict__ s) {
#pragma clang fp contract(off)   // squares rounded, then summed (no fused multiply-add): AT.^2 is an array in the reference
  const int j = blockIdx.x * BLK + threadIdx.x;
  if (j >= n_cols) return;
  double acc = 0.0;
  for (int k = tptr[j]; k < tptr[j + 1]; ++k) {
    const double a = val[tperm[k]];
    const double sq = a * a;
    acc = acc + sq;
  }
  s[j] = acc;
}
__global__ __launch_bounds__(BLK) void relax_spai(CsrDev A, double omega, const double* __restrict__ s,
                                                  double* __restrict__ d) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= A.n_rows) return;
  double diag = 0.0;
  for (int k = A.rowptr[i]; k < A.rowptr[i + 1]; ++k)
    if (A.colidx[k] == i) diag = A.val[k];
  d[i] = omega * (diag / s[i]);   // relaxParam * getSPAIprec(AT) (MGsetup.jl:148, 361): the quotient first
}

// ------------------------------------------------------------------------------------------------
// Block-vector helpers of the block Krylov drivers (blocks are row-major [n][k], k <= 16).
//   blk_gram_partial / blk_gram_final:  G = X' Y  (k x k), deterministic two-stage reduction
//   blk_comb:                           out[i,:] = s * add[i,:] + in[i,:] * C   (C k x k row-major; out may alias in / add)
// ------------------------------------------------------------------------------------------------
constexpr int BLK_KMAX = 16;
__global__ __launch_bounds__(BLK) void blk_gram_partial(const double* __restrict__ X, const double* __restrict__ Y,
                                                        long long n, int k, double* __restrict__ partial) {
  __shared__ double red[BLK / 64][BLK_KMAX];
  const int a = blockIdx.y;
  double acc[BLK_KMAX];
#pragma unroll
  for (int b = 0; b < BLK_KMAX; ++b) acc[b] = 0.0;
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
    const double xa = X[i * k + a];
#pragma unroll
    for (int b = 0; b < BLK_KMAX; ++b)
      if (b < k) acc[b] += xa * Y[i * k + b];
  }
#pragma unroll
  for (int b = 0; b < BLK_KMAX; ++b) {
    double t = acc[b];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][b] = t;
  }
  __syncthreads();
  if (threadIdx.x < k) {
    double t = 0.0;
    for (int w = 0; w < BLK / 64; ++w) t += red[w][threadIdx.x];
    partial[((size_t)blockIdx.x * k + a) * k + threadIdx.x] = t;
  }
}
// the same over the rows of a sub-box of an x-fastest grid (the owned box inside a rank's extended box: the Gram matrices of the sharded
// block Krylov drivers count every global row once)
__global__ __launch_bounds__(BLK) void blk_gram_box_partial(const double* __restrict__ X, const double* __restrict__ Y, BoxDev B, int k,
                                                            double* __restrict__ partial) {
  __shared__ double red[BLK / 64][BLK_KMAX];
  const int a = blockIdx.y;
  double acc[BLK_KMAX];
#pragma unroll
  for (int b = 0; b < BLK_KMAX; ++b) acc[b] = 0.0;
  const int lx = B.hi[0] - B.lo[0], ly = B.hi[1] - B.lo[1], lz = B.hi[2] - B.lo[2];
  const long long nown = (long long)lx * ly * lz, stride = (long long)gridDim.x * BLK;
  for (long long q = (long long)blockIdx.x * BLK + threadIdx.x; q < nown; q += stride) {
    const int z = (int)(q / ((long long)lx * ly)), rem = (int)(q - (long long)z * lx * ly), y = rem / lx, x = rem - y * lx;
    const long long i = ((long long)(z + B.lo[2]) * B.n2 + (y + B.lo[1])) * B.n1 + (x + B.lo[0]);
    const double xa = X[i * k + a];
#pragma unroll
    for (int b = 0; b < BLK_KMAX; ++b)
      if (b < k) acc[b] += xa * Y[i * k + b];
  }
#pragma unroll
  for (int b = 0; b < BLK_KMAX; ++b) {
    double t = acc[b];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][b] = t;
  }
  __syncthreads();
  if (threadIdx.x < k) {
    double t = 0.0;
    for (int w = 0; w < BLK / 64; ++w) t += red[w][threadIdx.x];
    partial[((size_t)blockIdx.x * k + a) * k + threadIdx.x] = t;
  }
}
// ghost layers of a row-major block [n][k]: dst[i][:] = src[idx[i]][:] / dst[idx[i]][:] = src[i][:]
__global__ __launch_bounds__(BLK) void ghost_pack_block(const double* __restrict__ src, const int* __restrict__ idx, double* __restrict__ dst, long long n, int k) {
  const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
  if (t >= n * k) return;
  const long long i = t / k;
  const int c = (int)(t - i * k);
  dst[t] = src[(long long)idx[i] * k + c];
}
__global__ __launch_bounds__(BLK) void ghost_unpack_block(const double* __restrict__ src, const int* __restrict__ idx, double* __restrict__ dst, long long n, int k) {
  const long long t = (long long)blockIdx.x * BLK + threadIdx.x;
  if (t >= n * k) return;
  const long long i = t / k;
  const int c = (int)(t - i * k);
  dst[(long long)idx[i] * k + c] = src[t];
}
__global__ __launch_bounds__(BLK) void blk_gram_final(const double* __restrict__ partial, int nb, int k,
                                                      double* __restrict__ out) {
  const int e = threadIdx.x;   // entry a*k + b
  if (e >= k * k) return;
  double t = 0.0;
  for (int p = 0; p < nb; ++p) t += partial[(size_t)p * k * k + e];
  out[e] = t;
}
__global__ __launch_bounds__(BLK) void blk_comb(double* out, const double* add, double s, const double* in,
                                                const double* __restrict__ Cm, long long n, int k) {
  __shared__ double sc[BLK_KMAX * BLK_KMAX];
  for (int t = threadIdx.x; t < k * k; t += BLK) sc[t] = Cm[t];
  __syncthreads();
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) {
    double row[BLK_KMAX], o[BLK_KMAX];
#pragma unroll
    for (int a = 0; a < BLK_KMAX; ++a) row[a] = a < k ? in[i * k + a] : 0.0;
#pragma unroll
    for (int b = 0; b < BLK_KMAX; ++b) o[b] = (b < k && add) ? s * add[i * k + b] : 0.0;
#pragma unroll
    for (int a = 0; a < BLK_KMAX; ++a)
      if (a < k) {
#pragma unroll
        for (int b = 0; b < BLK_KMAX; ++b)
          if (b < k) o[b] += row[a] * sc[a * k + b];
      }
#pragma unroll
    for (int b = 0; b < BLK_KMAX; ++b)
      if (b < k) out[i * k + b] = o[b];
  }
}
// float <-> double conversion of a block (mixed-precision preconditioner hook, SolveFuncs.jl:52-58)
__global__ __launch_bounds__(BLK) void f32_to_f64(const float* __restrict__ in, double* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) out[i] = (double)in[i];
}
__global__ __launch_bounds__(BLK) void f64_to_f32(const double* __restrict__ in, float* __restrict__ out, long long n) {
  const long long stride = (long long)gridDim.x * BLK;
  for (long long i = (long long)blockIdx.x * BLK + threadIdx.x; i < n; i += stride) out[i] = (float)in[i];
}

// ------------------------------------------------------------------------------------------------
// Hybrid Kaczmarz relaxation (reference native: deps/src/parRelax.h:7-43, applyHybridKaczmarz_FP64_INT64).
// Sub-domains in parallel, the rows listed for a sub-domain strictly in order: for row i
//   inner = (b_i - sum_k a_ik x_k) * invD_i ;  x_k += inner * a_ik  for every k of the row.
// One wavefront per sub-domain.  The lanes fetch the row's entries and products in parallel, lane 0 subtracts the
// products IN STORED ORDER (the reference's sequential loop, l.24-27), every product and sum is rounded separately
// (no FMA contraction: the reference is plain C), so with `domains_per_launch` = all (one wavefront walks every
// sub-domain in order) the result equals the reference binary run with one thread bit for bit.  x is read and written
// through L2 (agent-scope relaxed atomics): a row sees the updates of the rows before it, and - as with the
// reference's OpenMP threads - whatever the neighbouring sub-domains have written so far.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void hybrid_kaczmarz(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const double* __restrict__ val, const unsigned int* __restrict__ arr,
                                                      int num_domains, int domain_length, double* x,
                                                      const double* __restrict__ b, int nrhs, long long n,
                                                      const double* __restrict__ invD, int sequential) {
#pragma clang fp contract(off)   // every product and sum rounded separately, as in the reference's plain C (no FMA)
  const int lane = threadIdx.x;
  const int d0 = sequential ? 0 : blockIdx.x, d1 = sequential ? num_domains : blockIdx.x + 1;
  for (int dom = d0; dom < d1; ++dom) {
    for (int i = 0; i < domain_length; ++i) {
      const unsigned int row1 = arr[(size_t)dom * domain_length + i];
      if (row1 == 0) continue;   // zero padding (wave-uniform)
      const int row = (int)row1 - 1;
      const int s = rowptr[row], e = rowptr[row + 1];
      const double di = invD[row];
      for (int c = 0; c < nrhs; ++c) {
        double* xc = x + (size_t)c * n;
        double inner = b[(size_t)c * n + row];
        for (int k0 = s; k0 < e; k0 += 64) {
          const int k = k0 + lane;
          double prod = 0.0;
          if (k < e) prod = val[k] * __hip_atomic_load(&xc[col[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const int cnt = min(64, e - k0);
          for (int t = 0; t < cnt; ++t) inner = inner - __shfl(prod, t);   // stored order
        }
        inner = inner * di;
        for (int k = s + lane; k < e; k += 64) {
          const double old = __hip_atomic_load(&xc[col[k]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const double upd = inner * val[k];
          __hip_atomic_store(&xc[col[k]], old + upd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // the next row of this wavefront must see these stores: wait until L2 has acknowledged them (loads and stores
        // of one wavefront are not ordered against each other otherwise); the loads above bypass L1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// transposeHierarchy on the device (MGsetup.jl:274-318): CSR transpose of an operator in HBM - count the entries of every
// column, scan (host: n + 1 integers), scatter every entry to its column's segment (atomic cursors: any order), then sort
// each segment by row index (one lane per column, insertion sort: segments are as long as a row of the transposed operator) -
// the result is the stored-order CSR of the transpose, bit for bit what a host transpose gives, whatever the scatter's order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLK) void transpose_count(const int* __restrict__ col, long long nnz, int* __restrict__ cnt) {
  const long long stride = (long long)gridDim.x * BLK;
  for (long long k = (long long)blockIdx.x * BLK + threadIdx.x; k < nnz; k += stride) atomicAdd(cnt + col[k] + 1, 1);
}
__global__ __launch_bounds__(BLK) void transpose_fill(const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const double* __restrict__ val, int n_rows, const int* __restrict__ tptr,
                                                      int* __restrict__ cursor, int* __restrict__ tcol, double* __restrict__ tval) {
  const int i = blockIdx.x * BLK + threadIdx.x;
  if (i >= n_rows) return;
  for (int k = rowptr[i]; k < rowptr[i + 1]; ++k) {
    const int c = col[k];
    const int pos = tptr[c] + atomicAdd(cursor + c, 1);
    tcol[pos] = i;
    tval[pos] = val[k];
  }
}
__global__ __launch_bounds__(BLK) void transpose_sort(const int* __restrict__ tptr, int n_cols, int* __restrict__ tcol,
                                                      double* __restrict__ tval) {
  const int c = blockIdx.x * BLK + threadIdx.x;
  if (c >= n_cols) return;
  const int k0 = tptr[c], k1 = tptr[c + 1];
  for (int k = k0 + 1; k < k1; ++k) {
    const int r = tcol[k];
    const double v = tval[k];
    int j = k - 1;
    while (j >= k0 && tcol[j] > r) {
      tcol[j + 1] = tcol[j];
      tval[j + 1] = tval[j];
      --j;
    }
    tcol[j + 1] = r;
    tval[j + 1] = v;
  }
}
__global__ __launch_bounds__(BLK) void dense_transpose_inplace(double* __restrict__ a, int n) {
  const long long idx = (long long)blockIdx.x * BLK + threadIdx.x;
  const long long i = idx / n, j = idx - i * n;
  if (i < n && j < i) {
    const double t = a[i * n + j];
    a[i * n + j] = a[j * n + i];
    a[j * n + i] = t;
  }
}

}  // namespace mgk
