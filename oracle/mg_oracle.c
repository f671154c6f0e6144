/*
 * ORACLE - TEST INFRASTRUCTURE ONLY (see oracle/mg_oracle.py header; PARITY UNPINNED BY VALUE).
 *
 * Plain-C + OpenMP restatement of the reference's CPU cycle, written to mirror the path that
 * "Julia + ParSpMatVec OMP" executes: row-parallel CSR SpMV with Int64 1-based indices, one pass over
 * A per RHS column, sequential in-row accumulation, and the UNFUSED operation sequence of
 * src/Multigrid/MGcycle.jl:1-136 / SolveFuncs.jl:3-39 (separate axpy, separate x += d.*r, norm).
 * Used by tests as a second checker and by bench.py as the timed cpu_baseline ("port").
 * Compiled with the reference's own flags (deps/build.jl:30): gcc -O3 -fPIC -fopenmp -shared.
 *
 * Arrays are passed exactly as Julia holds the transposed CSC (= CSR of A): colptr/rowval 1-based
 * Int64, nzval Float64; dense blocks column-major n x nrhs.
 */
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef long long i64;

typedef struct {
  i64 n;  /* rows of A on this level */
  i64 nc; /* rows of the next coarser level (0 on the coarsest) */
  const i64 *A_colptr, *A_rowval;
  const double* A_nzval;
  const i64 *P_colptr, *P_rowval; /* CSR of P: n rows */
  const double* P_nzval;
  const i64 *R_colptr, *R_rowval; /* CSR of R: nc rows */
  const double* R_nzval;
  const double* d; /* relaxPrecs[level] */
  i64 npre, npost;
  double *b, *r, *x; /* CYCLEmem, n x nrhs each (b unused on level 1) */
} oracle_level;

/* target = beta*target + alpha*A*x  (SpMatMul.jl:4-13; ParSpMatVec.Ac_mul_B! semantics) */
void oracle_spmatmul_FP64_INT64(double alpha, const i64* colptr, const i64* rowval, const double* nzval,
                                i64 n_rows, i64 n_cols, const double* x, double beta, double* target,
                                i64 nrhs, i64 numCores) {
  omp_set_num_threads((int)numCores);
  for (i64 c = 0; c < nrhs; ++c) { /* outer loop over RHS columns: A is re-streamed per column */
    const double* xc = x + c * n_cols;
    double* tc = target + c * n_rows;
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n_rows; ++i) {
      double s = 0.0;
      for (i64 k = colptr[i] - 1; k < colptr[i + 1] - 1; ++k) s += nzval[k] * xc[rowval[k] - 1];
      tc[i] = (beta == 0.0) ? alpha * s : beta * tc[i] + alpha * s;
    }
  }
}

/* target += alpha*x  (SpMatMul.jl:29-37) */
static void add_vectors(double alpha, const double* x, double* target, i64 len) {
#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < len; ++i) target[i] += alpha * x[i];
}

static double norm2(const double* x, i64 len) {
  double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
  for (i64 i = 0; i < len; ++i) s += x[i] * x[i];
  return sqrt(s);
}

/* x .+= d.*r  (MGcycle.jl:129,134) */
static void dr_update(const double* d, const double* r, double* x, i64 n, i64 nrhs) {
  for (i64 c = 0; c < nrhs; ++c) {
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) x[c * n + i] += d[i] * r[c * n + i];
  }
}

static void spmm_A(double alpha, const oracle_level* L, const double* x, double beta, double* t, i64 nrhs, i64 nc) {
  oracle_spmatmul_FP64_INT64(alpha, L->A_colptr, L->A_rowval, L->A_nzval, L->n, L->n, x, beta, t, nrhs, nc);
}

/* relax (MGcycle.jl:122-136) */
static void relax(const oracle_level* L, double* r, double* x, const double* b, i64 numit, i64 nrhs, i64 numCores) {
  const i64 len = L->n * nrhs;
  for (i64 i = 1; i < numit; ++i) {
    dr_update(L->d, r, x, L->n, nrhs);
    spmm_A(-1.0, L, x, 0.0, r, nrhs, numCores);
    add_vectors(1.0, b, r, len);
  }
  dr_update(L->d, r, x, L->n, nrhs);
}

/* solveCoarsest default branch (MGcycle.jl:177): here with the explicit inverse, column-major nc x nc */
static void solve_coarsest(const double* Ainv, i64 n, const double* b, double* x, i64 nrhs) {
  for (i64 c = 0; c < nrhs; ++c) {
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) {
      double s = 0.0;
      for (i64 j = 0; j < n; ++j) s += Ainv[j * n + i] * b[c * n + j];
      x[c * n + i] = s;
    }
  }
}

/* recursiveCycle (MGcycle.jl:1-118); level is 1-based */
void oracle_recursive_cycle(const oracle_level* lev, i64 nlevels, const double* Ainv, i64 level, const double* b,
                            double* x, i64 nrhs, i64 cycleType, i64 numCores) {
  omp_set_num_threads((int)numCores);
  const oracle_level* L = &lev[level - 1];
  const i64 len = L->n * nrhs;
  if (level == nlevels) {
    memcpy(L->r, b, sizeof(double) * len);
    solve_coarsest(Ainv, L->n, L->r, x, nrhs);
    return;
  }
  double* r = L->r;
  memcpy(r, b, sizeof(double) * len);                                   /* l.26-28 */
  if (norm2(x, len) > 0.0) spmm_A(-1.0, L, x, 1.0, r, nrhs, numCores);  /* l.29-31 */
  relax(L, r, x, b, L->npre, nrhs, numCores);                           /* l.54 */
  spmm_A(-1.0, L, x, 0.0, r, nrhs, numCores);                           /* l.58 */
  add_vectors(1.0, b, r, len);                                          /* l.60 */
  const oracle_level* C = &lev[level];
  double* xc = C->x;
  memset(xc, 0, sizeof(double) * C->n * nrhs);                          /* l.63-64 */
  double* bc = C->b;
  oracle_spmatmul_FP64_INT64(1.0, L->R_colptr, L->R_rowval, L->R_nzval, L->nc, L->n, r, 0.0, bc, nrhs, numCores); /* l.66 */
  if (level == nlevels - 1) {
    solve_coarsest(Ainv, C->n, bc, xc, nrhs);                           /* l.67-69 */
  } else {
    oracle_recursive_cycle(lev, nlevels, Ainv, level + 1, bc, xc, nrhs, cycleType, numCores);
    if (cycleType == 'W') oracle_recursive_cycle(lev, nlevels, Ainv, level + 1, bc, xc, nrhs, 'W', numCores);
    else if (cycleType == 'F') oracle_recursive_cycle(lev, nlevels, Ainv, level + 1, bc, xc, nrhs, 'V', numCores);
  }
  oracle_spmatmul_FP64_INT64(1.0, L->P_colptr, L->P_rowval, L->P_nzval, L->n, L->nc, xc, 1.0, x, nrhs, numCores); /* l.90 */
  memcpy(r, b, sizeof(double) * len);                                   /* l.92 */
  spmm_A(-1.0, L, x, 1.0, r, nrhs, numCores);                           /* l.93 */
  relax(L, r, x, b, L->npost, nrhs, numCores);                          /* l.102 */
}

/* solveMG (SolveFuncs.jl:3-39); returns the iteration count, resvec[0..iter] */
i64 oracle_solveMG(const oracle_level* lev, i64 nlevels, const double* Ainv, const double* b, double* x, i64 nrhs,
                   double tol, i64 maxIter, i64 cycleType, i64 numCores, double* resvec) {
  omp_set_num_threads((int)numCores);
  const oracle_level* L = &lev[0];
  const i64 len = L->n * nrhs;
  double* r = L->r;
  memcpy(r, b, sizeof(double) * len);
  double res;
  if (norm2(x, len) == 0.0) {
    res = norm2(b, len);
  } else {
    spmm_A(-1.0, L, x, 1.0, r, nrhs, numCores);
    res = norm2(r, len);
  }
  const double res_init = res;
  if (resvec) resvec[0] = res_init;
  i64 iter = 0;
  for (i64 count = 1; count <= maxIter; ++count) {
    oracle_recursive_cycle(lev, nlevels, Ainv, 1, b, x, nrhs, cycleType, numCores);
    spmm_A(-1.0, L, x, 0.0, r, nrhs, numCores);
    add_vectors(1.0, b, r, len);
    ++iter;
    res = norm2(r, len);
    if (resvec) resvec[iter] = res;
    if (res / res_init < tol) break;
  }
  return iter;
}

i64 oracle_max_threads(void) { return (i64)omp_get_max_threads(); }

/* ---- NUMA placement for the timed baseline (bench.py cpu_baseline) ---------------------------------------------
 * numpy allocates and fills every array from ONE thread, so all pages of the hierarchy land on one NUMA node and the
 * row-parallel loops above pull them across the socket interconnect.  These helpers clone an array into fresh
 * (untouched) memory with the SAME static row partition the kernels use, so that each page is first touched - and
 * therefore placed - by the thread that will stream it. */
void* oracle_numa_clone_csr_FP64_INT64(i64 n_rows, const i64* colptr, const i64* rowval, const double* nzval,
                                       i64** out_colptr, i64** out_rowval, double** out_nzval, i64 numCores) {
  omp_set_num_threads((int)numCores);
  const i64 nnz = colptr[n_rows] - 1;
  i64* cp = (i64*)malloc(sizeof(i64) * (size_t)(n_rows + 1));
  i64* rv = (i64*)malloc(sizeof(i64) * (size_t)(nnz > 0 ? nnz : 1));
  double* nz = (double*)malloc(sizeof(double) * (size_t)(nnz > 0 ? nnz : 1));
  if (!cp || !rv || !nz) { free(cp); free(rv); free(nz); return NULL; }
#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < n_rows; ++i) {
    cp[i] = colptr[i];
    for (i64 k = colptr[i] - 1; k < colptr[i + 1] - 1; ++k) {
      rv[k] = rowval[k];
      nz[k] = nzval[k];
    }
  }
  cp[n_rows] = colptr[n_rows];
  *out_colptr = cp;
  *out_rowval = rv;
  *out_nzval = nz;
  return cp;
}
double* oracle_numa_clone_vec(const double* src, i64 n, i64 nrhs, i64 numCores) {
  omp_set_num_threads((int)numCores);
  double* v = (double*)malloc(sizeof(double) * (size_t)(n * nrhs > 0 ? n * nrhs : 1));
  if (!v) return NULL;
  for (i64 c = 0; c < nrhs; ++c) {
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) v[c * n + i] = src ? src[c * n + i] : 0.0;
  }
  return v;
}
void oracle_copy_vec(const double* src, double* dst, i64 n, i64 nrhs, i64 numCores) {
  omp_set_num_threads((int)numCores);
  for (i64 c = 0; c < nrhs; ++c) {
#pragma omp parallel for schedule(static)
    for (i64 i = 0; i < n; ++i) dst[c * n + i] = src[c * n + i];
  }
}
void oracle_free(void* p) { free(p); }
