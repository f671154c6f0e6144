// mg_host.cpp - native host-side (CPU) helpers of the hierarchy SETUP (not of the cycle).
//
// The reference builds its hierarchies on the CPU in serial Julia; BASELINE.json's north_star keeps
// that split ("the Julia host builds the GMG or SA-AMG hierarchy on CPU exactly as the reference
// does").  The aggregation of SA-AMG is an inherently sequential sweep over the rows, far too slow in
// interpreted Python at 256^3, so it lives here.  Arrays use the reference's convention: the CSC
// arrays of the (symmetric) strength matrix S, 1-based Int64.
//
//   mg_sa_aggregate_FP64_INT64  <->  neighborhoodAggregationNew, src/Multigrid/SA-AMG.jl:119-211
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <utility>
#include <vector>

#include <omp.h>

extern "C" {

// aggr_out[k] (1-based root index of the aggregate of node k), length n.
// Returns 0 on success.
// (I: index type; BASE: 1 for the reference's 1-based arrays, 0 for scipy's - CP / RV give the 1-based values either way)
}  // extern "C"
template <class I, int BASE>
static int sa_aggregate_t(long long n, const I* colptr_, const I* rowval_, const double* nzval, long long* aggr) {
  struct Ptr { const I* p; long long operator[](long long k) const { return (long long)p[k] + (1 - BASE); } } colptr{colptr_}, rowval{rowval_};
  const double tau = 3.0;  // l.121
  std::vector<double> aux((std::size_t)n + 1, 0.0);
  std::vector<long long> aux_count((std::size_t)n + 1, 0);
  for (long long k = 1; k <= n; ++k) aggr[k - 1] = 0;
  double avg_sparsity = 0.0;
  for (long long k = 1; k <= n; ++k) avg_sparsity += (double)(colptr[k] - colptr[k - 1]);  // l.128-131
  avg_sparsity /= (double)n;
  for (long long k = 1; k <= n; ++k)  // l.133-137: very dense rows are set aside
    if ((double)(colptr[k] - colptr[k - 1]) > tau * avg_sparsity) aux_count[k] = -1;
  // pass 1 (l.139-158): a node none of whose neighbours is aggregated roots a new aggregate
  for (long long k = 1; k <= n; ++k) {
    if (aux_count[k] == -1) continue;
    bool flag = false;
    for (long long g = colptr[k - 1]; g <= colptr[k] - 1; ++g)
      if (aggr[rowval[g - 1] - 1] != 0) { flag = true; break; }
    if (!flag)
      for (long long g = colptr[k - 1]; g <= colptr[k] - 1; ++g) {
        const long long j = rowval[g - 1];
        if (aux_count[j] != -1) { aggr[j - 1] = k; aux_count[k] += 1; }
      }
  }
  // pass 2 (l.160-178): the set-aside dense rows
  for (long long k = 1; k <= n; ++k) {
    if (aux_count[k] != -1) continue;
    aux_count[k] = 0;
    bool flag = false;
    for (long long g = colptr[k - 1]; g <= colptr[k] - 1; ++g)
      if (aggr[rowval[g - 1] - 1] != 0) { flag = true; break; }
    if (!flag)
      for (long long g = colptr[k - 1]; g <= colptr[k] - 1; ++g) {
        aggr[rowval[g - 1] - 1] = k;
        aux_count[k] += 1;
      }
  }
  // pass 3 (l.180-203): leftovers join the neighbouring ORIGINAL aggregate with the best average
  // connection.  The second loop is nested inside the first and aggr[k] is written inside the outer
  // loop, exactly as the reference's (mis-indented) source does - SURVEY.md note N2.
  for (long long k = 1; k <= n; ++k) {
    double chosen_score = 0.0;
    long long chosen = 0;
    if (aggr[k - 1] == 0) {
      for (long long g = colptr[k - 1]; g <= colptr[k] - 1; ++g) {
        if (aggr[rowval[g - 1] - 1] > 0) {
          const long long a = aggr[rowval[g - 1] - 1];
          aux[a] += nzval[g - 1];
        }
        for (long long g2 = colptr[k - 1]; g2 <= colptr[k] - 1; ++g2) {
          if (aggr[rowval[g2 - 1] - 1] > 0) {
            const long long a = aggr[rowval[g2 - 1] - 1];
            const double score = aux[a] / (double)aux_count[a];
            if (chosen_score < score) {
              chosen_score = score;
              chosen = a;
              aux[a] = 0;
            }
          }
        }
        aggr[k - 1] = -chosen;  // negative: the new member must not attract further nodes
      }
    }
  }
  for (long long k = 1; k <= n; ++k)  // l.205-209
    if (aggr[k - 1] < 0) aggr[k - 1] = -aggr[k - 1];
  return 0;
}
extern "C" {
int mg_sa_aggregate_FP64_INT64(long long n, const long long* colptr, const long long* rowval, const double* nzval, long long* aggr) {
  return sa_aggregate_t<long long, 1>(n, colptr, rowval, nzval, aggr);
}
// the same on scipy's 0-based 32-bit CSR arrays of the (symmetric) strength matrix as they are: no widened, shifted copies
int mg_sa_aggregate_FP64_INT32_BASE0(long long n, const int* ptr, const int* idx, const double* nzval, long long* aggr) {
  return sa_aggregate_t<int, 0>(n, ptr, idx, nzval, aggr);
}

// ------------------------------------------------------------------------------------------------
// C = A*B for CSR operands (0-based int64 pointers/indices, fp64 values), row-parallel Gustavson with a
// per-row open-addressing hash accumulator; output rows have sorted column indices and keep entries
// whose value cancels to zero (Julia's and scipy's sparse products keep them too).
// Used for the Galerkin products Ps[l]*AT*Rs[l] (MGsetup.jl:102, SA-AMG.jl:50), which dominate the
// reference's serial setup.  Two passes: spgemm_count fills C_ptr[i+1] = nnz of row i (caller prefix-sums),
// spgemm_fill writes indices/values.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr long long DENSE_MAX_COLS = 20000000;  // <= 240 MB of stamp+accumulator per thread (hosts of GPU nodes have TBs)
inline std::size_t table_size(long long products, long long ncols) {
  long long need = 2 * std::min(products, ncols) + 2;
  std::size_t t = 16;
  while ((long long)t < need) t <<= 1;
  return t;
}
inline std::size_t hash_slot(long long key, std::size_t mask) {
  return (std::size_t)((unsigned long long)key * 0x9E3779B97F4A7C15ull >> 20) & mask;
}
}  // namespace

extern "C++" {
template <class I>
static int spgemm_count_t(long long n_rows, long long ncols_B, const I* A_ptr, const I* A_idx, const I* B_ptr, const I* B_idx, long long* C_ptr,
                          long long nthreads) {
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  C_ptr[0] = 0;
  if (ncols_B <= DENSE_MAX_COLS) {  // dense stamp array per thread: no per-row clearing
#pragma omp parallel
    {
      std::vector<int> stamp((std::size_t)ncols_B, -1);
#pragma omp for schedule(dynamic, 16)
      for (long long i = 0; i < n_rows; ++i) {
        long long cnt = 0;
        for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) {
          const long long r = A_idx[k];
          for (long long q = B_ptr[r]; q < B_ptr[r + 1]; ++q) {
            const long long c = B_idx[q];
            if (stamp[(std::size_t)c] != (int)i) { stamp[(std::size_t)c] = (int)i; ++cnt; }
          }
        }
        C_ptr[i + 1] = cnt;
      }
    }
    return 0;
  }
#pragma omp parallel
  {
    std::vector<long long> keys;
#pragma omp for schedule(dynamic, 64)
    for (long long i = 0; i < n_rows; ++i) {
      long long products = 0;
      for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) products += B_ptr[A_idx[k] + 1] - B_ptr[A_idx[k]];
      if (products == 0) { C_ptr[i + 1] = 0; continue; }
      const std::size_t T = table_size(products, ncols_B), mask = T - 1;
      if (keys.size() < T) keys.resize(T);
      std::fill(keys.begin(), keys.begin() + T, -1LL);
      long long cnt = 0;
      for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) {
        const long long r = A_idx[k];
        for (long long q = B_ptr[r]; q < B_ptr[r + 1]; ++q) {
          const long long c = B_idx[q];
          std::size_t s = hash_slot(c, mask);
          while (keys[s] != -1 && keys[s] != c) s = (s + 1) & mask;
          if (keys[s] == -1) { keys[s] = c; ++cnt; }
        }
      }
      C_ptr[i + 1] = cnt;
    }
  }
  return 0;
}

template <class I>
static int spgemm_fill_t(long long n_rows, long long ncols_B, const I* A_ptr, const I* A_idx, const double* A_val, const I* B_ptr, const I* B_idx,
                         const double* B_val, const long long* C_ptr, I* C_idx, double* C_val, long long nthreads) {
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  if (ncols_B <= DENSE_MAX_COLS) {
#pragma omp parallel
    {
      std::vector<int> stamp((std::size_t)ncols_B, -1);
      std::vector<double> acc((std::size_t)ncols_B, 0.0);
      std::vector<long long> touched;
#pragma omp for schedule(dynamic, 16)
      for (long long i = 0; i < n_rows; ++i) {
        const long long out0 = C_ptr[i], cnt = C_ptr[i + 1] - C_ptr[i];
        if (cnt == 0) continue;
        touched.clear();
        for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) {
          const long long r = A_idx[k];
          const double a = A_val[k];
          for (long long q = B_ptr[r]; q < B_ptr[r + 1]; ++q) {
            const std::size_t c = (std::size_t)B_idx[q];
            if (stamp[c] != (int)i) { stamp[c] = (int)i; acc[c] = a * B_val[q]; touched.push_back((long long)c); }
            else acc[c] += a * B_val[q];
          }
        }
        std::sort(touched.begin(), touched.end());
        for (long long j = 0; j < cnt; ++j) {
          C_idx[out0 + j] = (I)touched[(std::size_t)j];
          C_val[out0 + j] = acc[(std::size_t)touched[(std::size_t)j]];
        }
      }
    }
    return 0;
  }
#pragma omp parallel
  {
    std::vector<long long> keys;
    std::vector<double> vals;
    std::vector<std::pair<long long, double>> row;
#pragma omp for schedule(dynamic, 64)
    for (long long i = 0; i < n_rows; ++i) {
      const long long out0 = C_ptr[i], cnt = C_ptr[i + 1] - C_ptr[i];
      if (cnt == 0) continue;
      long long products = 0;
      for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) products += B_ptr[A_idx[k] + 1] - B_ptr[A_idx[k]];
      const std::size_t T = table_size(products, ncols_B), mask = T - 1;
      if (keys.size() < T) { keys.resize(T); vals.resize(T); }
      std::fill(keys.begin(), keys.begin() + T, -1LL);
      for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) {  // accumulate in the order a serial Gustavson would
        const long long r = A_idx[k];
        const double a = A_val[k];
        for (long long q = B_ptr[r]; q < B_ptr[r + 1]; ++q) {
          const long long c = B_idx[q];
          std::size_t s = hash_slot(c, mask);
          while (keys[s] != -1 && keys[s] != c) s = (s + 1) & mask;
          if (keys[s] == -1) { keys[s] = c; vals[s] = a * B_val[q]; }
          else vals[s] += a * B_val[q];
        }
      }
      row.clear();
      for (std::size_t s = 0; s < T; ++s)
        if (keys[s] != -1) row.emplace_back(keys[s], vals[s]);
      std::sort(row.begin(), row.end(), [](const std::pair<long long, double>& x, const std::pair<long long, double>& y) { return x.first < y.first; });
      for (long long j = 0; j < cnt; ++j) { C_idx[out0 + j] = (I)row[(std::size_t)j].first; C_val[out0 + j] = row[(std::size_t)j].second; }
    }
  }
  return 0;
}

}  // extern "C++"

int mg_spgemm_count_INT64(long long n_rows, long long ncols_B, const long long* A_ptr, const long long* A_idx,
                          const long long* B_ptr, const long long* B_idx, long long* C_ptr, long long nthreads) {
  return spgemm_count_t<long long>(n_rows, ncols_B, A_ptr, A_idx, B_ptr, B_idx, C_ptr, nthreads);
}
int mg_spgemm_fill_FP64_INT64(long long n_rows, long long ncols_B, const long long* A_ptr, const long long* A_idx,
                              const double* A_val, const long long* B_ptr, const long long* B_idx,
                              const double* B_val, const long long* C_ptr, long long* C_idx, double* C_val,
                              long long nthreads) {
  return spgemm_fill_t<long long>(n_rows, ncols_B, A_ptr, A_idx, A_val, B_ptr, B_idx, B_val, C_ptr, C_idx, C_val, nthreads);
}
// the same with 32-bit indices in and out (scipy's default below 2^31 entries: no widening copies of the operands - at C3's size
// those copies were a quarter of the host setup); C_ptr stays 64-bit
int mg_spgemm_count_INT32(long long n_rows, long long ncols_B, const int* A_ptr, const int* A_idx, const int* B_ptr, const int* B_idx,
                          long long* C_ptr, long long nthreads) {
  return spgemm_count_t<int>(n_rows, ncols_B, A_ptr, A_idx, B_ptr, B_idx, C_ptr, nthreads);
}
int mg_spgemm_fill_FP64_INT32(long long n_rows, long long ncols_B, const int* A_ptr, const int* A_idx, const double* A_val, const int* B_ptr,
                              const int* B_idx, const double* B_val, const long long* C_ptr, int* C_idx, double* C_val, long long nthreads) {
  return spgemm_fill_t<int>(n_rows, ncols_B, A_ptr, A_idx, A_val, B_ptr, B_idx, B_val, C_ptr, C_idx, C_val, nthreads);
}

// ------------------------------------------------------------------------------------------------
// SpGEMM in a symbolic and a numeric phase (round 5): the Galerkin products of the SA-AMG middle levels are 10^10 - 10^11 products
// each, and the count + fill pair above walks them twice with a stamp test, a push and a sort in the inner loops.  Here the
// SYMBOLIC phase finds every output row's sorted pattern once - a per-thread bitmap (set a bit per product, then scan the words
// between the smallest and the largest column: sorted for free) where a row has many products, a stamped list + sort where it has
// few - and keeps it in per-thread chunks; the NUMERIC phase zeroes the accumulator at the row's pattern, adds the products with
// no test at all (same order as a serial Gustavson: same bits) and gathers.  int32 operands, at most DENSE_MAX_COLS columns.
// ------------------------------------------------------------------------------------------------
namespace {
struct SpgemmPlan {
  long long n_rows = 0, ncols = 0;
  std::vector<std::vector<int>> chunk;          // per thread: the patterns of the rows it took, back to back
  std::vector<int> owner;                        // [n_rows] thread that holds the row
  std::vector<long long> where;                  // [n_rows] offset in that thread's chunk
};
}  // namespace

void* mg_spgemm_symbolic_INT32(long long n_rows, long long ncols_B, const int* A_ptr, const int* A_idx, const int* B_ptr, const int* B_idx,
                               long long* C_ptr, long long nthreads) {
  if (ncols_B > DENSE_MAX_COLS || ncols_B <= 0 || n_rows < 0) return nullptr;
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  SpgemmPlan* plan = new SpgemmPlan();
  plan->n_rows = n_rows;
  plan->ncols = ncols_B;
  plan->owner.assign((std::size_t)n_rows, 0);
  plan->where.assign((std::size_t)n_rows, 0);
  plan->chunk.resize((std::size_t)omp_get_max_threads());
  const long long nwords = (ncols_B + 63) / 64;
  C_ptr[0] = 0;
  bool failed = false;
#pragma omp parallel
  {
    const int t = omp_get_thread_num();
    std::vector<int>& out = plan->chunk[(std::size_t)t];
    std::vector<unsigned long long> bits;
    std::vector<int> stamp;
    // (an exception must not leave the worksharing loop - OpenMP has no unwinding across it, the throwing thread would skip the
    // loop's barrier: every iteration catches its own, raises the shared flag, and the remaining iterations do nothing)
#pragma omp for schedule(dynamic, 16)
    for (long long i = 0; i < n_rows; ++i) {
      bool skip;
#pragma omp atomic read
      skip = failed;
      if (skip) continue;
      try {
        long long products = 0;
        for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) products += B_ptr[A_idx[k] + 1] - B_ptr[A_idx[k]];
        plan->owner[(std::size_t)i] = t;
        plan->where[(std::size_t)i] = (long long)out.size();
        if (products == 0) { C_ptr[i + 1] = 0; continue; }
        const std::size_t at = out.size();
        if (products * 4 >= nwords) {            // bitmap: the scan of at most nwords words is no more than the products it replaces a sort of
          if (bits.empty()) bits.assign((std::size_t)nwords, 0ull);
          int cmin = 0x7fffffff, cmax = -1;
          for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) {
            const int r = A_idx[k];
            const int q0 = B_ptr[r], q1 = B_ptr[r + 1];
            if (q1 > q0) {                       // (rows of B are sorted: their ends bound the columns)
              cmin = std::min(cmin, B_idx[q0]);
              cmax = std::max(cmax, B_idx[q1 - 1]);
            }
            // (a sorted row of B fills one word after the other: the word is built in a register and written once)
            unsigned wcur = q1 > q0 ? (unsigned)B_idx[q0] >> 6 : 0u;
            unsigned long long m = 0ull;
            for (int q = q0; q < q1; ++q) {
              const unsigned c = (unsigned)B_idx[q], wc = c >> 6;
              if (wc != wcur) { bits[wcur] |= m; m = 0ull; wcur = wc; }
              m |= 1ull << (c & 63u);
            }
            if (m) bits[wcur] |= m;
          }
          for (long long w = cmin >> 6; w <= (cmax >> 6); ++w) {
            unsigned long long v = bits[(std::size_t)w];
            if (!v) continue;
            bits[(std::size_t)w] = 0ull;
            while (v) {
              out.push_back((int)(w * 64 + __builtin_ctzll(v)));
              v &= v - 1;
            }
          }
        } else {
          if (stamp.empty()) stamp.assign((std::size_t)ncols_B, -1);
          for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) {
            const int r = A_idx[k];
            for (int q = B_ptr[r]; q < B_ptr[r + 1]; ++q) {
              const int c = B_idx[q];
              if (stamp[(std::size_t)c] != (int)i) { stamp[(std::size_t)c] = (int)i; out.push_back(c); }
            }
          }
          std::sort(out.begin() + (std::ptrdiff_t)at, out.end());
        }
        C_ptr[i + 1] = (long long)(out.size() - at);
      } catch (...) {
#pragma omp atomic write
        failed = true;
      }
    }
  }
  if (failed) { delete plan; return nullptr; }
  for (long long i = 0; i < n_rows; ++i) C_ptr[i + 1] += C_ptr[i];
  return plan;
}

void mg_spgemm_plan_free(void* plan) { delete static_cast<SpgemmPlan*>(plan); }

// C_ptr: the prefix sums the symbolic phase returned; C_idx / C_val: C_ptr[n_rows] entries.  Frees the plan.
int mg_spgemm_numeric_FP64_INT32(void* plan_, const int* A_ptr, const int* A_idx, const double* A_val, const int* B_ptr, const int* B_idx,
                                 const double* B_val, const long long* C_ptr, int* C_idx, double* C_val, long long nthreads) {
  SpgemmPlan* plan = static_cast<SpgemmPlan*>(plan_);
  if (!plan) return 1;
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  const long long n_rows = plan->n_rows;
  bool failed = false;
#pragma omp parallel
  {
    std::vector<double> acc;
    try {
      acc.assign((std::size_t)plan->ncols, 0.0);
    } catch (...) {          // (out of memory for a thread's accumulator: reported after the region, never std::terminate)
#pragma omp atomic write
      failed = true;
    }
#pragma omp barrier
#pragma omp for schedule(dynamic, 16)
    for (long long i = 0; i < n_rows; ++i) {
      bool skip;
#pragma omp atomic read
      skip = failed;
      if (skip) continue;
      const long long out0 = C_ptr[i], cnt = C_ptr[i + 1] - C_ptr[i];
      if (cnt == 0) continue;
      const int* pat = plan->chunk[(std::size_t)plan->owner[(std::size_t)i]].data() + plan->where[(std::size_t)i];
      int* ci = C_idx + out0;
      for (long long j = 0; j < cnt; ++j) { ci[j] = pat[j]; acc[(std::size_t)pat[j]] = 0.0; }
      for (long long k = A_ptr[i]; k < A_ptr[i + 1]; ++k) {
        const int r = A_idx[k];
        const double a = A_val[k];
        const int q1 = B_ptr[r + 1];
        for (int q = B_ptr[r]; q < q1; ++q) acc[(std::size_t)B_idx[q]] += a * B_val[q];
      }
      double* cv = C_val + out0;
      for (long long j = 0; j < cnt; ++j) cv[j] = acc[(std::size_t)ci[j]];
    }
  }
  delete plan;
  return failed ? 2 : 0;
}

// ------------------------------------------------------------------------------------------------
// Transposes of the setup (SA-AMG.jl:47 `R = P'`, l.115 `S + S'`): scipy's csc_tocsr is a serial scatter - 30 s of a 150 s setup
// at 96^3 cells.  mg_csr_transpose: counting sort by column with per-column cursors, then every output row sorted by its
// (distinct) column = input row index: the stored-order CSR of the transpose, whatever the thread count.
// mg_csr_add_transpose_symm: out = S + S' for a STRUCTURALLY symmetric S with sorted rows (the strength matrix: A symmetric):
// entry (i, j) looks (j, i) up by binary search in row j; returns 1 (nothing written is valid) if the pattern is not symmetric.
// ------------------------------------------------------------------------------------------------
int mg_csr_transpose_FP64_INT32(long long n_rows, long long n_cols, const int* ptr, const int* idx, const double* val, int* t_ptr, int* t_idx,
                                double* t_val, long long nthreads) {
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  const long long nnz = ptr[n_rows];
  std::vector<int> cnt((std::size_t)n_cols + 1, 0);
#pragma omp parallel for schedule(static)
  for (long long k = 0; k < nnz; ++k) {
#pragma omp atomic
    ++cnt[(std::size_t)idx[k] + 1];
  }
  t_ptr[0] = 0;
  for (long long j = 0; j < n_cols; ++j) t_ptr[j + 1] = t_ptr[j] + cnt[(std::size_t)j + 1];
  std::vector<int> cur((std::size_t)n_cols, 0);
#pragma omp parallel for schedule(dynamic, 1024)
  for (long long i = 0; i < n_rows; ++i)
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) {
      int pos;
#pragma omp atomic capture
      pos = cur[(std::size_t)idx[k]]++;
      const long long q = (long long)t_ptr[idx[k]] + pos;
      t_idx[q] = (int)i;
      t_val[q] = val[k];
    }
#pragma omp parallel
  {
    std::vector<std::pair<int, double>> row;
#pragma omp for schedule(dynamic, 256)
    for (long long j = 0; j < n_cols; ++j) {
      const int a = t_ptr[j], b = t_ptr[j + 1];
      bool sorted = true;
      for (int k = a + 1; k < b && sorted; ++k) sorted = t_idx[k - 1] < t_idx[k];
      if (sorted) continue;
      row.clear();
      for (int k = a; k < b; ++k) row.emplace_back(t_idx[k], t_val[k]);
      std::sort(row.begin(), row.end(), [](const std::pair<int, double>& x, const std::pair<int, double>& y) { return x.first < y.first; });
      for (int k = a; k < b; ++k) { t_idx[k] = row[(std::size_t)(k - a)].first; t_val[k] = row[(std::size_t)(k - a)].second; }
    }
  }
  return 0;
}
int mg_csr_add_transpose_symm_FP64_INT32(long long n, const int* ptr, const int* idx, const double* val, double* out, long long nthreads) {
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 256) reduction(| : bad)
  for (long long i = 0; i < n; ++i)
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) {
      const int j = idx[k];
      const int* lo = idx + ptr[j];
      const int* hi = idx + ptr[j + 1];
      const int* f = std::lower_bound(lo, hi, (int)i);
      if (f == hi || *f != (int)i) { bad = 1; out[k] = val[k]; continue; }
      out[k] = val[k] + val[f - idx];
    }
  return bad;
}

// Entries of a CSR matrix that are not zero, rows in place (scipy's eliminate_zeros into NEW arrays, thread-parallel):
// new_ptr[n + 1]; out_idx / out_val sized for nnz entries, the first new_ptr[n] are written.
int mg_csr_compact_nonzero_FP64_INT32(long long n, const int* ptr, const int* idx, const double* val, int* new_ptr, int* out_idx, double* out_val,
                                      long long nthreads) {
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  new_ptr[0] = 0;
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < n; ++i) {
    int c = 0;
    for (int k = ptr[i]; k < ptr[i + 1]; ++k) c += val[k] != 0.0;
    new_ptr[i + 1] = c;
  }
  for (long long i = 0; i < n; ++i) new_ptr[i + 1] += new_ptr[i];
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < n; ++i) {
    int q = new_ptr[i];
    for (int k = ptr[i]; k < ptr[i + 1]; ++k)
      if (val[k] != 0.0) { out_idx[q] = idx[k]; out_val[q] = val[k]; ++q; }
  }
  return 0;
}

// s[j] = sum_i A[i,j]^2 (the column sums of squares getSPAIprec divides the diagonal by, MGsetup.jl:359-362).  Every thread owns a
// contiguous range of COLUMNS and walks all rows (sorted: a binary search finds its part of a row), so every s[j] is summed in row
// order - the bits of the serial scatter it replaces, whatever the thread count.
int mg_csr_colsumsq_FP64_INT32(long long n_rows, long long n_cols, const int* ptr, const int* idx, const double* val, double* out, long long nthreads) {
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
#pragma omp parallel
  {
    const int t = omp_get_thread_num(), nt = omp_get_num_threads();
    const int c0 = (int)(n_cols * t / nt), c1 = (int)(n_cols * (t + 1) / nt);
    for (int j = c0; j < c1; ++j) out[j] = 0.0;
    if (c1 > c0)
      for (long long i = 0; i < n_rows; ++i) {
        const int* lo = idx + ptr[i];
        const int* hi = idx + ptr[i + 1];
        for (const int* q = std::lower_bound(lo, hi, c0); q < hi && *q < c1; ++q) {
          const double v = val[q - idx];
          out[*q] += v * v;
        }
      }
  }
  return 0;
}

// getStrengthMatrix before the symmetrisation (SA-AMG.jl:88-113), row-parallel: out = -val scaled per row by 1 / max(mm, largest
// entry of the row of -A), mm = 1e-16 * (largest entry of -A); diagonal := 1; entries < theta := 0.  The same double operations
// in the same order as the vectorised host code it replaces (negation, one reciprocal per row, one product per entry).
int mg_sa_strength_FP64_INT32(long long n, const int* ptr, const int* idx, const double* val, double theta, double* out, long long nthreads) {
  if (nthreads > 0) omp_set_num_threads((int)nthreads);
  const long long nnz = ptr[n];
  double gmax = 0.0;
  bool any = false;
#pragma omp parallel
  {
    double m = 0.0;
    bool have = false;
#pragma omp for schedule(static) nowait
    for (long long k = 0; k < nnz; ++k) {
      const double v = -val[k];
      if (!have || v > m) { m = v; have = true; }
    }
#pragma omp critical
    if (have && (!any || m > gmax)) { gmax = m; any = true; }
  }
  const double mm = 1e-16 * gmax;
#pragma omp parallel for schedule(dynamic, 1024)
  for (long long i = 0; i < n; ++i) {
    const int a = ptr[i], b = ptr[i + 1];
    if (b == a) continue;
    double rmax = -val[a];
    for (int k = a + 1; k < b; ++k) rmax = std::max(rmax, -val[k]);
    rmax = std::max(mm, rmax);
    const double sc = 1.0 / rmax;
    for (int k = a; k < b; ++k) {
      double v = (-val[k]) * sc;
      if (idx[k] == (int)i) v = 1.0;
      if (v < theta) v = 0.0;
      out[k] = v;
    }
  }
  return 0;
}

long long mg_host_max_threads(void) { return (long long)omp_get_max_threads(); }

}  // extern "C"
