// mg_host.cpp - native host-side (CPU) helpers of the hierarchy SETUP (not of the cycle).
//
// The reference builds its hierarchies on the CPU in serial Julia; BASELINE.json's north_star keeps
// that split ("the Julia host builds the GMG or SA-AMG hierarchy on CPU exactly as the reference
// does").  The aggregation of SA-AMG is an inherently sequential sweep over the rows, far too slow in
// interpreted Python at 256^3, so it lives here.  Arrays use the reference's convention: the CSC
// arrays of the (symmetric) strength matrix S, 1-based Int64.
//
//   mg_sa_aggregate_FP64_INT64  <->  neighborhoodAggregationNew, src/Multigrid/SA-AMG.jl:119-211
#include <cstddef>
#include <cstdint>
#include <vector>

extern "C" {

// aggr_out[k] (1-based root index of the aggregate of node k), length n.
// Returns 0 on success.
int mg_sa_aggregate_FP64_INT64(long long n, const long long* colptr, const long long* rowval,
                               const double* nzval, long long* aggr) {
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

}  // extern "C"
