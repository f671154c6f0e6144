// mg_small.hpp - kernels of the SMALL grid levels (a few thousand to a few hundred thousand rows), included by mgvcycle.hip.
//
// Below ~300 000 rows a launch of the cycle is latency, not bandwidth: 22 launches of levels 3-6 of C2 took 133 us of a
// 490 us step for 1.8 % of the rows - 6 us each where an empty launch inside a HIP graph costs 2.3 us (profiles/
// r03_gridbar_probe.txt).  The rest was each kernel's own dependent chain: row pointer -> column indices / values -> gather
// (streaming forms), or dictionary -> LDS -> barrier -> class id -> gather (row-class forms): three to four trips to L2 / HBM
// one behind the other.  These kernels have ONE: a lane owns a row of a 27-point grid operator (the Galerkin coarse levels;
// MGsetup.jl:102), derives the row's position class - first / interior / last node per direction, 27 codes - from its index
// by arithmetic, and issues the 27 record values (code * 28 doubles: L2-resident, the lanes of a wavefront mostly share
// them) and the 27 gathers of x at once; nothing waits for a table.  Same products in the same (dz, dy, dx) = CSR order as
// csr_rowclass_march27_spmv / the plane tiles (entries a row does not have carry the value 0 and multiply a valid, finite
// neighbour): same bits.  Restriction likewise over the coarse grid (MGcycle.jl:66), prolongation by pure arithmetic - the
// full-weighting weights are 1 / 1/2 per direction (GeometricTransferOperators.jl:27-29) - after the host has verified that
// the stored operator IS that tensor product (MGcycle.jl:90).  No LDS, no barrier, 64 or 256 lanes per workgroup.
#pragma once

namespace mgk {

struct Small27Dev {
  const double* rec;   // [27 codes][28]: v[(dz+1)*9 + (dy+1)*3 + (dx+1)] (0 where the rows of that code have no entry), [27] = relaxPrec
  int n1, n2, n3, P, n;
};
struct SmallRDev {
  const double* rec;   // [27 codes of the COARSE position][27]: weights of the fine nodes (2i+dx, 2j+dy, 2k+dz)
  int nc1, nc2, nc3, Pc, nc;
  int nf1, nf2, nf3, Pf, nf;
};
struct SmallPDev {
  int nf1, nf2, nf3, Pf, nf;
  int nc1, nc2, nc3, Pc, nc;
};

__device__ __forceinline__ int small_code1(int i, int n) { return i == 0 ? 0 : (i == n - 1 ? 2 : 1); }

// y = b - A x (RESID), y = x + d.*(b - A x) (SMOOTH), y = A x (AXPBY with alpha = 1, beta = 0)
template <int MODE>
__global__ __launch_bounds__(256) void grid27_small_spmv(Small27Dev T, const double* __restrict__ x, const double* __restrict__ b,
                                                         double* __restrict__ y) {
  const int row = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (row >= T.n) return;
  const int z = row / T.P, rem = row - z * T.P, yy = rem / T.n1, xx = rem - yy * T.n1;
  const int code = (small_code1(z, T.n3) * 3 + small_code1(yy, T.n2)) * 3 + small_code1(xx, T.n1);
  const double* __restrict__ rec = T.rec + code * 28;
  double rv[27], xv[27];
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int s = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
        int j = row + dz * T.P + dy * T.n1 + dx;
        j = j < 0 ? 0 : (j >= T.n ? T.n - 1 : j);     // (an entry the row does not have: value 0 times some valid, finite x)
        rv[s] = rec[s];
        xv[s] = x[j];
      }
  const double pb = MODE == AXPBY ? 0.0 : b[row];
  const double pd = MODE == SMOOTH ? rec[27] : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < 27; ++s) acc = acc + rv[s] * xv[s];
  double out;
  if (MODE == AXPBY) out = acc;
  else if (MODE == RESID) out = pb - acc;
  else out = xv[13] + pd * (pb - acc);
  y[row] = out;
}

// bc = R r over the coarse grid [, y2 = d2 .* bc: the coarse level's first update from x = 0]
__global__ __launch_bounds__(256) void grid27_small_restrict(SmallRDev T, const double* __restrict__ r, double* __restrict__ bc,
                                                             const double* __restrict__ d2, double* __restrict__ y2) {
  const int row = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (row >= T.nc) return;
  const int k = row / T.Pc, rem = row - k * T.Pc, j = rem / T.nc1, i = rem - j * T.nc1;
  const int code = (small_code1(k, T.nc3) * 3 + small_code1(j, T.nc2)) * 3 + small_code1(i, T.nc1);
  const double* __restrict__ rec = T.rec + code * 27;
  const int centre = (2 * k) * T.Pf + (2 * j) * T.nf1 + 2 * i;
  double rv[27], xv[27];
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int s = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
        int f = centre + dz * T.Pf + dy * T.nf1 + dx;
        f = f < 0 ? 0 : (f >= T.nf ? T.nf - 1 : f);
        rv[s] = rec[s];
        xv[s] = r[f];
      }
  const double pd = d2 ? d2[row] : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < 27; ++s) acc = acc + rv[s] * xv[s];
  bc[row] = acc;
  if (y2) y2[row] = pd * acc;
}

// x += P xc, P the full-weighting interpolation of a vertex-centred grid pair (fine = 2*coarse - 1 nodes per direction): weight 1
// from the coincident coarse node, 1/2 from each of the two neighbours of an odd coordinate; the (up to 8) products are added in
// ascending coarse column order, as the CSR row holds them.
__global__ __launch_bounds__(256) void grid_small_prolong(SmallPDev T, const double* __restrict__ xc, double* __restrict__ x) {
  const int row = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (row >= T.nf) return;
  const int z = row / T.Pf, rem = row - z * T.Pf, yy = rem / T.nf1, xx = rem - yy * T.nf1;
  const int oz = z & 1, oy = yy & 1, ox = xx & 1;
  const int c0 = (z >> 1) * T.Pc + (yy >> 1) * T.nc1 + (xx >> 1);
  const double wz = oz ? 0.5 : 1.0, wy = oy ? 0.5 : 1.0, wx = ox ? 0.5 : 1.0;
  const double w = wz * wy * wx;      // (every entry of the row carries the same weight: 1/2 per odd coordinate)
  double xv[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int dz = (s >> 2) & oz, dy = (s >> 1) & oy, dx = s & ox;   // (an absent neighbour repeats a present one: weight 0 below)
    xv[s] = xc[c0 + dz * T.Pc + dy * T.nc1 + dx];
  }
  const double px = x[row];
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const bool on = (((s >> 2) & 1) <= oz) && (((s >> 1) & 1) <= oy) && ((s & 1) <= ox);
    acc = acc + (on ? w : 0.0) * xv[s];
  }
  x[row] = 1.0 * acc + 1.0 * px;      // (the AXPBY epilogue with alpha = beta = 1: alpha*acc + beta*y)
}

}  // namespace mgk
