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
// A grid pair may be EMBEDDED (round 6; the sharded cycle's extended boxes, ghost_dist.ghost_boxes): the fine grid nf = 2*m - 1 nodes
// pairs with the SUB-BOX [o, o + m) of a larger coarse grid nc - coarse node (o1+i, o2+j, o3+k) coincides with fine node (2i, 2j, 2k);
// R's rows outside the sub-box are empty, P reads no column outside it.  An ordinary pair has o = 0, m = nc.
struct SmallRDev {
  const double* rec;   // [27 codes of the COARSE position][27]: weights of the fine nodes (2i+dx, 2j+dy, 2k+dz)  (ordinary pairs only)
  int nc1, nc2, nc3, Pc, nc;
  int nf1, nf2, nf3, Pf, nf;
  int o1, o2, o3, m1, m2, m3;
};
struct SmallPDev {
  int nf1, nf2, nf3, Pf, nf;
  int nc1, nc2, nc3, Pc, nc;
  int o1, o2, o3, m1, m2, m3;
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

// bc = R r for the full-weighting restriction R = scale * P' of a vertex-centred grid pair (weights scale * 2^-(|dz|+|dy|+|dx|) for the
// fine nodes (2i+dx, 2j+dy, 2k+dz) inside the grid: verified on the host), any size, with 9 memory instructions per coarse node instead
// of 27 (+ 27 record loads in the kernel above): lanes 1..62 of a wavefront own consecutive coarse nodes (lanes 0 and 63 are halo lanes
// that only load); per fine line (dz, dy) a lane loads ONE aligned 16-byte pair - (r[2i], r[2i+1]) where the line's first entry is even in
// memory, (r[2i-1], r[2i]) where it is odd - and takes the third value from its neighbour lane by a wavefront shuffle.  Every fine entry
// is loaded once per line it is used in, every load of a wavefront is one contiguous kilobyte.  Products added in ascending fine column
// order = the CSR row of R: the bits of the kernels above.
// One 16-byte pair (r[pb], r[pb + 1]), no branch, no bounds test (round 6: a test per load cost every wavefront ~8 instructions x 9 in a kernel
// whose 450 instructions per wavefront, not its bytes, set its time).  The only pair without a second entry starts at the LAST fine node,
// pb = nf - 1: nf is odd (2m - 1 nodes per direction), so with a 16-byte aligned vector that pair lies inside one aligned 16-byte granule
// - the 8 bytes behind the vector are in the same page, whoever owns them; the caller never USES that entry (wr_neighbour / the weight test).
__device__ __forceinline__ d2_t wr_load_pair(const double* __restrict__ r, long long pb) { return *reinterpret_cast<const d2_t*>(r + pb); }
// the neighbour lane's value by a wavefront shift on the vector pipe (DPP wave_shr:1 / wave_shl:1; lanes 0 / 63 keep 0: halo lanes)
template <int CTRL>
__device__ __forceinline__ double wr_neighbour(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u & 0xffffffffull), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo));
}
// fs (optional; round 6): ONE more workgroup at the end of the grid adds up the np per-workgroup partial sums of the pass that ran in front
// of this launch (sum_final_mirror's job: the solve loop's ||r||^2 of the four-stage pass - the fine restriction is the next launch of
// every step, and a launch of its own for 245 numbers cost the step 4 us of kernel + 5 us of idle queue behind it)
struct FinalSum {
  const double* partial;
  int np;
  double* out;
  double* host_mirror;
};
__global__ __launch_bounds__(256) void grid_wave_restrict(SmallRDev T, double scale, const double* __restrict__ r, double* __restrict__ bc,
                                                          const double* __restrict__ d2, double* __restrict__ y2, FinalSum fs) {
  const int lane = threadIdx.x & 63;
  const int nwg = (int)gridDim.x - (fs.partial ? 1 : 0);
  if ((int)blockIdx.x == nwg) {      // (uniform: the extra workgroup; the same code as sum_final_mirror - same bits)
    __shared__ double red[BLK / 64];
    double acc = 0.0;
    for (int i = threadIdx.x; i < fs.np; i += BLK) acc += fs.partial[i];
    const double s = block_sum(acc, red);
    if (threadIdx.x == 0) {
      fs.out[0] = s;
      fs.host_mirror[0] = s;
    }
    return;
  }
  // (workgroups take the coarse nodes in XCD bands: the fine lines two neighbouring coarse lines / planes share are re-read from that L2)
  const int wv = __builtin_amdgcn_readfirstlane(xcd_band((int)blockIdx.x, nwg) * 4 + (int)(threadIdx.x >> 6));
  const int c = wv * 62 + lane - 1;
  const bool own = lane >= 1 && lane <= 62 && c < T.nc;
  const int cc = c < 0 ? 0 : (c >= T.nc ? T.nc - 1 : c);     // (halo lanes beyond the ends: their values meet the weight 0)
  // (k, j, i) of the node: the two divisions once per WAVEFRONT on the scalar unit (its first node), the lanes step from there - a line
  // of >= 64 coarse nodes is left at most once inside a wavefront
  int kk, jj, ii;
  if (T.nc1 >= 64) {
    const int cb = wv * 62 - 1 < 0 ? 0 : (wv * 62 - 1 >= T.nc ? T.nc - 1 : wv * 62 - 1);   // (uniform)
    const int kb = cb / T.Pc, rb = cb - kb * T.Pc, jb = rb / T.nc1, ib = rb - jb * T.nc1;
    ii = ib + (cc - cb);
    jj = jb;
    kk = kb;
    if (ii >= T.nc1) { ii -= T.nc1; ++jj; }
    if (jj >= T.nc2) { jj -= T.nc2; ++kk; }
  } else {
    kk = cc / T.Pc;
    const int rem = cc - kk * T.Pc;
    jj = rem / T.nc1;
    ii = rem - jj * T.nc1;
  }
  // position inside the sub-box the fine grid pairs with (an ordinary pair: the whole coarse grid); a row outside it is empty: its
  // lanes load from the nearest node of the sub-box and every weight is 0
  const int k0 = kk - T.o3, j0 = jj - T.o2, i0 = ii - T.o1;
  const bool inside = k0 >= 0 && k0 < T.m3 && j0 >= 0 && j0 < T.m2 && i0 >= 0 && i0 < T.m1;
  const int k = k0 < 0 ? 0 : (k0 >= T.m3 ? T.m3 - 1 : k0), j = j0 < 0 ? 0 : (j0 >= T.m2 ? T.m2 - 1 : j0), i = i0 < 0 ? 0 : (i0 >= T.m1 ? T.m1 - 1 : i0);
  const long long centre = (long long)(2 * k) * T.Pf + (long long)(2 * j) * T.nf1 + 2 * i;     // (even: an aligned pair starts there)
  const double wxm = (inside && i > 0) ? 0.5 : 0.0, wxp = (inside && i < T.m1 - 1) ? 0.5 : 0.0;
  const bool lastnode = centre + 1 >= T.nf;      // (its aligned pairs have no second entry: never used, see wr_load_pair)
  double acc = 0.0;
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const bool ez = dz == 0 || (dz < 0 ? k > 0 : k < T.m3 - 1), ey = dy == 0 || (dy < 0 ? j > 0 : j < T.m2 - 1);
      const bool ex = inside && ez && ey;
      const double wl = ex ? scale * (dz ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) : 0.0;
      const bool aligned = ((dz + dy) & 1) == 0;                                // (Pf and nf1 are odd)
      // a line the node does not have: the centre line (aligned case) or the pair in front of the centre (its values meet the weight 0)
      const long long f = ex ? centre + (long long)dz * T.Pf + (long long)dy * T.nf1 : (aligned ? centre : (centre > 0 ? centre - 1 : 1));
      double vm, v0, vp;
      if (aligned) {
        const d2_t q = wr_load_pair(r, f);
        vm = wr_neighbour<0x138>(q.y);           // lane - 1's second entry
        v0 = q.x;
        vp = lastnode ? 0.0 : q.y;
      } else {
        const d2_t q = wr_load_pair(r, f - 1);
        vp = wr_neighbour<0x130>(q.x);           // lane + 1's first entry
        vm = q.x;
        v0 = q.y;
      }
      acc = acc + (wl * wxm) * vm;
      acc = acc + wl * v0;
      acc = acc + (wl * wxp) * vp;
    }
  if (own) {
    bc[c] = acc;
    if (y2) y2[c] = d2[c] * acc;
  }
}

// x += P xc, P the full-weighting interpolation of a vertex-centred grid pair (fine = 2*coarse - 1 nodes per direction): weight 1
// from the coincident coarse node, 1/2 from each of the two neighbours of an odd coordinate; the (up to 8) products are added in
// ascending coarse column order, as the CSR row holds them.
__global__ __launch_bounds__(256) void grid_small_prolong(SmallPDev T, const double* __restrict__ xc, double* __restrict__ x) {
  const int row = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (row >= T.nf) return;
  const int z = row / T.Pf, rem = row - z * T.Pf, yy = rem / T.nf1, xx = rem - yy * T.nf1;
  const int oz = z & 1, oy = yy & 1, ox = xx & 1;
  const int c0 = ((z >> 1) + T.o3) * T.Pc + ((yy >> 1) + T.o2) * T.nc1 + (xx >> 1) + T.o1;
  const double wz = oz ? 0.5 : 1.0, wy = oy ? 0.5 : 1.0, wx = ox ? 0.5 : 1.0;
  const double w = wz * wy * wx;      // (every entry of the row carries the same weight: 1/2 per odd coordinate)
  double xv[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {      // (only the neighbours the row has are loaded: 1, 2, 4 or 8 of them)
    const bool on = (((s >> 2) & 1) <= oz) && (((s >> 1) & 1) <= oy) && ((s & 1) <= ox);
    xv[s] = 0.0;
    if (on) xv[s] = xc[c0 + ((s >> 2) & 1) * T.Pc + ((s >> 1) & 1) * T.nc1 + (s & 1)];
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

// The same product with a lane per COARSE CELL: the up to 2 x 2 x 2 fine nodes (2i + dx, 2j + dy, 2k + dz) share the 8 coarse corners
// of the cell, so a lane issues 8 coarse loads + 8 loads and 8 stores of x for 8 fine nodes - 3 memory instructions per fine node
// where the lane-per-fine-row form above issues up to 10 (1-8 gathers + load + store) and the windowed form reads 4 bytes of
// descriptors per row.  Consecutive lanes = consecutive coarse columns: every x access of a wavefront covers one contiguous stretch
// of a fine line.  Per fine node the same products in the same order as above (the CSR row of P): same bits.
__global__ __launch_bounds__(256) void grid_cell_prolong(SmallPDev T, const double* __restrict__ xc, double* __restrict__ x) {
  const int cs = (int)(xcd_band((int)blockIdx.x, (int)gridDim.x) * 256 + threadIdx.x);   // (XCD bands: the coarse corners neighbouring cells share)
  const int Pm = T.m1 * T.m2;                                                  // cells of the sub-box the fine grid pairs with
  if (cs >= Pm * T.m3) return;
  // (k, j, i) of the cell: the two divisions once per WAVEFRONT on the scalar unit, the lanes step from its first cell (a line of >= 64 cells
  // is left at most once inside a wavefront)
  int k, j, i;
  if (T.m1 >= 64) {
    const int cb = __builtin_amdgcn_readfirstlane(cs);
    const int kb = cb / Pm, rb = cb - kb * Pm, jb = rb / T.m1, ib = rb - jb * T.m1;
    i = ib + (cs - cb);
    j = jb;
    k = kb;
    if (i >= T.m1) { i -= T.m1; ++j; }
    if (j >= T.m2) { j -= T.m2; ++k; }
  } else {
    k = cs / Pm;
    const int rem = cs - k * Pm;
    j = rem / T.m1;
    i = rem - j * T.m1;
  }
  const int c = (k + T.o3) * T.Pc + (j + T.o2) * T.nc1 + i + T.o1;
  const bool hx = i + 1 < T.m1, hy = j + 1 < T.m2, hz = k + 1 < T.m3;         // the cell has odd fine nodes in that direction
  const int ox = hx ? 1 : 0, oy = hy ? T.nc1 : 0, oz = hz ? T.Pc : 0;
  double cv[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) cv[s] = xc[c + ((s >> 2) & 1) * oz + ((s >> 1) & 1) * oy + (s & 1) * ox];
  const long long f0 = (long long)(2 * k) * T.Pf + (long long)(2 * j) * T.nf1 + 2 * i;
  // the cell's two fine nodes of a line (dx = 0, 1) travel as ONE 16-byte access (round 6; 8 loads + 8 stores of 8 bytes before): the lines
  // with dz + dy odd start at an odd index - a 16-byte access at an 8-byte aligned address, which the hardware takes (march_load_pair_raw)
  double px[8];
#pragma unroll
  for (int u = 0; u < 4; ++u) {          // u = (dz, dy)
    const int dz = u >> 1, dy = u & 1;
    const bool on0 = (!dz || hz) && (!dy || hy);
    const long long f = on0 ? f0 + (long long)dz * T.Pf + dy * T.nf1 : f0;
    const d2_t v = wr_load_pair(x, f);            // (the last cell of a line: its second entry is the next line's first node - loaded, not used)
    px[2 * u] = v.x;
    px[2 * u + 1] = v.y;
  }
  double out[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) {          // t = (dz, dy, dx) of the fine node inside the cell
    const int dz = (t >> 2) & 1, dy = (t >> 1) & 1, dx = t & 1;
    const double w = (dz ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) * (dx ? 0.5 : 1.0);
    double acc = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) {        // the corners this node takes: those not beyond it in any direction (the others carried the weight 0:
      const bool take = (((s >> 2) & 1) <= dz) && (((s >> 1) & 1) <= dy) && ((s & 1) <= dx);   // 37 of the 64 products of a cell - dropped)
      if (take) acc = acc + w * cv[s];
    }
    out[t] = 1.0 * acc + 1.0 * px[t];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int dz = u >> 1, dy = u & 1;
    const bool on0 = (!dz || hz) && (!dy || hy);
    const long long f = f0 + (long long)dz * T.Pf + dy * T.nf1;
    if (on0) {
      if (hx) *reinterpret_cast<d2_t*>(x + f) = d2_t{out[2 * u], out[2 * u + 1]};
      else x[f] = out[2 * u];
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Band-27: 27-point grid operators whose COEFFICIENTS differ from row to row - the Galerkin levels of a div-sigma-grad hierarchy
// (what jInv feeds the package: testGMG.jl:57-75; MGsetup.jl:226-270 exists because sigma changes every outer iteration).  No
// two rows are equal, so there are no row classes; the pattern-coded CSR kernels moved 8 B per non-zero + descriptors through an LDS
// phase.  Here the values live in 27 PLANAR arrays val[s][row] (s = (dz+1)*9 + (dy+1)*3 + (dx+1); 0 where a row has no such
// entry), filled ON THE DEVICE from the CSR arrays (band27_fill: again after mg_replace_values / mg_rap), and a lane owns a row:
// 27 coalesced value loads + 27 gathers of x issued at once, products added in (dz, dy, dx) = CSR order.  216 + 8..32 B per row.
// ------------------------------------------------------------------------------------------------------------------------------
struct Band27Dev {
  const double* val;   // [27][stride]; every slot is followed by zeros (stride >= n + P + n1 + 1)
  long long stride;
  int n1, n2, n3, P, n;
  int sym;             // the values are symmetric entry by entry - BIT FOR BIT by default (round 6; then reading a lower entry from the neighbour's
                       // upper one changes nothing), or, with the option band_sym_tol, up to the rounding of the Galerkin product R*(A*P) that made them
                       // (band27_sym_check: |A[i,j] - A[j,i]| <= 2^-50 |A[i,i]|; measured 1e-16 on div sigma grad): the 13 lower slots of row r are READ from the upper
                       // slots of the neighbour - slot (dz,dy,dx) of row r = slot (-dz,-dy,-dx) of row r + dz*P + dy*n1 + dx - so that
                       // 14 of the 27 planes are streamed from HBM; workgroups then take rows in XCD bands (the re-read lines sit in that L2)
};
// bad += rows whose lower entries differ from the neighbours' upper ones by more than the rounding of a Galerkin product (the two
// are sums of the same terms in different orders); an entry one side has and the other lacks (a 0) counts unless it is that small
// rtol = 0 (the default): bit for bit - the operator the kernel applies IS the stored one; rtol = 2^-50 (option band_sym_tol): up to the
// rounding of the Galerkin product, |A[i,j] - A[j,i]| <= rtol |A[i,i]|
__global__ __launch_bounds__(256) void band27_sym_check(Band27Dev T, double rtol, int* __restrict__ bad) {
  const int row = (int)(blockIdx.x * 256 + threadIdx.x);
  if (row >= T.n) return;
  const double tol = rtol * fabs(T.val[13LL * T.stride + row]);
  bool ok = true;
#pragma unroll
  for (int s = 0; s < 13; ++s) {
    const int dz = s / 9 - 1, dy = (s / 3) % 3 - 1, dx = s % 3 - 1;
    ok = ok && fabs(T.val[(long long)s * T.stride + row] - T.val[(long long)(26 - s) * T.stride + row + dz * T.P + dy * T.n1 + dx]) <= tol;
  }
  if (!ok) atomicAdd(bad, 1);
}
// flag |= 1 if a row has an entry outside the 3 x 3 x 3 neighbourhood of its node or out of ascending order (the form is then dropped)
__global__ __launch_bounds__(256) void band27_fill(CsrDev A, int n1, int n2, int n3, double* __restrict__ val, long long stride, int* __restrict__ flag) {
  const int row = (int)(blockIdx.x * 256 + threadIdx.x);
  if (row >= A.n_rows) return;
  const int P = n1 * n2;
  const int z = row / P, rem = row - z * P, y = rem / n1, x = rem - y * n1;
#pragma unroll
  for (int s = 0; s < 27; ++s) val[(long long)s * stride + row] = 0.0;
  int last = -1, bad = 0;
  for (int k = A.rowptr[row]; k < A.rowptr[row + 1]; ++k) {
    const int c = A.colidx[k];
    const int cz = c / P, cr = c - cz * P, cy = cr / n1, cx = cr - cy * n1;
    const int dz = cz - z, dy = cy - y, dx = cx - x;
    if (dz < -1 || dz > 1 || dy < -1 || dy > 1 || dx < -1 || dx > 1) { bad = 1; continue; }
    const int s = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
    if (s <= last) bad = 1;
    last = s;
    val[(long long)s * stride + row] = A.val[k];
  }
  if (bad) atomicOr(flag, 1);
}
// y = b - A x (RESID), y = x + d.*(b - A x) (SMOOTH), y = alpha A x + beta y (AXPBY)
// Lanes 1..62 of a wavefront own consecutive rows (0 and 63 are halo lanes): of the 27 entries of x a row reads, the 9 with dx = 0 are
// loaded (one coalesced load per (dz, dy) line) and the 18 with dx = -1 / +1 come from the neighbour lanes by a wavefront shuffle - 9
// memory instructions for x instead of 27 (where a line ends, the neighbour lane's value is not the row's neighbour: the band holds 0 there).
template <int MODE>
__global__ __launch_bounds__(256) void grid27_band_spmv(Band27Dev T, VecArgs v) {
  const int lane = threadIdx.x & 63;
  const long long wv = (long long)(T.sym ? xcd_band((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x) * 4 + (threadIdx.x >> 6);
  const long long rr = wv * 62 + lane - 1;
  const bool own = lane >= 1 && lane <= 62 && rr < T.n;
  const int row = (int)(rr < 0 ? 0 : (rr >= T.n ? (long long)T.n - 1 : rr));
  double rv[27], xv[27];
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      int j = row + dz * T.P + dy * T.n1;
      j = j < 0 ? 0 : (j >= T.n ? T.n - 1 : j);       // (an entry the row does not have: value 0 times some valid, finite x)
      const double xc = v.x[j];
      const int s0 = (dz + 1) * 9 + (dy + 1) * 3;
      xv[s0] = __shfl_up(xc, 1);
      xv[s0 + 1] = xc;
      xv[s0 + 2] = __shfl_down(xc, 1);
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const int s = s0 + dx + 1;
        const long long src = (s < 13 && T.sym) ? (long long)(26 - s) * T.stride + (dz * T.P + dy * T.n1 + dx) : (long long)s * T.stride;   // (uniform)
        rv[s] = T.val[src + row];
      }
    }
  double acc = 0.0;
#pragma unroll
  for (int s = 0; s < 27; ++s) acc = acc + rv[s] * xv[s];
  if (!own) return;
  double out;
  if (MODE == AXPBY) out = v.alpha * acc + (v.beta != 0.0 ? v.beta * v.y[row] : 0.0);
  else if (MODE == RESID) out = v.b[row] - acc;
  else out = v.xs[row] + v.d_full[row] * (v.b[row] - acc);
  v.y[row] = out;
}

// ------------------------------------------------------------------------------------------------------------------------------
// Two launches of a small level as ONE (round 6).  The cycle of a level below ~300 000 rows is a chain of launches of 4.5-9 us each
// for microseconds of work (levels 3-6 of C2: 20 % of the step for 1.9 % of the rows); two of its pairs have a consumer that reads
// nothing but the producer's 3 x 3 x 3 neighbourhood, so a workgroup can make what it needs itself, in LDS:
//   grid27_small_resid_restrict  r = b - A x (MGcycle.jl:58-60) on the 9^3 fine nodes around a 4 x 4 x 4 tile of coarse nodes, then
//                                bc = R r (l.66) on the tile [and the coarse level's first update d.*bc]: r never reaches memory;
//                                1.42 x the level's residuals are computed (the tiles' shared faces), from ONE staging of x
//   grid27_small_prolong_smooth  x + P xc (l.90) on the 10 x 10 x 6 nodes around an 8 x 8 x 4 tile of fine nodes, then one sweep
//                                y = (x + P xc) + d.*(b - A (x + P xc)) (l.92-102, 129-131) on the tile: the corrected x never reaches memory
// Same expressions, same order of the products as grid27_small_spmv / grid_wave_restrict / grid_small_prolong: same bits (a node
// outside the grid stages 0 where those kernels multiply a clamped neighbour by the value 0: a zero may change its sign).
// Ordinary grid pairs only (fine = 2 * coarse - 1 nodes; not the embedded pairs of the sharded cycle).
// ------------------------------------------------------------------------------------------------------------------------------
constexpr int SRR_CT = 4, SRR_XR = 2 * SRR_CT + 3, SRR_RR = 2 * SRR_CT + 1;     // coarse tile edge; edges of the staged x / r bricks
constexpr int SRR_NX = SRR_XR * SRR_XR * SRR_XR, SRR_NR = SRR_RR * SRR_RR * SRR_RR;
constexpr int SRR_XPL = (SRR_NX + 255) / 256, SRR_RPL = (SRR_NR + 255) / 256;   // entries of x / rows of r per lane
// (every global load a phase needs is issued BEFORE the barrier in front of it: the phases themselves touch LDS and registers only)
__global__ __launch_bounds__(256) void grid27_small_resid_restrict(Small27Dev A, SmallRDev R, double scale, const double* __restrict__ x,
                                                                   const double* __restrict__ b, double* __restrict__ bc,
                                                                   const double* __restrict__ d2, double* __restrict__ y2) {
  __shared__ double xs[SRR_NX];
  __shared__ double rs[SRR_NR];
  __shared__ double recs[27 * 28];
  const int tid = threadIdx.x;
  const int tx_n = (R.nc1 + SRR_CT - 1) / SRR_CT, ty_n = (R.nc2 + SRR_CT - 1) / SRR_CT;
  const int bz = (int)blockIdx.x / (tx_n * ty_n), brem = (int)blockIdx.x - bz * tx_n * ty_n, by = brem / tx_n, bx = brem - by * tx_n;
  const int cx0 = bx * SRR_CT, cy0 = by * SRR_CT, cz0 = bz * SRR_CT;
  const int fx0 = 2 * cx0 - 2, fy0 = 2 * cy0 - 2, fz0 = 2 * cz0 - 2;              // first node of the x brick
  double xr[SRR_XPL], pb[SRR_RPL], rc[3];
  int code[SRR_RPL];
#pragma unroll
  for (int u = 0; u < SRR_XPL; ++u) {
    const int i = tid + u * 256;
    const int uz = i / (SRR_XR * SRR_XR), ur = i - uz * SRR_XR * SRR_XR, uy = ur / SRR_XR, ux = ur - uy * SRR_XR;
    const int fz = fz0 + uz, fy = fy0 + uy, fx = fx0 + ux;
    const bool in = i < SRR_NX && fz >= 0 && fz < A.n3 && fy >= 0 && fy < A.n2 && fx >= 0 && fx < A.n1;
    xr[u] = in ? x[fz * A.P + fy * A.n1 + fx] : 0.0;
  }
#pragma unroll
  for (int u = 0; u < SRR_RPL; ++u) {
    const int i = tid + u * 256;
    const int tz = i / (SRR_RR * SRR_RR), tr = i - tz * SRR_RR * SRR_RR, ty = tr / SRR_RR, tx = tr - ty * SRR_RR;
    const int fz = fz0 + 1 + tz, fy = fy0 + 1 + ty, fx = fx0 + 1 + tx;
    const bool in = i < SRR_NR && fz >= 0 && fz < A.n3 && fy >= 0 && fy < A.n2 && fx >= 0 && fx < A.n1;
    pb[u] = in ? b[fz * A.P + fy * A.n1 + fx] : 0.0;
    code[u] = in ? (small_code1(fz, A.n3) * 3 + small_code1(fy, A.n2)) * 3 + small_code1(fx, A.n1) : -1;
  }
#pragma unroll
  for (int u = 0; u < 3; ++u) rc[u] = tid + u * 256 < 27 * 28 ? A.rec[tid + u * 256] : 0.0;
  // the coarse node of the last phase (lanes 0 .. 63)
  const int lz = (tid & 63) / (SRR_CT * SRR_CT), lr = (tid & 63) - lz * SRR_CT * SRR_CT, ly = lr / SRR_CT, lx = lr - ly * SRR_CT;
  const int k = cz0 + lz, j = cy0 + ly, ii = cx0 + lx;
  const bool cown = tid < SRR_CT * SRR_CT * SRR_CT && k < R.nc3 && j < R.nc2 && ii < R.nc1;
  const int crow = cown ? k * R.Pc + j * R.nc1 + ii : 0;
  const double pd = (cown && y2) ? d2[crow] : 0.0;
#pragma unroll
  for (int u = 0; u < SRR_XPL; ++u)
    if (tid + u * 256 < SRR_NX) xs[tid + u * 256] = xr[u];
#pragma unroll
  for (int u = 0; u < 3; ++u)
    if (tid + u * 256 < 27 * 28) recs[tid + u * 256] = rc[u];
  __syncthreads();
#pragma unroll
  for (int u = 0; u < SRR_RPL; ++u) {
    const int i = tid + u * 256;
    if (i >= SRR_NR) break;
    const int tz = i / (SRR_RR * SRR_RR), tr = i - tz * SRR_RR * SRR_RR, ty = tr / SRR_RR, tx = tr - ty * SRR_RR;
    double out = 0.0;
    if (code[u] >= 0) {
      const double* rec = recs + code[u] * 28;
      double rv[27], xv[27];
#pragma unroll
      for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const int s_ = (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1);
            rv[s_] = rec[s_];
            xv[s_] = xs[((tz + 1 + dz) * SRR_XR + (ty + 1 + dy)) * SRR_XR + (tx + 1 + dx)];
          }
      double acc = 0.0;
#pragma unroll
      for (int s_ = 0; s_ < 27; ++s_) acc = acc + rv[s_] * xv[s_];
      out = pb[u] - acc;
    }
    rs[i] = out;
  }
  __syncthreads();
  if (!cown) return;
  const double wxm = ii > 0 ? 0.5 : 0.0, wxp = ii < R.nc1 - 1 ? 0.5 : 0.0;
  double acc = 0.0;
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const bool ez = dz == 0 || (dz < 0 ? k > 0 : k < R.nc3 - 1), ey = dy == 0 || (dy < 0 ? j > 0 : j < R.nc2 - 1);
      const double wl = (ez && ey) ? scale * (dz ? 0.5 : 1.0) * (dy ? 0.5 : 1.0) : 0.0;
      const double* q = rs + ((2 * lz + 1 + dz) * SRR_RR + (2 * ly + 1 + dy)) * SRR_RR + 2 * lx + 1;
      acc = acc + (wl * wxm) * q[-1];
      acc = acc + wl * q[0];
      acc = acc + (wl * wxp) * q[1];
    }
  bc[crow] = acc;
  if (y2) y2[crow] = pd * acc;
}

constexpr int SPS_TX = 8, SPS_TY = 8, SPS_TZ = 4, SPS_XX = SPS_TX + 2, SPS_XY = SPS_TY + 2, SPS_XZ = SPS_TZ + 2;
constexpr int SPS_NX = SPS_XZ * SPS_XY * SPS_XX, SPS_XPL = (SPS_NX + 255) / 256;
__global__ __launch_bounds__(256) void grid27_small_prolong_smooth(Small27Dev A, SmallPDev T, const double* __restrict__ xc, const double* __restrict__ x,
                                                                   const double* __restrict__ b, double* __restrict__ y) {
  __shared__ double xs[SPS_NX];
  const int tid = threadIdx.x;
  const int tx_n = (A.n1 + SPS_TX - 1) / SPS_TX, ty_n = (A.n2 + SPS_TY - 1) / SPS_TY;
  const int bz = (int)blockIdx.x / (tx_n * ty_n), brem = (int)blockIdx.x - bz * tx_n * ty_n, by = brem / tx_n, bx = brem - by * tx_n;
  const int x0 = bx * SPS_TX, y0 = by * SPS_TY, z0 = bz * SPS_TZ;
  // the lane's own row and everything it needs from memory, in flight beside the staging loads
  const int lz = tid / (SPS_TY * SPS_TX), lr = tid - lz * SPS_TY * SPS_TX, ly = lr / SPS_TX, lx = lr - ly * SPS_TX;
  const int oz_ = z0 + lz, oy_ = y0 + ly, ox_ = x0 + lx;
  const bool own = oz_ < A.n3 && oy_ < A.n2 && ox_ < A.n1;
  const int orow = own ? oz_ * A.P + oy_ * A.n1 + ox_ : 0;
  const int ocode = own ? (small_code1(oz_, A.n3) * 3 + small_code1(oy_, A.n2)) * 3 + small_code1(ox_, A.n1) : 13;
  const double* __restrict__ rec = A.rec + ocode * 28;
  double rv[27];
#pragma unroll
  for (int s_ = 0; s_ < 27; ++s_) rv[s_] = rec[s_];
  const double pb = b[orow], pd = rec[27];
  double st[SPS_XPL];
#pragma unroll
  for (int u = 0; u < SPS_XPL; ++u) {
    const int i = tid + u * 256;
    const int uz = i / (SPS_XY * SPS_XX), ur = i - uz * SPS_XY * SPS_XX, uy = ur / SPS_XX, ux = ur - uy * SPS_XX;
    const int z = z0 - 1 + uz, yy = y0 - 1 + uy, xx = x0 - 1 + ux;
    double out = 0.0;
    if (i < SPS_NX && z >= 0 && z < A.n3 && yy >= 0 && yy < A.n2 && xx >= 0 && xx < A.n1) {      // (grid_small_prolong's row, kept in LDS)
      const int row = z * T.Pf + yy * T.nf1 + xx;
      const int oz = z & 1, oy = yy & 1, ox = xx & 1;
      const int c0 = ((z >> 1) + T.o3) * T.Pc + ((yy >> 1) + T.o2) * T.nc1 + (xx >> 1) + T.o1;
      const double wz = oz ? 0.5 : 1.0, wy = oy ? 0.5 : 1.0, wx = ox ? 0.5 : 1.0;
      const double w = wz * wy * wx;
      double xv[8];
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const bool on = (((s_ >> 2) & 1) <= oz) && (((s_ >> 1) & 1) <= oy) && ((s_ & 1) <= ox);
        xv[s_] = 0.0;
        if (on) xv[s_] = xc[c0 + ((s_ >> 2) & 1) * T.Pc + ((s_ >> 1) & 1) * T.nc1 + (s_ & 1)];
      }
      const double px = x[row];
      double acc = 0.0;
#pragma unroll
      for (int s_ = 0; s_ < 8; ++s_) {
        const bool on = (((s_ >> 2) & 1) <= oz) && (((s_ >> 1) & 1) <= oy) && ((s_ & 1) <= ox);
        acc = acc + (on ? w : 0.0) * xv[s_];
      }
      out = 1.0 * acc + 1.0 * px;
    }
    st[u] = out;
  }
#pragma unroll
  for (int u = 0; u < SPS_XPL; ++u)
    if (tid + u * 256 < SPS_NX) xs[tid + u * 256] = st[u];
  __syncthreads();
  if (!own) return;
  double xv[27];
#pragma unroll
  for (int dz = -1; dz <= 1; ++dz)
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        xv[(dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)] = xs[((lz + 1 + dz) * SPS_XY + (ly + 1 + dy)) * SPS_XX + (lx + 1 + dx)];
  double acc = 0.0;
#pragma unroll
  for (int s_ = 0; s_ < 27; ++s_) acc = acc + rv[s_] * xv[s_];
  y[orow] = xv[13] + pd * (pb - acc);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Long rows: one WAVEFRONT per row.  The Galerkin levels of an SA-AMG hierarchy on anisotropic diffusion carry rows of a few hundred
// to several thousand entries (SA-AMG.jl:44-50: the smoothed prolongation widens every coarse stencil; BASELINE config C3: levels 3-5
// hold 80 % of the hierarchy's 2.5 G non-zeros).  The LDS-staged segmented reduction of csr_stream_spmv serves them at 0.36-0.58 of
// the HBM peak: every product goes through LDS and two barriers although a row is far longer than a wavefront.  Here the 64 lanes
// stride over the row - 512-byte value and 256-byte index loads per wavefront instruction, four of each in flight per lane, the
// gathers of x behind them - accumulate in registers and meet in one shuffle tree per row: no LDS, no barrier.  A workgroup owns a
// contiguous band of rows (xcd_band: neighbouring rows share their gather window in the XCD's L2).  Summation order: per lane its
// strided entries in stored order, then a fixed tree - deterministic, fp64 reassociation against the CSR row loop (<= 1e-13).
// ------------------------------------------------------------------------------------------------------------------------------
// IDX16: the column indices are 16-bit offsets from the row's first column (ci16 / rowbase: 10 instead of 12 bytes per non-zero)
template <int MODE, bool NT, bool IDX16, typename AD = CsrDev>
__global__ __launch_bounds__(256) void csr_longrow_spmv(AD A, VecArgs v, int nwg, int rows_per_wg, const unsigned short* __restrict__ ci16,
                                                        const int* __restrict__ rowbase) {
  typedef typename AD::ptr_t K;      // position in the non-zero stream (int; long long for operators of >= 2^31 non-zeros)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int w = xcd_band((int)blockIdx.x, nwg);
  const int r0 = w * rows_per_wg, r1 = r0 + rows_per_wg < A.n_rows ? r0 + rows_per_wg : A.n_rows;
  typedef unsigned short us2_t __attribute__((ext_vector_type(2)));
  // A SHOT = 512 consecutive entries from an EVEN entry index sb: lane l takes the pairs sb + 2l + 128u, u = 0..3 - four 16-byte value
  // loads and four index-pair loads (4 bytes with 16-bit offsets) per lane, 2 memory instructions per non-zero with the gathers where
  // single entries took 3.  Entries outside the row [k0, k1) are loaded all the same (the arrays end in 512 spare entries) and
  // replaced by (value 0, column 0) before they are used.
#define LR_SHOT_LOAD(VV, C0, C1, sb)                                                                                   \
  _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                                   \
    const K p_ = (sb) + 2 * lane + 128 * u_;                                                                         \
    (VV)[u_] = NT ? __builtin_nontemporal_load(reinterpret_cast<const d2_t*>(A.val + p_)) : *reinterpret_cast<const d2_t*>(A.val + p_); \
    if (IDX16) {                                                                                                       \
      const us2_t c_ = NT ? __builtin_nontemporal_load(reinterpret_cast<const us2_t*>(ci16 + p_)) : *reinterpret_cast<const us2_t*>(ci16 + p_); \
      (C0)[u_] = (int)c_.x; (C1)[u_] = (int)c_.y;                                                                      \
    } else {                                                                                                           \
      const i2_t c_ = NT ? __builtin_nontemporal_load(reinterpret_cast<const i2_t*>(A.colidx + p_)) : *reinterpret_cast<const i2_t*>(A.colidx + p_); \
      (C0)[u_] = c_.x; (C1)[u_] = c_.y;                                                                                \
    }                                                                                                                  \
  }
#define LR_SHOT_USE(VV, C0, C1, sb, lo, hi)                                                                            \
  do {                                                                                                                 \
    double x0_[4], x1_[4];                                                                                             \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                                 \
      const K p_ = (sb) + 2 * lane + 128 * u_;                                                                       \
      const bool i0_ = p_ >= (lo) && p_ < (hi), i1_ = p_ + 1 >= (lo) && p_ + 1 < (hi);                                 \
      x0_[u_] = xb[i0_ ? (C0)[u_] : 0];                                                                                \
      x1_[u_] = xb[i1_ ? (C1)[u_] : 0];                                                                                \
    }                                                                                                                  \
    _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                                 \
      const K p_ = (sb) + 2 * lane + 128 * u_;                                                                       \
      const bool i0_ = p_ >= (lo) && p_ < (hi), i1_ = p_ + 1 >= (lo) && p_ + 1 < (hi);                                 \
      const double w0_ = i0_ ? (VV)[u_].x : 0.0, w1_ = i1_ ? (VV)[u_].y : 0.0;                                         \
      if (u_ & 1) { a2 = a2 + w0_ * x0_[u_]; a3 = a3 + w1_ * x1_[u_]; }                                                \
      else { a0 = a0 + w0_ * x0_[u_]; a1 = a1 + w1_ * x1_[u_]; }                                                       \
    }                                                                                                                  \
  } while (0)
  // Software pipeline over the wavefront's rows: the FIRST shot of the next row is in flight while the current row gathers, multiplies
  // and reduces; its row pointers one row further ahead.
  int row = r0 + wave;
  if (row >= r1) return;
  K k0 = uniform_first(A.rowptr[row]), k1 = uniform_first(A.rowptr[row + 1]);
  K nk0 = 0, nk1 = 0;
  if (row + 4 < r1) { nk0 = uniform_first(A.rowptr[row + 4]); nk1 = uniform_first(A.rowptr[row + 5]); }
  d2_t pv[4];
  int pc0[4], pc1[4];
  LR_SHOT_LOAD(pv, pc0, pc1, k0 & ~(K)1);
  for (; row < r1; row += 4) {
    const double* __restrict__ xb = IDX16 ? v.x + __builtin_amdgcn_readfirstlane(rowbase[row]) : v.x;
    const bool have_next = row + 4 < r1;                     // (uniform)
    K nnk0 = 0, nnk1 = 0;
    if (row + 8 < r1) { nnk0 = A.rowptr[row + 8]; nnk1 = A.rowptr[row + 9]; }
    d2_t nv[4];
    int nc0[4], nc1[4];
    if (have_next) { LR_SHOT_LOAD(nv, nc0, nc1, nk0 & ~(K)1); }
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    K sb = k0 & ~(K)1;                                       // (uniform)
    LR_SHOT_USE(pv, pc0, pc1, sb, k0, k1);
    for (sb += 512; sb < k1; sb += 512) {                    // rows beyond the first shot: shot by shot
      d2_t vv[4];
      int c0[4], c1[4];
      LR_SHOT_LOAD(vv, c0, c1, sb);
      LR_SHOT_USE(vv, c0, c1, sb, k0, k1);
    }
    double acc = (a0 + a1) + (a2 + a3);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
      double out;
      if (MODE == AXPBY) {
        out = v.alpha * acc + (v.beta != 0.0 ? v.beta * v.y[row] : 0.0);
        if (v.y2) v.y2[row] = v.d_full[row] * out;
      } else if (MODE == RESID) {
        out = v.b[row] - acc;
      } else {
        out = v.xs[row] + v.d_full[row] * (v.b[row] - acc);
      }
      v.y[row] = out;
    }
    // ---- rotate the pipeline --------------------------------------------------------------------------------------------------
    k0 = nk0; k1 = nk1;
    nk0 = uniform_first(nnk0); nk1 = uniform_first(nnk1);
#pragma unroll
    for (int u = 0; u < 4; ++u) { pv[u] = nv[u]; pc0[u] = nc0[u]; pc1[u] = nc1[u]; }
  }
#undef LR_SHOT_LOAD
#undef LR_SHOT_USE
}

}  // namespace mgk
