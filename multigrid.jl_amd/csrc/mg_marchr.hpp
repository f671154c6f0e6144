// mg_marchr.hpp - the restriction bc = R r (MGcycle.jl:66; R = P' * 0.5^dim of the trilinear P: GeometricTransferOperators.jl:5-36)
// of a vertex-centred grid pair (fine nodes 2*nc - 1 per direction) as a walk along z over the FINE planes.
//
// The gather form (csr_rowclass_lane_spmv: one lane per coarse row, 27 gathers of r, nine in flight) reads every fine line
// 2.25 times through L1 and waits for its gathers: 0.40 of the HBM peak while r still sits in the Infinity Cache (C2, 257^3 ->
// 129^3), 0.28-0.29 once it does not (400^3 and 512^3 cells: 0.34 / 0.68 ms, a fifth of the step).  Here a workgroup owns a
// tile of TX x TY COARSE columns and a run of coarse planes; the fine planes 2k-1 .. 2k+1 under them enter LDS once each
// (16-byte pair loads one iteration ahead, two slabs: the staging of the marching sweeps) and are read once: a lane = one coarse
// column reads the nine entries around (2y, 2x) of the fine plane and adds
//   fine plane 2k-1:  the dz = -1 block of the record to a new accumulator (coarse plane k)
//   fine plane 2k  :  the dz =  0 block
//   fine plane 2k+1:  the dz = +1 block, which completes coarse plane k - and starts k+1 with its dz = -1 block.
// Per coarse row the 27 products are added in ascending fine column = the order of R's CSR row: same bits as the gather form.
// Classes, records and the product map are the 27-point marching form's (mg_march27.hpp; the map is over the COARSE grid).
// Optional second output y2 = d2 .* bc: the coarse level's first damped-Jacobi update (x = 0 there: MGcycle.jl:26-31,134).
#pragma once

namespace mgk {

struct MarchRDev {
  const M27Class* cls;          // [ncls] (d unused)
  const unsigned short* cmap;   // cx[nc1] | cy[nc2] | cz[nc3] | tab[ncz*ncy*ncx]
  int ncx, ncy, ncz, ntab;
  int nc1, nc2, nc3, Pc;        // coarse grid, Pc = nc1*nc2
  int nf1, nf2, nf3, Pf;        // fine grid (2*nc - 1 per direction)
  int TX, TY, tiles_x, tiles_y; // coarse tile (TX*TY <= threads)
  int LY, NPL;                  // lines of a fine slab (2*TY + 1); 16-byte pairs per line
  int nblocks, segs, seglen;    // workgroup w = segment (w / tiles) of tile (w % tiles); seglen coarse planes
  int n_cols, ncls;             // fine rows = columns of R
};

template <int NT, int NPM, int PITCH>
__global__ __launch_bounds__(NT, 4) void csr_rowclass_marchr_spmv(March2Args a, MarchRDev T) {
  extern __shared__ double win[];
  const int tid = threadIdx.x;
  const int w = xcd_band(blockIdx.x, T.nblocks);
  const int XS = T.LY * PITCH;                  // doubles per slab
  const int XS8 = XS * 8;
  char* winb = reinterpret_cast<char*>(win);
  M27Class* dcl = reinterpret_cast<M27Class*>(win + 2 * XS);
  const unsigned short* cxG = T.cmap;
  const unsigned short* cyG = cxG + T.nc1;
  const unsigned short* czG = cyG + T.nc2;
  unsigned short* czL = reinterpret_cast<unsigned short*>(dcl + T.ncls);     // cz | tab
  unsigned short* tabL = czL + T.nc3;
  const int zstride = T.ncy * T.ncx;
  const int ntiles = T.tiles_x * T.tiles_y;
  const int seg = w / ntiles, c = w - seg * ntiles;
  const int k0 = seg * T.seglen, k1 = k0 + T.seglen < T.nc3 ? k0 + T.seglen : T.nc3;   // coarse planes of the run
  if (k1 <= k0) return;                         // (uniform)
  // ---- the lane's coarse column ------------------------------------------------------------------------------------------
  const int xx = tid % T.TX, j = tid / T.TX;
  const bool lane_ok = j < T.TY;
  const int ty = c / T.tiles_x, tx = c - ty * T.tiles_x;
  const int x0 = tx * T.TX, y0 = ty * T.TY;     // coarse
  const int fx0 = 2 * x0 - 1, fy0 = 2 * y0 - 1; // first fine column / line of the slab
  const int own8 = ((2 * (lane_ok ? j : 0) + 1) * PITCH + 2 * xx + 1) * 8;   // the fine node under the coarse one, inside a slab
  int pofs[NPM], pg[NPM];
  unsigned pflag = 0u;      // per m: bit 4m = the pair exists, bit 4m+1 = first pair of its line, bit 4m+2 = its line is inside the grid
#pragma unroll
  for (int m = 0; m < NPM; ++m) {
    const int pid = tid + m * NT;
    const int l = pid / T.NPL, i = pid - l * T.NPL;
    const int yl = fy0 + l;
    pofs[m] = l * PITCH + 2 * i;
    pg[m] = yl * T.nf1 + fx0 + 2 * i;
    if (pid < T.LY * T.NPL) pflag |= 1u << (4 * m);
    if (i == 0) pflag |= 2u << (4 * m);
    if (yl >= 0 && yl < T.nf2) pflag |= 4u << (4 * m);
  }
  const int gx = x0 + xx, gy = y0 + j;
  const bool live = lane_ok && gx < T.nc1 && gy < T.nc2;
  const int ip0 = gy * T.nc1 + gx;              // in-plane index of the coarse row
  double* sk = a.sink + ((size_t)(w & 31) * NT + tid);
  double rv[27];
  int rcls = -1;
#pragma unroll
  for (int u = 0; u < 27; ++u) rv[u] = 0.0;
#define MR_LOADRECS(cq)                                                                                                \
  do {                                                                                                                 \
    const M27Class* q_ = dcl + (cq);                                                                                   \
    _Pragma("unroll") for (int u_ = 0; u_ < 27; ++u_) rv[u_] = q_->v[u_];                                              \
    rcls = (cq);                                                                                                       \
  } while (0)
#define MR_LDS(off8) (*reinterpret_cast<const double*>(winb + (off8)))
#define MR_READ9(xv_, sb_)                                                                                             \
  _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_)                                                                  \
    _Pragma("unroll") for (int dx_ = 0; dx_ < 3; ++dx_)                                                                \
      (xv_)[dy_ * 3 + dx_] = MR_LDS((sb_) + own8 + ((dy_ - 1) * PITCH + (dx_ - 1)) * 8)
#define MR_FMA9(acc, blk, xv_) _Pragma("unroll") for (int u_ = 0; u_ < 9; ++u_) (acc) = (acc) + rv[(blk) * 9 + u_] * (xv_)[u_]
#define MR_CZ(q) ((int)czL[(q) < 0 ? 0 : ((q) >= T.nc3 ? T.nc3 - 1 : (q))])
#define MR_CLS(q) ((int)tabL[MR_CZ(q) * zstride + rp])
#define MR_PAR(p, m) ((int)(((long long)(p) * T.Pf + pg[m]) & 1LL))
#define MR_LOADPAIR(dst, p, m)                                                                                         \
  do {                                                                                                                 \
    const bool act_ = ((pflag >> (4 * (m))) & 5u) == 5u && (p) >= 0 && (p) < T.nf3;                                    \
    const long long e0_ = ((long long)(p) * T.Pf + pg[m]) & ~1LL;                                                      \
    (dst) = march_load_pair_raw(a.x, e0_, act_, T.n_cols);                                                             \
  } while (0)
#define MR_FIXPAIR(v, p, m)                                                                                            \
  do {                                                                                                                 \
    if ((p) == T.nf3 - 1) {                                                                          /* (uniform) */   \
      const bool act_ = ((pflag >> (4 * (m))) & 5u) == 5u;                                                             \
      const long long e0_ = ((long long)(p) * T.Pf + pg[m]) & ~1LL;                                                    \
      march_pair_fix((v), e0_, act_, T.n_cols);                                                                        \
    }                                                                                                                  \
  } while (0)
#define MR_STAGE(slot, p, m, v)                                                                                        \
  do {                                                                                                                 \
    if ((pflag >> (4 * (m))) & 1u) {                                                                                   \
      const int par_ = MR_PAR(p, m);                                                                                   \
      double* q_ = win + ((slot) * XS + pofs[m] - par_);                                                               \
      if (!(par_ && ((pflag >> (4 * (m))) & 2u))) q_[0] = (v).x;                                                       \
      q_[1] = (v).y;                                                                                                   \
    }                                                                                                                  \
  } while (0)
  {
    const int pA = 2 * k0 - 1, pE = 2 * k1 - 1;       // fine planes of the run (pA may be -1, pE may be nf3: no entries there)
    // ---- prologue: every global load goes out before the first wait ------------------------------------------------------
    d2_t q0[NPM], preb[NPM];
#pragma unroll
    for (int m = 0; m < NPM; ++m) MR_LOADPAIR(q0[m], pA, m);
#pragma unroll
    for (int m = 0; m < NPM; ++m) MR_LOADPAIR(preb[m], pA + 1, m);
    const int cyv = (int)cyG[live ? gy : 0], cxv = (int)cxG[live ? gx : 0];
    constexpr int ND = 4;
    const int nw = T.ncls * (int)(sizeof(M27Class) / 8), nm = T.nc3 + T.ntab;
    const double* srcd = reinterpret_cast<const double*>(T.cls);
    double dreg[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) dreg[u] = srcd[tid + u * NT < nw ? tid + u * NT : 0];
    const unsigned short creg = czG[tid < nm ? tid : 0];
    for (int i = tid; i < 2 * XS; i += NT) win[i] = 0.0;
    __syncthreads();
    {
      double* dstd = reinterpret_cast<double*>(dcl);
#pragma unroll
      for (int u = 0; u < ND; ++u)
        if (tid + u * NT < nw) dstd[tid + u * NT] = dreg[u];
      for (int i = tid + ND * NT; i < nw; i += NT) dstd[i] = srcd[i];
      if (tid < nm) czL[tid] = creg;
      for (int i = tid + NT; i < nm; i += NT) czL[i] = czG[i];
    }
#pragma unroll
    for (int m = 0; m < NPM; ++m) {
      MR_FIXPAIR(q0[m], pA, m);
      MR_STAGE(pA & 1, pA, m, q0[m]);
    }
    const int rp = live ? cyv * T.ncx + cxv : 0;
    double dv = 0.0;                                   // d2 of the coarse row being completed
    // as many stores as an iteration issues, behind the loads above (the loop's wait counts them)
#pragma unroll
    for (int i = 0; i < 2; ++i) sk[(size_t)i * 32 * NT] = 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll 1
    for (int p = pA; p <= pE; ++p) {
      d2_t cur[NPM];
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        cur[m] = preb[m];
        asm volatile("" : "+v"(cur[m].x), "+v"(cur[m].y));
      }
      double dcur = dv;
      asm volatile("" : "+v"(dcur));
      if (p + 1 <= pE) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) {
          MR_FIXPAIR(cur[m], p + 1, m);
          MR_STAGE((p + 1) & 1, p + 1, m, cur[m]);
        }
      }
      if (p + 2 <= pE) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) MR_LOADPAIR(preb[m], p + 2, m);
      }
      // p odd  = 2kc + 1: completes coarse plane kc = (p - 1) / 2 and starts kc + 1;  p even = 2kc: the middle block of kc
      const bool odd = (p & 1) != 0;                   // (uniform)
      const int kc = odd ? (p - 1) >> 1 : p >> 1;      // ((-1 - 1) >> 1 = -1: the plane in front of the grid)
      const int kn = odd ? kc + 1 : kc;                // the coarse plane whose record serves the block started / continued
      {
        const int rown = (kn >= k0 && kn < k1 && live) ? kn * T.Pc + ip0 : 0;
        dv = a.d ? a.d[rown] : 0.0;                    // (every iteration: a fixed number of loads in flight)
      }
      const bool fast = !odd || MR_CZ(kc) == MR_CZ(kn);
      if (fast) {
        const int cq = (int)tabL[MR_CZ(kn) * zstride + rp];
        if (cq != rcls) MR_LOADRECS(cq);
      }
      const int sb = (p & 1) * XS8;
      double outv = 0.0;
      if (fast) {
        double xv[9];
        MR_READ9(xv, sb);
        if (odd) {
          double ac = acc;
          MR_FMA9(ac, 2, xv);
          outv = ac;
          acc = 0.0;
          MR_FMA9(acc, 0, xv);
        } else {
          MR_FMA9(acc, 1, xv);
        }
      } else {   // (odd plane between two coarse planes of different z-class: both blocks from the LDS dictionary, rolled)
        const double *vC = dcl[MR_CLS(kc)].v + 18, *vN = dcl[MR_CLS(kn)].v;
        double sC = acc, sN = 0.0;
#pragma unroll 1
        for (int u = 0; u < 9; ++u) {
          const int dy = u / 3;
          const double xu = MR_LDS(sb + own8 + ((dy - 1) * PITCH + (u - 3 * dy - 1)) * 8);
          sC = sC + vC[u] * xu;
          sN = sN + vN[u] * xu;
        }
        outv = sC;
        acc = sN;
      }
      // ---- stores: every lane, every iteration (sink where there is nothing to store) --------------------------------------
      {
        const bool wr = odd && live && kc >= k0 && kc < k1;
        const long long row = (long long)kc * T.Pc + ip0;
        const double o1 = 1.0 * outv + 0.0;              // (alpha = 1, beta = 0: the gather form's epilogue)
        double* q_ = wr ? a.t + row : sk;
        *q_ = o1;
        double* q2_ = (wr && a.xn) ? a.xn + row : sk;
        *q2_ = dcur * o1;
      }
      __syncthreads();
    }
  }
#undef MR_LOADRECS
#undef MR_LDS
#undef MR_READ9
#undef MR_FMA9
#undef MR_CZ
#undef MR_CLS
#undef MR_PAR
#undef MR_LOADPAIR
#undef MR_FIXPAIR
#undef MR_STAGE
}

}  // namespace mgk
