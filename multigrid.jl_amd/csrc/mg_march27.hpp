// mg_march27.hpp - z-marching pass for grid operators with up to 27 entries per row (the Galerkin coarse levels of the
// 7-point operator: GeometricTransferOperators.jl:5-36 make every coarse operator a full 3 x 3 x 3 stencil).
//
//   TWO:   t = x + d.*(b - A x)  (MGcycle.jl:129-131)   and   r = b - A t  (MGcycle.jl:58-60)   in one walk along z
//   else:  ONE product per walk:  y = x + d.*(b - A x)  (MODE1 == SMOOTH)   or   y = b - A x  (MODE1 == RESID)
//   FZ (with TWO; round 6): the level is entered with x = 0 (MGcycle.jl:29: `norm(x) > 0` fails, r = b), so the first update is
//          x1 = d.*b and the walk is x1 = d.*b ; t = x1 + d.*(b - A x1) ; r = b - A t  - three stages, x is NOT read: the staging
//          loads take b and every staged entry is multiplied by the relaxPrec of ITS row's class (two LDS look-ups per entry
//          and plane); the restriction that produced b need not write x1.  Same products as the d.*b launch: same bits.
//
// Why another kernel (round 4): level 2 of C2 ran the plane-tile kernel (csr_rowclass_tile_spmv: six slabs staged per four
// planes, then 27 x {16-byte record read, 4 operand reads} per lane) at 84 rows/ns - half of what the fine level's marching
// kernels get per multiply-add - and took three launches per cycle (sweep, residual, sweep) for 13 % of the rows.  Here
//  * a workgroup owns a TX x TY tile of the plane and a run of planes (one segment); x planes enter LDS once, through
//    16-byte pair loads one iteration ahead of their use (the tile kernels' staging: raw clamped loads, fixed store count);
//  * every plane in LDS is read ONCE, nine operands per row: the nine values of plane p feed the three output planes they
//    belong to - q = p+1 (its dz = -1 entries), q = p (dz = 0), q = p-1 (dz = +1, which completes it) - through three
//    accumulators per row in registers.  A first version that read 27 operands per row and stage was bound by the LDS
//    pipe (7.8 cycles per wavefront read; pair 47 us against 49.5 for the two plane-tile launches: profiles/r04_march27_ab.md);
//  * a class is ONE record of 27 values in canonical order v[dz+1][dy+1][dx+1] (0 where the class has no such entry: the
//    product adds +-0) + relaxPrec; a lane keeps the block of nine it needs per output plane in registers and re-reads a
//    block only when that plane's z-class differs (first / last planes): no dictionary walk, no offsets - the operand
//    reads are LDS accesses at immediate offsets from the lane's own entry (the pitch is a template argument);
//  * class ids are not streamed: cls(x, y, z) = tab[cz[z]][cy[y]][cx[x]] (verified on the host, as for the tile forms);
//  * TWO: stage 1 runs on the tile + one ring, its t goes into a second ring of two slabs, stage 2 reads plane z-1 of t while
//    stage 1 completes plane z: r of plane q is complete in iteration q + 2; one barrier per plane.
// Per output row the products are added in ascending (dz, dy, dx) = ascending column order, the order of the CSR row and
// of every other kernel here, with the same epilogue expressions: bit-identical to the launches it replaces (a zero may
// change its sign).  One row per lane.
#pragma once

namespace mgk {

struct M27Class {               // 224 bytes per class
  double v[27];                 // v[(dz+1)*9 + (dy+1)*3 + (dx+1)]
  double d;                     // relaxPrec of the class
};
struct March27Dev {
  const M27Class* cls;          // [ncls]
  const unsigned short* cmap;   // cx[n1] | cy[n2] | cz[nplanes] | tab[ncz*ncy*ncx]
  int ncx, ncy, ncz, ntab;
  int n1, n2, nplanes, P;
  int TX, TY, tiles_x, tiles_y; // core tile
  int WX, SY;                   // stage-1 region: TX + 2 (TWO) or TX columns, SY = TY + 2 or TY lines; WX*SY <= threads
  int LY, NPL;                  // lines of a slab (SY + 2); 16-byte pairs per line
  int nblocks, segs, seglen;    // workgroup w = segment (w / tiles) of tile (w % tiles)
  int n_cols, ncls;
};

template <bool TWO, int MODE1, int NT, int NPM, int PITCH, bool FZ = false>
__global__ __launch_bounds__(NT, 4) void csr_rowclass_march27_spmv(March2Args a, March27Dev T) {
  static_assert(!FZ || (TWO && MODE1 == SMOOTH), "the from-zero form is the pair's");
  extern __shared__ double win[];
  const int tid = threadIdx.x;
  const int w = xcd_band(blockIdx.x, T.nblocks);
  constexpr int G1 = TWO ? 1 : 0;               // rings of the stage-1 region around the core
  constexpr int GX = G1 + 1;                    // halo of the staged x
  const int XS = T.LY * PITCH;                  // doubles per slab (x and t slabs alike: one set of offsets)
  const int XS8 = XS * 8;
  constexpr int NSL = TWO ? 4 : 2;              // x ring [2] | t ring [2]
  char* winb = reinterpret_cast<char*>(win);
  M27Class* dcl = reinterpret_cast<M27Class*>(win + NSL * XS);
  const unsigned short* cxG = T.cmap;
  const unsigned short* cyG = cxG + T.n1;
  const unsigned short* czG = cyG + T.n2;
  unsigned short* czL = reinterpret_cast<unsigned short*>(dcl + T.ncls);     // cz | tab
  unsigned short* tabL = czL + T.nplanes;
  const int zstride = T.ncy * T.ncx;
  const int ntiles = T.tiles_x * T.tiles_y;
  const int seg = w / ntiles, c = w - seg * ntiles;
  const int z0 = seg * T.seglen, z1 = z0 + T.seglen < T.nplanes ? z0 + T.seglen : T.nplanes;
  if (z1 <= z0) return;                         // (uniform: the last segments of a grid the segment length does not divide)
  // ---- the lane's row: column xx, line j of the stage-1 region --------------------------------------------------------
  const int xx = tid % T.WX, j = tid / T.WX;
  const bool lane_ok = j < T.SY;
  const int own8 = (((lane_ok ? j : 0) + 1) * PITCH + xx + 1) * 8;   // byte offset of the row's own entry inside a slab (lanes beyond the region: a valid one)
  const int ty = c / T.tiles_x, tx = c - ty * T.tiles_x;
  const int x0 = tx * T.TX, y0 = ty * T.TY;
  int pofs[NPM], pg[NPM];   // element index of the lane's pair inside a slab; in-plane index of its first element (before the even floor; may be negative)
  unsigned pflag = 0u;      // per m: bit 4m = the pair exists, bit 4m+1 = first pair of its line, bit 4m+2 = its line is inside the grid
  int prp[FZ ? NPM : 1][3]; // FZ: cy*ncx + cx of the columns xb-1, xb, xb+1 of the pair's line (xb = its first column before the even floor), clamped into the grid
#pragma unroll
  for (int m = 0; m < NPM; ++m) {
    const int pid = tid + m * NT;
    const int l = pid / T.NPL, i = pid - l * T.NPL;
    const int yl = y0 - GX + l;
    pofs[m] = l * PITCH + 2 * i;
    pg[m] = yl * T.n1 + x0 - GX + 2 * i;
    if constexpr (FZ) {
      const int yc = yl < 0 ? 0 : (yl >= T.n2 ? T.n2 - 1 : yl);
      const int cyl = (int)cyG[yc] * T.ncx;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int xg = x0 - GX + 2 * i - 1 + k;
        prp[m][k] = cyl + (int)cxG[xg < 0 ? 0 : (xg >= T.n1 ? T.n1 - 1 : xg)];
      }
    }
    if (pid < T.LY * T.NPL) pflag |= 1u << (4 * m);
    if (i == 0) pflag |= 2u << (4 * m);
    if (yl >= 0 && yl < T.n2) pflag |= 4u << (4 * m);
  }
  const int gx = x0 - G1 + xx, gy = y0 - G1 + j;
  const bool live1 = lane_ok && gx >= 0 && gx < T.n1 && gy >= 0 && gy < T.n2;             // stage 1 is computed on the row
  const bool core = live1 && xx >= G1 && xx < T.TX + G1 && j >= G1 && j < T.TY + G1;       // the row belongs to the core tile
  const int ip0 = gy * T.n1 + gx;                       // in-plane index of the row
  double* sk = a.sink + ((size_t)(w & 31) * NT + tid);
  // the lane's class record (registers): away from the first / last planes the three output planes an operand plane feeds
  // share the row's class - ONE record serves every block of both stages
  double rv[27], rd = 0.0;
  int rcls = -1;
#pragma unroll
  for (int u = 0; u < 27; ++u) rv[u] = 0.0;
#define M27_LOADRECS(cq)                                                                                               \
  do {                                                                                                                 \
    const M27Class* q_ = dcl + (cq);                                                                                   \
    _Pragma("unroll") for (int u_ = 0; u_ < 27; ++u_) rv[u_] = q_->v[u_];                                              \
    rd = q_->d;                                                                                                        \
    rcls = (cq);                                                                                                       \
  } while (0)
#define M27_LDS(off8) (*reinterpret_cast<const double*>(winb + (off8)))
  // the nine operands of the row in the plane whose slab starts at byte sb_
#define M27_READ9(xv_, sb_)                                                                                            \
  _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_)                                                                  \
    _Pragma("unroll") for (int dx_ = 0; dx_ < 3; ++dx_)                                                                \
      (xv_)[dy_ * 3 + dx_] = M27_LDS((sb_) + own8 + ((dy_ - 1) * PITCH + (dx_ - 1)) * 8)
#define M27_FMA9(acc, blk, xv_) _Pragma("unroll") for (int u_ = 0; u_ < 9; ++u_) (acc) = (acc) + rv[(blk) * 9 + u_] * (xv_)[u_]
// (first / last planes: the z-classes of the three output planes an operand plane feeds differ) the same three updates with
// every value and operand read from LDS in a rolled loop: slow, a handful of iterations per launch, and no registers taken
// from the marching loop.  cC_ / cM_ / cN_: classes of the completing / middle / new output plane; sb_: the operand slab
#define M27_SLOW(aC_, aM_, aN_, sb_, cC_, cM_, cN_)                                                                    \
  do {                                                                                                                 \
    const double *vC_ = dcl[(cC_)].v + 18, *vM_ = dcl[(cM_)].v + 9, *vN_ = dcl[(cN_)].v;                               \
    double sC_ = (aC_), sM_ = (aM_), sN_ = 0.0;                                                                        \
    _Pragma("unroll 1") for (int u_ = 0; u_ < 9; ++u_) {                                                               \
      const int dy_ = u_ / 3;                                                                                          \
      const double xu_ = M27_LDS((sb_) + own8 + ((dy_ - 1) * PITCH + (u_ - 3 * dy_ - 1)) * 8);                         \
      sC_ = sC_ + vC_[u_] * xu_;                                                                                       \
      sM_ = sM_ + vM_[u_] * xu_;                                                                                       \
      sN_ = sN_ + vN_[u_] * xu_;                                                                                       \
    }                                                                                                                  \
    (aC_) = sC_;                                                                                                       \
    (aM_) = sM_;                                                                                                       \
    (aN_) = sN_;                                                                                                       \
  } while (0)
#define M27_CZ(q) ((int)czL[(q) < 0 ? 0 : ((q) >= T.nplanes ? T.nplanes - 1 : (q))])
#define M27_CLS(q) ((int)tabL[M27_CZ(q) * zstride + rp])
#define M27_PAR(p, m) ((int)(((long long)(p) * T.P + pg[m]) & 1LL))
#define M27_LOADPAIR(dst, p, m)                                                                                        \
  do {                                                                                                                 \
    const bool act_ = ((pflag >> (4 * (m))) & 5u) == 5u && (p) >= 0 && (p) < T.nplanes;                                \
    const long long e0_ = ((long long)(p) * T.P + pg[m]) & ~1LL;                                                       \
    (dst) = march_load_pair_raw(FZ ? a.b : a.x, e0_, act_, T.n_cols);                                                  \
  } while (0)
  // FZ: the pair holds b of the entries (line, xb - par) and (line, xb - par + 1) of plane p: make it x1 = d.*b
#define M27_SCALE(v, p, m)                                                                                             \
  do {                                                                                                                 \
    if constexpr (FZ) {                                                                                                \
      const int par_ = M27_PAR(p, m);                                                                                  \
      const int zb_ = M27_CZ(p) * zstride;                                                                             \
      const int k0_ = par_ ? prp[m][0] : prp[m][1], k1_ = par_ ? prp[m][1] : prp[m][2];                                \
      (v).x = dcl[(int)tabL[zb_ + k0_]].d * (v).x;                                                                     \
      (v).y = dcl[(int)tabL[zb_ + k1_]].d * (v).y;                                                                     \
    }                                                                                                                  \
  } while (0)
#define M27_FIXPAIR(v, p, m)                                                                                           \
  do {                                                                                                                 \
    if ((p) == T.nplanes - 1) {                                                                      /* (uniform) */   \
      const bool act_ = ((pflag >> (4 * (m))) & 5u) == 5u;                                                             \
      const long long e0_ = ((long long)(p) * T.P + pg[m]) & ~1LL;                                                     \
      march_pair_fix((v), e0_, act_, T.n_cols);                                                                        \
    }                                                                                                                  \
  } while (0)
  // entry k of a slab line = in-plane index (line start) + k: a leading entry of an odd line start is dropped
#define M27_STAGE(slot, p, m, v)                                                                                       \
  do {                                                                                                                 \
    if ((pflag >> (4 * (m))) & 1u) {                                                                                   \
      const int par_ = M27_PAR(p, m);                                                                                  \
      double* q_ = win + ((slot) * XS + pofs[m] - par_);                                                               \
      if (!(par_ && ((pflag >> (4 * (m))) & 2u))) q_[0] = (v).x;                                                       \
      q_[1] = (v).y;                                                                                                   \
    }                                                                                                                  \
  } while (0)
#define M27_OPERANDS(zz)                                                                                               \
  do {                                                                                                                 \
    const bool pv_ = (zz) >= 0 && (zz) < T.nplanes && live1;                                                           \
    nbb = a.b[pv_ ? (zz) * T.P + ip0 : T.n_cols - 1];                                                                  \
  } while (0)
  {
    // Output planes: stage 1 on q1a .. q1b (TWO: one plane beyond the run on both sides, for stage 2), stage 2 on z0 .. z1-1.
    // Iteration z reads x plane z+1 and completes stage 1 of plane z; it reads t plane z-1 and completes stage 2 of plane z-2.
    const int q1a = TWO ? z0 - 1 : z0, q1b = TWO ? z1 : z1 - 1;
    const int zA = q1a - 2;                             // first iteration (it reads x plane q1a - 1)
    const int zE = TWO ? z1 + 1 : z1 - 1;               // last iteration
    const int zXe = q1b + 1;                            // last plane of x needed
    // ---- prologue: EVERY global load it needs goes out before the first wait (one trip to memory instead of four in a row:
    // measured, a launch of the first version spent 10-13 us before its first plane): x planes zA+1 (for its slab) and zA+2
    // (stays in registers), b of the first plane, the row's x / y class indices, this lane's share of the dictionaries
    d2_t q0[NPM], preb[NPM];
#pragma unroll
    for (int m = 0; m < NPM; ++m) M27_LOADPAIR(q0[m], zA + 1, m);
#pragma unroll
    for (int m = 0; m < NPM; ++m) M27_LOADPAIR(preb[m], zA + 2, m);
    double nbb;
    M27_OPERANDS(zA);
    const int cyv = (int)cyG[live1 ? gy : 0], cxv = (int)cxG[live1 ? gx : 0];
    constexpr int ND = 4;                               // dictionary doubles per lane in registers (the rest, if any, in a loop)
    const int nw = T.ncls * (int)(sizeof(M27Class) / 8), nm = T.nplanes + T.ntab;
    const double* srcd = reinterpret_cast<const double*>(T.cls);
    double dreg[ND];
#pragma unroll
    for (int u = 0; u < ND; ++u) dreg[u] = srcd[tid + u * NT < nw ? tid + u * NT : 0];
    const unsigned short creg = czG[tid < nm ? tid : 0];
    for (int i = tid; i < NSL * XS; i += NT) win[i] = 0.0;        // every slab entry finite from the start
    __syncthreads();                                              // (slabs cleared before anything is staged into them)
    {
      double* dstd = reinterpret_cast<double*>(dcl);
#pragma unroll
      for (int u = 0; u < ND; ++u)
        if (tid + u * NT < nw) dstd[tid + u * NT] = dreg[u];
      for (int i = tid + ND * NT; i < nw; i += NT) dstd[i] = srcd[i];
      if (tid < nm) czL[tid] = creg;
      for (int i = tid + NT; i < nm; i += NT) czL[i] = czG[i];
    }
    if (FZ) __syncthreads();                                      // (the first staged plane is scaled through the dictionaries)
#pragma unroll
    for (int m = 0; m < NPM; ++m) {
      M27_FIXPAIR(q0[m], zA + 1, m);
      M27_SCALE(q0[m], zA + 1, m);
      M27_STAGE((zA + 1) & 1, zA + 1, m, q0[m]);
    }
    const int rp = live1 ? cyv * T.ncx + cxv : 0;       // class of the row in plane z = tab[cz[z]*zstride + rp]
    // as many stores as an iteration issues, BEHIND the loads above: the wait for those loads at the top of the loop is then
    // s_waitcnt vmcnt(number of stores) on the entry path as well as on the back edge
#pragma unroll
    for (int i = 0; i < (TWO ? 2 : 1); ++i) sk[(size_t)i * 32 * NT] = 0.0;
    __syncthreads();
    double am = 0.0, an = 0.0;     // stage 1: accumulators of output planes z+1 (dz = -1, 0 added), z+2 (dz = -1 added)
    double bm = 0.0, bn = 0.0;     // stage 2: accumulators of output planes z-1, z
    double xo1 = 0.0;              // the row's own x of plane z (read with plane z in the previous iteration)
    double b1 = 0.0, b2 = 0.0;     // b of planes z-1, z-2
#pragma unroll 1
    for (int z = zA; z <= zE; ++z) {
      d2_t cur[NPM];
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        cur[m] = preb[m];
        asm volatile("" : "+v"(cur[m].x), "+v"(cur[m].y));     // the wait of this iteration: the loads, not the stores behind them
      }
      double b0 = nbb;
      asm volatile("" : "+v"(b0));
      // ---- x plane z+2 into its slot (that of plane z, read before the previous barrier) -----------------------------------
      if (z + 2 <= zXe) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) {
          M27_FIXPAIR(cur[m], z + 2, m);
          M27_SCALE(cur[m], z + 2, m);
          M27_STAGE((z + 2) & 1, z + 2, m, cur[m]);
        }
      }
      if (z + 3 <= zXe) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) M27_LOADPAIR(preb[m], z + 3, m);
      }
      if (z + 1 <= q1b) M27_OPERANDS(z + 1);
      // ---- stage 1: x plane z+1 feeds output planes z+2, z+1, z; plane z is complete ---------------------------------------
      // (uniform) do the planes this iteration touches - z-2 .. z+2 - share one z-class?  Then the lane's record serves them all
      const int zc_ = M27_CZ(z);
      const bool fast = M27_CZ(z + 2) == zc_ && M27_CZ(z + 1) == zc_ && (!TWO || (M27_CZ(z - 1) == zc_ && M27_CZ(z - 2) == zc_));
      if (fast) {
        const int cq = (int)tabL[zc_ * zstride + rp];
        if (cq != rcls) M27_LOADRECS(cq);
      }
      double tv = 0.0;
      {
        const int sb = ((z + 1) & 1) * XS8;
        double ac = am, dq = rd, xo1n;
        if (fast) {
          double xv[9];
          M27_READ9(xv, sb);
          M27_FMA9(ac, 2, xv);
          M27_FMA9(an, 1, xv);
          am = an;
          an = 0.0;
          M27_FMA9(an, 0, xv);
          xo1n = xv[4];
        } else {
          const int cC = M27_CLS(z);
          double aM = an, aN = 0.0;
          M27_SLOW(ac, aM, aN, sb, cC, M27_CLS(z + 1), M27_CLS(z + 2));
          am = aM;
          an = aN;
          dq = dcl[cC].d;
          xo1n = M27_LDS(sb + own8);
        }
        tv = (MODE1 == RESID && !TWO) ? b0 - ac : xo1 + dq * (b0 - ac);
        xo1 = xo1n;
        const bool s1 = live1 && z >= q1a && z <= q1b && z >= 0 && z < T.nplanes;
        if (TWO && s1) *reinterpret_cast<double*>(winb + (2 * XS8 + (z & 1) * XS8 + own8)) = tv;
      }
      // ---- stage 2: t plane z-1 feeds output planes z, z-1, z-2; plane z-2 is complete -------------------------------------
      double rr = 0.0;
      if (TWO) {
        const int sb = 2 * XS8 + ((z - 1) & 1) * XS8;
        double bc = bm;
        if (fast) {
          double tvv[9];
          M27_READ9(tvv, sb);
          M27_FMA9(bc, 2, tvv);
          M27_FMA9(bn, 1, tvv);
          bm = bn;
          bn = 0.0;
          M27_FMA9(bn, 0, tvv);
        } else {
          double aM = bn, aN = 0.0;
          M27_SLOW(bc, aM, aN, sb, M27_CLS(z - 2), M27_CLS(z - 1), M27_CLS(z));
          bm = aM;
          bn = aN;
        }
        rr = b2 - bc;
      }
      // ---- the stores of this iteration: every lane issues every store instruction (sink for lanes / planes with nothing) ----
      {
        const bool wt = core && z >= z0 && z < z1;
        double* q_ = wt ? a.t + ((long long)z * T.P + ip0) : sk;
        *q_ = tv;
        if (TWO) {
          const bool wr = core && z - 2 >= z0 && z - 2 < z1;
          double* q2_ = wr ? a.r + ((long long)(z - 2) * T.P + ip0) : sk;
          *q2_ = rr;
        }
      }
      b2 = b1;
      b1 = b0;
      __syncthreads();
    }
  }
#undef M27_LOADRECS
#undef M27_SLOW
#undef M27_CZ
#undef M27_LDS
#undef M27_READ9
#undef M27_FMA9
#undef M27_CLS
#undef M27_PAR
#undef M27_LOADPAIR
#undef M27_FIXPAIR
#undef M27_SCALE
#undef M27_STAGE
#undef M27_OPERANDS
}

}  // namespace mgk
