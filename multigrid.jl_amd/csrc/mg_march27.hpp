// mg_march27.hpp - z-marching pass for grid operators with up to 27 entries per row (the Galerkin coarse levels of the
// 7-point operator: GeometricTransferOperators.jl:5-36 make every coarse operator a full 3 x 3 x 3 stencil).
//
//   TWO:   t = x + d.*(b - A x)  (MGcycle.jl:129-131)   and   r = b - A t  (MGcycle.jl:58-60)   in one walk along z
//   else:  ONE product per walk:  y = x + d.*(b - A x)  (MODE1 == SMOOTH)   or   y = b - A x  (MODE1 == RESID)
//
// Why another kernel (round 4): level 2 of C2 ran the plane-tile kernel (csr_rowclass_tile_spmv: six slabs staged per four
// planes, then 27 x {16-byte record read, 4 operand reads} per lane) at 84 rows/ns - half of what the fine level's marching
// kernels get per multiply-add - and took three launches per cycle (sweep, residual, sweep) for 13 % of the rows.  Here
//  * a workgroup owns a TX x TY tile of the plane and a run of planes (one segment); x planes enter LDS once, through
//    16-byte pair loads one iteration ahead of their use (the tile kernels' staging: raw clamped loads, fixed store count);
//  * a class is ONE record of 27 values in canonical order v[dz+1][dy+1][dx+1] (0 where the class has no such entry: the
//    product adds +-0) + relaxPrec, held in the lane's registers: no dictionary walk, no offsets - the 27 operand reads
//    are LDS accesses at immediate offsets from three slab bases (the pitch is a template argument);
//  * class ids are not streamed: cls(x, y, z) = tab[cz[z]][cy[y]][cx[x]] (verified on the host, as for the tile forms);
//  * TWO: stage 1 runs on the tile + one ring, its t goes into a second ring of four slabs, stage 2 runs TWO planes behind
//    (plane z-2 needs t of z-3 .. z-1, all complete before the last barrier): one barrier per plane.
// Products in ascending (dz, dy, dx) = ascending column order, the order of the CSR row and of every other kernel here;
// same epilogue expressions: bit-identical to the launches it replaces (a zero may change its sign).
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
  int WX, SY;                   // width of the stage-1 region (TX + 2 if TWO else TX); lines of it per slot pass
  int LY, NPL;                  // lines of a slab (region + 2); 16-byte pairs per line
  int nblocks, segs, seglen;    // workgroup w = segment (w / tiles) of tile (w % tiles)
  int n_cols, ncls;
};

template <bool TWO, int MODE1, int NT, int K1, int NPM, int PITCH>
__global__ __launch_bounds__(NT) void csr_rowclass_march27_spmv(March2Args a, March27Dev T) {
  extern __shared__ double win[];
  const int tid = threadIdx.x;
  const int w = xcd_band(blockIdx.x, T.nblocks);
  constexpr int G1 = TWO ? 1 : 0;               // rings of the stage-1 region around the core
  constexpr int GX = G1 + 1;                    // halo of the staged x
  constexpr int P8 = PITCH * 8;
  const int XS = T.LY * PITCH;                  // doubles per slab (x and t slabs alike: one set of offsets)
  const int XS8 = XS * 8;
  constexpr int NSL = TWO ? 8 : 4;              // x ring [4] | t ring [4]
  char* winb = reinterpret_cast<char*>(win);
  M27Class* dcl = reinterpret_cast<M27Class*>(win + NSL * XS);
  const unsigned short* cxG = T.cmap;
  const unsigned short* cyG = cxG + T.n1;
  unsigned short* czL = reinterpret_cast<unsigned short*>(dcl + T.ncls);     // cz | tab
  unsigned short* tabL = czL + T.nplanes;
  {
    const int nw = T.ncls * (int)(sizeof(M27Class) / 8);
    const double* srcd = reinterpret_cast<const double*>(T.cls);
    double* dstd = reinterpret_cast<double*>(dcl);
    for (int i = tid; i < nw; i += NT) dstd[i] = srcd[i];
    const unsigned short* czG = cyG + T.n2;
    for (int i = tid; i < T.nplanes + T.ntab; i += NT) czL[i] = czG[i];
    for (int i = tid; i < NSL * XS; i += NT) win[i] = 0.0;        // every slab entry finite from the start
  }
  const int zstride = T.ncy * T.ncx;
  // ---- the lane's place: column xx of the stage-1 region, lines j + s*SY ------------------------------------------------
  const int xx = tid % T.WX, j = tid / T.WX;
  const bool lane_ok = j < T.SY;
  const int own8 = ((j + 1) * PITCH + xx + 1) * 8;      // byte offset of slot 0's own entry inside a slab
  const int sstride8 = T.SY * P8;                       // from slot s to slot s + 1
  int pofs[NPM], pline[NPM];
  unsigned pflag = 0u;  // per m: bit 4m = the pair exists, bit 4m+1 = first pair of its line, bit 4m+2 = its line is inside the grid
#pragma unroll
  for (int m = 0; m < NPM; ++m) {
    const int pid = tid + m * NT;
    const int l = pid / T.NPL, i = pid - l * T.NPL;
    pline[m] = l;
    pofs[m] = l * PITCH + 2 * i;
    if (pid < T.LY * T.NPL) pflag |= 1u << (4 * m);
    if (i == 0) pflag |= 2u << (4 * m);
  }
  const int ntiles = T.tiles_x * T.tiles_y;
  const int seg = w / ntiles, c = w - seg * ntiles;
  const int z0 = seg * T.seglen, z1 = z0 + T.seglen < T.nplanes ? z0 + T.seglen : T.nplanes;
  double* sk = a.sink + ((size_t)(w & 31) * NT + tid);
  // class record of the lane (registers)
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
  // acc = sum over (dz, dy, dx) ascending of v * operand; b0_, b1_, b2_: byte offsets of the row's own entry in the slabs of
  // planes z-1, z, z+1
#define M27_WALK(acc, b0_, b1_, b2_)                                                                                   \
  do {                                                                                                                 \
    _Pragma("unroll") for (int dz_ = 0; dz_ < 3; ++dz_) {                                                              \
      const int bb_ = dz_ == 0 ? (b0_) : (dz_ == 1 ? (b1_) : (b2_));                                                   \
      double xv_[9];                                                                                                   \
      _Pragma("unroll") for (int dy_ = 0; dy_ < 3; ++dy_)                                                              \
        _Pragma("unroll") for (int dx_ = 0; dx_ < 3; ++dx_)                                                            \
          xv_[dy_ * 3 + dx_] = M27_LDS(bb_ + ((dy_ - 1) * PITCH + (dx_ - 1)) * 8);                                     \
      _Pragma("unroll") for (int u_ = 0; u_ < 9; ++u_) (acc) = (acc) + rv[dz_ * 9 + u_] * xv_[u_];                     \
    }                                                                                                                  \
  } while (0)
  __syncthreads();   // dictionaries in place, slabs cleared
  if (z1 > z0) {
    const int ty = c / T.tiles_x, tx = c - ty * T.tiles_x;
    const int x0 = tx * T.TX, y0 = ty * T.TY;
    int pg[NPM];          // in-plane index of the pair's first element (before the even floor; may be negative)
#pragma unroll
    for (int m = 0; m < NPM; ++m) {
      const int yl = y0 - GX + pline[m];
      const int i2 = pofs[m] - pline[m] * PITCH;        // 2*i
      if (yl >= 0 && yl < T.n2) pflag |= 4u << (4 * m);
      pg[m] = yl * T.n1 + x0 - GX + i2;
    }
    const int gx = x0 - G1 + xx;
    const bool xin = lane_ok && gx >= 0 && gx < T.n1;
    const bool xcore = xx >= G1 && xx < T.TX + G1;
    const int ip0 = (y0 - G1 + j) * T.n1 + gx;          // in-plane index of slot 0's row; slot s: + s*SY*n1
    const int ipstride = T.SY * T.n1;
    unsigned live1 = 0u, core = 0u;                     // per slot: stage 1 is computed / the row belongs to the core tile
    int rp[K1];                                         // cy*ncx + cx of the slot's row (class = tab[cz*zstride + rp])
    const int cxo = xin ? (int)cxG[gx] : 0;
#pragma unroll
    for (int s = 0; s < K1; ++s) {
      const int yy = j + s * T.SY, gy = y0 - G1 + yy;
      const bool l1 = xin && yy < T.TY + 2 * G1 && gy >= 0 && gy < T.n2;
      live1 |= (l1 ? 1u : 0u) << s;
      core |= ((l1 && xcore && yy >= G1 && yy < T.TY + G1) ? 1u : 0u) << s;
      rp[s] = l1 ? (int)cyG[gy] * T.ncx + cxo : 0;
    }
#define M27_PAR(p, m) ((int)(((long long)(p) * T.P + pg[m]) & 1LL))
#define M27_LOADPAIR(dst, p, m)                                                                                        \
  do {                                                                                                                 \
    const bool act_ = ((pflag >> (4 * (m))) & 5u) == 5u && (p) >= 0 && (p) < T.nplanes;                                \
    const long long e0_ = ((long long)(p) * T.P + pg[m]) & ~1LL;                                                       \
    (dst) = march_load_pair_raw(a.x, e0_, act_, T.n_cols);                                                             \
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
    const int zF = TWO ? z0 - 2 : z0 - 1;               // first plane of x in the ring
    const int zS1a = TWO ? z0 - 1 : z0;                 // stage 1 runs on planes zS1a .. zS1b
    const int zS1b = TWO ? z1 : z1 - 1;
    const int zXe = zS1b + 1;                           // last plane of x needed
    const int zE = TWO ? z1 + 1 : z1 - 1;               // last iteration
    // ---- fill the ring: planes zF, zF+1, zF+2 (slot = plane & 3); plane zF+3 goes into registers ------------------------
#pragma unroll 1
    for (int pp = 0; pp < 3; ++pp) {
      d2_t q[NPM];
#pragma unroll
      for (int m = 0; m < NPM; ++m) M27_LOADPAIR(q[m], zF + pp, m);
#pragma unroll
      for (int m = 0; m < NPM; ++m) {
        M27_FIXPAIR(q[m], zF + pp, m);
        M27_STAGE((zF + pp) & 3, zF + pp, m, q[m]);
      }
    }
    d2_t preb[NPM];
#pragma unroll
    for (int m = 0; m < NPM; ++m) M27_LOADPAIR(preb[m], zF + 3, m);
    double nbb[K1];
#define M27_OPERANDS(zz)                                                                                               \
  do {                                                                                                                 \
    const bool pv_ = (zz) >= 0 && (zz) < T.nplanes;                                                                    \
    _Pragma("unroll") for (int s_ = 0; s_ < K1; ++s_) {                                                                \
      const int r_ = (pv_ && ((live1 >> s_) & 1u)) ? (zz) * T.P + ip0 + s_ * ipstride : T.n_cols - 1;                  \
      nbb[s_] = a.b[r_];                                                                                               \
    }                                                                                                                  \
  } while (0)
    M27_OPERANDS(zS1a);
    // as many stores as an iteration issues, BEHIND the loads above: the wait for those loads at the top of the loop is then
    // s_waitcnt vmcnt(number of stores) on the entry path as well as on the back edge
#pragma unroll
    for (int i = 0; i < K1 * (TWO ? 2 : 1); ++i) sk[(size_t)i * 32 * NT] = 0.0;
    __syncthreads();
    double b1[K1], b2[K1];         // TWO: b of planes z-1, z-2
#pragma unroll
    for (int s = 0; s < K1; ++s) b1[s] = b2[s] = 0.0;
#pragma unroll 1
    for (int z = zS1a; z <= zE; ++z) {
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
      // ---- x plane z+2 into its slot (that of plane z-2, last read before the previous barrier) --------------------------
      if (z + 2 <= zXe) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) {
          M27_FIXPAIR(cur[m], z + 2, m);
          M27_STAGE((z + 2) & 3, z + 2, m, cur[m]);
        }
      }
      if (z + 3 <= zXe) {
#pragma unroll
        for (int m = 0; m < NPM; ++m) M27_LOADPAIR(preb[m], z + 3, m);
      }
      if (z + 1 <= zS1b) M27_OPERANDS(z + 1);
      // ---- stage 1 on plane z ---------------------------------------------------------------------------------------------
      const bool s1 = z >= 0 && z < T.nplanes && z <= zS1b;    // (uniform)
      const int zb0 = s1 ? (int)czL[z] * zstride : 0;
      const int xb0 = ((z - 1) & 3) * XS8, xb1 = (z & 3) * XS8, xb2 = ((z + 1) & 3) * XS8;
      double tc[K1];
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        tc[s] = 0.0;
        if (s1 && ((live1 >> s) & 1u)) {
          const int o8 = own8 + s * sstride8;
          const int cq = (int)tabL[zb0 + rp[s]];
          if (cq != rcls) M27_LOADRECS(cq);
          double acc = 0.0;
          M27_WALK(acc, xb0 + o8, xb1 + o8, xb2 + o8);
          const double tv = (MODE1 == RESID && !TWO) ? b0[s] - acc : M27_LDS(xb1 + o8) + rd * (b0[s] - acc);
          if (TWO) *reinterpret_cast<double*>(winb + (4 * XS8 + (z & 3) * XS8 + o8)) = tv;
          tc[s] = tv;
        }
      }
      // ---- stage 2 on plane z-2: r = b - A t (t of planes z-3 .. z-1: written before the last barrier) ---------------------
      double st_r[K1];
      unsigned done2 = 0u;
      if (TWO) {
        const bool s2 = z - 2 >= z0 && z - 2 < z1;        // (uniform)
        const int zb2 = s2 ? (int)czL[z - 2] * zstride : 0;
        const int tb0 = 4 * XS8 + ((z - 3) & 3) * XS8, tb1 = 4 * XS8 + ((z - 2) & 3) * XS8, tb2 = 4 * XS8 + ((z - 1) & 3) * XS8;
#pragma unroll
        for (int s = 0; s < K1; ++s) {
          st_r[s] = 0.0;
          if (s2 && ((core >> s) & 1u)) {
            const int o8 = own8 + s * sstride8;
            const int cq = (int)tabL[zb2 + rp[s]];
            if (cq != rcls) M27_LOADRECS(cq);
            double acc = 0.0;
            M27_WALK(acc, tb0 + o8, tb1 + o8, tb2 + o8);
            st_r[s] = b2[s] - acc;
            done2 |= 1u << s;
          }
        }
      }
      // ---- the stores of this iteration: every lane issues every store instruction (sink for lanes / planes with nothing) ----
      {
        const bool wt = s1 && z >= z0 && z < z1;          // (uniform) plane z belongs to this run: its t (or y) is stored
#pragma unroll
        for (int s = 0; s < K1; ++s) {
          const int rowt = z * T.P + ip0 + s * ipstride;
          double* q_ = (wt && ((core >> s) & 1u)) ? a.t + rowt : sk;
          *q_ = tc[s];
          if (TWO) {
            double* q2_ = ((done2 >> s) & 1u) ? a.r + (rowt - 2 * T.P) : sk;
            *q2_ = st_r[s];
          }
        }
      }
#pragma unroll
      for (int s = 0; s < K1; ++s) {
        b2[s] = b1[s];
        b1[s] = b0[s];
      }
      __syncthreads();
    }
  }
#undef M27_LOADRECS
#undef M27_LDS
#undef M27_WALK
#undef M27_PAR
#undef M27_LOADPAIR
#undef M27_FIXPAIR
#undef M27_STAGE
#undef M27_OPERANDS
}

}  // namespace mgk
