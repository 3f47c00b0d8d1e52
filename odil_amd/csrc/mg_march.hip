// z-marching multigrid transfers for the layouts whose last three axes are cells: 'ccc' (the
// Poisson hot path), 'ncc' (node-centred marching axis: space-time fields with two space
// dimensions) and the 4-D '.ccc' / 'nccc' (batch or node-centred leading axis).
// Same arithmetic as mg_fast.hip / mg_transfer.hip; what changes is the data movement:
//
// P  : a thread owns CX adjacent coarse columns (jy, jx0 .. jx0+CX-1) and walks coarse planes,
//      holding the 3 x 3 x (CX+2) coarse neighbourhood in registers; every step emits the fine
//      planes 2jz, 2jz+1 as 16 B stores.  Fine traffic is touched exactly once.
// P^T: a thread owns the same columns; each FINE plane is reduced over its (y, x) window once
//      (rows as three packs) and the per-plane sums slide through a 6-entry register
//      window, so every fine value is loaded once per owner.
//
// CX = 1 for double; P in float uses CX = 2 so that every access of the fine array is 16 B per lane
// (256^3 -> 512^3: 3.0 -> 4.0 TB/s).  P^T stays at CX = 1: its register footprint (two windows of
// plane sums, three weight tables) halves the occupancy at CX = 2 and loses more than the wide
// loads gain (measured).
#include "mg_march.h"
#include "poisson.h"

namespace odil {

// s[sy][sx] += wz * sum_{ry, rx} wy wx v[sy+ry][...]: one z (or leading) tap of the 2 x 2CX fine
// outputs of a plane, in the reference's order (ry, rx), rx fastest; per axis the weight is
// parity == r ? 1 : 3.
template <typename T, int CX>
__device__ inline void acc_plane(T (&s)[2][2 * CX], const T (&v)[3][CX + 2], int wz) {
#pragma unroll
  for (int sy = 0; sy < 2; ++sy)
#pragma unroll
    for (int sx = 0; sx < 2 * CX; ++sx) {
      T t = s[sy][sx];
#pragma unroll
      for (int ry = 0; ry < 2; ++ry)
#pragma unroll
        for (int rx = 0; rx < 2; ++rx) {
          const int w = wz * (sy == ry ? 1 : 3) * ((sx & 1) == rx ? 1 : 3);
          t = t + T(w) * v[sy + ry][(sx >> 1) + (sx & 1) + rx];
        }
      s[sy][sx] = t;
    }
}

template <typename T, int CX>
__device__ inline void zero_plane(T (&s)[2][2 * CX]) {
#pragma unroll
  for (int sy = 0; sy < 2; ++sy)
#pragma unroll
    for (int sx = 0; sx < 2 * CX; ++sx) s[sy][sx] = T(0);
}

// fine rows 2jy, 2jy+1 of one fine plane: scale, add the fine-level term, 16 B stores
template <typename T, int CX>
__device__ inline void store_plane(T* __restrict__ fine, int64_t off, int fnx, const T (&s)[2][2 * CX], T rs,
                                   const PackN<T, 2 * CX> (&ad)[2], bool has_add, T ascale, bool nt) {
#pragma unroll
  for (int sy = 0; sy < 2; ++sy) {
    PackN<T, 2 * CX> pk;
#pragma unroll
    for (int sx = 0; sx < 2 * CX; ++sx) {
      T o = s[sy][sx] * rs;
      if (has_add) o = ascale * ad[sy].e[sx] + o;
      pk.e[sx] = o;
    }
    stream_st<T, 2 * CX>(fine + off + (int64_t)sy * fnx, pk, nt);
  }
}

template <typename T, int CX>
__device__ inline void load_add(const T* __restrict__ add, int64_t off, int fnx, PackN<T, 2 * CX> (&ad)[2], bool nt) {
#pragma unroll
  for (int sy = 0; sy < 2; ++sy) ad[sy] = stream_ld<T, 2 * CX>(add + off + (int64_t)sy * fnx, nt);
}

template <typename T, int CX>
__device__ inline void shift_window(T (&v)[3][3][CX + 2]) {
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < CX + 2; ++dx) {
      v[0][dy][dx] = v[1][dy][dx];
      v[1][dy][dx] = v[2][dy][dx];
    }
}

// Decodes the unit and the owned columns; false when this thread has nothing to do.
template <int CX>
__device__ inline bool march_decode(const MarchArgs& a, int& z0, int& z1, int& jy, int& jx0) {
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return false;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  jy = yt * a.ty + ly;
  jx0 = (xt * a.tx + lx) * CX;
  if (jy >= a.cn[1] || jx0 >= a.cn[2]) return false;
  z0 = zc * a.usched.ZC;
  z1 = z0 + a.usched.ZC < a.cn[0] ? z0 + a.usched.ZC : a.cn[0];
  return true;
}

// ------------------------------------------------------------------------------------
// P, all-cell layout 'ccc'
// ------------------------------------------------------------------------------------
template <typename T, int CX>
__global__ __launch_bounds__(kBlock) void k_interp_add_march(const T* __restrict__ coarse, const T* __restrict__ add,
                                                             T* __restrict__ fine, MarchArgs a, T cscale, T ascale) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int z0, z1, jy, jx0;
  if (!march_decode<CX>(a, z0, z1, jy, jx0)) return;
  const TapN<CX + 2> tx = tapn<CX + 2>(jx0, cnx);
  const TapN<3> ty = tapn<3>(jy, cny);
  T v[3][3][CX + 2];
  load_plane_shared<T, CX>(coarse, z0 - 1, cnz, cplane, cnx, ty, tx, cscale, v[0], a.tx, jx0);
  load_plane_shared<T, CX>(coarse, z0, cnz, cplane, cnx, ty, tx, cscale, v[1], a.tx, jx0);
  const T r64 = T(1) / T(64);  // exact: sum of weights 4*4*4
  for (int jz = z0; jz < z1; ++jz) {
    // issue the fine-grid addend loads first: they are the HBM stream of this kernel
    PackN<T, 2 * CX> ad[2][2];
    const int64_t fbase = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx0;
    if (add) {
      load_add<T, CX>(add, fbase, fnx, ad[0], a.nt);
      load_add<T, CX>(add, fbase + fplane, fnx, ad[1], a.nt);
    }
    load_plane_shared<T, CX>(coarse, jz + 1, cnz, cplane, cnx, ty, tx, cscale, v[2], a.tx, jx0);
#pragma unroll
    for (int sz = 0; sz < 2; ++sz) {
      T s[2][2 * CX];
      zero_plane<T, CX>(s);
#pragma unroll
      for (int rz = 0; rz < 2; ++rz) acc_plane<T, CX>(s, v[sz + rz], sz == rz ? 1 : 3);
      store_plane<T, CX>(fine, fbase + sz * fplane, fnx, s, r64, ad[sz], add != nullptr, ascale, a.nt);
    }
    shift_window<T, CX>(v);
  }
}

// ------------------------------------------------------------------------------------
// P, NODE-centred marching axis ('ncc': time-like axis of the space-time workloads): fine plane
// 2jz is coarse plane jz interpolated in (y, x) only, fine plane 2jz+1 the mean of the two
// neighbouring coarse planes; ghosts exist on the two cell axes only.
// ------------------------------------------------------------------------------------
template <typename T, int CX>
__global__ __launch_bounds__(kBlock) void k_interp_add_march_n(const T* __restrict__ coarse,
                                                               const T* __restrict__ add, T* __restrict__ fine,
                                                               MarchArgs a, T cscale, T ascale) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int z0, z1, jy, jx0;
  if (!march_decode<CX>(a, z0, z1, jy, jx0)) return;
  const TapN<CX + 2> tx = tapn<CX + 2>(jx0, cnx);
  const TapN<3> ty = tapn<3>(jy, cny);
  T v[2][3][CX + 2];
  load_plane_shared<T, CX>(coarse, z0, cnz, cplane, cnx, ty, tx, cscale, v[0], a.tx, jx0);
  const T r16 = T(1) / T(16), r32 = T(1) / T(32);
  for (int jz = z0; jz < z1; ++jz) {
    const bool odd = jz + 1 < cnz;  // the last coarse plane has no fine plane above it
    PackN<T, 2 * CX> ad[2][2];
    const int64_t fbase = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx0;
    if (add) {
      load_add<T, CX>(add, fbase, fnx, ad[0], a.nt);
      if (odd) load_add<T, CX>(add, fbase + fplane, fnx, ad[1], a.nt);
    }
    if (odd) load_plane_shared<T, CX>(coarse, jz + 1, cnz, cplane, cnx, ty, tx, cscale, v[1], a.tx, jx0);
    T s[2][2 * CX];
    zero_plane<T, CX>(s);
    acc_plane<T, CX>(s, v[0], 1);
    store_plane<T, CX>(fine, fbase, fnx, s, r16, ad[0], add != nullptr, ascale, a.nt);
    if (odd) {
      zero_plane<T, CX>(s);
      acc_plane<T, CX>(s, v[0], 1);
      acc_plane<T, CX>(s, v[1], 1);
      store_plane<T, CX>(fine, fbase + fplane, fnx, s, r32, ad[1], add != nullptr, ascale, a.nt);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < CX + 2; ++dx) v[0][dy][dx] = v[1][dy][dx];
  }
}

// ------------------------------------------------------------------------------------
// P, 4-D layouts ('.ccc' batches and 'nccc': space-time fields with three space dimensions): the
// walk of k_interp_add_march per FINE leading index f0 (blockIdx.y).  On a node-centred leading
// axis an odd f0 averages the two neighbouring coarse volumes: both windows are held and summed in
// the reference's order (leading tap outermost).
// ------------------------------------------------------------------------------------
// CNT0 = coarse volumes per fine leading index, a compile-time constant so that the single-volume
// launches (batches, even fine indices of a node axis) carry one window and keep the occupancy of the
// 3-D kernel; the fine leading index is f0 = f0_first + f0_step * blockIdx.y.
template <typename T, int CX, int CNT0>
__global__ __launch_bounds__(kBlock) void k_interp_add_march_lead(const T* __restrict__ coarse,
                                                                  const T* __restrict__ add, T* __restrict__ fine,
                                                                  MarchArgs a, T cscale, T ascale, int f0_first,
                                                                  int f0_step) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int64_t cvol = a.lead_cstride, fvol = (int64_t)a.fn[0] * fplane;
  const int f0 = f0_first + f0_step * (int)blockIdx.y;
  const bool node = a.lead_loc == kNode;
  constexpr int cnt0 = CNT0;
  const int c0 = node ? f0 >> 1 : f0;
  int z0, z1, jy, jx0;
  if (!march_decode<CX>(a, z0, z1, jy, jx0)) return;
  const TapN<CX + 2> tx = tapn<CX + 2>(jx0, cnx);
  const TapN<3> ty = tapn<3>(jy, cny);
  const T* cb[2] = {coarse + c0 * cvol, coarse + (c0 + cnt0 - 1) * cvol};
  add = add ? add + f0 * fvol : add;
  fine += f0 * fvol;
  T v[CNT0][3][3][CX + 2];
  auto load = [&](int q, int slot) {
#pragma unroll
    for (int r0 = 0; r0 < CNT0; ++r0)
      load_plane_shared<T, CX>(cb[r0], q, cnz, cplane, cnx, ty, tx, cscale, v[r0][slot], a.tx, jx0);
  };
  load(z0 - 1, 0);
  load(z0, 1);
  const T rs = T(1) / T(64 * cnt0);
  for (int jz = z0; jz < z1; ++jz) {
    PackN<T, 2 * CX> ad[2][2];
    const int64_t fbase = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx0;
    if (add) {
      load_add<T, CX>(add, fbase, fnx, ad[0], a.nt);
      load_add<T, CX>(add, fbase + fplane, fnx, ad[1], a.nt);
    }
    load(jz + 1, 2);
#pragma unroll
    for (int sz = 0; sz < 2; ++sz) {
      T s[2][2 * CX];
      zero_plane<T, CX>(s);
#pragma unroll
      for (int r0 = 0; r0 < CNT0; ++r0)
#pragma unroll
        for (int rz = 0; rz < 2; ++rz) acc_plane<T, CX>(s, v[r0][sz + rz], sz == rz ? 1 : 3);
      store_plane<T, CX>(fine, fbase + sz * fplane, fnx, s, rs, ad[sz], add != nullptr, ascale, a.nt);
    }
#pragma unroll
    for (int r0 = 0; r0 < CNT0; ++r0) shift_window<T, CX>(v[r0]);
  }
}

#ifndef ODIL_PAIR_PREFETCH
#define ODIL_PAIR_PREFETCH 1
#endif
// Node-centred leading axis, fine indices 2k AND 2k + 1 by one thread (blockIdx.y = k < lead_cn - 1): the window of
// coarse volume k serves both, the window of volume k + 1 the odd one.  Same sums in the same order as the two
// launches of k_interp_add_march_lead; the registers of the two-window launch (two waves per SIMD either way), but
// twice the fine bytes in flight per wave and one pass less over the coarse array.
template <typename T, int CX>
__global__ __launch_bounds__(kBlock) void k_interp_add_march_lead_pair(const T* __restrict__ coarse,
                                                                       const T* __restrict__ add, T* __restrict__ fine,
                                                                       MarchArgs a, T cscale, T ascale) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int64_t cvol = a.lead_cstride, fvol = (int64_t)a.fn[0] * fplane;
  const int c0 = (int)blockIdx.y;
  int z0, z1, jy, jx0;
  if (!march_decode<CX>(a, z0, z1, jy, jx0)) return;
  const TapN<CX + 2> tx = tapn<CX + 2>(jx0, cnx);
  const TapN<3> ty = tapn<3>(jy, cny);
  const T* cb[2] = {coarse + c0 * cvol, coarse + (c0 + 1) * cvol};
  add = add ? add + (int64_t)(2 * c0) * fvol : add;
  fine += (int64_t)(2 * c0) * fvol;
  T v[2][3][3][CX + 2];
#pragma unroll
  for (int r0 = 0; r0 < 2; ++r0) {
    load_plane_shared<T, CX>(cb[r0], z0 - 1, cnz, cplane, cnx, ty, tx, cscale, v[r0][0], a.tx, jx0);
    load_plane_shared<T, CX>(cb[r0], z0, cnz, cplane, cnx, ty, tx, cscale, v[r0][1], a.tx, jx0);
  }
  const T rs[2] = {T(1) / T(64), T(1) / T(128)};
  // the fine addend of step jz + 1 is requested before step jz is computed (ODIL_PAIR_PREFETCH): the march is a chain of
  // dependent steps and two waves per SIMD do not cover a step's memory latency on their own
  PackN<T, 2 * CX> nx[2][2][2];
  auto load_step = [&](int jz, PackN<T, 2 * CX> (&dst)[2][2][2]) {
    const int64_t fb = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      load_add<T, CX>(add + p * fvol, fb, fnx, dst[p][0], a.nt);
      load_add<T, CX>(add + p * fvol, fb + fplane, fnx, dst[p][1], a.nt);
    }
  };
#if ODIL_PAIR_PREFETCH
  if (add) load_step(z0, nx);
#endif
  for (int jz = z0; jz < z1; ++jz) {
    PackN<T, 2 * CX> ad[2][2][2];
    const int64_t fbase = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx0;
    if (add) {
#if ODIL_PAIR_PREFETCH
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 2; ++r) ad[p][q][r] = nx[p][q][r];
      if (jz + 1 < z1) load_step(jz + 1, nx);
#else
      load_step(jz, ad);
#endif
    }
#pragma unroll
    for (int r0 = 0; r0 < 2; ++r0) load_plane_shared<T, CX>(cb[r0], jz + 1, cnz, cplane, cnx, ty, tx, cscale, v[r0][2], a.tx, jx0);
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int sz = 0; sz < 2; ++sz) {
        T s[2][2 * CX];
        zero_plane<T, CX>(s);
#pragma unroll
        for (int r0 = 0; r0 <= p; ++r0)
#pragma unroll
          for (int rz = 0; rz < 2; ++rz) acc_plane<T, CX>(s, v[r0][sz + rz], sz == rz ? 1 : 3);
        store_plane<T, CX>(fine + p * fvol, fbase + sz * fplane, fnx, s, rs[p], ad[p][sz], add != nullptr, ascale, a.nt);
      }
#pragma unroll
    for (int r0 = 0; r0 < 2; ++r0) shift_window<T, CX>(v[r0]);
  }
}

// ------------------------------------------------------------------------------------
// P with a node-centred leading axis, float: the fine volumes 2k and 2k + 1 of a TILE of coarse cells, staged through LDS.
//
// The marching pair kernel above keeps two 3 x 3 x (CX + 2) coarse windows in registers (244 VGPRs, two waves per
// SIMD) and walks the z axis of the array -- 18 coarse planes on one rank of the tracer workload, two of them spent
// priming: 3.5 - 3.9 TB/s on the arrays where a device copy moves 4.75.  Here a workgroup stages the ghosted coarse
// neighbourhood of its tile for the volumes k and k + 1 ONCE (the joint ghost rule applied while staging:
// 2 c[clamp] - c[reflect] over all three cell axes, as load_plane forms it), then every thread forms 16-byte packs of
// fine values from LDS reads and streams them out: no dependent steps, ~70 registers, the fine addend in flight while
// the tile is staged.  The sums are formed in the order of acc_plane / store_plane (leading tap, z tap, then (ry, rx);
// weight * value added term by term; scaled by 1 / 64 or 1 / 128; addend last): bit-identical to the marching kernels.
// One thread per pair of coarse columns of the tile: its 2 x 2 x 4 fine values in both volumes, eight 16-byte packs.
// ------------------------------------------------------------------------------------
struct LeadTileArgs {
  int cn[3], fn[3];      // (z, y, x) coarse / fine extents of one volume
  int nty, ntx;          // tiles along y and x (z tiles = gridDim.x / (nty ntx))
  int nt;                // stream the fine arrays past the caches
  int64_t cvol, fvol;    // elements between leading indices of the coarse / fine arrays
};

// NV = 2: a node-centred leading axis (coarse volumes k, k + 1 -> fine volumes 2k, 2k + 1); NV = 1: one volume per
// leading index (a 3-D array, or a batch '.ccc'): coarse volume k -> fine volume k.
template <typename T, int NV, int TX, int TY, int TZ>
__global__ __launch_bounds__(kBlock) void k_interp_add_lead_tile(const T* __restrict__ coarse, const T* __restrict__ add,
                                                                 T* __restrict__ fine, LeadTileArgs a, T cscale, T ascale) {
  static_assert((TX / 2) * TY * TZ == kBlock, "one thread per pair of coarse columns of the tile");
  constexpr int LX = TX + 2, LY = TY + 2, LZ = TZ + 2, LVOL = LZ * LY * LX;
  __shared__ T sv[NV * LVOL];
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int k = (int)blockIdx.y;
  int tile = (int)blockIdx.x;
  const int tx_ = tile % a.ntx;
  tile /= a.ntx;
  const int ty_ = tile % a.nty, tz_ = tile / a.nty;
  const int z0 = tz_ * TZ, y0 = ty_ * TY, x0 = tx_ * TX;
  const T* cb = coarse + (int64_t)k * a.cvol;
  add = add ? add + (int64_t)(NV * k) * a.fvol : add;
  fine += (int64_t)(NV * k) * a.fvol;
  // the thread's coarse cell pair (jz, jy, 2 xp .. 2 xp + 1): its 2 x 2 fine rows of 4 values, in both fine volumes --
  // eight 16-byte packs; consecutive lanes hold consecutive packs of a fine row
  const int xp = threadIdx.x % (TX / 2), jyl = (threadIdx.x / (TX / 2)) % TY, jzl = threadIdx.x / ((TX / 2) * TY);
  const int jz = z0 + jzl, jy = y0 + jyl, jx = x0 + 2 * xp;
  const bool own = jz < cnz && jy < cny && jx < cnx;
  const int64_t base = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx;
  PackN<T, 4> ad[NV][2][2];
  if (add && own) {
#pragma unroll
    for (int pv = 0; pv < NV; ++pv)
#pragma unroll
      for (int sz = 0; sz < 2; ++sz)
#pragma unroll
        for (int sy = 0; sy < 2; ++sy)
          ad[pv][sz][sy] = stream_ld<T, 4>(add + pv * a.fvol + base + sz * fplane + sy * fnx, a.nt != 0);
  }
  // the ghosted coarse neighbourhood of the tile, volumes k and k + 1
  const bool inner = z0 >= 1 && z0 + TZ < cnz && y0 >= 1 && y0 + TY < cny && x0 >= 1 && x0 + TX < cnx;
  if (inner) {  // no wall (and no overhang) within reach: every staged position is a cell of the array
    for (int i = threadIdx.x; i < NV * LVOL; i += kBlock) {
      const int lv = i / LVOL, j = i - lv * LVOL;
      const int dx = j % LX, dy = (j / LX) % LY, dz = j / (LX * LY);
      sv[i] = cscale * cb[(int64_t)lv * a.cvol + (int64_t)(z0 - 1 + dz) * cplane + (int64_t)(y0 - 1 + dy) * cnx + (x0 - 1 + dx)];
    }
  } else {
    for (int i = threadIdx.x; i < NV * LVOL; i += kBlock) {
      const int lv = i / LVOL, j = i - lv * LVOL;
      const int dx = j % LX, dy = (j / LX) % LY, dz = j / (LX * LY);
      const int qz = z0 - 1 + dz, qy = y0 - 1 + dy, qx = x0 - 1 + dx;
      const bool out = qz < 0 || qz >= cnz || qy < 0 || qy >= cny || qx < 0 || qx >= cnx;
      const int zc = qz < 0 ? 0 : (qz >= cnz ? cnz - 1 : qz), zr = qz < 0 ? 1 : (qz >= cnz ? cnz - 2 : qz);
      const int yc = qy < 0 ? 0 : (qy >= cny ? cny - 1 : qy), yr = qy < 0 ? 1 : (qy >= cny ? cny - 2 : qy);
      const int xc = qx < 0 ? 0 : (qx >= cnx ? cnx - 1 : qx), xr = qx < 0 ? 1 : (qx >= cnx ? cnx - 2 : qx);
      const T* vol = cb + (int64_t)lv * a.cvol;
      const T val = cscale * vol[(int64_t)zc * cplane + (int64_t)yc * cnx + xc];
      T res = val;
      if (out) res = T(2) * val - cscale * vol[(int64_t)zr * cplane + (int64_t)yr * cnx + xr];
      sv[i] = res;
    }
  }
  __syncthreads();
  if (!own) return;
  // the 3 x 3 x 4 window of the thread's cell pair, per volume (index 0 <-> coarse j - 1)
  T v[NV][3][3][4];
#pragma unroll
  for (int lv = 0; lv < NV; ++lv)
#pragma unroll
    for (int dz = 0; dz < 3; ++dz)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const T* row = sv + lv * LVOL + ((jzl + dz) * LY + (jyl + dy)) * LX + 2 * xp;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[lv][dz][dy][c] = row[c];
      }
  const T rs[2] = {T(1) / T(64), T(1) / T(128)};
#pragma unroll
  for (int pv = 0; pv < NV; ++pv)
#pragma unroll
    for (int sz = 0; sz < 2; ++sz)
#pragma unroll
      for (int sy = 0; sy < 2; ++sy) {
        PackN<T, 4> pk;
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
          T t = T(0);
#pragma unroll
          for (int r0 = 0; r0 <= pv; ++r0)
#pragma unroll
            for (int rz = 0; rz < 2; ++rz)
#pragma unroll
              for (int ry = 0; ry < 2; ++ry)
#pragma unroll
                for (int rx = 0; rx < 2; ++rx) {
                  const int w = (sz == rz ? 1 : 3) * (sy == ry ? 1 : 3) * ((sx & 1) == rx ? 1 : 3);
                  t = t + T(w) * v[r0][sz + rz][sy + ry][(sx >> 1) + (sx & 1) + rx];
                }
          T o = t * rs[pv];
          if (add) o = ascale * ad[pv][sz][sy].e[sx] + o;
          pk.e[sx] = o;
        }
        stream_st<T, 4>(fine + pv * a.fvol + base + sz * fplane + sy * fnx, pk, a.nt != 0);
      }
}

static bool lead_tile_enabled() {
  const char* e = getenv("ODIL_LEAD_TILE");
  return !e || atoi(e) != 0;
}

// Launches the tiled kernel -- NV = 2: for the lead_cn - 1 pairs of fine volumes of a node-centred leading axis; NV = 1: for
// every volume of a 3-D array / a batch; false: the caller keeps its marching kernel.
template <typename T, int NV>
static bool lead_tile_launch(const T* coarse, const T* add, T* fine, const MarchArgs& m, T cscale, T ascale,
                             hipStream_t stream) {
  if (!lead_tile_enabled()) return false;
  for (int i = 0; i < 3; ++i)
    if (m.fn[i] != 2 * m.cn[i] || m.cn[i] < 2) return false;
  if (m.cn[2] % 2 || m.cn[2] < 32) return false;
  const uintptr_t pack = 4 * sizeof(T);  // the fine arrays are accessed in packs of four values
  if (reinterpret_cast<uintptr_t>(fine) % pack || (add && reinterpret_cast<uintptr_t>(add) % pack)) return false;
  LeadTileArgs a;
  for (int i = 0; i < 3; ++i) a.cn[i] = m.cn[i], a.fn[i] = m.fn[i];
  a.nt = m.nt;
  a.cvol = m.lead_cstride;
  a.fvol = (int64_t)m.fn[0] * m.fn[1] * m.fn[2];
  const int nlead = NV == 2 ? m.lead_cn - 1 : (m.lead_fn > 1 ? m.lead_fn : 1);
  if ((nlead > 1 || NV == 2) && a.fvol % 4) return false;  // every fine volume starts on a pack boundary
  // 32 x 8 x 2 coarse cells per workgroup: measured on the tracer rank's arrays (four fields, tools/mb_transfers_cfg5.py)
  // 1.92 ms against 1.99 (64 x 4 x 2), 1.97 (32 x 4 x 4), 2.24 (128 x 2 x 2: whole fine rows) and 2.41 (marching kernel)
  constexpr int tx = 32, ty = 8, tz = 2;
  a.ntx = (m.cn[2] + tx - 1) / tx;
  a.nty = (m.cn[1] + ty - 1) / ty;
  const int64_t tiles = (int64_t)a.ntx * a.nty * ((m.cn[0] + tz - 1) / tz);
  if (tiles >= ((int64_t)1 << 31) || nlead > 65535 || nlead < 1) return false;
  const dim3 grid((unsigned)tiles, (unsigned)nlead);
  hipLaunchKernelGGL((k_interp_add_lead_tile<T, NV, tx, ty, tz>), grid, dim3(kBlock), 0, stream, coarse, add, fine, a, cscale,
                     ascale);
  return true;
}

// ------------------------------------------------------------------------------------
// P^T
// ------------------------------------------------------------------------------------
// 1-D adjoint weights on a 'c' axis for coarse index J: window of 6 fine indices from 2J-2.
struct Adj6 {
  float wc[6], wr[6];
  bool special;
};

__device__ inline Adj6 adj6(int J, int n) {
  Adj6 t;
  const int F = 2 * n;
  const bool c_lo = J == 0, c_hi = J == n - 1, r_lo = J == 1, r_hi = J == n - 2;
  t.special = c_lo || c_hi || r_lo || r_hi;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int k = 2 * J - 2 + i;
    const float w = w_cell(J, k, F), lo = w_cell(-1, k, F), hi = w_cell(n, k, F);
    t.wc[i] = w + (c_lo ? lo : 0.f) + (c_hi ? hi : 0.f);
    t.wr[i] = w + (r_lo ? lo : 0.f) + (r_hi ? hi : 0.f);
  }
  return t;
}

// The arithmetic of the (y, x) reduction on loaded windows g[plane][row][pack] (see reduce_planes).
template <typename T, int CX, int NR, int NP>
__device__ inline void reduce_loaded(const PackN<T, 2 * CX> (&g)[NP][NR][3], const int (&f)[NP], int fnz,
                                     const Adj6& ay, const Adj6 (&ax)[CX], T (&rc)[NP][CX], T (&rr)[NP][CX]) {
  constexpr int R0 = (6 - NR) / 2;
  constexpr int NV = 2 * CX;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const bool inside = f[p] >= 0 && f[p] < fnz;
#pragma unroll
    for (int c = 0; c < CX; ++c) {
      T sc = T(0), sr = T(0);
#pragma unroll
      for (int r = 0; r < NR; ++r) {
        T xc = T(0), xr = T(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          const int e = NV - 2 + 2 * c + i;  // compile-time after unrolling
          const T val = g[p][r][e / NV].e[e % NV];
          xc = xc + T(ax[c].wc[i]) * val;
          xr = xr + T(ax[c].wr[i]) * val;
        }
        sc = sc + T(ay.wc[R0 + r]) * xc;
        sr = sr + T(ay.wr[R0 + r]) * xr;
      }
      rc[p][c] = inside ? sc : T(0);
      rr[p][c] = inside ? sr : T(0);
    }
  }
}

// (y, x) reduction of fine planes for the owned coarse columns: rc with the C weights, rr with the
// R weights.  The window of a row is three 16 B packs starting at fine x = 2 jx0 - 2 CX; column c
// uses its entries 2 CX - 2 + 2 c + (0..5).  All row loads of a call are issued back to back
// (addresses clamped into range; out-of-range rows / packs carry zero weights), and only then
// reduced: the kernel is bound by how many HBM requests a wave keeps in flight, so loads must not
// sit behind branches.  (Tried: one pack per lane + wave shuffles for the neighbours' values --
// 3x fewer loads, but the 32-64 ds_bpermute per step made it 15 % slower; the same with wave-wide DPP
// shifts instead of ds_bpermute and the segment-end lanes reading memory: 197 -> 284 VGPRs (the edge
// packs stay live across the divergent block), epoch 2.88 -> 3.32 ms; one plane at a time 3.00 ms.)
template <typename T, int CX, int NR, int NP>
__device__ inline void reduce_planes(const T* __restrict__ gfine, const int (&f)[NP], int fnz, int64_t fplane, int fny,
                                     int fnx, int jy, int jx0, const Adj6& ay, const Adj6 (&ax)[CX], T (&rc)[NP][CX],
                                     T (&rr)[NP][CX]) {
  constexpr int R0 = (6 - NR) / 2;  // first row of the window that is loaded (NR = 4: rows 1..4)
  constexpr int NV = 2 * CX;
  PackN<T, NV> g[NP][NR][3];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int fz = f[p] < 0 ? 0 : (f[p] >= fnz ? fnz - 1 : f[p]);
    const T* gp = gfine + (int64_t)fz * fplane;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      int fy = 2 * jy - 2 + R0 + r;
      fy = fy < 0 ? 0 : (fy >= fny ? fny - 1 : fy);
      const T* row = gp + (int64_t)fy * fnx;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        int fx = 2 * jx0 - NV + q * NV;
        fx = fx < 0 ? 0 : (fx >= fnx ? fnx - NV : fx);
        g[p][r][q] = *reinterpret_cast<const PackN<T, NV>*>(row + fx);
      }
    }
  }
  reduce_loaded<T, CX, NR, NP>(g, f, fnz, ay, ax, rc, rr);
}

template <typename T, int CX, int NP>
__device__ inline void reduce_dispatch(const T* __restrict__ gfine, const int (&f)[NP], int fnz, int64_t fplane,
                                       int fny, int fnx, int jy, int jx0, const Adj6& ay, const Adj6 (&ax)[CX],
                                       T (&rc)[NP][CX], T (&rr)[NP][CX]) {
  if (ay.special) {
    // boundary rows need the 6-row window: one plane at a time keeps the register count down
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int f1[1] = {f[p]};
      T c1[1][CX], r1[1][CX];
      reduce_planes<T, CX, 6, 1>(gfine, f1, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c1, r1);
#pragma unroll
      for (int c = 0; c < CX; ++c) {
        rc[p][c] = c1[0][c];
        rr[p][c] = r1[0][c];
      }
    }
  } else {
    reduce_planes<T, CX, 4, NP>(gfine, f, fnz, fplane, fny, fnx, jy, jx0, ay, ax, rc, rr);
  }
}

// Plane sums of two fine planes.  One batch of loads for a single column; for column pairs the
// 16 B packs of ONE plane already keep as many bytes in flight, and two planes at once would cost
// the registers that hold occupancy.
template <typename T, int CX>
__device__ inline void reduce_pair(const T* __restrict__ gfine, const int (&f)[2], int fnz, int64_t fplane, int fny,
                                   int fnx, int jy, int jx0, const Adj6& ay, const Adj6 (&ax)[CX], T (&rc)[2][CX],
                                   T (&rr)[2][CX]) {
  if constexpr (CX == 1) {
    reduce_dispatch<T, CX, 2>(gfine, f, fnz, fplane, fny, fnx, jy, jx0, ay, ax, rc, rr);
  } else {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int f1[1] = {f[p]};
      T c1[1][CX], r1[1][CX];
      reduce_dispatch<T, CX, 1>(gfine, f1, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c1, r1);
#pragma unroll
      for (int c = 0; c < CX; ++c) {
        rc[p][c] = c1[0][c];
        rr[p][c] = r1[0][c];
      }
    }
  }
}

// z-combination of one window of plane sums for coarse plane jz of a cell axis.
template <typename T>
__device__ inline T combine_z(const T (&wc)[6], const T (&wr)[6], int jz, int cnz, int fnz, bool xy_special,
                              int cut_lo, int cut_hi) {
  const bool z_special = ((jz == 0 || jz == 1) && !cut_lo) || ((jz == cnz - 2 || jz == cnz - 1) && !cut_hi);
  if (!z_special && !xy_special) return (T(0.25) * wc[1] + T(0.75) * wc[2]) + (T(0.75) * wc[3] + T(0.25) * wc[4]);
  if (!z_special) {
    // a column next to an x / y wall on a plane away from the z walls: the z weights are the constants
    // (0, 1, 3, 3, 1, 0) / 4 for both sums -- the general loop below with its weights known (same terms, same order);
    // forming them from the indices cost the waves that hold such columns more than the rest of their step
    T sc = T(0), sr = T(0);
    const T zw[6] = {T(0), T(0.25), T(0.75), T(0.75), T(0.25), T(0)};
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      sc = sc + zw[i] * wc[i];
      sr = sr + zw[i] * wr[i];
    }
    return T(2) * sc - sr;
  }
  T sc = T(0), sr = T(0);
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int k = 2 * jz - 2 + i;
    const float w = w_cell(jz, k, fnz), lo = w_cell(-1, k, fnz), hi = w_cell(cnz, k, fnz);
    const float zc_w = w + (jz == 0 && !cut_lo ? lo : 0.f) + (jz == cnz - 1 && !cut_hi ? hi : 0.f);
    const float zr_w = w + (jz == 1 && !cut_lo ? lo : 0.f) + (jz == cnz - 2 && !cut_hi ? hi : 0.f);
    sc = sc + T(zc_w) * wc[i];
    sr = sr + T(zr_w) * wr[i];
  }
  return T(2) * sc - sr;
}

// gradient of one coarse entry: store, optional scaled copy, optional Adam of this level's array
// by the lane that formed it
template <typename T>
__device__ inline void emit_coarse(T* __restrict__ gcoarse, T* __restrict__ gscaled, int64_t ci, T v, T scale,
                                   const AdamArgs<T>& ad) {
  gcoarse[ci] = v;
  if (gscaled) gscaled[ci] = scale * v;
  if (ad.x) {
    T xv = ad.x[ci], mv = ad.m[ci], vv = ad.v[ci];
    adam_update<T>(xv, mv, vv, gscaled ? scale * v : v, ad);
    ad.x[ci] = xv;
    ad.m[ci] = mv;
    ad.v[ci] = vv;
  }
}

template <typename T, int CX>
__device__ inline void put2(T (&wc)[CX][6], T (&wr)[CX][6], int at, const T (&c2)[2][CX], const T (&r2)[2][CX]) {
#pragma unroll
  for (int c = 0; c < CX; ++c) {
    wc[c][at] = c2[0][c], wc[c][at + 1] = c2[1][c];
    wr[c][at] = r2[0][c], wr[c][at + 1] = r2[1][c];
  }
}

template <typename T, int CX>
__device__ inline void slide(T (&wc)[CX][6], T (&wr)[CX][6]) {
#pragma unroll
  for (int c = 0; c < CX; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wc[c][i] = wc[c][i + 2];
      wr[c][i] = wr[c][i + 2];
    }
}

template <int CX>
__device__ inline void column_taps(int jx0, int cnx, Adj6 (&ax)[CX]) {
#pragma unroll
  for (int c = 0; c < CX; ++c) ax[c] = adj6(jx0 + c, cnx);
}

template <typename T, int CX>
__global__ __launch_bounds__(kBlock) void k_interp_adj_march(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                             T* __restrict__ gscaled, MarchArgs a, T scale,
                                                             AdamArgs<T> ad) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fnz = a.fn[0], fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int z0, z1, jy, jx0;
  if (!march_decode<CX>(a, z0, z1, jy, jx0)) return;
  Adj6 ax[CX];
  column_taps<CX>(jx0, cnx, ax);
  const Adj6 ay = adj6(jy, cny);
  // window of plane sums for fine planes 2jz-2 .. 2jz+3
  T wc[CX][6], wr[CX][6];
  {
    T c2[2][CX], r2[2][CX];
    const int fa[2] = {2 * z0 - 2, 2 * z0 - 1}, fb[2] = {2 * z0, 2 * z0 + 1};
    reduce_pair<T, CX>(gfine, fa, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c2, r2);
    put2<T, CX>(wc, wr, 0, c2, r2);
    reduce_pair<T, CX>(gfine, fb, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c2, r2);
    put2<T, CX>(wc, wr, 2, c2, r2);
  }
  for (int jz = z0; jz < z1; ++jz) {
    {
      T c2[2][CX], r2[2][CX];
      const int fn2[2] = {2 * jz + 2, 2 * jz + 3};
      reduce_pair<T, CX>(gfine, fn2, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c2, r2);
      put2<T, CX>(wc, wr, 4, c2, r2);
    }
    const int64_t ci = (int64_t)jz * cplane + (int64_t)jy * cnx + jx0;
#pragma unroll
    for (int c = 0; c < CX; ++c) {
      const T v = combine_z<T>(wc[c], wr[c], jz, cnz, fnz, ax[c].special || ay.special, a.cut_lo, a.cut_hi);
      emit_coarse<T>(gcoarse, gscaled, ci + c, v, scale, ad);
    }
    slide<T, CX>(wc, wr);
  }
}

// P^T for the large 'ccc' levels with the fine planes staged through LDS.  The register version above
// loads 12 packs per thread and fine plane for 2 packs of new data: HBM sees every line once, but the
// L2 serves six times the traffic (512^3 -> 256^3: 2.0 GB of HBM traffic in 0.59 ms).  Here a
// workgroup owns a tile of kTileY x kTileX coarse columns, reads the (2 kTileY + 4) x (2 kTileX + 4)
// window of each fine plane once with coalesced 16 B loads (1.27 x the tile's own data, the halo from
// L2), and every thread takes its 6 x 6 (4 x 6 away from the walls) window from LDS.  Same
// arithmetic, same order: the results are bit-identical to k_interp_adj_march.
#ifndef ODIL_ADJ_UNITS
#define ODIL_ADJ_UNITS kGridCap  // measured at 512^3: chain 0.80 / 0.69 / 0.72 ms for 1024 / 2048 / 4096
#endif
#ifndef ODIL_ADJ_UNITS_SMALL
#define ODIL_ADJ_UNITS_SMALL (kGridCap / 4)  // levels of <= 4 M coarse points: chain 0.690 -> 0.675 ms
#endif
#ifndef ODIL_TILE_Y
// 8 x 32 coarse columns per workgroup (rows of 64 fine cells = 512 B per array and plane): the fused adjoint
// kernel 1.83 -> 1.74 ms at 512^3 against 16 x 16 (tools/mb_tile_traffic.hip: the bare access pattern of the
// 16 x 16 tile is 5 % slower too); 4 x 64 needs 86 KB of LDS, one workgroup per CU: 3.4 ms
#define ODIL_TILE_Y 8
#define ODIL_TILE_X 32
#endif
#ifndef ODIL_TILE_UNITS
#define ODIL_TILE_UNITS (kGridCap / 2)  // 512^3 -> 256^3: 465 / 450 / 455 us for 2048 / 1024 / 512 units
#endif
constexpr int kTileY = ODIL_TILE_Y, kTileX = ODIL_TILE_X;
constexpr int kTileR = 2 * kTileY + 4, kTileC = 2 * kTileX + 4;       // fine rows / columns staged per plane
constexpr int kTilePacks = kTileR * (kTileC / 2);                       // packs of two values per plane
constexpr int kTileLoads = (kTilePacks + kBlock - 1) / kBlock;          // per thread and plane
static_assert(kTileY * kTileX == kBlock, "one thread per coarse column of the tile");
// Row stride of a staged plane in LDS, in packs: two more than the row holds, so that the two tile rows a group of
// sixteen lanes reads (eight lanes each, see tile_column) start 128 B apart modulo the 256 B of the banks.
constexpr int kTileRS = kTileC / 2 + 2;
constexpr int kTileLds = kTileR * kTileRS + 64;  // + a landing zone for the surplus packs of the last load round
// Coarse columns of the tile by thread: a WAVE owns a strip of kTileX / 4 columns over all kTileY rows.  The two
// columns next to a wall need the general 6 x 6 weights (ten times the arithmetic of the interior ones) and drag their
// whole wave through that path: with rows of the tile per wave every wave of a tile that touches an x wall had such
// lanes (planes of 256^2: every tile), with strips one wave in four has.
constexpr bool kTileStrips = (kBlock / 64) * (64 / kTileY) == kTileX;
__device__ __forceinline__ void tile_column(int& lx, int& ly) {
  if constexpr (kTileStrips) {
    constexpr int W = 64 / kTileY;  // columns per wave
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    lx = wave * W + lane % W;
    ly = lane / W;
  } else {
    lx = threadIdx.x % kTileX;
    ly = threadIdx.x / kTileX;
  }
}

template <typename T, int CX = 1>
struct TileVec {
  typedef T type __attribute__((ext_vector_type(2 * CX)));  // the fine cells under CX coarse columns
};

// This thread's packs of the fine planes fz0, fz0 + 1 (clamped into the array: planes beyond it carry
// zero weights), issued back to back.
template <typename T, int CX = 1>
__device__ __forceinline__ void tile_fetch(const T* __restrict__ gfine, int fz0, int fnz, int64_t fplane,
                                           const int64_t (&src)[kTileLoads],
                                           typename TileVec<T, CX>::type (&pre)[2][kTileLoads]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int fz = fz0 + q;
    fz = fz < 0 ? 0 : (fz >= fnz ? fnz - 1 : fz);
    const T* gp = gfine + (int64_t)fz * fplane;
#pragma unroll
    for (int i = 0; i < kTileLoads; ++i)
      pre[q][i] = *reinterpret_cast<const typename TileVec<T, CX>::type*>(gp + src[i]);
  }
}

// State of one thread of k_interp_adj_tile that the stages share.
template <typename T>
struct TileCtx {
  const T* gfine;
  T* gcoarse;
  T* gscaled;
  int cnz, cny, cnx, fnz, z0, jy, jx, lx, ly;
  bool sx, sy;  // the column / row is one of the two next to a wall (special weights)
  int cut_lo, cut_hi;
  int64_t cplane, fplane;
  bool owner;
  T scale;
};

// Weight tables of a thread parked in LDS: entry e of thread t at wt[e * kBlock + t] (conflict-free), order
// ay.wc, ay.wr, then wc, wr of each column.
template <int CX>
__device__ __forceinline__ void tile_tables_store(float* __restrict__ wt, const Adj6& ay, const Adj6 (&ax)[CX]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    wt[i * kBlock + threadIdx.x] = ay.wc[i];
    wt[(6 + i) * kBlock + threadIdx.x] = ay.wr[i];
#pragma unroll
    for (int cc = 0; cc < CX; ++cc) {
      wt[(12 + 12 * cc + i) * kBlock + threadIdx.x] = ax[cc].wc[i];
      wt[(18 + 12 * cc + i) * kBlock + threadIdx.x] = ax[cc].wr[i];
    }
  }
}
template <int CX>
__device__ __forceinline__ void tile_tables_load(const float* __restrict__ wt, Adj6& ay, Adj6 (&ax)[CX]) {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    ay.wc[i] = wt[i * kBlock + threadIdx.x];
    ay.wr[i] = wt[(6 + i) * kBlock + threadIdx.x];
#pragma unroll
    for (int cc = 0; cc < CX; ++cc) {
      ax[cc].wc[i] = wt[(12 + 12 * cc + i) * kBlock + threadIdx.x];
      ax[cc].wr[i] = wt[(18 + 12 * cc + i) * kBlock + threadIdx.x];
    }
  }
}

// Reduce the staged pair of fine planes 2 (z0 - 1 + k), + 1 over this thread's (y, x) window, slide it into
// the z-window and, from k = 2 on, emit coarse plane z0 + k - 2 (with the optional Adam update of that level).
// Away from the walls the weights are the constants (1, 3, 3, 1) / 4 per axis and the C and R sums coincide;
// the weight tables of the two columns / rows next to a wall are formed where they are needed instead of
// being carried in registers (24 VGPRs in double).  Zero-weight terms are skipped: the sums keep their bits.
// RS: row stride of the staged planes in packs; lxl: this thread's pack column in them (c.lx unless the caller stages
// a narrower window than the tile).
template <typename T, int CX = 1, int RS = kTileRS>
__device__ __forceinline__ void tile_reduce_emit(const TileCtx<T>& c,
                                                 const typename TileVec<T, CX>::type* __restrict__ tile0,
                                                 const typename TileVec<T, CX>::type* __restrict__ tile1, int k,
                                                 bool live, T (&wc)[CX][6], T (&wr)[CX][6], const AdamArgs<T>& ad,
                                                 const float* __restrict__ wt = nullptr,
                                                 const T* __restrict__ cpre = nullptr, int lxl = -1) {
  typedef typename TileVec<T, CX>::type P2;
  constexpr int NV = 2 * CX;
  if (lxl < 0) lxl = c.lx;
  const int f2[2] = {2 * (c.z0 - 1 + k), 2 * (c.z0 - 1 + k) + 1};
  T c2[2][CX], r2[2][CX];
  const bool xy_special = c.sx || c.sy;
  const bool wave_sy = __any(c.sy);  // (wave-uniform) some lane of this wave is next to a y wall
  if (xy_special && !wave_sy) {
    // next to an x wall only: the rows keep their constants (1, 3, 3, 1) / 4 and the two outer rows of the 6-row
    // window their zero weights (skipped: the sums keep their value), the columns take their tables -- a third of
    // the general path below, which is left to the waves that touch a y wall
    Adj6 ax[CX];
    if (wt) {
#pragma unroll
      for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int cc = 0; cc < CX; ++cc) {
          ax[cc].wc[i] = wt[(12 + 12 * cc + i) * kBlock + threadIdx.x];
          ax[cc].wr[i] = wt[(18 + 12 * cc + i) * kBlock + threadIdx.x];
        }
    } else {
      column_taps<CX>(c.jx, c.cnx, ax);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const P2* tq = q == 0 ? tile0 : tile1;
      const bool inside = f2[q] >= 0 && f2[q] < c.fnz;
      T sc[CX], sr[CX];
#pragma unroll
      for (int cc = 0; cc < CX; ++cc) sc[cc] = sr[cc] = T(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const P2* row = tq + (2 * c.ly + 1 + r) * RS + lxl;
        const P2 g[3] = {row[0], row[1], row[2]};
#pragma unroll
        for (int cc = 0; cc < CX; ++cc) {
          T xc = T(0), xr = T(0);
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            const int e = NV - 2 + 2 * cc + i;
            const T val = g[e / NV][e % NV];
            xc = xc + T(ax[cc].wc[i]) * val;
            xr = xr + T(ax[cc].wr[i]) * val;
          }
          const T wy = T((r == 0 || r == 3) ? 0.25 : 0.75);
          sc[cc] = sc[cc] + wy * xc;
          sr[cc] = sr[cc] + wy * xr;
        }
      }
#pragma unroll
      for (int cc = 0; cc < CX; ++cc) c2[q][cc] = inside ? sc[cc] : T(0), r2[q][cc] = inside ? sr[cc] : T(0);
    }
  } else if (xy_special) {
    // weight tables of this thread: from the LDS copy made at kernel start when there is one (tiles with a
    // wall send every wave through this branch, and forming three tables costs more than the reduction)
    Adj6 ax[CX], ay;
    if (wt) {
      tile_tables_load<CX>(wt, ay, ax);
      ay.special = c.sy;
#pragma unroll
      for (int cc = 0; cc < CX; ++cc) ax[cc].special = c.jx + cc < 2 || c.jx + cc >= c.cnx - 2;
    } else {
      column_taps<CX>(c.jx, c.cnx, ax);
      ay = adj6(c.jy, c.cny);
    }
    // one plane at a time keeps the register count down (the 6-row window)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const P2* tq = q == 0 ? tile0 : tile1;
      const int f1[1] = {f2[q]};
      T c1[1][CX], r1[1][CX];
      PackN<T, NV> g[1][6][3];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          const P2 t = tq[(2 * c.ly + r) * RS + lxl + w];
#pragma unroll
          for (int e = 0; e < NV; ++e) g[0][r][w].e[e] = t[e];
        }
      reduce_loaded<T, CX, 6, 1>(g, f1, c.fnz, ay, ax, c1, r1);
#pragma unroll
      for (int cc = 0; cc < CX; ++cc) c2[q][cc] = c1[0][cc], r2[q][cc] = r1[0][cc];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const P2* tq = q == 0 ? tile0 : tile1;
      const bool inside = f2[q] >= 0 && f2[q] < c.fnz;
      T sc[CX];
#pragma unroll
      for (int cc = 0; cc < CX; ++cc) sc[cc] = T(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // the window of the CX columns: the last value of the pack on the left, the own pack, the first value of the
        // pack on the right -- read as exactly that (one value, one pack, one value: 2/3 of the LDS bytes of three
        // packs in double, half in float; the reduction's time is its LDS reads, not its arithmetic)
        const T* rowv = reinterpret_cast<const T*>(tq + (2 * c.ly + 1 + r) * RS + lxl);
        const P2 own = *reinterpret_cast<const P2*>(rowv + NV);
        T val[NV + 2];
        val[0] = rowv[NV - 1];
#pragma unroll
        for (int e = 0; e < NV; ++e) val[1 + e] = own[e];
        val[NV + 1] = rowv[2 * NV];
#pragma unroll
        for (int cc = 0; cc < CX; ++cc) {
          // window entries 2 cc + (0 .. 3), weights (1, 3, 3, 1) / 4
          T xc = T(0);
          xc = xc + T(0.25) * val[2 * cc];
          xc = xc + T(0.75) * val[2 * cc + 1];
          xc = xc + T(0.75) * val[2 * cc + 2];
          xc = xc + T(0.25) * val[2 * cc + 3];
          sc[cc] = sc[cc] + T((r == 0 || r == 3) ? 0.25 : 0.75) * xc;
        }
      }
#pragma unroll
      for (int cc = 0; cc < CX; ++cc) c2[q][cc] = r2[q][cc] = inside ? sc[cc] : T(0);
    }
  }
  put2<T, CX>(wc, wr, 4, c2, r2);  // always the last two slots (a static position: no register indexing)
  if (k >= 2 && c.owner && live) {
    const int jz = c.z0 + k - 2;
    const int64_t ci = (int64_t)jz * c.cplane + (int64_t)c.jy * c.cnx + c.jx;
#pragma unroll
    for (int cc = 0; cc < CX; ++cc) {
      const T v = combine_z<T>(wc[cc], wr[cc], jz, c.cnz, c.fnz, xy_special, c.cut_lo, c.cut_hi);
      if (cpre && ad.x) {
        // this level's x, m, v of the entry were read at the start of the step (the caller's registers): the
        // update does not wait for a memory round trip at the end of every step
        c.gcoarse[ci + cc] = v;
        T xe = cpre[3 * cc], me = cpre[3 * cc + 1], ve = cpre[3 * cc + 2];
        adam_update<T>(xe, me, ve, v, ad);
        ad.x[ci + cc] = xe, ad.m[ci + cc] = me, ad.v[ci + cc] = ve;
      } else {
        emit_coarse<T>(c.gcoarse, c.gscaled, ci + cc, v, c.scale, ad);
      }
    }
  }
  slide<T, CX>(wc, wr);
}

// One pair of fine planes (2 (z0 - 1 + k), + 1): publish the staged packs, refill the staging registers
// with the pair `ahead` steps later, reduce this pair from LDS, and from k = 2 on emit coarse plane
// z0 + k - 2.  `live` is false for the padding step of an odd pair count (barriers only).
template <typename T, int CX = 1>
__device__ __forceinline__ void tile_stage(const TileCtx<T>& c, typename TileVec<T, CX>::type* __restrict__ tile0,
                                           typename TileVec<T, CX>::type* __restrict__ tile1, int k, int ahead,
                                           bool live, const int64_t (&src)[kTileLoads], const int (&dst)[kTileLoads],
                                           typename TileVec<T, CX>::type (&pre)[2][kTileLoads], T (&wc)[CX][6],
                                           T (&wr)[CX][6], const AdamArgs<T>& ad, const float* __restrict__ wt) {
  __syncthreads();  // the previous pair has been consumed
#pragma unroll
  for (int i = 0; i < kTileLoads; ++i) {
    tile0[dst[i]] = pre[0][i];
    tile1[dst[i]] = pre[1][i];
  }
  __syncthreads();
  // (unconditional: past the last pair it re-reads clamped planes, which keeps the staging registers out
  // of scratch memory)
  tile_fetch<T, CX>(c.gfine, 2 * (c.z0 - 1 + k + ahead), c.fnz, c.fplane, src, pre);
  tile_reduce_emit<T, CX>(c, tile0, tile1, k, live, wc, wr, ad, wt);
}

// CX coarse columns per thread: 1, or 2 for float so that every LDS / global access is 16 B per lane (the
// float kernel with one column ran at 1.8 TB/s on the space part of the 4-D tracer transposes).
template <typename T, int CX>
__global__ __launch_bounds__(kBlock) void k_interp_adj_tile(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                            T* __restrict__ gscaled, MarchArgs a, T scale,
                                                            AdamArgs<T> ad) {
  typedef typename TileVec<T, CX>::type P2;  // a native vector type: staged values stay in registers
  constexpr int NV = 2 * CX;
  // row-major over (row, pack) with rows of kTileRS packs; the loads are padded to a whole number of packs per
  // thread so that they and the LDS writes need no guards (the surplus packs re-read pack 0 and land behind the tile)
  __shared__ P2 tile[2][kTileLds];
  __shared__ float wtab[12 * (CX + 1) * kBlock];  // weight tables of the threads next to a wall
  const int cny = a.cn[1], fny = a.fn[1], fnx = a.fn[2];
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;  // whole workgroup
  TileCtx<T> c;
  c.cnz = a.cn[0], c.cnx = a.cn[2], c.fnz = a.fn[0];
  c.cut_lo = a.cut_lo, c.cut_hi = a.cut_hi;
  c.cplane = (int64_t)cny * c.cnx, c.fplane = (int64_t)fny * fnx;
  // a leading batch axis ('.ccc': the space part of the 4-D space-time transposes): one volume per blockIdx.y
  const int64_t cvol = (int64_t)blockIdx.y * c.cnz * c.cplane, fvol = (int64_t)blockIdx.y * c.fnz * c.fplane;
  gfine += fvol;
  c.gfine = gfine, c.gcoarse = gcoarse + cvol, c.gscaled = gscaled ? gscaled + cvol : nullptr;
  if (ad.x) ad.x += cvol, ad.m += cvol, ad.v += cvol;
  c.scale = scale;
  c.z0 = zc * a.usched.ZC;
  const int z1 = c.z0 + a.usched.ZC < c.cnz ? c.z0 + a.usched.ZC : c.cnz;
  tile_column(c.lx, c.ly);
  c.jy = yt * kTileY + c.ly, c.jx = (xt * kTileX + c.lx) * CX;  // first of this thread's CX columns
  c.owner = c.jy < cny && c.jx < c.cnx;  // (CX = 2 needs an even cnx: both columns exist or neither)
  c.cny = cny;
  c.sx = c.owner && (c.jx < 2 || c.jx + CX - 1 >= c.cnx - 2);
  c.sy = c.owner && (c.jy < 2 || c.jy >= cny - 2);
  if (c.sx || c.sy) {  // (read back by the same thread only: no barrier needed)
    Adj6 ax[CX];
    column_taps<CX>(c.jx, c.cnx, ax);
    tile_tables_store<CX>(wtab, adj6(c.jy, cny), ax);
  }
  // this thread's share of a staged plane: packs p = threadIdx.x + k kBlock of the tile, row-major; the window
  // of a row starts 2 CX fine cells left of the tile (pack aligned; 2 are needed)
  int64_t src[kTileLoads];
  int dst[kTileLoads];
#pragma unroll
  for (int k = 0; k < kTileLoads; ++k) {
    const int p = threadIdx.x + k * kBlock;
    const int r = p < kTilePacks ? p / (kTileC / 2) : 0, cc = p < kTilePacks ? p - r * (kTileC / 2) : 0;
    dst[k] = p < kTilePacks ? r * kTileRS + cc : kTileR * kTileRS + (threadIdx.x & 63);
    int fy = 2 * yt * kTileY - 2 + r, fx = NV * (xt * kTileX - 1 + cc);
    fy = fy < 0 ? 0 : (fy >= fny ? fny - 1 : fy);
    fx = fx < 0 ? 0 : (fx >= fnx ? fnx - NV : fx);
    src[k] = (int64_t)fy * fnx + fx;
  }
  T wc[CX][6], wr[CX][6];
#pragma unroll
  for (int cc = 0; cc < CX; ++cc)
#pragma unroll
    for (int i = 0; i < 6; ++i) wc[cc][i] = wr[cc][i] = T(0);
  // pairs of fine planes 2 (z0 - 1 + k), + 1 for k = 0 .. z1 - z0 + 1 (the first two prime the window); the
  // next pair is in flight while one is reduced.  (Two pairs ahead in two register sets: 256 VGPRs, no
  // faster -- 452 vs 448 us; three waves per SIMD spill and take 630 us.)
  const int npairs = z1 - c.z0 + 2;
  P2 pre[2][kTileLoads];
  tile_fetch<T, CX>(gfine, 2 * (c.z0 - 1), c.fnz, c.fplane, src, pre);
  for (int k = 0; k < npairs; ++k) {
    tile_stage<T, CX>(c, tile[0], tile[1], k, 1, true, src, dst, pre, wc, wr, ad, wtab);
  }
}

// ------------------------------------------------------------------------------------
// Stencil adjoint + first P^T in one kernel (Poisson hot path, 3-D): g0 = scale A^T fu is formed on the
// staged tile straight from fu, consumed by the Adam update of the finest level and by the transpose to
// the next level, and never written to memory -- the separate kernels write g0 (1 word per fine cell)
// and read it back.  Same tile as k_interp_adj_tile; fu is staged with one more halo cell per side in a
// ring of four planes (plane z of g0 needs fu planes z - 1 .. z + 1), g0 of the tile's halo ring is
// recomputed by each neighbouring workgroup (same expression, same bits).  g0 and g1 are bit-identical
// to k_poisson_adjoint followed by k_interp_adj_tile.
// ------------------------------------------------------------------------------------
constexpr int kFuR = 2 * kTileY + 6;   // fine rows from 2 jy0 - 3
constexpr int kFuC = kTileX + 4;       // packs of two: fine x from 2 jx0 - 4 (pack aligned)
constexpr int kFuPacks = kFuR * kFuC;
constexpr int kFuLoads = (kFuPacks + kBlock - 1) / kBlock;
constexpr int kGC = kTileC / 2;        // packs per row of the g0 tile (stored with the row stride kTileRS)
constexpr int kG0Steps = (kTileR * kFuC + kBlock - 1) / kBlock;  // g0 is formed walking rows of kFuC lanes
constexpr int kOwnLoads = 2 * kTileY * kTileX / kBlock;  // own packs per thread and plane

template <typename T>
__device__ __forceinline__ void fu_fetch(const T* __restrict__ fu, int fz0, int fnz, int64_t fplane,
                                         const int64_t (&src)[kFuLoads], typename TileVec<T>::type (&pre)[2][kFuLoads]) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int fz = fz0 + q;
    fz = fz < 0 ? 0 : (fz >= fnz ? fnz - 1 : fz);
    const T* gp = fu + (int64_t)fz * fplane;
#pragma unroll
    for (int i = 0; i < kFuLoads; ++i) pre[q][i] = *reinterpret_cast<const typename TileVec<T>::type*>(gp + src[i]);
  }
}

// The ring holds scale * fu: every cell of g0 reads seven staged values, and the product formed once per staged
// value is the product k_poisson_adjoint forms at each use (same operands, same rounding).
template <typename T>
__device__ __forceinline__ void fu_publish(typename TileVec<T>::type* __restrict__ ring, int fz0,
                                           const typename TileVec<T>::type (&pre)[2][kFuLoads], T scale) {
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    typename TileVec<T>::type* slot = ring + ((fz0 + q + 8) & 3) * kFuPacks;
#pragma unroll
    for (int i = 0; i < kFuLoads; ++i) {
      const int p = threadIdx.x + i * kBlock;
      if (p < kFuPacks) slot[p] = scale * pre[q][i];
    }
  }
}

// (s / h2) of one axis away from every wall: the interior row of adj_axis without its index tests.
template <typename T, bool MUL>
__device__ __forceinline__ T lap_inner(T fb, T fm, T fp, const H2<T>& h, int ax) {
  const T s = (fp + fm) + T(-2) * fb;
  if constexpr (MUL) return s * h.inv[ax];
  return div_h2<T>(s, h, ax);
}

// MUL: the three 1 / h^2 are exact (powers of two) and the division is a product -- known at launch, so the
// instantiation that runs carries no divide.
// G0: g0 is also written out (the callers that want the finest gradient itself); both levels share the Adam
// hyper-parameters (one set in scalar registers: the kernel spills them).
template <typename T, bool MUL, bool G0>
__global__ __launch_bounds__(kBlock) void k_poisson_adjoint_tile(const T* __restrict__ fu, T* __restrict__ g0out_,
                                                                 T* __restrict__ gcoarse, MarchArgs a, H2<T> h,
                                                                 T scale, AdamArgs<T> ad0, T* __restrict__ x1,
                                                                 T* __restrict__ m1, T* __restrict__ v1) {
  T* const g0out = G0 ? g0out_ : nullptr;
  AdamArgs<T> ad1 = ad0;
  ad1.x = x1, ad1.m = m1, ad1.v = v1;
  typedef typename TileVec<T>::type P2;
  __shared__ P2 ring[4 * kFuPacks];      // scale * fu, planes z & 3
  __shared__ P2 gt[2][kTileR * kTileRS]; // g0 of the current pair of planes
  const int cny = a.cn[1], fny = a.fn[1], fnx = a.fn[2];
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;  // whole workgroup
  if constexpr (MUL) h.mul_ok[0] = h.mul_ok[1] = h.mul_ok[2] = 1;  // (the wall rows lose their divide as well)
  // the step size of a replayed epoch lives in device memory: one read per workgroup, not one per cell
  if (ad0.alpha_dev) ad0.alpha = *ad0.alpha_dev;
  ad0.alpha_dev = nullptr;
  ad1.alpha = ad0.alpha, ad1.alpha_dev = nullptr;
  TileCtx<T> c;
  c.gfine = nullptr, c.gcoarse = gcoarse, c.gscaled = nullptr;
  c.cnz = a.cn[0], c.cnx = a.cn[2], c.fnz = a.fn[0];
  c.cut_lo = a.cut_lo, c.cut_hi = a.cut_hi;
  c.cplane = (int64_t)cny * c.cnx, c.fplane = (int64_t)fny * fnx;
  c.scale = T(1);
  c.z0 = zc * a.usched.ZC;
  const int z1 = c.z0 + a.usched.ZC < c.cnz ? c.z0 + a.usched.ZC : c.cnz;
  tile_column(c.lx, c.ly);
  c.jy = yt * kTileY + c.ly, c.jx = xt * kTileX + c.lx;
  c.owner = c.jy < cny && c.jx < c.cnx;
  c.cny = cny;
  c.sx = c.owner && (c.jx < 2 || c.jx >= c.cnx - 2);
  c.sy = c.owner && (c.jy < 2 || c.jy >= cny - 2);
  const int fy0 = 2 * yt * kTileY, fx0 = 2 * xt * kTileX;  // first own fine row / column of the tile
  // every g0 cell of the tile (own cells and the halo ring) has interior rows of the stencil on y and x
  const bool tile_inner = fy0 >= 4 && fy0 + 2 * kTileY + 4 <= fny && fx0 >= 4 && fx0 + 2 * kTileX + 4 <= fnx;
  // this thread's packs of a staged fu plane (row-major over the (kFuR, kFuC) window, clamped into the array)
  int64_t src[kFuLoads];
#pragma unroll
  for (int i = 0; i < kFuLoads; ++i) {
    int p = threadIdx.x + i * kBlock;
    p = p < kFuPacks ? p : kFuPacks - 1;
    const int r = p / kFuC, cc = p - r * kFuC;
    int fy = fy0 - 3 + r, fx = fx0 - 4 + 2 * cc;
    fy = fy < 0 ? 0 : (fy >= fny ? fny - 1 : fy);
    fx = fx < 0 ? 0 : (fx >= fnx ? fnx - 2 : fx);
    src[i] = (int64_t)fy * fnx + fx;
  }
  T wc[1][6], wr[1][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) wc[0][i] = wr[0][i] = T(0);
  // g0 planes of step k: zA = 2 m, zB = 2 m + 1 with m = z0 - 1 + k; they need the fu planes 2 m - 1 .. 2 m + 2,
  // staged as the pairs (2 m - 1, 2 m) and (2 m + 1, 2 m + 2)
  const int npairs = z1 - c.z0 + 2;
  P2 pre[2][kFuLoads];
  fu_fetch<T>(fu, 2 * (c.z0 - 1) - 1, c.fnz, c.fplane, src, pre);
  fu_publish<T>(ring, 2 * (c.z0 - 1) - 1, pre, scale);
  fu_fetch<T>(fu, 2 * (c.z0 - 1) + 1, c.fnz, c.fplane, src, pre);
  // own cells of the tile (the finest level's arrays): packs o = threadIdx.x + i kBlock, row-major; offset -1
  // marks a pack beyond the array (a per-lane flag array would cost scalar registers: the kernel spills them)
  int64_t own_off[kOwnLoads];
#pragma unroll
  for (int i = 0; i < kOwnLoads; ++i) {
    const int o = threadIdx.x + i * kBlock;
    const int row = o / kTileX, cc = o - row * kTileX;
    const int y = fy0 + row, x = fx0 + 2 * cc;
    own_off[i] = y < fny && x < fnx ? (int64_t)y * fnx + x : -1;
  }
  P2 xv[kOwnLoads], mv[kOwnLoads], vv[kOwnLoads];
#pragma unroll
  for (int i = 0; i < kOwnLoads; ++i) xv[i] = mv[i] = vv[i] = P2{T(0), T(0)};
  bool pending = false;  // the previous plane's update is outstanding
  for (int k = 0; k < npairs; ++k) {
    const int m = c.z0 - 1 + k, zA = 2 * m;
    // (no barrier here: the ring slots written now were last read by g0 of the previous step's planes, before
    // that step's last barrier; the transpose that ended the previous step reads gt only, and the next writer of
    // gt comes after the barrier below)
    fu_publish<T>(ring, zA + 1, pre, scale);
    __syncthreads();
    fu_fetch<T>(fu, zA + 3, c.fnz, c.fplane, src, pre);  // next pair of fu planes in flight
    // the coarse entry this step emits: its x, m, v travel while the step's planes are formed
    T cpre[3] = {T(0), T(0), T(0)};
    if (k >= 2 && c.owner && ad1.x) {
      const int64_t ci = (int64_t)(c.z0 + k - 2) * c.cplane + (int64_t)c.jy * c.cnx + c.jx;
      cpre[0] = ad1.x[ci], cpre[1] = ad1.m[ci], cpre[2] = ad1.v[ci];
    }
    // One plane at a time.  The update of the finest level runs one plane behind the plane whose g0 is
    // being formed: its loads (x, m, v of the own cells) are issued before g0 of its plane is formed and
    // consumed at the start of the next step, so they have a whole step to arrive (gt keeps two planes).
#pragma unroll 1
    for (int q = 0; q < 2; ++q) {
      const int z = zA + q;
      // (the update of plane z - 1 comes BEFORE plane z is formed: its readers of gt are then separated from
      // the next writer of that half by this step's barrier)
      if (pending) {  // plane z - 1: its g0 is in the other half of gt
        const P2* gprev = gt[q ^ 1];
        const int64_t zoff = (int64_t)(z - 1) * c.fplane;
#pragma unroll
        for (int i = 0; i < kOwnLoads; ++i) {
          const int o = threadIdx.x + i * kBlock;
          const int row = o / kTileX, cc = o - row * kTileX;
          if (own_off[i] < 0) continue;
          const P2 g = gprev[(row + 2) * kTileRS + cc + 1];
          const int64_t off = zoff + own_off[i];
          if (g0out) __builtin_nontemporal_store(g, reinterpret_cast<P2*>(g0out + off));
          if (ad0.x) {
            P2 xn = xv[i], mn = mv[i], vn = vv[i];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              T xe = xn[e], me = mn[e], ve = vn[e];
              adam_update<T>(xe, me, ve, g[e], ad0);
              xn[e] = xe, mn[e] = me, vn[e] = ve;
            }
            __builtin_nontemporal_store(xn, reinterpret_cast<P2*>(ad0.x + off));
            __builtin_nontemporal_store(mn, reinterpret_cast<P2*>(ad0.m + off));
            __builtin_nontemporal_store(vn, reinterpret_cast<P2*>(ad0.v + off));
          }
        }
      }
      pending = z >= 2 * c.z0 && z < 2 * z1;  // planes of this chunk (the others belong to its neighbours)
      if (pending && ad0.x) {
        const int64_t zoff = (int64_t)z * c.fplane;
#pragma unroll
        for (int i = 0; i < kOwnLoads; ++i) {
          const int64_t off = zoff + (own_off[i] < 0 ? 0 : own_off[i]);
          xv[i] = __builtin_nontemporal_load(reinterpret_cast<const P2*>(ad0.x + off));
          mv[i] = __builtin_nontemporal_load(reinterpret_cast<const P2*>(ad0.m + off));
          vv[i] = __builtin_nontemporal_load(reinterpret_cast<const P2*>(ad0.v + off));
        }
      }
      const P2* pm = ring + ((z - 1 + 8) & 3) * kFuPacks;
      const P2* pc = ring + ((z + 8) & 3) * kFuPacks;
      const P2* pp = ring + ((z + 1 + 8) & 3) * kFuPacks;
      P2* gq = gt[q];
      // slab interfaces (ghost planes beyond them, multi-GPU): the transposed stencil has its interior rows
      // there -- the wall rows of adj_axis sit two planes inside an end that is a wall
      const int zj = z + (c.cut_lo ? 2 : 0), zn = c.fnz + (c.cut_lo ? 2 : 0) + (c.cut_hi ? 2 : 0);
      const bool z_in = z >= 0 && z < c.fnz;
      // Lanes walk the fu window row by row in ITS row length (kFuC packs, of which the g0 tile takes kGC): the
      // seven window reads of consecutive lanes are then consecutive packs.  Walking the g0 tile's own rows
      // (kGC packs) made every wave straddle a row end with a two-pack gap in its window addresses, a two-way
      // LDS bank conflict in ~4 of 9 lane groups (SQ_LDS_BANK_CONFLICT 7.3e7 of 1.85e8 active cycles,
      // profiles/r02_v0_poisson_pmc.txt).
      if (tile_inner && z_in && zj >= 2 && zj < zn - 2) {
        // No wall within reach of any cell of this plane of the tile (a workgroup-uniform test: 3 of 4 tiles at
        // 512^3): the interior row of adj_axis on every axis, no index tests -- the same expression, term by
        // term.  The general form below costs ~3x the instructions (profiles/r02_v3_poisson_pmc.txt).
#pragma unroll 1
        for (int i = 0; i < kG0Steps; ++i) {
          const int w = threadIdx.x + i * kBlock;
          const int r = w / kFuC, cc = w - r * kFuC;
          if (r < kTileR && cc < kGC) {
            const int at = w + kFuC + 1;
            const P2 fc = pc[at], fl = pc[at - 1], fr = pc[at + 1];
            const P2 fym = pc[at - kFuC], fyp = pc[at + kFuC], fzm = pm[at], fzp = pp[at];
            P2 g;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
              const T fb = fc[e];
              T acc = T(0);
              acc = acc + lap_inner<T, MUL>(fb, fzm[e], fzp[e], h, 0);
              acc = acc + lap_inner<T, MUL>(fb, fym[e], fyp[e], h, 1);
              acc = acc + lap_inner<T, MUL>(fb, e == 0 ? fl[1] : fc[0], e == 0 ? fc[1] : fr[0], h, 2);
              g[e] = acc;
            }
            gq[r * kTileRS + cc] = g;
          }
        }
      } else {
#pragma unroll 1
        for (int i = 0; i < kG0Steps; ++i) {
          const int w = threadIdx.x + i * kBlock;
          const int r = w / kFuC, cc = w - r * kFuC;
          if (r < kTileR && cc < kGC) {
            const int y = fy0 - 2 + r, x = fx0 - 2 + 2 * cc;
            const int at = w + kFuC + 1;  // = (r + 1) * kFuC + cc + 1: the same cells in the fu window
            P2 g;
            g[0] = T(0), g[1] = T(0);
            if (z_in && y >= 0 && y < fny && x >= 0 && x < fnx) {
              const P2 fc = pc[at], fl = pc[at - 1], fr = pc[at + 1];
              const P2 fym = pc[at - kFuC], fyp = pc[at + kFuC], fzm = pm[at], fzp = pp[at];
#pragma unroll
              for (int e = 0; e < 2; ++e) {
                // the expression of k_poisson_adjoint, term by term
                const T fb = fc[e];
                T acc = T(0);
                acc = acc + adj_axis<T>(fb, fzm[e], fzp[e], zj, zn, h, 0);
                acc = acc + adj_axis<T>(fb, fym[e], fyp[e], y, fny, h, 1);
                acc = acc + adj_axis<T>(fb, e == 0 ? fl[1] : fc[0], e == 0 ? fc[1] : fr[0], x + e, fnx, h, 2);
                g[e] = acc;
              }
            }
            gq[r * kTileRS + cc] = g;
          }
        }
      }
      __syncthreads();
    }
    tile_reduce_emit<T>(c, gt[0], gt[1], k, true, wc, wr, ad1, nullptr, cpre);
  }
}

// P^T with a node-centred marching axis: coarse plane J collects fine plane 2J and half of the
// fine planes 2J-1 and 2J+1, each reduced over its (y, x) window with the two-cell-axis ghost rule.
template <typename T, int CX>
__global__ __launch_bounds__(kBlock) void k_interp_adj_march_n(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                               T* __restrict__ gscaled, MarchArgs a, T scale,
                                                               AdamArgs<T> ad) {
  const int cny = a.cn[1], cnx = a.cn[2];
  const int fnz = a.fn[0], fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int z0, z1, jy, jx0;
  if (!march_decode<CX>(a, z0, z1, jy, jx0)) return;
  Adj6 ax[CX];
  column_taps<CX>(jx0, cnx, ax);
  const Adj6 ay = adj6(jy, cny);
  T pc[CX], pr[CX];  // plane sums of the odd fine plane below
  {
    const int f1[1] = {2 * z0 - 1};
    T c1[1][CX], r1[1][CX];
    reduce_dispatch<T, CX, 1>(gfine, f1, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c1, r1);
#pragma unroll
    for (int c = 0; c < CX; ++c) pc[c] = c1[0][c], pr[c] = r1[0][c];
  }
  for (int jz = z0; jz < z1; ++jz) {
    T c2[2][CX], r2[2][CX];
    const int f2[2] = {2 * jz, 2 * jz + 1};
    reduce_pair<T, CX>(gfine, f2, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c2, r2);
    const int64_t ci = (int64_t)jz * cplane + (int64_t)jy * cnx + jx0;
#pragma unroll
    for (int c = 0; c < CX; ++c) {
      const T sc = c2[0][c] + T(0.5) * (pc[c] + c2[1][c]);
      T v = sc;
      if (ax[c].special || ay.special) {
        const T sr = r2[0][c] + T(0.5) * (pr[c] + r2[1][c]);
        v = T(2) * sc - sr;
      }
      pc[c] = c2[1][c], pr[c] = r2[1][c];
      emit_coarse<T>(gcoarse, gscaled, ci + c, v, scale, ad);
    }
  }
}

// P^T for the 4-D layouts: coarse volume J0 (blockIdx.y) collects the fine volumes 2 J0 and, halved,
// 2 J0 +- 1 on a node-centred leading axis (volume J0 alone on a batch axis); one z-window of plane
// sums per contributing fine volume.
template <typename T, int CX, int NTAP>
__global__ __launch_bounds__(kBlock) void k_interp_adj_march_lead(const T* __restrict__ gfine,
                                                                  T* __restrict__ gcoarse, T* __restrict__ gscaled,
                                                                  MarchArgs a, T scale, AdamArgs<T> ad) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fnz = a.fn[0], fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int64_t fvol = (int64_t)fnz * fplane;  // (the coarse volumes are a.lead_cstride apart)
  const int J0 = blockIdx.y;
  const bool node = a.lead_loc == kNode;
  int z0, z1, jy, jx0;
  if (!march_decode<CX>(a, z0, z1, jy, jx0)) return;
  Adj6 ax[CX];
  column_taps<CX>(jx0, cnx, ax);
  const Adj6 ay = adj6(jy, cny);
  // contributing fine volumes: (index, weight); missing ones get weight 0 and a valid index
  int fv[3];
  T wv[3];
  // NTAP: 3 contributing fine volumes on a node-centred leading axis, 1 on a batch axis (one window only)
  if (node) {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int f = 2 * J0 - 1 + t;
      const bool ok = f >= 0 && f < a.lead_fn;
      fv[t] = ok ? f : 2 * J0;
      wv[t] = ok ? (t == 1 ? T(1) : T(0.5)) : T(0);
    }
  } else {
    fv[0] = fv[1] = fv[2] = J0;
    wv[0] = T(1);
    wv[1] = wv[2] = T(0);
  }
  T wc[NTAP][CX][6], wr[NTAP][CX][6];
#pragma unroll
  for (int t = 0; t < NTAP; ++t) {
    const T* gf = gfine + fv[t] * fvol;
    T c2[2][CX], r2[2][CX];
    const int fa[2] = {2 * z0 - 2, 2 * z0 - 1}, fb[2] = {2 * z0, 2 * z0 + 1};
    reduce_pair<T, CX>(gf, fa, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c2, r2);
    put2<T, CX>(wc[t], wr[t], 0, c2, r2);
    reduce_pair<T, CX>(gf, fb, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c2, r2);
    put2<T, CX>(wc[t], wr[t], 2, c2, r2);
  }
  for (int jz = z0; jz < z1; ++jz) {
    T v[CX];
#pragma unroll
    for (int c = 0; c < CX; ++c) v[c] = T(0);
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
      T c2[2][CX], r2[2][CX];
      const int fn2[2] = {2 * jz + 2, 2 * jz + 3};
      reduce_pair<T, CX>(gfine + fv[t] * fvol, fn2, fnz, fplane, fny, fnx, jy, jx0, ay, ax, c2, r2);
      put2<T, CX>(wc[t], wr[t], 4, c2, r2);
#pragma unroll
      for (int c = 0; c < CX; ++c)
        v[c] = v[c] + wv[t] * combine_z<T>(wc[t][c], wr[t][c], jz, cnz, fnz, ax[c].special || ay.special, 0, 0);
      slide<T, CX>(wc[t], wr[t]);
    }
    const int64_t ci = J0 * a.lead_cstride + (int64_t)jz * cplane + (int64_t)jy * cnx + jx0;
#pragma unroll
    for (int c = 0; c < CX; ++c) emit_coarse<T>(gcoarse, gscaled, ci + c, v[c], scale, ad);
  }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static bool march_setup(MarchArgs& m, const InterpArgs& a, int cx, int64_t target_units = 2 * kGridCap) {
  // (1 | '.' batch | 'n'), ('c' | 'n'), 'c', 'c'
  if ((a.loc[1] != kCell && a.loc[1] != kNode) || a.loc[2] != kCell || a.loc[3] != kCell) return false;
  if (a.loc[1] == kNode && a.cut_axis >= 0) return false;
  const bool lead = a.fn[0] != 1;  // a real leading axis: '.ccc' batch or 'nccc'
  if (a.loc[0] == kCell || (lead && (a.loc[1] != kCell || a.cut_axis >= 0 || a.fn[0] > 65535))) return false;
  m.lead_loc = lead ? a.loc[0] : 0;
  m.lead_cn = (int)a.cn[0];
  m.lead_fn = (int)a.fn[0];
  m.lead_cstride = a.coarse_ld ? a.coarse_ld : a.cn[1] * a.cn[2] * a.cn[3];
  if (a.coarse_ld && !lead) return false;  // a strided coarse operand is a 4-D matter
  for (int i = 0; i < 3; ++i) {
    if (a.fn[i + 1] >= (1 << 30)) return false;
    m.cn[i] = (int)a.cn[i + 1];
    m.fn[i] = (int)a.fn[i + 1];
  }
  if (m.cn[0] < 4) return false;  // tiny levels: the per-plane kernel is as good
  m.nt = 0;
  m.cut_lo = a.cut_axis == 1 ? a.cut_lo : 0;
  m.cut_hi = a.cut_axis == 1 ? a.cut_hi : 0;
  if (a.cut_axis >= 0 && a.cut_axis != 1) return false;
  const int groups = (m.cn[2] + cx - 1) / cx;
  int tx = 1;
  while (tx < groups && tx < kBlock) tx *= 2;
  const int64_t xtiles = (groups + tx - 1) / tx;
  m.tx = tx;
  m.ty = kBlock / tx;
  const int64_t ytiles = (m.cn[1] + m.ty - 1) / m.ty;
  if ((int64_t)m.cn[0] * ytiles * xtiles >= ((int64_t)1 << 31)) return false;
  m.usched = make_unit_sched(m.cn[0], ytiles, xtiles, target_units);
  return true;
}

static inline bool aligned_to(const void* p, size_t bytes) { return (reinterpret_cast<uintptr_t>(p) % bytes) == 0; }

// columns per thread: 16 B per lane on the fine array when its rows allow it (pairs of float columns
// need an even coarse count and 16 B aligned arrays); 0: not even the narrow packs are aligned
template <typename T>
static int march_cx(const InterpArgs& a, const void* f0, const void* f1) {
  const bool wide = sizeof(T) == 4 && a.cn[3] % 2 == 0 && a.cn[3] >= 4;
  if (wide && aligned_to(f0, 16) && (!f1 || aligned_to(f1, 16))) return 2;
  return aligned_to(f0, 2 * sizeof(T)) && (!f1 || aligned_to(f1, 2 * sizeof(T))) ? 1 : 0;
}

template <typename T, int CX>
static int add_launch(const T* coarse, const T* add, T* fine, const InterpArgs& a, T cscale, T ascale,
                      hipStream_t stream) {
  MarchArgs m;
  if (!march_setup(m, a, CX)) return 0;
  m.nt = (int64_t)m.lead_fn * m.fn[0] * m.fn[1] * m.fn[2] * (int64_t)sizeof(T) > kStreamBytes;
  const dim3 grid(unit_grid(m.usched), m.lead_fn);
  if (m.lead_fn != 1) {
    if (m.lead_loc == kNode) {
      // even fine indices read one coarse volume, odd ones two
      // (one launch per parity instead: config 5 as one rank, P chain 3.25 ms against 2.70 with pairs)
      if (m.lead_fn == 2 * m.lead_cn - 1 && m.lead_cn >= 2) {
        // pairs (2k, 2k + 1) by one thread, then the last (even) index alone
        const dim3 pairs(unit_grid(m.usched), m.lead_cn - 1), last(unit_grid(m.usched), 1);
        if (!lead_tile_launch<T, 2>(coarse, add, fine, m, cscale, ascale, stream))
          hipLaunchKernelGGL((k_interp_add_march_lead_pair<T, CX>), pairs, dim3(kBlock), 0, stream, coarse, add, fine, m,
                             cscale, ascale);
        hipLaunchKernelGGL((k_interp_add_march_lead<T, CX, 1>), last, dim3(kBlock), 0, stream, coarse, add, fine, m,
                           cscale, ascale, m.lead_fn - 1, 2);
      } else {
        const dim3 even(unit_grid(m.usched), (m.lead_fn + 1) / 2), odd(unit_grid(m.usched), m.lead_fn / 2);
        hipLaunchKernelGGL((k_interp_add_march_lead<T, CX, 1>), even, dim3(kBlock), 0, stream, coarse, add, fine, m,
                           cscale, ascale, 0, 2);
        if (odd.y > 0)
          hipLaunchKernelGGL((k_interp_add_march_lead<T, CX, 2>), odd, dim3(kBlock), 0, stream, coarse, add, fine, m,
                             cscale, ascale, 1, 2);
      }
    } else if (!lead_tile_launch<T, 1>(coarse, add, fine, m, cscale, ascale, stream)) {
      hipLaunchKernelGGL((k_interp_add_march_lead<T, CX, 1>), grid, dim3(kBlock), 0, stream, coarse, add, fine, m,
                         cscale, ascale, 0, 1);
    }
  } else if (a.loc[1] == kNode)
    hipLaunchKernelGGL((k_interp_add_march_n<T, CX>), grid, dim3(kBlock), 0, stream, coarse, add, fine, m, cscale,
                       ascale);
  else if (!lead_tile_launch<T, 1>(coarse, add, fine, m, cscale, ascale, stream))
    hipLaunchKernelGGL((k_interp_add_march<T, CX>), grid, dim3(kBlock), 0, stream, coarse, add, fine, m, cscale,
                       ascale);
  const int e = check_launch("k_interp_add_march");
  return e ? e : 1;
}

// ------------------------------------------------------------------------------------
// P^T of float all-cell volumes whose rows hold <= 256 fine cells: ROW-marching kernel.
//
// out = 2 C - R, where C and R are SEPARABLE sums over the 6 x 6 x 6 fine window of a coarse cell with the per-axis
// weight tables wc / wr of adj6 (what k_interp_adj_march evaluates; away from the walls both are (1, 3, 3, 1) / 4 per
// axis and the result is C).  The tile kernel above stages fine planes through LDS and walks z: 235 VGPRs, 62 KB of
// LDS, two workgroups per compute unit, 2.3 TB/s on the array in float.  Here a lane owns TWO coarse columns of ONE
// coarse plane -- one 16 B pack of a fine row -- and walks the fine ROWS of a y chunk:
//   x: the two fine values left and right of the pack come from the adjacent lanes' registers (wave-wide DPP shift;
//      a row of 256 fine cells is exactly one wavefront, shorter rows share a wavefront between coarse planes);
//   z: the lane loads its pack from the 4 fine planes of its coarse plane (6 next to a z wall) and combines them at
//      once -- every fine value is requested by two waves of the SAME workgroup (or of the next one on the same XCD:
//      the work items of an XCD are consecutive z groups), so the second request is a cache hit;
//   y: the row sums slide through a 6-entry register window; a coarse row is emitted every two fine rows.
// No LDS, no barriers, ~100 VGPRs.  The sums are formed in the order x, z, y (the tile kernel: x, y, z): same terms,
// float rounding differs in the last bits.
// ------------------------------------------------------------------------------------
struct RowsArgs {
  int cn[3], fn[3];
  int lxlog;   // log2 of the lanes per fine row (fnx / 4)
  int yc;      // coarse rows per work item
  int nzg, nyc;
  int64_t total, per_xcd;  // work items (lead x y chunks x z groups), and per XCD
  int64_t fvol, cvol;
};

template <int NPL>
__device__ inline void rows_load(const float* __restrict__ vol, int fy, int cz, int fny, int fnz, int64_t fplane, int fnx,
                                 int lx4, PackN<float, 4> (&g)[NPL]) {
  constexpr int R0 = (6 - NPL) / 2;
  fy = fy < 0 ? 0 : (fy >= fny ? fny - 1 : fy);  // rows / planes beyond the array carry zero weights
  const float* row = vol + (int64_t)fy * fnx + lx4;
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    int fz = 2 * cz - 2 + R0 + p;
    fz = fz < 0 ? 0 : (fz >= fnz ? fnz - 1 : fz);
    g[p] = *reinterpret_cast<const PackN<float, 4>*>(row + (int64_t)fz * fplane);
  }
}

// Wall corrections of one axis: the weight tables of adj6 are the interior pattern (0, 1, 3, 3, 1, 0) / 4 restricted to
// the array, plus 1/4 on ONE tap next to a wall -- C: tap 2 for J = 0, tap 3 for J = n - 1; R: tap 0 for J = 1, tap 5
// for J = n - 2 (the ghost weights w_cell(-1, k) / w_cell(n, k) are 1/4 at k = 0 / k = F - 1 and zero elsewhere).
// Values beyond the array enter as zeros, so the pattern needs no range test.
struct WallTaps {
  float c2, c3, r0, r5;
};
__device__ inline WallTaps wall_taps(int J, int n) {
  WallTaps t;
  t.c2 = J == 0 ? 0.25f : 0.f;
  t.c3 = J == n - 1 ? 0.25f : 0.f;
  t.r0 = J == 1 ? 0.25f : 0.f;
  t.r5 = J == n - 2 ? 0.25f : 0.f;
  return t;
}
// (1, 3, 3, 1) / 4 of the four middle entries of a window of six
__device__ inline float mid4(float v1, float v2, float v3, float v4) {
  return __builtin_fmaf(0.75f, v2 + v3, 0.25f * (v1 + v4));
}

template <int NPL>
__device__ inline void rows_reduce(const PackN<float, 4> (&g)[NPL], bool lo_edge, bool hi_edge, const WallTaps (&xw)[2],
                                   const float (&zwc)[6], const float (&zwr)[6], float (&pc)[2], float (&pr)[2]) {
  float xc[NPL][2], xr[NPL][2];
#pragma unroll
  for (int p = 0; p < NPL; ++p) {
    float v[8];
    v[0] = from_prev_lane(g[p].e[2]);
    v[1] = from_prev_lane(g[p].e[3]);
    v[6] = from_next_lane(g[p].e[0]);
    v[7] = from_next_lane(g[p].e[1]);
    if (lo_edge) v[0] = v[1] = 0.f;  // beyond the row (or another row's / an undefined lane's value)
    if (hi_edge) v[6] = v[7] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[2 + i] = g[p].e[i];
    // the rows are even (cnx = fnx / 2, fnx a power of two): the column next to the low wall is the lane's first
    // (J = 0: tap c2) and the one next to the high wall its second (J = n - 1: c3); J = 1 (r0) is a second, J = n - 2
    // (r5) a first column -- the other four taps of wall_taps() are zero on every lane
    {
      const float s0 = mid4(v[1], v[2], v[3], v[4]), s1 = mid4(v[3], v[4], v[5], v[6]);
      xc[p][0] = __builtin_fmaf(xw[0].c2, v[2], s0);
      xr[p][0] = __builtin_fmaf(xw[0].r5, v[5], s0);
      xc[p][1] = __builtin_fmaf(xw[1].c3, v[5], s1);
      xr[p][1] = __builtin_fmaf(xw[1].r0, v[2], s1);
    }
  }
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if constexpr (NPL == 4) {  // no lane of the wavefront is next to a z wall
      pc[c] = mid4(xc[0][c], xc[1][c], xc[2][c], xc[3][c]);
      pr[c] = mid4(xr[0][c], xr[1][c], xr[2][c], xr[3][c]);
    } else {
      float sc = 0.f, sr = 0.f;
#pragma unroll
      for (int p = 0; p < 6; ++p) {
        sc = __builtin_fmaf(zwc[p], xc[p][c], sc);
        sr = __builtin_fmaf(zwr[p], xr[p][c], sr);
      }
      pc[c] = sc, pr[c] = sr;
    }
  }
}

template <int NPL>
__device__ inline void rows_march(const float* __restrict__ vol, float* __restrict__ gcoarse, float* __restrict__ gscaled,
                                  const RowsArgs& a, int cz, bool active, int y0, int y1, int lx, bool lo_edge, bool hi_edge,
                                  int64_t cbase, float scale, const AdamArgs<float>& ad) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fnz = a.fn[0], fny = a.fn[1], fnx = a.fn[2];
  const int64_t fplane = (int64_t)fny * fnx;
  const WallTaps xw[2] = {wall_taps(2 * lx, cnx), wall_taps(2 * lx + 1, cnx)};
  float zwc[6], zwr[6];
  {
    const Adj6 t = adj6(cz, cnz);
#pragma unroll
    for (int i = 0; i < 6; ++i) zwc[i] = t.wc[i], zwr[i] = t.wr[i];
  }
  float wc[2][6], wr[2][6];  // row sums of fine rows 2 cy - 2 .. 2 cy + 3
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int i = 0; i < 6; ++i) wc[c][i] = wr[c][i] = 0.f;
  // two steps ahead of the first coarse row: the window fills through the same loop body (a separate priming block
  // would hold 4 rows x NPL packs in flight at once -- 256 VGPRs)
#pragma unroll 1
  for (int cy = y0 - 2; cy < y1; ++cy) {
    {
      float pc[2], pr[2];
      if constexpr (NPL == 6) {
        // next to a z wall (few wavefronts): one row at a time, so that this path does not set the register count
#pragma unroll 1
        for (int half = 0; half < 2; ++half) {
          PackN<float, 4> ga[NPL];
          rows_load<NPL>(vol, 2 * cy + 2 + half, cz, fny, fnz, fplane, fnx, 4 * lx, ga);
          rows_reduce<NPL>(ga, lo_edge, hi_edge, xw, zwc, zwr, pc, pr);
          if (half == 0) {
#pragma unroll
            for (int c = 0; c < 2; ++c) wc[c][4] = pc[c], wr[c][4] = pr[c];
          } else {
#pragma unroll
            for (int c = 0; c < 2; ++c) wc[c][5] = pc[c], wr[c][5] = pr[c];
          }
        }
      } else {
        PackN<float, 4> ga[NPL], gb[NPL];
        rows_load<NPL>(vol, 2 * cy + 2, cz, fny, fnz, fplane, fnx, 4 * lx, ga);
        rows_load<NPL>(vol, 2 * cy + 3, cz, fny, fnz, fplane, fnx, 4 * lx, gb);
        rows_reduce<NPL>(ga, lo_edge, hi_edge, xw, zwc, zwr, pc, pr);
#pragma unroll
        for (int c = 0; c < 2; ++c) wc[c][4] = pc[c], wr[c][4] = pr[c];
        rows_reduce<NPL>(gb, lo_edge, hi_edge, xw, zwc, zwr, pc, pr);
#pragma unroll
        for (int c = 0; c < 2; ++c) wc[c][5] = pc[c], wr[c][5] = pr[c];
      }
    }
    if (cy >= y0) {
      float out[2];
      if (cy <= 1 || cy >= cny - 2) {  // next to a y wall (the whole wavefront)
        const Adj6 ay = adj6(cy, cny);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          float sc = 0.f, sr = 0.f;
#pragma unroll
          for (int i = 0; i < 6; ++i) {
            sc = __builtin_fmaf(ay.wc[i], wc[c][i], sc);
            sr = __builtin_fmaf(ay.wr[i], wr[c][i], sr);
          }
          out[c] = 2.f * sc - sr;
        }
      } else {
#pragma unroll
        for (int c = 0; c < 2; ++c)
          out[c] = 2.f * mid4(wc[c][1], wc[c][2], wc[c][3], wc[c][4]) - mid4(wr[c][1], wr[c][2], wr[c][3], wr[c][4]);
      }
      if (active) {
        const int64_t ci = cbase + (int64_t)cy * cnx + 2 * lx;
        if (gscaled || ad.x) {
          emit_coarse<float>(gcoarse, gscaled, ci, out[0], scale, ad);
          emit_coarse<float>(gcoarse, gscaled, ci + 1, out[1], scale, ad);
        } else {
          PackN<float, 2> pk;
          pk.e[0] = out[0], pk.e[1] = out[1];
          *reinterpret_cast<PackN<float, 2>*>(gcoarse + ci) = pk;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) wc[c][i] = wc[c][i + 2], wr[c][i] = wr[c][i + 2];
  }
}

__global__ __launch_bounds__(kBlock) void k_interp_adj_rows(const float* __restrict__ gfine, float* __restrict__ gcoarse,
                                                            float* __restrict__ gscaled, RowsArgs a, float scale,
                                                            AdamArgs<float> ad) {
  // consecutive work items (z groups fastest, then y chunks, then volumes) on ONE XCD: the fine planes and rows that
  // neighbouring items both read meet in that XCD's L2
  const int xcd = blockIdx.x % kNumXcd;
  const int64_t slot = blockIdx.x / kNumXcd, w = xcd * a.per_xcd + slot;
  if (w >= a.total) return;
  const int zg = (int)(w % a.nzg);
  const int64_t r = w / a.nzg;
  const int yc = (int)(r % a.nyc);
  const int64_t lead = r / a.nyc;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int sub = lane >> a.lxlog, lx = lane & ((1 << a.lxlog) - 1), pw = 64 >> a.lxlog;
  const int cz_first = (zg * (kBlock / 64) + wave) * pw;
  if (cz_first >= a.cn[0]) return;  // the whole wavefront is beyond the last coarse plane
  const bool active = cz_first + sub < a.cn[0];
  const int cz = active ? cz_first + sub : a.cn[0] - 1;  // (idle lanes keep the shifts of their neighbours defined)
  const int y0 = yc * a.yc, y1 = y0 + a.yc < a.cn[1] ? y0 + a.yc : a.cn[1];
  const bool lo_edge = lx == 0, hi_edge = lx == (1 << a.lxlog) - 1;
  const float* vol = gfine + lead * a.fvol;
  const int64_t cbase = lead * a.cvol + (int64_t)cz * a.cn[1] * a.cn[2];
  const bool z_wall = cz <= 1 || cz >= a.cn[0] - 2;
  // (the six-plane path as its own launch behind this one -- 83 / 132 VGPRs, five waves per SIMD for the interior --
  // was measured SLOWER, 0.363 against 0.297 ms at 128 x (32, 256, 256): its few workgroups are a serial tail; with
  // that path one row at a time the single kernel holds 120 VGPRs instead of 163: 0.293 ms)
  if (__any(z_wall))
    rows_march<6>(vol, gcoarse, gscaled, a, cz, active, y0, y1, lx, lo_edge, hi_edge, cbase, scale, ad);
  else
    rows_march<4>(vol, gcoarse, gscaled, a, cz, active, y0, y1, lx, lo_edge, hi_edge, cbase, scale, ad);
}

static bool adj_rows_enabled() {
  const char* e = getenv("ODIL_ADJ_ROWS");
  return !e || atoi(e) != 0;
}

// Launches k_interp_adj_rows when the volumes qualify; false: the caller keeps its kernel.
template <typename T>
static bool adj_rows_launch(const T* gfine, T* gcoarse, T* gscaled, const MarchArgs& m, T scale, hipStream_t stream,
                            const AdamArgs<T>& ad) {
  if constexpr (sizeof(T) != 4) {
    return false;
  } else {
    const int fnx = m.fn[2];
    if (!adj_rows_enabled() || m.cut_lo || m.cut_hi || m.lead_loc == kNode) return false;
    if (fnx != 64 && fnx != 128 && fnx != 256) return false;
    for (int i = 0; i < 3; ++i)
      if (m.fn[i] != 2 * m.cn[i] || m.cn[i] < 4) return false;
    if (!aligned_to(gfine, 16) || !aligned_to(gcoarse, 8)) return false;
    // results are stored as 8-byte packs at gcoarse + lead * cvol + ...: an odd leading stride (a strided view handed
    // to odil_interp_adj_ld) would misalign every odd leading index
    if (m.lead_fn > 1 && (m.lead_cstride % 2) != 0) return false;
    RowsArgs r;
    for (int i = 0; i < 3; ++i) r.cn[i] = m.cn[i], r.fn[i] = m.fn[i];
    r.lxlog = fnx == 256 ? 6 : (fnx == 128 ? 5 : 4);
    const int planes_per_group = (kBlock / 64) * (64 >> r.lxlog);
    r.nzg = (m.cn[0] + planes_per_group - 1) / planes_per_group;
    const int64_t lead = m.lead_fn > 1 ? m.lead_fn : 1;
    // long y chunks (every chunk loads four rows beyond its own) as long as the launch keeps >= 4096 work items
    r.yc = 32;
    while (r.yc > 8 && lead * r.nzg * ((m.cn[1] + r.yc - 1) / r.yc) < 4096) r.yc /= 2;
    r.nyc = (m.cn[1] + r.yc - 1) / r.yc;
    r.total = lead * r.nzg * r.nyc;
    r.per_xcd = (r.total + kNumXcd - 1) / kNumXcd;
    if (r.per_xcd * kNumXcd >= ((int64_t)1 << 31)) return false;
    r.fvol = (int64_t)m.fn[0] * m.fn[1] * m.fn[2];
    r.cvol = m.lead_cstride;
    hipLaunchKernelGGL(k_interp_adj_rows, dim3((unsigned)(r.per_xcd * kNumXcd)), dim3(kBlock), 0, stream, gfine, gcoarse,
                       gscaled, r, scale, ad);
    return true;
  }
}

// ODIL_ADJ_TILE=0 keeps the register-window kernel on every level (read per call: the tests compare both)
template <typename T>
static void launch_adj_tile(dim3 grid, bool wide, hipStream_t stream, const T* gfine, T* gcoarse, T* gscaled,
                            const MarchArgs& m, T scale, const AdamArgs<T>& ad) {
  if constexpr (sizeof(T) == 4) {
    if (wide) {
      hipLaunchKernelGGL((k_interp_adj_tile<T, 2>), grid, dim3(kBlock), 0, stream, gfine, gcoarse, gscaled, m, scale, ad);
      return;
    }
  }
  hipLaunchKernelGGL((k_interp_adj_tile<T, 1>), grid, dim3(kBlock), 0, stream, gfine, gcoarse, gscaled, m, scale, ad);
}

static bool adj_tile_enabled() {
  const char* e = getenv("ODIL_ADJ_TILE");
  return !e || atoi(e) != 0;
}

template <typename T, int CX>
static int adj_launch(const T* gfine, T* gcoarse, T* gscaled, const InterpArgs& a, T scale, hipStream_t stream,
                      const AdamArgs<T>& ad) {
  MarchArgs m;
  // every chunk primes its window with two extra fine-plane reductions: prefer chunks of >= 8 planes
  const bool small_level = a.cn[0] * a.cn[1] * a.cn[2] * a.cn[3] <= ((int64_t)1 << 22);
  if (!march_setup(m, a, CX, small_level ? ODIL_ADJ_UNITS_SMALL : ODIL_ADJ_UNITS)) return 0;
  const dim3 grid(unit_grid(m.usched), m.lead_cn);
  // (a strided coarse result is written by the kernels that index it through m.lead_cstride only, and never with a
  // scaled copy / an update of arrays of another layout: the caller checks the latter)
  const bool tile_ok = CX == 1 && m.cn[1] >= 2 * kTileY && m.cn[2] >= 2 * kTileX && adj_tile_enabled() && !a.coarse_ld;
  // float: two coarse columns per thread (16 B per lane everywhere) when the rows allow it
  const bool tile_wide = sizeof(T) == 4 && m.cn[2] % 2 == 0 && m.cn[2] >= 4 * kTileX && aligned_to(gfine, 16);
  const int tile_cols = kTileX * (tile_wide ? 2 : 1);
  if (m.lead_fn != 1) {
    if (m.lead_loc == kNode)
      hipLaunchKernelGGL((k_interp_adj_march_lead<T, CX, 3>), grid, dim3(kBlock), 0, stream, gfine, gcoarse, gscaled,
                         m, scale, ad);
    else if (adj_rows_launch<T>(gfine, gcoarse, gscaled, m, scale, stream, ad)) {
    } else if (tile_ok) {
      // batch of large all-cell volumes: the LDS-staged kernel, one volume per blockIdx.y
      const int64_t ytiles = (m.cn[1] + kTileY - 1) / kTileY, xtiles = (m.cn[2] + tile_cols - 1) / tile_cols;
      m.tx = kTileX;
      m.ty = kTileY;
      // the batch supplies the parallelism: long chunks (every chunk primes its window with two extra pairs of planes)
      const int64_t per_volume = kGridCap / m.lead_cn > 2 * ytiles * xtiles ? kGridCap / m.lead_cn : 2 * ytiles * xtiles;
      m.usched = make_unit_sched(m.cn[0], ytiles, xtiles, per_volume);
      launch_adj_tile<T>(dim3(unit_grid(m.usched), m.lead_cn), tile_wide, stream, gfine, gcoarse, gscaled, m, scale, ad);
    } else
      hipLaunchKernelGGL((k_interp_adj_march_lead<T, CX, 1>), grid, dim3(kBlock), 0, stream, gfine, gcoarse, gscaled,
                         m, scale, ad);
  } else if (a.loc[1] == kNode)
    hipLaunchKernelGGL((k_interp_adj_march_n<T, CX>), grid, dim3(kBlock), 0, stream, gfine, gcoarse, gscaled, m, scale,
                       ad);
  else if (adj_rows_launch<T>(gfine, gcoarse, gscaled, m, scale, stream, ad)) {
  } else if (tile_ok) {
    // large all-cell levels: fine planes staged through LDS
    const int64_t ytiles = (m.cn[1] + kTileY - 1) / kTileY, xtiles = (m.cn[2] + tile_cols - 1) / tile_cols;
    m.tx = kTileX;
    m.ty = kTileY;
    m.usched = make_unit_sched(m.cn[0], ytiles, xtiles, small_level ? ODIL_ADJ_UNITS_SMALL : ODIL_TILE_UNITS);
    launch_adj_tile<T>(dim3(unit_grid(m.usched)), tile_wide, stream, gfine, gcoarse, gscaled, m, scale, ad);
  } else
    hipLaunchKernelGGL((k_interp_adj_march<T, CX>), grid, dim3(kBlock), 0, stream, gfine, gcoarse, gscaled, m, scale,
                       ad);
  const int e = check_launch("k_interp_adj_march");
  return e ? e : 1;
}

template <typename T>
int poisson_adjoint_transpose(const T* fu, T* g0, T* g1, const int64_t* fshape, const T* h2, T scale,
                              const AdamArgs<T>& ad0, const AdamArgs<T>& ad1, int cut_lo, int cut_hi,
                              hipStream_t stream) {
  if (!fu || !g1 || !fshape || !h2) {
    set_error("poisson_adjoint_transpose: null pointer (fu, g1, shape, h2)");
    return ODIL_E_INVAL;
  }
  MarchArgs m;
  for (int i = 0; i < 3; ++i) {
    if (fshape[i] < 4 || fshape[i] % 2 || fshape[i] >= (1 << 30)) {
      set_error("poisson_adjoint_transpose: extent %lld on axis %d must be even, >= 4", (long long)fshape[i], i);
      return ODIL_E_INVAL;
    }
    m.fn[i] = (int)fshape[i];
    m.cn[i] = (int)(fshape[i] / 2);
  }
  const void* ptrs[] = {fu, g0, g1, ad0.x, ad0.m, ad0.v};
  for (const void* q : ptrs)
    if (q && !aligned_to(q, 2 * sizeof(T))) {
      set_error("poisson_adjoint_transpose: arrays must be aligned to %d bytes", (int)(2 * sizeof(T)));
      return ODIL_E_INVAL;
    }
  m.tx = kTileX, m.ty = kTileY;
  m.cut_lo = cut_lo, m.cut_hi = cut_hi;
  m.nt = 0;
  m.lead_loc = 0, m.lead_cn = 1, m.lead_fn = 1;
  const int64_t ytiles = (m.cn[1] + kTileY - 1) / kTileY, xtiles = (m.cn[2] + kTileX - 1) / kTileX;
  if ((int64_t)m.cn[0] * ytiles * xtiles >= ((int64_t)1 << 31)) {
    set_error("poisson_adjoint_transpose: grid too large for one launch");
    return ODIL_E_INVAL;
  }
  // workgroups per launch: 512^3 epoch 2.84 / 2.80 / 2.75 / 2.81 / 3.03 ms for 1024 / 2048 / 4096 / 8192 / 16384 (shorter
  // chunks re-prime their windows more often, longer ones leave the phases of resident workgroups in step)
  m.usched = make_unit_sched(m.cn[0], ytiles, xtiles, 2 * kGridCap);
  T h[3] = {h2[0], h2[1], h2[2]};
  const H2<T> hh = make_h2<T>(h);
  // (the C ABI passes one set of hyper-parameters for both levels)
  const bool mul = hh.mul_ok[0] && hh.mul_ok[1] && hh.mul_ok[2];
  const dim3 grid(unit_grid(m.usched));
#define ODIL_LAUNCH_ADJ_TILE(MUL, G0)                                                                             \
  hipLaunchKernelGGL((k_poisson_adjoint_tile<T, MUL, G0>), grid, dim3(kBlock), 0, stream, fu, g0, g1, m, hh, scale, \
                     ad0, ad1.x, ad1.m, ad1.v)
  if (mul && !g0) ODIL_LAUNCH_ADJ_TILE(true, false);
  else if (mul) ODIL_LAUNCH_ADJ_TILE(true, true);
  else if (!g0) ODIL_LAUNCH_ADJ_TILE(false, false);
  else ODIL_LAUNCH_ADJ_TILE(false, true);
#undef ODIL_LAUNCH_ADJ_TILE
  return check_launch("k_poisson_adjoint_tile");
}

template int poisson_adjoint_transpose<double>(const double*, double*, double*, const int64_t*, const double*, double,
                                               const AdamArgs<double>&, const AdamArgs<double>&, int, int, hipStream_t);
template int poisson_adjoint_transpose<float>(const float*, float*, float*, const int64_t*, const float*, float,
                                              const AdamArgs<float>&, const AdamArgs<float>&, int, int, hipStream_t);

template <typename T>
int interp_add_march(const T* coarse, const T* add, T* fine, const InterpArgs& a, T cscale, T ascale,
                     hipStream_t stream) {
  const int cx = march_cx<T>(a, fine, add);
  if constexpr (sizeof(T) == 4) {
    if (cx == 2) return add_launch<T, 2>(coarse, add, fine, a, cscale, ascale, stream);
  }
  return cx ? add_launch<T, 1>(coarse, add, fine, a, cscale, ascale, stream) : 0;
}

template <typename T>
int interp_adj_march(const T* gfine, T* gcoarse, T* gscaled, const InterpArgs& a, T scale, hipStream_t stream,
                     const AdamArgs<T>& ad) {
  return march_cx<T>(a, gfine, nullptr) ? adj_launch<T, 1>(gfine, gcoarse, gscaled, a, scale, stream, ad) : 0;
}

template int interp_add_march<double>(const double*, const double*, double*, const InterpArgs&, double, double,
                                      hipStream_t);
template int interp_add_march<float>(const float*, const float*, float*, const InterpArgs&, float, float, hipStream_t);
template int interp_adj_march<double>(const double*, double*, double*, const InterpArgs&, double, hipStream_t,
                                      const AdamArgs<double>&);
template int interp_adj_march<float>(const float*, float*, float*, const InterpArgs&, float, hipStream_t,
                                     const AdamArgs<float>&);

}  // namespace odil

using namespace odil;

extern "C" {
int odil_poisson_adjoint_transpose_adam_f64(const double* fu, double* g0, double* g1, const int64_t* fshape,
                                            const double* h2, double scale, double* x0, double* m0, double* v0,
                                            double* x1, double* m1, double* v1, double alpha, double one_minus_b1,
                                            double one_minus_b2, double eps, const double* alpha_dev, int cut_lo,
                                            int cut_hi, void* stream) {
  const AdamArgs<double> a0{x0, m0, v0, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev};
  const AdamArgs<double> a1{x1, m1, v1, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev};
  return poisson_adjoint_transpose<double>(fu, g0, g1, fshape, h2, scale, a0, a1, cut_lo, cut_hi, (hipStream_t)stream);
}
int odil_poisson_adjoint_transpose_adam_f32(const float* fu, float* g0, float* g1, const int64_t* fshape,
                                            const float* h2, float scale, float* x0, float* m0, float* v0, float* x1,
                                            float* m1, float* v1, float alpha, float one_minus_b1, float one_minus_b2,
                                            float eps, const float* alpha_dev, int cut_lo, int cut_hi, void* stream) {
  const AdamArgs<float> a0{x0, m0, v0, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev};
  const AdamArgs<float> a1{x1, m1, v1, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev};
  return poisson_adjoint_transpose<float>(fu, g0, g1, fshape, h2, scale, a0, a1, cut_lo, cut_hi, (hipStream_t)stream);
}
}
