// z-marching multigrid transfers for the 3-D layouts 'ccc' (the Poisson hot path) and 'ncc'
// (node-centred leading axis: the space-time fields of the tracer workload).
// Same arithmetic as mg_fast.hip / mg_transfer.hip; what changes is the data movement:
//
// P  : a thread owns a coarse column (jy, jx) and walks coarse planes, holding the 3x3x3
//      coarse neighbourhood in registers (9 new values per plane); every step emits the
//      two fine planes 2jz, 2jz+1 as 16 B stores.  Fine traffic is touched exactly once.
// P^T: a thread owns a coarse column and walks coarse planes; each FINE plane is reduced
//      over its (y, x) window once (rows as three 16 B packs) and the per-plane sums slide
//      through a 6-entry register window, so every fine value is loaded once per owner.
#include "mg_transfer.h"

namespace odil {

template <typename T>
struct alignas(2 * sizeof(T)) Pack2 {
  T a, b;
};

// touch-once streams (fine addend in, fine result out) bypass cache retention when the fine array is
// larger than the last-level cache could hand to the next kernel anyway (MarchArgs::nt)
template <typename T>
__device__ inline Pack2<T> stream_ld(const Pack2<T>* p, bool nt) {
  typedef T VT __attribute__((ext_vector_type(2)));
  const VT v = nt ? __builtin_nontemporal_load(reinterpret_cast<const VT*>(p)) : *reinterpret_cast<const VT*>(p);
  Pack2<T> r;
  r.a = v[0];
  r.b = v[1];
  return r;
}
template <typename T>
__device__ inline void stream_st(Pack2<T>* p, const Pack2<T>& x, bool nt) {
  typedef T VT __attribute__((ext_vector_type(2)));
  VT v;
  v[0] = x.a;
  v[1] = x.b;
  if (nt)
    __builtin_nontemporal_store(v, reinterpret_cast<VT*>(p));
  else
    *reinterpret_cast<VT*>(p) = v;
}

struct MarchArgs {
  int cn[3], fn[3];  // (z, y, x) coarse / fine extents
  int tx, ty;
  int cut_lo, cut_hi;  // z end is an interior slab interface (ghost planes), not a wall
  int nt;              // stream the fine array past the caches (it exceeds kStreamBytes)
  int lead_loc, lead_cn, lead_fn;  // 4-D layouts: leading axis kind ('.' batch or 'n') and its extents
  UnitSched usched;
};

struct Tap3 {
  int cl[3], rf[3];
  bool out[3];
};

__device__ inline Tap3 tap3(int j, int n) {
  Tap3 t;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int q = j + d - 1;
    t.out[d] = q < 0 || q >= n;
    t.cl[d] = q < 0 ? 0 : (q >= n ? n - 1 : q);
    t.rf[d] = q < 0 ? 1 : (q >= n ? n - 2 : q);
  }
  return t;
}

// 3x3 coarse values of (padded) plane q in [-1, n]: ghosts by the joint rule (core.py:640-643).
template <typename T>
__device__ inline void load_plane9(const T* __restrict__ coarse, int q, int cnz, int64_t cplane, int cnx,
                                   const Tap3& ty, const Tap3& tx, T cscale, T v[3][3]) {
  const bool oz = q < 0 || q >= cnz;
  const int zcl = q < 0 ? 0 : (q >= cnz ? cnz - 1 : q);
  const int zrf = q < 0 ? 1 : (q >= cnz ? cnz - 2 : q);
  const T* ccl = coarse + zcl * cplane;
  const T* crf = coarse + zrf * cplane;
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      T val = cscale * ccl[(int64_t)ty.cl[dy] * cnx + tx.cl[dx]];
      if (oz || ty.out[dy] || tx.out[dx]) val = T(2) * val - cscale * crf[(int64_t)ty.rf[dy] * cnx + tx.rf[dx]];
      v[dy][dx] = val;
    }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_add_march(const T* __restrict__ coarse, const T* __restrict__ add,
                                                             T* __restrict__ fine, MarchArgs a, T cscale, T ascale) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
  if (jy >= cny || jx >= cnx) return;
  const int z0 = zc * a.usched.ZC;
  const int z1 = z0 + a.usched.ZC < cnz ? z0 + a.usched.ZC : cnz;
  const Tap3 tx = tap3(jx, cnx), ty = tap3(jy, cny);
  T v[3][3][3];
  load_plane9<T>(coarse, z0 - 1, cnz, cplane, cnx, ty, tx, cscale, v[0]);
  load_plane9<T>(coarse, z0, cnz, cplane, cnx, ty, tx, cscale, v[1]);
  const T r64 = T(1) / T(64);  // exact: sum of weights 4*4*4
  for (int jz = z0; jz < z1; ++jz) {
    // issue the fine-grid addend loads first: they are the HBM stream of this kernel
    Pack2<T> ad[2][2];
    const int64_t fbase = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx;
    if (add) {
#pragma unroll
      for (int sz = 0; sz < 2; ++sz)
#pragma unroll
        for (int sy = 0; sy < 2; ++sy)
          ad[sz][sy] = stream_ld(reinterpret_cast<const Pack2<T>*>(add + fbase + sz * fplane + (int64_t)sy * fnx), a.nt);
    }
    load_plane9<T>(coarse, jz + 1, cnz, cplane, cnx, ty, tx, cscale, v[2]);
#pragma unroll
    for (int sz = 0; sz < 2; ++sz)
#pragma unroll
      for (int sy = 0; sy < 2; ++sy) {
        T o[2];
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          // reference order (rz, ry, rx), rx fastest; weights per axis: parity == r ? 1 : 3
          T s = T(0);
#pragma unroll
          for (int rz = 0; rz < 2; ++rz)
#pragma unroll
            for (int ry = 0; ry < 2; ++ry)
#pragma unroll
              for (int rx = 0; rx < 2; ++rx) {
                const int w = (sz == rz ? 1 : 3) * (sy == ry ? 1 : 3) * (sx == rx ? 1 : 3);
                s = s + T(w) * v[sz + rz][sy + ry][sx + rx];
              }
          o[sx] = s * r64;
        }
        if (add) {
          o[0] = ascale * ad[sz][sy].a + o[0];
          o[1] = ascale * ad[sz][sy].b + o[1];
        }
        Pack2<T> pk;
        pk.a = o[0];
        pk.b = o[1];
        stream_st(reinterpret_cast<Pack2<T>*>(fine + fbase + sz * fplane + (int64_t)sy * fnx), pk, a.nt);
      }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        v[0][dy][dx] = v[1][dy][dx];
        v[1][dy][dx] = v[2][dy][dx];
      }
  }
}

// 4-D layouts ('.ccc' batches and 'nccc': space-time fields with three space dimensions): the walk
// of k_interp_add_march per FINE leading index f0 (blockIdx.y).  On a node-centred leading axis an
// odd f0 averages the two neighbouring coarse volumes: both 3x3x3 windows are held and summed in
// the reference's order (leading tap outermost).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_add_march_lead(const T* __restrict__ coarse,
                                                                  const T* __restrict__ add, T* __restrict__ fine,
                                                                  MarchArgs a, T cscale, T ascale) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int64_t cvol = (int64_t)cnz * cplane, fvol = (int64_t)a.fn[0] * fplane;
  const int f0 = blockIdx.y;
  const bool node = a.lead_loc == kNode;
  const int cnt0 = node && (f0 & 1) ? 2 : 1;
  const int c0 = node ? f0 >> 1 : f0;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
  if (jy >= cny || jx >= cnx) return;
  const int z0 = zc * a.usched.ZC;
  const int z1 = z0 + a.usched.ZC < cnz ? z0 + a.usched.ZC : cnz;
  const Tap3 tx = tap3(jx, cnx), ty = tap3(jy, cny);
  const T* cb[2] = {coarse + c0 * cvol, coarse + (c0 + cnt0 - 1) * cvol};
  add = add ? add + f0 * fvol : add;
  fine += f0 * fvol;
  T v[2][3][3][3];
#pragma unroll
  for (int r0 = 0; r0 < 2; ++r0) {
    if (r0 >= cnt0) break;
    load_plane9<T>(cb[r0], z0 - 1, cnz, cplane, cnx, ty, tx, cscale, v[r0][0]);
    load_plane9<T>(cb[r0], z0, cnz, cplane, cnx, ty, tx, cscale, v[r0][1]);
  }
  const T rs = T(1) / T(64 * cnt0);
  for (int jz = z0; jz < z1; ++jz) {
    Pack2<T> ad[2][2];
    const int64_t fbase = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx;
    if (add) {
#pragma unroll
      for (int sz = 0; sz < 2; ++sz)
#pragma unroll
        for (int sy = 0; sy < 2; ++sy)
          ad[sz][sy] = stream_ld(reinterpret_cast<const Pack2<T>*>(add + fbase + sz * fplane + (int64_t)sy * fnx), a.nt);
    }
#pragma unroll
    for (int r0 = 0; r0 < 2; ++r0) {
      if (r0 >= cnt0) break;
      load_plane9<T>(cb[r0], jz + 1, cnz, cplane, cnx, ty, tx, cscale, v[r0][2]);
    }
#pragma unroll
    for (int sz = 0; sz < 2; ++sz)
#pragma unroll
      for (int sy = 0; sy < 2; ++sy) {
        T o[2];
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          T s = T(0);
#pragma unroll
          for (int r0 = 0; r0 < 2; ++r0) {
            if (r0 >= cnt0) break;
#pragma unroll
            for (int rz = 0; rz < 2; ++rz)
#pragma unroll
              for (int ry = 0; ry < 2; ++ry)
#pragma unroll
                for (int rx = 0; rx < 2; ++rx) {
                  const int w = (sz == rz ? 1 : 3) * (sy == ry ? 1 : 3) * (sx == rx ? 1 : 3);
                  s = s + T(w) * v[r0][sz + rz][sy + ry][sx + rx];
                }
          }
          o[sx] = s * rs;
        }
        if (add) {
          o[0] = ascale * ad[sz][sy].a + o[0];
          o[1] = ascale * ad[sz][sy].b + o[1];
        }
        Pack2<T> pk;
        pk.a = o[0];
        pk.b = o[1];
        stream_st(reinterpret_cast<Pack2<T>*>(fine + fbase + sz * fplane + (int64_t)sy * fnx), pk, a.nt);
      }
#pragma unroll
    for (int r0 = 0; r0 < 2; ++r0)
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          v[r0][0][dy][dx] = v[r0][1][dy][dx];
          v[r0][1][dy][dx] = v[r0][2][dy][dx];
        }
  }
}

// Same walk for a NODE-centred leading axis ('ncc': time-like axis of the space-time workloads):
// fine plane 2jz is coarse plane jz interpolated in (y, x) only, fine plane 2jz+1 the mean of the
// two neighbouring coarse planes; ghosts exist on the two cell axes only.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_add_march_n(const T* __restrict__ coarse,
                                                               const T* __restrict__ add, T* __restrict__ fine,
                                                               MarchArgs a, T cscale, T ascale) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
  if (jy >= cny || jx >= cnx) return;
  const int z0 = zc * a.usched.ZC;
  const int z1 = z0 + a.usched.ZC < cnz ? z0 + a.usched.ZC : cnz;
  const Tap3 tx = tap3(jx, cnx), ty = tap3(jy, cny);
  T v[2][3][3];
  load_plane9<T>(coarse, z0, cnz, cplane, cnx, ty, tx, cscale, v[0]);
  const T r16 = T(1) / T(16), r32 = T(1) / T(32);
  for (int jz = z0; jz < z1; ++jz) {
    const bool odd = jz + 1 < cnz;  // the last coarse plane has no fine plane above it
    Pack2<T> ad[2][2];
    const int64_t fbase = (int64_t)(2 * jz) * fplane + (int64_t)(2 * jy) * fnx + 2 * jx;
    if (add) {
#pragma unroll
      for (int sz = 0; sz < 2; ++sz)
#pragma unroll
        for (int sy = 0; sy < 2; ++sy)
          if (sz == 0 || odd)
            ad[sz][sy] = stream_ld(reinterpret_cast<const Pack2<T>*>(add + fbase + sz * fplane + (int64_t)sy * fnx), a.nt);
    }
    if (odd) load_plane9<T>(coarse, jz + 1, cnz, cplane, cnx, ty, tx, cscale, v[1]);
#pragma unroll
    for (int sz = 0; sz < 2; ++sz) {
      if (sz == 1 && !odd) break;
#pragma unroll
      for (int sy = 0; sy < 2; ++sy) {
        T o[2];
#pragma unroll
        for (int sx = 0; sx < 2; ++sx) {
          T s = T(0);
#pragma unroll
          for (int rz = 0; rz <= sz; ++rz)
#pragma unroll
            for (int ry = 0; ry < 2; ++ry)
#pragma unroll
              for (int rx = 0; rx < 2; ++rx) {
                const int w = (sy == ry ? 1 : 3) * (sx == rx ? 1 : 3);
                s = s + T(w) * v[rz][sy + ry][sx + rx];
              }
          o[sx] = s * (sz ? r32 : r16);
        }
        if (add) {
          o[0] = ascale * ad[sz][sy].a + o[0];
          o[1] = ascale * ad[sz][sy].b + o[1];
        }
        Pack2<T> pk;
        pk.a = o[0];
        pk.b = o[1];
        stream_st(reinterpret_cast<Pack2<T>*>(fine + fbase + sz * fplane + (int64_t)sy * fnx), pk, a.nt);
      }
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) v[0][dy][dx] = v[1][dy][dx];
  }
}

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

// (y, x) reduction of fine planes for the coarse column (jy, jx): rc with the C weights, rr
// with the R weights.  All row loads of a call are issued back to back (addresses clamped into
// range; out-of-range rows / pairs carry zero weights), and only then reduced: the kernel is
// bound by how many HBM requests a wave keeps in flight, so loads must not sit behind branches.
template <typename T, int NR, int NP>
__device__ inline void reduce_planes(const T* __restrict__ gfine, const int (&f)[NP], int fnz, int64_t fplane, int fny,
                                     int fnx, int jy, int jx, const Adj6& ay, const Adj6& ax, T (&rc)[NP],
                                     T (&rr)[NP]) {
  constexpr int R0 = (6 - NR) / 2;  // first row of the window that is loaded (NR = 4: rows 1..4)
  Pack2<T> g[NP][NR][3];
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
        int fx = 2 * (jx - 1 + q);
        fx = fx < 0 ? 0 : (fx >= fnx ? fnx - 2 : fx);
        g[p][r][q] = *reinterpret_cast<const Pack2<T>*>(row + fx);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    T sc = T(0), sr = T(0);
    const bool inside = f[p] >= 0 && f[p] < fnz;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      T xc = T(0), xr = T(0);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        xc = xc + T(ax.wc[2 * q]) * g[p][r][q].a;
        xc = xc + T(ax.wc[2 * q + 1]) * g[p][r][q].b;
        xr = xr + T(ax.wr[2 * q]) * g[p][r][q].a;
        xr = xr + T(ax.wr[2 * q + 1]) * g[p][r][q].b;
      }
      sc = sc + T(ay.wc[R0 + r]) * xc;
      sr = sr + T(ay.wr[R0 + r]) * xr;
    }
    rc[p] = inside ? sc : T(0);
    rr[p] = inside ? sr : T(0);
  }
}

template <typename T, int NP>
__device__ inline void reduce_dispatch(const T* __restrict__ gfine, const int (&f)[NP], int fnz, int64_t fplane,
                                       int fny, int fnx, int jy, int jx, const Adj6& ay, const Adj6& ax, T (&rc)[NP],
                                       T (&rr)[NP]) {
  if (ay.special) {
    // boundary rows need the 6-row window: one plane at a time keeps the register count down
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int f1[1] = {f[p]};
      T c1[1], r1[1];
      reduce_planes<T, 6, 1>(gfine, f1, fnz, fplane, fny, fnx, jy, jx, ay, ax, c1, r1);
      rc[p] = c1[0];
      rr[p] = r1[0];
    }
  } else {
    reduce_planes<T, 4, NP>(gfine, f, fnz, fplane, fny, fnx, jy, jx, ay, ax, rc, rr);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_adj_march(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                             T* __restrict__ gscaled, MarchArgs a, T scale,
                                                             AdamArgs<T> ad) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fnz = a.fn[0], fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
  if (jy >= cny || jx >= cnx) return;
  const int z0 = zc * a.usched.ZC;
  const int z1 = z0 + a.usched.ZC < cnz ? z0 + a.usched.ZC : cnz;
  const Adj6 ax = adj6(jx, cnx), ay = adj6(jy, cny);
  const bool xy_special = ax.special || ay.special;
  // window of plane sums for fine planes 2jz-2 .. 2jz+3
  T wc[6], wr[6];
  {
    T c2[2], r2[2];
    const int fa[2] = {2 * z0 - 2, 2 * z0 - 1}, fb[2] = {2 * z0, 2 * z0 + 1};
    reduce_dispatch<T, 2>(gfine, fa, fnz, fplane, fny, fnx, jy, jx, ay, ax, c2, r2);
    wc[0] = c2[0], wc[1] = c2[1], wr[0] = r2[0], wr[1] = r2[1];
    reduce_dispatch<T, 2>(gfine, fb, fnz, fplane, fny, fnx, jy, jx, ay, ax, c2, r2);
    wc[2] = c2[0], wc[3] = c2[1], wr[2] = r2[0], wr[3] = r2[1];
  }
  for (int jz = z0; jz < z1; ++jz) {
    {
      T c2[2], r2[2];
      const int fn2[2] = {2 * jz + 2, 2 * jz + 3};
      reduce_dispatch<T, 2>(gfine, fn2, fnz, fplane, fny, fnx, jy, jx, ay, ax, c2, r2);
      wc[4] = c2[0], wc[5] = c2[1], wr[4] = r2[0], wr[5] = r2[1];
    }
    const bool z_special = ((jz == 0 || jz == 1) && !a.cut_lo) || ((jz == cnz - 2 || jz == cnz - 1) && !a.cut_hi);
    T v;
    if (!z_special && !xy_special) {
      v = (T(0.25) * wc[1] + T(0.75) * wc[2]) + (T(0.75) * wc[3] + T(0.25) * wc[4]);
    } else {
      T sc = T(0), sr = T(0);
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float zc_w, zr_w;
        const int k = 2 * jz - 2 + i;
        const float w = w_cell(jz, k, fnz), lo = w_cell(-1, k, fnz), hi = w_cell(cnz, k, fnz);
        zc_w = w + (jz == 0 && !a.cut_lo ? lo : 0.f) + (jz == cnz - 1 && !a.cut_hi ? hi : 0.f);
        zr_w = w + (jz == 1 && !a.cut_lo ? lo : 0.f) + (jz == cnz - 2 && !a.cut_hi ? hi : 0.f);
        sc = sc + T(zc_w) * wc[i];
        sr = sr + T(zr_w) * wr[i];
      }
      v = T(2) * sc - sr;
    }
    const int64_t ci = (int64_t)jz * cplane + (int64_t)jy * cnx + jx;
    gcoarse[ci] = v;
    if (gscaled) gscaled[ci] = scale * v;
    if (ad.x) {  // Adam of this level's array by the lane that formed its gradient
      T xv = ad.x[ci], mv = ad.m[ci], vv = ad.v[ci];
      adam_update<T>(xv, mv, vv, gscaled ? scale * v : v, ad);
      ad.x[ci] = xv;
      ad.m[ci] = mv;
      ad.v[ci] = vv;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wc[i] = wc[i + 2];
      wr[i] = wr[i + 2];
    }
  }
}

// z-combination of one window of plane sums (the step of k_interp_adj_march).
template <typename T>
__device__ inline T combine_z(const T (&wc)[6], const T (&wr)[6], int jz, int cnz, int fnz, bool xy_special,
                              int cut_lo, int cut_hi) {
  const bool z_special = ((jz == 0 || jz == 1) && !cut_lo) || ((jz == cnz - 2 || jz == cnz - 1) && !cut_hi);
  if (!z_special && !xy_special) return (T(0.25) * wc[1] + T(0.75) * wc[2]) + (T(0.75) * wc[3] + T(0.25) * wc[4]);
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

// P^T for the 4-D layouts: coarse volume J0 (blockIdx.y) collects the fine volumes 2 J0 and, halved,
// 2 J0 +- 1 on a node-centred leading axis (volume J0 alone on a batch axis); one z-window of plane
// sums per contributing fine volume.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_adj_march_lead(const T* __restrict__ gfine,
                                                                  T* __restrict__ gcoarse, T* __restrict__ gscaled,
                                                                  MarchArgs a, T scale, AdamArgs<T> ad) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fnz = a.fn[0], fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int64_t cvol = (int64_t)cnz * cplane, fvol = (int64_t)fnz * fplane;
  const int J0 = blockIdx.y;
  const bool node = a.lead_loc == kNode;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
  if (jy >= cny || jx >= cnx) return;
  const int z0 = zc * a.usched.ZC;
  const int z1 = z0 + a.usched.ZC < cnz ? z0 + a.usched.ZC : cnz;
  const Adj6 ax = adj6(jx, cnx), ay = adj6(jy, cny);
  const bool xy_special = ax.special || ay.special;
  // contributing fine volumes: (index, weight); missing ones get weight 0 and a clamped index
  int fv[3];
  T wv[3];
  int ntap = 1;
  if (node) {
    ntap = 3;
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
  T wc[3][6], wr[3][6];
#pragma unroll
  for (int t = 0; t < 3; ++t) {
    if (t >= ntap) break;
    const T* gf = gfine + fv[t] * fvol;
    T c2[2], r2[2];
    const int fa[2] = {2 * z0 - 2, 2 * z0 - 1}, fb[2] = {2 * z0, 2 * z0 + 1};
    reduce_dispatch<T, 2>(gf, fa, fnz, fplane, fny, fnx, jy, jx, ay, ax, c2, r2);
    wc[t][0] = c2[0], wc[t][1] = c2[1], wr[t][0] = r2[0], wr[t][1] = r2[1];
    reduce_dispatch<T, 2>(gf, fb, fnz, fplane, fny, fnx, jy, jx, ay, ax, c2, r2);
    wc[t][2] = c2[0], wc[t][3] = c2[1], wr[t][2] = r2[0], wr[t][3] = r2[1];
  }
  for (int jz = z0; jz < z1; ++jz) {
    T v = T(0);
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      if (t >= ntap) break;
      T c2[2], r2[2];
      const int fn2[2] = {2 * jz + 2, 2 * jz + 3};
      reduce_dispatch<T, 2>(gfine + fv[t] * fvol, fn2, fnz, fplane, fny, fnx, jy, jx, ay, ax, c2, r2);
      wc[t][4] = c2[0], wc[t][5] = c2[1], wr[t][4] = r2[0], wr[t][5] = r2[1];
      v = v + wv[t] * combine_z<T>(wc[t], wr[t], jz, cnz, fnz, xy_special, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wc[t][i] = wc[t][i + 2];
        wr[t][i] = wr[t][i + 2];
      }
    }
    const int64_t ci = J0 * cvol + (int64_t)jz * cplane + (int64_t)jy * cnx + jx;
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
}

// P^T with a node-centred leading axis: coarse plane J collects fine plane 2J and half of the
// fine planes 2J-1 and 2J+1, each reduced over its (y, x) window with the two-cell-axis ghost rule.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_adj_march_n(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                               T* __restrict__ gscaled, MarchArgs a, T scale,
                                                               AdamArgs<T> ad) {
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int fnz = a.fn[0], fny = a.fn[1], fnx = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
  if (jy >= cny || jx >= cnx) return;
  const int z0 = zc * a.usched.ZC;
  const int z1 = z0 + a.usched.ZC < cnz ? z0 + a.usched.ZC : cnz;
  const Adj6 ax = adj6(jx, cnx), ay = adj6(jy, cny);
  const bool xy_special = ax.special || ay.special;
  T pc, pr;  // plane sums of the odd fine plane below
  {
    const int f1[1] = {2 * z0 - 1};
    T c1[1], r1[1];
    reduce_dispatch<T, 1>(gfine, f1, fnz, fplane, fny, fnx, jy, jx, ay, ax, c1, r1);
    pc = c1[0], pr = r1[0];
  }
  for (int jz = z0; jz < z1; ++jz) {
    T c2[2], r2[2];
    const int f2[2] = {2 * jz, 2 * jz + 1};
    reduce_dispatch<T, 2>(gfine, f2, fnz, fplane, fny, fnx, jy, jx, ay, ax, c2, r2);
    const T sc = c2[0] + T(0.5) * (pc + c2[1]);
    T v = sc;
    if (xy_special) {
      const T sr = r2[0] + T(0.5) * (pr + r2[1]);
      v = T(2) * sc - sr;
    }
    pc = c2[1], pr = r2[1];
    const int64_t ci = (int64_t)jz * cplane + (int64_t)jy * cnx + jx;
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
}

static bool march_setup(MarchArgs& m, const InterpArgs& a) {
  // exactly (1 | '.'), 'c' or 'n', 'c', 'c'
  if ((a.loc[1] != kCell && a.loc[1] != kNode) || a.loc[2] != kCell || a.loc[3] != kCell) return false;
  if (a.loc[1] == kNode && a.cut_axis >= 0) return false;
  const bool lead = a.fn[0] != 1;  // a real leading axis: '.ccc' batch or 'nccc'
  if (a.loc[0] == kCell || (lead && (a.loc[1] != kCell || a.cut_axis >= 0 || a.fn[0] > 65535))) return false;
  m.lead_loc = lead ? a.loc[0] : 0;
  m.lead_cn = (int)a.cn[0];
  m.lead_fn = (int)a.fn[0];
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
  int tx = 1;
  while (tx < m.cn[2] && tx < kBlock) tx *= 2;
  m.tx = tx;
  m.ty = kBlock / tx;
  const int64_t ytiles = (m.cn[1] + m.ty - 1) / m.ty, xtiles = (m.cn[2] + m.tx - 1) / m.tx;
  if ((int64_t)m.cn[0] * ytiles * xtiles >= ((int64_t)1 << 31)) return false;
  m.usched = make_unit_sched(m.cn[0], ytiles, xtiles);
  return true;
}

template <typename T>
int interp_add_march(const T* coarse, const T* add, T* fine, const InterpArgs& a, T cscale, T ascale,
                     hipStream_t stream) {
  MarchArgs m;
  if (!march_setup(m, a)) return 0;
  m.nt = (int64_t)m.lead_fn * m.fn[0] * m.fn[1] * m.fn[2] * (int64_t)sizeof(T) > kStreamBytes;
  if (m.lead_fn != 1)
    hipLaunchKernelGGL(k_interp_add_march_lead<T>, dim3(unit_grid(m.usched), m.lead_fn), dim3(kBlock), 0, stream,
                       coarse, add, fine, m, cscale, ascale);
  else if (a.loc[1] == kNode)
    hipLaunchKernelGGL(k_interp_add_march_n<T>, dim3(unit_grid(m.usched)), dim3(kBlock), 0, stream, coarse, add, fine,
                       m, cscale, ascale);
  else
    hipLaunchKernelGGL(k_interp_add_march<T>, dim3(unit_grid(m.usched)), dim3(kBlock), 0, stream, coarse, add, fine,
                       m, cscale, ascale);
  const int e = check_launch("k_interp_add_march");
  return e ? e : 1;
}

template <typename T>
int interp_adj_march(const T* gfine, T* gcoarse, T* gscaled, const InterpArgs& a, T scale, hipStream_t stream,
                     const AdamArgs<T>& ad) {
  MarchArgs m;
  if (!march_setup(m, a)) return 0;
  if (m.lead_fn != 1)
    hipLaunchKernelGGL(k_interp_adj_march_lead<T>, dim3(unit_grid(m.usched), m.lead_cn), dim3(kBlock), 0, stream,
                       gfine, gcoarse, gscaled, m, scale, ad);
  else if (a.loc[1] == kNode)
    hipLaunchKernelGGL(k_interp_adj_march_n<T>, dim3(unit_grid(m.usched)), dim3(kBlock), 0, stream, gfine, gcoarse,
                       gscaled, m, scale, ad);
  else
    hipLaunchKernelGGL(k_interp_adj_march<T>, dim3(unit_grid(m.usched)), dim3(kBlock), 0, stream, gfine, gcoarse,
                       gscaled, m, scale, ad);
  const int e = check_launch("k_interp_adj_march");
  return e ? e : 1;
}

template int interp_add_march<double>(const double*, const double*, double*, const InterpArgs&, double, double,
                                      hipStream_t);
template int interp_add_march<float>(const float*, const float*, float*, const InterpArgs&, float, float, hipStream_t);
template int interp_adj_march<double>(const double*, double*, double*, const InterpArgs&, double, hipStream_t,
                                      const AdamArgs<double>&);
template int interp_adj_march<float>(const float*, float*, float*, const InterpArgs&, float, hipStream_t,
                                     const AdamArgs<float>&);

}  // namespace odil
