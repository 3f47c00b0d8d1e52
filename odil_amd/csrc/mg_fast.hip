// Fast paths of the multigrid transfers for the layouts the workloads use: last axis 'c',
// second-to-last 'c' (or a '.' axis), arbitrary leading axes ('ccc', 'cc', 'c', 'ncc', ...).
// Same arithmetic and summation order as the generic kernels of mg_transfer.hip.
//
// P  : one thread per COARSE (jy, jx): it reads the 3x3 coarse neighbourhood once per
//      plane tap and produces the 2x2 fine outputs of a fine plane (16 B stores, lanes
//      contiguous along x).  Plane taps of the leading axes are uniform per workgroup.
// P^T: one thread per coarse (jy, jx): per fine row of the window it loads the 6 fine x
//      values [2jx-2, 2jx+3] as three 16 B packs, reduces them along x once and
//      accumulates rows/planes separably; the joint ghost rule costs a second weight set
//      only for cells within two of a boundary.
#include "mg_transfer.h"

namespace odil {

template <typename T>
struct alignas(2 * sizeof(T)) Pack2 {
  T a, b;
};

struct FastArgs {
  int cn[4], fn[4];
  int loc[4];
  int tx, ty;  // thread tile: tx lanes along coarse x, ty rows along coarse y
  int cut_axis, cut_lo, cut_hi;
  RowSched sched;
};

struct Tap3 {
  int cl[3], rf[3];
  bool out[3];
};

// Clamp / reflect indices of j-1, j, j+1 on a 'c' axis of n coarse cells.
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

template <typename T, bool YC>
__global__ __launch_bounds__(kBlock) void k_interp_add_fast(const T* __restrict__ coarse, const T* __restrict__ add,
                                                            T* __restrict__ fine, FastArgs a, T cscale, T ascale) {
  const int cnx = a.cn[3], cny = a.cn[2];
  const int fnx = a.fn[3], fny = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  RowIter it = sched_begin(a.sched);
  for (; it.t < it.count; it.t += it.step) {
    int p, yt, xt;
    sched_decode(a.sched, it, p, yt, xt);
    // plane taps (leading axes), uniform over the workgroup
    const int f0 = p / a.fn[1], f1 = p - f0 * a.fn[1];
    const Taps t0 = make_taps(a.loc[0], f0, a.cn[0]);
    const Taps t1 = make_taps(a.loc[1], f1, a.cn[1]);
    const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
    if (jy >= cny || jx >= cnx) continue;
    const Tap3 tx = tap3(jx, cnx);
    Tap3 ty;
    if (YC) {
      ty = tap3(jy, cny);
    } else {
      ty.cl[1] = ty.rf[1] = jy;
      ty.out[1] = false;
    }
    T acc[2][2] = {{T(0), T(0)}, {T(0), T(0)}};
#pragma unroll
    for (int r0 = 0; r0 < 2; ++r0)
#pragma unroll
      for (int r1 = 0; r1 < 2; ++r1) {
        if (r0 >= t0.cnt || r1 >= t1.cnt) continue;
        const int wl = t0.w[r0] * t1.w[r1];
        const T wl1 = T(wl), wl3 = T(3 * wl), wl9 = T(9 * wl);  // exact small integers
        const bool ol = t0.out[r0] || t1.out[r1];
        const T* ccl = coarse + (t0.cl[r0] * (int64_t)a.cn[1] + t1.cl[r1]) * cplane;
        const T* crf = coarse + (t0.rf[r0] * (int64_t)a.cn[1] + t1.rf[r1]) * cplane;
        T v[3][3];
#pragma unroll
        for (int dy = (YC ? 0 : 1); dy < (YC ? 3 : 2); ++dy)
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            T val = cscale * ccl[(int64_t)ty.cl[dy] * cnx + tx.cl[dx]];
            if (ol || ty.out[dy] || tx.out[dx])
              val = T(2) * val - cscale * crf[(int64_t)ty.rf[dy] * cnx + tx.rf[dx]];
            v[dy][dx] = val;
          }
          // reference order: (leading..., ry, rx) with rx fastest (core.py:675-687)
#pragma unroll
        for (int sy = 0; sy < (YC ? 2 : 1); ++sy)
#pragma unroll
          for (int sx = 0; sx < 2; ++sx) {
            T s = acc[sy][sx];
#pragma unroll
            for (int ry = 0; ry < (YC ? 2 : 1); ++ry)
#pragma unroll
              for (int rx = 0; rx < 2; ++rx) {
                // weights in r order on a 'c' axis: parity 0 -> (1, 3), parity 1 -> (3, 1)
                const int wy = YC ? (sy == ry ? 1 : 3) : 1;
                const int wx = sx == rx ? 1 : 3;
                const int dy = YC ? sy + ry : 1;
                const int dx = sx + rx;
                const int ww = wy * wx;  // compile-time after unrolling: 1, 3 or 9
                s = s + (ww == 1 ? wl1 : (ww == 3 ? wl3 : wl9)) * v[dy][dx];
              }
            acc[sy][sx] = s;
          }
      }
    // sum of weights is a power of two: multiplying by its reciprocal is exact (== the division)
    const T rdenom = T(1) / T(t0.sum * t1.sum * (YC ? 16 : 4));
    const int64_t fbase = (int64_t)p * fplane;
#pragma unroll
    for (int sy = 0; sy < (YC ? 2 : 1); ++sy) {
      const int fy = YC ? 2 * jy + sy : jy;
      const int64_t off = fbase + (int64_t)fy * fnx + 2 * jx;
      T o0 = acc[sy][0] * rdenom, o1 = acc[sy][1] * rdenom;
      if (add) {
        const Pack2<T> ad = *reinterpret_cast<const Pack2<T>*>(add + off);
        o0 = ascale * ad.a + o0;
        o1 = ascale * ad.b + o1;
      }
      Pack2<T> o;
      o.a = o0;
      o.b = o1;
      *reinterpret_cast<Pack2<T>*>(fine + off) = o;
    }
  }
}

// 1-D adjoint weights on a 'c' axis for coarse index J: window of 6 fine indices starting
// at 2J-2; wc / wr as in make_adj_taps (C / R sets of the joint ghost rule).
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

// Window and weights of a leading axis, computed on the fly (uniform per workgroup).
struct AdjWin {
  int k0, cnt;
  bool special;
};

__device__ inline AdjWin adj_window(int loc, int J, int n, bool cut_lo, bool cut_hi) {
  AdjWin w;
  w.special = false;
  if (loc == kCell) {
    w.special = ((J == 0 || J == 1) && !cut_lo) || ((J == n - 1 || J == n - 2) && !cut_hi);
    w.k0 = w.special ? 2 * J - 2 : 2 * J - 1;
    w.cnt = w.special ? 6 : 4;
  } else if (loc == kNode) {
    w.k0 = 2 * J - 1;
    w.cnt = 3;
  } else {
    w.k0 = J;
    w.cnt = 1;
  }
  return w;
}

__device__ inline void adj_weight(int loc, int J, int n, int F, int k, bool cut_lo, bool cut_hi, float& wc,
                                  float& wr) {
  if (loc == kCell) {
    const float w = w_cell(J, k, F), lo = w_cell(-1, k, F), hi = w_cell(n, k, F);
    wc = w + (J == 0 && !cut_lo ? lo : 0.f) + (J == n - 1 && !cut_hi ? hi : 0.f);
    wr = w + (J == 1 && !cut_lo ? lo : 0.f) + (J == n - 2 && !cut_hi ? hi : 0.f);
  } else if (loc == kNode) {
    const int d = k - 2 * J;
    wc = wr = (k >= 0 && k < F) ? (d == 0 ? 1.f : ((d == 1 || d == -1) ? 0.5f : 0.f)) : 0.f;
  } else {
    wc = wr = k == J ? 1.f : 0.f;
  }
}

template <typename T, bool YC>
__global__ __launch_bounds__(kBlock) void k_interp_adj_fast(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                            T* __restrict__ gscaled, FastArgs a, T scale,
                                                            AdamArgs<T> ad) {
  const int cnx = a.cn[3], cny = a.cn[2];
  const int fnx = a.fn[3], fny = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)fny * fnx;
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  RowIter it = sched_begin(a.sched);
  for (; it.t < it.count; it.t += it.step) {
    int p, yt, xt;
    sched_decode(a.sched, it, p, yt, xt);
    const int c0 = p / a.cn[1], c1 = p - c0 * a.cn[1];
    const bool l0 = a.cut_axis == 0 && a.cut_lo, h0 = a.cut_axis == 0 && a.cut_hi;
    const bool l1 = a.cut_axis == 1 && a.cut_lo, h1 = a.cut_axis == 1 && a.cut_hi;
    const AdjWin t0 = adj_window(a.loc[0], c0, a.cn[0], l0, h0);
    const AdjWin t1 = adj_window(a.loc[1], c1, a.cn[1], l1, h1);
    const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
    if (jy >= cny || jx >= cnx) continue;
    const Adj6 ax = adj6(jx, cnx);
    Adj6 ay;
    if (YC) {
      ay = adj6(jy, cny);
    } else {
      ay.special = false;
#pragma unroll
      for (int i = 0; i < 6; ++i) ay.wc[i] = ay.wr[i] = i == 2 ? 1.f : 0.f;
    }
    const bool special = t0.special || t1.special || ax.special || ay.special;
    T sc = T(0), sr = T(0);
    for (int i0 = 0; i0 < t0.cnt; ++i0) {
      float w0c, w0r;
      adj_weight(a.loc[0], c0, a.cn[0], a.fn[0], t0.k0 + i0, l0, h0, w0c, w0r);
      if (w0c == 0.f && w0r == 0.f) continue;
      for (int i1 = 0; i1 < t1.cnt; ++i1) {
        float w1c, w1r;
        adj_weight(a.loc[1], c1, a.cn[1], a.fn[1], t1.k0 + i1, l1, h1, w1c, w1r);
        if (w1c == 0.f && w1r == 0.f) continue;
        const T wcl = T(w0c * w1c), wrl = T(w0r * w1r);
        const T* gp = gfine + ((t0.k0 + i0) * (int64_t)a.fn[1] + (t1.k0 + i1)) * fplane;
#pragma unroll
        for (int iy = (YC ? 0 : 2); iy < (YC ? 6 : 3); ++iy) {
          if (ay.wc[iy] == 0.f && ay.wr[iy] == 0.f) continue;
          const int fy = YC ? 2 * jy - 2 + iy : jy;
          const T* row = gp + (int64_t)fy * fnx;
          // fine x window [2jx-2, 2jx+3] as three aligned pairs; out-of-range pairs have zero weight
          T g[6];
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            const int fx = 2 * (jx - 1 + q);
            if (fx >= 0 && fx < fnx) {
              const Pack2<T> pk = *reinterpret_cast<const Pack2<T>*>(row + fx);
              g[2 * q] = pk.a;
              g[2 * q + 1] = pk.b;
            } else {
              g[2 * q] = g[2 * q + 1] = T(0);
            }
          }
          T rc = T(0), rr = T(0);
          if (!ax.special) {
            // interior: {1,3,3,1}/4 on the middle four
            rc = (T(0.25) * g[1] + T(0.75) * g[2]) + (T(0.75) * g[3] + T(0.25) * g[4]);
            rr = rc;
          } else {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
              rc = rc + T(ax.wc[i]) * g[i];
              rr = rr + T(ax.wr[i]) * g[i];
            }
          }
          sc = sc + (wcl * T(ay.wc[iy])) * rc;
          if (special) sr = sr + (wrl * T(ay.wr[iy])) * rr;
        }
      }
    }
    const T v = special ? T(2) * sc - sr : sc;
    const int64_t ci = (int64_t)p * cplane + (int64_t)jy * cnx + jx;
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

static bool fast_setup(FastArgs& f, const InterpArgs& a, bool& yc) {
  if (a.loc[3] != kCell) return false;
  if (a.loc[2] == kNode) return false;
  yc = a.loc[2] == kCell;
  for (int i = 0; i < 4; ++i) {
    if (a.fn[i] >= (1 << 30)) return false;
    f.cn[i] = (int)a.cn[i];
    f.fn[i] = (int)a.fn[i];
    f.loc[i] = a.loc[i];
  }
  f.cut_axis = a.cut_axis;
  f.cut_lo = a.cut_lo;
  f.cut_hi = a.cut_hi;
  if (a.cut_axis >= 2) return false;  // cuts on the y / x axes go through the generic kernel
  // Thread tile: as many lanes along coarse x as useful (power of two, <= 256).
  int tx = 1;
  while (tx < f.cn[3] && tx < kBlock) tx *= 2;
  f.tx = tx;
  f.ty = kBlock / tx;
  const int64_t planes = a.fn[0] * a.fn[1];
  const int64_t ytiles = (f.cn[2] + f.ty - 1) / f.ty, xtiles = (f.cn[3] + f.tx - 1) / f.tx;
  if (!sched_ok(planes, ytiles, xtiles)) return false;
  f.sched = make_sched(planes, ytiles, xtiles);
  return true;
}

template <typename T>
int interp_add_fast(const T* coarse, const T* add, T* fine, const InterpArgs& a, T cscale, T ascale,
                    hipStream_t stream) {
  FastArgs f;
  bool yc;
  if (!fast_setup(f, a, yc)) return 0;
  const int grid = sched_grid(f.sched);
  if (yc)
    hipLaunchKernelGGL((k_interp_add_fast<T, true>), dim3(grid), dim3(kBlock), 0, stream, coarse, add, fine, f, cscale,
                       ascale);
  else
    hipLaunchKernelGGL((k_interp_add_fast<T, false>), dim3(grid), dim3(kBlock), 0, stream, coarse, add, fine, f,
                       cscale, ascale);
  const int e = check_launch("k_interp_add_fast");
  return e ? e : 1;
}

template <typename T>
int interp_adj_fast(const T* gfine, T* gcoarse, T* gscaled, const InterpArgs& a, T scale, hipStream_t stream,
                    const AdamArgs<T>& ad) {
  FastArgs f;
  bool yc;
  if (!fast_setup(f, a, yc)) return 0;
  // the schedule runs over COARSE planes here
  const int64_t planes = a.cn[0] * a.cn[1];
  const int64_t ytiles = (f.cn[2] + f.ty - 1) / f.ty, xtiles = (f.cn[3] + f.tx - 1) / f.tx;
  f.sched = make_sched(planes, ytiles, xtiles);
  const int grid = sched_grid(f.sched);
  if (yc)
    hipLaunchKernelGGL((k_interp_adj_fast<T, true>), dim3(grid), dim3(kBlock), 0, stream, gfine, gcoarse, gscaled, f,
                       scale, ad);
  else
    hipLaunchKernelGGL((k_interp_adj_fast<T, false>), dim3(grid), dim3(kBlock), 0, stream, gfine, gcoarse, gscaled, f,
                       scale, ad);
  const int e = check_launch("k_interp_adj_fast");
  return e ? e : 1;
}

template int interp_add_fast<double>(const double*, const double*, double*, const InterpArgs&, double, double,
                                     hipStream_t);
template int interp_add_fast<float>(const float*, const float*, float*, const InterpArgs&, float, float, hipStream_t);
template int interp_adj_fast<double>(const double*, double*, double*, const InterpArgs&, double, hipStream_t,
                                     const AdamArgs<double>&);
template int interp_adj_fast<float>(const float*, float*, float*, const InterpArgs&, float, hipStream_t,
                                    const AdamArgs<float>&);

}  // namespace odil
