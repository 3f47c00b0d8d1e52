// Poisson workload of the ODIL hot path on gfx950
// (reference examples/poisson/poisson.py:57-113; extrap_quadh core.py:1439-1445):
//   residual  fu = sum_i (u+ - 2u + u-)/h_i^2 - rhs, zero-Dirichlet ghosts, + mean(fu^2)
//   adjoint   gu = J^T (scale * fu)
//   jacobian  per-shift coefficient arrays (core.py:1313-1361 under distinct_shift)
//
// HBM-bound 7-point stencils.  Lanes run along x with 16 B per lane (2 x f64 / 4 x f32);
// the XCD-aware row schedule (common.h) keeps the y+-1 / z+-1 re-reads inside one XCD's
// L2, so HBM sees each of u, rhs once and fu once.  The loss is reduced in a fixed order
// (per-thread running sum -> wave shuffle -> LDS -> one partial per workgroup -> final
// kernel), so it is bit-reproducible run to run.
#include "poisson.h"

namespace odil {

struct StencilArgs {
  int64_t loss_z0, loss_z1;  // planes (canonical z) that enter the loss
  int64_t n[3];     // canonical (Z, Y, X) cell shape
  int active[3];    // axis takes part in the Laplacian
  UnitSched usched;
};

// z-marching 7-point kernels.  A workgroup owns one x-segment of one row and walks a chunk
// of planes; each lane keeps its own z-1 / z / z+1 values in registers, so every element of
// the marched array is loaded once by its owner (+ once per y-neighbour, an L2 hit inside
// the XCD).  Loads of the next plane are issued before the current plane is consumed.
template <typename T, bool FULL>
__global__ __launch_bounds__(kBlock) void k_poisson_residual(const T* __restrict__ u, const T* __restrict__ rhs,
                                                            T* __restrict__ fu, StencilArgs a, H2<T> h,
                                                            double* __restrict__ partials) {
  constexpr int V = VecOf<T>::N;
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  double local = 0.0;
  int zc, yi, xs;
  const bool have = unit_decode(a.usched, zc, yi, xs);
  const int64_t x0 = ((int64_t)xs * kBlock + threadIdx.x) * V;
  if (have && x0 < X) {
    const int64_t y = yi;
    const int64_t valid = X - x0 < V ? X - x0 : V;
    const int64_t z0 = (int64_t)zc * a.usched.ZC;
    const int64_t z1 = z0 + a.usched.ZC < Z ? z0 + a.usched.ZC : Z;
    const int64_t ym_off = (y == 0 ? Y - 1 : y - 1) * sy + x0, yp_off = (y == Y - 1 ? 0 : y + 1) * sy + x0;
    const int64_t c_off = y * sy + x0;
    const int64_t xl = y * sy + (x0 == 0 ? X - 1 : x0 - 1);
    const int64_t xr = y * sy + (x0 + valid >= X ? 0 : x0 + valid);
    T um[V], uc[V], up[V];
    if (a.active[0]) load_vec<T, V, FULL>(u + (z0 == 0 ? Z - 1 : z0 - 1) * sz + c_off, valid, um);
    load_vec<T, V, FULL>(u + z0 * sz + c_off, valid, uc);
    for (int64_t z = z0; z < z1; ++z) {
      const int64_t pz = z * sz;
      if (a.active[0]) load_vec<T, V, FULL>(u + (z == Z - 1 ? 0 : z + 1) * sz + c_off, valid, up);
      T r[V], ym[V], yp[V], out[V];
      load_vec<T, V, FULL, true>(rhs + pz + c_off, valid, r);
      if (a.active[1]) {
        load_vec<T, V, FULL>(u + pz + ym_off, valid, ym);
        load_vec<T, V, FULL>(u + pz + yp_off, valid, yp);
      }
      // x neighbours of the pack: periodic like mod.roll (core.py:963); the wrapped values
      // are discarded by the where() masks exactly as in the reference.
      T left, right;
      if (FULL) {
        // x neighbours of the pack from the adjacent lanes' registers; wave edges read memory
        left = from_prev_lane(uc[V - 1]);
        right = from_next_lane(uc[0]);
        if ((threadIdx.x & 63) == 0) left = u[pz + xl];
        if ((threadIdx.x & 63) == 63 || x0 + V >= X) right = u[pz + xr];
      } else {
        left = u[pz + xl];
        right = u[pz + xr];
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        if (i >= valid) break;
        const int64_t x = x0 + i;
        const T q = uc[i];
        T acc = T(0);
        bool first = true;
        if (a.active[0]) {
          acc = axis_term<T>(q, um[i], up[i], z == 0, z == Z - 1, h, 0);
          first = false;
        }
        if (a.active[1]) {
          const T t = axis_term<T>(q, ym[i], yp[i], y == 0, y == Y - 1, h, 1);
          acc = first ? t : acc + t;
          first = false;
        }
        {
          const T xm = i == 0 ? left : uc[i - 1];
          const T xp = (i == valid - 1) ? right : uc[i + 1 < V ? i + 1 : i];
          const T t = axis_term<T>(q, xm, xp, x == 0, x == X - 1, h, 2);
          acc = first ? t : acc + t;
        }
        const T f = acc - r[i];
        out[i] = f;
        if (z >= a.loss_z0 && z < a.loss_z1) local += (double)(f * f);
      }
      if (fu) store_vec<T, V, FULL, true>(fu + pz + c_off, valid, out);
#pragma unroll
      for (int i = 0; i < V; ++i) {
        um[i] = uc[i];
        uc[i] = up[i];
      }
    }
  }
  const double total = block_sum(local);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// One damped-Jacobi sweep of the same operator, x_out = x - omega (A x - b) / diag(A), in ONE pass
// (the smoother of the geometric multigrid that solves the Newton system, odil_amd/gmg.py): the walk
// of k_poisson_residual, with the diagonal -- sum over the active axes of (-2 - 2 [low wall] -
// 2 [high wall]) / h^2, see adj_axis -- formed from the indices instead of read from memory.
// Reads x and b, writes x_out (3 words per cell; residual + update as two kernels move 7).
// ZERO: the sweep starts from the zero vector (u is not read; the same arithmetic on zeros: the same bits).
template <typename T, bool FULL, bool ZERO = false>
__global__ __launch_bounds__(kBlock) void k_poisson_jacobi(const T* __restrict__ u, const T* __restrict__ rhs,
                                                          T* __restrict__ uout, StencilArgs a, H2<T> h, T omega) {
  constexpr int V = VecOf<T>::N;
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  int zc, yi, xs;
  if (!unit_decode(a.usched, zc, yi, xs)) return;
  const int64_t x0 = ((int64_t)xs * kBlock + threadIdx.x) * V;
  if (x0 >= X) return;
  const int64_t y = yi;
  const int64_t valid = X - x0 < V ? X - x0 : V;
  const int64_t z0 = (int64_t)zc * a.usched.ZC;
  const int64_t z1 = z0 + a.usched.ZC < Z ? z0 + a.usched.ZC : Z;
  const int64_t ym_off = (y == 0 ? Y - 1 : y - 1) * sy + x0, yp_off = (y == Y - 1 ? 0 : y + 1) * sy + x0;
  const int64_t c_off = y * sy + x0;
  const int64_t xl = y * sy + (x0 == 0 ? X - 1 : x0 - 1);
  const int64_t xr = y * sy + (x0 + valid >= X ? 0 : x0 + valid);
  // diagonal terms per axis: interior -2 / h^2, one more -2 / h^2 per wall touched
  T dterm[3];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) dterm[ax] = a.active[ax] ? div_h2<T>(T(-2), h, ax) : T(0);
  const T dy = a.active[1] ? dterm[1] * T(1 + (y == 0) + (y == Y - 1)) : T(0);
  // omega / diag takes four values along a unit's walk (z at a wall or not, x at a wall or not): four
  // divisions per lane instead of one ~35-instruction f64 divide per cell
  T wdiag[2][2];
#pragma unroll
  for (int zw = 0; zw < 2; ++zw)
#pragma unroll
    for (int xw = 0; xw < 2; ++xw) {
      const T dzv = a.active[0] ? dterm[0] * T(1 + zw) : T(0);
      wdiag[zw][xw] = omega / ((dzv + dy) + dterm[2] * T(1 + xw));
    }
  T um[V], uc[V], up[V];
  if constexpr (ZERO) {
#pragma unroll
    for (int i = 0; i < V; ++i) um[i] = uc[i] = up[i] = T(0);
  } else {
    if (a.active[0]) load_vec<T, V, FULL>(u + (z0 == 0 ? Z - 1 : z0 - 1) * sz + c_off, valid, um);
    load_vec<T, V, FULL>(u + z0 * sz + c_off, valid, uc);
  }
  for (int64_t z = z0; z < z1; ++z) {
    const int64_t pz = z * sz;
    if constexpr (!ZERO)
      if (a.active[0]) load_vec<T, V, FULL>(u + (z == Z - 1 ? 0 : z + 1) * sz + c_off, valid, up);
    T r[V], ym[V], yp[V], out[V];
    load_vec<T, V, FULL, true>(rhs + pz + c_off, valid, r);
    if constexpr (ZERO) {
#pragma unroll
      for (int i = 0; i < V; ++i) ym[i] = yp[i] = T(0);
    } else if (a.active[1]) {
      load_vec<T, V, FULL>(u + pz + ym_off, valid, ym);
      load_vec<T, V, FULL>(u + pz + yp_off, valid, yp);
    }
    T left, right;
    if constexpr (ZERO) {
      left = right = T(0);
    } else if (FULL) {
      const int lane = threadIdx.x & 63;
      left = from_prev_lane(uc[V - 1]);
      right = from_next_lane(uc[0]);
      if (lane == 0) left = u[pz + xl];
      if (lane == 63 || x0 + V >= X) right = u[pz + xr];
    } else {
      left = u[pz + xl];
      right = u[pz + xr];
    }
    // (a wall on both sides needs an extent of 1, which fill_args rejects)
    const bool zw = a.active[0] && (z == 0 || z == Z - 1);
    const T w_in = zw ? wdiag[1][0] : wdiag[0][0], w_wall = zw ? wdiag[1][1] : wdiag[0][1];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      if (i >= valid) break;
      const int64_t x = x0 + i;
      const T q = uc[i];
      T acc = T(0);
      if (a.active[0]) acc = axis_term<T>(q, um[i], up[i], z == 0, z == Z - 1, h, 0);
      if (a.active[1]) acc = acc + axis_term<T>(q, ym[i], yp[i], y == 0, y == Y - 1, h, 1);
      {
        const T xm = i == 0 ? left : uc[i - 1];
        const T xp = (i == valid - 1) ? right : uc[i + 1 < V ? i + 1 : i];
        acc = acc + axis_term<T>(q, xm, xp, x == 0, x == X - 1, h, 2);
      }
      out[i] = q - (acc - r[i]) * ((x == 0 || x == X - 1) ? w_wall : w_in);
    }
    store_vec<T, V, FULL, true>(uout + pz + c_off, valid, out);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      um[i] = uc[i];
      uc[i] = up[i];
    }
  }
}

// Residual of the same operator restricted to the next coarser grid in ONE pass, for the geometric
// multigrid of the Newton step (odil_amd/gmg.py): coarse[K, J, I] = scale * sum over the 2^3 fine cells
// of (A u - rhs), and mean((A u - rhs)^2) as the convergence measure.  The fine residual never reaches
// memory (separate residual + restriction kernels: 5 fine-grid words per cell, here 2 + 1/8).
// A workgroup owns an x-segment of a PAIR of rows (y = 2J, 2J + 1) and walks pairs of planes; row
// 2J + 1 is the y+1 neighbour of row 2J and vice versa, so the pair costs two y-neighbour loads, not
// four.  3-D, even extents, X a multiple of the pack width.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_poisson_residual_restrict(const T* __restrict__ u,
                                                                     const T* __restrict__ rhs,
                                                                     T* __restrict__ coarse, StencilArgs a, H2<T> h,
                                                                     T scale, double* __restrict__ partials) {
  constexpr int V = VecOf<T>::N;
  constexpr int C = V / 2;  // coarse cells per lane
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  const int64_t cX = X / 2, csz = (Y / 2) * cX;
  double local = 0.0;
  int zc, jj, xs;
  const bool have = unit_decode(a.usched, zc, jj, xs);  // units: (chunk of plane pairs, row pair, x-segment)
  const int64_t x0 = ((int64_t)xs * kBlock + threadIdx.x) * V;
  if (have && x0 < X) {
    const int64_t ya = 2 * (int64_t)jj, yb = ya + 1;
    const int64_t z0 = 2 * (int64_t)zc * a.usched.ZC;
    const int64_t z1 = z0 + 2 * a.usched.ZC < Z ? z0 + 2 * a.usched.ZC : Z;
    const int64_t a_off = ya * sy + x0, b_off = yb * sy + x0;
    const int64_t ym_off = (ya == 0 ? Y - 1 : ya - 1) * sy + x0, yp_off = (yb == Y - 1 ? 0 : yb + 1) * sy + x0;
    const int64_t xl = x0 == 0 ? X - 1 : x0 - 1, xr = x0 + V >= X ? 0 : x0 + V;
    T am[V], ac[V], ap[V], bm[V], bc[V], bp[V], acc[C];
    load_vec<T, V, true>(u + (z0 == 0 ? Z - 1 : z0 - 1) * sz + a_off, V, am);
    load_vec<T, V, true>(u + (z0 == 0 ? Z - 1 : z0 - 1) * sz + b_off, V, bm);
    load_vec<T, V, true>(u + z0 * sz + a_off, V, ac);
    load_vec<T, V, true>(u + z0 * sz + b_off, V, bc);
    for (int64_t z = z0; z < z1; ++z) {
      const int64_t pz = z * sz, pn = (z == Z - 1 ? 0 : z + 1) * sz;
      load_vec<T, V, true>(u + pn + a_off, V, ap);
      load_vec<T, V, true>(u + pn + b_off, V, bp);
      T ra[V], rb[V], ym[V], yp[V];
      load_vec<T, V, true, true>(rhs + pz + a_off, V, ra);
      load_vec<T, V, true, true>(rhs + pz + b_off, V, rb);
      load_vec<T, V, true>(u + pz + ym_off, V, ym);
      load_vec<T, V, true>(u + pz + yp_off, V, yp);
      // x neighbours of the packs: adjacent lanes' registers, memory at the wave edges
      T wa = from_prev_lane(ac[V - 1]), ea = from_next_lane(ac[0]);
      T wb = from_prev_lane(bc[V - 1]), eb = from_next_lane(bc[0]);
      if ((threadIdx.x & 63) == 0) {
        wa = u[pz + ya * sy + xl];
        wb = u[pz + yb * sy + xl];
      }
      if ((threadIdx.x & 63) == 63 || x0 + V >= X) {
        ea = u[pz + ya * sy + xr];
        eb = u[pz + yb * sy + xr];
      }
      const bool zlo = z == 0, zhi = z == Z - 1;
      const bool counted = z >= a.loss_z0 && z < a.loss_z1;  // (slab form: the norm of the rank's own planes)
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int64_t x = x0 + i;
        const bool xlo = x == 0, xhi = x == X - 1;
        T fa = axis_term<T>(ac[i], am[i], ap[i], zlo, zhi, h, 0);
        fa = fa + axis_term<T>(ac[i], ym[i], bc[i], ya == 0, false, h, 1);
        fa = fa + axis_term<T>(ac[i], i == 0 ? wa : ac[i - 1], i == V - 1 ? ea : ac[i + 1 < V ? i + 1 : i], xlo, xhi,
                               h, 2);
        fa = fa - ra[i];
        T fb = axis_term<T>(bc[i], bm[i], bp[i], zlo, zhi, h, 0);
        fb = fb + axis_term<T>(bc[i], ac[i], yp[i], false, yb == Y - 1, h, 1);
        fb = fb + axis_term<T>(bc[i], i == 0 ? wb : bc[i - 1], i == V - 1 ? eb : bc[i + 1 < V ? i + 1 : i], xlo, xhi,
                               h, 2);
        fb = fb - rb[i];
        if (counted) local += (double)(fa * fa) + (double)(fb * fb);
        const T pair = fa + fb;
        if ((i & 1) == 0)
          acc[i / 2] = ((z & 1) ? acc[i / 2] : T(0)) + pair;
        else
          acc[i / 2] = acc[i / 2] + pair;
      }
      if (z & 1) {
        T out[C];
#pragma unroll
        for (int c = 0; c < C; ++c) out[c] = scale * acc[c];
        store_vec<T, C, true>(coarse + (z / 2) * csz + (int64_t)jj * cX + x0 / 2, C, out);
      }
#pragma unroll
      for (int i = 0; i < V; ++i) {
        am[i] = ac[i];
        ac[i] = ap[i];
        bm[i] = bc[i];
        bc[i] = bp[i];
      }
    }
  }
  const double total = block_sum(local);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

template <typename T, bool FULL>
__global__ __launch_bounds__(kBlock) void k_poisson_adjoint(const T* __restrict__ fu, T* __restrict__ gu,
                                                           StencilArgs a, H2<T> h, T scale, AdamArgs<T> ad) {
  constexpr int V = VecOf<T>::N;
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  int zc, yi, xs;
  if (!unit_decode(a.usched, zc, yi, xs)) return;
  const int64_t x0 = ((int64_t)xs * kBlock + threadIdx.x) * V;
  if (x0 >= X) return;
  const int64_t y = yi;
  const int64_t valid = X - x0 < V ? X - x0 : V;
  const int64_t z0 = (int64_t)zc * a.usched.ZC;
  const int64_t z1 = z0 + a.usched.ZC < Z ? z0 + a.usched.ZC : Z;
  const int64_t ym_off = (y == 0 ? Y - 1 : y - 1) * sy + x0, yp_off = (y == Y - 1 ? 0 : y + 1) * sy + x0;
  const int64_t c_off = y * sy + x0;
  const int64_t xl = y * sy + (x0 == 0 ? X - 1 : x0 - 1);
  const int64_t xr = y * sy + (x0 + valid >= X ? 0 : x0 + valid);
  T fm[V], fc[V], fp[V];
  if (a.active[0]) load_vec<T, V, FULL>(fu + (z0 == 0 ? Z - 1 : z0 - 1) * sz + c_off, valid, fm);
  load_vec<T, V, FULL>(fu + z0 * sz + c_off, valid, fc);
  for (int64_t z = z0; z < z1; ++z) {
    const int64_t pz = z * sz;
    if (a.active[0]) load_vec<T, V, FULL>(fu + (z == Z - 1 ? 0 : z + 1) * sz + c_off, valid, fp);
    T ym[V], yp[V], out[V];
    if (a.active[1]) {
      load_vec<T, V, FULL>(fu + pz + ym_off, valid, ym);
      load_vec<T, V, FULL>(fu + pz + yp_off, valid, yp);
    }
    T left, right;
    if (FULL) {
      // x neighbours of the pack from the adjacent lanes' registers; wave edges read memory
      left = from_prev_lane(fc[V - 1]);
      right = from_next_lane(fc[0]);
      if ((threadIdx.x & 63) == 0) left = fu[pz + xl];
      if ((threadIdx.x & 63) == 63 || x0 + V >= X) right = fu[pz + xr];
    } else {
      left = fu[pz + xl];
      right = fu[pz + xr];
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      if (i >= valid) break;
      const int64_t x = x0 + i;
      const T fb = scale * fc[i];
      T g = T(0);
      if (a.active[0]) g = g + adj_axis<T>(fb, scale * fm[i], scale * fp[i], z, Z, h, 0);
      if (a.active[1]) g = g + adj_axis<T>(fb, scale * ym[i], scale * yp[i], y, Y, h, 1);
      const T xm = scale * (i == 0 ? left : fc[i - 1]);
      const T xp = scale * ((i == valid - 1) ? right : fc[i + 1 < V ? i + 1 : i]);
      g = g + adj_axis<T>(fb, xm, xp, x, X, h, 2);
      out[i] = g;
    }
    if (gu) store_vec<T, V, FULL, true>(gu + pz + c_off, valid, out);
    if (ad.x) {
      // x, m, v are touched exactly once per launch: streaming (non-temporal) accesses keep them
      // from evicting the fu rows / planes the stencil re-reads from L2
      T xv[V], mv[V], vv[V];
      load_vec<T, V, FULL, true>(ad.x + pz + c_off, valid, xv);
      load_vec<T, V, FULL, true>(ad.m + pz + c_off, valid, mv);
      load_vec<T, V, FULL, true>(ad.v + pz + c_off, valid, vv);
#pragma unroll
      for (int i = 0; i < V; ++i) adam_update<T>(xv[i], mv[i], vv[i], out[i], ad);
      store_vec<T, V, FULL, true>(ad.x + pz + c_off, valid, xv);
      store_vec<T, V, FULL, true>(ad.m + pz + c_off, valid, mv);
      store_vec<T, V, FULL, true>(ad.v + pz + c_off, valid, vv);
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      fm[i] = fc[i];
      fc[i] = fp[i];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_poisson_jac(T* __restrict__ coeffs, StencilArgs a, int ndim, T h2z, T h2y,
                                                       T h2x) {
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t size = Z * Y * X;
  const T h2[3] = {h2z, h2y, h2x};
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += (int64_t)gridDim.x * kBlock) {
    const int64_t x = i % X, y = (i / X) % Y, z = i / (X * Y);
    const int64_t idx[3] = {z, y, x};
    T c0 = T(0);
    int slot = 1;
    for (int ax = 0; ax < 3; ++ax) {
      if (!a.active[ax]) continue;
      const bool lo = idx[ax] == 0, hi = idx[ax] == a.n[ax] - 1;
      const T one = T(1);
      const T cm = (lo ? T(0) : one) + (hi ? one / T(3) : T(0));
      const T cp = (hi ? T(0) : one) + (lo ? one / T(3) : T(0));
      const T cc = T(-2) * one + (lo ? T(-2) * one : T(0)) + (hi ? T(-2) * one : T(0));
      coeffs[(int64_t)slot * size + i] = cm / h2[ax];
      coeffs[(int64_t)(slot + 1) * size + i] = cp / h2[ax];
      c0 = c0 + cc / h2[ax];
      slot += 2;
    }
    coeffs[i] = c0;
  }
}

// Is a set of coefficient arrays the Jacobian of the Poisson operator?  One pass over the arrays against the values
// k_poisson_jac would WRITE (the same expressions), no reference arrays: per array k the two maxima max |a_k - e_k| and
// max |e_k| (the recognition of a Newton system, odil_amd/gmg.py: recognise_poisson -- it used to generate the 2 d + 1
// reference arrays and compare them pair by pair: 3.6 ms at 512^3 for what is 1.4 ms of reading).
template <typename T>
struct JacPtrs {
  const T* p[7];
};

template <typename T>
__global__ __launch_bounds__(kBlock) void k_poisson_jac_match(JacPtrs<T> arr, StencilArgs a, T h2z, T h2y, T h2x,
                                                             double* __restrict__ partials) {
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t size = Z * Y * X;
  const T h2[3] = {h2z, h2y, h2x};
  double md[7], mb[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) md[k] = mb[k] = 0.0;
  auto upd = [&](int k, T got, T want) {
    const double d = fabs((double)got - (double)want), w = fabs((double)want);
    md[k] = d > md[k] || d != d ? d : md[k];  // (a NaN difference must not pass for a match)
    mb[k] = w > mb[k] ? w : mb[k];
  };
  const int64_t per = (size + gridDim.x - 1) / gridDim.x;
  const int64_t lo_i = (int64_t)blockIdx.x * per, hi_i = lo_i + per < size ? lo_i + per : size;
  for (int64_t i = lo_i + threadIdx.x; i < hi_i; i += kBlock) {
    const int64_t x = i % X, y = (i / X) % Y, z = i / (X * Y);
    const int64_t idx[3] = {z, y, x};
    T c0 = T(0);
    int slot = 1;
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      if (!a.active[ax]) continue;
      const bool lo = idx[ax] == 0, hi = idx[ax] == a.n[ax] - 1;
      const T one = T(1);
      const T cm = (lo ? T(0) : one) + (hi ? one / T(3) : T(0));
      const T cp = (hi ? T(0) : one) + (lo ? one / T(3) : T(0));
      const T cc = T(-2) * one + (lo ? T(-2) * one : T(0)) + (hi ? T(-2) * one : T(0));
      upd(slot, arr.p[slot][i], cm / h2[ax]);
      upd(slot + 1, arr.p[slot + 1][i], cp / h2[ax]);
      c0 = c0 + cc / h2[ax];
      slot += 2;
    }
    upd(0, arr.p[0][i], c0);
  }
  __shared__ double sm[kBlock / 64][14];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 7; ++k) {
    for (int off = 32; off > 0; off >>= 1) {
      const double od = __shfl_xor(md[k], off, 64), ob = __shfl_xor(mb[k], off, 64);
      md[k] = od > md[k] || od != od ? od : md[k];
      mb[k] = ob > mb[k] ? ob : mb[k];
    }
    if (lane == 0) sm[wave][2 * k] = md[k], sm[wave][2 * k + 1] = mb[k];
  }
  __syncthreads();
  if (threadIdx.x < 14) {
    double m = sm[0][threadIdx.x];
    for (int w = 1; w < kBlock / 64; ++w) {
      const double o = sm[w][threadIdx.x];
      m = o > m || o != o ? o : m;
    }
    partials[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] = m;  // row q of the partial maxima: 14 rows of gridDim.x
  }
}

template <typename T>
__global__ __launch_bounds__(64) void k_rows_max_final(const double* __restrict__ partials, int count, T* __restrict__ out) {
  const double* p = partials + (int64_t)blockIdx.x * count;
  double m = 0.0;
  for (int i = threadIdx.x; i < count; i += 64) {
    const double v = p[i];
    m = v > m || v != v ? v : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_xor(m, off, 64);
    m = o > m || o != o ? o : m;
  }
  if (threadIdx.x == 0) out[blockIdx.x] = (T)m;
}

template <typename T>
static int fill_args(StencilArgs& a, const int64_t* shape, int ndim, const T* h2, T h[3]) {
  if (ndim < 1 || ndim > 3 || !shape || !h2) {
    set_error("poisson: ndim=%d out of range [1,3] or null shape/h2", ndim);
    return ODIL_E_INVAL;
  }
  for (int i = 0; i < 3; ++i) {
    a.n[i] = 1;
    a.active[i] = 0;
    h[i] = T(1);
  }
  // Canonical (Z, Y, X): the slowest real axis is the marched one.  ndim 3 -> (n0, n1, n2),
  // ndim 2 -> (n0, 1, n1), ndim 1 -> (1, 1, n0).  Axis terms are still summed in axis order.
  static const int map3[3][3] = {{2, 0, 0}, {0, 2, 0}, {0, 1, 2}};
  for (int i = 0; i < ndim; ++i) {
    const int c = map3[ndim - 1][i];
    a.n[c] = shape[i];
    a.active[c] = 1;
    h[c] = h2[i];
    if (shape[i] < 2) {
      set_error("poisson: extent %lld on axis %d must be >= 2", (long long)shape[i], i);
      return ODIL_E_INVAL;
    }
  }
  const int per = kBlock * VecOf<T>::N;
  const int64_t XS = (a.n[2] + per - 1) / per;
  if (a.n[0] * a.n[1] * XS >= ((int64_t)1 << 31)) {
    set_error("poisson: grid too large for one launch");
    return ODIL_E_INVAL;
  }
  a.usched = make_unit_sched(a.n[0], a.n[1], XS);
  if (unit_grid(a.usched) > kMaxPartials) {
    set_error("poisson: %d workgroups exceed the reduction workspace", unit_grid(a.usched));
    return ODIL_E_INVAL;
  }
  return 0;
}

template <typename T>
static int poisson_residual(const T* u, const T* rhs, T* fu, const int64_t* shape, int ndim, const T* h2,
                            double* partials, T* loss, void* stream, int64_t z0 = 0, int64_t z1 = -1,
                            double denom = 0.0) {
  StencilArgs a;
  T h[3];
  if (int e = fill_args<T>(a, shape, ndim, h2, h)) return e;
  a.loss_z0 = z0;
  a.loss_z1 = z1 < 0 ? a.n[0] : z1;
  if (ndim < 3 && (z0 != 0 || z1 >= 0) && ndim != 2) {
    set_error("poisson_residual_slab: a plane range needs ndim >= 2");
    return ODIL_E_INVAL;
  }
  if (!u || !rhs || !partials || !loss) {
    set_error("poisson_residual: null pointer");
    return ODIL_E_INVAL;
  }
  const int grid = unit_grid(a.usched);
  if (a.n[2] % VecOf<T>::N == 0)
    hipLaunchKernelGGL((k_poisson_residual<T, true>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, u, rhs, fu, a,
                       make_h2<T>(h), partials);
  else
    hipLaunchKernelGGL((k_poisson_residual<T, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, u, rhs, fu,
                       a, make_h2<T>(h), partials);
  if (int e = check_launch("k_poisson_residual")) return e;
  const double size = denom > 0.0 ? denom : (double)(a.n[0] * a.n[1] * a.n[2]);
  return launch_final_reduce<T>(partials, grid, 0, 1, size, loss, (hipStream_t)stream);
}

template <typename T>
static int poisson_adjoint(const T* fu, T* gu, const int64_t* shape, int ndim, const T* h2, T scale, void* stream,
                           AdamArgs<T> ad = AdamArgs<T>{nullptr, nullptr, nullptr, T(0), T(0), T(0), T(0), nullptr}) {
  StencilArgs a;
  T h[3];
  if (int e = fill_args<T>(a, shape, ndim, h2, h)) return e;
  if (!fu || (!gu && !ad.x)) {  // with the Adam update fused in, the gradient itself need not be stored
    set_error("poisson_adjoint: null pointer");
    return ODIL_E_INVAL;
  }
  if (a.n[2] % VecOf<T>::N == 0)
    hipLaunchKernelGGL((k_poisson_adjoint<T, true>), dim3(unit_grid(a.usched)), dim3(kBlock), 0, (hipStream_t)stream,
                       fu, gu, a, make_h2<T>(h), scale, ad);
  else
    hipLaunchKernelGGL((k_poisson_adjoint<T, false>), dim3(unit_grid(a.usched)), dim3(kBlock), 0, (hipStream_t)stream,
                       fu, gu, a, make_h2<T>(h), scale, ad);
  return check_launch("k_poisson_adjoint");
}

template <typename T>
static int poisson_jacobi(const T* u, const T* rhs, T* uout, const int64_t* shape, int ndim, const T* h2, T omega,
                          void* stream) {
  StencilArgs a;
  T h[3];
  if (int e = fill_args<T>(a, shape, ndim, h2, h)) return e;
  if (!rhs || !uout || u == uout) {  // (u == NULL: the sweep starts from the zero vector)
    set_error("poisson_jacobi: null pointer, or the sweep in place (x_out must differ from x)");
    return ODIL_E_INVAL;
  }
  a.loss_z0 = 0;
  a.loss_z1 = 0;
  if (!u && a.n[2] % VecOf<T>::N == 0)
    hipLaunchKernelGGL((k_poisson_jacobi<T, true, true>), dim3(unit_grid(a.usched)), dim3(kBlock), 0, (hipStream_t)stream,
                       u, rhs, uout, a, make_h2<T>(h), omega);
  else if (!u)
    hipLaunchKernelGGL((k_poisson_jacobi<T, false, true>), dim3(unit_grid(a.usched)), dim3(kBlock), 0,
                       (hipStream_t)stream, u, rhs, uout, a, make_h2<T>(h), omega);
  else if (a.n[2] % VecOf<T>::N == 0)
    hipLaunchKernelGGL((k_poisson_jacobi<T, true>), dim3(unit_grid(a.usched)), dim3(kBlock), 0, (hipStream_t)stream, u,
                       rhs, uout, a, make_h2<T>(h), omega);
  else
    hipLaunchKernelGGL((k_poisson_jacobi<T, false>), dim3(unit_grid(a.usched)), dim3(kBlock), 0, (hipStream_t)stream,
                       u, rhs, uout, a, make_h2<T>(h), omega);
  return check_launch("k_poisson_jacobi");
}

template <typename T>
static int poisson_residual_restrict(const T* u, const T* rhs, T* coarse, const int64_t* shape, int ndim, const T* h2,
                                     T scale, double* partials, T* loss, void* stream, int64_t z0 = 0, int64_t z1 = -1,
                                     double denom = 0.0) {
  StencilArgs a;
  T h[3];
  if (int e = fill_args<T>(a, shape, ndim, h2, h)) return e;
  if (!u || !rhs || !coarse || !partials) {  // (loss == NULL: the norm is nobody's measure -- no reduction launch)
    set_error("poisson_residual_restrict: null pointer");
    return ODIL_E_INVAL;
  }
  if (ndim != 3 || a.n[0] % 2 || a.n[1] % 2 || a.n[2] % VecOf<T>::N) {
    set_error("poisson_residual_restrict: needs ndim = 3, even extents and n[2] %% %d == 0", VecOf<T>::N);
    return ODIL_E_INVAL;
  }
  a.loss_z0 = z1 < 0 ? 0 : z0;
  a.loss_z1 = z1 < 0 ? a.n[0] : z1;
  if (denom <= 0.0) denom = (double)(a.n[0] * a.n[1] * a.n[2]);
  const int per = kBlock * VecOf<T>::N;
  a.usched = make_unit_sched(a.n[0] / 2, a.n[1] / 2, (a.n[2] + per - 1) / per);
  const int grid = unit_grid(a.usched);
  hipLaunchKernelGGL((k_poisson_residual_restrict<T>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, u, rhs, coarse,
                     a, make_h2<T>(h), scale, partials);
  if (int e = check_launch("k_poisson_residual_restrict")) return e;
  if (!loss) return 0;
  return launch_final_reduce<T>(partials, grid, 0, 1, denom, loss, (hipStream_t)stream);
}

template <typename T>
static int poisson_jac(T* coeffs, const int64_t* shape, int ndim, const T* h2, void* stream) {
  StencilArgs a;
  T h[3];
  if (int e = fill_args<T>(a, shape, ndim, h2, h)) return e;
  if (!coeffs) {
    set_error("poisson_jac_coeffs: null pointer");
    return ODIL_E_INVAL;
  }
  const int64_t size = a.n[0] * a.n[1] * a.n[2];
  hipLaunchKernelGGL(k_poisson_jac<T>, dim3(grid_flat(size, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, coeffs, a,
                     ndim, h[0], h[1], h[2]);
  return check_launch("k_poisson_jac");
}

template <typename T>
static int poisson_jac_match(const T* const* arrays, const int64_t* shape, int ndim, const T* h2, double* partials, T* out,
                             void* stream) {
  StencilArgs a;
  T h[3];
  if (int e = fill_args<T>(a, shape, ndim, h2, h)) return e;
  if (!arrays || !partials || !out) {
    set_error("poisson_jac_match: null pointer");
    return ODIL_E_INVAL;
  }
  JacPtrs<T> ptrs;
  for (int k = 0; k < 7; ++k) ptrs.p[k] = nullptr;
  for (int k = 0; k < 2 * ndim + 1; ++k) {
    if (!arrays[k]) {
      set_error("poisson_jac_match: coefficient array %d is null", k);
      return ODIL_E_INVAL;
    }
    ptrs.p[k] = arrays[k];
  }
  const int64_t size = a.n[0] * a.n[1] * a.n[2];
  int grid = grid_for(size, kBlock * 8);
  if (grid > kMaxPartials / 14) grid = kMaxPartials / 14;
  hipLaunchKernelGGL(k_poisson_jac_match<T>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, ptrs, a, h[0], h[1], h[2],
                     partials);
  if (int e = check_launch("k_poisson_jac_match")) return e;
  hipLaunchKernelGGL(k_rows_max_final<T>, dim3(2 * (2 * ndim + 1)), dim3(64), 0, (hipStream_t)stream, partials, grid, out);
  return check_launch("k_rows_max_final");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_poisson_jac_match_f64(const double* const* arrays, const int64_t* shape, int ndim, const double* h2,
                               double* partials, double* out, void* stream) {
  return poisson_jac_match<double>(arrays, shape, ndim, h2, partials, out, stream);
}
int odil_poisson_jac_match_f32(const float* const* arrays, const int64_t* shape, int ndim, const float* h2, double* partials,
                               float* out, void* stream) {
  return poisson_jac_match<float>(arrays, shape, ndim, h2, partials, out, stream);
}
int odil_poisson_residual_f64(const double* u, const double* rhs, double* fu, const int64_t* shape, int ndim,
                              const double* h2, double* partials, double* loss, void* stream) {
  return poisson_residual<double>(u, rhs, fu, shape, ndim, h2, partials, loss, stream);
}
int odil_poisson_residual_f32(const float* u, const float* rhs, float* fu, const int64_t* shape, int ndim,
                              const float* h2, double* partials, float* loss, void* stream) {
  return poisson_residual<float>(u, rhs, fu, shape, ndim, h2, partials, loss, stream);
}
int odil_poisson_residual_slab_f64(const double* u, const double* rhs, double* fu, const int64_t* shape, int ndim,
                                   const double* h2, int64_t z0, int64_t z1, double denom, double* partials,
                                   double* loss, void* stream) {
  return poisson_residual<double>(u, rhs, fu, shape, ndim, h2, partials, loss, stream, z0, z1, denom);
}
int odil_poisson_residual_slab_f32(const float* u, const float* rhs, float* fu, const int64_t* shape, int ndim,
                                   const float* h2, int64_t z0, int64_t z1, double denom, double* partials,
                                   float* loss, void* stream) {
  return poisson_residual<float>(u, rhs, fu, shape, ndim, h2, partials, loss, stream, z0, z1, denom);
}
int odil_poisson_residual_restrict_f64(const double* u, const double* rhs, double* coarse, const int64_t* shape,
                                       int ndim, const double* h2, double scale, double* partials, double* loss,
                                       void* stream) {
  return poisson_residual_restrict<double>(u, rhs, coarse, shape, ndim, h2, scale, partials, loss, stream);
}
int odil_poisson_residual_restrict_f32(const float* u, const float* rhs, float* coarse, const int64_t* shape, int ndim,
                                       const float* h2, float scale, double* partials, float* loss, void* stream) {
  return poisson_residual_restrict<float>(u, rhs, coarse, shape, ndim, h2, scale, partials, loss, stream);
}
int odil_poisson_residual_restrict_slab_f64(const double* u, const double* rhs, double* coarse, const int64_t* shape,
                                            int ndim, const double* h2, double scale, int64_t z0, int64_t z1,
                                            double denom, double* partials, double* loss, void* stream) {
  return poisson_residual_restrict<double>(u, rhs, coarse, shape, ndim, h2, scale, partials, loss, stream, z0, z1, denom);
}
int odil_poisson_residual_restrict_slab_f32(const float* u, const float* rhs, float* coarse, const int64_t* shape,
                                            int ndim, const float* h2, float scale, int64_t z0, int64_t z1, double denom,
                                            double* partials, float* loss, void* stream) {
  return poisson_residual_restrict<float>(u, rhs, coarse, shape, ndim, h2, scale, partials, loss, stream, z0, z1, denom);
}
int odil_poisson_adjoint_f64(const double* fu, double* gu, const int64_t* shape, int ndim, const double* h2,
                             double scale, void* stream) {
  return poisson_adjoint<double>(fu, gu, shape, ndim, h2, scale, stream);
}
int odil_poisson_adjoint_f32(const float* fu, float* gu, const int64_t* shape, int ndim, const float* h2,
                             float scale, void* stream) {
  return poisson_adjoint<float>(fu, gu, shape, ndim, h2, scale, stream);
}
int odil_poisson_adjoint_adam_f64(const double* fu, double* gu, double* x, double* m, double* v, const int64_t* shape,
                                  int ndim, const double* h2, double scale, double alpha, double one_minus_b1,
                                  double one_minus_b2, double eps, const double* alpha_dev, void* stream) {
  if (!x || !m || !v) {
    set_error("poisson_adjoint_adam: null pointer");
    return ODIL_E_INVAL;
  }
  return poisson_adjoint<double>(fu, gu, shape, ndim, h2, scale, stream,
                                 AdamArgs<double>{x, m, v, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev});
}
int odil_poisson_adjoint_adam_f32(const float* fu, float* gu, float* x, float* m, float* v, const int64_t* shape,
                                  int ndim, const float* h2, float scale, float alpha, float one_minus_b1,
                                  float one_minus_b2, float eps, const float* alpha_dev, void* stream) {
  if (!x || !m || !v) {
    set_error("poisson_adjoint_adam: null pointer");
    return ODIL_E_INVAL;
  }
  return poisson_adjoint<float>(fu, gu, shape, ndim, h2, scale, stream,
                                AdamArgs<float>{x, m, v, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev});
}
int odil_poisson_jacobi_f64(const double* u, const double* rhs, double* uout, const int64_t* shape, int ndim,
                            const double* h2, double omega, void* stream) {
  return poisson_jacobi<double>(u, rhs, uout, shape, ndim, h2, omega, stream);
}
int odil_poisson_jacobi_f32(const float* u, const float* rhs, float* uout, const int64_t* shape, int ndim,
                            const float* h2, float omega, void* stream) {
  return poisson_jacobi<float>(u, rhs, uout, shape, ndim, h2, omega, stream);
}
int odil_poisson_jac_coeffs_f64(double* coeffs, const int64_t* shape, int ndim, const double* h2, void* stream) {
  return poisson_jac<double>(coeffs, shape, ndim, h2, stream);
}
int odil_poisson_jac_coeffs_f32(float* coeffs, const int64_t* shape, int ndim, const float* h2, void* stream) {
  return poisson_jac<float>(coeffs, shape, ndim, h2, stream);
}
}  // extern "C"
