// Poisson residual with the LAST prolongation of the multigrid synthesis fused in:
//   u = w_0 + P s_1   (reference core.py:245-263, last step)   is never written to memory,
//   fu = Lap(u) - rhs (reference examples/poisson/poisson.py:57-113) is evaluated from it directly.
//
// The walk is the one of k_interp_add_march: a thread owns a coarse column (jy, jx), keeps the
// 3 x 3 x 3 ghosted coarse neighbourhood in registers and steps through the coarse planes.  That
// neighbourhood determines u on the 4 x 4 x 4 fine patch around the thread's 2 x 2 x 2 fine cells,
// which is everything the 7-point stencil of those cells reads: own values for the planes below and
// above slide through registers, edge neighbours are recomputed (P costs 8 multiply-adds per
// value; the kernel stays HBM-bound), and only w_0 is loaded at those positions.  Every u is formed
// by exactly the arithmetic of the transfer kernel (same order, same constants), so fu is
// bit-identical to the two-kernel path.  Saves writing and re-reading u: 2 of the 16.4 words per
// grid-point update of the epoch.
#include "mg_march.h"
#include "poisson.h"

namespace odil {

// Interpolated coarse contribution at fine offset (ez, ey, ex) in {-1, 0, 1, 2}^3 relative to the
// fine cell (2jz, 2jy, 2jx): coarse base (e + 2) / 2 - 1 and parity e & 1 per axis; reference
// order (rz, ry, rx), weights parity == r ? 1 : 3, scaled by the exact 1/64.
template <typename T, int EZ, int EY, int EX>
__device__ inline T synth_val(const T (&v)[3][3][3]) {
  constexpr int bz = (EZ + 2) / 2 - 1, by = (EY + 2) / 2 - 1, bx = (EX + 2) / 2 - 1;
  constexpr int sz = EZ & 1, sy = EY & 1, sx = EX & 1;
  T s = T(0);
#pragma unroll
  for (int rz = 0; rz < 2; ++rz)
#pragma unroll
    for (int ry = 0; ry < 2; ++ry)
#pragma unroll
      for (int rx = 0; rx < 2; ++rx) {
        const int w = (sz == rz ? 1 : 3) * (sy == ry ? 1 : 3) * (sx == rx ? 1 : 3);
        s = s + T(w) * v[bz + sz + rz][by + sy + ry][bx + sx + rx];  // window index 0..2 <-> coarse j-1..j+1
      }
  return s * (T(1) / T(64));
}

// own 2 x 2 of fine plane 2jz + EZ: u = w0 + P
template <typename T, int EZ>
__device__ inline void synth_own(const T (&v)[3][3][3], const PackN<T, 2> (&w)[2], T (&u)[2][2]) {
  u[0][0] = T(1) * w[0].e[0] + synth_val<T, EZ, 0, 0>(v);
  u[0][1] = T(1) * w[0].e[1] + synth_val<T, EZ, 0, 1>(v);
  u[1][0] = T(1) * w[1].e[0] + synth_val<T, EZ, 1, 0>(v);
  u[1][1] = T(1) * w[1].e[1] + synth_val<T, EZ, 1, 1>(v);
}

struct SynthArgs {
  MarchArgs m;
  int64_t loss_z0, loss_z1;  // fine planes that enter the loss
};

// Residual of the four own cells of fine plane fz = 2jz + EZ (EZ in {0, 1}).
//   uc: own values, ub / ua: own values of the planes below / above,
//   wy[2]: w0 packs of rows 2jy-1 and 2jy+2, wx[2][2]: w0 at x = 2jx-1 / 2jx+2 of the two own rows.
// JAC: f is the damped-Jacobi update q - (A u - rhs) wd instead of the residual, wd[iy][ix] = omega / diag of the plane.
template <typename T, int EZ, bool JAC = false>
__device__ inline void residual_plane(const T (&v)[3][3][3], const T (&uc)[2][2], const T (&ub)[2][2],
                                      const T (&ua)[2][2], const PackN<T, 2> (&wy)[2], const T (&wx)[2][2],
                                      const PackN<T, 2> (&r)[2], int fz, int fy0, int fx0, int FZ, int FY, int FX,
                                      const H2<T>& h, T (&f)[2][2], const T (*wd)[2][2] = nullptr) {
  // edge neighbours: rows 2jy-1 and 2jy+2 at the own x, columns 2jx-1 and 2jx+2 at the own rows
  T ylo[2], yhi[2], xlo[2], xhi[2];
  ylo[0] = T(1) * wy[0].e[0] + synth_val<T, EZ, -1, 0>(v);
  ylo[1] = T(1) * wy[0].e[1] + synth_val<T, EZ, -1, 1>(v);
  yhi[0] = T(1) * wy[1].e[0] + synth_val<T, EZ, 2, 0>(v);
  yhi[1] = T(1) * wy[1].e[1] + synth_val<T, EZ, 2, 1>(v);
  xlo[0] = T(1) * wx[0][0] + synth_val<T, EZ, 0, -1>(v);
  xlo[1] = T(1) * wx[1][0] + synth_val<T, EZ, 1, -1>(v);
  xhi[0] = T(1) * wx[0][1] + synth_val<T, EZ, 0, 2>(v);
  xhi[1] = T(1) * wx[1][1] + synth_val<T, EZ, 1, 2>(v);
#pragma unroll
  for (int iy = 0; iy < 2; ++iy)
#pragma unroll
    for (int ix = 0; ix < 2; ++ix) {
      const int y = fy0 + iy, x = fx0 + ix;
      const T q = uc[iy][ix];
      const T ym = iy == 0 ? ylo[ix] : uc[0][ix], yp = iy == 0 ? uc[1][ix] : yhi[ix];
      const T xm = ix == 0 ? xlo[iy] : uc[iy][0], xp = ix == 0 ? uc[iy][1] : xhi[iy];
      T acc = axis_term<T>(q, ub[iy][ix], ua[iy][ix], fz == 0, fz == FZ - 1, h, 0);
      acc = acc + axis_term<T>(q, ym, yp, y == 0, y == FY - 1, h, 1);
      acc = acc + axis_term<T>(q, xm, xp, x == 0, x == FX - 1, h, 2);
      if constexpr (JAC) {  // k_poisson_jacobi's expression
        const bool zw = fz == 0 || fz == FZ - 1, yw = y == 0 || y == FY - 1, xw = x == 0 || x == FX - 1;
        const T w = zw ? (yw ? (xw ? wd[1][1][1] : wd[1][1][0]) : (xw ? wd[1][0][1] : wd[1][0][0]))
                       : (yw ? (xw ? wd[0][1][1] : wd[0][1][0]) : (xw ? wd[0][0][1] : wd[0][0][0]));
        f[iy][ix] = q - (acc - r[iy].e[ix]) * w;
      }
      else
        f[iy][ix] = acc - r[iy].e[ix];
    }
}

// The same value in every lane, kept in scalar registers.
__device__ __forceinline__ double uniform_value(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float uniform_value(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

// omega / diag by walls touched, w[z wall][y wall][x wall] (k_poisson_jacobi: the diagonal is the sum over the axes of
// (-2 / h^2) (1 + walls touched)): eight values that are the same in every lane -- scalar registers; one set per own
// cell in vector registers pushed this kernel over its register budget (255 + 20 spilled).
template <typename T>
__device__ __forceinline__ void jacobi_weight_table(const H2<T>& h, T omega, T (&w)[2][2][2]) {
#pragma unroll
  for (int zw = 0; zw < 2; ++zw)
#pragma unroll
    for (int yw = 0; yw < 2; ++yw)
#pragma unroll
      for (int xw = 0; xw < 2; ++xw) {
        const T dz = div_h2<T>(T(-2), h, 0) * T(1 + zw), dy = div_h2<T>(T(-2), h, 1) * T(1 + yw);
        w[zw][yw][xw] = uniform_value(omega / ((dz + dy) + div_h2<T>(T(-2), h, 2) * T(1 + xw)));
      }
}

// JAC: one damped-Jacobi sweep of the same operator on u = w0 + P coarse instead of its residual (the first
// post-smoothing sweep of a V-cycle with the coarse-grid correction formed in registers: x + P x_c is never
// stored, 3 1/8 words per cell instead of 5 1/8); fu receives the new iterate, no loss.
template <typename T, bool JAC = false>
__global__ __launch_bounds__(kBlock) void k_poisson_residual_synth(const T* __restrict__ coarse,
                                                                   const T* __restrict__ w0,
                                                                   const T* __restrict__ rhs, T* __restrict__ fu,
                                                                   SynthArgs sa, H2<T> h,
                                                                   double* __restrict__ partials, T omega = T(0)) {
  const MarchArgs& a = sa.m;
  const int cnz = a.cn[0], cny = a.cn[1], cnx = a.cn[2];
  const int FZ = a.fn[0], FY = a.fn[1], FX = a.fn[2];
  const int64_t cplane = (int64_t)cny * cnx, fplane = (int64_t)FY * FX;
  double local = 0.0;
  int zc, yt, xt;
  const bool have = unit_decode(a.usched, zc, yt, xt);
  const int lx = threadIdx.x % a.tx, ly = threadIdx.x / a.tx;
  const int jy = yt * a.ty + ly, jx = xt * a.tx + lx;
  if (have && jy < cny && jx < cnx) {
    const int z0 = zc * a.usched.ZC;
    const int z1 = z0 + a.usched.ZC < cnz ? z0 + a.usched.ZC : cnz;
    const TapN<3> tx = tapn<3>(jx, cnx), ty = tapn<3>(jy, cny);
    const int fy0 = 2 * jy, fx0 = 2 * jx;
    // clamped positions of the edge neighbours (values beyond a wall are discarded by axis_term)
    const int64_t row0 = (int64_t)fy0 * FX + fx0, row1 = row0 + FX;
    const int64_t rowm = (int64_t)(fy0 == 0 ? 0 : fy0 - 1) * FX + fx0;
    const int64_t rowp = (int64_t)(fy0 + 2 >= FY ? FY - 1 : fy0 + 2) * FX + fx0;
    const int xm = fx0 == 0 ? 0 : -1, xp = fx0 + 2 >= FX ? 1 : 2;
    // omega / diag of the four own cells on a plane away from the z walls (k_poisson_jacobi: the diagonal is
    // sum over axes of (-2 / h^2) (1 + walls touched)); the two wall planes of the array form theirs where needed
    T wdt[2][2][2];
    if constexpr (JAC) jacobi_weight_table<T>(h, omega, wdt);
    T v[3][3][3];
    load_plane<T, 1>(coarse, z0 - 1, cnz, cplane, cnx, ty, tx, T(1), v[0]);
    load_plane<T, 1>(coarse, z0, cnz, cplane, cnx, ty, tx, T(1), v[1]);
    // own values of the fine planes 2 z0 - 1 and 2 z0 (the first is beyond the wall when z0 == 0)
    T uA[2][2], uB[2][2];
    {
      // relative to coarse plane z0 these planes have offsets -1 and 0: both read the window rows
      // 0 and 1 (coarse planes z0 - 1, z0), which are loaded; row 2 is not touched yet
      PackN<T, 2> wa[2], wb[2];
      const int64_t pa = (int64_t)(z0 == 0 ? 0 : 2 * z0 - 1) * fplane, pb = (int64_t)(2 * z0) * fplane;
      wa[0] = stream_ld<T, 2>(w0 + pa + row0, false);
      wa[1] = stream_ld<T, 2>(w0 + pa + row1, false);
      wb[0] = stream_ld<T, 2>(w0 + pb + row0, false);
      wb[1] = stream_ld<T, 2>(w0 + pb + row1, false);
      synth_own<T, -1>(v, wa, uA);
      synth_own<T, 0>(v, wb, uB);
    }
    for (int jz = z0; jz < z1; ++jz) {
      const int fzB = 2 * jz, fzC = 2 * jz + 1, fzD = 2 * jz + 2;
      const int64_t pB = (int64_t)fzB * fplane, pC = (int64_t)fzC * fplane;
      const int64_t pD = (int64_t)(fzD >= FZ ? FZ - 1 : fzD) * fplane;
      // the HBM streams of this step first: w0 of the two new own planes, the edge rows / columns of
      // the two planes that are finalised, rhs
      PackN<T, 2> wC[2], wD[2], wyB[2], wyC[2], rB[2], rC[2];
      T wxB[2][2], wxC[2][2];
      // (w0 rows are re-read as edge rows by the neighbouring threads of the XCD: cached loads;
      // rhs and fu are touched once: streamed)
      wC[0] = stream_ld<T, 2>(w0 + pC + row0, false);
      wC[1] = stream_ld<T, 2>(w0 + pC + row1, false);
      wD[0] = stream_ld<T, 2>(w0 + pD + row0, false);
      wD[1] = stream_ld<T, 2>(w0 + pD + row1, false);
      wyB[0] = stream_ld<T, 2>(w0 + pB + rowm, false);
      wyB[1] = stream_ld<T, 2>(w0 + pB + rowp, false);
      wyC[0] = stream_ld<T, 2>(w0 + pC + rowm, false);
      wyC[1] = stream_ld<T, 2>(w0 + pC + rowp, false);
      rB[0] = stream_ld<T, 2>(rhs + pB + row0, true);
      rB[1] = stream_ld<T, 2>(rhs + pB + row1, true);
      rC[0] = stream_ld<T, 2>(rhs + pC + row0, true);
      rC[1] = stream_ld<T, 2>(rhs + pC + row1, true);
#pragma unroll
      for (int iy = 0; iy < 2; ++iy) {
        wxB[iy][0] = w0[pB + row0 + iy * FX + xm];
        wxB[iy][1] = w0[pB + row0 + iy * FX + xp];
        wxC[iy][0] = w0[pC + row0 + iy * FX + xm];
        wxC[iy][1] = w0[pC + row0 + iy * FX + xp];
      }
      load_plane<T, 1>(coarse, jz + 1, cnz, cplane, cnx, ty, tx, T(1), v[2]);
      T uC[2][2], uD[2][2];
      synth_own<T, 1>(v, wC, uC);
      synth_own<T, 2>(v, wD, uD);
      T fB[2][2], fC[2][2];
      if constexpr (JAC) {
        residual_plane<T, 0, true>(v, uB, uA, uC, wyB, wxB, rB, fzB, fy0, fx0, FZ, FY, FX, h, fB, wdt);
        residual_plane<T, 1, true>(v, uC, uB, uD, wyC, wxC, rC, fzC, fy0, fx0, FZ, FY, FX, h, fC, wdt);
      } else {
        residual_plane<T, 0>(v, uB, uA, uC, wyB, wxB, rB, fzB, fy0, fx0, FZ, FY, FX, h, fB);
        residual_plane<T, 1>(v, uC, uB, uD, wyC, wxC, rC, fzC, fy0, fx0, FZ, FY, FX, h, fC);
      }
      if (fu) {
#pragma unroll
        for (int iy = 0; iy < 2; ++iy) {
          PackN<T, 2> o;
          o.e[0] = fB[iy][0], o.e[1] = fB[iy][1];
          stream_st<T, 2>(fu + pB + row0 + iy * FX, o, true);
          o.e[0] = fC[iy][0], o.e[1] = fC[iy][1];
          stream_st<T, 2>(fu + pC + row0 + iy * FX, o, true);
        }
      }
      const bool inB = fzB >= sa.loss_z0 && fzB < sa.loss_z1, inC = fzC >= sa.loss_z0 && fzC < sa.loss_z1;
#pragma unroll
      for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
          if (inB) local += (double)(fB[iy][ix] * fB[iy][ix]);
          if (inC) local += (double)(fC[iy][ix] * fC[iy][ix]);
        }
#pragma unroll
      for (int iy = 0; iy < 2; ++iy)
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
          uA[iy][ix] = uC[iy][ix];
          uB[iy][ix] = uD[iy][ix];
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
  if constexpr (!JAC) {
    const double total = block_sum(local);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
  }
}

template <typename T>
static int poisson_residual_synth(const T* coarse, const T* w0, const T* rhs, T* fu, const int64_t* cshape,
                                  const T* h2, int64_t z0, int64_t z1, double denom, double* partials, T* loss,
                                  void* stream, bool jacobi = false, T omega = T(0)) {
  if (!coarse || !w0 || !rhs || (!jacobi && (!partials || !loss)) || (jacobi && (!fu || fu == w0))) {
    set_error("poisson_residual_synth: null pointer (or the Jacobi sweep in place)");
    return ODIL_E_INVAL;
  }
  SynthArgs sa;
  MarchArgs& m = sa.m;
  for (int i = 0; i < 3; ++i) {
    if (cshape[i] < 2 || cshape[i] >= (1 << 29)) {
      set_error("poisson_residual_synth: coarse extent %lld on axis %d", (long long)cshape[i], i);
      return ODIL_E_INVAL;
    }
    m.cn[i] = (int)cshape[i];
    m.fn[i] = 2 * m.cn[i];
  }
  if (!((reinterpret_cast<uintptr_t>(w0) | reinterpret_cast<uintptr_t>(rhs) | reinterpret_cast<uintptr_t>(fu)) %
            (2 * sizeof(T)) ==
        0)) {
    set_error("poisson_residual_synth: arrays must be aligned to %d bytes", (int)(2 * sizeof(T)));
    return ODIL_E_INVAL;
  }
  m.cut_lo = m.cut_hi = 0;
  m.lead_loc = 0;
  m.lead_cn = m.lead_fn = 1;
  m.nt = (int64_t)m.fn[0] * m.fn[1] * m.fn[2] * (int64_t)sizeof(T) > kStreamBytes;
  int tx = 1;
  while (tx < m.cn[2] && tx < kBlock) tx *= 2;
  m.tx = tx;
  m.ty = kBlock / tx;
  const int64_t ytiles = (m.cn[1] + m.ty - 1) / m.ty, xtiles = (m.cn[2] + m.tx - 1) / m.tx;
  if ((int64_t)m.cn[0] * ytiles * xtiles >= ((int64_t)1 << 31)) {
    set_error("poisson_residual_synth: grid too large for one launch");
    return ODIL_E_INVAL;
  }
  m.usched = make_unit_sched(m.cn[0], ytiles, xtiles);
  const int grid = unit_grid(m.usched);
  if (grid > kMaxPartials) {
    set_error("poisson_residual_synth: %d workgroups exceed the reduction workspace", grid);
    return ODIL_E_INVAL;
  }
  sa.loss_z0 = z0;
  sa.loss_z1 = z1 < 0 ? m.fn[0] : z1;
  T hh[3] = {h2[0], h2[1], h2[2]};
  if (jacobi) {
    hipLaunchKernelGGL((k_poisson_residual_synth<T, true>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, coarse,
                       w0, rhs, fu, sa, make_h2<T>(hh), partials, omega);
    return check_launch("k_poisson_residual_synth<jacobi>");
  }
  hipLaunchKernelGGL((k_poisson_residual_synth<T, false>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, coarse, w0,
                     rhs, fu, sa, make_h2<T>(hh), partials, T(0));
  if (int e = check_launch("k_poisson_residual_synth")) return e;
  const double size = denom > 0.0 ? denom : (double)m.fn[0] * m.fn[1] * m.fn[2];
  return launch_final_reduce<T>(partials, grid, 0, 1, size, loss, (hipStream_t)stream);
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_poisson_residual_synth_f64(const double* coarse, const double* w0, const double* rhs, double* fu,
                                    const int64_t* cshape, const double* h2, int64_t z0, int64_t z1, double denom,
                                    double* partials, double* loss, void* stream) {
  return poisson_residual_synth<double>(coarse, w0, rhs, fu, cshape, h2, z0, z1, denom, partials, loss, stream);
}
int odil_poisson_residual_synth_f32(const float* coarse, const float* w0, const float* rhs, float* fu,
                                    const int64_t* cshape, const float* h2, int64_t z0, int64_t z1, double denom,
                                    double* partials, float* loss, void* stream) {
  return poisson_residual_synth<float>(coarse, w0, rhs, fu, cshape, h2, z0, z1, denom, partials, loss, stream);
}
int odil_poisson_jacobi_synth_f64(const double* coarse, const double* x, const double* rhs, double* xout,
                                  const int64_t* cshape, const double* h2, double omega, void* stream) {
  return poisson_residual_synth<double>(coarse, x, rhs, xout, cshape, h2, 0, -1, 0.0, nullptr, nullptr, stream, true,
                                        omega);
}
int odil_poisson_jacobi_synth_f32(const float* coarse, const float* x, const float* rhs, float* xout,
                                  const int64_t* cshape, const float* h2, float omega, void* stream) {
  return poisson_residual_synth<float>(coarse, x, rhs, xout, cshape, h2, 0, -1, 0.0, nullptr, nullptr, stream, true,
                                       omega);
}
}  // extern "C"
