// TWO damped-Jacobi sweeps in ONE pass over memory (temporal blocking of the multigrid smoothers of the Newton solve,
// odil_amd/gmg.py; the reference hands the Newton system to SuperLU / pyamg, src/odil/linsolver.py:17-26, 61-72).
//
// A sweep is HBM-bound: x and b in, x' out (3 words per cell for the constant-coefficient Poisson stencil, 2 d + 4 with
// coefficient arrays).  Two sweeps as two launches move that twice; here the intermediate iterate y1 never leaves the
// CU, so a PAIR of sweeps costs what one did.  The result is bit-identical to two single sweeps (same expressions, same
// operation order) -- asserted by the tests.
//
// Mapping.  A WAVE owns one row segment of 64 packs (16 bytes per lane) and marches along z; a workgroup is 16 waves =
// 16 consecutive rows of one x-window, of which the inner 14 are owned (their second sweep is stored).
//   x neighbours  adjacent lanes' registers (wave-wide DPP shifts, common.h); an x-window that is not the whole row
//                 carries one halo pack per side, whose first sweep is recomputed here (62 owned packs of 64)
//   y neighbours  first sweep: rows y +- 1 of x from memory (L2 hits of the neighbouring waves' own rows, as in
//                 k_poisson_jacobi); second sweep: rows y +- 1 of y1 through LDS (one plane, double-buffered: one
//                 barrier per plane)
//   z neighbours  registers: x planes p - 1 .. p + 2 and y1 planes p - 2 .. p of the lane's own pack
// Step t forms y1 on plane p = z0 - 1 + t and emits the second sweep of plane p - 1; loads run one plane ahead of
// their use and are issued first in every step.
#include "poisson.h"

namespace odil {

constexpr int kS2Waves = 16;            // rows per workgroup
constexpr int kS2Own = kS2Waves - 2;    // rows whose second sweep is stored

struct Smooth2Args {
  int64_t n[3];    // canonical (Z, Y, X) cells
  int active[3];
  int packs;       // X / V
  int own_x;       // packs owned per x-window (the whole row: packs)
  int halo_x;      // 1: windows of 64 packs with a halo pack per side; 0: one window = the whole row
  int stream;      // non-temporal stores of the result
  UnitSched usched;  // units (z-chunk, y-tile, x-window)
};

// q - (A q - r) w of k_poisson_jacobi for the V cells of a pack, term by term
template <typename T, int V>
__device__ __forceinline__ void jacobi_pack(const T (&qc)[V], const T (&zm)[V], const T (&zp)[V], const T (&ym)[V],
                                            const T (&yp)[V], T left, T right, const T (&r)[V], int64_t z, int64_t y,
                                            int64_t x0, const Smooth2Args& a, const H2<T>& h, T w_in, T w_wall,
                                            T (&out)[V]) {
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int64_t x = x0 + i;
    const T q = qc[i];
    T acc = T(0);
    if (a.active[0]) acc = axis_term<T>(q, zm[i], zp[i], z == 0, z == Z - 1, h, 0);
    if (a.active[1]) acc = acc + axis_term<T>(q, ym[i], yp[i], y == 0, y == Y - 1, h, 1);
    {
      const T xm = i == 0 ? left : qc[i - 1 >= 0 ? i - 1 : 0];
      const T xp = i == V - 1 ? right : qc[i + 1 < V ? i + 1 : i];
      acc = acc + axis_term<T>(q, xm, xp, x == 0, x == X - 1, h, 2);
    }
    out[i] = q - (acc - r[i]) * ((x == 0 || x == X - 1) ? w_wall : w_in);
  }
}

template <typename T, bool HASY>
__global__ __launch_bounds__(HASY ? 64 * kS2Waves : 64) void k_poisson_jacobi2(const T* __restrict__ u,
                                                                               const T* __restrict__ rhs,
                                                                               T* __restrict__ uout, Smooth2Args a,
                                                                               H2<T> h, T omega1, T omega2) {
  constexpr int V = VecOf<T>::N;
  constexpr int NW = HASY ? kS2Waves : 1;
  __shared__ T ybuf[HASY ? 2 * NW * 64 * V : 1];
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;  // whole workgroup
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  // the lane's pack: window position -> pack index (wrapped into the row: every address is valid, what lies beyond a
  // wall or outside the owned range is computed and discarded)
  int64_t xp = (int64_t)xt * a.own_x - a.halo_x + lane;
  const int own_here = (int64_t)(xt + 1) * a.own_x <= a.packs ? a.own_x : a.packs - xt * a.own_x;
  const bool lane_own = a.halo_x ? (lane >= 1 && lane <= own_here) : lane < a.packs;
  xp = ((xp % a.packs) + a.packs) % a.packs;
  const int64_t x0 = xp * V;
  int64_t y = HASY ? (int64_t)yt * kS2Own - 1 + row : 0;
  const bool row_own = HASY ? (row >= 1 && row <= kS2Own && y < Y) : true;
  y = ((y % Y) + Y) % Y;
  const int64_t z0 = (int64_t)zc * a.usched.ZC;
  const int64_t z1 = z0 + a.usched.ZC < Z ? z0 + a.usched.ZC : Z;
  const int64_t c_off = y * sy + x0;
  const int64_t ym_off = (y == 0 ? Y - 1 : y - 1) * sy + x0, yp_off = (y == Y - 1 ? 0 : y + 1) * sy + x0;
  // omega / diag by (z at a wall, x at a wall) for both sweeps -- the expressions of k_poisson_jacobi
  T dterm[3];
#pragma unroll
  for (int ax = 0; ax < 3; ++ax) dterm[ax] = a.active[ax] ? div_h2<T>(T(-2), h, ax) : T(0);
  const T dy = a.active[1] ? dterm[1] * T(1 + (y == 0) + (y == Y - 1)) : T(0);
  T wd1[2][2], wd2[2][2];
#pragma unroll
  for (int zw = 0; zw < 2; ++zw)
#pragma unroll
    for (int xw = 0; xw < 2; ++xw) {
      const T dzv = a.active[0] ? dterm[0] * T(1 + zw) : T(0);
      const T diag = (dzv + dy) + dterm[2] * T(1 + xw);
      wd1[zw][xw] = omega1 / diag;
      wd2[zw][xw] = omega2 / diag;
    }
  auto wrapz = [Z](int64_t q) { return ((q % Z) + Z) % Z; };
  const int64_t p0 = z0 - 1;
  T u_m[V], u_c[V], u_p[V], u_n[V];      // own row: planes p - 1, p, p + 1, p + 2
  T nm_c[V], np_c[V], nm_n[V], np_n[V];  // rows y -+ 1 of planes p, p + 1
  T b_m[V], b_c[V], b_n[V];              // rhs of planes p - 1, p, p + 1
  T y_m[V], y_c[V];                      // first sweep of planes p - 2, p - 1
  T ym_c[V], yp_c[V];                    // rows y -+ 1 of the first sweep of plane p - 1
  load_vec<T, V, true>(u + wrapz(p0 - 1) * sz + c_off, V, u_m);
  load_vec<T, V, true>(u + wrapz(p0) * sz + c_off, V, u_c);
  load_vec<T, V, true>(u + wrapz(p0 + 1) * sz + c_off, V, u_p);
  load_vec<T, V, true>(rhs + wrapz(p0) * sz + c_off, V, b_c);
  if (HASY) {
    load_vec<T, V, true>(u + wrapz(p0) * sz + ym_off, V, nm_c);
    load_vec<T, V, true>(u + wrapz(p0) * sz + yp_off, V, np_c);
  }
#pragma unroll
  for (int i = 0; i < V; ++i) y_m[i] = y_c[i] = ym_c[i] = yp_c[i] = b_m[i] = T(0);
  const int nt = (int)(z1 - z0) + 2;
  for (int t = 0; t < nt; ++t) {
    const int64_t p = p0 + t;
    // (1) the loads of this step: they are consumed by the NEXT step
    {
      const int64_t pa = wrapz(p + 2) * sz, pb = wrapz(p + 1) * sz;
      load_vec<T, V, true>(u + pa + c_off, V, u_n);
      load_vec<T, V, true>(rhs + pb + c_off, V, b_n);
      if (HASY) {
        load_vec<T, V, true>(u + pb + ym_off, V, nm_n);
        load_vec<T, V, true>(u + pb + yp_off, V, np_n);
      }
    }
    // (2) first sweep on plane p
    const int64_t pw = wrapz(p);
    T y_p[V];
    {
      T left = from_prev_lane(u_c[V - 1]), right = from_next_lane(u_c[0]);
      const bool zw = a.active[0] && (pw == 0 || pw == Z - 1);
      jacobi_pack<T, V>(u_c, u_m, u_p, nm_c, np_c, left, right, b_c, pw, y, x0, a, h, zw ? wd1[1][0] : wd1[0][0],
                        zw ? wd1[1][1] : wd1[0][1], y_p);
    }
    T ym_p[V], yp_p[V];
    if (HASY) {
      T* buf = ybuf + (t & 1) * (NW * 64 * V);
      store_vec<T, V, true>(buf + (row * 64 + lane) * V, V, y_p);
      __syncthreads();
      const int rm = row == 0 ? 0 : row - 1, rp = row == NW - 1 ? NW - 1 : row + 1;
      load_vec<T, V, true>(buf + (rm * 64 + lane) * V, V, ym_p);
      load_vec<T, V, true>(buf + (rp * 64 + lane) * V, V, yp_p);
    }
    // (3) second sweep on plane p - 1
    if (t >= 2) {
      const int64_t z = p - 1;  // in [z0, z1)
      T out[V];
      T left = from_prev_lane(y_c[V - 1]), right = from_next_lane(y_c[0]);
      const bool zw = a.active[0] && (z == 0 || z == Z - 1);
      jacobi_pack<T, V>(y_c, y_m, y_p, ym_c, yp_c, left, right, b_m, z, y, x0, a, h, zw ? wd2[1][0] : wd2[0][0],
                        zw ? wd2[1][1] : wd2[0][1], out);
      if (row_own && lane_own) {
        if (a.stream)
          store_vec<T, V, true, true>(uout + z * sz + c_off, V, out);
        else
          store_vec<T, V, true, false>(uout + z * sz + c_off, V, out);
      }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      u_m[i] = u_c[i], u_c[i] = u_p[i], u_p[i] = u_n[i];
      b_m[i] = b_c[i], b_c[i] = b_n[i];
      y_m[i] = y_c[i], y_c[i] = y_p[i];
      if (HASY) {
        nm_c[i] = nm_n[i], np_c[i] = np_n[i];
        ym_c[i] = ym_p[i], yp_c[i] = yp_p[i];
      }
    }
  }
}

template <typename T>
static int smooth2_args(Smooth2Args& a, const int64_t* shape, int ndim, const T* h2, T h[3], int zc_hint,
                        const char* what) {
  constexpr int V = VecOf<T>::N;
  if (ndim < 1 || ndim > 3 || !shape || !h2) {
    set_error("%s: ndim=%d out of range [1,3] or null shape/h2", what, ndim);
    return ODIL_E_INVAL;
  }
  for (int i = 0; i < 3; ++i) {
    a.n[i] = 1;
    a.active[i] = 0;
    h[i] = T(1);
  }
  static const int map3[3][3] = {{2, 0, 0}, {0, 2, 0}, {0, 1, 2}};  // as poisson.hip: fill_args
  for (int i = 0; i < ndim; ++i) {
    const int c = map3[ndim - 1][i];
    a.n[c] = shape[i];
    a.active[c] = 1;
    h[c] = h2[i];
    if (shape[i] < 2) {
      set_error("%s: extent %lld on axis %d must be >= 2", what, (long long)shape[i], i);
      return ODIL_E_INVAL;
    }
  }
  if (a.n[2] % V) {
    set_error("%s: the last extent must be a multiple of %d (use two single sweeps)", what, V);
    return ODIL_E_INVAL;
  }
  a.packs = (int)(a.n[2] / V);
  int64_t nxt;
  if (a.packs <= 64) {
    a.halo_x = 0, a.own_x = a.packs, nxt = 1;
  } else {
    a.halo_x = 1;
    nxt = (a.packs + 61) / 62;
    a.own_x = (int)((a.packs + nxt - 1) / nxt);
  }
  const int64_t nyt = a.active[1] ? (a.n[1] + kS2Own - 1) / kS2Own : 1;
  if (a.n[0] * nyt * nxt >= ((int64_t)1 << 31)) {
    set_error("%s: grid too large for one launch", what);
    return ODIL_E_INVAL;
  }
  a.stream = a.n[0] * a.n[1] * a.n[2] * (int64_t)sizeof(T) > kStreamBytes;
  // chunks of planes: every chunk primes its window with two extra planes of the first sweep, so they are long --
  // as long as the launch still has a few workgroups per CU
  const int64_t per_plane = nyt * nxt;
  int64_t zc = zc_hint > 0 ? zc_hint : 64;
  while (zc_hint <= 0 && zc > 8 && ((a.n[0] + zc - 1) / zc) * per_plane < 1024) zc /= 2;
  if (zc > a.n[0]) zc = a.n[0];
  a.usched = make_unit_sched(a.n[0], nyt, nxt, 1);
  // (make_unit_sched derives the chunk length from a target count; here it is set directly)
  a.usched.ZC = (int)zc;
  a.usched.ZCH = (int)((a.n[0] + zc - 1) / zc);
  if (a.usched.axis == 1) {
    a.usched.per_xcd = a.usched.ZCH * a.usched.chunk * a.usched.XS;
  } else if (a.usched.ZCH >= kNumXcd) {
    a.usched.axis = 0;
    a.usched.chunk = 0;
    a.usched.per_xcd = (int)(((int64_t)a.usched.ZCH * a.usched.Y * a.usched.XS + kNumXcd - 1) / kNumXcd);
  } else {
    a.usched.axis = -1;
    a.usched.chunk = 0;
    a.usched.per_xcd = a.usched.ZCH * a.usched.Y * a.usched.XS;
  }
  return 0;
}

template <typename T>
static int poisson_jacobi2(const T* u, const T* rhs, T* uout, const int64_t* shape, int ndim, const T* h2, T omega1,
                           T omega2, int zc_hint, void* stream) {
  Smooth2Args a;
  T h[3];
  if (int e = smooth2_args<T>(a, shape, ndim, h2, h, zc_hint, "poisson_jacobi2")) return e;
  if (!u || !rhs || !uout || u == uout) {
    set_error("poisson_jacobi2: null pointer, or the sweeps in place (x_out must differ from x)");
    return ODIL_E_INVAL;
  }
  const int grid = unit_grid(a.usched);
  if (a.active[1])
    hipLaunchKernelGGL((k_poisson_jacobi2<T, true>), dim3(grid), dim3(64 * kS2Waves), 0, (hipStream_t)stream, u, rhs,
                       uout, a, make_h2<T>(h), omega1, omega2);
  else
    hipLaunchKernelGGL((k_poisson_jacobi2<T, false>), dim3(grid), dim3(64), 0, (hipStream_t)stream, u, rhs, uout, a,
                       make_h2<T>(h), omega1, omega2);
  return check_launch("k_poisson_jacobi2");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_poisson_jacobi2_f64(const double* u, const double* rhs, double* uout, const int64_t* shape, int ndim,
                             const double* h2, double omega1, double omega2, int zc_hint, void* stream) {
  return poisson_jacobi2<double>(u, rhs, uout, shape, ndim, h2, omega1, omega2, zc_hint, stream);
}
int odil_poisson_jacobi2_f32(const float* u, const float* rhs, float* uout, const int64_t* shape, int ndim,
                             const float* h2, float omega1, float omega2, int zc_hint, void* stream) {
  return poisson_jacobi2<float>(u, rhs, uout, shape, ndim, h2, omega1, omega2, zc_hint, stream);
}
}  // extern "C"
