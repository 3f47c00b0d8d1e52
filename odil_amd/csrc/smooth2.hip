// TWO damped-Jacobi sweeps in ONE pass over memory (temporal blocking of the multigrid smoothers of the Newton solve,
// odil_amd/gmg.py; the reference hands the Newton system to SuperLU / pyamg, src/odil/linsolver.py:17-26, 61-72).
//
// A sweep is HBM-bound: x and b in, x' out (3 words per cell for the constant-coefficient Poisson stencil, 2 d + 4 with
// coefficient arrays).  Two sweeps as two launches move that twice; here the intermediate iterate y1 never leaves the
// CU, so a PAIR of sweeps costs what one did.  The result is bit-identical to two single sweeps (same expressions, same
// operation order) -- asserted by the tests.
//
// Mapping.  A WAVE owns one row segment of 64 packs (16 bytes per lane) and marches along z; a workgroup is 16 waves =
// 16 consecutive rows of one x-window: all of them load their row of x, rows 1 .. 14 form the first sweep, rows 2 .. 13
// (the 12 owned rows) the second, which is stored.
//   x neighbours  adjacent lanes' registers (wave-wide DPP shifts, common.h); an x-window that is not the whole row
//                 carries one halo pack per side, whose first sweep is recomputed here (62 owned packs of 64)
//   y neighbours  LDS: every wave publishes its row of x (plane p + 1) and of y1 (plane p), two buffers each, ONE
//                 barrier per plane
//   z neighbours  registers: x planes p - 1 .. p + 1 and y1 planes p - 2 .. p of the lane's own pack
// Step t forms y1 on plane p = z0 - 1 + t and emits the second sweep of plane p - 1.  Loads (one of x, one of b per
// wave and step) are issued first and consumed kS2Ahead steps later: register rings with compile-time indices -- the
// march is unrolled by the ring length, so nothing is copied and no load is waited for before its use.
#include "poisson.h"

namespace odil {

constexpr int kS2Waves = 16;            // rows per workgroup
constexpr int kS2Own = kS2Waves - 4;    // rows whose second sweep is stored
constexpr int kS2Ring = 6;              // register ring (planes) = unroll factor of the march
constexpr int kS2Ahead = 2;             // steps between a load and its use (<= kS2Ring - 4)
#ifndef S2_SVAR_DC
#define S2_SVAR_DC 0  // steps the coefficient loads of the variable-coefficient pair run ahead
#endif
constexpr int kS2CRows = kS2Waves / 2 + 2;  // coarse rows behind a workgroup's fine rows (fused prolongation)

struct Smooth2Args {
  int64_t n[3];    // canonical (Z, Y, X) cells
  int active[3];
  int packs;       // X / V
  int own_x;       // packs owned per x-window (the whole row: packs)
  int halo_x;      // 1: windows of 64 packs with a halo pack per side; 0: one window = the whole row
  int nxt;         // x-windows per row
  int stream;      // non-temporal stores of the result
  int slot[3];     // (variable coefficients) index of the -e_a array among the coefficient arrays, +e_a = slot + 1
  int64_t size;    // cells of the array
  UnitSched usched;  // units (z-chunk, y-tile, x-window)
};

__device__ __forceinline__ double s2_uniform(double x) {
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
__device__ __forceinline__ float s2_uniform(float x) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(x)));
}

// q - (A q - r) w of k_poisson_jacobi for the V cells of a pack, term by term.  WALLS false: no cell of the wave touches
// a wall on this plane (a wave-uniform fact) -- the same expressions without their index tests.
template <typename T, bool MUL>
__device__ __forceinline__ T s2_div_h2(T v, const H2<T>& h, int ax) {
  if constexpr (MUL) return v * h.inv[ax];
  return div_h2<T>(v, h, ax);
}

// MUL: all 1 / h^2 are exact (powers of two), known at launch -- the instantiation that runs carries no divide
template <typename T, int V, bool WALLS, bool MUL>
__device__ __forceinline__ void jacobi_pack(const T (&qc)[V], const T (&zm)[V], const T (&zp)[V], const T (&ym)[V],
                                            const T (&yp)[V], T left, T right, const T (&r)[V], int z, int y, int x0,
                                            const Smooth2Args& a, const H2<T>& h, T w_in, T w_wall, T (&out)[V]) {
  const int Z = (int)a.n[0], Y = (int)a.n[1], X = (int)a.n[2];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int x = x0 + i;
    const T q = qc[i];
    const T xm = i == 0 ? left : qc[i - 1 >= 0 ? i - 1 : 0];
    const T xp = i == V - 1 ? right : qc[i + 1 < V ? i + 1 : i];
    T acc = T(0);
    if constexpr (WALLS) {
      if (a.active[0]) acc = axis_term<T>(q, zm[i], zp[i], z == 0, z == Z - 1, h, 0);
      if (a.active[1]) acc = acc + axis_term<T>(q, ym[i], yp[i], y == 0, y == Y - 1, h, 1);
      acc = acc + axis_term<T>(q, xm, xp, x == 0, x == X - 1, h, 2);
      out[i] = q - (acc - r[i]) * ((x == 0 || x == X - 1) ? w_wall : w_in);
    } else {
      if (a.active[0]) acc = s2_div_h2<T, MUL>(zp[i] - T(2) * q + zm[i], h, 0);
      if (a.active[1]) acc = acc + s2_div_h2<T, MUL>(yp[i] - T(2) * q + ym[i], h, 1);
      acc = acc + s2_div_h2<T, MUL>(xp - T(2) * q + xm, h, 2);
      out[i] = q - (acc - r[i]) * w_in;
    }
  }
}

// ---- the prolongation of the coarse-grid correction as stage 0 (SYNTH) ------------------------------------------------
// u = x + P x_c for the lane's pack of TWO fine cells (x = 2 j, 2 j + 1 <-> coarse column j) of fine row y, plane s:
// exactly the arithmetic of k_interp_add_march (mg_march.hip: acc_plane / store_plane; reference core.py:606-700 with the
// joint ghost rule core.py:640-643) -- t = sum over (rz, ry, rx) of T(wz wy wx) v, then x + t / 64.
//   c[a][b], r[a][b]   coarse value at (plane a, row b) of the two planes / rows the cell reads, at the CLAMPED resp.
//                      REFLECTED plane / row index, own column (r == c unless that plane or row lies beyond a wall)
//   o[a][b]            that plane or row lies beyond a wall
//   wz0, wy0           weight of the first tap (1: even fine index, 3: odd), the second gets 4 - it
// Columns j -+ 1 are the adjacent lanes' values (lane shifts); beyond an x wall: clamp = own column, reflect = the
// neighbour on the other side.
template <typename T, bool ZY>
__device__ __forceinline__ void synth_pack(const T (&c)[2][2], const T (&r)[2][2], const bool (&o)[2][2], int wz0, int wy0,
                                           bool jlo, bool jhi, T (&x)[2]) {
  T t0 = T(0), t1 = T(0);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const T cc = c[a][b];
      const T clm = from_prev_lane(cc), clp = from_next_lane(cc);
      T rr = cc, rfm = clm, rfp = clp;
      if constexpr (ZY) {
        rr = r[a][b];
        rfm = from_prev_lane(rr);
        rfp = from_next_lane(rr);
      }
      const bool oo = ZY ? o[a][b] : false;
      const T v0 = oo ? T(2) * cc - rr : cc;
      T vm, vp;
      {
        const T cl = jlo ? cc : clm, rf = jlo ? rfp : rfm;
        vm = (oo || jlo) ? T(2) * cl - rf : cl;
      }
      {
        const T cl = jhi ? cc : clp, rf = jhi ? rfm : rfp;
        vp = (oo || jhi) ? T(2) * cl - rf : cl;
      }
      const int wzy = (a == 0 ? wz0 : 4 - wz0) * (b == 0 ? wy0 : 4 - wy0);
      // even fine column: columns (j - 1, j) with weights (1, 3); odd: (j, j + 1) with (3, 1)
      t0 = t0 + T(wzy) * vm;
      t0 = t0 + T(3 * wzy) * v0;
      t1 = t1 + T(3 * wzy) * v0;
      t1 = t1 + T(wzy) * vp;
    }
  const T r64 = T(1) / T(64);
  x[0] = T(1) * x[0] + t0 * r64;
  x[1] = T(1) * x[1] + t1 * r64;
}

// V: cells per lane (16 bytes; 8 bytes for float with SYNTH, where a lane is one coarse column).
// ZERO: the iterate the sweeps start from is the zero vector (every coarse level of a V-cycle): x is not read -- the ring
// holds zeros, the same arithmetic on them gives the same bits (0 - w (A 0 - b) = w b exactly) --, nothing of x goes through
// LDS, and the caller need not zero an array first: 2 words per cell instead of 3 + 1.
template <typename T, int V, bool HASY, bool MUL, bool SYNTH, int D = kS2Ahead, bool ZERO = false>
__global__ __launch_bounds__(HASY ? 64 * kS2Waves : 64) void k_poisson_jacobi2(const T* __restrict__ u,
                                                                               const T* __restrict__ rhs,
                                                                               T* __restrict__ uout,
                                                                               const T* __restrict__ coarse,
                                                                               Smooth2Args a, H2<T> h, T omega1,
                                                                               T omega2) {
  static_assert(!SYNTH || (V == 2 && HASY), "the fused prolongation is 3-D with one coarse column per lane");
  static_assert(!(SYNTH && ZERO), "x + P x_c is not a zero start");
  constexpr int NW = HASY ? kS2Waves : 1;
  constexpr int HY = HASY ? 2 : 0;  // halo rows per side
  __shared__ T ubuf[HASY && !ZERO ? 2 * NW * 64 * V : 1];
  __shared__ T ybuf[HASY ? 2 * NW * 64 * V : 1];
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;  // whole workgroup
  if constexpr (MUL) h.mul_ok[0] = h.mul_ok[1] = h.mul_ok[2] = 1;  // (the wall rows lose their divide as well)
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  // the lane's pack: window position -> pack index (wrapped into the row: every address is valid, what lies beyond a
  // wall or outside the owned range is computed and discarded)
  int64_t xp = (int64_t)xt * a.own_x - a.halo_x + lane;
  const int own_here = (int64_t)(xt + 1) * a.own_x <= a.packs ? a.own_x : a.packs - xt * a.own_x;
  const bool lane_own = a.halo_x ? (lane >= a.halo_x && lane < a.halo_x + own_here) : lane < a.packs;
  // (lanes beyond the window load too -- wrapped, valid addresses: masking them off measured SLOWER, 0.78 -> 0.88 ms)
  xp = ((xp % a.packs) + a.packs) % a.packs;
  const int x0 = (int)xp * V;
  int64_t y = HASY ? (int64_t)yt * kS2Own - HY + row : 0;
  const bool row_own = HASY ? (row >= HY && row < HY + kS2Own && y < Y) : true;
  const bool row_s1 = HASY ? (row >= 1 && row <= NW - 2) : true;  // rows that form the first sweep
  y = ((y % Y) + Y) % Y;
  const int64_t z0 = (int64_t)zc * a.usched.ZC;
  const int64_t z1 = z0 + a.usched.ZC < Z ? z0 + a.usched.ZC : Z;
  const uint32_t c_off = (uint32_t)(y * sy + x0);  // (a plane has < 2^31 cells: smooth2_args)
  // no cell of this wave at a wall of y or x (wave-uniform)
  const bool yx_inner = (!a.active[1] || (y != 0 && y != Y - 1)) && (a.halo_x ? (xt != 0 && xt != a.nxt - 1) : false);
  // omega / diag by (z at a wall, x at a wall) for both sweeps -- the expressions of k_poisson_jacobi; the same in every
  // lane of a wave (y is): scalar registers
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
      wd1[zw][xw] = s2_uniform(omega1 / diag);
      wd2[zw][xw] = s2_uniform(omega2 / diag);
    }
  // plane indices wrap periodically (what lies beyond a wall is discarded by the wall rows); a chunk starts two planes
  // early and the ring runs D + 1 planes ahead
  const int Zi = (int)Z;
  auto wrapz = [Zi](int q) {
    q %= Zi;
    return q < 0 ? q + Zi : q;
  };
  const int p0 = (int)z0 - 1;
  // Rings indexed by the step number modulo R: entry (k mod R) of `uo` holds plane p0 + k of x (SYNTH: of x + P x_c once
  // its step has formed it), of `bb` the rhs of plane p0 + k, of `y1` the first sweep of plane p0 + k.  Every index below
  // is a constant after unrolling.
  constexpr int R = kS2Ring;
  T uo[R][V], bb[R][V], y1[R][V];
  // ---- SYNTH: the coarse planes in an LDS ring of three, the kS2CRows coarse rows the workgroup's fine rows read ------
  // (fine rows 12 yt - 2 .. 12 yt + 13 read the coarse rows 6 yt - 2 .. 6 yt + 7; wave w < kS2CRows fetches row w of a new
  // coarse plane every second step, clamped into the array: the `cl` values of the joint ghost rule)
  const int cnz = Zi / 2, cny = (int)(Y / 2), cnx = a.packs;
  const int64_t cplane = (int64_t)cny * cnx;
  __shared__ T cbuf[SYNTH ? 3 * kS2CRows * 64 : 1];
  const bool jlo = xp == 0, jhi = xp == cnx - 1;
  int crow[2] = {0, 0};  // LDS rows of the two coarse rows this fine row reads
  int cq[2] = {0, 0};    // their coarse row numbers (may lie beyond a wall)
  int srw0 = 1;          // weight of the first row tap
  int cfetch = 0;        // offset of the coarse row this wave fetches + own column
  if constexpr (SYNTH) {
    const int yu = yt * kS2Own - HY + row;  // not wrapped
    const int jy = yu >> 1, ey = yu & 1;
    srw0 = ey ? 3 : 1;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      cq[b] = jy - 1 + ey + b;
      crow[b] = (cq[b] - (yt * (kS2Own / 2) - 2)) * 64 + lane;
    }
    const int qf = yt * (kS2Own / 2) - 2 + row;
    cfetch = (qf < 0 ? 0 : (qf >= cny ? cny - 1 : qf)) * cnx + (int)xp;
  }
  auto zcl = [cnz](int q) { return q < 0 ? 0 : (q >= cnz ? cnz - 1 : q); };
  auto zrf = [cnz](int q) { return q < 0 ? 1 : (q >= cnz ? (cnz >= 2 ? cnz - 2 : 0) : q); };
  auto yrf = [cny](int q) { return q < 0 ? 1 : (q >= cny ? (cny >= 2 ? cny - 2 : 0) : q); };
  // u = x + P x_c in place on the raw pack `xv` of fine plane s; the coarse planes (qa, qa + 1) it reads lie in the ring
  // entries ea, eb
  auto synth = [&](int s, int ea, int eb, T (&xv)[V]) {
    if constexpr (SYNTH) {
      const int qa = (s >> 1) - 1 + (s & 1);
      const bool oza = qa < 0 || qa >= cnz, ozb = qa + 1 < 0 || qa + 1 >= cnz;
      const bool oya = cq[0] < 0 || cq[0] >= cny, oyb = cq[1] < 0 || cq[1] >= cny;
      const int wz0 = (s & 1) ? 3 : 1;
      T c[2][2];
      c[0][0] = cbuf[ea * (kS2CRows * 64) + crow[0]];
      c[0][1] = cbuf[ea * (kS2CRows * 64) + crow[1]];
      c[1][0] = cbuf[eb * (kS2CRows * 64) + crow[0]];
      c[1][1] = cbuf[eb * (kS2CRows * 64) + crow[1]];
      if (oza || ozb || oya || oyb) {  // (wave-uniform: planes / rows next to a wall)
        T r[2][2];
        bool o[2][2];
#pragma unroll
        for (int pa = 0; pa < 2; ++pa)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            r[pa][b] = coarse[(int64_t)zrf(qa + pa) * cplane + (int64_t)yrf(cq[b]) * cnx + xp];
            o[pa][b] = (pa ? ozb : oza) || (b ? oyb : oya);
          }
        synth_pack<T, true>(c, r, o, wz0, srw0, jlo, jhi, xv);
      } else {
        const bool o[2][2] = {{false, false}, {false, false}};
        synth_pack<T, false>(c, c, o, wz0, srw0, jlo, jhi, xv);
      }
    }
  };
  if constexpr (ZERO) {
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) uo[k][i] = T(0);
  } else {
    load_vec<T, V, true>(u + wrapz(p0 - 1) * sz + c_off, V, uo[R - 1]);
#pragma unroll
    for (int k = 0; k <= D; ++k) load_vec<T, V, true>(u + wrapz(p0 + k) * sz + c_off, V, uo[k]);
  }
#pragma unroll
  for (int k = 0; k < D; ++k) load_vec<T, V, true>(rhs + wrapz(p0 + k) * sz + c_off, V, bb[k]);
  T cnew = T(0);  // the coarse value this wave fetched, on its way to the LDS ring
  if constexpr (SYNTH) {
    // planes z0 - 2 (even) and z0 - 1 (odd) of u for the first step read the coarse planes js0 - 2 .. js0: entries 1, 2, 0
    // of the ring (plane js0 - 2 sits where js0 + 1 belongs and is replaced below)
    const int js0 = (int)(z0 >> 1);
    if (row < kS2CRows) {
      cbuf[(1 * kS2CRows + row) * 64 + lane] = coarse[(int64_t)zcl(js0 - 2) * cplane + cfetch];
      cbuf[(2 * kS2CRows + row) * 64 + lane] = coarse[(int64_t)zcl(js0 - 1) * cplane + cfetch];
      cbuf[(0 * kS2CRows + row) * 64 + lane] = coarse[(int64_t)zcl(js0) * cplane + cfetch];
      cnew = coarse[(int64_t)zcl(js0 + 1) * cplane + cfetch];
    }
    __syncthreads();
    synth((int)z0 - 2, 1, 2, uo[R - 1]);
    synth((int)z0 - 1, 2, 0, uo[0]);
    __syncthreads();
    if (row < kS2CRows) cbuf[(1 * kS2CRows + row) * 64 + lane] = cnew;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) y1[R - 2][i] = y1[R - 1][i] = bb[R - 1][i] = T(0);
  const int slot = (row * 64 + lane) * V;
  const int slot_m = ((row == 0 ? 0 : row - 1) * 64 + lane) * V, slot_p = ((row == NW - 1 ? NW - 1 : row + 1) * 64 + lane) * V;
  if (HASY && !ZERO) {  // plane p0 of x for the first step's neighbours
    store_vec<T, V, true>(ubuf + slot, V, uo[0]);
    __syncthreads();
  }
  // whole groups of R steps (no exit inside the unrolled body: straight-line code); the steps beyond the chunk's last
  // plane load valid (wrapped) planes and store nothing
  const int nt = (int)(z1 - z0) + 2;
  int pw = wrapz(p0);                 // plane p, wrapped
  int pa = wrapz(p0 + 1 + D);         // plane p + 1 + D, wrapped
  int pb = wrapz(p0 + D);             // plane p + D, wrapped
  for (int t0 = 0; t0 < nt; t0 += R) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int t = t0 + k;
      // (1) the loads of this step, consumed D steps later
      // (uniform plane pointer + 32-bit lane offset: the scalar-base form of the load, no 64-bit lane arithmetic)
      if constexpr (!ZERO) load_vec<T, V, true>(u + pa * sz + c_off, V, uo[(k + 1 + D) % R]);
      if (row_s1) load_vec<T, V, true>(rhs + pb * sz + c_off, V, bb[(k + D) % R]);
      if constexpr (SYNTH) {
        // fine plane s = z0 + t (parity of k: z0 and t0 are even) reads the coarse planes js - 1, js (even) or js, js + 1
        // (odd), js = s / 2 in ring entry (k / 2) % 3.  An odd step fetches plane js + 2, the next (even) step puts it
        // into the entry of plane js - 1 -- last read by the even step before -- and the odd step after that reads it.
        if ((k & 1) && row < kS2CRows) cnew = coarse[(int64_t)zcl((int)(z0 >> 1) + (t >> 1) + 2) * cplane + cfetch];
        if (!(k & 1) && t > 0 && row < kS2CRows) cbuf[((((k >> 1) + 1) % 3) * kS2CRows + row) * 64 + lane] = cnew;
        const int s = (int)z0 + t;
        if (k & 1)
          synth(s, (k >> 1) % 3, ((k >> 1) + 1) % 3, uo[(k + 1) % R]);
        else
          synth(s, ((k >> 1) + 2) % 3, (k >> 1) % 3, uo[(k + 1) % R]);
      }
      // (2) plane p + 1 of x for the next step's y neighbours
      if (HASY && !ZERO) store_vec<T, V, true>(ubuf + ((k + 1) & 1) * (NW * 64 * V) + slot, V, uo[(k + 1) % R]);
      if (row_s1) {
        // (3) first sweep on plane p
        {
          const T (&qc)[V] = uo[k];
          T nm[V], np[V];
          if constexpr (ZERO) {
#pragma unroll
            for (int i = 0; i < V; ++i) nm[i] = np[i] = T(0);
          } else if (HASY) {
            const T* buf = ubuf + (k & 1) * (NW * 64 * V);
            load_vec<T, V, true>(buf + slot_m, V, nm);
            load_vec<T, V, true>(buf + slot_p, V, np);
          }
          T left = from_prev_lane(qc[V - 1]), right = from_next_lane(qc[0]);
          const bool zw = a.active[0] && (pw == 0 || pw == Zi - 1);
          if (yx_inner && !zw)
            jacobi_pack<T, V, false, MUL>(qc, uo[(k + R - 1) % R], uo[(k + 1) % R], nm, np, left, right, bb[k], pw, (int)y, x0, a,
                                          h, wd1[0][0], wd1[0][0], y1[k]);
          else
            jacobi_pack<T, V, true, MUL>(qc, uo[(k + R - 1) % R], uo[(k + 1) % R], nm, np, left, right, bb[k], pw, (int)y, x0, a,
                                         h, zw ? wd1[1][0] : wd1[0][0], zw ? wd1[1][1] : wd1[0][1], y1[k]);
        }
        if (HASY) store_vec<T, V, true>(ybuf + (k & 1) * (NW * 64 * V) + slot, V, y1[k]);
        // (4) second sweep on plane p - 1 (its y neighbours were published by the previous step)
        {
          const int z = p0 + t - 1;  // not wrapped: a plane of this chunk when t >= 2 and z < z1
          const T (&qc)[V] = y1[(k + R - 1) % R];
          T nm[V], np[V], out[V];
          if (HASY) {
            const T* buf = ybuf + ((k + 1) & 1) * (NW * 64 * V);
            load_vec<T, V, true>(buf + slot_m, V, nm);
            load_vec<T, V, true>(buf + slot_p, V, np);
          }
          T left = from_prev_lane(qc[V - 1]), right = from_next_lane(qc[0]);
          const bool zw = a.active[0] && (z == 0 || z == Zi - 1);
          if (yx_inner && !zw)
            jacobi_pack<T, V, false, MUL>(qc, y1[(k + R - 2) % R], y1[k], nm, np, left, right, bb[(k + R - 1) % R], z, (int)y, x0,
                                          a, h, wd2[0][0], wd2[0][0], out);
          else
            jacobi_pack<T, V, true, MUL>(qc, y1[(k + R - 2) % R], y1[k], nm, np, left, right, bb[(k + R - 1) % R], z, (int)y, x0,
                                         a, h, zw ? wd2[1][0] : wd2[0][0], zw ? wd2[1][1] : wd2[0][1], out);
          if (t >= 2 && z < (int)z1 && row_own && lane_own) {
            if (a.stream)
              store_vec<T, V, true, true>(uout + (int64_t)z * sz + c_off, V, out);
            else
              store_vec<T, V, true, false>(uout + (int64_t)z * sz + c_off, V, out);
          }
        }
      }
      if (HASY) __syncthreads();
      // (the steps stay apart: hoisting the loads of all R unrolled steps to the top costs the registers of R planes)
#ifndef S2_NO_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
      pw = pw + 1 == Zi ? 0 : pw + 1;
      pa = pa + 1 == Zi ? 0 : pa + 1;
      pb = pb + 1 == Zi ? 0 : pb + 1;
    }
  }
}

// ---- two sweeps of the VARIABLE-coefficient smoother (stencil_mg.hip: k_svar_smooth) in one pass ------------------------
// x' = x - w (A x - b) / c0 with A x = c0 x + c_-z x[-z] + c_-y x[-y] + c_+y x[+y] + c_-x x[-x] + c_+x x[+x] + c_+z x[+z]
// (this order: the +z term LAST, in the single-sweep kernel too).  The coefficient arrays are 7 of the 10 words a sweep
// moves, so they must be read ONCE for both sweeps: step t loads the coefficients of plane p, forms the first sweep
// y1[p] with them, and -- after the barrier that publishes y1[p] to the neighbouring rows -- everything of the SECOND
// sweep of plane p that does not need y1[p + 1]: the partial sum of A y1 without its +z term.  What the next step needs
// to finish that cell (partial sum, c_+z, c0, b) is four values; the coefficients themselves are dead when the step ends.
// Same mapping as k_poisson_jacobi2 (a wave per row of 64 packs, 16 rows per workgroup of which 12 are owned, lane shifts
// along x, LDS along y, registers along z); every window carries a halo pack per side, whose addresses wrap
// periodically like every index of this operator (wall rows carry zero coefficients instead).
// ZERO: the sweeps start from the zero vector -- x is not read (the ring holds zeros: the same products, the same bits).
template <typename T, int V, bool HASY, int DC, bool ZERO = false>
__global__ __launch_bounds__(HASY ? 64 * kS2Waves : 64) void k_svar_smooth2(const T* __restrict__ c,
                                                                            const T* __restrict__ x,
                                                                            const T* __restrict__ rhs,
                                                                            T* __restrict__ xout, Smooth2Args a,
                                                                            T omega1, T omega2) {
  constexpr int NW = HASY ? kS2Waves : 1;
  constexpr int HY = HASY ? 2 : 0;
  constexpr int R = kS2Ring, D = 1;  // x runs one step ahead of its use, the coefficient set DC steps
  __shared__ T ubuf[HASY && !ZERO ? 2 * NW * 64 * V : 1];
  __shared__ T ybuf[HASY ? 2 * NW * 64 * V : 1];
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  int zc, yt, xt;
  if (!unit_decode(a.usched, zc, yt, xt)) return;  // whole workgroup
  const int lane = threadIdx.x & 63, row = threadIdx.x >> 6;
  int64_t xp = (int64_t)xt * a.own_x - 1 + lane;
  const int own_here = (int64_t)(xt + 1) * a.own_x <= a.packs ? a.own_x : a.packs - xt * a.own_x;
  const bool lane_own = lane >= 1 && lane <= own_here;
  xp = ((xp % a.packs) + a.packs) % a.packs;
  int64_t y = HASY ? (int64_t)yt * kS2Own - HY + row : 0;
  const bool row_own = HASY ? (row >= HY && row < HY + kS2Own && y < Y) : true;
  const bool row_s1 = HASY ? (row >= 1 && row <= NW - 2) : true;
  y = ((y % Y) + Y) % Y;
  const int64_t z0 = (int64_t)zc * a.usched.ZC;
  const int64_t z1 = z0 + a.usched.ZC < Z ? z0 + a.usched.ZC : Z;
  const uint32_t c_off = (uint32_t)(y * sy + xp * V);
  const int Zi = (int)Z;
  auto wrapz = [Zi](int q) {
    q %= Zi;
    return q < 0 ? q + Zi : q;
  };
  const int p0 = (int)z0 - 1;
  const bool hz = a.active[0], hy = a.active[1];
  // the coefficient arrays (0, -z, +z, -y, +y, -x, +x as far as the axes exist) and the right-hand side
  const T* cz = c + (int64_t)a.slot[0] * a.size;
  const T* cy = c + (int64_t)a.slot[1] * a.size;
  const T* cx = c + (int64_t)a.slot[2] * a.size;
  T uo[R][V], y1[R][V];
  // coefficient sets in a ring of DC + 1: entry (k % (DC + 1)) for plane p0 + k -- [0] c0, [1] -z, [2] +z, [3] -y, [4] +y,
  // [5] -x, [6] +x, [7] b
  T cs[DC + 1][8][V];
  T hp[V], hzp[V], hc0[V], hb[V];  // second sweep of plane p - 1, waiting for y1[p]: partial sum, c_+z, c0, b
  auto load_set = [&](int pl, T (&s)[8][V]) {
    const int64_t off = (int64_t)pl * sz + c_off;
    load_vec<T, V, true>(c + off, V, s[0]);
    if (hz) {
      load_vec<T, V, true>(cz + off, V, s[1]);
      load_vec<T, V, true>(cz + a.size + off, V, s[2]);
    }
    if (hy) {
      load_vec<T, V, true>(cy + off, V, s[3]);
      load_vec<T, V, true>(cy + a.size + off, V, s[4]);
    }
    load_vec<T, V, true>(cx + off, V, s[5]);
    load_vec<T, V, true>(cx + a.size + off, V, s[6]);
    load_vec<T, V, true>(rhs + off, V, s[7]);
  };
  if constexpr (ZERO) {
#pragma unroll
    for (int k = 0; k < R; ++k)
#pragma unroll
      for (int i = 0; i < V; ++i) uo[k][i] = T(0);
  } else {
    load_vec<T, V, true>(x + wrapz(p0 - 1) * sz + c_off, V, uo[R - 1]);
#pragma unroll
    for (int k = 0; k <= D; ++k) load_vec<T, V, true>(x + wrapz(p0 + k) * sz + c_off, V, uo[k]);
  }
#pragma unroll
  for (int k = 0; k < DC; ++k) load_set(wrapz(p0 + k), cs[k % (DC + 1)]);
#pragma unroll
  for (int i = 0; i < V; ++i) y1[R - 1][i] = hp[i] = hzp[i] = hb[i] = T(0), hc0[i] = T(1);
  const int slot = (row * 64 + lane) * V;
  const int slot_m = ((row == 0 ? 0 : row - 1) * 64 + lane) * V, slot_p = ((row == NW - 1 ? NW - 1 : row + 1) * 64 + lane) * V;
  if (HASY && !ZERO) {
    store_vec<T, V, true>(ubuf + slot, V, uo[0]);
    __syncthreads();
  }
  const int nt = (int)(z1 - z0) + 2;
  int pw = wrapz(p0 + DC), pa = wrapz(p0 + 1 + D);  // the planes whose coefficients / x this step loads
  for (int t0 = 0; t0 < nt; t0 += R) {
#pragma unroll
    for (int k = 0; k < R; ++k) {
      const int t = t0 + k;
      T (&cc)[8][V] = cs[k % (DC + 1)];  // the coefficients of plane p
      // (1) loads: x of plane p + 2, the coefficient set of plane p + DC
      if constexpr (!ZERO) load_vec<T, V, true>(x + pa * sz + c_off, V, uo[(k + 1 + D) % R]);
      if (row_s1) load_set(pw, cs[(k + DC) % (DC + 1)]);
      // (2) plane p + 1 of x for the next step's y neighbours
      if (HASY && !ZERO) store_vec<T, V, true>(ubuf + ((k + 1) & 1) * (NW * 64 * V) + slot, V, uo[(k + 1) % R]);
      if (row_s1) {
        // (3) first sweep on plane p
        {
          const T (&qc)[V] = uo[k];
          T nm[V], np[V];
          if constexpr (ZERO) {
#pragma unroll
            for (int i = 0; i < V; ++i) nm[i] = np[i] = T(0);
          } else if (HASY) {
            const T* buf = ubuf + (k & 1) * (NW * 64 * V);
            load_vec<T, V, true>(buf + slot_m, V, nm);
            load_vec<T, V, true>(buf + slot_p, V, np);
          }
          const T left = from_prev_lane(qc[V - 1]), right = from_next_lane(qc[0]);
#pragma unroll
          for (int i = 0; i < V; ++i) {
            T acc = cc[0][i] * qc[i];
            if (hz) acc = acc + cc[1][i] * uo[(k + R - 1) % R][i];
            if (hy) {
              acc = acc + cc[3][i] * nm[i];
              acc = acc + cc[4][i] * np[i];
            }
            acc = acc + cc[5][i] * (i == 0 ? left : qc[i - 1 >= 0 ? i - 1 : 0]);
            acc = acc + cc[6][i] * (i == V - 1 ? right : qc[i + 1 < V ? i + 1 : i]);
            if (hz) acc = acc + cc[2][i] * uo[(k + 1) % R][i];
            y1[k][i] = qc[i] - omega1 * (acc - cc[7][i]) / cc[0][i];
          }
        }
        if (HASY) store_vec<T, V, true>(ybuf + (k & 1) * (NW * 64 * V) + slot, V, y1[k]);
        // (4) the second sweep of plane p - 1 gets its +z term
        {
          const int z = p0 + t - 1;
          const T (&qc)[V] = y1[(k + R - 1) % R];
          T out[V];
#pragma unroll
          for (int i = 0; i < V; ++i) {
            T acc = hp[i];
            if (hz) acc = acc + hzp[i] * y1[k][i];
            out[i] = qc[i] - omega2 * (acc - hb[i]) / hc0[i];
          }
          if (t >= 2 && z < (int)z1 && row_own && lane_own) {
            if (a.stream)
              store_vec<T, V, true, true>(xout + (int64_t)z * sz + c_off, V, out);
            else
              store_vec<T, V, true, false>(xout + (int64_t)z * sz + c_off, V, out);
          }
        }
      }
      if (HASY) __syncthreads();
      if (row_s1) {
        // (5) the second sweep of plane p without its +z term
        const T (&qc)[V] = y1[k];
        T nm[V], np[V];
        if (HASY) {
          const T* buf = ybuf + (k & 1) * (NW * 64 * V);
          load_vec<T, V, true>(buf + slot_m, V, nm);
          load_vec<T, V, true>(buf + slot_p, V, np);
        }
        const T left = from_prev_lane(qc[V - 1]), right = from_next_lane(qc[0]);
#pragma unroll
        for (int i = 0; i < V; ++i) {
          T acc = cc[0][i] * qc[i];
          if (hz) acc = acc + cc[1][i] * y1[(k + R - 1) % R][i];
          if (hy) {
            acc = acc + cc[3][i] * nm[i];
            acc = acc + cc[4][i] * np[i];
          }
          acc = acc + cc[5][i] * (i == 0 ? left : qc[i - 1 >= 0 ? i - 1 : 0]);
          acc = acc + cc[6][i] * (i == V - 1 ? right : qc[i + 1 < V ? i + 1 : i]);
          hp[i] = acc, hzp[i] = cc[2][i], hc0[i] = cc[0][i], hb[i] = cc[7][i];
        }
      }
#ifndef S2_NO_SCHED_BARRIER
      __builtin_amdgcn_sched_barrier(0);
#endif
      pw = pw + 1 == Zi ? 0 : pw + 1;
      pa = pa + 1 == Zi ? 0 : pa + 1;
    }
  }
}

// V: cells per lane; halo: packs of x-halo per side of a window that is not the whole row
template <typename T>
static int smooth2_args(Smooth2Args& a, const int64_t* shape, int ndim, const T* h2, T h[3], int zc_hint, int V,
                        int halo, const char* what) {
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
    set_error("%s: the last extent must be a multiple of %d (use single sweeps)", what, V);
    return ODIL_E_INVAL;
  }
  a.packs = (int)(a.n[2] / V);
  int64_t nxt;
  if (a.packs <= 64) {
    a.halo_x = 0, a.own_x = a.packs, nxt = 1;
  } else {
    a.halo_x = halo;
    nxt = (a.packs + 63 - 2 * halo) / (64 - 2 * halo);
    a.own_x = (int)((a.packs + nxt - 1) / nxt);
  }
  a.nxt = (int)nxt;
  const int64_t nyt = a.active[1] ? (a.n[1] + kS2Own - 1) / kS2Own : 1;
  if (a.n[0] * nyt * nxt >= ((int64_t)1 << 31) || a.n[1] * a.n[2] >= ((int64_t)1 << 31) || a.n[0] >= ((int64_t)1 << 30)) {
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
  constexpr int V = VecOf<T>::N;
  Smooth2Args a;
  T h[3];
  if (int e = smooth2_args<T>(a, shape, ndim, h2, h, zc_hint, V, 1, "poisson_jacobi2")) return e;
  if (!rhs || !uout || u == uout) {  // (u == NULL: the sweeps start from the zero vector)
    set_error("poisson_jacobi2: null pointer, or the sweeps in place (x_out must differ from x)");
    return ODIL_E_INVAL;
  }
  const int grid = unit_grid(a.usched);
  const H2<T> hh = make_h2<T>(h);
  const bool mul = hh.mul_ok[0] && hh.mul_ok[1] && hh.mul_ok[2];
  const T* none = nullptr;
#define ODIL_LAUNCH_J2(HASY, MUL, THREADS)                                                                              \
  do {                                                                                                                  \
    if (u)                                                                                                              \
      hipLaunchKernelGGL((k_poisson_jacobi2<T, V, HASY, MUL, false>), dim3(grid), dim3(THREADS), 0, (hipStream_t)stream, \
                         u, rhs, uout, none, a, hh, omega1, omega2);                                                    \
    else                                                                                                                \
      hipLaunchKernelGGL((k_poisson_jacobi2<T, V, HASY, MUL, false, kS2Ahead, true>), dim3(grid), dim3(THREADS), 0,      \
                         (hipStream_t)stream, u, rhs, uout, none, a, hh, omega1, omega2);                               \
  } while (0)
  if (a.active[1]) {
    if (mul)
      ODIL_LAUNCH_J2(true, true, 64 * kS2Waves);
    else
      ODIL_LAUNCH_J2(true, false, 64 * kS2Waves);
  } else {
    if (mul)
      ODIL_LAUNCH_J2(false, true, 64);
    else
      ODIL_LAUNCH_J2(false, false, 64);
  }
#undef ODIL_LAUNCH_J2
  return check_launch("k_poisson_jacobi2");
}

// x + P coarse, then two sweeps: the coarse-grid correction of a V-cycle and its post-smoothing in one pass
template <typename T>
static int poisson_jacobi2_synth(const T* coarse, const T* x, const T* rhs, T* xout, const int64_t* cshape, const T* h2,
                                 T omega1, T omega2, int zc_hint, void* stream) {
  if (!coarse || !x || !rhs || !xout || x == xout || !cshape) {
    set_error("poisson_jacobi2_synth: null pointer, or the sweeps in place");
    return ODIL_E_INVAL;
  }
  int64_t shape[3];
  for (int i = 0; i < 3; ++i) {
    if (cshape[i] < 2 || cshape[i] >= (1 << 29)) {
      set_error("poisson_jacobi2_synth: coarse extent %lld on axis %d", (long long)cshape[i], i);
      return ODIL_E_INVAL;
    }
    shape[i] = 2 * cshape[i];
  }
  Smooth2Args a;
  T h[3];
  if (zc_hint > 0) zc_hint += zc_hint & 1;  // (chunks of an even number of planes: the parity of a step is its plane's)
  if (int e = smooth2_args<T>(a, shape, 3, h2, h, zc_hint, 2, 2, "poisson_jacobi2_synth")) return e;
  if (a.usched.ZC & 1) {  // (only when the whole array is one odd chunk -- impossible: fine extents are even)
    set_error("poisson_jacobi2_synth: odd chunk length");
    return ODIL_E_INVAL;
  }
  const int grid = unit_grid(a.usched);
  const H2<T> hh = make_h2<T>(h);
  if (hh.mul_ok[0] && hh.mul_ok[1] && hh.mul_ok[2])
    hipLaunchKernelGGL((k_poisson_jacobi2<T, 2, true, true, true, 1>), dim3(grid), dim3(64 * kS2Waves), 0,
                       (hipStream_t)stream, x, rhs, xout, coarse, a, hh, omega1, omega2);
  else
    hipLaunchKernelGGL((k_poisson_jacobi2<T, 2, true, false, true, 1>), dim3(grid), dim3(64 * kS2Waves), 0,
                       (hipStream_t)stream, x, rhs, xout, coarse, a, hh, omega1, omega2);
  return check_launch("k_poisson_jacobi2<synth>");
}

template <typename T>
static int svar_smooth2(const T* coeffs, const T* x, const T* b, T* out, const int64_t* shape, int ndim, T omega1, T omega2,
                        int zc_hint, void* stream) {
  constexpr int V = 2;  // (two cells per lane in either precision: with four floats the coefficient set spills)
  Smooth2Args a;
  T h[3];
  const T ones[3] = {T(1), T(1), T(1)};
  if (int e = smooth2_args<T>(a, shape, ndim, ones, h, zc_hint, V, 1, "stencil_var_smooth2")) return e;
  if (!coeffs || !b || !out || x == out) {  // (x == NULL: the sweeps start from the zero vector)
    set_error("stencil_var_smooth2: null pointer or in-place sweeps");
    return ODIL_E_INVAL;
  }
  // every window carries its halo pack (the indices of this operator wrap periodically: also a window that is the whole row)
  a.halo_x = 1;
  a.nxt = (a.packs + 61) / 62;
  a.own_x = (a.packs + a.nxt - 1) / a.nxt;
  {
    const int64_t nyt = a.active[1] ? (a.n[1] + kS2Own - 1) / kS2Own : 1;
    const int zc = a.usched.ZC;
    a.usched = make_unit_sched(a.n[0], nyt, a.nxt, 1);
    a.usched.ZC = zc;
    a.usched.ZCH = (int)((a.n[0] + zc - 1) / zc);
    if (a.usched.axis == 1) {
      a.usched.per_xcd = a.usched.ZCH * a.usched.chunk * a.usched.XS;
    } else if (a.usched.ZCH >= kNumXcd) {
      a.usched.axis = 0, a.usched.chunk = 0;
      a.usched.per_xcd = (int)(((int64_t)a.usched.ZCH * a.usched.Y * a.usched.XS + kNumXcd - 1) / kNumXcd);
    } else {
      a.usched.axis = -1, a.usched.chunk = 0;
      a.usched.per_xcd = a.usched.ZCH * a.usched.Y * a.usched.XS;
    }
  }
  a.size = a.n[0] * a.n[1] * a.n[2];
  // array axis i is canonical axis map3[ndim - 1][i] (smooth2_args: the slowest real axis is the marched one); its two
  // coefficient arrays follow the diagonal in array-axis order
  {
    static const int map3[3][3] = {{2, 0, 0}, {0, 2, 0}, {0, 1, 2}};
    for (int d = 0; d < 3; ++d) a.slot[d] = 0;
    for (int i = 0; i < ndim; ++i) a.slot[map3[ndim - 1][i]] = 1 + 2 * i;
  }
  const int grid = unit_grid(a.usched);
  if (a.active[1] && x)
    hipLaunchKernelGGL((k_svar_smooth2<T, V, true, S2_SVAR_DC>), dim3(grid), dim3(64 * kS2Waves), 0, (hipStream_t)stream,
                       coeffs, x, b, out, a, omega1, omega2);
  else if (a.active[1])  // (x == NULL: from the zero vector)
    hipLaunchKernelGGL((k_svar_smooth2<T, V, true, S2_SVAR_DC, true>), dim3(grid), dim3(64 * kS2Waves), 0,
                       (hipStream_t)stream, coeffs, x, b, out, a, omega1, omega2);
  else if (x)
    hipLaunchKernelGGL((k_svar_smooth2<T, V, false, S2_SVAR_DC>), dim3(grid), dim3(64), 0, (hipStream_t)stream, coeffs, x, b,
                       out, a, omega1, omega2);
  else
    hipLaunchKernelGGL((k_svar_smooth2<T, V, false, S2_SVAR_DC, true>), dim3(grid), dim3(64), 0, (hipStream_t)stream, coeffs,
                       x, b, out, a, omega1, omega2);
  return check_launch("k_svar_smooth2");
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
int odil_poisson_jacobi2_synth_f64(const double* coarse, const double* x, const double* rhs, double* xout,
                                   const int64_t* cshape, const double* h2, double omega1, double omega2, int zc_hint,
                                   void* stream) {
  return poisson_jacobi2_synth<double>(coarse, x, rhs, xout, cshape, h2, omega1, omega2, zc_hint, stream);
}
int odil_poisson_jacobi2_synth_f32(const float* coarse, const float* x, const float* rhs, float* xout,
                                   const int64_t* cshape, const float* h2, float omega1, float omega2, int zc_hint,
                                   void* stream) {
  return poisson_jacobi2_synth<float>(coarse, x, rhs, xout, cshape, h2, omega1, omega2, zc_hint, stream);
}
int odil_stencil_var_smooth2_f64(const double* coeffs, const double* x, const double* b, double* out, const int64_t* shape,
                                 int ndim, double omega1, double omega2, int zc_hint, void* stream) {
  return svar_smooth2<double>(coeffs, x, b, out, shape, ndim, omega1, omega2, zc_hint, stream);
}
int odil_stencil_var_smooth2_f32(const float* coeffs, const float* x, const float* b, float* out, const int64_t* shape,
                                 int ndim, float omega1, float omega2, int zc_hint, void* stream) {
  return svar_smooth2<float>(coeffs, x, b, out, shape, ndim, omega1, omega2, zc_hint, stream);
}
}  // extern "C"
