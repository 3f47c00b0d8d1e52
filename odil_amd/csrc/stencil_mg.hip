// Geometric multigrid for ANY (2 d + 1)-point operator with variable coefficients on a cell-centred grid (d <= 3):
// the kernels behind gmg.StencilGMG, which solves the Newton system M delta = -r of a single-field operator whose
// Jacobian `Problem.linearize` delivers as per-shift coefficient arrays (reference src/odil/core.py:1113-1217; the
// reference hands M^T M to SuperLU / pyamg, src/odil/linsolver.py:17-26, 61-72 -- neither scales past ~1e6 unknowns).
//
// Row r of A:   (A u)[r] = c0[r] u[r] + sum_a ( cm_a[r] u[r - e_a] + cp_a[r] u[r + e_a] ),  indices wrap periodically
// (the roll of Context.field, core.py:962-963; wall rows simply carry a zero coefficient towards the wall).
// `coeffs` = the 2 d + 1 arrays one after another in the order (0, -e_0, +e_0, -e_1, +e_1, ...), each of `shape`.
//
//   k_svar_smooth<MODE>        x' = x - omega (A x - b) / c0  (damped Jacobi sweep)  |  r = b - A x
//   k_svar_residual_restrict   b_c = scale * sum over the 2^d children of (b - A x); sum (A x - b)^2 (deterministic)
//   k_svar_coarsen             the coarse-grid operator, again 2 d + 1 coefficient arrays (below)
//
// HBM-bound by construction: a sweep streams 2 d + 1 coefficient arrays + x + b in and x' out (10 words per cell in
// 3-D where the constant-coefficient Poisson sweep needs 3): that is the price of a general operator.
//
// Coarse operator (k_svar_coarsen).  Aggregates of 2^d cells, piecewise-constant Galerkin products R A P0 with R = mean
// of the children, applied to three parts of A that scale differently with the mesh width:
//   A = A2 + A1 + A0,  A1 = the matrix-antisymmetric part of every coupling pair (first derivatives: exact under R A P0),
//   A0 = the row sums of what is left on rows that have all their neighbours (reaction terms: exact under R A P0),
//   A2 = the rest (second derivatives and wall closures: R A P0 makes them twice too stiff -- the known factor of
//        cell-centred aggregation -- so they enter with 1/2);
// a coupling whose first-order part outweighs its second-order part (cell Peclet number > 1) has the latter raised to
// it, which keeps the coarse operators diagonally dominant (upwinding on the coarse grids).  With linear interpolation
// for the corrections (odil_interp_add) V(2,2) cycles contract by ~0.15 - 0.2 per cycle for constant and strongly
// varying diffusion (1 : 1000 jumps), reaction-diffusion and convection-diffusion alike (tests/test_stencil_gmg_*.py).
#include "common.h"

namespace odil {

struct SvarArgs {
  int64_t n[3];   // canonical (Z, Y, X); leading extents 1 for d < 3
  int64_t size;   // Z Y X
  int has[3];     // axis present (its two coefficient arrays exist)
  int slot[3];    // index of cm_a among the coefficient arrays (cp_a = slot + 1)
  int halve[3];   // (coarsening) two cells of this axis are merged; 0: the axis keeps its cells (semi-coarsening)
};

template <typename T>
struct Vec2 {
  typedef T type __attribute__((ext_vector_type(2)));
};

__device__ inline void svar_decode(int64_t i, const SvarArgs& a, int64_t (&id)[3]) {
  if (i < ((int64_t)1 << 31)) {
    uint32_t r = (uint32_t)i;
    const uint32_t X = (uint32_t)a.n[2], Y = (uint32_t)a.n[1];
    id[2] = r % X;
    r /= X;
    id[1] = r % Y;
    id[0] = r / Y;
  } else {
    id[2] = i % a.n[2];
    const int64_t r = i / a.n[2];
    id[1] = r % a.n[1];
    id[0] = r / a.n[1];
  }
}

// A x at cell (id) with flat index i
template <typename T>
__device__ inline T svar_apply(const T* __restrict__ c, const T* __restrict__ x, const SvarArgs& a, int64_t i,
                               const int64_t (&id)[3]) {
  // order of the terms: the diagonal, then per axis -e, +e -- except that the +e term of the SLOWEST existing axis comes
  // LAST, so that the one-pass pair of sweeps (smooth2.hip: k_svar_smooth2), which marches along that axis and must
  // finish a cell one plane late, forms bit for bit the same sum
  T acc = c[i] * x[i];
  const int64_t stride[3] = {a.n[1] * a.n[2], a.n[2], 1};
  const int first = a.has[0] ? 0 : (a.has[1] ? 1 : 2);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (!a.has[d]) continue;
    const int64_t n = a.n[d];
    const int64_t im = id[d] == 0 ? i + (n - 1) * stride[d] : i - stride[d];
    const int64_t ip = id[d] == n - 1 ? i - (n - 1) * stride[d] : i + stride[d];
    acc = acc + c[(int64_t)a.slot[d] * a.size + i] * x[im];
    if (d != first) acc = acc + c[(int64_t)(a.slot[d] + 1) * a.size + i] * x[ip];
  }
  {
    const int d = first;
    const int64_t n = a.n[d];
    const int64_t ip = id[d] == n - 1 ? i - (n - 1) * stride[d] : i + stride[d];
    acc = acc + c[(int64_t)(a.slot[d] + 1) * a.size + i] * x[ip];
  }
  return acc;
}

template <typename T, int MODE>
__global__ __launch_bounds__(kBlock) void k_svar_smooth(const T* __restrict__ c, const T* __restrict__ x,
                                                       const T* __restrict__ b, T* __restrict__ out, SvarArgs a,
                                                       T omega) {
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.size) return;
  int64_t id[3];
  svar_decode(i, a, id);
  if (MODE == 2) {  // the sweep from the ZERO vector: x is not read (A 0 = 0; finite coefficients: the same bits)
    out[i] = T(0) - omega * (T(0) - b[i]) / c[i];
    return;
  }
  const T ax = svar_apply<T>(c, x, a, i, id);
  if (MODE == 0) {
    out[i] = x[i] - omega * (ax - b[i]) / c[i];
  } else {
    out[i] = b[i] - ax;
  }
}

// one thread per COARSE cell; its two children along x are one 16-byte (f64) pack of every stream: the coefficient
// arrays, b and x are read once, fully coalesced (adjacent threads, adjacent packs); the y / z neighbour rows of x come
// as packs too (cache hits of the neighbouring threads' own rows), the two x neighbours outside the pack as scalars
template <typename T>
__global__ __launch_bounds__(kBlock) void k_svar_residual_restrict(const T* __restrict__ c, const T* __restrict__ x,
                                                                  const T* __restrict__ b, T* __restrict__ coarse,
                                                                  SvarArgs a, SvarArgs ca, T scale,
                                                                  double* __restrict__ partials, int64_t lz0,
                                                                  int64_t lz1) {
  typedef typename Vec2<T>::type T2;
  // a contiguous chunk of coarse cells per workgroup: the order of the partial sums does not depend on the grid
  const int64_t per = (ca.size + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  const int64_t hi = lo + per < ca.size ? lo + per : ca.size;
  const int64_t Z = a.n[0], Y = a.n[1], X = a.n[2];
  const int64_t sy = X, sz = Y * X;
  double local = 0.0;
  for (int64_t I = lo + threadIdx.x; I < hi; I += kBlock) {
    int64_t cid[3];
    svar_decode(I, ca, cid);
    T sum = T(0);
    const int k0 = a.has[0] ? 2 : 1, k1 = a.has[1] ? 2 : 1;
    const int64_t x0 = 2 * cid[2];
    const int64_t xl = x0 == 0 ? X - 1 : x0 - 1, xr = x0 + 2 >= X ? 0 : x0 + 2;
    for (int p = 0; p < k0; ++p)
      for (int q = 0; q < k1; ++q) {
        const int64_t z = a.has[0] ? 2 * cid[0] + p : 0, y = a.has[1] ? 2 * cid[1] + q : 0;
        const int64_t row = z * sz + y * sy, i = row + x0;
        const T2 xc = *(const T2*)(x + i);
        const T2 c0 = *(const T2*)(c + i);
        const T2 bb = *(const T2*)(b + i);
        const T2 cxm = *(const T2*)(c + (int64_t)a.slot[2] * a.size + i);
        const T2 cxp = *(const T2*)(c + (int64_t)(a.slot[2] + 1) * a.size + i);
        const T wl = x[row + xl], er = x[row + xr];
        T2 ax = c0 * xc;
        ax.x = ax.x + cxm.x * wl;
        ax.x = ax.x + cxp.x * xc.y;
        ax.y = ax.y + cxm.y * xc.x;
        ax.y = ax.y + cxp.y * er;
        if (a.has[1]) {
          const int64_t ym = (y == 0 ? Y - 1 : y - 1) * sy + z * sz + x0, yp = (y == Y - 1 ? 0 : y + 1) * sy + z * sz + x0;
          const T2 cm = *(const T2*)(c + (int64_t)a.slot[1] * a.size + i);
          const T2 cp = *(const T2*)(c + (int64_t)(a.slot[1] + 1) * a.size + i);
          ax = ax + cm * *(const T2*)(x + ym);
          ax = ax + cp * *(const T2*)(x + yp);
        }
        if (a.has[0]) {
          const int64_t zm = (z == 0 ? Z - 1 : z - 1) * sz + y * sy + x0, zp = (z == Z - 1 ? 0 : z + 1) * sz + y * sy + x0;
          const T2 cm = *(const T2*)(c + (int64_t)a.slot[0] * a.size + i);
          const T2 cp = *(const T2*)(c + (int64_t)(a.slot[0] + 1) * a.size + i);
          ax = ax + cm * *(const T2*)(x + zm);
          ax = ax + cp * *(const T2*)(x + zp);
        }
        const T2 r = bb - ax;
        if (z >= lz0 && z < lz1) local += (double)(r.x * r.x) + (double)(r.y * r.y);  // (slab form: own planes only)
        sum = sum + r.x;
        sum = sum + r.y;
      }
    coarse[I] = scale * sum;
  }
  const double total = block_sum(local);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

// one thread per coarse cell: the coarse coefficients (see the header of this file)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_svar_coarsen(const T* __restrict__ c, T* __restrict__ cc, SvarArgs a,
                                                        SvarArgs ca) {
  const int64_t I = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (I >= ca.size) return;
  int64_t cid[3];
  svar_decode(I, ca, cid);
  const int64_t stride[3] = {a.n[1] * a.n[2], a.n[2], 1};
  const int k0 = a.has[0] ? 2 : 1, k1 = a.has[1] ? 2 : 1, k2 = a.has[2] ? 2 : 1;
  // Galerkin sums over the children: diagonal of A2, A1 (always 0: antisymmetric internal couplings are ADDED below), A0;
  // face couplings of A2 and A1 per axis and side
  T d2 = T(0), d1 = T(0), d0 = T(0);
  T s2m[3] = {T(0), T(0), T(0)}, s2p[3] = {T(0), T(0), T(0)}, n1m[3] = {T(0), T(0), T(0)}, n1p[3] = {T(0), T(0), T(0)};
  for (int p = 0; p < k0; ++p)
    for (int q = 0; q < k1; ++q)
      for (int s = 0; s < k2; ++s) {
        const int bit[3] = {p, q, s};
        const int64_t id[3] = {a.has[0] ? 2 * cid[0] + p : 0, a.has[1] ? 2 * cid[1] + q : 0, a.has[2] ? 2 * cid[2] + s : 0};
        const int64_t i = (id[0] * a.n[1] + id[1]) * a.n[2] + id[2];
        const T c0 = c[i];
        T sym_sum = c0;   // row sum of the symmetric part
        bool complete = true;
        T row_sm[3], row_sp[3], row_nm[3], row_np[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          row_sm[d] = row_sp[d] = row_nm[d] = row_np[d] = T(0);
          if (!a.has[d]) continue;
          const int64_t n = a.n[d];
          const int64_t im = id[d] == 0 ? i + (n - 1) * stride[d] : i - stride[d];
          const int64_t ip = id[d] == n - 1 ? i - (n - 1) * stride[d] : i + stride[d];
          const T* cmA = c + (int64_t)a.slot[d] * a.size;
          const T* cpA = cmA + a.size;
          const T cm = cmA[i], cp = cpA[i];
          const T cmn = cmA[ip];  // the reverse coupling (i + e) -> i
          const T cpp = cpA[im];  // the reverse coupling (i - e) -> i
          const bool pair_p = cp != T(0) && cmn != T(0), pair_m = cm != T(0) && cpp != T(0);
          row_sp[d] = pair_p ? T(0.5) * (cp + cmn) : cp;
          row_np[d] = pair_p ? T(0.5) * (cp - cmn) : T(0);
          row_sm[d] = pair_m ? T(0.5) * (cm + cpp) : cm;
          row_nm[d] = pair_m ? T(0.5) * (cm - cpp) : T(0);
          complete = complete && cm != T(0) && cp != T(0);
          sym_sum = sym_sum + row_sm[d] + row_sp[d];
        }
        const T z = complete ? sym_sum : T(0);
        d0 = d0 + z;
        d2 = d2 + (c0 - z);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (!a.has[d]) continue;
          if (bit[d] == 0) {  // low child: its minus coupling crosses the aggregate's face, its plus coupling is internal
            s2m[d] = s2m[d] + row_sm[d];
            n1m[d] = n1m[d] + row_nm[d];
            d2 = d2 + row_sp[d];
            d1 = d1 + row_np[d];
          } else {
            s2p[d] = s2p[d] + row_sp[d];
            n1p[d] = n1p[d] + row_np[d];
            d2 = d2 + row_sm[d];
            d1 = d1 + row_nm[d];
          }
        }
      }
  const T w = T(1) / T(k0 * k1 * k2);   // R = mean of the children
  const T half = T(0.5);
  T diag2 = half * w * d2;
  const T diag10 = w * (d1 + d0);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (!a.has[d]) continue;
    T sm = half * w * s2m[d], sp = half * w * s2p[d];
    const T nm = w * n1m[d], np = w * n1p[d];
    // cell Peclet limiter: |first-order| <= |second-order| on every coupling, the difference taken from the diagonal
    const T sgn_d = diag2 > T(0) ? T(-1) : T(1);
    if (fabs((double)nm) > fabs((double)sm)) {
      const T sgn = sm != T(0) ? (sm > T(0) ? T(1) : T(-1)) : sgn_d;
      const T snew = sgn * (T)fabs((double)nm);
      diag2 = diag2 - (snew - sm);
      sm = snew;
    }
    if (fabs((double)np) > fabs((double)sp)) {
      const T sgn = sp != T(0) ? (sp > T(0) ? T(1) : T(-1)) : sgn_d;
      const T snew = sgn * (T)fabs((double)np);
      diag2 = diag2 - (snew - sp);
      sp = snew;
    }
    cc[(int64_t)ca.slot[d] * ca.size + I] = sm + nm;
    cc[(int64_t)(ca.slot[d] + 1) * ca.size + I] = sp + np;
  }
  cc[I] = diag2 + diag10;
}

// SEMI-coarsening: only the axes with a.halve are merged (the strongly coupled axes of an anisotropic operator; the
// others keep their cells until the couplings are of one size).  Same split of A; the factor 1/2 of the second-order part
// belongs to the axes whose spacing doubles, so A2 is taken apart axis by axis: the couplings of axis d with their share
// -(s_m + s_p) of the diagonal get f_d = 1/2 (merged) or 1 (kept); what is left of the diagonal on an incomplete row (a
// wall closure) goes with the incomplete axis -- 1/2 when one of those is merged.  With every axis merged this is
// k_svar_coarsen (which stays the kernel of that case: bit-identical hierarchies for grids of cubes).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_svar_coarsen_axes(const T* __restrict__ c, T* __restrict__ cc, SvarArgs a,
                                                             SvarArgs ca) {
  const int64_t I = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (I >= ca.size) return;
  int64_t cid[3];
  svar_decode(I, ca, cid);
  const int64_t stride[3] = {a.n[1] * a.n[2], a.n[2], 1};
  int kk[3];
  T f[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    kk[d] = (a.has[d] && a.halve[d]) ? 2 : 1;
    f[d] = a.halve[d] ? T(0.5) : T(1);
  }
  T dd2 = T(0), d1 = T(0), d0 = T(0);
  T s2m[3] = {T(0), T(0), T(0)}, s2p[3] = {T(0), T(0), T(0)}, n1m[3] = {T(0), T(0), T(0)}, n1p[3] = {T(0), T(0), T(0)};
  for (int p = 0; p < kk[0]; ++p)
    for (int q = 0; q < kk[1]; ++q)
      for (int s = 0; s < kk[2]; ++s) {
        const int bit[3] = {p, q, s};
        int64_t id[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) id[d] = !a.has[d] ? 0 : (a.halve[d] ? 2 * cid[d] + bit[d] : cid[d]);
        const int64_t i = (id[0] * a.n[1] + id[1]) * a.n[2] + id[2];
        const T c0 = c[i];
        T sym_sum = c0;
        bool complete = true, half_row = false;
        T row_sm[3], row_sp[3], row_nm[3], row_np[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          row_sm[d] = row_sp[d] = row_nm[d] = row_np[d] = T(0);
          if (!a.has[d]) continue;
          const int64_t n = a.n[d];
          const int64_t im = id[d] == 0 ? i + (n - 1) * stride[d] : i - stride[d];
          const int64_t ip = id[d] == n - 1 ? i - (n - 1) * stride[d] : i + stride[d];
          const T* cmA = c + (int64_t)a.slot[d] * a.size;
          const T* cpA = cmA + a.size;
          const T cm = cmA[i], cp = cpA[i];
          const T cmn = cmA[ip], cpp = cpA[im];  // the reverse couplings
          const bool pair_p = cp != T(0) && cmn != T(0), pair_m = cm != T(0) && cpp != T(0);
          row_sp[d] = pair_p ? T(0.5) * (cp + cmn) : cp;
          row_np[d] = pair_p ? T(0.5) * (cp - cmn) : T(0);
          row_sm[d] = pair_m ? T(0.5) * (cm + cpp) : cm;
          row_nm[d] = pair_m ? T(0.5) * (cm - cpp) : T(0);
          const bool both = cm != T(0) && cp != T(0);
          complete = complete && both;
          half_row = half_row || (!both && a.halve[d]);
          sym_sum = sym_sum + row_sm[d] + row_sp[d];
        }
        const T z = complete ? sym_sum : T(0);
        d0 = d0 + z;
        T rem = c0 - z;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (!a.has[d]) continue;
          const T share = -(row_sm[d] + row_sp[d]);
          rem = rem - share;
          dd2 = dd2 + f[d] * share;
          if (!a.halve[d]) {  // both couplings cross the aggregate's faces
            s2m[d] = s2m[d] + row_sm[d];
            s2p[d] = s2p[d] + row_sp[d];
            n1m[d] = n1m[d] + row_nm[d];
            n1p[d] = n1p[d] + row_np[d];
          } else if (bit[d] == 0) {
            s2m[d] = s2m[d] + row_sm[d];
            n1m[d] = n1m[d] + row_nm[d];
            dd2 = dd2 + f[d] * row_sp[d];
            d1 = d1 + row_np[d];
          } else {
            s2p[d] = s2p[d] + row_sp[d];
            n1p[d] = n1p[d] + row_np[d];
            dd2 = dd2 + f[d] * row_sm[d];
            d1 = d1 + row_nm[d];
          }
        }
        dd2 = dd2 + (half_row ? T(0.5) : T(1)) * rem;
      }
  const T w = T(1) / T(kk[0] * kk[1] * kk[2]);
  T diag2 = w * dd2;
  const T diag10 = w * (d1 + d0);
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (!a.has[d]) continue;
    T sm = f[d] * w * s2m[d], sp = f[d] * w * s2p[d];
    const T nm = w * n1m[d], np = w * n1p[d];
    const T sgn_d = diag2 > T(0) ? T(-1) : T(1);
    if (fabs((double)nm) > fabs((double)sm)) {
      const T sgn = sm != T(0) ? (sm > T(0) ? T(1) : T(-1)) : sgn_d;
      const T snew = sgn * (T)fabs((double)nm);
      diag2 = diag2 - (snew - sm);
      sm = snew;
    }
    if (fabs((double)np) > fabs((double)sp)) {
      const T sgn = sp != T(0) ? (sp > T(0) ? T(1) : T(-1)) : sgn_d;
      const T snew = sgn * (T)fabs((double)np);
      diag2 = diag2 - (snew - sp);
      sp = snew;
    }
    cc[(int64_t)ca.slot[d] * ca.size + I] = sm + nm;
    cc[(int64_t)(ca.slot[d] + 1) * ca.size + I] = sp + np;
  }
  cc[I] = diag2 + diag10;
}

// max |a - b| and max |b| over n entries (recognition of a known operator from its coefficient arrays: one pass over
// both instead of several elementwise launches); one partial pair per workgroup, combined by a second tiny launch
template <typename T>
__global__ __launch_bounds__(kBlock) void k_max_abs_diff(const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                                        double* __restrict__ partials) {
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double md = 0.0, mb = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
    const double vb = (double)b[i], vd = fabs((double)a[i] - vb);
    md = vd > md || vd != vd ? vd : md;  // (a NaN difference must not pass for a match)
    mb = fabs(vb) > mb ? fabs(vb) : mb;
  }
  __shared__ double sd[kBlock], sb[kBlock];
  sd[threadIdx.x] = md;
  sb[threadIdx.x] = mb;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const double od = sd[threadIdx.x + off], ob = sb[threadIdx.x + off];
      if (od > sd[threadIdx.x] || od != od) sd[threadIdx.x] = od;
      if (ob > sb[threadIdx.x]) sb[threadIdx.x] = ob;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partials[2 * blockIdx.x] = sd[0];
    partials[2 * blockIdx.x + 1] = sb[0];
  }
}

template <typename T>
__global__ __launch_bounds__(64) void k_max_abs_final(const double* __restrict__ partials, int count, T* __restrict__ out) {
  double md = 0.0, mb = 0.0;
  for (int i = threadIdx.x; i < count; i += 64) {  // one wavefront: strided maxima, then a butterfly over the lanes
    const double vd = partials[2 * i], vb = partials[2 * i + 1];
    md = vd > md || vd != vd ? vd : md;
    mb = vb > mb ? vb : mb;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double od = __shfl_xor(md, off, 64), ob = __shfl_xor(mb, off, 64);
    md = od > md || od != od ? od : md;
    mb = ob > mb ? ob : mb;
  }
  if (threadIdx.x == 0) {
    out[0] = (T)md;
    out[1] = (T)mb;
  }
}

template <typename T>
static int max_abs_diff(const T* a, const T* b, int64_t n, double* partials, T* out, void* stream) {
  if (!a || !b || !partials || !out || n < 1) {
    set_error("max_abs_diff: null pointer or n < 1");
    return ODIL_E_INVAL;
  }
  int grid = grid_for(n, kBlock * 8);
  if (2 * grid > kMaxPartials) grid = kMaxPartials / 2;
  hipLaunchKernelGGL((k_max_abs_diff<T>), dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a, b, n, partials);
  if (int e = check_launch("k_max_abs_diff")) return e;
  hipLaunchKernelGGL((k_max_abs_final<T>), dim3(1), dim3(64), 0, (hipStream_t)stream, partials, grid, out);
  return check_launch("k_max_abs_final");
}

// out[r] = max |a[r n .. (r + 1) n)| for the rows of one array in TWO launches (the set-up of the multigrid hierarchy asks for
// the largest coupling of every direction on every level: 2 d arrays per level, one read-back)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_max_abs_rows(const T* __restrict__ a, int64_t n, double* __restrict__ partials) {
  const T* row = a + (int64_t)blockIdx.y * n;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
  double m = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
    const double v = fabs((double)row[i]);
    m = v > m || v != v ? v : m;  // (a NaN stays visible)
  }
  __shared__ double sm[kBlock];
  sm[threadIdx.x] = m;
  __syncthreads();
  for (int off = kBlock / 2; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const double o = sm[threadIdx.x + off];
      if (o > sm[threadIdx.x] || o != o) sm[threadIdx.x] = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = sm[0];
}

template <typename T>
__global__ __launch_bounds__(64) void k_max_abs_rows_final(const double* __restrict__ partials, int count, T* __restrict__ out) {
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
static int max_abs_rows(const T* a, int nrows, int64_t n, double* partials, T* out, void* stream) {
  if (!a || !partials || !out || n < 1 || nrows < 1 || nrows > 64) {
    set_error("max_abs_rows: null pointer, n < 1 or nrows outside 1 .. 64");
    return ODIL_E_INVAL;
  }
  int grid = grid_for(n, kBlock * 8);
  if (grid > kMaxPartials / nrows) grid = kMaxPartials / nrows;
  hipLaunchKernelGGL((k_max_abs_rows<T>), dim3(grid, nrows), dim3(kBlock), 0, (hipStream_t)stream, a, n, partials);
  if (int e = check_launch("k_max_abs_rows")) return e;
  hipLaunchKernelGGL((k_max_abs_rows_final<T>), dim3(nrows), dim3(64), 0, (hipStream_t)stream, partials, grid, out);
  return check_launch("k_max_abs_rows_final");
}

static int svar_fill(SvarArgs& a, const int64_t* shape, int ndim, const char* what) {
  if (ndim < 1 || ndim > 3 || !shape) {
    set_error("%s: ndim %d (1..3 supported)", what, ndim);
    return ODIL_E_INVAL;
  }
  for (int d = 0; d < 3; ++d) {
    const int i = d - (3 - ndim);
    a.n[d] = i >= 0 ? shape[i] : 1;
    a.has[d] = i >= 0 ? 1 : 0;
    a.slot[d] = i >= 0 ? 1 + 2 * i : 0;
    a.halve[d] = a.has[d];
    if (a.n[d] < 1) {
      set_error("%s: empty extent", what);
      return ODIL_E_INVAL;
    }
  }
  a.size = a.n[0] * a.n[1] * a.n[2];
  if (a.size >= (int64_t)1 << 40) {
    set_error("%s: too many cells", what);
    return ODIL_E_INVAL;
  }
  return 0;
}

static int svar_coarse(const SvarArgs& a, SvarArgs& ca, const char* what) {
  ca = a;
  for (int d = 0; d < 3; ++d) {
    if (!a.has[d] || !a.halve[d]) continue;
    if (a.n[d] % 2 || a.n[d] < 2) {
      set_error("%s: extent %lld of axis %d is not even", what, (long long)a.n[d], d);
      return ODIL_E_INVAL;
    }
    ca.n[d] = a.n[d] / 2;
  }
  ca.size = ca.n[0] * ca.n[1] * ca.n[2];
  return 0;
}

template <typename T>
static int svar_smooth(const T* coeffs, const T* x, const T* b, T* out, const int64_t* shape, int ndim, T omega,
                       int mode, void* stream) {
  SvarArgs a;
  if (int e = svar_fill(a, shape, ndim, "stencil_var_smooth")) return e;
  if (!coeffs || (!x && mode != 0) || !b || !out || x == out) {  // (x == NULL with mode 0: the sweep starts from zero)
    set_error("stencil_var_smooth: null pointer or in-place sweep");
    return ODIL_E_INVAL;
  }
  const int64_t nb = (a.size + kBlock - 1) / kBlock;
  if (!x)
    hipLaunchKernelGGL((k_svar_smooth<T, 2>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, coeffs, x, b, out,
                       a, omega);
  else if (mode == 0)
    hipLaunchKernelGGL((k_svar_smooth<T, 0>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, coeffs, x, b, out,
                       a, omega);
  else
    hipLaunchKernelGGL((k_svar_smooth<T, 1>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, coeffs, x, b, out,
                       a, omega);
  return check_launch("k_svar_smooth");
}

template <typename T>
static int svar_residual_restrict(const T* coeffs, const T* x, const T* b, T* coarse, const int64_t* shape, int ndim,
                                  T scale, double* partials, T* loss, void* stream, int64_t z0 = 0, int64_t z1 = -1,
                                  double denom = 0.0) {
  SvarArgs a, ca;
  if (int e = svar_fill(a, shape, ndim, "stencil_var_residual_restrict")) return e;
  if (int e = svar_coarse(a, ca, "stencil_var_residual_restrict")) return e;
  if (!coeffs || !x || !b || !coarse || !partials) {  // (loss == NULL: no reduction launch)
    set_error("stencil_var_residual_restrict: null pointer");
    return ODIL_E_INVAL;
  }
  int64_t nb = (ca.size + kBlock - 1) / kBlock;
  if (nb > kMaxPartials) nb = kMaxPartials;  // (one partial sum per workgroup in the reduction workspace)
  if (z1 < 0) z0 = 0, z1 = a.n[0];
  if (denom <= 0.0) denom = (double)a.size;
  hipLaunchKernelGGL((k_svar_residual_restrict<T>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, coeffs, x,
                     b, coarse, a, ca, scale, partials, z0, z1);
  if (int e = check_launch("k_svar_residual_restrict")) return e;
  if (!loss) return 0;
  return launch_final_reduce<T>(partials, (int)nb, 0, 1, denom, loss, (hipStream_t)stream);
}

template <typename T>
static int svar_coarsen(const T* coeffs, T* coarse, const int64_t* shape, int ndim, const int* halve, void* stream) {
  SvarArgs a, ca;
  if (int e = svar_fill(a, shape, ndim, "stencil_var_coarsen")) return e;
  bool all = true, any = false;
  if (halve)
    for (int i = 0; i < ndim; ++i) {
      a.halve[i + 3 - ndim] = halve[i] != 0;
      all = all && halve[i] != 0;
      any = any || halve[i] != 0;
    }
  if (halve && !any) {
    set_error("stencil_var_coarsen: no axis to merge");
    return ODIL_E_INVAL;
  }
  if (int e = svar_coarse(a, ca, "stencil_var_coarsen")) return e;
  if (!coeffs || !coarse) {
    set_error("stencil_var_coarsen: null pointer");
    return ODIL_E_INVAL;
  }
  const int64_t nb = (ca.size + kBlock - 1) / kBlock;
  if (all)
    hipLaunchKernelGGL((k_svar_coarsen<T>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, coeffs, coarse, a, ca);
  else
    hipLaunchKernelGGL((k_svar_coarsen_axes<T>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, coeffs, coarse, a,
                       ca);
  return check_launch("k_svar_coarsen");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_stencil_var_smooth_f64(const double* coeffs, const double* x, const double* b, double* out, const int64_t* shape,
                                int ndim, double omega, int mode, void* stream) {
  return svar_smooth<double>(coeffs, x, b, out, shape, ndim, omega, mode, stream);
}
int odil_stencil_var_smooth_f32(const float* coeffs, const float* x, const float* b, float* out, const int64_t* shape,
                                int ndim, float omega, int mode, void* stream) {
  return svar_smooth<float>(coeffs, x, b, out, shape, ndim, omega, mode, stream);
}
int odil_stencil_var_residual_restrict_f64(const double* coeffs, const double* x, const double* b, double* coarse,
                                           const int64_t* shape, int ndim, double scale, double* partials, double* loss,
                                           void* stream) {
  return svar_residual_restrict<double>(coeffs, x, b, coarse, shape, ndim, scale, partials, loss, stream);
}
int odil_stencil_var_residual_restrict_f32(const float* coeffs, const float* x, const float* b, float* coarse,
                                           const int64_t* shape, int ndim, float scale, double* partials, float* loss,
                                           void* stream) {
  return svar_residual_restrict<float>(coeffs, x, b, coarse, shape, ndim, scale, partials, loss, stream);
}
int odil_stencil_var_residual_restrict_slab_f64(const double* coeffs, const double* x, const double* b, double* coarse,
                                                const int64_t* shape, int ndim, double scale, int64_t z0, int64_t z1,
                                                double denom, double* partials, double* loss, void* stream) {
  return svar_residual_restrict<double>(coeffs, x, b, coarse, shape, ndim, scale, partials, loss, stream, z0, z1, denom);
}
int odil_stencil_var_residual_restrict_slab_f32(const float* coeffs, const float* x, const float* b, float* coarse,
                                                const int64_t* shape, int ndim, float scale, int64_t z0, int64_t z1,
                                                double denom, double* partials, float* loss, void* stream) {
  return svar_residual_restrict<float>(coeffs, x, b, coarse, shape, ndim, scale, partials, loss, stream, z0, z1, denom);
}
int odil_max_abs_rows_f64(const double* a, int nrows, int64_t n, double* partials, double* out, void* stream) {
  return max_abs_rows<double>(a, nrows, n, partials, out, stream);
}
int odil_max_abs_rows_f32(const float* a, int nrows, int64_t n, double* partials, float* out, void* stream) {
  return max_abs_rows<float>(a, nrows, n, partials, out, stream);
}
int odil_max_abs_diff_f64(const double* a, const double* b, int64_t n, double* partials, double* out, void* stream) {
  return max_abs_diff<double>(a, b, n, partials, out, stream);
}
int odil_max_abs_diff_f32(const float* a, const float* b, int64_t n, double* partials, float* out, void* stream) {
  return max_abs_diff<float>(a, b, n, partials, out, stream);
}
int odil_stencil_var_coarsen_f64(const double* coeffs, double* coarse, const int64_t* shape, int ndim, void* stream) {
  return svar_coarsen<double>(coeffs, coarse, shape, ndim, nullptr, stream);
}
int odil_stencil_var_coarsen_f32(const float* coeffs, float* coarse, const int64_t* shape, int ndim, void* stream) {
  return svar_coarsen<float>(coeffs, coarse, shape, ndim, nullptr, stream);
}
int odil_stencil_var_coarsen_axes_f64(const double* coeffs, double* coarse, const int64_t* shape, int ndim, const int* halve,
                                      void* stream) {
  return svar_coarsen<double>(coeffs, coarse, shape, ndim, halve, stream);
}
int odil_stencil_var_coarsen_axes_f32(const float* coeffs, float* coarse, const int64_t* shape, int ndim, const int* halve,
                                      void* stream) {
  return svar_coarsen<float>(coeffs, coarse, shape, ndim, halve, stream);
}
}
