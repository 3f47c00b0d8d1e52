// Multigrid transfers of the ODIL hot path on gfx950:
//   P   : interp_to_finer   (reference src/odil/core.py:606-700, 'stack' summation order)
//   P^T : its transpose     (what autodiff yields, core.py:1100 / :1062)
//   R   : restrict_to_coarser (core.py:703-755)
// plus the level loops of Domain.multigrid_to_regular (core.py:245-263) and its adjoint.
//
// All three are HBM-bound streaming kernels: each workgroup owns row segments handed out
// by the XCD-aware schedule of common.h, lanes run along the contiguous last axis, every
// lane writes 2 consecutive fine values (16 B for f64).  The coarse operand is 1/2^d of
// the fine traffic and is served from L1/L2.
#include "mg_transfer.h"

namespace odil {

template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_add(const T* __restrict__ coarse, const T* __restrict__ add,
                                                       T* __restrict__ fine, InterpArgs a, T cscale, T ascale) {
  const int64_t cs2 = a.cn[3], cs1 = a.cn[2] * cs2, cs0 = a.cn[1] * cs1;
  const int64_t fs2 = a.fn[3], fs1 = a.fn[2] * fs2, fs0 = a.fn[1] * fs1;
  const int64_t npairs = (a.fn[3] + 1) / 2;
  RowIter it = sched_begin(a.sched);
  for (; it.t < it.count; it.t += it.step) {
    int64_t zz, y, xs;
    sched_decode(a.sched, it, zz, y, xs);
    const int64_t f0 = zz / a.fn[1], f1 = zz - f0 * a.fn[1], f2 = y;
    // Leading-axis taps: uniform over the workgroup.
    const Taps t0 = make_taps(a.loc[0], f0, a.cn[0]);
    const Taps t1 = make_taps(a.loc[1], f1, a.cn[1]);
    const Taps t2 = make_taps(a.loc[2], f2, a.cn[2]);
    const int64_t p = xs * kBlock + threadIdx.x;
    if (p >= npairs) continue;
    const int64_t k0 = 2 * p;
    const int nout = (k0 + 1 < a.fn[3]) ? 2 : 1;
    Taps tx[2];
    tx[0] = make_taps(a.loc[3], k0, a.cn[3]);
    tx[1] = make_taps(a.loc[3], k0 + 1 < a.fn[3] ? k0 + 1 : k0, a.cn[3]);
    T acc[2] = {T(0), T(0)};
    for (int r0 = 0; r0 < t0.cnt; ++r0)
      for (int r1 = 0; r1 < t1.cnt; ++r1)
        for (int r2 = 0; r2 < t2.cnt; ++r2) {
          const int wl = t0.w[r0] * t1.w[r1] * t2.w[r2];
          const bool ol = t0.out[r0] || t1.out[r1] || t2.out[r2];
          const int64_t bcl = t0.cl[r0] * cs0 + t1.cl[r1] * cs1 + t2.cl[r2] * cs2;
          const int64_t brf = t0.rf[r0] * cs0 + t1.rf[r1] * cs1 + t2.rf[r2] * cs2;
#pragma unroll
          for (int o = 0; o < 2; ++o) {
            if (o >= nout) break;
            for (int r3 = 0; r3 < tx[o].cnt; ++r3) {
              T val;
              if (ol || tx[o].out[r3]) {
                val = T(2) * (cscale * coarse[bcl + tx[o].cl[r3]]) - cscale * coarse[brf + tx[o].rf[r3]];
              } else {
                val = cscale * coarse[bcl + tx[o].cl[r3]];
              }
              acc[o] = acc[o] + T(wl * tx[o].w[r3]) * val;
            }
          }
        }
    const int sl = t0.sum * t1.sum * t2.sum;
    const int64_t fbase = f0 * fs0 + f1 * fs1 + f2 * fs2 + k0;
    for (int o = 0; o < nout; ++o) {
      T v = acc[o] / T(sl * tx[o].sum);
      if (add) v = ascale * add[fbase + o] + v;
      fine[fbase + o] = v;
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_adj(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                       T* __restrict__ gscaled, InterpArgs a, T scale) {
  const int64_t cs2 = a.cn[3], cs1 = a.cn[2] * cs2, cs0 = a.cn[1] * cs1;
  const int64_t fs2 = a.fn[3], fs1 = a.fn[2] * fs2, fs0 = a.fn[1] * fs1;
  RowIter it = sched_begin(a.sched);
  for (; it.t < it.count; it.t += it.step) {
    int64_t zz, y, xs;
    sched_decode(a.sched, it, zz, y, xs);
    const int64_t c0 = zz / a.cn[1], c1 = zz - c0 * a.cn[1], c2 = y;
    const AdjTaps t0 = make_adj_taps(a.loc[0], c0, a.cn[0], a.fn[0], a.cut_axis == 0 && a.cut_lo,
                                        a.cut_axis == 0 && a.cut_hi);
    const AdjTaps t1 = make_adj_taps(a.loc[1], c1, a.cn[1], a.fn[1], a.cut_axis == 1 && a.cut_lo,
                                        a.cut_axis == 1 && a.cut_hi);
    const AdjTaps t2 = make_adj_taps(a.loc[2], c2, a.cn[2], a.fn[2], a.cut_axis == 2 && a.cut_lo,
                                        a.cut_axis == 2 && a.cut_hi);
    const int64_t c3 = xs * kBlock + threadIdx.x;
    if (c3 >= a.cn[3]) continue;
    const AdjTaps t3 = make_adj_taps(a.loc[3], c3, a.cn[3], a.fn[3], a.cut_axis == 3 && a.cut_lo,
                                        a.cut_axis == 3 && a.cut_hi);
    const bool special = t0.special || t1.special || t2.special || t3.special;
    T sc = T(0), sr = T(0);
    for (int i0 = 0; i0 < t0.cnt; ++i0) {
      if (t0.wc[i0] == 0.f && t0.wr[i0] == 0.f) continue;
      for (int i1 = 0; i1 < t1.cnt; ++i1) {
        if (t1.wc[i1] == 0.f && t1.wr[i1] == 0.f) continue;
        for (int i2 = 0; i2 < t2.cnt; ++i2) {
          if (t2.wc[i2] == 0.f && t2.wr[i2] == 0.f) continue;
          const T wcl = T(t0.wc[i0] * t1.wc[i1] * t2.wc[i2]);
          const T wrl = T(t0.wr[i0] * t1.wr[i1] * t2.wr[i2]);
          const int64_t base = (t0.k0 + i0) * fs0 + (t1.k0 + i1) * fs1 + (t2.k0 + i2) * fs2;
          for (int i3 = 0; i3 < t3.cnt; ++i3) {
            if (t3.wc[i3] == 0.f && t3.wr[i3] == 0.f) continue;
            const T g = gfine[base + t3.k0 + i3];
            sc = sc + (wcl * T(t3.wc[i3])) * g;
            if (special) sr = sr + (wrl * T(t3.wr[i3])) * g;
          }
        }
      }
    }
    const T v = special ? T(2) * sc - sr : sc;
    const int64_t ci = c0 * cs0 + c1 * cs1 + c2 * cs2 + c3;
    gcoarse[ci] = v;
    if (gscaled) gscaled[ci] = scale * v;
  }
}

// ------------------------------------------------------------------------------------
// Restriction: 'c' [1,1]/2, 'n' [1,2,1]/4 on linearly extrapolated ghosts (joint rule
// 2*u[clamp]-u[reflect] over the 'n' axes, core.py:736-739), stride 2 VALID on every
// axis including '.' (backend.py:118-119).
// ------------------------------------------------------------------------------------
struct ResTaps {
  int cnt;
  int64_t cl[3], rf[3];
  bool out[3];
  float w[3];
};

__device__ inline ResTaps make_res_taps(int loc, int64_t j, int64_t n) {
  ResTaps t;
  if (loc == kCell) {
    t.cnt = 2;
    for (int i = 0; i < 2; ++i) {
      t.cl[i] = t.rf[i] = 2 * j + i;
      t.out[i] = false;
      t.w[i] = 0.5f;
    }
  } else if (loc == kNode) {
    t.cnt = 3;
    for (int i = 0; i < 3; ++i) {
      int64_t q = 2 * j - 1 + i;
      t.out[i] = q < 0 || q >= n;
      t.cl[i] = q < 0 ? 0 : (q >= n ? n - 1 : q);
      t.rf[i] = q < 0 ? 1 : (q >= n ? n - 2 : q);
      t.w[i] = i == 1 ? 0.5f : 0.25f;
    }
  } else {
    t.cnt = 1;
    t.cl[0] = t.rf[0] = 2 * j;
    t.out[0] = false;
    t.w[0] = 1.f;
  }
  return t;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_restrict(const T* __restrict__ fine, T* __restrict__ coarse,
                                                     InterpArgs a) {
  const int64_t cs2 = a.cn[3], cs1 = a.cn[2] * cs2, cs0 = a.cn[1] * cs1;
  const int64_t fs2 = a.fn[3], fs1 = a.fn[2] * fs2, fs0 = a.fn[1] * fs1;
  RowIter it = sched_begin(a.sched);
  for (; it.t < it.count; it.t += it.step) {
    int64_t zz, y, xs;
    sched_decode(a.sched, it, zz, y, xs);
    const int64_t c0 = zz / a.cn[1], c1 = zz - c0 * a.cn[1], c2 = y;
    const ResTaps t0 = make_res_taps(a.loc[0], c0, a.fn[0]);
    const ResTaps t1 = make_res_taps(a.loc[1], c1, a.fn[1]);
    const ResTaps t2 = make_res_taps(a.loc[2], c2, a.fn[2]);
    const int64_t c3 = xs * kBlock + threadIdx.x;
    if (c3 >= a.cn[3]) continue;
    const ResTaps t3 = make_res_taps(a.loc[3], c3, a.fn[3]);
    T acc = T(0);
    for (int i0 = 0; i0 < t0.cnt; ++i0)
      for (int i1 = 0; i1 < t1.cnt; ++i1)
        for (int i2 = 0; i2 < t2.cnt; ++i2)
          for (int i3 = 0; i3 < t3.cnt; ++i3) {
            const bool o = t0.out[i0] || t1.out[i1] || t2.out[i2] || t3.out[i3];
            const int64_t icl = t0.cl[i0] * fs0 + t1.cl[i1] * fs1 + t2.cl[i2] * fs2 + t3.cl[i3];
            T val = fine[icl];
            if (o) {
              const int64_t irf = t0.rf[i0] * fs0 + t1.rf[i1] * fs1 + t2.rf[i2] * fs2 + t3.rf[i3];
              val = T(2) * val - fine[irf];
            }
            acc = acc + T(t0.w[i0] * t1.w[i1] * t2.w[i2] * t3.w[i3]) * val;
          }
    coarse[c0 * cs0 + c1 * cs1 + c2 * cs2 + c3] = acc;
  }
}

// Restriction of all-cell layouts ('c', 'cc', 'ccc', 'cccc'): mean of the 2^d fine cells.  A lane reads
// its pair of fine x-neighbours as ONE pack per fine row (the generic kernel issues one-element loads at
// stride 2: 0.7 TB/s at 512^3); the sum runs in the generic kernel's order with its weights, so the
// results are bit-identical.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_restrict_cells(const T* __restrict__ fine, T* __restrict__ coarse,
                                                           InterpArgs a) {
  typedef T P2 __attribute__((ext_vector_type(2)));
  const int64_t fs2 = a.fn[3], fs1 = a.fn[2] * fs2, fs0 = a.fn[1] * fs1;
  const int64_t rows = a.cn[0] * a.cn[1] * a.cn[2];
  const int64_t per_row = (a.cn[3] + kBlock - 1) / kBlock;
  const int n0 = a.loc[0] == kCell ? 2 : 1, n1 = a.loc[1] == kCell ? 2 : 1, n2 = a.loc[2] == kCell ? 2 : 1;
  const T w = T((n0 == 2 ? 0.5f : 1.f) * (n1 == 2 ? 0.5f : 1.f) * (n2 == 2 ? 0.5f : 1.f) * 0.5f);
  for (int64_t u = blockIdx.x; u < rows * per_row; u += gridDim.x) {
    const int64_t row = u / per_row, c3 = (u - row * per_row) * kBlock + threadIdx.x;
    if (c3 >= a.cn[3]) continue;
    const int64_t c2 = row % a.cn[2], r1 = row / a.cn[2], c1 = r1 % a.cn[1], c0 = r1 / a.cn[1];
    const T* base = fine + (n0 * c0) * fs0 + (n1 * c1) * fs1 + (n2 * c2) * fs2 + 2 * c3;
    P2 v[2][2][2];
#pragma unroll
    for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
      for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          if (i0 < n0 && i1 < n1 && i2 < n2) v[i0][i1][i2] = *reinterpret_cast<const P2*>(base + i0 * fs0 + i1 * fs1 + i2 * fs2);
    T acc = T(0);
#pragma unroll
    for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
      for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          if (i0 < n0 && i1 < n1 && i2 < n2) {
            acc = acc + w * v[i0][i1][i2][0];
            acc = acc + w * v[i0][i1][i2][1];
          }
    coarse[row * a.cn[3] + c3] = acc;
  }
}

// ------------------------------------------------------------------------------------
// R^T (cotangent of the restriction; `poisson --mgloss` differentiates through R).  Gather
// form per FINE element K: per axis the coarse taps (j, w) with 2j + t (- 1 on 'n' axes) = q for
// q in C(K) / R(K), the clamp / reflect pre-images of K under the joint ghost rule on 'n' axes.
// ------------------------------------------------------------------------------------
struct RAdjTaps {
  int cnt;
  int64_t j[4];
  float wc[4], wr[4];
  bool special;
};

__device__ inline void radj_add(RAdjTaps& t, int64_t q, int64_t nout, bool in_c, bool in_r) {
  for (int k = 0; k < 3; ++k) {
    const int64_t num = q + 1 - k;
    if (num & 1) continue;
    const int64_t j = num / 2;
    if (num < 0 || j >= nout) continue;
    const float w = k == 1 ? 0.5f : 0.25f;
    int slot = -1;
    for (int i = 0; i < t.cnt; ++i)
      if (t.j[i] == j) slot = i;
    if (slot < 0) {
      slot = t.cnt++;
      t.j[slot] = j;
      t.wc[slot] = t.wr[slot] = 0.f;
    }
    if (in_c) t.wc[slot] += w;
    if (in_r) t.wr[slot] += w;
  }
}

__device__ inline RAdjTaps make_radj_taps(int loc, int64_t K, int64_t n, int64_t nout) {
  RAdjTaps t;
  t.cnt = 0;
  t.special = false;
  for (int i = 0; i < 4; ++i) {
    t.j[i] = 0;
    t.wc[i] = t.wr[i] = 0.f;
  }
  if (loc == kCell) {
    const int64_t j = K >> 1;
    if (j < nout) {
      t.cnt = 1;
      t.j[0] = j;
      t.wc[0] = t.wr[0] = 0.5f;
    }
  } else if (loc == kNone) {
    if ((K & 1) == 0 && K / 2 < nout) {
      t.cnt = 1;
      t.j[0] = K / 2;
      t.wc[0] = t.wr[0] = 1.f;
    }
  } else {
    t.special = K == 0 || K == 1 || K == n - 1 || K == n - 2;
    radj_add(t, K, nout, true, true);
    if (K == 0) radj_add(t, -1, nout, true, false);
    if (K == n - 1) radj_add(t, n, nout, true, false);
    if (K == 1) radj_add(t, -1, nout, false, true);
    if (K == n - 2) radj_add(t, n, nout, false, true);
  }
  return t;
}

// R^T of all-cell layouts: every fine cell receives 1 / 2^d of its coarse cell.  One coarse value per lane,
// written as 16 B packs to its 2^(d-1) fine rows (the generic gather kernel: per-element tap tables and
// 64-bit divisions, 0.17 TB/s at 512^3).  The single product w * g is what the generic sum reduces to.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_restrict_adj_cells(const T* __restrict__ gcoarse, T* __restrict__ gfine,
                                                               InterpArgs a) {
  typedef T P2 __attribute__((ext_vector_type(2)));
  const int64_t fs2 = a.fn[3], fs1 = a.fn[2] * fs2, fs0 = a.fn[1] * fs1;
  const int64_t rows = a.cn[0] * a.cn[1] * a.cn[2];
  const int64_t per_row = (a.cn[3] + kBlock - 1) / kBlock;
  const int n0 = a.loc[0] == kCell ? 2 : 1, n1 = a.loc[1] == kCell ? 2 : 1, n2 = a.loc[2] == kCell ? 2 : 1;
  const T w = T((n0 == 2 ? 0.5f : 1.f) * (n1 == 2 ? 0.5f : 1.f) * (n2 == 2 ? 0.5f : 1.f) * 0.5f);
  for (int64_t u = blockIdx.x; u < rows * per_row; u += gridDim.x) {
    const int64_t row = u / per_row, c3 = (u - row * per_row) * kBlock + threadIdx.x;
    if (c3 >= a.cn[3]) continue;
    const int64_t c2 = row % a.cn[2], r1 = row / a.cn[2], c1 = r1 % a.cn[1], c0 = r1 / a.cn[1];
    const T v = w * gcoarse[row * a.cn[3] + c3];
    P2 pk;
    pk[0] = v, pk[1] = v;
    T* base = gfine + (n0 * c0) * fs0 + (n1 * c1) * fs1 + (n2 * c2) * fs2 + 2 * c3;
#pragma unroll
    for (int i0 = 0; i0 < 2; ++i0)
#pragma unroll
      for (int i1 = 0; i1 < 2; ++i1)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          if (i0 < n0 && i1 < n1 && i2 < n2) *reinterpret_cast<P2*>(base + i0 * fs0 + i1 * fs1 + i2 * fs2) = pk;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_restrict_adj(const T* __restrict__ gcoarse, T* __restrict__ gfine,
                                                         InterpArgs a) {
  const int64_t cs2 = a.cn[3], cs1 = a.cn[2] * cs2, cs0 = a.cn[1] * cs1;
  const int64_t total = a.fn[0] * a.fn[1] * a.fn[2] * a.fn[3];
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += nthreads) {
    int64_t rem = i;
    int64_t K[4];
    for (int d = 3; d >= 0; --d) {
      K[d] = rem % a.fn[d];
      rem /= a.fn[d];
    }
    const RAdjTaps t0 = make_radj_taps(a.loc[0], K[0], a.fn[0], a.cn[0]);
    const RAdjTaps t1 = make_radj_taps(a.loc[1], K[1], a.fn[1], a.cn[1]);
    const RAdjTaps t2 = make_radj_taps(a.loc[2], K[2], a.fn[2], a.cn[2]);
    const RAdjTaps t3 = make_radj_taps(a.loc[3], K[3], a.fn[3], a.cn[3]);
    const bool special = t0.special || t1.special || t2.special || t3.special;
    T sc = T(0), sr = T(0);
    for (int i0 = 0; i0 < t0.cnt; ++i0)
      for (int i1 = 0; i1 < t1.cnt; ++i1)
        for (int i2 = 0; i2 < t2.cnt; ++i2)
          for (int i3 = 0; i3 < t3.cnt; ++i3) {
            const T g = gcoarse[t0.j[i0] * cs0 + t1.j[i1] * cs1 + t2.j[i2] * cs2 + t3.j[i3]];
            sc = sc + T(t0.wc[i0] * t1.wc[i1] * t2.wc[i2] * t3.wc[i3]) * g;
            sr = sr + T(t0.wr[i0] * t1.wr[i1] * t2.wr[i2] * t3.wr[i3]) * g;
          }
    gfine[i] = special ? T(2) * sc - sr : sc;
  }
}

// P^T when only the LEADING axis is refined and it is node-centred (loc 'n...'): the second half of the
// two-step transpose of the 4-D space-time layouts (ops.mg_synth_adj), a pure stream:
// gc[J] = gf[2J] + (gf[2J-1] + gf[2J+1]) / 2 over volumes of `vol` contiguous elements.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_interp_adj_lead_node(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                                T* __restrict__ gscaled, int64_t cn0, int64_t fn0,
                                                                int64_t vol, T scale, int64_t ld) {
  // ld: elements between leading indices of gcoarse (vol unless it is a view of a larger array; no gscaled then)
  const int64_t total = cn0 * vol;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += nthreads) {
    const int64_t J = i / vol, r = i - J * vol;
    const int64_t k = 2 * J;
    T sides = T(0);
    if (k - 1 >= 0) sides = sides + gfine[(k - 1) * vol + r];
    if (k + 1 < fn0) sides = sides + gfine[(k + 1) * vol + r];
    const T v = gfine[k * vol + r] + T(0.5) * sides;
    gcoarse[J * ld + r] = v;
    if (gscaled) gscaled[i] = scale * v;
  }
}

// The same with the coarse leading index in blockIdx.y and NV consecutive entries of a volume per thread (16 B
// accesses, no 64-bit division per entry): volumes that are a multiple of NV long, aligned arrays.  Same sums.
template <typename T, int NV>
__global__ __launch_bounds__(kBlock) void k_interp_adj_lead_node_wide(const T* __restrict__ gfine, T* __restrict__ gcoarse,
                                                                     T* __restrict__ gscaled, int64_t fn0, int64_t vol,
                                                                     T scale, int64_t ld) {
  typedef T VT __attribute__((ext_vector_type(NV)));
  const int64_t J = blockIdx.y, k = 2 * J;
  const int64_t r = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * NV;
  if (r >= vol) return;
  const VT mid = *reinterpret_cast<const VT*>(gfine + k * vol + r);
  VT lo, hi;
  const bool has_lo = k - 1 >= 0, has_hi = k + 1 < fn0;
  if (has_lo) lo = *reinterpret_cast<const VT*>(gfine + (k - 1) * vol + r);
  if (has_hi) hi = *reinterpret_cast<const VT*>(gfine + (k + 1) * vol + r);
  VT out, sc;
#pragma unroll
  for (int e = 0; e < NV; ++e) {
    T sides = T(0);
    if (has_lo) sides = sides + lo[e];
    if (has_hi) sides = sides + hi[e];
    out[e] = mid[e] + T(0.5) * sides;
    sc[e] = scale * out[e];
  }
  *reinterpret_cast<VT*>(gcoarse + J * ld + r) = out;
  if (gscaled) *reinterpret_cast<VT*>(gscaled + J * vol + r) = sc;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scale_copy(const T* __restrict__ x, T* __restrict__ y, int64_t n, T a) {
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) y[i] = a * x[i];
}

// ------------------------------------------------------------------------------------
// Host side
// ------------------------------------------------------------------------------------
static int fill_interp_args(InterpArgs& a, const int64_t* cshape, int ndim, const char* loc) {
  if (ndim < 1 || ndim > ODIL_MAX_NDIM) {
    set_error("ndim=%d out of range [1,%d]", ndim, ODIL_MAX_NDIM);
    return ODIL_E_INVAL;
  }
  if (parse_loc(loc, ndim, a.loc)) {
    set_error("invalid loc='%s' for ndim=%d", loc ? loc : "(null)", ndim);
    return ODIL_E_INVAL;
  }
  canon_shape(cshape, ndim, a.cn);
  a.cut_axis = -1;
  a.cut_lo = a.cut_hi = 0;
  a.coarse_ld = 0;
  for (int i = 0; i < 4; ++i) {
    if (a.cn[i] < 1 || (a.loc[i] != kNone && a.cn[i] < 2)) {
      set_error("coarse extent %lld on a refined axis must be >= 2", (long long)a.cn[i]);
      return ODIL_E_INVAL;
    }
    a.fn[i] = a.loc[i] == kCell ? 2 * a.cn[i] : (a.loc[i] == kNode ? 2 * a.cn[i] - 1 : a.cn[i]);
  }
  return 0;
}

template <typename T>
static int interp_add(const T* coarse, const T* add, T* fine, const int64_t* cshape, int ndim, const char* loc,
                      T cscale, T ascale, void* stream, int64_t coarse_ld = 0) {
  InterpArgs a;
  if (int e = fill_interp_args(a, cshape, ndim, loc)) return e;
  if (!coarse || !fine) {
    set_error("interp_add: null pointer");
    return ODIL_E_INVAL;
  }
  if (coarse_ld) {
    if (coarse_ld < a.cn[1] * a.cn[2] * a.cn[3]) {
      set_error("interp_add: leading stride %lld of the coarse array is less than its volume", (long long)coarse_ld);
      return ODIL_E_INVAL;
    }
    a.coarse_ld = coarse_ld;
  }
  if (int r = interp_add_march<T>(coarse, add, fine, a, cscale, ascale, (hipStream_t)stream)) return r < 0 ? r : 0;
  if (coarse_ld) return 1;  // only the marching kernels of the 4-D layouts take a strided operand: nothing launched
  if (int r = interp_add_fast<T>(coarse, add, fine, a, cscale, ascale, (hipStream_t)stream)) return r < 0 ? r : 0;
  const int64_t npairs = (a.fn[3] + 1) / 2;
  a.sched = make_sched(a.fn[0] * a.fn[1], a.fn[2], (npairs + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_interp_add<T>, dim3(sched_grid(a.sched)), dim3(kBlock), 0, (hipStream_t)stream, coarse, add,
                     fine, a, cscale, ascale);
  return check_launch("k_interp_add");
}

template <typename T>
static int interp_adj(const T* gfine, T* gcoarse, T* gscaled, const int64_t* cshape, int ndim, const char* loc,
                      T scale, void* stream, int cut_lo = 0, int cut_hi = 0,
                      AdamArgs<T> ad = AdamArgs<T>{nullptr, nullptr, nullptr, T(0), T(0), T(0), T(0)},
                      int64_t coarse_ld = 0) {
  InterpArgs a;
  if (int e = fill_interp_args(a, cshape, ndim, loc)) return e;
  if (coarse_ld) {
    if (coarse_ld < a.cn[1] * a.cn[2] * a.cn[3] || gscaled || ad.x || cut_lo || cut_hi) {
      set_error("interp_adj: a strided result takes no scaled copy, update or cut, and a leading stride >= its volume");
      return ODIL_E_INVAL;
    }
    a.coarse_ld = coarse_ld;
  }
  if (cut_lo || cut_hi) {
    a.cut_axis = 4 - ndim;
    a.cut_lo = cut_lo;
    a.cut_hi = cut_hi;
  }
  if (!gfine || !gcoarse) {
    set_error("interp_adj: null pointer");
    return ODIL_E_INVAL;
  }
  if (a.loc[0] == kNode && a.loc[1] == kNone && a.loc[2] == kNone && a.loc[3] == kNone && a.cut_axis < 0) {
    const int64_t vol = a.cn[1] * a.cn[2] * a.cn[3];
    constexpr int NV = 16 / sizeof(T);
    const int64_t ld = coarse_ld ? coarse_ld : vol;
    const auto al = [](const void* q) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % 16 == 0; };
    if (vol % NV == 0 && ld % NV == 0 && a.cn[0] <= 65535 && al(gfine) && al(gcoarse) && al(gscaled) && vol >= 4096) {
      const dim3 grid((unsigned)((vol / NV + kBlock - 1) / kBlock), (unsigned)a.cn[0]);
      hipLaunchKernelGGL((k_interp_adj_lead_node_wide<T, NV>), grid, dim3(kBlock), 0, (hipStream_t)stream, gfine, gcoarse,
                         gscaled, a.fn[0], vol, scale, ld);
    } else {
      hipLaunchKernelGGL(k_interp_adj_lead_node<T>, dim3(grid_flat(a.cn[0] * vol, kBlock)), dim3(kBlock), 0,
                         (hipStream_t)stream, gfine, gcoarse, gscaled, a.cn[0], a.fn[0], vol, scale, ld);
    }
    if (int e = check_launch("k_interp_adj_lead_node")) return e;
    if (ad.x)
      return adam_launch<T>(ad.x, ad.m, ad.v, gscaled ? gscaled : gcoarse, prod4(a.cn), ad.alpha, ad.omb1, ad.omb2,
                            ad.eps, (hipStream_t)stream, ad.alpha_dev);
    return 0;
  }
  if (int r = interp_adj_march<T>(gfine, gcoarse, gscaled, a, scale, (hipStream_t)stream, ad)) return r < 0 ? r : 0;
  if (coarse_ld) return 1;  // only the marching kernels of the 4-D layouts write a strided result: nothing launched
  if (int r = interp_adj_fast<T>(gfine, gcoarse, gscaled, a, scale, (hipStream_t)stream, ad)) return r < 0 ? r : 0;
  a.sched = make_sched(a.cn[0] * a.cn[1], a.cn[2], (a.cn[3] + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_interp_adj<T>, dim3(sched_grid(a.sched)), dim3(kBlock), 0, (hipStream_t)stream, gfine,
                     gcoarse, gscaled, a, scale);
  if (int e = check_launch("k_interp_adj")) return e;
  if (ad.x)  // the generic kernel does not fuse the update: plain launch on this level
    return adam_launch<T>(ad.x, ad.m, ad.v, gscaled ? gscaled : gcoarse, prod4(a.cn), ad.alpha, ad.omb1, ad.omb2, ad.eps,
                          (hipStream_t)stream, ad.alpha_dev);
  return 0;
}

template <typename T>
static int restrict_(const T* fine, T* coarse, const int64_t* fshape, int ndim, const char* loc, void* stream) {
  InterpArgs a;
  if (ndim < 1 || ndim > ODIL_MAX_NDIM || parse_loc(loc, ndim, a.loc)) {
    set_error("restrict: invalid ndim=%d / loc", ndim);
    return ODIL_E_INVAL;
  }
  canon_shape(fshape, ndim, a.fn);
  for (int i = 0; i < 4; ++i) {
    const int64_t n = a.fn[i];
    // VALID stride-2 output sizes on the padded array: 'c' (n-2)/2+1, 'n' (n+2-3)/2+1, '.' (n-1)/2+1.
    a.cn[i] = a.loc[i] == kCell ? (n - 2) / 2 + 1 : (n - 1) / 2 + 1;
    if (a.loc[i] == kCell && n < 2) {
      set_error("restrict: extent %lld too small", (long long)n);
      return ODIL_E_INVAL;
    }
  }
  if (!fine || !coarse) {
    set_error("restrict: null pointer");
    return ODIL_E_INVAL;
  }
  bool cells = a.loc[3] == kCell && a.fn[3] % 2 == 0 && (reinterpret_cast<uintptr_t>(fine) % (2 * sizeof(T))) == 0;
  for (int i = 0; i < 3; ++i) cells = cells && ((a.loc[i] == kCell && a.fn[i] % 2 == 0) || (a.fn[i] == 1 && a.loc[i] != kNode));
  if (cells) {
    const int64_t units = a.cn[0] * a.cn[1] * a.cn[2] * ((a.cn[3] + kBlock - 1) / kBlock);
    const int grid = (int)(units < 16 * kGridCap ? units : 16 * kGridCap);
    hipLaunchKernelGGL(k_restrict_cells<T>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, fine, coarse, a);
    return check_launch("k_restrict_cells");
  }
  a.sched = make_sched(a.cn[0] * a.cn[1], a.cn[2], (a.cn[3] + kBlock - 1) / kBlock);
  hipLaunchKernelGGL(k_restrict<T>, dim3(sched_grid(a.sched)), dim3(kBlock), 0, (hipStream_t)stream, fine, coarse, a);
  return check_launch("k_restrict");
}

template <typename T>
static int restrict_adj(const T* gcoarse, T* gfine, const int64_t* fshape, int ndim, const char* loc, void* stream) {
  InterpArgs a;
  if (ndim < 1 || ndim > ODIL_MAX_NDIM || parse_loc(loc, ndim, a.loc)) {
    set_error("restrict_adj: invalid ndim=%d / loc", ndim);
    return ODIL_E_INVAL;
  }
  canon_shape(fshape, ndim, a.fn);
  a.cut_axis = -1;
  a.cut_lo = a.cut_hi = 0;
  for (int i = 0; i < 4; ++i) {
    const int64_t n = a.fn[i];
    a.cn[i] = a.loc[i] == kCell ? (n - 2) / 2 + 1 : (n - 1) / 2 + 1;
    if (a.loc[i] == kCell && n < 2) {
      set_error("restrict_adj: extent %lld too small", (long long)n);
      return ODIL_E_INVAL;
    }
  }
  if (!gcoarse || !gfine) {
    set_error("restrict_adj: null pointer");
    return ODIL_E_INVAL;
  }
  bool cells = a.loc[3] == kCell && a.fn[3] % 2 == 0 && (reinterpret_cast<uintptr_t>(gfine) % (2 * sizeof(T))) == 0;
  for (int i = 0; i < 3; ++i) cells = cells && ((a.loc[i] == kCell && a.fn[i] % 2 == 0) || (a.fn[i] == 1 && a.loc[i] != kNode));
  if (cells) {
    const int64_t units = a.cn[0] * a.cn[1] * a.cn[2] * ((a.cn[3] + kBlock - 1) / kBlock);
    const int grid = (int)(units < 16 * kGridCap ? units : 16 * kGridCap);
    hipLaunchKernelGGL(k_restrict_adj_cells<T>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, gcoarse, gfine, a);
    return check_launch("k_restrict_adj_cells");
  }
  hipLaunchKernelGGL(k_restrict_adj<T>, dim3(grid_flat(prod4(a.fn), kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     gcoarse, gfine, a);
  return check_launch("k_restrict_adj");
}

static int check_levels(const int64_t* shapes, int nlvl, int ndim, const char* loc) {
  if (nlvl < 1 || nlvl > ODIL_MAX_LEVELS || !shapes) {
    set_error("nlvl=%d out of range [1,%d]", nlvl, ODIL_MAX_LEVELS);
    return ODIL_E_INVAL;
  }
  int l4[4];
  if (ndim < 1 || ndim > ODIL_MAX_NDIM || parse_loc(loc, ndim, l4)) {
    set_error("invalid ndim=%d / loc", ndim);
    return ODIL_E_INVAL;
  }
  for (int l = 1; l < nlvl; ++l)
    for (int d = 0; d < ndim; ++d) {
      const int64_t c = shapes[l * ndim + d], f = shapes[(l - 1) * ndim + d];
      const char t = loc[d];
      const int64_t expect = t == 'c' ? 2 * c : (t == 'n' ? 2 * c - 1 : c);
      if (f != expect) {
        set_error("level %d axis %d: fine extent %lld does not refine coarse extent %lld (loc '%c')", l, d,
                  (long long)f, (long long)c, t);
        return ODIL_E_INVAL;
      }
    }
  return 0;
}

// res_{L-1} = f_{L-1} w_{L-1};  res_l = f_l w_l + P(res_{l+1})   (core.py:258-262)
template <typename T>
static int mg_synth(const T* const* terms, const T* factors, T* const* work, T* u, const int64_t* shapes, int nlvl,
                    int ndim, const char* loc, void* stream) {
  if (int e = check_levels(shapes, nlvl, ndim, loc)) return e;
  if (!terms || !u) {
    set_error("mg_synth: null pointer");
    return ODIL_E_INVAL;
  }
  if (nlvl == 1) {  // u = f_0 * w_0
    int64_t n0 = 1;
    for (int d = 0; d < ndim; ++d) n0 *= shapes[d];
    hipLaunchKernelGGL(k_scale_copy<T>, dim3(grid_flat(n0, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, terms[0],
                       u, n0, factors ? factors[0] : T(1));
    return check_launch("k_scale_copy");
  }
  const T* coarse = terms[nlvl - 1];
  T cscale = factors ? factors[nlvl - 1] : T(1);
  for (int l = nlvl - 2; l >= 0; --l) {
    T* out = l == 0 ? u : (work ? work[l] : nullptr);
    if (!out) {
      set_error("mg_synth: work[%d] is null", l);
      return ODIL_E_INVAL;
    }
    if (int e = interp_add<T>(coarse, terms[l], out, shapes + (l + 1) * ndim, ndim, loc, cscale,
                              factors ? factors[l] : T(1), stream))
      return e;
    coarse = out;
    cscale = T(1);
  }
  return 0;
}

// g'_0 = gu; g'_{l+1} = P^T g'_l; grads_l = f_l g'_l.
template <typename T>
static int mg_synth_adj(const T* gu, T* const* grads, const T* factors, T* const* work, const int64_t* shapes,
                        int nlvl, int ndim, const char* loc, void* stream, T* const* ax = nullptr,
                        T* const* am = nullptr, T* const* av = nullptr, T alpha = T(0), T omb1 = T(0), T omb2 = T(0),
                        T eps = T(0), const T* alpha_dev = nullptr) {
  if (int e = check_levels(shapes, nlvl, ndim, loc)) return e;
  if (!gu || !grads) {
    set_error("mg_synth_adj: null pointer");
    return ODIL_E_INVAL;
  }
  // Level 0: grads[0] = f_0 * gu (grads[0] may alias gu when f_0 == 1).
  const T f0 = factors ? factors[0] : T(1);
  if (grads[0] && (grads[0] != gu || f0 != T(1))) {
    const int64_t* s0 = shapes;
    int64_t n0 = 1;
    for (int d = 0; d < ndim; ++d) n0 *= s0[d];
    hipLaunchKernelGGL(k_scale_copy<T>, dim3(grid_flat(n0, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, gu,
                       grads[0], n0, f0);
    if (int e = check_launch("k_scale_copy")) return e;
  }
  const T* gfine = gu;  // the chain itself is unscaled
  for (int l = 1; l < nlvl; ++l) {
    const T f = factors ? factors[l] : T(1);
    T* unscaled = grads[l];
    T* scaled = nullptr;
    if (f != T(1)) {
      // later levels read the unscaled cotangent: keep it in work[l]
      if (!work || !work[l]) {
        set_error("mg_synth_adj: work[%d] needed for factor != 1", l);
        return ODIL_E_INVAL;
      }
      unscaled = work[l];
      scaled = grads[l];
    }
    AdamArgs<T> ad{nullptr, nullptr, nullptr, alpha, omb1, omb2, eps, alpha_dev};
    if (ax && ax[l]) {
      ad.x = ax[l];
      ad.m = am[l];
      ad.v = av[l];
    }
    if (int e = interp_adj<T>(gfine, unscaled, scaled, shapes + l * ndim, ndim, loc, f, stream, 0, 0, ad)) return e;
    gfine = unscaled;
  }
  return 0;
}

}  // namespace odil

using namespace odil;

extern "C" {

int odil_interp_add_f64(const double* coarse, const double* add, double* fine, const int64_t* cshape, int ndim,
                        const char* loc, double coarse_scale, double add_scale, void* stream) {
  return interp_add<double>(coarse, add, fine, cshape, ndim, loc, coarse_scale, add_scale, stream);
}
int odil_interp_add_f32(const float* coarse, const float* add, float* fine, const int64_t* cshape, int ndim,
                        const char* loc, float coarse_scale, float add_scale, void* stream) {
  return interp_add<float>(coarse, add, fine, cshape, ndim, loc, coarse_scale, add_scale, stream);
}
int odil_interp_add_ld_f64(const double* coarse, int64_t coarse_ld, const double* add, double* fine, const int64_t* cshape,
                           int ndim, const char* loc, double coarse_scale, double add_scale, void* stream) {
  return interp_add<double>(coarse, add, fine, cshape, ndim, loc, coarse_scale, add_scale, stream, coarse_ld);
}
int odil_interp_add_ld_f32(const float* coarse, int64_t coarse_ld, const float* add, float* fine, const int64_t* cshape,
                           int ndim, const char* loc, float coarse_scale, float add_scale, void* stream) {
  return interp_add<float>(coarse, add, fine, cshape, ndim, loc, coarse_scale, add_scale, stream, coarse_ld);
}
int odil_interp_adj_ld_f64(const double* gfine, double* gcoarse, int64_t gcoarse_ld, const int64_t* cshape, int ndim,
                           const char* loc, void* stream) {
  return interp_adj<double>(gfine, gcoarse, nullptr, cshape, ndim, loc, 1.0, stream, 0, 0,
                            AdamArgs<double>{nullptr, nullptr, nullptr, 0, 0, 0, 0}, gcoarse_ld);
}
int odil_interp_adj_ld_f32(const float* gfine, float* gcoarse, int64_t gcoarse_ld, const int64_t* cshape, int ndim,
                           const char* loc, void* stream) {
  return interp_adj<float>(gfine, gcoarse, nullptr, cshape, ndim, loc, 1.0f, stream, 0, 0,
                           AdamArgs<float>{nullptr, nullptr, nullptr, 0, 0, 0, 0}, gcoarse_ld);
}
int odil_interp_adj_f64(const double* gfine, double* gcoarse, double* gscaled, const int64_t* cshape, int ndim,
                        const char* loc, double scale, void* stream) {
  return interp_adj<double>(gfine, gcoarse, gscaled, cshape, ndim, loc, scale, stream);
}
int odil_interp_adj_f32(const float* gfine, float* gcoarse, float* gscaled, const int64_t* cshape, int ndim,
                        const char* loc, float scale, void* stream) {
  return interp_adj<float>(gfine, gcoarse, gscaled, cshape, ndim, loc, scale, stream);
}
int odil_interp_adj_cut_f64(const double* gfine, double* gcoarse, double* gscaled, const int64_t* cshape, int ndim,
                            const char* loc, double scale, int cut_lo, int cut_hi, void* stream) {
  return interp_adj<double>(gfine, gcoarse, gscaled, cshape, ndim, loc, scale, stream, cut_lo, cut_hi);
}
int odil_interp_adj_cut_f32(const float* gfine, float* gcoarse, float* gscaled, const int64_t* cshape, int ndim,
                            const char* loc, float scale, int cut_lo, int cut_hi, void* stream) {
  return interp_adj<float>(gfine, gcoarse, gscaled, cshape, ndim, loc, scale, stream, cut_lo, cut_hi);
}
int odil_interp_adj_cut_adam_f64(const double* gfine, double* gcoarse, const int64_t* cshape, int ndim, const char* loc,
                                 int cut_lo, int cut_hi, double* x, double* m, double* v, double alpha,
                                 double one_minus_b1, double one_minus_b2, double eps, const double* alpha_dev, void* stream) {
  return interp_adj<double>(gfine, gcoarse, nullptr, cshape, ndim, loc, 1.0, stream, cut_lo, cut_hi,
                            AdamArgs<double>{x, m, v, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev});
}
int odil_interp_adj_cut_adam_f32(const float* gfine, float* gcoarse, const int64_t* cshape, int ndim, const char* loc,
                                 int cut_lo, int cut_hi, float* x, float* m, float* v, float alpha,
                                 float one_minus_b1, float one_minus_b2, float eps, const float* alpha_dev, void* stream) {
  return interp_adj<float>(gfine, gcoarse, nullptr, cshape, ndim, loc, 1.0f, stream, cut_lo, cut_hi,
                           AdamArgs<float>{x, m, v, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev});
}
int odil_restrict_f64(const double* fine, double* coarse, const int64_t* fshape, int ndim, const char* loc,
                      void* stream) {
  return restrict_<double>(fine, coarse, fshape, ndim, loc, stream);
}
int odil_restrict_f32(const float* fine, float* coarse, const int64_t* fshape, int ndim, const char* loc,
                      void* stream) {
  return restrict_<float>(fine, coarse, fshape, ndim, loc, stream);
}
int odil_restrict_adj_f64(const double* gcoarse, double* gfine, const int64_t* fshape, int ndim, const char* loc,
                          void* stream) {
  return restrict_adj<double>(gcoarse, gfine, fshape, ndim, loc, stream);
}
int odil_restrict_adj_f32(const float* gcoarse, float* gfine, const int64_t* fshape, int ndim, const char* loc,
                          void* stream) {
  return restrict_adj<float>(gcoarse, gfine, fshape, ndim, loc, stream);
}
int odil_mg_synth_f64(const double* const* terms, const double* factors, double* const* work, double* u,
                      const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream) {
  return mg_synth<double>(terms, factors, work, u, shapes, nlvl, ndim, loc, stream);
}
int odil_mg_synth_f32(const float* const* terms, const float* factors, float* const* work, float* u,
                      const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream) {
  return mg_synth<float>(terms, factors, work, u, shapes, nlvl, ndim, loc, stream);
}
int odil_mg_synth_adj_f64(const double* gu, double* const* grads, const double* factors, double* const* work,
                          const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream) {
  return mg_synth_adj<double>(gu, grads, factors, work, shapes, nlvl, ndim, loc, stream);
}
int odil_mg_synth_adj_f32(const float* gu, float* const* grads, const float* factors, float* const* work,
                          const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream) {
  return mg_synth_adj<float>(gu, grads, factors, work, shapes, nlvl, ndim, loc, stream);
}
int odil_mg_synth_adj_adam_f64(const double* gu, double* const* grads, const double* factors, double* const* work,
                               const int64_t* shapes, int nlvl, int ndim, const char* loc, double* const* x,
                               double* const* m, double* const* v, double alpha, double one_minus_b1,
                               double one_minus_b2, double eps, const double* alpha_dev, void* stream) {
  return mg_synth_adj<double>(gu, grads, factors, work, shapes, nlvl, ndim, loc, stream, x, m, v, alpha, one_minus_b1,
                              one_minus_b2, eps, alpha_dev);
}
int odil_mg_synth_adj_adam_f32(const float* gu, float* const* grads, const float* factors, float* const* work,
                               const int64_t* shapes, int nlvl, int ndim, const char* loc, float* const* x,
                               float* const* m, float* const* v, float alpha, float one_minus_b1, float one_minus_b2,
                               float eps, const float* alpha_dev, void* stream) {
  return mg_synth_adj<float>(gu, grads, factors, work, shapes, nlvl, ndim, loc, stream, x, m, v, alpha, one_minus_b1,
                             one_minus_b2, eps, alpha_dev);
}

}  // extern "C"
