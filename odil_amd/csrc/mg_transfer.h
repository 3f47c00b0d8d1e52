// Shared tap tables of the multigrid transfer kernels (mg_transfer.hip, mg_fast.hip).
#pragma once
#include "common.h"

namespace odil {

// ------------------------------------------------------------------------------------
// Taps of the 1-D prolongation for one fine index k on an axis of type `loc` with n
// coarse points.  In the reference's (r) order (core.py:675-687): r = 0 first.
// 'c': k = 2i+s reads padded index (i+s)+r-1, weights s ? (3,1) : (1,3), sum 4.
// 'n': k = 2i+s reads i (+ i+1 if s), weights 1, sum 1+s.   '.': identity.
// Out-of-range 'c' indices (-1, n) are ghosts: 2*u[clamp] - u[reflect], evaluated
// JOINTLY over all 'c' axes (core.py:640-643), hence the per-tap clamp/reflect pair.
// ------------------------------------------------------------------------------------
struct Taps {
  int cnt;
  int64_t cl[2], rf[2];
  int w[2];
  bool out[2];
  int sum;
};

__device__ __host__ inline Taps make_taps(int loc, int64_t k, int64_t n) {
  Taps t;
  t.cnt = 1;
  t.cl[0] = t.rf[0] = k;
  t.cl[1] = t.rf[1] = 0;
  t.w[0] = 1;
  t.w[1] = 0;
  t.out[0] = t.out[1] = false;
  t.sum = 1;
  if (loc == kCell) {
    int64_t i = k >> 1;
    int s = (int)(k & 1);
    t.cnt = 2;
    t.sum = 4;
    t.w[0] = s ? 3 : 1;
    t.w[1] = s ? 1 : 3;
    for (int r = 0; r < 2; ++r) {
      int64_t j = i + s + r - 1;
      bool o = j < 0 || j >= n;
      t.out[r] = o;
      t.cl[r] = j < 0 ? 0 : (j >= n ? n - 1 : j);
      t.rf[r] = j < 0 ? 1 : (j >= n ? n - 2 : j);
    }
  } else if (loc == kNode) {
    int64_t i = k >> 1;
    int s = (int)(k & 1);
    t.cl[0] = t.rf[0] = i;
    if (s) {
      t.cnt = 2;
      t.sum = 2;
      t.w[1] = 1;
      t.cl[1] = t.rf[1] = i + 1;
    }
  }
  return t;
}

struct InterpArgs {
  int64_t cn[4];  // coarse array shape (canonical 4-D)
  int64_t fn[4];  // fine array shape
  int loc[4];
  // slab decomposition: array axis 0 (canonical index cut_axis) is cut at its low / high end,
  // i.e. that end is an interior interface with ghost planes, not a physical boundary
  int cut_axis, cut_lo, cut_hi;
  // elements between consecutive LEADING indices of the coarse array when it is a view whose leading stride exceeds
  // its volume (a ghost-extended level array of the slab paths without its outer ghost planes); 0: contiguous
  int64_t coarse_ld;
  RowSched sched;
};

// ------------------------------------------------------------------------------------
// P^T in gather form.  With gpad = (tensor-product transpose onto the padded coarse
// grid), the joint ghost rule upad = 2*u[clamp] - u[reflect] gives
//   gc[J] = 2 * sum_{j: clamp(j)=J} gpad[j] - sum_{j: reflect(j)=J} gpad[j],
// and both sums stay separable: per axis the 1-D weights W(j,k) are summed over
//   C(J) = {J} + {-1 if J==0} + {n if J==n-1}   resp.   R(J) = {J} + {-1 if J==1} + {n if J==n-2}.
// 1-D weights: 'c' W(j,k) = {1,3,3,1}/4 at k-2j = -1..2; 'n' {1/2,1,1/2} at k-2j = -1..1.
// ------------------------------------------------------------------------------------
struct AdjTaps {
  int64_t k0;  // first fine index of the window
  int cnt;     // window length (<= 6)
  float wc[6], wr[6];
  bool special;  // wc != wr somewhere
};

__device__ inline float w_cell(int64_t j, int64_t k, int64_t F) {
  if (k < 0 || k >= F) return 0.f;
  int64_t d = k - 2 * j;
  return (d == 0 || d == 1) ? 0.75f : ((d == -1 || d == 2) ? 0.25f : 0.f);
}
// (the march / tile kernels index with int: a 64-bit compare is several instructions, and the wall paths evaluate
// dozens of these per step)
__device__ inline float w_cell(int j, int k, int F) {
  if (k < 0 || k >= F) return 0.f;
  const int d = k - 2 * j;
  return (d == 0 || d == 1) ? 0.75f : ((d == -1 || d == 2) ? 0.25f : 0.f);
}

__device__ inline AdjTaps make_adj_taps(int loc, int64_t J, int64_t n, int64_t F, bool cut_lo = false,
                                       bool cut_hi = false) {
  AdjTaps t;
  t.special = false;
  if (loc == kCell) {
    const bool c_lo = J == 0 && !cut_lo, c_hi = J == n - 1 && !cut_hi, r_lo = J == 1 && !cut_lo,
               r_hi = J == n - 2 && !cut_hi;
    t.special = c_lo || c_hi || r_lo || r_hi;
    if (t.special) {
      t.k0 = 2 * J - 2;
      t.cnt = 6;
    } else {
      t.k0 = 2 * J - 1;
      t.cnt = 4;
    }
    for (int i = 0; i < 6; ++i) {
      int64_t k = t.k0 + i;
      float w = i < t.cnt ? w_cell(J, k, F) : 0.f;
      float lo = w_cell((int64_t)-1, k, F), hi = w_cell(n, k, F);
      t.wc[i] = i < t.cnt ? w + (c_lo ? lo : 0.f) + (c_hi ? hi : 0.f) : 0.f;
      t.wr[i] = i < t.cnt ? w + (r_lo ? lo : 0.f) + (r_hi ? hi : 0.f) : 0.f;
    }
  } else if (loc == kNode) {
    t.k0 = 2 * J - 1;
    t.cnt = 3;
    for (int i = 0; i < 6; ++i) {
      int64_t k = t.k0 + i;
      float w = (i < 3 && k >= 0 && k < F) ? (i == 1 ? 1.f : 0.5f) : 0.f;
      t.wc[i] = t.wr[i] = w;
    }
  } else {
    t.k0 = J;
    t.cnt = 1;
    for (int i = 0; i < 6; ++i) t.wc[i] = t.wr[i] = i == 0 ? 1.f : 0.f;
  }
  return t;
}

// Fast paths (mg_fast.hip): last axis 'c', second-to-last 'c' or '.', any leading axes.
// Return 1 if they handled the call, 0 if the generic kernel must run, <0 on error.
template <typename T>
int interp_add_fast(const T* coarse, const T* add, T* fine, const InterpArgs& a, T cscale, T ascale,
                    hipStream_t stream);
template <typename T>
int interp_adj_fast(const T* gfine, T* gcoarse, T* gscaled, const InterpArgs& a, T scale, hipStream_t stream,
                    const AdamArgs<T>& ad);

// z-marching variants (mg_march.hip): exactly 'ccc' (3-D, all cell-centred).
template <typename T>
int interp_add_march(const T* coarse, const T* add, T* fine, const InterpArgs& a, T cscale, T ascale,
                     hipStream_t stream);
template <typename T>
int interp_adj_march(const T* gfine, T* gcoarse, T* gscaled, const InterpArgs& a, T scale, hipStream_t stream,
                     const AdamArgs<T>& ad);

}  // namespace odil
