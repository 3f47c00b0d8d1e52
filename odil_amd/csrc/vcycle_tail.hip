// The COARSE TAIL of a multigrid V-cycle in ONE launch (gmg.PoissonGMG / gmg.StencilGMG: the Newton solve; the reference
// hands that system to SuperLU / pyamg, src/odil/linsolver.py:17-26, 61-72).
//
// Levels of a few thousand cells cost a launch each for every sweep, residual, restriction and prolongation -- ~7
// dependent launches per level and cycle, 5 - 12 us apiece on a GPU that finishes the work itself in under a microsecond:
// at 512^3 the levels from 16^3 down are 28 of a cycle's 57 launches and ~0.3 of its 3.3 ms.  Here ONE workgroup walks
// the whole tail -- pre-smoothing, residual + restriction, the dense solve on the coarsest grid, prolongation +
// post-smoothing, level after level -- with a workgroup barrier where a launch boundary used to be.
//
// The operator of every level is given by coefficient arrays (the layout of stencil_mg.hip: (2 d + 1) arrays per level,
// order 0, -e_0, +e_0, ...; neighbours wrap periodically, wall rows carry a zero towards the wall), which serves both
// cycles: the constant-coefficient Poisson hierarchy passes odil_poisson_jac_coeffs of its rediscretised levels.
//   sweep         x' = x - w (A x - b) / c0
//   coarse rhs    b_c = mean over the merged children of (b - A x)
//   correction    x += P x_c, P = the multigrid decomposition's prolongation with its joint ghost rule
//                 (reference core.py:606-700, 640-643) on the merged axes, identity on the others
//   coarsest      x = inv b, the (pseudo-)inverse the host built once
// fmg: the nested-iteration start instead of one cycle -- b is restricted to every level, the coarsest problem solved,
// every finer level starts one cycle from the interpolated solution of the level below.
#include "common.h"

namespace odil {

constexpr int kTailMaxLev = 8;
constexpr int kTailThreads = 1024;

struct TailLevel {
  int n[3];       // canonical (Z, Y, X), leading extents 1 for d < 3
  float rn[3];    // 1 / n
  int halve[3];   // the transition to the next coarser level merges pairs of cells of this axis
  int size;
  int64_t c;      // offset of this level's coefficient arrays in `coeffs`
  int64_t x, t, b;  // offsets of its iterate / spare / right-hand side in `work`
};

template <typename T>
struct TailArgs {
  int nlev, ndim;
  int has[3], slot[3];
  TailLevel lv[kTailMaxLev];
  int npre, npost, fmg, ninv;
  T wpre[4], wpost[4];
};

// i / n for 0 <= i < 2^22 without the ~40-instruction integer divide: the float quotient is off by at most one
__device__ __forceinline__ int tail_div(int i, int n, float rn) {
  int q = (int)((float)i * rn);
  q = q * n > i ? q - 1 : q;
  q = (q + 1) * n <= i ? q + 1 : q;
  return q;
}

__device__ __forceinline__ void tail_decode(int i, const TailLevel& L, int (&id)[3]) {
  const int r = tail_div(i, L.n[2], L.rn[2]);
  id[2] = i - r * L.n[2];
  id[0] = tail_div(r, L.n[1], L.rn[1]);
  id[1] = r - id[0] * L.n[1];
}

// (A x)[i]; zero: x is the zero vector (the start of every coarse-grid correction) and is not read
template <typename T>
__device__ __forceinline__ T tail_apply(const T* __restrict__ c, const T* __restrict__ x, bool zero, const TailLevel& L,
                                        const TailArgs<T>& a, int i, const int (&id)[3]) {
  if (zero) return T(0);
  T acc = c[i] * x[i];
  const int stride[3] = {L.n[1] * L.n[2], L.n[2], 1};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    if (!a.has[d]) continue;
    const int n = L.n[d];
    const int im = id[d] == 0 ? i + (n - 1) * stride[d] : i - stride[d];
    const int ip = id[d] == n - 1 ? i - (n - 1) * stride[d] : i + stride[d];
    acc = acc + c[(int64_t)a.slot[d] * L.size + i] * x[im];
    acc = acc + c[(int64_t)(a.slot[d] + 1) * L.size + i] * x[ip];
  }
  return acc;
}

template <typename T>
__device__ void tail_sweep(const T* __restrict__ c, const T* __restrict__ x, bool zero, const T* __restrict__ b,
                           T* __restrict__ out, T w, const TailLevel& L, const TailArgs<T>& a) {
#pragma unroll 4
  for (int i = threadIdx.x; i < L.size; i += kTailThreads) {
    int id[3];
    tail_decode(i, L, id);
    const T ax = tail_apply<T>(c, x, zero, L, a, i, id);
    out[i] = (zero ? T(0) : x[i]) - w * (ax - b[i]) / c[i];
  }
  __syncthreads();
}

// bc = mean over the merged children of (b - A x)  (x == nullptr: of b itself -- the restriction of a right-hand side)
template <typename T>
__device__ void tail_restrict(const T* __restrict__ c, const T* __restrict__ x, bool zero, const T* __restrict__ b,
                              T* __restrict__ bc, const TailLevel& L, const TailLevel& C, const TailArgs<T>& a, bool residual) {
  const int k0 = L.halve[0] ? 2 : 1, k1 = L.halve[1] ? 2 : 1, k2 = L.halve[2] ? 2 : 1;
  const T scale = T(1) / T(k0 * k1 * k2);
  for (int I = threadIdx.x; I < C.size; I += kTailThreads) {
    int cid[3];
    tail_decode(I, C, cid);
    T sum = T(0);
    for (int p = 0; p < k0; ++p)
      for (int q = 0; q < k1; ++q)
        for (int r = 0; r < k2; ++r) {
          int id[3] = {L.halve[0] ? 2 * cid[0] + p : cid[0], L.halve[1] ? 2 * cid[1] + q : cid[1],
                       L.halve[2] ? 2 * cid[2] + r : cid[2]};
          const int i = (id[0] * L.n[1] + id[1]) * L.n[2] + id[2];
          T v = b[i];
          if (residual) v = v - tail_apply<T>(c, x, zero, L, a, i, id);
          sum = sum + v;
        }
    bc[I] = scale * sum;
  }
  __syncthreads();
}

// P xc (add: + xin) on level L from level C; result to out (not xin)
template <typename T>
__device__ void tail_prolong(const T* __restrict__ xc, const T* __restrict__ xin, bool add, T* __restrict__ out,
                             const TailLevel& L, const TailLevel& C) {
  for (int i = threadIdx.x; i < L.size; i += kTailThreads) {
    int id[3];
    tail_decode(i, L, id);
    // per axis: two taps (clamped, reflected index, weight, beyond a wall) on a merged axis, one on the others
    int cl[3][2], rf[3][2], w[3][2], cnt[3];
    bool out_[3][2];
    int den = 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (L.halve[d]) {
        const int j0 = id[d] >> 1, s = id[d] & 1, n = C.n[d];
        cnt[d] = 2;
        den *= 4;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int j = j0 + s + r - 1;
          out_[d][r] = j < 0 || j >= n;
          cl[d][r] = j < 0 ? 0 : (j >= n ? n - 1 : j);
          rf[d][r] = j < 0 ? (n > 1 ? 1 : 0) : (j >= n ? (n > 1 ? n - 2 : 0) : j);
          w[d][r] = (s == r) ? 1 : 3;
        }
      } else {
        cnt[d] = 1;
        cl[d][0] = rf[d][0] = id[d];
        w[d][0] = 1;
        out_[d][0] = false;
        cl[d][1] = rf[d][1] = 0, w[d][1] = 0, out_[d][1] = false;
      }
    }
    T s = T(0);
#pragma unroll
    for (int r0 = 0; r0 < 2; ++r0)
#pragma unroll
      for (int r1 = 0; r1 < 2; ++r1)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
          if (r0 >= cnt[0] || r1 >= cnt[1] || r2 >= cnt[2]) continue;
          const bool o = out_[0][r0] || out_[1][r1] || out_[2][r2];
          const T vc = xc[(cl[0][r0] * C.n[1] + cl[1][r1]) * C.n[2] + cl[2][r2]];
          T v = vc;
          if (o) v = T(2) * vc - xc[(rf[0][r0] * C.n[1] + rf[1][r1]) * C.n[2] + rf[2][r2]];
          s = s + T(w[0][r0] * w[1][r1] * w[2][r2]) * v;
        }
    s = s * (T(1) / T(den));
    out[i] = add ? xin[i] + s : s;
  }
  __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kTailThreads) void k_vcycle_tail(const T* __restrict__ coeffs, const T* xin,
                                                              const T* __restrict__ btop, T* xout, T* work,
                                                              const T* __restrict__ inv, TailArgs<T> a) {
  const int last = a.nlev - 1;
  // where the iterate of a level lives (uniform bookkeeping: every thread holds the same pointers)
  // (in LDS, every thread writing the same values: indexed by a run-time level, private arrays would live in scratch
  // memory and every phase would start with a chain of scratch loads)
  __shared__ const T* cur[kTailMaxLev];
  __shared__ T* bufA[kTailMaxLev];
  __shared__ T* bufB[kTailMaxLev];
  __shared__ const T* rhs[kTailMaxLev];
  __shared__ T* rhsw[kTailMaxLev];
  for (int l = 0; l < kTailMaxLev; ++l) {
    bufA[l] = l == 0 ? xout : work + a.lv[l].x;
    bufB[l] = work + a.lv[l].t;
    rhsw[l] = work + a.lv[l].b;
    rhs[l] = l == 0 ? btop : rhsw[l];
    cur[l] = nullptr;
  }
  auto other = [&](int l) -> T* { return cur[l] == bufA[l] ? bufB[l] : bufA[l]; };
  auto coarsest = [&]() {
    const TailLevel& L = a.lv[last];
    T* out = bufA[last];
    const T* b = rhs[last];
    for (int r = threadIdx.x; r < L.size; r += kTailThreads) {
      T s = T(0);
      for (int j = 0; j < L.size; ++j) s = s + inv[(int64_t)r * L.size + j] * b[j];
      out[r] = s;
    }
    __syncthreads();
    cur[last] = out;
  };
  // one V-cycle on level `top` of the tail; cur[top] == nullptr: from the zero iterate
  auto vcycle = [&](int top) {
    for (int l = top; l < last; ++l) {
      const TailLevel& L = a.lv[l];
      const T* c = coeffs + L.c;
      if (l > top) cur[l] = nullptr;
      for (int k = 0; k < a.npre; ++k) {
        T* dst = other(l);
        tail_sweep<T>(c, cur[l], cur[l] == nullptr, rhs[l], dst, a.wpre[k], L, a);
        cur[l] = dst;
      }
      tail_restrict<T>(c, cur[l], cur[l] == nullptr, rhs[l], rhsw[l + 1], L, a.lv[l + 1], a, true);
    }
    coarsest();
    for (int l = last - 1; l >= top; --l) {
      const TailLevel& L = a.lv[l];
      const T* c = coeffs + L.c;
      if (cur[l] == nullptr) {
        tail_prolong<T>(cur[l + 1], nullptr, false, bufA[l], L, a.lv[l + 1]);
        cur[l] = bufA[l];
      } else {
        T* dst = other(l);
        tail_prolong<T>(cur[l + 1], cur[l], true, dst, L, a.lv[l + 1]);
        cur[l] = dst;
      }
      for (int k = 0; k < a.npost; ++k) {
        T* dst = other(l);
        tail_sweep<T>(c, cur[l], false, rhs[l], dst, a.wpost[k], L, a);
        cur[l] = dst;
      }
    }
  };
  if (a.fmg) {
    for (int l = 0; l < last; ++l)
      tail_restrict<T>(nullptr, nullptr, true, rhs[l], rhsw[l + 1], a.lv[l], a.lv[l + 1], a, false);
    if (last == 0) {
      coarsest();
    } else {
      // the coarsest problem, then every finer level: the interpolated solution of the level below as the start of one cycle
      coarsest();
      for (int l = last - 1; l >= 0; --l) {
        tail_prolong<T>(cur[l + 1], nullptr, false, bufA[l], a.lv[l], a.lv[l + 1]);
        cur[l] = bufA[l];
        vcycle(l);
      }
    }
  } else {
    cur[0] = xin;
    if (last == 0)
      coarsest();
    else
      vcycle(0);
  }
  if (cur[0] != xout) {
    for (int i = threadIdx.x; i < a.lv[0].size; i += kTailThreads) xout[i] = cur[0][i];
  }
}

template <typename T>
static int vcycle_tail(const T* coeffs, const int64_t* shapes, const int* halve, int nlev, int ndim, const T* xin,
                       const T* b, T* xout, T* work, int64_t work_len, const T* inv, int ninv, const T* wpre, int npre,
                       const T* wpost, int npost, int fmg, void* stream) {
  if (nlev < 1 || nlev > kTailMaxLev || ndim < 1 || ndim > 3 || !shapes || (nlev > 1 && !halve) || !coeffs || !b ||
      !xout || !work || !inv || npre < 0 || npre > 4 || npost < 0 || npost > 4 || (npre && !wpre) || (npost && !wpost)) {
    set_error("stencil_vcycle_tail: bad arguments (1..%d levels, ndim 1..3, at most 4 sweeps per side)", kTailMaxLev);
    return ODIL_E_INVAL;
  }
  if (xin == xout || b == xout) {
    set_error("stencil_vcycle_tail: xout must not alias xin or b");
    return ODIL_E_INVAL;
  }
  TailArgs<T> a;
  a.nlev = nlev, a.ndim = ndim;
  for (int d = 0; d < 3; ++d) {
    const int i = d - (3 - ndim);
    a.has[d] = i >= 0 ? 1 : 0;
    a.slot[d] = i >= 0 ? 1 + 2 * i : 0;
  }
  int64_t coff = 0, woff = 0;
  for (int l = 0; l < kTailMaxLev; ++l) {
    TailLevel& L = a.lv[l];
    L.size = 0, L.c = 0, L.x = L.t = L.b = 0;
    for (int d = 0; d < 3; ++d) L.n[d] = 1, L.halve[d] = 0, L.rn[d] = 1.0f;
    if (l >= nlev) continue;
    int64_t size = 1;
    for (int d = 0; d < 3; ++d) {
      const int i = d - (3 - ndim);
      if (i < 0) continue;
      const int64_t n = shapes[l * ndim + i];
      if (n < 1 || n > 4096) {
        set_error("stencil_vcycle_tail: extent %lld of level %d", (long long)n, l);
        return ODIL_E_INVAL;
      }
      L.n[d] = (int)n;
      L.rn[d] = 1.0f / (float)n;
      size *= n;
      if (l + 1 < nlev) {
        L.halve[d] = halve[l * ndim + i] != 0;
        const int64_t nc = shapes[(l + 1) * ndim + i];
        if ((L.halve[d] && (n % 2 || nc != n / 2)) || (!L.halve[d] && nc != n)) {
          set_error("stencil_vcycle_tail: level %d does not follow from level %d along axis %d", l + 1, l, i);
          return ODIL_E_INVAL;
        }
      }
    }
    if (size > (1 << 20)) {
      set_error("stencil_vcycle_tail: level %d has %lld cells (one workgroup walks the tail)", l, (long long)size);
      return ODIL_E_INVAL;
    }
    L.size = (int)size;
    L.c = coff;
    coff += (int64_t)(2 * ndim + 1) * size;
    L.x = woff, L.t = woff + size, L.b = woff + 2 * size;
    woff += 3 * size;
  }
  if (woff > work_len) {
    set_error("stencil_vcycle_tail: work array of %lld values, %lld needed", (long long)work_len, (long long)woff);
    return ODIL_E_INVAL;
  }
  if (ninv != a.lv[nlev - 1].size) {
    set_error("stencil_vcycle_tail: the coarsest level has %d unknowns, the inverse %d", a.lv[nlev - 1].size, ninv);
    return ODIL_E_INVAL;
  }
  a.ninv = ninv;
  a.npre = npre, a.npost = npost, a.fmg = fmg != 0;
  for (int k = 0; k < 4; ++k) {
    a.wpre[k] = k < npre ? wpre[k] : T(0);
    a.wpost[k] = k < npost ? wpost[k] : T(0);
  }
  hipLaunchKernelGGL((k_vcycle_tail<T>), dim3(1), dim3(kTailThreads), 0, (hipStream_t)stream, coeffs, xin, b, xout, work,
                     inv, a);
  return check_launch("k_vcycle_tail");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_stencil_vcycle_tail_f64(const double* coeffs, const int64_t* shapes, const int* halve, int nlev, int ndim,
                                 const double* xin, const double* b, double* xout, double* work, int64_t work_len,
                                 const double* inv, int ninv, const double* wpre, int npre, const double* wpost,
                                 int npost, int fmg, void* stream) {
  return vcycle_tail<double>(coeffs, shapes, halve, nlev, ndim, xin, b, xout, work, work_len, inv, ninv, wpre, npre,
                             wpost, npost, fmg, stream);
}
int odil_stencil_vcycle_tail_f32(const float* coeffs, const int64_t* shapes, const int* halve, int nlev, int ndim,
                                 const float* xin, const float* b, float* xout, float* work, int64_t work_len,
                                 const float* inv, int ninv, const float* wpre, int npre, const float* wpost, int npost,
                                 int fmg, void* stream) {
  return vcycle_tail<float>(coeffs, shapes, halve, nlev, ndim, xin, b, xout, work, work_len, inv, ninv, wpre, npre, wpost,
                            npost, fmg, stream);
}
}  // extern "C"
