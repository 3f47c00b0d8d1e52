// Shared pieces of the Poisson stencil kernels (poisson.hip, poisson_fused.hip):
// reference examples/poisson/poisson.py:57-113, extrap_quadh core.py:1439-1445.
#pragma once
#include <math.h>

#include "common.h"

namespace odil {

template <typename T>
struct VecOf;
template <>
struct VecOf<double> {
  static constexpr int N = 2;
};
template <>
struct VecOf<float> {
  static constexpr int N = 4;
};

template <typename T, int N>
struct alignas(N * sizeof(T)) Pack {
  T v[N];
};

// Row-segment loads / stores of N consecutive values.  FULL (every active lane owns N valid
// cells; decided on the host from X % N) compiles to one unconditional wide access per call, so
// the loads of a step sit in one basic block and are all in flight together; global memory on
// gfx950 needs only element alignment for a wide access.  !FULL is the ragged-tail path.
template <typename T, int N, bool FULL, bool STREAM = false>
__device__ inline void load_vec(const T* __restrict__ p, int64_t valid, T out[N]) {
  if (FULL) {
    typedef T VT __attribute__((ext_vector_type(N), aligned(sizeof(T))));
    const VT q = STREAM ? __builtin_nontemporal_load(reinterpret_cast<const VT*>(p)) : *reinterpret_cast<const VT*>(p);
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = q[i];
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = i < valid ? p[i] : T(0);
  }
}

template <typename T, int N, bool FULL, bool STREAM = false>
__device__ inline void store_vec(T* __restrict__ p, int64_t valid, const T in[N]) {
  if (FULL) {
    typedef T VT __attribute__((ext_vector_type(N), aligned(sizeof(T))));
    VT q;
#pragma unroll
    for (int i = 0; i < N; ++i) q[i] = in[i];
    if (STREAM)
      __builtin_nontemporal_store(q, reinterpret_cast<VT*>(p));
    else
      *reinterpret_cast<VT*>(p) = q;
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i)
      if (i < valid) p[i] = in[i];
  }
}

// x / h2 -- as a multiplication when h2 is an exact power of two (bit-identical), else a
// true division: the f64 divide is ~35 VALU instructions and three of them per cell are
// enough to make this HBM-bound kernel VALU-bound.
template <typename T>
struct H2 {
  T h2[3], inv[3];
  int mul_ok[3];
};

template <typename T>
__device__ inline T div_h2(T v, const H2<T>& h, int ax) {
  return h.mul_ok[ax] ? v * h.inv[ax] : v / h.h2[ax];
}

// x / 3, correctly rounded, without the ~35-instruction IEEE divide: q = RN(x * RN(1/3)),
// exact remainder r = x - 3q by FMA, result RN(q + r * RN(1/3)) (Markstein's correction; for
// the divisor 3 it reproduces the correctly rounded quotient -- checked against exact
// rational arithmetic on 3e5 random doubles incl. random bit patterns, and by the GPU
// parity tests that demand bit equality with NumPy's x / 3).  The wall extrapolation sits in
// a per-lane branch, so a wave containing x == 0 or x == X-1 would otherwise execute two full
// divides for every cell of the wave.
__device__ inline double div3(double x) {
  const double r = 1.0 / 3.0;
  const double q = x * r;
  return fma(fma(-3.0, q, x), r, q);
}
__device__ inline float div3(float x) {
  const float r = 1.0f / 3.0f;
  const float q = x * r;
  return fmaf(fmaf(-3.0f, q, x), r, q);
}

// One axis of poisson.py:57-68 + :112: ghosts by extrap_quadh(q+-, q, 0), then (qp - 2q + qm)/h2.
// The extrapolation (a division by 3) is evaluated only where a boundary is touched.
template <typename T>
__device__ inline T axis_term(T q, T qwm, T qwp, bool lo, bool hi, const H2<T>& h, int ax) {
  T qm = qwm, qp = qwp;
  if (lo || hi) {
    if (lo) qm = div3(qwp - T(6) * q);
    if (hi) qp = div3(qwm - T(6) * q);
  }
  return div_h2<T>(qp - T(2) * q + qm, h, ax);
}

// Row i of the 1-D operator: cm(i) u[i-1] + c0(i) u[i] + cp(i) u[i+1], all / h2, with
//   cm(i) = [i != 0] + [i == n-1]/3,  cp(i) = [i != n-1] + [i == 0]/3,
//   c0(i) = -2 - 2[i == 0] - 2[i == n-1]          (poisson.py:57-68).
// Transpose: g[j] = cm(j+1) fb[j+1] + c0(j) fb[j] + cp(j-1) fb[j-1]  (periodic indices;
// the masked coefficients vanish exactly where the roll wraps).
template <typename T>
__device__ inline T adj_axis(T fb, T fbm, T fbp, int64_t j, int64_t n, const H2<T>& h, int ax) {
  T s;
  if (j >= 2 && j < n - 2) {
    // interior: (fb[j+1] + fb[j-1]) - 2 fb[j]
    s = (fbp + fbm) + T(-2) * fb;
  } else {
    const int64_t jp = j == n - 1 ? 0 : j + 1;
    const int64_t jm = j == 0 ? n - 1 : j - 1;
    s = T(0);
    // from row jp: cm(jp) * fb[jp]
    if (jp != 0) s = s + fbp;
    if (jp == n - 1) s = s + div3(fbp);
    // from row jm: cp(jm) * fb[jm]
    if (jm != n - 1) s = s + fbm;
    if (jm == 0) s = s + div3(fbm);
    // from row j
    T c0 = T(-2);
    if (j == 0) c0 = c0 - T(2);
    if (j == n - 1) c0 = c0 - T(2);
    s = s + c0 * fb;
  }
  return div_h2<T>(s, h, ax);
}

template <typename T>
inline H2<T> make_h2(const T h[3]) {
  H2<T> r;
  for (int i = 0; i < 3; ++i) {
    int e;
    r.h2[i] = h[i];
    r.inv[i] = T(1) / h[i];
    r.mul_ok[i] = frexp((double)h[i], &e) == 0.5 && r.inv[i] * h[i] == T(1);
  }
  return r;
}

}  // namespace odil
