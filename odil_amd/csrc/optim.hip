// Optimizer updates and flat-vector algebra of the ODIL hot path on gfx950
// (reference src/odil/optimizer.py:256-341; L-BFGS / CG building blocks).
// Pure HBM streaming: 16 B per lane, grid capped at 8 workgroups per CU, grid-stride.
#include "common.h"

namespace odil {

template <typename T>
struct Vec16;
template <>
struct Vec16<double> {
  static constexpr int N = 2;
  typedef double type __attribute__((ext_vector_type(2)));
};
template <>
struct Vec16<float> {
  static constexpr int N = 4;
  typedef float type __attribute__((ext_vector_type(4)));
};

template <typename T>
__device__ inline void adam_one(T& x, T& m, T& v, T g, T alpha, T omb1, T omb2, T eps) {
  // optimizer.py:316-318
  m = m + (g - m) * omb1;
  v = v + (g * g - v) * omb2;
  x = x - (m * alpha) / (sqrt(v) + eps);
}

// NT: the arrays are far larger than the caches (chosen by the launcher): streaming loads and stores, +4-6 % on
// top of the uncapped grid (6.3 / 6.5 TB/s for f32 / f64 at 1 G / 128 M elements).
template <typename T, bool NT>
__global__ __launch_bounds__(kBlock) void k_adam(T* __restrict__ x, T* __restrict__ m, T* __restrict__ v,
                                                const T* __restrict__ g, int64_t n, T alpha, T omb1, T omb2, T eps,
                                                int vec_ok, const T* __restrict__ alpha_dev) {
  if (alpha_dev) alpha = *alpha_dev;
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type VT;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  if (vec_ok) {
    const int64_t nv = n / V;
    for (int64_t i = tid; i < nv; i += nthreads) {
      VT xv, mv, vv, gv;
      if constexpr (NT) {
        xv = __builtin_nontemporal_load(reinterpret_cast<VT*>(x) + i);
        mv = __builtin_nontemporal_load(reinterpret_cast<VT*>(m) + i);
        vv = __builtin_nontemporal_load(reinterpret_cast<VT*>(v) + i);
        gv = __builtin_nontemporal_load(reinterpret_cast<const VT*>(g) + i);
      } else {
        xv = reinterpret_cast<VT*>(x)[i], mv = reinterpret_cast<VT*>(m)[i], vv = reinterpret_cast<VT*>(v)[i];
        gv = reinterpret_cast<const VT*>(g)[i];
      }
      T* xp = reinterpret_cast<T*>(&xv);
      T* mp = reinterpret_cast<T*>(&mv);
      T* vp = reinterpret_cast<T*>(&vv);
      const T* gp = reinterpret_cast<const T*>(&gv);
#pragma unroll
      for (int k = 0; k < V; ++k) adam_one<T>(xp[k], mp[k], vp[k], gp[k], alpha, omb1, omb2, eps);
      if constexpr (NT) {
        __builtin_nontemporal_store(xv, reinterpret_cast<VT*>(x) + i);
        __builtin_nontemporal_store(mv, reinterpret_cast<VT*>(m) + i);
        __builtin_nontemporal_store(vv, reinterpret_cast<VT*>(v) + i);
      } else {
        reinterpret_cast<VT*>(x)[i] = xv;
        reinterpret_cast<VT*>(m)[i] = mv;
        reinterpret_cast<VT*>(v)[i] = vv;
      }
    }
    for (int64_t i = nv * V + tid; i < n; i += nthreads) adam_one<T>(x[i], m[i], v[i], g[i], alpha, omb1, omb2, eps);
  } else {
    for (int64_t i = tid; i < n; i += nthreads) adam_one<T>(x[i], m[i], v[i], g[i], alpha, omb1, omb2, eps);
  }
}

static inline int aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
int adam_launch(T* x, T* m, T* v, const T* g, int64_t n, T alpha, T omb1, T omb2, T eps, hipStream_t stream,
                const T* alpha_dev);

template <typename T>
static int adam_step(T* x, T* m, T* v, const T* g, int64_t n, T alpha, T omb1, T omb2, T eps, const T* alpha_dev,
                     void* stream) {
  return adam_launch<T>(x, m, v, g, n, alpha, omb1, omb2, eps, (hipStream_t)stream, alpha_dev);
}

template <typename T>
int adam_launch(T* x, T* m, T* v, const T* g, int64_t n, T alpha, T omb1, T omb2, T eps, hipStream_t stream,
                const T* alpha_dev) {
  if (!x || !m || !v || !g || n < 0) {
    set_error("adam_step: null pointer or n < 0");
    return ODIL_E_INVAL;
  }
  if (n == 0) return 0;
  const int vec_ok = aligned16(x) && aligned16(m) && aligned16(v) && aligned16(g);
  if (7 * n * (int64_t)sizeof(T) > kStreamBytes)
    hipLaunchKernelGGL((k_adam<T, true>), dim3(grid_flat(n, kBlock * Vec16<T>::N)), dim3(kBlock), 0, stream, x, m, v, g,
                       n, alpha, omb1, omb2, eps, vec_ok, alpha_dev);
  else
    hipLaunchKernelGGL((k_adam<T, false>), dim3(grid_flat(n, kBlock * Vec16<T>::N)), dim3(kBlock), 0, stream, x, m, v, g,
                       n, alpha, omb1, omb2, eps, vec_ok, alpha_dev);
  return check_launch("k_adam");
}
template int adam_launch<double>(double*, double*, double*, const double*, int64_t, double, double, double, double,
                                 hipStream_t, const double*);
template int adam_launch<float>(float*, float*, float*, const float*, int64_t, float, float, float, float,
                                hipStream_t, const float*);

// Adam on a strided family of contiguous pieces: elements o * stride + offset + j, o < gridDim.y, j < count (the
// planes next to the slab interfaces of a ghost-extended array whose sharded axis is not the leading one: the rest
// of the array was updated by the launch that formed its gradient, these wait for the neighbour's contribution).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_adam_pieces(T* __restrict__ x, T* __restrict__ m, T* __restrict__ v,
                                                       const T* __restrict__ g, int64_t stride, int64_t offset,
                                                       int64_t count, T alpha, T omb1, T omb2, T eps, int vec_ok,
                                                       const T* __restrict__ alpha_dev) {
  if (alpha_dev) alpha = *alpha_dev;
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type VT;
  const int64_t base = (int64_t)blockIdx.y * stride + offset;
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  if (vec_ok) {
    for (int64_t i = tid; i < count / V; i += nthreads) {
      const int64_t at = base + i * V;
      VT xv = *reinterpret_cast<VT*>(x + at), mv = *reinterpret_cast<VT*>(m + at), vv = *reinterpret_cast<VT*>(v + at);
      const VT gv = *reinterpret_cast<const VT*>(g + at);
      T* xp = reinterpret_cast<T*>(&xv);
      T* mp = reinterpret_cast<T*>(&mv);
      T* vp = reinterpret_cast<T*>(&vv);
      const T* gp = reinterpret_cast<const T*>(&gv);
#pragma unroll
      for (int k = 0; k < V; ++k) adam_one<T>(xp[k], mp[k], vp[k], gp[k], alpha, omb1, omb2, eps);
      *reinterpret_cast<VT*>(x + at) = xv;
      *reinterpret_cast<VT*>(m + at) = mv;
      *reinterpret_cast<VT*>(v + at) = vv;
    }
  } else {
    for (int64_t i = tid; i < count; i += nthreads)
      adam_one<T>(x[base + i], m[base + i], v[base + i], g[base + i], alpha, omb1, omb2, eps);
  }
}

template <typename T>
static int adam_pieces(T* x, T* m, T* v, const T* g, int64_t npieces, int64_t stride, int64_t offset, int64_t count,
                       T alpha, T omb1, T omb2, T eps, const T* alpha_dev, void* stream) {
  if (!x || !m || !v || !g || npieces < 0 || count < 0 || offset < 0 || npieces > 65535) {
    set_error("adam_step_pieces: null pointer, negative size or more than 65535 pieces");
    return ODIL_E_INVAL;
  }
  if (npieces == 0 || count == 0) return 0;
  constexpr int V = Vec16<T>::N;
  const int vec_ok = aligned16(x) && aligned16(m) && aligned16(v) && aligned16(g) && stride % V == 0 &&
                     offset % V == 0 && count % V == 0;
  const dim3 grid(grid_flat(count, kBlock * V), (unsigned)npieces);
  hipLaunchKernelGGL(k_adam_pieces<T>, grid, dim3(kBlock), 0, (hipStream_t)stream, x, m, v, g, stride, offset, count,
                     alpha, omb1, omb2, eps, vec_ok, alpha_dev);
  return check_launch("k_adam_pieces");
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_axpy(T* __restrict__ y, const T* __restrict__ x, int64_t n, T a) {
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) y[i] = y[i] + a * x[i];
}

template <typename T>
static int axpy(T* y, const T* x, int64_t n, T a, void* stream) {
  if (!y || !x || n < 0) {
    set_error("axpy: null pointer or n < 0");
    return ODIL_E_INVAL;
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_axpy<T>, dim3(grid_flat(n, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, y, x, n, a);
  return check_launch("k_axpy");
}

// Mixed-precision iterative refinement (gmg.py: float32 V-cycles inside a float64 residual loop): the two conversions
// with their scalings, one pass each.  narrow: y32 = (a / sqrt(*msq)) * x64 (the residual normalised by its own RMS, read
// from the device scalar the residual kernel just wrote: no host round trip); widen: y64 += (a * sqrt(*msq)) * x32.
__global__ __launch_bounds__(kBlock) void k_narrow_scale(const double* __restrict__ x, float* __restrict__ y, int64_t n,
                                                        double a, const double* __restrict__ msq) {
  const double s = msq ? a / sqrt(*msq > 0.0 ? *msq : 1.0) : a;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) y[i] = (float)(s * x[i]);
}
__global__ __launch_bounds__(kBlock) void k_widen_axpy(double* __restrict__ y, const float* __restrict__ x, int64_t n,
                                                      double a, const double* __restrict__ msq) {
  const double s = msq ? a * sqrt(*msq > 0.0 ? *msq : 1.0) : a;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) y[i] = y[i] + s * (double)x[i];
}

// y = a * (adev ? *adev : 1) * x   (cotangent of mean(x^2): 2/n * gout * x, core.py:1093)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_scale(const T* __restrict__ x, T* __restrict__ y, int64_t n, T a,
                                                 const T* __restrict__ adev) {
  const T f = adev ? a * adev[0] : a;
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) y[i] = f * x[i];
}

template <typename T>
static int scale(const T* x, T* y, int64_t n, T a, const T* adev, void* stream) {
  if (!x || !y || n < 0) {
    set_error("scale: null pointer or n < 0");
    return ODIL_E_INVAL;
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_scale<T>, dim3(grid_flat(n, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, x, y, n, a, adev);
  return check_launch("k_scale");
}

// y (+)= a (.) b
template <typename T>
__global__ __launch_bounds__(kBlock) void k_addcmul(T* __restrict__ y, const T* __restrict__ a, const T* __restrict__ b,
                                                   int64_t n, int accumulate) {
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
    const T v = a[i] * b[i];
    y[i] = accumulate ? y[i] + v : v;
  }
}

template <typename T>
static int addcmul(T* y, const T* a, const T* b, int64_t n, int accumulate, void* stream) {
  if (!y || !a || !b || n < 0) {
    set_error("addcmul: null pointer or n < 0");
    return ODIL_E_INVAL;
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_addcmul<T>, dim3(grid_flat(n, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, y, a, b, n,
                     accumulate);
  return check_launch("k_addcmul");
}

// out[k] = <a_k, b>: blockIdx.y = k; contiguous chunk per workgroup; fixed order.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dots(const T* __restrict__ a, int64_t lda, const T* __restrict__ b,
                                                int64_t n, double* __restrict__ partials) {
  const T* ak = a + (int64_t)blockIdx.y * lda;
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  int64_t hi = lo + per;
  if (hi > n) hi = n;
  double local = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) local += (double)ak[i] * (double)b[i];
  const double total = block_sum(local);
  if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * kDotPartials + blockIdx.x] = total;
}

template <typename T>
static int dots(const T* a, int64_t lda, int nvec, const T* b, int64_t n, double* partials, T* out, void* stream) {
  if (!a || !b || !partials || !out || n < 1 || nvec < 1 || nvec > 65535) {
    set_error("dots: null pointer, n < 1 or nvec=%d out of range", nvec);
    return ODIL_E_INVAL;
  }
  int grid = grid_for(n, kBlock * 8);
  if (grid > kDotPartials) grid = kDotPartials;
  hipLaunchKernelGGL(k_dots<T>, dim3(grid, nvec), dim3(kBlock), 0, (hipStream_t)stream, a, lda, b, n, partials);
  if (int e = check_launch("k_dots")) return e;
  return launch_final_reduce<T>(partials, grid, kDotPartials, nvec, 1.0, out, (hipStream_t)stream);
}

// out[j][k] = <a_k, b_j> for up to three right-hand vectors in ONE pass over the a_k (the L-BFGS
// history is by far the largest operand: S^T y, S^T s and S^T g cost one read of S instead of three).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dots3(const T* __restrict__ a, int64_t lda, const T* __restrict__ b0,
                                                 const T* __restrict__ b1, const T* __restrict__ b2, int64_t n,
                                                 int nvec, double* __restrict__ partials, int vec_ok) {
  constexpr int V = Vec16<T>::N;
  typedef typename Vec16<T>::type VT;
  const T* ak = a + (int64_t)blockIdx.y * lda;
  // chunk boundaries on multiples of V so that every block reads whole 16 B packs
  const int64_t per = (((n + gridDim.x - 1) / gridDim.x) + V - 1) / V * V;
  const int64_t lo = (int64_t)blockIdx.x * per;
  int64_t hi = lo + per;
  if (hi > n) hi = n;
  double l0 = 0.0, l1 = 0.0, l2 = 0.0;
  int64_t i = lo;
  if (vec_ok) {
    // 16 B per lane and two packs in flight per stream: the history is read once at HBM speed, the
    // right-hand vectors come from the last-level cache
    const int64_t nv = lo < hi ? (hi - lo) / V : 0;
    for (int64_t p = threadIdx.x; p < nv; p += kBlock) {
      const VT av = *reinterpret_cast<const VT*>(ak + lo + p * V);
      const VT x0 = *reinterpret_cast<const VT*>(b0 + lo + p * V);
      VT x1 = x0, x2 = x0;
      if (b1) x1 = *reinterpret_cast<const VT*>(b1 + lo + p * V);
      if (b2) x2 = *reinterpret_cast<const VT*>(b2 + lo + p * V);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const double ad = (double)av[k];
        l0 += ad * (double)x0[k];
        l1 += ad * (double)x1[k];
        l2 += ad * (double)x2[k];
      }
    }
    i = lo + nv * V;
  }
  for (i += threadIdx.x; i < hi; i += kBlock) {
    const double av = (double)ak[i];
    l0 += av * (double)b0[i];
    if (b1) l1 += av * (double)b1[i];
    if (b2) l2 += av * (double)b2[i];
  }
  if (!b1) l1 = 0.0;
  if (!b2) l2 = 0.0;
  const double t0 = block_sum(l0), t1 = block_sum(l1), t2 = block_sum(l2);
  if (threadIdx.x == 0) {
    const int64_t base = (int64_t)blockIdx.y * kDotPartials + blockIdx.x;
    partials[base] = t0;
    partials[(int64_t)nvec * kDotPartials + base] = t1;
    partials[(int64_t)2 * nvec * kDotPartials + base] = t2;
  }
}

// The same products with the right-hand vectors read ONCE: a workgroup owns a chunk of elements for
// all history vectors, keeps its part of b0 / b1 / b2 in registers and walks the history; per vector
// the three wave sums go to that wave's slot of an LDS table (fixed order, no atomics).  The kernel
// above re-reads b0 / b1 / b2 for every a_k -- four streams per history element, of which three come
// from the last-level cache: 2.2 TB/s on the history of L-BFGS (m = 50, 1.4 M unknowns).
constexpr int kDots3MaxVec = 128;  // S and Y of L-BFGS (m = 50) interleaved in one matrix
template <typename T, int E>
__global__ __launch_bounds__(kBlock) void k_dots3_once(const T* __restrict__ a, int64_t lda, const T* __restrict__ b0,
                                                      const T* __restrict__ b1, const T* __restrict__ b2, int64_t n,
                                                      int nvec, double* __restrict__ partials) {
  constexpr int V = Vec16<T>::N;
  constexpr int NP = E / V;  // 16 B packs per thread and chunk
  typedef typename Vec16<T>::type VT;
  __shared__ double acc[3][kBlock / 64][kDots3MaxVec];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int q = threadIdx.x; q < 3 * (kBlock / 64) * kDots3MaxVec; q += kBlock) (&acc[0][0][0])[q] = 0.0;
  __syncthreads();
  const int64_t chunk = (int64_t)kBlock * E;
  for (int64_t base = (int64_t)blockIdx.x * chunk; base < n; base += (int64_t)gridDim.x * chunk) {
    VT x0[NP], x1[NP], x2[NP];
    bool ok[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int64_t i = base + ((int64_t)p * kBlock + threadIdx.x) * V;
      ok[p] = i + V <= n;  // n is a multiple of V on this path
      const int64_t j = ok[p] ? i : 0;
      x0[p] = *reinterpret_cast<const VT*>(b0 + j);
      x1[p] = b1 ? *reinterpret_cast<const VT*>(b1 + j) : x0[p];
      x2[p] = b2 ? *reinterpret_cast<const VT*>(b2 + j) : x0[p];
    }
    for (int k = 0; k < nvec; ++k) {
      const T* ak = a + (int64_t)k * lda;
      VT av[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int64_t i = base + ((int64_t)p * kBlock + threadIdx.x) * V;
        av[p] = *reinterpret_cast<const VT*>(ak + (ok[p] ? i : 0));
      }
      double l0 = 0.0, l1 = 0.0, l2 = 0.0;
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < V; ++e) {
          const double ad = ok[p] ? (double)av[p][e] : 0.0;
          l0 += ad * (double)x0[p][e];
          l1 += ad * (double)x1[p][e];
          l2 += ad * (double)x2[p][e];
        }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        l0 += __shfl_down(l0, off, 64);
        l1 += __shfl_down(l1, off, 64);
        l2 += __shfl_down(l2, off, 64);
      }
      if (lane == 0) {
        acc[0][wave][k] += l0;
        acc[1][wave][k] += l1;
        acc[2][wave][k] += l2;
      }
    }
  }
  __syncthreads();
  for (int q = threadIdx.x; q < 3 * nvec; q += kBlock) {
    const int j = q / nvec, k = q - j * nvec;
    double t = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) t += acc[j][w][k];
    if ((j == 1 && !b1) || (j == 2 && !b2)) t = 0.0;
    partials[(int64_t)q * kDotPartials + blockIdx.x] = t;
  }
}

template <typename T>
static int dots3(const T* a, int64_t lda, int nvec, const T* b0, const T* b1, const T* b2, int64_t n, double* partials,
                 T* out, void* stream) {
  if (!a || !b0 || !partials || !out || n < 1 || nvec < 1 || nvec > 65535) {
    set_error("dots3: null pointer, n < 1 or nvec=%d out of range", nvec);
    return ODIL_E_INVAL;
  }
  // every block ends with three block reductions: give it enough elements (64 per thread) to hide them
  int grid = grid_for(n, kBlock * 64);
  if (grid > kDotPartials) grid = kDotPartials;
  const int vec_ok = aligned16(a) && aligned16(b0) && (!b1 || aligned16(b1)) && (!b2 || aligned16(b2)) &&
                     (lda * (int64_t)sizeof(T)) % 16 == 0;
  if (vec_ok && nvec <= kDots3MaxVec && n % Vec16<T>::N == 0 && n >= (int64_t)kBlock * 64) {
    constexpr int E = 4 * Vec16<T>::N;  // four 16 B packs per thread, stream and chunk
    int chunks = grid_for(n, kBlock * E);
    if (chunks > kDotPartials) chunks = kDotPartials;
    hipLaunchKernelGGL((k_dots3_once<T, E>), dim3(chunks), dim3(kBlock), 0, (hipStream_t)stream, a, lda, b0, b1, b2, n,
                       nvec, partials);
    if (int e = check_launch("k_dots3_once")) return e;
    return launch_final_reduce<T>(partials, chunks, kDotPartials, 3 * nvec, 1.0, out, (hipStream_t)stream);
  }
  hipLaunchKernelGGL(k_dots3<T>, dim3(grid, nvec), dim3(kBlock), 0, (hipStream_t)stream, a, lda, b0, b1, b2, n, nvec,
                     partials, vec_ok);
  if (int e = check_launch("k_dots3")) return e;
  return launch_final_reduce<T>(partials, grid, kDotPartials, 3 * nvec, 1.0, out, (hipStream_t)stream);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_lincomb(T* __restrict__ y, T beta, const T* __restrict__ a, int64_t lda,
                                                   int nvec, const T* __restrict__ coef, int64_t n) {
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += nthreads) {
    T acc = beta == T(0) ? T(0) : beta * y[i];
    for (int k = 0; k < nvec; ++k) acc = acc + coef[k] * a[(int64_t)k * lda + i];
    y[i] = acc;
  }
}

template <typename T>
static int lincomb(T* y, T beta, const T* a, int64_t lda, int nvec, const T* coef, int64_t n, void* stream) {
  if (!y || n < 0 || nvec < 0 || (nvec > 0 && (!a || !coef))) {
    set_error("lincomb: null pointer or negative size");
    return ODIL_E_INVAL;
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_lincomb<T>, dim3(grid_for(n, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, y, beta, a, lda,
                     nvec, coef, n);
  return check_launch("k_lincomb");
}

// One pass over the new gradient for everything the L-BFGS line search reads on the host after an
// evaluation: <g, d> (Wolfe test), <g, g> and max |g| (projected-gradient stopping test); fixed
// summation order.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_lbfgs_probe(const T* __restrict__ g, const T* __restrict__ d, int64_t n,
                                                       double* __restrict__ partials) {
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  int64_t hi = lo + per;
  if (hi > n) hi = n;
  double gd = 0.0, gg = 0.0, gm = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
    const double gi = (double)g[i];
    gd += gi * (double)d[i];
    gg += gi * gi;
    gm = fmax(gm, fabs(gi));
  }
  const double t0 = block_sum(gd), t1 = block_sum(gg);
  __shared__ double wave_max[kBlock / 64];
  for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_down(gm, off, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) gm = fmax(gm, wave_max[w]);
    partials[blockIdx.x] = t0;
    partials[kDotPartials + blockIdx.x] = t1;
    partials[2 * kDotPartials + blockIdx.x] = gm;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_lbfgs_probe_final(const double* __restrict__ partials, int count,
                                                             T* __restrict__ out) {
  double gd = 0.0, gg = 0.0, gm = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) {
    gd += partials[i];
    gg += partials[kDotPartials + i];
    gm = fmax(gm, partials[2 * kDotPartials + i]);
  }
  const double t0 = block_sum(gd), t1 = block_sum(gg);
  __shared__ double wave_max[kBlock / 64];
  for (int off = 32; off > 0; off >>= 1) gm = fmax(gm, __shfl_down(gm, off, 64));
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = gm;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < kBlock / 64; ++w) gm = fmax(gm, wave_max[w]);
    out[0] = (T)t0;
    out[1] = (T)t1;
    out[2] = (T)gm;
  }
}

template <typename T>
static int lbfgs_probe(const T* g, const T* d, int64_t n, double* partials, T* out, void* stream) {
  if (!g || !d || !partials || !out || n < 1) {
    set_error("lbfgs_probe: null pointer or n < 1");
    return ODIL_E_INVAL;
  }
  int grid = grid_for(n, kBlock * 8);
  if (grid > kDotPartials) grid = kDotPartials;
  hipLaunchKernelGGL(k_lbfgs_probe<T>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, g, d, n, partials);
  if (int e = check_launch("k_lbfgs_probe")) return e;
  hipLaunchKernelGGL(k_lbfgs_probe_final<T>, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, partials, grid, out);
  return check_launch("k_lbfgs_probe_final");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_adam_step_f64(double* x, double* m, double* v, const double* g, int64_t n, double alpha,
                       double one_minus_b1, double one_minus_b2, double eps, const double* alpha_dev, void* stream) {
  return adam_step<double>(x, m, v, g, n, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev, stream);
}
int odil_adam_step_f32(float* x, float* m, float* v, const float* g, int64_t n, float alpha, float one_minus_b1,
                       float one_minus_b2, float eps, const float* alpha_dev, void* stream) {
  return adam_step<float>(x, m, v, g, n, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev, stream);
}
int odil_adam_step_pieces_f64(double* x, double* m, double* v, const double* g, int64_t npieces, int64_t stride,
                              int64_t offset, int64_t count, double alpha, double one_minus_b1, double one_minus_b2,
                              double eps, const double* alpha_dev, void* stream) {
  return adam_pieces<double>(x, m, v, g, npieces, stride, offset, count, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev,
                             stream);
}
int odil_adam_step_pieces_f32(float* x, float* m, float* v, const float* g, int64_t npieces, int64_t stride,
                              int64_t offset, int64_t count, float alpha, float one_minus_b1, float one_minus_b2,
                              float eps, const float* alpha_dev, void* stream) {
  return adam_pieces<float>(x, m, v, g, npieces, stride, offset, count, alpha, one_minus_b1, one_minus_b2, eps, alpha_dev,
                            stream);
}
int odil_narrow_scale(const double* x, float* y, int64_t n, double a, const double* msq, void* stream) {
  if (!x || !y || n < 0) {
    set_error("narrow_scale: null pointer or n < 0");
    return ODIL_E_INVAL;
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_narrow_scale, dim3(grid_flat(n, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, x, y, n, a, msq);
  return check_launch("k_narrow_scale");
}
int odil_widen_axpy(double* y, const float* x, int64_t n, double a, const double* msq, void* stream) {
  if (!x || !y || n < 0) {
    set_error("widen_axpy: null pointer or n < 0");
    return ODIL_E_INVAL;
  }
  if (n == 0) return 0;
  hipLaunchKernelGGL(k_widen_axpy, dim3(grid_flat(n, kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream, y, x, n, a, msq);
  return check_launch("k_widen_axpy");
}
int odil_axpy_f64(double* y, const double* x, int64_t n, double a, void* stream) {
  return axpy<double>(y, x, n, a, stream);
}
int odil_axpy_f32(float* y, const float* x, int64_t n, float a, void* stream) {
  return axpy<float>(y, x, n, a, stream);
}
int odil_scale_f64(const double* x, double* y, int64_t n, double a, const double* adev, void* stream) {
  return scale<double>(x, y, n, a, adev, stream);
}
int odil_scale_f32(const float* x, float* y, int64_t n, float a, const float* adev, void* stream) {
  return scale<float>(x, y, n, a, adev, stream);
}
int odil_addcmul_f64(double* y, const double* a, const double* b, int64_t n, int accumulate, void* stream) {
  return addcmul<double>(y, a, b, n, accumulate, stream);
}
int odil_addcmul_f32(float* y, const float* a, const float* b, int64_t n, int accumulate, void* stream) {
  return addcmul<float>(y, a, b, n, accumulate, stream);
}
int odil_dots_f64(const double* a, int64_t lda, int nvec, const double* b, int64_t n, double* partials,
                  double* out, void* stream) {
  return dots<double>(a, lda, nvec, b, n, partials, out, stream);
}
int odil_dots_f32(const float* a, int64_t lda, int nvec, const float* b, int64_t n, double* partials, float* out,
                  void* stream) {
  return dots<float>(a, lda, nvec, b, n, partials, out, stream);
}
int odil_lbfgs_probe_f64(const double* g, const double* d, int64_t n, double* partials, double* out, void* stream) {
  return lbfgs_probe<double>(g, d, n, partials, out, stream);
}
int odil_lbfgs_probe_f32(const float* g, const float* d, int64_t n, double* partials, float* out, void* stream) {
  return lbfgs_probe<float>(g, d, n, partials, out, stream);
}
int odil_dots3_f64(const double* a, int64_t lda, int nvec, const double* b0, const double* b1, const double* b2,
                   int64_t n, double* partials, double* out, void* stream) {
  return dots3<double>(a, lda, nvec, b0, b1, b2, n, partials, out, stream);
}
int odil_dots3_f32(const float* a, int64_t lda, int nvec, const float* b0, const float* b1, const float* b2, int64_t n,
                   double* partials, float* out, void* stream) {
  return dots3<float>(a, lda, nvec, b0, b1, b2, n, partials, out, stream);
}
int odil_lincomb_f64(double* y, double beta, const double* a, int64_t lda, int nvec, const double* coef, int64_t n,
                     void* stream) {
  return lincomb<double>(y, beta, a, lda, nvec, coef, n, stream);
}
int odil_lincomb_f32(float* y, float beta, const float* a, int64_t lda, int nvec, const float* coef, int64_t n,
                     void* stream) {
  return lincomb<float>(y, beta, a, lda, nvec, coef, n, stream);
}
}
