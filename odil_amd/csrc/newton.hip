// Newton building blocks of the ODIL hot path on gfx950: the sparse Jacobian that
// Problem.linearize / field_to_matrix assemble on the host with scipy.sparse
// (reference src/odil/core.py:1113-1217) is kept as per-shift coefficient arrays
// (core.py:1313-1361) and applied matrix-free, M and M^T, for the normal equations
// M^T M d = -M^T r (linsolver.py:17-26); a CSR export is provided for parity checks
// and host solvers.
#include "common.h"

namespace odil {

constexpr int kMaxShifts = 32;

struct ShiftArgs {
  int64_t n[4];
  int nshift;
  int64_t shift[kMaxShifts][4];
};

// Coordinates of a flat index, taken apart ONCE per element (the 64-bit divisions used to be repeated for
// every shift: 56 of them per cell of a 7-point stencil, which held odil_stencil_apply at 1.6 TB/s).
struct Coords {
  int64_t id[4];
};
__device__ inline Coords coords_of(const ShiftArgs& a, int64_t i) {
  Coords c;
  int64_t rem = i;
  for (int d = 3; d >= 0; --d) {
    if (a.n[d] == 1) {
      c.id[d] = 0;
    } else if (rem < (int64_t(1) << 31) && a.n[d] < (int64_t(1) << 31)) {
      const uint32_t r = (uint32_t)rem, n = (uint32_t)a.n[d];
      c.id[d] = r % n;
      rem = r / n;
    } else {
      c.id[d] = rem % a.n[d];
      rem /= a.n[d];
    }
  }
  return c;
}
// Flat index of the periodic neighbour; the shifts are stored reduced to [0, n) (fill_shifts).
__device__ inline int64_t shifted_index(const ShiftArgs& a, const Coords& c, int s, int sign) {
  int64_t idx = 0, stride = 1;
#pragma unroll
  for (int d = 3; d >= 0; --d) {
    int64_t p = c.id[d] + sign * a.shift[s][d];
    if (p >= a.n[d]) p -= a.n[d];
    if (p < 0) p += a.n[d];
    idx += p * stride;
    stride *= a.n[d];
  }
  return idx;
}

// y[r] = sum_s c_s[r] x[r + shift_s]   (cols = roll(arange, -shift), core.py:1158)
// y[j] = sum_s c_s[j - shift_s] x[j - shift_s]   (transpose)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_stencil_apply(const T* __restrict__ coeffs, const T* __restrict__ x,
                                                         T* __restrict__ y, ShiftArgs a, int transpose) {
  const int64_t size = a.n[0] * a.n[1] * a.n[2] * a.n[3];
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += nthreads) {
    T acc = T(0);
    const Coords c = coords_of(a, i);
    for (int s = 0; s < a.nshift; ++s) {
      if (!transpose) {
        acc = acc + coeffs[(int64_t)s * size + i] * x[shifted_index(a, c, s, +1)];
      } else {
        const int64_t r = shifted_index(a, c, s, -1);
        acc = acc + coeffs[(int64_t)s * size + r] * x[r];
      }
    }
    y[i] = acc;
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_csr_assemble(const T* __restrict__ coeffs, ShiftArgs a, int64_t col_offset,
                                                        int64_t* __restrict__ indptr, int64_t* __restrict__ indices,
                                                        T* __restrict__ data) {
  const int64_t size = a.n[0] * a.n[1] * a.n[2] * a.n[3];
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < size; i += nthreads) {
    indptr[i] = i * a.nshift;
    if (i == size - 1) indptr[size] = size * a.nshift;
    const Coords c = coords_of(a, i);
    for (int s = 0; s < a.nshift; ++s) {
      indices[i * a.nshift + s] = col_offset + shifted_index(a, c, s, +1);
      data[i * a.nshift + s] = coeffs[(int64_t)s * size + i];
    }
  }
}

// One level of a triangular solve along `axis`: for the rows whose `axis` coordinate is `level`,
//   x[r] = (b[r] - sum_{s != diag} c_s[r] x[r + shift_s]) / c_diag[r],
// every off-diagonal neighbour lying in a level already solved (the host checks the shifts).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_stencil_march_level(const T* __restrict__ coeffs, const T* __restrict__ b,
                                                               T* __restrict__ x, ShiftArgs a, int diag, int axis,
                                                               int64_t level) {
  int64_t plane = 1;
  for (int d = 0; d < 4; ++d)
    if (d != axis) plane *= a.n[d];
  const int64_t size = a.n[0] * a.n[1] * a.n[2] * a.n[3];
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < plane; p += nthreads) {
    Coords c;
    int64_t rem = p, i = 0, stride = 1;
    for (int d = 3; d >= 0; --d) {
      if (d == axis) {
        c.id[d] = level;
      } else {
        c.id[d] = rem % a.n[d];
        rem /= a.n[d];
      }
      i += c.id[d] * stride;
      stride *= a.n[d];
    }
    T acc = b[i];
    // a zero coefficient contributes nothing WHATEVER its neighbour holds: the rows at the ends of `axis` have
    // neighbours that wrap periodically into levels not solved yet (recognise_marching only proves their
    // coefficients zero), i.e. memory this solve has not written -- 0 * NaN must not enter the substitution
    for (int s = 0; s < a.nshift; ++s) {
      if (s == diag) continue;
      const T cs = coeffs[(int64_t)s * size + i];
      if (cs != T(0)) acc = acc - cs * x[shifted_index(a, c, s, +1)];
    }
    x[i] = acc / coeffs[(int64_t)diag * size + i];
  }
}

static int fill_shifts(ShiftArgs& a, const int64_t* shifts, int nshift, const int64_t* shape, int ndim) {
  if (ndim < 1 || ndim > ODIL_MAX_NDIM || !shape || !shifts || nshift < 1 || nshift > kMaxShifts) {
    set_error("stencil: invalid ndim=%d or nshift=%d (max %d)", ndim, nshift, kMaxShifts);
    return ODIL_E_INVAL;
  }
  canon_shape(shape, ndim, a.n);
  a.nshift = nshift;
  for (int s = 0; s < nshift; ++s)
    for (int d = 0; d < 4; ++d) {
      const int i = d - (4 - ndim);
      const int64_t sh = i >= 0 ? shifts[s * ndim + i] : 0;
      a.shift[s][d] = ((sh % a.n[d]) + a.n[d]) % a.n[d];  // periodic (mod.roll): reduced to [0, n)
    }
  return 0;
}

template <typename T>
static int stencil_apply(const T* coeffs, const int64_t* shifts, int nshift, const T* x, T* y, const int64_t* shape,
                         int ndim, int transpose, void* stream) {
  ShiftArgs a;
  if (int e = fill_shifts(a, shifts, nshift, shape, ndim)) return e;
  if (!coeffs || !x || !y) {
    set_error("stencil_apply: null pointer");
    return ODIL_E_INVAL;
  }
  hipLaunchKernelGGL(k_stencil_apply<T>, dim3(grid_flat(prod4(a.n), kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     coeffs, x, y, a, transpose);
  return check_launch("k_stencil_apply");
}

template <typename T>
static int csr_assemble(const T* coeffs, const int64_t* shifts, int nshift, const int64_t* shape, int ndim,
                        int64_t col_offset, int64_t* indptr, int64_t* indices, T* data, void* stream) {
  ShiftArgs a;
  if (int e = fill_shifts(a, shifts, nshift, shape, ndim)) return e;
  if (!coeffs || !indptr || !indices || !data) {
    set_error("csr_assemble: null pointer");
    return ODIL_E_INVAL;
  }
  hipLaunchKernelGGL(k_csr_assemble<T>, dim3(grid_flat(prod4(a.n), kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                     coeffs, a, col_offset, indptr, indices, data);
  return check_launch("k_csr_assemble");
}

// M x = b for a stencil matrix that is TRIANGULAR along one axis with a diagonal block on the diagonal (an operator
// that is explicit in time: every coefficient of the newest time level sits on the unknown itself): forward
// (direction > 0: all other shifts point to lower levels) or backward substitution, one small launch per level.
template <typename T>
static int stencil_march(const T* coeffs, const int64_t* shifts, int nshift, int diag, const T* b, T* x,
                         const int64_t* shape, int ndim, int axis, int direction, void* stream) {
  ShiftArgs a;
  if (int e = fill_shifts(a, shifts, nshift, shape, ndim)) return e;
  if (!coeffs || !b || !x || diag < 0 || diag >= nshift || axis < 0 || axis >= ndim) {
    set_error("stencil_march: null pointer, diagonal slot %d or axis %d out of range", diag, axis);
    return ODIL_E_INVAL;
  }
  const int ax = axis + (4 - ndim);
  for (int d = 0; d < 4; ++d)
    if (a.shift[diag][d] != 0) {
      set_error("stencil_march: slot %d is not the diagonal", diag);
      return ODIL_E_INVAL;
    }
  const int64_t levels = a.n[ax], plane = prod4(a.n) / levels;
  for (int64_t k = 0; k < levels; ++k) {
    const int64_t level = direction > 0 ? k : levels - 1 - k;
    hipLaunchKernelGGL(k_stencil_march_level<T>, dim3(grid_for(plane, kBlock)), dim3(kBlock), 0, (hipStream_t)stream,
                       coeffs, b, x, a, diag, ax, level);
  }
  return check_launch("k_stencil_march_level");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_stencil_apply_f64(const double* coeffs, const int64_t* shifts, int nshift, const double* x, double* y,
                           const int64_t* shape, int ndim, int transpose, void* stream) {
  return stencil_apply<double>(coeffs, shifts, nshift, x, y, shape, ndim, transpose, stream);
}
int odil_stencil_apply_f32(const float* coeffs, const int64_t* shifts, int nshift, const float* x, float* y,
                           const int64_t* shape, int ndim, int transpose, void* stream) {
  return stencil_apply<float>(coeffs, shifts, nshift, x, y, shape, ndim, transpose, stream);
}
int odil_stencil_march_f64(const double* coeffs, const int64_t* shifts, int nshift, int diag, const double* b, double* x,
                           const int64_t* shape, int ndim, int axis, int direction, void* stream) {
  return stencil_march<double>(coeffs, shifts, nshift, diag, b, x, shape, ndim, axis, direction, stream);
}
int odil_stencil_march_f32(const float* coeffs, const int64_t* shifts, int nshift, int diag, const float* b, float* x,
                           const int64_t* shape, int ndim, int axis, int direction, void* stream) {
  return stencil_march<float>(coeffs, shifts, nshift, diag, b, x, shape, ndim, axis, direction, stream);
}
int odil_csr_assemble_f64(const double* coeffs, const int64_t* shifts, int nshift, const int64_t* shape, int ndim,
                          int64_t col_offset, int64_t* indptr, int64_t* indices, double* data, void* stream) {
  return csr_assemble<double>(coeffs, shifts, nshift, shape, ndim, col_offset, indptr, indices, data, stream);
}
int odil_csr_assemble_f32(const float* coeffs, const int64_t* shifts, int nshift, const int64_t* shape, int ndim,
                          int64_t col_offset, int64_t* indptr, int64_t* indices, float* data, void* stream) {
  return csr_assemble<float>(coeffs, shifts, nshift, shape, ndim, col_offset, indptr, indices, data, stream);
}
}
