// Stencil access of the ODIL hot path on gfx950: Context.field(key, *shift, loc)
// (reference src/odil/core.py:910-975) as ONE gather instead of pad + roll + slice, and
// its transpose (the cotangent autodiff routes back to the source array).
#include "common.h"

namespace odil {

struct AccessArgs {
  int64_t sn[4];   // source array shape (canonical 4-D)
  int64_t on[4];   // output array shape
  int64_t np[4];   // padded extent = sn + pad
  int64_t pad[4];  // 1 where 'c' -> 'n' (zero-pad at the low end, core.py:956-960)
  int64_t shift[4];
};

// i = (((i0 * n1 + i1) * n2 + i2) * n3 + i3) taken apart; 32-bit divisions whenever the values fit (a 64-bit
// division is ~40 instructions, and these kernels did twelve per element).
__device__ inline void split4(int64_t i, const int64_t (&n)[4], int64_t (&id)[4]) {
  int64_t rem = i;
#pragma unroll
  for (int d = 3; d >= 0; --d) {
    if (n[d] == 1) {
      id[d] = 0;
    } else if (rem < (int64_t(1) << 31) && n[d] < (int64_t(1) << 31)) {
      const uint32_t r = (uint32_t)rem, m = (uint32_t)n[d];
      id[d] = r % m;
      rem = r / m;
    } else {
      id[d] = rem % n[d];
      rem /= n[d];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_field_gather(const T* __restrict__ src, T* __restrict__ out,
                                                        AccessArgs a) {
  const int64_t total = a.on[0] * a.on[1] * a.on[2] * a.on[3];
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += nthreads) {
    int64_t sidx = 0, stride = 1, id[4];
    bool zero = false;
    split4(i, a.on, id);
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      int64_t p = id[d] + a.shift[d];  // roll by -shift (core.py:963); shift reduced to [0, np) on the host
      if (p >= a.np[d]) p -= a.np[d];
      const int64_t q = p - a.pad[d];
      if (q < 0) zero = true;
      sidx += (q < 0 ? 0 : q) * stride;
      stride *= a.sn[d];
    }
    out[i] = zero ? T(0) : src[sidx];
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_field_scatter(const T* __restrict__ g, T* __restrict__ gsrc, AccessArgs a,
                                                         int accumulate) {
  const int64_t total = a.sn[0] * a.sn[1] * a.sn[2] * a.sn[3];
  const int64_t nthreads = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += nthreads) {
    int64_t oidx = 0, stride = 1, q[4];
    bool none = false;
    split4(i, a.sn, q);
#pragma unroll
    for (int d = 3; d >= 0; --d) {
      int64_t id = q[d] + a.pad[d] - a.shift[d];  // in (-np, np]
      if (id < 0) id += a.np[d];
      if (id >= a.np[d]) id -= a.np[d];
      if (id >= a.on[d]) none = true;  // trimmed away (core.py:965-969)
      oidx += (id >= a.on[d] ? 0 : id) * stride;
      stride *= a.on[d];
    }
    const T v = none ? T(0) : g[oidx];
    gsrc[i] = accumulate ? gsrc[i] + v : v;
  }
}

static int fill_access(AccessArgs& a, const int64_t* sshape, int ndim, const char* field_loc, const char* loc,
                       const int64_t* shift) {
  int fl[4], tl[4];
  if (ndim < 1 || ndim > ODIL_MAX_NDIM || !sshape || parse_loc(field_loc, ndim, fl) || parse_loc(loc, ndim, tl)) {
    set_error("field access: invalid ndim=%d, shape or loc", ndim);
    return ODIL_E_INVAL;
  }
  canon_shape(sshape, ndim, a.sn);
  for (int d = 0; d < 4; ++d) {
    const int i = d - (4 - ndim);
    if ((fl[d] == kNone) != (tl[d] == kNone) && i >= 0) {
      set_error("field access: loc '%s' -> '%s' mixes '.' with c/n", field_loc, loc);
      return ODIL_E_INVAL;
    }
    a.pad[d] = (fl[d] == kCell && tl[d] == kNode) ? 1 : 0;
    const int64_t trim = (fl[d] == kNode && tl[d] == kCell) ? 1 : 0;
    a.np[d] = a.sn[d] + a.pad[d];
    a.on[d] = a.np[d] - trim;
    if (a.sn[d] < 1 || a.on[d] < 1) {
      set_error("field access: empty extent on axis %d", d);
      return ODIL_E_INVAL;
    }
    const int64_t sh = (i >= 0 && shift) ? shift[i] : 0;
    a.shift[d] = ((sh % a.np[d]) + a.np[d]) % a.np[d];  // periodic roll: reduced to [0, np)
  }
  return 0;
}

template <typename T>
static int field_gather(const T* src, T* out, const int64_t* sshape, int ndim, const char* field_loc,
                        const char* loc, const int64_t* shift, void* stream) {
  AccessArgs a;
  if (int e = fill_access(a, sshape, ndim, field_loc, loc, shift)) return e;
  if (!src || !out) {
    set_error("field_gather: null pointer");
    return ODIL_E_INVAL;
  }
  hipLaunchKernelGGL(k_field_gather<T>, dim3(grid_flat(prod4(a.on), kBlock * 2)), dim3(kBlock), 0, (hipStream_t)stream,
                     src, out, a);
  return check_launch("k_field_gather");
}

template <typename T>
static int field_scatter(const T* g, T* gsrc, const int64_t* sshape, int ndim, const char* field_loc,
                         const char* loc, const int64_t* shift, int accumulate, void* stream) {
  AccessArgs a;
  if (int e = fill_access(a, sshape, ndim, field_loc, loc, shift)) return e;
  if (!g || !gsrc) {
    set_error("field_scatter: null pointer");
    return ODIL_E_INVAL;
  }
  hipLaunchKernelGGL(k_field_scatter<T>, dim3(grid_for(prod4(a.sn), kBlock * 2)), dim3(kBlock), 0,
                     (hipStream_t)stream, g, gsrc, a, accumulate);
  return check_launch("k_field_scatter");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_field_gather_f64(const double* src, double* out, const int64_t* sshape, int ndim, const char* field_loc,
                          const char* loc, const int64_t* shift, void* stream) {
  return field_gather<double>(src, out, sshape, ndim, field_loc, loc, shift, stream);
}
int odil_field_gather_f32(const float* src, float* out, const int64_t* sshape, int ndim, const char* field_loc,
                          const char* loc, const int64_t* shift, void* stream) {
  return field_gather<float>(src, out, sshape, ndim, field_loc, loc, shift, stream);
}
int odil_field_scatter_f64(const double* g, double* gsrc, const int64_t* sshape, int ndim, const char* field_loc,
                           const char* loc, const int64_t* shift, int accumulate, void* stream) {
  return field_scatter<double>(g, gsrc, sshape, ndim, field_loc, loc, shift, accumulate, stream);
}
int odil_field_scatter_f32(const float* g, float* gsrc, const int64_t* sshape, int ndim, const char* field_loc,
                           const char* loc, const int64_t* shift, int accumulate, void* stream) {
  return field_scatter<float>(g, gsrc, sshape, ndim, field_loc, loc, shift, accumulate, stream);
}
}
