// Library plumbing: thread-local error string, device query, and the reductions
// (final stage shared by every reducing kernel; mean / mean-square, core.py:1093-1095).
#include <stdarg.h>

#include "common.h"

namespace odil {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_error, sizeof(g_error), fmt, ap);
  va_end(ap);
}

// out[q] = T( sum(partials[q*stride .. q*stride+count)) / denom ), fixed summation order.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_final_reduce(const double* __restrict__ partials, int count, int stride,
                                                        double denom, T* __restrict__ out) {
  const double* p = partials + (int64_t)blockIdx.x * stride;
  double local = 0.0;
  for (int i = threadIdx.x; i < count; i += kBlock) local += p[i];
  const double total = block_sum(local);
  if (threadIdx.x == 0) out[blockIdx.x] = T(total / denom);
}

template <typename T>
int launch_final_reduce(const double* partials, int count, int stride, int nq, double denom, T* out,
                        hipStream_t stream) {
  hipLaunchKernelGGL(k_final_reduce<T>, dim3(nq), dim3(kBlock), 0, stream, partials, count, stride, denom, out);
  return check_launch("k_final_reduce");
}
template int launch_final_reduce<double>(const double*, int, int, int, double, double*, hipStream_t);
template int launch_final_reduce<float>(const double*, int, int, int, double, float*, hipStream_t);

template <typename T>
__global__ __launch_bounds__(kBlock) void k_mean_partial(const T* __restrict__ x, int64_t n, int square,
                                                        double* __restrict__ partials) {
  // Contiguous chunk per workgroup, so the summation order does not depend on the grid stride.
  const int64_t per = (n + gridDim.x - 1) / gridDim.x;
  const int64_t lo = (int64_t)blockIdx.x * per;
  int64_t hi = lo + per;
  if (hi > n) hi = n;
  double local = 0.0;
  for (int64_t i = lo + threadIdx.x; i < hi; i += kBlock) {
    const T v = x[i];
    local += square ? (double)(v * v) : (double)v;
  }
  const double total = block_sum(local);
  if (threadIdx.x == 0) partials[blockIdx.x] = total;
}

template <typename T>
static int mean_reduce(const T* x, int64_t n, int square, double* partials, T* out, void* stream) {
  if (!x || !partials || !out || n < 1) {
    set_error("mean_reduce: null pointer or n=%lld < 1", (long long)n);
    return ODIL_E_INVAL;
  }
  const int grid = grid_for(n, kBlock * 8);
  hipLaunchKernelGGL(k_mean_partial<T>, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, x, n, square, partials);
  if (int e = check_launch("k_mean_partial")) return e;
  return launch_final_reduce<T>(partials, grid, 0, 1, (double)n, out, (hipStream_t)stream);
}

}  // namespace odil

using namespace odil;

extern "C" {
const char* odil_last_error(void) { return g_error; }
int odil_version(void) { return 100; }
int odil_device_count(void) {
  int n = 0;
  hipError_t err = hipGetDeviceCount(&n);
  if (err != hipSuccess) {
    set_error("hipGetDeviceCount: %s", hipGetErrorString(err));
    return ODIL_E_NODEV;
  }
  return n;
}
size_t odil_reduce_workspace_bytes(void) { return (size_t)kMaxPartials * sizeof(double); }
size_t odil_dots_workspace_bytes(int nvec) { return (size_t)(nvec < 1 ? 1 : nvec) * kDotPartials * sizeof(double); }

int odil_mean_reduce_f64(const double* x, int64_t n, int square, double* partials, double* out, void* stream) {
  return mean_reduce<double>(x, n, square, partials, out, stream);
}
int odil_mean_reduce_f32(const float* x, int64_t n, int square, double* partials, float* out, void* stream) {
  return mean_reduce<float>(x, n, square, partials, out, stream);
}
}
