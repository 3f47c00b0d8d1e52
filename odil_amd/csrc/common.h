// Shared host/device helpers for the gfx950 kernels of the ODIL hot path.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "odil_hip.h"

namespace odil {

// MI355X: 8 XCDs x 32 CUs; a launch needs >> 256 workgroups, and each XCD has a private L2.
constexpr int kNumXcd = 8;
constexpr int kBlock = 256;          // 4 waves of 64
constexpr int kMaxPartials = 4096;   // doubles of reduction scratch per reduced quantity
constexpr int kGridCap = 2048;       // 256 CUs x 8 resident workgroups

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t err = hipGetLastError();
  if (err != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(err));
    return ODIL_E_LAUNCH;
  }
  return 0;
}

// Canonical 4-D view (leading axes padded with size 1 / loc '.').
struct Dims4 {
  int64_t n[4];
};

enum Loc : int { kNone = 0, kCell = 1, kNode = 2 };

inline int parse_loc(const char* loc, int ndim, int out[4]) {
  if (!loc || (int)strlen(loc) != ndim) return ODIL_E_INVAL;
  for (int i = 0; i < 4; ++i) out[i] = kNone;
  for (int i = 0; i < ndim; ++i) {
    char c = loc[i];
    int v = c == 'c' ? kCell : c == 'n' ? kNode : c == '.' ? kNone : -1;
    if (v < 0) return ODIL_E_INVAL;
    out[4 - ndim + i] = v;
  }
  return 0;
}

inline void canon_shape(const int64_t* shape, int ndim, int64_t out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = 1;
  for (int i = 0; i < ndim; ++i) out[4 - ndim + i] = shape[i];
}

inline int64_t prod4(const int64_t n[4]) { return n[0] * n[1] * n[2] * n[3]; }

inline int grid_for(int64_t work_items, int per_block) {
  int64_t nb = (work_items + per_block - 1) / per_block;
  if (nb > kGridCap) nb = kGridCap;
  if (nb < 1) nb = 1;
  return (int)nb;
}

// ---------------------------------------------------------------------------
// XCD-aware row schedule.
//
// Work is a list of row segments (z, y, xs) of a (Z, Y, X) array.  Workgroup b is
// observed to run on XCD b % 8 (performance only, never correctness), so XCD k is
// given the y-chunk [k*Yc, (k+1)*Yc) of EVERY plane and walks it plane by plane:
// the y+-1 and z+-1 neighbours a stencil re-reads are then served by that XCD's own
// 4 MiB L2 instead of being fetched once per XCD.  Within an XCD the workgroups take
// items round-robin, so the items in flight are consecutive.  Each thread's running
// sums therefore accumulate in a fixed order: reductions are deterministic.
// ---------------------------------------------------------------------------
struct RowSched {
  int64_t Z, Y, XS;  // planes, rows per plane, x-segments per row
  int64_t Yc;        // rows per XCD chunk
};

inline RowSched make_sched(int64_t Z, int64_t Y, int64_t XS) {
  RowSched s;
  s.Z = Z;
  s.Y = Y;
  s.XS = XS;
  s.Yc = (Y + kNumXcd - 1) / kNumXcd;
  return s;
}

inline int sched_grid(const RowSched& s) {
  // Same number of workgroups per XCD; enough to cover the largest chunk, capped.
  int64_t per_xcd = s.Z * s.Yc * s.XS;
  int64_t cap = kGridCap / kNumXcd;
  if (per_xcd > cap) per_xcd = cap;
  if (per_xcd < 1) per_xcd = 1;
  return (int)(per_xcd * kNumXcd);
}

#ifdef __HIPCC__
struct RowIter {
  int64_t t, step, count, y0, ny;
};

__device__ inline RowIter sched_begin(const RowSched& s) {
  RowIter it;
  int k = blockIdx.x % kNumXcd;
  it.t = blockIdx.x / kNumXcd;
  it.step = gridDim.x / kNumXcd;
  it.y0 = (int64_t)k * s.Yc;
  int64_t y1 = it.y0 + s.Yc;
  if (y1 > s.Y) y1 = s.Y;
  it.ny = y1 > it.y0 ? y1 - it.y0 : 0;
  it.count = s.Z * it.ny * s.XS;
  return it;
}

__device__ inline void sched_decode(const RowSched& s, const RowIter& it, int64_t& z, int64_t& y, int64_t& xs) {
  int64_t per_plane = it.ny * s.XS;
  z = it.t / per_plane;
  int64_t r = it.t - z * per_plane;
  int64_t yl = r / s.XS;
  xs = r - yl * s.XS;
  y = it.y0 + yl;
}

// Block-wide sum of one double per thread; result valid on thread 0.
__device__ inline double block_sum(double v) {
  __shared__ double wave_sums[kBlock / 64];
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) wave_sums[wave] = v;
  __syncthreads();
  double total = 0;
  if (threadIdx.x == 0) {
    for (int w = 0; w < kBlock / 64; ++w) total += wave_sums[w];
  }
  return total;
}
#endif

// Final stage of every reduction: out[q] = scale * sum(partials[q*stride .. +count)).
template <typename T>
int launch_final_reduce(const double* partials, int count, int stride, int nq, double scale, T* out,
                        hipStream_t stream);

}  // namespace odil
