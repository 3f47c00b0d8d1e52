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
constexpr int kMaxPartials = 65536;  // doubles of reduction scratch per reduced quantity
constexpr int kDotPartials = 1024;   // partial sums per vector in odil_dots
// arrays above this size are streamed with non-temporal accesses; smaller ones are left to the
// 256 MB last-level cache, where the consumer kernel of the same epoch finds them
constexpr int64_t kStreamBytes = (int64_t)128 << 20;
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

// Arrays are handled in a canonical 4-D view: leading axes padded with size 1 / loc '.'.
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

// Flat streaming kernels (optimizer update, axpy, scale ...) get ONE chunk per workgroup: with the grid capped at
// kGridCap and a grid-stride loop k_adam moved 4.4-4.9 TB/s (f64) / 4.8-5.5 (f32), uncapped 5.6-6.1 / 5.2-6.1 on the same
// boxes (the streams of a looping workgroup are gridDim x 4 KB apart; tools/mb_tile_traffic.hip shows the same
// for a bare copy).
inline int flat_grid_cap() { return 1 << 20; }

inline int grid_flat(int64_t work_items, int per_block) {
  int64_t nb = (work_items + per_block - 1) / per_block;
  if (nb > flat_grid_cap()) nb = flat_grid_cap();
  if (nb < 1) nb = 1;
  return (int)nb;
}

// (reductions keep the capped grid: their partial sums live in a workspace of kMaxPartials entries and their
// summation order is part of the reproducibility contract)
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
  int Z, Y, XS;  // planes, rows per plane, x-segments per row
  int Yc;        // rows per XCD chunk
};

inline RowSched make_sched(int64_t Z, int64_t Y, int64_t XS) {
  RowSched s;
  s.Z = (int)Z;
  s.Y = (int)Y;
  s.XS = (int)XS;
  s.Yc = (int)((Y + kNumXcd - 1) / kNumXcd);
  return s;
}

// Items must be countable in 31 bits (checked by the host wrappers via sched_ok).
inline bool sched_ok(int64_t Z, int64_t Y, int64_t XS) { return Z * Y * XS < (int64_t)1 << 31; }

inline int sched_grid(const RowSched& s) {
  // Same number of workgroups per XCD; enough to cover the largest chunk, capped.
  int64_t per_xcd = (int64_t)s.Z * s.Yc * s.XS;
  int64_t cap = kGridCap / kNumXcd;
  if (per_xcd > cap) per_xcd = cap;
  if (per_xcd < 1) per_xcd = 1;
  return (int)(per_xcd * kNumXcd);
}

// ---------------------------------------------------------------------------
// One-unit-per-workgroup schedule for the z-marching kernels.
//
// A unit is (zc, y, xs): a z-chunk of planes, one row (or row tile), one x-segment.  The
// workgroup keeps the z-1 / z / z+1 values of its column in registers while it marches, so
// z re-reads never leave the CU; rows y+-1 belong to workgroups of the SAME XCD (XCD k owns
// y-chunk k) that march in step, so they are L2 hits.  When Y is too small to split eight
// ways the units are split in their (zc, y, xs) order instead; tiny problems fall back to plain order.
// ---------------------------------------------------------------------------
struct UnitSched {
  int ZCH, Y, XS;   // number of z-chunks, rows, x-segments
  int ZC;           // planes per z-chunk
  int axis;         // 1: XCD-chunk along Y, 0: along ZCH, -1: none
  int chunk;        // chunk length along `axis`
  int per_xcd;      // units per XCD (grid = 8 * per_xcd), or total units when axis < 0
};

// `target_units`: how many workgroups the launch should at least be cut into.  Kernels that prime a
// window of planes before their first step (P^T: two extra fine-plane reductions per chunk) ask for
// fewer, longer chunks.
inline UnitSched make_unit_sched(int64_t Z, int64_t Y, int64_t XS, int64_t target_units = 2 * kGridCap) {
  UnitSched s;
  int64_t zc = (Z * Y * XS) / target_units;
  if (zc < 1) zc = 1;
  if (zc > 64) zc = 64;
  if (zc > Z) zc = Z;
  s.ZC = (int)zc;
  s.ZCH = (int)((Z + zc - 1) / zc);
  s.Y = (int)Y;
  s.XS = (int)XS;
  if (Y >= 4 * kNumXcd) {
    s.axis = 1;
    s.chunk = (int)((Y + kNumXcd - 1) / kNumXcd);
    s.per_xcd = s.ZCH * s.chunk * s.XS;
  } else if (s.ZCH >= kNumXcd) {
    // contiguous ranges of the (zc, y, xs) order, equal to within one unit: whole z-chunks per XCD left two
    // XCDs idle when their number was not a multiple of eight (17 chunks -> 3, 3, 3, 3, 3, 2, 0, 0)
    s.axis = 0;
    s.chunk = 0;
    s.per_xcd = (int)(((int64_t)s.ZCH * s.Y * s.XS + kNumXcd - 1) / kNumXcd);
  } else {
    s.axis = -1;
    s.chunk = 0;
    s.per_xcd = s.ZCH * s.Y * s.XS;
  }
  return s;
}

inline int unit_grid(const UnitSched& s) { return s.axis < 0 ? s.per_xcd : s.per_xcd * kNumXcd; }

#ifdef __HIPCC__
// Returns false when this workgroup has no unit (padding of an uneven chunk).
__device__ inline bool unit_decode(const UnitSched& s, int& zc, int& y, int& xs) {
  if (s.axis < 0) {
    const int i = blockIdx.x;
    xs = i % s.XS;
    const int r = i / s.XS;
    y = r % s.Y;
    zc = r / s.Y;
    return true;
  }
  const int k = blockIdx.x % kNumXcd, i = blockIdx.x / kNumXcd;
  xs = i % s.XS;
  const int r = i / s.XS;
  if (s.axis == 1) {
    y = k * s.chunk + r % s.chunk;
    zc = r / s.chunk;
    return y < s.Y;
  }
  // axis 0: XCD k owns units [k per_xcd, (k + 1) per_xcd) of the zc-major order
  const int u = k * s.per_xcd + i;
  xs = u % s.XS;
  const int ru = u / s.XS;
  y = ru % s.Y;
  zc = ru / s.Y;
  return zc < s.ZCH;
}
#endif

#ifdef __HIPCC__
struct RowIter {
  int t, step, count, y0, ny;
};

__device__ inline RowIter sched_begin(const RowSched& s) {
  RowIter it;
  const int k = blockIdx.x % kNumXcd;
  it.t = blockIdx.x / kNumXcd;
  it.step = gridDim.x / kNumXcd;
  it.y0 = k * s.Yc;
  int y1 = it.y0 + s.Yc;
  if (y1 > s.Y) y1 = s.Y;
  it.ny = y1 > it.y0 ? y1 - it.y0 : 0;
  it.count = s.Z * it.ny * s.XS;
  return it;
}

template <typename I>
__device__ inline void sched_decode(const RowSched& s, const RowIter& it, I& z, I& y, I& xs) {
  const int per_plane = it.ny * s.XS;
  const int zi = it.t / per_plane;
  const int r = it.t - zi * per_plane;
  const int yl = r / s.XS;
  z = zi;
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

#ifdef __HIPCC__
// Value held by the previous / next lane of the wavefront (wave-wide DPP shift: no LDS, no memory).
// Undefined for lane 0 / lane 63, which read their x neighbour from memory instead.  The x-1 / x+1
// neighbours of a lane's pack are the edge values of the adjacent lanes' packs: taking them from
// registers removes two half-efficiency (one element per lane) loads per plane from kernels whose
// limit is the number of memory instructions, not HBM.
__device__ inline int lane_shift_word(int v, bool up) {
  return up ? __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false)   // wave_shr:1 -> from lane - 1
            : __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);  // wave_shl:1 -> from lane + 1
}
__device__ inline float from_prev_lane(float v) { return __int_as_float(lane_shift_word(__float_as_int(v), true)); }
__device__ inline float from_next_lane(float v) { return __int_as_float(lane_shift_word(__float_as_int(v), false)); }
__device__ inline double from_prev_lane(double v) {
  const int lo = lane_shift_word(__double2loint(v), true), hi = lane_shift_word(__double2hiint(v), true);
  return __hiloint2double(hi, lo);
}
__device__ inline double from_next_lane(double v) {
  const int lo = lane_shift_word(__double2loint(v), false), hi = lane_shift_word(__double2hiint(v), false);
  return __hiloint2double(hi, lo);
}
#endif

// Optional Adam update fused into the kernel that FORMS a gradient array (the lane that writes
// g[i] also owns x[i], m[i], v[i]): saves re-reading g in a separate optimizer launch.
template <typename T>
struct AdamArgs {
  T* x;
  T* m;
  T* v;
  T alpha, omb1, omb2, eps;
  const T* alpha_dev;  // optional: step size read from device memory (hipGraph replay of an epoch)
};

#ifdef __HIPCC__
template <typename T>
__device__ inline void adam_update(T& x, T& m, T& v, T g, const AdamArgs<T>& a) {
  // reference optimizer.py:316-318
  m = m + (g - m) * a.omb1;
  v = v + (g * g - v) * a.omb2;
  const T alpha = a.alpha_dev ? *a.alpha_dev : a.alpha;
  x = x - (m * alpha) / (sqrt(v) + a.eps);
}
#endif

// Plain Adam launch on a flat range (optim.hip), for paths that cannot fuse it.
template <typename T>
int adam_launch(T* x, T* m, T* v, const T* g, int64_t n, T alpha, T omb1, T omb2, T eps, hipStream_t stream,
                const T* alpha_dev = nullptr);

// Final stage of every reduction: out[q] = scale * sum(partials[q*stride .. +count)).
template <typename T>
int launch_final_reduce(const double* partials, int count, int stride, int nq, double scale, T* out,
                        hipStream_t stream);

}  // namespace odil
