// Device helpers shared by the z-marching transfer kernels (mg_march.hip) and the kernels that fuse
// the last prolongation into a stencil (poisson_synth.hip): wide packs, streaming accesses, and the
// ghosted coarse neighbourhood of the joint ghost rule (reference core.py:640-643).
#pragma once
#include "mg_transfer.h"

namespace odil {

template <typename T, int NV>
struct alignas(NV * sizeof(T)) PackN {
  T e[NV];
};

// touch-once streams (fine addend in, fine result out) bypass cache retention when the fine array is
// larger than the last-level cache could hand to the next kernel anyway (MarchArgs::nt)
template <typename T, int NV>
__device__ inline PackN<T, NV> stream_ld(const T* p, bool nt) {
  typedef T VT __attribute__((ext_vector_type(NV)));
  const VT v = nt ? __builtin_nontemporal_load(reinterpret_cast<const VT*>(p)) : *reinterpret_cast<const VT*>(p);
  PackN<T, NV> r;
#pragma unroll
  for (int k = 0; k < NV; ++k) r.e[k] = v[k];
  return r;
}
template <typename T, int NV>
__device__ inline void stream_st(T* p, const PackN<T, NV>& x, bool nt) {
  typedef T VT __attribute__((ext_vector_type(NV)));
  VT v;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = x.e[k];
  if (nt)
    __builtin_nontemporal_store(v, reinterpret_cast<VT*>(p));
  else
    *reinterpret_cast<VT*>(p) = v;
}

struct MarchArgs {
  int cn[3], fn[3];  // (z, y, x) coarse / fine extents
  int tx, ty;        // thread tile: tx column groups along x, ty rows along y
  int cut_lo, cut_hi;  // z end is an interior slab interface (ghost planes), not a wall
  int nt;              // stream the fine array past the caches (it exceeds kStreamBytes)
  int lead_loc, lead_cn, lead_fn;  // 4-D layouts: leading axis kind ('.' batch or 'n') and its extents
  int64_t lead_cstride;            // elements between leading indices of the COARSE array (its volume unless a view)
  UnitSched usched;
};

// Clamp / reflect indices of the N coarse positions j0-1 .. j0+N-2 on a 'c' axis of n cells.
template <int N>
struct TapN {
  int cl[N], rf[N];
  bool out[N];
};

template <int N>
__device__ inline TapN<N> tapn(int j0, int n) {
  TapN<N> t;
#pragma unroll
  for (int d = 0; d < N; ++d) {
    const int q = j0 + d - 1;
    t.out[d] = q < 0 || q >= n;
    t.cl[d] = q < 0 ? 0 : (q >= n ? n - 1 : q);
    t.rf[d] = q < 0 ? 1 : (q >= n ? n - 2 : q);
  }
  return t;
}

// Coarse values of (padded) plane q in [-1, n] around the owned columns: ghosts by the joint rule
// (core.py:640-643), 2 u[clamp] - u[reflect].  Branch-free: the clamp AND the reflect value of every
// position are loaded (the same address away from the walls, so the second load is a cache hit) and
// the ghost is selected afterwards -- a branch around the second load makes the compiler drain all
// outstanding loads (s_waitcnt vmcnt(0)) nine times per plane, serialising the HBM streams of the
// step behind these cache hits.
template <typename T, int CX>
__device__ inline void load_plane(const T* __restrict__ coarse, int q, int cnz, int64_t cplane, int cnx,
                                  const TapN<3>& ty, const TapN<CX + 2>& tx, T cscale, T (&v)[3][CX + 2]) {
  const bool oz = q < 0 || q >= cnz;
  const int zcl = q < 0 ? 0 : (q >= cnz ? cnz - 1 : q);
  const int zrf = q < 0 ? 1 : (q >= cnz ? cnz - 2 : q);
  const T* ccl = coarse + zcl * cplane;
  const T* crf = coarse + zrf * cplane;
  T cl[3][CX + 2], rf[3][CX + 2];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < CX + 2; ++dx) {
      cl[dy][dx] = ccl[(int64_t)ty.cl[dy] * cnx + tx.cl[dx]];
      rf[dy][dx] = crf[(int64_t)ty.rf[dy] * cnx + tx.rf[dx]];
    }
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < CX + 2; ++dx) {
      const T val = cscale * cl[dy][dx];
      const T ghost = T(2) * val - cscale * rf[dy][dx];
      v[dy][dx] = (oz || ty.out[dy] || tx.out[dx]) ? ghost : val;
    }
}

#ifndef ODIL_PLANE_DPP
#define ODIL_PLANE_DPP 1
#endif

// The same neighbourhood with ONE load per row and source (clamp / reflect) for the lane's own columns;
// the columns left and right of them are the own columns of the adjacent lanes (wave-wide DPP shift).
// load_plane issues 2 x 3 x (CX + 2) one-element loads per plane -- with two coarse volumes per fine
// volume (node-centred leading axis) 48 of them per step, which bounds the kernel by the number of memory
// instructions (1.5 TB/s), not by bytes.  Lanes at the ends of a row segment (and at the walls, where the
// ghost rule picks other columns) read their outer columns from memory as before.  `ntx`: lanes per row
// segment (MarchArgs::tx), jx0: first own column.  4-D tracer (32 x 256^3, four 'nccc' fields): 60.0 -> 56.5
// ms / epoch.  Not used by the fused prolongation + residual kernel (poisson_synth.hip): at 225 VGPRs the
// extra live values cost it 0.66 -> 0.74 ms.
template <typename T, int CX>
__device__ inline void load_plane_shared(const T* __restrict__ coarse, int q, int cnz, int64_t cplane, int cnx,
                                         const TapN<3>& ty, const TapN<CX + 2>& tx, T cscale, T (&v)[3][CX + 2],
                                         int ntx, int jx0) {
#if ODIL_PLANE_DPP
  const bool oz = q < 0 || q >= cnz;
  const int zcl = q < 0 ? 0 : (q >= cnz ? cnz - 1 : q);
  const int zrf = q < 0 ? 1 : (q >= cnz ? cnz - 2 : q);
  const T* ccl = coarse + zcl * cplane;
  const T* crf = coarse + zrf * cplane;
  const int lx = threadIdx.x % ntx, lane = threadIdx.x & 63;
  const bool lo_edge = lx == 0 || lane == 0 || jx0 == 0;
  const bool hi_edge = lx == ntx - 1 || lane == 63 || jx0 + CX >= cnx;
  T cl[3][CX + 2], rf[3][CX + 2];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const T* rc = ccl + (int64_t)ty.cl[dy] * cnx + jx0;  // own columns are inside the array: clamp == reflect == jx0 + c
    const T* rr = crf + (int64_t)ty.rf[dy] * cnx + jx0;
#pragma unroll
    for (int c = 0; c < CX; ++c) {
      cl[dy][1 + c] = rc[c];
      rf[dy][1 + c] = rr[c];
    }
  }
  if (lo_edge) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      cl[dy][0] = ccl[(int64_t)ty.cl[dy] * cnx + tx.cl[0]];
      rf[dy][0] = crf[(int64_t)ty.rf[dy] * cnx + tx.rf[0]];
    }
  }
  if (hi_edge) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      cl[dy][CX + 1] = ccl[(int64_t)ty.cl[dy] * cnx + tx.cl[CX + 1]];
      rf[dy][CX + 1] = crf[(int64_t)ty.rf[dy] * cnx + tx.rf[CX + 1]];
    }
  }
#pragma unroll
  for (int dy = 0; dy < 3; ++dy) {
    const T pc = from_prev_lane(cl[dy][CX]), pr = from_prev_lane(rf[dy][CX]);
    const T nc = from_next_lane(cl[dy][1]), nr = from_next_lane(rf[dy][1]);
    if (!lo_edge) cl[dy][0] = pc, rf[dy][0] = pr;
    if (!hi_edge) cl[dy][CX + 1] = nc, rf[dy][CX + 1] = nr;
  }
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < CX + 2; ++dx) {
      const T val = cscale * cl[dy][dx];
      const T ghost = T(2) * val - cscale * rf[dy][dx];
      v[dy][dx] = (oz || ty.out[dy] || tx.out[dx]) ? ghost : val;
    }
#else
  load_plane<T, CX>(coarse, q, cnz, cplane, cnx, ty, tx, cscale, v);
#endif
}

}  // namespace odil
