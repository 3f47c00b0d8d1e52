// WHOLE EPOCHS of the multigrid Poisson problem in ONE launch, for states that fit one workgroup's reach (1-D and 2-D
// grids: the sizes the reference's own examples run at, examples/poisson/poisson.py:133-147 -- 1-D N = 256 is 510
// unknowns in 8 levels).  As separate kernels an epoch of that problem is 17 dependent launches (7 prolongations, the
// residual and its two-stage reduction, the adjoint, 7 transposed prolongations with their Adam updates) of a few hundred
// cells each: 121 us eager, 58 us replayed as a hipGraph -- launch latency, not work.  Here ONE workgroup of 1024 threads
// walks the same phases with a barrier where a launch boundary used to be, for E epochs in a row, the state resident in
// LDS when it fits:
//   synthesis   u = w_0 + P(w_1 + P(...))                        reference core.py:245-263, 606-700
//   residual    fu = Lap(u) - rhs, loss = mean(fu^2)             examples/poisson/poisson.py:57-113, core.py:1093-1095
//   adjoint     g_0 = (2 / n) A^T fu, Adam on level 0            core.py:1100, optimizer.py:311-319
//   transposes  g_l = P^T g_{l-1}, Adam on level l
// Every phase repeats the ARITHMETIC of the kernel it replaces, operation for operation -- k_interp_add_fast /
// k_interp_adj_fast (mg_fast.hip), k_poisson_residual / k_poisson_adjoint (poisson.hip) with axis_term / adj_axis /
// adam_update from the shared headers -- and the loss is summed in the order of the two-stage reduction (per-thread running
// sums of the residual kernel's virtual workgroups, wave shuffle tree, four wave sums, k_final_reduce's tree over the
// partials), so the trajectory is BIT-IDENTICAL to the multi-launch path (tests/test_trajectories.py).
#include "mg_transfer.h"
#include "poisson.h"

namespace odil {

constexpr int kSmallThreads = 1024;  // (largest launch; tiny problems run with 256: fewer waves at every barrier)
constexpr int kSmallMaxLev = 12;

struct SmallLevel {
  int nz, nx;     // canonical (Z, X): 1-D arrays have nz = 1
  int xshift;     // log2(nx) when nx is a power of two, else -1
  float rnx;      // 1 / nx
  int size;
  int off;        // offset of the level in the packed vectors (x, m, v, g) and in the synthesis scratch
};

template <typename T>
struct SmallArgs {
  int nlvl, ndim, nepochs;
  SmallLevel lv[kSmallMaxLev];
  int total;          // unknowns of all levels
  T scale;            // 2 / n, the cotangent of the mean square
  T omb1, omb2, eps;  // Adam: 1 - beta_1, 1 - beta_2, epsilon (outside the square root: Keras convention)
  double denom;       // cells of the finest level
  UnitSched usched;   // the residual kernel's unit schedule (poisson.hip: fill_args) -- the loss is summed in ITS order
  int grid;           // its workgroups
  int lds;            // the state is staged in LDS
};

// i / nx without the integer divide (i < 2^22: the float quotient is off by at most one)
__device__ __forceinline__ int small_row(int i, const SmallLevel& L) {
  if (L.xshift >= 0) return i >> L.xshift;
  int q = (int)((float)i * L.rnx);
  q = q * L.nx > i ? q - 1 : q;
  q = (q + 1) * L.nx <= i ? q + 1 : q;
  return q;
}

// ---- P: k_interp_add_fast<T, YC> for one coarse cell (jy, jx): the 2 (x 2) fine cells it owns --------------------------
template <typename T>
__device__ __forceinline__ void small_interp_add(const T* coarse, const T* add, T* fine, int cny, int cnx, bool yc, int jy,
                                                 int jx) {
  const int fnx = 2 * cnx;
  int xcl[3], xrf[3], ycl[3], yrf[3];
  bool xo[3], yo[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const int q = jx + d - 1;
    xo[d] = q < 0 || q >= cnx;
    xcl[d] = q < 0 ? 0 : (q >= cnx ? cnx - 1 : q);
    xrf[d] = q < 0 ? 1 : (q >= cnx ? cnx - 2 : q);
    const int r = jy + d - 1;
    yo[d] = yc && (r < 0 || r >= cny);
    ycl[d] = !yc ? jy : (r < 0 ? 0 : (r >= cny ? cny - 1 : r));
    yrf[d] = !yc ? jy : (r < 0 ? 1 : (r >= cny ? cny - 2 : r));
  }
  T v[3][3];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      if (!yc && dy != 1) {
        v[dy][dx] = T(0);
        continue;
      }
      T val = T(1) * coarse[ycl[dy] * cnx + xcl[dx]];
      if (yo[dy] || xo[dx]) val = T(2) * val - T(1) * coarse[yrf[dy] * cnx + xrf[dx]];
      v[dy][dx] = val;
    }
  const T wl1 = T(1), wl3 = T(3), wl9 = T(9);
  const T rdenom = T(1) / T(yc ? 16 : 4);
#pragma unroll
  for (int sy = 0; sy < 2; ++sy) {
    if (!yc && sy) continue;
    T o[2];
#pragma unroll
    for (int sx = 0; sx < 2; ++sx) {
      T s = T(0);
#pragma unroll
      for (int ry = 0; ry < 2; ++ry)
#pragma unroll
        for (int rx = 0; rx < 2; ++rx) {
          if (!yc && ry) continue;
          const int wy = yc ? (sy == ry ? 1 : 3) : 1;
          const int wx = sx == rx ? 1 : 3;
          const int dy = yc ? sy + ry : 1;
          const int dx = sx + rx;
          const int ww = wy * wx;
          s = s + (ww == 1 ? wl1 : (ww == 3 ? wl3 : wl9)) * v[dy][dx];
        }
      o[sx] = s * rdenom;
    }
    const int fy = yc ? 2 * jy + sy : jy;
    const int off = fy * fnx + 2 * jx;
    fine[off] = T(1) * add[off] + o[0];
    fine[off + 1] = T(1) * add[off + 1] + o[1];
  }
}

// 1-D adjoint weights of k_interp_adj_fast (mg_fast.hip: adj6)
struct SmallAdj6 {
  float wc[6], wr[6];
  bool special;
};

__device__ __forceinline__ SmallAdj6 small_adj6(int J, int n) {
  SmallAdj6 t;
  const int F = 2 * n;
  const bool c_lo = J == 0, c_hi = J == n - 1, r_lo = J == 1, r_hi = J == n - 2;
  t.special = c_lo || c_hi || r_lo || r_hi;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int k = 2 * J - 2 + i;
    const float w = w_cell(J, k, F), lo = w_cell(-1, k, F), hi = w_cell(n, k, F);
    t.wc[i] = w + (c_lo ? lo : 0.f) + (c_hi ? hi : 0.f);
    t.wr[i] = w + (r_lo ? lo : 0.f) + (r_hi ? hi : 0.f);
  }
  return t;
}

// ---- P^T: k_interp_adj_fast<T, YC> for one coarse cell ------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T small_interp_adj(const T* gfine, int cny, int cnx, bool yc, int jy, int jx) {
  const int fnx = 2 * cnx;
  const SmallAdj6 ax = small_adj6(jx, cnx);
  SmallAdj6 ay;
  if (yc) {
    ay = small_adj6(jy, cny);
  } else {
    ay.special = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) ay.wc[i] = ay.wr[i] = i == 2 ? 1.f : 0.f;
  }
  const bool special = ax.special || ay.special;
  T sc = T(0), sr = T(0);
  const T wcl = T(1.f * 1.f), wrl = T(1.f * 1.f);
#pragma unroll
  for (int iy = 0; iy < 6; ++iy) {
    if (!yc && iy != 2) continue;
    if (ay.wc[iy] == 0.f && ay.wr[iy] == 0.f) continue;
    const int fy = yc ? 2 * jy - 2 + iy : jy;
    const T* row = gfine + fy * fnx;
    T g[6];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int fx = 2 * (jx - 1 + q);
      if (fx >= 0 && fx < fnx) {
        g[2 * q] = row[fx];
        g[2 * q + 1] = row[fx + 1];
      } else {
        g[2 * q] = g[2 * q + 1] = T(0);
      }
    }
    T rc = T(0), rr = T(0);
    if (!ax.special) {
      rc = (T(0.25) * g[1] + T(0.75) * g[2]) + (T(0.75) * g[3] + T(0.25) * g[4]);
      rr = rc;
    } else {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        rc = rc + T(ax.wc[i]) * g[i];
        rr = rr + T(ax.wr[i]) * g[i];
      }
    }
    sc = sc + (wcl * T(ay.wc[iy])) * rc;
    if (special) sr = sr + (wrl * T(ay.wr[iy])) * rr;
  }
  return special ? T(2) * sc - sr : sc;
}

// bytes of the LDS-resident state (x, m, v, g = u of all levels; fu, rhs of the finest), before the reduction workspace
template <typename T>
__host__ __device__ inline size_t small_state_bytes(int total, int n0) {
  return ((size_t)(4 * (size_t)total + 2 * (size_t)n0) * sizeof(T) + 15) / 16 * 16;
}

constexpr size_t kSmallLdsBytes = 160 * 1024 - 512;  // a workgroup's LDS on gfx950, less the static arrays

// the wave-level stage of block_sum (common.h): lane 0 holds the sum
__device__ __forceinline__ double small_wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// LDS: the state lives in shared memory for the whole launch -- a separate instantiation, so that its pointers are
// LDS pointers to the compiler (ds_read / ds_write) and not generic ones (flat accesses cost a phase ~3x the latency)
template <typename T, bool LDS>
__global__ __launch_bounds__(kSmallThreads) void k_poisson_small_epochs(T* xg, T* mg, T* vg, T* gg, T* ug, T* fug,
                                                                        const T* rhsg, const T* __restrict__ alphas,
                                                                        T* __restrict__ losses, T* __restrict__ norms,
                                                                        double* partials,
                                                                        SmallArgs<T> a, H2<T> h) {
  constexpr int V = VecOf<T>::N;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ double wave_sums[kSmallThreads / 64];
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int n0 = a.lv[0].size;
  T *x, *m, *v, *g, *u, *fu;
  const T* rhs;
  double* part;
  if constexpr (LDS) {
    // layout: x | m | v | g = u | fu | rhs | partials.  (The synthesis scratch u is dead once the residual is formed and
    // the gradient is written after that: one array serves both.)
    T* s = reinterpret_cast<T*>(smem_raw);
    x = s, m = s + a.total, v = s + 2 * a.total, g = s + 3 * a.total, u = g;
    fu = s + 4 * a.total;
    T* r = fu + n0;
    part = reinterpret_cast<double*>(smem_raw + small_state_bytes<T>(a.total, n0));
    for (int i = tid; i < a.total; i += nthr) x[i] = xg[i], m[i] = mg[i], v[i] = vg[i];
    for (int i = tid; i < n0; i += nthr) r[i] = rhsg[i];
    rhs = r;
    __syncthreads();
  } else {
    x = xg, m = mg, v = vg, g = gg, u = ug, fu = fug, rhs = rhsg, part = partials;
  }
  const bool yc = a.ndim == 2;
  const int Z = a.lv[0].nz, X = a.lv[0].nx;
  for (int e = 0; e < a.nepochs; ++e) {
    AdamArgs<T> ad{nullptr, nullptr, nullptr, alphas[e], a.omb1, a.omb2, a.eps, nullptr};
    // ---- synthesis: res_{L-1} = w_{L-1}; res_l = w_l + P res_{l+1} (mg_transfer.hip: mg_synth) ----------------------------
    const T* field = x;  // one level: u is the unknown itself
    for (int l = a.nlvl - 2; l >= 0; --l) {
      const SmallLevel& C = a.lv[l + 1];
      const T* coarse = l == a.nlvl - 2 ? x + C.off : u + C.off;
      T* out = u + a.lv[l].off;
      for (int i = tid; i < C.size; i += nthr) {
        const int jy = small_row(i, C), jx = i - jy * C.nx;
        small_interp_add<T>(coarse, x + a.lv[l].off, out, C.nz, C.nx, yc, jy, jx);
      }
      __syncthreads();
      field = u;
    }
    // ---- residual + loss in the order of k_poisson_residual's workgroups (poisson.hip) -----------------------------------
    // a task = one WAVE of one virtual workgroup (its running sums, its shuffle tree); the waves of this launch take them
    // in turn without meeting, then every virtual workgroup's four wave sums are added in block_sum's order
    {
      const int W = nthr >> 6, wv = tid >> 6, ln = tid & 63;
      double* vws = part + a.grid;  // [virtual workgroup][wave]
      const UnitSched& s = a.usched;
      // (the waves of a virtual workgroup beyond the end of its row segment hold zeros -- sum +0.0, written without being
      // walked; the tasks are the waves that do reach into the row, so that they spread over the waves of this launch)
      const int aw = (((X < kBlock * V ? X : kBlock * V) + 64 * V - 1) / (64 * V));  // waves per virtual workgroup with cells
      for (int i = tid; i < a.grid * (kBlock / 64); i += nthr)
        if (i % (kBlock / 64) >= aw) vws[i] = 0.0;
      for (int at = wv; at < a.grid * aw; at += W) {
        const int vb = at / aw, task = vb * (kBlock / 64) + at % aw, vt = (at % aw) * 64 + ln;
        double local = 0.0;
        // unit_decode for workgroup vb
        int zc, xs;
        bool have = true;
        if (s.axis < 0) {
          xs = vb % s.XS;
          zc = (vb / s.XS) / s.Y;
        } else {
          const int k = vb % kNumXcd, i = vb / kNumXcd;
          const int un = k * s.per_xcd + i;  // (axis 0: Y = 1 rules out the y-chunk schedule)
          xs = un % s.XS;
          zc = (un / s.XS) / s.Y;
          have = zc < s.ZCH;
        }
        const int x0 = (xs * kBlock + vt) * V;
        if (have && x0 < X) {
          const int valid = X - x0 < V ? X - x0 : V;
          const int z0 = zc * s.ZC, z1 = z0 + s.ZC < Z ? z0 + s.ZC : Z;
          for (int z = z0; z < z1; ++z) {
            const int pz = z * X;
            const int zm = (z == 0 ? Z - 1 : z - 1) * X, zp = (z == Z - 1 ? 0 : z + 1) * X;
            for (int i = 0; i < valid; ++i) {
              const int xx = x0 + i;
              const T q = field[pz + xx];
              const T xm = field[pz + (xx == 0 ? X - 1 : xx - 1)], xp = field[pz + (xx == X - 1 ? 0 : xx + 1)];
              T acc;
              const T tx = axis_term<T>(q, xm, xp, xx == 0, xx == X - 1, h, 2);
              if (yc) {
                acc = axis_term<T>(q, field[zm + xx], field[zp + xx], z == 0, z == Z - 1, h, 0);
                acc = acc + tx;
              } else {
                acc = tx;
              }
              const T f = acc - rhs[pz + xx];
              fu[pz + xx] = f;
              local += (double)(f * f);
            }
          }
        }
        const double ws = small_wave_sum(local);
        if (ln == 0) vws[task] = ws;
      }
      __syncthreads();
      for (int vb = tid; vb < a.grid; vb += nthr) {
        double total = 0;
        for (int w = 0; w < kBlock / 64; ++w) total += vws[vb * (kBlock / 64) + w];
        part[vb] = total;
      }
      __syncthreads();
    }
    // k_final_reduce: 256 threads stride over the partials, block_sum, / denom
    {
      double local = 0.0;
      if (tid < kBlock)
        for (int i = tid; i < a.grid; i += kBlock) local += part[i];
      const double ws = small_wave_sum(local);
      if ((tid & 63) == 0) wave_sums[tid >> 6] = ws;
      __syncthreads();
      if (tid == 0) {
        double total = 0;
        for (int w = 0; w < kBlock / 64; ++w) total += wave_sums[w];
        const T loss = T(total / a.denom);
        losses[e] = loss;
        norms[e] = sqrt(loss);  // (what the report calls the residual norm, core.py:1095)
      }
    }
    // ---- adjoint + Adam of level 0 (k_poisson_adjoint) ---------------------------------------------------------------------
    for (int i = tid; i < n0; i += nthr) {
      const int z = small_row(i, a.lv[0]), xx = i - z * X;
      const int pz = z * X;
      const T fb = a.scale * fu[i];
      T gv = T(0);
      if (yc) {
        const T fm = a.scale * fu[(z == 0 ? Z - 1 : z - 1) * X + xx], fp = a.scale * fu[(z == Z - 1 ? 0 : z + 1) * X + xx];
        gv = gv + adj_axis<T>(fb, fm, fp, z, Z, h, 0);
      }
      const T xm = a.scale * fu[pz + (xx == 0 ? X - 1 : xx - 1)], xp = a.scale * fu[pz + (xx == X - 1 ? 0 : xx + 1)];
      gv = gv + adj_axis<T>(fb, xm, xp, xx, X, h, 2);
      g[i] = gv;
      T xv = x[i], mv = m[i], vv = v[i];
      adam_update<T>(xv, mv, vv, gv, ad);
      x[i] = xv, m[i] = mv, v[i] = vv;
    }
    __syncthreads();
    // ---- transposes + Adam of the coarser levels (k_interp_adj_fast) -----------------------------------------------------
    for (int l = 1; l < a.nlvl; ++l) {
      const SmallLevel& C = a.lv[l];
      const T* gfine = g + a.lv[l - 1].off;
      for (int i = tid; i < C.size; i += nthr) {
        const int jy = small_row(i, C), jx = i - jy * C.nx;
        const T gv = small_interp_adj<T>(gfine, C.nz, C.nx, yc, jy, jx);
        const int ci = C.off + i;
        g[ci] = gv;
        T xv = x[ci], mv = m[ci], vv = v[ci];
        adam_update<T>(xv, mv, vv, gv, ad);
        x[ci] = xv, m[ci] = mv, v[ci] = vv;
      }
      __syncthreads();
    }
  }
  if constexpr (LDS) {
    for (int i = tid; i < a.total; i += nthr) xg[i] = x[i], mg[i] = m[i], vg[i] = v[i], gg[i] = g[i];
    for (int i = tid; i < n0; i += nthr) fug[i] = fu[i];
  }
}

template <typename T>
static int poisson_small_epochs(T* x, T* m, T* v, T* g, T* u, T* fu, const T* rhs, const int64_t* shapes, int nlvl,
                                int ndim, const T* h2, const T* alphas, int nepochs, T omb1, T omb2, T eps, T* losses,
                                T* norms, double* partials, void* stream) {
  if (!x || !m || !v || !g || !u || !fu || !rhs || !shapes || !h2 || !alphas || !losses || !norms || !partials) {
    set_error("poisson_small_epochs: null pointer");
    return ODIL_E_INVAL;
  }
  if (ndim < 1 || ndim > 2 || nlvl < 1 || nlvl > kSmallMaxLev || nepochs < 1) {
    set_error("poisson_small_epochs: ndim %d (1 or 2), %d levels (<= %d), %d epochs", ndim, nlvl, kSmallMaxLev, nepochs);
    return ODIL_E_INVAL;
  }
  SmallArgs<T> a;
  a.nlvl = nlvl, a.ndim = ndim, a.nepochs = nepochs;
  int off = 0;
  for (int l = 0; l < kSmallMaxLev; ++l) {
    SmallLevel& L = a.lv[l];
    L.nz = L.nx = 1, L.size = 0, L.off = 0, L.xshift = 0, L.rnx = 1.0f;
    if (l >= nlvl) continue;
    const int64_t nz = ndim == 2 ? shapes[l * ndim] : 1, nx = shapes[l * ndim + ndim - 1];
    if (nz < (ndim == 2 ? 2 : 1) || nx < 2 || nz * nx > (1 << 22)) {
      set_error("poisson_small_epochs: level %d of %lld x %lld cells", l, (long long)nz, (long long)nx);
      return ODIL_E_INVAL;
    }
    if (l > 0 && (2 * nx != a.lv[l - 1].nx || (ndim == 2 && 2 * nz != a.lv[l - 1].nz))) {
      set_error("poisson_small_epochs: level %d does not halve level %d", l, l - 1);
      return ODIL_E_INVAL;
    }
    L.nz = (int)nz, L.nx = (int)nx, L.size = (int)(nz * nx), L.off = off;
    L.rnx = 1.0f / (float)nx;
    L.xshift = -1;
    for (int sft = 0; sft < 30; ++sft)
      if (((int64_t)1 << sft) == nx) L.xshift = sft;
    off += L.size;
  }
  a.total = off;
  const int64_t n0 = a.lv[0].size;
  a.denom = (double)n0;
  a.scale = T(2) / T(n0);
  a.omb1 = omb1, a.omb2 = omb2, a.eps = eps;
  // the residual kernel's schedule for this array (poisson.hip: fill_args): canonical (Z, 1, X)
  const int per = kBlock * VecOf<T>::N;
  const int64_t XS = (a.lv[0].nx + per - 1) / per;
  a.usched = make_unit_sched(a.lv[0].nz, 1, XS);
  a.grid = unit_grid(a.usched);
  if (a.grid * 5 > kMaxPartials) {
    set_error("poisson_small_epochs: %d partial sums exceed the reduction workspace", a.grid);
    return ODIL_E_INVAL;
  }
  T hh[3] = {T(1), T(1), T(1)};
  if (ndim == 2) hh[0] = h2[0], hh[2] = h2[1];
  if (ndim == 1) hh[2] = h2[0];
  const size_t need = small_state_bytes<T>(a.total, (int)n0) + (size_t)a.grid * 5 * sizeof(double);
  a.lds = need <= kSmallLdsBytes;
  const int threads = (ndim == 1 && n0 <= 512) ? kBlock : kSmallThreads;
  if (a.lds) {
    static bool raised = false;  // (more than the 64 KB a launch gets by default: once per process and instantiation)
    if (!raised) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_poisson_small_epochs<T, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSmallLdsBytes) != hipSuccess) {
        set_error("poisson_small_epochs: cannot raise the dynamic LDS limit");
        return ODIL_E_LAUNCH;
      }
      raised = true;
    }
    hipLaunchKernelGGL((k_poisson_small_epochs<T, true>), dim3(1), dim3(threads), need, (hipStream_t)stream, x, m, v, g, u, fu,
                       rhs, alphas, losses, norms, partials, a, make_h2<T>(hh));
  } else {
    hipLaunchKernelGGL((k_poisson_small_epochs<T, false>), dim3(1), dim3(threads), 0, (hipStream_t)stream, x, m, v, g, u, fu,
                       rhs, alphas, losses, norms, partials, a, make_h2<T>(hh));
  }
  return check_launch("k_poisson_small_epochs");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_poisson_small_epochs_resident(const int64_t* shapes, int nlvl, int ndim, int elem_size) {
  if (!shapes || nlvl < 1 || nlvl > kSmallMaxLev || ndim < 1 || ndim > 2 || (elem_size != 4 && elem_size != 8)) return 0;
  int64_t total = 0;
  for (int l = 0; l < nlvl; ++l) {
    int64_t size = 1;
    for (int d = 0; d < ndim; ++d) size *= shapes[l * ndim + d];
    total += size;
  }
  int64_t n0 = 1;
  for (int d = 0; d < ndim; ++d) n0 *= shapes[d];
  if (total > (1 << 22)) return 0;
  const int per = kBlock * (16 / elem_size);
  const UnitSched us = make_unit_sched(ndim == 2 ? shapes[0] : 1, 1, (shapes[ndim - 1] + per - 1) / per);
  const size_t state = elem_size == 8 ? small_state_bytes<double>((int)total, (int)n0) : small_state_bytes<float>((int)total, (int)n0);
  return state + (size_t)unit_grid(us) * 5 * sizeof(double) <= kSmallLdsBytes ? 1 : 0;
}
int odil_poisson_small_epochs_f64(double* x, double* m, double* v, double* g, double* u, double* fu, const double* rhs,
                                  const int64_t* shapes, int nlvl, int ndim, const double* h2, const double* alphas,
                                  int nepochs, double one_minus_b1, double one_minus_b2, double eps, double* losses,
                                  double* norms, double* partials, void* stream) {
  return poisson_small_epochs<double>(x, m, v, g, u, fu, rhs, shapes, nlvl, ndim, h2, alphas, nepochs, one_minus_b1,
                                      one_minus_b2, eps, losses, norms, partials, stream);
}
int odil_poisson_small_epochs_f32(float* x, float* m, float* v, float* g, float* u, float* fu, const float* rhs,
                                  const int64_t* shapes, int nlvl, int ndim, const float* h2, const float* alphas,
                                  int nepochs, float one_minus_b1, float one_minus_b2, float eps, float* losses,
                                  float* norms, double* partials, void* stream) {
  return poisson_small_epochs<float>(x, m, v, g, u, fu, rhs, shapes, nlvl, ndim, h2, alphas, nepochs, one_minus_b1,
                                     one_minus_b2, eps, losses, norms, partials, stream);
}
}  // extern "C"
