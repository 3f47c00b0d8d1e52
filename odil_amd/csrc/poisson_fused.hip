// Fused Poisson loss + gradient on gfx950:  gu = scale * J^T (J u - rhs),  loss = mean((J u - rhs)^2)
// in ONE pass over u and rhs (reference examples/poisson/poisson.py:89-113 + the reverse pass
// of core.py:1100).  The residual fu is never written to HBM: 3 words per cell instead of
// the 5 of residual + adjoint.
//
// A workgroup owns TY rows of a plane (all of x, <= 512 cells, 2 per lane) and marches z.
// Each lane keeps, for its two columns,
//   u   at planes zf-1, zf, zf+1 on rows y0-2 .. y0+TY+1      (registers)
//   fu  at planes zf-2, zf-1, zf on rows y0-1 .. y0+TY        (registers; the two halo rows
//       are recomputed instead of communicated)
// x-neighbours come from adjacent lanes by wave shuffles; only the two edge lanes of a wave
// fetch u from memory, and fu edges cross waves through a 3-deep LDS ring of edge values
// (one barrier per plane).  Every step: fu[zf] from u (same operation order as the
// two-kernel path), then gu[zf-1] from fu.  Loss partial sums are deterministic.
#include "poisson.h"

namespace odil {

struct FusedArgs {
  int Z, Y, X;
  UnitSched usched;
};

template <typename T>
struct alignas(2 * sizeof(T)) Pair {
  T a, b;
};

template <typename T>
__device__ inline T shfl_up1(T v) {
  return __shfl_up(v, 1, 64);
}
template <typename T>
__device__ inline T shfl_down1(T v) {
  return __shfl_down(v, 1, 64);
}

template <typename T, int TY, int C, int NT>
__global__ __launch_bounds__(NT) void k_poisson_loss_grad(const T* __restrict__ u, const T* __restrict__ rhs,
                                                             T* __restrict__ gu, FusedArgs a, H2<T> h, T scale,
                                                             double* __restrict__ partials) {
  constexpr int NU = TY + 4, NF = TY + 2;
  constexpr int NW = NT / 64;
  __shared__ T edge[3][TY][NW][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Z = a.Z, Y = a.Y, X = a.X;
  const int64_t sy = X, sz = (int64_t)Y * X;
  int zc, yt, xs;
  const bool have = unit_decode(a.usched, zc, yt, xs);
  const int x0 = tid * C;
  const bool active = have && x0 < X;
  const int valid = active ? (X - x0 < C ? X - x0 : C) : 0;
  const int y0 = have ? yt * TY : 0;
  const int z0 = have ? zc * a.usched.ZC : 0;
  const int z1 = have ? (z0 + a.usched.ZC < Z ? z0 + a.usched.ZC : Z) : 0;
  double local = 0.0;

  // wrapped row offsets (periodic like mod.roll; the wrapped values are masked by the where()s)
  int64_t yoff[NU];
#pragma unroll
  for (int j = 0; j < NU; ++j) {
    int y = y0 - 2 + j;
    y = y < 0 ? y + Y : (y >= Y ? y - Y : y);
    yoff[j] = (int64_t)y * sy;
  }

  // Registers.  U*: u planes zf-1 / zf / zf+1; Un, Rn: the NEXT step's plane and rhs rows,
  // whose loads are issued one full step ahead (software prefetch: at two waves per SIMD the
  // HBM latency must be covered by the step's own arithmetic).  F*: scale * fu on planes
  // zf-2 / zf-1 / zf (the products scale*f are exactly those of the two-kernel path).
  T Um[NU][C], Uc[NU][C], Up[NU][C], Un[NU][C];
  T Fm[NF][C], Fc[NF][C], Fp[NF][C];
  T R[NF][C], Rn[NF][C];
#pragma unroll
  for (int i = 0; i < NF; ++i)
    for (int c = 0; c < C; ++c) Fm[i][c] = Fc[i][c] = T(0);
  // x-edge values of u cross waves through LDS as well (published one step ahead), so the
  // only global loads are the coalesced row loads, all issued a full step before use.
  __shared__ T uedge[2][NF][NW][2];
  auto publish_u_edges = [&](int buf, const T P[NU][C]) {
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (lane == 0) uedge[buf][i][wave][0] = P[i + 1][0];
      if (lane == 63) uedge[buf][i][wave][1] = P[i + 1][C - 1];
    }
  };

  auto load_plane = [&](int zp, T out[NU][C]) {
    const int zz = zp < 0 ? zp + Z : (zp >= Z ? zp - Z : zp);
    const T* base = u + (int64_t)zz * sz + x0;
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      if (C == 2 && valid == 2) {
        const Pair<T> p = *reinterpret_cast<const Pair<T>*>(base + yoff[j]);
        out[j][0] = p.a;
        out[j][C - 1] = p.b;
      } else {
        out[j][C - 1] = T(0);
        out[j][0] = valid ? base[yoff[j]] : T(0);
      }
    }
  };
  auto load_rhs = [&](int zp, T out[NF][C]) {
    const int zz = zp < 0 ? zp + Z : (zp >= Z ? zp - Z : zp);
    const T* rb = rhs + (int64_t)zz * sz + x0;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      if (C == 2 && valid == 2) {
        const Pair<T> p = *reinterpret_cast<const Pair<T>*>(rb + yoff[i + 1]);
        out[i][0] = p.a;
        out[i][C - 1] = p.b;
      } else {
        out[i][C - 1] = T(0);
        out[i][0] = valid ? rb[yoff[i + 1]] : T(0);
      }
    }
  };

  if (have) {
    load_plane(z0 - 2, Um);
    load_plane(z0 - 1, Uc);
    load_plane(z0, Up);
    load_rhs(z0 - 1, R);
    publish_u_edges(0, Uc);
  }
  __syncthreads();
  const int nsteps = have ? z1 - z0 + 2 : 0;
  // every workgroup runs the same number of barriers per step; idle workgroups run none
  for (int s = 0; s < nsteps; ++s) {
    const int zf = z0 - 1 + s;                                   // plane whose fu is formed now
    const int zfw = zf < 0 ? zf + Z : (zf >= Z ? zf - Z : zf);   // wrapped
    if (s + 1 < nsteps) {  // prefetch for step s+1
      load_plane(zf + 2, Un);
      load_rhs(zf + 1, Rn);
    }
    publish_u_edges((s + 1) & 1, Up);  // read at step s+1, after this step's barrier
    const bool zlo = zfw == 0, zhi = zfw == Z - 1;
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      const int yy = y0 - 1 + i;
      const int yw = yy < 0 ? yy + Y : (yy >= Y ? yy - Y : yy);
      const bool ylo = yw == 0, yhi = yw == Y - 1;
      // x neighbours of the pair on row i (u row i+1): adjacent lanes, edge lanes from memory
      T left = shfl_up1<T>(Uc[i + 1][C - 1]);
      T right = shfl_down1<T>(Uc[i + 1][0]);
      if (lane == 0) left = wave > 0 ? uedge[s & 1][i][wave - 1][1] : T(0);
      if (lane == 63) right = wave < NW - 1 ? uedge[s & 1][i][wave + 1][0] : T(0);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const int x = x0 + c;
        const T q = Uc[i + 1][c];
        T acc = axis_term<T>(q, Um[i + 1][c], Up[i + 1][c], zlo, zhi, h, 0);
        acc = acc + axis_term<T>(q, Uc[i][c], Uc[i + 2][c], ylo, yhi, h, 1);
        const T xm = c == 0 ? left : Uc[i + 1][0];
        const T xp = (c == C - 1 || valid == 1) ? right : Uc[i + 1][C - 1];
        acc = acc + axis_term<T>(q, xm, xp, x == 0, x == X - 1, h, 2);
        const T f = acc - R[i][c];
        Fp[i][c] = scale * f;
        const bool own = i >= 1 && i <= TY && yy < Y && c < valid && zf >= z0 && zf < z1;
        if (own) local += (double)(f * f);
      }
    }
    // publish this wave's edge values of (scaled) fu on the own rows
#pragma unroll
    for (int i = 1; i <= TY; ++i) {
      if (lane == 0) edge[s % 3][i - 1][wave][0] = Fp[i][0];
      if (lane == 63) edge[s % 3][i - 1][wave][1] = Fp[i][C - 1];
    }
    __syncthreads();
    // gu on plane zg = zf - 1: centre Fc (published at step s-1), below Fm, above Fp
    const int zg = zf - 1;
    if (zg >= z0 && zg < z1) {
      const int eb = (s + 2) % 3;  // == (s - 1) mod 3
#pragma unroll
      for (int i = 1; i <= TY; ++i) {
        const int y = y0 + i - 1;
        T left = shfl_up1<T>(Fc[i][C - 1]);
        T right = shfl_down1<T>(Fc[i][0]);
        if (lane == 0) left = wave > 0 ? edge[eb][i - 1][wave - 1][1] : T(0);
        if (lane == 63) right = wave < NW - 1 ? edge[eb][i - 1][wave + 1][0] : T(0);
        if (y < Y && active) {
          T out[C];
#pragma unroll
          for (int c = 0; c < C; ++c) {
            const int x = x0 + c;
            const T fb = Fc[i][c];
            T g = adj_axis<T>(fb, Fm[i][c], Fp[i][c], zg, Z, h, 0);
            g = g + adj_axis<T>(fb, Fc[i - 1][c], Fc[i + 1][c], y, Y, h, 1);
            const T xm = c == 0 ? left : Fc[i][0];
            const T xp = (c == C - 1 || valid == 1) ? right : Fc[i][C - 1];
            g = g + adj_axis<T>(fb, xm, xp, x, X, h, 2);
            out[c] = g;
          }
          T* dst = gu + (int64_t)zg * sz + (int64_t)y * sy + x0;
          if (C == 2 && valid == 2) {
            Pair<T> p;
            p.a = out[0];
            p.b = out[C - 1];
            *reinterpret_cast<Pair<T>*>(dst) = p;
          } else {
            dst[0] = out[0];
          }
        }
      }
    }
#pragma unroll
    for (int j = 0; j < NU; ++j) {
      for (int c = 0; c < C; ++c) {
        Um[j][c] = Uc[j][c];
        Uc[j][c] = Up[j][c];
        Up[j][c] = Un[j][c];
      }
    }
#pragma unroll
    for (int i = 0; i < NF; ++i) {
      for (int c = 0; c < C; ++c) {
        Fm[i][c] = Fc[i][c];
        Fc[i][c] = Fp[i][c];
        R[i][c] = Rn[i][c];
      }
    }
  }
  // block-wide sum over NT threads (fixed order)
  __shared__ double wsum[NW];
  double v = local;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  if (lane == 0) wsum[wave] = v;
  __syncthreads();
  if (tid == 0) {
    double total = 0;
    for (int w = 0; w < NW; ++w) total += wsum[w];
    partials[blockIdx.x] = total;
  }
}

static bool fused_supported(const int64_t* shape, int ndim) {
  if (ndim != 3 || !shape) return false;
  if (shape[2] > 512 || shape[2] < 4 || shape[1] < 4 || shape[0] < 4) return false;
  if (shape[0] >= (1 << 30) || shape[1] >= (1 << 30)) return false;
  return true;
}

template <typename T>
static int poisson_loss_grad(const T* u, const T* rhs, T* gu, const int64_t* shape, int ndim, const T* h2,
                             double* partials, T* loss, void* stream) {
  if (!fused_supported(shape, ndim)) {
    set_error("poisson_loss_grad: unsupported shape (3-D, 4 <= extents, last extent <= %d); use residual + adjoint",
              512);
    return ODIL_E_INVAL;
  }
  if (!u || !rhs || !gu || !h2 || !partials || !loss) {
    set_error("poisson_loss_grad: null pointer");
    return ODIL_E_INVAL;
  }
  constexpr int TY = 2;
  FusedArgs a;
  a.Z = (int)shape[0];
  a.Y = (int)shape[1];
  a.X = (int)shape[2];
  const int64_t ytiles = (a.Y + TY - 1) / TY;
  a.usched = make_unit_sched(a.Z, ytiles, 1);
  const int grid = unit_grid(a.usched);
  if (grid > kMaxPartials) {
    set_error("poisson_loss_grad: %d workgroups exceed the reduction workspace", grid);
    return ODIL_E_INVAL;
  }
  T hh[3] = {h2[0], h2[1], h2[2]};
  const double size = (double)shape[0] * (double)shape[1] * (double)shape[2];
  const T scale = T(2) / T(size);
  if (a.X > 256) {
    // one cell per lane, 8 waves: half the registers per lane -> twice the waves per SIMD
    hipLaunchKernelGGL((k_poisson_loss_grad<T, TY, 1, 512>), dim3(grid), dim3(512), 0, (hipStream_t)stream, u, rhs, gu,
                       a, make_h2<T>(hh), scale, partials);
  } else {
    hipLaunchKernelGGL((k_poisson_loss_grad<T, TY, 1, 256>), dim3(grid), dim3(256), 0, (hipStream_t)stream, u, rhs, gu,
                       a, make_h2<T>(hh), scale, partials);
  }
  if (int e = check_launch("k_poisson_loss_grad")) return e;
  return launch_final_reduce<T>(partials, grid, 0, 1, size, loss, (hipStream_t)stream);
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_poisson_loss_grad_supported(const int64_t* shape, int ndim) { return fused_supported(shape, ndim) ? 1 : 0; }
int odil_poisson_loss_grad_f64(const double* u, const double* rhs, double* gu, const int64_t* shape, int ndim,
                               const double* h2, double* partials, double* loss, void* stream) {
  return poisson_loss_grad<double>(u, rhs, gu, shape, ndim, h2, partials, loss, stream);
}
int odil_poisson_loss_grad_f32(const float* u, const float* rhs, float* gu, const int64_t* shape, int ndim,
                               const float* h2, double* partials, float* loss, void* stream) {
  return poisson_loss_grad<float>(u, rhs, gu, shape, ndim, h2, partials, loss, stream);
}
}
