// The dense block of the Newton normal equations on the matrix cores.
//
// `Problem.linearize` gives `Array` / `NeuralNet` unknowns DENSE Jacobian columns (reference
// src/odil/core.py:1189-1203): D is (rows = residual values) x (p = a few dozen parameters).  The normal equations
// (reference src/odil/linsolver.py:17-23) need D^T D, D^T r and, for the Schur complement against the stencil part,
// (S^T D)^T Z -- all products X^T Y of two tall, skinny matrices.  One workgroup of four waves walks a contiguous
// range of rows four at a time: lane l holds X[r + l/16][16 ti + l%16] and Y[r + l/16][16 tj + l%16], which ARE the
// A and B operands of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 for the 16 x 16 tile (ti, tj) of the result,
// so every pair of column tiles is one MFMA per four rows with no shuffling.  Partial results: waves of a workgroup
// are summed through LDS in wave order, workgroups in index order by a second kernel -- bit-reproducible.
#include "common.h"

namespace odil {

constexpr int kGramTile = 16;
constexpr int kGramMaxTiles = 4;   // up to 64 columns per operand
constexpr int kGramBlocks = 256;   // workgroups (= partial results)
constexpr int kGramWaves = kBlock / 64;

template <typename T> struct Acc4 { typedef T type __attribute__((ext_vector_type(4))); };

__device__ inline Acc4<double>::type mfma_16x16x4(double a, double b, Acc4<double>::type c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ inline Acc4<float>::type mfma_16x16x4(float a, float b, Acc4<float>::type c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
// row of accumulator register `reg` held by `lane` (cdna_hip_programming.md, fragment layout): f64 and f32 differ
template <typename T> __device__ inline int acc_row(int lane, int reg);
template <> __device__ inline int acc_row<double>(int lane, int reg) { return (lane >> 4) + 4 * reg; }
template <> __device__ inline int acc_row<float>(int lane, int reg) { return (lane >> 4) * 4 + reg; }

// partial[b][i][j] = sum over the rows of workgroup b of X[r][i] Y[r][j]   (i < 16 TX, j < 16 TY, zero padded)
template <typename T, int TX, int TY>
__global__ __launch_bounds__(kBlock) void k_xty_partial(const T* __restrict__ x, const T* __restrict__ y, int64_t n,
                                                       int px, int py, int64_t ldx, int64_t ldy,
                                                       T* __restrict__ partial) {
  typedef typename Acc4<T>::type A4;
  __shared__ T red[kGramWaves][TX * TY][4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, krow = lane >> 4;
  // contiguous row range of this workgroup, a multiple of 4 * kGramWaves rows long
  const int64_t quads = (n + 3) / 4;
  const int64_t per = (quads + gridDim.x - 1) / gridDim.x;
  const int64_t q0 = (int64_t)blockIdx.x * per, q1 = q0 + per < quads ? q0 + per : quads;
  A4 acc[TX][TY];
#pragma unroll
  for (int i = 0; i < TX; ++i)
#pragma unroll
    for (int j = 0; j < TY; ++j) acc[i][j] = A4{T(0), T(0), T(0), T(0)};
  for (int64_t q = q0 + wave; q < q1; q += kGramWaves) {
    const int64_t r = 4 * q + krow;
    T xv[TX], yv[TY];
#pragma unroll
    for (int i = 0; i < TX; ++i) {
      const int c = kGramTile * i + col;
      xv[i] = (r < n && c < px) ? x[r * ldx + c] : T(0);
    }
#pragma unroll
    for (int j = 0; j < TY; ++j) {
      const int c = kGramTile * j + col;
      yv[j] = (r < n && c < py) ? y[r * ldy + c] : T(0);
    }
#pragma unroll
    for (int i = 0; i < TX; ++i)
#pragma unroll
      for (int j = 0; j < TY; ++j) acc[i][j] = mfma_16x16x4(xv[i], yv[j], acc[i][j]);
  }
#pragma unroll
  for (int i = 0; i < TX; ++i)
#pragma unroll
    for (int j = 0; j < TY; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][i * TY + j][e][lane] = acc[i][j][e];
  __syncthreads();
  // waves summed in order by the threads of wave 0's shape: every thread takes some (tile, reg, lane) entries
  T* out = partial + (int64_t)blockIdx.x * (kGramTile * TX) * (kGramTile * TY);
  for (int k = threadIdx.x; k < TX * TY * 4 * 64; k += kBlock) {
    const int l = k & 63, e = (k >> 6) & 3, t = k >> 8;
    T s = red[0][t][e][l];
#pragma unroll
    for (int w = 1; w < kGramWaves; ++w) s = s + red[w][t][e][l];
    const int ti = t / TY, tj = t - ti * TY;
    const int row = kGramTile * ti + acc_row<T>(l, e), cc = kGramTile * tj + (l & 15);
    out[row * (kGramTile * TY) + cc] = s;
  }
}

// out[i][j] = sum_b partial[b][i][j] in index order, accumulated in double
template <typename T>
__global__ __launch_bounds__(kBlock) void k_xty_final(const T* __restrict__ partial, int nblocks, int ppx, int ppy,
                                                     int px, int py, T* __restrict__ out) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= px * py) return;
  const int i = k / py, j = k - i * py;
  double s = 0.0;
  for (int b = 0; b < nblocks; ++b) s += (double)partial[((int64_t)b * ppx + i) * ppy + j];
  out[k] = T(s);
}

template <typename T, int TX>
static int xty_launch_y(const T* x, const T* y, int64_t n, int px, int py, int64_t ldx, int64_t ldy, T* partial,
                        int ty, hipStream_t stream) {
#define ODIL_XTY_CASE(TY)                                                                                          \
  case TY:                                                                                                         \
    hipLaunchKernelGGL((k_xty_partial<T, TX, TY>), dim3(kGramBlocks), dim3(kBlock), 0, stream, x, y, n, px, py, ldx, \
                       ldy, partial);                                                                              \
    break;
  switch (ty) {
    ODIL_XTY_CASE(1) ODIL_XTY_CASE(2) ODIL_XTY_CASE(3) ODIL_XTY_CASE(4)
    default: return ODIL_E_INVAL;
  }
#undef ODIL_XTY_CASE
  return check_launch("k_xty_partial");
}

template <typename T>
static int dense_block_xty(const T* x, const T* y, int64_t n, int px, int py, int64_t ldx, int64_t ldy, T* out,
                           T* workspace, void* stream) {
  if (!x || !y || !out || !workspace || n < 1 || px < 1 || py < 1 || px > kGramTile * kGramMaxTiles ||
      py > kGramTile * kGramMaxTiles || ldx < px || ldy < py) {
    set_error("dense_block_xty: null pointer, n < 1 or column counts (%d, %d) outside 1..%d", px, py,
              kGramTile * kGramMaxTiles);
    return ODIL_E_INVAL;
  }
  const int tx = (px + kGramTile - 1) / kGramTile, ty = (py + kGramTile - 1) / kGramTile;
  hipStream_t s = (hipStream_t)stream;
  int e;
  switch (tx) {
    case 1: e = xty_launch_y<T, 1>(x, y, n, px, py, ldx, ldy, workspace, ty, s); break;
    case 2: e = xty_launch_y<T, 2>(x, y, n, px, py, ldx, ldy, workspace, ty, s); break;
    case 3: e = xty_launch_y<T, 3>(x, y, n, px, py, ldx, ldy, workspace, ty, s); break;
    default: e = xty_launch_y<T, 4>(x, y, n, px, py, ldx, ldy, workspace, ty, s); break;
  }
  if (e) return e;
  hipLaunchKernelGGL(k_xty_final<T>, dim3((px * py + kBlock - 1) / kBlock), dim3(kBlock), 0, s, workspace, kGramBlocks,
                     kGramTile * tx, kGramTile * ty, px, py, out);
  return check_launch("k_xty_final");
}

}  // namespace odil

using namespace odil;

extern "C" {
size_t odil_dense_block_workspace_bytes(void) {
  return (size_t)kGramBlocks * (kGramTile * kGramMaxTiles) * (kGramTile * kGramMaxTiles) * sizeof(double);
}
int odil_dense_block_xty_f64(const double* x, const double* y, int64_t n, int px, int py, int64_t ldx, int64_t ldy,
                             double* out, double* workspace, void* stream) {
  return dense_block_xty<double>(x, y, n, px, py, ldx, ldy, out, workspace, stream);
}
int odil_dense_block_xty_f32(const float* x, const float* y, int64_t n, int px, int py, int64_t ldx, int64_t ldy,
                             float* out, float* workspace, void* stream) {
  return dense_block_xty<float>(x, y, n, px, py, ldx, ldy, out, workspace, stream);
}
int odil_dense_block_gram_f64(const double* d, int64_t n, int p, int64_t ld, double* out, double* workspace,
                              void* stream) {
  return dense_block_xty<double>(d, d, n, p, p, ld, ld, out, workspace, stream);
}
int odil_dense_block_gram_f32(const float* d, int64_t n, int p, int64_t ld, float* out, float* workspace,
                              void* stream) {
  return dense_block_xty<float>(d, d, n, p, p, ld, ld, out, workspace, stream);
}
}
