// Small strided VALID correlations on gfx950: what `mod.convolution` and `mod.conv_transpose` of the reference's
// backend do for the multigrid transfers (reference src/odil/backend.py:112-126 -> jax.lax.conv, :165-172 ->
// jax.lax.conv_transpose; called by core.py:656-662 and :744-751 with kron weights of extent 1 - 4 per axis and
// stride 1 or 2).  One output element per thread, every tap a gather (no atomics, fixed summation order: taps in
// C order of the kernel).  HBM-bound: each input element is read once from memory, the other taps hit L2 / L1.
//
//   corr   : out[o] = sum_k w[k] in[o * s + k]                                  (o over the VALID extent)
//   corr_t : out[p] = sum_k w[k] in[(p - k) / s]  where s | (p - k), 0 <= (p - k) / s < n_in
//            -- the transpose of corr (cotangent of `convolution`), and with the kernel flipped by the caller the
//            forward pass of `conv_transpose` (out extent (n_in - 1) s + K).
#include "common.h"

namespace odil {

constexpr int kConvMaxTaps = 256;  // 4 taps per axis, 4 axes

struct ConvArgs {
  int64_t in[4];   // input shape (canonical 4-D)
  int64_t out[4];  // output shape
  int32_t k[4];    // kernel extents
  int32_t s[4];    // strides
};

__device__ inline void conv_split4(int64_t i, const int64_t (&n)[4], int64_t (&id)[4]) {
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

template <typename T, bool TRANSPOSED>
__global__ __launch_bounds__(kBlock) void k_conv_valid(const T* __restrict__ in, const T* __restrict__ w,
                                                      T* __restrict__ out, ConvArgs a) {
  const int64_t total = a.out[0] * a.out[1] * a.out[2] * a.out[3];
  const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= total) return;
  int64_t o[4];
  conv_split4(i, a.out, o);
  T acc = T(0);
  int tap = 0;
  for (int k0 = 0; k0 < a.k[0]; ++k0)
    for (int k1 = 0; k1 < a.k[1]; ++k1)
      for (int k2 = 0; k2 < a.k[2]; ++k2)
        for (int k3 = 0; k3 < a.k[3]; ++k3, ++tap) {
          const int kk[4] = {k0, k1, k2, k3};
          int64_t idx = 0;
          bool ok = true;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            int64_t q;
            if (TRANSPOSED) {
              const int64_t p = o[d] - kk[d];
              q = p / a.s[d];
              ok = ok && p >= 0 && q * a.s[d] == p && q < a.in[d];
            } else {
              q = o[d] * a.s[d] + kk[d];  // inside by construction of the VALID extent
            }
            idx = idx * a.in[d] + q;
          }
          if (ok) acc = acc + w[tap] * in[idx];
        }
  out[i] = acc;
}

template <typename T>
static int conv_valid(const T* in, const T* w, T* out, const int64_t* ishape, const int64_t* wshape,
                      const int64_t* strides, const int64_t* oshape, int ndim, int transposed, void* stream) {
  if (ndim < 1 || ndim > ODIL_MAX_NDIM || !in || !w || !out || !ishape || !wshape || !strides || !oshape) {
    set_error("conv_valid: invalid ndim=%d or null argument", ndim);
    return ODIL_E_INVAL;
  }
  ConvArgs a;
  int64_t ks[4], ss[4];
  canon_shape(ishape, ndim, a.in);
  canon_shape(oshape, ndim, a.out);
  canon_shape(wshape, ndim, ks);
  canon_shape(strides, ndim, ss);
  int64_t taps = 1;
  for (int d = 0; d < 4; ++d) {
    if (ks[d] < 1 || ks[d] > 4 || ss[d] < 1 || ss[d] > 4 || a.in[d] < 1 || a.out[d] < 1) {
      set_error("conv_valid: axis %d: kernel extent %lld, stride %lld (1..4 supported), extents %lld -> %lld", d,
                (long long)ks[d], (long long)ss[d], (long long)a.in[d], (long long)a.out[d]);
      return ODIL_E_INVAL;
    }
    const int64_t want = transposed ? (a.in[d] - 1) * ss[d] + ks[d] : (a.in[d] - ks[d]) / ss[d] + 1;
    // corr_t may be asked for a LONGER output (cotangent of a VALID correlation that left a remainder: zeros there)
    if (a.in[d] < ks[d] && !transposed) {
      set_error("conv_valid: input extent %lld shorter than the kernel %lld", (long long)a.in[d], (long long)ks[d]);
      return ODIL_E_INVAL;
    }
    if (transposed ? a.out[d] < want : a.out[d] != want) {
      set_error("conv_valid: axis %d output extent %lld, expected %s%lld", d, (long long)a.out[d],
                transposed ? ">= " : "", (long long)want);
      return ODIL_E_INVAL;
    }
    a.k[d] = (int32_t)ks[d];
    a.s[d] = (int32_t)ss[d];
    taps *= ks[d];
  }
  if (taps > kConvMaxTaps) {
    set_error("conv_valid: %lld taps (at most %d)", (long long)taps, kConvMaxTaps);
    return ODIL_E_INVAL;
  }
  const int64_t total = prod4(a.out);
  const int64_t nb = (total + kBlock - 1) / kBlock;
  if (nb >= (int64_t)1 << 31) {
    set_error("conv_valid: too many elements");
    return ODIL_E_INVAL;
  }
  if (transposed)
    hipLaunchKernelGGL((k_conv_valid<T, true>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, in, w, out, a);
  else
    hipLaunchKernelGGL((k_conv_valid<T, false>), dim3((unsigned)nb), dim3(kBlock), 0, (hipStream_t)stream, in, w, out, a);
  return check_launch("k_conv_valid");
}

}  // namespace odil

using namespace odil;

extern "C" {
int odil_conv_valid_f64(const double* in, const double* w, double* out, const int64_t* ishape, const int64_t* wshape,
                        const int64_t* strides, const int64_t* oshape, int ndim, int transposed, void* stream) {
  return conv_valid<double>(in, w, out, ishape, wshape, strides, oshape, ndim, transposed, stream);
}
int odil_conv_valid_f32(const float* in, const float* w, float* out, const int64_t* ishape, const int64_t* wshape,
                        const int64_t* strides, const int64_t* oshape, int ndim, int transposed, void* stream) {
  return conv_valid<float>(in, w, out, ishape, wshape, strides, oshape, ndim, transposed, stream);
}
}
