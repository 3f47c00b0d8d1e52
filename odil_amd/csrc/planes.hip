// Planes of ghost-extended arrays <-> contiguous message buffers of the slab exchanges (odil_amd/slab.py,
// slab_traced.py; no reference counterpart: the reference is single-device).
//
// A "plane" of an array cut along axis a is, in the packed state vector, `outer` runs of `inner` contiguous
// elements, run o at base + o * ostride (outer = product of the extents before a, inner = of those after it,
// ostride = extent(a) * inner); in the message it is one contiguous range at `boff`.  One launch moves every plane of
// every level of every field of an exchange: blockIdx.y is the plane, blockIdx.x walks it in 16-byte packs.  Pure
// HBM streaming: 2 words per element (the index_select / index_copy_ this replaces read an 8-byte index per element
// on top, and kept those index tables -- 8 bytes per exchanged element -- resident).
#include "common.h"

namespace odil {

struct PlaneDesc {
  int64_t base, outer, ostride, inner, boff;
};

enum PlaneMode : int { kPack = 0, kUnpack = 1, kUnpackAdd = 2 };

template <typename T, int MODE, int VEC>
__global__ __launch_bounds__(kBlock) void k_planes(T* __restrict__ arr, T* __restrict__ buf,
                                                   const PlaneDesc* __restrict__ descs) {
  const PlaneDesc d = descs[blockIdx.y];
  const int64_t count = d.outer * d.inner / VEC;
  const int64_t per = d.inner / VEC;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < count; e += (int64_t)gridDim.x * kBlock) {
    const int64_t o = e / per, i = (e - o * per) * VEC;
    T* a = arr + d.base + o * d.ostride + i;
    T* b = buf + d.boff + e * VEC;
    typedef T V __attribute__((ext_vector_type(VEC), aligned(sizeof(T))));
    if (MODE == kPack) {
      __builtin_nontemporal_store(*(const V*)a, (V*)b);
    } else if (MODE == kUnpack) {
      *(V*)a = __builtin_nontemporal_load((const V*)b);
    } else {
      *(V*)a = *(const V*)a + __builtin_nontemporal_load((const V*)b);
    }
  }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kBlock) void k_planes_scalar(T* __restrict__ arr, T* __restrict__ buf,
                                                          const PlaneDesc* __restrict__ descs, int mode) {
  const PlaneDesc d = descs[blockIdx.y];
  const int64_t count = d.outer * d.inner;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < count; e += (int64_t)gridDim.x * kBlock) {
    const int64_t o = e / d.inner, i = e - o * d.inner;
    T* a = arr + d.base + o * d.ostride + i;
    T* b = buf + d.boff + e;
    if (mode == kPack) *b = *a;
    else if (mode == kUnpack) *a = *b;
    else *a = *a + *b;
  }
}

template <typename T>
static int planes_copy(T* arr, T* buf, const int64_t* descs, int ndesc, int64_t max_count, int vec_ok, int mode,
                       void* stream) {
  if (ndesc == 0) return 0;
  if (!arr || !buf || !descs || ndesc < 0 || ndesc > 65535 || max_count < 1 || mode < 0 || mode > 2) {
    set_error("planes_copy: invalid arguments (ndesc=%d, mode=%d)", ndesc, mode);
    return ODIL_E_INVAL;
  }
  const PlaneDesc* d = reinterpret_cast<const PlaneDesc*>(descs);
  hipStream_t s = (hipStream_t)stream;
  if (vec_ok) {
    const int64_t packs = (max_count + 3) / 4;
    const unsigned gx = (unsigned)((packs + kBlock - 1) / kBlock < 4096 ? (packs + kBlock - 1) / kBlock : 4096);
    const dim3 grid(gx ? gx : 1, ndesc);
    if (mode == kPack) hipLaunchKernelGGL((k_planes<T, kPack, 4>), grid, dim3(kBlock), 0, s, arr, buf, d);
    else if (mode == kUnpack) hipLaunchKernelGGL((k_planes<T, kUnpack, 4>), grid, dim3(kBlock), 0, s, arr, buf, d);
    else hipLaunchKernelGGL((k_planes<T, kUnpackAdd, 4>), grid, dim3(kBlock), 0, s, arr, buf, d);
  } else {
    const unsigned gx = (unsigned)((max_count + kBlock - 1) / kBlock < 4096 ? (max_count + kBlock - 1) / kBlock : 4096);
    hipLaunchKernelGGL((k_planes_scalar<T, 1>), dim3(gx ? gx : 1, ndesc), dim3(kBlock), 0, s, arr, buf, d, mode);
  }
  return check_launch("k_planes");
}

}  // namespace odil

extern "C" {
int odil_planes_copy_f64(double* arr, double* buf, const int64_t* descs, int ndesc, int64_t max_count, int vec_ok,
                         int mode, void* stream) {
  return odil::planes_copy<double>(arr, buf, descs, ndesc, max_count, vec_ok, mode, stream);
}
int odil_planes_copy_f32(float* arr, float* buf, const int64_t* descs, int ndesc, int64_t max_count, int vec_ok,
                         int mode, void* stream) {
  return odil::planes_copy<float>(arr, buf, descs, ndesc, max_count, vec_ok, mode, stream);
}
}
