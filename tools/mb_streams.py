"""Many-stream write microbenchmark: what the generated forward kernel's store pattern can reach.
One thread handles VEC consecutive floats of NR input arrays and NW output arrays (grid-stride over blocks)."""
import ctypes, sys, torch
sys.path.insert(0, '.')
from odil_amd.stencil_codegen import _compile
SRC = r"""
#include <hip/hip_runtime.h>
typedef float V __attribute__((ext_vector_type(@VEC@)));
struct Args { const float* in[@NR@]; float* out[@NW@]; long n; int nblocks; };
extern "C" __global__ __launch_bounds__(256) void k(const Args a) {
#if @CHUNK@
  for (long l = (long)blockIdx.x * 256 * @CHUNK@ + threadIdx.x, e = l + 256L * @CHUNK@; l < e && l * @VEC@ < a.n; l += 256) {
#else
  for (long l = (long)blockIdx.x * 256 + threadIdx.x; l * @VEC@ < a.n; l += (long)a.nblocks * 256) {
#endif
    V s = (V)(0.0f);
#pragma unroll
    for (int r = 0; r < @NR@; ++r) {
      s += *reinterpret_cast<const V*>(a.in[r] + l * @VEC@);
#if @SHIFTS@
      const long offs[8] = {-1, 1, -256, 256, -65536, 65536, -16777216, -16777216 + 1};
#pragma unroll
      for (int k = 0; k < @SHIFTS@; ++k) {  // one (unaligned) vector load per shifted read
        long j = l * @VEC@ + offs[k];
        j = j < 0 ? j + a.n : j;
        j = j + @VEC@ > a.n ? a.n - @VEC@ : j;  // (stays inside the array)
        V t;
        __builtin_memcpy(&t, a.in[r] + j, sizeof(V));
        s += t;
      }
#endif
    }
#pragma unroll
    for (int w = 0; w < @NW@; ++w) {
      V v = s * (float)(w + 1);
      if (@NT@) __builtin_nontemporal_store(v, reinterpret_cast<V*>(a.out[w] + l * @VEC@));
      else *reinterpret_cast<V*>(a.out[w] + l * @VEC@) = v;
    }
  }
}
extern "C" int run(const Args* a, void* stream) { hipLaunchKernelGGL(k, dim3(a->nblocks), dim3(256), 0, (hipStream_t)stream, *a); return (int)hipGetLastError(); }
"""
dev = torch.device("cuda:0")
n = 1 << 28  # 1 GB per array
NR, NW = 4, 17
ins = [torch.randn(n, device=dev) for _ in range(NR)]
outs = [torch.empty(n, device=dev) for _ in range(NW)]
for vec in (1, 2, 4):
    for nt in (1,):
        for chunk, shifts in ((0, 0), (0, 8)):
            nblocks = 65536 if chunk == 0 else (n // 256 + chunk - 1) // chunk
            src = SRC.replace("@VEC@", str(vec)).replace("@NR@", str(NR)).replace("@NW@", str(NW)).replace("@NT@", str(nt)).replace("@CHUNK@", str(chunk)).replace("@SHIFTS@", str(shifts))
            lib, _ = _compile(src)
            class Args(ctypes.Structure):
                _fields_ = [("inp", ctypes.c_void_p * NR), ("out", ctypes.c_void_p * NW), ("n", ctypes.c_long), ("nblocks", ctypes.c_int)]
            a = Args()
            for i, t in enumerate(ins): a.inp[i] = t.data_ptr()
            for i, t in enumerate(outs): a.out[i] = t.data_ptr()
            a.n = n; a.nblocks = min(nblocks, (n // vec + 255) // 256)
            lib.run.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            lib.run(ctypes.byref(a), s); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): lib.run(ctypes.byref(a), s)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 3
            print("vec %d, %d shifted reads per input: %.3f ms  %.2f TB/s (on %d + %d words)" % (vec, shifts, ms, (NR + NW) * n * 4 / ms / 1e9, NR, NW))
