"""HBM ceilings for read:write mixes with a plain streaming kernel (16 B per lane, nontemporal stores):
what the hand-written kernels can be held to.  python3 tools/mb_mix.py"""
import ctypes, sys, torch
sys.path.insert(0, '.')
from odil_amd.stencil_codegen import _compile
SRC = r"""
#include <hip/hip_runtime.h>
typedef double V __attribute__((ext_vector_type(2)));
struct Args { const double* in[8]; double* out[8]; long n; };
template <int NR, int NW> __global__ __launch_bounds__(256) void k(const Args a) {
  const long l = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
  if (l >= a.n) return;
  V s = (V)(0.0);
#pragma unroll
  for (int r = 0; r < NR; ++r) s += __builtin_nontemporal_load(reinterpret_cast<const V*>(a.in[r] + l));
#pragma unroll
  for (int w = 0; w < NW; ++w) __builtin_nontemporal_store(s * (double)(w + 1), reinterpret_cast<V*>(a.out[w] + l));
}
extern "C" int run(int nr, int nw, const Args* a, void* stream) {
  const unsigned g = (unsigned)((a->n / 2 + 255) / 256);
#define C(R, W) if (nr == R && nw == W) hipLaunchKernelGGL((k<R, W>), dim3(g), dim3(256), 0, (hipStream_t)stream, *a);
  C(1, 0) C(0, 1) C(1, 1) C(2, 1) C(4, 3) C(3, 1) C(1, 3) C(7, 7)
  return (int)hipGetLastError();
}
"""
if __name__ == "__main__":
    lib, _ = _compile(SRC)
    dev = torch.device("cuda:0")
    n = 1 << 27  # 1 GB per f64 array
    arrs = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(14)]
    class Args(ctypes.Structure):
        _fields_ = [("inp", ctypes.c_void_p * 8), ("out", ctypes.c_void_p * 8), ("n", ctypes.c_long)]
    a = Args()
    for i in range(7): a.inp[i] = arrs[i].data_ptr(); a.out[i] = arrs[7 + i].data_ptr()
    a.n = n
    lib.run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for nr, nw in [(1, 0), (0, 1), (1, 1), (2, 1), (3, 1), (4, 3), (1, 3), (7, 7)]:
        lib.run(nr, nw, ctypes.byref(a), s); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): lib.run(nr, nw, ctypes.byref(a), s)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("%d read : %d written streams: %.3f ms  %.2f TB/s" % (nr, nw, ms, (nr + nw) * n * 8 / ms / 1e9))
