// Ceiling of a write-heavy many-stream kernel (the traced k_fwd: 4 fields in, 17 cotangent arrays out, float):
// hipcc -O3 --offload-arch=gfx950 tools/mb_streams17.hip -o tools/bin/mb_streams17
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
struct Ptrs { const float* in[4]; float* out[17]; };

template <int NOUT, int V, bool NT>
__global__ __launch_bounds__(256) void k(Ptrs p, long n) {
  typedef float VT __attribute__((ext_vector_type(V)));
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n / V; i += stride) {
    VT s = 0;
#pragma unroll
    for (int a = 0; a < 4; ++a) s += reinterpret_cast<const VT*>(p.in[a])[i];
#pragma unroll
    for (int o = 0; o < NOUT; ++o) {
      const VT val = s * (float)(o + 1);
      if (NT) __builtin_nontemporal_store(val, reinterpret_cast<VT*>(p.out[o]) + i);
      else reinterpret_cast<VT*>(p.out[o])[i] = val;
    }
  }
}

template <int NOUT, int V, bool NT>
void run(Ptrs p, long n, int grid) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<NOUT, V, NT>), dim3(grid), dim3(256), 0, 0, p, n);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<NOUT, V, NT>), dim3(grid), dim3(256), 0, 0, p, n);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  printf("out %2d  %2d B/lane  nt %d  grid %7d: %.3f ms  %.2f TB/s\n", NOUT, 4 * V, (int)NT, grid, ms,
         (4.0 + NOUT) * n * 4 / ms / 1e9);
}

int main() {
  const long n = 1L << 28;  // 268 M points, as one slab rank of config 5
  Ptrs p;
  for (int a = 0; a < 4; ++a) { float* q; CHECK(hipMalloc(&q, n * 4)); CHECK(hipMemset(q, 0, n * 4)); p.in[a] = q; }
  for (int o = 0; o < 17; ++o) { CHECK(hipMalloc(&p.out[o], n * 4)); CHECK(hipMemset(p.out[o], 0, n * 4)); }
  for (int grid : {65536, 1 << 20}) {
    run<17, 1, false>(p, n, grid);
    run<17, 1, true>(p, n, grid);
    run<17, 4, false>(p, n, grid / 4);
    run<17, 4, true>(p, n, grid / 4);
    run<4, 1, false>(p, n, grid);
    run<4, 4, false>(p, n, grid / 4);
    run<1, 1, false>(p, n, grid);
    run<1, 4, true>(p, n, grid / 4);
  }
  return 0;
}
