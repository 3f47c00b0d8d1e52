// Access-pattern ceilings for the fused adjoint kernel (k_poisson_adjoint_tile): the same bytes with almost no
// arithmetic.  hipcc -O3 --offload-arch=gfx950 tools/mb_tile_traffic.hip -o gpurun_out/mb_tile_traffic
//   A  flat stream, 4 arrays read / 3 written, 16 B per lane
//   B  tile walk (TY x TX coarse columns = 2TY x 2TX fine cells per plane, z-chunks of ZC coarse planes), own cells
//      only, no LDS, loads of plane z + 1 issued before plane z is consumed
//   C  B + the fu window with its halo staged through an LDS ring, two barriers per plane, 7-point sum from LDS
// Each variant at 1, 2 and (where LDS allows) 3+ workgroups per CU (dynamic LDS padding).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef double P2 __attribute__((ext_vector_type(2)));
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void k_flat(const double* __restrict__ fu, double* __restrict__ x,
                                                 double* __restrict__ m, double* __restrict__ v, int64_t npacks) {
  extern __shared__ char pad[];
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < npacks; i += stride) {
    const P2 f = __builtin_nontemporal_load(reinterpret_cast<const P2*>(fu) + i);
    P2 a = __builtin_nontemporal_load(reinterpret_cast<const P2*>(x) + i);
    P2 b = __builtin_nontemporal_load(reinterpret_cast<const P2*>(m) + i);
    P2 c = __builtin_nontemporal_load(reinterpret_cast<const P2*>(v) + i);
    b = b + 0.1 * (f - b);
    c = c + 0.001 * (f * f - c);
    a = a - 1e-3 * b;
    __builtin_nontemporal_store(a, reinterpret_cast<P2*>(x) + i);
    __builtin_nontemporal_store(b, reinterpret_cast<P2*>(m) + i);
    __builtin_nontemporal_store(c, reinterpret_cast<P2*>(v) + i);
  }
}


template <bool NT>
__device__ __forceinline__ P2 ld(const double* p) {
  if constexpr (NT) return __builtin_nontemporal_load(reinterpret_cast<const P2*>(p));
  return *reinterpret_cast<const P2*>(p);
}
template <bool NT>
__device__ __forceinline__ void st(P2 v, double* p) {
  if constexpr (NT) __builtin_nontemporal_store(v, reinterpret_cast<P2*>(p));
  else *reinterpret_cast<P2*>(p) = v;
}

// unit u -> (zc, yt, xt): XCD k = blockIdx % 8 owns units [k per, (k + 1) per) of the zc-major order
__device__ int g_map;  // 0: as above; 1: XCD k owns the y-tiles [k YT/8, (k+1) YT/8), order (xt, yt, zc); 2: plain order
__device__ inline bool decode(int ZCH, int YT, int XT, int& zc, int& yt, int& xt) {
  const int per = (ZCH * YT * XT + 7) / 8;
  const int k = blockIdx.x % 8, i = blockIdx.x / 8;
  if (g_map == 1) {
    const int chunk = YT / 8;
    xt = i % XT;
    const int r = i / XT;
    yt = k * chunk + r % chunk;
    zc = r / chunk;
    return zc < ZCH;
  }
  if (g_map == 2) {
    const int u = blockIdx.x;
    xt = u % XT;
    yt = (u / XT) % YT;
    zc = u / XT / YT;
    return zc < ZCH;
  }
  const int u = k * per + i;
  xt = u % XT;
  const int r = u / XT;
  yt = r % YT;
  zc = r / YT;
  return i < per && zc < ZCH;
}

template <int TY, int TX, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__(kBlock) void k_tile_own(const double* __restrict__ fu, double* __restrict__ x,
                                                     double* __restrict__ m, double* __restrict__ v, int fnz, int fny,
                                                     int fnx, int ZC) {
  extern __shared__ char pad[];
  constexpr int NL = 2 * TY * TX / kBlock;  // own packs per thread and plane
  const int YT = fny / (2 * TY), XT = fnx / (2 * TX), ZCH = (fnz / 2 + ZC - 1) / ZC;
  int zc, yt, xt;
  if (!decode(ZCH, YT, XT, zc, yt, xt)) return;
  const int64_t fplane = (int64_t)fny * fnx;
  int64_t off[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int o = threadIdx.x + i * kBlock;
    const int row = o / TX, cc = o - row * TX;
    off[i] = (int64_t)(2 * yt * TY + row) * fnx + 2 * xt * TX + 2 * cc;
  }
  const int z0 = 2 * zc * ZC, z1 = min(fnz, z0 + 2 * ZC);
  P2 f[NL], a[NL], b[NL], c[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int64_t o = (int64_t)z0 * fplane + off[i];
    f[i] = ld<NTL>(fu + o);
    a[i] = ld<NTL>(x + o);
    b[i] = ld<NTL>(m + o);
    c[i] = ld<NTL>(v + o);
  }
  for (int z = z0; z < z1; ++z) {
    P2 fn[NL], an[NL], bn[NL], cn[NL];
    const int zn = z + 1 < z1 ? z + 1 : z;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int64_t o = (int64_t)zn * fplane + off[i];
      fn[i] = ld<NTL>(fu + o);
      an[i] = ld<NTL>(x + o);
      bn[i] = ld<NTL>(m + o);
      cn[i] = ld<NTL>(v + o);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int64_t o = (int64_t)z * fplane + off[i];
      b[i] = b[i] + 0.1 * (f[i] - b[i]);
      c[i] = c[i] + 0.001 * (f[i] * f[i] - c[i]);
      a[i] = a[i] - 1e-3 * b[i];
      st<NTS>(a[i], x + o);
      st<NTS>(b[i], m + o);
      st<NTS>(c[i], v + o);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) f[i] = fn[i], a[i] = an[i], b[i] = bn[i], c[i] = cn[i];
  }
}

// C: fu window (2TY + 2 rows, TX + 2 packs) per plane in a 4-slot LDS ring; g = 7-point sum; planes z0 - 1 .. z1
template <int TY, int TX, bool NTL = true, bool NTS = true>
__global__ __launch_bounds__(kBlock) void k_tile_ring(const double* __restrict__ fu, double* __restrict__ x,
                                                      double* __restrict__ m, double* __restrict__ v, int fnz, int fny,
                                                      int fnx, int ZC) {
  extern __shared__ char smem[];
  constexpr int NL = 2 * TY * TX / kBlock;
  constexpr int WR = 2 * TY + 2, WC = TX + 2, WP = WR * WC, WL = (WP + kBlock - 1) / kBlock;
  P2* ring = reinterpret_cast<P2*>(smem);
  const int YT = fny / (2 * TY), XT = fnx / (2 * TX), ZCH = (fnz / 2 + ZC - 1) / ZC;
  int zc, yt, xt;
  if (!decode(ZCH, YT, XT, zc, yt, xt)) return;
  const int64_t fplane = (int64_t)fny * fnx;
  int64_t off[NL], src[WL];
  int at[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int o = threadIdx.x + i * kBlock;
    const int row = o / TX, cc = o - row * TX;
    off[i] = (int64_t)(2 * yt * TY + row) * fnx + 2 * xt * TX + 2 * cc;
    at[i] = (row + 1) * WC + cc + 1;
  }
#pragma unroll
  for (int i = 0; i < WL; ++i) {
    int p = threadIdx.x + i * kBlock;
    p = p < WP ? p : WP - 1;
    const int r = p / WC, cc = p - r * WC;
    int fy = 2 * yt * TY - 1 + r, fx = 2 * xt * TX - 2 + 2 * cc;
    fy = fy < 0 ? 0 : (fy >= fny ? fny - 1 : fy);
    fx = fx < 0 ? 0 : (fx >= fnx ? fnx - 2 : fx);
    src[i] = (int64_t)fy * fnx + fx;
  }
  const int z0 = 2 * zc * ZC, z1 = min(fnz, z0 + 2 * ZC);
  auto clampz = [&](int z) { return z < 0 ? 0 : (z >= fnz ? fnz - 1 : z); };
  P2 pre[WL];
  // prime: planes z0 - 1 and z0 into the ring, z0 + 1 in flight
  for (int z = z0 - 1; z <= z0; ++z) {
#pragma unroll
    for (int i = 0; i < WL; ++i) pre[i] = *reinterpret_cast<const P2*>(fu + (int64_t)clampz(z) * fplane + src[i]);
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int p = threadIdx.x + i * kBlock;
      if (p < WP) ring[((z + 4) & 3) * WP + p] = pre[i];
    }
  }
#pragma unroll
  for (int i = 0; i < WL; ++i) pre[i] = *reinterpret_cast<const P2*>(fu + (int64_t)clampz(z0 + 1) * fplane + src[i]);
  P2 a[NL], b[NL], c[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) {
    const int64_t o = (int64_t)z0 * fplane + off[i];
    a[i] = ld<NTL>(x + o);
    b[i] = ld<NTL>(m + o);
    c[i] = ld<NTL>(v + o);
  }
  for (int z = z0; z < z1; ++z) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WL; ++i) {
      const int p = threadIdx.x + i * kBlock;
      if (p < WP) ring[((z + 1 + 4) & 3) * WP + p] = pre[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < WL; ++i) pre[i] = *reinterpret_cast<const P2*>(fu + (int64_t)clampz(z + 2) * fplane + src[i]);
    P2 an[NL], bn[NL], cn[NL];
    const int zn = z + 1 < z1 ? z + 1 : z;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int64_t o = (int64_t)zn * fplane + off[i];
      an[i] = ld<NTL>(x + o);
      bn[i] = ld<NTL>(m + o);
      cn[i] = ld<NTL>(v + o);
    }
    const P2* pm = ring + ((z - 1 + 4) & 3) * WP;
    const P2* pc = ring + ((z + 4) & 3) * WP;
    const P2* pp = ring + ((z + 1 + 4) & 3) * WP;
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int64_t o = (int64_t)z * fplane + off[i];
      const P2 fc = pc[at[i]], fl = pc[at[i] - 1], fr = pc[at[i] + 1];
      P2 g = pm[at[i]] + pp[at[i]] + pc[at[i] - WC] + pc[at[i] + WC] - 6.0 * fc;
      g[0] += fl[1] + fc[1];
      g[1] += fc[0] + fr[0];
      b[i] = b[i] + 0.1 * (g - b[i]);
      c[i] = c[i] + 0.001 * (g * g - c[i]);
      a[i] = a[i] - 1e-3 * b[i];
      st<NTS>(a[i], x + o);
      st<NTS>(b[i], m + o);
      st<NTS>(c[i], v + o);
    }
#pragma unroll
    for (int i = 0; i < NL; ++i) a[i] = an[i], b[i] = bn[i], c[i] = cn[i];
  }
}

template <typename F>
static float time_ms(F launch, int reps = 10) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  CHECK(hipGetLastError());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int N = 512;
  const int64_t n = (int64_t)N * N * N;
  double *fu, *x, *m, *v;
  CHECK(hipMalloc(&fu, n * 8));
  CHECK(hipMalloc(&x, n * 8));
  CHECK(hipMalloc(&m, n * 8));
  CHECK(hipMalloc(&v, n * 8));
  CHECK(hipMemset(fu, 0, n * 8));
  CHECK(hipMemset(x, 0, n * 8));
  CHECK(hipMemset(m, 0, n * 8));
  CHECK(hipMemset(v, 0, n * 8));
  const double gb = 7.0 * n * 8 / 1e9;
  CHECK(hipFuncSetAttribute((const void*)k_flat, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  for (int grid : {4096, 16384, 65536})
    for (int pad : {0, 78 * 1024}) {
      const float ms = time_ms([&] { hipLaunchKernelGGL(k_flat, dim3(grid), dim3(kBlock), pad, 0, fu, x, m, v, n / 2); });
      printf("A flat        grid %5d lds %3d KB: %.3f ms  %.2f TB/s (7 words)\n", grid, pad / 1024, ms, gb / ms);
    }
  auto set_map = [&](int mp) { CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_map), &mp, sizeof(int))); };
#define RUN(KERNEL, NAME, TY, TX, NTL, NTS, BASE_LDS)                                                                \
  for (int mp : {0, 1, 2})                                                                                           \
    for (int zc : {16, 64})                                                                                          \
      for (int extra : {0, 40 * 1024, 70 * 1024}) {                                                                  \
        if (mp == 1 && (N / (2 * TY)) % 8) continue;                                                                 \
        const int lds = BASE_LDS + extra;                                                                            \
        if (lds < 52 * 1024 && extra) continue;                                                                      \
        set_map(mp);                                                                                                 \
        CHECK(hipFuncSetAttribute((const void*)KERNEL<TY, TX, NTL, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                  160 * 1024));                                                                      \
        const int units = (N / 2 / zc) * (N / (2 * TY)) * (N / (2 * TX));                                            \
        const int grid = ((units + 7) / 8) * 8;                                                                      \
        const float ms = time_ms([&] {                                                                               \
          hipLaunchKernelGGL((KERNEL<TY, TX, NTL, NTS>), dim3(grid), dim3(kBlock), lds, 0, fu, x, m, v, N, N, N, zc); \
        });                                                                                                          \
        printf("%s %2dx%-3d nt %d%d map %d ZC %2d lds %3d KB: %.3f ms  %.2f TB/s\n", NAME, TY, TX, NTL, NTS, mp, zc, \
               lds / 1024, ms, gb / ms);                                                                             \
      }
#define RING_LDS(TY, TX) (4 * (2 * TY + 2) * (TX + 2) * 16)
  RUN(k_tile_own, "B own ", 16, 16, true, true, 52 * 1024)
  RUN(k_tile_own, "B own ", 8, 32, true, true, 52 * 1024)
  RUN(k_tile_own, "B own ", 8, 32, false, false, 52 * 1024)
  RUN(k_tile_own, "B own ", 8, 32, true, false, 52 * 1024)
  RUN(k_tile_own, "B own ", 8, 32, false, true, 52 * 1024)
  RUN(k_tile_own, "B own ", 4, 64, true, true, 52 * 1024)
  RUN(k_tile_own, "B own ", 2, 128, true, true, 52 * 1024)
  RUN(k_tile_own, "B own ", 1, 256, true, true, 52 * 1024)
  RUN(k_tile_ring, "C ring", 16, 16, true, true, RING_LDS(16, 16))
  RUN(k_tile_ring, "C ring", 8, 32, true, true, RING_LDS(8, 32))
  RUN(k_tile_ring, "C ring", 8, 32, false, false, RING_LDS(8, 32))
  RUN(k_tile_ring, "C ring", 4, 64, true, true, RING_LDS(4, 64))
  return 0;
}
