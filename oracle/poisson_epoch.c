/* TEST / MEASUREMENT INFRASTRUCTURE -- never linked into or called by the product (odil_amd/).
 *
 * Plain-C restatement of the reference's hot loop for the headline workload: ONE Adam epoch of the 3-D Poisson problem
 * with the multigrid decomposition, float64, cell-centred fields, one thread, no SIMD intrinsics:
 *
 *   u    = sum_l P^l w_l                      reference src/odil/core.py:245-263, :606-700 (joint ghost rule :640-643)
 *   fu   = Lap(u) - rhs, zero-Dirichlet       reference examples/poisson/poisson.py:57-68, :89-113; core.py:1439-1445
 *   loss = mean(fu^2)                         reference core.py:1093-1095
 *   g_0  = A^T (2 fu / n), g_l = P^T g_{l-1}  what reverse mode yields, core.py:1100
 *   Adam                                      reference src/odil/optimizer.py:311-319
 *
 * It exists so that bench.py's `cpu_baseline` can be timed at the headline's own size (512^3: the NumPy oracle needs
 * minutes per epoch there, most of it page faults of temporaries); it is pinned against the NumPy oracle
 * (oracle/odil_np.py, itself pinned on the reference's golden vectors) by tests/test_oracle_c.py to round-off.
 *
 *   cc -O3 -shared -fPIC -o _build/libpoisson_epoch.so poisson_epoch.c -lm      (oracle/Makefile)
 *   cc -O3 -DODIL_C_MAIN -o _build/poisson_epoch poisson_epoch.c -lm;  ./poisson_epoch N seconds [start_unix_time [epochs]]
 *   cc -O3 -fopenmp -DODIL_C_MAIN -o _build/poisson_epoch_omp ...: the same loops shared among OMP_NUM_THREADS threads --
 *   ONE problem on all host cores (bench.py's `cpu_baseline.all_cores`).  Without -fopenmp every pragma is ignored and the
 *   code is the serial restatement the tests pin; with it only the ORDER of a few sums changes (the two-colour transpose of
 *   the prolongation, the reduction of the loss): tests/test_oracle_c.py holds the threaded binary to the serial one at 1e-12.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef int64_t i64;

static inline i64 clampi(i64 j, i64 n) { return j < 0 ? 0 : (j >= n ? n - 1 : j); }
static inline i64 reflecti(i64 j, i64 n) { return j < 0 ? -j : (j >= n ? 2 * n - 2 - j : j); } /* -1 -> 1, n -> n - 2 */

/* upad = 2 * pad_symmetric(u) - pad_reflect(u), one ghost layer on every axis (core.py:640-643) */
static void make_upad(const double* u, i64 n0, i64 n1, i64 n2, double* up) {
  const i64 p1 = n1 + 2, p2 = n2 + 2;
#pragma omp parallel for schedule(static)
  for (i64 a = -1; a <= n0; ++a)
    for (i64 b = -1; b <= n1; ++b)
      for (i64 c = -1; c <= n2; ++c) {
        const int inside = a >= 0 && a < n0 && b >= 0 && b < n1 && c >= 0 && c < n2;
        double v;
        if (inside) {
          v = u[(a * n1 + b) * n2 + c];
        } else {
          const double s = u[(clampi(a, n0) * n1 + clampi(b, n1)) * n2 + clampi(c, n2)];
          const double r = u[(reflecti(a, n0) * n1 + reflecti(b, n1)) * n2 + reflecti(c, n2)];
          v = 2 * s - r;
        }
        up[((a + 1) * p1 + (b + 1)) * p2 + (c + 1)] = v;
      }
}

/* fine = add + P(coarse); coarse (n0, n1, n2) -> fine (2 n0, 2 n1, 2 n2); work: (n0 + 2)(n1 + 2)(n2 + 2) doubles.
 * Fine index 2 i + s reads padded indices (i + s) + r, r in {0, 1}, weight 3 where r == 1 - s else 1 (core.py:675-687). */
void odil_c_interp_add(const double* coarse, i64 n0, i64 n1, i64 n2, const double* add, double* fine, double* work) {
  make_upad(coarse, n0, n1, n2, work);
  const i64 p1 = n1 + 2, p2 = n2 + 2, f1 = 2 * n1, f2 = 2 * n2;
#pragma omp parallel for schedule(static)
  for (i64 a = 0; a < 2 * n0; ++a) {
    const i64 ia = a >> 1, sa = a & 1;
    for (i64 b = 0; b < f1; ++b) {
      const i64 ib = b >> 1, sb = b & 1;
      for (i64 c = 0; c < f2; ++c) {
        const i64 ic = c >> 1, sc = c & 1;
        double acc = 0;
        for (int ra = 0; ra < 2; ++ra)
          for (int rb = 0; rb < 2; ++rb)
            for (int rc = 0; rc < 2; ++rc) {
              const int w = (ra == 1 - sa ? 3 : 1) * (rb == 1 - sb ? 3 : 1) * (rc == 1 - sc ? 3 : 1);
              acc += w * work[((ia + sa + ra) * p1 + (ib + sb + rb)) * p2 + (ic + sc + rc)];
            }
        const i64 f = (a * f1 + b) * f2 + c;
        fine[f] = (add ? add[f] : 0.0) + acc / 64;
      }
    }
  }
}

/* gcoarse = P^T gfine: the exact transpose of the above (scatter onto the padded grid, then the ghost rule's transpose) */
void odil_c_interp_adj(const double* gfine, i64 n0, i64 n1, i64 n2, double* gcoarse, double* work) {
  const i64 p0 = n0 + 2, p1 = n1 + 2, p2 = n2 + 2, f1 = 2 * n1, f2 = 2 * n2;
  memset(work, 0, sizeof(double) * p0 * p1 * p2);
#ifdef _OPENMP
  /* fine planes 2 k - 1 and 2 k scatter onto the padded planes k and k + 1 only: the units k of one parity touch disjoint
   * planes and run concurrently, the two parities one after the other */
  for (int colour = 0; colour < 2; ++colour) {
#pragma omp parallel for schedule(static)
    for (i64 k = colour; k <= n0; k += 2)
      for (i64 a = (2 * k - 1 < 0 ? 0 : 2 * k - 1); a <= 2 * k && a < 2 * n0; ++a) {
        const i64 ia = a >> 1, sa = a & 1;
        for (i64 b = 0; b < f1; ++b) {
          const i64 ib = b >> 1, sb = b & 1;
          for (i64 c = 0; c < f2; ++c) {
            const i64 ic = c >> 1, sc = c & 1;
            const double g = gfine[(a * f1 + b) * f2 + c] / 64;
            for (int ra = 0; ra < 2; ++ra)
              for (int rb = 0; rb < 2; ++rb)
                for (int rc = 0; rc < 2; ++rc) {
                  const int w = (ra == 1 - sa ? 3 : 1) * (rb == 1 - sb ? 3 : 1) * (rc == 1 - sc ? 3 : 1);
                  work[((ia + sa + ra) * p1 + (ib + sb + rb)) * p2 + (ic + sc + rc)] += w * g;
                }
          }
        }
      }
  }
#else
  for (i64 a = 0; a < 2 * n0; ++a) {
    const i64 ia = a >> 1, sa = a & 1;
    for (i64 b = 0; b < f1; ++b) {
      const i64 ib = b >> 1, sb = b & 1;
      for (i64 c = 0; c < f2; ++c) {
        const i64 ic = c >> 1, sc = c & 1;
        const double g = gfine[(a * f1 + b) * f2 + c] / 64;
        for (int ra = 0; ra < 2; ++ra)
          for (int rb = 0; rb < 2; ++rb)
            for (int rc = 0; rc < 2; ++rc) {
              const int w = (ra == 1 - sa ? 3 : 1) * (rb == 1 - sb ? 3 : 1) * (rc == 1 - sc ? 3 : 1);
              work[((ia + sa + ra) * p1 + (ib + sb + rb)) * p2 + (ic + sc + rc)] += w * g;
            }
      }
    }
  }
#endif
  memset(gcoarse, 0, sizeof(double) * n0 * n1 * n2);
#ifdef _OPENMP
  /* a padded plane 0 <= a < n0 folds onto coarse plane a only: those run concurrently; the ghost planes a = -1 and a = n0
   * fold onto the planes 0, 1 / n0 - 1, n0 - 2 and run last, alone */
#pragma omp parallel for schedule(static)
  for (i64 a = 0; a < n0; ++a)
    for (i64 b = -1; b <= n1; ++b)
      for (i64 c = -1; c <= n2; ++c) {
        const double g = work[((a + 1) * p1 + (b + 1)) * p2 + (c + 1)];
        const int inside = b >= 0 && b < n1 && c >= 0 && c < n2;
        if (inside) {
          gcoarse[(a * n1 + b) * n2 + c] += g;
        } else {
          gcoarse[(a * n1 + clampi(b, n1)) * n2 + clampi(c, n2)] += 2 * g;
          gcoarse[(a * n1 + reflecti(b, n1)) * n2 + reflecti(c, n2)] -= g;
        }
      }
  for (i64 a = -1; a <= n0; a += n0 + 1)
#else
  for (i64 a = -1; a <= n0; ++a)
#endif
    for (i64 b = -1; b <= n1; ++b)
      for (i64 c = -1; c <= n2; ++c) {
        const double g = work[((a + 1) * p1 + (b + 1)) * p2 + (c + 1)];
        const int inside = a >= 0 && a < n0 && b >= 0 && b < n1 && c >= 0 && c < n2;
        if (inside) {
          gcoarse[(a * n1 + b) * n2 + c] += g;
        } else {
          gcoarse[(clampi(a, n0) * n1 + clampi(b, n1)) * n2 + clampi(c, n2)] += 2 * g;
          gcoarse[(reflecti(a, n0) * n1 + reflecti(b, n1)) * n2 + reflecti(c, n2)] -= g;
        }
      }
}

/* one axis of the Laplacian with the quadratic zero-Dirichlet ghosts (poisson.py:57-68, core.py:1439-1445) */
static inline double axis_term(double q, double qwm, double qwp, int lo, int hi, double h2) {
  const double qm = lo ? (qwp - 6 * q + 8 * 0.0) / 3 : qwm;
  const double qp = hi ? (qwm - 6 * q + 8 * 0.0) / 3 : qwp;
  return (qp - 2 * q + qm) / h2;
}

/* fu = Lap(u) - rhs; returns sum(fu^2).  The rolls of the reference are periodic: the wrapped values are read and then
 * discarded by the wall masks, as here. */
double odil_c_residual(const double* u, const double* rhs, i64 n0, i64 n1, i64 n2, const double* h2, double* fu) {
  double sum = 0;
#pragma omp parallel for schedule(static) reduction(+ : sum)
  for (i64 a = 0; a < n0; ++a)
    for (i64 b = 0; b < n1; ++b)
      for (i64 c = 0; c < n2; ++c) {
        const i64 i = (a * n1 + b) * n2 + c;
        const double q = u[i];
        const double am = u[(((a + n0 - 1) % n0) * n1 + b) * n2 + c], ap = u[(((a + 1) % n0) * n1 + b) * n2 + c];
        const double bm = u[(a * n1 + (b + n1 - 1) % n1) * n2 + c], bp = u[(a * n1 + (b + 1) % n1) * n2 + c];
        const double cm = u[(a * n1 + b) * n2 + (c + n2 - 1) % n2], cp = u[(a * n1 + b) * n2 + (c + 1) % n2];
        const double f = axis_term(q, am, ap, a == 0, a == n0 - 1, h2[0]) + axis_term(q, bm, bp, b == 0, b == n1 - 1, h2[1]) +
                         axis_term(q, cm, cp, c == 0, c == n2 - 1, h2[2]) - (rhs ? rhs[i] : 0.0);
        fu[i] = f;
        sum += f * f;
      }
  return sum;
}

/* coefficients of one axis at index i of n: on u[i - 1], u[i], u[i + 1] (oracle/odil_np.py poisson_jac_coeffs) */
static inline void axis_coeffs(i64 i, i64 n, double h2, double* cm, double* c0, double* cp) {
  const int lo = i == 0, hi = i == n - 1;
  *cm = ((lo ? 0.0 : 1.0) + (hi ? 1.0 / 3 : 0.0)) / h2;
  *cp = ((hi ? 0.0 : 1.0) + (lo ? 1.0 / 3 : 0.0)) / h2;
  *c0 = (-2.0 + (lo ? -2.0 : 0.0) + (hi ? -2.0 : 0.0)) / h2;
}

/* gu = scale * A^T fu (gather form: row j collects what the rows j - 1, j, j + 1 of A hold in column j) */
void odil_c_adjoint(const double* fu, i64 n0, i64 n1, i64 n2, const double* h2, double scale, double* gu) {
  const i64 n[3] = {n0, n1, n2}, st[3] = {n1 * n2, n2, 1};
#pragma omp parallel for schedule(static)
  for (i64 a = 0; a < n0; ++a)
    for (i64 b = 0; b < n1; ++b)
      for (i64 c = 0; c < n2; ++c) {
        const i64 idx[3] = {a, b, c}, i = (a * n1 + b) * n2 + c;
        double acc = 0;
        for (int d = 0; d < 3; ++d) {
          double cm, c0, cp, t0, t1, t2;
          axis_coeffs(idx[d], n[d], h2[d], &cm, &c0, &cp);
          acc += c0 * fu[i];
          if (idx[d] + 1 < n[d]) {  /* row j + 1 reads u[j] with its c_m */
            axis_coeffs(idx[d] + 1, n[d], h2[d], &t0, &t1, &t2);
            acc += t0 * fu[i + st[d]];
          }
          if (idx[d] >= 1) {  /* row j - 1 reads u[j] with its c_p */
            axis_coeffs(idx[d] - 1, n[d], h2[d], &t0, &t1, &t2);
            acc += t2 * fu[i - st[d]];
          }
        }
        gu[i] = scale * acc;
      }
}

void odil_c_adam(double* x, double* m, double* v, const double* g, i64 n, double alpha, double b1, double b2, double eps) {
#pragma omp parallel for schedule(static)
  for (i64 i = 0; i < n; ++i) {
    m[i] = m[i] + (g[i] - m[i]) * (1 - b1);
    v[i] = v[i] + (g[i] * g[i] - v[i]) * (1 - b2);
    x[i] = x[i] - (m[i] * alpha) / (sqrt(v[i]) + eps);
  }
}

/* One epoch on an N^3 grid with nlvl levels (N, N/2, ...): x, m, v, g are arrays of level pointers; u, fu, work: N^3,
 * N^3 and (N/2 + 2)^3 ... scratch (work needs (N/2 + 2)^3 doubles, lvl: two buffers of (N/2)^3 for the chain).
 * Returns the loss; `epoch` is the 1-based local epoch of the bias correction (optimizer.py:313-315). */
double odil_c_epoch(i64 N, int nlvl, double** x, double** m, double** v, double** g, const double* rhs, double* u, double* fu,
                    double* work, double* lvl_a, double* lvl_b, int epoch, double lr) {
  i64 n[32];
  for (int l = 0; l < nlvl; ++l) n[l] = N >> l;
  /* synthesis, coarsest first */
  const double* res = x[nlvl - 1];
  for (int l = nlvl - 2; l >= 0; --l) {
    double* out = l == 0 ? u : ((nlvl - l) % 2 ? lvl_a : lvl_b);
    odil_c_interp_add(res, n[l + 1], n[l + 1], n[l + 1], x[l], out, work);
    res = out;
  }
  if (nlvl == 1) memcpy(u, x[0], sizeof(double) * N * N * N);
  const double h = 1.0 / (double)N, h2[3] = {h * h, h * h, h * h};
  const double cells = (double)N * (double)N * (double)N;
  const double loss = odil_c_residual(u, rhs, N, N, N, h2, fu) / cells;
  odil_c_adjoint(fu, N, N, N, h2, 2.0 / cells, g[0]);
  for (int l = 1; l < nlvl; ++l) odil_c_interp_adj(g[l - 1], n[l], n[l], n[l], g[l], work);
  const double b1 = 0.9, b2 = 0.999, eps = 1e-7;
  const double alpha = lr * sqrt(1 - pow(b2, (double)epoch)) / (1 - pow(b1, (double)epoch));
  for (int l = 0; l < nlvl; ++l) odil_c_adam(x[l], m[l], v[l], g[l], n[l] * n[l] * n[l], alpha, b1, b2, eps);
  return loss;
}

#ifdef ODIL_C_MAIN
#include <sys/mman.h>

/* zeroed, 2-MiB aligned, transparent huge pages requested, every page touched before the timed region (first-touch
 * page faults are the host's cost, not the algorithm's) */
static double* zalloc(i64 count) {
  const size_t huge = (size_t)2 << 20, bytes = (((size_t)count * 8 + huge - 1) / huge) * huge;
  void* p = NULL;
  if (posix_memalign(&p, huge, bytes)) exit(2);
  madvise(p, bytes, MADV_HUGEPAGE);
  /* first touch by the threads that will work on the pages (static schedule, as the loops): NUMA placement */
#pragma omp parallel for schedule(static)
  for (i64 pg = 0; pg < (i64)(bytes / huge); ++pg) memset((char*)p + (size_t)pg * huge, 0, huge);
  return (double*)p;
}

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

/* the synthetic inputs of the headline: ref_u = 'hat' (poisson.py:18-24), rhs = its discrete Laplacian (:71-86), zero start */
int main(int argc, char** argv) {
  const i64 N = argc > 1 ? atoll(argv[1]) : 64;
  const double budget = argc > 2 ? atof(argv[2]) : 5.0;
  int nlvl = 0;
  while ((N >> nlvl) >= 2 && ((N >> nlvl) << nlvl) == N) ++nlvl; /* round(log2 N) levels: N ... 2 (core.py:66-73) */
  const i64 cells = N * N * N;
  double *x[32], *m[32], *v[32], *g[32];
  for (int l = 0; l < nlvl; ++l) {
    const i64 k = (N >> l) * (N >> l) * (N >> l);
    x[l] = zalloc(k), m[l] = zalloc(k), v[l] = zalloc(k), g[l] = zalloc(k);
  }
  double* u = zalloc(cells), *fu = zalloc(cells), *rhs = zalloc(cells);
  const i64 half = N / 2;
  double* work = zalloc((half + 2) * (half + 2) * (half + 2)), *la = zalloc(half * half * half), *lb = zalloc(half * half * half);
  const double h = 1.0 / (double)N, h2[3] = {h * h, h * h, h * h};
#pragma omp parallel for schedule(static)
  for (i64 a = 0; a < N; ++a)
    for (i64 b = 0; b < N; ++b)
      for (i64 c = 0; c < N; ++c) {
        const double xa = (a + 0.5) * h, xb = (b + 0.5) * h, xc = (c + 0.5) * h;
        const double w = ((1 - xa) * xa * 5) * ((1 - xb) * xb * 5) * ((1 - xc) * xc * 5), w5 = pow(w, 5);
        u[(a * N + b) * N + c] = pow(w5 / (1 + w5), 0.2);
      }
  odil_c_residual(u, NULL, N, N, N, h2, rhs);
  int done = 0;
  double loss = 0;
  if (cells < 100000000) loss = odil_c_epoch(N, nlvl, x, m, v, g, rhs, u, fu, work, la, lb, ++done, 0.005); /* warm-up */
  if (argc > 3) {
    const double start = atof(argv[3]);
    while (now() < start) {
    }
  }
  const int fixed = argc > 4 ? atoi(argv[4]) : 0; /* exactly this many epochs (tests), else by the time budget */
  const double t0 = now();
  int k = 0;
  double el;
  do {
    loss = odil_c_epoch(N, nlvl, x, m, v, g, rhs, u, fu, work, la, lb, ++done, 0.005);
    ++k;
    el = now() - t0;
  } while (fixed ? k < fixed : (el < budget && k < 200));
  printf("{\"cells\": %lld, \"epochs\": %d, \"seconds\": %.6f, \"loss\": %.17g, \"levels\": %d}\n", (long long)cells, k, el, loss, nlvl);
  return 0;
}
#endif
