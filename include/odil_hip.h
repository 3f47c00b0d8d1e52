/*
 * odil_hip.h -- C-ABI of the MI355X-native (gfx950) ODIL hot path.
 *
 * The reference (cselab/odil v0.1.8) is pure Python and has no FFI boundary of its
 * own: its device arithmetic goes through the `mod` namespace into XLA / TensorFlow
 * (reference src/odil/backend.py:12-317).  This library is what a ROCm `mod` +
 * `Problem` bind instead.  Each entry point below cites the reference code whose
 * arithmetic it replaces (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - `extern "C"`, plain pointers and sizes; no torch / C++ types.
 *   - every data pointer is a DEVICE pointer into caller-owned memory (the Python
 *     host passes torch tensor storage); the library never allocates, frees or
 *     retains device memory; scratch space is passed in explicitly.
 *   - arrays are C-order, contiguous; `shape` is the ARRAY shape (not cells),
 *     `ndim` <= ODIL_MAX_NDIM; `loc` is ODIL's location string, one of 'c' (cell),
 *     'n' (node), '.' (axis not refined) per axis (reference core.py:606-616).
 *   - `stream` is a `hipStream_t` passed as `void*` (NULL = default stream).
 *     Kernels are enqueued and the call returns; no host synchronisation inside.
 *   - return value: 0 on success, <0 on error (ODIL_E_*); the message is
 *     available from `odil_last_error()` (thread-local).
 *   - suffix `_f32` / `_f64` selects the arithmetic type (reference runtime.py:77).
 */
#ifndef ODIL_HIP_H
#define ODIL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODIL_MAX_NDIM 4
#define ODIL_MAX_LEVELS 32
#define ODIL_E_INVAL (-1)  /* bad argument */
#define ODIL_E_LAUNCH (-2) /* HIP launch / runtime failure */
#define ODIL_E_NODEV (-3)  /* no usable gfx950 device */

/* ---- library ---------------------------------------------------------------- */
const char* odil_last_error(void);
int odil_version(void);
/* Number of HIP devices visible, or <0.  Does not create a context. */
int odil_device_count(void);
/* Bytes of scratch every reduction-carrying call needs (`partials`). */
size_t odil_reduce_workspace_bytes(void);

/* ---- multigrid transfers (reference core.py:606-755) ------------------------- */
/* fine = add_scale * add + P(coarse_scale * coarse); `add` may be NULL (then fine =
 * P(coarse)).  Replaces `interp_to_finer(method="stack")` (core.py:606-700) and one
 * step of `Domain.multigrid_to_regular` (core.py:258-262).  `cshape` = coarse array
 * shape; fine shape per axis: 'c' 2n, 'n' 2n-1, '.' n. */
int odil_interp_add_f64(const double* coarse, const double* add, double* fine, const int64_t* cshape, int ndim,
                        const char* loc, double coarse_scale, double add_scale, void* stream);
int odil_interp_add_f32(const float* coarse, const float* add, float* fine, const int64_t* cshape, int ndim,
                        const char* loc, float coarse_scale, float add_scale, void* stream);
/* The same with the coarse operand a VIEW: `coarse_ld` elements lie between consecutive indices of its leading axis
 * (>= the product of its other extents; a ghost-extended level array of the slab paths without its outer ghost planes --
 * no reference counterpart).  Served by the marching kernels of the 4-D layouts ('nccc', '.ccc'); returns 1,
 * having launched nothing, for any other layout or size (the caller then passes a contiguous copy to odil_interp_add). */
int odil_interp_add_ld_f64(const double* coarse, int64_t coarse_ld, const double* add, double* fine, const int64_t* cshape,
                           int ndim, const char* loc, double coarse_scale, double add_scale, void* stream);
int odil_interp_add_ld_f32(const float* coarse, int64_t coarse_ld, const float* add, float* fine, const int64_t* cshape,
                           int ndim, const char* loc, float coarse_scale, float add_scale, void* stream);
/* gcoarse = P^T gfine (the cotangent autodiff produces for core.py:606-700);
 * if `gscaled` != NULL also gscaled = scale * gcoarse. */
int odil_interp_adj_f64(const double* gfine, double* gcoarse, double* gscaled, const int64_t* cshape, int ndim,
                        const char* loc, double scale, void* stream);
int odil_interp_adj_f32(const float* gfine, float* gcoarse, float* gscaled, const int64_t* cshape, int ndim,
                        const char* loc, float scale, void* stream);
/* gcoarse = P^T gfine written into a VIEW with `gcoarse_ld` elements between consecutive leading indices (see
 * odil_interp_add_ld; also the time part 'n...' of the two-step transpose of 'nccc' arrays).  Returns 1, having launched
 * nothing, where the layout is not served. */
int odil_interp_adj_ld_f64(const double* gfine, double* gcoarse, int64_t gcoarse_ld, const int64_t* cshape, int ndim,
                           const char* loc, void* stream);
int odil_interp_adj_ld_f32(const float* gfine, float* gcoarse, int64_t gcoarse_ld, const int64_t* cshape, int ndim,
                           const char* loc, void* stream);
/* Slab decomposition (no reference counterpart): the same transpose for an array
 * whose axis 0 is CUT at its low / high end -- that end carries ghost planes of the
 * neighbouring rank instead of being a wall, so the boundary (ghost-rule) weights are not
 * applied there. */
int odil_interp_adj_cut_f64(const double* gfine, double* gcoarse, double* gscaled, const int64_t* cshape, int ndim,
                            const char* loc, double scale, int cut_lo, int cut_hi, void* stream);
int odil_interp_adj_cut_f32(const float* gfine, float* gcoarse, float* gscaled, const int64_t* cshape, int ndim,
                            const char* loc, float scale, int cut_lo, int cut_hi, void* stream);
/* ... and with the Adam update of the coarse array (x, m, v of gcoarse's shape) fused in, as in
 * odil_mg_synth_adj_adam. */
int odil_interp_adj_cut_adam_f64(const double* gfine, double* gcoarse, const int64_t* cshape, int ndim, const char* loc,
                                 int cut_lo, int cut_hi, double* x, double* m, double* v, double alpha,
                                 double one_minus_b1, double one_minus_b2, double eps, const double* alpha_dev,
                                 void* stream);
int odil_interp_adj_cut_adam_f32(const float* gfine, float* gcoarse, const int64_t* cshape, int ndim, const char* loc,
                                 int cut_lo, int cut_hi, float* x, float* m, float* v, float alpha,
                                 float one_minus_b1, float one_minus_b2, float eps, const float* alpha_dev,
                                 void* stream);
/* coarse = R(fine): full weighting `restrict_to_coarser(method="conv")`
 * (core.py:703-755, backend.py:112-126).  `fshape` = fine array shape. */
int odil_restrict_f64(const double* fine, double* coarse, const int64_t* fshape, int ndim, const char* loc,
                      void* stream);
int odil_restrict_f32(const float* fine, float* coarse, const int64_t* fshape, int ndim, const char* loc,
                      void* stream);

/* gfine = R^T gcoarse: the cotangent autodiff routes through `restrict_to_coarser`
 * (`poisson --mgloss`, examples/poisson/poisson.py:116-122).  `fshape` = fine array shape. */
int odil_restrict_adj_f64(const double* gcoarse, double* gfine, const int64_t* fshape, int ndim, const char* loc,
                          void* stream);
int odil_restrict_adj_f32(const float* gcoarse, float* gfine, const int64_t* fshape, int ndim, const char* loc,
                          void* stream);

/* Strided VALID correlations with a small dense kernel: `mod.convolution` (backend.py:112-126 -> jax.lax.conv; called by
 * restrict_to_coarser, core.py:744-751) and `mod.conv_transpose` (backend.py:165-172 -> jax.lax.conv_transpose; called
 * by interp_to_finer(method="conv"), core.py:656-662) for user operators that call them directly (the framework's own
 * transfers use the dedicated kernels above).  Kernel extents and strides 1..4 per axis, ndim <= 4, taps summed in C
 * order of the kernel.
 *   transposed == 0:  out[o] = sum_k w[k] in[o * s + k],                 oshape = (ishape - wshape) / s + 1
 *   transposed != 0:  out[p] = sum_k w[k] in[(p - k) / s] over the taps with s | (p - k) and the quotient inside `in`
 *                     (the transpose of the above; oshape >= (ishape - 1) s + wshape, zeros beyond). */
int odil_conv_valid_f64(const double* in, const double* w, double* out, const int64_t* ishape, const int64_t* wshape,
                        const int64_t* strides, const int64_t* oshape, int ndim, int transposed, void* stream);
int odil_conv_valid_f32(const float* in, const float* w, float* out, const int64_t* ishape, const int64_t* wshape,
                        const int64_t* strides, const int64_t* oshape, int ndim, int transposed, void* stream);

/* u = sum_l P^l (factor_l * w_l): `Domain.multigrid_to_regular` (core.py:245-263).
 * terms / work / grads are HOST arrays of device pointers, factors / shapes HOST arrays.
 * terms[l]: level arrays fine->coarse, shapes[l*ndim..]: their array shapes,
 * work[l] (1 <= l <= nlvl-2): scratch of level l's size for the partial sums
 * (work[0] and work[nlvl-1] are ignored); `u` has level 0's shape. */
int odil_mg_synth_f64(const double* const* terms, const double* factors, double* const* work, double* u,
                      const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream);
int odil_mg_synth_f32(const float* const* terms, const float* factors, float* const* work, float* u,
                      const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream);
/* grads[l] = factor_l * (P^T)^l gu: transpose of the above (what jax.value_and_grad /
 * tf.GradientTape return for the level arrays, core.py:1100 / :1062).  work[l]
 * (1 <= l <= nlvl-1) is needed only for levels whose factor != 1. */
int odil_mg_synth_adj_f64(const double* gu, double* const* grads, const double* factors, double* const* work,
                          const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream);
int odil_mg_synth_adj_f32(const float* gu, float* const* grads, const float* factors, float* const* work,
                          const int64_t* shapes, int nlvl, int ndim, const char* loc, void* stream);

/* The same chain with the Adam update (optimizer.py:316-318) of every level l >= 1 whose x[l] is
 * non-NULL applied by the lane that forms grads[l][i] (x, m, v: HOST arrays of device pointers,
 * entry 0 ignored): the optimizer launch over the coarse levels and its re-read of the
 * gradients disappear.  grads are still written. */
int odil_mg_synth_adj_adam_f64(const double* gu, double* const* grads, const double* factors, double* const* work,
                               const int64_t* shapes, int nlvl, int ndim, const char* loc, double* const* x,
                               double* const* m, double* const* v, double alpha, double one_minus_b1,
                               double one_minus_b2, double eps, const double* alpha_dev, void* stream);
int odil_mg_synth_adj_adam_f32(const float* gu, float* const* grads, const float* factors, float* const* work,
                               const int64_t* shapes, int nlvl, int ndim, const char* loc, float* const* x,
                               float* const* m, float* const* v, float alpha, float one_minus_b1, float one_minus_b2,
                               float eps, const float* alpha_dev, void* stream);

/* ---- stencil access: Context.field (reference core.py:910-975) ---------------- */
/* out = trim(roll(pad(src), -shift)): 'c'->'n' zero-pad at the low end, periodic roll,
 * 'n'->'c' drop last (core.py:956-969).  `sshape` = source array shape. */
int odil_field_gather_f64(const double* src, double* out, const int64_t* sshape, int ndim, const char* field_loc,
                          const char* loc, const int64_t* shift, void* stream);
int odil_field_gather_f32(const float* src, float* out, const int64_t* sshape, int ndim, const char* field_loc,
                          const char* loc, const int64_t* shift, void* stream);
/* gsrc (+)= transpose of the gather applied to g; accumulate != 0 adds into gsrc. */
int odil_field_scatter_f64(const double* g, double* gsrc, const int64_t* sshape, int ndim, const char* field_loc,
                           const char* loc, const int64_t* shift, int accumulate, void* stream);
int odil_field_scatter_f32(const float* g, float* gsrc, const int64_t* sshape, int ndim, const char* field_loc,
                           const char* loc, const int64_t* shift, int accumulate, void* stream);

/* ---- loss reduction (reference core.py:1093-1095) ----------------------------- */
/* out[0] = mean(x^2) (square != 0) or mean(x) over n elements, accumulated in f64 in
 * a fixed order (deterministic).  `partials`: odil_reduce_workspace_bytes() scratch. */
int odil_mean_reduce_f64(const double* x, int64_t n, int square, double* partials, double* out, void* stream);
int odil_mean_reduce_f32(const float* x, int64_t n, int square, double* partials, float* out, void* stream);

/* ---- Poisson workload (reference examples/poisson/poisson.py:57-113) ---------- */
/* fu = sum_i (u+ - 2u + u-)/h2[i] - rhs with zero-Dirichlet ghosts by extrap_quadh
 * (poisson.py:57-68, core.py:1439-1445); loss[0] = mean(fu^2) (core.py:1093).
 * `fu` may be NULL (loss only).  `shape`: cell shape, ndim <= 3; h2[i] = step_i^2. */
int odil_poisson_residual_f64(const double* u, const double* rhs, double* fu, const int64_t* shape, int ndim,
                              const double* h2, double* partials, double* loss, void* stream);
int odil_poisson_residual_f32(const float* u, const float* rhs, float* fu, const int64_t* shape, int ndim,
                              const float* h2, double* partials, float* loss, void* stream);
/* Slab variant: fu on every plane of the (ghost-extended) local array, but
 * loss[0] = sum_{z0 <= z < z1} fu^2 / denom  (the rank's own planes over the GLOBAL size). */
int odil_poisson_residual_slab_f64(const double* u, const double* rhs, double* fu, const int64_t* shape, int ndim,
                                   const double* h2, int64_t z0, int64_t z1, double denom, double* partials,
                                   double* loss, void* stream);
int odil_poisson_residual_slab_f32(const float* u, const float* rhs, float* fu, const int64_t* shape, int ndim,
                                   const float* h2, int64_t z0, int64_t z1, double denom, double* partials,
                                   float* loss, void* stream);
/* Residual restricted to the next coarser grid in one pass (3-D, even extents, shape[2] a multiple of the
 * 16-byte pack): coarse[K,J,I] = scale * sum over the 2x2x2 fine cells of (A u - rhs), *loss = mean((A u -
 * rhs)^2).  Replaces odil_poisson_residual + odil_restrict in the V-cycle that solves the Newton system
 * (reference linsolver.py:61-72 uses pyamg there); `partials`: odil_reduce_workspace_bytes() scratch.  loss == NULL
 * (coarse levels of a cycle, whose norm nobody reads): the reduction launch is skipped. */
int odil_poisson_residual_restrict_f64(const double* u, const double* rhs, double* coarse, const int64_t* shape,
                                       int ndim, const double* h2, double scale, double* partials, double* loss,
                                       void* stream);
int odil_poisson_residual_restrict_f32(const float* u, const float* rhs, float* coarse, const int64_t* shape, int ndim,
                                       const float* h2, float scale, double* partials, float* loss, void* stream);
/* Slab variant (multi-GPU V-cycles, odil_amd/slab_solvers.py): the same pass over a ghost-extended local array, the
 * convergence measure restricted to the rank's own planes: *loss = sum over planes z0 <= z < z1 of (A u - rhs)^2 / denom
 * (z1 < 0: all planes; denom <= 0: the array size). */
int odil_poisson_residual_restrict_slab_f64(const double* u, const double* rhs, double* coarse, const int64_t* shape,
                                            int ndim, const double* h2, double scale, int64_t z0, int64_t z1,
                                            double denom, double* partials, double* loss, void* stream);
int odil_poisson_residual_restrict_slab_f32(const float* u, const float* rhs, float* coarse, const int64_t* shape,
                                            int ndim, const float* h2, float scale, int64_t z0, int64_t z1, double denom,
                                            double* partials, float* loss, void* stream);
/* Stencil adjoint and the first transposed prolongation in one pass (3-D, even extents >= 4):
 * g0 = scale * A^T fu (stored only when g0 != NULL), g1 = P^T g0 on the grid of half the extents, and the Adam
 * update of the finest level (x0, m0, v0; all NULL: no update, g0 required) and of the next level (x1, m1, v1;
 * optional) by the lanes that form their gradients.  g0 never makes the round trip through memory that
 * odil_poisson_adjoint_adam + odil_mg_synth_adj_adam need (reference core.py:1100 / optimizer.py:311-319);
 * results are bit-identical to that pair.  cut_lo / cut_hi: that end of axis 0 is a slab interface with ghost
 * planes beyond it (multi-GPU), not a wall, as in odil_interp_adj_cut. */
int odil_poisson_adjoint_transpose_adam_f64(const double* fu, double* g0, double* g1, const int64_t* fshape,
                                            const double* h2, double scale, double* x0, double* m0, double* v0,
                                            double* x1, double* m1, double* v1, double alpha, double one_minus_b1,
                                            double one_minus_b2, double eps, const double* alpha_dev, int cut_lo,
                                            int cut_hi, void* stream);
int odil_poisson_adjoint_transpose_adam_f32(const float* fu, float* g0, float* g1, const int64_t* fshape,
                                            const float* h2, float scale, float* x0, float* m0, float* v0, float* x1,
                                            float* m1, float* v1, float alpha, float one_minus_b1, float one_minus_b2,
                                            float eps, const float* alpha_dev, int cut_lo, int cut_hi, void* stream);
/* One damped-Jacobi sweep of that operator: uout = u - omega (A u - rhs) / diag(A), uout != u.  The
 * smoother of the geometric multigrid that solves the Newton system of the Poisson stencil
 * (reference linsolver.py:61-72 hands that system to pyamg).  u == NULL here and in the two-sweep form below: the sweeps
 * start from the ZERO vector (every coarse level of a V-cycle) -- u is neither read nor need it be zeroed by the caller;
 * the results are those of the same call on an array of zeros, bit for bit. */
int odil_poisson_jacobi_f64(const double* u, const double* rhs, double* uout, const int64_t* shape, int ndim,
                            const double* h2, double omega, void* stream);
int odil_poisson_jacobi_f32(const float* u, const float* rhs, float* uout, const int64_t* shape, int ndim,
                            const float* h2, float omega, void* stream);

/* TWO sweeps of odil_poisson_jacobi, with the weights omega1 and then omega2, in ONE pass over memory: the intermediate iterate
 * stays on the CU (registers along z, lane shifts along x, LDS along y), so a pair of sweeps moves the 3 words per cell
 * of one.  uout != u; bit-identical to two calls of odil_poisson_jacobi.  The last extent must be a multiple of the
 * 16-byte pack (2 doubles / 4 floats).  zc_hint: planes per workgroup chunk, <= 0: automatic.  (Multigrid smoother of
 * the Newton solve; the reference hands that system to SuperLU / pyamg, linsolver.py:17-26, 61-72.) */
int odil_poisson_jacobi2_f64(const double* u, const double* rhs, double* uout, const int64_t* shape, int ndim,
                             const double* h2, double omega1, double omega2, int zc_hint, void* stream);
int odil_poisson_jacobi2_f32(const float* u, const float* rhs, float* uout, const int64_t* shape, int ndim,
                             const float* h2, float omega1, float omega2, int zc_hint, void* stream);

/* The coarse-grid correction of a V-cycle and BOTH post-smoothing sweeps in one pass: two sweeps of odil_poisson_jacobi
 * (weights omega1, omega2) of u = x + P coarse, the prolongation formed on the fly (never stored), the first sweep kept
 * on the CU.  Bit-identical to odil_interp_add followed by odil_poisson_jacobi2.  3-D; coarse: shape cshape, x / rhs /
 * xout: 2 * cshape, xout != x; h2: squared FINE steps.  3 1/8 words per fine cell (separately: 2 1/8 + 6). */
int odil_poisson_jacobi2_synth_f64(const double* coarse, const double* x, const double* rhs, double* xout,
                                   const int64_t* cshape, const double* h2, double omega1, double omega2, int zc_hint,
                                   void* stream);
int odil_poisson_jacobi2_synth_f32(const float* coarse, const float* x, const float* rhs, float* xout,
                                   const int64_t* cshape, const float* h2, float omega1, float omega2, int zc_hint,
                                   void* stream);

/* The same residual with the LAST prolongation of the multigrid synthesis fused in: u = w0 + P coarse
 * (reference core.py:245-263, last step) is formed in registers and never stored.  `coarse`: the
 * synthesised level-1 array of shape cshape (3-D, all axes cell-centred), w0 / rhs / fu: the fine
 * arrays of shape 2 * cshape, h2: squared FINE steps.  fu is bit-identical to
 * odil_interp_add + odil_poisson_residual.  Loss = sum over the fine planes z0 <= z < z1 (z1 < 0:
 * all) of fu^2 / denom (denom <= 0: the array size), as odil_poisson_residual_slab. */
int odil_poisson_residual_synth_f64(const double* coarse, const double* w0, const double* rhs, double* fu,
                                    const int64_t* cshape, const double* h2, int64_t z0, int64_t z1, double denom,
                                    double* partials, double* loss, void* stream);
int odil_poisson_residual_synth_f32(const float* coarse, const float* w0, const float* rhs, float* fu,
                                    const int64_t* cshape, const float* h2, int64_t z0, int64_t z1, double denom,
                                    double* partials, float* loss, void* stream);

/* One damped-Jacobi sweep (odil_poisson_jacobi) of u = x + P coarse with the prolongation formed in registers:
 * the coarse-grid correction of a V-cycle and its first post-smoothing sweep in one pass, xout != x.  Equal, bit for
 * bit, to odil_interp_add followed by odil_poisson_jacobi.  Shapes as odil_poisson_residual_synth.  (The reference
 * hands the Newton system to pyamg, linsolver.py:61-72.) */
int odil_poisson_jacobi_synth_f64(const double* coarse, const double* x, const double* rhs, double* xout,
                                  const int64_t* cshape, const double* h2, double omega, void* stream);
int odil_poisson_jacobi_synth_f32(const float* coarse, const float* x, const float* rhs, float* xout,
                                  const int64_t* cshape, const float* h2, float omega, void* stream);

/* gu = J^T (scale * fu): cotangent of the operator above; scale = 2/size gives
 * d mean(fu^2)/du (core.py:1093-1101). */
int odil_poisson_adjoint_f64(const double* fu, double* gu, const int64_t* shape, int ndim, const double* h2,
                             double scale, void* stream);
int odil_poisson_adjoint_f32(const float* fu, float* gu, const int64_t* shape, int ndim, const float* h2,
                             float scale, void* stream);
/* The same adjoint with the Adam update of the array that gu is the gradient of fused in
 * (AdamNativeOptimizer._step, optimizer.py:316-318, applied by the lane that forms gu[i]):
 * gu is still written (the P^T chain reads it); x, m, v are updated in place. */
int odil_poisson_adjoint_adam_f64(const double* fu, double* gu, double* x, double* m, double* v, const int64_t* shape,
                                  int ndim, const double* h2, double scale, double alpha, double one_minus_b1,
                                  double one_minus_b2, double eps, const double* alpha_dev, void* stream);
int odil_poisson_adjoint_adam_f32(const float* fu, float* gu, float* x, float* m, float* v, const int64_t* shape,
                                  int ndim, const float* h2, float scale, float alpha, float one_minus_b1,
                                  float one_minus_b2, float eps, const float* alpha_dev, void* stream);
/* Per-shift Jacobian coefficient arrays d(sum fu)/d u_shift as
 * `Problem.eval_operator_grad` returns them under `distinct_shift`
 * (core.py:1313-1361): coeffs holds 2*ndim+1 arrays of `shape`, order
 * [centre, (-1 axis 0), (+1 axis 0), (-1 axis 1), ...]. */
int odil_poisson_jac_coeffs_f64(double* coeffs, const int64_t* shape, int ndim, const double* h2, void* stream);
int odil_poisson_jac_coeffs_f32(float* coeffs, const int64_t* shape, int ndim, const float* h2, void* stream);
/* Are 2 ndim + 1 coefficient arrays (a HOST array of device pointers, the order above) that Jacobian?  One pass over
 * them against the values odil_poisson_jac_coeffs would write, no reference arrays: out[2 k] = max |a_k - e_k|,
 * out[2 k + 1] = max |e_k| (device, 2 (2 ndim + 1) numbers; a NaN stays visible).  The recognition step of the general
 * Newton route (odil_amd/gmg.py: recognise_poisson; the reference has no counterpart: it factorises whatever it is given,
 * linsolver.py:17-26). */
int odil_poisson_jac_match_f64(const double* const* arrays, const int64_t* shape, int ndim, const double* h2,
                               double* partials, double* out, void* stream);
int odil_poisson_jac_match_f32(const float* const* arrays, const int64_t* shape, int ndim, const float* h2, double* partials,
                               float* out, void* stream);

/* ---- optimizers (reference optimizer.py:256-341) ------------------------------ */
/* AdamNativeOptimizer._step (optimizer.py:311-319) on a flat vector:
 *   m += (g-m)*one_minus_b1; v += (g^2-v)*one_minus_b2; x -= (m*alpha)/(sqrt(v)+eps).
 * In this and every *_adam entry point `alpha_dev`, when non-NULL, is a device scalar that replaces
 * `alpha`: the bias-corrected step size changes every epoch, and an epoch captured into a hipGraph
 * must read it from memory (odil_amd/optimizer.py: graph replay of launch-bound epochs). */
int odil_adam_step_f64(double* x, double* m, double* v, const double* g, int64_t n, double alpha,
                       double one_minus_b1, double one_minus_b2, double eps, const double* alpha_dev, void* stream);
int odil_adam_step_f32(float* x, float* m, float* v, const float* g, int64_t n, float alpha, float one_minus_b1,
                       float one_minus_b2, float eps, const float* alpha_dev, void* stream);
/* The same update on `npieces` (<= 65535) contiguous pieces of `count` elements, piece o at o * stride + offset: the
 * planes next to the slab interfaces of a ghost-extended array whose sharded axis is not the leading one (the rest of
 * such an array is updated by the launch that forms its gradient; these planes wait for the neighbour's share). */
int odil_adam_step_pieces_f64(double* x, double* m, double* v, const double* g, int64_t npieces, int64_t stride,
                              int64_t offset, int64_t count, double alpha, double one_minus_b1, double one_minus_b2,
                              double eps, const double* alpha_dev, void* stream);
int odil_adam_step_pieces_f32(float* x, float* m, float* v, const float* g, int64_t npieces, int64_t stride,
                              int64_t offset, int64_t count, float alpha, float one_minus_b1, float one_minus_b2,
                              float eps, const float* alpha_dev, void* stream);
/* Planes of ghost-extended arrays <-> contiguous message buffers of the slab exchanges (odil_amd/slab.py,
 * slab_traced.py; no reference counterpart).  `descs`: DEVICE array of ndesc (<= 65535) records
 * of five int64 {base, outer, ostride, inner, boff}: plane k of the message is `outer` runs of `inner` contiguous
 * elements of `arr`, run o at base + o * ostride, and the contiguous range [boff, boff + outer * inner) of `buf`.
 * mode 0: buf <- arr (pack); 1: arr <- buf (unpack); 2: arr += buf (halo-add: the transpose of the pack).
 * max_count: the largest outer * inner; vec_ok != 0: every inner is a multiple of 4 (16-byte packs).  One launch. */
int odil_planes_copy_f64(double* arr, double* buf, const int64_t* descs, int ndesc, int64_t max_count, int vec_ok,
                         int mode, void* stream);
int odil_planes_copy_f32(float* arr, float* buf, const int64_t* descs, int ndesc, int64_t max_count, int vec_ok,
                         int mode, void* stream);
/* y += a * x  (GdOptimizer: x -= lr*g, optimizer.py:270; Newton update util.py:177). */
int odil_axpy_f64(double* y, const double* x, int64_t n, double a, void* stream);
int odil_axpy_f32(float* y, const float* x, int64_t n, float a, void* stream);
/* y = a * (adev ? adev[0] : 1) * x: the cotangent of the loss terms, d mean(f^2)/df =
 * (2/n) * gout * f (core.py:1093-1101); `adev` is an optional DEVICE scalar. */
int odil_scale_f64(const double* x, double* y, int64_t n, double a, const double* adev, void* stream);
int odil_scale_f32(const float* x, float* y, int64_t n, float a, const float* adev, void* stream);
/* y = a (.) b (accumulate == 0) or y += a (.) b: one CSR block of the Jacobian applied as
 * coefficient array times gathered field (core.py:1144-1171). */
int odil_addcmul_f64(double* y, const double* a, const double* b, int64_t n, int accumulate, void* stream);
int odil_addcmul_f32(float* y, const float* a, const float* b, int64_t n, int accumulate, void* stream);
/* out[k] = sum_i a[k*lda + i] * b[i], k < nvec, deterministic, f64 accumulation.
 * `partials`: odil_dots_workspace_bytes(nvec) scratch. */
size_t odil_dots_workspace_bytes(int nvec);
int odil_dots_f64(const double* a, int64_t lda, int nvec, const double* b, int64_t n, double* partials,
                  double* out, void* stream);
int odil_dots_f32(const float* a, int64_t lda, int nvec, const float* b, int64_t n, double* partials, float* out,
                  void* stream);
/* out[j*nvec + k] = <a_k, b_j>, j < 3 (b1, b2 may be NULL -> zeros), in one pass over the a_k.
 * `partials`: odil_dots_workspace_bytes(3 * nvec) scratch; `out`: 3 * nvec values. */
int odil_dots3_f64(const double* a, int64_t lda, int nvec, const double* b0, const double* b1, const double* b2,
                   int64_t n, double* partials, double* out, void* stream);
int odil_dots3_f32(const float* a, int64_t lda, int nvec, const float* b0, const float* b1, const float* b2, int64_t n,
                   double* partials, float* out, void* stream);
/* out[0] = <g, d>, out[1] = <g, g>, out[2] = max_i |g[i]| in one pass: what the L-BFGS-B line search and
 * stopping test read after an evaluation (reference optimizer.py:95-105 -> SciPy lnsrlb / projgr).
 * `partials`: odil_dots_workspace_bytes(3) scratch. */
int odil_lbfgs_probe_f64(const double* g, const double* d, int64_t n, double* partials, double* out, void* stream);
int odil_lbfgs_probe_f32(const float* g, const float* d, int64_t n, double* partials, float* out, void* stream);
/* y = beta * y + sum_k coef[k] * a[k*lda + :]  (coef on DEVICE, length nvec). */
int odil_lincomb_f64(double* y, double beta, const double* a, int64_t lda, int nvec, const double* coef, int64_t n,
                     void* stream);
int odil_lincomb_f32(float* y, float beta, const float* a, int64_t lda, int nvec, const float* coef, int64_t n,
                     void* stream);

/* ---- Newton: matrix-free normal equations (reference core.py:1113-1217,
 *      linsolver.py:17-26) ------------------------------------------------------- */
/* y = M x  where row r of M has coefficients coeffs[s][r] on columns roll(-shift_s):
 * the CSR matrix `field_to_matrix` builds (core.py:1144-1171), kept as coefficient
 * arrays.  `shifts`: nshift*ndim int64. transpose != 0 applies M^T. */
int odil_stencil_apply_f64(const double* coeffs, const int64_t* shifts, int nshift, const double* x, double* y,
                           const int64_t* shape, int ndim, int transpose, void* stream);
int odil_stencil_apply_f32(const float* coeffs, const int64_t* shifts, int nshift, const float* x, float* y,
                           const int64_t* shape, int ndim, int transpose, void* stream);
/* x = M^-1 b for such a matrix when it is triangular along `axis` with the coefficient array `diag` (zero shift)
 * alone on the diagonal block -- the Jacobian of an operator that is explicit in time (wave): every other shift has
 * a negative (direction > 0, forward substitution) or positive (direction < 0) component along `axis` and no
 * coefficient that wraps around the ends.  M d = -r then has the solution of the normal equations
 * M^T M d = -M^T r the reference hands to SuperLU (linsolver.py:17-26).  One launch per level of `axis`. */
int odil_stencil_march_f64(const double* coeffs, const int64_t* shifts, int nshift, int diag, const double* b, double* x,
                           const int64_t* shape, int ndim, int axis, int direction, void* stream);
int odil_stencil_march_f32(const float* coeffs, const int64_t* shifts, int nshift, int diag, const float* b, float* x,
                           const int64_t* shape, int ndim, int axis, int direction, void* stream);
/* Geometric multigrid for a (2 ndim + 1)-point operator with VARIABLE coefficients on a cell-centred grid, ndim <= 3 (the
 * Newton system of any single-field operator whose Jacobian `Problem.linearize` delivers as per-shift coefficient arrays,
 * core.py:1113-1217; the reference hands M^T M to SuperLU / pyamg, linsolver.py:17-26, 61-72).  `coeffs`: the 2 ndim + 1
 * arrays one after another in the order (0, -e_0, +e_0, -e_1, +e_1, ...), each of `shape`; neighbours wrap periodically
 * (core.py:962-963), a wall row carries a zero coefficient towards the wall.
 *   smooth, mode 0:  out = x - omega (A x - b) / c0   (one damped-Jacobi sweep; out != x).  x == NULL (here and in
 *           odil_stencil_var_smooth2): the sweeps start from the ZERO vector, x is not read -- for finite coefficients the
 *           bits of the same call on an array of zeros
 *           mode 1:  out = b - A x
 *   residual_restrict: coarse = scale * sum over the 2^ndim children of (b - A x), loss = mean((A x - b)^2)
 *           (deterministic two-stage sum; `partials`: odil_reduce_workspace_bytes(); loss == NULL: no reduction launch);
 *           even extents
 *   coarsen: the coarse-grid operator as 2 ndim + 1 arrays of shape / 2: aggregates of 2^ndim cells, piecewise-constant
 *           Galerkin products of the second-order part (x 1/2), the matrix-antisymmetric part and the row sums
 *           (csrc/stencil_mg.hip). */
int odil_stencil_var_smooth_f64(const double* coeffs, const double* x, const double* b, double* out, const int64_t* shape,
                                int ndim, double omega, int mode, void* stream);
int odil_stencil_var_smooth_f32(const float* coeffs, const float* x, const float* b, float* out, const int64_t* shape,
                                int ndim, float omega, int mode, void* stream);
int odil_stencil_var_residual_restrict_f64(const double* coeffs, const double* x, const double* b, double* coarse,
                                           const int64_t* shape, int ndim, double scale, double* partials, double* loss,
                                           void* stream);
int odil_stencil_var_residual_restrict_f32(const float* coeffs, const float* x, const float* b, float* coarse,
                                           const int64_t* shape, int ndim, float scale, double* partials, float* loss,
                                           void* stream);
/* slab variant: the norm over the planes z0 <= z < z1 only, divided by denom (conventions of the Poisson one above) */
int odil_stencil_var_residual_restrict_slab_f64(const double* coeffs, const double* x, const double* b, double* coarse,
                                                const int64_t* shape, int ndim, double scale, int64_t z0, int64_t z1,
                                                double denom, double* partials, double* loss, void* stream);
int odil_stencil_var_residual_restrict_slab_f32(const float* coeffs, const float* x, const float* b, float* coarse,
                                                const int64_t* shape, int ndim, float scale, int64_t z0, int64_t z1,
                                                double denom, double* partials, float* loss, void* stream);
int odil_stencil_var_coarsen_f64(const double* coeffs, double* coarse, const int64_t* shape, int ndim, void* stream);
int odil_stencil_var_coarsen_f32(const float* coeffs, float* coarse, const int64_t* shape, int ndim, void* stream);
/* ... merging two cells along the axes with halve[i] != 0 only (semi-coarsening: the strongly coupled axes of an anisotropic
 * operator; coarse shape = shape / 2 on those axes, shape on the others).  The factor 1/2 of the second-order part goes with
 * the merged axes; with every axis merged this is odil_stencil_var_coarsen. */
int odil_stencil_var_coarsen_axes_f64(const double* coeffs, double* coarse, const int64_t* shape, int ndim, const int* halve,
                                      void* stream);
int odil_stencil_var_coarsen_axes_f32(const float* coeffs, float* coarse, const int64_t* shape, int ndim, const int* halve,
                                      void* stream);

/* WHOLE EPOCHS of the multigrid Poisson problem (1-D or 2-D, all axes cell-centred) in ONE launch: one workgroup walks
 * synthesis, residual + loss, adjoint, transposed prolongations and the Adam update of every level -- the 17 dependent
 * launches of a 1-D N = 256 epoch -- `nepochs` times, the state resident in LDS when it fits.  Every phase repeats the
 * arithmetic of the kernel it replaces and the loss is summed in the order of the two-stage reduction: the trajectory is
 * bit-identical to the multi-launch path.
 *   x, m, v, g   packed vectors of all levels, finest first (unknowns, Adam moments, gradient: all updated)
 *   u            scratch of the same length (the synthesised field of every level; level 0 = the field u)
 *   fu, rhs      residual (out) and right-hand side on the finest level
 *   shapes       nlvl x ndim extents, each level half the one before; h2: squared steps of the finest level
 *   alphas       DEVICE array of nepochs step sizes lr sqrt(1 - b2^t) / (1 - b1^t); losses: DEVICE array, the loss each
 *                epoch evaluated (before its update), norms: their square roots; partials: the reduction workspace
 * Reference: core.py:245-263, 606-700; examples/poisson/poisson.py:57-113; core.py:1093-1100; optimizer.py:311-336. */
/* 1 when odil_poisson_small_epochs keeps the state of these levels in LDS for the whole launch (elem_size 4 / 8 bytes):
 * x, m, v, the gradient (= the synthesis scratch) of all levels, residual and right-hand side of the finest -- 156 KB. */
int odil_poisson_small_epochs_resident(const int64_t* shapes, int nlvl, int ndim, int elem_size);
int odil_poisson_small_epochs_f64(double* x, double* m, double* v, double* g, double* u, double* fu, const double* rhs,
                                  const int64_t* shapes, int nlvl, int ndim, const double* h2, const double* alphas,
                                  int nepochs, double one_minus_b1, double one_minus_b2, double eps, double* losses,
                                  double* norms, double* partials, void* stream);
int odil_poisson_small_epochs_f32(float* x, float* m, float* v, float* g, float* u, float* fu, const float* rhs,
                                  const int64_t* shapes, int nlvl, int ndim, const float* h2, const float* alphas,
                                  int nepochs, float one_minus_b1, float one_minus_b2, float eps, float* losses,
                                  float* norms, double* partials, void* stream);

/* TWO sweeps of odil_stencil_var_smooth in mode 0, with the weights omega1 and then omega2, in ONE pass: the coefficient arrays -- 7 of
 * the 10 words a sweep moves in 3-D -- are read once for both, the intermediate iterate stays on the CU.  out != x;
 * bit-identical to two calls of odil_stencil_var_smooth.  The last extent must be even.  zc_hint: planes per workgroup
 * chunk, <= 0: automatic. */
int odil_stencil_var_smooth2_f64(const double* coeffs, const double* x, const double* b, double* out, const int64_t* shape,
                                 int ndim, double omega1, double omega2, int zc_hint, void* stream);
int odil_stencil_var_smooth2_f32(const float* coeffs, const float* x, const float* b, float* out, const int64_t* shape,
                                 int ndim, float omega1, float omega2, int zc_hint, void* stream);

/* The COARSE TAIL of a multigrid V-cycle in one launch: one workgroup walks `nlev` <= 8 small levels (pre-smoothing,
 * residual + restriction, dense solve on the coarsest, prolongation + post-smoothing) with a workgroup barrier where a
 * launch boundary used to be (levels of a few thousand cells cost ~7 dependent launches per level and cycle otherwise).
 *   coeffs   the (2 ndim + 1) coefficient arrays of every level, finest first, back to back (layout of
 *            odil_stencil_var_smooth; the Poisson hierarchy passes odil_poisson_jac_coeffs of its levels)
 *   shapes   nlev x ndim extents; halve: (nlev - 1) x ndim, 1 where the transition to the next level merges cell pairs
 *   xin      start iterate on the finest of these levels (NULL: zero), b its right-hand side, xout the result (!= xin, b)
 *   work     >= 3 * (cells of all levels) values of scratch;  inv: ninv x ninv (pseudo-)inverse of the coarsest operator
 *   wpre / wpost   HOST arrays of npre / npost (<= 4) Jacobi weights;  sweep: x' = x - w (A x - b) / c0
 *   fmg      0: one V-cycle from xin;  1: the nested-iteration start (b restricted to every level, every level one cycle
 *            from the interpolated solution of the level below), xin ignored
 * Transfers: mean of the merged children down, the multigrid decomposition's prolongation (reference core.py:606-700,
 * joint ghost rule core.py:640-643) up.  (Newton solve; the reference: SuperLU / pyamg, linsolver.py:17-26, 61-72.) */
int odil_stencil_vcycle_tail_f64(const double* coeffs, const int64_t* shapes, const int* halve, int nlev, int ndim,
                                 const double* xin, const double* b, double* xout, double* work, int64_t work_len,
                                 const double* inv, int ninv, const double* wpre, int npre, const double* wpost,
                                 int npost, int fmg, void* stream);
int odil_stencil_vcycle_tail_f32(const float* coeffs, const int64_t* shapes, const int* halve, int nlev, int ndim,
                                 const float* xin, const float* b, float* xout, float* work, int64_t work_len,
                                 const float* inv, int ninv, const float* wpre, int npre, const float* wpost, int npost,
                                 int fmg, void* stream);
/* Mixed-precision iterative refinement of the multigrid Newton solve (gmg.py; no reference counterpart -- the reference's
 * direct solver is double throughout): float32 V-cycles inside a float64 residual loop.  narrow_scale: y32 = s x64 with
 * s = a / sqrt(*msq) (msq: device scalar, the mean square the residual kernel just wrote; NULL: s = a); widen_axpy:
 * y64 += s x32 with s = a sqrt(*msq). */
int odil_narrow_scale(const double* x, float* y, int64_t n, double a, const double* msq, void* stream);
int odil_widen_axpy(double* y, const float* x, int64_t n, double a, const double* msq, void* stream);
/* out[r] = max |a[r n .. (r + 1) n)| for the nrows <= 64 rows of one array, two launches in all (the multigrid set-up reads
 * the largest coupling of every direction on every level: the 2 d coefficient arrays of a level are rows of one buffer). */
int odil_max_abs_rows_f64(const double* a, int nrows, int64_t n, double* partials, double* out, void* stream);
int odil_max_abs_rows_f32(const float* a, int nrows, int64_t n, double* partials, float* out, void* stream);
/* out[0] = max |a - b|, out[1] = max |b| over n entries in one pass (NaN differences propagate): the comparison by which
 * a linearised operator's coefficient arrays are recognised as a known stencil (gmg.recognise_poisson) -- no reference
 * counterpart.  `partials`: odil_reduce_workspace_bytes(). */
int odil_max_abs_diff_f64(const double* a, const double* b, int64_t n, double* partials, double* out, void* stream);
int odil_max_abs_diff_f32(const float* a, const float* b, int64_t n, double* partials, float* out, void* stream);
/* CSR assembly of the same matrix (core.py:1144-1171, :1214): indptr[n+1], indices,
 * data of nnz = nshift*n entries, columns offset by `col_offset`; rows keep ODIL's
 * order (ascending shift index within a row, not sorted by column). */
int odil_csr_assemble_f64(const double* coeffs, const int64_t* shifts, int nshift, const int64_t* shape, int ndim,
                          int64_t col_offset, int64_t* indptr, int64_t* indices, double* data, void* stream);
int odil_csr_assemble_f32(const float* coeffs, const int64_t* shifts, int nshift, const int64_t* shape, int ndim,
                          int64_t col_offset, int64_t* indptr, int64_t* indices, float* data, void* stream);

/* ---- Newton: the dense block of the normal equations on the matrix cores --------------------------------
 * `Array` / `NeuralNet` unknowns enter `Problem.linearize` as DENSE Jacobian columns (reference core.py:1189-1203);
 * the normal equations (linsolver.py:17-23) need D^T D, D^T r and, for the Schur complement against the
 * matrix-free stencil part, (S^T D)^T Z.  out (px x py, row-major) = X^T Y for row-major X (n x px, row stride
 * ldx) and Y (n x py, row stride ldy), px, py <= 64, by v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 tiles
 * with a fixed-order two-stage reduction (bit-reproducible).  `workspace`: odil_dense_block_workspace_bytes(). */
size_t odil_dense_block_workspace_bytes(void);
int odil_dense_block_xty_f64(const double* x, const double* y, int64_t n, int px, int py, int64_t ldx, int64_t ldy,
                             double* out, double* workspace, void* stream);
int odil_dense_block_xty_f32(const float* x, const float* y, int64_t n, int px, int py, int64_t ldx, int64_t ldy,
                             float* out, float* workspace, void* stream);
/* The Gram matrix D^T D (p x p) of one block: odil_dense_block_xty with X = Y = D. */
int odil_dense_block_gram_f64(const double* d, int64_t n, int p, int64_t ld, double* out, double* workspace,
                              void* stream);
int odil_dense_block_gram_f32(const float* d, int64_t n, int p, int64_t ld, float* out, float* workspace,
                              void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ODIL_HIP_H */
