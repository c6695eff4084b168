/*
 * ica_hip.h -- C ABI of libica_hip.so, the MI355X (gfx950) inverse-compositional alignment engine.
 *
 * The reference (mfournigault/inverse_compositional_algorithm) has no FFI: its boundary for this
 * path is three Python functions in src/inverse_compositional_algorithm.py
 *     inverse_compositional_algorithm()            :17-133
 *     robust_inverse_compositional_algorithm()     :135-261
 *     pyramidal_inverse_compositional_algorithm()  :264-374
 * plus the operators they call.  A maintainer who wanted to bind a native implementation of that
 * path would bind exactly the entry points below (ctypes stub: INTEGRATION.md).  Every entry point
 * cites the reference code it replaces.  All file:line citations are relative to the reference root.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types appear in any signature (a stream is a void*).
 *   - images are planar: [batch][channels][height][width], contiguous; float32, or uint8 when
 *     ica_problem.storage == ICA_STORAGE_U8 (the reference's data is 8-bit and it casts on entry, ref :63-65, :314-316:
 *     such images stay one byte per pixel in HBM; the pyramid levels built from them are float32 either way, and the
 *     two storages give bit-identical results on the same pixel values).
 *     (The reference holds (H,W,3) float64 arrays; the Python shim transposes and converts.)
 *   - transform parameters are float64, 8 per pair, zero padded: p[b*8 + k], k < nparams.
 *   - "_dev" functions take DEVICE pointers and enqueue on the context's stream without
 *     synchronising (except where noted); the others take HOST pointers, copy, run and synchronise.
 *     ica_solve_batch_dev is the exception that matters: it never calls a device-wide synchronise, but it is a
 *     HOST-BLOCKING call.  The number of pairs still iterating sizes every next launch, and the host reads it from a
 *     pinned board the kernels post to, staying exactly one iteration ahead of the GPU: the calling thread polls
 *     (yielding the core; sleeping between polls after 4 ms) for the whole solve and returns about one iteration
 *     before the GPU finishes -- the outputs are still only valid after a sync.  Budget one host thread per ctx for
 *     the duration of a solve (8 on an 8-GPU node); ICA_BOARD_TIMEOUT_MS (default 10000) bounds each wait.
 *   - layout: every image / output argument is dense [batch][channels][height][width] with no padding (row stride =
 *     width elements, plane stride = height * width): ica_problem carries no stride fields.  Strided or interleaved
 *     (H, W, C) data is repacked by the caller (the Python layer does).
 *   - every function returns ICA_OK (0) or a negative ica_status.
 *   - one ica_ctx per GPU; a ctx is not thread-safe, different ctxs are independent.
 */
#ifndef ICA_HIP_H
#define ICA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICA_ABI_VERSION 4   /* 3: ica_problem.storage, void* image pointers, ica_op_ref_*; 4: ica_stats.shader_clock_ghz */
#define ICA_MAX_PARAMS 8
#define ICA_MAX_SCALES 16
#define ICA_MAX_CHANNELS 4
#define ICA_TRACE_STRIDE 10 /* per iteration: |dp|, p[0..7], lambda */

typedef enum ica_status {
  ICA_OK = 0,
  ICA_EINVAL_SHAPE = -1,    /* -> ValueError("I1 and I2 must be RGB images ...")   ref :48-49, :300-301 */
  ICA_EINVAL_DIMS = -2,     /* -> ValueError("I1 and I2 must have the same dimensions") ref :55-56      */
  ICA_EINVAL_TOL = -3,      /* -> ValueError("TOL must be positive and very small ...")  ref :59-60      */
  ICA_EINVAL_TRANSFORM = -4,/* unknown transform type          transformation.py:31-32                   */
  ICA_EINVAL_ROBUST = -5,   /* unknown robust error function   image_optimisation.py:50-51               */
  ICA_EINVAL_ARG = -6,      /* null pointer / non-positive size / too many scales                       */
  ICA_ENOMEM = -7,          /* device allocation failed                                                 */
  ICA_EHIP = -8,            /* a HIP runtime call failed (see ica_last_error)                           */
  ICA_ENODEVICE = -9        /* no gfx950 device / bad device index                                      */
} ica_status;

/* transformation.py:8-13 */
typedef enum ica_transform {
  ICA_TRANSLATION = 1, ICA_EUCLIDEAN = 2, ICA_SIMILARITY = 3, ICA_AFFINITY = 4, ICA_HOMOGRAPHY = 5
} ica_transform;

/* image_optimisation.py:10-15 */
typedef enum ica_robust {
  ICA_QUADRATIC = 0, ICA_TRUNCATED_QUADRATIC = 1, ICA_GERMAN_MCCLURE = 2, ICA_LORENTZIAN = 3,
  ICA_CHARBONNIER = 4
} ica_robust;

/* element type of the level-0 images handed to ica_solve_batch*_ / ica_op_normal_equations_dev */
typedef enum ica_storage { ICA_STORAGE_F32 = 0, ICA_STORAGE_U8 = 1 } ica_storage;

enum {
  /* compositional update: use the exact matrix composition instead of the reference's closed forms,
   * which differ in AFFINITY p[1] (transformation.py:106) and HOMOGRAPHY p[4] (:136).  Default (0)
   * reproduces the reference. */
  ICA_FLAG_EXACT_UPDATE = 1,
};

typedef struct ica_problem {
  int32_t batch;          /* B independent image pairs                                                  */
  int32_t height, width;  /* level-0 size                                                               */
  int32_t channels;       /* 1..ICA_MAX_CHANNELS (L2 / pyramidal reference entries accept 3 only, ref :48;   */
                          /* its robust entry any count, ref :168-177); 1 and 3 run the LDS-staged kernel    */
  int32_t transform_type; /* ica_transform                                                              */
  int32_t robust_type;    /* ica_robust; QUADRATIC selects the L2 loop (ref :343-355)                   */
  int32_t nscales;        /* pyramid levels, >= 1 (config.ini pyramid_levels)                           */
  int32_t max_iter;       /* constants.py:1 MAX_ITER = 30                                               */
  int32_t nanifoutside;   /* ref :85                                                                    */
  int32_t delta;          /* ref :85-93                                                                 */
  int32_t flags;          /* ICA_FLAG_*                                                                 */
  int32_t storage;        /* ica_storage of I1 / I2 (uint8: 1 or 3 channels); outputs are float32 regardless    */
  double tol;             /* TOL, must be < 0.01 (ref :59)                                              */
  double lambda;          /* robust threshold; <= 0 selects the schedule below (ref :223)              */
  double nu;              /* pyramid downsampling factor (config.ini nu)                                */
  double lambda0;         /* constants.py:2 LAMBDA_0 = 80                                               */
  double lambdaN;         /* constants.py:3 LAMBDA_N = 5                                                */
  double lambda_ratio;    /* constants.py:4 LAMBDA_RATIO = 0.9                                          */
} ica_problem;

/* Filled by ica_solve_batch*_ when stats != NULL (the call then synchronises).
 * The iteration work of a scale is split by the kernel that did it: `iter_*` is the steady-state kernel
 * (L2: ica_iterate_kernel<TT, b-only, C, ..>, iterations >= 1; robust: <TT, weighted H+b, C, rho, ..>, every iteration), `first_*`
 * is the L2 path's first iteration of a level, which also accumulates the Hessian (<TT, unweighted H+b, C>; zeros for
 * robust). */
typedef struct ica_stats {
  int64_t pair_iterations[ICA_MAX_SCALES]; /* executed (pair, iteration) units per scale (0 = finest) */
  int64_t pair_iterations_total;
  int64_t iter_launches[ICA_MAX_SCALES];   /* launches of the steady-state iteration kernel per scale   */
  double iter_kernel_ms[ICA_MAX_SCALES];   /* HIP-event time of those launches (profiling on), else 0   */
  double iter_kernel_bytes[ICA_MAX_SCALES];/* algorithmic bytes they processed: 2*H_l*W_l*C*4 per unit  */
                                           /* (the 4-byte convention of SURVEY 8(d), also for uint8 storage) */
  int32_t level_width[ICA_MAX_SCALES], level_height[ICA_MAX_SCALES];
  int64_t first_launches[ICA_MAX_SCALES];  /* same three figures for the L2 first-iteration launches    */
  double first_kernel_ms[ICA_MAX_SCALES];
  double first_kernel_bytes[ICA_MAX_SCALES];
  double shader_clock_ghz;                 /* clock the iteration kernel ran at (profiling on): cycles / 100 MHz ticks */
                                           /* of one workgroup's lifetime in the middle of every launch; else 0        */
} ica_stats;

typedef struct ica_ctx ica_ctx;

/* ---- library / context ------------------------------------------------------------------------ */
int ica_version(void);                 /* ICA_ABI_VERSION                                            */
int ica_device_count(void);            /* number of visible HIP devices (>= 0)                       */
const char* ica_strerror(int status);  /* static string                                              */
const char* ica_last_error(const ica_ctx* ctx); /* last HIP error text of this ctx ("" if none)      */
int ica_create(int device, ica_ctx** out);
int ica_destroy(ica_ctx* ctx);
int ica_set_stream(ica_ctx* ctx, void* hip_stream); /* NULL = default stream                        */
int ica_synchronize(ica_ctx* ctx);
int ica_set_profiling(ica_ctx* ctx, int on);       /* HIP events around every iteration-kernel launch */
/* Measurement aid (no reference counterpart): ONE wave on this ctx's stream compares the shader clock counter
 * (s_memtime) with the constant 100 MHz counter (s_memrealtime) for `duration_us` microseconds and returns the
 * shader clock it saw, in GHz.  Run on a second ctx / stream while another ctx computes, it reports the clock the
 * chip sustains UNDER THAT LOAD (the driver's sysfs reading does not track it).  Blocks until the wave is done.
 * Caution: two HIP streams may share a hardware queue, in which case the probe runs BETWEEN the other context's
 * kernels, delays them and reads the idle clock; ica_stats.shader_clock_ghz (measured inside the iteration kernel,
 * profiling on) has neither problem and is what bench.py reports. */
int ica_op_shader_clock_dev(ica_ctx* ctx, int duration_us, double* ghz_out);
int ica_nparams(int transform_type);   /* transformation.py:16-32; < 0 if unknown                    */
/* zoom.py:8-22: level sizes, int(np.round(n*nu)) (round-half-even). */
int ica_zoom_size(int width, int height, double nu, int* out_width, int* out_height);
/* Pre-allocate every workspace a later ica_solve_batch*_(prob) needs (so the solve itself never
 * allocates).  Called implicitly by the solve entry points when needed. */
int ica_prepare(ica_ctx* ctx, const ica_problem* prob);

/* ---- the hot path ------------------------------------------------------------------------------
 * Replaces pyramidal_inverse_compositional_algorithm() (ref :264-374) for a batch of pairs; with
 * nscales == 1 it is inverse_compositional_algorithm() (:17-133, robust_type == QUADRATIC) or
 * robust_inverse_compositional_algorithm() (:135-261).
 *   I1, I2     [B][C][H][W] float32, or uint8 when prob->storage == ICA_STORAGE_U8
 *   p_inout    [B][8] float64: initial parameters in, solution out.  As in the reference, the initial
 *              value is only used when nscales == 1 (ref :327,:337,:372: coarser levels start at 0
 *              and ps[0] is overwritten).
 *   error_out  [B] float64: |dp| of the last iteration at scale 0 (ref :374)
 *   niter_out  [B][nscales] int32: iterations executed per scale (index 0 = finest)
 *   DI_out, Iw_out  nullable [B][C][H][W] float32 with NaN where the reference has NaN: the error
 *              image and warped image of the LAST iteration at scale 0 (ref :133,:261,:374)
 *   trace_out  nullable [B][nscales][max_iter][ICA_TRACE_STRIDE] float64, what verbose=True prints
 *              (ref :125-129, :253-257): |dp|, p after the update, lambda after its decay
 *   stats      nullable HOST pointer
 */
int ica_solve_batch_dev(ica_ctx* ctx, const ica_problem* prob, const void* I1, const void* I2,
                        double* p_inout, double* error_out, int32_t* niter_out, float* DI_out,
                        float* Iw_out, double* trace_out, ica_stats* stats);
int ica_solve_batch(ica_ctx* ctx, const ica_problem* prob, const void* I1, const void* I2,
                    double* p_inout, double* error_out, int32_t* niter_out, float* DI_out,
                    float* Iw_out, double* trace_out, ica_stats* stats);

/* ---- per-operator hooks (device pointers) ------------------------------------------------------
 * One-to-one with the reference operators; used by the parity tests and by data generation. */

/* bicubic_interpolation.py:154-206 (skimage.transform.warp order=3, constant, cval=nan, clip):
 * Iw = clip(bicubic I2(M(p).x)), NaN where any of the 16 taps is outside.  If I1 != NULL also
 * DI = Iw - I1 (ref :114).  p is [B][8].  minmax (nullable, [B][2] float32) supplies the clip range;
 * NULL computes it from I2 first (min/max over the whole [C][H][W] image, as skimage does). */
int ica_op_warp_dev(ica_ctx* ctx, int batch, int channels, int height, int width, int transform_type,
                    const float* I2, const double* p, const float* minmax, const float* I1,
                    float* Iw_out, float* DI_out);

/* min / max of each [C][H][W] image -> out[b][0..1] (the clip range of skimage's warp / rescale). */
int ica_op_minmax_dev(ica_ctx* ctx, int batch, int channels, int height, int width, const float* I,
                      float* out);

/* One iteration's normal equations for the current p (no solve, no update):
 *   b = sum rho' * DIJ^T DI   image_optimisation.py:82-143
 *   H = sum rho' * DIJ^T DIJ  derivatives.py:73-107
 * with the gradient / NaN frame of ref :81-93, the Jacobian of derivatives.py:7-70, the steepest
 * descent images of image_optimisation.py:158-194 and rho' of image_optimisation.py:17-79 all
 * recomputed on the fly.  lambda_it is [B] (ignored for QUADRATIC).  H_out [B][8][8], b_out [B][8].
 * I1 / I2 are float32 or uint8 as prob->storage says. */
int ica_op_normal_equations_dev(ica_ctx* ctx, const ica_problem* prob, const void* I1,
                                const void* I2, const double* p, const double* lambda_it,
                                double* H_out, double* b_out);

/* derivatives.py:110-130 + image_optimisation.py:146-155 + transformation.py:35-141:
 * dp = inv(H) b (zeros if singular), error = |dp|, p <- p o dp^-1 in place.
 * H [B][8][8], b [B][8], p_inout [B][8], dp_out [B][8] nullable, error_out [B] nullable. */
int ica_op_solve_update_dev(ica_ctx* ctx, int batch, int transform_type, int flags, const double* H,
                            const double* b, double* p_inout, double* dp_out, double* error_out);

/* One pyramid step, skimage.transform.rescale(I, nu, mode='constant', cval=0, order=3,
 * anti_aliasing=True, preserve_range=True) as called at ref :333-336: Gaussian sigma=(f-1)/2 with zero
 * padding, cubic B-spline zoom (grid_mode), clip to the input range.  out is
 * [B][C][round(H*nu)][round(W*nu)] (ica_zoom_size). */
int ica_op_pyramid_down_dev(ica_ctx* ctx, int batch, int channels, int height, int width, double nu,
                            const float* in, float* out);

/* zoom.py:62-125: p at the coarse size (w,h) -> p at the fine size (ww,hh), [B][8] in place. */
int ica_op_zoom_in_parameters_dev(ica_ctx* ctx, int batch, int transform_type, int w, int h, int ww,
                                  int hh, double* p_inout);

/* ---- the reference's per-operator ARRAY functions in their own shapes -------------------------------------------
 * float64, channel-last, materialised tensors -- what the fused iteration kernel never builds.  They exist so that code
 * written against the reference's operator functions (its unit tests test/test_derivatives.py, the profiling notebook)
 * runs against this library; `derivatives.py` / `image_optimisation.py` of the Python package call them.
 * nparams is the LAST dimension of DIJ (1..8, not tied to a transform type, as in the reference). */

/* derivatives.py:7-70 jacobian(transform_type, nx, ny) -> J [ny][nx][2 * nparams]. */
int ica_op_ref_jacobian_dev(ica_ctx* ctx, int transform_type, int nx, int ny, double* J_out);

/* image_optimisation.py:159-200 steepest_descent_images(Ix, Iy, J, nparams): Ix, Iy [ny][nx][nz], J [ny][nx][2n]
 * -> DIJ [ny][nx][nz][n]. */
int ica_op_ref_steepest_descent_dev(ica_ctx* ctx, int ny, int nx, int nz, int nparams, const double* Ix,
                                    const double* Iy, const double* J, double* DIJ_out);

/* derivatives.py:73-107 hessian / hessian_robust and image_optimisation.py:82-143 independent_vector(_robust):
 * H = sum rho * DIJ^T DIJ ([n][n]), b = sum rho * DIJ^T DI ([n]); non-finite entries of DIJ / DI count as zero.
 * DI, rho ([ny][nx]), H_out, b_out are nullable (rho NULL = weights 1; b_out requires DI). */
int ica_op_ref_normal_dev(ica_ctx* ctx, int ny, int nx, int nz, int nparams, const double* DIJ, const double* DI,
                          const double* rho, double* H_out, double* b_out);

/* image_optimisation.py:16-53 rhop(t2, lambda, type), element-wise over `count` values. */
int ica_op_ref_rhop_dev(ica_ctx* ctx, size_t count, const double* t2, double lambda, int robust_type, double* out);

/* image_optimisation.py:56-78 robust_error_function(DI [ny][nx][nz], lambda, type) -> rho [ny][nx]. */
int ica_op_ref_robust_error_dev(ica_ctx* ctx, int ny, int nx, int nz, const double* DI, double lambda,
                                int robust_type, double* rho_out);

/* transformation.py:266-318 transform_image (the reference's test-data generator): out = bilinear I(M.x) with
 * skimage.transform.warp(image, tform.inverse, preserve_range=True) semantics (order 1, mode constant, cval 0: a tap
 * outside the image contributes 0; clipped to the input range).  M is [B][9], row-major 3x3, the matrix that maps
 * OUTPUT pixel (col,row,1) to the INPUT position, i.e. inv(params2matrix(gt)) as the reference passes tform.inverse. */
int ica_op_transform_image_dev(ica_ctx* ctx, int batch, int channels, int height, int width, const float* I,
                               const double* M, float* out);

#ifdef __cplusplus
}
#endif
#endif /* ICA_HIP_H */
