/* so3proj.h -- C ABI of libso3proj.so: batched 3x3 SVD -> SO(3) projection on AMD MI355X (gfx950).
 *
 * Drop-in boundary for the hot path of henrikgruner/PoseEstimation.  The reference has no FFI (it is
 * pure Python bottoming out in ATen); each entry point below replaces one ATen op chain of the
 * reference, cited as <file>:<lines> relative to the reference repository root.  A Python binding
 * with the reference's own function names lives in poseestimation_amd/rotation_representation.py;
 * the ctypes stub a maintainer would add to the reference is shown in INTEGRATION.md.
 *
 * Conventions (all entry points)
 *   - Plain pointers and sizes only; no torch / HIP C++ types.  `stream` is a hipStream_t passed as
 *     void* (NULL = the legacy default stream).
 *   - All pointers are DEVICE pointers owned by the caller.  The library never allocates or frees
 *     device memory and keeps no pointer after the call returns.
 *   - Enqueue-only: every call launches on `stream` and returns without synchronising; no hidden
 *     hipMalloc / hipMemcpy / hipDeviceSynchronize, so calls are hipGraph-capturable.  The caller
 *     selects the device (hipSetDevice) -- the library holds no global mutable state and is
 *     re-entrant from several host threads and devices.
 *   - Layout: row-major contiguous 3x3 blocks; (B,9) == (B,3,3); element (i,j) of matrix b at
 *     offset 9*b + 3*i + j.  16-byte aligned base pointers take the vectorised path, any 4-byte
 *     (2-byte for bf16) aligned pointer is accepted.
 *   - Return value: 0 on success, otherwise a hipError_t value (or SO3_ERR_INVALID for a bad
 *     argument); so3_last_error() gives a thread-local description.  Nothing throws.
 *   - NaN/Inf in -> NaN out (the reference's CPU path raises from LAPACK instead; the reference's GPU
 *     path returns NaN; documented divergence).
 *   - B == 0 is a no-op that returns 0.
 */
#ifndef SO3PROJ_H_
#define SO3PROJ_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SO3PROJ_VERSION 210          /* 0.2.0: the reducing entry points exist once (*_v2: workspace nullable, a flags word), the float64 ones
                                        included (so3_angle_error_v2_f64, so3_frob_loss_v2_f64: round 4 had changed their arguments under the old
                                        names); so3_angle_stats's workspace is zero-filled once by the caller.  0.2.1 (210) ADDS the metrics'
                                        backward (so3_angle_bwd_f32 / _f64), so3_geodesic_eps_f64 and two flags; nothing of 200 changed its
                                        arguments, but the float64 entries now reject flag bits they do not know.  A binding checks so3_version(). */
#define SO3_ERR_INVALID (-1)

/* Library version (SO3PROJ_VERSION the library was built with).  A caller compiled against another version of this header must not
 * call the library: argument lists changed between 100 and 200 without new symbol names for every function (210 only adds to 200). */
int so3_version(void);

/* Thread-local description of the last non-zero return on this thread ("" if none). */
const char *so3_last_error(void);

/* Name of the streaming-engine kernel this thread launched last, as a profiler prints it
 * ("so3::k_rows<so3::OpProject<4, false, 100, true>, 2, 3, 256, false, false, 1>"; "" if none yet): the runtime's name
 * of the kernel behind the launch, demangled, so a bench line or a log names the instantiation that actually ran. */
const char *so3_last_kernel(void);

/* ---- K1: symmetric orthogonalization -------------------------------------------------------------
 * R_b = U diag(1,1,det(U V^T)) V^T for M_b = U S V^T.
 * Replaces rotation_representation.py:192-206 (view -> torch.svd -> transpose -> matmul -> det ->
 * cat -> matmul; copies at 3D-Pose/main2.py:34-48, point_cloud/model_fetch.py:13-27, ...).
 *   M    in   B*9 elements (f32, or bf16 bits for the _bf16 variant)
 *   R    out  B*9 float32 (always float32: a bf16 rotation is not orthogonal to 1e-5)
 *   flip out  optional (NULL to skip), B bytes: 1 where det(U V^T) < 0 (<=> det M < 0), else 0
 */
int so3_project_fwd_f32(const float *M, float *R, uint8_t *flip, int64_t B, void *stream);
int so3_project_fwd_bf16(const void *M, float *R, uint8_t *flip, int64_t B, void *stream);

/* ---- K2: backward of K1 ---------------------------------------------------------------------------
 * dM_b = U' Bm V^T with the signed SVD M = U' diag(s') V^T (U', V in SO(3)), A = U'^T G V,
 * Bm_ij = (A_ij - A_ji)/(s'_i + s'_j), Bm_ii = 0.  The SVD is recomputed from M (nothing else is
 * saved by the forward).  Replaces autograd's svd_backward + the backward of the glue ops triggered
 * at 3D-Pose/main.py:90, UPNA/main.py:63, Comparison/main.py:62.
 * Denominators are clamped at 1e-12*s1 (the reference yields inf/NaN at s'_i + s'_j = 0).
 *   M  in  B*9 (f32 / bf16),  G in B*9 float32 (dL/dR),  dM out B*9 (f32 / bf16)
 */
int so3_project_bwd_f32(const float *M, const float *G, float *dM, int64_t B, void *stream);
int so3_project_bwd_bf16(const void *M, const float *G, void *dM, int64_t B, void *stream);

/* float64 variants (the reference's functions accept double tensors; callers on the hot path never pass them):
 * M, R, G, dM are B*9 float64.  Same algorithm in float64 arithmetic, sweeps repeated until the residual is below
 * 1e-14; one row per thread, not tuned.  flip (nullable) as above. */
int so3_project_fwd_f64(const double *M, double *R, uint8_t *flip, int64_t B, void *stream);
int so3_project_bwd_f64(const double *M, const double *G, double *dM, int64_t B, void *stream);

/* ---- the reducing entry points: K3, K3', K4, K1+K4 ----------------------------------------------------------------
 * Each reduces over the batch (loss_sum; sum_count, range_flag) and exists ONCE; how the reduction is finished is chosen by
 * `workspace` and `flags`:
 *   workspace  NULL, or so3_reduce_workspace_bytes() bytes of device memory owned by the caller, ZERO-FILLED ONCE before its
 *              first use (every call leaves it zeroed), used by ONE stream at a time.  With it every workgroup parks its
 *              partial in a slot of its own and the last one to finish (a ticket) sums the slots in a fixed order and writes
 *              the results with plain stores: one launch, and the same input gives the same bits whatever order the
 *              workgroups retire in.  Without it the accumulators are zeroed by a memset / 1-thread launch in front of the
 *              kernel (unless SO3_PREZEROED) and every workgroup adds to them with one atomic.  (Batches of <= 1024 rows run
 *              as one workgroup and never touch it; input that cannot take the streaming engine -- e.g. a bfloat16 view
 *              starting at an odd row -- falls back to the atomics path.)
 *   flags      SO3_RADIANS    angles in radians (default: degrees)                                        [K4, K1+K4]
 *              SO3_PREZEROED  the caller guarantees loss_sum[0] / sum_count[0] / *range_flag are 0 on entry (e.g. fresh slots
 *                             of a zero-filled pool): no init launch, the kernels add to them as they are and one workgroup
 *                             stores the row count into sum_count[1]
 *              SO3_EXACT_F64  K1+K4's sum without per-row angles: the reference's float64 arithmetic on EVERY row (default:
 *                             float32 trace and acos for rows whose cosine is at least 5e-7 away from +-1, float64 inside
 *                             that band; see so3_project_angle_error_v2_f32)
 */
#define SO3_RADIANS 0x1u
#define SO3_PREZEROED 0x2u
#define SO3_EXACT_F64 0x4u
#define SO3_GRAD_SCALAR 0x8u         /* so3_angle_bwd_*: `grad` is ONE element in device memory shared by every row */
#define SO3_F64_MATH 0x10u           /* so3_angle_bwd_f32: angle_error's spelling (float64 arithmetic on float32 data, float64 `grad`) */
size_t so3_reduce_workspace_bytes(void);

/* ---- K3: fused head forward + Frobenius loss + backward (config #4) ------------------------------
 * loss = mean_b ||Rtrue_b - R_b||_F  (3D-Pose/loss.py:7-11; NOT squared),  dM = dloss/dM.
 * Replaces the chain 3D-Pose/main.py:60 (head), :85 (loss), :90 (backward) in one launch.
 *   M         in   B*9 (f32 / bf16)
 *   Rtrue     in   B*9 float32
 *   R         out  optional B*9 float32
 *   dM        out  optional B*9 (f32 / bf16): d(mean loss)/dM, i.e. already divided by B
 *   loss_sum  out  1 double: sum_b ||Rtrue_b - R_b||_F (may be NULL for B <= 1024 when loss_mean is given)
 *   loss_mean out  optional 1 float: (float)(loss_sum / B) -- what loss_frobenius returns (3D-Pose/loss.py:11), so the host
 *                  side needs no launch of its own to turn the float64 sum into the float32 mean
 * A row whose difference is exactly zero contributes zero gradient (the reference gives NaN).
 */
int so3_frob_fwd_bwd_v2_f32(const float *M, const float *Rtrue, float *R, float *dM, double *loss_sum, float *loss_mean,
                            void *workspace, unsigned flags, int64_t B, void *stream);
int so3_frob_fwd_bwd_v2_bf16(const void *M, const float *Rtrue, float *R, void *dM, double *loss_sum, float *loss_mean,
                             void *workspace, unsigned flags, int64_t B, void *stream);

/* K3': stand-alone Frobenius loss for a caller that already holds R_pred (3D-Pose/loss.py:7-11; copies at
 * Comparison/main.py:12-16, UPNA/main.py:27-31, Iterative/loss.py:4-7):
 *   loss_sum out 1 double: sum_b ||Rtrue_b - Rpred_b||_F;  loss_mean out optional 1 float, as above
 *   dRpred   out optional B*9 float32: d(mean loss)/dRpred = (Rpred - Rtrue)/(B ||.||_F); d/dRtrue is its negative.
 */
int so3_frob_loss_v2_f32(const float *Rpred, const float *Rtrue, float *dRpred, double *loss_sum, float *loss_mean,
                         void *workspace, unsigned flags, int64_t B, void *stream);

/* ---- K4: geodesic angle error ----------------------------------------------------------------------
 * theta_b = acos(clamp((tr(R1_b^T R2_b) - 1)/2, -1, 1)) evaluated in float64 on float32 data.
 * Replaces rotation_representation.py:230-242 (angle_error; copies at Comparison/main.py:19-31,
 * Iterative/utility.py:35-47, ...).
 *   deg        out optional B doubles: angle in degrees (radians with SO3_RADIANS)
 *   sum_count  out optional 2 doubles: (sum_b theta_b, B), accumulated on the device.  This pair is what one RCCL
 *                  all-reduce sums across GPUs.
 *   range_flag out optional 1 int32: set to 1 if any cos is outside [-1.1, 1.1] -- the condition
 *                  on which the reference raises ValueError (:237-239); 0 otherwise.
 */
int so3_angle_error_v2(const float *R1, const float *R2, double *deg, double *sum_count, int32_t *range_flag,
                       void *workspace, unsigned flags, int64_t B, void *stream);

/* K1 + K4 fused: theta_b = angle(proj(M_b), Rtrue_b), R not materialised unless requested -- the evaluation step
 * `angle_error(func[rot_rep](out), R).mean()` of 3D-Pose/main.py:60-62,110-112 and UPNA/main.py:54-57 in one
 * launch reading 72 B per row.  deg / sum_count / range_flag as in so3_angle_error_v2; R optional (required only when B is
 * not a multiple of 64 or a pointer is not 16-byte aligned: the tail then runs as K1 followed by K4 and needs the buffer).
 * Arithmetic of the metric: with `deg` every row's angle is the reference's float64 expression on the float32 rotation
 * (1e-9 degrees against so3_angle_error_v2 on the materialised R).  The SUM ALONE (deg == NULL, batches above 1024 rows on the
 * streaming engine) evaluates the same expression in float32 -- trace, cosine, acos -- for every row whose cosine is at least
 * 5e-7 away from +-1, accumulates in float64, and runs the float64 expression for the rows inside that band (angles within
 * 0.057 degrees of 0 or 180, where acos would amplify the float32 trace's round-off): a row then differs from its float64 angle
 * by at most 2e-7 / sin(theta) rad, without bias; measured |difference of the means| 3e-8 degrees over 1M and 16M Haar pairs and
 * <= 2e-6 degrees when every angle is 0.3 degrees (tests/test_gpu_parity.py).  SO3_EXACT_F64 selects float64 for every row. */
int so3_project_angle_error_v2_f32(const float *M, const float *Rtrue, float *R, double *deg, double *sum_count,
                                   int32_t *range_flag, void *workspace, unsigned flags, int64_t B, void *stream);

/* Diagnostic: K1 one row per thread (the same arithmetic, bit for bit, as so3_project_fwd_f32) plus, per row, the device's own
 * verdict: hard[b] = 1 where the quaternion fast path did not certify its result and the row was redone by the Jacobi path.
 * For tests that search for inputs the certificate wrongly accepts (tests/test_gpu_parity.py); not a production entry point. */
int so3_project_fwd_diag_f32(const float *M, float *R, uint8_t *hard, int64_t B, void *stream);

/* dst[i] = src[i] * (*factor), factor a float32 scalar IN DEVICE MEMORY, n elements (float32 / bfloat16; dst may be src).
 * The last step of the chain rule for K3's stored gradient: `loss.backward()` hands the upstream factor over as a 0-dim device
 * tensor, and scaling by it is one launch here instead of a float() / mul / to(bfloat16) chain of the host framework. */
int so3_scale_f32(const float *src, const float *factor, float *dst, int64_t n, void *stream);
int so3_scale_bf16(const void *src, const float *factor, void *dst, int64_t n, void *stream);

/* float64 arguments (the reference's metric and loss functions accept double tensors; rotation_representation.py:232-233 even
 * casts to double itself): the same quantities from float64 data in float64 arithmetic, one row per thread and trip.
 * so3_geodesic_f64 returns float64 radians, so3_frob_loss_v2_f64's dRpred and loss_mean are float64 (loss_mean = loss_sum / B),
 * everything else as in the float32 functions.  `workspace` (nullable) as for the float32 reductions: with it -- and for any batch
 * of <= 1024 rows -- a call is ONE launch whose last workgroup writes sum, count, flag / loss and mean; without it the
 * accumulators are zeroed by a launch in front of the kernel (and the mean written by one behind it).  flags: SO3_RADIANS. */
int so3_angle_error_v2_f64(const double *R1, const double *R2, double *deg, double *sum_count, int32_t *range_flag, void *workspace,
                           unsigned flags, int64_t B, void *stream);
int so3_geodesic_f64(const double *R1, const double *R2, double *theta, int64_t B, void *stream);
int so3_frob_loss_v2_f64(const double *Rpred, const double *Rtrue, double *dRpred, double *loss_sum, double *loss_mean, void *workspace,
                         unsigned flags, int64_t B, void *stream);

/* Float32 radians variant: tr(m1 m2^T), hard clamp to [-1,1], no range check.
 * Replaces rotation_representation.py:209-227 (compute_geodesic_distance_from_two_matrices; copy at
 * point_cloud/main.py:43-57). */
int so3_geodesic_f32(const float *R1, const float *R2, float *theta, int64_t B, void *stream);

/* The same angle with the clamp drawn in by eps and reduced over the batch: geodesic(R1, R2, reduction) of
 * point_cloud/main.py:61-73 (eps = 1e-7: acos(clamp((tr(R1 R2^T) - 1)/2, -1 + eps, 1 - eps)), float32; never called by the
 * reference's loops, kept beside compute_geodesic_distance_from_two_matrices there).
 *   theta  out optional B floats (reduction "none")
 *   sum    optional 1 double of scratch: zeroed by the call, receives sum_b theta_b (float64 accumulation)
 *   result out optional 1 float (needs `sum`): (float) of that sum, or of sum / B when mean != 0 -- written on `stream` behind the kernels
 *   workspace optional so3_reduce_workspace_bytes() of device memory (zeroed once, private to the stream): the reduction is
 *          finished by the kernel's last workgroup -- one launch instead of memset + kernel + mean, and a sum that does not
 *          depend on the order the workgroups finish in
 */
int so3_geodesic_eps_f32(const float *R1, const float *R2, float *theta, double *sum, float *result, int mean, float eps,
                         void *workspace, int64_t B, void *stream);

/* float64 twin of so3_geodesic_eps_f32 (the reference's geodesic() computes in its arguments' dtype): theta B doubles (nullable),
 * sum 1 double of scratch (nullable; zeroed by the call), result 1 double (nullable, needs sum): sum or sum / B.  One row per thread,
 * memset + kernel (+ a 1-thread launch for result); not a benchmark path. */
int so3_geodesic_eps_f64(const double *R1, const double *R2, double *theta, double *sum, double *result, int mean, double eps,
                         int64_t B, void *stream);

/* ---- K4b: backward of the metrics ----------------------------------------------------------------------------------
 * The reference's three metric spellings are plain differentiable tensor code, and its training loops take any of them as the
 * loss (`lossfunc`: point_cloud/main.py:194-197, UPNA/main.py:56-59; geodesic()'s eps = 1e-7 exists for this gradient,
 * point_cloud/main.py:61-73 and the comment at :64).  With c_b = (sum_ij R1_b,ij R2_b,ij - 1)/2 and
 * theta_b = unit * acos(clamp(c_b, -1 + eps, 1 - eps)) -- tr(R1 R2^T) = tr(R1^T R2), so one formula serves
 * rotation_representation.py:209-227 (eps 0, radians), :230-242 (eps 0, degrees, float64 arithmetic) and geodesic (eps 1e-7) --
 *     dR1_b = h_b R2_b,   dR2_b = h_b R1_b,   h_b = (grad_b / grad_div) * unit * (-1 / sqrt(1 - c_b^2)) / 2
 * for rows with -1 + eps <= c_b <= 1 - eps, and 0 outside the clamp: torch.clamp's / torch.min's / torch.max's backward FILL the
 * gradient with 0 there, the reference never multiplies by acos' infinite slope.  (Divergence: a row with c_b = +-1 EXACTLY
 * and eps = 0 gets -+inf from the reference and 0 here.)  NaN rows give NaN.
 *   R1, R2    in  B*9
 *   grad      in  the upstream gradient d loss / d theta_b: B elements, or ONE element with SO3_GRAD_SCALAR (the 0-dim tensor
 *                 autograd hands to a "mean" / "sum" reduction; a device pointer either way, no host sync).  float32 for
 *                 so3_angle_bwd_f32, float64 with SO3_F64_MATH and for so3_angle_bwd_f64.
 *   grad_div      every gradient element is divided by it first (B for reduction "mean": torch divides, it does not multiply
 *                 by 1/B; 1 otherwise)
 *   eps           the clamp is [-1 + eps, 1 - eps], evaluated in the arithmetic of the spelling (float32 unless SO3_F64_MATH)
 *   dR1, dR2  out B*9 each; either may be NULL (not both)
 *   flags         SO3_RADIANS (default: theta in degrees, as K4), SO3_GRAD_SCALAR, SO3_F64_MATH
 * so3_angle_bwd_f32 without SO3_F64_MATH follows the float32 graph operation for operation (trace summed as so3_geodesic_f32
 * sums it, 1 - c c with both roundings, rsqrt, gradient, mask, / 2, times the other rotation); with it, angle_error's: both
 * rotations cast to float64, every step float64, ONE rounding to float32 at the end.  Streaming engine: 72 B read + 36 B (72 B with
 * both gradients) written per row (+ 4 / 8 B of per-row gradient).  so3_angle_bwd_f64: float64 data, one row per thread. */
int so3_angle_bwd_f32(const float *R1, const float *R2, const void *grad, double grad_div, double eps, unsigned flags,
                      float *dR1, float *dR2, int64_t B, void *stream);
int so3_angle_bwd_f64(const double *R1, const double *R2, const double *grad, double grad_div, double eps, unsigned flags,
                      double *dR1, double *dR2, int64_t B, void *stream);

/* ---- next row: the SE(3) pose update fused after the head ------------------------
 * Replaces calculate_T_pred, Iterative/utility.py:90-128 (the head at :105, einsum at :124, the translation
 * update at :116-121 and the 4x4 assembly the reference's `combine`, :63-71, intends):
 *   dR = proj(out[:, :9]);  R_new = dR R_k;  z_new = vz z_k;  x_new = (vx/fx + x_k/z_k) z_new;  y likewise;
 *   T_pred = [[R_new, (x_new, y_new, z_new)^T], [0, 0, 0, 1]].
 *   out12 in B*12 float32 (network output);  Tinit in B*16 float32 (row-major 4x4);  Tpred out B*16 float32;
 *   fx, fy: focal lengths in pixels (get_scene_parameters, utility.py:73-88: 50 / (36/320) = 444.44).
 * The backward gives dL/dout12 for upstream G = dL/dTpred (B*16); T_init is treated as a constant, as the
 * reference's loop detaches it (Iterative/main.py:97).
 */
int so3_se3_update_f32(const float *out12, const float *Tinit, float *Tpred, float fx, float fy, int64_t B,
                       void *stream);
int so3_se3_update_bwd_f32(const float *out12, const float *Tinit, const float *G, float *dout12, float fx,
                           float fy, int64_t B, void *stream);

/* ---- next row: the 6D Gram-Schmidt head and its backward -------------------------
 * x = a/|a|, z = (x x b)/|x x b|, y = z x x, R = [x y z] as columns; (a, b) = the two halves of each 6-vector.
 * Replaces rotation_representation.py:21-36 (compute_rotation_matrix_from_ortho6d; duplicate at :174-189),
 * the reference's main comparison head (transform_output['6D'], Comparison/models.py:19).
 *   X in B*6 float32;  R out B*9 float32;  G in B*9 float32 (dL/dR);  dX out B*6 float32.
 * No epsilon in the norms, as in the reference: a zero or parallel pair gives Inf/NaN.
 */
int so3_ortho6d_fwd_f32(const float *X, float *R, int64_t B, void *stream);
int so3_ortho6d_bwd_f32(const float *X, const float *G, float *dX, int64_t B, void *stream);

/* ---- next row f5: the remaining heads of the reference's dispatch tables -------------
 * Model.func / Model.dimension (Comparison/models.py:18-19, point_cloud/model_fetch.py:153-154), func
 * (3D-Pose/main.py:46,101) and transform_output (rotation_representation.py:323-324) map a key to a head; with
 * these four plus the SVD and 6D heads above every key is served natively ("Direct" is a reshape).
 *   quat     X B*4  (w,x,y,z) -> n = q / max(|q|, 1e-8) -> R(n)      rotation_representation.py:39-50, 137-171
 *   euler    X B*3  R from (c1,s1)=e0, (c2,s2)=e2, (c3,s3)=e1         rotation_representation.py:92-113
 *   ortho5d  X B*5  stereographic un-projection of X[2:5] * (1+sqrt2, 1+sqrt2, sqrt2), then the 6D head
 *                                                                     rotation_representation.py:69-90, 118-134
 *   expmap   X B*3  so(3) exponential map, theta = sqrt(max(|v|^2, 1e-4))   rotation_representation.py:245-275,
 *                   reached as vec_3d_to_SO3 (:309-321, transform_output['3D'])
 * Forward: R out B*9 float32.  Backward: G in B*9 float32 (dL/dR), dX out shaped like X (what the reference
 * gets from autograd through the same formulas).  Degenerate input behaves like the reference's float32 graph
 * (zero 5D tail or zero 6D halves: Inf/NaN), except that an exactly-zero quaternion gives the clamped
 * gradient G-terms/1e-8 where autograd yields NaN.
 */
int so3_quat_fwd_f32(const float *X, float *R, int64_t B, void *stream);
int so3_quat_bwd_f32(const float *X, const float *G, float *dX, int64_t B, void *stream);
int so3_euler_fwd_f32(const float *X, float *R, int64_t B, void *stream);
int so3_euler_bwd_f32(const float *X, const float *G, float *dX, int64_t B, void *stream);
int so3_ortho5d_fwd_f32(const float *X, float *R, int64_t B, void *stream);
int so3_ortho5d_bwd_f32(const float *X, const float *G, float *dX, int64_t B, void *stream);
int so3_expmap_fwd_f32(const float *X, float *R, int64_t B, void *stream);
int so3_expmap_bwd_f32(const float *X, const float *G, float *dX, int64_t B, void *stream);

/* ---- row a7: the cloud side of the point-cloud path -------------------------------------
 * so3_rotate_clouds_f32 replaces the pairing rule of the training loop, point_cloud/main.py:173-181
 *   (expand gt_rmat to every point, bmm, view) and, with transposed != 0, the `.transpose(1, 2)` at :183 as well:
 *     P B*N*3, R B*9  ->  out[b][i][:] = R_b p_i   (transposed: out[b][:][i], the (B,3,N) layout the network reads).
 * so3_pc_normalize_f32 replaces pc_normalize, point_cloud/prepare.py:51-56, for a batch of clouds:
 *     centre = (max + min)/2 per axis, scale = |max - min| of the centred cloud, out = (P - centre)/scale;
 *     centroid (B*3) and scale (B) are optional outputs.  float32 (the reference runs it in numpy float64).
 */
int so3_rotate_clouds_f32(const float *P, const float *R, float *out, int transposed, int64_t B, int32_t N, void *stream);
int so3_pc_normalize_f32(const float *P, float *out, float *centroid, float *scale, int64_t B, int32_t N, void *stream);

/* ---- next row f6: the ADD-L1 losses that consume calculate_T_pred's output, with their gradient ----------------
 * Replaces Iterative/loss.py:10-26 (compute_ADD_L1_loss), :29-48 (compute_disentangled_ADD_L1_loss) and :51-70
 * (transform_pts), called at Iterative/main.py:94-95,150-151,196-197 right after calculate_T_pred, plus the autograd
 * of the loss w.r.t. the predicted pose (which so3_se3_update_bwd_f32 then takes to the network output).
 *   Tgt, Tpred   in  B*16 float32, row-major 4x4 poses
 *   points       in  B*N*3 float32 model points (the reference's `verts`), N >= 1
 *   so3_add_l1_f32:  dist_b = mean over points and coordinates of |T_gt p - T_pred p|
 *     dists      out optional B float32 (use_batch_mean=False);  loss_sum out optional double[1] = sum_b dist_b
 *   so3_add_l1_disentangled_f32:  the three terms of the disentangled loss per sample -- rotation (R_pred with
 *     T_gt's translation), translation x,y and depth z (each with T_gt's rotation; these two do not depend on the
 *     points: the transformed clouds differ by a constant vector)
 *     loss_sum   out double[3] = sum_b (rot_b, transl_b, depth_b);  the reference's value is their total / B
 *   dTpred       out optional B*16 float32: grad_scale * d(sum_b loss_b)/dTpred  (grad_scale = 1/B for the batch
 *                mean; the bottom row is 0; d|x|/dx = sgn(x) with sgn(0) = 0, as autograd)
 * loss_sum is zeroed by the call.  One pass over the points (12 B per point: HBM-read bound).
 */
int so3_add_l1_f32(const float *Tgt, const float *Tpred, const float *points, float *dists, double *loss_sum,
                   float *dTpred, float grad_scale, int64_t B, int32_t N, void *stream);
int so3_add_l1_disentangled_f32(const float *Tpred, const float *Tgt, const float *points, double *loss_sum,
                                float *dTpred, float grad_scale, int64_t B, int32_t N, void *stream);

/* ---- next row: per-class evaluation statistics on K4's angles ----------------------
 * Replaces the host-side numpy block of 3D-Pose/test_per_class.py:174-216 (np.mean / np.median / np.std / np.max
 * and the accuracy thresholds (x < 30|15|7.5).sum()/len(x)), which the reference feeds one sample at a time.
 *   deg       in  B float64 angles (degrees), e.g. so3_angle_error's per-row output
 *   cls       in  optional B int32 class ids in [0, ncls) (NULL: one class); rows with other ids are ignored
 *   stats     out ncls x 8 doubles: count, mean, std (population, as np.std), max, median (EXACT: radix select
 *                 on the float64 bits, the two middle elements averaged as np.median does), acc<30, acc<15, acc<7.5
 *   workspace     caller-owned scratch of so3_angle_stats_workspace_bytes() bytes (~10 MB: histograms and a buffer for the
 *                 candidates of the medians), ZERO-FILLED ONCE before its first use -- every call leaves its sums and histograms
 *                 zeroed and re-initialises its control words in its first launch --, used by ONE stream at a time (re-zero it
 *                 after a call that returned an error).  A call whose launches are delayed by other work on the device (its
 *                 workgroups wait for each other, boundedly) is slower, never wrong, and leaves the workspace as usable as before.
 * ncls <= 64.  A class containing a NaN angle reports NaN for mean/std/max/median, as numpy does.
 * Two launches, two passes over the rows: a histogram pass, and a pass that compacts the rows of the medians' 1/16-octave bins and
 * then, one workgroup per class behind a ticket, selects among them.
 */
size_t so3_angle_stats_workspace_bytes(void);
int so3_angle_stats(const double *deg, const int32_t *cls, int32_t ncls, double *stats, void *workspace,
                    int64_t B, void *stream);

/* ---- K5: fused Kabsch (config #3) ------------------------------------------------------------------
 * H_b = sum_i q_bi p_bi^T (= bmm(Q^T, P)),  R_b = proj_SO(3)(H_b) = argmin_R sum_i |R p_bi - q_bi|^2.
 * No centring: the reference's pairing rule q = R p has no translation (point_cloud/main.py:173-181).
 * The reference has no closed-form solve (it learns R with PointNet++); the oracle is
 * symmetric_orthogonalization(bmm(Q^T, P)).
 *   P, Q in  B*N*3 float32, cloud-major then point-major (the (B,1024,3) tensors of
 *            point_cloud/main.py:171);  R out B*9 float32;  H out optional B*9 float32.
 */
int so3_kabsch_f32(const float *P, const float *Q, float *R, float *H, int64_t B, int32_t N,
                   void *stream);

/* ---- next row: on-device pair synthesis for Kabsch ------------------------------
 * so3_rotations_axis_angle_f32: the arithmetic of the reference's sampler, point_cloud/prepare.py:21-49
 *   (normalize_vector :12-18, quaternion (cos theta, axis sin theta) -> matrix :27-47), given the random
 *   draws theta (B) and axis (B,3) (the reference draws them with numpy / torch.randn: RNG stays with the caller).
 * so3_kabsch_synth_f32: K5 with the second cloud synthesised on the fly,
 *       q_bi = Rgt_b p_bi + sigma * n(seed, b, i)        (pairing rule point_cloud/main.py:173-181, plus noise)
 *   so only P (12 B per point) is read from HBM instead of P and Q.  n is a stateless counter-based standard
 *   normal (32-bit mix -> three Box-Muller pairs per PAIR of points 64 apart, csrc/so3proj.hip `synth_normal3x2`; restated by the
 *   test oracle, oracle/ `synth_normal_np`).  The stream a seed names belongs to the library's build, not to this
 *   ABI: it is the same on every device and launch shape, and it changed between rounds of this library (last: round 6).
 *   sigma = 0 skips the generator.  R, H as in so3_kabsch_f32.
 */
int so3_rotations_axis_angle_f32(const float *theta, const float *axis, float *R, int64_t B, void *stream);
int so3_kabsch_synth_f32(const float *P, const float *Rgt, float sigma, uint32_t seed, float *R, float *H,
                         int64_t B, int32_t N, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SO3PROJ_H_ */
