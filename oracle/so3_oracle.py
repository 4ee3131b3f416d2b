"""CPU oracle for the SVD -> SO(3) hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, on the CPU, what the reference computes on the path named by BASELINE.json.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product (`poseestimation_amd/`) never does and fails loudly when its HIP library is missing.

Two flavours of every function:

* `*_np`   -- numpy, float64 LAPACK (`np.linalg.svd`).  The mathematical answer; used as the
              "truth" that both the reference's fp32 path and the HIP kernels are measured against.
* `*_torch`-- the reference's own ATen call sequence (torch.linalg.svd -> det -> scale last row
              of Vh -> matmul) in the input dtype, on CPU tensors.  `torch.svd` (what the reference
              literally calls) and `torch.linalg.svd` are bitwise identical on torch 2.10 (SURVEY
              section 8c), so this is the "port" that `bench.py` times as `cpu_baseline`.

Pinned (tests/test_oracle_golden.py) against golden vectors produced by importing the reference's
own `rotation_representation.py` in the build container (tools/gen_golden.py, tests/golden/*.npz).

Reference lines followed (all under /root/reference):
  symmetric_orthogonalization          rotation_representation.py:192-206
  compute_geodesic_distance_from_...   rotation_representation.py:209-227
  angle_error                          rotation_representation.py:230-242
  loss_frobenius                       3D-Pose/loss.py:7-11
  rotation sampler (Kabsch pairs)      point_cloud/prepare.py:21-49, point_cloud/main.py:173-181
  6D Gram-Schmidt head (next row f2)   rotation_representation.py:21-36
  per-class statistics (next row f3)   3D-Pose/test_per_class.py:174-175,206-216
  SE(3) pose update (next row f1)      Iterative/utility.py:63-128
  other heads (next row f5)            rotation_representation.py:39-50,69-171,245-321
  ADD-L1 losses (next row f6)          Iterative/loss.py:10-70
  cloud pairing / pc_normalize (a7)    point_cloud/main.py:173-183, point_cloud/prepare.py:51-56
"""
from __future__ import annotations

import numpy as np

try:  # torch is only needed by the *_torch flavour
    import torch
except Exception:  # pragma: no cover
    torch = None


# --------------------------------------------------------------------------------------------
# numpy / float64
# --------------------------------------------------------------------------------------------
def symmetric_orthogonalization_np(x, return_parts=False):
    """R = U diag(1,1,det(U V^T)) V^T for every 3x3 block of x (rotation_representation.py:199-206).

    float64 LAPACK regardless of the input dtype.  Returns (B,3,3) float64; with return_parts also
    the singular values (B,3) and the flip sign d = det(U V^T) (B,).
    """
    m = np.asarray(x, dtype=np.float64).reshape(-1, 3, 3)          # :199  x.view(-1, 3, 3)
    u, s, vt = np.linalg.svd(m)                                    # :200  torch.svd (v -> vt, :201)
    d = np.linalg.det(u @ vt)                                      # :202
    vt = vt.copy()
    vt[:, 2, :] *= d[:, None]                                      # :204  last row of vt times det
    r = u @ vt                                                     # :205
    if return_parts:
        return r, s, d
    return r


def flip_flag_np(x):
    """det(U V^T) < 0  <=>  det(M) < 0, evaluated in float64 (rotation_representation.py:202)."""
    m = np.asarray(x, dtype=np.float64).reshape(-1, 3, 3)
    return np.linalg.det(m) < 0


def angle_error_np(r1, r2, check=True):
    """Geodesic angle in float64 degrees, tr(R1^T R2) (rotation_representation.py:230-242)."""
    a = np.asarray(r1, dtype=np.float64).reshape(-1, 3, 3)
    b = np.asarray(r2, dtype=np.float64).reshape(-1, 3, 3)
    tr = np.einsum("bji,bji->b", a, b)                              # trace(R1^T R2)  :232-235
    cos = (tr - 1.0) / 2.0                                          # :236
    if check and (np.any(cos < -1.1) or np.any(cos > 1.1)):         # :237-239
        raise ValueError("angle out of range, input probably not proper rotation matrices")
    cos = np.clip(cos, -1.0, 1.0)                                   # :240
    return np.arccos(cos) * (180.0 / np.pi)                         # :241-242


def geodesic_np(m1, m2, dtype=np.float32):
    """Radians, input dtype, tr(m1 m2^T), hard clamp (rotation_representation.py:209-227)."""
    a = np.asarray(m1, dtype=dtype).reshape(-1, 3, 3)
    b = np.asarray(m2, dtype=dtype).reshape(-1, 3, 3)
    m = a @ b.transpose(0, 2, 1)                                    # :215
    cos = (m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2] - dtype(1)) / dtype(2)   # :217
    cos = np.minimum(cos, dtype(1))                                 # :218
    cos = np.maximum(cos, dtype(-1))                                # :219
    return np.arccos(cos)                                           # :222


def geodesic_eps_np(r1, r2, reduction="mean", eps=1e-7):
    """geodesic(R1, R2, reduction), point_cloud/main.py:61-73: float32, tr(R1 R2^T), clamp to [-1 + eps, 1 - eps], acos;
    "none" -> (B,), "mean" / "sum" -> scalar; any other string -> None (the reference's if-chain falls through)."""
    a = np.asarray(r1, dtype=np.float32).reshape(-1, 3, 3)
    b = np.asarray(r2, dtype=np.float32).reshape(-1, 3, 3)
    diffs = a @ b.transpose(0, 2, 1)                                # :63
    traces = np.trace(diffs, axis1=-2, axis2=-1).astype(np.float32)  # :65
    lo, hi = np.float32(-1 + eps), np.float32(1 - eps)              # torch.clamp's scalars on a float32 tensor
    dists = np.arccos(np.clip((traces - np.float32(1)) / np.float32(2), lo, hi))   # :66-67
    if reduction == "none":
        return dists
    if reduction == "mean":
        return np.float32(dists.astype(np.float64).mean())
    if reduction == "sum":
        return np.float32(dists.astype(np.float64).sum())
    return None


def metric_backward_np(r1, r2, upstream, eps=0.0, unit=1.0, divisor=1.0, clamp_dtype=np.float64):
    """Gradient of theta_b = unit * acos(clamp(c_b, -1 + eps, 1 - eps)), c_b = (sum_ij R1_ij R2_ij - 1)/2, with respect to both
    rotations, in float64 -- what autograd computes through geodesic (point_cloud/main.py:61-73: eps 1e-7, unit 1),
    compute_geodesic_distance_from_two_matrices (rotation_representation.py:209-227: eps 0, unit 1) and angle_error (:230-242:
    eps 0, unit 180/pi): tr(R1 R2^T) = tr(R1^T R2) = sum_ij R1_ij R2_ij, d acos(c)/dc = -1/sqrt(1 - c^2), and torch.clamp's /
    torch.min's / torch.max's backward fill the gradient with 0 outside the clamp.  `upstream` = d loss / d theta per row (or one
    number for all rows), divided by `divisor` (the mean's B).  clamp_dtype: the dtype the clamp's bounds are rounded to (float32 for
    geodesic on float32 tensors).  Rows with c = +-1 exactly get 0 (the reference: -+inf); documented divergence.
    Returns (dR1, dR2, c)."""
    a = np.asarray(r1, dtype=np.float64).reshape(-1, 3, 3)
    b = np.asarray(r2, dtype=np.float64).reshape(-1, 3, 3)
    c = ((a * b).sum(axis=(1, 2)) - 1.0) / 2.0
    lo = float(clamp_dtype(-1) + clamp_dtype(eps))
    hi = float(clamp_dtype(1) - clamp_dtype(eps))
    om = 1.0 - c * c
    inside = (c >= lo) & (c <= hi) & (om > 0)
    g = np.broadcast_to(np.asarray(upstream, dtype=np.float64), c.shape) / divisor
    with np.errstate(divide="ignore", invalid="ignore"):
        h = np.where(inside, g * unit * (-1.0 / np.sqrt(np.where(inside, om, 1.0))) / 2.0, 0.0)
    h = np.where(np.isnan(c), np.nan, h)
    return h[:, None, None] * b, h[:, None, None] * a, c


def loss_frobenius_np(r_pred, r_true):
    """mean_b ||R_true - R_pred||_F, not squared (3D-Pose/loss.py:7-11)."""
    d = np.asarray(r_true, np.float64).reshape(-1, 3, 3) - np.asarray(r_pred, np.float64).reshape(-1, 3, 3)
    return np.sqrt((d * d).sum(axis=(1, 2))).mean()


def projection_backward_np(x, g):
    """dL/dM for R = proj(M), given G = dL/dR; closed form.

    With the *signed* SVD M = U' diag(s') V^T, U',V in SO(3), s' = (s1, s2, d*s3):
        A = U'^T G V,   B_ij = (A_ij - A_ji) / (s'_i + s'_j),  B_ii = 0,   dM = U' B V^T.
    Checked against torch autograd through the reference function in tests/test_oracle_golden.py.
    """
    m = np.asarray(x, dtype=np.float64).reshape(-1, 3, 3)
    g = np.asarray(g, dtype=np.float64).reshape(-1, 3, 3)
    u, s, vt = np.linalg.svd(m)
    d = np.linalg.det(u @ vt)
    u = u.copy()
    u[:, :, 2] *= d[:, None]            # U' = U diag(1,1,d)  (V kept as LAPACK returns it)
    sp = s.copy()
    sp[:, 2] *= d                       # s' so that M = U' diag(s') V^T exactly
    v = vt.transpose(0, 2, 1)
    a = u.transpose(0, 2, 1) @ g @ v
    den = sp[:, :, None] + sp[:, None, :]
    with np.errstate(divide="ignore", invalid="ignore"):
        b = (a - a.transpose(0, 2, 1)) / den
    idx = np.arange(3)
    b[:, idx, idx] = 0.0
    return u @ b @ vt


def frobenius_fwd_bwd_np(x, r_true):
    """loss = mean_b ||R_true - proj(M_b)||_F and dloss/dM (3D-Pose/main.py:60,85,90 chain)."""
    r = symmetric_orthogonalization_np(x)
    t = np.asarray(r_true, np.float64).reshape(-1, 3, 3)
    diff = r - t
    nrm = np.sqrt((diff * diff).sum(axis=(1, 2)))
    b = r.shape[0]
    with np.errstate(divide="ignore", invalid="ignore"):
        g = diff / (nrm[:, None, None] * b)
    return nrm.mean(), projection_backward_np(x, g), r


def sample_rotations_axis_angle_np(rng, batch):
    """The reference's pair generator (point_cloud/prepare.py:21-49), numpy/float64, CPU.

    theta ~ U(-pi, pi), axis = normalised N(0,I); quaternion (cos theta, axis sin theta), i.e. the
    rotation angle is 2*theta -- reproduced as written.
    """
    theta = rng.uniform(-1.0, 1.0, batch) * np.pi                   # :23
    sin = np.sin(theta)                                             # :24
    axis = rng.standard_normal((batch, 3))                          # :25
    axis = axis / np.maximum(np.linalg.norm(axis, axis=1, keepdims=True), 1e-8)   # :26 / :12-18
    qw = np.cos(theta)                                              # :27
    qx, qy, qz = axis[:, 0] * sin, axis[:, 1] * sin, axis[:, 2] * sin   # :28-30
    xx, yy, zz = qx * qx, qy * qy, qz * qz
    xy, xz, yz = qx * qy, qx * qz, qy * qz
    xw, yw, zw = qx * qw, qy * qw, qz * qw
    row0 = np.stack((1 - 2 * yy - 2 * zz, 2 * xy - 2 * zw, 2 * xz + 2 * yw), 1)   # :43
    row1 = np.stack((2 * xy + 2 * zw, 1 - 2 * xx - 2 * zz, 2 * yz - 2 * xw), 1)   # :44
    row2 = np.stack((2 * xz - 2 * yw, 2 * yz + 2 * xw, 1 - 2 * xx - 2 * yy), 1)   # :45
    return np.stack((row0, row1, row2), 1)                          # :47


def rotations_from_draws_np(theta, axis):
    """The arithmetic of the reference sampler (point_cloud/prepare.py:24-47) for given draws theta (B,), axis (B,3)."""
    theta = np.asarray(theta, np.float64)
    axis = np.asarray(axis, np.float64)
    axis = axis / np.maximum(np.linalg.norm(axis, axis=1, keepdims=True), 1e-8)
    sin, qw = np.sin(theta), np.cos(theta)
    qx, qy, qz = axis[:, 0] * sin, axis[:, 1] * sin, axis[:, 2] * sin
    xx, yy, zz, xy, xz, yz = qx * qx, qy * qy, qz * qz, qx * qy, qx * qz, qy * qz
    xw, yw, zw = qx * qw, qy * qw, qz * qw
    return np.stack((np.stack((1 - 2 * yy - 2 * zz, 2 * xy - 2 * zw, 2 * xz + 2 * yw), 1),
                     np.stack((2 * xy + 2 * zw, 1 - 2 * xx - 2 * zz, 2 * yz - 2 * xw), 1),
                     np.stack((2 * xz - 2 * yw, 2 * yz + 2 * xw, 1 - 2 * xx - 2 * yy), 1)), 1)


def _mix32(x):
    x = np.asarray(x, np.uint64) & 0xFFFFFFFF
    x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF
    x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def synth_pair_uniforms_np(seed, cloud, pair):
    """The six uniforms of a PAIR of points behind synth_normal_np (csrc/so3proj.hip `synth_normal3x2`): one full 32-bit mix of (seed, cloud,
    pair), four single-multiply rounds of it, 23 bits each -- radii u_rA, u_rB, u_rC in (0, 1], angles u_A, u_B, u_C in [0, 1) turns; u_C from
    the low bytes of the first four hashes.  Returned in the order (u_rA, u_A, u_rB, u_B, u_rC, u_C)."""
    cloud, pair = np.asarray(cloud, np.uint64), np.asarray(pair, np.uint64)
    m = np.uint64(0xFFFFFFFF)
    key = _mix32(np.uint64(seed) ^ _mix32((cloud * 0x9e3779b9 + 0x85ebca6b) & m))
    h0 = _mix32(key ^ ((pair * 0x9e3779b9 + 0xc2b2ae35) & m))

    def round_(h, s1, mul, s2):
        h = h ^ (h >> np.uint64(s1)); h = (h * np.uint64(mul)) & m
        return h ^ (h >> np.uint64(s2))
    h1 = round_((h0 + 0x27d4eb2f) & m, 16, 0x7feb352d, 15)
    h2 = round_(h0 ^ np.uint64(0x165667b1), 15, 0x2c1b3c6d, 16)
    h3 = round_((h0 + 0x9e3779b1) & m, 17, 0x297a2d39, 14)
    h4 = round_(h0 ^ np.uint64(0x85ebca77), 14, 0xc2b2ae3d, 17)
    unit = lambda h: (h >> 9).astype(np.float64) / 8388608.0                     # the device's float in [1, 2), minus one
    low = ((h0 & 0xFF) << 24) | ((h1 & 0xFF) << 16) | ((h2 & 0xFF) << 8) | (h3 & 0xFF)
    return 1.0 - unit(h0), unit(h1), 1.0 - unit(h2), unit(h3), 1.0 - unit(h4), unit(low)


def synth_normal_np(seed, cloud, point, comp):
    """The stateless standard normals of so3_kabsch_synth_f32, restated: three Box-Muller pairs per PAIR of points.  Point p belongs to pair
    (p >> 7) * 64 + (p & 63) and is its point a or b by bit 6 (the two points a lane of the kernel holds in neighbouring trips);
    a: (r_A cos, r_A sin)(2 pi u_A), r_C cos(2 pi u_C);  b: (r_B cos, r_B sin)(2 pi u_B), r_C sin(2 pi u_C)."""
    comp, point = np.asarray(comp), np.asarray(point, np.int64)
    pair, second = (point >> 7) * 64 + (point & 63), ((point >> 6) & 1).astype(bool)
    ura, ua, urb, ub, urc, uc = synth_pair_uniforms_np(seed, cloud, pair)
    r01 = np.sqrt(-2.0 * np.log(np.where(second, urb, ura)))
    t01 = 2 * np.pi * np.where(second, ub, ua)
    rc, tc = np.sqrt(-2.0 * np.log(urc)), 2 * np.pi * uc
    return np.where(comp == 0, r01 * np.cos(t01), np.where(comp == 1, r01 * np.sin(t01), rc * np.where(second, np.sin(tc), np.cos(tc))))


def synth_pairs_np(p, r_gt, sigma, seed):
    """q_bi = R_gt_b p_bi + sigma n(seed, b, i, c)  (pairing rule point_cloud/main.py:173-181 plus noise)."""
    p = np.asarray(p, np.float64)
    b, n, _ = p.shape
    q = np.einsum("bac,bic->bia", np.asarray(r_gt, np.float64).reshape(b, 3, 3), p)
    if sigma != 0:
        cb, pi, cc = np.meshgrid(np.arange(b), np.arange(n), np.arange(3), indexing="ij")
        q = q + sigma * synth_normal_np(seed, cb, pi, cc)
    return q


def cross_covariance_np(p, q):
    """H_b = sum_i q_i p_i^T = bmm(Q^T, P), float64 (config #3)."""
    p = np.asarray(p, np.float64)
    q = np.asarray(q, np.float64)
    return np.einsum("bia,bic->bac", q, p)


def rotate_clouds_np(p, r, transposed=False):
    """q_i = R_b p_i for every point (point_cloud/main.py:173-181); transposed -> (B,3,N) as main.py:183 feeds the net."""
    q = np.einsum("bac,bic->bia", np.asarray(r, np.float64).reshape(-1, 3, 3), np.asarray(p, np.float64))
    return q.transpose(0, 2, 1) if transposed else q


def pc_normalize_np(pc):
    """point_cloud/prepare.py:51-56 for one (N,3) cloud or a (B,N,3) batch, float64: (pc, centroid, scale)."""
    pc = np.asarray(pc, np.float64)
    centroid = (np.max(pc, axis=-2, keepdims=True) + np.min(pc, axis=-2, keepdims=True)) / 2          # :52
    pc = pc - centroid                                                                                # :53
    scale = np.linalg.norm(np.max(pc, axis=-2) - np.min(pc, axis=-2), axis=-1)                        # :54
    pc = pc / scale[..., None, None]                                                                  # :55
    return pc, centroid.squeeze(-2), scale


def kabsch_np(p, q):
    """argmin_R sum_i |R p_i - q_i|^2 over SO(3) = proj(H) (no centring, as the pairing rule
    point_cloud/main.py:173-181 has no translation)."""
    return symmetric_orthogonalization_np(cross_covariance_np(p, q))


def se3_update_np(model_output, t_init, fx=50 / (36 / 320), fy=50 / (36 / 320)):
    """calculate_T_pred (Iterative/utility.py:90-128) in float64; focal lengths from get_scene_parameters (:73-88)."""
    o = np.asarray(model_output, np.float64)
    t = np.asarray(t_init, np.float64).reshape(-1, 4, 4)
    dr = symmetric_orthogonalization_np(o[:, :9])                    # :105
    vx, vy, vz = o[:, 9], o[:, 10], o[:, 11]                         # :106-108
    r_k = t[:, :3, :3]                                               # :110
    z_k, x_k, y_k = t[:, 2, 3], t[:, 0, 3], t[:, 1, 3]               # :114,118,119
    z_new = vz * z_k                                                 # :116
    x_new = (vx / fx + x_k / z_k) * z_new                            # :120
    y_new = (vy / fy + y_k / z_k) * z_new                            # :121
    tp = np.ones((o.shape[0], 4, 4))                                 # combine, :63-71 (as intended)
    tp[:, :3, :3] = np.einsum("bij,bjk->bik", dr, r_k)               # :124
    tp[:, 0, 3], tp[:, 1, 3], tp[:, 2, 3] = x_new, y_new, z_new
    tp[:, 3, :3] = 0
    return tp


def se3_update_backward_np(model_output, t_init, g, fx=50 / (36 / 320), fy=50 / (36 / 320)):
    """dL/dmodel_output[:, :12] for upstream G = dL/dT_pred (T_init constant); checked against the reference's autograd."""
    o = np.asarray(model_output, np.float64)
    t = np.asarray(t_init, np.float64).reshape(-1, 4, 4)
    g = np.asarray(g, np.float64).reshape(-1, 4, 4)
    gdr = np.einsum("bik,bjk->bij", g[:, :3, :3], t[:, :3, :3])      # G_R R_k^T
    d = np.zeros((o.shape[0], 12))
    d[:, :9] = projection_backward_np(o[:, :9], gdr).reshape(-1, 9)
    z_k, x_k, y_k = t[:, 2, 3], t[:, 0, 3], t[:, 1, 3]
    z_new = o[:, 11] * z_k
    ax, ay = o[:, 9] / fx + x_k / z_k, o[:, 10] / fy + y_k / z_k
    d[:, 9] = g[:, 0, 3] * z_new / fx
    d[:, 10] = g[:, 1, 3] * z_new / fy
    d[:, 11] = z_k * (g[:, 2, 3] + g[:, 0, 3] * ax + g[:, 1, 3] * ay)
    return d


def angle_statistics_np(angles, class_ids=None, num_classes=1):
    """Per-class evaluation statistics exactly as 3D-Pose/test_per_class.py:174-175,206-216 computes them with
    numpy on the host: np.mean, np.median, np.std (population), np.max, and (x < t).sum()/len(x) for t = 30, 15, 7.5."""
    a = np.asarray(angles, np.float64).reshape(-1)
    c = np.zeros(a.shape[0], np.int64) if class_ids is None else np.asarray(class_ids).reshape(-1)
    out = {k: np.full(num_classes, np.nan) for k in ("count", "mean", "std", "max", "median", "acc30", "acc15", "acc7.5")}
    for k in range(num_classes):
        x = a[c == k]
        out["count"][k] = len(x)
        if len(x) == 0:
            continue
        out["mean"][k], out["median"][k], out["std"][k], out["max"][k] = np.mean(x), np.median(x), np.std(x), np.max(x)   # :174-175
        out["acc30"][k] = (x < 30).sum() / len(x)       # :214
        out["acc15"][k] = (x < 15).sum() / len(x)       # :215
        out["acc7.5"][k] = (x < 7.5).sum() / len(x)     # :216
    return out


def ortho6d_np(poses):
    """6D head: x = a/|a|, z = (x x b)/|.|, y = z x x, columns (x, y, z)  (rotation_representation.py:21-36)."""
    p = np.asarray(poses, np.float64)
    a, b = p[..., 0:3], p[..., 3:6]                                  # :29-30
    x = a / np.linalg.norm(a, axis=-1, keepdims=True)                # :31
    z = np.cross(x, b)                                               # :32
    z = z / np.linalg.norm(z, axis=-1, keepdims=True)                # :33
    y = np.cross(z, x)                                               # :34
    return np.stack((x, y, z), -1)                                   # :35


def ortho6d_backward_np(poses, g):
    """dL/dposes for the 6D head given G = dL/dR (closed form; checked against the reference's autograd)."""
    p = np.asarray(poses, np.float64).reshape(-1, 6)
    g = np.asarray(g, np.float64).reshape(-1, 3, 3)
    a, b = p[:, 0:3], p[:, 3:6]
    na = np.linalg.norm(a, axis=1, keepdims=True)
    x = a / na
    w = np.cross(x, b)
    nw = np.linalg.norm(w, axis=1, keepdims=True)
    z = w / nw
    gx, gy, gz = g[:, :, 0], g[:, :, 1], g[:, :, 2]
    gzt = gz + np.cross(x, gy)                                       # y = z x x
    gxt = gx + np.cross(gy, z)
    gw = (gzt - z * (z * gzt).sum(1, keepdims=True)) / nw            # z = w/|w|
    gxt = gxt + np.cross(b, gw)                                      # w = x x b
    gb = np.cross(gw, x)
    ga = (gxt - x * (x * gxt).sum(1, keepdims=True)) / na            # x = a/|a|
    return np.concatenate((ga, gb), 1)


# --------------------------------------------------------------------------------------------
# next row f5: the other heads of the dispatch tables.  Restated op for op with torch (float64 by default) so
# that the backward is autograd through the same graph the reference differentiates -- no closed forms here.
# --------------------------------------------------------------------------------------------
def quat_torch(q):
    """(B,4) (w,x,y,z) -> (B,3,3); rotation_representation.py:39-50 (normalize_vector) and :137-171."""
    mag = torch.sqrt(q.pow(2).sum(1))                                            # :46
    mag = torch.max(mag, torch.tensor([1e-8], dtype=q.dtype))                    # :47
    n = q / mag.view(-1, 1)                                                      # :48-49
    w, x, y, z = n[:, 0:1], n[:, 1:2], n[:, 2:3], n[:, 3:4]                      # :148-151
    xx, yy, zz, xy, xz, yz, xw, yw, zw = x * x, y * y, z * z, x * y, x * z, y * z, x * w, y * w, z * w   # :154-162
    r0 = torch.cat((1 - 2 * yy - 2 * zz, 2 * xy - 2 * zw, 2 * xz + 2 * yw), 1)    # :164
    r1 = torch.cat((2 * xy + 2 * zw, 1 - 2 * xx - 2 * zz, 2 * yz - 2 * xw), 1)    # :165
    r2 = torch.cat((2 * xz - 2 * yw, 2 * yz + 2 * xw, 1 - 2 * xx - 2 * yy), 1)    # :166
    return torch.stack((r0, r1, r2), 1)                                          # :168


def euler_torch(e):
    """(B,3) -> (B,3,3); rotation_representation.py:92-113 (note: c2,s2 come from e[:,2] and c3,s3 from e[:,1])."""
    c1, s1 = torch.cos(e[:, 0:1]), torch.sin(e[:, 0:1])                          # :101-102
    c2, s2 = torch.cos(e[:, 2:3]), torch.sin(e[:, 2:3])                          # :103-104
    c3, s3 = torch.cos(e[:, 1:2]), torch.sin(e[:, 1:2])                          # :105-106
    r0 = torch.cat((c2 * c3, -s2, c2 * s3), 1)                                   # :108
    r1 = torch.cat((c1 * s2 * c3 + s1 * s3, c1 * c2, c1 * s2 * s3 - s1 * c3), 1)  # :109
    r2 = torch.cat((s1 * s2 * c3 - c1 * s3, s1 * c2, s1 * s2 * s3 + c1 * c3), 1)  # :110
    return torch.stack((r0, r1, r2), 1)                                          # :112


def ortho6d_torch(p):
    a, b = p[..., 0:3], p[..., 3:6]
    x = a / torch.norm(a, p=2, dim=-1, keepdim=True)
    z = torch.cross(x, b, dim=-1)
    z = z / torch.norm(z, p=2, dim=-1, keepdim=True)
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), -1)


def ortho5d_torch(a):
    """(B,5) -> (B,3,3); rotation_representation.py:118-134 with stereographic_unproject(axis=0) of :69-90."""
    scale = torch.tensor([np.sqrt(2) + 1, np.sqrt(2) + 1, np.sqrt(2)], dtype=a.dtype).view(1, 3)   # :126-127
    v = a[:, 2:5] * scale
    s2 = v.pow(2).sum(1)                                                         # :81
    unproj = 2 * v / (s2 + 1).view(-1, 1)                                        # :83
    u = torch.cat((((s2 - 1) / (s2 + 1)).view(-1, 1), unproj), 1)                # :84-87 with axis = 0
    norm = torch.sqrt(u[:, 1:].pow(2).sum(1))                                    # :130
    u = u / norm.view(-1, 1)                                                     # :131
    return ortho6d_torch(torch.cat((a[:, 0:2], u), 1))                           # :132-133


def expmap_torch(v, eps=1e-4):
    """(B,3) -> (B,3,3); rotation_representation.py:245-275 (so3_exp_map) with hat() of :278-306."""
    nrms = (v * v).sum(1)                                                        # :258
    ang = torch.clamp(nrms, eps).sqrt()                                          # :260
    inv = 1.0 / ang                                                              # :261
    fac1 = inv * ang.sin()                                                       # :262
    fac2 = inv * inv * (1.0 - ang.cos())                                         # :263
    x, y, z = v.unbind(1)
    zero = torch.zeros_like(x)
    k = torch.stack((torch.stack((zero, -z, y), 1), torch.stack((z, zero, -x), 1), torch.stack((-y, x, zero), 1)), 1)   # :299-305
    return fac1[:, None, None] * k + fac2[:, None, None] * torch.bmm(k, k) + torch.eye(3, dtype=v.dtype)[None]   # :267-273


_HEADS_TORCH = {"quat": (4, quat_torch), "euler": (3, euler_torch), "ortho5d": (5, ortho5d_torch), "expmap": (3, expmap_torch)}


def head_np(name, x):
    """Head `name` ('quat' | 'euler' | 'ortho5d' | 'expmap') in float64 -> (B,3,3) numpy."""
    n, fn = _HEADS_TORCH[name]
    return fn(torch.as_tensor(np.asarray(x, np.float64).reshape(-1, n))).numpy()


def head_backward_np(name, x, g):
    """dL/dx for upstream G = dL/dR: autograd through the float64 restatement above."""
    n, fn = _HEADS_TORCH[name]
    xt = torch.as_tensor(np.asarray(x, np.float64).reshape(-1, n)).clone().requires_grad_(True)
    fn(xt).backward(torch.as_tensor(np.asarray(g, np.float64).reshape(-1, 3, 3)))
    return xt.grad.numpy()


# --------------------------------------------------------------------------------------------
# next row f6: the ADD-L1 losses on the output of calculate_T_pred (Iterative/loss.py), op for op in torch
# --------------------------------------------------------------------------------------------
def transform_pts_torch(t, pts):
    """(B,4,4), (B,N,3) -> (B,N,3): R p + t   (Iterative/loss.py:51-70, the 3-D `T` branch)."""
    return (t.unsqueeze(-3)[..., :3, :3] @ pts.unsqueeze(-1) + t.unsqueeze(-3)[..., :3, [-1]]).squeeze(-1)   # :69-70


def add_l1_torch(t_gt, t_pred, points, use_batch_mean=True):
    """Iterative/loss.py:10-26."""
    dists = (transform_pts_torch(t_gt, points) - transform_pts_torch(t_pred, points)).abs().mean(dim=(-1, -2))   # :21-22
    return dists.mean() if use_batch_mean else dists                                                            # :23-26


def add_l1_disentangled_torch(t_pred, t_gt, points, terms=False):
    """Iterative/loss.py:29-48; `terms=True` also returns the (rot, transl, depth) summands."""
    rot, transl, depth = t_gt.clone(), t_gt.clone(), t_gt.clone()                # :35-37
    rot[:, :3, :3] = t_pred[:, :3, :3]                                           # :39
    transl[:, :2, 3] = t_pred[:, :2, 3]                                          # :40
    depth[:, 2, 3] = t_pred[:, 2, 3]                                             # :41
    parts = (add_l1_torch(t_gt, rot, points), add_l1_torch(t_gt, transl, points), add_l1_torch(t_gt, depth, points))   # :43-45
    total = parts[0] + parts[1] + parts[2]                                       # :47
    return (total, parts) if terms else total


def se3_update_torch(model_output, t_init, fx=50 / (36 / 320), fy=50 / (36 / 320)):
    """calculate_T_pred (Iterative/utility.py:90-128) as a differentiable torch graph (any float dtype)."""
    o, t = model_output, t_init
    dr = symmetric_orthogonalization_torch(o[:, :9])                            # :105
    z_new = o[:, 11] * t[:, 2, 3]                                               # :116
    x_new = (o[:, 9] / fx + t[:, 0, 3] / t[:, 2, 3]) * z_new                    # :120
    y_new = (o[:, 10] / fy + t[:, 1, 3] / t[:, 2, 3]) * z_new                   # :121
    top = torch.cat((torch.matmul(dr, t[:, :3, :3]), torch.stack((x_new, y_new, z_new), 1).unsqueeze(-1)), 2)   # :124, combine :63-71
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=o.dtype).expand(o.shape[0], 1, 4)
    return torch.cat((top, bottom), 1)


def add_l1_np(t_gt, t_pred, points, disentangled=False):
    """float64: (loss, dL/dT_pred, per-sample dists or the three disentangled terms)."""
    tg = torch.as_tensor(np.asarray(t_gt, np.float64))
    tp = torch.as_tensor(np.asarray(t_pred, np.float64)).clone().requires_grad_(True)
    pts = torch.as_tensor(np.asarray(points, np.float64))
    if disentangled:
        loss, parts = add_l1_disentangled_torch(tp, tg, pts, terms=True)
        extra = np.array([p.item() for p in parts])
    else:
        loss = add_l1_torch(tg, tp, pts)
        extra = add_l1_torch(tg, tp.detach(), pts, use_batch_mean=False).numpy()
    loss.backward()
    return loss.item(), tp.grad.numpy(), extra


# --------------------------------------------------------------------------------------------
# torch / input dtype: the reference's ATen call sequence ("port" timed as cpu_baseline)
# --------------------------------------------------------------------------------------------
def symmetric_orthogonalization_torch(x):
    m = x.reshape(-1, 3, 3)
    u, _, vh = torch.linalg.svd(m)                                  # == torch.svd + transpose(v)
    det = torch.det(torch.matmul(u, vh)).view(-1, 1, 1)
    vh = torch.cat((vh[:, :2, :], vh[:, -1:, :] * det), 1)
    return torch.matmul(u, vh)


def angle_error_torch(r1, r2):
    off = torch.matmul(r1.transpose(1, 2).double(), r2.double())
    cos = (off.diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
    if torch.any(cos < -1.1) or torch.any(cos > 1.1):
        raise ValueError("angle out of range, input probably not proper rotation matrices")
    return torch.acos(torch.clamp(cos, -1, 1)) * (180 / np.pi)


def loss_frobenius_torch(r_pred, r_true):
    return torch.linalg.matrix_norm(r_true - r_pred, ord="fro").mean()


def kabsch_torch(p, q):
    return symmetric_orthogonalization_torch(torch.bmm(q.transpose(1, 2), p))
