#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Run once, here, with /root/reference mounted:   python tools/gen_golden.py
The reference never travels: only the inputs and the outputs it produced are committed
(SURVEY.md section 8c).  Nothing under tests/, bench.py or the package imports this script.

How the reference code is executed
  * rotation_representation.py imports cleanly (torch, numpy only) -> imported as a module.
  * loss_frobenius (3D-Pose/loss.py:7-11) and the rotation sampler
    (point_cloud/prepare.py:12-49) live in files whose *module-level imports* need packages that
    are not installed (spatialmath, trimesh).  Their function bodies need only torch/numpy, so the
    function definitions are compiled straight from the reference files with `ast` and executed --
    the reference's own code runs, nothing is retyped.
  * Two reference functions hard-code `.cuda()` (rotation_representation.py:218-219,
    point_cloud/prepare.py:23,25).  There is no GPU here, so `Tensor.cuda` is made the identity
    for the duration of this script; arithmetic is unchanged.
"""
import ast
import os
import sys

import numpy as np

sys.dont_write_bytecode = True
REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self          # CPU container: see docstring
torch.set_num_threads(1)                                # deterministic LAPACK path

sys.path.insert(0, REF)
import rotation_representation as rr  # noqa: E402


def functions_from(path, names):
    """Compile the named top-level functions of a reference file without importing the file."""
    with open(path) as fh:
        tree = ast.parse(fh.read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in keep} == set(names), (path, names)
    ns = {"torch": torch, "np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


(loss_frobenius,) = functions_from(os.path.join(REF, "3D-Pose", "loss.py"), ["loss_frobenius"])
normalize_vector, sample_rot = functions_from(
    os.path.join(REF, "point_cloud", "prepare.py"),
    ["normalize_vector", "get_sampled_rotation_matrices_by_axisAngle"])


def svd_parts(x):
    """s and det(u v^T) exactly as the reference computes them (rotation_representation.py:200-202)."""
    m = x.view(-1, 3, 3)
    u, s, v = torch.svd(m)
    det = torch.det(torch.matmul(u, torch.transpose(v, 1, 2)))
    return s, det


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


def g7_ortho6d():
    """Next row f2: the 6D Gram-Schmidt head (rotation_representation.py:21-36), forward + autograd backward."""
    torch.manual_seed(6)
    p = torch.randn(300, 6, requires_grad=True)
    g = torch.randn(300, 3, 3)
    r = rr.compute_rotation_matrix_from_ortho6d(p)
    r.backward(g)
    pd = p.detach().double().requires_grad_(True)
    rd = rr.compute_rotation_matrix_from_ortho6d(pd)
    rd.backward(g.double())
    shaped = torch.randn(2, 5, 6)
    save("g7_ortho6d.npz", p=p.detach(), r=r.detach(), g=g, dp=p.grad, r_f64=rd.detach(), dp_f64=pd.grad,
         p_shaped=shaped, r_shaped=rr.compute_rotation_matrix_from_ortho6d(shaped))


def g8_se3_update():
    """Next row f1: calculate_T_pred (Iterative/utility.py:90-128) forward + autograd backward.

    The reference's helper `combine` (utility.py:63-71) reads two names that are not in its scope
    (`model_output`, `R_new`) and raises NameError as committed; the function compiled from the reference file is
    therefore run with a `combine` that does what that helper plainly intends (ones(B,4,4); [:3,:3]=R; column 3 =
    (tx,ty,tz); [3,:3]=0).  Everything else -- head, focal lengths, translation update, einsum -- is the
    reference's own code."""
    def combine(R, tx, ty, tz, device="cpu"):
        T = torch.ones((R.shape[0], 4, 4))
        T[:, :3, :3] = R
        T[:, 0, 3], T[:, 1, 3], T[:, 2, 3] = tx, ty, tz
        T[:, 3, :3] = 0
        return T
    calc, scene = functions_from(os.path.join(REF, "Iterative", "utility.py"), ["calculate_T_pred", "get_scene_parameters"])
    calc.__globals__.update(symmetric_orthogonalization=rr.symmetric_orthogonalization, combine=combine, get_scene_parameters=scene)
    torch.manual_seed(8)
    b = 200
    out = torch.randn(b, 12)
    out[:, 9:11] *= 20.0                                   # pixel-scale offsets
    out[:, 11] = 1.0 + 0.1 * torch.randn(b)                 # depth ratio near 1
    t_init = torch.zeros(b, 4, 4)
    t_init[:, :3, :3] = rr.symmetric_orthogonalization(torch.randn(b, 9))
    t_init[:, :3, 3] = torch.tensor([0.0, 0.0, 2.5]) + 0.3 * torch.randn(b, 3)
    t_init[:, 3, 3] = 1.0
    g = torch.randn(b, 4, 4)
    o = out.clone().requires_grad_(True)
    tp = calc(o, t_init, "cpu")
    tp.backward(g)
    # (no float64 run: the reference casts R_k with .float(), utility.py:123, so a double input raises)
    save("g8_se3_update.npz", out=out, t_init=t_init, g=g, t_pred=tp.detach(), dout=o.grad, fx=scene()[0], fy=scene()[1])


def g9_sampler():
    """Next row f4: the rotation sampler (point_cloud/prepare.py:21-49) with its random draws recorded.
    The sampler takes theta from numpy's global RNG and the axis from torch's; both are re-seeded so the draws
    can be replayed and stored next to the matrices the reference made from them."""
    b = 500
    np.random.seed(11)
    theta = np.random.uniform(-1, 1, b) * np.pi                      # what prepare.py:23 will draw
    torch.manual_seed(11)
    axis = torch.randn(b, 3)                                         # what prepare.py:25 will draw
    np.random.seed(11)
    torch.manual_seed(11)
    r = sample_rot(b)
    save("g9_sampler.npz", theta=theta.astype(np.float32), axis=axis, r=r)


def g11_add_l1():
    """Next row f6: compute_ADD_L1_loss / compute_disentangled_ADD_L1_loss (Iterative/loss.py:10-70) and their
    autograd w.r.t. the predicted pose, float32 as the training loop runs them and float64."""
    add_l1, add_l1_dis, _ = functions_from(os.path.join(REF, "Iterative", "loss.py"),
                                           ["compute_ADD_L1_loss", "compute_disentangled_ADD_L1_loss", "transform_pts"])
    add_l1.__globals__["transform_pts"] = _
    torch.manual_seed(21)
    b, n = 24, 200                                                        # N not a multiple of 64: ragged tail
    def poses(noise):
        t = torch.eye(4).repeat(b, 1, 1)
        t[:, :3, :3] = rr.symmetric_orthogonalization(torch.randn(b, 9))
        t[:, :3, 3] = torch.randn(b, 3) * 0.3 + torch.tensor([0.0, 0.0, 2.0])
        return t
    t_gt = poses(0)
    t_pred = t_gt.clone()
    t_pred[:, :3, :3] = torch.matmul(rr.symmetric_orthogonalization(torch.eye(3).reshape(1, 9) + 0.2 * torch.randn(b, 9)), t_gt[:, :3, :3])
    t_pred[:, :3, 3] += 0.1 * torch.randn(b, 3)
    t_pred[0] = t_gt[0]                                                   # an exact hit: every |.| sits at its kink
    t_pred[1, :3, 3] = t_gt[1, :3, 3]                                     # rotation error only
    t_pred[2, :3, :3] = t_gt[2, :3, :3]                                   # translation error only
    pts = torch.randn(b, n, 3) * 0.2
    out = {"t_gt": t_gt, "t_pred": t_pred, "points": pts}
    for tag, dt in (("", torch.float32), ("_f64", torch.float64)):
        tg, p = t_gt.to(dt), pts.to(dt)
        tp = t_pred.to(dt).clone().requires_grad_(True)
        loss = add_l1(tg, tp, p)
        loss.backward()
        out.update({"add" + tag: loss.detach(), "add_grad" + tag: tp.grad, "add_dists" + tag: add_l1(tg, tp.detach(), p, use_batch_mean=False)})
        tp = t_pred.to(dt).clone().requires_grad_(True)
        loss = add_l1_dis(tp, tg, p)
        loss.backward()
        out.update({"dis" + tag: loss.detach(), "dis_grad" + tag: tp.grad})
    save("g11_add_l1.npz", **out)


def g12_clouds():
    """Row a7: pc_normalize (point_cloud/prepare.py:51-56, the reference function, numpy float64) and the training
    loop's pairing rule (point_cloud/main.py:173-183), whose four statements are replayed here verbatim on CPU."""
    (pc_normalize,) = functions_from(os.path.join(REF, "point_cloud", "prepare.py"), ["pc_normalize"])
    rng = np.random.RandomState(12)
    clouds = (rng.rand(6, 200, 3) - 0.5) * np.array([1.0, 2.0, 0.5]) + np.array([0.3, -1.0, 2.0])
    norm, cent, scale = zip(*(pc_normalize(c) for c in clouds))
    torch.manual_seed(12)
    np.random.seed(12)
    batch, point_num = 6, 200
    pc1 = torch.tensor(np.stack(norm)).float()
    gt_rmat = sample_rot(batch)
    gt_rmats = gt_rmat.contiguous().view(batch, 1, 3, 3).expand(batch, point_num, 3, 3).contiguous().view(-1, 3, 3)   # :176-177
    pc2 = torch.bmm(gt_rmats, pc1.view(-1, 3, 1))                                                                      # :180
    pc_out = pc2.view(batch, point_num, 3)                                                                             # :181
    gg = pc_out.transpose(1, 2)                                                                                        # :183
    save("g12_clouds.npz", clouds=clouds, norm=np.stack(norm), centroid=np.stack(cent), scale=np.array(scale),
         pc1=pc1, gt_rmat=gt_rmat, pc_out=pc_out, gg=gg.contiguous())


def g10_heads():
    """Next row f5: the quaternion / Euler / 5D / exp-map heads (rotation_representation.py:39-171, 245-321) and
    their autograd, float32 as the reference runs them; float64 too where the reference's code keeps float64."""
    heads = {"quat": (4, rr.compute_rotation_matrix_from_quaternion), "euler": (3, rr.compute_rotation_matrix_from_euler),
             "ortho5d": (5, rr.compute_rotation_matrix_from_ortho5d), "expmap": (3, rr.vec_3d_to_SO3)}
    out = {}
    for seed, (name, (n, fn)) in enumerate(sorted(heads.items())):
        torch.manual_seed(100 + seed)
        b = 192
        x = torch.randn(b, n)
        x[:32] *= 4.0                                       # large angles / magnitudes
        x[32:64] *= 0.05                                    # small ones
        if name == "expmap":
            x[64:80] *= 0.004                               # |v|^2 < 1e-4: the clamped branch
            x[80:96] = x[80:96] / x[80:96].norm(dim=1, keepdim=True) * torch.linspace(0.9, 1.1, 16).view(-1, 1)   # around theta = 1
        g = torch.randn(b, 3, 3)
        xf = x.clone().requires_grad_(True)
        r = fn(xf)
        r.backward(g)
        out.update({name + "_x": x, name + "_r": r.detach(), name + "_g": g, name + "_dx": xf.grad})
        if name != "ortho5d":                               # its float64 run goes through float32 zeros (:82)
            xd = x.double().requires_grad_(True)
            rd = fn(xd)
            rd.backward(g.double())
            assert rd.dtype == torch.float64
            out.update({name + "_r_f64": rd.detach(), name + "_dx_f64": xd.grad})
    save("g10_heads.npz", **out)


def g13_dtype_fidelity():
    """Output dtypes and float64 values of the loss and the two metrics (3D-Pose/loss.py:7-11,
    rotation_representation.py:209-242) for float32 and float64 arguments."""
    torch.manual_seed(13)
    r1 = rr.symmetric_orthogonalization(torch.randn(300, 9).double())
    r2 = rr.symmetric_orthogonalization(torch.randn(300, 9).double())
    out = {"r1": r1, "r2": r2}
    for tag, cast in (("f32", torch.float32), ("f64", torch.float64)):
        a, b = r1.to(cast), r2.to(cast)
        ar = a.clone().requires_grad_(True)
        loss = loss_frobenius(ar, b)
        loss.backward()
        geo = rr.compute_geodesic_distance_from_two_matrices(a, b)
        ang = rr.angle_error(a, b)
        out.update({"loss_" + tag: loss.detach(), "dloss_" + tag: ar.grad, "geo_" + tag: geo, "ang_" + tag: ang,
                    "dtypes_" + tag: np.array([str(loss.dtype), str(ar.grad.dtype), str(geo.dtype), str(ang.dtype)])})
    save("g13_dtype_fidelity.npz", **out)


def g14_geodesic_reduction():
    """geodesic(R1, R2, reduction) of point_cloud/main.py:61-73 (the eps-clamped geodesic with "none" / "mean" / "sum"), run from
    the reference file on G3's pairs (0 and 180 degrees included: the clamp is what differs from the other geodesic) and on 2000
    Haar-like pairs."""
    (geodesic,) = functions_from(os.path.join(REF, "point_cloud", "main.py"), ["geodesic"])
    g3 = np.load(os.path.join(OUT, "g3_angles.npz"))
    r1, r2 = torch.from_numpy(g3["r1"]), torch.from_numpy(g3["r2"])
    torch.manual_seed(14)
    a = rr.symmetric_orthogonalization(torch.randn(2000, 9))
    b = rr.symmetric_orthogonalization(torch.randn(2000, 9))
    out = {"a": a, "b": b}
    for tag, (p, q) in (("g3", (r1, r2)), ("haar", (a, b))):
        out.update({tag + "_none": geodesic(p, q, "none"), tag + "_mean": geodesic(p, q, "mean"), tag + "_sum": geodesic(p, q, "sum")})
    assert geodesic(a, b, "median") is None                 # any other string falls through the if-chain
    out["dtypes"] = np.array([str(out["haar_none"].dtype), str(out["haar_mean"].dtype), str(out["haar_sum"].dtype)])
    save("g14_geodesic_reduction.npz", **out)


def g15_metric_gradients():
    """Autograd through the three metric spellings, as the reference's training loops can use them (`lossfunc` hooks:
    point_cloud/main.py:194-197, UPNA/main.py:56-59): geodesic(R1, R2, reduction) (point_cloud/main.py:61-73, whose eps exists for
    this gradient), compute_geodesic_distance_from_two_matrices and angle_error (rotation_representation.py:209-242).  Pairs: G3's
    (their first 64: 0 and 180 degrees, 1e-4 and 3e-3 rad included; outside the clamp the reference's masked fill gives exactly 0) and the first 128 of G14's Haar-like pairs; float32 and float64; gradients with respect to BOTH arguments; a per-row
    upstream gradient for the unreduced forms."""
    (geodesic,) = functions_from(os.path.join(REF, "point_cloud", "main.py"), ["geodesic"])
    g3 = np.load(os.path.join(OUT, "g3_angles.npz"))
    g14 = np.load(os.path.join(OUT, "g14_geodesic_reduction.npz"))
    sets = {"g3": (torch.from_numpy(g3["r1"][:64]), torch.from_numpy(g3["r2"][:64])),       # rows 0-4: 0, 180, 179.96, 0.044, 0.155 degrees
            "haar": (torch.from_numpy(g14["a"][:128]), torch.from_numpy(g14["b"][:128]))}
    out = {}
    for tag, (p, q) in sets.items():
        torch.manual_seed(15)
        w = torch.randn(p.shape[0])                        # upstream gradient of the per-row forms
        out[tag + "_r1"], out[tag + "_r2"], out[tag + "_w"] = p, q, w
        for dt_tag, dt in (("f32", torch.float32), ("f64", torch.float64)):
            def run(fn, upstream):
                a = p.to(dt).clone().requires_grad_(True)
                b = q.to(dt).clone().requires_grad_(True)
                y = fn(a, b)
                if upstream is None:
                    y.backward()
                else:
                    y.backward(upstream.to(y.dtype))
                return y.detach(), a.grad, b.grad
            cases = {
                "geo_mean": (lambda a, b: geodesic(a, b, "mean"), None),
                "geo_sum": (lambda a, b: geodesic(a, b, "sum"), None),
                "geo_none": (lambda a, b: geodesic(a, b, "none"), w),
                "cgd": (rr.compute_geodesic_distance_from_two_matrices, w),
                "ang": (rr.angle_error, w),
                "ang_mean": (lambda a, b: rr.angle_error(a, b).mean(), None),
            }
            for name, (fn, up) in cases.items():
                y, da, db = run(fn, up)
                key = "%s_%s_%s" % (tag, name, dt_tag)
                out[key + "_y"], out[key + "_d1"], out[key + "_d2"] = y, da, db
    out["dtypes"] = np.array([str(out["haar_ang_f32_y"].dtype), str(out["haar_ang_f32_d1"].dtype), str(out["haar_geo_mean_f32_d1"].dtype),
                              str(out["haar_cgd_f64_d1"].dtype)])
    save("g15_metric_gradients.npz", **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "g15":
        return g15_metric_gradients()
    if len(sys.argv) > 1 and sys.argv[1] == "g14":
        return g14_geodesic_reduction()
    if len(sys.argv) > 1 and sys.argv[1] == "g13":
        return g13_dtype_fidelity()
    if len(sys.argv) > 1 and sys.argv[1] == "g9":
        return g9_sampler()
    if len(sys.argv) > 1 and sys.argv[1] == "g7":        # regenerate one fixture without touching the others
        return g7_ortho6d()
    if len(sys.argv) > 1 and sys.argv[1] == "g8":
        return g8_se3_update()
    if len(sys.argv) > 1 and sys.argv[1] == "g10":
        return g10_heads()
    if len(sys.argv) > 1 and sys.argv[1] == "g11":
        return g11_add_l1()
    if len(sys.argv) > 1 and sys.argv[1] == "g12":
        return g12_clouds()
    g7_ortho6d()
    g8_se3_update()
    g9_sampler()
    g10_heads()
    g11_add_l1()
    g12_clouds()
    g13_dtype_fidelity()
    # ---- G1: config #1, 256 Gaussian rows ------------------------------------------------------
    torch.manual_seed(0)
    x = torch.randn(256, 9)
    r = rr.symmetric_orthogonalization(x)
    s, det = svd_parts(x)
    r64 = rr.symmetric_orthogonalization(x.double())
    save("g1_gaussian256.npz", x=x, r=r, s=s, det=det, r_f64=r64)

    # ---- G2: adversarial inputs ------------------------------------------------------------------
    torch.manual_seed(2)
    q = rr.symmetric_orthogonalization(torch.randn(4, 9))
    cases, names = [], []

    def add(name, m):
        names.append(name)
        cases.append(torch.as_tensor(m, dtype=torch.float32).reshape(3, 3))
    add("zero", torch.zeros(3, 3))
    add("identity", torch.eye(3))
    add("reflection_z", torch.diag(torch.tensor([1., 1., -1.])))
    add("neg_identity", -torch.eye(3))
    add("rank1_e1", torch.diag(torch.tensor([2., 0., 0.])))
    add("rank1_ones", torch.ones(3, 3))
    add("rank2_diag", torch.diag(torch.tensor([1., 1., 0.])))
    add("rank2_rot", q[0] @ torch.diag(torch.tensor([3., 0.5, 0.])) @ q[1].T)
    add("rotation", q[2])
    add("rotation_scaled_1e-20", q[2] * 1e-20)
    add("rotation_scaled_1e+15", q[2] * 1e15)
    add("improper_rotation", q[3] @ torch.diag(torch.tensor([1., 1., -1.])))
    add("near_equal_sv", q[0] @ torch.diag(torch.tensor([1.0, 1.0 - 1e-6, 1.0 - 2e-6])) @ q[1].T)
    add("near_equal_sv_flip", q[0] @ torch.diag(torch.tensor([1.0, 0.7, -0.3])) @ q[1].T)
    add("flip_close_s2_s3", q[0] @ torch.diag(torch.tensor([1.0, 0.5, -0.499])) @ q[1].T)
    add("tiny_s3_pos", q[2] @ torch.diag(torch.tensor([1.0, 0.5, 1e-6])) @ q[3].T)
    add("tiny_s3_neg", q[2] @ torch.diag(torch.tensor([1.0, 0.5, -1e-6])) @ q[3].T)
    add("graded", q[1] @ torch.diag(torch.tensor([1e3, 1.0, 1e-3])) @ q[0].T)
    add("upper_triangular", torch.tensor([[1., 2., 3.], [0., 4., 5.], [0., 0., 6.]]))
    add("permutation_odd", torch.tensor([[0., 1., 0.], [1., 0., 0.], [0., 0., 1.]]))
    add("permutation_even", torch.tensor([[0., 1., 0.], [0., 0., 1.], [1., 0., 0.]]))
    xa = torch.stack(cases).reshape(-1, 9)
    ra = rr.symmetric_orthogonalization(xa)
    sa, deta = svd_parts(xa)
    save("g2_adversarial.npz", x=xa, r=ra, s=sa, det=deta, names=np.array(names),
         r_f64=rr.symmetric_orthogonalization(xa.double()))
    # view semantics: any shape with numel % 9 == 0 (rotation_representation.py:199)
    torch.manual_seed(3)
    xs = torch.randn(2, 5, 9)
    save("g2_shape_2x5x9.npz", x=xs, r=rr.symmetric_orthogonalization(xs))

    # ---- G3: angle_error / geodesic ----------------------------------------------------------------
    torch.manual_seed(4)
    r1 = rr.symmetric_orthogonalization(torch.randn(256, 9))
    r2 = rr.symmetric_orthogonalization(torch.randn(256, 9))
    r2[0] = r1[0]                                                   # 0 degrees
    r2[1] = r1[1] @ torch.diag(torch.tensor([1., -1., -1.]))        # 180 degrees
    r2[2] = r1[2] @ torch.diag(torch.tensor([-1., -1., 1.]))        # 180 degrees
    small = rr.so3_exp_map(torch.tensor([[1e-4, 0., 0.], [0., 3e-3, 0.]])) if hasattr(rr, "so3_exp_map") else None
    if small is not None and small.shape == (2, 3, 3):
        r2[3] = r1[3] @ small[0].float()
        r2[4] = r1[4] @ small[1].float()
    deg = rr.angle_error(r1, r2)
    rad = rr.compute_geodesic_distance_from_two_matrices(r1, r2)
    bad1 = torch.eye(3).repeat(3, 1, 1)
    bad2 = torch.eye(3).repeat(3, 1, 1)
    bad2[1] = 1.5 * torch.eye(3)          # trace 4.5 -> cos 1.75 > 1.1 : must raise
    raised = False
    try:
        rr.angle_error(bad1, bad2)
    except ValueError as exc:
        raised = True
        msg = str(exc)
    assert raised
    nearly = torch.eye(3).repeat(2, 1, 1)
    nearly2 = nearly.clone()
    nearly2[1] = 1.05 * torch.eye(3)      # cos = 1.075: inside the tolerance band, clamped, no raise
    deg_nearly = rr.angle_error(nearly, nearly2)
    save("g3_angles.npz", r1=r1, r2=r2, deg=deg, rad=rad, bad1=bad1, bad2=bad2, raise_msg=np.array(msg),
         nearly1=nearly, nearly2=nearly2, deg_nearly=deg_nearly)

    # ---- G4: config #4, 512 rows bf16-rounded, Frobenius loss forward + backward ------------------
    torch.manual_seed(0)
    x4 = torch.randn(512, 9).bfloat16()
    torch.manual_seed(1)
    rt = rr.symmetric_orthogonalization(torch.randn(512, 9))
    xf = x4.float().requires_grad_(True)
    out = rr.symmetric_orthogonalization(xf)
    loss = loss_frobenius(rt, out)                      # call order of 3D-Pose/main.py:85 (R, out)
    loss.backward()
    # generic upstream gradient through the head alone (K2)
    torch.manual_seed(5)
    g = torch.randn(512, 3, 3)
    xg = x4.float().requires_grad_(True)
    rr.symmetric_orthogonalization(xg).backward(g)
    # float64 versions of both gradients (the reference code, double input)
    xd = x4.double().requires_grad_(True)
    lossd = loss_frobenius(rt.double(), rr.symmetric_orthogonalization(xd))
    lossd.backward()
    xgd = x4.double().requires_grad_(True)
    rr.symmetric_orthogonalization(xgd).backward(g.double())
    save("g4_frobenius512.npz", x_bf16_bits=x4.view(torch.int16), r_true=rt, r=out.detach(), loss=loss.detach(),
         dx=xf.grad, g=g, dx_g=xg.grad, loss_f64=lossd.detach(), dx_f64=xd.grad, dx_g_f64=xgd.grad)

    # ---- G5: Kabsch pairs (config #3 contract), 6 clouds x 1024 + 24 clouds x 64 --------------------
    for tag, (b, n) in {"6x1024": (6, 1024), "24x64": (24, 64)}.items():
        torch.manual_seed(7)
        np.random.seed(7)
        p = torch.rand(b, n, 3) - 0.5
        r_gt = sample_rot(b)                                         # point_cloud/prepare.py:21-49
        qpts = torch.bmm(r_gt, p.transpose(1, 2)).transpose(1, 2).contiguous()   # point_cloud/main.py:176-181
        qn = qpts + 0.01 * torch.randn(b, n, 3)
        h = torch.bmm(qn.transpose(1, 2), p)
        save("g5_kabsch_%s.npz" % tag, p=p, q=qn, r_gt=r_gt, h=h, r=rr.symmetric_orthogonalization(h),
             r_f64=rr.symmetric_orthogonalization(torch.bmm(qn.double().transpose(1, 2), p.double())))

    # ---- G6: scalar statistics at config #2's full size (inputs regenerated from the seeds) -------
    torch.set_num_threads(8)
    torch.manual_seed(0)
    xb = torch.randn(1_000_000, 9)
    torch.manual_seed(1)
    tb = rr.symmetric_orthogonalization(torch.randn(1_000_000, 9))
    rb = rr.symmetric_orthogonalization(xb)
    sb, detb = svd_parts(xb)
    ang = rr.angle_error(rb, tb)
    rb64 = rr.symmetric_orthogonalization(xb.double())
    ang64 = rr.angle_error(rb64, tb.double())
    orth = torch.linalg.matrix_norm(rb.transpose(1, 2) @ rb - torch.eye(3), ord="fro")
    save("g6_stats_1m.npz", n=1_000_000, seed_x=0, seed_t=1,
         mean_angle_deg=ang.mean(), mean_angle_deg_f64=ang64.mean(),
         flip_count=(detb < 0).sum(), max_orth_err=orth.max(),
         x_head=xb[:64], r_head=rb[:64], t_head=tb[:64], x_checksum=xb.double().sum(), t_checksum=tb.double().sum(),
         flip_bits=np.packbits((detb < 0).numpy()))


if __name__ == "__main__":
    main()
