"""The oracle is pinned here: every flavour (numpy f64, torch f32 port, C f64) against the golden
vectors that tools/gen_golden.py produced by RUNNING the reference (tests/golden/*.npz).
CPU only; no GPU, no HIP library."""
import numpy as np
import pytest
import torch

from conftest import METRIC_GRAD_CASES, load_golden, metric_grad_check, orth_err, well_conditioned
from oracle import so3_oracle as so

# names in g2_adversarial whose projection is unique (rank >= 2 and no s2==s3 flip degeneracy)
# (improper_rotation, reflection_z, permutation_odd, neg_identity have s2 == s3 WITH a flip: any rotation of the
#  (u2,u3)/(v2,v3) planes is a valid SVD and gives a different R -- implementation-defined, checked for
#  orthogonality/det only, as SURVEY.md section 8c prescribes.)
UNIQUE = {"identity", "rank2_diag", "rank2_rot", "rotation", "rotation_scaled_1e-20",
          "rotation_scaled_1e+15", "near_equal_sv", "near_equal_sv_flip", "flip_close_s2_s3",
          "tiny_s3_pos", "tiny_s3_neg", "graded", "upper_triangular", "permutation_even"}


def test_g1_numpy_oracle_matches_reference_f64():
    g = load_golden("g1_gaussian256.npz")
    r = so.symmetric_orthogonalization_np(g["x"])
    assert np.abs(r - g["r_f64"]).max() < 1e-12            # same LAPACK, same formula
    # the reference's own float32 result sits within float32 conditioning of the float64 answer
    ok = well_conditioned(g["s"], g["det"])
    assert np.abs(r[ok] - g["r"][ok]).max() < 2e-5
    assert ok.sum() > 240


def test_g1_flip_is_det_sign_and_count():
    g = load_golden("g1_gaussian256.npz")
    assert ((g["det"] < 0) == so.flip_flag_np(g["x"])).all()
    assert int((g["det"] < 0).sum()) == 132                 # SURVEY.md section 8c [probed]
    assert orth_err(g["r"]).max() < 1e-5


def test_g1_torch_port_is_bitwise_the_reference():
    g = load_golden("g1_gaussian256.npz")
    torch.set_num_threads(1)
    r = so.symmetric_orthogonalization_torch(torch.from_numpy(g["x"]))
    assert np.array_equal(r.numpy(), g["r"])


def test_g1_c_oracle(c_oracle):
    g = load_golden("g1_gaussian256.npz")
    r, flip = c_oracle.project(g["x"].astype(np.float64), want_flip=True)
    assert np.abs(r - g["r_f64"]).max() < 1e-11
    assert (flip == (g["det"] < 0)).all()


def test_g2_adversarial(c_oracle):
    g = load_golden("g2_adversarial.npz")
    names = [str(n) for n in g["names"]]
    r_np = so.symmetric_orthogonalization_np(g["x"])
    r_c = c_oracle.project(g["x"].astype(np.float64))
    for r in (r_np, r_c, g["r"]):
        assert orth_err(r).max() < 1e-5
        assert np.abs(np.linalg.det(np.asarray(r, np.float64)) - 1).max() < 1e-5
    for i, n in enumerate(names):
        if n in UNIQUE:
            # the reference's OWN float32 result is only as good as float32 LAPACK on that conditioning
            tol = {"flip_close_s2_s3": 5e-3, "near_equal_sv": 5e-3, "graded": 5e-4, "tiny_s3_pos": 5e-4,
                   "tiny_s3_neg": 5e-4, "rank2_rot": 5e-4}.get(n, 2e-5)
            assert np.abs(r_np[i] - g["r_f64"][i]).max() < 1e-9, n
            assert np.abs(r_c[i] - g["r_f64"][i]).max() < 1e-7, n
            assert np.abs(g["r"][i] - g["r_f64"][i]).max() < tol, n
    # documented reference behaviours
    assert np.allclose(g["r"][names.index("zero")], np.eye(3))
    assert np.allclose(g["r"][names.index("reflection_z")], np.eye(3))
    assert np.allclose(r_c[names.index("zero")], np.eye(3))
    assert np.allclose(r_c[names.index("reflection_z")], np.eye(3))


def test_g2_view_semantics():
    g = load_golden("g2_shape_2x5x9.npz")
    assert g["r"].shape == (10, 3, 3)
    assert np.abs(so.symmetric_orthogonalization_np(g["x"]) - g["r"]).max() < 2e-5


def test_g3_angle_error(c_oracle):
    g = load_golden("g3_angles.npz")
    deg = so.angle_error_np(g["r1"], g["r2"])
    assert np.abs(deg - g["deg"]).max() < 1e-9
    deg_c, bad = c_oracle.angle_error(g["r1"], g["r2"])
    assert not bad and np.abs(deg_c - g["deg"]).max() < 1e-9
    assert np.abs(so.angle_error_torch(torch.from_numpy(g["r1"]), torch.from_numpy(g["r2"])).numpy() - g["deg"]).max() == 0
    assert g["deg"][1] > 179.9 and g["deg"][2] > 179.9 and g["deg"][0] < 0.2
    with pytest.raises(ValueError, match="angle out of range"):
        so.angle_error_np(g["bad1"], g["bad2"])
    assert str(g["raise_msg"]) == "angle out of range, input probably not proper rotation matrices"
    assert c_oracle.angle_error(g["bad1"], g["bad2"])[1]
    assert np.abs(so.angle_error_np(g["nearly1"], g["nearly2"]) - g["deg_nearly"]).max() < 1e-12


def test_g3_geodesic_radians():
    g = load_golden("g3_angles.npz")
    rad = so.geodesic_np(g["r1"], g["r2"])
    assert rad.dtype == np.float32
    # acos is ill-conditioned at 0 and pi: compare cosines (float32 trace arithmetic)
    assert np.abs(np.cos(rad.astype(np.float64)) - np.cos(g["rad"].astype(np.float64))).max() < 1e-6
    assert np.abs(np.degrees(g["rad"].astype(np.float64)) - g["deg"])[5:].max() < 0.05


def test_g14_geodesic_with_reduction():
    """geodesic(R1, R2, reduction) (point_cloud/main.py:61-73), golden from the reference function: the eps-clamped angles (no exact
    0 or pi: acos(1 - 1.19e-7) = 4.9e-4 rad at the ends), their float32 mean and sum, and None for an unknown reduction."""
    g = load_golden("g14_geodesic_reduction.npz")
    g3 = load_golden("g3_angles.npz")
    for tag, (a, b) in (("g3", (g3["r1"], g3["r2"])), ("haar", (g["a"], g["b"]))):
        rad = so.geodesic_eps_np(a, b, "none")
        assert rad.dtype == np.float32
        assert np.abs(np.cos(rad.astype(np.float64)) - np.cos(g[tag + "_none"].astype(np.float64))).max() < 1e-6
        assert rad.min() >= 4.8e-4 and rad.max() <= np.pi - 4.8e-4        # the clamp keeps every angle off 0 and pi
        assert abs(float(so.geodesic_eps_np(a, b, "mean")) - float(g[tag + "_mean"])) < 2e-6 * float(g[tag + "_mean"]) + 1e-5
        assert abs(float(so.geodesic_eps_np(a, b, "sum")) - float(g[tag + "_sum"])) < 2e-6 * float(g[tag + "_sum"]) + 3e-3
    assert so.geodesic_eps_np(g["a"], g["b"], "median") is None
    assert list(g["dtypes"]) == ["torch.float32"] * 3


def test_g15_metric_gradients():
    """The closed form of the metrics' gradient (oracle.metric_backward_np: what K4b computes) against autograd through the
    reference's three spellings, both arguments, float32 and float64 graphs, 0 / 180 degrees and 1e-4 rad included."""
    g = load_golden("g15_metric_gradients.npz")
    assert list(g["dtypes"]) == ["torch.float64", "torch.float32", "torch.float32", "torch.float64"]
    ambiguous = 0
    for tag in ("g3", "haar"):
        n = g[tag + "_r1"].shape[0]
        for name, (eps, unit, div, up) in METRIC_GRAD_CASES.items():
            for dt, np_dt in (("f32", np.float32), ("f64", np.float64)):
                key = "%s_%s_%s" % (tag, name, dt)
                a, b = g[tag + "_r1"].astype(np_dt), g[tag + "_r2"].astype(np_dt)
                f64_graph = dt == "f64" or name.startswith("ang")            # angle_error casts to float64 itself
                # geodesic's clamp bounds are float32 scalars on a float32 tensor
                d1, d2, c = so.metric_backward_np(a, b, g[tag + "_w"].astype(np_dt) if up == "w" else 1.0, eps=eps, unit=unit,
                                                  divisor=float(n) if div == "B" else 1.0, clamp_dtype=np.float64 if f64_graph else np.float32)
                assert g[key + "_d1"].dtype == np_dt and g[key + "_d1"].shape == (n, 3, 3)
                ambiguous += metric_grad_check(d1, d2, g[key + "_d1"], g[key + "_d2"], c, eps, 1e-15 if f64_graph else 4e-7,
                                               1e-12 if dt == "f64" else 3e-7, key)
    assert ambiguous < 40          # of 2304 row checks: the 0 / 180 degree pairs of G3


def test_g4_loss_and_gradients(c_oracle):
    g = load_golden("g4_frobenius512.npz")
    x = torch.from_numpy(g["x_bf16_bits"]).view(torch.bfloat16).float().numpy()
    loss, dx, r = so.frobenius_fwd_bwd_np(x, g["r_true"])
    assert abs(loss - float(g["loss_f64"])) < 1e-12
    assert abs(loss - float(g["loss"])) < 1e-6
    scale = np.abs(g["dx_f64"]).max()
    assert np.abs(dx.reshape(512, 9) - g["dx_f64"]).max() < 1e-9 * max(scale, 1)
    # closed-form K2 backward vs autograd through the reference (float64), generic upstream gradient
    dxg = so.projection_backward_np(x, g["g"]).reshape(512, 9)
    assert np.abs(dxg - g["dx_g_f64"]).max() < 1e-8 * np.abs(g["dx_g_f64"]).max()
    # C oracle, float32 in/out
    dxc = c_oracle.project_bwd(x, g["g"]).reshape(512, 9)
    rel = np.abs(dxc - g["dx_g_f64"]).max(1) / (1e-3 + np.abs(g["dx_g_f64"]).max(1))
    assert np.quantile(rel, 0.99) < 1e-5
    # the reference's float32 autograd is itself only conditioning-accurate
    rel32 = np.abs(g["dx_g"] - g["dx_g_f64"]).max(1) / (1e-3 + np.abs(g["dx_g_f64"]).max(1))
    assert np.median(rel32) < 1e-5


@pytest.mark.parametrize("tag", ["6x1024", "24x64"])
def test_g5_kabsch(c_oracle, tag):
    g = load_golden("g5_kabsch_%s.npz" % tag)
    h = so.cross_covariance_np(g["p"], g["q"])
    assert np.abs(h - g["h"]).max() < 2e-4 * np.abs(g["h"]).max()        # reference bmm is float32
    r = so.kabsch_np(g["p"], g["q"])
    assert np.abs(r - g["r_f64"]).max() < 1e-12
    assert np.abs(r - g["r"]).max() < 5e-6
    rc, hc = c_oracle.kabsch(g["p"], g["q"], want_h=True)
    assert np.abs(hc - h).max() < 1e-10
    assert np.abs(rc - g["r_f64"]).max() < 2e-7
    # noise is 1%: the solve recovers the generating rotation to a fraction of a degree
    assert so.angle_error_np(r, g["r_gt"]).max() < 1.0


def test_g6_full_size_statistics(c_oracle):
    """Config #2's 1M rows, regenerated from the seeds; pins the oracle at BASELINE.json's size."""
    g = load_golden("g6_stats_1m.npz")
    n = int(g["n"])
    torch.manual_seed(int(g["seed_x"]))
    x = torch.randn(n, 9)
    torch.manual_seed(int(g["seed_t"]))
    t_in = torch.randn(n, 9)
    assert abs(float(x.double().sum()) - float(g["x_checksum"])) < 1e-7  # same RNG stream as the generator
    assert np.array_equal(x[:64].numpy(), g["x_head"])
    r, flip = c_oracle.project(x.numpy(), want_flip=True)
    t = c_oracle.project(t_in.numpy())
    assert np.abs(t[:64] - g["t_head"]).max() < 2e-5
    assert int(flip.sum()) == int(g["flip_count"])
    assert np.array_equal(np.packbits(flip), g["flip_bits"])             # every one of the 1M flags
    deg, bad = c_oracle.angle_error(r, t)
    assert not bad
    assert abs(deg.mean() - float(g["mean_angle_deg_f64"])) < 2e-5      # targets differ (f32 ref vs f64 oracle) at 1e-6 level
    assert abs(deg.mean() - float(g["mean_angle_deg"])) < 1e-4
    assert float(g["max_orth_err"]) < 1e-5 and orth_err(r[:100000]).max() < 1e-6


def test_rotation_sampler_matches_reference_formula():
    g = load_golden("g5_kabsch_24x64.npz")
    r = so.sample_rotations_axis_angle_np(np.random.default_rng(0), 1000)
    assert orth_err(r).max() < 1e-12 and np.abs(np.linalg.det(r) - 1).max() < 1e-12
    assert orth_err(g["r_gt"]).max() < 1e-5


def test_g7_ortho6d_oracle():
    """Next row f2: the 6D head restatement and its closed-form backward against the reference + autograd."""
    g = load_golden("g7_ortho6d.npz")
    r = so.ortho6d_np(g["p"])
    assert np.abs(r - g["r_f64"]).max() < 1e-12
    assert np.abs(r - g["r"]).max() < 2e-5
    dp = so.ortho6d_backward_np(g["p"], g["g"])
    assert np.abs(dp - g["dp_f64"]).max() < 1e-9 * max(1.0, np.abs(g["dp_f64"]).max())
    assert so.ortho6d_np(g["p_shaped"]).shape == (2, 5, 3, 3)
    assert np.abs(so.ortho6d_np(g["p_shaped"]) - g["r_shaped"]).max() < 2e-5
    assert orth_err(g["r"]).max() < 1e-5


@pytest.mark.parametrize("name", ["quat", "euler", "ortho5d", "expmap"])
def test_g10_heads_oracle(name):
    """Next row f5: each remaining head restated (float64) against the reference function's output and autograd."""
    g = load_golden("g10_heads.npz")
    x, gr = g[name + "_x"], g[name + "_g"]
    r = so.head_np(name, x)
    dx = so.head_backward_np(name, x, gr)
    tol = 1e-8 if name == "expmap" else 1e-12          # below the clamp (|v|^2 < 1e-4) the map is only nearly orthogonal
    assert orth_err(r).max() < tol and np.abs(np.linalg.det(r) - 1).max() < tol
    if name + "_r_f64" in g:
        assert np.abs(r - g[name + "_r_f64"]).max() < 1e-12
        ref = g[name + "_dx_f64"]
        assert np.abs(dx - ref).max() < 1e-9 * max(1.0, np.abs(ref).max())
    # the reference's own float32 run: forward to float32 round-off; the gradient to a few 1e-4 of its scale
    # (the exp-map factors (1 - cos t)/t^2 and d/dt cancel in float32 for small t; 1/|q| amplifies for small q)
    assert np.abs(r - g[name + "_r"]).max() < 5e-6
    ref32 = g[name + "_dx"].astype(np.float64)
    scale = np.maximum(np.abs(ref32).max(axis=1, keepdims=True), 1.0)
    assert (np.abs(dx - ref32) / scale).max() < 2e-3


def test_g11_add_l1_oracle():
    """Next row f6: the ADD-L1 losses restated (float64) against the reference functions and their autograd."""
    g = load_golden("g11_add_l1.npz")
    loss, grad, dists = so.add_l1_np(g["t_gt"], g["t_pred"], g["points"])
    assert abs(loss - g["add_f64"]) < 1e-14 and np.abs(grad - g["add_grad_f64"]).max() < 1e-15
    assert np.abs(dists - g["add_dists_f64"]).max() < 1e-14
    assert abs(loss - g["add"]) < 1e-6 and np.abs(grad - g["add_grad"]).max() < 1e-6        # the float32 run
    loss, grad, parts = so.add_l1_np(g["t_gt"], g["t_pred"], g["points"], disentangled=True)
    assert abs(loss - g["dis_f64"]) < 1e-14 and np.abs(grad - g["dis_grad_f64"]).max() < 1e-15
    assert abs(loss - g["dis"]) < 1e-6 and np.abs(grad - g["dis_grad"]).max() < 1e-6
    assert abs(parts.sum() - loss) < 1e-14
    # the translation / depth terms do not depend on the points: (|dtx| + |dty|)/3 and |dtz|/3 per sample
    dt = g["t_gt"][:, :3, 3].astype(np.float64) - g["t_pred"][:, :3, 3].astype(np.float64)
    assert abs(parts[1] - ((np.abs(dt[:, 0]) + np.abs(dt[:, 1])) / 3).mean()) < 1e-12
    assert abs(parts[2] - (np.abs(dt[:, 2]) / 3).mean()) < 1e-12
    assert np.all(g["dis_grad_f64"][0] == 0)                                                 # exact hit: sgn(0) = 0


def test_g12_cloud_prep_oracle():
    """Row a7: pc_normalize against the reference function; the pairing rule against the loop's own statements."""
    g = load_golden("g12_clouds.npz")
    n, c, sc = so.pc_normalize_np(g["clouds"])
    assert np.abs(n - g["norm"]).max() < 1e-15 and np.abs(c - g["centroid"]).max() < 1e-15 and np.abs(sc - g["scale"]).max() < 1e-15
    one = so.pc_normalize_np(g["clouds"][3])
    assert one[0].shape == (200, 3) and np.abs(one[0] - g["norm"][3]).max() < 1e-15
    assert np.abs(so.rotate_clouds_np(g["pc1"], g["gt_rmat"]) - g["pc_out"]).max() < 2e-7
    assert np.abs(so.rotate_clouds_np(g["pc1"], g["gt_rmat"], transposed=True) - g["gg"]).max() < 2e-7


def test_g8_se3_update_oracle():
    """Next row f1: calculate_T_pred restated (float64) against the reference function's float32 output and autograd."""
    g = load_golden("g8_se3_update.npz")
    assert abs(float(g["fx"]) - 50 / (36 / 320)) < 1e-9 and float(g["fx"]) == float(g["fy"])
    tp = so.se3_update_np(g["out"], g["t_init"])
    assert np.abs(tp - g["t_pred"]).max() < 2e-5
    assert np.abs(tp[:, 3] - np.array([0, 0, 0, 1.0])).max() == 0
    d = so.se3_update_backward_np(g["out"], g["t_init"], g["g"])
    rel = np.abs(d - g["dout"]).max(1) / (1e-3 + np.abs(g["dout"]).max(1))
    assert np.median(rel) < 2e-6 and rel.max() < 1e-4


def test_g9_sampler_and_synthesis_oracle():
    """Next row f4: the sampler's arithmetic against the reference's output for recorded draws; the stateless
    normal generator has unit variance and the synthesised pairs reproduce the pairing rule."""
    g = load_golden("g9_sampler.npz")
    assert np.abs(so.rotations_from_draws_np(g["theta"], g["axis"]) - g["r"]).max() < 2e-6
    z = so.synth_normal_np(3, *np.meshgrid(np.arange(50), np.arange(400), np.arange(3), indexing="ij"))
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02
    flat = z.reshape(-1, 3)
    c = np.corrcoef(flat.T)                                          # the three components of a point, and neighbouring points, do not correlate
    assert np.abs(c - np.eye(3)).max() < 0.02 and abs(np.corrcoef(flat[:-1, 0], flat[1:, 0])[0, 1]) < 0.02
    assert abs((flat ** 3).mean()) < 0.05 and abs((flat ** 4).mean() - 3.0) < 0.1      # skewness 0, kurtosis 3
    z2 = so.synth_normal_np(4, *np.meshgrid(np.arange(50), np.arange(400), np.arange(3), indexing="ij"))
    assert abs(np.corrcoef(z.ravel(), z2.ravel())[0, 1]) < 0.02     # another seed: another stream
    # points 64 apart share a Box-Muller pair for their third component (cosine and sine of one angle): independent all the same
    zz = so.synth_normal_np(5, *np.meshgrid(np.arange(400), np.arange(128), np.arange(3), indexing="ij"))          # 25 600 pairs: sigma 0.006
    for ca, cb in ((2, 2), (0, 0), (0, 2), (2, 1)):
        assert abs(np.corrcoef(zz[:, :64, ca].ravel(), zz[:, 64:, cb].ravel())[0, 1]) < 0.025, (ca, cb)
        assert abs(np.corrcoef((zz[:, :64, ca] ** 2).ravel(), (zz[:, 64:, cb] ** 2).ravel())[0, 1]) < 0.025, (ca, cb)
    # the six uniforms behind a PAIR of points' normals come from ONE full hash and four single-multiply rounds of it: pairwise independence
    # on a 64 x 64 grid, plain and with the low bits magnified 256-fold, against each other and against the next pair's (chi-square as a
    # z-score over 4M pairs; two rounds of the same shape fail this at 8-9 sigma)
    cl, pt = np.meshgrid(np.arange(4096), np.arange(1024), indexing="ij")
    u = so.synth_pair_uniforms_np(3, cl.ravel(), pt.ravel())
    assert all(0.0 < u[i].min() and u[i].max() <= 1.0 for i in (0, 2, 4)) and all(0.0 <= u[i].min() and u[i].max() < 1.0 for i in (1, 3, 5))

    def z_of(a, b, k=64):
        h = np.histogram2d(a % 1.0, b % 1.0, bins=k, range=[[0, 1], [0, 1]])[0]
        e = a.size / (k * k)
        return (((h - e) ** 2 / e).sum() - (k * k - 1)) / np.sqrt(2.0 * (k * k - 1))
    zs = []
    for i in range(6):
        for j in range(i + 1, 6):
            zs += [z_of(u[i], u[j]), z_of(u[i] * 256, u[j] * 256)]
        zs.append(z_of(u[i].reshape(4096, 1024)[:, :-1].ravel(), u[i].reshape(4096, 1024)[:, 1:].ravel()))
    assert max(abs(z) for z in zs) < 4.0, zs
    rng = np.random.default_rng(0)
    p = rng.random((4, 100, 3)) - 0.5
    q = so.synth_pairs_np(p, g["r"][:4], 0.0, 1)
    assert np.abs(q - np.einsum("bac,bic->bia", g["r"][:4].astype(np.float64), p)).max() < 1e-12
