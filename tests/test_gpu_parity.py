"""GPU parity tests: the HIP kernels, called through the C ABI (via the ctypes binding), against
(1) the golden vectors produced by the reference, (2) the oracle on the same seeded inputs, and
(3) size-independent properties at BASELINE.json's full sizes.

Tolerances (float32 path):
  * flip flags: bit-exact.
  * orthogonality ||R^T R - I||_F < 1e-5 on every row.
  * |mean angle - reference mean angle| < 1e-4 degrees.
  * per-row |R - R_ref|: float32 conditioning -- eps * s1/gap with gap = s2+s3 (no flip) or s2-s3
    (flip); asserted as  err * gap/s1 < 3e-6  and  < 1e-5 absolute on well-conditioned rows.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

from conftest import METRIC_GRAD_CASES, ROOT, load_golden, metric_grad_check, orth_err, well_conditioned

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.fixture(scope="module")
def pa():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    import poseestimation_amd as pa_
    from poseestimation_amd import _lib
    _lib.load()                                   # fail loudly if the HIP extension is missing
    return pa_


@pytest.fixture(scope="module")
def rr(pa):
    from poseestimation_amd import rotation_representation
    return rotation_representation


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(DEV, dtype)


def conditioned_err(x, r, r_ref):
    """(|dR| per row, |dR| * gap / s1 per row) for input rows x: gap = s2 + s3 without flip, s2 - s3 with flip -- the measure in
    which a rotation can be judged independently of how well its row determines it (float32 round-off is ~1e-7 there)."""
    m = np.asarray(x, np.float64).reshape(-1, 3, 3)
    s = np.linalg.svd(m, compute_uv=False)
    gap = np.where(np.linalg.det(m) < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2]) / np.maximum(s[:, 0], 1e-300)
    err = np.abs(np.asarray(r, np.float64).reshape(len(m), -1) - np.asarray(r_ref, np.float64).reshape(len(m), -1)).max(1)
    return err, err * gap


def cond_scaled_err(r, r_ref, s, det):
    s = np.asarray(s, np.float64)
    gap = np.where(np.asarray(det) < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2])
    err = np.abs(np.asarray(r, np.float64) - r_ref).reshape(len(s), -1).max(1)
    return err, err * gap / np.maximum(s[:, 0], 1e-300)


# ------------------------------------------------------------------------------------------------
# K1 against the reference's golden vectors
# ------------------------------------------------------------------------------------------------
def test_g1_config1_256_rows(rr):
    g = load_golden("g1_gaussian256.npz")
    r, flip = rr.symmetric_orthogonalization_with_flip(dev(g["x"]))
    r = r.cpu().numpy()
    assert r.shape == (256, 3, 3) and r.dtype == np.float32
    assert np.array_equal(flip.cpu().numpy(), g["det"] < 0)            # det-sign flip: bit-exact
    assert orth_err(r).max() < 1e-5
    err, scaled = cond_scaled_err(r, g["r_f64"], g["s"], g["det"])
    assert scaled.max() < 3e-6
    ok = well_conditioned(g["s"], g["det"])
    assert np.abs(r[ok] - g["r"][ok]).max() < 1e-5                      # vs the reference's own float32 output
    # closer to the float64 answer than the reference's float32 path is, on aggregate
    err_ref = np.abs(g["r"] - g["r_f64"]).reshape(256, -1).max(1)
    assert np.median(err) <= np.median(err_ref)


def test_g2_adversarial(rr):
    g = load_golden("g2_adversarial.npz")
    names = [str(n) for n in g["names"]]
    r = rr.symmetric_orthogonalization(dev(g["x"])).cpu().numpy()
    assert np.isfinite(r).all()
    assert orth_err(r).max() < 1e-5
    assert np.abs(np.linalg.det(r.astype(np.float64)) - 1).max() < 1e-5
    unique = {"identity": 1e-6, "rank2_diag": 1e-6, "rank2_rot": 2e-5, "rotation": 1e-6, "rotation_scaled_1e-20": 1e-6,
              "rotation_scaled_1e+15": 1e-6, "near_equal_sv": 5e-6, "near_equal_sv_flip": 2e-6, "flip_close_s2_s3": 5e-4,
              "tiny_s3_pos": 2e-6, "tiny_s3_neg": 2e-6, "graded": 1e-4,   # graded: s1/gap = 1e3 -> eps*1e3 (reference f32: 4e-5)
              "upper_triangular": 2e-6, "permutation_even": 1e-6}
    for n, tol in unique.items():
        i = names.index(n)
        assert np.abs(r[i] - g["r_f64"][i]).max() < tol, n
    # behaviours of the reference a caller may rely on
    assert np.allclose(r[names.index("zero")], np.eye(3), atol=1e-7)
    assert np.allclose(r[names.index("reflection_z")], np.eye(3), atol=1e-7)


def test_view_semantics_and_bad_shapes(rr):
    g = load_golden("g2_shape_2x5x9.npz")
    r = rr.symmetric_orthogonalization(dev(g["x"]))
    assert tuple(r.shape) == (10, 3, 3)
    assert np.abs(r.cpu().numpy() - g["r"]).max() < 1e-5
    with pytest.raises(RuntimeError, match=r"shape '\[-1, 3, 3\]' is invalid for input of size 10"):
        rr.symmetric_orthogonalization(torch.zeros(10, device=DEV))
    with pytest.raises(TypeError):
        rr.symmetric_orthogonalization(torch.zeros(2, 9, device=DEV, dtype=torch.int32))
    z64 = rr.symmetric_orthogonalization(torch.zeros(2, 9, device=DEV, dtype=torch.float64))       # zero -> identity, in double
    assert z64.dtype == torch.float64 and torch.equal(z64, torch.eye(3, device=DEV, dtype=torch.float64).expand(2, 3, 3))
    assert tuple(rr.symmetric_orthogonalization(torch.zeros(0, 9, device=DEV)).shape) == (0, 3, 3)


@pytest.mark.parametrize("b", [1, 63, 64, 65, 255, 256, 257, 511, 513, 1000])
def test_ragged_batches_and_unaligned_pointers(rr, c_oracle, b):
    rng = np.random.default_rng(b)
    x = rng.standard_normal((b + 3, 9)).astype(np.float32)
    xd = dev(x)
    for off in (0, 1, 3):                                   # row offset 1 -> base pointer 36 B off: not 16-B aligned
        r = rr.symmetric_orthogonalization(xd[off:off + b]).cpu().numpy()
        ref, _ = c_oracle.project(x[off:off + b], want_flip=True)
        assert np.quantile(np.abs(r - ref), 0.9) < 1e-6
        assert orth_err(r).max() < 1e-5
    # non-contiguous input (every other row)
    r = rr.symmetric_orthogonalization(xd[::2]).cpu().numpy()
    assert np.quantile(np.abs(r - c_oracle.project(x[::2])), 0.9) < 1e-6


@pytest.mark.parametrize("off", [1, 2, 3])
def test_offset_views_row_by_row_against_the_oracle(rr, c_oracle, off):
    """Views that start at row 1, 2 or 3 of a tensor (base pointers 4 / 8 / 12 bytes off 16-byte alignment: the shards of an
    uneven split) stream through the engine's 16-byte buffer loads and stores.  EVERY row of K1, K2, K3 and K4 on such a
    view is compared with the float64 oracle in the conditioned measure -- a shifted or stale element anywhere in a misaligned
    block shows up as an O(1) error in that row -- and a bfloat16 view at an even row offset likewise."""
    from oracle import so3_oracle as so
    b = 64 * 37 + 11                                         # whole units and a remainder
    rng = np.random.default_rng(100 + off)
    x = rng.standard_normal((b + 4, 9)).astype(np.float32)
    t = so.symmetric_orthogonalization_np(rng.standard_normal((b + 4, 9))).astype(np.float32).reshape(-1, 9)
    g = rng.standard_normal((b + 4, 9)).astype(np.float32)
    xs, ts, gs = x[off:off + b], t[off:off + b], g[off:off + b]
    ref, s, d = so.symmetric_orthogonalization_np(xs, return_parts=True)
    gap = np.where(d < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2]) / s[:, 0]
    cond = lambda got, want: (np.abs(np.asarray(got, np.float64).reshape(b, -1) - np.asarray(want).reshape(b, -1)).max(1) * gap).max()
    xd, td, gd = dev(x), dev(t), dev(g)
    assert xd[off:off + b].data_ptr() % 16 == (36 * off) % 16 != 0
    # K1 (+ flip flags), K2
    xv = xd[off:off + b].clone().requires_grad_(True) if False else xd[off:off + b].detach().requires_grad_(True)
    r = rr.symmetric_orthogonalization(xv)
    assert cond(r.detach().cpu().numpy(), ref) < 2e-6 and orth_err(r.detach().cpu().numpy()).max() < 1e-5
    r2, flip = rr.symmetric_orthogonalization_with_flip(xd[off:off + b])
    assert torch.equal(r2, r.detach()) and np.array_equal(flip.cpu().numpy(), d < 0)
    r.backward(gd[off:off + b].view(b, 3, 3))
    dref = so.projection_backward_np(xs.astype(np.float64), gs.astype(np.float64))
    rel = np.abs(xv.grad.cpu().numpy().reshape(b, 9) - dref.reshape(b, 9)).max(1) * gap * gap * s[:, 0]
    assert rel.max() < 2e-5
    # K3 on the views (R, dM, loss), K3' stand-alone
    xk = xd[off:off + b].detach().requires_grad_(True)
    loss, rk = rr.frobenius_head(xk, td[off:off + b])
    loss.backward()
    lref, dxref, _ = so.frobenius_fwd_bwd_np(xs, ts)
    assert torch.equal(rk, r.detach()) and abs(loss.item() - lref) < 2e-6
    assert (np.abs(xk.grad.cpu().numpy().reshape(b, 9) - dxref.reshape(b, 9)).max(1) * gap * gap * s[:, 0] * b).max() < 5e-5
    # K4 per row and fused, K4', K1+K4
    deg = rr.angle_error(r.detach().view(b, 9)[:], td[off:off + b])
    dref_deg = so.angle_error_np(r.detach().cpu().numpy(), ts)
    assert np.abs(deg.cpu().numpy() - dref_deg).max() < 1e-9
    sc = rr.angle_error_sum_count(r.detach(), td[off:off + b])
    assert sc[1].item() == b and abs(sc[0].item() - dref_deg.sum()) < 1e-8 * b
    fused = rr.head_angle_error(xd[off:off + b], td[off:off + b])
    assert np.abs(fused.cpu().numpy() - dref_deg).max() < 1e-9
    # bfloat16 storage at an even row offset (dword aligned: the engine) and an odd one (the tile kernels)
    xb = dev(x).bfloat16()
    for o2 in (2 * (off // 2 + 1), 2 * (off // 2) + 1):
        xbv = xb[o2:o2 + b - 4]
        rb = rr.symmetric_orthogonalization(xbv).cpu().numpy()
        refb, sb, db = so.symmetric_orthogonalization_np(xbv.float().cpu().numpy(), return_parts=True)
        gapb = np.where(db < 0, sb[:, 1] - sb[:, 2], sb[:, 1] + sb[:, 2]) / sb[:, 0]
        assert (np.abs(rb - refb).reshape(len(rb), -1).max(1) * gapb).max() < 2e-6, o2


def test_float64_head_and_backward(rr):
    """Double tensors go through the float64 kernels (the reference's function accepts them) and come back double."""
    from oracle import so3_oracle as so
    g1 = load_golden("g1_gaussian256.npz")
    x = dev(g1["x"], torch.float64)
    r = rr.symmetric_orthogonalization(x)
    assert r.dtype == torch.float64 and tuple(r.shape) == (256, 3, 3)
    assert np.abs(r.cpu().numpy() - g1["r_f64"]).max() < 1e-11                 # vs the reference run in float64
    gen = np.random.default_rng(64)
    xs = gen.standard_normal((100_003, 9))
    xs[:1000] *= np.array([1, 1, 1, 1e-3, 1e-3, 1e-3, 1e-6, 1e-6, 1e-6])       # graded rows
    xs[1000] = 0.0                                                              # zero -> identity
    xs[1001] = np.diag([1.0, 1.0, -1.0]).ravel()                                # reflection -> identity
    xs[1002] = np.outer([1.0, 2.0, 3.0], [0.5, -1.0, 2.0]).ravel()              # rank one: any rotation, but a rotation
    xt = torch.as_tensor(xs).to(DEV).requires_grad_(True)
    rt, flip = rr.symmetric_orthogonalization_with_flip(xt.detach())
    ref, s, d = so.symmetric_orthogonalization_np(xs, return_parts=True)
    out = rt.cpu().numpy()
    assert orth_err(out).max() < 1e-13 and np.abs(np.linalg.det(out) - 1).max() < 1e-13
    assert np.array_equal(out[1000], np.eye(3)) and np.abs(out[1001] - np.eye(3)).max() < 1e-15
    gap = np.where(d < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2]) / np.maximum(s[:, 0], 1e-300)
    ok = gap > 1e-6
    ok[1000:1003] = False
    assert (np.abs(out - ref).reshape(len(xs), -1).max(1) * gap)[ok].max() < 1e-13
    assert np.array_equal(flip.cpu().numpy()[ok], d[ok] < 0)
    g = gen.standard_normal((len(xs), 3, 3))
    rr.symmetric_orthogonalization(xt).backward(torch.as_tensor(g).to(DEV))
    assert xt.grad.dtype == torch.float64
    wc = gap > 1e-3
    wc[1000:1003] = False
    refg = so.projection_backward_np(xs[wc], g[wc])
    err = np.abs(xt.grad.cpu().numpy()[wc].reshape(-1, 9) - refg.reshape(-1, 9)).max(1)
    assert (err * gap[wc] * s[wc, 0]).max() < 1e-11


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_numerically_rank_deficient_input_still_gives_rotations(rr, dtype):
    """Exactly or numerically rank-one / rank-two input (small integers, outer products, nine equal network outputs):
    the SVD is not unique there and the answer is implementation-defined, but it must be a rotation, as LAPACK's is."""
    gen = torch.Generator(device=DEV).manual_seed(123)
    n = 500_000
    a = torch.randn(n, 3, 3, device=DEV, generator=gen)
    families = {
        "small integers": torch.randint(-3, 4, (n, 3, 3), device=DEV, generator=gen).float(),
        "outer products": torch.randn(n, 3, 1, device=DEV, generator=gen) @ torch.randn(n, 1, 3, device=DEV, generator=gen),
        "integer outer products": torch.randint(-3, 4, (n, 3, 1), device=DEV, generator=gen).float() @ torch.randint(-3, 4, (n, 1, 3), device=DEV, generator=gen).float(),
        "nine equal entries": torch.randn(n, 1, 1, device=DEV, generator=gen).expand(n, 3, 3).contiguous(),
        "rank two": torch.cat((a[:, :2], a[:, :1] + a[:, 1:2]), 1),
        "rank two + 1e-7": torch.cat((a[:, :2], a[:, :1] + a[:, 1:2]), 1) + 1e-7 * torch.randn(n, 3, 3, device=DEV, generator=gen),
        "outer + 1e-7": torch.randn(n, 3, 1, device=DEV, generator=gen) @ torch.randn(n, 1, 3, device=DEV, generator=gen) + 1e-7 * a,
        # what the quaternion fast path must hand to the Jacobi path: exact double roots at the top of K's spectrum (s2 = s3,
        # det < 0: frequent with entries in -1..1), reflections (a triple root), scales outside its window
        "entries in -1..1": torch.randint(-1, 2, (n, 3, 3), device=DEV, generator=gen).float(),
        "reflections": rr.symmetric_orthogonalization(torch.randn(n, 9, device=DEV, generator=gen)) * torch.tensor([1.0, 1.0, -1.0], device=DEV),
        "scaled 1e-5": 1e-5 * a,
        "scaled 1e+5": 1e5 * a,
    }
    tol = 1e-5 if dtype == torch.float32 else 1e-12
    for name, m in families.items():
        x = m.to(dtype).requires_grad_(True)
        r = rr.symmetric_orthogonalization(x)
        cols = [(r[:, :, i] * r[:, :, j]).sum(1) - (1.0 if i == j else 0.0) for i in range(3) for j in range(3)]
        orth = torch.stack(cols, 1).norm(dim=1)
        det = torch.linalg.det(r.detach().double())
        assert orth.max().item() < tol and (det - 1).abs().max().item() < 10 * tol, (name, orth.max().item())
        # R maximises tr(R^T M) over SO(3): compare the objective with float64 LAPACK's (unique even where R is not)
        s = torch.linalg.svdvals(m.double())
        best = s[:, 0] + s[:, 1] + torch.where(torch.linalg.det(m.double()) < 0, -s[:, 2], s[:, 2])
        got = (r.detach().double() * m.double()).sum((1, 2))
        assert ((best - got) / s[:, 0].clamp_min(1e-30)).max().item() < (2e-6 if dtype == torch.float32 else 1e-12), name
        r.backward(torch.randn(n, 3, 3, device=DEV, generator=gen).to(dtype))
        assert torch.isfinite(x.grad).all(), name               # floored denominators: large but finite


def test_a_row_does_not_depend_on_its_neighbours(rr):
    """Batch invariance, bit for bit: proj(x)[i] == proj(x[i:i+1]) whatever else is in the batch, whichever kernel
    (packed streaming engine, one-row-per-lane tile kernel, aligned or not) ends up processing the row.  The adaptive
    sweep is decided per wave but applied per matrix, so LAPACK-like per-matrix semantics hold."""
    gen = torch.Generator(device=DEV).manual_seed(77)
    n = 100_000
    x = torch.randn(n, 9, device=DEV, generator=gen)
    # degenerate rows among the ordinary ones: zeros, exact rank one, huge and tiny scales (absolute constants of the
    # algorithm -- the 1e-18 in the rotation, the rank thresholds -- must not make a row depend on its wave-mates)
    x[5::97] = 0.0
    x[11::89] = (torch.randint(-3, 4, (len(x[11::89]), 3, 1), device=DEV, generator=gen).float()
                 @ torch.randint(-3, 4, (len(x[11::89]), 1, 3), device=DEV, generator=gen).float()).reshape(-1, 9)
    x[17::83] *= 1e18
    x[23::79] *= 1e-18
    gup = torch.randn(n, 3, 3, device=DEV, generator=gen)
    full_x = x.clone().requires_grad_(True)
    full = rr.symmetric_orthogonalization(full_x)
    full.backward(gup)
    for lo, hi in ((0, 1), (63, 65), (1000, 1064), (12_345, 54_321), (n - 7, n)):
        for offset in (0, 1):                                   # 16-byte aligned, and 4-byte aligned only
            buf = torch.empty((hi - lo) * 9 + offset, device=DEV)
            part_x = buf[offset:].view(hi - lo, 9)
            part_x.copy_(x[lo:hi])
            part_x.requires_grad_(True)
            part = rr.symmetric_orthogonalization(part_x)
            part.backward(gup[lo:hi])
            assert torch.equal(part, full[lo:hi]), (lo, hi, offset)
            assert torch.equal(part_x.grad, full_x.grad[lo:hi]), (lo, hi, offset)
    shuffled = torch.randperm(n, device=DEV, generator=gen)
    assert torch.equal(rr.symmetric_orthogonalization(x[shuffled]), full.detach()[shuffled])


@pytest.mark.parametrize("bf16", [False, True])
def test_deferred_hard_rows_are_the_rows_the_tile_kernel_gives(rr, bf16):
    """K1's engine queues the hard rows of a round that holds few of them and runs the Jacobi path once per wave, patching the
    rows it had already stored; a round dense in hard rows takes it on the spot; a full queue too.  Every row, bit for bit,
    against the one-row-per-thread kernel (same arithmetic, no queue), on a batch whose share of hard rows runs from none to
    all by region, with and without flip flags, 16-byte aligned and not, and against the float64 oracle's verdict per family."""
    from poseestimation_amd import _lib
    import importlib.util
    spec = importlib.util.spec_from_file_location("k1_hard_rows", os.path.join(ROOT, "tools", "k1_hard_rows.py"))
    hr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hr)
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(123)
    d = torch.device(DEV)
    n = 64 * 6000 + 37                                                       # 6000 whole units for the engine + a remainder for the tile kernel
    x = torch.randn(n, 9, device=DEV, generator=gen)
    families = ("near-reflection", "entries in {-1,0,1}", "generic ties", "rank one", "all zero")
    region = n // 6
    for j, share in enumerate((0.0, 0.003, 0.03, 0.2, 0.6, 1.0)):         # sparse (queued), mixed, dense (on the spot)
        lo, hi = j * region, (j + 1) * region
        pick = torch.nonzero(torch.rand(hi - lo, device=DEV, generator=gen) < share).flatten() + lo
        for f, name in enumerate(families):
            idx = pick[f::len(families)]
            if idx.numel():
                x[idx] = hr.family(name, idx.numel(), d, gen).reshape(-1, 9)
    if bf16:
        x = x.bfloat16().float()                                              # values a bfloat16 tensor can hold
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.empty(n, 9, device=DEV)
    hard = torch.empty(n, dtype=torch.uint8, device=DEV)
    assert lib.so3_project_fwd_diag_f32(x.data_ptr(), ref.data_ptr(), hard.data_ptr(), n, st) == 0
    # the mixture really is hard where it is meant to be (bfloat16 rounding breaks some ties; the all-zero rows are answered by the
    # forward itself since round 4 and are not hard any more)
    assert 0.07 * n < int(hard.sum().item()) < 0.45 * n
    fn = lib.so3_project_fwd_bf16 if bf16 else lib.so3_project_fwd_f32
    for offset in (0, 1, 2):                                               # rows 36 B apart: 16-, 4- and 8-byte aligned starts
        src = torch.empty((n + 1) * 9, device=DEV, dtype=torch.bfloat16 if bf16 else torch.float32)
        xin = src[(2 * offset if bf16 else offset):][:n * 9].view(n, 9)     # bfloat16: even element offsets keep dword alignment
        xin.copy_(x)
        for want_flip in (False, True):
            out = torch.full((n * 9 + 8,), 7.0, device=DEV)
            r = out[offset:offset + n * 9].view(n, 9)
            flip = torch.empty(n, dtype=torch.uint8, device=DEV) if want_flip else None
            assert fn(xin.data_ptr(), r.data_ptr(), flip.data_ptr() if want_flip else None, n, st) == 0
            assert torch.equal(r, ref), (offset, want_flip, int((r != ref).any(dim=1).sum().item()))
            assert (out[:offset] == 7).all() and (out[offset + n * 9:] == 7).all()
            if want_flip:
                det = torch.linalg.det(x.double().view(n, 3, 3))
                sure = det.abs() > 1e-6
                assert torch.equal(flip.bool()[sure], (det < 0)[sure])
    # orthogonal, det +1, and optimal wherever the answer is unique (the float64 oracle's conditioned error)
    r3 = ref.view(n, 3, 3).double()
    assert (r3.transpose(1, 2) @ r3 - torch.eye(3, device=DEV, dtype=torch.float64)).abs().amax() < 1e-5
    assert (torch.linalg.det(r3) - 1).abs().max() < 1e-5


@pytest.mark.parametrize("plan", [("sparse", "reflection", "reflection", "ties", "sparse"), ("rank one", "rank one", "sparse", "sparse", "reflection"),
                                  ("crowded", "crowded", "crowded", "crowded", "crowded"), ("ties", "ties", "reflection", "reflection", "zero"),
                                  ("zero", "zero", "rank one", "mixed", "mixed")])
def test_hard_row_queue_over_the_rounds_of_a_wave(rr, plan):
    """What K1's engine does with hard rows depends on what the WAVE met before: its queue fills over its rounds (and a round it
    has no room for takes the Jacobi path on the spot), a round after a dense one is asked whether all its rows are hard by their
    invariants (reflections, rank one: yes, no fast path; ties: no, and the wave does not ask again).  The result may not: five
    passes of the grid (wave w takes rounds w, w + 3072, ...), each pass of its own kind, every row bit for bit against the
    one-row-per-thread kernel."""
    from poseestimation_amd import _lib
    import importlib.util
    spec = importlib.util.spec_from_file_location("k1_hard_rows", os.path.join(ROOT, "tools", "k1_hard_rows.py"))
    hr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hr)
    lib = _lib.load()
    d = torch.device(DEV)
    gen = torch.Generator(device=DEV).manual_seed(sum(len(k) * (i + 1) for i, k in enumerate(plan)))
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    waves = cus * 4 * 3                                                     # K1: three waves per SIMD, 128 rows per wave and round
    per_pass = waves * 128
    n = per_pass * len(plan) + 64 * 3 + 11                                # + an odd tail of units + a remainder for the tile kernel
    x = torch.randn(n, 9, device=DEV, generator=gen)
    families = ("near-reflection", "entries in {-1,0,1}", "generic ties", "rank one", "all zero")

    def fill(lo, hi, kind):
        if kind in ("sparse", "crowded", "mixed"):
            share = {"sparse": 0.05, "crowded": 0.22, "mixed": 0.5}[kind]   # 0.22: ~28 of a round's 128 rows -- queued, until the queue is full
            pick = torch.nonzero(torch.rand(hi - lo, device=DEV, generator=gen) < share).flatten() + lo
            for f, name in enumerate(families):
                idx = pick[f::len(families)]
                if idx.numel():
                    x[idx] = hr.family(name, idx.numel(), d, gen).reshape(-1, 9)
        else:
            name = {"reflection": "near-reflection", "ties": "generic ties", "rank one": "rank one", "zero": "all zero"}[kind]
            x[lo:hi] = hr.family(name, hi - lo, d, gen).reshape(-1, 9)

    for p_, kind in enumerate(plan):
        fill(p_ * per_pass, (p_ + 1) * per_pass, kind)
    fill(per_pass * len(plan), n, "mixed")
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.empty(n, 9, device=DEV)
    hard = torch.empty(n, dtype=torch.uint8, device=DEV)
    assert lib.so3_project_fwd_diag_f32(x.data_ptr(), ref.data_ptr(), hard.data_ptr(), n, st) == 0
    out = torch.full((n * 9 + 8,), 7.0, device=DEV)
    r = out[4:4 + n * 9].view(n, 9)
    assert lib.so3_project_fwd_f32(x.data_ptr(), r.data_ptr(), None, n, st) == 0
    bad = (r != ref).any(dim=1)
    assert not bool(bad.any()), (plan, int(bad.sum().item()), torch.nonzero(bad).flatten()[:8].tolist(), int(hard.sum().item()))
    assert (out[:4] == 7).all() and (out[4 + n * 9:] == 7).all()
    # the passes are what they are meant to be: every row of a reflection / rank-one / zero pass is hard, a sparse one has a few per cent
    for p_, kind in enumerate(plan):
        share = hard[p_ * per_pass:(p_ + 1) * per_pass].float().mean().item()
        if kind in ("reflection", "rank one", "ties"):
            assert share > 0.999, (kind, share)
        elif kind == "zero":                                                 # a dead head is answered by the forward itself (round 4): the identity, not hard
            assert share == 0.0 and torch.equal(r[p_ * per_pass:(p_ + 1) * per_pass], torch.eye(3, device=DEV).reshape(1, 9).expand(per_pass, 9))
        elif kind == "sparse":
            assert 0.01 < share < 0.06, (kind, share)


def test_parked_and_dense_hard_rows_in_the_two_input_kernels(rr):
    """K2, K3 and K1+K4 park the hard rows of a round that holds few of them (both inputs and the row number, in the workgroup's LDS
    list) and redo them one matrix per lane behind the loop; a dense round runs the packed Jacobi path on the spot.  Every row of a
    batch whose share of hard rows runs from none to all by region, bit for bit against the same rows through the one-row-per-thread
    kernels (K2: the tile kernel in 63-row calls; K3, K1+K4: the one-workgroup kernels in 1024-row calls -- 1/16 of the batch, so the
    1/B in the gradient differs by an exact power of two), every output of every variant; the loss and the angle sum to round-off."""
    from poseestimation_amd import _lib
    import importlib.util
    spec = importlib.util.spec_from_file_location("k1_hard_rows", os.path.join(ROOT, "tools", "k1_hard_rows.py"))
    hr = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hr)
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(321)
    d = torch.device(DEV)
    n, chunk = 16384, 1024
    x = torch.randn(n, 9, device=DEV, generator=gen)
    families = ("near-reflection", "entries in {-1,0,1}", "generic ties", "rank one", "all zero")
    region = n // 8
    for j, share in enumerate((0.0, 0.005, 0.03, 0.1, 0.2, 0.35, 0.6, 1.0)):   # parked (sparse), the list filling up, dense
        lo, hi = j * region, (j + 1) * region
        pick = torch.nonzero(torch.rand(hi - lo, device=DEV, generator=gen) < share).flatten() + lo
        for f, name in enumerate(families):
            idx = pick[f::len(families)]
            if idx.numel():
                x[idx] = hr.family(name, idx.numel(), d, gen).reshape(-1, 9)
    g = torch.randn(n, 9, device=DEV, generator=gen)
    t = hr.haar(n, d, gen).reshape(n, 9).contiguous()
    st = torch.cuda.current_stream().cuda_stream
    p = lambda a: None if a is None else a.data_ptr()
    new = lambda *shape, dt=torch.float32: torch.full(shape, float("nan"), device=DEV, dtype=dt)
    # K2
    dm_e, dm_r = new(n, 9), new(n, 9)
    assert lib.so3_project_bwd_f32(p(x), p(g), p(dm_e), n, st) == 0
    for lo in range(0, n, 63):
        m = min(63, n - lo)
        assert lib.so3_project_bwd_f32(p(x[lo:]), p(g[lo:]), p(dm_r[lo:]), m, st) == 0
    same = lambda a, b_: bool(((a == b_) | (torch.isnan(a) & torch.isnan(b_))).all())
    assert same(dm_e, dm_r), int((~((dm_e == dm_r) | (torch.isnan(dm_e) & torch.isnan(dm_r)))).any(dim=1).sum().item())
    # K3, every variant
    for want_r, want_dm in ((True, True), (False, True), (True, False), (False, False)):
        r_e, d_e = (new(n, 9) if want_r else None), (new(n, 9) if want_dm else None)
        r_r, d_r = (new(n, 9) if want_r else None), (new(n, 9) if want_dm else None)
        ls_e, ls_r = new(1, dt=torch.float64), new(1, dt=torch.float64)
        assert lib.so3_frob_fwd_bwd_v2_f32(p(x), p(t), p(r_e), p(d_e), p(ls_e), None, None, 0, n, st) == 0
        total = 0.0
        for lo in range(0, n, chunk):
            sub = lambda a: None if a is None else a[lo:].data_ptr()
            assert lib.so3_frob_fwd_bwd_v2_f32(sub(x), sub(t), sub(r_r), sub(d_r), p(ls_r), None, None, 0, chunk, st) == 0
            total += ls_r.item()
        if want_r:
            assert same(r_e, r_r), (want_r, want_dm)
        if want_dm:
            assert same(d_e * float(n // chunk), d_r), (want_r, want_dm)      # 1/B: an exact power of two apart
        assert abs(ls_e.item() - total) < 1e-9 * total
    # bfloat16 storage (config #4's): K2 and K3 with M read and dM written as bfloat16 -- a parked row's dM is overwritten element by element
    xb = x.to(torch.bfloat16)
    db_e, db_r = new(n, 9, dt=torch.bfloat16), new(n, 9, dt=torch.bfloat16)
    assert lib.so3_project_bwd_bf16(p(xb), p(g), p(db_e), n, st) == 0
    for lo in range(0, n, 63):
        assert lib.so3_project_bwd_bf16(p(xb[lo:]), p(g[lo:]), p(db_r[lo:]), min(63, n - lo), st) == 0
    assert same(db_e.float(), db_r.float())
    r_e, r_r, db_e, db_r = new(n, 9), new(n, 9), new(n, 9, dt=torch.bfloat16), new(n, 9, dt=torch.bfloat16)
    ls_e, ls_r = new(1, dt=torch.float64), new(1, dt=torch.float64)
    assert lib.so3_frob_fwd_bwd_v2_bf16(p(xb), p(t), p(r_e), p(db_e), p(ls_e), None, None, 0, n, st) == 0
    total = 0.0
    for lo in range(0, n, chunk):
        assert lib.so3_frob_fwd_bwd_v2_bf16(p(xb[lo:]), p(t[lo:]), p(r_r[lo:]), p(db_r[lo:]), p(ls_r), None, None, 0, chunk, st) == 0
        total += ls_r.item()
    # (1/B enters the gradient before the bfloat16 rounding: a power of two commutes with it, short of underflow)
    assert same(r_e, r_r) and same(db_e.float() * float(n // chunk), db_r.float()) and abs(ls_e.item() - total) < 1e-9 * total
    # K1+K4: per-row angles + R (float64 on every row), and the sum
    deg_e, deg_r, rr_e, rr_r = new(n, dt=torch.float64), new(n, dt=torch.float64), new(n, 9), new(n, 9)
    fl = torch.zeros(1, dtype=torch.int32, device=DEV)
    assert lib.so3_project_angle_error_v2_f32(p(x), p(t), p(rr_e), p(deg_e), None, p(fl), None, 0, n, st) == 0
    sc = new(2, dt=torch.float64)
    for lo in range(0, n, chunk):
        assert lib.so3_project_angle_error_v2_f32(x[lo:].data_ptr(), t[lo:].data_ptr(), rr_r[lo:].data_ptr(), deg_r[lo:].data_ptr(), p(sc), p(fl), None, 0,
                                                  chunk, st) == 0
    assert same(rr_e, rr_r) and same(deg_e, deg_r)
    for flags in (_lib.EXACT_F64, 0):
        assert lib.so3_project_angle_error_v2_f32(p(x), p(t), None, None, p(sc), p(fl), None, flags, n, st) == 0
        assert sc[1].item() == n and abs(sc[0].item() - deg_r.sum().item()) < (1e-9 if flags else 1e-5) * n
    # and what the hard rows got is a rotation
    r3 = rr_e.view(n, 3, 3).double()
    assert (r3.transpose(1, 2) @ r3 - torch.eye(3, device=DEV, dtype=torch.float64)).abs().amax() < 1e-5


def test_device_rows_match_the_host_model_of_the_same_templates(rr):
    """csrc/so3_device.h compiled for the host (oracle/kernel_model.cpp) against the device: same algorithm, the only
    difference being 1-ulp v_rsq/v_sqrt/v_rcp versus correctly rounded libm."""
    from oracle import kernel_model as km
    from oracle import so3_oracle as so
    if km.clangxx() is None:
        pytest.skip("clang++ is not available")
    rng = np.random.default_rng(99)
    x = rng.standard_normal((200_000, 9)).astype(np.float32)
    x[:1000] = (rng.standard_normal((1000, 3, 1)) @ rng.standard_normal((1000, 1, 3))).reshape(1000, 9)   # rank one
    x[1000:2000] = rng.integers(-3, 4, (1000, 9))
    r_dev, flip_dev = rr.symmetric_orthogonalization_with_flip(dev(x))
    r_mod, flip_mod = km.project(x, want_flip=True)
    assert np.array_equal(flip_dev.cpu().numpy(), flip_mod)
    _, s, d = so.symmetric_orthogonalization_np(x, return_parts=True)
    gap = np.where(d < 0, s[:, 1] - s[:, 2], s[:, 1] + s[:, 2]) / np.maximum(s[:, 0], 1e-300)
    diff = np.abs(r_dev.cpu().numpy() - r_mod).reshape(len(x), -1).max(1)
    ok = gap > 1e-6
    assert (diff * gap)[ok].max() < 3e-6                      # a different rounding of rsq may flip the adaptive-sweep decision
    assert np.median(diff[ok]) < 2e-7


def test_nan_rows_stay_local(rr):
    x = torch.randn(300, 9, device=DEV)
    x[7, 4] = float("nan")
    x[200, 0] = float("inf")
    r = rr.symmetric_orthogonalization(x)
    bad = torch.isnan(r).reshape(300, -1).any(1)
    assert bad[7] and bad[200] and int(bad.sum()) == 2


def test_bf16_input(rr, c_oracle):
    g = load_golden("g4_frobenius512.npz")
    xb = torch.from_numpy(g["x_bf16_bits"]).view(torch.bfloat16).to(DEV)
    r = rr.symmetric_orthogonalization(xb)
    assert r.dtype == torch.float32
    assert np.quantile(np.abs(r.cpu().numpy() - g["r"]), 0.99) < 5e-6
    r16 = rr.symmetric_orthogonalization(xb.float().half())
    assert orth_err(r16.cpu().numpy()).max() < 1e-5


# ------------------------------------------------------------------------------------------------
# K1 + K4 at BASELINE.json's full size (config #2: 1M rows)
# ------------------------------------------------------------------------------------------------
def test_config2_one_million_rows(rr, c_oracle):
    g = load_golden("g6_stats_1m.npz")
    n = int(g["n"])
    torch.manual_seed(int(g["seed_x"]))
    x = torch.randn(n, 9)
    torch.manual_seed(int(g["seed_t"]))
    t_in = torch.randn(n, 9)
    assert np.array_equal(x[:64].numpy(), g["x_head"])
    r, flip = rr.symmetric_orthogonalization_with_flip(x.to(DEV))
    t = rr.symmetric_orthogonalization(t_in.to(DEV))
    # every one of the 1M flip flags equals the reference's
    assert np.array_equal(np.packbits(flip.cpu().numpy()), g["flip_bits"])
    rc = r.cpu().numpy()
    assert orth_err(rc).max() < 1e-5
    assert np.abs(rc[:64] - g["r_head"]).max() < 1e-5
    # mean geodesic angle vs the decoy target: the BASELINE metric
    deg = rr.angle_error(r, t)
    assert deg.dtype == torch.float64 and tuple(deg.shape) == (n,)
    mean = deg.mean().item()
    assert abs(mean - float(g["mean_angle_deg"])) < 1e-4            # vs the reference (float32 torch.svd)
    assert abs(mean - float(g["mean_angle_deg_f64"])) < 1e-4        # vs the reference run in float64
    sc = rr.angle_error_sum_count(r, t)
    assert sc[1].item() == n and abs(sc[0].item() / n - mean) < 1e-9
    # row by row against the float64 C oracle
    ref = c_oracle.project(x.numpy())
    err, scaled = conditioned_err(x.numpy(), rc, ref)
    assert np.median(err) < 2e-7 and np.quantile(err, 0.99) < 2e-6 and scaled.max() < 2e-6       # 1.2e-7, 9e-7, 8e-7 measured
    deg_or, _ = c_oracle.angle_error(rc, t.cpu().numpy())           # K4 against the oracle on identical inputs
    assert np.abs(deg.cpu().numpy() - deg_or).max() < 1e-9


def test_properties_at_full_size(rr):
    n = 1_000_000
    gen = torch.Generator(device=DEV).manual_seed(11)
    m = torch.randn(n, 3, 3, device=DEV, generator=gen)
    r = rr.symmetric_orthogonalization(m)
    # idempotence: a rotation projects to itself
    rr2 = rr.symmetric_orthogonalization(r)
    assert (rr2 - r).abs().max().item() < 2e-6
    # scale invariance, including power-of-two and non-power-of-two factors
    for f in (2.0 ** -40, 3.7e5):
        assert (rr.symmetric_orthogonalization(m * f) - r).abs().median().item() < 2e-7
    # equivariance proj(Q M P) = Q proj(M) P for rotations Q, P
    q = rr.symmetric_orthogonalization(torch.randn(n, 9, device=DEV, generator=gen))
    p = rr.symmetric_orthogonalization(torch.randn(n, 9, device=DEV, generator=gen))
    lhs = rr.symmetric_orthogonalization(q @ m @ p)
    rhs = q @ r @ p
    d = (lhs - rhs).abs().reshape(n, -1).max(1).values
    assert d.median().item() < 5e-7 and d.quantile(0.999).item() < 2e-5
    # det = +1 and optimality: tr(R^T M) >= tr(T^T M) for any other rotation T
    assert (torch.linalg.det(r.double()) - 1).abs().max().item() < 1e-5
    gain = (r * m).sum((1, 2)) - (q * m).sum((1, 2))
    assert gain.min().item() > -1e-4


# ------------------------------------------------------------------------------------------------
# K4 details
# ------------------------------------------------------------------------------------------------
def test_g3_angle_error_and_raise(rr):
    g = load_golden("g3_angles.npz")
    deg = rr.angle_error(dev(g["r1"]), dev(g["r2"]))
    assert np.abs(deg.cpu().numpy() - g["deg"]).max() < 1e-9
    with pytest.raises(ValueError, match="angle out of range, input probably not proper rotation matrices"):
        rr.angle_error(dev(g["bad1"]), dev(g["bad2"]))
    assert tuple(rr.angle_error(dev(g["bad1"]), dev(g["bad2"]), check=False).shape) == (3,)
    assert np.abs(rr.angle_error(dev(g["nearly1"]), dev(g["nearly2"])).cpu().numpy() - g["deg_nearly"]).max() < 1e-12
    assert tuple(rr.angle_error(torch.zeros(0, 3, 3, device=DEV), torch.zeros(0, 3, 3, device=DEV)).shape) == (0,)


def test_g3_geodesic_radians(rr):
    g = load_golden("g3_angles.npz")
    rad = rr.compute_geodesic_distance_from_two_matrices(dev(g["r1"]), dev(g["r2"]))
    assert rad.dtype == torch.float32 and tuple(rad.shape) == (256,)
    got = rad.cpu().numpy().astype(np.float64)
    assert np.abs(np.cos(got) - np.cos(g["rad"].astype(np.float64))).max() < 1e-6
    assert np.abs(got[5:] - g["rad"][5:]).max() < 2e-5


# ------------------------------------------------------------------------------------------------
# K2 / K3: backward and the fused loss (config #4)
# ------------------------------------------------------------------------------------------------
def test_g14_geodesic_with_reduction(rr):
    """geodesic(R1, R2, reduction) of point_cloud/main.py:61-73 against the golden the reference function produced: per-row angles
    (cosines to 1e-6: float32 trace; the eps clamp keeps them off 0 and pi), mean and sum as 0-dim float32, None for anything else;
    ragged sizes and offset views go through the tile kernel's share of the sum."""
    from oracle import so3_oracle as so
    g = load_golden("g14_geodesic_reduction.npz")
    g3 = load_golden("g3_angles.npz")
    for tag, (a, b) in (("g3", (g3["r1"], g3["r2"])), ("haar", (g["a"], g["b"]))):
        da, db = dev(a), dev(b)
        rad = rr.geodesic(da, db, "none")
        assert rad.dtype == torch.float32 and rad.shape == (len(a),)
        assert np.abs(np.cos(rad.double().cpu().numpy()) - np.cos(g[tag + "_none"].astype(np.float64))).max() < 1e-6
        assert rad.min().item() >= 4.8e-4 and rad.max().item() <= np.pi - 4.8e-4
        mean, total = rr.geodesic(da, db), rr.geodesic(da, db, "sum")
        assert mean.dtype == torch.float32 and mean.dim() == 0 and total.dim() == 0
        assert abs(mean.item() - float(g[tag + "_mean"])) < 2e-6 * float(g[tag + "_mean"]) + 1e-5
        assert abs(total.item() - float(g[tag + "_sum"])) < 2e-6 * float(g[tag + "_sum"]) + 3e-3
        assert abs(total.item() - rad.double().sum().item()) < 1e-3
    assert rr.geodesic(dev(g["a"]), dev(g["b"]), "median") is None
    gen = torch.Generator(device=DEV).manual_seed(14)
    big_a, big_b = _haar_rows(100_003, gen), _haar_rows(100_003, gen)
    for lo in (0, 1):                                                      # 16-byte aligned and a view 36 bytes in
        a, b = big_a[lo:], big_b[lo:]
        ref = so.geodesic_eps_np(a.cpu().numpy(), b.cpu().numpy(), "none").astype(np.float64)
        rad = rr.geodesic(a, b, "none").double().cpu().numpy()
        assert np.abs(np.cos(rad) - np.cos(ref)).max() < 1e-6
        assert abs(rr.geodesic(a, b, "sum").item() - ref.sum()) < 2e-6 * ref.sum() + 0.05
        assert abs(rr.geodesic(a, b, "mean").item() - ref.mean()) < 1e-5
    # the C ABI with and without the reduction workspace: one launch (the last workgroup writes sum and mean) against
    # memset + kernel + mean; the same float64 sum to its last bits' worth of ordering, twice in a row on one workspace
    from poseestimation_amd import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.zeros(lib.so3_reduce_workspace_bytes(), dtype=torch.uint8, device=DEV)
    for n in (100_003, 65_536, 1025):
        a, b = big_a[:n].contiguous(), big_b[:n].contiguous()
        got = []
        for w in (ws, ws, None):
            acc = torch.full((1,), float("nan"), dtype=torch.float64, device=DEV)
            out = torch.full((), float("nan"), dtype=torch.float32, device=DEV)
            rc = lib.so3_geodesic_eps_f32(a.data_ptr(), b.data_ptr(), None, acc.data_ptr(), out.data_ptr(), 1, 1e-7,
                                          w.data_ptr() if w is not None else None, n, st)
            assert rc == 0
            got.append((acc.item(), out.item()))
        assert got[0] == got[1]                                            # the workspace is left as it was found
        assert abs(got[0][0] - got[2][0]) < 1e-9 * got[2][0] and abs(got[0][1] - got[2][1]) < 1e-6
        assert abs(got[0][1] - np.float32(got[0][0] / n)) < 1e-7


def _haar_rows(n, gen):
    q = torch.randn(n, 4, device=DEV, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                        2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1)


# ------------------------------------------------------------------------------------------------
# K4b: the metrics are differentiable, like the reference's tensor code (G15)
# ------------------------------------------------------------------------------------------------
def _metric_call(rr, name, a, b):
    if name.startswith("geo_"):
        return rr.geodesic(a, b, name[4:])
    if name == "cgd":
        return rr.compute_geodesic_distance_from_two_matrices(a, b)
    out = rr.angle_error(a, b)
    return out.mean() if name == "ang_mean" else out


def test_g15_metric_gradients_against_the_reference(rr):
    """loss.backward() through geodesic / compute_geodesic_distance_from_two_matrices / angle_error against autograd through the
    reference's own functions (tools/gen_golden.py g15): both arguments, float32 and float64 tensors, per-row and reduced forms,
    pairs at 0 and 180 degrees (gradient exactly 0 on the clamp, as the reference's masked fill gives) and at 1e-4 rad.
    Tolerance: the slope's conditioning times a few float32 ulps of the cosine (conftest.metric_grad_check); float64 graphs 1e-12."""
    g = load_golden("g15_metric_gradients.npz")
    ambiguous = 0
    for tag in ("g3", "haar"):
        n = g[tag + "_r1"].shape[0]
        c64 = (np.einsum("bij,bij->b", g[tag + "_r1"].astype(np.float64), g[tag + "_r2"].astype(np.float64)) - 1) / 2
        for name, (eps, unit, div, up) in METRIC_GRAD_CASES.items():
            for dt, tdt in (("f32", torch.float32), ("f64", torch.float64)):
                key = "%s_%s_%s" % (tag, name, dt)
                a = dev(g[tag + "_r1"], tdt).requires_grad_(True)
                b = dev(g[tag + "_r2"], tdt).requires_grad_(True)
                y = _metric_call(rr, name, a, b)
                assert y.requires_grad and y.grad_fn is not None, key
                assert y.dtype == (torch.float64 if name.startswith("ang") else tdt), key
                ref_y = g[key + "_y"].astype(np.float64)
                got_y = y.detach().double().cpu().numpy()
                if name in ("geo_mean", "geo_sum", "ang_mean"):
                    assert abs(got_y - ref_y) <= 3e-6 * abs(ref_y) + 1e-5, key
                else:
                    assert np.abs(np.cos(got_y / unit) - np.cos(ref_y / unit)).max() < 1e-6, key
                if up == "w":
                    y.backward(dev(g[tag + "_w"], y.dtype))
                else:
                    y.backward()
                assert a.grad.dtype == tdt and a.grad.shape == a.shape and b.grad.shape == b.shape
                f64_graph = dt == "f64" or name.startswith("ang")
                ambiguous += metric_grad_check(a.grad.cpu().numpy(), b.grad.cpu().numpy(), g[key + "_d1"], g[key + "_d2"], c64, eps,
                                               1e-15 if f64_graph else 6e-7, 1e-12 if dt == "f64" else 4e-7, key)
                # rows 0, 1 of G3: 0 and 180 degrees -- finite everywhere, and 0 where the reference has 0
                assert torch.isfinite(a.grad).all() and torch.isfinite(b.grad).all(), key
                if tag == "g3":
                    for row in (0, 1):
                        if not np.any(g[key + "_d1"][row]):
                            assert not a.grad[row].any() and not b.grad[row].any(), (key, row)
    assert ambiguous < 40


def test_metric_gradients_on_large_ragged_batches_and_views(rr):
    """K4b on the streaming engine and its remainder kernel against the float64 closed form (oracle.metric_backward_np): 100 003
    rows (an odd number of 64-row units plus 35 rows), a view 36 bytes into its tensor, one gradient alone (either argument),
    an upstream gradient per row and one expanded by .sum() / .mean(), and the three spellings' units and clamps."""
    from oracle import so3_oracle as so
    gen = torch.Generator(device=DEV).manual_seed(150)
    n = 100_003
    big_a, big_b = _haar_rows(n + 1, gen), _haar_rows(n + 1, gen)
    big_b[5] = big_a[5]                                                   # 0 degrees
    big_b[6] = (big_a[6].view(3, 3) @ torch.diag(torch.tensor([1.0, -1.0, -1.0], device=DEV))).reshape(9)     # 180 degrees
    w = torch.randn(n + 1, device=DEV, generator=gen)
    for lo in (0, 1):
        an, bn, wn = big_a[lo:lo + n].cpu().numpy(), big_b[lo:lo + n].cpu().numpy(), w[lo:lo + n].cpu().numpy()
        for name, (eps, unit, div, up) in METRIC_GRAD_CASES.items():
            a = big_a[lo:lo + n].detach().requires_grad_(True)
            b = big_b[lo:lo + n].detach().requires_grad_(True)
            y = _metric_call(rr, name, a, b)
            if up == "w":
                y.backward(w[lo:lo + n].to(y.dtype))
            else:
                y.backward()
            f64_graph = name.startswith("ang")
            d1, d2, c = so.metric_backward_np(an, bn, wn if up == "w" else 1.0, eps=eps, unit=unit, divisor=float(n) if div == "B" else 1.0,
                                              clamp_dtype=np.float64 if f64_graph else np.float32)
            metric_grad_check(a.grad.cpu().numpy(), b.grad.cpu().numpy(), d1, d2, c, eps, 1e-15 if f64_graph else 6e-7, 4e-7, (name, lo))
            assert torch.isfinite(a.grad).all() and torch.isfinite(b.grad).all()
    # one gradient alone: the first argument's, then the second's (the library swaps the pair), bit for bit the two-gradient launch's
    a = big_a[:n].detach().requires_grad_(True)
    b = big_b[:n].detach().requires_grad_(True)
    rr.geodesic(a, b, "sum").backward()
    a1 = big_a[:n].detach().requires_grad_(True)
    rr.geodesic(a1, big_b[:n], "sum").backward()
    b1 = big_b[:n].detach().requires_grad_(True)
    rr.geodesic(big_a[:n], b1, "sum").backward()
    assert torch.equal(a1.grad, a.grad) and torch.equal(b1.grad, b.grad)
    # .sum() / .mean() on the per-row forms hand an EXPANDED gradient over: it travels as one element, and gives what the reduced forms give
    a2 = big_a[:n].detach().requires_grad_(True)
    rr.geodesic(a2, big_b[:n], "none").sum().backward()
    assert torch.equal(a2.grad, a.grad)
    a3 = big_a[:n].detach().requires_grad_(True)
    (3.0 * rr.angle_error(a3, big_b[:n]).mean()).backward()
    d1, _, c = so.metric_backward_np(big_a[:n].cpu().numpy(), big_b[:n].cpu().numpy(), 3.0, unit=180.0 / np.pi, divisor=float(n))
    metric_grad_check(a3.grad.cpu().numpy(), a3.grad.cpu().numpy(), d1, d1, c, 0.0, 1e-15, 4e-7, "3 * angle_error.mean()")
    # shapes, dtypes and the no-graph path
    x9 = big_a[:64].reshape(64, 9).detach().requires_grad_(True)
    rr.compute_geodesic_distance_from_two_matrices(x9.view(64, 3, 3), big_b[:64]).sum().backward()
    assert x9.grad.shape == (64, 9)
    h = big_a[:128].half().requires_grad_(True)
    rr.geodesic(h, big_b[:128]).backward()
    assert h.grad.dtype == torch.float16 and h.grad.shape == h.shape
    with torch.no_grad():
        assert rr.geodesic(a, b).grad_fn is None and not rr.angle_error(a, b).requires_grad
    assert not rr.geodesic(big_a[:n], big_b[:n]).requires_grad
    # the build's own fused evaluation calls are metrics only -- and say so ONCE when an argument requires grad, instead of dropping it silently
    rr._WARNED.clear()
    xq = torch.randn(256, 9, device=DEV, requires_grad=True)
    with pytest.warns(RuntimeWarning, match="evaluation call"):
        out = rr.head_angle_error(xq, big_b[:256].reshape(256, 3, 3))
    assert not out.requires_grad
    with pytest.warns(RuntimeWarning, match="angle_error\\(...\\).sum\\(\\)"):
        rr.angle_error_sum_count(a[:256], b[:256])
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                    # once per process: the second call is quiet; without grad it never speaks
        rr.head_angle_error(xq, big_b[:256].reshape(256, 3, 3))
        rr.head_angle_error(xq.detach(), big_b[:256].reshape(256, 3, 3))
    e = torch.zeros(0, 3, 3, device=DEV, requires_grad=True)
    rr.geodesic(e, torch.zeros(0, 3, 3, device=DEV), "sum").backward()
    assert e.grad.shape == (0, 3, 3)
    # a second derivative is refused loudly, never returned wrong: the backward's result is a constant of the graph
    with pytest.raises(RuntimeError, match="differentiate twice|does not require grad"):
        a4 = big_a[:64].detach().requires_grad_(True)
        (ga,) = torch.autograd.grad(rr.geodesic(a4, big_b[:64]), a4, create_graph=True)
        ga.sum().backward()


def test_a_geodesic_loss_trains_the_head_like_the_reference_graph(rr):
    """`lossfunc` = geodesic in the reference's loops (point_cloud/main.py:194-197, UPNA/main.py:56-59): head -> metric -> backward,
    against the same chain in float64 torch (the oracle's port of the head and autograd through the reference's expression)."""
    from oracle import so3_oracle as so
    torch.manual_seed(77)
    x = torch.randn(4096, 9)
    t = so.symmetric_orthogonalization_torch(torch.randn(4096, 9))
    xd = x.to(DEV).requires_grad_(True)
    loss = rr.loss_frobenius(t.to(DEV), rr.symmetric_orthogonalization(xd)) + 0.5 * rr.geodesic(t.to(DEV), rr.symmetric_orthogonalization(xd))
    loss.backward()
    xr = x.double().requires_grad_(True)
    rd = so.symmetric_orthogonalization_torch(xr)
    cos = ((t.double() @ rd.transpose(1, 2)).diagonal(dim1=-2, dim2=-1).sum(-1) - 1) / 2
    ref = so.loss_frobenius_torch(rd, t.double()) + 0.5 * torch.acos(torch.clamp(cos, -1 + 1e-7, 1 - 1e-7)).mean()
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-5
    err = (xd.grad.cpu().double() - xr.grad).abs().reshape(4096, -1).amax(1)
    scale = xr.grad.abs().reshape(4096, -1).amax(1)
    assert torch.median(err / scale) < 2e-6 and (err <= 2e-3 * scale + 1e-9).float().mean() > 0.995


def test_g4_backward_generic_gradient(rr):
    g = load_golden("g4_frobenius512.npz")
    x = torch.from_numpy(g["x_bf16_bits"]).view(torch.bfloat16).float().to(DEV).requires_grad_(True)
    rr.symmetric_orthogonalization(x).backward(dev(g["g"]))
    got = x.grad.cpu().numpy()
    ref = g["dx_g_f64"]
    rel = np.abs(got - ref).max(1) / (1e-3 + np.abs(ref).max(1))
    rel_ref = np.abs(g["dx_g"] - ref).max(1) / (1e-3 + np.abs(ref).max(1))   # the reference's float32 autograd
    assert np.median(rel) < 1e-6 and np.quantile(rel, 0.99) < 1e-4
    assert np.median(rel) <= 2 * np.median(rel_ref) + 1e-7


def test_g4_config4_fused_head_loss_backward(rr):
    g = load_golden("g4_frobenius512.npz")
    xb = torch.from_numpy(g["x_bf16_bits"]).view(torch.bfloat16).to(DEV)
    rt = dev(g["r_true"])
    # (a) bf16 storage end to end: gradient comes back in bf16
    x = xb.clone().requires_grad_(True)
    loss, r = rr.frobenius_head(x, rt)
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 2e-6
    assert abs(loss.item() - float(g["loss_f64"])) < 2e-6
    assert np.quantile(np.abs(r.cpu().numpy() - g["r"]), 0.99) < 5e-6
    assert x.grad.dtype == torch.bfloat16
    ref = g["dx_f64"]
    got = x.grad.float().cpu().numpy()
    assert np.median(np.abs(got - ref).max(1) / (1e-6 + np.abs(ref).max(1))) < 4e-3      # bf16 rounding: 2^-8
    # (b) float32 storage: tight against the float64 autograd of the reference
    xf = xb.float().requires_grad_(True)
    loss_f, _ = rr.frobenius_head(xf, rt)
    (2.0 * loss_f).backward()                                # upstream scale must be honoured
    got = xf.grad.cpu().numpy() / 2.0
    rel = np.abs(got - ref).max(1) / (1e-6 + np.abs(ref).max(1))
    assert np.median(rel) < 1e-6 and np.quantile(rel, 0.99) < 1e-4
    # (c) unfused path gives the same numbers
    xu = xb.float().requires_grad_(True)
    lu = rr.loss_frobenius(rt, rr.symmetric_orthogonalization(xu))
    lu.backward()
    assert abs(lu.item() - loss_f.item()) < 1e-6
    assert (xu.grad - xf.grad / 2.0).abs().max().item() < 1e-6 * max(1.0, float(np.abs(ref).max()))


def test_backward_matches_oracle_on_random_batch(rr, c_oracle):
    rng = np.random.default_rng(3)
    x = rng.standard_normal((5000, 9)).astype(np.float32)
    gup = rng.standard_normal((5000, 9)).astype(np.float32)
    xd = dev(x).requires_grad_(True)
    rr.symmetric_orthogonalization(xd).backward(dev(gup).view(-1, 3, 3))
    ref = c_oracle.project_bwd(x, gup).reshape(5000, 9)
    rel = np.abs(xd.grad.cpu().numpy() - ref).max(1) / (1e-3 + np.abs(ref).max(1))
    assert np.median(rel) < 1e-6 and np.quantile(rel, 0.99) < 2e-4
    # the gradient is orthogonal to M's radial direction: proj(M) is scale invariant
    radial = (xd.grad * xd.detach()).sum(1)
    assert radial.abs().max().item() < 1e-3 * xd.grad.abs().max().item()


# ------------------------------------------------------------------------------------------------
# K5: fused Kabsch (config #3 contract)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["6x1024", "24x64"])
def test_g5_kabsch_golden(rr, tag):
    g = load_golden("g5_kabsch_%s.npz" % tag)
    r, h = rr.kabsch_rotation(dev(g["p"]), dev(g["q"]), return_h=True)
    assert np.abs(h.cpu().numpy() - g["h"]).max() < 3e-6 * np.abs(g["h"]).max() + 1e-5
    assert np.abs(r.cpu().numpy() - g["r_f64"]).max() < 5e-6
    assert np.abs(r.cpu().numpy() - g["r"]).max() < 5e-6
    assert orth_err(r.cpu().numpy()).max() < 1e-5


@pytest.mark.parametrize("b,n", [(1, 1), (3, 7), (5, 64), (17, 65), (70, 200), (300, 1024), (4100, 256)])
def test_kabsch_ragged(rr, c_oracle, b, n):
    rng = np.random.default_rng(b * 1000 + n)
    p = (rng.random((b, n, 3)) - 0.5).astype(np.float32)
    q = (rng.random((b, n, 3)) - 0.5).astype(np.float32)
    r, h = rr.kabsch_rotation(dev(p), dev(q), return_h=True)
    ro, ho = c_oracle.kabsch(p, q, want_h=True)
    assert np.abs(h.cpu().numpy() - ho).max() < 1e-5 * max(1.0, np.abs(ho).max())
    assert orth_err(r.cpu().numpy()).max() < 1e-5
    if n >= 64:                                             # H well conditioned -> R comparable
        assert np.quantile(np.abs(r.cpu().numpy() - ro), 0.9) < 5e-5


def test_kabsch_degenerate_clouds(rr):
    """Coplanar clouds (rank-two H: R still unique thanks to det = +1) and collinear clouds (rank-one H: R is only
    determined on the line) -- the answer must be a rotation that reproduces the second cloud."""
    gen = torch.Generator(device=DEV).manual_seed(31)
    b, n = 4096, 257
    r_gt = rr.get_sampled_rotation_matrices_by_axisAngle(b, DEV)
    basis = rr.get_sampled_rotation_matrices_by_axisAngle(b, DEV)
    coef = torch.randn(b, n, 3, device=DEV, generator=gen)
    for rank in (2, 1):
        c = coef.clone()
        c[:, :, rank:] = 0.0
        p = torch.bmm(c, basis.transpose(1, 2))                       # points in a plane / on a line through the origin
        q = torch.bmm(p, r_gt.transpose(1, 2))
        r = rr.kabsch_rotation(p, q)
        cols = [(r[:, :, i] * r[:, :, j]).sum(1) - (1.0 if i == j else 0.0) for i in range(3) for j in range(3)]
        assert torch.stack(cols, 1).norm(dim=1).max().item() < 1e-5
        assert (torch.linalg.det(r.double()) - 1).abs().max().item() < 1e-5
        resid = (torch.bmm(p, r.transpose(1, 2)) - q).norm(dim=2).max().item()
        assert resid < 2e-5 * p.norm(dim=2).max().item(), (rank, resid)
        if rank == 2:
            assert (r - r_gt).abs().max().item() < 2e-5


def test_config3_kabsch_recovers_rotations(rr):
    """Round trip at a large size: Q = R_gt P (no noise) -> Kabsch returns R_gt."""
    b, n = 8192, 1024
    gen = torch.Generator(device=DEV).manual_seed(7)
    p = torch.rand(b, n, 3, device=DEV, generator=gen) - 0.5
    r_gt = rr.symmetric_orthogonalization(torch.randn(b, 9, device=DEV, generator=gen))
    q = torch.bmm(p, r_gt.transpose(1, 2))
    r = rr.kabsch_rotation(p, q)
    assert (r - r_gt).abs().max().item() < 5e-6
    via_head = rr.symmetric_orthogonalization(torch.bmm(q.transpose(1, 2), p))     # the unfused spelling
    assert (r - via_head).abs().max().item() < 5e-6


@pytest.mark.parametrize("sigma", [0.0, 0.01])
def test_config3_at_its_own_geometry_65536_clouds_of_1024(rr, c_oracle, sigma):
    """BASELINE config #3 as bench.py runs it: 65 536 clouds x 1024 points, i.e. 16 clouds per wave in `k_kabsch`
    (clouds_per_wave = B / (CUs * 16): the keep-loop that parks cloud j's H in lane j runs with j up to 15 here).
    Data contract: point_cloud/main.py:171-181 (q = R_gt p, no translation), rotations from prepare.py:21-49.
    Checked against the C oracle on EVERY cloud (max |R - R_ref| <= 5e-6)."""
    b, n = 65536, 1024
    gen = torch.Generator(device=DEV).manual_seed(7)
    p = torch.rand(b, n, 3, device=DEV, generator=gen) - 0.5
    r_gt = rr.get_sampled_rotation_matrices_by_axisAngle(b, DEV)
    q = torch.bmm(p, r_gt.transpose(1, 2))
    if sigma:
        q = q + sigma * torch.randn(b, n, 3, device=DEV, generator=gen)
    r, h = rr.kabsch_rotation(p, q, return_h=True)
    ro, ho = c_oracle.kabsch(p.cpu().numpy(), q.cpu().numpy(), want_h=True)
    assert np.abs(h.cpu().numpy() - ho).max() < 2e-5 * np.abs(ho).max()        # 1024-term float32 sums vs float64
    assert np.abs(r.cpu().numpy() - ro).max() < 5e-6
    assert orth_err(r.cpu().numpy()).max() < 1e-5
    if sigma == 0.0:
        assert (r - r_gt).abs().max().item() < 5e-6                              # the round trip
    # the same clouds with the second cloud made in the kernel (row f4): q = R_gt p + sigma n(seed, cloud, point)
    from oracle import so3_oracle as so
    rs, hs = rr.kabsch_rotation_synthetic(p, r_gt, sigma=sigma, seed=11, return_h=True)
    assert orth_err(rs.cpu().numpy()).max() < 1e-5
    if sigma == 0.0:
        assert (rs - r_gt).abs().max().item() < 5e-6 and (rs - r).abs().max().item() < 5e-6
    # the generator is stateless in (seed, cloud, point): restate it for the first and the last 48 clouds (three waves' worth)
    for lo in (0, b - 48):
        ids = np.arange(lo, lo + 48)
        pn = p[lo:lo + 48].cpu().numpy().astype(np.float64)
        qn = np.einsum("bac,bic->bia", r_gt[lo:lo + 48].cpu().numpy().astype(np.float64), pn)
        if sigma:
            cb, pi, cc = np.meshgrid(ids, np.arange(n), np.arange(3), indexing="ij")
            qn = qn + sigma * so.synth_normal_np(11, cb, pi, cc)
        href = so.cross_covariance_np(pn, qn)
        assert np.abs(hs[lo:lo + 48].cpu().numpy() - href).max() < 3e-5 * np.abs(href).max()
        assert np.abs(rs[lo:lo + 48].cpu().numpy() - so.symmetric_orthogonalization_np(href)).max() < 5e-6


@pytest.mark.parametrize("b,n", [(65536, 64), (262144, 64), (300_000, 33)])
def test_many_small_clouds_sixteen_and_sixty_four_per_wave(rr, c_oracle, b, n):
    """The per-wave loops of every cloud kernel at the geometries large batches select: 16 clouds (samples) per wave at
    B = 65 536 and the cap of 64 at B >= 262 144 (clouds_per_wave = B / (CUs * 16)); 300 000 x 33 adds a ragged last wave
    and a point count that is not a multiple of the lane count.  Everything against the CPU oracles on all clouds."""
    from oracle import so3_oracle as so
    gen = torch.Generator(device=DEV).manual_seed(b + n)
    p = torch.rand(b, n, 3, device=DEV, generator=gen) - 0.5
    r_gt = rr.get_sampled_rotation_matrices_by_axisAngle(b, DEV)
    pn, rn = p.cpu().numpy(), r_gt.cpu().numpy()
    # a7: pairing (rotate) and normalisation
    q = rr.rotate_point_clouds(p, r_gt)
    assert np.abs(q.cpu().numpy() - so.rotate_clouds_np(pn, rn)).max() < 2e-6
    qt = rr.rotate_point_clouds(p, r_gt, transposed=True)
    assert np.abs(qt.cpu().numpy() - so.rotate_clouds_np(pn, rn).transpose(0, 2, 1)).max() < 2e-6
    nn, cc, ss = rr.pc_normalize(p)
    on, oc, os_ = so.pc_normalize_np(pn)
    assert np.abs(nn.cpu().numpy() - on).max() < 2e-6 and np.abs(cc.cpu().numpy() - oc).max() < 1e-6
    assert np.abs(ss.cpu().numpy() - os_).max() < 2e-6
    # K5 and its synthesising variant
    r, h = rr.kabsch_rotation(p, q, return_h=True)
    ro, ho = c_oracle.kabsch(pn, q.cpu().numpy(), want_h=True)
    assert np.abs(h.cpu().numpy() - ho).max() < 1e-5 * np.abs(ho).max()
    assert np.quantile(np.abs(r.cpu().numpy() - ro), 0.999) < 5e-6 and (r - r_gt).abs().max().item() < 3e-5   # 33/64 points: H less well conditioned
    rs = rr.kabsch_rotation_synthetic(p, r_gt, sigma=0.0, seed=3)
    assert (rs - r).abs().max().item() < 3e-5
    # f6: ADD-L1 with one wave handling 16 / 64 samples
    def poses(rot):
        t = torch.eye(4, device=DEV).repeat(b, 1, 1)
        t[:, :3, :3] = rot
        t[:, :3, 3] = torch.randn(b, 3, device=DEV, generator=gen)
        return t
    t_gt, t_pred = poses(r_gt), poses(rr.get_sampled_rotation_matrices_by_axisAngle(b, DEV))
    d = rr.compute_ADD_L1_loss(t_gt, t_pred, p, use_batch_mean=False)
    pts_gt = torch.bmm(p.double(), t_gt[:, :3, :3].double().transpose(1, 2)) + t_gt[:, None, :3, 3].double()
    pts_pr = torch.bmm(p.double(), t_pred[:, :3, :3].double().transpose(1, 2)) + t_pred[:, None, :3, 3].double()
    ref = (pts_gt - pts_pr).abs().mean(dim=(1, 2))                              # Iterative/loss.py:24-25, float64
    assert (d.double() - ref).abs().max().item() < 3e-6


def test_g13_output_dtypes_follow_the_reference(rr):
    """loss_frobenius / compute_geodesic_distance_from_two_matrices / angle_error for float32 and float64 arguments:
    dtypes and values as the reference returned them (3D-Pose/loss.py:7-11, rotation_representation.py:209-242)."""
    g = load_golden("g13_dtype_fidelity.npz")
    for tag, dt, tol in (("f32", torch.float32, 2e-6), ("f64", torch.float64, 1e-12)):
        a = dev(g["r1"], dt).requires_grad_(True)
        b = dev(g["r2"], dt)
        loss = rr.loss_frobenius(a, b)
        loss.backward()
        geo = rr.compute_geodesic_distance_from_two_matrices(a.detach(), b)
        ang = rr.angle_error(a.detach(), b)
        assert [str(loss.dtype), str(a.grad.dtype), str(geo.dtype), str(ang.dtype)] == [str(x) for x in g["dtypes_" + tag]]
        assert abs(loss.item() - float(g["loss_" + tag])) < tol and np.abs(a.grad.cpu().numpy() - g["dloss_" + tag]).max() < tol
        # float32 acos near 0 and pi is conditioned like 1/sin(theta): judge the float32 path against the float64 answer
        gerr = np.abs(geo.cpu().numpy().astype(np.float64) - g["geo_f64"]) * np.sin(g["geo_f64"])
        assert gerr.max() < (2e-6 if dt == torch.float32 else 1e-12)
        assert np.abs(ang.cpu().numpy() - g["ang_" + tag]).max() < (1e-9 if dt == torch.float32 else 1e-9)
    # the float64 head's output keeps its precision through the loss and the fused spelling
    x = torch.randn(300, 9, device=DEV, dtype=torch.float64, requires_grad=True)
    t = dev(g["r2"], torch.float64)
    l2 = rr.loss_frobenius(rr.symmetric_orthogonalization(x), t)
    l3, r3 = rr.frobenius_head(x, t)
    assert l2.dtype == torch.float64 and l3.dtype == torch.float64 and r3.dtype == torch.float64 and abs(l2.item() - l3.item()) < 1e-14
    with pytest.raises(ValueError, match="angle out of range"):
        rr.angle_error(1.7 * t, t)


def test_fused_head_backward_twice_over_one_graph(rr):
    """loss.backward(retain_graph=True) twice with an upstream factor != 1: the stored gradient must not be scaled in
    place (the second pass would return dm * g * g) and the returned tensor must not alias it."""
    gen = torch.Generator(device=DEV).manual_seed(4)
    x = torch.randn(640, 9, device=DEV, generator=gen, requires_grad=True)
    t = rr.symmetric_orthogonalization(torch.randn(640, 9, device=DEV, generator=gen))
    loss, _ = rr.frobenius_head(x, t)
    (3.0 * loss).backward(retain_graph=True)
    g1 = x.grad.clone()
    x.grad = None
    (3.0 * loss).backward()
    assert torch.equal(g1, x.grad)
    x2 = x.detach().clone().requires_grad_(True)
    rr.loss_frobenius(t, rr.symmetric_orthogonalization(x2)).backward()
    assert (g1 - 3.0 * x2.grad).abs().max().item() < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_scale_by_a_device_scalar(rr, dtype):
    """so3_scale_*: dst = src * (*factor) with the factor in device memory -- what _FrobeniusHead.backward launches instead of
    torch's float() / mul / to(bfloat16) chain.  Bit-equal to that chain, for ragged lengths and unaligned views."""
    from poseestimation_amd import _lib
    lib = _lib.load()
    fn = lib.so3_scale_bf16 if dtype == torch.bfloat16 else lib.so3_scale_f32
    gen = torch.Generator(device=DEV).manual_seed(9)
    base = torch.randn(40_000, device=DEV, generator=gen).to(dtype)
    factor = torch.tensor(-0.37, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    for off, n in ((0, 4608), (0, 4607), (1, 4608), (3, 1), (0, 0), (2, 39_990)):
        src = base[off:off + n]
        dst = torch.full((n + 8,), 7.0, device=DEV).to(dtype)
        assert fn(src.data_ptr(), factor.data_ptr(), dst[3:].data_ptr(), n, st) == 0
        assert torch.equal(dst[3:3 + n], (src.float() * factor).to(dtype)) and (dst[:3].float() == 7).all() and (dst[3 + n:].float() == 7).all()
    # through autograd: an upstream factor that is not 1, bfloat16 and float32 storage
    x = torch.randn(512, 9, device=DEV, generator=gen).to(dtype).requires_grad_(True)
    t = rr.symmetric_orthogonalization(torch.randn(512, 9, device=DEV, generator=gen))
    loss, _ = rr.frobenius_head(x, t)
    (loss * 2.5).backward()
    x1 = x.detach().clone().requires_grad_(True)
    rr.frobenius_head(x1, t)[0].backward()
    assert x.grad.dtype == dtype and torch.equal(x.grad, (x1.grad.float() * 2.5).to(dtype))


@pytest.mark.parametrize("b", [512, 4000])
def test_fused_head_is_differentiable_in_the_target_too(rr, b):
    """The reference's loss is differentiable in both arguments (3D-Pose/loss.py:7-11: plain tensor arithmetic):
    frobenius_head(x, R_true) hands R_true its gradient -(R - R_true) / (B ||.||_F) like loss_frobenius does, with or
    without a gradient for x, and whether or not the rotation is asked for."""
    gen = torch.Generator(device=DEV).manual_seed(40 + b)
    x = torch.randn(b, 9, device=DEV, generator=gen)
    t = rr.symmetric_orthogonalization(torch.randn(b, 9, device=DEV, generator=gen))
    # the two-call spelling with torch's own autograd through the loss expression
    ta = t.clone().requires_grad_(True)
    xa = x.clone().requires_grad_(True)
    la = torch.linalg.matrix_norm(ta - rr.symmetric_orthogonalization(xa), ord="fro").mean()
    (2.0 * la).backward()
    for x_grad in (True, False):
        for want_r in (True, False):
            tb = t.clone().requires_grad_(True)
            xb = x.clone().requires_grad_(x_grad)
            out = rr.frobenius_head(xb, tb, return_rotation=want_r)
            loss = out[0] if want_r else out
            assert loss.requires_grad
            (2.0 * loss).backward()
            assert tb.grad is not None and tb.grad.shape == t.shape
            assert (tb.grad - ta.grad).abs().max().item() < 2e-7 * 512 / b + 1e-9, (x_grad, want_r)
            if x_grad:
                assert (xb.grad - xa.grad).abs().max().item() < 1e-5
            else:
                assert xb.grad is None
    # a target without requires_grad gets none (and costs nothing)
    xc = x.clone().requires_grad_(True)
    tc = t.clone()
    rr.frobenius_head(xc, tc)[0].backward()
    assert tc.grad is None and torch.equal(xc.grad * 2.0, xa.grad) or (xc.grad * 2.0 - xa.grad).abs().max().item() < 1e-5


@pytest.mark.parametrize("dtype,b", [(torch.bfloat16, 512), (torch.float32, 512), (torch.float32, 1), (torch.bfloat16, 3000), (torch.float32, 70_000)])
def test_cpp_autograd_node_equals_the_python_function(rr, dtype, b):
    """frobenius_head goes through csrc/autograd_node.cpp when the arguments are its case; the Python class _FrobeniusHead is
    what it replaces.  Same launches, so: loss, rotation and gradient bit for bit, with and without the rotation, with an upstream
    factor, twice over one graph; what the node declines (a strided x, float16, a target that wants its gradient) still works;
    backward of backward fails loudly."""
    if rr._node() is None:
        pytest.skip("_so3node not built")
    gen = torch.Generator(device=DEV).manual_seed(b)
    x = torch.randn(b, 9, device=DEV, generator=gen).to(dtype)
    t = rr.symmetric_orthogonalization(torch.randn(b, 9, device=DEV, generator=gen))
    for want_r in (True, False):
        for shape in ((b, 9), (b, 3, 3)):
            xn = x.clone().view(shape).requires_grad_(True)
            xp = x.clone().view(shape).requires_grad_(True)
            out = rr.frobenius_head(xn, t, return_rotation=want_r)
            assert "FrobeniusHeadNode" in (out[0] if want_r else out).grad_fn.name()
            box = []
            lp = rr._FrobeniusHead.apply(xp, t, want_r, box)
            ln = out[0] if want_r else out
            assert ln.dtype == torch.float32 and ln.dim() == 0 and torch.equal(ln, lp)
            if want_r:
                assert out[1].shape == (b, 3, 3) and not out[1].requires_grad and torch.equal(out[1], box[0])
            (ln * 1.5).backward(retain_graph=True)
            (lp * 1.5).backward()
            assert xn.grad.shape == shape and xn.grad.dtype == dtype and torch.equal(xn.grad, xp.grad)
            g1 = xn.grad.clone()
            xn.grad = None
            (ln * 1.5).backward()                                                  # the stored gradient was not scaled in place
            assert torch.equal(xn.grad, g1)
    # no gradient wanted: nothing saved, nothing returned
    ln, rn = rr.frobenius_head(x, t)
    box = []
    lp = rr._FrobeniusHead.apply(x, t, True, box)
    assert not ln.requires_grad and ln.grad_fn is None and torch.equal(ln, lp) and torch.equal(rn, box[0])
    # declined cases take the Python class and give the same numbers
    xs = torch.randn(b, 18, device=DEV, generator=gen).to(dtype)[:, ::2]           # strided
    xs.requires_grad_(True)
    ls, _ = rr.frobenius_head(xs, t)
    assert "FrobeniusHeadNode" not in ls.grad_fn.name()
    tg = t.clone().requires_grad_(True)
    xg = x.clone().requires_grad_(True)
    lg, _ = rr.frobenius_head(xg, tg)
    assert "FrobeniusHeadNode" not in lg.grad_fn.name()
    lg.backward()
    assert tg.grad is not None and xg.grad is not None
    # double backward is refused under once_differentiable's own condition -- grad mode on AND a cotangent that requires grad;
    # create_graph=True alone (a gradient penalty elsewhere in the graph) gives the ordinary gradient, as a constant
    for head in (lambda xx: rr.frobenius_head(xx, t)[0], lambda xx: rr._FrobeniusHead.apply(xx, t, True, [])):
        xd = x.clone().requires_grad_(True)
        (g_plain,) = torch.autograd.grad(head(xd), xd)
        (g_cg,) = torch.autograd.grad(head(xd), xd, create_graph=True)
        assert torch.equal(g_cg, g_plain) and not g_cg.requires_grad
        go = torch.ones((), device=DEV, requires_grad=True)
        with pytest.raises(RuntimeError, match="differentiate twice"):
            torch.autograd.grad(head(xd), xd, grad_outputs=go, create_graph=True)


@pytest.mark.parametrize("dtype,b", [(torch.bfloat16, 512), (torch.float32, 512), (torch.float32, 3), (torch.float32, 70_000)])
def test_cpp_nodes_of_the_two_call_spelling_equal_the_python_functions(rr, dtype, b):
    """The reference's own spelling -- out = symmetric_orthogonalization(out); loss = loss_frobenius(R, out); loss.backward()
    (3D-Pose/main.py:60,85,90) -- goes through csrc/autograd_node.cpp's ProjectNode and FrobLossNode; the Python classes are
    what they replace.  Same launches: rotation, loss and every gradient bit for bit, in either argument order of the loss."""
    if rr._node() is None:
        pytest.skip("_so3node not built")
    gen = torch.Generator(device=DEV).manual_seed(7 + b)
    x = torch.randn(b, 9, device=DEV, generator=gen).to(dtype)
    t = rr.symmetric_orthogonalization(torch.randn(b, 9, device=DEV, generator=gen))
    for swapped in (True, False):                                          # the reference passes (R, out): the prediction second
        xn = x.clone().requires_grad_(True)
        xp = x.clone().requires_grad_(True)
        rn = rr.symmetric_orthogonalization(xn)
        rp = rr._SymmetricOrthogonalization.apply(xp)
        assert "ProjectNode" in rn.grad_fn.name() and rn.shape == (b, 3, 3) and rn.dtype == torch.float32 and torch.equal(rn, rp)
        ln = rr.loss_frobenius(t, rn) if swapped else rr.loss_frobenius(rn, t)
        lp = rr._LossFrobenius.apply(t, rp) if swapped else rr._LossFrobenius.apply(rp, t)
        assert "FrobLossNode" in ln.grad_fn.name() and ln.dim() == 0 and ln.dtype == torch.float32 and torch.equal(ln, lp)
        (ln * 0.75).backward()
        (lp * 0.75).backward()
        assert xn.grad.dtype == dtype and xn.grad.shape == (b, 9) and torch.equal(xn.grad, xp.grad)
    # both arguments of the loss want their gradient; a (2,b/2..)-shaped head input; a non-contiguous upstream gradient of the head
    a = rr.symmetric_orthogonalization(x.float()).clone().requires_grad_(True)
    c = t.clone().requires_grad_(True)
    (rr.loss_frobenius(a, c) * 2.0).backward()
    a2, c2 = a.detach().clone().requires_grad_(True), c.detach().clone().requires_grad_(True)
    (rr._LossFrobenius.apply(a2, c2) * 2.0).backward()
    assert torch.equal(a.grad, a2.grad) and torch.equal(c.grad, c2.grad) and torch.equal(c.grad, -a.grad)
    xs = x.float().clone().requires_grad_(True)
    w = torch.randn(3, 3, b, device=DEV, generator=gen).permute(2, 0, 1)                                   # strided cotangent
    rr.symmetric_orthogonalization(xs).backward(w)
    xs2 = x.float().clone().requires_grad_(True)
    rr._SymmetricOrthogonalization.apply(xs2).backward(w)
    assert torch.equal(xs.grad, xs2.grad)
    # declined: float16 / float64 / strided input, no gradient wanted, double backward refused
    assert "ProjectNode" not in rr.symmetric_orthogonalization(x.half().requires_grad_(True)).grad_fn.name()
    assert rr.symmetric_orthogonalization(x).grad_fn is None
    xd = x.float().clone().requires_grad_(True)
    (g_plain,) = torch.autograd.grad(rr.symmetric_orthogonalization(xd).sum(), xd)
    (g_cg,) = torch.autograd.grad(rr.symmetric_orthogonalization(xd).sum(), xd, create_graph=True)      # .sum()'s backward hands over a constant
    assert torch.equal(g_cg, g_plain) and not g_cg.requires_grad
    go = torch.ones((), device=DEV, requires_grad=True)
    with pytest.raises(RuntimeError, match="differentiate twice"):
        torch.autograd.grad(rr.symmetric_orthogonalization(xd).sum(), xd, grad_outputs=go, create_graph=True)
    with pytest.raises(RuntimeError, match="differentiate twice"):
        torch.autograd.grad(rr.loss_frobenius(t, a), a, grad_outputs=go, create_graph=True)
    # a gradient penalty on ANOTHER branch of the graph: the heads' backward runs under create_graph=True and must not object
    w = torch.randn(9, device=DEV, generator=gen, requires_grad=True)
    xd2 = x.float().clone().requires_grad_(True)
    total = rr.loss_frobenius(t, rr.symmetric_orthogonalization(xd2)) + ((xd2 * w).sum()) ** 2
    (gx,) = torch.autograd.grad(total, xd2, create_graph=True)
    assert gx.requires_grad                                               # through the penalty branch only
    gx.pow(2).sum().backward()
    assert w.grad is not None and torch.isfinite(w.grad).all()


@pytest.mark.parametrize("name,width,cls", [("compute_rotation_matrix_from_ortho6d", 6, "_Ortho6d"), ("compute_rotation_matrix_from_quaternion", 4, "_Quat"),
                                            ("compute_rotation_matrix_from_euler", 3, "_Euler"), ("compute_rotation_matrix_from_ortho5d", 5, "_Ortho5d"),
                                            ("so3_exp_map", 3, "_ExpMap")])
def test_cpp_node_of_the_row_heads_equals_the_python_functions(rr, name, width, cls):
    """The heads that are plain row operations share csrc/autograd_node.cpp's RowHeadNode; the Python classes are what it replaces
    (and still serve float16 / strided / no-grad input).  Same launches: rotation and gradient bit for bit."""
    if rr._node() is None:
        pytest.skip("_so3node not built")
    gen = torch.Generator(device=DEV).manual_seed(width)
    x = torch.randn(777, width, device=DEV, generator=gen)
    w = torch.randn(777, 3, 3, device=DEV, generator=gen)
    xn, xp = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    rn, rp = getattr(rr, name)(xn), getattr(rr, cls).apply(xp)
    assert "RowHeadNode" in rn.grad_fn.name() and rn.shape == (777, 3, 3) and torch.equal(rn, rp)
    (rn * w).sum().backward()
    (rp * w).sum().backward()
    assert torch.equal(xn.grad, xp.grad)
    assert "RowHeadNode" not in getattr(rr, name)(x.half().requires_grad_(True)).grad_fn.name()
    assert getattr(rr, name)(x).grad_fn is None
    if width == 6:                                                          # the 6D head takes (..., 6)
        x3 = x[:776].reshape(2, 388, 6).clone().requires_grad_(True)
        r3 = rr.compute_rotation_matrix_from_ortho6d(x3)
        assert r3.shape == (2, 388, 3, 3) and torch.equal(r3.reshape(-1, 3, 3), rp[:776])
        r3.sum().backward()
        assert x3.grad.shape == (2, 388, 6)


@pytest.mark.parametrize("dtype,b", [(torch.bfloat16, 512), (torch.float32, 512), (torch.float32, 1000), (torch.bfloat16, 3000)])
def test_recorded_training_step_matches_the_autograd_spelling(rr, dtype, b):
    """FrobeniusHeadStep (one hipGraph replay: config #4's launch-bound step) against frobenius_head + backward; sizes on
    both sides of the one-workgroup kernel's limit (1024 rows)."""
    gen = torch.Generator(device=DEV).manual_seed(b)
    x = torch.randn(b, 9, device=DEV, generator=gen).to(dtype)
    t = rr.symmetric_orthogonalization(torch.randn(b, 9, device=DEV, generator=gen))
    step = rr.FrobeniusHeadStep(b, dtype=dtype, device=DEV)
    for rep in range(2):                                                         # replayed twice: buffers are reused
        step.x.copy_(x)
        step.r_true.copy_(t)
        loss, dx, r = step()
        xa = x.clone().requires_grad_(True)
        la, ra = rr.frobenius_head(xa, t)
        la.backward()
        assert loss.dtype == torch.float32 and loss.dim() == 0 and abs(loss.item() - la.item()) < 1e-6
        assert dx.dtype == dtype and (dx.float() - xa.grad.float()).abs().max().item() <= (0 if dtype == torch.float32 else 1e-3)
        assert (r - ra).abs().max().item() == 0
        x = -x                                                                   # new data for the second replay


def test_fused_evaluation_on_an_offset_view(rr):
    """head_angle_error on a contiguous view whose base pointer is not 16-byte aligned (x[1:65]: 64 rows, 36 B off)."""
    gen = torch.Generator(device=DEV).manual_seed(6)
    x = torch.randn(200, 9, device=DEV, generator=gen)
    t = rr.symmetric_orthogonalization(torch.randn(200, 9, device=DEV, generator=gen))
    for lo, hi in ((1, 65), (3, 131), (0, 64)):
        d = rr.head_angle_error(x[lo:hi], t[lo:hi])
        ref = rr.angle_error(rr.symmetric_orthogonalization(x[lo:hi]), t[lo:hi])
        assert (d - ref).abs().max().item() < 1e-9
    with pytest.raises(ValueError, match="reduce must be"):
        rr.head_angle_error(x[:64], t[:64], reduce="median")


# ------------------------------------------------------------------------------------------------
# the C ABI itself: streams, nullable outputs, error codes
# ------------------------------------------------------------------------------------------------
def test_c_abi_direct_on_side_stream(pa, c_oracle):
    from poseestimation_amd import _lib
    lib = _lib.load()
    x = torch.randn(10_000, 9, device=DEV)
    r = torch.empty(10_000, 9, device=DEV)
    flip = torch.empty(10_000, dtype=torch.uint8, device=DEV)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        rc = lib.so3_project_fwd_f32(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(r.data_ptr()),
                                     ctypes.c_void_p(flip.data_ptr()), 10_000, ctypes.c_void_p(side.cuda_stream))
    assert rc == 0
    side.synchronize()
    ref, f = c_oracle.project(x.cpu().numpy(), want_flip=True)
    assert np.quantile(np.abs(r.cpu().numpy().reshape(-1, 3, 3) - ref), 0.99) < 2e-6
    assert np.array_equal(flip.cpu().numpy().astype(bool), f)
    # sum_count / range_flag are zeroed by the call itself: call twice, get the same answer
    sc = torch.full((2,), 123.0, dtype=torch.float64, device=DEV)
    fl = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    for _ in range(2):
        rc = lib.so3_angle_error_v2(ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(r.data_ptr()), None, ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(fl.data_ptr()), None, 0, 10_000, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    assert sc[1].item() == 10_000 and fl.item() == 0 and sc[0].item() / 10_000 < 0.2


@pytest.mark.parametrize("n", [1025, 4096 + 37, 1_000_000, 1_000_003])
def test_workspace_reductions_match_the_atomic_path_and_repeat_bit_for_bit(rr, n):
    """so3_*_ws entry points (include/so3proj.h): a caller-owned workspace replaces the zero-fill launch and the atomics on
    the result.  Same sums as the entry points without it (to round-off: the order of the additions differs), the SAME BITS
    from call to call (the atomics' order varies, the ticket's fixed-order sum does not), the float32 mean written by the
    kernel, the range flag raised and cleared, and the workspace left zeroed -- on sizes with and without a remainder."""
    from poseestimation_amd import _lib
    lib = _lib.load()
    gen = torch.Generator(device=DEV).manual_seed(n)
    x = torch.randn(n, 9, device=DEV, generator=gen)
    t = rr.symmetric_orthogonalization(torch.randn(n, 9, device=DEV, generator=gen)).reshape(n, 9)
    r = torch.empty(n, 9, device=DEV)
    dm = torch.empty(n, 9, device=DEV)
    ws = torch.zeros(lib.so3_reduce_workspace_bytes(), dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    p = lambda a: a.data_ptr() if a is not None else None
    f64 = lambda k: torch.full((k,), 777.0, dtype=torch.float64, device=DEV)

    # K3: fused head + loss + backward
    ls0, r0, dm0 = f64(1), torch.empty_like(r), torch.empty_like(dm)
    assert lib.so3_frob_fwd_bwd_v2_f32(p(x), p(t), p(r0), p(dm0), p(ls0), None, None, 0, n, st) == 0
    runs = []
    for _ in range(3):
        ls, mean = f64(1), torch.full((), -1.0, device=DEV)
        assert lib.so3_frob_fwd_bwd_v2_f32(p(x), p(t), p(r), p(dm), p(ls), p(mean), p(ws), 0, n, st) == 0
        runs.append((ls.item(), mean.item()))
        assert torch.equal(r, r0) and torch.equal(dm, dm0)
    assert runs[0] == runs[1] == runs[2]
    assert abs(runs[0][0] - ls0.item()) <= 1e-12 * ls0.item()
    assert runs[0][1] == float(np.float32(runs[0][0] * (1.0 / n)))
    # ... and without a workspace the mean comes from the finishing launch
    ls, mean = f64(1), torch.full((), -1.0, device=DEV)
    assert lib.so3_frob_fwd_bwd_v2_f32(p(x), p(t), None, p(dm), p(ls), p(mean), None, 0, n, st) == 0
    assert mean.item() == float(np.float32(ls.item() * (1.0 / n))) and abs(ls.item() - ls0.item()) <= 1e-12 * ls0.item()

    # K3': stand-alone loss
    ls0 = f64(1)
    assert lib.so3_frob_loss_v2_f32(p(r0), p(t), p(dm0), p(ls0), None, None, 0, n, st) == 0
    runs = []
    for _ in range(2):
        ls, mean = f64(1), torch.full((), -1.0, device=DEV)
        assert lib.so3_frob_loss_v2_f32(p(r0), p(t), p(dm), p(ls), p(mean), p(ws), 0, n, st) == 0
        runs.append((ls.item(), mean.item()))
        assert torch.equal(dm, dm0)
    assert runs[0] == runs[1] and abs(runs[0][0] - ls0.item()) <= 1e-12 * ls0.item()
    assert runs[0][1] == float(np.float32(runs[0][0] * (1.0 / n)))

    # K4 and K1+K4: (sum, count), the flag, per-row angles untouched by the way the sum is formed
    sc0, fl0, deg0 = f64(2), torch.full((1,), 9, dtype=torch.int32, device=DEV), f64(n)
    assert lib.so3_angle_error_v2(p(r0), p(t), p(deg0), p(sc0), p(fl0), None, 0, n, st) == 0
    for fused in (False, True):
        runs = []
        for _ in range(2):
            sc, fl, deg = f64(2), torch.full((1,), 9, dtype=torch.int32, device=DEV), f64(n)
            if fused:
                assert lib.so3_project_angle_error_v2_f32(p(x), p(t), p(r), p(deg), p(sc), p(fl), p(ws), 4, n, st) == 0
            else:
                assert lib.so3_angle_error_v2(p(r0), p(t), p(deg), p(sc), p(fl), p(ws), 0, n, st) == 0
            runs.append(sc.tolist())
            assert torch.equal(deg, deg0) and fl.item() == 0 and sc[1].item() == n
        assert runs[0] == runs[1] and abs(runs[0][0] - sc0[0].item()) <= 1e-12 * sc0[0].item()
        # sum only, flag only
        sc = f64(2)
        if fused:
            assert lib.so3_project_angle_error_v2_f32(p(x), p(t), p(r), None, p(sc), None, p(ws), 4, n, st) == 0
        else:
            assert lib.so3_angle_error_v2(p(r0), p(t), None, p(sc), None, p(ws), 0, n, st) == 0
        assert sc.tolist() == runs[0]
    # accumulators zeroed by the caller (so3_*_acc): one launch, the same numbers up to the order of the atomics
    for fused in (False, True):
        sc, fl = torch.zeros(2, dtype=torch.float64, device=DEV), torch.zeros(1, dtype=torch.int32, device=DEV)
        if fused:
            assert lib.so3_project_angle_error_v2_f32(p(x), p(t), p(r), None, p(sc), p(fl), None, 6, n, st) == 0
        else:
            assert lib.so3_angle_error_v2(p(r0), p(t), None, p(sc), p(fl), None, 2, n, st) == 0
        assert sc[1].item() == n and fl.item() == 0 and abs(sc[0].item() - sc0[0].item()) <= 1e-12 * sc0[0].item()
    bad = t.clone()
    bad[n - 3] = 3.0 * r0[n - 3]                                 # a row of the remainder when there is one: tr = 9, cosine 4
    bad[5] = 3.0 * r0[5]
    for rows in (bad, t):                                         # raised, then cleared again by the next call
        fl = torch.full((1,), 9, dtype=torch.int32, device=DEV)
        assert lib.so3_angle_error_v2(p(r0), p(rows), None, None, p(fl), p(ws), 0, n, st) == 0
        assert fl.item() == (1 if rows is bad else 0)
        fl = torch.zeros(1, dtype=torch.int32, device=DEV)
        assert lib.so3_angle_error_v2(p(r0), p(rows), None, None, p(fl), None, 2, n, st) == 0
        assert fl.item() == (1 if rows is bad else 0)
    torch.cuda.synchronize()
    assert int(torch.count_nonzero(ws).item()) == 0              # slots, flag and ticket are left as they were found


def test_graph_capture_of_the_head(pa):
    """Enqueue-only contract: the C-ABI launches can be captured into a hipGraph and replayed."""
    from poseestimation_amd import _lib
    lib = _lib.load()
    x = torch.randn(4096, 9, device=DEV)
    r = torch.zeros(4096, 9, device=DEV)
    sc = torch.zeros(2, dtype=torch.float64, device=DEV)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        assert lib.so3_project_fwd_f32(ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(r.data_ptr()), None, 4096, st) == 0
        assert lib.so3_angle_error_v2(ctypes.c_void_p(r.data_ptr()), ctypes.c_void_p(r.data_ptr()), None, ctypes.c_void_p(sc.data_ptr()), None, None, 0, 4096, st) == 0
    x.normal_()
    graph.replay()
    torch.cuda.synchronize()
    assert orth_err(r.cpu().numpy()).max() < 1e-5 and sc[1].item() == 4096


def test_loss_frobenius_standalone_matches_definition(rr):
    """a3: mean_b ||R_true - R_pred||_F and both gradients against the 3-line definition (3D-Pose/loss.py:7-11)."""
    gen = torch.Generator(device=DEV).manual_seed(21)
    p = torch.randn(1000, 3, 3, device=DEV, generator=gen, requires_grad=True)
    t = torch.randn(1000, 3, 3, device=DEV, generator=gen, requires_grad=True)
    loss = rr.loss_frobenius(p, t)
    (3.0 * loss).backward()
    p2 = p.detach().clone().requires_grad_(True)
    t2 = t.detach().clone().requires_grad_(True)
    ref = (t2 - p2).reshape(-1, 9).norm(dim=1).mean()          # the definition, composed from torch ops
    (3.0 * ref).backward()
    assert abs(loss.item() - ref.item()) < 1e-6
    assert (p.grad - p2.grad).abs().max().item() < 1e-8 and (t.grad - t2.grad).abs().max().item() < 1e-8
    # a zero difference gives a zero gradient instead of the reference's NaN (documented)
    q = torch.eye(3, device=DEV).repeat(4, 1, 1).requires_grad_(True)
    rr.loss_frobenius(q, torch.eye(3, device=DEV).repeat(4, 1, 1)).backward()
    assert torch.isfinite(q.grad).all() and q.grad.abs().max().item() == 0


# ------------------------------------------------------------------------------------------------
# next row f2: the 6D Gram-Schmidt head (rotation_representation.py:21-36)
# ------------------------------------------------------------------------------------------------
def test_g7_ortho6d_head_forward_backward(rr, pa):
    from oracle import so3_oracle as so
    g = load_golden("g7_ortho6d.npz")
    p = dev(g["p"]).requires_grad_(True)
    r = rr.compute_rotation_matrix_from_ortho6d(p)
    assert tuple(r.shape) == (300, 3, 3)
    assert np.abs(r.detach().cpu().numpy() - g["r_f64"]).max() < 2e-6
    assert np.abs(r.detach().cpu().numpy() - g["r"]).max() < 2e-6        # vs the reference's float32 output
    r.backward(dev(g["g"]))
    ref = g["dp_f64"]
    rel = np.abs(p.grad.cpu().numpy() - ref).max(1) / (1e-3 + np.abs(ref).max(1))
    assert np.median(rel) < 1e-6 and rel.max() < 2e-4
    # shape rule (..., 6) -> (..., 3, 3), dispatch table, ragged sizes through the remainder kernel
    rs = rr.compute_rotation_matrix_from_ortho6d(dev(g["p_shaped"]))
    assert tuple(rs.shape) == (2, 5, 3, 3) and np.abs(rs.cpu().numpy() - g["r_shaped"]).max() < 2e-6
    assert pa.transform_output["6D"][0] == 6 and pa.transform_output["6D"][1] is rr.compute_rotation_matrix_from_ortho6d
    with pytest.raises(AssertionError):
        rr.compute_rotation_matrix_from_ortho6d(torch.zeros(4, 9, device=DEV))
    gen6 = torch.Generator(device=DEV).manual_seed(66)
    for b in (1, 63, 64, 65, 1000, 100_003):
        x = torch.randn(b, 6, device=DEV, generator=gen6)
        out = rr.compute_rotation_matrix_from_ortho6d(x).cpu().numpy()
        # Gram-Schmidt is as well conditioned as the two 3-vectors are far from parallel: scale the error by sin(angle)
        xn = x.cpu().numpy().astype(np.float64)
        sin = np.linalg.norm(np.cross(xn[:, :3], xn[:, 3:]), axis=1) / (np.linalg.norm(xn[:, :3], axis=1) * np.linalg.norm(xn[:, 3:], axis=1))
        err = np.abs(out - so.ortho6d_np(x.cpu().numpy())).reshape(b, -1).max(1)
        assert (err * sin).max() < 2e-6 and np.median(err) < 5e-7
        assert orth_err(out).max() < 1e-5
    big = torch.randn(1_000_000, 6, device=DEV, requires_grad=True)
    rb = rr.compute_rotation_matrix_from_ortho6d(big)
    rb.backward(torch.randn(1_000_000, 3, 3, device=DEV))
    assert torch.isfinite(big.grad).all() and (torch.linalg.det(rb.detach().double()) - 1).abs().max().item() < 1e-5


# ------------------------------------------------------------------------------------------------
# next row f5: the quaternion / Euler / 5D / exp-map heads (rotation_representation.py:39-171, 245-321)
# ------------------------------------------------------------------------------------------------
HEADS = {"quat": (4, "compute_rotation_matrix_from_quaternion"), "euler": (3, "compute_rotation_matrix_from_euler"),
         "ortho5d": (5, "compute_rotation_matrix_from_ortho5d"), "expmap": (3, "vec_3d_to_SO3")}


@pytest.mark.parametrize("name", sorted(HEADS))
def test_g10_heads_forward_backward(rr, name):
    from oracle import so3_oracle as so
    n, fn_name = HEADS[name]
    fn = getattr(rr, fn_name)
    g = load_golden("g10_heads.npz")
    x = dev(g[name + "_x"]).requires_grad_(True)
    r = fn(x)
    assert tuple(r.shape) == (192, 3, 3) and r.dtype == torch.float32
    r64 = so.head_np(name, g[name + "_x"])
    assert np.abs(r.detach().cpu().numpy() - r64).max() < 2e-6           # vs the float64 restatement
    assert np.abs(r.detach().cpu().numpy() - g[name + "_r"]).max() < 5e-6  # vs the reference's float32 output
    r.backward(dev(g[name + "_g"]))
    ref = so.head_backward_np(name, g[name + "_x"], g[name + "_g"])
    scale = np.maximum(np.abs(ref).max(axis=1), 1.0)
    err = np.abs(x.grad.cpu().numpy() - ref).max(axis=1) / scale
    ref_err = np.abs(g[name + "_dx"] - ref).max(axis=1) / scale          # how far the reference's own float32 autograd is
    assert np.median(err) < 2e-6 and err.max() < max(2e-5, 2.0 * ref_err.max()), (err.max(), ref_err.max())
    # ragged sizes: streaming units + the one-row-per-thread remainder; unaligned views
    for b in (1, 63, 64, 65, 1000, 100_003):
        xb = torch.randn(b, n, device=DEV, requires_grad=True)
        out = fn(xb)
        gb = torch.randn(b, 3, 3, device=DEV)
        out.backward(gb)
        e = np.abs(out.detach().cpu().numpy() - so.head_np(name, xb.detach().cpu().numpy())).reshape(b, -1).max(1)
        # float32 conditioning: rows whose two Gram-Schmidt vectors are nearly parallel (5D) amplify round-off
        assert np.median(e) < 3e-7 and np.quantile(e, 0.999) < 5e-6 and e.max() < 5e-4, (np.median(e), e.max())
        refb = so.head_backward_np(name, xb.detach().cpu().numpy(), gb.cpu().numpy())
        sc = np.maximum(np.abs(refb).max(axis=1), 1.0)
        assert (np.abs(xb.grad.cpu().numpy() - refb).max(axis=1) / sc).max() < 5e-4
    base = torch.randn(1001 * n + 1, device=DEV)
    odd = base[1:].view(1001, n)                                           # 4-byte aligned only
    assert np.abs(fn(odd).cpu().numpy() - so.head_np(name, odd.cpu().numpy())).max() < 5e-6
    big = torch.randn(1_000_000, n, device=DEV, requires_grad=True)
    rb = fn(big)
    rb.backward(torch.randn(1_000_000, 3, 3, device=DEV))
    assert torch.isfinite(big.grad).all()
    tol = 1e-5 if name != "expmap" else 1e-4                               # exp map below the clamp is only nearly orthogonal
    assert (torch.linalg.det(rb.detach().double()) - 1).abs().max().item() < tol


def test_head_tables_and_argument_errors(rr, pa):
    assert set(pa.transform_output) == {"SVD", "6D", "3D"} and pa.transform_output["3D"] == (3, rr.vec_3d_to_SO3)
    assert pa.head_dimensions == {"SVD": 9, "6D": 6, "5D": 5, "Quat": 4, "Euler": 3, "Direct": 9}
    for key, width in pa.head_dimensions.items():
        if key == "Direct":
            continue
        out = pa.head_functions[key](torch.randn(10, width, device=DEV))
        assert tuple(out.shape) == (10, 3, 3) and orth_err(out.cpu().numpy()).max() < 1e-5
    assert pa.head_functions["quat"] is pa.head_functions["Quat"]
    with pytest.raises(ValueError, match="Nx3"):
        rr.so3_exp_map(torch.zeros(4, 4, device=DEV))
    with pytest.raises(RuntimeError):
        rr.compute_rotation_matrix_from_quaternion(torch.zeros(4, 3, device=DEV))
    with pytest.raises(RuntimeError):
        rr.compute_rotation_matrix_from_euler(torch.zeros(3, device=DEV))
    # identity cases
    eye = np.eye(3)
    assert np.abs(rr.compute_rotation_matrix_from_quaternion(torch.tensor([[2.0, 0, 0, 0]], device=DEV)).cpu().numpy()[0] - eye).max() == 0
    assert np.abs(rr.compute_rotation_matrix_from_euler(torch.zeros(1, 3, device=DEV)).cpu().numpy()[0] - eye).max() == 0
    assert np.abs(rr.vec_3d_to_SO3(torch.zeros(1, 3, device=DEV)).cpu().numpy()[0] - eye).max() == 0


# ------------------------------------------------------------------------------------------------
# row a7: cloud pairing (point_cloud/main.py:173-183) and pc_normalize (point_cloud/prepare.py:51-56)
# ------------------------------------------------------------------------------------------------
def test_g12_cloud_pairing_and_normalisation(rr):
    from oracle import so3_oracle as so
    g = load_golden("g12_clouds.npz")
    out = rr.rotate_point_clouds(dev(g["pc1"]), dev(g["gt_rmat"]))
    assert tuple(out.shape) == (6, 200, 3) and np.abs(out.cpu().numpy() - g["pc_out"]).max() < 3e-7
    outt = rr.rotate_point_clouds(dev(g["pc1"]), dev(g["gt_rmat"]), transposed=True)
    assert tuple(outt.shape) == (6, 3, 200) and outt.is_contiguous() and np.abs(outt.cpu().numpy() - g["gg"]).max() < 3e-7
    n, c, s = rr.pc_normalize(dev(g["clouds"]))
    assert np.abs(n.cpu().numpy() - g["norm"]).max() < 3e-7 and np.abs(c.cpu().numpy() - g["centroid"]).max() < 3e-7
    assert np.abs(s.cpu().numpy() - g["scale"]).max() < 5e-7
    n1, c1, s1 = rr.pc_normalize(dev(g["clouds"][2]))                       # the reference's single-cloud call
    assert tuple(n1.shape) == (200, 3) and tuple(c1.shape) == (3,) and s1.dim() == 0
    assert np.abs(n1.cpu().numpy() - g["norm"][2]).max() < 3e-7
    for b, npts in ((1, 1), (3, 63), (2, 64), (5, 1000), (4099, 70), (256, 1024), (3, 3001)):   # 3001 > 1024: the two-pass variant
        p = torch.randn(b, npts, 3, device=DEV)
        r = rr.get_sampled_rotation_matrices_by_axisAngle(b, DEV)
        ref = so.rotate_clouds_np(p.cpu().numpy(), r.cpu().numpy())
        assert np.abs(rr.rotate_point_clouds(p, r).cpu().numpy() - ref).max() < 2e-6
        assert np.abs(rr.rotate_point_clouds(p, r, transposed=True).cpu().numpy() - ref.transpose(0, 2, 1)).max() < 2e-6
        nn, cc, ss = rr.pc_normalize(p)
        rn, rc, rs = so.pc_normalize_np(p.cpu().numpy())
        if npts > 1:                                                         # a single point has a zero box: 0/0 as in numpy
            assert np.abs(nn.cpu().numpy() - rn).max() < 2e-6 and np.abs(ss.cpu().numpy() - rs).max() < 2e-6 * max(1.0, rs.max())
        assert np.abs(cc.cpu().numpy() - rc).max() < 1e-6
    # the loop's pairing feeds Kabsch: rotate, then recover the rotation
    p = torch.rand(512, 1024, 3, device=DEV) - 0.5
    r = rr.get_sampled_rotation_matrices_by_axisAngle(512, DEV)
    assert (rr.kabsch_rotation(p, rr.rotate_point_clouds(p, r)) - r).abs().max().item() < 5e-6


# ------------------------------------------------------------------------------------------------
# next row f6: the ADD-L1 losses after calculate_T_pred (Iterative/loss.py:10-70)
# ------------------------------------------------------------------------------------------------
def test_g11_add_l1_losses_and_gradients(rr):
    from oracle import so3_oracle as so
    g = load_golden("g11_add_l1.npz")
    tg, pts = dev(g["t_gt"]), dev(g["points"])
    tp = dev(g["t_pred"]).requires_grad_(True)
    loss = rr.compute_ADD_L1_loss(tg, tp, pts)
    loss.backward()
    assert loss.dtype == torch.float32 and loss.dim() == 0
    assert abs(loss.item() - float(g["add_f64"])) < 2e-7 and abs(loss.item() - float(g["add"])) < 1e-6
    assert np.abs(tp.grad.cpu().numpy() - g["add_grad_f64"]).max() < 2e-7
    d = rr.compute_ADD_L1_loss(tg, tp.detach(), pts, use_batch_mean=False)
    assert tuple(d.shape) == (24,) and np.abs(d.cpu().numpy() - g["add_dists_f64"]).max() < 2e-7
    # per-sample mode is differentiable too: weights w_b -> sum_b w_b dist_b
    tp2 = dev(g["t_pred"]).requires_grad_(True)
    w = torch.linspace(0.5, 2.0, 24, device=DEV)
    (rr.compute_ADD_L1_loss(tg, tp2, pts, use_batch_mean=False) * w).sum().backward()
    assert np.abs(tp2.grad.cpu().numpy() - 24 * g["add_grad_f64"] * w.cpu().numpy()[:, None, None]).max() < 5e-6
    tp3 = dev(g["t_pred"]).requires_grad_(True)
    ldis = rr.compute_disentangled_ADD_L1_loss(tp3, tg, pts)
    ldis.backward()
    assert abs(ldis.item() - float(g["dis_f64"])) < 2e-7 and abs(ldis.item() - float(g["dis"])) < 1e-6
    assert np.abs(tp3.grad.cpu().numpy() - g["dis_grad_f64"]).max() < 2e-7
    assert np.all(tp3.grad[0].cpu().numpy() == 0)                               # exact hit: sgn(0) = 0, as autograd
    with pytest.raises(AssertionError):
        rr.compute_ADD_L1_loss(tg, tp.detach()[:, :3], pts)
    with pytest.raises(AssertionError):
        rr.compute_disentangled_ADD_L1_loss(tp.detach(), tg, pts[..., :2])


@pytest.mark.parametrize("b,n", [(1, 1), (3, 63), (5, 64), (7, 513), (300, 1024), (4097, 100)])
def test_add_l1_ragged_sizes_against_oracle(rr, b, n):
    from oracle import so3_oracle as so
    gen = torch.Generator().manual_seed(b * 1000 + n)
    def poses():
        t = torch.eye(4).repeat(b, 1, 1)
        t[:, :3, :3] = so.symmetric_orthogonalization_torch(torch.randn(b, 9, generator=gen))
        t[:, :3, 3] = torch.randn(b, 3, generator=gen)
        return t
    t_gt, t_pred, pts = poses(), poses(), torch.randn(b, n, 3, generator=gen)
    for dis in (False, True):
        ref_loss, ref_grad, _ = so.add_l1_np(t_gt.numpy(), t_pred.numpy(), pts.numpy(), disentangled=dis)
        tp = t_pred.to(DEV).requires_grad_(True)
        loss = rr.compute_disentangled_ADD_L1_loss(tp, t_gt.to(DEV), pts.to(DEV)) if dis else rr.compute_ADD_L1_loss(t_gt.to(DEV), tp, pts.to(DEV))
        loss.backward()
        assert abs(loss.item() - ref_loss) < 3e-6 * max(1.0, abs(ref_loss))
        # a coordinate difference within float32 round-off of its kink may take the other sign: allow a handful of points
        err = np.abs(tp.grad.cpu().numpy() - ref_grad)
        assert err.max() < 3e-6 + 8.0 * 3.0 / (3 * n * b), err.max()


def test_iterative_step_head_update_loss_backward(rr):
    """The refiner's training step end to end (Iterative/main.py:88-99): network output -> calculate_T_pred ->
    disentangled ADD-L1 -> backward to the network output, all through the library; against float64 autograd."""
    from oracle import so3_oracle as so
    gen = torch.Generator().manual_seed(5)
    b, n = 48, 300
    out = torch.randn(b, 12, generator=gen)
    out[:, 11] = 1.0 + 0.05 * torch.randn(b, generator=gen)
    t_init = torch.eye(4).repeat(b, 1, 1)
    t_init[:, :3, :3] = so.symmetric_orthogonalization_torch(torch.randn(b, 9, generator=gen))
    t_init[:, :3, 3] = torch.tensor([0.0, 0.0, 2.0]) + 0.2 * torch.randn(b, 3, generator=gen)
    t_gt = t_init.clone()
    t_gt[:, :3, 3] += 0.05 * torch.randn(b, 3, generator=gen)
    pts = 0.2 * torch.randn(b, n, 3, generator=gen)
    o = out.to(DEV).requires_grad_(True)
    t_pred = rr.calculate_T_pred(o, t_init.to(DEV), DEV)
    loss = rr.compute_disentangled_ADD_L1_loss(t_pred, t_gt.to(DEV), pts.to(DEV))
    loss.backward()
    fx, fy = rr.get_scene_parameters()
    od = out.double().requires_grad_(True)
    tp64 = so.se3_update_torch(od, t_init.double(), fx, fy)
    l64 = so.add_l1_disentangled_torch(tp64, t_gt.double(), pts.double())
    l64.backward()
    assert abs(loss.item() - l64.item()) < 2e-6
    ref = od.grad.numpy()
    assert np.abs(o.grad.cpu().numpy() - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


# ------------------------------------------------------------------------------------------------
# next row f3: per-class evaluation statistics (3D-Pose/test_per_class.py:174-216)
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,ncls", [(1, 1), (2, 1), (1000, 1), (1001, 3), (250_000, 10), (1_000_000, 10),
                                    (100_001, 16), (100_001, 17), (300_000, 40), (300_000, 64)])   # > 16 classes: the 128-KB histograms
def test_angle_error_statistics_match_numpy(rr, n, ncls):
    from oracle import so3_oracle as so
    rng = np.random.default_rng(n + ncls)
    ang = np.abs(rng.standard_normal(n)) * 25.0
    ang[rng.integers(0, n, max(1, n // 50))] = 0.0                      # ties and exact zeros
    ang = np.minimum(ang, 180.0)
    cls = rng.integers(0, ncls, n) if ncls > 1 else None
    got = rr.angle_error_statistics(dev(ang, torch.float64), None if cls is None else dev(cls, torch.int32), ncls)
    ref = so.angle_statistics_np(ang, cls, ncls)
    for k in ("count", "max", "median", "acc30", "acc15", "acc7.5"):    # exact quantities
        assert np.array_equal(got[k].cpu().numpy(), ref[k]), k
    assert np.abs(got["mean"].cpu().numpy() - ref["mean"]).max() < 1e-10
    assert np.abs(got["std"].cpu().numpy() - ref["std"]).max() < 1e-8
    # sliced views: deg[1:] is 8 bytes, cls[1:] 4 bytes off the alignment of the kernel's 16- / 8-byte loads (a peeled first
    # row), and views whose offsets disagree take scalar loads -- the same exact answers
    ad, cd = dev(ang, torch.float64), None if cls is None else dev(cls, torch.int32)
    for lo_a, lo_c in ((1, 1), (1, 0), (2, 1), (3, 3)):
        m = n - 3
        if m < 1:
            continue
        a_view = ad[lo_a:lo_a + m]
        c_view = None if cd is None else cd[lo_c:lo_c + m]
        got = rr.angle_error_statistics(a_view, c_view, ncls)
        ref = so.angle_statistics_np(ang[lo_a:lo_a + m], None if cls is None else cls[lo_c:lo_c + m], ncls)
        for k in ("count", "max", "median", "acc30", "acc15", "acc7.5"):
            assert np.array_equal(got[k].cpu().numpy(), ref[k], equal_nan=True), (k, lo_a, lo_c)
        assert np.allclose(got["mean"].cpu().numpy(), ref["mean"], rtol=0, atol=1e-10, equal_nan=True)


@pytest.mark.parametrize("case", ["one class of a million", "two classes of a million each", "a third of the rows tie at the median",
                                  "every row the same", "all rows inside one bin", "one bin, two classes, NaN in one"])
def test_angle_error_statistics_far_candidates_and_overflow(rr, case):
    """The exact median where the candidates of a class do not fit the finishing workgroup's LDS (one class of 1M rows: 44 000 rows in its
    middle 1/16-octave bin -- they move into LDS once the first digit has thinned them out) and where a collecting workgroup's region
    overflows (ties, a distribution narrower than a bin: the finishing workgroup then selects over the rows themselves, and again
    moves to LDS when few enough are left): bit-equal to np.median every time."""
    from oracle import so3_oracle as so
    rng = np.random.default_rng(len(case))
    n, ncls = 1_000_000, 1
    if case == "one class of a million":
        ang = np.minimum(np.abs(rng.standard_normal(n)) * 25.0, 180.0)
    elif case == "two classes of a million each":
        n, ncls = 2_000_000, 2
        ang = np.minimum(np.abs(rng.standard_normal(n)) * 25.0, 180.0)
    elif case == "a third of the rows tie at the median":
        ang = np.minimum(np.abs(rng.standard_normal(n)) * 25.0, 180.0)
        ang[rng.integers(0, n, n // 3)] = float(np.median(ang))
    elif case == "every row the same":
        ang = np.full(n, 12.5)
    elif case == "all rows inside one bin":
        ang = 20.0 + 1e-3 * rng.random(n)                              # 16 < x < 17: one bin of the window, every digit but the last few alike
    else:
        n, ncls = 1_000_001, 2
        ang = 20.0 + 1e-9 * rng.random(n)
        ang[5] = np.nan
    cls = rng.integers(0, ncls, n) if ncls > 1 else None
    if case.endswith("NaN in one"):
        cls[5] = 1
    got = rr.angle_error_statistics(dev(ang, torch.float64), None if cls is None else dev(cls, torch.int32), ncls)
    ref = so.angle_statistics_np(ang, cls, ncls)
    for k in ("count", "max", "median", "acc30", "acc15", "acc7.5"):
        assert np.array_equal(got[k].cpu().numpy(), ref[k], equal_nan=True), (k, got[k], ref[k])
    assert np.allclose(got["mean"].cpu().numpy(), ref["mean"], rtol=0, atol=1e-9, equal_nan=True)


_STAT_ACC_BYTES = 4 * 64 * 4 * 8                   # StatWork (csrc/so3proj.hip): acc[replicas][classes][4] float64 ...
_STAT_HIST_AT = _STAT_ACC_BYTES + 16 + 64 * 4      # ... overflow, ticket, 2 spare words, class_cursor[64]; then the histograms


def test_angle_error_statistics_leave_their_workspace_zeroed(rr):
    """include/so3proj.h: the caller zero-fills so3_angle_stats's workspace ONCE and every call leaves its counted part zeroed -- sums and
    histograms (all replicas, both bin widths); the control words between them (overflow flag, ticket, cursors: 272 bytes) are
    re-initialised by every call's first launch, and the candidate buffer behind is scratch.  One workspace
    through few classes (bins of 1/64 octave), many (1/16), the 64-class histograms, NaN rows, an overflowing staging area (every row the
    same) and back: all zeros after every call, and the answers do not depend on what ran before."""
    from oracle import so3_oracle as so
    from poseestimation_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(77)
    total = lib.so3_angle_stats_workspace_bytes()
    counted = total - 9 * (1 << 20)                                    # tag (1 B) and cand (8 B) of 2^20 candidates close the layout
    assert 0 < counted < total
    work = torch.zeros(total, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    n = 300_000
    cases = [(1, "normal"), (3, "nan"), (10, "normal"), (4, "same"), (40, "normal"), (64, "nan"), (2, "normal"), (10, "same"), (1, "normal")]
    for ncls, kind in cases:
        ang = np.minimum(np.abs(rng.standard_normal(n)) * 25.0, 180.0)
        if kind == "same":
            ang[:] = 12.5
        if kind == "nan":
            ang[rng.integers(0, n, 7)] = np.nan
        cls = rng.integers(0, ncls, n).astype(np.int32)
        a, c = dev(ang, torch.float64), dev(cls, torch.int32)
        stats = torch.empty(ncls, 8, dtype=torch.float64, device=DEV)
        assert lib.so3_angle_stats(a.data_ptr(), c.data_ptr(), ncls, stats.data_ptr(), work.data_ptr(), n, st) == 0
        torch.cuda.synchronize()
        assert int(torch.count_nonzero(work[:_STAT_ACC_BYTES]).item()) == 0 and int(torch.count_nonzero(work[_STAT_HIST_AT:counted]).item()) == 0, (ncls, kind)
        ref = so.angle_statistics_np(ang, cls, ncls)
        got = stats.cpu().numpy()
        for i, k in enumerate(("count", "mean", "std", "max", "median", "acc30", "acc15", "acc7.5")):
            if k in ("mean", "std"):
                assert np.allclose(got[:, i], ref[k], rtol=0, atol=1e-9, equal_nan=True), (ncls, kind, k)
            else:
                assert np.array_equal(got[:, i], ref[k], equal_nan=True), (ncls, kind, k)


def test_angle_stats_survives_timed_out_waits():
    """The advisor's round-5 finding: a finishing workgroup of so3_angle_stats whose launch-mates do not arrive within its wait
    finishes its class over the rows themselves -- and round 5 then cleared ticket / overflow / cursors while the late mates were
    still to come, so the NEXT call on that workspace started from a non-zero ticket and could trust a half-filled candidate buffer.
    The library built with SO3_STAT_SPINS=0 and four workgroups per CU (poseestimation_amd/libso3proj_spins0.so, build.TEST_VARIANTS)
    makes every wait time out at once and three quarters of the launch start after the finishing workgroups have cleared their
    classes: several calls in a row on ONE workspace, exact medians every time, sums and histograms zero after every call -- then
    the shipped library on the same workspace, and once more with the control words deliberately scribbled on (they are the
    call's own to initialise)."""
    from oracle import so3_oracle as so
    from poseestimation_amd import _lib, build
    path = os.path.join(os.path.dirname(build.LIB), "libso3proj_spins0.so")
    if not os.path.exists(path):
        build.build_test_variant("spins0", build.TEST_VARIANTS["spins0"])          # (hipcc is on the GPU box too)
    slow = ctypes.CDLL(path)
    slow.so3_angle_stats.restype = ctypes.c_int
    slow.so3_angle_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    assert slow.so3_version() == _lib.ABI_VERSION
    lib = _lib.load()
    rng = np.random.default_rng(91)
    total = lib.so3_angle_stats_workspace_bytes()
    counted = total - 9 * (1 << 20)
    work = torch.zeros(total, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    plan = [(slow, 10, 1_000_000), (slow, 10, 1_000_000), (slow, 1, 300_000), (slow, 40, 300_000), (lib, 10, 1_000_000), (slow, 3, 50_000),
            ("scribble", 0, 0), (lib, 10, 1_000_000), (slow, 64, 200_000), (lib, 1, 1_000_000)]
    for which, ncls, n in plan:
        if which == "scribble":
            work[_STAT_ACC_BYTES:_STAT_HIST_AT] = 0xA5
            continue
        ang = np.minimum(np.abs(rng.standard_normal(n)) * 25.0, 180.0)
        cls = rng.integers(0, ncls, n).astype(np.int32)
        a, c = dev(ang, torch.float64), dev(cls, torch.int32)
        stats = torch.full((ncls, 8), float("nan"), dtype=torch.float64, device=DEV)
        assert which.so3_angle_stats(a.data_ptr(), c.data_ptr(), ncls, stats.data_ptr(), work.data_ptr(), n, st) == 0
        torch.cuda.synchronize()
        assert int(torch.count_nonzero(work[:_STAT_ACC_BYTES]).item()) == 0 and int(torch.count_nonzero(work[_STAT_HIST_AT:counted]).item()) == 0
        ref = so.angle_statistics_np(ang, cls, ncls)
        got = stats.cpu().numpy()
        for i, k in enumerate(("count", "mean", "std", "max", "median", "acc30", "acc15", "acc7.5")):
            if k in ("mean", "std"):
                assert np.allclose(got[:, i], ref[k], rtol=0, atol=1e-9, equal_nan=True), (ncls, n, k)
            else:
                assert np.array_equal(got[:, i], ref[k], equal_nan=True), (which is slow, ncls, n, k)


def test_angle_error_statistics_end_to_end_and_nan(rr):
    """K1 -> K4 -> statistics without leaving the device, against the oracle chain; NaN propagates like numpy."""
    from oracle import so3_oracle as so
    gen = torch.Generator(device=DEV).manual_seed(31)
    r1 = rr.symmetric_orthogonalization(torch.randn(50_000, 9, device=DEV, generator=gen))
    r2 = rr.symmetric_orthogonalization(torch.randn(50_000, 9, device=DEV, generator=gen))
    cls = torch.randint(0, 10, (50_000,), device=DEV, generator=gen)
    deg = rr.angle_error(r1, r2)
    got = rr.angle_error_statistics(deg, cls, 10)
    ref = so.angle_statistics_np(deg.cpu().numpy(), cls.cpu().numpy(), 10)
    assert np.array_equal(got["median"].cpu().numpy(), ref["median"])
    assert np.abs(got["mean"].cpu().numpy() - ref["mean"]).max() < 1e-10
    bad = deg.clone()
    bad[123] = float("nan")
    s = rr.angle_error_statistics(bad, None, 1)
    assert torch.isnan(s["median"]).all() and torch.isnan(s["mean"]).all() and s["count"].item() == 50_000


# ------------------------------------------------------------------------------------------------
# robustness: host threads, large batches, every dtype path of the engine at a non-trivial size
# ------------------------------------------------------------------------------------------------
def test_reentrant_from_two_host_threads(rr, c_oracle):
    """No global mutable state: two host threads on their own streams get their own correct answers."""
    import threading
    xs = [torch.randn(200_000, 9, device=DEV, generator=torch.Generator(device=DEV).manual_seed(s)) for s in (1, 2)]
    outs = [None, None]

    def work(i):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            for _ in range(5):
                outs[i] = rr.symmetric_orthogonalization(xs[i])
        st.synchronize()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(2):
        ref = c_oracle.project(xs[i].cpu().numpy())
        err, scaled = conditioned_err(xs[i].cpu().numpy(), outs[i].cpu().numpy(), ref)
        assert np.median(err) < 2e-7 and scaled.max() < 2e-6


def test_config5_shard_size_sixteen_million_rows(rr):
    """Config #5's global batch (16M rows) on one GPU: properties that do not need the oracle at this size."""
    n = 16_000_000
    gen = torch.Generator(device=DEV).manual_seed(5)
    x = torch.randn(n, 9, device=DEV, generator=gen)
    r, flip = rr.symmetric_orthogonalization_with_flip(x)
    det_neg = torch.linalg.det(x.view(-1, 3, 3).double()) < 0
    assert int((flip != det_neg).sum()) == 0                                   # flip == (det M < 0), all 16M rows
    rtr = torch.bmm(r.transpose(1, 2), r)
    assert (rtr - torch.eye(3, device=DEV)).flatten(1).norm(dim=1).max().item() < 1e-5
    del rtr
    assert (rr.symmetric_orthogonalization(r) - r).abs().max().item() < 3e-6    # idempotence
    sc = rr.angle_error_sum_count(r, r)
    assert sc[1].item() == n and sc[0].item() / n < 0.05                        # ~0 degrees up to acos noise


def test_offsets_past_four_gigabytes(rr, c_oracle):
    """120M rows: each array is 4.32 GB, so unit byte offsets cross 2^32.  Checked in windows (start, around the
    2^32-byte boundary, end) against the C oracle, and by the device-side properties on every row."""
    n = 120_000_001                                         # odd: the last unit is a 1-row remainder for the tile kernel
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = torch.empty(n, 9, device=DEV)
    for lo in range(0, n, 20_000_000):                      # fill in slabs: randn of 1e9 elements at once needs no extra 4 GB
        hi = min(n, lo + 20_000_000)
        x[lo:hi] = torch.randn(hi - lo, 9, device=DEV, generator=gen)
    r, flip = rr.symmetric_orthogonalization_with_flip(x)
    boundary = (1 << 32) // 36
    for lo in (0, boundary - 500, n - 1000):
        xs = x[lo:lo + 1000].cpu().numpy()
        ref = c_oracle.project(xs)
        err = np.abs(r[lo:lo + 1000].cpu().numpy() - ref).reshape(1000, -1).max(1)
        assert np.median(err) < 5e-7 and err.max() < 1e-3, (lo, err.max())
        assert np.array_equal(flip[lo:lo + 1000].cpu().numpy(), np.linalg.det(xs.reshape(-1, 3, 3).astype(np.float64)) < 0)
    worst, mismatches = 0.0, 0
    # Orthogonality element-wise instead of torch.bmm(r^T, r): probed on this image (torch 2.10 + ROCm 7.0 rocBLAS), a batched
    # 3x3 float32 bmm works up to a batch of 2^24 = 16 777 216 and at 20 000 000 the process dies with SIGABRT after
    # "GPU core dump created" (a memory fault inside the library's batched-GEMM kernel; nothing of libso3proj is running).
    # Callers of the reference's bmm-based metrics at this scale have to chunk the batch the same way.
    for lo in range(0, n, 20_000_000):
        blk = r[lo:lo + 20_000_000]
        e = torch.zeros(blk.shape[0], device=DEV)
        for i in range(3):
            for j in range(i, 3):
                d = (blk[:, :, i] * blk[:, :, j]).sum(1) - (1.0 if i == j else 0.0)
                e += (1.0 if i == j else 2.0) * d * d
        worst = max(worst, e.sqrt().max().item())
        m = x[lo:lo + 20_000_000].view(-1, 3, 3).double()
        det = (m[:, 0, 0] * (m[:, 1, 1] * m[:, 2, 2] - m[:, 1, 2] * m[:, 2, 1]) - m[:, 0, 1] * (m[:, 1, 0] * m[:, 2, 2] - m[:, 1, 2] * m[:, 2, 0])
               + m[:, 0, 2] * (m[:, 1, 0] * m[:, 2, 1] - m[:, 1, 1] * m[:, 2, 0]))
        mismatches += int(((det < 0) != flip[lo:lo + 20_000_000]).sum())
        del blk, e, m, det
    assert worst < 1e-5 and mismatches == 0
    sc = rr.angle_error_sum_count(r, r)
    assert sc[1].item() == n and sc[0].item() / n < 0.05
    del x, r, flip
    torch.cuda.empty_cache()


def test_engine_dtype_paths_at_size(rr, c_oracle):
    """bf16 in / bf16 gradients and the flip variant through the streaming engine (not the remainder kernels)."""
    n = 100_000 + 37                                                           # 1562 units + a 69-row remainder
    gen = torch.Generator(device=DEV).manual_seed(9)
    x = torch.randn(n, 9, device=DEV, generator=gen)
    xb = x.bfloat16()
    ref = c_oracle.project(xb.float().cpu().numpy())
    r = rr.symmetric_orthogonalization(xb)
    err, scaled = conditioned_err(xb.float().cpu().numpy(), r.cpu().numpy(), ref)
    assert np.median(err) < 2e-7 and scaled.max() < 2e-6
    g = torch.randn(n, 3, 3, device=DEV, generator=gen)
    xg = xb.clone().requires_grad_(True)
    rr.symmetric_orthogonalization(xg).backward(g)
    gref = c_oracle.project_bwd(xb.float().cpu().numpy(), g.cpu().numpy()).reshape(n, 9)
    rel = np.abs(xg.grad.float().cpu().numpy() - gref).max(1) / (1e-3 + np.abs(gref).max(1))
    assert xg.grad.dtype == torch.bfloat16 and np.median(rel) < 4e-3           # bf16 rounding of the gradient
    xf = xb.float().requires_grad_(True)
    rr.symmetric_orthogonalization(xf).backward(g)
    rel = np.abs(xf.grad.cpu().numpy() - gref).max(1) / (1e-3 + np.abs(gref).max(1))
    assert np.median(rel) < 1e-6 and np.quantile(rel, 0.99) < 2e-4
    rt = rr.symmetric_orthogonalization(torch.randn(n, 9, device=DEV, generator=gen))
    xl = xb.float().requires_grad_(True)
    loss, rf = rr.frobenius_head(xl, rt)
    loss.backward()
    xu = xb.float().requires_grad_(True)
    lu = rr.loss_frobenius(rt, rr.symmetric_orthogonalization(xu))
    lu.backward()
    assert abs(loss.item() - lu.item()) < 2e-6 and (xl.grad - xu.grad).abs().max().item() < 1e-7
    # K3 hands out the rotation of its Jacobi frames, K1 the quaternion fast path's: two algorithms, equal up to the
    # conditioning of the row (s1 / gap) times float32 round-off
    sv = np.linalg.svd(xb.float().cpu().numpy().reshape(n, 3, 3).astype(np.float64), compute_uv=False)
    flips = np.linalg.det(xb.float().cpu().numpy().reshape(n, 3, 3).astype(np.float64)) < 0
    gap = np.where(flips, sv[:, 1] - sv[:, 2], sv[:, 1] + sv[:, 2]) / sv[:, 0]
    diff = (rf - r).abs().flatten(1).amax(1).cpu().numpy()
    assert (diff * gap).max() < 3e-6 and np.median(diff) < 3e-7


# ------------------------------------------------------------------------------------------------
# next row f1: the SE(3) pose update of the iterative refiner (Iterative/utility.py:90-128)
# ------------------------------------------------------------------------------------------------
def test_g8_se3_update_forward_backward(rr):
    from oracle import so3_oracle as so
    g = load_golden("g8_se3_update.npz")
    out = dev(g["out"]).requires_grad_(True)
    tp = rr.calculate_T_pred(out, dev(g["t_init"]), DEV)
    assert tuple(tp.shape) == (200, 4, 4)
    # the reference never reads `rot_repr` (Iterative/utility.py:90-105 always runs the SVD head): neither do we
    for name in ("6D", "Quat", "anything"):
        assert torch.equal(rr.calculate_T_pred(out.detach(), dev(g["t_init"]), DEV, rot_repr=name), tp.detach())
    # the rotation block inherits the head's conditioning (s1 / gap of the 3x3 in out[:, :9]): judge the error scaled by it
    m9 = g["out"][:, :9].astype(np.float64).reshape(-1, 3, 3)
    sv = np.linalg.svd(m9, compute_uv=False)
    gap = np.where(np.linalg.det(m9) < 0, sv[:, 1] - sv[:, 2], sv[:, 1] + sv[:, 2]) / sv[:, 0]
    e32 = np.abs(tp.detach().cpu().numpy() - g["t_pred"]).reshape(200, -1).max(1)             # vs the reference (float32)
    e64 = np.abs(tp.detach().cpu().numpy() - so.se3_update_np(g["out"], g["t_init"])).reshape(200, -1).max(1)
    assert (e32 * gap).max() < 1e-5 and (e64 * gap).max() < 5e-6 and np.median(e64) < 1e-6
    tp.backward(dev(g["g"]))
    ref = so.se3_update_backward_np(g["out"], g["t_init"], g["g"])
    rel = np.abs(out.grad.cpu().numpy() - ref).max(1) / (1e-3 + np.abs(ref).max(1))
    rel_ref = np.abs(g["dout"] - ref).max(1) / (1e-3 + np.abs(ref).max(1))                    # the reference's float32 autograd
    assert np.median(rel) < 1e-6 and rel.max() < 1e-4 and np.median(rel) <= 2 * np.median(rel_ref) + 1e-7
    # ragged sizes (remainder kernel), extra network outputs beyond 12 columns get zero gradient
    rng = np.random.default_rng(0)
    for b in (1, 65, 1000, 100_003):
        o = rng.standard_normal((b, 14)).astype(np.float32)
        o[:, 11] = 1.0 + 0.1 * o[:, 11]
        t = np.zeros((b, 4, 4), np.float32)
        t[:, :3, :3] = so.symmetric_orthogonalization_np(rng.standard_normal((b, 9)))
        t[:, :3, 3] = [0, 0, 2.5] + 0.3 * rng.standard_normal((b, 3))
        t[:, 3, 3] = 1
        od = dev(o).requires_grad_(True)
        res = rr.calculate_T_pred(od, dev(t), DEV)
        ref_t = so.se3_update_np(o[:, :12], t)
        assert np.quantile(np.abs(res.detach().cpu().numpy() - ref_t), 0.999) < 5e-6
        res.sum().backward()
        assert od.grad[:, 12:].abs().max().item() == 0 and torch.isfinite(od.grad).all()


def test_orthogonality_holds_for_ill_conditioned_input(rr):
    """s = (1, 10^-k2, +-10^-k3), k2 in [0,7]: R stays a rotation to 1e-5 however badly M is conditioned
    (the final Gram-Schmidt steps, not the sweeps, guarantee it); see tests/manual/illcond_check.py."""
    from oracle import so3_oracle as so
    rng = np.random.default_rng(0)
    n = 200_000
    u = so.symmetric_orthogonalization_np(rng.standard_normal((n, 9)))
    v = so.symmetric_orthogonalization_np(rng.standard_normal((n, 9)))
    k2 = rng.uniform(0, 7, n)
    k3 = k2 + rng.uniform(0, 3, n)
    s = np.stack([np.ones(n), 10.0 ** -k2, np.where(rng.random(n) < 0.5, -1.0, 1.0) * 10.0 ** -k3], 1)
    m = (u * s[:, None, :]) @ v.transpose(0, 2, 1)
    r = rr.symmetric_orthogonalization(dev(m.reshape(n, 9))).cpu().numpy()
    assert np.isfinite(r).all() and orth_err(r).max() < 1e-5
    assert np.abs(np.linalg.det(r.astype(np.float64)) - 1).max() < 1e-5
    well = k2 < 1.5
    assert np.quantile(np.abs(r[well] - so.symmetric_orthogonalization_np(m[well].astype(np.float32))), 0.999) < 2e-5


# ------------------------------------------------------------------------------------------------
# next row f4: on-device pair synthesis for Kabsch (point_cloud/prepare.py:21-49, main.py:173-181)
# ------------------------------------------------------------------------------------------------
def test_g9_sampler_kernel_and_synthesised_kabsch(rr, c_oracle):
    from oracle import so3_oracle as so
    g = load_golden("g9_sampler.npz")
    r = rr.rotations_from_axis_angle_draws(dev(g["theta"]), dev(g["axis"])).cpu().numpy()
    assert np.abs(r - g["r"]).max() < 2e-6                                  # vs the reference sampler's own output
    rs = rr.get_sampled_rotation_matrices_by_axisAngle(10_000, DEV).cpu().numpy()
    assert orth_err(rs).max() < 1e-5 and np.abs(np.linalg.det(rs.astype(np.float64)) - 1).max() < 1e-5
    # synthesised pairs: H and R against the oracle's restatement of the generator, noise-free and noisy
    rng = np.random.default_rng(4)
    for b, n, sigma in ((7, 64, 0.0), (5, 1024, 0.0), (33, 200, 0.02), (300, 1024, 0.01)):
        p = (rng.random((b, n, 3)) - 0.5).astype(np.float32)
        rg = so.symmetric_orthogonalization_np(rng.standard_normal((b, 9))).astype(np.float32)
        rk, h = rr.kabsch_rotation_synthetic(dev(p), dev(rg), sigma, 1234, return_h=True)
        q = so.synth_pairs_np(p, rg, sigma, 1234)
        h_ref = so.cross_covariance_np(p, q)
        assert np.abs(h.cpu().numpy() - h_ref).max() < 2e-5 * max(1.0, np.abs(h_ref).max())
        r_ref = so.symmetric_orthogonalization_np(h_ref)
        assert np.quantile(np.abs(rk.cpu().numpy() - r_ref), 0.99) < 2e-5
        if sigma == 0.0:
            assert np.abs(rk.cpu().numpy() - rg).max() < 5e-6               # noise-free: the generating rotation comes back
        else:
            assert so.angle_error_np(rk.cpu().numpy(), rg).max() < 2.0
    # the fused form equals the two-array Kabsch on the same (materialised) pairs
    p = torch.rand(2000, 1024, 3, device=DEV) - 0.5
    rg = rr.get_sampled_rotation_matrices_by_axisAngle(2000, DEV)
    q = torch.bmm(p, rg.transpose(1, 2))
    assert (rr.kabsch_rotation_synthetic(p, rg) - rr.kabsch_rotation(p, q)).abs().max().item() < 5e-6


# ------------------------------------------------------------------------------------------------
# K1 + K4 fused (evaluation step, 3D-Pose/main.py:60-62): no rotation written
# ------------------------------------------------------------------------------------------------
def test_fused_head_angle_error_matches_two_kernel_path(rr):
    g = load_golden("g6_stats_1m.npz")
    n = int(g["n"])
    torch.manual_seed(int(g["seed_x"]))
    x = torch.randn(n, 9).to(DEV)
    torch.manual_seed(int(g["seed_t"]))
    t = rr.symmetric_orthogonalization(torch.randn(n, 9).to(DEV))
    deg = rr.head_angle_error(x, t)                                          # (B,) degrees, R never written
    two = rr.angle_error(rr.symmetric_orthogonalization(x), t)
    assert (deg - two).abs().max().item() < 1e-9
    mean = rr.head_angle_error(x, t, reduce="mean").item()
    assert abs(mean - float(g["mean_angle_deg"])) < 1e-4                      # the BASELINE parity metric, one launch
    sc = rr.head_angle_error(x, t, reduce="sum_count")
    assert sc[1].item() == n and abs(sc[0].item() / n - mean) < 1e-12
    d2, r = rr.head_angle_error(x[:1000], t[:1000], return_rotation=True)      # ragged size: tail through K1 + K4
    assert (d2 - two[:1000]).abs().max().item() < 1e-9
    assert (r - rr.symmetric_orthogonalization(x[:1000])).abs().max().item() == 0
    with pytest.raises(ValueError, match="angle out of range"):
        rr.head_angle_error(x[:128], 1.7 * t[:128])


def _haar(n, gen):
    q = torch.randn(n, 4, device=DEV, generator=gen)
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w), 2 * (x * y + z * w), 1 - 2 * (x * x + z * z),
                        2 * (y * z - x * w), 2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], dim=1)


def test_fused_evaluation_sum_float32_outside_the_band_float64_inside(rr):
    """The reduced forms of head_angle_error (3D-Pose/main.py:60-62: `angle_error(func[rot_rep](out), R).mean()`) run the
    reference's expression in float32 for rows whose cosine is at least 5e-7 away from +-1 and in float64 inside that band
    (so3::angle_sum_f32).  Against `angle_error(...).mean()` -- float64 on every row, pinned to the reference by G3 / G6:
      * 1M Haar-like pairs (G6's rows) and 16M rows: |difference of the means| <= 1e-6 degrees;
      * rows AT the ill-conditioned ends (G3's 0 / 180 degree pairs and its 1e-4 rad pair: inside the band): equal to 1e-9;
      * G3 as a whole (it also holds a pair 3e-3 rad apart, outside the band, where one float32 ulp of the trace is worth
        2e-3 degrees on that row): <= 2e-5 degrees on the mean;
      * every pair 0.3 / 3 / 30 degrees apart (a trained network's regime): <= 5e-6 degrees;
      * exact=True: 1e-9 everywhere;  the range check raises on G3's bad pairs, not on its nearly-proper ones;  NaN stays NaN."""
    g6 = load_golden("g6_stats_1m.npz")
    n = int(g6["n"])
    torch.manual_seed(int(g6["seed_x"]))
    x = torch.randn(n, 9).to(DEV)
    torch.manual_seed(int(g6["seed_t"]))
    t = rr.symmetric_orthogonalization(torch.randn(n, 9).to(DEV))
    ref = rr.angle_error(rr.symmetric_orthogonalization(x), t).mean().item()
    got = rr.head_angle_error(x, t, reduce="mean").item()
    exact = rr.head_angle_error(x, t, reduce="mean", exact=True).item()
    report = {"1M": got - ref}
    assert abs(exact - ref) < 1e-9 and abs(got - ref) <= 1e-6, (got - ref, exact - ref)
    assert abs(got - float(g6["mean_angle_deg"])) < 1e-4                        # and the BASELINE parity metric against the reference's number
    # 16M rows (config #5's size on one device), in four slices for the float64 reference's per-row vector
    gen = torch.Generator(device=DEV).manual_seed(99)
    n16 = 16_000_000
    x16 = torch.randn(n16, 9, device=DEV, generator=gen)
    t16 = _haar(n16, gen)
    sc = rr.head_angle_error(x16, t16, reduce="sum_count")
    ref_sum = 0.0
    for lo in range(0, n16, 4_000_000):
        ref_sum += rr.angle_error(rr.symmetric_orthogonalization(x16[lo:lo + 4_000_000]), t16[lo:lo + 4_000_000]).sum().item()
    report["16M"] = sc[0].item() / n16 - ref_sum / n16
    assert sc[1].item() == n16 and abs(report["16M"]) <= 1e-6, report
    del x16, t16
    # G3's pairs on the engine (tiled to 2048 rows; M = r1, whose projection is the rotation itself up to round-off)
    g3 = load_golden("g3_angles.npz")
    r1, r2 = dev(g3["r1"]).reshape(-1, 9), dev(g3["r2"]).reshape(-1, 9)

    def both(a, b_, reps):
        a, b_ = a.repeat(reps, 1).contiguous(), b_.repeat(reps, 1).contiguous()
        return (rr.head_angle_error(a, b_, reduce="mean").item(), rr.angle_error(rr.symmetric_orthogonalization(a), b_).mean().item(),
                rr.head_angle_error(a, b_, reduce="mean", exact=True).item())
    ends = both(r1[:4], r2[:4], 512)                    # 0, 180, 180 degrees and 1e-4 rad: all inside the band -> float64
    assert abs(ends[0] - ends[1]) < 1e-9 and abs(ends[2] - ends[1]) < 1e-9, ends
    whole = both(r1, r2, 8)
    report["G3"] = whole[0] - whole[1]
    assert abs(whole[0] - whole[1]) <= 2e-5 and abs(whole[2] - whole[1]) < 1e-9, whole
    # a trained network's regime: every pair the same small angle apart
    for deg_apart in (0.3, 3.0, 30.0):
        m = 1 << 20
        base = _haar(m, gen)
        axis = torch.randn(m, 3, device=DEV, generator=gen)
        axis = axis / axis.norm(dim=1, keepdim=True) * (deg_apart * torch.pi / 180.0)
        tt = torch.bmm(base.view(m, 3, 3), rr.so3_exp_map(axis)).reshape(m, 9).contiguous()
        f32 = rr.head_angle_error(base, tt, reduce="mean").item()
        f64 = rr.angle_error(rr.symmetric_orthogonalization(base), tt).mean().item()
        report["%g deg" % deg_apart] = f32 - f64
        assert abs(f32 - f64) <= 5e-6 and abs(f64 - deg_apart) < 2e-2, (deg_apart, f32, f64, report)
    # the reference's range check, on the float32 cosine
    bad1, bad2 = dev(g3["bad1"]).reshape(-1, 9).repeat(700, 1).contiguous(), dev(g3["bad2"]).reshape(-1, 9).repeat(700, 1).contiguous()
    with pytest.raises(ValueError, match="angle out of range, input probably not proper rotation matrices"):
        rr.head_angle_error(bad1, bad2, reduce="mean")
    near1, near2 = dev(g3["nearly1"]).reshape(-1, 9).repeat(1024, 1).contiguous(), dev(g3["nearly2"]).reshape(-1, 9).repeat(1024, 1).contiguous()
    assert rr.head_angle_error(near1, near2, reduce="mean").item() == 0.0          # cos 1.075: inside the tolerance band, clamped
    xn = x[:4096].clone()
    xn[77, 3] = float("nan")
    assert np.isnan(rr.head_angle_error(xn, t[:4096], reduce="mean").item())
    # the caller-owned workspace finishes the float32 sum like the atomics do (same partials, a fixed order)
    from poseestimation_amd import _lib
    lib = _lib.load()
    ws = torch.zeros(lib.so3_reduce_workspace_bytes(), dtype=torch.uint8, device=DEV)
    sc_ws = torch.empty(2, dtype=torch.float64, device=DEV)
    fl = torch.empty(1, dtype=torch.int32, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    for flags in (0, _lib.RADIANS):
        assert lib.so3_project_angle_error_v2_f32(x.data_ptr(), t.data_ptr(), None, None, sc_ws.data_ptr(), fl.data_ptr(), ws.data_ptr(), flags, n, st) == 0
        unit = 1.0 if flags else 180.0 / np.pi
        assert sc_ws[1].item() == n and fl.item() == 0 and abs(sc_ws[0].item() / n - got / (180.0 / np.pi) * unit) < 1e-10
    assert int(torch.count_nonzero(ws).item()) == 0
    print("float32 angle sum minus float64, degrees:", report)


def test_zero_pool_is_per_stream(rr):
    """The metric kernels' pre-zeroed accumulator slots come from a pool per (device, stream): a kernel on a side stream must
    never add into slots whose zero-fill was enqueued on another stream (the advisor's round-3 finding)."""
    gen = torch.Generator(device=DEV).manual_seed(3)
    a, b_ = _haar(200_000, gen), _haar(200_000, gen)
    ref = rr.angle_error_sum_count(a, b_)[0].item()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.current_stream().synchronize()
    outs = []
    with torch.cuda.stream(side):
        for _ in range(300):                                  # more than one pool's worth of slots, all on the side stream
            outs.append(rr.angle_error_sum_count(a, b_, check=False))
    side.synchronize()
    pools = rr._ZERO_POOL.pools
    assert any(k[1] == side.cuda_stream for k in pools) and any(k[1] == torch.cuda.current_stream().cuda_stream for k in pools)
    assert all(abs(o[0].item() - ref) < 1e-6 and o[1].item() == 200_000 for o in outs)


# ------------------------------------------------------------------------------------------------
# drop-in check: the reference's comparison experiment (Comparison/models.py:14-43, Comparison/main.py:43-66)
# trained with the library's heads and with the oracle's ATen restatements, same weights, same data
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key", ["SVD", "6D", "5D", "Quat", "Euler"])
def test_comparison_experiment_trains_identically(rr, pa, key):
    from oracle import so3_oracle as so
    ref_heads = {"SVD": so.symmetric_orthogonalization_torch, "6D": so.ortho6d_torch, "5D": so.ortho5d_torch,
                 "Quat": so.quat_torch, "Euler": so.euler_torch}

    def make_model():
        torch.manual_seed(3)
        return torch.nn.Sequential(torch.nn.Linear(9, 128), torch.nn.ReLU(), torch.nn.Linear(128, 64), torch.nn.ReLU(),
                                   torch.nn.Linear(64, pa.head_dimensions[key])).double()

    gen = torch.Generator().manual_seed(17)
    target = so.symmetric_orthogonalization_torch(torch.randn(20, 128, 9, generator=gen).double()).reshape(20, 128, 3, 3)   # (steps, batch, 3, 3)
    inputs = target.reshape(20, 128, 9) + 0.05 * torch.randn(20, 128, 9, generator=gen).double()

    # Step by step: the float64 reference model (CPU, the reference's ops) leads; at every step the library model
    # (float32, device) starts from the same weights, and its loss and parameter gradients must agree.  (Free-running
    # float32 and float64 trajectories drift apart after ~10 steps through the head's ill-conditioned rows.)
    m_ref = make_model()
    m = make_model().float().to(DEV)
    opt = torch.optim.SGD(m_ref.parameters(), lr=0.01)
    first, last = None, None
    for x, r in zip(inputs, target):
        m.load_state_dict({k: v.float() for k, v in m_ref.state_dict().items()})
        opt.zero_grad()
        loss_ref = so.loss_frobenius_torch(ref_heads[key](m_ref(x)), r)
        loss_ref.backward()
        m.zero_grad()
        loss = rr.loss_frobenius(pa.head_functions[key](m(x.float().to(DEV))), r.float().to(DEV))
        loss.backward()
        assert abs(loss.item() - loss_ref.item()) < 2e-6 * max(1.0, loss_ref.item())
        for a, b in zip(m.parameters(), m_ref.parameters()):
            ga, gb = a.grad.detach().cpu().double(), b.grad.detach()
            assert (ga - gb).norm().item() < 1e-3 * gb.norm().item() + 1e-9, (key, (ga - gb).norm().item(), gb.norm().item())
        opt.step()
        first = loss_ref.item() if first is None else first
        last = loss_ref.item()
    assert last < first                                       # and it does train


# ------------------------------------------------------------------------------------------------
# launch-bound batches of the metric (B <= 1024): one launch, the kernel writes (sum, count) and the flag itself
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("b", [1, 63, 64, 65, 512, 1000, 1024, 1025])
def test_small_batch_metric_is_one_launch_with_the_same_angles(rr, b):
    gen = torch.Generator(device=DEV).manual_seed(700 + b)
    big = 4096                                                  # the same rows inside a batch that takes the streaming path
    x = torch.randn(big, 9, device=DEV, generator=gen)
    t = rr.symmetric_orthogonalization(torch.randn(big, 9, device=DEV, generator=gen))
    r = rr.symmetric_orthogonalization(x)
    ref_deg = rr.angle_error(r, t)
    deg = rr.angle_error(r[:b], t[:b])
    assert torch.equal(deg, ref_deg[:b])                        # per-row angles: the same operations in the same order
    sc = rr.angle_error_sum_count(r[:b], t[:b])
    assert sc[1].item() == b and abs(sc[0].item() - ref_deg[:b].sum().item()) < 1e-9 * max(b, 1)
    fused = rr.head_angle_error(x[:b], t[:b])
    assert (fused - ref_deg[:b]).abs().max().item() < 1e-9
    fsc = rr.head_angle_error(x[:b], t[:b], reduce="sum_count", exact=True)
    assert fsc[1].item() == b and abs(fsc[0].item() - ref_deg[:b].sum().item()) < 1e-8 * max(b, 1)
    fsc = rr.head_angle_error(x[:b], t[:b], reduce="sum_count")         # above 1024 rows: float32 trace outside the band around cos = +-1
    assert fsc[1].item() == b and abs(fsc[0].item() - ref_deg[:b].sum().item()) < (1e-8 if b <= 1024 else 5e-6) * max(b, 1)
    d2, r2 = rr.head_angle_error(x[:b], t[:b], return_rotation=True)
    assert torch.equal(r2, r[:b]) and (d2 - ref_deg[:b]).abs().max().item() < 1e-9
    with pytest.raises(ValueError, match="angle out of range"):
        rr.angle_error(r[:b], 3.0 * r[:b])                      # tr = 9, cosine 4: out of range whatever the rows are
    with pytest.raises(ValueError, match="angle out of range"):
        rr.head_angle_error(x[:b], 3.0 * r[:b], reduce="mean")
    bad = t[:b].clone()
    bad[b // 2] *= float("nan")                                 # NaN is not "out of range" (torch.any of a false comparison)
    assert torch.isnan(rr.angle_error(r[:b], bad)[b // 2])


@pytest.mark.parametrize("b", [1, 64, 65, 512, 1024, 1025])
def test_small_batch_stand_alone_loss_is_one_launch_with_the_same_rows(rr, b):
    gen = torch.Generator(device=DEV).manual_seed(900 + b)
    big = 4096
    p_all = rr.symmetric_orthogonalization(torch.randn(big, 9, device=DEV, generator=gen))
    t_all = rr.symmetric_orthogonalization(torch.randn(big, 9, device=DEV, generator=gen))
    pb = p_all.detach().clone().requires_grad_(True)
    rr.loss_frobenius(pb, t_all).backward()                    # streaming path: per-row gradients scaled by 1 / big
    ps = p_all[:b].detach().clone().requires_grad_(True)
    loss = rr.loss_frobenius(ps, t_all[:b])
    loss.backward()
    ref = torch.linalg.matrix_norm((t_all[:b] - p_all[:b]).double()).mean().item()
    assert abs(loss.item() - ref) < 2e-6 * max(ref, 1.0)
    assert torch.allclose(ps.grad * b, pb.grad[:b] * big, rtol=2e-6, atol=1e-7)   # same row arithmetic, another 1/B
    same = p_all[:b].detach().clone().requires_grad_(True)     # zero difference: zero loss, zero gradient, no NaN
    z = rr.loss_frobenius(same, p_all[:b].detach())
    z.backward()
    assert z.item() == 0.0 and float(same.grad.abs().max()) == 0.0
