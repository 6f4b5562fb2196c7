"""Pins the CPU oracle (oracle/gs_oracle.c) before anything trusts it.

1. the reference tests' own fixtures on this path (SH polynomials, quaternion convention),
2. the known-answer values recorded from the reference's kernels,
3. float64 finite differences of the oracle's forward against its hand-derived backward.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
APX = json.load(open(os.path.join(HERE, "golden", "survey_appendix_c.json")))
FIX = json.load(open(os.path.join(HERE, "golden", "reference_test_fixtures.json")))

# Signed constant tables as the reference's tests use them (ShUtilsTests.swift via ShUtils.swift:4-32)
C0, C1 = 0.28209479177387814, 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]
C4 = [2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431,
      -0.6690465435572892, 0.47308734787878004, -1.7701307697799304, 0.6258357354491761]


def _expected_sh(deg, sh, d):
    """The polynomial the reference's tests expand by hand (ShUtilsTests.swift:30-150)."""
    x, y, z = d
    res = C0 * sh[0]
    if deg > 0:
        res += -C1 * y * sh[1] + C1 * z * sh[2] - C1 * x * sh[3]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    if deg > 1:
        res += (C2[0] * xy * sh[4] + C2[1] * yz * sh[5] + C2[2] * (2 * zz - xx - yy) * sh[6] + C2[3] * xz * sh[7]
                + C2[4] * (xx - yy) * sh[8])
    if deg > 2:
        res += (C3[0] * y * (3 * xx - yy) * sh[9] + C3[1] * xy * z * sh[10] + C3[2] * y * (4 * zz - xx - yy) * sh[11]
                + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[12] + C3[4] * x * (4 * zz - xx - yy) * sh[13]
                + C3[5] * z * (xx - yy) * sh[14] + C3[6] * x * (xx - 3 * yy) * sh[15])
    if deg > 3:
        res += (C4[0] * xy * (xx - yy) * sh[16] + C4[1] * yz * (3 * xx - yy) * sh[17] + C4[2] * xy * (7 * zz - 1) * sh[18]
                + C4[3] * yz * (7 * zz - 3) * sh[19] + C4[4] * (zz * (35 * zz - 30) + 3) * sh[20]
                + C4[5] * xz * (7 * zz - 3) * sh[21] + C4[6] * (xx - yy) * (7 * zz - 1) * sh[22]
                + C4[7] * xz * (xx - 3 * yy) * sh[23] + C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy)) * sh[24])
    return res


def _identity_cam(W=800, H=800, fx=1111.11):
    view = np.eye(4, dtype=np.float32)
    fov = 2 * np.arctan(np.float32(W) / (2 * np.float32(fx)))
    from gaussiansplattingmlx_amd.camera import getProjectionMatrix
    proj = getProjectionMatrix(0.1, 100.0, float(fov), float(fov)).astype(np.float32)
    return view, proj, float(fov)


@pytest.mark.parametrize("case", FIX["sh"])
def test_sh_matches_reference_test_polynomials(oracle32, oracle64, case):
    deg = case["deg"]
    sh = case.get("sh") or list(np.arange(*case["sh_arange"]))[: (deg + 1) ** 2]
    d = case["dir"]
    want = _expected_sh(deg, sh, d)
    for orc, tol in ((oracle32, 2e-6), (oracle64, 1e-12)):
        b = orc.sh_basis(deg, *d)
        got = float(np.dot(b[: len(sh)].astype(np.float64), np.asarray(sh, np.float64)))
        assert abs(got - want) <= tol * max(1.0, abs(want))
    # and through the projection kernel: colour = max(sum + 0.5, 0) with direction = means3d - camCenter
    K = 25
    shs = np.zeros((1, K, 3), np.float32)
    shs[0, : len(sh), :] = np.asarray(sh, np.float32)[:, None]
    view, proj, fov = _identity_cam()
    out = oracle32.projection_forward(np.full((1, 3), 0.01), [[1, 0, 0, 0]], [d], shs, [0, 0, 0], view, proj, fov, fov,
                                      1111.11, 1111.11, 800, 800, deg)
    np.testing.assert_allclose(out["color"][0], max(want + 0.5, 0.0), rtol=2e-6, atol=2e-6)


def test_sh_degree0(oracle32):
    f = FIX["sh_deg0"]
    for s, d in zip(f["sh"], f["dirs"]):
        b = oracle32.sh_basis(0, *d)
        assert abs(b[0] * s - C0 * s) < 1e-7 and np.all(b[1:] == 0)


def test_quaternion_convention(oracle32):
    for c in FIX["build_rotation"]:
        _, rot = oracle32.cov3d([1, 1, 1], c["q"])
        np.testing.assert_allclose(rot, np.asarray(c["R"], np.float32), atol=1e-6)
    c = FIX["build_scaling_rotation"]
    cov, rot = oracle32.cov3d(c["s"], c["q"])
    L = rot * np.asarray(c["s"], np.float32)[None, :]
    np.testing.assert_allclose(L, np.asarray(c["L"], np.float32), atol=1e-6)
    np.testing.assert_allclose(cov, L @ L.T, atol=1e-6)


def test_camera_matches_oracle(oracle32):
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    c2w = look_at_c2w([2.0, -3.0, 1.5])
    cam = Camera(800, 600, 1111.11, 1000.0, c2w)
    view, proj, fx, fy, cc = oracle32.camera_build(c2w, 1111.11, 1000.0, 800, 600)
    np.testing.assert_allclose(view, cam.worldViewTransform, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(proj, cam.projectionMatrix, rtol=1e-6, atol=1e-7)
    assert fx == float(cam.FoVx) and fy == float(cam.FoVy)
    np.testing.assert_allclose(cc, cam.cameraCenter, rtol=1e-7)
    # row-vector convention: the camera centre maps to the view-space origin
    pv = np.append(cam.cameraCenter, 1.0) @ cam.worldViewTransform.astype(np.float64)
    np.testing.assert_allclose(pv[:3], 0.0, atol=1e-5)


# ---------------------------------------------------------------- Appendix C
def _apx_projection_inputs():
    P = APX["projection"]
    N, K = P["N"], P["K"]
    shs = (0.3 * np.sin(0.37 * np.arange(N * K * 3) + 1)).reshape(N, K, 3)
    view = np.eye(4)
    view[3, :] = P["view_row3"]
    W, H, fx = P["W"], P["H"], P["fx"]
    fov = 2 * np.arctan(np.float32(W) / (2 * np.float32(fx)))
    t = np.tan(float(fov) / 2) * P["znear"]
    n, f = P["znear"], P["zfar"]
    proj = np.array([[2 * n / (2 * t), 0, 0, 0], [0, 2 * n / (2 * t), 0, 0], [0, 0, f / (f - n), 1],
                     [0, 0, -n * f / (f - n), 0]])
    return dict(scales=np.array(P["scales"]), rot=np.array(P["rotations"]), means=np.array(P["means"]), shs=shs,
                cam=np.array(P["camCenter"], float), view=view, proj=proj, fov=float(fov), fx=fx, W=W, H=H,
                degree=P["degree"])


def test_appendix_c_projection_forward(oracle32):
    i = _apx_projection_inputs()
    out = oracle32.projection_forward(i["scales"], i["rot"], i["means"], i["shs"], i["cam"], i["view"], i["proj"],
                                      i["fov"], i["fov"], i["fx"], i["fx"], i["W"], i["H"], i["degree"])
    F = APX["projection"]["forward"]
    np.testing.assert_allclose(out["means2d"], F["means2d"], atol=6e-3)
    np.testing.assert_allclose(out["depths"], F["depth"], atol=1e-3)
    np.testing.assert_allclose(out["color"], F["color"], rtol=1e-5, atol=1e-3)
    con = out["conic"].reshape(-1, 4)
    np.testing.assert_allclose(con[:, [0, 1, 3]], F["conic_c00_c01_c11"], atol=6e-6)
    np.testing.assert_allclose(con[:, 1], con[:, 2], rtol=1e-3)
    np.testing.assert_array_equal(out["radii"], F["radius"])
    np.testing.assert_allclose(np.rint(out["rectMin"]), F["rectMin"], atol=0)   # printed to the nearest integer
    np.testing.assert_allclose(np.rint(out["rectMax"]), F["rectMax"], atol=0)


def test_appendix_c_projection_backward(oracle32):
    i = _apx_projection_inputs()
    N = 3
    cotConic = np.zeros((N, 4)); cotConic[1, 0] = 1
    cotM = np.zeros((N, 2)); cotM[1, 0] = 1
    cotC = np.zeros((N, 3)); cotC[1, 0] = 1
    out = oracle32.projection_backward(i["scales"], i["rot"], i["means"], i["shs"], i["cam"], i["view"], i["proj"],
                                       i["fov"], i["fov"], i["fx"], i["fx"], i["W"], i["H"], i["degree"],
                                       np.zeros(N), cotM, np.zeros((N, 4)), cotC, cotConic)
    B = APX["projection"]["backward"]
    np.testing.assert_allclose(out["gradMeans3d"][1], B["gradMeans3d"], rtol=2e-5)
    np.testing.assert_allclose(out["gradScales"][1], B["gradScales"], rtol=2e-3)
    np.testing.assert_allclose(out["gradRot"][1], B["gradRotations"], rtol=3e-3)
    np.testing.assert_allclose(out["gradCamCenterPoint"][1], B["gradCameraCenterPoint"], rtol=2e-5)
    # untouched Gaussians get exact zeros
    assert not np.any(out["gradMeans3d"][[0, 2]]) and not np.any(out["gradShs"][[0, 2]])


def test_appendix_c_blend_forward(oracle32):
    b = APX["blend_forward"]
    packed = np.array(b["packed"], np.float32)
    idx = np.array(b["packedTileIndices"], np.uint32).reshape(-1)
    ranges = np.array([[0, 2], [2, 4]], np.uint32)
    color, depth, alpha, last = oracle32.blend_forward(packed, idx, ranges, b["W"], b["H"], b["tileW"], b["tileH"],
                                                       b["whiteBg"])
    x, y = b["pixel"]
    p = y * b["W"] + x
    np.testing.assert_allclose(color[p], b["color"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(depth[p], b["depth"], rtol=2e-6)
    np.testing.assert_allclose(alpha[p], b["alpha"], rtol=2e-6)
    assert last[p] == b["nContrib"]
    # dense-table view of the same lists (build_packed_tile_indices)
    dense = oracle32.build_packed_tile_indices(idx, ranges, 2)
    np.testing.assert_array_equal(dense, b["packedTileIndices"])


# ------------------------------------------------------- finite differences
def _rand_scene(rng, N=24, K=25):
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    cam = Camera(64, 48, 70.0, 75.0, look_at_c2w([1.8, -2.4, 1.6]))
    p = dict(xyz=rng.uniform(-0.6, 0.6, (N, 3)), features_dc=rng.normal(0, 1, (N, 1, 3)),
             features_rest=rng.normal(0, 0.1, (N, K - 1, 3)), scales=rng.normal(np.log(0.08), 0.4, (N, 3)),
             rotation=rng.normal(0, 1, (N, 4)), opacity=rng.normal(0.5, 1.0, N))
    return p, cam


def _fd(f, x, h=1e-6):
    g = np.zeros_like(x)
    flat, gf = x.reshape(-1), g.reshape(-1)
    for i in range(flat.size):
        old = flat[i]
        flat[i] = old + h; a = f()
        flat[i] = old - h; b = f()
        flat[i] = old
        gf[i] = (a - b) / (2 * h)
    return g


def test_projection_backward_matches_fd(oracle64):
    rng = np.random.default_rng(1)
    p, cam = _rand_scene(rng, N=6)
    c = cam.as_dict()
    o = oracle64
    op, sc, rt = o.activations_forward(p["opacity"], p["scales"], p["rotation"])
    shs = np.concatenate([p["features_dc"], p["features_rest"]], 1)
    N = 6
    cots = dict(depths=rng.normal(size=N), means2d=rng.normal(size=(N, 2)), cov2d=rng.normal(size=(N, 2, 2)),
                color=rng.normal(size=(N, 3)), conic=rng.normal(size=(N, 2, 2)) * 1e2)
    args = dict(scales=sc, rot=rt, means=p["xyz"].copy(), shs=shs, cam=c["camCenter"].astype(np.float64))

    def fwd():
        out = o.projection_forward(args["scales"], args["rot"], args["means"], args["shs"], args["cam"], c["view"],
                                   c["proj"], c["fovX"], c["fovY"], c["focalX"], c["focalY"], 64, 48, 4)
        return float(sum(np.sum(out[k] * cots[k]) for k in cots))

    bw = o.projection_backward(sc, rt, p["xyz"], shs, args["cam"], c["view"], c["proj"], c["fovX"], c["fovY"],
                               c["focalX"], c["focalY"], 64, 48, 4, cots["depths"], cots["means2d"], cots["cov2d"],
                               cots["color"], cots["conic"])
    for name, key in (("gradMeans3d", "means"), ("gradScales", "scales"), ("gradRot", "rot"), ("gradShs", "shs")):
        fd = _fd(fwd, args[key])
        scale = np.abs(fd).max() + 1e-12
        np.testing.assert_allclose(bw[name] / scale, fd / scale, atol=2e-6, err_msg=name)
    fd = _fd(fwd, args["cam"])
    np.testing.assert_allclose(bw["gradCamCenterPoint"].sum(0), fd, rtol=1e-5, atol=1e-6 * np.abs(fd).max())


def test_activations_backward_matches_fd(oracle64):
    rng = np.random.default_rng(2)
    N = 5
    a = dict(o=rng.normal(size=N), s=rng.normal(size=(N, 3)), q=rng.normal(size=(N, 4)))
    g = dict(o=rng.normal(size=N), s=rng.normal(size=(N, 3)), q=rng.normal(size=(N, 4)))

    def fwd():
        op, sc, rt = oracle64.activations_forward(a["o"], a["s"], a["q"])
        return float(np.sum(op * g["o"]) + np.sum(sc * g["s"]) + np.sum(rt * g["q"]))

    do, ds, dq = oracle64.activations_backward(a["o"], a["s"], a["q"], g["o"], g["s"], g["q"])
    np.testing.assert_allclose(do, _fd(fwd, a["o"]), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(ds, _fd(fwd, a["s"]), rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(dq, _fd(fwd, a["q"]), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("white", [False, True])
def test_blend_backward_matches_fd(oracle64, white):
    rng = np.random.default_rng(3)
    W, H, tw, th = 32, 16, 16, 16
    N = 10
    packed = np.zeros((N, 11))
    packed[:, 0] = rng.uniform(0, W, N); packed[:, 1] = rng.uniform(0, H, N)
    a = rng.uniform(0.01, 0.05, N); c = rng.uniform(0.01, 0.05, N); b = rng.uniform(-0.005, 0.005, (N, 2))
    packed[:, 2], packed[:, 3], packed[:, 4], packed[:, 5] = a, b[:, 0], b[:, 1], c
    packed[:, 6:9] = rng.uniform(0, 1, (N, 3)); packed[:, 9] = rng.uniform(0.2, 0.9, N)
    packed[:, 10] = rng.uniform(1, 5, N)
    idx = np.concatenate([rng.permutation(N), rng.permutation(N)]).astype(np.uint32)
    ranges = np.array([[0, N], [N, 2 * N]], np.uint32)
    cC, cD, cA = rng.normal(size=(W * H, 3)), rng.normal(size=W * H), rng.normal(size=W * H)
    o = oracle64

    def fwd():
        col, dep, alp, _ = o.blend_forward(packed, idx, ranges, W, H, tw, th, white)
        return float(np.sum(col * cC) + np.sum(dep * cD) + np.sum(alp * cA))

    col, dep, alp, last = o.blend_forward(packed, idx, ranges, W, H, tw, th, white)
    assert (last == N).all()      # no early termination: the forward is smooth here
    g = o.blend_backward(packed, idx, ranges, W, H, tw, th, white, cC, cD, cA, col, dep, alp, last)
    fd = _fd(fwd, packed)
    np.testing.assert_allclose(g, fd, rtol=2e-5, atol=1e-6 * np.abs(fd).max())


def test_blend_alpha_clamp_blocks_gradient(oracle64):
    # opacity*exp > 0.99 -> alpha clamps, and the reference's derivative is then 0 for mean/conic/opacity
    W, H = 16, 16
    packed = np.array([[8.0, 8.0, 1e-4, 0, 0, 1e-4, 1, 1, 1, 1.0, 2.0]])
    idx, ranges = np.array([0], np.uint32), np.array([[0, 1]], np.uint32)
    col, dep, alp, last = oracle64.blend_forward(packed, idx, ranges, W, H, 16, 16, False)
    np.testing.assert_allclose(alp, 0.99, rtol=1e-12)
    g = oracle64.blend_backward(packed, idx, ranges, W, H, 16, 16, False, np.ones((W * H, 3)), np.zeros(W * H),
                                np.zeros(W * H), col, dep, alp, last)
    assert np.all(g[0, [0, 1, 2, 3, 4, 5, 9]] == 0) and np.all(g[0, 6:9] > 0)


def test_blend_early_termination(oracle32):
    # 40 opaque splats on one pixel: T < 1e-4 after two (0.01^2 == 1e-4 is not < 1e-4 in exact arithmetic, f32 decides)
    W, H = 16, 16
    N = 40
    packed = np.tile(np.array([[8.0, 8.0, 1e-4, 0, 0, 1e-4, 0.5, 0.5, 0.5, 1.0, 2.0]], np.float32), (N, 1))
    idx, ranges = np.arange(N, dtype=np.uint32), np.array([[0, N]], np.uint32)
    col, dep, alp, last = oracle32.blend_forward(packed, idx, ranges, W, H, 16, 16, False)
    assert last.max() <= 3 and last.min() >= 2
    T = np.float32(1.0)
    for k in range(int(last[8 * 16 + 8])):
        T = T * (np.float32(1.0) - np.float32(0.99))
    assert T < 1e-4 and np.float32(1.0) - T == alp[8 * 16 + 8]


def test_ssim_backward_matches_fd(oracle64):
    rng = np.random.default_rng(4)
    H, W = 14, 13
    a, b = rng.uniform(0, 1, (H, W, 3)), rng.uniform(0, 1, (H, W, 3))
    up = rng.normal(size=(H, W, 3))
    o = oracle64

    def fwd():
        return float(np.sum(o.ssim_forward(a, b)[0] * up))

    outs = o.ssim_forward(a, b)
    g1, g2 = o.ssim_backward(up, a, b, outs[1:])
    np.testing.assert_allclose(g1, _fd(fwd, a), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(g2, _fd(fwd, b), rtol=1e-5, atol=1e-7)


def test_ssim_window_is_off_centre_and_identical_images(oracle32):
    w = oracle32.ssim_window().reshape(11, 11)
    assert abs(w.sum() - 1) < 1e-6
    g = w.sum(0)
    assert g.argmax() in (5, 6) and abs(g[5] - g[6]) < 1e-7 and g[0] < g[10]   # centre 5.5: asymmetric
    img = np.random.default_rng(5).uniform(0, 1, (20, 20, 3)).astype(np.float32)
    s = oracle32.ssim_forward(img, img)[0]
    np.testing.assert_allclose(s, 1.0, atol=2e-5)
    ones, zeros = np.ones((32, 32, 3), np.float32), np.zeros((32, 32, 3), np.float32)
    assert oracle32.ssim_forward(ones, zeros)[0].mean() < 0.01      # TrainTests.swift:55-80 cases
    assert abs(oracle32.ssim_forward(ones, ones)[0].mean() - 1) < 1e-5


def test_loss_cotangent_matches_fd(oracle64):
    rng = np.random.default_rng(6)
    H, W = 12, 12
    r, t = rng.uniform(0, 1, (H, W, 3)), rng.uniform(0, 1, (H, W, 3))
    rd, td = rng.uniform(1, 3, (H, W)), rng.uniform(1, 3, (H, W))
    mask = rng.uniform(size=(H, W)) > 0.4
    o = oracle64

    def fwd():
        return o.loss_forward_backward(r, t, 0.2, rd, td, mask, 0.3)[0]

    loss, cc, cd, l1, ss = o.loss_forward_backward(r, t, 0.2, rd, td, mask, 0.3)
    assert abs(loss - (0.8 * l1 + 0.2 * (1 - ss) + 0.3 * (np.abs(rd - td) * mask).sum() / mask.sum())) < 1e-12
    np.testing.assert_allclose(cc, _fd(fwd, r), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(cd, _fd(fwd, rd), rtol=1e-5, atol=1e-8)


def test_full_render_backward_matches_fd(oracle64):
    """raw parameters -> image -> scalar, through activations, projection, packing, binning and blending."""
    rng = np.random.default_rng(7)
    p, cam = _rand_scene(rng, N=12)
    c = cam.as_dict()
    W, H = 64, 48
    o = oracle64
    cC, cD, cA = rng.normal(size=(W * H, 3)), rng.normal(size=W * H) * 0.1, rng.normal(size=W * H)
    fw = o.render_forward(p, c, W, H, 16, 16, 4)
    assert fw["bin"].M > 0

    def fwd():
        f = o.render_forward(p, c, W, H, 16, 16, 4)
        return float(np.sum(f["color"] * cC) + np.sum(f["depth"] * cD) + np.sum(f["alpha"] * cA))

    g = o.render_backward(p, c, W, H, 16, 16, 4, fw, cC, cD, cA)
    for k in ("xyz", "scales", "rotation", "opacity", "features_dc"):
        fd = _fd(fwd, p[k], h=1e-6)
        scale = np.abs(fd).max() + 1e-12
        np.testing.assert_allclose(g[k] / scale, fd / scale, atol=5e-5, err_msg=k)


# ---------------------------------------------------------------- the reference's remaining vectors on section-8 rows
def test_inverse_sigmoid_reference_vector():
    """GaussianSplattingMlxTests.swift:36-53 -- pins the opacity initialisation of model_init.create_from_pcd (row f4)."""
    from gaussiansplattingmlx_amd.model_init import inverse_sigmoid
    f = FIX["inverse_sigmoid"]
    x = np.array(f["values"], np.float32)
    want = np.log(x / (np.float32(1) - x))
    np.testing.assert_allclose(inverse_sigmoid(x), want, atol=f["atol"])
    np.testing.assert_allclose(inverse_sigmoid(x), [-2.1972246, 0.0, 2.1972246], atol=f["atol"])


def test_get_rays_from_images_reference_vector():
    """PointCloudUtilsTests.swift:15-61 -- pins pointcloud.getRaysFromImages (row f4): 2x2 image, identity intrinsics and
    pose: origins 0, directions (u, v, 1) in u-minor order."""
    from gaussiansplattingmlx_amd.pointcloud import getRaysFromImages
    f = FIX["get_rays_from_images"]
    eye = np.eye(4, dtype=np.float32)[None]
    o, d = getRaysFromImages(f["H"], f["W"], eye, eye)
    assert o.shape == (1, 4, 3) and d.shape == (1, 4, 3)
    np.testing.assert_allclose(o[0], f["origins"], atol=f["atol"])
    np.testing.assert_allclose(d[0], f["directions"], atol=f["atol"])


def test_simd_row_major_layout_reference_vector(oracle32):
    """TinyTests.swift:145-187 -- pins row a1's matrix convention: the array the kernels get is a[r][c] = simd[c][r], so
    simd's row-vector product v * m equals v @ a; and products commute with the conversion.  Then the same convention
    on the camera itself: p_view = [p, 1] @ worldViewTransform, in camera.py and in the oracle's camera_build."""
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w, simd_to_row_major
    f = FIX["simd_layout"]
    cols = np.array([[i * 10 + j for j in range(4)] for i in range(4)], np.float32)     # cols[i][j] = mat[i, j]
    a = simd_to_row_major(cols)
    v = np.array(f["row_vector"], np.float32)
    np.testing.assert_array_equal(v @ a, np.array(f["row_vector_times_matrix"], np.float32))
    # simd v * m = sum_j v[j] * m[i, j] per output i (row vector times matrix, columns stored)
    np.testing.assert_array_equal(np.array([sum(v[j] * cols[i][j] for j in range(4)) for i in range(4)]), v @ a)
    colsB = np.array([[100 + i * 10 + j for j in range(4)] for i in range(4)], np.float32)
    # simd A * B (column storage): (A*B)[c][r] = sum_k A[k][r] * B[c][k]
    prod = np.array([[sum(cols[k][r] * colsB[c][k] for k in range(4)) for r in range(4)] for c in range(4)], np.float32)
    np.testing.assert_array_equal(simd_to_row_major(prod), a @ simd_to_row_major(colsB))
    # the camera: a world point goes to view space as a row vector, translation in row 3 of the array
    c2w = look_at_c2w([2.0, -1.5, 1.0])
    cam = Camera(64, 48, 60.0, 60.0, c2w)
    p = np.array([0.3, -0.2, 0.4, 1.0])
    want = np.linalg.inv(c2w) @ p                                   # column-vector form of the same transform
    np.testing.assert_allclose(p.astype(np.float32) @ cam.worldViewTransform, want, atol=1e-6)
    np.testing.assert_allclose(cam.worldViewTransform[3, :3], np.linalg.inv(c2w)[:3, 3], atol=1e-6)
    view, proj, fx, fy, centre = oracle32.camera_build(c2w, 60.0, 60.0, 64, 48)
    np.testing.assert_allclose(view, cam.worldViewTransform, atol=1e-7)        # two f64 inversions, then the f32 cast
    np.testing.assert_allclose(proj, cam.projectionMatrix, rtol=2e-7)


def test_dist_topk_reference_vector_pins_the_brute_force():
    """GaussianModelTests.swift:16-36: 4 points, k = 2 -> 0.5 each.  Pins the numpy brute force that
    tests/test_loaders.py holds gs_dist_topk to (the GPU kernel itself meets the same vector in test_gpu_parity.py)."""
    f = FIX["dist_topk"]
    X = np.array(f["points"], np.float64)
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1)
    np.testing.assert_allclose(np.sort(d2, axis=1)[:, :f["k"]].mean(1), f["expect"])
