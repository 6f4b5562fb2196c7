"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): integer/index work bit-exact; rendered RGB within 1e-4 L-inf; gradients
within 1e-3 relative (relative to the largest magnitude of the tensor, since float atomics reorder sums).
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
RGB_TOL = 1e-4
GRAD_RTOL = 1e-3


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X box")


def _renderer(W, H, tile=(16, 16), white=False, degree=4):
    _need_gpu()
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    return GaussianRenderer(degree, W, H, tile, white)


def _scene(seed, N, W, H, K=25, spread=0.9, scale=0.05, focal=None):
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    rng = np.random.default_rng(seed)
    focal = focal or 0.9 * W
    cam = Camera(W, H, focal, focal * 1.02, look_at_c2w([2.2, -2.6, 1.7]))
    p = dict(xyz=rng.uniform(-spread, spread, (N, 3)), features_dc=rng.normal(0, 1, (N, 1, 3)),
             features_rest=rng.normal(0, 0.08, (N, K - 1, 3)), scales=rng.normal(np.log(scale), 0.5, (N, 3)),
             rotation=rng.normal(0, 1, (N, 4)), opacity=rng.normal(0.3, 1.5, N))
    p = {k: np.ascontiguousarray(v, np.float32) for k, v in p.items()}
    return p, cam


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def _np(t):
    return t.detach().cpu().numpy()


GRAD_KEYS = ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity")


def _share_beyond(a, b):
    """BASELINE.md section 3's bar as written -- "max relative gradient error ... with an absolute floor for near-zero
    entries" -- taken ELEMENT by element: the share of a tensor's elements with |a_i - b_i| > 1e-3 max(|b_i|, 1e-4 max|b|).
    (The max-norm bar `_rel` lets an element 100x smaller than the tensor's largest be 10 % off; with float atomics a pure
    element-wise bar of zero exceptions is not holdable -- the float32 and float64 ORACLES do not meet it either -- so the
    share is reported and bounded against that pair's.)"""
    a, b = np.asarray(a, np.float64).reshape(-1), np.asarray(b, np.float64).reshape(-1)
    floor = 1e-4 * (np.abs(b).max() + 1e-300)
    return float((np.abs(a - b) > GRAD_RTOL * np.maximum(np.abs(b), floor)).mean())


def _elementwise_gradient_bar(tag, got, want32, oracle64, params, c, W, H, tgt, cot_fixed=None, tile=(16, 16)):
    """Holds the HIP gradients to the element-wise bar relative to what float32 arithmetic itself can hold: per tensor, the
    share of elements beyond 1e-3 (floored) between HIP and the float32 oracle may not exceed 1.5 x the share between the
    float32 and the float64 oracle on the same view + 5e-4 (the yardstick tests/test_gpu_trajectory.py uses for ten steps).
    The float64 oracle runs the whole chain itself (its own forward, its own loss cotangent unless one is prescribed).
    Writes the numbers to gpurun_out/gradient_elementwise_<tag>.json.

    Measured (MI355X; share HIP vs float32 oracle / share float32 vs float64 oracle, worst tensor):
      c3 (300 k, 800x800, loss cotangent)   1.9e-4 / 1.0e-3      c3 raw colours   1.3e-5 / 8.6e-4
      c2 (100 k, 800x800, random cotangent) 3.9e-3 / 5.4e-2      c2 raw colours   2.8e-4 / 5.0e-2
      c1 (10 k, 400x400), four-wave forward 1.3e-2 / 1.6e-2      c1, one-wave forward  1.4e-2 / 1.6e-2
    -- HIP sits 5 to 100 times BELOW the oracle pair's share on c2 / c3 and level with it on c1.
    Round 5 carried a special case here: c1 under the one-wave forward lay ABOVE the pair (opacity: 7 % of the elements
    against 1.5 %).  Every pixel of that scene blends ~1100 splats of opacity 0.1, and the segment-parallel backward takes
    what the rest of a list still owes from the stored image and a stored checkpoint; it was formed as cot . C_final -
    cot . C_checkpoint, two float32 dot products at the size of the total whose difference is small deep in a list, from sums
    the forward had accumulated entry by entry at that size.  Round 6 removed the cause instead of keeping the wider bar: the
    forward keeps its sums as base + chunk (one rounding at the size of the total per 64 positions instead of one per entry)
    and the backward differences the channels first (blend_v2.hip): 7.1 % -> 1.35 % on that case, below the pair's 1.5 %."""
    o64 = oracle64
    p64 = {k: np.asarray(v, np.float64) for k, v in params.items()}
    fw64 = o64.render_forward(p64, c, W, H, tile[0], tile[1], 4)
    if cot_fixed is None:
        _, cc64, _, _, _ = o64.loss_forward_backward(fw64["color"].reshape(H, W, 3), np.asarray(tgt, np.float64), 0.2)
        cot64 = cc64.reshape(-1, 3)
    else:
        cot64 = np.asarray(cot_fixed, np.float64)
    z = np.zeros(W * H, np.float64)
    want64 = o64.render_backward(p64, c, W, H, tile[0], tile[1], 4, fw64, cot64, z, z)
    report = dict(tag=tag, bar="share of elements with |a - b| > 1e-3 max(|b_i|, 1e-4 max|b|)", tensors={})
    for k in GRAD_KEYS:
        g, w32, w64 = _np(got[k]).astype(np.float64), np.asarray(want32[k], np.float64), np.asarray(want64[k], np.float64)
        w32 = w32.reshape(g.shape)
        hip = _share_beyond(g, w32)
        pair = _share_beyond(w32.reshape(-1), w64.reshape(-1))
        scale = np.abs(w32).max() + 1e-300
        beyond = np.abs(g - w32) > GRAD_RTOL * np.maximum(np.abs(w32), 1e-4 * scale)
        report["tensors"][k] = dict(hip_vs_oracle32=hip, oracle32_vs_oracle64=pair, max_norm_rel=float(_rel(g, w32)),
                                    hip_vs_oracle64=_share_beyond(g, w64.reshape(g.shape)),
                                    worst_abs_error_of_the_elements_beyond_over_max=float(np.abs(g - w32)[beyond].max() / scale) if beyond.any() else 0.0,
                                    median_size_of_the_elements_beyond_over_max=float(np.median(np.abs(w32)[beyond]) / scale) if beyond.any() else 0.0)
    os.makedirs(os.path.join(HERE, "..", "gpurun_out"), exist_ok=True)
    with open(os.path.join(HERE, "..", "gpurun_out", f"gradient_elementwise_{tag}.json"), "w") as f:
        json.dump(report, f, indent=1)
    for k, v in report["tensors"].items():
        assert v["hip_vs_oracle32"] <= 1.5 * v["oracle32_vs_oracle64"] + 5e-4, (tag, k, v)
    return report


def _ncontrib_close(got, want, slack_pixels=2):
    """nContrib is an integer cut at T < 1e-4: a pixel whose T lands within an ulp of the threshold can stop a splat or
    two earlier or later when exp() differs in the last bit (device v_exp_f32 vs libm).  Bar: at most 2e-5 of the pixels
    (never fewer than `slack_pixels`, for small images), and never by more than 8 list positions."""
    got, want = np.asarray(got).reshape(-1).astype(np.int64), np.asarray(want).reshape(-1).astype(np.int64)
    bad = int((got != want).sum())
    assert bad <= max(slack_pixels, int(2e-5 * got.size)), (bad, got.size)
    assert np.abs(got - want).max() <= 8


def _lists_trimmed(r):
    """Does this renderer's fused forward bin on TRIMMED rects (GS_TUNE_TRIM_RECTS, default 1, 16 x 16 tiles: the reference's
    3-sigma square cut by the box of the ellipse q <= 40.3, include/gsplat.h)?  Then M, nContrib and the exported lists count
    positions in lists that leave out the entries no pixel of the tile can see; with 0 they are the reference's, position
    for position."""
    return r.getTuning("trim_rects") != 0 and (r.TILE_SIZE.w, r.TILE_SIZE.h) == (16, 16)


def _pairs_match(r, want_M):
    """The fused forward's pair count against the oracle's: equal on the reference's lists, never more on trimmed ones."""
    M = r.stats()["M"]
    if _lists_trimmed(r):
        assert M <= want_M, (M, want_M)
    else:
        assert M == want_M, (M, want_M)
    return M


def _fused_lists(r, W, H):
    """(M, Gaussian index per pair, [start, end) per tile, count per tile) of the fused forward just run (gs_tile_bin_export)."""
    import ctypes as C
    M, T = r.stats()["M"], ((W + 15) // 16) * ((H + 15) // 16)
    idx = torch.zeros(max(M, 1), dtype=torch.int32, device=r.device)
    rng_ = torch.zeros(T, 2, dtype=torch.int32, device=r.device)
    cnt = torch.zeros(T, dtype=torch.int32, device=r.device)
    r._check(r.lib.gs_tile_bin_export(r.ctx, C.c_void_p(idx.data_ptr()), C.c_void_p(rng_.data_ptr()), C.c_void_p(cnt.data_ptr())))
    return M, _np(idx)[:M].astype(np.int64), _np(rng_).astype(np.int64), _np(cnt).astype(np.int64)


def _ncontrib_match(r, fw, W, H, nc=None, slack_pixels=2, max_tiles=400):
    """nContrib of the fused forward just run against the oracle's forward `fw`.  Reference lists: position for position
    (_ncontrib_close).  Trimmed lists (_lists_trimmed): positions count the trimmed list, so what is compared is what the
    position MEANS -- a pixel that terminated (T < 1e-4) stopped at the same Gaussian; a pixel live at the end went through
    its whole list (nContrib = the list's length, on either side) -- under the same slack for pixels whose T lands within an
    ulp of the threshold; and every tile's list (all of them, or `max_tiles` sampled) is the oracle's with entries left out,
    order kept."""
    nc = _np(r.lastContrib()) if nc is None else np.asarray(nc)
    if not _lists_trimmed(r):
        return _ncontrib_close(nc, fw["last"], slack_pixels)
    M, idx, rng_, cnt = _fused_lists(r, W, H)
    bn = fw["bin"]
    o_idx = np.asarray(bn.sortedIdx).astype(np.int64)
    o_rng = np.asarray(bn.tileRanges).astype(np.int64).reshape(-1, 2)
    o_cnt = np.asarray(bn.tileCounts).astype(np.int64)
    assert M == int(cnt.sum()) and M <= int(o_cnt.sum()) and (cnt <= o_cnt).all()
    gw = (W + 15) // 16
    ys, xs = np.divmod(np.arange(W * H), W)
    tile = (ys // 16) * gw + xs // 16
    got, want = nc.reshape(-1).astype(np.int64), np.asarray(fw["last"]).reshape(-1).astype(np.int64)
    T = 1.0 - np.asarray(fw["alpha"], np.float64).reshape(-1)
    dead, live = T < 0.9e-4, T > 1.1e-4
    pad = lambda a: np.concatenate([a, [-1]])          # (an empty list set: index 0 of nothing)
    gid = np.where(got > 0, pad(idx)[np.clip(rng_[tile, 0] + got - 1, 0, max(M - 1, 0)) if M else np.full(got.size, -1)], -1)
    wid = np.where(want > 0, pad(o_idx)[np.clip(o_rng[tile, 0] + want - 1, 0, max(o_idx.size - 1, 0)) if o_idx.size else np.full(want.size, -1)], -1)
    bad = int((gid != wid)[dead].sum()) + int((got != cnt[tile])[live].sum())
    assert bad <= max(slack_pixels, int(2e-5 * got.size)), (bad, got.size)
    has = np.nonzero(o_cnt > 0)[0]
    if has.size > max_tiles:
        has = np.unique(np.concatenate([np.random.default_rng(5).choice(has, max_tiles, replace=False), has[np.argsort(o_cnt[has])[-8:]]]))
    for t in has:
        a, b = idx[rng_[t, 0]:rng_[t, 1]], o_idx[o_rng[t, 0]:o_rng[t, 1]]
        keep = np.isin(b, a)
        assert int(keep.sum()) == a.size and np.array_equal(b[keep], a), f"tile {t}: not the oracle's list with entries left out"


# ------------------------------------------------------------------------------------------ projection
@pytest.mark.parametrize("degree", [0, 2, 4])
def test_projection_forward_backward(oracle32, degree):
    W, H, N = 200, 152, 4000
    p, cam = _scene(11, N, W, H)
    c = cam.as_dict()
    o = oracle32
    op, sc, rt = o.activations_forward(p["opacity"], p["scales"], p["rotation"])
    shs = np.concatenate([p["features_dc"], p["features_rest"]], 1)
    want = o.projection_forward(sc, rt, p["xyz"], shs, c["camCenter"], c["view"], c["proj"], c["fovX"], c["fovY"],
                                c["focalX"], c["focalY"], W, H, degree)
    r = _renderer(W, H, degree=degree)
    gcam = r._camera(c["view"], c["proj"], c["camCenter"], c["fovX"], c["fovY"], c["focalX"], c["focalY"])
    got = r.projectionScreenFused(sc, rt, p["xyz"], shs, gcam)
    # mean / depth / rect arithmetic is compiled without FMA contraction: bit-exact
    for k in ("means2d", "depths", "rectMin", "rectMax", "radii"):
        np.testing.assert_array_equal(_np(got[k]), want[k], err_msg=k)
    np.testing.assert_allclose(_np(got["color"]), want["color"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(_np(got["cov2d"]), want["cov2d"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(_np(got["conic"]), want["conic"], rtol=1e-5, atol=1e-9)

    rng = np.random.default_rng(5)
    cots = dict(depths=rng.normal(size=N), means2d=rng.normal(size=(N, 2)), cov2d=rng.normal(size=(N, 2, 2)),
                color=rng.normal(size=(N, 3)), conic=rng.normal(size=(N, 2, 2)) * 100)
    cots = {k: v.astype(np.float32) for k, v in cots.items()}
    wb = o.projection_backward(sc, rt, p["xyz"], shs, c["camCenter"], c["view"], c["proj"], c["fovX"], c["fovY"],
                               c["focalX"], c["focalY"], W, H, degree, cots["depths"], cots["means2d"], cots["cov2d"],
                               cots["color"], cots["conic"])
    gb = r.projectionScreenFusedVJP(sc, rt, p["xyz"], shs, gcam, cots["means2d"], cots["depths"], cots["color"],
                                    cots["cov2d"], cots["conic"])
    for k in ("gradScales", "gradRot", "gradMeans3d", "gradShs", "gradCamCenterPoint"):
        np.testing.assert_allclose(_np(gb[k]), wb[k], rtol=2e-4, atol=2e-5 * np.abs(wb[k]).max(), err_msg=k)
    coeff = (degree + 1) ** 2
    assert not np.any(_np(gb["gradShs"])[:, coeff:, :])


def test_projection_appendix_c_through_abi():
    A = json.load(open(os.path.join(HERE, "golden", "survey_appendix_c.json")))["projection"]
    from tests.test_oracle_pins import _apx_projection_inputs
    i = _apx_projection_inputs()
    r = _renderer(800, 800)
    cam = r._camera(i["view"], i["proj"], i["cam"], i["fov"], i["fov"], i["fx"], i["fx"])
    out = r.projectionScreenFused(i["scales"], i["rot"], i["means"], i["shs"], cam)
    F = A["forward"]
    np.testing.assert_allclose(_np(out["means2d"]), F["means2d"], atol=6e-3)
    np.testing.assert_allclose(_np(out["color"]), F["color"], rtol=1e-5, atol=1e-3)
    np.testing.assert_array_equal(_np(out["radii"]), F["radius"])
    N = 3
    cotConic = np.zeros((N, 4), np.float32); cotConic[1, 0] = 1
    cotM = np.zeros((N, 2), np.float32); cotM[1, 0] = 1
    cotC = np.zeros((N, 3), np.float32); cotC[1, 0] = 1
    z = lambda *s: np.zeros(s, np.float32)
    g = r.projectionScreenFusedVJP(i["scales"], i["rot"], i["means"], i["shs"], cam, cotM, z(N), cotC, z(N, 4), cotConic)
    B = A["backward"]
    np.testing.assert_allclose(_np(g["gradMeans3d"])[1], B["gradMeans3d"], rtol=5e-5)
    np.testing.assert_allclose(_np(g["gradScales"])[1], B["gradScales"], rtol=2e-3)
    np.testing.assert_allclose(_np(g["gradRot"])[1], B["gradRotations"], rtol=3e-3)
    np.testing.assert_allclose(_np(g["gradCamCenterPoint"])[1], B["gradCameraCenterPoint"], rtol=5e-5)


# --------------------------------------------------------------------------------------------- binning
@pytest.mark.parametrize("wide", [1, 0])
@pytest.mark.parametrize("W,H,tile,N", [(200, 152, (16, 16), 6000), (400, 400, (100, 100), 3000),
                                        (64, 48, (16, 16), 50), (640, 360, (16, 16), 40000), (1024, 1024, (16, 16), 20000),
                                        (1040, 1024, (16, 16), 2000), (200, 152, (16, 16), 16384), (200, 152, (16, 16), 16385),
                                        (96, 64, (16, 16), 1), (96, 64, (16, 16), 1025)])
def test_tile_bin_bit_exact(oracle32, W, H, tile, N, wide):
    """Depth sort: up to 16384 records in one workgroup (radix_sort_tiny_kernel; 16384 is its last size, 16385 the first of
    the two-launches-per-pass path), 40000 through that path.  wide = 1: the one-pass tile sort (up to 4096 tiles: 1024x1024 is exactly 4096, 1040x1024 one column more and falls
    back); wide = 0: the two 8-bit passes + range kernel.  Same lists, bit for bit."""
    p, cam = _scene(21, N, W, H, scale=0.04)
    c = cam.as_dict()
    o = oracle32
    fw = o.render_forward(p, c, W, H, tile[0], tile[1], 4)
    pr, bn = fw["proj"], fw["bin"]
    r = _renderer(W, H, tile)
    r.setTuning(wide_tile_sort=wide)
    info = r.buildGlobalTileSliceInfo((pr["rectMin"], pr["rectMax"]), pr["radii"], pr["depths"], want_dense=True)
    assert info["M"] == bn.M and info["maxTilePairs"] == bn.B
    np.testing.assert_array_equal(_np(info["sortedGaussIdx"]).astype(np.uint32), bn.sortedIdx)
    np.testing.assert_array_equal(_np(info["tileCounts"]).astype(np.uint32), bn.tileCounts)
    rng_ = _np(info["tileRanges"]).astype(np.uint32)
    nz = bn.tileCounts > 0
    np.testing.assert_array_equal(rng_[nz], bn.tileRanges[nz])
    np.testing.assert_array_equal(_np(info["packedTileIndices"]),
                                  o.build_packed_tile_indices(bn.sortedIdx, bn.tileRanges, bn.B))


def test_tile_bin_equal_depth_ties_and_empty(oracle32):
    # identical depths in one tile: order must fall back to the Gaussian index (stable sort)
    W, H, N = 64, 64, 300
    rng = np.random.default_rng(3)
    rectMin = rng.uniform(0, 40, (N, 2)).astype(np.float32)
    rectMax = (rectMin + rng.uniform(1, 20, (N, 2))).astype(np.float32)
    radii = np.where(rng.uniform(size=N) < 0.2, 0.0, 5.0).astype(np.float32)
    depths = rng.choice(np.array([1.0, 1.5, 2.0, 2.5], np.float32), N)
    bn = oracle32.tile_bin(rectMin, rectMax, radii, depths, W, H, 16, 16)
    r = _renderer(W, H)
    for wide in (1, 0):
        r.setTuning(wide_tile_sort=wide)
        info = r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths)
        assert info["M"] == bn.M
        np.testing.assert_array_equal(_np(info["sortedGaussIdx"]).astype(np.uint32), bn.sortedIdx)
        np.testing.assert_array_equal(_np(info["tileCounts"]).astype(np.uint32), bn.tileCounts)
    # nothing visible -> M = 0, all tiles empty
    info0 = r.buildGlobalTileSliceInfo((rectMin, rectMax), np.zeros(N, np.float32), depths)
    assert info0["M"] == 0 and info0["maxTilePairs"] == 0 and not _np(info0["tileCounts"]).any()
    # N = 0
    e = np.zeros((0, 2), np.float32)
    info1 = r.buildGlobalTileSliceInfo((e, e), np.zeros(0, np.float32), np.zeros(0, np.float32))
    assert info1["M"] == 0


@pytest.mark.parametrize("N", [1, 63, 64, 65, 4097, 16383, 16384, 16385])
def test_tile_bin_small_depth_sorts_with_ties(oracle32, N):
    """The depth sort of up to 16384 records is a sort by rank on the whole chip (rank_sort_kernel: 64 records per workgroup, all
    keys in LDS, sixteen waves counting a sixteenth of the keys each), from 16385 on the splitter buckets / LSD passes: the
    lists at both sides of the boundary, at workgroup and wave boundaries, with 40 % of the depths tied (the tie-break is the
    Gaussian index) and a fifth of the Gaussians invisible, against the oracle's -- and the same under the one-workgroup
    radix sort the rank sort replaces (GSPLAT_RANK_SORT=0 is an environment switch of the library: not reachable from here,
    so that form is covered by what it sorted in rounds 2-3)."""
    W, H = 96, 80
    rng = np.random.default_rng(100 + N)
    rectMin = rng.uniform(0, 70, (N, 2)).astype(np.float32)
    rectMax = (rectMin + rng.uniform(1, 24, (N, 2))).astype(np.float32)
    radii = np.where(rng.uniform(size=N) < 0.2, 0.0, 5.0).astype(np.float32)
    depths = rng.uniform(1.0, 6.0, N).astype(np.float32)
    tied = rng.uniform(size=N) < 0.4
    depths[tied] = rng.choice(np.array([1.25, 2.0, 3.5, 4.0], np.float32), int(tied.sum()))
    bn = oracle32.tile_bin(rectMin, rectMax, radii, depths, W, H, 16, 16)
    r = _renderer(W, H)
    for visit in range(2):          # (the second call has the first one's splitters where the splitter sort runs)
        info = r.buildGlobalTileSliceInfo((rectMin, rectMax), radii, depths)
        assert info["M"] == bn.M
        np.testing.assert_array_equal(_np(info["sortedGaussIdx"]).astype(np.uint32), bn.sortedIdx)
        np.testing.assert_array_equal(_np(info["tileCounts"]).astype(np.uint32), bn.tileCounts)


# ----------------------------------------------------------------------------------------------- blend
def _blend_case(oracle32, W, H, tile, white, N=5000, seed=31, scale=0.05):
    p, cam = _scene(seed, N, W, H, scale=scale)
    c = cam.as_dict()
    fw = oracle32.render_forward(p, c, W, H, tile[0], tile[1], 4, white)
    return p, c, fw


@pytest.mark.parametrize("W,H,tile,white", [(200, 152, (16, 16), False), (200, 152, (16, 16), True),
                                            (128, 96, (32, 32), False), (120, 90, (30, 30), False),
                                            (400, 400, (100, 100), False)])
@pytest.mark.parametrize("ppl", [1, 2, 4])
def test_blend_forward_backward(oracle32, W, H, tile, white, ppl):
    p, c, fw = _blend_case(oracle32, W, H, tile, white)
    pr, bn = fw["proj"], fw["bin"]
    r = _renderer(W, H, tile, white)
    r.setTuning(op_fwd_ppl=ppl, op_bwd_ppl=ppl)
    try:
        r.buildGlobalTileSliceInfo((pr["rectMin"], pr["rectMax"]), pr["radii"], pr["depths"])
        color, depth, alpha = r.globalTileComposite(fw["packed"])
        assert np.abs(_np(color) - fw["color"]).max() <= RGB_TOL
        assert np.abs(_np(alpha) - fw["alpha"]).max() <= RGB_TOL
        np.testing.assert_allclose(_np(depth), fw["depth"], rtol=1e-4, atol=1e-4)
        _ncontrib_close(_np(r._saved["last"]), fw["last"])
        rng = np.random.default_rng(8)
        cC = rng.normal(size=(W * H, 3)).astype(np.float32)
        cD = (rng.normal(size=W * H) * 0.1).astype(np.float32)
        cA = rng.normal(size=W * H).astype(np.float32)
        # same saved forward state on both sides: the oracle's
        saved = dict(packed=r._t(fw["packed"]), color=r._t(fw["color"]), depth=r._t(fw["depth"]),
                     alpha=r._t(fw["alpha"]), last=torch.as_tensor(fw["last"].astype(np.int32), device=r.device))
        got = _np(r.globalTileCompositeVJP(cC, cD, cA, saved=saved))
        want = oracle32.blend_backward(fw["packed"], bn.sortedIdx, bn.tileRanges, W, H, tile[0], tile[1], white, cC,
                                       cD, cA, fw["color"], fw["depth"], fw["alpha"], fw["last"])
        for col in range(11):
            assert _rel(got[:, col], want[:, col]) <= GRAD_RTOL, col
        # default training case: depth/alpha cotangents absent
        got0 = _np(r.globalTileCompositeVJP(cC, None, None, saved=saved))
        z = np.zeros(W * H, np.float32)
        want0 = oracle32.blend_backward(fw["packed"], bn.sortedIdx, bn.tileRanges, W, H, tile[0], tile[1], white, cC,
                                        z, z, fw["color"], fw["depth"], fw["alpha"], fw["last"])
        assert _rel(got0, want0) <= GRAD_RTOL
    finally:
        r.setTuning(op_fwd_ppl=1, op_bwd_ppl=1)


def test_blend_appendix_c_through_abi():
    b = json.load(open(os.path.join(HERE, "golden", "survey_appendix_c.json")))["blend_forward"]
    r = _renderer(b["W"], b["H"])
    # two splats, both tiles list both: rect covering the whole image reproduces the Appendix-C lists
    # (tile 0: [0,1] by depth 1.5 < 2.5; the appendix lists tile 1 as [1,0], so tile 1 is checked separately)
    packed = np.array(b["packed"], np.float32)
    rmin = np.zeros((2, 2), np.float32)
    rmax = np.tile(np.array([[b["W"] - 1, b["H"] - 1]], np.float32), (2, 1))
    r.buildGlobalTileSliceInfo((rmin, rmax), np.ones(2, np.float32), packed[:, 10])
    color, depth, alpha = r.globalTileComposite(packed)
    x, y = b["pixel"]
    pix = y * b["W"] + x
    np.testing.assert_allclose(_np(color)[pix], b["color"], rtol=5e-6, atol=1e-7)
    np.testing.assert_allclose(_np(depth)[pix], b["depth"], rtol=5e-6)
    np.testing.assert_allclose(_np(alpha)[pix], b["alpha"], rtol=5e-6)
    assert int(_np(r._saved["last"])[pix]) == b["nContrib"]


def test_blend_deep_list_early_termination(oracle32):
    # many opaque splats stacked on a few pixels: exercises multi-chunk lists, saturation and the wave-level exit
    W, H = 64, 48
    N = 3000
    rng = np.random.default_rng(4)
    packed = np.zeros((N, 11), np.float32)
    packed[:, 0] = rng.uniform(10, 50, N); packed[:, 1] = rng.uniform(8, 40, N)
    packed[:, 2] = packed[:, 5] = rng.uniform(0.002, 0.02, N)
    packed[:, 6:9] = rng.uniform(0, 1, (N, 3)); packed[:, 9] = rng.uniform(0.05, 0.6, N)
    packed[:, 10] = rng.uniform(1, 9, N)
    rmin = np.zeros((N, 2), np.float32)
    rmax = np.tile(np.array([[W - 1, H - 1]], np.float32), (N, 1))
    radii = np.ones(N, np.float32)
    bn = oracle32.tile_bin(rmin, rmax, radii, packed[:, 10], W, H, 16, 16)
    col, dep, alp, last = oracle32.blend_forward(packed, bn.sortedIdx, bn.tileRanges, W, H, 16, 16, False)
    assert last.max() < N and last.min() > 5
    r = _renderer(W, H)
    for ppl in (1, 2, 4):
        r.setTuning(op_fwd_ppl=ppl, op_bwd_ppl=ppl)
        r.buildGlobalTileSliceInfo((rmin, rmax), radii, packed[:, 10])
        c, d, a = r.globalTileComposite(packed)
        assert np.abs(_np(c) - col).max() <= RGB_TOL
        assert (np.abs(_np(r._saved["last"]).astype(np.int64) - last.astype(np.int64)) <= 1).all()
        cC = rng.normal(size=(W * H, 3)).astype(np.float32)
        saved = dict(packed=r._t(packed), color=r._t(col), depth=r._t(dep), alpha=r._t(alp),
                     last=torch.as_tensor(last.astype(np.int32), device=r.device))
        got = _np(r.globalTileCompositeVJP(cC, None, None, saved=saved))
        z = np.zeros(W * H, np.float32)
        want = oracle32.blend_backward(packed, bn.sortedIdx, bn.tileRanges, W, H, 16, 16, False, cC, z, z, col, dep,
                                       alp, last)
        assert _rel(got, want) <= GRAD_RTOL
    r.setTuning(op_fwd_ppl=1, op_bwd_ppl=1)


# ------------------------------------------------------------------------------------------------ SSIM
@pytest.mark.parametrize("H,W", [(48, 64), (37, 53), (9, 7)])
def test_ssim_forward_backward(oracle32, H, W):
    rng = np.random.default_rng(41)
    a = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    b = np.clip(a + rng.normal(0, 0.1, a.shape), 0, 1).astype(np.float32)
    up = rng.normal(size=a.shape).astype(np.float32)
    r = _renderer(64, 64)
    win = oracle32.ssim_window()
    np.testing.assert_array_equal(_np(r.ssimWindow), win)
    want = oracle32.ssim_forward(a, b)
    got = r.ssim(a, b)
    for g, w in zip(got, want):
        np.testing.assert_allclose(_np(g), w, rtol=1e-5, atol=2e-6)
    w1, w2 = oracle32.ssim_backward(up, a, b, want[1:])
    g1, g2 = r.ssimVJP(up)
    assert _rel(_np(g1), w1) <= GRAD_RTOL and _rel(_np(g2), w2) <= GRAD_RTOL


def test_loss_forward_backward(oracle32):
    H, W = 152, 200
    rng = np.random.default_rng(43)
    ren = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    tgt = np.clip(ren + rng.normal(0, 0.15, ren.shape), 0, 1).astype(np.float32)
    rd, td = rng.uniform(1, 4, (H, W)).astype(np.float32), rng.uniform(1, 4, (H, W)).astype(np.float32)
    mask = rng.uniform(size=(H, W)) > 0.5
    r = _renderer(W, H)
    loss, cc, cd, l1, ss = oracle32.loss_forward_backward(ren, tgt, 0.2)
    lo, gc, gd = r.lossForwardBackward(ren, tgt, 0.2)
    lo = _np(lo)
    assert abs(lo[0] - loss) < 2e-6 and abs(lo[1] - l1) < 2e-6 and abs(lo[2] - ss) < 2e-6
    assert _rel(_np(gc), cc) <= GRAD_RTOL and gd is None
    loss, cc, cd, l1, ss = oracle32.loss_forward_backward(ren, tgt, 0.2, rd, td, mask, 0.3)
    lo, gc, gd = r.lossForwardBackward(ren, tgt, 0.2, torch.as_tensor(rd), torch.as_tensor(td),
                                       torch.as_tensor(mask), 0.3)
    assert abs(_np(lo)[0] - loss) < 5e-6
    assert _rel(_np(gc), cc) <= GRAD_RTOL and _rel(_np(gd), cd) <= GRAD_RTOL


@pytest.mark.parametrize("H,W", [(120, 160), (37, 53), (800, 800)])
def test_loss_target_statistics_cache_is_bit_identical(H, W):
    """gs_set_loss_target_cache: the target's windowed statistics kept at the first loss of a view and read back at the
    later ones.  The cache holds the very floats the kernel computes, so loss and cotangent must be BIT-identical without
    a cache, while filling it and when reading it -- for other renders against the same target too; a key whose target
    tensor changed is refilled."""
    r = _renderer(W, H)
    g = torch.Generator().manual_seed(5)
    tgt = torch.rand(H, W, 3, generator=g).to(r.device)
    tgt2 = torch.rand(H, W, 3, generator=g).to(r.device)
    rens = [torch.rand(H, W, 3, generator=g).to(r.device) * s for s in (1.0, 0.5, 2.0)]

    def loss(ren, t, key):
        lo, gc, _ = r.lossForwardBackward(ren, t, 0.2, targetKey=key)
        return lo.clone(), gc.clone()
    want = [loss(x, tgt, None) for x in rens]
    want2 = [loss(x, tgt2, None) for x in rens]
    for i, x in enumerate(rens):                       # first call fills, the others read
        lo, gc = loss(x, tgt, "view")
        assert torch.equal(lo, want[i][0]) and torch.equal(gc, want[i][1]), i
    assert r._target_cache["view"][2] == 1
    lo, gc = loss(rens[0], tgt2, "view")               # same key, another target tensor: refilled, not reused
    assert torch.equal(lo, want2[0][0]) and torch.equal(gc, want2[0][1])
    lo, gc = loss(rens[1], tgt2, "view")
    assert torch.equal(lo, want2[1][0]) and torch.equal(gc, want2[1][1])
    lo, gc = loss(rens[2], tgt, None)                  # and no key = no cache
    assert torch.equal(lo, want[2][0]) and torch.equal(gc, want[2][1])
    r.close()


def test_target_statistics_cache_never_evicts_the_key_it_serves():
    """Round 4's advisor: a key that is REFILLED (its target changed) kept its old place in the LRU order; with the cache over
    its byte cap at that moment (the cap lowered since) the eviction loop popped that very key and the step died in a
    KeyError.  The key being served is now the most recent one before anything is evicted."""
    H, W = 37, 53
    r = _renderer(W, H)
    rng = np.random.default_rng(5)
    render = torch.as_tensor(rng.uniform(0, 1, (H, W, 3)).astype(np.float32), device=r.device)
    tg = [torch.as_tensor(rng.uniform(0, 1, (H, W, 3)).astype(np.float32), device=r.device) for _ in range(3)]
    for k in range(3):
        r.lossForwardBackward(render, tg[k], 0.2, targetKey=k)
    r.targetStatsCacheBytes = 2 * 4 * 6 * H * W + 1          # room for two views' statistics: three are held
    tg[0].add_(0.01)                                          # the OLDEST key's target rewritten in place: a refill
    lo, _, _ = r.lossForwardBackward(render, tg[0], 0.2, targetKey=0)
    want, _, _ = r.lossForwardBackward(render, tg[0], 0.2)
    assert torch.equal(lo, want) and 0 in r._target_cache and len(r._target_cache) <= 2
    r.close()


def test_colour_riders_leave_the_same_bits():
    """GS_TUNE_COLOUR_RIDERS: the SH colours of a K = 25 forward computed (0) in the projection kernel, (2) in a kernel of
    their own in front of the blend, (3) in the projection kernel behind the geometry, rows of unseen Gaussians left out,
    (1, default) by workgroups riding in the depth sort's and the tile sort's launches
    (gs_rider.h) -- where the splitter depth sort runs, i.e. from the context's second forward on and above 16384
    Gaussians.  Image, nContrib and gradients must not depend on the setting: same arithmetic, same order.  The scene
    has Gaussians behind the camera and off screen (rows the riders do not even fetch) and a count that is no multiple
    of 64."""
    W, H, N = 320, 240, 30011
    p, cam = _scene(83, N, W, H, spread=2.5)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    cot = (torch.rand(H, W, 3, generator=torch.Generator().manual_seed(4)) - 0.5)
    ref = None
    for mode in (0, 2, 3, 1):
        r = _renderer(W, H)
        r.setTuning(colour_riders=mode)
        cotd = cot.to(r.device)
        for visit in range(3):          # the first forward of a context has no splitters yet: one projection kernel
            res = r.renderForward(tp, cam)
            img, nc = res.render.clone(), r.lastContrib().clone()
            g = {k: v.clone() for k, v in r.renderBackward(cotd).items()}
            st = r.stats()
            if ref is None:
                ref = (img, nc, g, st["M"])
                assert 0 < st["N_visible"] < N          # some rows are never fetched
            assert torch.equal(img, ref[0]) and torch.equal(nc, ref[1]) and st["M"] == ref[3], (mode, visit)
            for k in g:      # (sums of float atomics: the order of the adds differs from launch to launch)
                scale = float(ref[2][k].abs().max()) + 1e-30
                assert float((g[k] - ref[2][k]).abs().max()) <= 1e-4 * scale, (mode, visit, k)
        r.close()


def test_colour_riders_at_the_bench_size():
    """The same on the bench scene (300 k Gaussians, 800x800): the third forward of a context -- riders in the splitter sort's
    launches -- against the interleaved projection kernel, image and nContrib bit for bit."""
    from gaussiansplattingmlx_amd.scenes import make_config
    params, cams, (W, H) = make_config("c3_300k_800", n_views=1)
    ref = None
    for mode in (0, 1, 3):
        r = _renderer(W, H)
        r.setTuning(colour_riders=mode)
        tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
        for _ in range(3):
            res = r.renderForward(tp, cams[0])
        got = (res.render.clone(), r.lastContrib().clone())
        if ref is None:
            ref = got
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), mode
        r.close()


@pytest.mark.parametrize("tile", [(16, 16), (50, 38)])
def test_forward_without_a_depth_image(tile):
    """gs_render_forward with out_depth NULL (renderForward(wantDepth=False), what the trainer's steps use): colour, alpha
    and nContrib are the same bits as with a depth image, the gradients of a colour cotangent too, and a depth cotangent
    for such a forward is refused."""
    from gaussiansplattingmlx_amd._lib import GsplatError
    W, H, N = 200, 152, 6000
    p, cam = _scene(77, N, W, H)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    r = _renderer(W, H, tile)
    cot = (torch.rand(H, W, 3, generator=torch.Generator().manual_seed(3)) - 0.5).to(r.device)
    res = r.renderForward(tp, cam)
    img, alpha, nc = res.render.clone(), res.alpha.clone(), r.lastContrib().clone()
    assert res.depth is not None and float(res.depth.abs().max()) > 0
    g0 = {k: v.clone() for k, v in r.renderBackward(cot).items()}
    res = r.renderForward(tp, cam, wantDepth=False)
    assert res.depth is None
    assert torch.equal(res.render, img) and torch.equal(res.alpha, alpha) and torch.equal(r.lastContrib(), nc)
    with pytest.raises(GsplatError):
        r.renderBackward(cot, cotDepth=torch.ones(H, W, device=r.device))
    res = r.renderForward(tp, cam, wantDepth=False)
    g1 = r.renderBackward(cot)
    for k in g0:      # (sums of float atomics: the order of the adds differs from launch to launch)
        scale = float(g0[k].abs().max()) + 1e-30
        assert float((g1[k] - g0[k]).abs().max()) <= 1e-4 * scale, k
    r.close()


# ------------------------------------------------------------------------------------- fused end to end
@pytest.mark.parametrize("W,H,tile,N,white", [(200, 152, (16, 16), 6000, False), (200, 152, (16, 16), 6000, True),
                                              (400, 400, (100, 100), 3000, False), (800, 800, (200, 200), 1500, False)])
def test_fused_render_forward_backward(oracle32, W, H, tile, N, white):
    """(800, 800, (200, 200)): the tile size the reference APP constructs its renderer with, TILE_SIZE = (W/4, H/4)
    (Data/ColmapDataLoader.swift:495-498, UI/TrainView.swift:184-190) -- not a multiple of 16, so the generic blend
    kernels (blend.hip) behind the same fused entry points; `bench.py --tile 200` times them."""
    from gaussiansplattingmlx_amd.scenes import perturb
    p, cam = _scene(51, N, W, H)
    c = cam.as_dict()
    o = oracle32
    fw = o.render_forward(p, c, W, H, tile[0], tile[1], 4, white)
    tgt = o.render_forward(perturb(p, 99), c, W, H, tile[0], tile[1], 4, white)["color"].reshape(H, W, 3)
    r = _renderer(W, H, tile, white)
    res = r.renderForward({k: torch.as_tensor(v) for k, v in p.items()}, cam, want_radii=True)
    st = r.stats()
    assert st["N_visible"] == int((fw["proj"]["radii"] > 0).sum())
    if tile[0] % 16 == 0 and tile[1] % 16 == 0:
        _pairs_match(r, fw["bin"].M)
    else:       # block lists (include/gsplat.h): the fused path bins per 16 x 16 block of a tile, M counts (Gaussian, block) pairs
        assert st["M"] >= fw["bin"].M
    img = _np(res.render)
    assert np.abs(img.reshape(-1, 3) - fw["color"]).max() <= RGB_TOL
    np.testing.assert_array_equal(_np(res.radii), fw["proj"]["radii"])
    # loss on the HIP image vs the oracle's loss on the oracle image
    loss, cc, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)
    lo, gc, _ = r.lossForwardBackward(res.render, tgt, 0.2)
    assert abs(_np(lo)[0] - loss) < 1e-5
    want = o.render_backward(p, c, W, H, tile[0], tile[1], 4, fw, cc.reshape(-1, 3), np.zeros(W * H, np.float32),
                             np.zeros(W * H, np.float32), white)
    got = r.renderBackward(gc)
    for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
        assert _rel(_np(got[k]), want[k].reshape(_np(got[k]).shape)) <= GRAD_RTOL, k
    # arbitrary cotangents incl. depth and alpha
    rng = np.random.default_rng(6)
    cC, cD, cA = (rng.normal(size=(W * H, 3)).astype(np.float32), rng.normal(size=W * H).astype(np.float32) * 0.1,
                  rng.normal(size=W * H).astype(np.float32))
    # hand the oracle the HIP forward's saved state so both sides undo the same T
    fw2 = dict(fw); fw2["alpha"] = _np(res.alpha).reshape(-1)
    want = o.render_backward(p, c, W, H, tile[0], tile[1], 4, fw2, cC, cD, cA, white)
    got = r.renderBackward(cC, cD, cA)
    for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
        assert _rel(_np(got[k]), want[k].reshape(_np(got[k]).shape)) <= GRAD_RTOL, k


@pytest.mark.parametrize("W,H,tile,N,white", [(250, 170, (100, 70), 3000, False), (130, 100, (24, 40), 2500, True),
                                              (200, 152, (50, 38), 4000, False), (97, 61, (200, 200), 800, False),
                                              (1237, 822, (310, 206), 5000, False)])
def test_block_lists_match_the_oracle_at_odd_tile_sizes(oracle32, oracle64, W, H, tile, N, white):
    """Tile sizes that are not multiples of 16 run the fused kernels on BLOCK lists (include/gsplat.h, gs_ctx.h GsVirtGeom):
    the 16 x 16 blocks are enumerated per tile (last column / row of a tile narrower), a block's list holds the Gaussians of
    its tile's list that can reach it.  Cases: a last tile cut by the image in both directions with a tile height that leaves
    a 6-row last block; tiles of 24 x 40 (16 + 8 columns, 16 + 16 + 8 rows) on a white background; 50 x 38; one tile larger
    than the image; the garden image at the app's W/4 x H/4 tiles (80 x 52 = 4160 blocks: beyond the one-pass tile sort's 4096
    bins, the key + value sort).  Image, depth, alpha and every gradient (with depth and alpha cotangents) against the oracle at that tile
    size; a second visit of the view under its hints and depth cuts gives the same bits; GSPLAT_BLOCK_LISTS=0 (the generic
    kernels scanning the tile's list per block) stays within the same bars."""
    from gaussiansplattingmlx_amd.scenes import perturb
    p, cam = _scene(77, N, W, H)
    p["features_rest"] *= 0.3
    c = cam.as_dict()
    o = oracle32
    fw = o.render_forward(p, c, W, H, tile[0], tile[1], 4, white)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    rng = np.random.default_rng(8)
    cC, cD, cA = (rng.normal(size=(W * H, 3)).astype(np.float32), rng.normal(size=W * H).astype(np.float32) * 0.1,
                  rng.normal(size=W * H).astype(np.float32))

    def run(r, key=None):
        res = r.renderChecked(tp, cam, want_radii=True, viewKey=key)
        img, dep, alp = _np(res.render).reshape(-1, 3).copy(), _np(res.depth).reshape(-1).copy(), _np(res.alpha).reshape(-1).copy()
        g = {k: _np(v).copy() for k, v in r.renderBackward(cC, cD, cA).items()}
        return img, dep, alp, g, _np(res.radii).copy()

    r = _renderer(W, H, tile, white)
    img, dep, alp, g, radii = run(r, key=0)
    assert np.abs(img - fw["color"]).max() <= RGB_TOL
    assert np.abs(alp - fw["alpha"]).max() <= RGB_TOL
    np.testing.assert_allclose(dep, fw["depth"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(radii, fw["proj"]["radii"])
    fw2 = dict(fw); fw2["alpha"] = alp
    want = o.render_backward(p, c, W, H, tile[0], tile[1], 4, fw2, cC, cD, cA, white)
    for k in GRAD_KEYS:
        assert _rel(g[k], want[k].reshape(g[k].shape)) <= GRAD_RTOL, k
    # ... and element by element against what the float32 / float64 oracle pair holds at this tile size (colour cotangent of
    # the L1 / DSSIM loss, no depth or alpha cotangent)
    if not white and W * H <= 50000:
        tgt = o.render_forward(perturb(p, 99), c, W, H, tile[0], tile[1], 4, white)["color"].reshape(H, W, 3)
        _, cc, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)
        w32 = o.render_backward(p, c, W, H, tile[0], tile[1], 4, fw, cc.reshape(-1, 3), np.zeros(W * H, np.float32), np.zeros(W * H, np.float32), white)
        r.renderChecked(tp, cam, viewKey=0)
        lo, gc, _ = r.lossForwardBackward(r._fused["color"].view(H, W, 3), tgt, 0.2)
        gl = {k: v.clone() for k, v in r.renderBackward(gc).items()}
        _elementwise_gradient_bar(f"block_lists_{W}x{H}_tile{tile[0]}x{tile[1]}", gl, w32, oracle64, p, c, W, H, tgt, tile=tile)
    # the view's second and third visit: launch order from its hints, then (forced) its depth cuts -- the same image bits
    r.cutMinDropped = 0
    for visit in range(2):
        img2, dep2, alp2, g2, _ = run(r, key=0)
        np.testing.assert_array_equal(img2, img)
        np.testing.assert_array_equal(alp2, alp)
        for k in GRAD_KEYS:
            assert _rel(g2[k], want[k].reshape(g2[k].shape)) <= GRAD_RTOL, (visit, k)
    r.close()
    # round 3's form of the same entry points
    os.environ["GSPLAT_BLOCK_LISTS"] = "0"
    try:
        r0 = _renderer(W, H, tile, white)
        img0, dep0, alp0, g0, _ = run(r0)
        r0.close()
    finally:
        del os.environ["GSPLAT_BLOCK_LISTS"]
    assert np.abs(img0 - fw["color"]).max() <= RGB_TOL
    assert np.abs(img0 - img).max() <= 2e-5
    for k in GRAD_KEYS:
        assert _rel(g0[k], want[k].reshape(g0[k].shape)) <= GRAD_RTOL, k


def test_block_lists_overflow_is_reported_and_regrown():
    """The reserved-capacity contract (test_reserved_overflow_is_reported_and_never_applied) on block lists: the pair count a
    reserve has to hold is the count of (Gaussian, block) pairs -- more than the tile lists' -- and an overflow reports THAT
    count, renders nothing, and is regrown from by the trainer."""
    from gaussiansplattingmlx_amd._lib import GsplatError
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H, N, tile = 200, 152, 6000, (50, 38)
    p, cam = _scene(81, N, W, H)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    r0 = _renderer(W, H, tile)
    ref = r0.renderForward(tp, cam)
    M = r0.stats()["M"]
    tgt = ref.render.clone()
    r = _renderer(W, H, tile)
    r.reserve(N, M // 3)                                   # too small on purpose
    res = r.renderForward(tp, cam)
    with pytest.raises(GsplatError) as ei:
        r.sync()
    assert ei.value.code == 3 and str(M) in str(ei.value)
    assert r.stats()["overflow"] == 1 and r.stats()["M"] == M and not bool(res.render.any())
    model = GaussModel(p, r.device)
    tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
    loss = tr.trainStep(cam, tgt, viewKey=0)
    assert tr.overflowRecoveries == 1 and r.stats()["overflow"] == 0 and r.stats()["M"] == M
    assert np.isfinite(float(loss[0])) and float(loss[0]) < 1e-3          # (the target is this very render)
    r.close(); r0.close()


@pytest.mark.parametrize("W,H,N", [(200, 152, 6000), (640, 600, 20000)])
def test_render_only_forward_is_the_same_image_and_refuses_a_backward(W, H, N):
    """GS_TUNE_RENDER_ONLY: the fused forward keeps no checkpoints (both forward kernels: four waves per quadrant at 200 x 152,
    one at 640 x 600).  Image, depth, alpha and nContrib are the normal forward's bits; a backward of such a forward is
    refused (GS_ERR_NO_FORWARD), and switching the knob off again gives a differentiable forward."""
    from gaussiansplattingmlx_amd._lib import GsplatError
    p, cam = _scene(91, N, W, H)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    r = _renderer(W, H)
    ref = r.renderForward(tp, cam)
    img, dep, alp, nc = ref.render.clone(), ref.depth.clone(), ref.alpha.clone(), r.lastContrib().clone()
    assert int(nc.max()) > 64                                     # (lists deep enough to have checkpoints at all)
    cot = torch.ones(W * H, 3, device=r.device)
    g0 = {k: v.clone() for k, v in r.renderBackward(cot).items()}
    r.setTuning(render_only=1)
    res = r.renderForward(tp, cam)
    assert torch.equal(res.render, img) and torch.equal(res.depth, dep) and torch.equal(res.alpha, alp) and torch.equal(r.lastContrib(), nc)
    with pytest.raises(GsplatError) as ei:
        r.renderBackward(cot)
    assert ei.value.code == 5 and "render-only" in str(ei.value)
    lo, gc, _ = r.lossForwardBackward(res.render, img, 0.2)       # the loss of a render-only forward carries no backward preparation
    with pytest.raises(GsplatError):
        r.renderBackward(gc)
    r.setTuning(render_only=0)
    r.renderForward(tp, cam)
    g1 = r.renderBackward(cot)
    for k in g0:
        assert _rel(_np(g1[k]), _np(g0[k])) <= 1e-4, k
    r.close()


def test_error_behaviour():
    from gaussiansplattingmlx_amd._lib import GsplatError
    r = _renderer(64, 48)
    with pytest.raises(GsplatError):          # backward without forward
        r._fused = dict(params={k: torch.zeros(1, device=r.device) for k in
                                ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity")})
        r.renderBackward(torch.zeros(64 * 48, 3))
    with pytest.raises(GsplatError):          # image size mismatch (reference precondition, GaussianRenderer.swift:789)
        z = torch.zeros(4, 2)
        r.render(32, 32, z, None, None, None, None, torch.zeros(4), None, (z, z))
    with pytest.raises(GsplatError):          # blend without binning
        r2 = _renderer(64, 48)
        r2.globalTileComposite(torch.zeros(4, 11))



def _assert_lists_ordered_and_complete(oracle32, r, params, cam, radii, idx_n, rng_n, cnt_n, W, H, n_sample=300, seed=0):
    """What the full-size property tests say about the tile lists, checked on `n_sample` tiles: inside a tile
    (depth bits, Gaussian index) strictly increases (the reference's stable sort of (tile, depth) keys emitted in
    Gaussian order, slang/gaussian_tile_global_kernels.slang:73-126, 151-305), and the tile's Gaussians are exactly those
    whose tile rect covers it (count_tiles_per_gaussian, :17-58).  Depth and screen position depend on xyz and the camera
    only and are bit-exact between the oracle and the library (test_projection_forward_backward), so they come from the
    oracle; the radius goes through exp / sigmoid (libm vs device: a ceil() edge may differ), so it is the library's own
    (renderForward(want_radii=True)) -- what is checked is the binning, not the projection."""
    c = cam.as_dict()
    o = oracle32
    N = params["xyz"].shape[0]
    op, sc, rt = o.activations_forward(params["opacity"], params["scales"], params["rotation"])
    shs = np.ascontiguousarray(params["features_dc"])                    # the colour is not looked at: degree 0, K = 1
    pr = o.projection_forward(sc, rt, params["xyz"], shs, c["camCenter"], c["view"], c["proj"], c["fovX"], c["fovY"],
                              c["focalX"], c["focalY"], W, H, 0)
    keys = np.ascontiguousarray(pr["depths"], np.float32).view(np.uint32).astype(np.int64)
    m2d = pr["means2d"].astype(np.float32)
    radii = np.asarray(radii, np.float32)
    # the rect of kernels.slang:158-172 from the screen position and the radius, in f32
    vis = radii > 0
    m2d = np.where(vis[:, None], m2d, np.float32(0.0))          # (points on the camera plane project to inf / NaN: invisible)
    rmin = np.maximum(m2d - radii[:, None], np.float32(0.0))
    rmax = np.minimum(m2d + radii[:, None], np.array([W - 1.0, H - 1.0], np.float32))
    gw, gh = (W + 15) // 16, (H + 15) // 16
    f = lambda a: np.floor(np.nan_to_num(a / np.float32(16.0), nan=0.0, posinf=1e9, neginf=-1e9)).astype(np.int64)
    x0, y0 = np.clip(f(rmin[:, 0]), 0, gw), np.clip(f(rmin[:, 1]), 0, gh)
    x1, y1 = np.clip(f(rmax[:, 0]) + 1, 0, gw), np.clip(f(rmax[:, 1]) + 1, 0, gh)
    trimmed = _lists_trimmed(r)
    if trimmed:
        assert int(cnt_n.sum()) <= int(((x1 - x0) * (y1 - y0))[vis].sum())
    else:
        assert int(((x1 - x0) * (y1 - y0))[vis].sum()) == int(cnt_n.sum())      # M itself
    con = np.asarray(pr["conic"], np.float64).reshape(N, 4)
    rng = np.random.default_rng(seed)
    has = np.nonzero(cnt_n > 0)[0]
    tiles = rng.choice(has, min(n_sample, has.size), replace=False)
    deepest = has[np.argsort(cnt_n[has])[-8:]]                                  # always include the longest lists
    for t in np.unique(np.concatenate([tiles, deepest])):
        s, e = rng_n[t]
        lst = idx_n[s:e]
        k = keys[lst]
        assert np.all((k[1:] > k[:-1]) | ((k[1:] == k[:-1]) & (lst[1:] > lst[:-1]))), f"tile {t}: list out of order"
        ty, tx = divmod(int(t), gw)
        want = np.nonzero(vis & (x0 <= tx) & (tx < x1) & (y0 <= ty) & (ty < y1))[0]
        if not trimmed:
            assert np.array_equal(np.sort(lst), want), f"tile {t}: wrong Gaussians"
            continue
        # trimmed rects (GS_TUNE_TRIM_RECTS): a sub-list of the reference's, and what was left out reaches no pixel of the tile
        # -- its quadratic form stays above the blend's cull bound (gs_cull.h: q > 40, weight < 2^-29) on every pixel centre
        assert np.isin(lst, want).all(), f"tile {t}: a Gaussian whose rect does not cover the tile"
        out = np.setdiff1d(want, lst)
        if out.size:
            px = np.arange(16 * tx, min(16 * tx + 16, W), dtype=np.float64)
            py = np.arange(16 * ty, min(16 * ty + 16, H), dtype=np.float64)
            dx = px[None, :, None] - m2d[out, 0].astype(np.float64)[:, None, None]
            dy = py[None, None, :] - m2d[out, 1].astype(np.float64)[:, None, None]
            cc = con[out]
            q = cc[:, 0, None, None] * dx * dx + (cc[:, 1] + cc[:, 2])[:, None, None] * dx * dy + cc[:, 3, None, None] * dy * dy
            assert q.reshape(out.size, -1).min(axis=1).min() > 40.0, f"tile {t}: a Gaussian left out that reaches a pixel"
    return len(tiles)


# ----------------------------------------------------------- full-size properties (BASELINE configs[1])
def test_full_size_properties(oracle32):
    from gaussiansplattingmlx_amd.scenes import make_config
    params, cams, (W, H) = make_config("c2_100k_800", n_views=1)
    r = _renderer(W, H)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
    res = r.renderForward(tp, cams[0], want_radii=True)
    radii = _np(res.radii)
    img1 = res.render.clone()
    st = r.stats()
    assert st["M"] > 0 and st["overflow"] == 0
    # binning invariants at full size: ranges partition [0, M), lists sorted by (depth, index)
    M, T = st["M"], ((W + 15) // 16) * ((H + 15) // 16)
    idx = torch.empty(M, dtype=torch.int32, device=r.device)
    rng_ = torch.empty(T, 2, dtype=torch.int32, device=r.device)
    cnt = torch.empty(T, dtype=torch.int32, device=r.device)
    import ctypes as C
    r._check(r.lib.gs_tile_bin_export(r.ctx, C.c_void_p(idx.data_ptr()), C.c_void_p(rng_.data_ptr()),
                                      C.c_void_p(cnt.data_ptr())))
    cnt_n, rng_n, idx_n = _np(cnt).astype(np.int64), _np(rng_).astype(np.int64), _np(idx).astype(np.int64)
    assert cnt_n.sum() == M
    nz = cnt_n > 0
    assert (rng_n[nz, 1] - rng_n[nz, 0] == cnt_n[nz]).all()
    starts = np.sort(rng_n[nz, 0]); ends = np.sort(rng_n[nz, 1])
    assert starts[0] == 0 and ends[-1] == M and (starts[1:] == ends[:-1]).all()
    assert _assert_lists_ordered_and_complete(oracle32, r, params, cams[0], radii, idx_n, rng_n, cnt_n, W, H) >= 300
    # determinism of the forward (no atomics on that path) and idempotence
    res2 = r.renderForward(tp, cams[0])
    assert torch.equal(img1, res2.render)
    # linearity of the backward in the cotangent: g(a*c1 + c2) == a*g(c1) + g(c2)
    g = torch.Generator(device="cpu").manual_seed(1)
    c1 = torch.randn(W * H, 3, generator=g).to(r.device)
    c2 = torch.randn(W * H, 3, generator=g).to(r.device)
    g1 = {k: v.clone() for k, v in r.renderBackward(c1).items()}
    g2 = {k: v.clone() for k, v in r.renderBackward(c2).items()}
    g3 = r.renderBackward(2.5 * c1 + c2)
    for k in g1:
        ref = 2.5 * g1[k] + g2[k]
        assert (g3[k] - ref).abs().max() <= 2e-3 * ref.abs().max() + 1e-12, k


def test_forward_queue_count_leaves_the_same_bits():
    """GS_TUNE_FWD_QUEUES (round 4: eight work queues, one per XCD, instead of one) changes which wave sweeps which quadrant
    and when, nothing else: image, alpha, nContrib and the per-block sweep lengths are the same bits for 1, 2, 4 and 8 queues,
    with and without a view hint (deepest-first launch order), also on an image whose block count is no multiple of 8."""
    W, H, N = 200, 152, 6000            # 13 x 10 = 130 pixel blocks
    p, cam = _scene(23, N, W, H)
    out = {}
    for nq in (1, 2, 4, 8):
        r = _renderer(W, H)
        r.setTuning(fwd_queues=nq)
        tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
        for visit in range(2):          # the second visit runs in the order the first one's sweep lengths give
            res = r.renderForward(tp, cam, viewKey=0)
            out[(nq, visit)] = (res.render.clone(), res.alpha.clone(), r.lastContrib().clone())
        r.close()
    for k, v in out.items():
        for a, b in zip(v, out[(1, 0)]):
            assert torch.equal(a, b), k
    with pytest.raises(Exception):
        _renderer(W, H).setTuning(fwd_queues=3)


@pytest.mark.parametrize("W,H,N,scale", [(200, 152, 6000, 0.05), (640, 600, 20000, 0.03), (400, 400, 10000, 0.12)])
def test_staging_wave_forward_leaves_the_same_bits(W, H, N, scale):
    """GS_TUNE_FWD_PAIR (round 6: blend_fwd_v2p_kernel -- a second wave per quadrant loads, culls and compacts chunk c + 1 into
    LDS while the first blends chunk c; one workgroup barrier per chunk) changes who prepares a chunk, not what is blended nor
    in which order: image, depth, alpha, nContrib, the per-block sweep lengths and -- through the checkpoints it leaves -- the
    gradients are those of the one-wave kernel, for 12, 14 and 3 workgroups per CU (3: every workgroup takes many items), with
    a view hint, with and without a depth image, on deep lists (the third scene: ~1000 entries per tile, most culled per
    quadrant)."""
    p, cam = _scene(29, N, W, H, scale=scale)
    rng = np.random.default_rng(4)
    cC = rng.normal(size=(H, W, 3)).astype(np.float32)
    out = {}
    for pair in (0, 1, 14, 3):
        r = _renderer(W, H)
        r.setTuning(fwd_four_waves=0, fwd_pair=pair)
        tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
        for visit in range(2):
            res = r.renderForward(tp, cam, viewKey=0, wantDepth=visit == 0)
            g = r.renderBackward(cC)
            out[(pair, visit)] = (res.render.clone(), res.alpha.clone(), r.lastContrib().clone(), r.blockWork().clone(),
                                  None if res.depth is None else res.depth.clone(), {k: _np(v).copy() for k, v in g.items()})
        assert r.stats()["overflow"] == 0
        r.close()
    for (pair, visit), v in out.items():
        ref = out[(0, visit)]
        for a, b in zip(v[:4], ref[:4]):
            assert torch.equal(a, b), (pair, visit)
        assert (v[4] is None) == (ref[4] is None) and (v[4] is None or torch.equal(v[4], ref[4])), (pair, visit)
        for k in GRAD_KEYS:          # (float atomics: not the same bits run to run)
            assert _rel(v[5][k], ref[5][k]) <= 1e-4, (pair, visit, k)
    with pytest.raises(Exception):
        _renderer(W, H).setTuning(fwd_pair=17)


def test_four_waves_per_quadrant_forward_against_the_one_wave_forward():
    """blend_fwd_v2w_kernel (images with fewer quadrants than wave slots; GS_TUNE_FWD_FOUR_WAVES) against blend_fwd_v2q_kernel on a
    dense scene whose pixels finish at all depths of lists of ~1100 entries (rounds of four chunks with pixels crossing
    T < 1e-4 inside a part that was swept from T = 1): image, depth and alpha within 1e-5 of the largest value (sums composed
    across chunks instead of accumulated), nContrib the same but for pixels on the threshold, gradients of a random
    cotangent within 1e-4 -- and the four-wave forward itself the same bits at every visit, with 1, 2, 4 or 8 queues, hinted
    or not (the parts depend on the list position only)."""
    W, H, N = 160, 120, 20000
    p, cam = _scene(77, N, W, H, spread=0.5, scale=0.12)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    rng = np.random.default_rng(3)
    cot = [torch.as_tensor(rng.normal(size=s).astype(np.float32)) for s in ((H * W, 3), (H * W,), (H * W,))]
    out = {}
    for mode in (0, 1):
        r = _renderer(W, H)
        r.setTuning(fwd_four_waves=mode)
        res = r.renderForward(tp, cam)
        nc = r.lastContrib().clone()
        g = r.renderBackward(*cot)
        out[mode] = (res.render.clone(), res.depth.clone(), res.alpha.clone(), nc, {k: v.clone() for k, v in g.items()})
        if mode == 1:
            assert int(nc.max()) > 256          # lists deep enough for several rounds
            for nq, key in ((1, None), (2, "v"), (4, "v"), (8, "v"), (8, "v")):
                r.setTuning(fwd_queues=nq)
                again = r.renderForward(tp, cam, viewKey=key)
                assert torch.equal(again.render, res.render) and torch.equal(r.lastContrib(), nc), (nq, key)
        r.close()
    a, b = out[0], out[1]
    for i, name in enumerate(("render", "depth", "alpha")):
        scale = float(a[i].abs().max())
        assert float((a[i] - b[i]).abs().max()) <= 1e-5 * max(scale, 1.0), name
    diff = (a[3] != b[3])
    assert float(diff.float().mean()) <= 1e-3 and int((a[3].long() - b[3].long()).abs().max()) <= 64
    for k in a[4]:
        ga, gb = a[4][k].double(), b[4][k].double()
        assert float((ga - gb).abs().max()) <= 1e-4 * float(ga.abs().max()), k


@pytest.mark.parametrize("scale_permille", [1000, 900, 500, 50])
def test_four_wave_forward_pixels_that_come_back_live_from_their_second_take(oracle32, scale_permille):
    """Round 4's advisor finding on blend_fwd_v2w_kernel: the fold reads "this pixel crosses T < 1e-4 inside part w" off the
    COMPOSED product T_prefix x T_part, the second take of the part multiplies in sequence and rounds differently, so a pixel
    within ~1e-6 of the threshold can come back from it still live -- and round 4's kernel had by then cut it out of the
    round's later parts (their entries skipped for it, their checkpoint lanes never written, the next round blending on): a
    backward that reads checkpoint lanes no forward wrote.  Once in ~1e7 pixel-parts by itself, so the test forces it:
    GS_TUNE_FWD_FOLD_TEST_SCALE scales the composed product in that one test, which sends every pixel whose T at a part's end
    lies within [1e-4, 1e-4 / scale) through a second take that it survives (at 0.05: every pixel below 2e-3, hundreds per
    image, many of them twice or three times in one round), and GS_TUNE_POISON_CHECKPOINTS fills the checkpoint arena with NaN
    in front of the forward, so that any checkpoint lane the backward reads and this forward did not write makes a NaN
    gradient.  Against the ORACLE at the suite's bars (image, nContrib, all six gradients), and against the one-wave
    kernel; scale 1000 = the shipped arithmetic under the same poison."""
    from gaussiansplattingmlx_amd.scenes import perturb
    W, H, N = 160, 120, 20000
    p, cam = _scene(77, N, W, H, spread=0.5, scale=0.12)
    p["features_rest"] *= 0.05
    c = cam.as_dict()
    o = oracle32
    fw = o.render_forward(p, c, W, H, 16, 16, 4, False)
    tgt = o.render_forward(perturb(p, 5), c, W, H, 16, 16, 4, False)["color"].reshape(H, W, 3)
    loss, cc, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)
    z = np.zeros(W * H, np.float32)
    want = o.render_backward(p, c, W, H, 16, 16, 4, fw, cc.reshape(-1, 3), z, z, False)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    got = {}
    for mode, permille in ((1, scale_permille), (0, 1000)):
        r = _renderer(W, H)
        r.setTuning(fwd_four_waves=mode, fwd_fold_test_scale=permille, poison_checkpoints=1)
        res = r.renderForward(tp, cam)
        nc = _np(r.lastContrib())
        assert int(nc.max()) > 256                     # several rounds of four parts
        assert np.abs(_np(res.render).reshape(-1, 3) - fw["color"]).max() <= RGB_TOL
        _ncontrib_match(r, fw, W, H, nc)
        lo, gc, _ = r.lossForwardBackward(res.render, tgt, 0.2)
        g = r.renderBackward(gc)
        for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
            a = _np(g[k])
            assert np.isfinite(a).all(), (k, mode)      # a NaN = a checkpoint lane nobody wrote
            assert _rel(a, want[k].reshape(a.shape)) <= GRAD_RTOL, (k, mode)
        got[mode] = (nc, {k: _np(v).astype(np.float64) for k, v in g.items()})
        r.close()
    assert float((got[0][0] != got[1][0]).mean()) <= 1e-3
    for k in got[0][1]:
        assert np.abs(got[0][1][k] - got[1][1][k]).max() <= 1e-4 * np.abs(got[0][1][k]).max(), k
    with pytest.raises(Exception):
        _renderer(W, H).setTuning(fwd_fold_test_scale=0)


# ------------------------------------------------------------------------------ next row: Adam + train step
def test_adam_step_matches_numpy():
    import ctypes as C
    r = _renderer(64, 48)
    rng = np.random.default_rng(9)
    n = 10007                                    # odd length: exercises the scalar tail
    p, g = rng.normal(size=n).astype(np.float32), rng.normal(size=n).astype(np.float32)
    m, v = np.zeros(n, np.float32), np.zeros(n, np.float32)
    seg_end = np.array([1000, 4000, 4001, 9000, 10000, n], np.int64)
    lrs = np.array([1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 2.5e-2], np.float32)
    tp, tg, tm, tv = (torch.as_tensor(a, device=r.device) for a in (p, g, m, v))
    b1, b2, eps, scale = 0.9, 0.999, 1e-15, 0.5
    lr_el = np.zeros(n, np.float32)
    prev = 0
    for e, lr in zip(seg_end, lrs):
        lr_el[prev:e] = lr
        prev = e
    for _ in range(3):
        r._check(r.lib.gs_adam_step(r.ctx, n, C.c_void_p(tp.data_ptr()), C.c_void_p(tg.data_ptr()),
                                    C.c_void_p(tm.data_ptr()), C.c_void_p(tv.data_ptr()), 6,
                                    seg_end.ctypes.data_as(C.c_void_p), lrs.ctypes.data_as(C.c_void_p),
                                    C.c_float(b1), C.c_float(b2), C.c_float(eps), C.c_float(scale)))
        gs = g * np.float32(scale)
        # (1 - beta) in f32, as mlx-swift's Adam computes it from its Float betas
        one = np.float32(1)
        m = np.float32(b1) * m + (one - np.float32(b1)) * gs
        v = np.float32(b2) * v + (one - np.float32(b2)) * gs * gs
        p = p - lr_el * m / (np.sqrt(v) + np.float32(eps))
    np.testing.assert_allclose(_np(tp), p, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(_np(tm), m, rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(_np(tv), v, rtol=2e-6, atol=1e-12)


def test_train_steps_reduce_the_loss(oracle32):
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H, N = 160, 120, 4000
    p, cam = _scene(61, N, W, H, scale=0.06)
    p["features_rest"] *= 0.05                       # colours of order 1: a realistic photometric loss
    tgt = oracle32.render_forward(perturb(p, 5, 0.1), cam.as_dict(), W, H, 16, 16, 4)["color"].reshape(H, W, 3)
    r = _renderer(W, H)
    model = GaussModel(p, r.device)
    tr = GaussianTrainer(model, r, iterationCount=1000)
    target = torch.as_tensor(tgt, device=r.device)
    losses = []
    for _ in range(40):
        losses.append(float(tr.trainStep(cam, target)[0]))
    assert np.isfinite(losses).all()
    assert losses[-1] < 0.8 * losses[0], (losses[0], losses[-1])
    # first-step loss equals the oracle's loss on the oracle's image
    fw = oracle32.render_forward(p, cam.as_dict(), W, H, 16, 16, 4)
    want = oracle32.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)[0]
    assert abs(losses[0] - want) < 1e-5


# ------------------------------------------------------------------------------------------ data-parallel exchange
@pytest.mark.parametrize("degree,K", [(4, 25), (2, 25), (0, 1), (3, 16)])
def test_sh_compressed_backward_matches_summed_view_gradients(degree, K):
    """gs_render_backward_dp + gs_sh_grad_from_views over R = 3 views == sum of gs_render_backward over the views."""
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    W, H, N = 176, 128, 3001                       # N not a multiple of the 128-thread block nor of 64
    p, cam0 = _scene(71, N, W, H, K=K)
    cams = [cam0, Camera(W, H, 0.8 * W, 0.8 * W, look_at_c2w([-2.4, 1.9, 1.2])),
            Camera(W, H, 1.1 * W, 1.1 * W, look_at_c2w([0.4, 2.9, -1.6]))]
    r = _renderer(W, H, degree=degree)
    rng = np.random.default_rng(5)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
    want = {k: np.zeros_like(v, np.float64) for k, v in p.items()}
    ccs, geom = [], ("xyz", "scales", "rotation", "opacity")
    for cam in cams:
        cot = torch.as_tensor(rng.normal(0, 1, (H, W, 3)).astype(np.float32), device=r.device)
        r.renderForward(tp, cam)
        g = r.renderBackward(cot)
        for k in want:
            want[k] += _np(g[k]).astype(np.float64)
        g2, cc = r.renderBackwardDP(cot)
        for k in geom:                             # same kernel code; only the blend's atomic summation order differs
            np.testing.assert_allclose(_np(g2[k]), _np(g[k]), rtol=1e-3, atol=1e-5 * np.abs(_np(g[k])).max())
        ccs.append(cc.clone())
    sh = r.shGradFromViews(tp["xyz"], torch.stack(ccs), np.stack([c.cameraCenter for c in cams]), K)
    for k in ("features_dc", "features_rest"):
        got = _np(sh[k])
        assert got.shape == p[k].shape
        if want[k].size:
            assert np.abs(want[k]).max() > 0
            np.testing.assert_allclose(got, want[k], rtol=1e-3, atol=1e-5 * np.abs(want[k]).max())
    if K > (degree + 1) ** 2:                      # inactive bands get exactly zero, as in the direct path
        assert not _np(sh["features_rest"])[:, (degree + 1) ** 2 - 1:, :].any()


def test_sh_grad_from_views_rejects_bad_arguments():
    r = _renderer(64, 64)
    x = torch.zeros(8, 3, device=r.device)
    cc = torch.zeros(17, 8, 3, device=r.device)
    with pytest.raises(Exception):
        r.shGradFromViews(x, cc, np.zeros((17, 3), np.float32), 25)          # R > 16
    with pytest.raises(Exception):
        r.shGradFromViews(x, cc[:1], np.zeros((1, 3), np.float32), 9)         # K < (degree+1)^2


@pytest.mark.parametrize("workload", ["small", "c4_300k_800"])
def test_trainer_exchanges_agree_on_a_one_rank_rccl_group(oracle32, workload):
    """Runs the real collectives (RCCL, 1-rank group) of both exchanges -- the gradient all-reduce, the colour-cotangent
    all-gather + geometry all-reduce, each with the step's gate word riding in it -- and checks they leave the same parameters as
    the exchange-free step.  "c4_300k_800" is BASELINE configs[3]'s per-rank workload (the bench scene, 300 k Gaussians,
    800x800: a 103-MB arena, 28.8-MB gathers at 8 ranks) on the one rank this box has; the 8-rank run itself needs the
    8-GPU node (bench.py --gpus 8)."""
    import os
    import socket
    import torch.distributed as dist
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    if workload == "small":
        W, H, N = 160, 120, 3000
        p, cam = _scene(62, N, W, H, scale=0.06)
        r = _renderer(W, H)
        target = torch.rand(H, W, 3, device=r.device)
    else:
        from gaussiansplattingmlx_amd.scenes import make_config, perturb
        p, cams, (W, H) = make_config("c3_300k_800", n_views=1)
        cam = cams[0]
        r = _renderer(W, H)
        r.reserve(p["xyz"].shape[0], 24 << 20)
        target = r.renderForward({k: torch.as_tensor(v, device=r.device) for k, v in perturb(p, 12345).items()}, cam).render.clone()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=r.device)
    try:
        out = {}
        for mode in ("none", "allreduce", "sh_compressed"):
            model = GaussModel(p, r.device)
            tr = GaussianTrainer(model, r, iterationCount=1000, process_group=None if mode == "none" else dist.group.WORLD,
                                 dp_exchange="allreduce" if mode == "none" else mode, exchange_when_single=True)
            for _ in range(3):
                tr.trainStep(cam, target, stepCameras=[cam])
            out[mode] = _np(model.arena).copy()
        with pytest.raises(ValueError):
            tr.trainStep(cam, target)                                          # sh_compressed needs the step's cameras
    finally:
        dist.destroy_process_group()
    a = out["allreduce"] - _np(GaussModel(p, r.device).arena)
    b = out["none"] - _np(GaussModel(p, r.device).arena)
    assert np.mean(np.abs(a - b) > 1e-3 * np.abs(b).max()) < 1e-3              # atomics: not bit-reproducible run to run
    # Adam's first steps are ~lr * sign(g): compare the parameter movement, not the parameters
    a = out["sh_compressed"] - _np(GaussModel(p, r.device).arena)
    b = out["none"] - _np(GaussModel(p, r.device).arena)
    assert np.mean(np.abs(a - b) > 1e-3 * np.abs(b).max()) < 1e-3


@pytest.mark.parametrize("workload", ["small", "c4_300k_800"])
def test_native_rccl_exchange_matches_the_torch_exchange(oracle32, workload):
    """Row e through the C ABI: gs_dp_unique_id + gs_dp_init make a 1-rank RCCL communicator INSIDE the library (no
    torch.distributed anywhere in this test), gs_dp_step runs backward + all-gather / all-reduce + Adam on the library's
    side stream, gs_dp_allreduce_sum carries the densify statistic.  Three steps of each mode must leave the parameters the
    exchange-free single-device steps leave (the same bar as the torch.distributed path's test above)."""
    import ctypes as C
    from gaussiansplattingmlx_amd import _lib
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    if workload == "small":
        W, H, N = 160, 120, 3000
        p, cam = _scene(62, N, W, H, scale=0.06)
        r = _renderer(W, H)
        target = torch.rand(H, W, 3, device=r.device)
    else:
        from gaussiansplattingmlx_amd.scenes import make_config, perturb
        p, cams, (W, H) = make_config("c3_300k_800", n_views=1)
        cam = cams[0]
        r = _renderer(W, H)
        r.reserve(p["xyz"].shape[0], 24 << 20)
        target = r.renderForward({k: torch.as_tensor(v, device=r.device) for k, v in perturb(p, 12345).items()}, cam).render.clone()
    out, accum = {}, {}
    for mode in ("none", "allreduce", "sh_compressed"):
        model = GaussModel(p, r.device)
        boot = None
        if mode != "none":
            uid = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
            assert r.lib.gs_dp_unique_id(uid) == 0
            boot = (uid.raw, 0, 1)
        tr = GaussianTrainer(model, r, iterationCount=1000, dp_exchange="allreduce" if mode == "none" else mode,
                             exchange_when_single=True, exchange_impl="native", dp_bootstrap=boot)
        if mode != "none":
            rank, world = C.c_int(-1), C.c_int(-1)
            r._check(r.lib.gs_dp_info(r.ctx, C.byref(rank), C.byref(world)))
            assert (rank.value, world.value) == (0, 1) and tr._native
        for _ in range(3):
            tr.trainStep(cam, target, stepCameras=[cam])
        if mode == "sh_compressed":
            with pytest.raises(ValueError):
                tr.trainStep(cam, target)                                      # needs the step's cameras
        out[mode] = _np(model.arena).copy()
        accum[mode] = _np(tr.xyzGradAccumulation).copy()
        if mode != "none":
            # a sum over one rank is the identity, and nothing was gated
            buf = torch.arange(1000, dtype=torch.float32, device=r.device)
            r._check(r.lib.gs_dp_allreduce_sum(r.ctx, C.c_void_p(buf.data_ptr()), 1000))
            assert torch.equal(buf, torch.arange(1000, dtype=torch.float32, device=r.device))
            assert tr._collectiveOverflowCheck() is False
            tr.closeExchange()
            r._check(r.lib.gs_dp_info(r.ctx, None, C.byref(world)))
            assert world.value == 0
    base = _np(GaussModel(p, r.device).arena)
    for mode in ("allreduce", "sh_compressed"):
        a, b = out[mode] - base, out["none"] - base
        assert np.abs(b).max() > 0
        assert np.mean(np.abs(a - b) > 1e-3 * np.abs(b).max()) < 1e-3, mode    # atomics: not bit-reproducible run to run
        np.testing.assert_allclose(accum[mode], accum["none"], rtol=2e-3, atol=1e-4 * np.abs(accum["none"]).max())
    # without a communicator the step is refused, not run single-handed
    a = _lib.gs_dp_step_args()
    assert r.lib.gs_dp_step(r.ctx, 0, C.byref(a)) == 1


@pytest.mark.parametrize("mode", ["sh_compressed", "allreduce"])
@pytest.mark.parametrize("impl", ["native", "torch"])
def test_exchange_gates_and_regrows_after_an_overflow(oracle32, impl, mode):
    """A forward that does not fit the pair reserve inside a data-parallel step: no host error (no rank may leave a step
    alone), the step's gate -- round 5: the rank's overflow word riding behind its colour cotangents in the all-gather, or
    behind the gradient arena in the all-reduce, no collective of its own -- skips the update in every optimizer kernel of
    the step, the collective look every rank would take at the same step (gs_dp_check_overflow / _collectiveOverflowCheck:
    the `seen` word an optimizer kernel raises) agrees on the need and regrows the reserve, and training carries on.  Both
    issuers of the collectives (the library's own RCCL calls; torch.distributed on a 1-rank nccl group), both exchanges.
    The replica check of SURVEY 8(e) runs its kernels and its collective on the way (one rank: it passes)."""
    import ctypes as C
    import os
    import socket
    import torch.distributed as dist
    from gaussiansplattingmlx_amd import _lib
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H, N = 160, 120, 3000
    p, cam = _scene(62, N, W, H, scale=0.06)
    r0 = _renderer(W, H)
    r0.renderForward({k: torch.as_tensor(v, device=r0.device) for k, v in p.items()}, cam)
    need = r0.stats()["M"]
    r0.close()
    r = _renderer(W, H)
    target = torch.rand(H, W, 3, device=r.device)
    r.reserve(N, need // 2)                                    # too small on purpose (a reserve only ever grows)
    assert r.stats()["capM"] == need // 2
    model = GaussModel(p, r.device)
    if impl == "native":
        uid = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
        assert r.lib.gs_dp_unique_id(uid) == 0
        kw = dict(exchange_impl="native", dp_bootstrap=(uid.raw, 0, 1))
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=r.device)
        kw = dict(process_group=dist.group.WORLD)
    try:
        tr = GaussianTrainer(model, r, iterationCount=1000, dp_exchange=mode, exchange_when_single=True, densify=False, **kw)
        before = _np(model.arena).copy()
        for _ in range(3):
            tr.trainStep(cam, target, stepCameras=[cam])           # no viewKey: no first-visit check, nothing raises
        torch.cuda.synchronize()
        np.testing.assert_array_equal(_np(model.arena), before)    # every step was gated
        assert not _np(model.m).any() and not _np(model.v).any()
        assert tr._collectiveOverflowCheck() is True and tr.overflowRecoveries == 1
        assert r.stats()["capM"] >= need
        for _ in range(2):
            tr.trainStep(cam, target, stepCameras=[cam])
        torch.cuda.synchronize()
        assert np.abs(_np(model.arena) - before).max() > 0         # ... and now it trains
        assert tr._collectiveOverflowCheck() is False
        tr.checkReplicas()
        tr.closeExchange()
    finally:
        if impl == "torch":
            dist.destroy_process_group()


# ------------------------------------------------------------------------------ next row: densify / prune
def _densify_scene(N, seed=9, K=25):
    rng = np.random.default_rng(seed)
    p = dict(xyz=rng.normal(size=(N, 3)), features_dc=rng.normal(size=(N, 1, 3)),
             features_rest=rng.normal(size=(N, K - 1, 3)), scales=rng.normal(np.log(0.01), 0.5, (N, 3)),
             rotation=rng.normal(size=(N, 4)), opacity=rng.normal(-3, 3, N))
    p = {k: np.ascontiguousarray(v, np.float32) for k, v in p.items()}
    acc = (np.abs(rng.normal(0, 3e-4, N)) * 7).astype(np.float32)
    return p, acc


@pytest.mark.parametrize("N", [1, 1023, 70001])
def test_densify_kernels_match_the_oracle(oracle32, N):
    o = oracle32
    r = _renderer(64, 64)
    p, acc = _densify_scene(N)
    # accum_grad_norm: sqrt and adds are correctly rounded on both sides -> bit-exact
    g = np.random.default_rng(1).normal(0, 1e-3, (N, 3)).astype(np.float32)
    got = r.accumGradNorm(g, acc)
    np.testing.assert_array_equal(_np(got), o.accum_grad_norm(g, acc))
    np.testing.assert_array_equal(_np(r.accumGradNorm(g)), o.accum_grad_norm(g))
    # classify: integer output; exp() may differ by an ulp between libm and the device, so entries whose decision
    # value sits within 4 ulp of a threshold are excluded (and must be rare)
    for allow in (True, False):
        wa, wc = o.classify_gaussians(acc, 7.0, p["scales"], p["opacity"], allowDensify=allow)
        ga, gc = r.classifyGaussians(acc, 7.0, p["scales"], p["opacity"], allowDensify=allow)
        ms = np.exp(p["scales"].astype(np.float64)).max(1)
        op = 1 / (1 + np.exp(-p["opacity"].astype(np.float64)))
        near = (np.abs(ms - 0.01) < 4e-7 * 0.01) | (np.abs(op - 0.005) < 4e-7 * 0.005)
        assert near.mean() < 1e-3
        np.testing.assert_array_equal(_np(ga)[~near], wa[~near])
        np.testing.assert_array_equal(_np(gc)[~near], wc[~near])
    # scan + histogram + map on the ORACLE's classification: bit-exact
    wa, wc = o.classify_gaussians(acc, 7.0, p["scales"], p["opacity"])
    woff, wst = o.densify_offsets(wa, wc)
    ta, tc = torch.as_tensor(wa, device=r.device), torch.as_tensor(wc, device=r.device)
    goff, gst = r.densifyOffsets(ta, tc)
    assert gst == wst
    np.testing.assert_array_equal(_np(goff), woff)
    if wst["total"] == 0:
        return
    wg, wm = o.build_densify_output_map(wa, woff, wst["total"])
    gg, gm = r.buildDensifyOutputMap(ta, goff, wst["total"])
    np.testing.assert_array_equal(_np(gg), wg)
    np.testing.assert_array_equal(_np(gm), wm)
    # gather + per-slot modification
    nz = np.random.default_rng(2).normal(size=(wst["total"], 3)).astype(np.float32)
    want = o.densify_gather(p, wg, wm, nz)
    got = r.densifyGather({k: torch.as_tensor(v, device=r.device) for k, v in p.items()}, gg, gm, nz)
    for k in ("features_dc", "features_rest", "rotation", "opacity", "scales"):
        np.testing.assert_array_equal(_np(got[k]), want[k].reshape(_np(got[k]).shape), err_msg=k)
    np.testing.assert_allclose(_np(got["xyz"]), want["xyz"], rtol=1e-6, atol=1e-7)       # exp() inside the noise scale
    got0 = r.densifyGather({k: torch.as_tensor(v, device=r.device) for k, v in p.items()}, gg, gm, None)
    for k in p:
        np.testing.assert_array_equal(_np(got0[k]), p[k][wg].reshape(_np(got0[k]).shape), err_msg=k)


@pytest.mark.parametrize("N", [1, 1023, 70001])
def test_planned_densify_kernels_match_the_unplanned_ones(oracle32, N):
    """The event with the count left on the device (gs_densify_plan / ..._output_map_planned / gs_densify_gather_planned;
    round 5) against the kernels that take the count from the host, on the same classification: the plan's words are the
    scan's totals, the map is the same map in `capacity` zero-filled slots, the gather writes the same rows [0, new count)
    -- its noise, drawn inside the kernel from (seed, row), is gs_densify_noise's tensor -- and leaves the rows behind
    alone; an event that changes nothing (no split, clone or prune: the reference's early-out, GaussianTrainer.swift:819-847)
    plans the identity.  The generator's output looks like a standard normal."""
    o = oracle32
    r = _renderer(64, 64)
    p, acc = _densify_scene(N)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
    wa, wc = o.classify_gaussians(acc, 7.0, p["scales"], p["opacity"])
    woff, wst = o.densify_offsets(wa, wc)
    ta, tc = torch.as_tensor(wa, device=r.device), torch.as_tensor(wc, device=r.device)
    off = r.densifyPlan(ta, tc)
    plan = r.densifyPlanRead()
    applies = wst["total"] > 0 and (wst["split"] or wst["clone"] or wst["prune"])
    assert {k: plan[k] for k in wst} == wst and plan["N"] == N and plan["applies"] == int(bool(applies))
    assert plan["N_new"] == (wst["total"] if applies else N)
    np.testing.assert_array_equal(_np(off), woff)
    cap = 2 * N + 7
    gg, gm = r.buildDensifyOutputMapPlanned(ta, off, cap)
    if applies:
        wg, wm = o.build_densify_output_map(wa, woff, wst["total"])
        np.testing.assert_array_equal(_np(gg)[:wst["total"]], wg)
        np.testing.assert_array_equal(_np(gm)[:wst["total"]], wm)
        assert not _np(gg)[wst["total"]:].any() and not _np(gm)[wst["total"]:].any()
        seed = 20260313 + 600
        nz = r.densifyNoise(seed, wst["total"])
        want = r.densifyGather(tp, torch.as_tensor(wg, device=r.device), torch.as_tensor(wm, device=r.device), nz)
        out = {k: torch.full((cap,) + tuple(v.shape[1:]), -7.0, device=r.device) for k, v in tp.items()}
        r.densifyGatherPlanned(tp, gg, gm, seed, out, cap)
        for k in out:
            assert torch.equal(out[k][:wst["total"]], want[k]), k
            assert bool((out[k][wst["total"]:] == -7.0).all()), k          # rows beyond the new count are left alone
        if wst["total"] > 3000:
            z = _np(nz).astype(np.float64)
            assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02 and abs((z ** 3).mean()) < 0.05 and np.abs(z).max() < 6.5
            assert abs(np.corrcoef(z[:, 0], z[:, 1])[0, 1]) < 0.03 and abs(np.corrcoef(z[:-1, 0], z[1:, 0])[0, 1]) < 0.03
    # an event that changes nothing: every action "keep" -> the identity, whatever the count
    keep = torch.zeros(N, dtype=torch.int32, device=r.device)
    ones = torch.ones(N, dtype=torch.int32, device=r.device)
    off = r.densifyPlan(keep, ones)
    plan = r.densifyPlanRead()
    assert plan["applies"] == 0 and plan["N_new"] == N and plan["total"] == N and plan["keep"] == N
    gg, gm = r.buildDensifyOutputMapPlanned(keep, off, cap)
    out = {k: torch.full((cap,) + tuple(v.shape[1:]), -7.0, device=r.device) for k, v in tp.items()}
    r.densifyGatherPlanned(tp, gg, gm, 1, out, cap)
    for k in out:
        assert torch.equal(out[k][:N], tp[k]) and bool((out[k][N:] == -7.0).all()), k
    r.close()


def test_packed_planned_gather_of_a_plan_that_does_not_fit_stays_inside_its_buffer():
    """gs_densify_gather_planned_packed lays the six tensors out on the device from the plan's new count.  When that count
    is beyond the staging buffer's capacity -- the host learns it from the plan words only afterwards, regrows and gathers
    again -- the first gather must still write inside the buffer it was given: the starts are laid out for min(N_new, capacity)
    rows.  (Round 6, found by tools/soak_dp.py: laid out for N_new they pushed the last tensors past the buffer's end --
    silent corruption at a run's first regrow, a GPU write fault at its second.  The event tests compare models, which came out
    right after the second gather; this one looks at the bytes behind the buffer.)"""
    from gaussiansplattingmlx_amd.trainer import ARENA_ORDER
    N = 20011
    p, acc = _densify_scene(N)
    r = _renderer(64, 64)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
    actions, counts = r.classifyGaussians(acc, 7.0, tp["scales"], tp["opacity"].reshape(-1), allowDensify=True)
    offsets = r.densifyPlan(actions, counts)
    plan = r.densifyPlanRead()
    cap = N                                                    # the model's own size: the event grows it
    assert plan["applies"] and plan["N_new"] > cap + 1000
    per = dict(xyz=3, scales=3, rotation=4, opacity=1, features_dc=3, features_rest=72)
    packed = sum((cap * per[k] + 3) & ~3 for k in ARENA_ORDER)
    guard = 4 * (plan["N_new"] - cap) * 86 + 4096              # more than the overrun of a layout for N_new would reach
    base = torch.full((packed + guard,), -7.0, device=r.device)
    gather, mode = r.buildDensifyOutputMapPlanned(actions, offsets, cap)
    r.densifyGatherPlannedPacked(tp, gather, mode, 1234, base, cap, ARENA_ORDER)
    torch.cuda.synchronize()
    assert bool((base[packed:] == -7.0).all()), int((base[packed:] != -7.0).sum())
    assert bool((base[:packed] != -7.0).any())
    r.close()


@pytest.mark.parametrize("form", ["single", "dp1_native", "dp1_torch", "local_two_views"])
def test_planned_densify_event_leaves_the_model_the_unplanned_event_leaves(oracle32, form):
    """(form, round 6: the same through the DATA-PARALLEL step's event -- a 1-rank RCCL communicator inside the library, a 1-rank
    nccl process group, and two views per step without a group: the planned gather then writes the PACKED layout, tensor starts
    computed on the device behind the plan (gs_densify_gather_planned_packed), the ranks compare plans in a fixed-size
    collective on a side stream and the arena checksum is queued and judged later.)

    trainer.split_and_prune with the count left on the device (plannedDensify, the default of a single-device trainer:
    classify, plan, map and gather into a capacity-strided layout, the optimizer reset -- all queued before the host waits,
    and it waits for the plan alone) against the reference's sequence (read the count, then size and queue everything), both
    with the library's noise and from the same state (no training steps in between: float atomics are not reproducible run
    to run): the same N, the same six tensors bit for bit, zeroed moments, the same action counts -- through an event that
    splits, clones and prunes, the same with a new count beyond the capacity (the gather is repeated into a larger buffer),
    a prune-only event and one that changes nothing (the identity); then a training step on the strided layout."""
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    N, W, H = 20011, 64, 64
    p, acc = _densify_scene(N)
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    cam = Camera(W, H, 60.0, 60.0, look_at_c2w([4.0, -5.0, 3.0]))
    out = {}
    kw, step_kw, pg_up = {}, {}, False
    if form == "dp1_torch":
        import socket
        import torch.distributed as dist
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        pg_up = True
        kw = dict(process_group=dist.group.WORLD, exchange_when_single=True)
    try:
        _planned_event_cases(form, kw, p, acc, N, W, H, cam, out)
    finally:
        if pg_up:
            dist.destroy_process_group()
    ref = out[(False, 3 * N)]
    assert ref[0][1]["split"] > 0 and ref[0][1]["clone"] > 0 and ref[0][1]["prune"] > 0 and ref[0][0] > N
    assert ref[1][1]["split"] == 0 and ref[1][1]["clone"] == 0 and ref[1][1]["prune"] >= 10 and ref[1][0] < ref[0][0]
    assert ref[2][1]["prune"] == 0 and ref[2][0] == ref[1][0]
    for key in ((True, 3 * N), (True, N)):
        for e, (a, b) in enumerate(zip(ref, out[key])):
            assert a[0] == b[0] and a[1] == b[1], (key, e, a[0], b[0], a[1], b[1])
            for k in a[2]:
                assert np.array_equal(a[2][k], b[2][k]), (key, e, k)
            assert b[5] == b[0] and b[6] == 0.0 and b[7] == 0, (key, e)           # accumulators reset, at the new size
        # the optimizer state is re-created at every cadence, changed or not (:1098-1110): the planned event does it inside
        assert all(b[3] == 0.0 and b[4] == 0.0 for b in out[key]), key
    assert ref[0][3] == 0.0 and ref[1][3] == 0.0          # (a committed event of the reference sequence resets it as well)


def _planned_event_cases(form, kw, p, acc, N, W, H, cam, out):
    import ctypes as C
    from gaussiansplattingmlx_amd import _lib
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    for planned, cap in ((False, 3 * N), (True, 3 * N), (True, N)):          # (capacity N: the first event does not fit)
        r = _renderer(W, H)
        r.reserve(3 * N, 4 << 20)
        model = GaussModel(p, r.device, capacity=cap)
        if form == "dp1_native":
            uid = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
            assert r.lib.gs_dp_unique_id(uid) == 0
            kw = dict(exchange_impl="native", dp_bootstrap=(uid.raw, 0, 1), exchange_when_single=True)
        elif form == "local_two_views":
            kw = dict(views_per_rank=2)
        tr = GaussianTrainer(model, r, iterationCount=1000, **kw)
        assert tr._dp == (form != "single") and tr._exchange == form.startswith("dp1")
        tr.plannedDensify, tr.noiseSource = planned, "library"
        snaps = []

        def event(it):
            st = tr.split_and_prune(it)
            torch.cuda.synchronize()
            snaps.append((model.N, dict(st), {k: _np(v).copy() for k, v in model.getParams().items()},
                          float(model.m.abs().max()), float(model.v.abs().max()), int(tr.xyzGradAccumulation.shape[0]),
                          float(tr.xyzGradAccumulation.abs().max()), tr.denomGradAccumulation))

        model.m.fill_(0.5); model.v.fill_(0.25)                       # an optimizer state for the event to reset
        tr.xyzGradAccumulation = torch.as_tensor(acc, device=r.device).clone()
        tr.denomGradAccumulation = 7
        event(600)                                                    # splits, clones, prunes
        tr.maxGaussians = 1                                           # from here on the budget allows pruning only (:785)
        model.getParams()["opacity"][:10] = -9.0
        model.m.fill_(0.5)
        tr.denomGradAccumulation = 3
        event(700)
        tr.minOpacity = 0.0                                           # ... and now there is nothing to prune either
        model.m.fill_(0.5)
        event(800)
        out[(planned, cap)] = snaps
        # a single-device planned event lays the model out at capacity strides; the data-parallel form's is PACKED (it
        # all-reduces the arena's leading geometry slice), its pads zero
        assert model.stride == (model.capacity if planned and form == "single" else model.N)
        if form != "single":
            from gaussiansplattingmlx_amd.trainer import ARENA_ORDER
            arena = _np(model.arena)
            for k, lo, hi in zip(ARENA_ORDER, model.seg_start, model.seg_end):
                assert not arena[int(lo) + model.N * model._per[k]:int(hi)].any(), k
        target = torch.rand(H, W, 3, device=r.device)
        for _ in range(2):          # (the scene is a test pattern for the classifier, not a picture: the steps only have to run)
            if form == "local_two_views":
                loss = tr.trainStep([cam, cam], [target, target], stepCameras=[cam, cam])
            else:
                loss = tr.trainStep(cam, target, stepCameras=[cam])
        assert np.isfinite(float(loss[0])) and model.getParams()["xyz"].shape[0] == model.N
        tr.closeExchange()          # (takes the verdict of the event's deferred replica check)
        r.close()


def test_trainer_split_and_prune_follows_the_reference_sequence(oracle32):
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H, N = 160, 120, 4000
    p, cam = _scene(63, N, W, H, scale=0.06)
    p["features_rest"] *= 0.05
    p["opacity"][:50] = -8.0                                       # sigma < 0.005 -> pruned
    tgt = oracle32.render_forward(perturb(p, 5, 0.1), cam.as_dict(), W, H, 16, 16, 4)["color"].reshape(H, W, 3)
    r = _renderer(W, H)
    r.reserve(N, 1 << 20)
    model = GaussModel(p, r.device)
    tr = GaussianTrainer(model, r, iterationCount=1000)
    tr.densifyFromIter, tr.split_and_prune_per_iteration = 4, 4
    tr.gradientThreshold = 2e-6
    target = torch.as_tensor(tgt, device=r.device)
    # iterations 0..3: outside the window at it = 0 (returns early, :767), accumulating
    for _ in range(4):
        tr.trainStep(cam, target)
    assert model.N == N and tr.denomGradAccumulation == 4 and float(model.m.abs().max()) > 0
    before = {k: _np(v).copy() for k, v in model.getParams().items()}
    # iteration 4 densifies after its Adam step: replay that step's inputs on the oracle
    acc_before = _np(tr.xyzGradAccumulation).copy()
    tr.trainStep(cam, target)
    st = tr.lastDensifyStats
    assert st["prune"] >= 50 and st["split"] + st["clone"] > 0
    assert model.N == st["total"] == st["keep"] + 2 * (st["split"] + st["clone"])
    assert tr.denomGradAccumulation == 0 and not _np(tr.xyzGradAccumulation).any()
    assert not _np(model.m).any() and not _np(model.v).any()       # optimizer state re-created (:1104-1109)
    assert r.stats()["capN"] >= model.N
    assert acc_before.max() > 0 and before["xyz"].shape[0] == N
    # training continues on the new model
    l0 = float(tr.trainStep(cam, target)[0])
    assert np.isfinite(l0) and model.getGrads()["xyz"].shape[0] == model.N
    # prune-only pass: densification disabled by the budget (:785)
    tr.maxGaussians = 1
    model.getParams()["opacity"][:10] = -9.0
    n0 = model.N
    while tr.iteration % 4 != 1:
        tr.trainStep(cam, target)
    st = tr.lastDensifyStats
    assert st["split"] == 0 and st["clone"] == 0 and st["prune"] >= 10 and model.N == n0 - st["prune"]


@pytest.mark.parametrize("K,degree,N", [(25, 4, 3001), (25, 4, 3000), (16, 3, 1500), (9, 2, 1501), (4, 1, 700), (1, 0, 333)])
def test_fused_backward_adam_matches_backward_then_adam(oracle32, K, degree, N):
    """gs_render_backward_adam == gs_render_backward + gs_adam_step (same arithmetic, no gradient arena), for every SH
    size: K = 25 and 9 take the float4 rows kept in registers from the staging load to the update, K = 16 and 4 (rows of
    45 and 9 floats) the scalar staging, K = 1 has no rows; odd and even counts (the arena pads every tensor to 16 B)."""
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H = 160, 120
    p, cam = _scene(64, N, W, H, K=K, scale=0.06)
    r = _renderer(W, H, degree=degree)
    target = torch.rand(H, W, 3, device=r.device, generator=torch.Generator(device=r.device).manual_seed(3))
    out = {}
    for fuse in (False, True):
        model = GaussModel(p, r.device)
        tr = GaussianTrainer(model, r, iterationCount=1000, fuse_adam=fuse)
        for _ in range(3):
            tr.trainStep(cam, target)
        out[fuse] = (_np(model.arena).copy(), _np(model.m).copy(), _np(model.v).copy(), _np(tr.xyzGradAccumulation).copy())
    start = _np(GaussModel(p, r.device).arena)
    a, b = out[True][0] - start, out[False][0] - start
    assert np.abs(b).max() > 0
    assert np.mean(np.abs(a - b) > 1e-3 * np.abs(b).max()) < 1e-3              # atomics: not bit-reproducible run to run
    for k in (1, 2, 3):
        ref = out[False][k]
        assert np.mean(np.abs(out[True][k] - ref) > 1e-3 * np.abs(ref).max()) < 1e-3, k
    # the arena check: tensors outside the arena are refused
    from gaussiansplattingmlx_amd._lib import GsplatError
    model = GaussModel(p, r.device)
    r.renderForward({k: v.clone() for k, v in model.getParams().items()}, cam)
    with pytest.raises(GsplatError):
        r.renderBackwardAdam(torch.zeros(H, W, 3, device=r.device), model.arena, model.m, model.v, [1e-3] * 6)


# ------------------------------------------------------- the bench workload itself against the oracle (BASELINE configs[2])
@pytest.mark.parametrize("sh_rest_scale", [0.02, 1.0])
def test_bench_workload_parity_300k_800(oracle32, oracle64, sh_rest_scale):
    """One view of the benchmark scene (synthetic Lego 800x800, 300 k Gaussians, K = 25) against the float32 oracle:
    pair count and radii exact, per-pixel nContrib exact but for threshold ties, loss and every gradient tensor within 1e-3.

    Image bar: the north star's 1e-4 L-inf ABSOLUTE, on the raw SURVEY 8(d) scene (SH-rest ~ N(0, 0.05^2) with the
    reference's un-normalised view direction: colours up to ~27, so 1e-4 absolute is 4e-6 relative) as well as with the
    SH-rest scaled to physical colours (<= ~1.4).  Measured: 4.3e-5 and 2.5e-6.  What it took (tools/full_size_parity.py,
    DESIGN.md section 2): the forward's exp(-q/2) carries the rounding error of its exponent product along; with the
    plain v_exp_f32(q * const) the raw scene sat at 3.4e-4, and the reference's two-rounding colour accumulation instead
    of fmaf changed nothing.  For scale: the float32 and float64 oracles differ by 3.6e-3 on the raw scene."""
    from gaussiansplattingmlx_amd.scenes import make_config, perturb
    params, cams, (W, H) = make_config("c3_300k_800", n_views=1)
    params = dict(params)
    params["features_rest"] = (params["features_rest"] * np.float32(sh_rest_scale)).astype(np.float32)
    cam = cams[0]
    o = oracle32
    c = cam.as_dict()
    fw = o.render_forward(params, c, W, H, 16, 16, 4)
    r = _renderer(W, H)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
    res = r.renderForward(tp, cam, want_radii=True)
    st = r.stats()
    _pairs_match(r, fw["bin"].M)
    assert st["overflow"] == 0
    err = np.abs(_np(res.render).reshape(-1, 3) - fw["color"])
    cmax = float(fw["color"].max())
    assert err.max() <= RGB_TOL, (cmax, err.max())
    assert (cmax < 2.0) == (sh_rest_scale < 1.0)
    np.testing.assert_array_equal(_np(res.radii), fw["proj"]["radii"])
    # nContrib is an integer cut at T < 1e-4: a pixel whose T lands within an ulp of the threshold can stop a splat or
    # two earlier or later when exp() differs in the last bit (device v_exp_f32 vs libm) -- a handful of 640 000
    _ncontrib_match(r, fw, W, H, slack_pixels=0)       # (at most 2e-5 of the pixels, a handful of 640 000)
    tgt = o.render_forward(perturb(params, 12345), c, W, H, 16, 16, 4)["color"].reshape(H, W, 3)
    loss, cc, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)
    lo, gc, _ = r.lossForwardBackward(res.render, tgt, 0.2)
    assert abs(_np(lo)[0] - loss) < 1e-5 * max(1.0, abs(loss))
    z = np.zeros(W * H, np.float32)
    want = o.render_backward(params, c, W, H, 16, 16, 4, fw, cc.reshape(-1, 3), z, z)
    got = r.renderBackward(gc)
    for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
        assert _rel(_np(got[k]), want[k].reshape(_np(got[k]).shape)) <= GRAD_RTOL, k
    # ... and element by element (round 5): no tensor with a larger share of its elements beyond 1e-3 (floored at 1e-4 of
    # the tensor's largest) than 1.5 x what separates the float32 from the float64 ORACLE on this view, + 5e-4
    _elementwise_gradient_bar(f"c3_300k_800_sh{sh_rest_scale}", got, want, oracle64, params, c, W, H, tgt)


@pytest.mark.parametrize("scene", ["c2_100k_800", "wide_splats_203x157_white"])
def test_trimmed_rects_change_the_lists_and_nothing_else(oracle32, scene):
    """GS_TUNE_TRIM_RECTS (include/gsplat.h; default 1): at 16 x 16 tiles the fused forward bins a Gaussian on the tiles of the
    reference's 3-sigma square that the axis-aligned box of its ellipse q <= 40.3 reaches.  What is left out is what the blend's
    staging drops for every quadrant of the tile anyway (weight < 2^-29), so against the same forward on the reference's lists
    (knob 0: M the oracle's): fewer pairs; the radii and -- one-wave forward: a running product, no sums -- alpha the same bits; colour and depth
    within the rounding of their per-chunk sums (a chunk is 64 LIST positions, so the kept entries group differently: one
    rounding of the size of the total per chunk); nContrib pointing at the same Gaussians and every list the
    reference's with entries left out (_ncontrib_match); every gradient inside the bar two runs of the SAME forward hold
    (float atomics: test_full_size_properties)."""
    from gaussiansplattingmlx_amd.scenes import make_config
    if scene == "c2_100k_800":
        params, cams, (W, H) = make_config("c2_100k_800", n_views=1)
        cam, white = cams[0], False
    else:
        W, H, white = 203, 157, True
        params, cam = _scene(77, 3000, W, H, spread=1.1, scale=0.12)
        params["features_rest"] *= 0.05
    fw = oracle32.render_forward(params, cam.as_dict(), W, H, 16, 16, 4, white)
    r = _renderer(W, H, (16, 16), white)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
    cot = torch.as_tensor(np.random.default_rng(4).normal(size=(W * H, 3)).astype(np.float32), device=r.device)
    out = {}
    for trim in (0, 1, 2, 0):       # (the last pass: back on the reference's lists, the same bits as the first)
        r.setTuning(trim_rects=trim)
        assert _lists_trimmed(r) == bool(trim)
        res = r.renderForward(tp, cam, want_radii=True)
        M = _pairs_match(r, fw["bin"].M)
        _ncontrib_match(r, fw, W, H)
        g = {k: v.clone() for k, v in r.renderBackward(cot).items()}
        got = (M, res.render.clone(), res.depth.clone(), res.alpha.clone(), res.radii.clone(), g)
        if trim in out:
            assert out[trim][0] == M and all(torch.equal(a, b) for a, b in zip(out[trim][1:5], got[1:5]))
        out[trim] = got
    # 1 = the square cut by the ellipse's box, 2 (default) = the box cut further into four row groups: fewer pairs again, and
    # against the reference's lists the same bars
    assert out[2][0] < 0.97 * out[1][0] and out[1][0] < 0.97 * out[0][0]
    for k in (1, 2):
        assert torch.equal(out[k][4], out[0][4])
        assert (out[k][1] - out[0][1]).abs().max().item() <= (1e-6 if scene == "c2_100k_800" else 2e-6) * max(1.0, out[0][1].abs().max().item())
    (M0, img0, dep0, alp0, rad0, g0), (M1, img1, dep1, alp1, rad1, g1) = out[0], out[2]
    assert M0 == fw["bin"].M and M1 < 0.97 * M0
    assert torch.equal(rad0, rad1)
    if scene == "c2_100k_800":      # the one-wave forward: T is one running product down the list
        assert torch.equal(alp0, alp1)
    else:                           # an image this small takes the four-wave forward, which composes T across chunks as well
        assert (alp0 - alp1).abs().max().item() <= 2e-6            # (measured 6e-7: ten ulps of a T near 1)
    tol = 1e-6 if scene == "c2_100k_800" else 2e-6            # (measured 4.4e-7: four ulps of a depth of 4.3)
    assert (img0 - img1).abs().max().item() <= tol * max(1.0, img0.abs().max().item())
    assert (dep0 - dep1).abs().max().item() <= tol * max(1.0, dep0.abs().max().item())
    assert np.abs(_np(img1).reshape(-1, 3) - fw["color"]).max() <= RGB_TOL * max(1.0, float(np.abs(fw["color"]).max()))
    for k in g0:
        a, b = g1[k].double(), g0[k].double()
        assert float((a - b).norm()) <= 2e-3 * float(b.norm()) + 1e-12, k
        assert float((a - b).abs().max()) <= 2e-2 * float(b.abs().max()) + 1e-12, k


@pytest.mark.parametrize("seed,scale,aniso", [(1, 0.05, 30.0), (2, 0.15, 100.0), (3, 0.4, 10.0)])
def test_row_groups_drop_no_tile_a_needle_reaches(oracle32, seed, scale, aniso):
    """The trimmed rects' row groups (gs_math.h rect_row_groups4) on what they are for: elongated, tilted splats -- one scale
    `aniso` times the other two, random rotations, 2-D ellipses tens of tiles long at every angle.  EVERY tile of the image is
    checked (_assert_lists_ordered_and_complete): its list holds no Gaussian whose rect misses the tile, and every Gaussian of
    the reference's list that it leaves out has q > 40 on all 256 pixel centres of the tile in float64.  The image is the one the
    same forward gives on the reference's lists (knob 0) to the rounding of the chunk sums, and the groups do cut (on the 30 : 1
    needles 1.11 M pairs -> 0.97 M in the box -> 0.35 M in the groups).
    (Against the ORACLE these scenes sit outside the 1e-4 image bar at every setting of the knob alike, 2.5e-4 at 30 : 1 and 1.2e-3
    at 100 : 1: q = a dx^2 + 2 b dx dy + c dy^2 of such a conic is a difference of terms 1e3 .. 1e4 times its size, and two float32
    evaluations of it -- the oracle's libm build, the device's -- differ in the third digit of alpha.  Found by this test; the
    bar here is 2e-3.)"""
    W, H, N = 400, 304, 2500
    p, cam = _scene(400 + seed, N, W, H, spread=1.0, scale=scale)
    rng = np.random.default_rng(seed)
    p["scales"][:, 0] += np.float32(np.log(aniso))                 # needles
    p["scales"][:, 1:] -= np.float32(0.5 * np.log(aniso) * rng.uniform(0.0, 1.0, (N, 2)))
    p["features_rest"] *= 0.05
    fw = oracle32.render_forward(p, cam.as_dict(), W, H, 16, 16, 4)
    r = _renderer(W, H)
    # (the one-wave forward: an image this small would take the four-wave kernel, which composes T across list chunks -- the
    # chunks move with the lists, and a pixel whose T sits at the 1e-4 threshold then blends one splat more or less: one pixel
    # of this scene, 1.5e-5)
    r.setTuning(fwd_four_waves=0)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
    Ms, img0 = {}, None
    for trim in (0, 1, 2):
        r.setTuning(trim_rects=trim)
        res = r.renderForward(tp, cam, want_radii=True)
        Ms[trim] = _pairs_match(r, fw["bin"].M)
        assert np.abs(_np(res.render).reshape(-1, 3) - fw["color"]).max() <= 2e-3 * max(1.0, float(np.abs(fw["color"]).max()))
        if trim == 0:
            img0 = res.render.clone()
            continue
        assert (res.render - img0).abs().max().item() <= 2e-6 * max(1.0, img0.abs().max().item())
        M, idx_n, rng_n, cnt_n = _fused_lists(r, W, H)
        assert _assert_lists_ordered_and_complete(oracle32, r, p, cam, _np(res.radii), idx_n, rng_n, cnt_n, W, H, n_sample=10 ** 9) > 0
    assert Ms[0] == fw["bin"].M and Ms[2] < 0.9 * Ms[1] and Ms[1] < Ms[0], (Ms, fw["bin"].M)


def test_a_needle_whose_conic_is_not_positive_definite_leaves_no_nan(oracle32):
    """tools/soak.py, iteration 10971: a Gaussian of scales (8.2, 0.007, 0.006) -- a needle tens of thousands of pixels long on
    screen, cov2d = (4.2e7, -3.8e7; -3.8e7, 3.5e7) -- whose determinant cancels in float32: its conic comes out NEGATIVE
    (-0.26, -0.29; -0.29, -0.32; the oracle's float32 projection gives the same), q < 0 without bound, exp(-q/2) overflows, and
    the blend backward's branch-free gates (alpha x 0 past a pixel's nContrib) made 0 x inf = NaN in that splat's accumulator
    row; its geometry parameters and moments were NaN from then on.  The row is dropped in the projection backward
    (projection.hip, drop_nonfinite_row): every gradient of the step is finite, the needle's are zero, and the others'
    are what they are without the needle in the scene wherever it changes no pixel."""
    from gaussiansplattingmlx_amd.scenes import make_config
    _, cams, (W, H) = make_config("c3_300k_800", n_views=8)
    cam = cams[1]
    p, _ = _scene(5, 400, W, H, spread=0.8, scale=0.03)
    needle = dict(xyz=[-1.4522207975387573, 0.6518905162811279, -1.2091460227966309],
                  scales=[2.0996572971343994, -4.9795122146606445, -5.073225975036621],
                  rotation=[-2.58821177482605, 0.7762980461120605, -0.24294103682041168, -0.2615210711956024],
                  opacity=[9.125700950622559], features_dc=[[-3.0830109119415283, -0.8318700194358826, -1.586165189743042]])
    for k, v in needle.items():
        p[k][0] = np.asarray(v, np.float32).reshape(p[k][0].shape)
    p["features_rest"][0] = 0.0
    pr = oracle32.render_forward(p, cam.as_dict(), W, H, 16, 16, 4)["proj"]
    con = np.asarray(pr["conic"]).reshape(-1, 4)[0]
    assert con[0] < 0 and con[3] < 0 and float(pr["radii"][0]) > 10000          # the degenerate projection itself
    r = _renderer(W, H)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
    cot = torch.as_tensor(np.random.default_rng(9).normal(size=(W * H, 3)).astype(np.float32), device=r.device)
    res = r.renderForward(tp, cam)
    assert bool(torch.isfinite(res.render).all())
    g = r.renderBackward(cot)
    for k in GRAD_KEYS:
        assert bool(torch.isfinite(g[k]).all()), k
    for k in ("xyz", "scales", "rotation", "opacity"):
        assert not bool(g[k][0].any()), k
    # ... and through the fused backward + Adam: parameters and moments stay finite
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    model = GaussModel(p, r.device)
    tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
    tr.iteration = 1
    for _ in range(3):
        tr.trainStep(cam, res.render.detach().clone() * 0.5, viewKey=0)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(model.arena).all()) and bool(torch.isfinite(model.m).all()) and bool(torch.isfinite(model.v).all())


def test_garden_2m_properties(oracle32):
    """BASELINE configs[4] (2 M Gaussians, 1237x822, partial edge tiles, ~95 M pairs): no oracle render at this size; the
    size-independent properties instead -- ranges partition [0, M), lists sorted by (depth bits, index) inside every
    sampled tile and made of exactly the Gaussians whose rect covers the tile (_assert_lists_ordered_and_complete; the
    same two-word pair path is held to the oracle bit for bit on synthetic rects in test_gpu_binning_large.py), forward
    deterministic, automatic workspace growth, backward finite and linear in the cotangent."""
    from gaussiansplattingmlx_amd.scenes import make_config
    params, cams, (W, H) = make_config("c5_garden_2m", n_views=1)
    r = _renderer(W, H)                                           # no reserve: the library sizes the workspace itself
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
    res = r.renderForward(tp, cams[0], want_radii=True)
    radii = _np(res.radii)
    img1 = res.render.clone()
    st = r.stats()
    M, T = st["M"], ((W + 15) // 16) * ((H + 15) // 16)
    assert M > 40_000_000 and st["overflow"] == 0 and st["capM"] >= M      # (95 M on the reference's squares, 49 M on trimmed rects)
    idx = torch.empty(M, dtype=torch.int32, device=r.device)
    rng_ = torch.empty(T, 2, dtype=torch.int32, device=r.device)
    cnt = torch.empty(T, dtype=torch.int32, device=r.device)
    import ctypes as C
    r._check(r.lib.gs_tile_bin_export(r.ctx, C.c_void_p(idx.data_ptr()), C.c_void_p(rng_.data_ptr()),
                                      C.c_void_p(cnt.data_ptr())))
    cnt_n, rng_n = _np(cnt).astype(np.int64), _np(rng_).astype(np.int64)
    assert cnt_n.sum() == M
    nz = cnt_n > 0
    starts, ends = np.sort(rng_n[nz, 0]), np.sort(rng_n[nz, 1])
    assert starts[0] == 0 and ends[-1] == M and (starts[1:] == ends[:-1]).all()
    idx_n = _np(idx).astype(np.int64)
    del idx
    assert _assert_lists_ordered_and_complete(oracle32, r, params, cams[0], radii, idx_n, rng_n, cnt_n, W, H) >= 300
    del idx_n
    assert torch.equal(img1, r.renderForward(tp, cams[0]).render)
    assert torch.isfinite(img1).all()
    g = torch.Generator(device="cpu").manual_seed(2)
    c1 = torch.randn(W * H, 3, generator=g).to(r.device)
    g1 = {k: v.clone() for k, v in r.renderBackward(c1).items()}
    g2 = r.renderBackward(-2.0 * c1)
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        ref = -2.0 * g1[k]
        # two runs differ by the order of ~1e8 float atomics: compare in the 2-norm, and loosely in the max-norm
        assert float((g2[k] - ref).double().norm()) <= 2e-3 * float(ref.double().norm()) + 1e-12, k
        assert (g2[k] - ref).abs().max() <= 2e-2 * ref.abs().max() + 1e-12, k
    # depth cuts at this size (sliced expansion, two-level segment prefix): a hinted visit, its backward, a cut visit
    nc1 = r.lastContrib().clone()
    r.cutMinDropped = 0
    assert torch.equal(r.renderForward(tp, cams[0], viewKey="g").render, img1) and not r.forwardMissed()
    r.renderBackward(c1)
    cut = r.renderChecked(tp, cams[0], viewKey="g")
    assert r.stats()["M"] < M // 4
    assert torch.equal(cut.render, img1) and torch.equal(r.lastContrib(), nc1)


@pytest.mark.parametrize("trim", [2, 1, 0])
@pytest.mark.parametrize("four_waves", [-1, 0])
@pytest.mark.parametrize("seed", range(12))
def test_randomized_small_scenes(oracle32, seed, four_waves, trim):
    """Random image sizes (partial edge tiles, images smaller than a tile), Gaussian counts from 1 up, random scale and
    spread, white or black background: fused forward / loss / backward against the oracle.  four_waves: -1 = the default
    (images this small take the four-waves-per-quadrant forward, blend_fwd_v2w_kernel), 0 = the one-wave kernel.
    trim: GS_TUNE_TRIM_RECTS -- 1 = the default (lists without the entries no pixel of the tile can see: _pairs_match,
    _ncontrib_match), 0 = the reference's lists, M and nContrib position for position."""
    from gaussiansplattingmlx_amd.scenes import perturb
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.integers(9, 97)), int(rng.integers(9, 97))
    N = int(rng.choice([1, 2, 7, 63, 64, 65, 300, 1500]))
    white = bool(rng.integers(0, 2))
    p, cam = _scene(int(rng.integers(0, 10 ** 6)), N, W, H, spread=float(rng.uniform(0.2, 1.2)),
                    scale=float(rng.uniform(0.02, 0.3)))
    p["features_rest"] *= 0.05
    c = cam.as_dict()
    o = oracle32
    fw = o.render_forward(p, c, W, H, 16, 16, 4, white)
    r = _renderer(W, H, (16, 16), white)
    r.setTuning(fwd_four_waves=four_waves, trim_rects=trim)
    res = r.renderForward({k: torch.as_tensor(v) for k, v in p.items()}, cam, want_radii=True, viewKey=seed)
    _pairs_match(r, fw["bin"].M)
    assert np.abs(_np(res.render).reshape(-1, 3) - fw["color"]).max() <= RGB_TOL
    _ncontrib_match(r, fw, W, H)
    tgt = rng.uniform(0, 1, (H, W, 3)).astype(np.float32)
    loss, cc, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)
    lo, gc, _ = r.lossForwardBackward(res.render, tgt, 0.2)
    assert abs(_np(lo)[0] - loss) < 1e-5
    z = np.zeros(W * H, np.float32)
    want = o.render_backward(p, c, W, H, 16, 16, 4, fw, cc.reshape(-1, 3), z, z, white)
    got = r.renderBackward(gc)
    for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
        w_ = want[k].reshape(_np(got[k]).shape)
        if np.abs(w_).max() > 0:
            assert _rel(_np(got[k]), w_) <= GRAD_RTOL, (k, W, H, N)
        else:
            assert not _np(got[k]).any()
    # a second forward of the same view (now with the block-work hint) gives the identical image
    res2 = r.renderForward({k: torch.as_tensor(v) for k, v in p.items()}, cam, viewKey=seed)
    assert torch.equal(res.render, res2.render)


@pytest.mark.gpu
@pytest.mark.parametrize("four_waves,trim", [(-1, 2), (0, 2), (0, 1), (0, 0)])
@pytest.mark.parametrize("seed", range(36))
def test_adversarial_small_scenes(seed, four_waves, trim):
    """tools/fuzz_parity.py: camera inside the cloud, Gaussians straddling the z >= 0.2 visibility plane, screen-filling and
    sub-pixel scales, near-zero quaternions, saturated opacities, wide and long lenses, SH degrees 0-4, tiles up to 100 px
    (larger than the image), depth / alpha cotangents: pair count, image (1e-4 of the largest colour), nContrib, the
    finite / non-finite pattern and every gradient (1e-3 relative) against the oracle.  300 further seeds were run once
    through the tool with nothing outside the bars."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "..", "tools", "fuzz_parity.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    assert fz.run_case(5000 + seed, tuning=dict(fwd_four_waves=four_waves, trim_rects=trim)) == []


# ------------------------------------------------------- depth cuts: a prefix of every tile list, results unchanged
@pytest.mark.parametrize("tile", [(16, 16), (200, 200)])
def test_depth_cuts_are_exact_and_misses_are_caught(tile):
    """gs_set_view_hints: the second forward of a view bins each tile only as deep as the first one needed it (+ margin).
    Outputs must be IDENTICAL to the uncut forward (the binned list is a prefix of the full one and every pixel has
    terminated inside it), the pair count must drop, the gradients must agree, and a cut that is too shallow must be
    reported by gs_forward_missed so that the forward is repeated in full.  (200, 200): the same per BLOCK of the block lists
    the fused path keeps at tile sizes that are not multiples of 16 (the reference app's W/4)."""
    from gaussiansplattingmlx_amd.scenes import make_config
    params, cams, (W, H) = make_config("c2_100k_800", n_views=2)
    r0 = _renderer(W, H, tile)
    tp = {k: torch.as_tensor(v, device=r0.device) for k, v in params.items()}
    cot = torch.as_tensor(np.random.default_rng(5).standard_normal((H * W, 3)).astype(np.float32), device=r0.device)
    ref = r0.renderForward(tp, cams[0])
    img0, dep0, alp0 = ref.render.clone(), ref.depth.clone(), ref.alpha.clone()
    nc0 = r0.lastContrib().clone()
    M0 = r0.stats()["M"]
    g0 = {k: v.clone() for k, v in r0.renderBackward(cot).items()}

    r = _renderer(W, H, tile)
    r.cutMinDropped = 0                                            # always cut (the default policy wants >= 6 M pairs left out)
    first = r.renderForward(tp, cams[0], viewKey="a")             # no cuts yet: the full lists
    assert not r.forwardMissed() and r.stats()["M"] == M0
    assert torch.equal(first.render, img0)
    r.renderBackward(cot)                                          # the backward's item kernel records the cuts
    hints = r._work_hints["a"]
    nblk = int(r.blockWork().numel())                              # (16 x 16 tiles: (W / 16) (H / 16); block lists: the blocks per tile)
    assert nblk == ((W // 16) * (H // 16) if tile == (16, 16) else 52 * 52) and hints.numel() == 2 * nblk
    cuts = hints[nblk:]
    # (a tile gets a cut where its list goes on beyond twice its sweep + 128 entries: fewer do since the lists are trimmed rects')
    assert int((cuts != 0).sum()) > nblk // 6, "the scene should saturate a good part of its tiles"

    second = r.renderForward(tp, cams[0], viewKey="a")            # under cuts
    assert not r.forwardMissed()
    M1 = r.stats()["M"]
    assert M1 < 0.95 * M0, (M1, M0)
    # what the cut policy is told: pairs kept / pairs a full binning makes -- incl. those of the Gaussians the projection dropped
    # whole because they lie beyond the deepest cut of every 4 x 4 tiles they touch (round 5)
    assert r._cut_policy["a"].last_dropped == M0 - M1
    assert torch.equal(second.render, img0) and torch.equal(second.depth, dep0) and torch.equal(second.alpha, alp0)
    assert torch.equal(r.lastContrib(), nc0)
    g1 = r.renderBackward(cot)
    for k in g0:
        a, b = _np(g1[k]), _np(g0[k])
        assert np.mean(np.abs(a - b) > 1e-3 * np.abs(b).max()) < 1e-4, k       # float atomics: not bit-reproducible
    third = r.renderForward(tp, cams[0], viewKey="a")             # cuts renewed from a cut list: still exact
    assert not r.forwardMissed() and torch.equal(third.render, img0)

    # cuts far too shallow: everything beyond depth 0.5 dropped -> tiles end with live pixels -> missed
    cuts.fill_(int(np.int32(np.uint32(0xFFFFFFFF - np.float32(0.5).view(np.uint32)).view(np.int32))))
    r.renderForward(tp, cams[0], viewKey="a")
    assert r.forwardMissed()
    # a caller that goes straight to the backward without asking gs_forward_missed is refused, not served gradients
    # from truncated lists
    cuts.fill_(int(np.int32(np.uint32(0xFFFFFFFF - np.float32(0.5).view(np.uint32)).view(np.int32))))
    r.renderForward(tp, cams[0], viewKey="a")
    from gaussiansplattingmlx_amd._lib import GsplatError
    with pytest.raises(GsplatError) as ei:
        r.renderBackward(cot)
    assert ei.value.code == 5 and "missed" in str(ei.value)
    cuts.fill_(int(np.int32(np.uint32(0xFFFFFFFF - np.float32(0.5).view(np.uint32)).view(np.int32))))
    again = r.renderChecked(tp, cams[0], viewKey="a")             # repeats without cuts
    assert torch.equal(again.render, img0) and torch.equal(r.lastContrib(), nc0) and r.stats()["M"] == M0
    r.renderBackward(cot)                                          # renews the cuts from the full lists
    # default policy: cuts that leave out this few pairs do not pay; the view sits out the next visits
    r.cutMinDropped = 8_000_000
    r.renderForward(tp, cams[0], viewKey="a")
    pol = r._cut_policy["a"]
    assert not r.forwardMissed() and pol.sit_out == r.cutProbeInterval and 0 < pol.last_dropped < M0
    r.renderForward(tp, cams[0], viewKey="a")
    assert r.stats()["M"] == M0 and pol.sit_out == r.cutProbeInterval - 1
    # a hint buffer without room for the cuts is refused
    small = torch.zeros(nblk, dtype=torch.int32, device=r.device)
    assert r.lib.gs_set_view_hints(r.ctx, small.data_ptr(), int(small.numel())) != 0
    # another view has its own buffer and starts without cuts
    other = r.renderForward(tp, cams[1], viewKey="b")
    assert not r.forwardMissed()
    assert torch.equal(other.render, r0.renderForward(tp, cams[1]).render)


@pytest.mark.gpu
def test_a_view_that_was_only_rendered_does_not_put_the_cut_policy_to_sleep():
    """Cuts are written by the preparation of a backward.  A view that is first rendered without one (a preview, the
    bench's capacity pre-visit) has none; its next forward must neither count as "cut" nor, finding nothing left out,
    make the policy sit out the next probe_interval visits (the garden bench lost its cuts that way: 85 M pairs binned
    instead of 3 M).  After the first real step the cuts are there and the following forward uses them."""
    from gaussiansplattingmlx_amd.scenes import make_config
    params, cams, (W, H) = make_config("c2_100k_800", n_views=1)
    r = _renderer(W, H)
    r.cutMinDropped = 1
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
    cot = torch.as_tensor(np.random.default_rng(6).standard_normal((H * W, 3)).astype(np.float32), device=r.device)
    r.renderForward(tp, cams[0], viewKey="v")                       # render only
    assert r._cut_view is None
    r.renderForward(tp, cams[0], viewKey="v")                       # still no backward has run: still uncut
    assert r._cut_view is None and not r.forwardMissed()
    M0 = r.stats()["M"]
    r.renderBackward(cot)
    pol = r._cut_policy["v"]
    assert pol.sit_out == 0 and pol.since_empty == 1
    r.renderForward(tp, cams[0], viewKey="v")                       # now under cuts
    assert r._cut_view == "v" and not r.forwardMissed()
    assert r.stats()["M"] < 0.95 * M0 and pol.sit_out == 0


def test_depth_cuts_hold_through_training():
    """40 training steps (Adam moving every parameter, a densify event in the middle) with the cuts forced on: before
    each step the forward the trainer is about to do is compared, bit for bit, with an uncut forward of the same
    parameters on a second context."""
    from gaussiansplattingmlx_amd.scenes import make_config, perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    params, cams, (W, H) = make_config("c2_100k_800", n_views=4)
    r, r2 = _renderer(W, H), _renderer(W, H)
    r.cutMinDropped = 0
    dev = r.device
    tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 7).items()}
    targets = [r2.renderForward(tp, c).render.clone() for c in cams]
    model = GaussModel(params, dev, capacity=int(params["xyz"].shape[0] * 1.5))
    tr = GaussianTrainer(model, r, iterationCount=30000)
    tr.iteration = 480                                       # densify event at iteration 500
    cut_forwards = 0
    for i in range(40):
        v = i % 4
        got = r.renderChecked(model.getParams(), cams[v], viewKey=v)
        img = got.render.clone(); nc = r.lastContrib().clone(); M_cut = r.stats()["M"]
        want = r2.renderForward(model.getParams(), cams[v])
        assert torch.equal(img, want.render), (i, v)
        assert torch.equal(nc, r2.lastContrib()), (i, v)
        cut_forwards += int(M_cut < r2.stats()["M"])
        tr.trainStep(cams[v], targets[v], viewKey=v)
    assert cut_forwards >= 10, cut_forwards                 # the cuts were actually in force for a good part of the run
    assert bool(torch.isfinite(model.arena).all())


# ------------------------------------------------------------ the reference's own vector for distTopK (row f4)
def test_dist_topk_reference_vector():
    """GaussianModelTests.swift:16-36: four points, k = 2 -> 0.5 each, through gs_dist_topk."""
    from gaussiansplattingmlx_amd.model_init import distTopK
    f = json.load(open(os.path.join(HERE, "golden", "reference_test_fixtures.json")))["dist_topk"]
    r = _renderer(64, 48)
    X = np.array(f["points"], np.float32)
    for stride in (True, False):
        got = _np(distTopK(r, X, f["k"], reference_stride=stride))
        np.testing.assert_allclose(got, f["expect"], rtol=1e-6)


# ----------------------------------------------------------------------- a5: buildPackedGaussians at op level
def test_build_packed_gaussians_matches_oracle(oracle32):
    """gs_pack_gaussians (GaussianRenderer.swift:85-99, column map :45-51) against the oracle's packed rows: a pure
    interleave, bit-exact; N = 0 and N = 1 included."""
    W, H = 200, 152
    p, cam = _scene(61, 5000, W, H)
    fw = oracle32.render_forward(p, cam.as_dict(), W, H, 16, 16, 4)
    pr = fw["proj"]
    r = _renderer(W, H)
    got = _np(r.buildPackedGaussians(pr["means2d"], pr["conic"], pr["color"], fw["opacity"], pr["depths"]))
    np.testing.assert_array_equal(got, fw["packed"])
    assert got.shape == (5000, 11)
    np.testing.assert_array_equal(got[:, 0:2], pr["means2d"])
    np.testing.assert_array_equal(got[:, 2:6], pr["conic"].reshape(-1, 4))
    np.testing.assert_array_equal(got[:, 6:9], pr["color"])
    np.testing.assert_array_equal(got[:, 9], fw["opacity"])
    np.testing.assert_array_equal(got[:, 10], pr["depths"])
    one = _np(r.buildPackedGaussians(pr["means2d"][:1], pr["conic"][:1], pr["color"][:1], fw["opacity"][:1], pr["depths"][:1]))
    np.testing.assert_array_equal(one, fw["packed"][:1])
    z = np.zeros
    assert tuple(r.buildPackedGaussians(z((0, 2)), z((0, 2, 2)), z((0, 3)), z(0), z(0)).shape) == (0, 11)


# ------------------------------------------------- a9: the reference's own call sequence, one kernel at a time
@pytest.mark.parametrize("W,H,tile,white,N", [(200, 152, (16, 16), False, 6000), (200, 152, (16, 16), True, 6000),
                                              (120, 90, (30, 30), False, 3000), (400, 400, (100, 100), True, 3000)])
def test_op_level_chain_matches_oracle_and_fused_path(oracle32, W, H, tile, white, N):
    """forward(camera, activated inputs) = projection -> gs_pack_gaussians -> gs_tile_bin -> gs_blend_forward
    (GaussianRenderer.swift:823-934 -> 769-821) and its VJP chain blend VJP -> split of gradPacked -> projection VJP
    -> activation VJPs (GaussianRenderer.swift:149-185, 605-701, 936-963), each through its own op-level entry point:
    the 5-tuple against the oracle, and the chain against the fused gs_render_forward / gs_render_backward."""
    p, cam = _scene(71, N, W, H)
    p["features_rest"] *= 0.3
    c = cam.as_dict()
    o = oracle32
    fw = o.render_forward(p, c, W, H, tile[0], tile[1], 4, white)
    r = _renderer(W, H, tile, white)
    dev = r.device
    raw = {k: torch.as_tensor(v, device=dev).requires_grad_(True) for k, v in p.items()}
    # activations on the host framework, as in the reference (get_*_from, GaussianRenderer.swift:936-963)
    means3d = r.get_xyz_from(raw["xyz"])
    opacity = r.get_opacity_from(raw["opacity"])
    scales = r.get_scales_from(raw["scales"])
    rotations = r.get_rotation_from(raw["rotation"])
    shs = r.get_features_from(raw["features_dc"], raw["features_rest"])
    np.testing.assert_allclose(_np(opacity), fw["opacity"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(_np(scales), fw["scales"], rtol=2e-6)
    np.testing.assert_allclose(_np(rotations), fw["rot"], rtol=2e-6, atol=1e-7)
    res = r.forward(cam, means3d.detach(), shs.detach(), opacity.detach(), scales.detach(), rotations.detach())
    # the 5-tuple (render, depth, alpha, visibility, radii) in the reference's shapes (:803-820)
    assert tuple(res.render.shape) == (H, W, 3) and tuple(res.depth.shape) == (H, W, 1) and tuple(res.alpha.shape) == (H, W, 1)
    assert res.visiility_filter.dtype == torch.bool and tuple(res.radii.shape) == (N,)
    assert np.abs(_np(res.render).reshape(-1, 3) - fw["color"]).max() <= RGB_TOL
    assert np.abs(_np(res.alpha).reshape(-1) - fw["alpha"]).max() <= RGB_TOL
    np.testing.assert_allclose(_np(res.depth).reshape(-1), fw["depth"], rtol=1e-4, atol=1e-4)
    radii_want = fw["proj"]["radii"]
    # the activations above are torch's exp / sigmoid, the oracle's are libm's: a radius may sit on a ceil() edge
    assert (_np(res.radii) != radii_want).mean() <= 1e-3
    np.testing.assert_array_equal(_np(res.visiility_filter), _np(res.radii) > 0)
    # (torch's exp / sigmoid vs libm's in the activations: a few more pixels may sit on the threshold than with
    # bit-identical inputs)
    _ncontrib_close(_np(r._saved["last"]), fw["last"], slack_pixels=8)
    # ... and against the fused path on the raw tensors
    r2 = _renderer(W, H, tile, white)
    fused = r2.renderForward({k: v.detach() for k, v in raw.items()}, cam, want_radii=True)
    # (tiles larger than a block: a pixel's list holds every Gaussian of its tile, most of them far from it; the op-level
    # blend adds them all up, the fused path drops -- per 8 x 8 quadrant -- those whose weight stays below 2^-29: thousands
    # of terms of up to 2e-9 each)
    both = 2e-5 if tile == (16, 16) else 4e-5
    assert (res.render - fused.render).abs().max().item() <= both
    assert (res.alpha - fused.alpha).abs().max().item() <= both
    assert (_np(res.radii) != _np(fused.radii)).mean() <= 1e-3
    if _lists_trimmed(r2):                              # (the fused path's rects are cut to what the blend can see: fewer pairs)
        assert 0.5 * r.stats()["M"] <= r2.stats()["M"] <= r.stats()["M"] * (1 + 2e-3)
    elif tile[0] % 16 == 0 and tile[1] % 16 == 0:       # (otherwise the fused path counts (Gaussian, block) pairs: block lists)
        assert r.stats()["M"] == pytest.approx(r2.stats()["M"], rel=2e-3)

    # VJP chain
    rng = np.random.default_rng(3)
    cC = torch.as_tensor(rng.normal(size=(H, W, 3)).astype(np.float32), device=dev)
    cD = torch.as_tensor((rng.normal(size=(H, W)) * 0.1).astype(np.float32), device=dev)
    cA = torch.as_tensor(rng.normal(size=(H, W)).astype(np.float32), device=dev)
    g = r.forwardWithCameraParamsVJP(cC.view(-1, 3), cD.view(-1), cA.view(-1))
    # activation VJPs by the host framework's autodiff (MLX in the reference, torch here)
    torch.autograd.backward([means3d, shs, opacity, scales, rotations],
                            [g["means3d"], g["shs"], g["opacity"].view_as(opacity), g["scales"], g["rotations"]])
    got = {k: _np(raw[k].grad) for k in raw}
    fused_g = r2.renderBackward(cC.view(-1, 3), cD.view(-1), cA.view(-1))
    fw2 = dict(fw); fw2["alpha"] = _np(res.alpha).reshape(-1)
    want = o.render_backward(p, c, W, H, tile[0], tile[1], 4, fw2, _np(cC).reshape(-1, 3), _np(cD).reshape(-1),
                             _np(cA).reshape(-1), white)
    for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
        w_ = want[k].reshape(got[k].shape)
        assert _rel(got[k], w_) <= GRAD_RTOL, ("chain vs oracle", k)
        assert _rel(got[k], _np(fused_g[k]).reshape(got[k].shape)) <= GRAD_RTOL, ("chain vs fused", k)
    # gradPacked itself, column by column, against the oracle's blend VJP
    for col in range(11):
        assert _rel(_np(g["gradPacked"])[:, col], want["gradPacked"][:, col]) <= GRAD_RTOL, col
    # size precondition of render() (GaussianRenderer.swift:789-792)
    from gaussiansplattingmlx_amd._lib import GsplatError
    with pytest.raises(GsplatError):
        r.forwardWithCameraParams(c["view"], c["proj"], c["camCenter"], c["fovX"], c["fovY"], c["focalX"], c["focalY"],
                                  W + 1, H, means3d.detach(), shs.detach(), opacity.detach(), scales.detach(),
                                  rotations.detach())


# -------------------------------------------------------- BASELINE configs[0] and configs[1] against the oracle
def _config_parity(oracle32, name, with_loss, sh_rest_scale=1.0, four_waves=-1, oracle64=None):
    from gaussiansplattingmlx_amd.scenes import make_config, perturb
    params, cams, (W, H) = make_config(name, n_views=1)
    if sh_rest_scale != 1.0:
        params = dict(params)
        params["features_rest"] = (params["features_rest"] * np.float32(sh_rest_scale)).astype(np.float32)
    cam = cams[0]
    o = oracle32
    c = cam.as_dict()
    fw = o.render_forward(params, c, W, H, 16, 16, 4)
    r = _renderer(W, H)
    r.setTuning(fwd_four_waves=four_waves)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
    res = r.renderForward(tp, cam, want_radii=True)
    st = r.stats()
    _pairs_match(r, fw["bin"].M)
    assert st["overflow"] == 0 and st["N_visible"] == int((fw["proj"]["radii"] > 0).sum())
    err = np.abs(_np(res.render).reshape(-1, 3) - fw["color"])
    cmax = float(fw["color"].max())
    np.testing.assert_array_equal(_np(res.radii), fw["proj"]["radii"])
    assert np.abs(_np(res.alpha).reshape(-1) - fw["alpha"]).max() <= RGB_TOL
    _ncontrib_match(r, fw, W, H, slack_pixels=0)       # (at most 2e-5 of the pixels, a handful of 640 000)
    tgt = o.render_forward(perturb(params, 12345), c, W, H, 16, 16, 4)["color"].reshape(H, W, 3)
    if with_loss:
        loss, cc, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)
        lo, gc, _ = r.lossForwardBackward(res.render, tgt, 0.2)
        assert abs(_np(lo)[0] - loss) < 1e-5 * max(1.0, abs(loss))
        cot_o, cot_g = cc.reshape(-1, 3), gc
    else:       # forward + backward only: the same cotangent on both sides
        cot_o = np.random.default_rng(17).normal(size=(W * H, 3)).astype(np.float32)
        cot_g = torch.as_tensor(cot_o, device=r.device)
    z = np.zeros(W * H, np.float32)
    want = o.render_backward(params, c, W, H, 16, 16, 4, fw, cot_o, z, z)
    got = r.renderBackward(cot_g)
    for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
        w_ = want[k].reshape(_np(got[k]).shape)
        if np.abs(w_).max() > 0:
            assert _rel(_np(got[k]), w_) <= GRAD_RTOL, k
        else:
            assert not _np(got[k]).any(), k
    if oracle64 is not None:      # the element-wise bar against the float32 / float64 oracle pair (_elementwise_gradient_bar)
        _elementwise_gradient_bar(f"{name}_sh{sh_rest_scale}_fw{four_waves}", got, want, oracle64, params, c, W, H, tgt,
                                  cot_fixed=None if with_loss else cot_o)
    return err, cmax


@pytest.mark.parametrize("four_waves", [-1, 0])
def test_config1_10k_400_forward_loss_backward(oracle32, oracle64, four_waves):
    """BASELINE.json configs[0] at full size (10 k random-init Gaussians, 400x400, one view; every pixel blends ~1100
    splats of opacity 0.1): forward, loss and backward against the float32 oracle.  Colours are <= 1 here, so the image
    bar is the north star's 1e-4 ABSOLUTE.  four_waves = -1: the default, which at this size is the four-waves-per-quadrant
    forward (2500 quadrants on 4096 wave slots); 0: the one-wave kernel."""
    err, cmax = _config_parity(oracle32, "c1_10k_400", with_loss=True, four_waves=four_waves, oracle64=oracle64)
    assert cmax <= 1.5 and err.max() <= RGB_TOL, (cmax, err.max())


@pytest.mark.parametrize("sh_rest_scale", [0.02, 1.0])
def test_config2_100k_800_forward_backward(oracle32, oracle64, sh_rest_scale):
    """BASELINE.json configs[1] at full size (100 k Gaussians, 800x800, projection + tile blend forward and backward of
    one view, no loss) against the float32 oracle: counts, radii, nContrib and gradients at the bench workload's bars.

    Image, sh_rest_scale = 0.02 (physical colours, <= ~2, as test_bench_workload_parity_300k_800 has them): the north
    star's bar as written, 1e-4 L-inf ABSOLUTE.
    Image, raw SURVEY 8(d) scene: this view reaches colours of 97.7 (un-normalised view directions, degree-4 basis
    ~ |d|^4), where 1e-4 absolute is 1e-6 relative -- the float32 and float64 ORACLES differ by 1.1e-3 here.  Bar: 1e-4
    relative to the largest colour (measured 2.4e-4 absolute = 2.5e-6 relative), and all but 5e-5 of the values inside the
    absolute bar too (measured 42 of 1.92 M = 2.2e-5; arithmetic variants move neither number, tools/full_size_parity.py)."""
    err, cmax = _config_parity(oracle32, "c2_100k_800", with_loss=False, sh_rest_scale=sh_rest_scale, oracle64=oracle64)
    if sh_rest_scale < 1.0:
        assert cmax < 3.0 and err.max() <= RGB_TOL, (cmax, err.max())
        return
    assert err.max() <= RGB_TOL * max(1.0, cmax), (cmax, err.max())
    assert err.max() <= 5e-4 and (err > RGB_TOL).mean() <= 5e-5


# --------------------------------------------------------------- reserved-capacity overflow is loud and harmless
def test_reserved_overflow_is_reported_and_never_applied(oracle32):
    """A forward whose pairs exceed gs_ctx_reserve's max_pairs renders nothing.  Contract (include/gsplat.h, "Overflow"):
    no optimizer step is taken from it (parameters and moments bit-identical), the error surfaces as
    GS_ERR_WORKSPACE_OVERFLOW at the next call that sees it and at gs_sync at the latest, and the trainer regrows the
    reserve and carries on."""
    from gaussiansplattingmlx_amd._lib import GsplatError
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H, N = 200, 152, 6000
    p, cam = _scene(81, N, W, H)
    c = cam.as_dict()
    fw = oracle32.render_forward(p, c, W, H, 16, 16, 4)
    r0 = _renderer(W, H)
    r0.renderForward({k: torch.as_tensor(v) for k, v in p.items()}, cam)
    M = _pairs_match(r0, fw["bin"].M)                      # (the fused path's own count: trimmed rects, GS_TUNE_TRIM_RECTS)
    r0.close()
    tgt = oracle32.render_forward(perturb(p, 5), c, W, H, 16, 16, 4)["color"].reshape(H, W, 3)
    r = _renderer(W, H)
    r.reserve(N, M // 3)                                   # too small on purpose
    model = GaussModel(p, r.device)
    model.m.fill_(0.25); model.v.fill_(0.5)                # momentum that WOULD move the parameters on a zero gradient
    before = (model.arena.clone(), model.m.clone(), model.v.clone())
    res = r.renderForward(model.getParams(), cam)
    try:                                                   # may or may not have seen the flag yet: both are in contract
        lo, gc, _ = r.lossForwardBackward(res.render, tgt, 0.2)
        r.renderBackwardAdam(gc, model.arena, model.m, model.v, [1e-2] * 6)
    except GsplatError as e:
        assert e.code == 3
    with pytest.raises(GsplatError) as ei:
        r.sync()
    assert ei.value.code == 3 and str(M) in str(ei.value)
    assert r.stats()["overflow"] == 1 and r.stats()["M"] == M
    for a, b in zip(before, (model.arena, model.m, model.v)):
        assert torch.equal(a, b)                           # nothing was applied
    assert not bool(res.render.any())                      # background only (black)
    # stand-alone Adam after an overflowed forward is gated the same way
    import ctypes as C
    g = torch.ones_like(model.arena)
    rc = r.lib.gs_adam_step(r.ctx, model.numel, model.arena.data_ptr(), g.data_ptr(), model.m.data_ptr(), model.v.data_ptr(), 1,
                            (C.c_longlong * 1)(model.numel), (C.c_float * 1)(1e-2), C.c_float(0.9), C.c_float(0.999),
                            C.c_float(1e-15), C.c_float(1.0))
    assert rc == 0
    r.lib.gs_sync(r.ctx)
    assert torch.equal(before[0], model.arena)
    # the trainer: first visit of the view -> checked -> reserve regrown -> the step is taken from a full render
    model.m.zero_(); model.v.zero_()
    tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
    tr.trainStep(cam, tgt, viewKey=0)
    r.sync()
    assert tr.overflowRecoveries == 1 and r.stats()["overflow"] == 0 and r.stats()["capM"] >= M
    assert not torch.equal(before[0], model.arena)
    assert abs(float(tr._loss[0]) - oracle32.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)[0]) < 1e-5
    # ... and an overflow that shows up later (a view already visited whose pair count grew): caught at the first call
    # that sees the device's flag, nothing applied in between
    r3 = _renderer(W, H)
    r3.reserve(N, M // 2)
    m3 = GaussModel(p, r3.device)
    tr3 = GaussianTrainer(m3, r3, iterationCount=30000, densify=False)
    tr3._checked_views.add(0)                              # as if the view had fitted on its first visit
    a0 = m3.arena.clone()
    for _ in range(4):
        tr3.trainStep(cam, tgt, viewKey=0)                 # the first overflows on the device; a later one sees the flag,
    r3.sync()                                              # regrows the reserve and repeats itself
    assert tr3.overflowRecoveries == 1 and r3.stats()["overflow"] == 0 and r3.stats()["capM"] >= M
    assert not torch.equal(a0, m3.arena) and bool(torch.isfinite(m3.arena).all())


def test_checkpoint_arena_overflow_is_reported_and_regrown():
    """The fused forward takes its checkpoint slots from an arena sized from use, not from list lengths (blend_v2.hip;
    reserved mode: one 8x8-quadrant slot per 80 reserved pairs, at least 65536).  BASELINE configs[1] (100 k Gaussians,
    800x800: deep sweeps, ~82 k slots for 2.4 M pairs) at a pair reserve that just fits its pairs needs more slots than
    that: the forward's IMAGE is complete and correct, but the step is gated and the overflow reported like a pair
    overflow; gs_ctx_reserve regrows the arena from what the forward asked for, and the repeated step gives the gradients
    of an unreserved context (whose arena holds the full bound)."""
    from gaussiansplattingmlx_amd._lib import GsplatError
    from gaussiansplattingmlx_amd.scenes import make_config
    params, cams, (W, H) = make_config("c2_100k_800", n_views=1)
    cam, N = cams[0], params["xyz"].shape[0]
    ref = _renderer(W, H)                                    # no reserve: full-bound arena
    ref.setTuning(trim_rects=0)                              # (the reference's lists: on trimmed rects the scene fits the 65536 slots)
    tp = {k: torch.as_tensor(v, device=ref.device) for k, v in params.items()}
    img = ref.renderForward(tp, cam).render.clone()
    M = ref.stats()["M"]
    cot = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(3)).to(ref.device)
    want = {k: v.clone() for k, v in ref.renderBackward(cot).items()}
    ws_full = int(ref.lib.gs_workspace_bytes(ref.ctx))
    ref.close()
    r = _renderer(W, H)
    r.setTuning(trim_rects=0)
    r.reserve(N, M + 4096)                                   # pairs fit; the arena gets 65536 slots, the scene needs ~82 k
    ws_small = int(r.lib.gs_workspace_bytes(r.ctx))
    res = r.renderForward(tp, cam)
    with pytest.raises(GsplatError) as ei:
        r.sync()
    assert ei.value.code == 3 and "checkpoint" in str(ei.value)
    assert torch.equal(res.render, img)                      # the image itself is complete
    with pytest.raises(GsplatError):
        r.renderBackward(cot)                                # ... but no backward is taken from it
    r.reserve(N, M + 4096)                                   # same sizes: regrows the arena from the forward's own count
    assert int(r.lib.gs_workspace_bytes(r.ctx)) > ws_small
    assert torch.equal(r.renderForward(tp, cam).render, img)
    got = r.renderBackward(cot)
    r.sync()
    for k in want:
        assert (got[k] - want[k]).abs().max() <= 2e-3 * want[k].abs().max() + 1e-12, k
    assert int(r.lib.gs_workspace_bytes(r.ctx)) < ws_full            # ... and still smaller than the full bound


def test_reference_param_reload_switch(oracle32):
    """GaussianTrainer.swift:1098-1110 re-reads `params` from the model after EVERY split_and_prune call; outside the densify
    window (and at every cadence without a change) the model still holds its last committed tensors, so the reference's
    training falls back to them.  Off by default (training keeps what it has learnt); referenceParamReload = True
    reproduces the reference's trajectory."""
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H, N = 160, 120, 3000
    p, cam = _scene(62, N, W, H, scale=0.06)
    r = _renderer(W, H)
    target = torch.rand(H, W, 3, device=r.device)
    out = {}
    for reload_ in (False, True):
        model = GaussModel(p, r.device)
        start = model.arena.clone()
        tr = GaussianTrainer(model, r, iterationCount=1000)
        tr.referenceParamReload = reload_
        tr.iteration = 95                                  # iterations 95 .. 100: the cadence (100) lies outside [500, 15000]
        for _ in range(5):
            tr.trainStep(cam, target)
        assert not torch.equal(model.arena, start)         # five steps have moved the parameters
        tr.trainStep(cam, target)                          # iteration 100: split_and_prune returns early, nothing committed
        torch.cuda.synchronize()
        out[reload_] = model.arena.clone()
        assert torch.equal(model.arena, start) == reload_  # the reference: back to the model's tensors; default: kept
        assert not _np(model.m).any() and not _np(model.v).any()      # optimizer state re-created either way (:1104-1109)
        tr.trainStep(cam, target)
        torch.cuda.synchronize()
        assert not torch.equal(model.arena, out[reload_])  # ... and training goes on from there


def test_interval_profiler_reports_under_the_reference_section_names(oracle32):
    """`var profiler` of the preserved Swift surface (GaussianRenderer.swift:66-68, 157-172, 579-600;
    GaussianTrainer.swift:122-243, 962-966): a profiled iteration yields the reference's report format, host sections
    plus the library's stage timers under the reference's section names."""
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    W, H, N = 200, 152, 4000
    p, cam = _scene(91, N, W, H)
    r = _renderer(W, H)
    tgt = r.renderForward({k: torch.as_tensor(v) for k, v in perturb(p, 2).items()}, cam).render.clone()
    tr = GaussianTrainer(GaussModel(p, r.device), r, iterationCount=30000, densify=False)
    tr.enableIntervalProfiling, tr.profilingLogInterval = True, 2
    logs = []
    tr.log = logs.append
    for _ in range(3):
        tr.trainStep(cam, tgt, viewKey=0)
    assert len(logs) == 2 and logs[0].startswith("[Profile] iter=0 wall=")
    dev = tr.lastProfiler.deviceSections()
    for name in ("train.forward", "train.loss.ssim", "bwd.globalTileComposite", "bwd.projectionScreenFused"):
        assert dev[name][0] > 0.0 and dev[name][1] >= 1, name
    assert "train.valueAndGrad.execute" in tr.lastProfiler.metrics and "train.forward" in tr.lastProfiler.metrics
    assert r.profiler is None
    # the op-level VJPs sit in the reference's sections when a profiler is set
    from gaussiansplattingmlx_amd.profiler import IntervalProfiler
    r.profiler = IntervalProfiler(True)
    fw = oracle32.render_forward(p, cam.as_dict(), W, H, 16, 16, 4)
    res = r.forward(cam, p["xyz"], fw["shs"], fw["opacity"], fw["scales"], fw["rot"])
    r.forwardWithCameraParamsVJP(torch.ones(W * H, 3))
    assert set(r.profiler.metrics) == {"bwd.globalTileComposite", "bwd.projectionScreenFused"}


def test_depth_gradient_knob_drops_the_depth_checkpoints_only(oracle32):
    """GS_TUNE_DEPTH_GRADIENT = 0 (what the trainer sets: its loss has no depth term): the forward checkpoints four
    planes instead of five; outputs identical, colour-only gradients identical to the five-plane run within atomics
    noise, and a backward that does bring a depth cotangent is refused."""
    from gaussiansplattingmlx_amd._lib import GsplatError
    W, H, N = 200, 152, 6000
    p, cam = _scene(101, N, W, H, scale=0.12)            # long lists: several 64-entry segments per block
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    r5, r4 = _renderer(W, H), _renderer(W, H)
    r4.setTuning(depth_gradient=0)
    a, b = r5.renderForward(tp, cam), r4.renderForward(tp, cam)
    assert torch.equal(a.render, b.render) and torch.equal(a.depth, b.depth) and torch.equal(a.alpha, b.alpha)
    assert int(r5.lastContrib().max()) > 3 * 64
    cot = torch.as_tensor(np.random.default_rng(1).normal(size=(H * W, 3)).astype(np.float32), device=r5.device)
    g5, g4 = r5.renderBackward(cot), r4.renderBackward(cot)
    for k in g5:
        assert _rel(_np(g4[k]), _np(g5[k])) <= 1e-4, k
    cD = torch.ones(H * W, device=r4.device)
    with pytest.raises(GsplatError) as ei:
        r4.renderBackward(cot, cD)
    assert ei.value.code == 1 and "GS_TUNE_DEPTH_GRADIENT" in str(ei.value)
    r5.renderBackward(cot, cD)                            # the default context takes it


@pytest.mark.parametrize("W,H", [(1024, 1024), (1040, 1024)])
def test_fused_render_is_the_same_under_both_tile_sorts(oracle32, W, H):
    """The fused forward / backward with the one-pass tile sort (4096 tiles: the last size it takes) and with the two
    8-bit passes (forced by the knob, or by the 4160 tiles of the wider image -- there the blend forward's bookkeeping
    also runs as its own kernel again instead of inside the sort's launch): identical images and nContrib, and the
    oracle's pair count."""
    N = 20000
    p, cam = _scene(111, N, W, H, scale=0.03)
    tp = {k: torch.as_tensor(v) for k, v in p.items()}
    fw = oracle32.render_forward(p, cam.as_dict(), W, H, 16, 16, 4)
    out = []
    for wide in (1, 0):
        r = _renderer(W, H)
        r.setTuning(wide_tile_sort=wide)
        res = r.renderForward(tp, cam, viewKey=0)
        _pairs_match(r, fw["bin"].M)
        img, nc = res.render.clone(), r.lastContrib().clone()
        g = r.renderBackward(torch.ones(H * W, 3, device=r.device))
        res2 = r.renderForward(tp, cam, viewKey=0)                # second visit: launch order from the view hint
        assert torch.equal(res2.render, img)
        out.append((img, nc, {k: v.clone() for k, v in g.items()}))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    assert np.abs(_np(out[0][0]).reshape(-1, 3) - fw["color"]).max() <= RGB_TOL
    for k in out[0][2]:
        assert _rel(_np(out[0][2][k]), _np(out[1][2][k])) <= 1e-4, k
