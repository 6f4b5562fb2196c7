"""Trajectory parity of the thing the bench times: K consecutive FUSED train steps against an oracle-driven loop.

The reference's iteration is lossFn -> valueAndGrad -> per-tensor Adam (GaussianTrainer.swift:958-1086, learning rates
GaussianModel.swift:56-65).  Every stage is held to the oracle on its own elsewhere; this file drives
`GaussianTrainer.trainStep` for 10 consecutive steps over 3 views with `viewKey` set -- so that from the views' second
visits on the sweep-length hints, the target-statistics cache, the depth sort's splitters, the colour riders and the fused
backward + Adam are all warm -- and compares, step by step, with a loop that calls the oracle's render_forward /
loss_forward_backward / render_backward and the numpy Adam of test_adam_step_matches_numpy.

Bars.  Per-step loss: 1e-5 absolute (measured 2e-7; the float32 and float64 ORACLE loops differ by 4e-6).  Adam moments
after the last step: 1e-3 of the tensor's largest magnitude on all but 1e-3 of the elements (measured: at most 1.1e-4 of
them beyond, largest deviation 2.8e-3).  Parameters: 1e-3 of the tensor's largest magnitude -- on all but a share of the
elements that is set by the ORACLE, not by a constant.  With eps = 1e-15 Adam's step is lr m / sqrt(v), which does not
shrink with the gradient; an element whose gradient is a cancelling sum (|sum| << sum of |terms|) has a float32 gradient
good to a few digits at best on BOTH sides, its sign can differ, and a differing sign moves the element by up to
2 x 3.16 lr in the first step whatever its size (opacity, lr 0.025: 0.16).  That is the reference's update rule, not either
implementation: the float64 oracle loop leaves 6.3e-3 of the opacities and 2.1e-3 of the SH-rest coefficients further than
1e-3 from the float32 oracle loop, with the SAME largest deviation (0.184 on an opacity) the HIP loop shows against it
(5.3e-3 and 6.6e-4 of them).  So: the HIP loop may not leave a larger share of a tensor beyond 1e-3 of the float32 oracle
loop than 1.5 x the float64 oracle loop does (+ 5e-4), and no element further than sign flips can take it (2 x 3.17 lr per
step taken).  gpurun_out/trajectory_*.json keeps every number of a run.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
KEYS = ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity")
STEPS, TOTAL = 10, 1000
LOSS_TOL, REL_TOL, MOMENT_SHARE = 1e-5, 1e-3, 1e-3


def _scene(seed, N, W, H, scale):
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    rng = np.random.default_rng(seed)
    K = 25
    p = dict(xyz=rng.uniform(-0.9, 0.9, (N, 3)), features_dc=rng.normal(0, 1, (N, 1, 3)),
             features_rest=rng.normal(0, 0.004, (N, K - 1, 3)), scales=rng.normal(np.log(scale), 0.5, (N, 3)),
             rotation=rng.normal(0, 1, (N, 4)), opacity=rng.normal(0.3, 1.5, N))
    p = {k: np.ascontiguousarray(v, np.float32) for k, v in p.items()}
    focal = 0.9 * W
    cams = [Camera(W, H, focal, focal * 1.02, look_at_c2w(eye)) for eye in ([2.2, -2.6, 1.7], [-2.9, 1.4, 1.2], [0.6, 3.1, 2.0])]
    return p, cams


def _oracle_loop(o, p0, cams, targets, W, H, steps=STEPS, lam=0.2, snapshots=None, tile=(16, 16)):
    """lossFn -> gradients -> Adam with getLearningRates, all on the CPU: the oracle's kernels + numpy's float32 Adam
    ((1 - beta) in f32, no bias correction: mlx-swift 0.30.6, as test_adam_step_matches_numpy)."""
    from gaussiansplattingmlx_amd.trainer import PARAM_ORDER, getLearningRates
    dt = o.dtype
    p = {k: v.astype(dt).copy() for k, v in p0.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(x) for k, x in p.items()}
    b1, b2, eps, one = dt.type(0.9), dt.type(0.999), dt.type(1e-15), dt.type(1)
    z = np.zeros(W * H, dt)
    losses = []
    for it in range(steps):
        cam = cams[it % len(cams)].as_dict()
        fw = o.render_forward(p, cam, W, H, tile[0], tile[1], 4)
        loss, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), targets[it % len(cams)].astype(dt), lam)
        g = o.render_backward(p, cam, W, H, tile[0], tile[1], 4, fw, cot.reshape(-1, 3), z, z)
        losses.append(loss)
        lr = dict(zip(PARAM_ORDER, getLearningRates(it, TOTAL)))
        for k in KEYS:
            gk = np.asarray(g[k], dt).reshape(p[k].shape)
            m[k] = b1 * m[k] + (one - b1) * gk
            v[k] = b2 * v[k] + (one - b2) * gk * gk
            p[k] = (p[k] - dt.type(lr[k]) * m[k] / (np.sqrt(v[k]) + eps)).astype(dt)
        if snapshots is not None:
            snapshots.append(tuple({k: x[k].copy() for k in KEYS} for x in (p, m, v)))
    return losses, p, m, v


def _oracle_loop_multi(o, p0, cams, targets, W, H, V, steps=STEPS, lam=0.2):
    """BASELINE config 4's iteration ("8 views/step is a semantic extension: loss = mean over views") on the CPU:
    per step V views, each through the reference's lossFn and its VJP, ONE Adam update from the gradient of the mean loss --
    the sum of the V view gradients at grad_scale 1 / V (the product in float32 as the optimizer kernels form it)."""
    from gaussiansplattingmlx_amd.trainer import PARAM_ORDER, getLearningRates
    dt = o.dtype
    p = {k: v.astype(dt).copy() for k, v in p0.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(x) for k, x in p.items()}
    b1, b2, eps, one = dt.type(0.9), dt.type(0.999), dt.type(1e-15), dt.type(1)
    z = np.zeros(W * H, dt)
    losses = []
    for it in range(steps):
        gsum = {k: np.zeros_like(x) for k, x in p.items()}
        lsum = 0.0
        for j in range(V):
            vi = (it * V + j) % len(cams)
            cam = cams[vi].as_dict()
            fw = o.render_forward(p, cam, W, H, 16, 16, 4)
            loss, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), targets[vi].astype(dt), lam)
            g = o.render_backward(p, cam, W, H, 16, 16, 4, fw, cot.reshape(-1, 3), z, z)
            lsum += float(loss)
            for k in KEYS:
                gsum[k] += np.asarray(g[k], dt).reshape(p[k].shape)
        losses.append(lsum / V)
        lr = dict(zip(PARAM_ORDER, getLearningRates(it, TOTAL)))
        for k in KEYS:
            gk = gsum[k] * dt.type(1.0 / V)
            m[k] = b1 * m[k] + (one - b1) * gk
            v[k] = b2 * v[k] + (one - b2) * gk * gk
            p[k] = (p[k] - dt.type(lr[k]) * m[k] / (np.sqrt(v[k]) + eps)).astype(dt)
    return losses, p, m, v


def _scene_views(W, H, n):
    """n cameras on a ring around _scene's cloud (the multi-view variants need more views than _scene's three)."""
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    focal = 0.9 * W
    eyes = [[3.4 * np.cos(2 * np.pi * i / n + 0.3), 3.4 * np.sin(2 * np.pi * i / n + 0.3), 1.2 + 0.9 * ((i * 5) % n) / n] for i in range(n)]
    return [Camera(W, H, focal, focal * 1.02, look_at_c2w(e)) for e in eyes]


def _hip_loop_multi(r, p0, cams, targets, V, steps=STEPS, **kw):
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    model = GaussModel(p0, r.device)
    tr = GaussianTrainer(model, r, iterationCount=TOTAL, densify=False, views_per_rank=V, **kw)
    tg = [torch.as_tensor(t, device=r.device) for t in targets]
    losses = []
    for it in range(steps):
        vs = [(it * V + j) % len(cams) for j in range(V)]
        loss = tr.trainStep([cams[v] for v in vs], [tg[v] for v in vs], viewKey=vs, stepCameras=[cams[v] for v in vs])
        losses.append(float(loss[0]))
    N = model.N
    params = {k: model.getParams()[k].detach().cpu().numpy().copy() for k in KEYS}
    mom = {k: model._carve(model.m, N)[k].detach().cpu().numpy().copy() for k in KEYS}
    var = {k: model._carve(model.v, N)[k].detach().cpu().numpy().copy() for k in KEYS}
    return losses, params, mom, var, tr


@pytest.mark.parametrize("variant", ["local8", "local8_unfused"])
def test_eight_views_one_update_matches_the_oracle_loop(oracle32, oracle64, variant):
    """BASELINE config 4's step -- eight views, ONE update -- on one card (round 6; the verdict's first item): the
    data-parallel form of the backward per view (colour-cotangent block + gate word, geometry slice, |grad xyz|), the local
    buffer where the all-gather goes, the SH rebuild + Adam over EIGHT blocks at grad_scale 1/8, the geometry Adam on the sum
    of eight slices -- ten steps over twelve views against an oracle loop with the mean-of-8 loss, inside the bars of
    test_train_trajectory_matches_the_oracle_loop."""
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import PARAM_ORDER, getLearningRates
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X box")
    W, H, N, V = 160, 120, 3000, 8
    p0, _ = _scene(71, N, W, H, 0.06)
    cams = _scene_views(W, H, 12)
    tp = perturb(p0, 5, 0.1)
    targets = [oracle32.render_forward(tp, c.as_dict(), W, H, 16, 16, 4)["color"].reshape(H, W, 3).copy() for c in cams]
    want_l, want_p, want_m, want_v = _oracle_loop_multi(oracle32, p0, cams, targets, W, H, V)
    ref_l, ref_p, ref_m, ref_v = _oracle_loop_multi(oracle64, p0, cams, targets, W, H, V)
    r = GaussianRenderer(4, W, H, (16, 16), False)
    got_l, got_p, got_m, got_v, tr = _hip_loop_multi(r, p0, cams, targets, V, fuse_adam=variant == "local8")
    assert r.stats()["overflow"] == 0 and tr.forwardMisses == 0
    report = dict(variant=variant, N=N, loss_hip=got_l, loss_oracle32=want_l, loss_oracle64=ref_l)
    _compare("param", got_p, want_p, p0, report)
    _compare("m", got_m, want_m, p0, report)
    _compare("v", got_v, want_v, p0, report)
    _compare("oracle32_vs_64.param", {k: ref_p[k] for k in KEYS}, want_p, p0, report)
    _compare("oracle32_vs_64.m", {k: ref_m[k] for k in KEYS}, want_m, p0, report)
    _compare("oracle32_vs_64.v", {k: ref_v[k] for k in KEYS}, want_v, p0, report)
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        try:
            json.dump(report, open(os.path.join(out, f"trajectory_{variant}_{N}.json"), "w"), indent=1)
        except OSError:
            pass
    dl = np.abs(np.asarray(got_l) - np.asarray(want_l))
    assert got_l[-1] < got_l[0] and dl.max() <= LOSS_TOL, (dl.tolist(), got_l, want_l)
    lr = dict(zip(PARAM_ORDER, getLearningRates(0, TOTAL)))
    for k in KEYS:
        for tag in ("m", "v"):
            # the single-view variants' bar -- all but 1e-3 of the elements within 1e-3, none further than 2e-2 -- with the
            # oracle pair as the yardstick for the largest deviation: a moment of the SUM of eight views' gradients is a
            # cancelling sum more often than one view's (features_rest here: scale 8e-3), and float32 summation order alone
            # moves such an element (measured, local8: m.features_rest 2.5e-2 on one element, share beyond 2e-4)
            e, ref = report[f"{tag}.{k}"], report[f"oracle32_vs_64.{tag}.{k}"]
            assert e["share_beyond"] <= MOMENT_SHARE and e["max_rel"] <= max(2e-2, 2.0 * ref["max_rel"]), (tag, k, e, ref)
        e, ref = report[f"param.{k}"], report[f"oracle32_vs_64.param.{k}"]
        assert e["share_beyond"] <= 1.5 * ref["share_beyond"] + 5e-4, (k, e, ref)
        assert e["max_abs"] <= 2 * 3.17 * lr[k] * STEPS * 1.01 + 1e-6, (k, e)


def _hip_loop(r, p0, cams, targets, variant, steps=STEPS):
    import ctypes as C
    from gaussiansplattingmlx_amd import _lib
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    model = GaussModel(p0, r.device)
    kw = dict(iterationCount=TOTAL, densify=False)
    if variant == "unfused":
        kw["fuse_adam"] = False
    elif variant.startswith("native"):
        uid = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
        assert r.lib.gs_dp_unique_id(uid) == 0
        kw.update(exchange_impl="native", dp_bootstrap=(uid.raw, 0, 1), exchange_when_single=True,
                  dp_exchange="sh_compressed" if variant == "native_sh" else "allreduce")
    tr = GaussianTrainer(model, r, **kw)
    tg = [torch.as_tensor(t, device=r.device) for t in targets]
    losses = []
    try:
        for it in range(steps):
            v = it % len(cams)
            loss = tr.trainStep(cams[v], tg[v], viewKey=v, stepCameras=[cams[v]])
            losses.append(float(loss[0]))          # (waits for the step: the loss buffer is reused)
    finally:
        tr.closeExchange()
    N = model.N
    params = {k: model.getParams()[k].detach().cpu().numpy().copy() for k in KEYS}
    mom = {k: model._carve(model.m, N)[k].detach().cpu().numpy().copy() for k in KEYS}
    var = {k: model._carve(model.v, N)[k].detach().cpu().numpy().copy() for k in KEYS}
    return losses, params, mom, var, tr


def _compare(tag, got, want, p0, report):
    """Largest relative deviation (max norm) and the share of elements beyond REL_TOL, per tensor."""
    for k in KEYS:
        a, b = np.asarray(got[k], np.float64).reshape(-1), np.asarray(want[k], np.float64).reshape(-1)
        scale = np.abs(b).max() + 1e-30
        d = np.abs(a - b) / scale
        report[f"{tag}.{k}"] = dict(max_rel=float(d.max()), share_beyond=float((d > REL_TOL).mean()), scale=float(scale),
                                    max_abs=float(np.abs(a - b).max()))


@pytest.mark.parametrize("variant,N,scale", [("fused", 3000, 0.06), ("fused", 20000, 0.03), ("unfused", 3000, 0.06),
                                             ("native_sh", 3000, 0.06), ("native_allreduce", 3000, 0.06),
                                             ("fused_tiles_50x38", 3000, 0.06)])
def test_train_trajectory_matches_the_oracle_loop(oracle32, oracle64, variant, N, scale):
    """(fused_tiles_50x38: a tile size that is not a multiple of 16 -- the fused path on block lists, include/gsplat.h -- against
    the oracle loop at THAT tile size: the reference's semantics there, every Gaussian of a pixel's tile blended.)"""
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import PARAM_ORDER, getLearningRates
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X box")
    W, H = 160, 120
    p0, cams = _scene(71, N, W, H, scale)
    tp = perturb(p0, 5, 0.1)
    tile = (50, 38) if variant == "fused_tiles_50x38" else (16, 16)
    targets = [oracle32.render_forward(tp, c.as_dict(), W, H, tile[0], tile[1], 4)["color"].reshape(H, W, 3).copy() for c in cams]
    want_l, want_p, want_m, want_v = _oracle_loop(oracle32, p0, cams, targets, W, H, tile=tile)
    ref_l, ref_p, _, _ = _oracle_loop(oracle64, p0, cams, targets, W, H, tile=tile)          # how far float32 rounding alone takes the loop
    r = GaussianRenderer(4, W, H, tile, False)
    assert r.blockLists == (tile != (16, 16))
    got_l, got_p, got_m, got_v, tr = _hip_loop(r, p0, cams, targets, variant)
    st = r.stats()
    assert st["overflow"] == 0 and tr.forwardMisses == 0
    if N > 16384:      # the depth sort took the splitter buckets and the colours rode in it from the second forward on
        assert r.lib is not None and r.getTuning("splitter_depth_sort") == 1 and r.getTuning("colour_riders") == 1
    report = dict(variant=variant, N=N, M=int(st["M"]), loss_hip=got_l, loss_oracle32=want_l, loss_oracle64=ref_l)
    _compare("param", got_p, want_p, p0, report)
    _compare("m", got_m, want_m, p0, report)
    _compare("v", got_v, want_v, p0, report)
    _compare("oracle32_vs_64.param", {k: ref_p[k] for k in KEYS}, want_p, p0, report)
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        try:
            json.dump(report, open(os.path.join(out, f"trajectory_{variant}_{N}.json"), "w"), indent=1)
        except OSError:
            pass
    # 1. the loss of every step
    dl = np.abs(np.asarray(got_l) - np.asarray(want_l))
    assert got_l[-1] < got_l[0]                                              # the steps train
    lr = dict(zip(PARAM_ORDER, getLearningRates(0, TOTAL)))
    if N > 16384:
        # The dense scene is chaotic when left to run free: at step 5 a discontinuity of the reference's own forward takes
        # the two loops to different sides (test_teacher_forced_steps_on_a_dense_scene holds every step of it to 2e-6 /
        # 2e-4 from identical states).  What is asserted here is only that the loops stay in each other's neighbourhood
        # (measured: loss 1.0e-4 at step 5, 1.4 % of the opacities beyond 1e-3 after ten steps).
        assert dl[:5].max() <= LOSS_TOL and dl.max() <= 1e-3, dl.tolist()
        for k in KEYS:
            assert report[f"param.{k}"]["share_beyond"] <= 0.05, (k, report[f"param.{k}"])
            assert report[f"param.{k}"]["max_abs"] <= 2 * 3.17 * lr[k] * STEPS * 1.01 + 1e-6, (k, report[f"param.{k}"])
        return
    assert dl.max() <= LOSS_TOL, (dl.tolist(), got_l, want_l)
    # 2. parameters and moments after the last step
    for k in KEYS:
        for tag in ("m", "v"):
            e = report[f"{tag}.{k}"]
            # (block lists: the first moment of features_dc, 9000 elements at a scale of 1e-4, has 4 .. 10 elements beyond 1e-3
            # from one run to the next -- float atomics on sums that cancel: 4.4e-4 .. 1.1e-3 of the tensor)
            share_bar = MOMENT_SHARE if tile == (16, 16) else 2.5 * MOMENT_SHARE
            assert e["share_beyond"] <= share_bar and e["max_rel"] <= 2e-2, (tag, k, e)
        e, ref = report[f"param.{k}"], report[f"oracle32_vs_64.param.{k}"]
        if tile != (16, 16):
            # Tiles larger than a block: the reference blends -- and differentiates -- every Gaussian of a pixel's tile, also
            # those whose weight on the block never reaches 2^-29, which the fused path leaves out of the block's list.  Their
            # gradients are below 1e-15 of the tensor's scale (measured on this scene: 4 % of the Gaussians, at most 9e-16), and Adam with eps = 1e-15 (no bias correction) turns ANY non-zero
            # gradient into a step of about the learning rate: the reference random-walks such elements, the fused path leaves
            # them where they are.  Measured (50 x 38 tiles, ten steps): every step's loss within 2e-6, both moments within 1e-3
            # on all but 4e-4 of the elements, and 0.4 % of the opacity / SH-rest elements further than 1e-3 against 0.07 - 0.1 %
            # between the float32 and float64 oracle loops.  Bar: 1.5 x the measured 0.4 % (round 5 held it to a flat 1 %: a
            # regression of the reach bound would have had room to hide; include/gsplat.h states the consequence).
            assert e["share_beyond"] <= 6e-3, (k, e, ref)
        else:
            assert e["share_beyond"] <= 1.5 * ref["share_beyond"] + 5e-4, (k, e, ref)
        # what sign flips of cancelling gradients can do at most: 2 x 3.17 lr per step
        assert e["max_abs"] <= 2 * 3.17 * lr[k] * STEPS * 1.01 + 1e-6, (k, e)

def test_teacher_forced_steps_on_a_dense_scene(oracle32):
    """The same ten steps on a DENSE scene (20 000 Gaussians on 160x120: lists of ~1100 per tile, depth sort through the
    splitter buckets, colours as riders), one step at a time from the oracle loop's own state: before step t the
    parameters and both moments of the float32 oracle loop after step t - 1 are loaded into the model, the trainer takes its
    step -- forward under the view's hints / cuts / caches, loss, fused backward + Adam --, and loss, parameters and moments
    are compared with the oracle loop's step t.  Free-running, this scene is chaotic: at the second visit of its third view
    a discontinuity of the reference's own forward (integer radii, tile membership, the T < 1e-4 cut) takes the HIP loop
    and the oracle loop, 1e-6 apart by then, to different sides, 290 opacity gradients change at once and the loss differs
    by 1e-4 (tools/traj_debug3.py; the float64 oracle loop happens to stay on the float32 one's side).  Fed the same state,
    every step must agree: loss 2e-6; all but 2e-4 of each tensor's elements within 1e-3 of its largest magnitude (the
    rest: Adam's step for gradients around its eps, see the module docstring)."""
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X box")
    W, H, N = 160, 120, 20000
    p0, cams = _scene(71, N, W, H, 0.03)
    tp = perturb(p0, 5, 0.1)
    targets = [oracle32.render_forward(tp, c.as_dict(), W, H, 16, 16, 4)["color"].reshape(H, W, 3).copy() for c in cams]
    snaps = []
    want_l, _, _, _ = _oracle_loop(oracle32, p0, cams, targets, W, H, snapshots=snaps)
    r = GaussianRenderer(4, W, H, (16, 16), False)
    model = GaussModel(p0, r.device)
    tr = GaussianTrainer(model, r, iterationCount=TOTAL, densify=False)
    tg = [torch.as_tensor(t, device=r.device) for t in targets]
    worst = dict(loss=0.0, share=0.0, max_rel=0.0)
    for it in range(STEPS):
        if it > 0:        # the oracle loop's state after step it - 1
            for views, src in zip((model.getParams(), model._carve(model.m, N), model._carve(model.v, N)), snaps[it - 1]):
                for k in KEYS:
                    views[k].copy_(torch.as_tensor(src[k]).reshape(views[k].shape))
        tr.iteration = it
        v = it % len(cams)
        loss = float(tr.trainStep(cams[v], tg[v], viewKey=v)[0])
        assert abs(loss - want_l[it]) <= 2e-6, (it, loss, want_l[it])
        worst["loss"] = max(worst["loss"], abs(loss - want_l[it]))
        for views, src, tag in zip((model.getParams(), model._carve(model.m, N), model._carve(model.v, N)), snaps[it], "pmv"):
            for k in KEYS:
                a, b = views[k].detach().cpu().numpy().reshape(-1).astype(np.float64), src[k].reshape(-1).astype(np.float64)
                d = np.abs(a - b) / (np.abs(b).max() + 1e-30)
                share = float((d > REL_TOL).mean())
                worst["share"], worst["max_rel"] = max(worst["share"], share), max(worst["max_rel"], float(d.max()))
                assert share <= 2e-4, (it, tag, k, share, float(d.max()))
    assert tr.forwardMisses == 0
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        json.dump(worst, open(os.path.join(out, "trajectory_teacher_forced_20000.json"), "w"))
