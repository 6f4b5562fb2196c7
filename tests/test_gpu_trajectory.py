"""Trajectory parity of the thing the bench times: K consecutive FUSED train steps against an oracle-driven loop.

The reference's iteration is lossFn -> valueAndGrad -> per-tensor Adam (GaussianTrainer.swift:958-1086, learning rates
GaussianModel.swift:56-65).  Every stage is held to the oracle on its own elsewhere; this file drives
`GaussianTrainer.trainStep` for 10 consecutive steps over 3 views with `viewKey` set -- so that from the views' second
visits on the sweep-length hints, the target-statistics cache, the depth sort's splitters, the colour riders and the fused
backward + Adam are all warm -- and compares, step by step, with a loop that calls the oracle's render_forward /
loss_forward_backward / render_backward and the numpy Adam of test_adam_step_matches_numpy.

Bars.  Per-step loss: 1e-5 absolute.  Parameters and Adam moments after the last step: 1e-3 of the tensor's largest
magnitude -- on all but a small share of the elements.  Why a share: with eps = 1e-15 Adam's step is lr m / sqrt(v), a
function of the gradient's history that does not shrink with the gradient; an element whose gradient is a cancelling sum
(|sum| << sum of |terms|) has a float32 gradient good to a few digits at best on BOTH sides, its sign can differ, and a
differing sign moves the element by up to 2 x 3.16 lr in the first step whatever its size (opacity: 0.16).  That is a
property of the reference's update rule, not of either implementation -- the float32 and the float64 oracle loops differ
in the same way -- so the bar is: at most 2e-3 of a tensor's elements beyond 1e-3 (measured: see the assertion messages /
gpurun_out/trajectory_*.json), and never beyond what sign flips can do (2 x 3.17 lr per step taken).
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
LOSS_TOL, REL_TOL, SHARE_TOL = 1e-5, 1e-3, 2e-3


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


def _oracle_loop(o, p0, cams, targets, W, H, steps=STEPS, lam=0.2):
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
        fw = o.render_forward(p, cam, W, H, 16, 16, 4)
        loss, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), targets[it % len(cams)].astype(dt), lam)
        g = o.render_backward(p, cam, W, H, 16, 16, 4, fw, cot.reshape(-1, 3), z, z)
        losses.append(loss)
        lr = dict(zip(PARAM_ORDER, getLearningRates(it, TOTAL)))
        for k in KEYS:
            gk = np.asarray(g[k], dt).reshape(p[k].shape)
            m[k] = b1 * m[k] + (one - b1) * gk
            v[k] = b2 * v[k] + (one - b2) * gk * gk
            p[k] = (p[k] - dt.type(lr[k]) * m[k] / (np.sqrt(v[k]) + eps)).astype(dt)
    return losses, p, m, v


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
                                             ("native_sh", 3000, 0.06), ("native_allreduce", 3000, 0.06)])
def test_train_trajectory_matches_the_oracle_loop(oracle32, oracle64, variant, N, scale):
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    from gaussiansplattingmlx_amd.scenes import perturb
    from gaussiansplattingmlx_amd.trainer import PARAM_ORDER, getLearningRates
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on an MI355X box")
    W, H = 160, 120
    p0, cams = _scene(71, N, W, H, scale)
    tp = perturb(p0, 5, 0.1)
    targets = [oracle32.render_forward(tp, c.as_dict(), W, H, 16, 16, 4)["color"].reshape(H, W, 3).copy() for c in cams]
    want_l, want_p, want_m, want_v = _oracle_loop(oracle32, p0, cams, targets, W, H)
    ref_l, ref_p, _, _ = _oracle_loop(oracle64, p0, cams, targets, W, H)          # how far float32 rounding alone takes the loop
    r = GaussianRenderer(4, W, H, (16, 16), False)
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
    assert dl.max() <= LOSS_TOL, (dl.tolist(), got_l, want_l)
    assert got_l[-1] < got_l[0]                                              # and the steps train
    # 2. parameters and moments after the last step
    lr = dict(zip(PARAM_ORDER, getLearningRates(0, TOTAL)))
    for k in KEYS:
        for tag in ("param", "m", "v"):
            e = report[f"{tag}.{k}"]
            assert e["share_beyond"] <= SHARE_TOL, (tag, k, e, report[f"oracle32_vs_64.param.{k}"])
        # what sign flips of cancelling gradients can do at most: 2 x 3.17 lr per step
        assert report[f"param.{k}"]["max_abs"] <= 2 * 3.17 * lr[k] * STEPS * 1.01 + 1e-6, (k, report[f"param.{k}"])
