"""Diagnostic: the soak (tools/soak.py) up to the first step after which a parameter is not finite; the step is found by re-running
from a snapshot, and the Gaussian's parameters in front of it, its projection and its gradient row (unfused backward) are printed.
usage: python tools/soak_nonfinite.py [steps] (GSPLAT_TRIM_RECTS=0|1|2)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 9500
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
if os.environ.get("GSPLAT_TRIM_RECTS"):
    r.setTuning(trim_rects=int(os.environ["GSPLAT_TRIM_RECTS"]))
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev)
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 450
CHECK = 10
KEYS = ("xyz", "scales", "rotation", "opacity", "features_dc", "features_rest")
def bad_rows():
    anyb = None
    for k, v in model.getParams().items():
        b = ~torch.isfinite(v.reshape(v.shape[0], -1)).all(dim=1)
        anyb = b if anyb is None else (anyb | b)
    return anyb
snap = None
i = 0
while i < steps:
    if i % CHECK == 0 and i >= 7000:
        snap = (i, tr.iteration, model.N, model.arena.clone(), model.m.clone(), model.v.clone(),
                tr.xyzGradAccumulation.clone(), tr.denomGradAccumulation)
    v = i % 8
    n_before = model.N
    loss = tr.trainStep(cams[v], targets[v], viewKey=v)
    i += 1
    if i % 1000 == 0:
        print(f"it {tr.iteration} N {model.N} loss {float(loss[0]):.4f}", flush=True)
    if i >= 7000 and (i % CHECK == 0) and model.N == n_before and snap is not None and snap[2] == model.N:
        b = bad_rows()
        if bool(b.any()):
            rows = b.nonzero().reshape(-1).tolist()
            print(f"non-finite rows {rows[:8]} after step {i} (iteration {tr.iteration - 1}); re-running from the snapshot at step {snap[0]}")
            i0, it0, n0, a0, m0, v0, acc0, den0 = snap
            model.arena.copy_(a0); model.m.copy_(m0); model.v.copy_(v0); tr.iteration = it0
            tr.xyzGradAccumulation.copy_(acc0); tr.denomGradAccumulation = den0
            g = rows[0]
            for j in range(i0, i):
                vv = j % 8
                before = {k: model.getParams()[k][g].clone() for k in KEYS}
                mb = model.m.clone(); vb = model.v.clone(); ab = model.arena.clone()
                tr.trainStep(cams[vv], targets[vv], viewKey=vv)
                if bool(bad_rows().any()):
                    print(f"step {j} (iteration {tr.iteration - 1}, view {vv}) makes row {g} non-finite.  The row in front of the step:")
                    for k in KEYS:
                        print(f"   {k}: {[float(x) for x in before[k].reshape(-1)[:8].cpu()]}")
                    # the same step's gradient, unfused
                    model.arena.copy_(ab); model.m.copy_(mb); model.v.copy_(vb)
                    res = r.renderForward(model.getParams(), cams[vv], want_radii=True)
                    lo, cot, _ = r.lossForwardBackward(res.render, targets[vv], 0.2)
                    gr = r.renderBackward(cot)
                    print("   radius", float(res.radii[g]), "loss", [float(x) for x in lo.cpu()])
                    for k in KEYS:
                        t = gr[k][g].reshape(-1)
                        print(f"   grad {k}: {[float(x) for x in t[:8].cpu()]}  non-finite {int((~torch.isfinite(t)).sum())}")
                    nb = {k: int((~torch.isfinite(gr[k].reshape(gr[k].shape[0], -1)).all(dim=1)).sum()) for k in KEYS}
                    print("   rows with a non-finite gradient, per tensor:", nb)
                    # the oracle's projection of this one Gaussian
                    from oracle.oracle import Oracle
                    o = Oracle(np.float32)
                    one = {k: before[k].reshape((1,) + tuple(model.getParams()[k].shape[1:])).cpu().numpy() for k in KEYS}
                    c = cams[vv].as_dict()
                    op, sc, rt = o.activations_forward(one["opacity"], one["scales"], one["rotation"])
                    shs = np.concatenate([one["features_dc"], one["features_rest"]], 1)
                    pr = o.projection_forward(sc, rt, one["xyz"], shs, c["camCenter"], c["view"], c["proj"], c["fovX"], c["fovY"], c["focalX"], c["focalY"], W, H, 4)
                    print("   oracle projection:", {k: np.asarray(pr[k]).reshape(-1)[:4].tolist() for k in ("means2d", "depths", "cov2d", "conic", "radii")}, "opacity", op.tolist(), "scales", sc.tolist())
                    sys.exit(0)
            print("the re-run from the snapshot stayed finite (not reproducible: float atomics)")
            sys.exit(0)
print("no non-finite parameter in", steps, "steps")
