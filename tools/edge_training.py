"""Edge cases of a training run, none of which may raise, fault or leave a non-finite value: a model of ONE Gaussian, a camera that
sees nothing, an event that prunes everything (N = 0 from then on), every step form.
usage: python tools/edge_training.py"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_gaussians, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
W, H = 160, 120
dev = torch.device("cuda", 0)
cam_in = Camera(W, H, 150.0, 150.0, look_at_c2w([3.0, -2.5, 2.0]))
away = look_at_c2w([3.0, -2.5, 2.0]).copy(); away[:3, 2] *= -1.0; away[:3, 0] *= -1.0        # looking away from the cloud
cam_out = Camera(W, H, 150.0, 150.0, away)
ok = True
def run(tag, params, cams, forms=("single", "unfused", "local2"), prune_all=False, steps=70):
    global ok
    for form in forms:
        r = GaussianRenderer(4, W, H, (16, 16), False)
        tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 3).items()}
        targets = [r.renderForward(tp, c).render.clone() for c in cams]
        model = GaussModel(params, dev)
        kw = dict(fuse_adam=False) if form == "unfused" else (dict(views_per_rank=2) if form == "local2" else {})
        tr = GaussianTrainer(model, r, iterationCount=30000, **kw)
        tr.iteration = 470
        if prune_all:
            tr.minOpacity = 2.0                 # every Gaussian's opacity is below it: the first event prunes them all
        try:
            for i in range(steps):
                v = i % len(cams)
                if form == "local2":
                    vs = [v, (v + 1) % len(cams)]
                    loss = tr.trainStep([cams[j] for j in vs], [targets[j] for j in vs], viewKey=vs, stepCameras=[cams[j] for j in vs])
                else:
                    loss = tr.trainStep(cams[v], targets[v], viewKey=v)
            torch.cuda.synchronize()
            fin = bool(torch.isfinite(model.arena).all()) and bool(torch.isfinite(loss).all())
            print(f"{tag} {form}: N {params['xyz'].shape[0]} -> {model.N} loss {[round(float(x), 5) for x in loss.cpu()]} finite {fin} last event {tr.lastDensifyStats}", flush=True)
            ok = ok and fin
        except Exception as e:
            print(f"{tag} {form}: EXCEPTION {type(e).__name__}: {e}", flush=True)
            ok = False
        r.close()
p1 = make_gaussians(1, "trained_like", 7); p1["scales"] += 1.5
run("one Gaussian", p1, [cam_in])
p50 = make_gaussians(50, "trained_like", 8); p50["scales"] += 1.2
run("camera that sees nothing", p50, [cam_out, cam_out])
run("one view sees, one does not", p50, [cam_in, cam_out])
run("event prunes everything", p50, [cam_in, cam_in], prune_all=True)
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 3)
