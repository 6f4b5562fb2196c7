"""What a densify / prune event costs a training run in wall time: 100 views, steps around iteration 600, the run's time with
the event minus the same steps without it.  Also the host-side pieces of the event.  usage: python tools/event_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
V = 100
params, cams, _ = make_config(name, n_views=V)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 << 20)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
gc = [r._camera(c.worldViewTransform, c.projectionMatrix, c.cameraCenter, c.FoVx, c.FoVy, c.focalX, c.focalY) for c in cams]
out = {}
for with_event in (False, True, False, True):
    model = GaussModel(params, dev, capacity=int(N * 1.5))
    tr = GaussianTrainer(model, r, iterationCount=30000, densify=True)
    tr.iteration = 560 if with_event else 501          # 40 steps: with the event at iteration 600 in them, or with none
    tr.prewarmDensify()
    for v in range(V):
        res = r.renderForward(model.getParams(), gc[v], viewKey=v, wantDepth=False)
        r.lossForwardBackward(res.render, targets[v], 0.2, out=dict(loss=tr._loss, cotColor=tr._cot), targetKey=v)
    tr._checked_views.update(range(V))
    for i in range(20):
        tr.trainStep(gc[i], targets[i], viewKey=i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20, 60):
        tr.trainStep(gc[i], targets[i], viewKey=i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    print(f"event in the 40 steps: {with_event}   total {dt:.2f} ms   N after {model.N}", flush=True)
