"""Diagnostic: host-side timeline of a training step under depth cuts (where does the host wait, how long do the
launch sequences take?)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel, getLearningRates

name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
tr.iteration = 450
for i in range(24):
    tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8)
torch.cuda.synchronize()
acc = dict(fwd=0.0, loss=0.0, wait=0.0, bwd=0.0)
steps = 80
t_all = time.perf_counter()
for i in range(steps):
    v = i % 8
    t0 = time.perf_counter()
    res = r.renderForward(model.getParams(), cams[v], viewKey=v)
    t1 = time.perf_counter()
    r.lossForwardBackward(res.render, targets[v], tr.lambda_dssim, out=dict(loss=tr._loss, cotColor=tr._cot))
    t2 = time.perf_counter()
    missed = r.forwardMissed()
    t3 = time.perf_counter()
    if missed:
        res = r.renderForward(model.getParams(), cams[v], viewKey=v, depthCuts=False)
        r.lossForwardBackward(res.render, targets[v], tr.lambda_dssim, out=dict(loss=tr._loss, cotColor=tr._cot))
    r.renderBackwardAdam(tr._cot, model.arena, model.m, model.v, getLearningRates(tr.iteration, tr.iterationCount))
    tr.iteration += 1
    t4 = time.perf_counter()
    acc["fwd"] += t1 - t0; acc["loss"] += t2 - t1; acc["wait"] += t3 - t2; acc["bwd"] += t4 - t3
torch.cuda.synchronize()
t_all = time.perf_counter() - t_all
print("per step ms: total %.3f | host: launch forward %.3f, launch loss %.3f, wait for the forward %.3f, launch backward+adam %.3f"
      % (t_all / steps * 1e3, *(acc[k] / steps * 1e3 for k in ("fwd", "loss", "wait", "bwd"))))
