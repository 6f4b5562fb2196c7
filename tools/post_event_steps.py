"""Are the steps right after a densify event slower than the same views' next visits?  Eight views cycled; per-step stage times
(stage events) for two rounds before the event, the round with the event in it and three rounds after.
usage: python tools/post_event_steps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
idx, N, W, H, kind = CONFIGS["c3_300k_800"]
V = 8
params, cams, _ = make_config("c3_300k_800", n_views=V)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 << 20)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.prewarmDensify()
tr.iteration = 600 - 4 * V - 3
rows = []
for rnd in range(8):
    for v in range(V):
        it = tr.iteration
        r.profile(True)
        tr.trainStep(cams[v], targets[v], viewKey=v)
        pr = r.profileRead(); r.profile(False)
        rows.append((rnd, v, it, model.N, {k: round(1e3 * a / max(n, 1)) for k, (a, n) in pr.items() if n}))
for rnd, v, it, n, st in rows:
    print(rnd, v, it, n, st, "EVENT" if it == 600 else "")
