"""What the bench scene looks like once the reference's own schedule has grown it to its cap (maxGaussians = 1 M): runs the
trainer from iteration 450 for `steps` iterations, then prints quantiles of the activated opacities and scales, pair counts,
sweep depths and stage times -- the numbers scenes.py's "trained_like_grown" generator is fitted to.
usage: python tools/grown_stats.py [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1150
name = sys.argv[2] if len(sys.argv) > 2 else "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev)
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 450


def describe(tag):
    p = model.getParams()
    q = torch.tensor([0.01, 0.05, 0.1, 0.25, 0.5, 0.75, 0.9, 0.95, 0.99], device=dev)
    op = torch.sigmoid(p["opacity"].reshape(-1))
    sc = torch.exp(p["scales"])
    out = {"tag": tag, "N": model.N, "it": tr.iteration,
           "opacity_q": [round(float(x), 4) for x in torch.quantile(op[:4_000_000], q)],
           "scale_max_q": [round(float(x), 5) for x in torch.quantile(sc.max(dim=1).values[:4_000_000], q)],
           "scale_min_q": [round(float(x), 5) for x in torch.quantile(sc.min(dim=1).values[:4_000_000], q)],
           "log_scale_mean_std": [round(float(p["scales"].mean()), 4), round(float(p["scales"].std()), 4)],
           "f_dc_std": round(float(p["features_dc"].std()), 4), "f_rest_std": round(float(p["features_rest"].std()), 4)}
    views = []
    for v in range(len(cams)):
        r.renderForward(p, cams[v], viewKey=v, depthCuts=False)
        st = r.stats()
        last = r.lastContrib().to(torch.int64)
        bm = last.view(H // 16, 16, W // 16, 16).amax(dim=(1, 3))
        views.append({"M": st["M"], "M_eff": int(bm.sum()), "max_list": st["max_tile_list"], "mean_nContrib": round(float(last.double().mean()), 1),
                      "N_visible": st["N_visible"]})
    out["views"] = views
    print(json.dumps(out), flush=True)


describe("start")
i = 0
while i < steps:
    tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8); i += 1
torch.cuda.synchronize()
describe("grown")
r.profile(True)
for _ in range(40):
    tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8); i += 1
torch.cuda.synchronize()
pr = r.profileRead(); r.profile(False)
print(json.dumps({"it": tr.iteration, "N": model.N, "stages_ms": {k: round(v[0] / 40, 4) for k, v in pr.items()}}), flush=True)
