"""Diagnostic: per-view step time of the bench workload (the 8-GPU step time is the slowest view's)."""
import sys, time
import torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

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
for v in range(8):
    for _ in range(5):
        tr.trainStep(cams[v], targets[v], viewKey=v)
ms = []
for v in range(8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30):
        tr.trainStep(cams[v], targets[v], viewKey=v)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 30 * 1e3)
print("per-view step ms:", [round(x, 3) for x in ms], "mean %.3f max %.3f max/mean %.3f" % (sum(ms) / 8, max(ms), max(ms) * 8 / sum(ms)))
