"""Diagnostic: host time to enqueue a train step vs. device time to run it."""
import sys, time
import torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

name = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
gc = [r._camera(c.worldViewTransform, c.projectionMatrix, c.cameraCenter, c.FoVx, c.FoVy, c.focalX, c.focalY) for c in cams]
for i in range(20):
    tr.trainStep(gc[i % 8], targets[i % 8], viewKey=i % 8)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
for i in range(n):
    tr.trainStep(gc[i % 8], targets[i % 8], viewKey=i % 8)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue ms/step %.3f   total ms/step %.3f" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
