"""Diagnostic: host cost of one trainStep (tiny scene, so the device never limits)."""
import sys, time, cProfile, pstats
import numpy as np, torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_gaussians
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
W = H = 64
p = make_gaussians(200, "trained_like", 3)
cam = Camera(W, H, 60.0, 60.0, look_at_c2w([3.0, -2.5, 2.0]))
r = GaussianRenderer(4, W, H)
r.reserve(1000, 1 << 16)
model = GaussModel(p, r.device, capacity=1000)
tr = GaussianTrainer(model, r, iterationCount=30000, densify=True)
tr.densifyFromIter = 10 ** 9
gc = r._camera(cam.worldViewTransform, cam.projectionMatrix, cam.cameraCenter, cam.FoVx, cam.FoVy, cam.focalX, cam.focalY)
tgt = torch.rand(H, W, 3, device=r.device)
for _ in range(50): tr.trainStep(gc, tgt, viewKey=0)
torch.cuda.synchronize()
n = 500
t0 = time.perf_counter()
for _ in range(n): tr.trainStep(gc, tgt, viewKey=0)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host ms/step %.3f (incl. device drain %.3f)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(200): tr.trainStep(gc, tgt, viewKey=0)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
