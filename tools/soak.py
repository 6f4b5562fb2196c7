"""Soak: many train steps on the bench scene with the densify cadence; prints N, loss and speed every 100 iterations.
usage: python tools/soak.py [steps] [tile]"""
import sys, time
import torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
tile = int(sys.argv[2]) if len(sys.argv) > 2 else 16         # e.g. 200: the reference app's W/4 (block lists)
import os
name = os.environ.get("SOAK_CONFIG", "c3_300k_800")          # e.g. c5_garden_2m
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
white = os.environ.get("SOAK_WHITE", "0") != "0"            # white background
r = GaussianRenderer(4, W, H, (tile, tile), white)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev)
tr = GaussianTrainer(model, r, iterationCount=30000, fuse_adam=os.environ.get("SOAK_FUSE_ADAM", "1") != "0")
tr.iteration = 450
t0 = time.perf_counter()
for i in range(steps):
    v = i % 8
    loss = tr.trainStep(cams[v], targets[v], viewKey=v)
    if (i + 1) % 100 == 0:
        l = [float(x) for x in loss.cpu()]
        t1 = time.perf_counter()
        st = r.stats()
        print(f"it {tr.iteration} N {model.N} loss {l[0]:.4f} l1 {l[1]:.4f} ssim {l[2]:.4f} views/s {100 / (t1 - t0):.0f} "
              f"M {st['M']} capN {st['capN']} capM {st['capM']} finite {bool(all(bool(torch.isfinite(v).all()) for v in model.getParams().values()))} densify {tr.lastDensifyStats}", flush=True)
        t0 = time.perf_counter()
