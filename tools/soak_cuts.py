"""Soak with the depth cuts forced on (bench scene, densify cadence): N, loss, speed, repeated forwards."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 800
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.cutMinDropped = 0
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev)
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 450
t0 = time.perf_counter()
for i in range(steps):
    v = i % 8
    loss = tr.trainStep(cams[v], targets[v], viewKey=v)
    if (i + 1) % 100 == 0:
        l = [float(x) for x in loss.cpu()]
        t1 = time.perf_counter()
        st = r.stats()
        print(f"it {tr.iteration} N {model.N} loss {l[0]:.4f} views/s {100 / (t1 - t0):.0f} M {st['M']} repeated forwards so far "
              f"{tr.forwardMisses} finite {bool(all(bool(torch.isfinite(v).all()) for v in model.getParams().values()))}", flush=True)
        t0 = time.perf_counter()
