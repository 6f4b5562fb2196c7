"""Soak of the reserved-capacity path with a reserve that is too tight on purpose: the model grows through the densify
cadence, the pair count outgrows the reserve again and again, and every time the trainer has to notice (the device
gate keeps the optimizer from stepping on a blank render), regrow and carry on.  Prints N, loss, speed, recoveries."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
M0 = max(r.stats()["M"], 1)
r.reserve(int(N * 4), int(M0 * 1.02))     # room for every Gaussian to come, but pairs for the first views only
model = GaussModel(params, dev, capacity=int(N * 4))
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 450
t0 = time.perf_counter()
first = None
for i in range(steps):
    v = i % 8
    loss = tr.trainStep(cams[v], targets[v], viewKey=v)
    if (i + 1) % 100 == 0:
        l = [float(x) for x in loss.cpu()]
        first = first or l[0]
        t1 = time.perf_counter()
        st = r.stats()
        print(f"it {tr.iteration} N {model.N} loss {l[0]:.4f} views/s {100 / (t1 - t0):.0f} M {st['M']} capM {st['capM']} overflow now "
              f"{st['overflow']} recoveries {tr.overflowRecoveries} finite {bool(all(bool(torch.isfinite(v).all()) for v in model.getParams().values()))}", flush=True)
        t0 = time.perf_counter()
r.sync()
assert bool(all(bool(torch.isfinite(v).all()) for v in model.getParams().values()))
print("done: recoveries", tr.overflowRecoveries, "first/last loss", first, l[0])
