"""Times one split_and_prune event of the c3 workload (diagnostic; not part of the product path)."""
import sys, time
import torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

name = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
headroom = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * headroom), int(16 * 1024 * 1024 * headroom))
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * headroom))
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 561
tr.prewarmDensify()
for i in range(39):
    tr.trainStep(cams[i % 8], targets[i % 8])
torch.cuda.synchronize()
tr.densify = False
t0 = time.perf_counter(); tr.trainStep(cams[0], targets[0]); torch.cuda.synchronize(); t1 = time.perf_counter()
tr.densify = True
tr.iteration = 600
ta = time.perf_counter(); tr.trainStep(cams[1], targets[1]); torch.cuda.synchronize(); tb = time.perf_counter()
print("plain step ms", (t1 - t0) * 1e3, "step with event ms", (tb - ta) * 1e3, tr.lastDensifyStats)
# stage timing of the event itself on the new model
tr.iteration = 700
for i in range(5):
    tr.trainStep(cams[i % 8], targets[i % 8])
torch.cuda.synchronize()
t = [time.perf_counter()]
p = model.getParams()
a, c = r.classifyGaussians(tr.xyzGradAccumulation, 5.0, p["scales"], p["opacity"].reshape(-1)); torch.cuda.synchronize(); t.append(time.perf_counter())
off, st = r.densifyOffsets(a, c); t.append(time.perf_counter())
g, m = r.buildDensifyOutputMap(a, off, st["total"]); torch.cuda.synchronize(); t.append(time.perf_counter())
nz = torch.randn(st["total"], 3, device=dev); torch.cuda.synchronize(); t.append(time.perf_counter())
new = r.densifyGather(p, g, m, nz); torch.cuda.synchronize(); t.append(time.perf_counter())
model.commit(new); torch.cuda.synchronize(); t.append(time.perf_counter())
print("classify, offsets, map, noise, gather, commit ms:", [round((y - x) * 1e3, 3) for x, y in zip(t, t[1:])], st)
