"""Host wall time of the pieces of one densify / prune event (each followed by a device wait: the sum is an upper bound)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 << 20)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 560
tr.prewarmDensify()
for i in range(40):
    tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8)
torch.cuda.synchronize()
m = model
T = []
SYNC = len(sys.argv) > 1
def lap(name):
    if SYNC: torch.cuda.synchronize()
    T.append((name, time.perf_counter()))
lap("start")
p = m.getParams()
actions, counts = r.classifyGaussians(tr.xyzGradAccumulation, float(tr.denomGradAccumulation), p["scales"], p["opacity"].reshape(-1), tr.gradientThreshold, tr.maxScale, tr.minOpacity, True); lap("classify")
offsets, st = r.densifyOffsets(actions, counts); lap("offsets (+ count read)")
total = st["total"]
gather, mode = r.buildDensifyOutputMap(actions, offsets, total); lap("output map")
gen = torch.Generator(device=r.device); gen.manual_seed(1234)
noise = torch.randn(total, 3, generator=gen, device=r.device, dtype=torch.float32); lap("noise")
r.densifyGather(p, gather, mode, noise, out=m.stagingViews(total)); lap("gather")
m.commitStaged(); lap("commit (3 arena memsets)")
r.dropDepthCuts(); tr._alloc_exchange_buffers(); tr.resetGradientAccumulation(); lap("rest")
for (a, ta), (b, tb) in zip(T, T[1:]):
    print(f"{b:28s} {(tb - ta) * 1e3:7.3f} ms")
torch.cuda.synchronize(); print("total (host)", (T[-1][1] - T[0][1]) * 1e3, "until the device is done", (time.perf_counter() - T[0][1]) * 1e3, st)
