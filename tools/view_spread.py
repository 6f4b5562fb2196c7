"""Per-view step time of the bench workload, and what the slowest-of-R rule of a data-parallel step makes of it: an R-rank
step takes as long as its slowest rank's view.  Measures every view's single-device step time (100 views, 20 steps each, no
densify), then averages max-over-the-step's-views for the index order and for trainer.balanced_view_order (views sorted by the
block-entries their forward traverses, zigzag) -- a one-card estimate of the straggler tax at R = 2, 4, 8.
usage: python tools/view_spread.py [views]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel, balanced_view_order, view_for

V = int(sys.argv[1]) if len(sys.argv) > 1 else 100
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=V)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
start = model.arena.clone(); m0 = model.m.clone(); v0 = model.v.clone()
for v in range(V):
    for _ in range(3):
        tr.trainStep(cams[v], targets[v], viewKey=v)
nblk = ((W + 15) // 16) * ((H + 15) // 16)
ms = []
for v in range(V):
    model.arena.copy_(start); model.m.copy_(m0); model.v.copy_(v0)          # every view timed on the same scene
    tr.trainStep(cams[v], targets[v], viewKey=v)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        tr.trainStep(cams[v], targets[v], viewKey=v)
    torch.cuda.synchronize(); ms.append((time.perf_counter() - t0) / 20 * 1e3)
model.arena.copy_(start)
costs = []
for v in range(V):
    r.renderForward(model.getParams(), cams[v], viewKey=v, depthCuts=False)
    costs.append(int(r._work_hints[v][:nblk].to(torch.int64).sum().item()))
ms = np.array(ms); costs = np.array(costs, np.float64)
out = {"views": V, "step_ms_mean": round(float(ms.mean()), 4), "p10": round(float(np.percentile(ms, 10)), 4), "p90": round(float(np.percentile(ms, 90)), 4),
       "min": round(float(ms.min()), 4), "max": round(float(ms.max()), 4),
       "corr_step_ms_vs_traversed_block_entries": round(float(np.corrcoef(ms, costs)[0, 1]), 3)}
bal = balanced_view_order(list(costs))
for R in (2, 4, 8):
    steps = range(V * R // np.gcd(V, R) // R)          # whole passes
    def mean_max(order):
        return float(np.mean([max(ms[order[view_for(s, q, R, V)]] for q in range(R)) for s in steps]))
    a, b = mean_max(list(range(V))), mean_max(bal)
    out[f"R{R}"] = {"index_order_ms": round(a, 4), "balanced_ms": round(b, 4), "tax_index_order": round(a / ms.mean() - 1, 4),
                    "tax_balanced": round(b / ms.mean() - 1, 4)}
print(json.dumps(out))
