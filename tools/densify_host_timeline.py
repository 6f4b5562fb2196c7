"""Where the host spends its time around a planned densify event (diagnostic).  usage: python tools/densify_host_timeline.py [planned 0|1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

planned = (sys.argv[1] if len(sys.argv) > 1 else "1") != "0"
idx, N, W, H, kind = CONFIGS["c3_300k_800"]
params, cams, _ = make_config("c3_300k_800", n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 << 20)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.plannedDensify = planned
tr.prewarmDensify()
marks = []
orig_read, orig_fwd = r.densifyPlanRead, r.renderForward
def read(*a, **k):
    t0 = time.perf_counter(); out = orig_read(*a, **k); marks.append(("plan wait", t0, time.perf_counter())); return out
def fwd(*a, **k):
    t0 = time.perf_counter(); out = orig_fwd(*a, **k); marks.append(("renderForward", t0, time.perf_counter())); return out
r.densifyPlanRead, r.renderForward = read, fwd
orig_off = r.densifyOffsets
def off(*a, **k):
    t0 = time.perf_counter(); out = orig_off(*a, **k); marks.append(("offsets wait", t0, time.perf_counter())); return out
r.densifyOffsets = off
for rep in range(3):
    tr.iteration = 557 + 100 * rep      # (39 + 4 steps later the counter stands at 600: the event's step)
    for i in range(39):
        tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8)
    torch.cuda.synchronize()
    del marks[:]
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for i in range(4):
        tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8)
    e1.record()
    t_ev0 = time.perf_counter()
    tr.trainStep(cams[4], targets[4], viewKey=4)        # iteration 600 + 100 rep: the event
    t_ev1 = time.perf_counter()
    for i in range(5, 9):
        tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8)
    e2.record()
    torch.cuda.synchronize()
    waits = [(n, a, b) for n, a, b in marks if "wait" in n]
    n, a, b = waits[-1]
    nxt = [m for m in marks if m[0] == "renderForward" and m[1] > b][0]
    print(f"planned={planned} rep {rep} N={model.N}: 4 plain steps {e0.elapsed_time(e1):.3f} ms, event step + 4 {e1.elapsed_time(e2):.3f} ms "
          f"-> event {e1.elapsed_time(e2) - e0.elapsed_time(e1) * 5 / 4:.3f} ms over plain; host: step call {1e3 * (t_ev1 - t_ev0):.3f} ms, "
          f"waited {1e3 * (b - a):.3f} ms, wait return -> next forward queued {1e3 * (nxt[2] - b):.3f} ms (of which the forward call {1e3 * (nxt[2] - nxt[1]):.3f})")
