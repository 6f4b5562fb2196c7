"""Diagnostic: per-item timeline of the persistent blend-forward kernel (start/end shader clocks, iterations)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config

params, cams, (W, H) = make_config("c3_300k_800", n_views=2)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(300000, 16 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
for _ in range(3):
    r.renderForward(tp, cams[0])
nItems = ((W + 15) // 16) * ((H + 15) // 16) * 2
buf = torch.zeros(nItems * 4, dtype=torch.int64, device=r.device)
r.lib.gs_debug_set_fwd_trace(C.c_void_p(buf.data_ptr()))
r.renderForward(tp, cams[0])
torch.cuda.synchronize()
r.lib.gs_debug_set_fwd_trace(None)
t = buf.cpu().numpy().reshape(-1, 4)
t0, t1, it = t[:, 0], t[:, 1], t[:, 2]
ok = (t1 > 0) & (t0 > 0)
base = t0[ok].min()
dur = (t1 - t0)[ok]
print("items", ok.sum(), "kernel span cycles", t1[ok].max() - base)
hw = t[:, 3][ok]
print("distinct hw ids", len(np.unique(hw)))
ev = torch.cuda.Event(enable_timing=True); ev2 = torch.cuda.Event(enable_timing=True)
ev.record(); r.renderForward(tp, cams[0]); ev2.record(); torch.cuda.synchronize(); print("whole forward ms", ev.elapsed_time(ev2))
print("sum iterations", it[ok].sum(), "max", it[ok].max())
cyc_per_it = dur / np.maximum(it[ok], 1)
big = it[ok] > 200
print("cycles/iteration (items > 200 its): mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f" % (
    cyc_per_it[big].mean(), *np.percentile(cyc_per_it[big], [10, 50, 90])))
order = np.argsort(-(t1[ok] - base))[:10]
print("last finishing items: (end, start, iters, cyc/it)")
for o in order:
    print(int(t1[ok][o] - base), int(t0[ok][o] - base), int(it[ok][o]), round(float(cyc_per_it[o]), 1))
# concurrency over time
ends = np.sort(t1[ok] - base)
span = ends[-1]
for frac in (0.25, 0.5, 0.75, 0.9):
    tt = span * frac
    running = ((t0[ok] - base) <= tt) & ((t1[ok] - base) > tt)
    print("t=%.0f%% running waves %d" % (frac * 100, running.sum()))
print("fixed overhead per item (items with 0 iterations): mean cycles", dur[it[ok] == 0].mean() if (it[ok] == 0).any() else None)
