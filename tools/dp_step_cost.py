"""Diagnostic (one GPU): compute cost of the data-parallel step's code path against the single-device step.
torch.distributed is replaced by stand-ins that move the same bytes on the device (all-gather = R copies of the local
slice, all-reduce = an in-place scale), so what is timed is everything of an R-rank step except the wire:
blend/projection backward in DP form, the SH-gradient rebuild over R views and the split Adam."""
import sys, time
import numpy as np, torch
import torch.distributed as dist
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = 60
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)


class _Done:
    def wait(self): return True


def fake_all_gather(out, inp, group=None, async_op=False):
    out.view(R, -1)[:] = inp.view(1, -1)
    return _Done()


def fake_all_reduce(t, op=None, group=None, async_op=False):
    if t.is_floating_point(): t.mul_(1.0)          # same bytes touched in place; the values stay one view's (scaled by 1/R in Adam)
    return _Done()


def run(dp):
    r = GaussianRenderer(4, W, H, (16, 16), False)
    r.reserve(int(N * 1.5), 24 * 1024 * 1024)
    tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
    targets = [r.renderForward(tp, c).render.clone() for c in cams]
    model = GaussModel(params, dev, capacity=int(N * 1.5))
    if dp:
        dist.get_world_size = lambda g=None: R
        dist.get_rank = lambda g=None: 0
        dist.all_gather_into_tensor = fake_all_gather
        dist.all_reduce = fake_all_reduce
        tr = GaussianTrainer(model, r, iterationCount=30000, process_group=object(), densify=False)
    else:
        tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
    tr.iteration = 450
    def step(i):
        v = i % 8
        sc = [cams[(v + k) % 8] for k in range(R)] if dp else None
        tr.trainStep(cams[v], targets[v], stepCameras=sc, viewKey=v)
    for i in range(16): step(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): step(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


t1 = run(False)
tR = run(True)
print(f"single-device step {t1:.4f} ms ; {R}-rank code path without the wire {tR:.4f} ms ; overhead {tR - t1:.4f} ms "
      f"-> weak-scaling ceiling {t1 / tR * R:.2f}x of {R} before any exposed communication")
