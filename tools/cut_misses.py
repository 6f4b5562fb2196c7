"""Diagnostic: how often does a forward under depth cuts miss on the bench scene, and how many pairs do the cuts leave?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

name = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 * 1024 * 1024 if N <= 400_000 else 96 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000, densify=False)
tr.iteration = 450
nblk = ((W + 15) // 16) * ((H + 15) // 16)
for i in range(steps):
    v = i % 8
    before = tr.forwardMisses
    if v in r._work_hints:          # look at the forward the step is about to do: which tiles would miss, and by how much?
        used = r._work_hints[v].clone()
        res = r.renderForward(model.getParams(), cams[v], viewKey=v)
        if r.forwardMissed():
            import ctypes as C
            cut = used[nblk:] != 0
            rng_ = torch.empty(nblk, 2, dtype=torch.int32, device=dev); cnt_ = torch.empty(nblk, dtype=torch.int32, device=dev)
            r._check(r.lib.gs_tile_bin_export(r.ctx, None, C.c_void_p(rng_.data_ptr()), C.c_void_p(cnt_.data_ptr())))
            len_cut = cnt_.clone()
            work_cut = r._work_hints[v][:nblk].clone()
            bad = cut & (work_cut >= len_cut) & (len_cut > 0)       # a pixel live at the end of the list has nContrib = len
            full = r.renderForward(model.getParams(), cams[v], viewKey=v, depthCuts=False)
            work_full = r._work_hints[v][:nblk].clone()
            work_prev = used[:nblk]
            ratio = (work_full[bad].float() / work_prev[bad].float().clamp(min=1))
            wp = work_prev[bad].float()
            print(f"   on missed tiles: cut list length median {float(len_cut[bad].float().median()):.0f}, expected >= 1.25 prev + 65 median {float((wp * 1.25 + 65).median()):.0f}; "
                  f"sweep on the cut list median {float(work_cut[bad].float().median()):.0f}; tiles whose cut list is shorter than prev sweep: {int((len_cut[bad] < work_prev[bad]).sum())}")
            print(f"   missed tiles {int(bad.sum())} of {int(cut.sum())} under a cut; sweep now / sweep then on those: "
                  f"median {float(ratio.median()):.2f} max {float(ratio.max()):.2f}; prev sweeps there: median {float(work_prev[bad].float().median()):.0f}")
        r._work_hints[v].copy_(used)
    tr.trainStep(cams[v], targets[v], viewKey=v)
    st = r.stats()
    cuts = r._work_hints[v][nblk:]
    print(f"step {i} view {v} missed {tr.forwardMisses - before} M {st['M']} tiles under a cut next time {int((cuts != 0).sum())}", flush=True)
