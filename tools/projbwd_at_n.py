"""Fused projection backward + Adam: time and HBM rate against the Gaussian count (c3's generator at several N, every
Gaussian in view) and on the garden scene (48 % of the Gaussians touch no tile).  usage: python tools/projbwd_at_n.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_gaussians, lego_cameras, garden_cameras
from gaussiansplattingmlx_amd.trainer import GaussModel, getLearningRates
for kind, N, W, H in (("trained_like", 300_000, 800, 800), ("trained_like", 1_000_000, 800, 800), ("trained_like", 2_000_000, 800, 800),
                      ("garden", 2_000_000, 1237, 822)):
    params = make_gaussians(N, kind, 5)
    if kind == "trained_like" and N > 300_000:
        params["scales"] -= np.float32(np.log((N / 300_000) ** 0.5))        # keep the pair count in bounds
    cam = (garden_cameras if kind == "garden" else lego_cameras)(1, W, H, 7)[0]
    r = GaussianRenderer(4, W, H, (16, 16), False)
    r.setTuning(depth_gradient=0)
    model = GaussModel(params, r.device)
    cot = (torch.rand(H, W, 3, device=r.device) - 0.5) * 1e-6
    lrs = getLearningRates(100, 30000)
    for _ in range(3):
        r.renderForward(model.getParams(), cam, viewKey=0, wantDepth=False, depthCuts=False); r.renderBackwardAdam(cot, model.arena, model.m, model.v, lrs)
    r.profile(["proj_bwd"])
    for _ in range(20):
        r.renderForward(model.getParams(), cam, viewKey=0, wantDepth=False, depthCuts=False); r.renderBackwardAdam(cot, model.arena, model.m, model.v, lrs)
    pr = r.profileRead(); r.profile(False)
    ms = pr["proj_bwd"][0] / pr["proj_bwd"][1]
    st = r.stats()
    bytes_ = N * (344 + 64 + 86 * 24)
    print(f"{kind:13s} N {N:8d} visible {st['N_visible']:8d} M {st['M']:9d}  proj_bwd+Adam {ms:.4f} ms  {bytes_ / ms / 1e9:.2f} TB/s of designed bytes", flush=True)
    r.close(); del model
