"""Does the fused projection backward + Adam pay for a partly filled last round of waves?  185 VGPRs and 18.7 KB of LDS per wave
hold 2 waves per SIMD = 2048 on the chip; 300 k Gaussians are 4688 waves = 2.29 rounds.  Times the kernel (stage events) for N
on both sides of whole rounds.
usage: python tools/proj_bwd_rounds.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
params, cams, (W, H) = make_config("c3_300k_800", n_views=1)
big = {k: np.concatenate([v, v], 0) for k, v in params.items()}
r = GaussianRenderer(4, W, H)
r.reserve(600000, 48 << 20)
out = {}
for N in (131072, 196608, 262144, 270000, 300000, 330000, 393216, 400000, 524288):
    model = GaussModel({k: np.ascontiguousarray(v[:N]) for k, v in big.items()}, r.device)
    tr = GaussianTrainer(model, r, iterationCount=30000)
    tr.densifyFromIter = 10 ** 9
    target = r.renderForward(model.getParams(), cams[0]).render.clone()
    for _ in range(5):
        tr.trainStep(cams[0], target, viewKey=0)
    r.profile(["proj_bwd"])
    for _ in range(30):
        tr.trainStep(cams[0], target, viewKey=0)
    pr = r.profileRead(); r.profile(False)
    ms = pr["proj_bwd"][0] / pr["proj_bwd"][1]
    out[N] = dict(waves=(N + 63) // 64, rounds=round((N + 63) // 64 / 2048, 2), ms=round(ms, 4), ns_per_gaussian=round(ms * 1e6 / N, 2))
    print(N, out[N], flush=True)
print(json.dumps(out, indent=1))
