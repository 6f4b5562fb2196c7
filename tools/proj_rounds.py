"""Does the fused projection forward pay for its second round of workgroups?  Its LDS staging (19 KB per 128-thread block) holds 8
blocks per CU = 2048 on the chip; 300 k Gaussians are 2344 blocks.  Times the kernel (stage events) for N on both sides of 2048 blocks.
usage: python tools/proj_rounds.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
params, cams, (W, H) = make_config("c3_300k_800", n_views=1)
r = GaussianRenderer(4, W, H)
r.reserve(400000, 24 << 20)
out = {}
for N in (196608, 229376, 262144, 270000, 300000):
    tp = {k: torch.as_tensor(v[:N], device=r.device).contiguous() for k, v in params.items()}
    for _ in range(3):
        r.renderForward(tp, cams[0])
    r.profile(["proj_fwd"])
    for _ in range(20):
        r.renderForward(tp, cams[0])
    pr = r.profileRead(); r.profile(False)
    ms = pr["proj_fwd"][0] / pr["proj_fwd"][1]
    out[N] = dict(blocks=(N + 127) // 128, ms=round(ms, 4), ns_per_gaussian=round(ms * 1e6 / N, 2))
print(json.dumps(out, indent=1))
