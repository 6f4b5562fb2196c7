"""Forwards only (fused path) of a config, for per-kernel timing under rocprofv3: python tools/fwd_only.py [config] [reps] [views]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
V = int(sys.argv[3]) if len(sys.argv) > 3 else 4
params, cams, (W, H) = make_config(cfg, n_views=V)
r = GaussianRenderer(4, W, H)
r.depthCuts = False
r.reserve(params["xyz"].shape[0], (96 if params["xyz"].shape[0] > 1_000_000 else 24) << 20)
t = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
for i in range(reps):
    r.renderForward(t, cams[i % V], viewKey=i % V, wantDepth=False)
torch.cuda.synchronize()
print("M", r.stats()["M"])
