"""A/B of the four-wave forward (GSPLAT_FWD_WIDE) on trained-like scenes rendered SMALL (fewer quadrants than wave slots), where
pixels terminate early: python tools/fwd_wide_ab.py [config] [size]   (run once with GSPLAT_FWD_WIDE=0 and once with =1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config, lego_cameras
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 400
params, _, _ = make_config(cfg, n_views=1)
cams = lego_cameras(4, S, S, 4242)
r = GaussianRenderer(4, S, S)
r.depthCuts = False
r.reserve(params["xyz"].shape[0], 24 << 20)
t = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
for i in range(8):
    res = r.renderForward(t, cams[i % 4], viewKey=i % 4)
r.profile(["proj_fwd", "bin", "blend_fwd"])
for i in range(40):
    res = r.renderForward(t, cams[i % 4], viewKey=i % 4)
torch.cuda.synchronize()
pr = r.profileRead(); r.profile(False)
st = r.stats()
print(cfg, S, "wide", os.environ.get("GSPLAT_FWD_WIDE"), {k: round(pr[k][0] / max(pr[k][1], 1), 4) for k in ("proj_fwd", "bin", "blend_fwd")},
      "M", st["M"], "checksum", float(res.render.double().sum()), int(r.lastContrib().long().sum()))
