"""Blend-backward stage time of the fused path for the library GSPLAT_LIB selects.  usage: python tools/bwd_ab.py [config] [waves_per_cu ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
params, cams, (W, H) = make_config(cfg, n_views=1)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.setTuning(depth_gradient=0)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
cot = (torch.rand(W * H, 3, device=r.device) - 0.5) * 1e-6
out = {"lib": os.environ.get("GSPLAT_LIB", "default"), "config": cfg}
for w in [int(a) for a in sys.argv[2:]] or [16]:
    r.setTuning(bwd_waves_per_cu=w)
    for _ in range(3):
        r.renderForward(tp, cams[0], viewKey=0); r.renderBackward(cot)
    r.profile(["blend_bwd", "blend_fwd"])
    for _ in range(30):
        r.renderForward(tp, cams[0], viewKey=0); r.renderBackward(cot)
    pr = r.profileRead(); r.profile(False)
    out[f"bwd_ms_w{w}"] = round(pr["blend_bwd"][0] / max(pr["blend_bwd"][1], 1), 4)
    out[f"fwd_ms_w{w}"] = round(pr["blend_fwd"][0] / max(pr["blend_fwd"][1], 1), 4)
print(json.dumps(out))
