"""Blend-forward stage time against the number of persistent waves per SIMD (GS_TUNE_FWD_WAVES_PER_SIMD), fixed scene, hinted order.
usage: python tools/fwd_waves.py [config]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
params, cams, (W, H) = make_config(cfg, n_views=2)
r = GaussianRenderer(4, W, H)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
out = {}
for wps in (1, 2, 3, 4, 5, 6, 8):
    r.setTuning(fwd_waves_per_simd=wps)
    for _ in range(4):
        r.renderForward(tp, cams[0], viewKey=0)
    r.profile(["blend_fwd"])
    for _ in range(30):
        r.renderForward(tp, cams[0], viewKey=0)
    pr = r.profileRead(); r.profile(False)
    out[wps] = round(pr["blend_fwd"][0] / pr["blend_fwd"][1], 4)
print(json.dumps(out))
