import json, os, sys
sys.path.insert(0, "/root/repo")
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
params, cams, (W, H) = make_config("c3_300k_800", n_views=1)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.setTuning(depth_gradient=0)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
cot = (torch.rand(W * H, 3, device=r.device) - 0.5) * 1e-6
for f in (3, 4, 5, 6):
    for b in (12, 14, 16, 18, 20):
        r.setTuning(fwd_waves_per_simd=f, bwd_waves_per_cu=b)
        for _ in range(3):
            r.renderForward(tp, cams[0], viewKey=0); r.renderBackward(cot)
        r.profile(["blend_bwd", "blend_fwd"])
        for _ in range(30):
            r.renderForward(tp, cams[0], viewKey=0); r.renderBackward(cot)
        pr = r.profileRead(); r.profile(False)
        print(f"fwd waves/SIMD {f} bwd waves/CU {b}: fwd {pr['blend_fwd'][0] / pr['blend_fwd'][1]:.4f} bwd {pr['blend_bwd'][0] / pr['blend_bwd'][1]:.4f}", flush=True)
