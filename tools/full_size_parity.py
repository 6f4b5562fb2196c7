import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from oracle.oracle import Oracle
params, cams, (W, H) = make_config("c3_300k_800", n_views=1)
for scale in (1.0, 0.02):
    p = dict(params); p["features_rest"] = params["features_rest"] * scale
    o = Oracle(np.float32); o64 = Oracle(np.float64)
    c = cams[0].as_dict()
    fw = o.render_forward(p, c, W, H, 16, 16, 4)
    fw64 = o64.render_forward(p, c, W, H, 16, 16, 4)
    r = GaussianRenderer(4, W, H)
    res = r.renderForward({k: torch.as_tensor(v, device=r.device) for k, v in p.items()}, cams[0])
    img = res.render.cpu().numpy().reshape(-1, 3)
    d = np.abs(img - fw["color"]); d64 = np.abs(img - fw64["color"]); o_vs_64 = np.abs(fw["color"] - fw64["color"])
    print("scale", scale, "max colour", fw["color"].max(), "HIP-oracle32 Linf", d.max(), "frac>1e-4", (d > 1e-4).mean(),
          "| HIP-oracle64 Linf", d64.max(), "| oracle32-oracle64 Linf", o_vs_64.max(), "| rel", d.max() / fw["color"].max())
