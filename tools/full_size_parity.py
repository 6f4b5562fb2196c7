"""The image bar at full size, per build variant (GSPLAT_LIB selects the library): L-inf / tail / RMS of the HIP render
against the float32 and float64 oracles on the raw SURVEY 8(d) scenes, plus the blend stage times.
usage: [GSPLAT_LIB=...so] python tools/full_size_parity.py [config ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from oracle.oracle import Oracle
out = {"lib": os.environ.get("GSPLAT_LIB", "default")}
for cfg in (sys.argv[1:] or ["c2_100k_800", "c3_300k_800"]):
    params, cams, (W, H) = make_config(cfg, n_views=1)
    for scale in (1.0, 0.02):
        p = dict(params); p["features_rest"] = (params["features_rest"] * np.float32(scale)).astype(np.float32)
        o, o64 = Oracle(np.float32), Oracle(np.float64)
        c = cams[0].as_dict()
        fw = o.render_forward(p, c, W, H, 16, 16, 4)
        fw64 = o64.render_forward(p, c, W, H, 16, 16, 4)
        r = GaussianRenderer(4, W, H)
        tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
        res = r.renderForward(tp, cams[0])
        img = res.render.cpu().numpy().reshape(-1, 3).astype(np.float64)
        d, d64, o_vs_64 = np.abs(img - fw["color"]), np.abs(img - fw64["color"]), np.abs(fw["color"] - fw64["color"])
        last = r.lastContrib().cpu().numpy().reshape(-1).astype(np.int64)
        cot = torch.ones(W * H, 3, device=r.device)
        r.profile(["blend_fwd", "blend_bwd"])
        for _ in range(20):
            r.renderForward(tp, cams[0], viewKey=0); r.renderBackward(cot)
        pr = r.profileRead(); r.profile(False)
        out[f"{cfg} sh_rest x{scale}"] = {
            "max_colour": float(fw["color"].max()), "hip_vs_oracle32_linf": float(d.max()), "values_over_1e-4": int((d > 1e-4).sum()),
            "frac_over_1e-4": float((d > 1e-4).mean()), "hip_vs_oracle32_rms": float(np.sqrt((d ** 2).mean())),
            "hip_vs_oracle64_linf": float(d64.max()), "hip_vs_oracle64_rms": float(np.sqrt((d64 ** 2).mean())),
            "oracle32_vs_oracle64_linf": float(o_vs_64.max()), "oracle32_vs_oracle64_rms": float(np.sqrt((o_vs_64 ** 2).mean())),
            "nContrib_mismatches": int((last != fw["last"].astype(np.int64)).sum()),
            "blend_fwd_ms": pr["blend_fwd"][0] / max(pr["blend_fwd"][1], 1), "blend_bwd_ms": pr["blend_bwd"][0] / max(pr["blend_bwd"][1], 1)}
        r.close()
print(json.dumps(out, indent=1))
