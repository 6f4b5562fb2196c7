"""Runs only the op-level binning (gs_tile_bin) of the bench scene, for per-kernel timing under rocprofv3 and for build
variants of binning.hip whose sorted output is not meant to be consumed.  usage: python tools/bin_only.py [config] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.renderer import GaussianRenderer, _p
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
params, cams, (W, H) = make_config(cfg, n_views=1)
r = GaussianRenderer(4, W, H)
if os.environ.get("TWO_PASS"):
    r.setTuning(wide_tile_sort=0)
r.reserve(params["xyz"].shape[0], 24 << 20)
t = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
cam = cams[0]
gc = r._camera(cam.worldViewTransform, cam.projectionMatrix, cam.cameraCenter, cam.FoVx, cam.FoVy, cam.focalX, cam.focalY)
o = r.projectionScreenFused(torch.exp(t["scales"]), r.get_rotation_from(t["rotation"]), t["xyz"],
                            torch.cat([t["features_dc"], t["features_rest"]], 1), gc)
N = t["xyz"].shape[0]
torch.cuda.synchronize()
for _ in range(reps):
    r._check(r.lib.gs_tile_bin(r.ctx, N, _p(o["rectMin"]), _p(o["rectMax"]), _p(o["radii"]), _p(o["depths"])))
torch.cuda.synchronize()
print("M", r.stats()["M"])
