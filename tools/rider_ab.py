"""A/B of the colour riders (GS_TUNE_COLOUR_RIDERS): forward stage times per setting on a fixed scene, and the images compared bit
for bit.  0: one projection kernel; 2: geometry kernel + all colours in a kernel of their own in front of the blend; 1: the colours
as riders of the binning kernels.  usage: python tools/rider_ab.py [config]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
params, cams, (W, H) = make_config(cfg, n_views=2)
r = GaussianRenderer(4, W, H)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
ref = None
modes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 2, 1, 0, 1]
for mode in modes:
    r.setTuning(colour_riders=mode)
    for _ in range(4):
        res = r.renderForward(tp, cams[0], viewKey=0, depthCuts=False)
    img = res.render.clone(); nc = r.lastContrib().clone()
    if ref is None:
        ref = (img, nc)
    same = bool(torch.equal(img, ref[0]) and torch.equal(nc, ref[1]))
    r.profile(["proj_fwd", "bin", "blend_fwd"])
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(30):
        r.renderForward(tp, cams[0], viewKey=0, depthCuts=False)
    ev1.record(); torch.cuda.synchronize()
    pr = r.profileRead(); r.profile(False)
    print(json.dumps({"colour_riders": mode, "identical_to_mode_0": same, "forward_ms_incl_events": round(ev0.elapsed_time(ev1) / 30, 4),
                      **{k: round(pr[k][0] / max(pr[k][1], 1), 4) for k in ("proj_fwd", "bin", "blend_fwd")}}), flush=True)
