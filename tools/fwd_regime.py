"""Where the blend forward's time goes on a GROWN scene (the soak's ~1 M Gaussians) against the 300 k bench scene: entries staged
per quadrant (its sweep length, from nContrib), entries kept by the staging cull (from the kernel's trace), wave-cycles per kept
entry.  usage: python tools/fwd_regime.py [train_steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1300
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev)
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 450

def analyse(tag, p):
    for _ in range(3):
        r.renderForward(p, cams[0], viewKey=0)
    nItems = ((W + 15) // 16) * ((H + 15) // 16) * 4
    buf = torch.zeros(nItems * 4, dtype=torch.int64, device=dev)
    r.setTuning(fwd_trace_buffer=buf.data_ptr())
    r.renderForward(p, cams[0], viewKey=0)
    torch.cuda.synchronize()
    r.setTuning(fwd_trace_buffer=0)
    t = buf.cpu().numpy().reshape(-1, 4)
    its = t[:, 2]; dur = t[:, 1] - t[:, 0]
    nc = r.lastContrib().to(torch.int64)
    Hp, Wp = -(-H // 16) * 16, -(-W // 16) * 16
    pad = torch.zeros(Hp, Wp, dtype=torch.int64, device=dev); pad[:H, :W] = nc.view(H, W)
    quad = pad.view(Hp // 16, 2, 8, Wp // 16, 2, 8).amax(dim=(2, 5))
    staged = int(quad.sum()); chunks = int(((quad + 63) // 64).sum())
    r.profile(["blend_fwd", "blend_bwd"])
    for _ in range(10):
        r.renderForward(p, cams[0], viewKey=0)
    pr = r.profileRead(); r.profile(False)
    st = r.stats()
    print(json.dumps({"scene": tag, "N": int(p["xyz"].shape[0]), "M": st["M"], "entries_staged": staged, "chunks": chunks,
                      "entries_kept": int(its.sum()), "kept_share": round(float(its.sum() / max(staged, 1)), 3),
                      "live_share_of_staged_pixel_splats": round(float(pad.sum()) / max(float(staged * 64), 1.0), 3),
                      "busy_wave_cycles": int(dur.sum()), "cycles_per_kept_entry": round(float(dur.sum() / max(its.sum(), 1)), 1),
                      "cycles_per_staged_entry": round(float(dur.sum() / max(staged, 1)), 1),
                      "blend_fwd_ms": round(pr["blend_fwd"][0] / pr["blend_fwd"][1], 4)}), flush=True)

analyse("bench scene", {k: torch.as_tensor(v, device=dev) for k, v in params.items()})
for i in range(steps):
    tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8)
torch.cuda.synchronize()
analyse("after %d train steps" % steps, {k: v.clone() for k, v in model._views.items()})
