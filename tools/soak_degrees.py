"""Short training runs (through one densify event) at SH degrees 0 - 3 (K = 1, 4, 9, 16): the fused backward + Adam and the
data-parallel form's kernels away from K = 25, where the row spans are not float4-addressable and the riders do not travel.
usage: python tools/soak_degrees.py"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
idx, N, W, H, kind = CONFIGS["c2_100k_800"]
params, cams, _ = make_config("c2_100k_800", n_views=4)
dev = torch.device("cuda", 0)
ok = True
for degree in (0, 1, 2, 3):
    K = (degree + 1) ** 2
    p = {k: (np.ascontiguousarray(v[:, :K - 1]) if k == "features_rest" else v) for k, v in params.items()}
    for form in ("single", "unfused", "local2"):
        r = GaussianRenderer(degree, W, H, (16, 16), False)
        tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(p, 12345).items()}
        targets = [r.renderForward(tp, c).render.clone() for c in cams]
        model = GaussModel(p, dev)
        kw = dict(fuse_adam=False) if form == "unfused" else (dict(views_per_rank=2) if form == "local2" else {})
        tr = GaussianTrainer(model, r, iterationCount=30000, **kw)
        tr.iteration = 440
        for i in range(130):
            v = i % 4
            if form == "local2":
                vs = [v, (v + 1) % 4]
                loss = tr.trainStep([cams[j] for j in vs], [targets[j] for j in vs], viewKey=vs, stepCameras=[cams[j] for j in vs])
            else:
                loss = tr.trainStep(cams[v], targets[v], viewKey=v)
        torch.cuda.synchronize()
        fin = bool(torch.isfinite(model.arena).all()) and bool(torch.isfinite(model.m).all()) and bool(torch.isfinite(model.v).all())
        l = [float(x) for x in loss.cpu()]
        print(f"degree {degree} K {K} {form}: N {N} -> {model.N} loss {l[0]:.4f} ssim {l[2]:.4f} finite {fin} events {tr.lastDensifyStats}", flush=True)
        ok = ok and fin and model.N != N and l[0] < 0.5
        r.close()
print("OK" if ok else "FAILED")
sys.exit(0 if ok else 3)
