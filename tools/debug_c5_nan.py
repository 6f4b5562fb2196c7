import sys; sys.path.insert(0, ".")
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
params, cams, (W, H) = make_config("c5_garden_2m", n_views=1)
r = GaussianRenderer(4, W, H)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
res = r.renderForward(tp, cams[0], want_radii=True)
g = torch.Generator(device="cpu").manual_seed(2)
c1 = torch.randn(W * H, 3, generator=g).to(r.device)
gr = r.renderBackward(c1)
for k, v in gr.items():
    bad = ~torch.isfinite(v.reshape(v.shape[0], -1)).all(dim=1)
    print(k, "non-finite rows", int(bad.sum()))
bad = ~torch.isfinite(gr["xyz"]).all(dim=1)
ids = torch.nonzero(bad).reshape(-1)[:8].cpu().numpy()
print("ids", ids)
cam = cams[0]
V = np.asarray(cam.worldViewTransform, np.float64)
for i in ids:
    x = params["xyz"][i].astype(np.float64)
    pv = np.array([*x, 1.0]) @ V
    print(i, "xyz", x, "view z", pv[2], "scales", np.exp(params["scales"][i]), "radius", float(res.radii[i]), "grad", gr["xyz"][i].cpu().numpy(), "gscale", gr["scales"][i].cpu().numpy(), "gop", float(gr["opacity"].reshape(-1)[i]))
