"""One fuzz case in detail: python tools/fuzz_one.py <seed>   (prints the forward and every gradient tensor against the oracle)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tools.fuzz_parity as fz
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
from oracle.oracle import Oracle
s = int(sys.argv[1])
print(fz.run_case(s))
o = Oracle(np.float32)
rng = np.random.default_rng(77000 + s)
W, H = int(rng.integers(9, 130)), int(rng.integers(9, 130))
N = int(rng.choice([1, 3, 64, 65, 200, 900, 4000]))
K = int(rng.choice([1, 4, 9, 16, 25])); deg = {1: 0, 4: 1, 9: 2, 16: 3, 25: 4}[K]
tile = tuple(int(rng.choice([16, 32, 48, 100])) for _ in range(2)) if rng.random() < 0.3 else (16, 16)
white = bool(rng.integers(0, 2))
mode = s % 6
eye = np.array([2.2, -2.6, 1.7]) * (rng.uniform(0.05, 0.4) if mode == 0 else 1.0)
fmul = float(rng.choice([0.25, 0.9, 3.0]))
cam = Camera(W, H, fmul * W, fmul * 1.02 * W, look_at_c2w(list(eye)))
xyz = rng.uniform(-1, 1, (N, 3)); scales = rng.normal(np.log(0.05), 0.5, (N, 3)); rot = rng.normal(0, 1, (N, 4)); opac = rng.normal(0.3, 1.5, N)
assert mode == 0, "this helper rebuilds mode-0 cases only"
p = dict(xyz=xyz, features_dc=rng.normal(0, 1, (N, 1, 3)), features_rest=rng.normal(0, 0.05, (N, K - 1, 3)), scales=scales, rotation=rot, opacity=opac)
p = {k: np.ascontiguousarray(v, np.float32) for k, v in p.items()}
c = cam.as_dict()
fw = o.render_forward(p, c, W, H, tile[1], tile[0], deg, white)
print("W H", W, H, "tile", tile, "N", N, "M", fw["bin"].M, "radii", fw["proj"]["radii"], "means2d", fw["proj"]["means2d"], "conic", fw["proj"]["conic"].reshape(-1, 4))
print("opacity act", 1 / (1 + np.exp(-opac)), "nContrib max", fw["last"].max(), "alpha max", fw["alpha"].max() if "alpha" in fw else None)
cot = rng.normal(0, 1, (H * W, 3)).astype(np.float32); z = np.zeros(W * H, np.float32)
want = o.render_backward(p, c, W, H, tile[1], tile[0], deg, fw, cot, z, z, white)
for tl in (tile, (16, 16)):
    r = GaussianRenderer(deg, W, H, (tl[1], tl[0]), white)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
    res = r.renderForward(tp, cam)
    got = r.renderBackward(torch.as_tensor(cot, device=r.device))
    print("tile", tl)
    for k in got:
        print(" ", k, got[k].cpu().numpy().reshape(-1)[:6], want[k].reshape(-1)[:6])
    r.close()
