"""Step-0 gradients of the dense trajectory scene: where do HIP and the oracle differ in SIGN or in being zero?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_trajectory as T
from oracle.oracle import Oracle
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import perturb

o32, o64 = Oracle(np.float32), Oracle(np.float64)
W, H, N, scale = 160, 120, int(sys.argv[1]) if len(sys.argv) > 1 else 20000, float(sys.argv[2]) if len(sys.argv) > 2 else 0.03
p0, cams = T._scene(71, N, W, H, scale)
tp = perturb(p0, 5, 0.1)
cam = cams[0]; c = cam.as_dict()
tgt = o32.render_forward(tp, c, W, H, 16, 16, 4)["color"].reshape(H, W, 3).copy()
z = np.zeros(W * H, np.float32)
def oracle_grads(o):
    dt = o.dtype
    p = {k: v.astype(dt) for k, v in p0.items()}
    fw = o.render_forward(p, c, W, H, 16, 16, 4)
    loss, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt.astype(dt), 0.2)
    return o.render_backward(p, c, W, H, 16, 16, 4, fw, cot.reshape(-1, 3), z.astype(dt), z.astype(dt)), fw
g32, fw32 = oracle_grads(o32)
g64, _ = oracle_grads(o64)
r = GaussianRenderer(4, W, H, (16, 16), False)
res = r.renderForward({k: torch.as_tensor(v) for k, v in p0.items()}, cam)
lo, gc, _ = r.lossForwardBackward(res.render, tgt, 0.2)
gh = {k: v.cpu().numpy() for k, v in r.renderBackward(gc).items()}
last = r.lastContrib().cpu().numpy().reshape(-1)
print("nContrib differs at", int((last != np.asarray(fw32["last"]).reshape(-1)).sum()), "pixels of", last.size)
for k in ("opacity", "scales", "xyz", "features_dc"):
    a = gh[k].reshape(-1).astype(np.float64); b = np.asarray(g32[k]).reshape(-1).astype(np.float64); d = np.asarray(g64[k]).reshape(-1)
    mx = np.abs(b).max()
    for name, x in (("hip", a), ("o64", d)):
        zero_x, zero_b = x == 0, b == 0
        sign = (np.sign(x) != np.sign(b)) & ~zero_x & ~zero_b
        rel = np.abs(x - b) / (np.abs(b) + 1e-300)
        print(k, name, "max|g|", f"{mx:.3e}", "x=0,b!=0:", int((zero_x & ~zero_b).sum()), "x!=0,b=0:", int((~zero_x & zero_b).sum()), "sign flips:", int(sign.sum()),
              "elements with rel err > 1e-2 (b != 0):", int(((rel > 1e-2) & ~zero_b).sum()), "> 1e-1:", int(((rel > 1e-1) & ~zero_b).sum()), "of", int((~zero_b).sum()),
              "| median |b| of the >1e-2 ones / max:", f"{np.median(np.abs(b[(rel > 1e-2) & ~zero_b])) / mx if ((rel > 1e-2) & ~zero_b).any() else 0:.2e}")
print("--- Adam-relevant deviations (first step: lr 0.1 g / (0.0316 |g| + 1e-15), in units of lr)")
def adam1(g): return 0.1 * g / (np.sqrt(0.001) * np.abs(g) + 1e-15)
for k in ("opacity", "scales", "xyz", "features_dc", "features_rest", "rotation"):
    a = gh[k].reshape(-1).astype(np.float64); b = np.asarray(g32[k]).reshape(-1).astype(np.float64); d = np.asarray(g64[k]).reshape(-1)
    for name, x in (("hip", a), ("o64", d)):
        dev = np.abs(adam1(x) - adam1(b)) > 0.05
        if dev.any():
            bb, xx = np.abs(b[dev]), np.abs(x[dev])
            print(k, name, "steps off by > 0.05 lr:", int(dev.sum()), "of", b.size, "| x == 0 among them:", int((xx == 0).sum()),
                  "| |b| quantiles (10/50/90 %):", [f"{q:.1e}" for q in np.quantile(bb, [0.1, 0.5, 0.9])], "| |x| quantiles:", [f"{q:.1e}" for q in np.quantile(xx, [0.1, 0.5, 0.9])])
        else:
            print(k, name, "none")
