"""Per-step growth of the deviation between the HIP loop and the float32 oracle loop (dense trajectory scene)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_trajectory as T
from oracle.oracle import Oracle
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel, PARAM_ORDER, getLearningRates

o32, o64 = Oracle(np.float32), Oracle(np.float64)
W, H, N, scale = 160, 120, 20000, 0.03
p0, cams = T._scene(71, N, W, H, scale)
tp = perturb(p0, 5, 0.1)
targets = [o32.render_forward(tp, c.as_dict(), W, H, 16, 16, 4)["color"].reshape(H, W, 3).copy() for c in cams]

def oracle_steps(o):
    dt = o.dtype
    p = {k: v.astype(dt).copy() for k, v in p0.items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}; v = {k: np.zeros_like(x) for k, x in p.items()}
    b1, b2, eps, one = dt.type(0.9), dt.type(0.999), dt.type(1e-15), dt.type(1)
    z = np.zeros(W * H, dt); snaps = []; grads = []
    for it in range(T.STEPS):
        cam = cams[it % 3].as_dict()
        fw = o.render_forward(p, cam, W, H, 16, 16, 4)
        loss, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), targets[it % 3].astype(dt), 0.2)
        g = o.render_backward(p, cam, W, H, 16, 16, 4, fw, cot.reshape(-1, 3), z, z)
        lr = dict(zip(PARAM_ORDER, getLearningRates(it, T.TOTAL)))
        for k in T.KEYS:
            gk = np.asarray(g[k], dt).reshape(p[k].shape)
            m[k] = b1 * m[k] + (one - b1) * gk; v[k] = b2 * v[k] + (one - b2) * gk * gk
            p[k] = (p[k] - dt.type(lr[k]) * m[k] / (np.sqrt(v[k]) + eps)).astype(dt)
        snaps.append({k: p[k].copy() for k in T.KEYS}); grads.append({k: np.asarray(g[k]).copy() for k in ("opacity",)})
    return snaps, grads
s32, g32 = oracle_steps(o32)
s64, g64 = oracle_steps(o64)
r = GaussianRenderer(4, W, H, (16, 16), False)
model = GaussModel(p0, r.device)
tr = GaussianTrainer(model, r, iterationCount=T.TOTAL, densify=False, fuse_adam=False)
tg = [torch.as_tensor(t, device=r.device) for t in targets]
sh, ghip = [], []
for it in range(T.STEPS):
    tr.trainStep(cams[it % 3], tg[it % 3], viewKey=it % 3)
    torch.cuda.synchronize()
    sh.append({k: model.getParams()[k].cpu().numpy().copy() for k in T.KEYS}); ghip.append(model.getGrads()["opacity"].cpu().numpy().copy())
for it in range(T.STEPS):
    row = [f"step {it}"]
    for k in ("opacity", "scales", "features_rest"):
        sc = np.abs(s32[it][k]).max()
        dh = np.abs(sh[it][k].reshape(-1) - s32[it][k].reshape(-1)) / sc; d6 = np.abs(s64[it][k].reshape(-1) - s32[it][k].reshape(-1)) / sc
        row.append(f"{k}: hip {float((dh > 1e-3).mean()):.5f} o64 {float((d6 > 1e-3).mean()):.5f}")
    # this step's opacity gradients where the PREVIOUS parameters still agreed: who deviates?
    gb, gx, gd = g32[it]["opacity"].reshape(-1).astype(np.float64), ghip[it].reshape(-1).astype(np.float64), g64[it]["opacity"].reshape(-1)
    def off(x): return int((np.abs(x - gb) > 0.05 * np.abs(gb) + 1e-16).sum())
    row.append(f"opacity grads off by > 5 % (and > 1e-16): hip {off(gx)} o64 {off(gd)}")
    print(" | ".join(row), flush=True)
