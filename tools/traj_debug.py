"""Which warm path moves the dense trajectory scene's loss at step 5?  (tests/test_gpu_trajectory.py, N = 20000)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_trajectory as T
from oracle.oracle import Oracle
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

o32 = Oracle(np.float32)
W, H, N, scale = 160, 120, 20000, 0.03
p0, cams = T._scene(71, N, W, H, scale)
tp = perturb(p0, 5, 0.1)
targets = [o32.render_forward(tp, c.as_dict(), W, H, 16, 16, 4)["color"].reshape(H, W, 3).copy() for c in cams]
want_l, want_p, _, _ = T._oracle_loop(o32, p0, cams, targets, W, H)
for mode in ("default", "nocuts", "nokey", "noriders", "nosplit"):
    r = GaussianRenderer(4, W, H, (16, 16), False)
    if mode == "nocuts": r.depthCuts = False
    if mode == "noriders": r.setTuning(colour_riders=0)
    if mode == "nosplit": r.setTuning(splitter_depth_sort=0)
    model = GaussModel(p0, r.device)
    tr = GaussianTrainer(model, r, iterationCount=T.TOTAL, densify=False)
    tg = [torch.as_tensor(t, device=r.device) for t in targets]
    losses = []
    for it in range(T.STEPS):
        v = it % 3
        losses.append(float(tr.trainStep(cams[v], tg[v], viewKey=None if mode == "nokey" else v)[0]))
    d = np.abs(np.array(losses) - np.array(want_l))
    op = model.getParams()["opacity"].cpu().numpy().reshape(-1)
    dev = np.abs(op - want_p["opacity"].reshape(-1)) / np.abs(want_p["opacity"]).max()
    print(mode, "misses", tr.forwardMisses, "loss diff", d.round(7).tolist(), "opacity share", float((dev > 1e-3).mean()), flush=True)
    r.close()
