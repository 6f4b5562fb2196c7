"""Diagnostic: pixel work of the blend backward if it culled per 8x8 quadrant instead of per 16x8 half.
For every binned (tile, Gaussian) pair: how many halves / quadrants of the 16x16 block can the splat reach
(minimum of the quadratic form over the rectangle <= 58)?  Early termination is ignored."""
import sys, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
from oracle.oracle import Oracle
name = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
params, cams, (W, H) = make_config(name, n_views=1)
r = GaussianRenderer(4, W, H)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
r.renderForward(tp, cams[0])
st = r.stats(); M = st["M"]; gw = (W + 15) // 16; T = gw * ((H + 15) // 16)
idx = torch.empty(M, dtype=torch.int32, device=r.device); rng_ = torch.empty(T, 2, dtype=torch.int32, device=r.device); cnt = torch.empty(T, dtype=torch.int32, device=r.device)
r._check(r.lib.gs_tile_bin_export(r.ctx, C.c_void_p(idx.data_ptr()), C.c_void_p(rng_.data_ptr()), C.c_void_p(cnt.data_ptr())))
o = Oracle(np.float32); c = cams[0].as_dict()
op, sc, rt = o.activations_forward(params["opacity"], params["scales"], params["rotation"])
shs = np.concatenate([params["features_dc"], params["features_rest"]], 1)
pr = o.projection_forward(sc, rt, params["xyz"], shs, c["camCenter"], c["view"], c["proj"], c["fovX"], c["fovY"], c["focalX"], c["focalY"], W, H, 4)
m2 = torch.as_tensor(pr["means2d"], device=r.device); con = torch.as_tensor(pr["conic"], device=r.device).reshape(-1, 4)
tile_of = torch.repeat_interleave(torch.arange(T, device=r.device), cnt.long())
g = idx.long()
tx, ty = (tile_of % gw).float() * 16, (tile_of // gw).float() * 16
mx, my = m2[g, 0], m2[g, 1]
c00, c01, c10, c11 = con[g, 0], con[g, 1], con[g, 2], con[g, 3]
b = 0.5 * (c01 + c10)
pd = (c00 > 0) & (c11 > 0) & (c00 * c11 > b * b)
def reach(x0, x1, y0, y1, thr=58.0):
    X0, X1, Y0, Y1 = tx + x0 - mx, tx + x1 - mx, ty + y0 - my, ty + y1 - my
    inside = (X0 <= 0) & (X1 >= 0) & (Y0 <= 0) & (Y1 >= 0)
    def ex(X): dy = torch.minimum(torch.maximum(-b / c11 * X, Y0), Y1); return c00 * X * X + 2 * b * X * dy + c11 * dy * dy
    def ey(Y): dx = torch.minimum(torch.maximum(-b / c00 * Y, X0), X1); return c00 * dx * dx + 2 * b * dx * Y + c11 * Y * Y
    q = torch.minimum(torch.minimum(ex(X0), ex(X1)), torch.minimum(ey(Y0), ey(Y1)))
    q = torch.where(inside, torch.zeros_like(q), q)
    return ~(pd & (q > thr))
halves = reach(0, 15, 0, 7).float() + reach(0, 15, 8, 15).float()
quads = sum(reach(x, x + 7, y, y + 7).float() for x in (0, 8) for y in (0, 8))
print(name, "pairs", M)
print("pixel work per pair, in 64-pixel units: no culling 4.00, halves %.3f, quadrants %.3f  (quadrants / halves = %.3f)" % (
    float(2 * halves.mean()), float(quads.mean()), float(quads.sum() / (2 * halves.sum()))))
print("pairs reaching no half: %.4f ; histogram of quadrants reached:" % float((halves == 0).float().mean()),
      [round(float((quads == k).float().mean()), 4) for k in range(5)])
for thr in (58.0, 48.0, 40.0, 32.0, 24.0):
    hv = reach(0, 15, 0, 7, thr).float() + reach(0, 15, 8, 15, thr).float()
    qd = sum(reach(x, x + 7, y, y + 7, thr).float() for x in (0, 8) for y in (0, 8))
    print("threshold %.0f: halves %.3f quadrants %.3f" % (thr, float(2 * hv.mean()), float(qd.mean())))
