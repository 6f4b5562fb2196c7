"""Diagnostic: what fraction of the binned (tile, Gaussian) pairs cannot reach any pixel of their 16x16 tile
(minimum of the quadratic form over the tile rectangle > 58, i.e. exponent < 2^-41)?"""
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
# projected quantities from the oracle (same arithmetic)
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
X0, X1, Y0, Y1 = tx - mx, tx + 15 - mx, ty - my, ty + 15 - my
inside = (X0 <= 0) & (X1 >= 0) & (Y0 <= 0) & (Y1 >= 0)
def ex(X): dy = torch.minimum(torch.maximum(-b / c11 * X, Y0), Y1); return c00 * X * X + 2 * b * X * dy + c11 * dy * dy
def ey(Y): dx = torch.minimum(torch.maximum(-b / c00 * Y, X0), X1); return c00 * dx * dx + 2 * b * dx * Y + c11 * Y * Y
q = torch.minimum(torch.minimum(ex(X0), ex(X1)), torch.minimum(ey(Y0), ey(Y1)))
q = torch.where(inside, torch.zeros_like(q), q)
pd = (c00 > 0) & (c11 > 0) & (c00 * c11 > b * b)
far = pd & (q > 58.0)
print(name, "pairs", M, "unreachable fraction %.4f" % float(far.float().mean()))
# which rect sizes do the pairs (and the unreachable ones) come from?
rect_w = torch.as_tensor(pr["rectMax"], device=r.device).reshape(-1, 2)
rmin = torch.as_tensor(pr["rectMin"], device=r.device).reshape(-1, 2)
tx0 = torch.floor(rmin[:, 0] / 16).clamp(0, gw); tx1 = (torch.floor(rect_w[:, 0] / 16) + 1).clamp(0, gw)
gh = (H + 15) // 16
ty0 = torch.floor(rmin[:, 1] / 16).clamp(0, gh); ty1 = (torch.floor(rect_w[:, 1] / 16) + 1).clamp(0, gh)
area = ((tx1 - tx0) * (ty1 - ty0))[g]
for lim in (16, 32, 64, 128, 256, 10 ** 9):
    sel = area <= lim
    print("rect area <= %d: %.3f of pairs, %.3f of the unreachable pairs" % (lim, float(sel.float().mean()), float((far & sel).float().sum() / far.float().sum())))
