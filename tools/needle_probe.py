"""Diagnostic: on a scene of needles, which entries do trimmed rects drop that the reference-lists forward blends visibly?"""
import sys, numpy as np, torch
sys.path.insert(0, '.')
import importlib.util
spec = importlib.util.spec_from_file_location("tp", "tests/test_gpu_parity.py"); tp = importlib.util.module_from_spec(spec); spec.loader.exec_module(tp)
from oracle.oracle import Oracle
o = Oracle(np.float32)
seed, scale, aniso = 1, 0.05, 30.0
W, H, N = 400, 304, 2500
p, cam = tp._scene(400 + seed, N, W, H, spread=1.0, scale=scale)
rng = np.random.default_rng(seed)
p["scales"][:, 0] += np.float32(np.log(aniso)); p["scales"][:, 1:] -= np.float32(0.5 * np.log(aniso) * rng.uniform(0, 1, (N, 2))); p["features_rest"] *= 0.05
fw = o.render_forward(p, cam.as_dict(), W, H, 16, 16, 4)
pr = fw["proj"]
r = tp._renderer(W, H)
t = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
imgs, lists = {}, {}
for trim in (0, 1, 2):
    r.setTuning(trim_rects=trim)
    res = r.renderForward(t, cam)
    imgs[trim] = res.render.cpu().numpy().reshape(H, W, 3)
    lists[trim] = tp._fused_lists(r, W, H)
m2d = np.asarray(pr["means2d"], np.float64); con = np.asarray(pr["conic"], np.float64).reshape(-1, 4); cov = np.asarray(pr["cov2d"], np.float64).reshape(-1, 4)
for trim in (1, 2):
    d = np.abs(imgs[trim] - imgs[0]).max(axis=2)
    y, x = np.unravel_index(d.argmax(), d.shape)
    print("trim", trim, "max diff", d.max(), "at pixel", x, y, "pixels beyond 2e-6:", int((d > 2e-6).sum()))
    tile = (y // 16) * ((W + 15) // 16) + x // 16
    M0, idx0, rng0, cnt0 = lists[0]; M1, idx1, rng1, cnt1 = lists[trim]
    a = idx0[rng0[tile, 0]:rng0[tile, 1]]; b = idx1[rng1[tile, 0]:rng1[tile, 1]]
    out = np.setdiff1d(a, b)
    print("  tile", tile, "list", a.size, "->", b.size, "dropped", out.size)
    rows = []
    for g in out:
        dx, dy = x - m2d[g, 0], y - m2d[g, 1]
        q = con[g, 0] * dx * dx + (con[g, 1] + con[g, 2]) * dx * dy + con[g, 3] * dy * dy
        # min q over the tile's pixels
        px = np.arange(16 * (x // 16), min(16 * (x // 16) + 16, W)); py = np.arange(16 * (y // 16), min(16 * (y // 16) + 16, H))
        DX, DY = np.meshgrid(px - m2d[g, 0], py - m2d[g, 1])
        Q = con[g, 0] * DX * DX + (con[g, 1] + con[g, 2]) * DX * DY + con[g, 3] * DY * DY
        rows.append((q, Q.min(), g))
    rows.sort()
    for q, qmin, g in rows[:6]:
        print(f"   dropped Gaussian {g}: q at the pixel {q:.3f}, min q over the tile {qmin:.3f}, mean2d {m2d[g]}, cov2d {cov[g]}, conic {con[g]}, radius {float(pr['radii'][g])}")
