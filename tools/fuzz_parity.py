"""Exploratory: adversarial small scenes (camera inside the cloud, Gaussians straddling the z >= 0.2 visibility plane,
screen-filling and sub-pixel scales, near-zero quaternions, saturated opacities, tiles larger than the image) through the
fused forward / backward against the oracle.  Prints every case that leaves the test-suite bars.
2000 seeds (20000-21999) on MI355X: 5 flagged, none a defect -- three are the documented deviation (a Gaussian within
~1e-3 of the camera plane that no pixel blended: the reference's J^T 0 is 0 * inf = NaN, the fused backward returns the
exact 0), one is the rotation gradient of a quaternion of norm 1e-9 (5 % apart: the normalisation's
Jacobian is ~1e8 there), one a single Gaussian whose colour gradient is a sum of +-1 terms cancelling to 3e-4 (1.01e-3
apart in one run, 0.99e-3 in the next: float atomics).
800 seeds (40000-40799, round 3's kernels): 3 flagged -- two of the J^T 0 kind, and seed 40410: ONE Gaussian whose mean lies 22 px
outside a 33x27 image; every gradient is ~1e-9 (against cotangents of order 1) and 8-10 % off, at 16x16 and at 48x48 tiles alike:
what the oracle sums there is the tail beyond q = 40 that the staging cull drops by design (weights below 2^-29;
tools/fuzz_one.py prints a case in full).
1000 seeds (50000-50999, round 4's kernels: per-XCD forward / backward queues, arena parts): 6 flagged, none a defect -- two of the
J^T 0 kind (50088, 50724); three single Gaussians far outside the image whose every gradient is 1e-26 .. 1e-45 against
cotangents of order 1 (50144, 50568, 50690: the fused path returns the exact 0 for the tail the staging cull drops); and 50742,
the 40410 kind (one Gaussian, gradients ~1e-5, 0.2-0.5 % off).  `python tools/fuzz_parity.py detail <seed> ...` prints such cases.
1000 seeds (80000-80999, the round's last kernels: these images are small, so every forward is the four-waves-per-quadrant
kernel): 4 flagged, three of the J^T 0 kind and one single Gaussian with gradients in the denormal range (80120).
Round 5: FUZZ_TILES=24,40,100,200,7,33 puts every case on tile sizes that are not multiples of 16 (block lists: image and gradients
against the oracle at that tile size; M and nContrib refer to block lists there and are not compared).
Round 6: at 16 x 16 tiles the fused path bins on rects cut to what the blend can see (GS_TUNE_TRIM_RECTS, default): M <= the oracle's,
nContrib compared by the Gaussian it points at (trimmed_ncontrib_mismatches); FUZZ_TRIM=0 runs the reference's lists.
usage: python tools/fuzz_parity.py [n_cases] [first_seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
from oracle.oracle import Oracle

_oracle = None


def trimmed_ncontrib_mismatches(r, fw, W, H, last):
    """Pixels whose nContrib, a position in the fused path's TRIMMED list, does not mean what the oracle's position means: a pixel
    that terminated (T < 1e-4) stopped at another Gaussian, or a pixel live at the end did not go through its whole list; plus the
    tiles whose list is not the oracle's with entries left out, order kept (tests/test_gpu_parity.py, _ncontrib_match)."""
    import ctypes as C
    M, T = r.stats()["M"], ((W + 15) // 16) * ((H + 15) // 16)
    idx = torch.zeros(max(M, 1), dtype=torch.int32, device=r.device)
    rng_ = torch.zeros(T, 2, dtype=torch.int32, device=r.device)
    cnt = torch.zeros(T, dtype=torch.int32, device=r.device)
    r._check(r.lib.gs_tile_bin_export(r.ctx, C.c_void_p(idx.data_ptr()), C.c_void_p(rng_.data_ptr()), C.c_void_p(cnt.data_ptr())))
    idx, rng_, cnt = (np.concatenate([idx.cpu().numpy()[:M].astype(np.int64), [-1]]), rng_.cpu().numpy().astype(np.int64),
                      cnt.cpu().numpy().astype(np.int64))
    bn = fw["bin"]
    o_idx = np.concatenate([np.asarray(bn.sortedIdx).astype(np.int64), [-1]])
    o_rng = np.asarray(bn.tileRanges).astype(np.int64).reshape(-1, 2)
    ys, xs = np.divmod(np.arange(W * H), W)
    tile = (ys // 16) * ((W + 15) // 16) + xs // 16
    want = fw["last"].reshape(-1).astype(np.int64)
    Tr = 1.0 - np.asarray(fw["alpha"], np.float64).reshape(-1)
    gid = np.where(last > 0, idx[np.clip(rng_[tile, 0] + last - 1, 0, M)], -1)
    wid = np.where(want > 0, o_idx[np.clip(o_rng[tile, 0] + want - 1, 0, o_idx.size - 1)], -1)
    bad = int((gid != wid)[Tr < 0.9e-4].sum()) + int((last != cnt[tile])[Tr > 1.1e-4].sum())
    for t in np.nonzero(np.asarray(bn.tileCounts) > 0)[0]:
        a, b = idx[rng_[t, 0]:rng_[t, 1]], o_idx[o_rng[t, 0]:o_rng[t, 1]]
        keep = np.isin(b, a)
        bad += 0 if (int(keep.sum()) == a.size and np.array_equal(b[keep], a)) else 1000
    return bad


def run_case(s, detail=False, tuning=None):
    """One adversarial case; returns the list of bars it leaves (empty = fine)."""
    global _oracle
    if _oracle is None:
        _oracle = Oracle(np.float32)
    o = _oracle
    rng = np.random.default_rng(77000 + s)
    W, H = int(rng.integers(9, 130)), int(rng.integers(9, 130))
    N = int(rng.choice([1, 3, 64, 65, 200, 900, 4000]))
    K = int(rng.choice([1, 4, 9, 16, 25])); deg = {1: 0, 4: 1, 9: 2, 16: 3, 25: 4}[K]
    tile = tuple(int(rng.choice([16, 32, 48, 100])) for _ in range(2)) if rng.random() < 0.3 else (16, 16)
    if os.environ.get("FUZZ_TILES"):        # e.g. FUZZ_TILES=24,40,100,200,7: every case on tile sizes drawn from this list (block lists)
        ch = [int(t) for t in os.environ["FUZZ_TILES"].split(",")]
        tile = (int(ch[(s * 7 + 1) % len(ch)]), int(ch[(s * 3 + 2) % len(ch)]))
    white = bool(rng.integers(0, 2))
    mode = s % 6
    eye = np.array([2.2, -2.6, 1.7]) * (rng.uniform(0.05, 0.4) if mode == 0 else 1.0)      # mode 0: camera inside the cloud
    fmul = float(rng.choice([0.25, 0.9, 3.0]))                                               # wide, normal, long lens
    cam = Camera(W, H, fmul * W, fmul * 1.02 * W, look_at_c2w(list(eye)))
    xyz = rng.uniform(-1, 1, (N, 3))
    scales = rng.normal(np.log(0.05), 0.5, (N, 3))
    rot = rng.normal(0, 1, (N, 4))
    opac = rng.normal(0.3, 1.5, N)
    if mode == 1: scales += np.log(40.0) * (rng.random((N, 1)) < 0.2)                        # screen-filling splats
    if mode == 2: scales -= np.log(200.0)                                                    # sub-pixel splats
    if mode == 3: rot[rng.random(N) < 0.3] *= 1e-9                                           # near-zero quaternions
    if mode == 4: opac = np.where(rng.random(N) < 0.5, 20.0, -20.0)                          # saturated opacities
    if mode == 5:                                                                            # points on the z = 0.2 plane
        c = cam.as_dict(); V = np.asarray(c["view"], np.float64).reshape(4, 4)
        # move a third of the points so that their view-space z is 0.2 +- 1e-3
        pv = np.c_[xyz, np.ones(N)] @ V
        fwd = V[:3, 2] / (np.linalg.norm(V[:3, 2]) ** 2)
        sel = rng.random(N) < 0.33
        xyz[sel] += np.outer((0.2 + rng.uniform(-1e-3, 1e-3, sel.sum()) - pv[sel, 2]), fwd)
    p = dict(xyz=xyz, features_dc=rng.normal(0, 1, (N, 1, 3)), features_rest=rng.normal(0, 0.05, (N, K - 1, 3)),
             scales=scales, rotation=rot, opacity=opac)
    p = {k: np.ascontiguousarray(v, np.float32) for k, v in p.items()}
    c = cam.as_dict()
    try:
        fw = o.render_forward(p, c, W, H, tile[1], tile[0], deg, white)
        r = GaussianRenderer(deg, W, H, (tile[1], tile[0]), white)
        if tuning:
            r.setTuning(**tuning)
        if os.environ.get("FUZZ_TRIM"):
            r.setTuning(trim_rects=int(os.environ["FUZZ_TRIM"]))
        tp = {k: torch.as_tensor(v, device=r.device) for k, v in p.items()}
        res = r.renderForward(tp, cam)
        img = res.render.cpu().numpy().reshape(-1, 3)
        msg = []
        # (a tile size that is not a multiple of 16: the fused path works on block lists -- its M counts (Gaussian, block) pairs and
        # its nContrib positions in a block's list; the image and the gradients are what is held to the oracle there)
        block_lists = r.blockLists
        # (16 x 16 tiles under GS_TUNE_TRIM_RECTS, the default: lists without the entries no pixel of the tile can see -- never more
        # pairs than the oracle's, and nContrib compared by what the position means: trimmed_ncontrib_mismatches)
        trimmed = r.getTuning("trim_rects") != 0 and tuple(tile) == (16, 16)
        if not block_lists and (r.stats()["M"] > fw["bin"].M if trimmed else r.stats()["M"] != fw["bin"].M):
            msg.append(f"M {r.stats()['M']} != {fw['bin'].M}")
        fin = np.isfinite(fw["color"]).all(1) & np.isfinite(img).all(1)
        if (np.isfinite(fw["color"]).all(1) != np.isfinite(img).all(1)).any(): msg.append("finite masks differ")
        d = np.abs(img[fin] - fw["color"][fin]).max() if fin.any() else 0.0
        scale = max(1.0, float(np.abs(fw["color"][fin]).max())) if fin.any() else 1.0
        if d > 1e-4 * scale: msg.append(f"rgb {d:.3g} (max colour {scale:.3g})")
        last = r.lastContrib().cpu().numpy().reshape(-1).astype(np.int64)
        nb = trimmed_ncontrib_mismatches(r, fw, W, H, last) if trimmed else int((last != fw["last"].astype(np.int64)).sum())
        if nb > 2 and not block_lists: msg.append(f"nContrib differs on {nb} px")
        cot = rng.normal(0, 1, (H * W, 3)).astype(np.float32)
        z = np.zeros(W * H, np.float32)
        cd, ca = z, z
        if s % 2:                                                                            # depth and alpha cotangents too
            cd = rng.normal(0, 1, H * W).astype(np.float32); ca = rng.normal(0, 1, H * W).astype(np.float32)
        want = o.render_backward(p, c, W, H, tile[1], tile[0], deg, fw, cot, cd, ca, white)
        dev = lambda a: torch.as_tensor(a, device=r.device)
        got = r.renderBackward(dev(cot), dev(cd) if s % 2 else None, dev(ca) if s % 2 else None)
        for k in ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity"):
            g = got[k].cpu().numpy().astype(np.float64); w_ = want[k].reshape(g.shape).astype(np.float64)
            fm = np.isfinite(w_) & np.isfinite(g)
            if (np.isfinite(w_) != np.isfinite(g)).any(): msg.append(f"{k}: finite masks differ ({int((~np.isfinite(w_)).sum())} vs {int((~np.isfinite(g)).sum())} non-finite)")
            if fm.any() and np.abs(w_[fm]).max() > 0:
                rel = np.abs(g[fm] - w_[fm]).max() / np.abs(w_[fm]).max()
                if rel > 1e-3: msg.append(f"{k}: rel {rel:.3g}")
                if detail:
                    i = int(np.abs(np.where(fm, g - w_, 0)).argmax()); gi = i // max(1, int(np.prod(g.shape[1:])))
                    print(f"  {k}: max|want| {np.abs(w_[fm]).max():.3g} max|got| {np.abs(g[fm]).max():.3g} worst at Gaussian {gi}: "
                          f"want {w_.reshape(-1)[i]:.6g} got {g.reshape(-1)[i]:.6g}; rows with any nonzero want {int((np.abs(w_).reshape(len(w_), -1).max(1) > 0).sum())} "
                          f"got {int((np.abs(g).reshape(len(g), -1).max(1) > 0).sum())}")
        if detail:
            st = r.stats()
            print(f"  W {W} H {H} tile {tile} N {N} K {K} mode {mode} white {white} fmul {fmul} M {st['M']} oracle M {fw['bin'].M} "
                  f"nContrib max {int(last.max())} px with nContrib>0 {int((last > 0).sum())} rgb diff {d:.3g}")
        r.close()
    except Exception as e:      # noqa
        msg = [f"EXCEPTION {type(e).__name__}: {e}"]
    return msg


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "detail":          # python tools/fuzz_parity.py detail <seed> [<seed> ...]
        for s in sys.argv[2:]:
            print(f"seed {s}:"); print("  ", run_case(int(s), detail=True))
        sys.exit(0)
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    bad = 0
    for s in range(seed0, seed0 + n_cases):
        msg = run_case(s)
        if msg:
            bad += 1
            print(f"seed {s}: " + "; ".join(msg), flush=True)
    print(f"{n_cases} cases, {bad} outside the bars")
