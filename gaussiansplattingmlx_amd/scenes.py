"""Synthetic workloads of BASELINE.json's configs.

There is no network, so "Lego" is the public NeRF-synthetic camera geometry
(camera_angle_x = 0.6911112, radius 4.0311, upper hemisphere, look-at origin)
over seeded synthetic Gaussians.  seed = 20260313 + config index.
"""
from __future__ import annotations

import numpy as np

from .camera import Camera, look_at_c2w

LEGO_ANGLE_X = 0.6911112
LEGO_RADIUS = 4.0311
C0 = 0.28209479177387814

CONFIGS = {
    # name: (config index, N, W, H, kind)
    "c1_10k_400": (0, 10_000, 400, 400, "random_init"),
    "c2_100k_800": (1, 100_000, 800, 800, "trained_like"),
    "c3_300k_800": (2, 300_000, 800, 800, "trained_like"),
    "c5_garden_2m": (4, 2_000_000, 1237, 822, "garden"),
    # not a BASELINE config: c3's scene, which bench.py then GROWS with the trainer's own schedule before it times anything
    # (GROW_ITERATIONS untimed iterations from iteration 450; the reference's densification -- GaussianTrainer.swift:297-300:
    # every 100 iterations from 500, maxGaussians = 1_000_000 -- takes it to the cap by iteration ~1500 and keeps it there):
    # the regime the reference's own 30 000-iteration schedule spends most of its time in.  What the scene looks like then
    # (tools/grown_stats.py): ~16 M pairs of which ~6.7 M block-entries are traversed, mean nContrib 2300 against 375 --
    # training drives most opacities low (median 0.11), which no closed-form generator of ours reproduced.
    "c3_grown_1m": (2, 300_000, 800, 800, "trained_like"),
}


GROW_ITERATIONS = 1150      # c3_grown_1m: iterations 450 .. 1600 of the trainer's own schedule, untimed


def lego_cameras(n_views: int, W: int, H: int, seed: int):
    rng = np.random.default_rng(seed)
    focal = 0.5 * W / np.tan(0.5 * LEGO_ANGLE_X)
    cams = []
    for _ in range(n_views):
        th = rng.uniform(0.0, 2 * np.pi)
        ph = rng.uniform(np.deg2rad(10.0), np.deg2rad(80.0))   # elevation
        eye = LEGO_RADIUS * np.array([np.cos(ph) * np.cos(th), np.cos(ph) * np.sin(th), np.sin(ph)])
        cams.append(Camera(W, H, focal, focal, look_at_c2w(eye)))
    return cams


def garden_cameras(n_views: int, W: int, H: int, seed: int):
    rng = np.random.default_rng(seed)
    cams = []
    for _ in range(n_views):
        th = rng.uniform(0.0, 2 * np.pi)
        eye = np.array([4.0 * np.cos(th), 4.0 * np.sin(th), 1.5])
        cams.append(Camera(W, H, 961.2, 961.2, look_at_c2w(eye, target=(0.0, 0.0, 0.5))))
    return cams


def _knn_log_scale(xyz: np.ndarray) -> np.ndarray:
    """log sqrt(mean of 3 smallest squared NN distances incl. self, floor 1e-7)
    (GaussianModel.swift:105-110), true kNN for every point."""
    from scipy.spatial import cKDTree
    d, _ = cKDTree(xyz).query(xyz, k=3)
    d2 = np.maximum((d ** 2).mean(axis=1), 1e-7)
    return np.log(np.sqrt(d2))


def make_gaussians(N: int, kind: str, seed: int, sh_degree: int = 4) -> dict:
    rng = np.random.default_rng(seed)
    K = (sh_degree + 1) ** 2
    if kind == "random_init":
        xyz = rng.uniform(-1.3, 1.3, (N, 3))
        rgb = rng.uniform(0.0, 1.0, (N, 3))
        f_dc = ((rgb - 0.5) / C0)[:, None, :]
        f_rest = np.zeros((N, K - 1, 3))
        ls = _knn_log_scale(xyz)
        scales = np.repeat(ls[:, None], 3, axis=1)
        rot = np.zeros((N, 4)); rot[:, 0] = 1.0
        opacity = np.full((N,), np.log(0.1 / 0.9))
    else:
        if kind == "garden":
            lo, hi = np.array([-6.0, -6.0, -1.0]), np.array([6.0, 6.0, 3.0])
            ls_mu, ls_sd = np.log(0.02), 0.8
        else:
            lo, hi = np.array([-1.3] * 3), np.array([1.3] * 3)
            ls_mu, ls_sd = np.log(0.012), 0.6
        n_shell = int(0.7 * N)
        # thin shells / boxes inside the bbox: points on the surfaces of a few nested boxes + spheres
        u = rng.uniform(-1.0, 1.0, (n_shell, 3))
        which = rng.integers(0, 4, n_shell)
        r = np.array([0.35, 0.6, 0.8, 0.95])[which]
        sph = which % 2 == 0
        nrm = np.linalg.norm(u, axis=1, keepdims=True) + 1e-9
        on_sphere = u / nrm
        ax = rng.integers(0, 3, n_shell)
        on_box = u.copy()
        on_box[np.arange(n_shell), ax] = np.sign(on_box[np.arange(n_shell), ax] + 1e-12)
        shell = np.where(sph[:, None], on_sphere, on_box) * r[:, None]
        shell += rng.normal(0.0, 0.004, shell.shape)
        ctr, half = 0.5 * (lo + hi), 0.5 * (hi - lo)
        shell = ctr + shell * half
        uni = rng.uniform(lo, hi, (N - n_shell, 3))
        xyz = np.concatenate([shell, uni], axis=0)
        xyz = xyz[rng.permutation(N)]
        scales = rng.normal(ls_mu, ls_sd, (N, 3))
        rot = rng.normal(0.0, 1.0, (N, 4))
        rot /= np.linalg.norm(rot, axis=1, keepdims=True)
        op = np.clip(rng.beta(0.5, 0.5, N), 0.01, 0.99)
        opacity = np.log(op / (1 - op))
        f_dc = rng.normal(0.0, 1.0, (N, 1, 3))
        f_rest = rng.normal(0.0, 0.05, (N, K - 1, 3))
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    return dict(xyz=f32(xyz), features_dc=f32(f_dc), features_rest=f32(f_rest), scales=f32(scales),
                rotation=f32(rot), opacity=f32(opacity))


def perturb(params: dict, seed: int, amount: float = 0.05) -> dict:
    """A perturbed copy of the scene; its render is the training target, so gradients are non-trivial."""
    rng = np.random.default_rng(seed)
    out = {}
    for k, v in params.items():
        sd = amount * (float(np.std(v)) + 1e-3)
        out[k] = (v + rng.normal(0.0, sd, v.shape)).astype(np.float32)
    return out


def make_config(name: str, n_views: int = 8, sh_degree: int = 4):
    idx, N, W, H, kind = CONFIGS[name]
    seed = 20260313 + idx
    params = make_gaussians(N, kind, seed, sh_degree)
    cams = (garden_cameras if kind == "garden" else lego_cameras)(n_views, W, H, seed + 1000)
    return params, cams, (W, H)
