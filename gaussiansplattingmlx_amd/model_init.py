"""Host-side mirror of the model initialisation in Trainer/GaussianModel.swift (distTopK :11-31, create_from_pcd
:87-125) over the C ABI: the kNN distances come from the HIP kernel, the rest is shaping."""
from __future__ import annotations

import numpy as np
import torch

from .renderer import GaussianRenderer, _p

C0 = 0.28209479177387814           # ShUtils.swift


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def inverse_sigmoid(x):
    return np.log(x / (1 - x))


def distTopK(renderer: GaussianRenderer, X, k: int = 3, reference_stride: bool = True) -> torch.Tensor:
    """Mean of the k smallest squared distances (self included) per point, [N] on the device.

    reference_stride=True reproduces the reference's loop exactly (GaussianModel.swift:13-18): with
    iterationSize = N/256 + 1 it runs `for i in stride(from: 0, to: iterationSize, by: 256)`, i.e. it only fills
    the chunks [i, i + 256) for i = 0, 256, ... < N/256 + 1 -- the first 256 points for any N < 65280 -- and leaves
    every other entry 0 (create_from_pcd then floors them at 1e-7).  reference_stride=False covers all points."""
    r = renderer
    X = r._t(X).reshape(-1, 3)
    N = int(X.shape[0])
    out = r._empty(N).zero_()
    if reference_stride:
        ranges = [(i, min(256, N - i)) for i in range(0, N // 256 + 1, 256) if i < N]
    else:
        ranges = [(0, N)]
    for b, c in ranges:
        r._check(r.lib.gs_dist_topk(r.ctx, N, int(k), int(b), int(c), _p(X), _p(out)))
    return out


def create_from_pcd(renderer: GaussianRenderer, pcd, sh_degree: int = 3, reference_stride: bool = True) -> dict:
    """GaussModel.create_from_pcd: the six raw parameter tensors (device f32) in the trainer's shapes --
    xyz [N,3], features_dc [N,1,3], features_rest [N,K-1,3], scales [N,3], rotation [N,4], opacity [N,1]."""
    r = renderer
    points = np.asarray(pcd.coords, np.float32)
    colors = (pcd.select_channels(["R", "G", "B"]) / np.float32(255.0)).astype(np.float32)
    N, K = points.shape[0], (sh_degree + 1) ** 2
    fused_color = RGB2SH(colors).astype(np.float32)
    dist2 = torch.clamp_min(distTopK(r, points, 3, reference_stride), 1e-7)
    scales = torch.log(torch.sqrt(dist2)).reshape(N, 1).repeat(1, 3)
    rots = torch.zeros(N, 4, device=r.device)
    rots[:, 0] = 1.0
    opac = float(inverse_sigmoid(np.float32(0.1)))
    return dict(xyz=r._t(points), features_dc=r._t(fused_color.reshape(N, 1, 3)),
                features_rest=torch.zeros(N, K - 1, 3, device=r.device), scales=scales.contiguous(), rotation=rots,
                opacity=torch.full((N, 1), opac, device=r.device))
