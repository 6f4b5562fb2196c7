"""Host-side mirror of Trainer/PointCloudUtil.swift (point clouds behind the loaders) -- numpy, float32.

Nothing here is on the per-iteration path; it feeds model_init.create_from_pcd once per training run."""
from __future__ import annotations

import numpy as np


def inv3x3(m):
    """PointCloudUtil.swift:13-46: adjugate / (det + 1e-10), batched."""
    m = np.asarray(m, np.float32)
    a, b, c = m[..., 0, 0], m[..., 0, 1], m[..., 0, 2]
    d, e, f = m[..., 1, 0], m[..., 1, 1], m[..., 1, 2]
    g, h, i = m[..., 2, 0], m[..., 2, 1], m[..., 2, 2]
    det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g)
    de = det + np.float32(1e-10)
    inv = np.zeros_like(m)
    inv[..., 0, 0] = (e * i - f * h) / de; inv[..., 0, 1] = -(b * i - c * h) / de; inv[..., 0, 2] = (b * f - c * e) / de
    inv[..., 1, 0] = -(d * i - f * g) / de; inv[..., 1, 1] = (a * i - c * g) / de; inv[..., 1, 2] = -(a * f - c * d) / de
    inv[..., 2, 0] = (d * h - e * g) / de; inv[..., 2, 1] = -(a * h - b * g) / de; inv[..., 2, 2] = (a * e - b * d) / de
    return inv


def getRaysFromImages(H: int, W: int, intrinsics, c2w, renderStride: int = 1):
    """PointCloudUtil.swift:47-94: per image, origin and (un-normalised) direction of every pixel's ray; pixels in
    meshgrid 'xy' order (v major, u minor)."""
    intrinsics, c2w = np.asarray(intrinsics, np.float32), np.asarray(c2w, np.float32)
    u = np.arange(0, W, renderStride, dtype=np.float32)
    v = np.arange(0, H, renderStride, dtype=np.float32)
    ug, vg = np.meshgrid(u, v, indexing="xy")
    pixels = np.stack([ug.reshape(-1), vg.reshape(-1), np.ones(ug.size, np.float32)], 0)       # [3, HW]
    inv = inv3x3(intrinsics[:, :3, :3])
    rot = c2w[:, :3, :3]
    rays_d = np.matmul(np.matmul(rot, inv), pixels[None]).transpose(0, 2, 1)                 # [B, HW, 3]
    rays_o = np.repeat(c2w[:, None, :3, 3], rays_d.shape[1], axis=1)
    return rays_o.astype(np.float32), rays_d.astype(np.float32)


class PointCloud:
    """PointCloudUtil.swift:127-191."""
    COLORS = {"R", "G", "B", "A"}

    def __init__(self, coords, channels: dict):
        self.coords = np.asarray(coords, np.float32)
        self.channels = {k: np.asarray(v, np.float32) for k, v in channels.items()}

    def preprocess(self, data, channel: str):
        return np.round(data * np.float32(255.0)) if channel in self.COLORS else data

    def select_channels(self, channel_names):
        return np.stack([self.preprocess(self.channels[n], n) for n in channel_names], axis=-1)

    def randomSample(self, numPoints: int, rng=None):
        n = self.coords.shape[0]
        if n <= numPoints:
            return self
        pick = (rng or np.random.default_rng()).permutation(n)[:numPoints]
        return PointCloud(self.coords[pick], {k: v[pick] for k, v in self.channels.items()})

    def centering(self, data, outlierSigma: float = 3.0):
        """:164-190: subtract the mean of the points from the points AND from every camera position, then drop points
        outside +-outlierSigma standard deviations (population std) on any axis."""
        center = self.coords.mean(axis=0, dtype=np.float32)
        data.c2wArray = np.array(data.c2wArray, np.float32, copy=True)
        data.c2wArray[:, :3, 3] -= center
        self.coords = self.coords - center
        std = self.coords.std(axis=0, dtype=np.float32)
        s = np.float32(outlierSigma)
        keep = np.all((self.coords > -s * std) & (self.coords < s * std), axis=1)
        self.coords = self.coords[keep]
        self.channels = {k: v[keep] for k, v in self.channels.items()}


def getPointCloudsFromTrainData(trainData) -> PointCloud:
    """PointCloudUtil.swift:95-126: back-project every fully opaque pixel (alpha == 1) along its ray by its depth."""
    if trainData.depthArray is None:
        raise ValueError("unexpected nil depth")
    Hs, Ws, intrinsics, c2ws = trainData.getCameraParams()
    H, W = int(Hs[0]), int(Ws[0])
    depths, alphas, rgbs = trainData.depthArray, trainData.alphaArray, trainData.rgbArray
    assert depths.shape == alphas.shape
    rays_o, rays_d = getRaysFromImages(H, W, intrinsics, c2ws)
    pts = rays_o + rays_d * depths.reshape(rays_o.shape[0], -1, 1)
    rgba = np.concatenate([rgbs, alphas[..., None]], axis=-1).reshape(-1, 4)
    mask = np.nonzero(alphas.reshape(-1) == 1)[0]
    sel = rgba[mask]
    return PointCloud(pts.reshape(-1, 3)[mask], dict(R=sel[:, 0], G=sel[:, 1], B=sel[:, 2], A=sel[:, 3]))
