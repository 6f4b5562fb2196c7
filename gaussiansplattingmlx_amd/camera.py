"""Host-side camera, mirroring the reference's `Camera`
(Trainer/CameraUtil.swift:5-102, Trainer/simd+ext.swift:45-66).

view  = (c2w^-1)^T stored row-major, so p_view = [p, 1] @ view (translation in row 3)
proj  = rows (2n/(r-l),0,0,0), (0,2n/(t-b),0,0), (0,0,f/(f-n),1), (0,0,-nf/(f-n),0)
FoV   = 2 atan(pixels / (2 focal)) evaluated in f32; matrices built in f64, cast to f32.
The principal point is ignored, as in the reference (CameraUtil.swift:26-27).
"""
from __future__ import annotations

import numpy as np


def simd_to_row_major(columns) -> np.ndarray:
    """simd_float4x4 / simd_double4x4 `toMLXArray()` (Trainer/simd+ext.swift:45-55): simd stores COLUMNS
    (`m[c][r]`, `m[c, r]` is element (r, c)); the array handed to the kernels is row-major with a[r][c] = m[c][r].
    With that layout a simd `v * m` (row vector times matrix) is `v @ a`, which is why the kernels compute
    p_view = [p, 1] @ view with the translation in row 3."""
    cols = np.asarray(columns)
    return np.ascontiguousarray(cols.T)


def focal2fov(focal: float, pixels: float) -> np.float32:
    return np.float32(2.0) * np.arctan(np.float32(pixels) / (np.float32(2.0) * np.float32(focal)))


def fov2focal(fov: float, pixels: float) -> float:
    return pixels / (2.0 * np.tan(fov / 2.0))


def getProjectionMatrix(znear: float, zfar: float, fovX: float, fovY: float) -> np.ndarray:
    """Row-major array the reference hands to its kernels (P.transpose.toMLXArray())."""
    tanHalfY, tanHalfX = np.tan(fovY / 2.0), np.tan(fovX / 2.0)
    top, right = tanHalfY * znear, tanHalfX * znear
    bottom, left = -top, -right
    P = np.zeros((4, 4), np.float64)
    P[0, 0] = 2 * znear / (right - left)
    P[1, 1] = 2 * znear / (top - bottom)
    P[2, 0] = (right + left) / (right - left)
    P[2, 1] = (top + bottom) / (top - bottom)
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = 1.0
    P[3, 2] = -znear * zfar / (zfar - znear)
    return P


class Camera:
    def __init__(self, width: int, height: int, focalX: float, focalY: float, c2w, znear: float = 0.1,
                 zfar: float = 100.0):
        c2w = np.asarray(c2w, dtype=np.float64).reshape(4, 4)
        self.imageWidth, self.imageHeight = int(width), int(height)
        self.focalX, self.focalY = np.float32(focalX), np.float32(focalY)
        self.FoVx = focal2fov(self.focalX, float(width))
        self.FoVy = focal2fov(self.focalY, float(height))
        # simd_double4x4.fromMlxArray(c2w).inverse.transpose.toMLXArray() (CameraUtil.swift:30, :34): fromMlxArray /
        # toMLXArray are the row-major <-> column-storage pair above, so in array terms this is inv(c2w)^T
        self.worldViewTransform = np.ascontiguousarray(np.linalg.inv(c2w).T.astype(np.float32))
        self.projectionMatrix = np.ascontiguousarray(
            getProjectionMatrix(znear, zfar, float(self.FoVx), float(self.FoVy)).astype(np.float32))
        self.cameraCenter = c2w[:3, 3].copy()

    def as_dict(self):
        return dict(view=self.worldViewTransform, proj=self.projectionMatrix, fovX=float(self.FoVx),
                    fovY=float(self.FoVy), focalX=float(self.focalX), focalY=float(self.focalY),
                    camCenter=self.cameraCenter.astype(np.float32))


def look_at_c2w(eye, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)) -> np.ndarray:
    """OpenCV-convention camera-to-world (x right, y down, z forward)."""
    eye = np.asarray(eye, np.float64)
    fwd = np.asarray(target, np.float64) - eye
    fwd /= np.linalg.norm(fwd)
    upv = np.asarray(up, np.float64)
    right = np.cross(fwd, upv)
    if np.linalg.norm(right) < 1e-8:
        right = np.cross(fwd, np.array([0.0, 1.0, 0.0]))
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    c2w = np.eye(4)
    c2w[:3, 0], c2w[:3, 1], c2w[:3, 2], c2w[:3, 3] = right, down, fwd, eye
    return c2w
