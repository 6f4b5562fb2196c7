"""Host-side mirror of the reference's dataset loaders (Data/ColmapDataLoader.swift, Data/NerfStudioDataLoader.swift,
Data/BlenderDataLoader.swift): file formats, pose / intrinsics conventions, the TrainData container.

What is mirrored exactly: the binary / JSON / PLY parsing, quaternion -> rotation, world-to-camera inversion, the
OpenGL -> OpenCV flip (rows 1-2 of w2c negated), intrinsics scaling by resizeFactor, white-background compositing,
the tile size rule (W/4, H/4).  What is not: image decoding and resampling (the reference goes through UIKit /
CoreGraphics; here PIL, bilinear) and the downloads / unzipping of the demo sets (no network: loaders take paths)."""
from __future__ import annotations

import json
import os
import struct
from collections import namedtuple
from dataclasses import dataclass

import numpy as np

from .pointcloud import PointCloud, getPointCloudsFromTrainData

TILE_SIZE_H_W = namedtuple("TILE_SIZE_H_W", ["w", "h"])


@dataclass
class TrainData:
    Hs: np.ndarray
    Ws: np.ndarray
    intrinsicArray: np.ndarray
    c2wArray: np.ndarray
    rgbArray: np.ndarray
    alphaArray: np.ndarray
    depthArray: np.ndarray | None = None

    def getCameraParams(self):
        return self.Hs, self.Ws, self.intrinsicArray, self.c2wArray


# ---- images ---------------------------------------------------------------------------------------------------
def _load_image(path, scale: float, mode: str):
    from PIL import Image
    img = Image.open(path).convert(mode)
    if scale != 1.0:
        img = img.resize((int(img.size[0] * scale), int(img.size[1] * scale)), Image.BILINEAR)
    return np.asarray(img, np.float32) / np.float32(255.0)


def readImageRGBA(path, resizeFactor: float = 1.0):
    """ColmapDataLoader.readImage / NerfStudio readImage: rgb [H,W,3], alpha [H,W], H, W (floats in [0,1])."""
    rgba = _load_image(path, resizeFactor, "RGBA")
    return rgba[..., :3], rgba[..., 3], float(rgba.shape[0]), float(rgba.shape[1])


def _stack_frames(frames, whiteBackground: bool):
    rgbs = np.stack([f[0] for f in frames]); alphas = np.stack([f[1] for f in frames])
    if whiteBackground:
        rgbs = alphas[..., None] * rgbs + (1 - alphas)[..., None]
    Hs = np.array([f[2] for f in frames], np.float32); Ws = np.array([f[3] for f in frames], np.float32)
    return Hs, Ws, rgbs.astype(np.float32), alphas.astype(np.float32)


# ---- COLMAP (ColmapDataLoader.swift:165-500) ---------------------------------------------------------------------
ColmapCamera = namedtuple("ColmapCamera", "id width height fx fy cx cy k1 k2 p1 p2")
ImageWithPose = namedtuple("ImageWithPose", "filePath pose cameraId")


def quatToRotMat(q):
    """:55-58: qvec = (w, x, y, z) -> 3x3 rotation (simd_quatd(ix: q1, iy: q2, iz: q3, r: q0))."""
    w, x, y, z = (float(v) for v in q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], np.float64)


def _intrinsics3(fx, fy, cx, cy):
    return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)


def colmapReadCamerasAndPoses(binRoot, imageRoot):
    """cameras.bin + images.bin -> ({camera id: ColmapCamera}, [ImageWithPose]); pose = camera-to-world
    [R^T | -R^T t] of the stored world-to-camera (qvec, tvec) (:258-296).  Unknown camera models parse as PINHOLE
    (:199)."""
    cam_path, img_path = os.path.join(binRoot, "cameras.bin"), os.path.join(binRoot, "images.bin")
    if not (os.path.exists(cam_path) and os.path.exists(img_path)):
        raise FileNotFoundError("Colmap files missing")
    camMap = {}
    with open(cam_path, "rb") as f:
        rd = lambda fmt: struct.unpack("<" + fmt, f.read(struct.calcsize("<" + fmt)))
        (n,) = rd("Q")
        for _ in range(n):
            camId, model = rd("Ii")
            width, height = rd("QQ")
            k1 = k2 = p1 = p2 = None
            if model == 0:                      # SIMPLE_PINHOLE
                fx, cx, cy = rd("ddd"); fy = fx
            elif model == 2:                    # SIMPLE_RADIAL
                fx, cx, cy, k1 = rd("dddd"); fy = fx
            elif model == 3:                    # OPENCV
                fx, fy, cx, cy, k1, k2, p1, p2 = rd("dddddddd")
            else:                               # PINHOLE, and anything unknown
                fx, fy, cx, cy = rd("dddd")
            camMap[camId] = ColmapCamera(camId, width, height, fx, fy, cx, cy, k1, k2, p1, p2)
    poses = []
    with open(img_path, "rb") as f:
        rd = lambda fmt: struct.unpack("<" + fmt, f.read(struct.calcsize("<" + fmt)))
        (n,) = rd("Q")
        for _ in range(n):
            rd("I")                                              # image id
            q = rd("dddd"); t = np.array(rd("ddd"), np.float64)
            Rinv = quatToRotMat(q).T
            (camId,) = rd("I")
            name = bytearray()
            while True:
                ch = f.read(1)
                if not ch or ch == b"\x00":
                    break
                name += ch
            pose = np.eye(4, dtype=np.float64)
            pose[:3, :3] = Rinv
            pose[:3, 3] = -(Rinv @ t)
            (np2d,) = rd("Q")
            f.seek(24 * np2d, os.SEEK_CUR)                       # x, y (f64), point3D id (u64)
            poses.append(ImageWithPose(os.path.join(imageRoot, name.decode("utf-8")), pose, camId))
    return camMap, poses


def colmapReadPointSet(points3DPath):
    """points3D.bin -> (points [N,3] f64, colors [N,3] u8) (:398-440)."""
    pts, cols = [], []
    with open(points3DPath, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            rec = f.read(8 + 24 + 3 + 8 + 8)
            _, x, y, z, r, g, b, _err, track = struct.unpack("<QdddBBBdQ", rec)
            pts.append((x, y, z)); cols.append((r, g, b))
            f.seek(8 * track, os.SEEK_CUR)
    return np.array(pts, np.float64).reshape(-1, 3), np.array(cols, np.uint8).reshape(-1, 3)


class ColmapDataLoader:
    def __init__(self, binRoot, imageRoot):
        self.binRoot, self.imageRoot = binRoot, imageRoot

    def getOriginalImageSize(self):
        camMap, poses = colmapReadCamerasAndPoses(self.binRoot, self.imageRoot)
        cam = camMap[poses[0].cameraId]
        return int(cam.width), int(cam.height)

    def load(self, resizeFactor: float = 1.0, whiteBackground: bool = False, readImage=readImageRGBA):
        """loadTrainDataAndPointCloud (:441-500) -> (TrainData, PointCloud, TILE_SIZE_H_W)."""
        camMap, poses = colmapReadCamerasAndPoses(self.binRoot, self.imageRoot)
        intr = np.stack([_intrinsics3(*(camMap[p.cameraId][3:7])) for p in poses])
        if resizeFactor != 1.0:
            intr[:, :2, :3] *= np.float32(resizeFactor)
        c2ws = np.stack([p.pose.astype(np.float32) for p in poses])
        Hs, Ws, rgbs, alphas = _stack_frames([readImage(p.filePath, resizeFactor) for p in poses], whiteBackground)
        pts, cols = colmapReadPointSet(os.path.join(self.binRoot, "points3D.bin"))
        ch = cols.astype(np.float32) / np.float32(255.0)
        pcd = PointCloud(pts.astype(np.float32), dict(R=ch[:, 0], G=ch[:, 1], B=ch[:, 2]))
        return (TrainData(Hs, Ws, intr, c2ws, rgbs, alphas, None), pcd,
                TILE_SIZE_H_W(w=int(Ws[0]) // 4, h=int(Hs[0]) // 4))


# ---- OpenGL -> OpenCV -------------------------------------------------------------------------------------------
def opengl_c2w_to_opencv(c2w):
    """NerfStudioDataLoader.swift:357-366 / BlenderDataLoader.swift:84-88: invert, negate rows 1 and 2 of the
    world-to-camera matrix, invert back (all in f64)."""
    w2c = np.linalg.inv(np.asarray(c2w, np.float64))
    w2c[1:3, :] *= -1
    return np.linalg.inv(w2c)


# ---- NerfStudio (NerfStudioDataLoader.swift) ----------------------------------------------------------------------
def parsePLY(path):
    """:98-215: vertex positions (f32) and colours (u8) of an ascii or binary (x y z r g b, 15-byte vertices) PLY."""
    blob = open(path, "rb").read()
    end = blob.find(b"end_header\n")
    if end < 0:
        raise ValueError("No end_header")
    end += len(b"end_header\n")
    header = blob[:end].decode("ascii")
    line = next((l for l in header.split("\n") if l.startswith("element vertex")), None)
    if line is None:
        raise ValueError("No vertex count")
    n = int(line.split(" ")[-1])
    if "format ascii" in header:
        xyz, rgb = [], []
        for ln in blob[end:].decode("ascii").splitlines()[:n]:
            v = ln.split(" ")
            if len(v) < 6:
                continue
            xyz.append([float(v[0]), float(v[1]), float(v[2])]); rgb.append([int(v[3]), int(v[4]), int(v[5])])
        return np.array(xyz, np.float32).reshape(-1, 3), np.array(rgb, np.uint8).reshape(-1, 3)
    if end + n * 15 > len(blob):
        raise ValueError("File too small")
    rec = np.frombuffer(blob, np.dtype([("p", "<f4", 3), ("c", "u1", 3)]), count=n, offset=end)
    return rec["p"].copy(), rec["c"].copy()


class NerfStudioDataLoader:
    def __init__(self, directory):
        self.directory = directory

    def load(self, resizeFactor: float = 1.0, whiteBackground: bool = False, readImage=readImageRGBA):
        """loadTrainDataAndPointCloud (:385-416): transforms.json + its ply_file_path."""
        meta = json.load(open(os.path.join(self.directory, "transforms.json")))
        xyz, rgb = parsePLY(os.path.join(self.directory, meta["ply_file_path"]))
        ch = rgb.astype(np.float32) / np.float32(255.0)
        pcd = PointCloud(xyz, dict(R=ch[:, 0], G=ch[:, 1], B=ch[:, 2]))

        def intrinsic(d):
            return _intrinsics3(d["fl_x"], d["fl_y"], d["cx"], d["cy"]) if all(k in d for k in ("fl_x", "fl_y", "cx", "cy")) else None
        intr, c2ws, frames = [], [], []
        for fr in meta["frames"]:
            K = intrinsic(fr)
            K = intrinsic(meta) if K is None else K
            if K is None:
                raise ValueError("Failed to load intrinsic matrix")
            if resizeFactor != 1.0:
                K[:2, :3] *= np.float32(resizeFactor)
            intr.append(K)
            frames.append(readImage(os.path.join(self.directory, fr["file_path"]), resizeFactor))
            # the reference reads transform_matrix as Float before widening (:359)
            c2ws.append(opengl_c2w_to_opencv(np.asarray(fr["transform_matrix"], np.float32).astype(np.float64)).astype(np.float32))
        Hs, Ws, rgbs, alphas = _stack_frames(frames, whiteBackground)
        return (TrainData(Hs, Ws, np.stack(intr), np.stack(c2ws), rgbs[..., :3], alphas, None), pcd,
                TILE_SIZE_H_W(w=int(Ws[0]) // 4, h=int(Hs[0]) // 4))


# ---- Blender demo set (BlenderDataLoader.swift) --------------------------------------------------------------------
class BlenderDemoDataLoader:
    def __init__(self, folder):
        self.folder = folder

    def readCamera(self):
        """:71-96: info.json -> rgb paths, OpenCV camera-to-world poses, 4x4 intrinsics, max depth."""
        info = json.load(open(os.path.join(self.folder, "info.json")))
        files, poses, intr = [], [], []
        for img in info["images"]:
            files.append(os.path.join(self.folder, img["rgb"]))
            poses.append(opengl_c2w_to_opencv(np.asarray(img["pose"], np.float64)))
            I = np.zeros((4, 4), np.float64)
            I[:3, :3] = np.asarray(img["intrinsic"], np.float64)[:3, :3]
            I[3, 3] = 1
            intr.append(I)
        maxDepth = info["images"][0]["max_depth"] if info["images"] else 1.0
        return files, poses, intr, maxDepth

    def load(self, resizeFactor: float = 1.0, whiteBackground: bool = False):
        """readAll + getPointCloudsFromTrainData (:97-295, :318-341): rgb, <name>_depth.png * max_depth and
        <name>_alpha.png per view; the point cloud is back-projected from the opaque pixels."""
        files, poses, intr, maxDepth = self.readCamera()
        Hs, Ws, rgbs, alphas, depths, Ks = [], [], [], [], [], []
        for path, K in zip(files, intr):
            rgb = _load_image(path, resizeFactor, "RGB")
            base = os.path.splitext(os.path.basename(path))[0].split("_")[0]
            d = os.path.dirname(path)
            depth = _load_image(os.path.join(d, f"{base}_depth.png"), resizeFactor, "L") * np.float32(maxDepth)
            alpha = _load_image(os.path.join(d, f"{base}_alpha.png"), resizeFactor, "L")
            K = K.astype(np.float32)
            if resizeFactor != 1.0:
                K[:2, :3] *= np.float32(resizeFactor)
            if whiteBackground:
                rgb = alpha[..., None] * rgb + (1 - alpha)[..., None]
            Hs.append(rgb.shape[0]); Ws.append(rgb.shape[1]); rgbs.append(rgb); alphas.append(alpha); depths.append(depth); Ks.append(K)
        data = TrainData(np.array(Hs, np.float32), np.array(Ws, np.float32), np.stack(Ks),
                         np.stack([p.astype(np.float32) for p in poses]), np.stack(rgbs).astype(np.float32),
                         np.stack(alphas).astype(np.float32), np.stack(depths).astype(np.float32))
        return data, getPointCloudsFromTrainData(data), TILE_SIZE_H_W(w=int(Ws[0]) // 4, h=int(Hs[0]) // 4)
