"""Dataset loaders and point-cloud initialisation.  The parsers are pure host code: known-answer
tests on files written here.  distTopK / create_from_pcd run on the GPU against a numpy brute force."""
import json
import struct

import numpy as np
import pytest

from gaussiansplattingmlx_amd import data as D
from gaussiansplattingmlx_amd.pointcloud import PointCloud, getPointCloudsFromTrainData, getRaysFromImages, inv3x3


def _write_colmap(root, cams, images, points):
    with open(root / "cameras.bin", "wb") as f:
        f.write(struct.pack("<Q", len(cams)))
        for cid, model, w, h, params in cams:
            f.write(struct.pack("<IiQQ", cid, model, w, h) + struct.pack("<%dd" % len(params), *params))
    with open(root / "images.bin", "wb") as f:
        f.write(struct.pack("<Q", len(images)))
        for iid, q, t, cid, name, n2d in images:
            f.write(struct.pack("<I4d3dI", iid, *q, *t, cid) + name.encode() + b"\x00" + struct.pack("<Q", n2d))
            f.write(b"\x07" * (24 * n2d))
    with open(root / "points3D.bin", "wb") as f:
        f.write(struct.pack("<Q", len(points)))
        for pid, xyz, rgb, track in points:
            f.write(struct.pack("<Q3d3BdQ", pid, *xyz, *rgb, 0.5, track) + b"\x01" * (8 * track))


def test_colmap_binary_parsing_and_pose_convention(tmp_path):
    s = np.sqrt(0.5)
    cams = [(1, 1, 800, 600, (500.0, 510.0, 400.0, 300.0)), (2, 0, 640, 480, (450.0, 320.0, 240.0)),
            (3, 2, 640, 480, (450.0, 320.0, 240.0, 0.01)), (4, 3, 64, 48, (50.0, 51.0, 32.0, 24.0, .1, .2, .3, .4))]
    images = [(10, (1.0, 0.0, 0.0, 0.0), (1.0, 2.0, 3.0), 1, "a.png", 2),
              (11, (s, 0.0, 0.0, s), (0.5, -1.0, 2.0), 2, "sub/b.png", 0)]        # 90 degrees about z
    points = [(7, (0.1, 0.2, 0.3), (255, 128, 0), 3), (8, (-1.0, 2.0, 5.0), (1, 2, 3), 0)]
    _write_colmap(tmp_path, cams, images, points)
    camMap, poses = D.colmapReadCamerasAndPoses(str(tmp_path), "/img")
    assert camMap[1].fx == 500.0 and camMap[1].fy == 510.0 and camMap[2].fy == camMap[2].fx == 450.0
    assert camMap[3].k1 == 0.01 and camMap[4].p2 == 0.4 and camMap[4].width == 64
    assert poses[0].filePath == "/img/a.png" and poses[1].filePath == "/img/sub/b.png" and poses[1].cameraId == 2
    # identity rotation: c2w = [I | -t]
    np.testing.assert_allclose(poses[0].pose, np.array([[1, 0, 0, -1], [0, 1, 0, -2], [0, 0, 1, -3], [0, 0, 0, 1.0]]))
    # R = Rz(90): w2c x = R x + t  ->  c2w = [R^T | -R^T t]
    R = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1.0]])
    np.testing.assert_allclose(D.quatToRotMat((s, 0, 0, s)), R, atol=1e-12)
    np.testing.assert_allclose(poses[1].pose[:3, :3], R.T, atol=1e-12)
    np.testing.assert_allclose(poses[1].pose[:3, 3], -R.T @ np.array([0.5, -1.0, 2.0]), atol=1e-12)
    pts, cols = D.colmapReadPointSet(str(tmp_path / "points3D.bin"))
    np.testing.assert_array_equal(pts, [[0.1, 0.2, 0.3], [-1.0, 2.0, 5.0]])
    np.testing.assert_array_equal(cols, [[255, 128, 0], [1, 2, 3]])

    fake = lambda path, rf: (np.full((int(60 * rf), int(80 * rf), 3), 0.25, np.float32),
                             np.full((int(60 * rf), int(80 * rf)), 0.5, np.float32), 60.0 * rf, 80.0 * rf)
    td, pcd, tile = D.ColmapDataLoader(str(tmp_path), "/img").load(0.5, True, readImage=fake)
    np.testing.assert_allclose(td.intrinsicArray[0], [[250, 0, 200], [0, 255, 150], [0, 0, 1]])       # scaled rows 0-1
    assert td.c2wArray.shape == (2, 4, 4) and td.depthArray is None and tile == D.TILE_SIZE_H_W(w=10, h=7)
    np.testing.assert_allclose(td.rgbArray, 0.5 * 0.25 + 0.5)                                          # white background
    np.testing.assert_allclose(pcd.select_channels(["R", "G", "B"])[0], [255, 128, 0])
    assert D.ColmapDataLoader(str(tmp_path), "/img").getOriginalImageSize() == (800, 600)
    with pytest.raises(FileNotFoundError):
        D.colmapReadCamerasAndPoses(str(tmp_path / "nope"), "/img")


def test_opengl_to_opencv_flip():
    rng = np.random.default_rng(0)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    c2w = np.eye(4); c2w[:3, :3] = D.quatToRotMat(q); c2w[:3, 3] = [1, 2, 3]
    out = D.opengl_c2w_to_opencv(c2w)
    # flipping rows 1, 2 of w2c = flipping the camera's y and z axes: columns 1, 2 of the c2w rotation, same centre
    np.testing.assert_allclose(out[:3, 0], c2w[:3, 0], atol=1e-12)
    np.testing.assert_allclose(out[:3, 1:3], -c2w[:3, 1:3], atol=1e-12)
    np.testing.assert_allclose(out[:3, 3], c2w[:3, 3], atol=1e-12)


def test_nerfstudio_and_ply_pointcloud(tmp_path):
    xyz = np.array([[0, 0, 0], [1, 2, 3], [-1, 0.5, 2]], np.float32)
    rgb = np.array([[255, 0, 0], [0, 255, 0], [10, 20, 30]], np.uint8)
    rec = np.zeros(3, np.dtype([("p", "<f4", 3), ("c", "u1", 3)])); rec["p"] = xyz; rec["c"] = rgb
    head = ("ply\nformat binary_little_endian 1.0\nelement vertex 3\nproperty float x\nproperty float y\n"
            "property float z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
    (tmp_path / "pts.ply").write_bytes(head.encode() + rec.tobytes())
    p, c = D.parsePLY(str(tmp_path / "pts.ply"))
    np.testing.assert_array_equal(p, xyz); np.testing.assert_array_equal(c, rgb)
    (tmp_path / "a.ply").write_text(head.replace("binary_little_endian", "ascii") +
                                    "0 0 0 255 0 0\n1 2 3 0 255 0\nbad line\n-1 0.5 2 10 20 30\n")
    p2, c2 = D.parsePLY(str(tmp_path / "a.ply"))            # prefix(vertexCount) lines, short lines skipped (:160-170)
    np.testing.assert_array_equal(p2, xyz[:2]); np.testing.assert_array_equal(c2, rgb[:2])
    T = np.eye(4); T[:3, 3] = [0, 0, 4]
    meta = dict(ply_file_path="pts.ply", fl_x=100.0, fl_y=110.0, cx=16.0, cy=12.0,
                frames=[dict(file_path="f0.png", transform_matrix=T.tolist()),
                        dict(file_path="f1.png", transform_matrix=T.tolist(), fl_x=50.0, fl_y=55.0, cx=8.0, cy=6.0)])
    (tmp_path / "transforms.json").write_text(json.dumps(meta))
    fake = lambda path, rf: (np.zeros((24, 32, 3), np.float32), np.ones((24, 32), np.float32), 24.0, 32.0)
    td, pcd, tile = D.NerfStudioDataLoader(str(tmp_path)).load(1.0, False, readImage=fake)
    np.testing.assert_allclose(td.intrinsicArray[0], [[100, 0, 16], [0, 110, 12], [0, 0, 1]])        # file-level
    np.testing.assert_allclose(td.intrinsicArray[1], [[50, 0, 8], [0, 55, 6], [0, 0, 1]])            # per-frame wins
    np.testing.assert_allclose(td.c2wArray[0], np.diag([1, -1, -1, 1.0]) + np.array([[0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 4], [0, 0, 0, 0]]))
    assert tile == D.TILE_SIZE_H_W(w=8, h=6) and pcd.coords.shape == (3, 3)


def test_blender_demo_loader_and_backprojection(tmp_path):
    from PIL import Image
    H, W, f = 8, 12, 20.0
    pose = np.eye(4); pose[:3, 3] = [0.5, -0.25, 1.0]
    info = dict(backend="x", light_mode="y", fast_mode=True, format_version=1, channels=["rgb"], scale=1.0, bbox=[[0] * 3] * 2,
                images=[dict(intrinsic=[[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], pose=pose.tolist(), rgb="0_rgb.png",
                             depth="0_depth.png", alpha="0_alpha.png", max_depth=4.0, HW=[H, W])])
    (tmp_path / "info.json").write_text(json.dumps(info))
    Image.fromarray(np.full((H, W, 3), 128, np.uint8)).save(tmp_path / "0_rgb.png")
    depth = np.full((H, W), 255, np.uint8); alpha = np.zeros((H, W), np.uint8); alpha[2:4, 3:6] = 255
    Image.fromarray(depth).save(tmp_path / "0_depth.png"); Image.fromarray(alpha).save(tmp_path / "0_alpha.png")
    td, pcd, tile = D.BlenderDemoDataLoader(str(tmp_path)).load(1.0, False)
    assert td.depthArray.shape == (1, H, W) and float(td.depthArray.max()) == 4.0 and tile == D.TILE_SIZE_H_W(w=3, h=2)
    assert pcd.coords.shape == (6, 3)                                  # the six opaque pixels
    # pixel (u=3, v=2), OpenCV camera of the flipped pose: p = o + R K^-1 (u, v, 1) * depth
    c2w = D.opengl_c2w_to_opencv(pose)
    want = c2w[:3, 3] + c2w[:3, :3] @ (np.array([(3 - W / 2) / f, (2 - H / 2) / f, 1.0]) * 4.0)
    np.testing.assert_allclose(pcd.coords[0], want, rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(pcd.select_channels(["R"])[0], 128)


def test_pointcloud_utils():
    m = np.array([[[2, 0, 1], [0, 3, 0], [1, 0, 4.0]]], np.float32)
    np.testing.assert_allclose(inv3x3(m)[0] @ m[0], np.eye(3), atol=1e-6)
    K = np.array([[[10, 0, 2, 0], [0, 10, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]], np.float32)
    c2w = np.eye(4, dtype=np.float32)[None].copy(); c2w[0, :3, 3] = [1, 2, 3]
    o, d = getRaysFromImages(2, 3, K, c2w)
    assert o.shape == d.shape == (1, 6, 3)
    np.testing.assert_allclose(d[0, 4], [(1 - 2) / 10, (1 - 1) / 10, 1], atol=1e-6)      # pixel index 4 = (u=1, v=1)
    np.testing.assert_allclose(o[0, 0], [1, 2, 3])
    rng = np.random.default_rng(1)
    pts = rng.normal(size=(500, 3)).astype(np.float32); pts[0] = [40, 0, 0]
    pc = PointCloud(pts.copy(), dict(R=rng.uniform(size=500)))

    class TD:
        c2wArray = np.eye(4, dtype=np.float32)[None].repeat(2, 0)
    td = TD()
    mean = pts.mean(0)
    pc.centering(td)
    np.testing.assert_allclose(td.c2wArray[:, :3, 3], np.tile(-mean, (2, 1)), atol=1e-6)
    assert pc.coords.shape[0] < 500 and np.abs(pc.coords[:, 0]).max() < 40 and pc.channels["R"].shape[0] == pc.coords.shape[0]
    assert pc.randomSample(10, np.random.default_rng(0)).coords.shape == (10, 3) and pc.randomSample(10 ** 6) is pc


@pytest.mark.gpu
def test_dist_topk_and_create_from_pcd():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test on a box without a GPU")
    from gaussiansplattingmlx_amd.model_init import C0, create_from_pcd, distTopK
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    r = GaussianRenderer(3, 64, 64)
    rng = np.random.default_rng(5)
    N = 3000
    X = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    d2 = ((X[:, None, :].astype(np.float64) - X[None, :, :]) ** 2).sum(-1)
    want = np.sort(d2, axis=1)[:, :3].mean(1)                         # self distance 0 is one of the three
    got_all = distTopK(r, X, 3, reference_stride=False).cpu().numpy()
    np.testing.assert_allclose(got_all, want, rtol=2e-5, atol=1e-9)
    got_ref = distTopK(r, X, 3, reference_stride=True).cpu().numpy()  # N/256+1 = 12 -> only i = 0: first 256 points
    np.testing.assert_array_equal(got_ref[:256], got_all[:256])
    assert not got_ref[256:].any()
    big = distTopK(r, np.zeros((70000, 3), np.float32) + X[:1], 2, reference_stride=True).cpu().numpy()
    assert big.shape == (70000,)                                       # N/256+1 = 274 -> chunks at 0 and 256
    pcd = PointCloud(X, dict(R=rng.uniform(size=N), G=rng.uniform(size=N), B=rng.uniform(size=N)))
    p = create_from_pcd(r, pcd, sh_degree=3)
    assert {k: tuple(v.shape) for k, v in p.items()} == dict(xyz=(N, 3), features_dc=(N, 1, 3), features_rest=(N, 15, 3),
                                                             scales=(N, 3), rotation=(N, 4), opacity=(N, 1))
    col = np.round(np.stack([pcd.channels[c] for c in "RGB"], -1) * 255) / 255
    np.testing.assert_allclose(p["features_dc"].cpu().numpy()[:, 0], (col - 0.5) / C0, rtol=1e-5, atol=1e-6)
    sc = p["scales"].cpu().numpy()
    np.testing.assert_allclose(sc[:256, 0], np.log(np.sqrt(np.maximum(want[:256], 1e-7))), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sc[256:], np.log(np.sqrt(1e-7)), rtol=1e-6)              # the stride quirk's floor
    assert (sc[:, 0] == sc[:, 1]).all() and (sc[:, 0] == sc[:, 2]).all()
    np.testing.assert_allclose(p["opacity"].cpu().numpy(), np.log(0.1 / 0.9), rtol=1e-6)
    np.testing.assert_array_equal(p["rotation"].cpu().numpy(), np.tile([1, 0, 0, 0], (N, 1)))
    with pytest.raises(Exception):
        r._check(r.lib.gs_dist_topk(r.ctx, 10, 9, 0, 10, None, None))                  # k > 8
