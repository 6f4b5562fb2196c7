"""Snapshot format (Data/PlyWriter.swift): oracle known-answers on CPU, byte parity on the GPU."""
import struct

import numpy as np
import pytest

from oracle import ply_oracle


def _params(N, M, seed=0):
    rng = np.random.default_rng(seed)
    return dict(positions=rng.normal(size=(N, 3)), features_dc=rng.normal(size=(N, 1, 3)),
                features_rest=rng.normal(size=(N, M, 3)), opacities=rng.normal(size=(N, 1)),
                scales=rng.normal(size=(N, 3)), rotations=rng.normal(size=(N, 4)))


def test_oracle_header_and_vertex_layout_follow_the_writer_text():
    p = _params(2, 2)
    p = {k: v.astype(np.float32) for k, v in p.items()}
    blob = ply_oracle.write_gaussian_binary(**p)
    head, _, data = blob.partition(b"end_header\n")
    lines = head.decode().split("\n")
    assert lines[:4] == ["ply", "format binary_little_endian 1.0", "comment features_rest_shape 2 3", "element vertex 2"]
    names = [l.split()[2] for l in lines[4:] if l]
    assert names == (["x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(6)] +
                     ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    assert len(data) == 2 * 20 * 4
    v1 = struct.unpack("<20f", data[80:160])                       # second vertex
    assert v1[:3] == tuple(p["positions"][1]) and v1[3:6] == tuple(p["features_dc"][1, 0])
    # f_rest is coefficient-major, channel-minor: f_rest_3 is coefficient 1, channel 0
    assert v1[6 + 3] == p["features_rest"][1, 1, 0] and v1[6 + 2] == p["features_rest"][1, 0, 2]
    assert v1[12] == p["opacities"][1, 0] and v1[13:16] == tuple(p["scales"][1]) and v1[16:20] == tuple(p["rotations"][1])


def test_oracle_round_trip_and_reordered_header():
    p = {k: v.astype(np.float32) for k, v in _params(5, 3, 1).items()}
    blob = ply_oracle.write_gaussian_binary(**p)
    back = ply_oracle.load_gaussian_binary_ply(blob)
    for k in p:
        np.testing.assert_array_equal(back[k], p[k])
    with pytest.raises(ValueError):
        ply_oracle.load_gaussian_binary_ply(blob.replace(b"comment features_rest_shape 3 3\n", b""))
    with pytest.raises(ValueError):
        ply_oracle.load_gaussian_binary_ply(blob.replace(b"end_header\n", b"end_headr\n"))


def _reordered(blob, p):
    """Same vertices with opacity moved first and a non-float property line, as a foreign writer might emit."""
    N, M = p["positions"].shape[0], p["features_rest"].shape[1]
    F = 14 + 3 * M
    head, _, data = blob.partition(b"end_header\n")
    rows = np.frombuffer(data, "<f4").reshape(N, F)
    lines = head.decode().split("\n")
    props = [l for l in lines if l.startswith("property")]
    other = [l for l in lines if l and not l.startswith("property")]
    op = props.index("property float opacity")
    order = [op] + [i for i in range(F) if i != op]
    new_head = "\n".join(other + ["property uchar ignored"] + [props[i] for i in order]) + "\n"
    return new_head.encode() + b"end_header\n" + np.ascontiguousarray(rows[:, order]).tobytes()


@pytest.mark.gpu
@pytest.mark.parametrize("N,M", [(0, 24), (1, 0), (1000, 24), (70001, 15)])
def test_hip_writer_bytes_equal_the_reference_layout(tmp_path, N, M):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test on a box without a GPU")
    from gaussiansplattingmlx_amd.ply import PlyWriter
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    r = GaussianRenderer(4 if M >= 24 else 0, 64, 64)
    w = PlyWriter(r)
    p = {k: v.astype(np.float32) for k, v in _params(N, M, 7).items()}
    path = tmp_path / "sub" / "dir" / f"iteration_{N}.ply"          # parent directories are created (:106-111)
    w.writeGaussianBinary(p["positions"], p["features_dc"], p["features_rest"], p["opacities"], p["scales"],
                          p["rotations"], to=path)
    blob = path.read_bytes()
    assert blob == ply_oracle.write_gaussian_binary(**p)
    assert w.probe(path) == (N, M, 3)
    back = w.loadGaussianBinaryPLYAsMLX(path)
    for k in p:
        assert tuple(back[k].shape) == p[k].shape
        np.testing.assert_array_equal(back[k].cpu().numpy(), p[k])
    if N:
        rows = w.packRows(**p).cpu().numpy()
        assert rows.tobytes() == blob.partition(b"end_header\n")[2]
        # a header in another order loads to the same tensors
        path2 = tmp_path / "reordered.ply"
        path2.write_bytes(_reordered(blob, p))
        back2 = w.loadGaussianBinaryPLYAsMLX(path2)
        for k in p:
            np.testing.assert_array_equal(back2[k].cpu().numpy(), p[k])


@pytest.mark.gpu
def test_hip_ply_errors(tmp_path):
    import torch
    from gaussiansplattingmlx_amd._lib import GsplatError
    from gaussiansplattingmlx_amd.ply import PlyWriter
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    w = PlyWriter(GaussianRenderer(0, 64, 64))
    with pytest.raises(GsplatError):
        w.probe(tmp_path / "missing.ply")
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nend_header\n")
    with pytest.raises(GsplatError):                                  # no features_rest_shape comment (:184-188)
        w.probe(bad)
    p = {k: v.astype(np.float32) for k, v in _params(3, 2).items()}
    blob = ply_oracle.write_gaussian_binary(**p)
    trunc = tmp_path / "trunc.ply"
    trunc.write_bytes(blob[:-8])
    with pytest.raises(GsplatError):
        w.loadGaussianBinaryPLYAsMLX(trunc)
    with pytest.raises(ValueError):                                   # attribute size mismatch (:34-43)
        w.writeGaussianBinary(p["positions"], p["features_dc"][:2], p["features_rest"], p["opacities"], p["scales"],
                              p["rotations"], to=tmp_path / "x.ply")
