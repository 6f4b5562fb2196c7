"""TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference snapshot format (Data/PlyWriter.swift).

write_gaussian_binary follows writeGaussianBinary (:22-113): header text line by line, then per vertex
x y z, f_dc_0..2, features_rest[i] flattened [M][3] (coefficient-major), opacity, scale_0..2, rot_0..3 as
little-endian float32.  load_gaussian_binary_ply follows loadGaussianBinaryPLY (:149-233): fields are looked up by
name among the header's `property float` lines.  Parity unpinned: the reference has no PLY test or fixture; the
byte layout is pinned by the writer's source text only."""
import numpy as np


def header(num_points: int, M: int) -> bytes:
    h = "ply\nformat binary_little_endian 1.0\n"
    h += f"comment features_rest_shape {M} 3\n"
    h += f"element vertex {num_points}\n"
    for n in ("x", "y", "z", "f_dc_0", "f_dc_1", "f_dc_2"):
        h += f"property float {n}\n"
    for i in range(M * 3):
        h += f"property float f_rest_{i}\n"
    for n in ("opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"):
        h += f"property float {n}\n"
    h += "end_header\n"
    return h.encode("ascii")


def write_gaussian_binary(positions, features_dc, features_rest, opacities, scales, rotations) -> bytes:
    pos = np.asarray(positions, "<f4").reshape(-1, 3)
    N = pos.shape[0]
    rest = np.asarray(features_rest, "<f4")
    M = rest.shape[1] if rest.ndim == 3 else 0
    rows = np.concatenate([pos, np.asarray(features_dc, "<f4").reshape(N, 3), rest.reshape(N, M * 3),
                           np.asarray(opacities, "<f4").reshape(N, 1), np.asarray(scales, "<f4").reshape(N, 3),
                           np.asarray(rotations, "<f4").reshape(N, 4)], axis=1).astype("<f4")
    return header(N, M) + rows.tobytes()


def load_gaussian_binary_ply(blob: bytes) -> dict:
    end = blob.find(b"end_header\n")
    if end < 0:
        raise ValueError("No end_header")
    end += len(b"end_header\n")
    num_points, shape, fields = 0, None, []
    for line in blob[:end].decode("ascii").split("\n"):
        parts = line.split()
        if line.startswith("comment features_rest_shape") and len(parts) >= 4:
            shape = (int(parts[2]), int(parts[3]))
        if len(parts) >= 3 and parts[0] == "element" and parts[1] == "vertex":
            num_points = int(parts[2])
        elif len(parts) == 3 and parts[0] == "property" and parts[1] == "float":
            fields.append(parts[2])
    if shape is None:
        raise ValueError("No features_rest_shape comment")
    M, D = shape
    F = len(fields)
    rows = np.frombuffer(blob, "<f4", count=num_points * F, offset=end).reshape(num_points, F)
    col = lambda names: rows[:, [fields.index(n) for n in names]]
    return dict(positions=col(["x", "y", "z"]), features_dc=col(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(-1, 1, 3),
                features_rest=col([f"f_rest_{i}" for i in range(M * D)]).reshape(-1, M, D),
                opacities=col(["opacity"]), scales=col(["scale_0", "scale_1", "scale_2"]),
                rotations=col(["rot_0", "rot_1", "rot_2", "rot_3"]))
