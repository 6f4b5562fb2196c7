"""CPU-side checks: the C-ABI library loads and exports what include/gsplat.h declares; host logic."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from gaussiansplattingmlx_amd import build
    build.build()
    from gaussiansplattingmlx_amd import _lib
    return _lib.load()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "gsplat.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(gs_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    names = _declared_functions()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in gsplat.h but not exported"
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "gaussiansplattingmlx_amd",
                                                                     "libgsplat_hip.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (gs_[a-z0-9_]+)", out))
    assert set(names) <= exported
    from gaussiansplattingmlx_amd import _lib
    assert set(_lib.exported_symbols()) == set(names)       # the Python binding covers the whole header


def test_library_contains_gfx950_code():
    so = os.path.join(ROOT, "gaussiansplattingmlx_amd", "libgsplat_hip.so")
    data = open(so, "rb").read()
    assert b"gfx950" in data and b"blend_fwd_kernel" in data and b"radix_scatter_kernel" in data


def test_abi_version_and_host_only_entry_points(lib):
    assert lib.gs_abi_version() == 6
    w = np.zeros(121, np.float32)
    assert lib.gs_ssim_window(11, ctypes.c_float(1.5), w.ctypes.data_as(ctypes.c_void_p)) == 0
    from oracle.oracle import Oracle
    np.testing.assert_array_equal(w, Oracle(np.float32).ssim_window())
    assert lib.gs_ssim_window(0, ctypes.c_float(1.5), w.ctypes.data_as(ctypes.c_void_p)) == 1   # invalid arg
    assert lib.gs_ctx_destroy(None) == 0


def test_no_gpu_means_loud_failure(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    rc = lib.gs_ctx_create(0, 64, 64, 16, 16, 4, 0, ctypes.byref(ctx))
    assert rc == 6 and not ctx.value                     # GS_ERR_NO_DEVICE, never a silent CPU path
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    with pytest.raises(RuntimeError):
        GaussianRenderer(4, 64, 64)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "gaussiansplattingmlx_amd")
    banned = ("import oracle", "from oracle", "gs_oracle", "libgs_oracle", "oracle/")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, f)).read()
                for b in banned:
                    assert b not in txt, (dirpath, f, b)


def test_scene_generator_is_seeded_and_shaped():
    from gaussiansplattingmlx_amd.scenes import make_gaussians, lego_cameras
    a = make_gaussians(2000, "trained_like", 5)
    b = make_gaussians(2000, "trained_like", 5)
    for k in a:
        np.testing.assert_array_equal(a[k], b[k])
    assert a["features_rest"].shape == (2000, 24, 3) and a["rotation"].shape == (2000, 4)
    c = make_gaussians(500, "random_init", 1)
    assert np.allclose(c["opacity"], np.log(0.1 / 0.9)) and (c["rotation"][:, 0] == 1).all()
    cams = lego_cameras(3, 800, 800, 1)
    assert abs(float(cams[0].focalX) - 1111.11) < 0.01
    for cam in cams:
        assert abs(np.linalg.norm(cam.cameraCenter) - 4.0311) < 1e-6


def test_c_header_is_plain_c_and_the_cpp_host_links(tmp_path, lib):
    """include/gsplat.h compiles as C99; host/GaussianRenderer.hpp (the C++ mirror of the reference class) compiles and
    a program using only the header and the .so links, reports the ABI version and -- on a box without a GPU -- gets
    GS_ERR_NO_DEVICE from gs_ctx_create (exit code 10) rather than any fallback."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not shutil.which("gcc") or not shutil.which("g++"):
        pytest.skip("no host compiler")
    c_src = tmp_path / "use.c"
    c_src.write_text('#include "include/gsplat.h"\nint main(void) { return gs_abi_version() == GSPLAT_ABI_VERSION ? 0 : 1; }\n')
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", root, str(c_src)], check=True)
    from gaussiansplattingmlx_amd import _lib
    so = _lib.LIB_PATH
    exe = tmp_path / "abi_check"
    subprocess.run(["g++", "-std=c++17", "-Wall", "-I", root, os.path.join(root, "host", "abi_check.cpp"), "-o", str(exe),
                    so, "-Wl,-rpath," + os.path.dirname(so)], check=True)
    import torch
    env = dict(os.environ)
    # the .so needs libamdhip64: the one torch ships (what _lib.load() resolves by importing torch first)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    res = subprocess.run([str(exe)], env=env, capture_output=True, text=True, timeout=120)
    assert "window sum 1.0000" in res.stdout, res.stdout + res.stderr
    assert res.returncode in (0, 10), (res.returncode, res.stdout, res.stderr)
    if not torch.cuda.is_available():
        assert res.returncode == 10


def test_cut_policy():
    """renderer.CutPolicy (pure host logic): no cuts while the view's cuts are empty -- they are written by the
    preparation of a backward, a view that was only rendered has none -- a view whose cuts leave out too little sits
    out probe_interval forwards and is then tried again, a missed forward changes nothing."""
    from gaussiansplattingmlx_amd.renderer import CutPolicy
    p = CutPolicy(min_dropped=1000, probe_interval=3)
    assert p.begin(True) is False                 # first visit: the buffer holds no cuts yet
    assert p.begin(True) is False                 # rendered again without a backward in between: still none
    p.renewed()                                   # the loss / backward of that forward has been queued
    assert p.begin(True) is True
    p.report(missed=False, kept=100, full=5000)   # 4900 left out: worth it
    p.renewed()
    assert p.begin(True) is True
    p.report(missed=True, kept=10, full=5000)     # a miss says nothing about the savings
    assert p.begin(True) is True
    p.report(missed=False, kept=4500, full=5000)  # 500 left out: sit out
    assert [p.begin(True) for _ in range(3)] == [False, False, False]
    assert p.begin(True) is True                  # probe again
    p.report(missed=False, kept=0, full=5000)
    p.cuts_cleared()                              # densify: the cuts are gone
    assert p.begin(True) is False
    p.renewed()
    assert p.begin(True) is True
    assert p.begin(False) is False                # the caller asked for an uncut forward: no bookkeeping
    assert p.begin(True) is True


def test_arena_segments_start_on_16_byte_boundaries_for_any_count():
    """trainer.GaussModel: every tensor's slice of the flat arena is padded to a multiple of four floats, whatever N is
    (a densify event leaves an odd count as often as not): the fused kernels take the float4 / kept-in-registers path
    for the SH rows only on 16-byte boundaries.  The pads are zero in all four arenas and take part in nothing."""
    import numpy as np
    import torch
    from gaussiansplattingmlx_amd.trainer import ARENA_ORDER, GaussModel
    rng = np.random.default_rng(0)
    for N in (1, 2, 7, 301):
        p = dict(xyz=rng.normal(size=(N, 3)), scales=rng.normal(size=(N, 3)), rotation=rng.normal(size=(N, 4)),
                 opacity=rng.normal(size=N), features_dc=rng.normal(size=(N, 1, 3)), features_rest=rng.normal(size=(N, 24, 3)))
        p = {k: v.astype(np.float32) for k, v in p.items()}
        m = GaussModel(p, torch.device("cpu"), capacity=N + 3)
        base = m.arena.data_ptr()
        covered = np.zeros(m.numel, bool)
        for k in ARENA_ORDER:
            v = m.getParams()[k]
            assert (v.data_ptr() - base) % 16 == 0, (N, k)
            np.testing.assert_array_equal(v.numpy().reshape(p[k].shape), p[k])
            off = (v.data_ptr() - base) // 4
            covered[off:off + v.numel()] = True
            assert (m.getGrads()[k].data_ptr() - m.grad.data_ptr()) == (v.data_ptr() - base)
        assert m.numel % 4 == 0 and int(m.seg_end[-1]) == m.numel and m.geom_numel == int(m.seg_end[3])
        assert not m.arena.numpy()[~covered].any() and (~covered).sum() == m.numel - N * 86
        # a densify-style flip to another count keeps the property
        q = {k: np.concatenate([v, v[:1]], 0) for k, v in p.items()}
        m.commit(q)
        for k in ARENA_ORDER:
            assert (m.getParams()[k].data_ptr() - m.arena.data_ptr()) % 16 == 0
            np.testing.assert_array_equal(m.getParams()[k].numpy().reshape(q[k].shape), q[k])


def _block_pixels(nbx, nby, tw, th, bx, by, W, H):
    """gs_block_pixels (csrc/gs_ctx.h) restated: pixel origin and exclusive limits of block (bx, by) of the block-list grid."""
    tx, ix = divmod(bx, nbx)
    ty, iy = divmod(by, nby)
    X0, Y0 = tx * tw + ix * 16, ty * th + iy * 16
    return X0, Y0, min(W, X0 + 16, (tx + 1) * tw), min(H, Y0 + 16, (ty + 1) * th)


@pytest.mark.parametrize("W,H,tw,th", [(800, 800, 200, 200), (1237, 822, 310, 206), (250, 170, 100, 70), (130, 100, 24, 40),
                                       (97, 61, 200, 200), (64, 48, 7, 33), (200, 152, 50, 38)])
def test_block_list_geometry_partitions_the_image(W, H, tw, th):
    """Block lists (include/gsplat.h, gs_ctx_create): tiles whose size is not a multiple of 16 are cut into 16 x 16 blocks
    enumerated PER TILE, the last column / row of a tile narrower.  The rule the kernels use (gs_block_pixels), restated:
    every pixel of the image lies in exactly one block, no block straddles two tiles, and a pixel's block column follows from
    its x alone as (x div tw) nbx + (x mod tw) div 16 -- what block_rect_of_splat (gs_math.h) relies on."""
    nbx, nby = -(-tw // 16), -(-th // 16)
    gridW, gridH = -(-W // tw), -(-H // th)
    cover = np.zeros((H, W), np.int32)
    for by in range(gridH * nby):
        for bx in range(gridW * nbx):
            X0, Y0, XL, YL = _block_pixels(nbx, nby, tw, th, bx, by, W, H)
            if XL <= X0 or YL <= Y0:
                continue                                # (a block beyond the image, or the empty remainder of a last tile)
            assert XL - X0 <= 16 and YL - Y0 <= 16
            assert X0 // tw == (XL - 1) // tw and Y0 // th == (YL - 1) // th, "a block lies in one tile"
            cover[Y0:YL, X0:XL] += 1
            xs, ys = np.arange(X0, XL), np.arange(Y0, YL)
            assert ((xs // tw) * nbx + (xs % tw) // 16 == bx).all() and ((ys // th) * nby + (ys % th) // 16 == by).all()
    assert (cover == 1).all()
