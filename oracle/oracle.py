"""numpy/ctypes front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY -- see the header of gs_oracle.c.  Importable from
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force: bool = False) -> None:
    """Compile the float32 and float64 oracle libraries with gcc."""
    src = os.path.join(_HERE, "gs_oracle.c")
    outs = [os.path.join(_BUILD, "libgs_oracle.so"), os.path.join(_BUILD, "libgs_oracle64.so")]
    if not force and all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs):
        return
    subprocess.check_call(["make", "-C", _HERE, "-B", "all"], stdout=subprocess.DEVNULL)


def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def build_native() -> str:
    """The -O3 -march=native build that bench.py's cpu_baseline times (float32 only).  -march=native code must run on
    the CPU it was built on, so the library is rebuilt whenever the recorded CPU model differs from this machine's."""
    src = os.path.join(_HERE, "gs_oracle.c")
    out = os.path.join(_BUILD, "libgs_oracle_native.so")
    stamp = out + ".cpu"
    cpu = _cpu_model()
    fresh = (os.path.exists(out) and os.path.exists(stamp) and open(stamp).read() == cpu
             and os.path.getmtime(out) >= os.path.getmtime(src))
    if not fresh:
        subprocess.check_call(["make", "-C", _HERE, "-B", "native"], stdout=subprocess.DEVNULL)
        with open(stamp, "w") as f:
            f.write(cpu)
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


@dataclass
class Binning:
    M: int
    B: int
    tilesTouched: np.ndarray
    offsets: np.ndarray
    keysHigh: np.ndarray
    keysLow: np.ndarray
    gaussIdx: np.ndarray
    sortedHigh: np.ndarray
    sortedLow: np.ndarray
    sortedIdx: np.ndarray
    tileRanges: np.ndarray
    tileCounts: np.ndarray


class Oracle:
    def __init__(self, dtype=np.float32, native: bool = False):
        build()
        self.dtype = np.dtype(dtype)
        if native and self.dtype != np.float32:
            raise ValueError("the native (timed) build exists in float32 only")
        if self.dtype == np.float32:
            self.lib = C.CDLL(build_native() if native else os.path.join(_BUILD, "libgs_oracle.so"))
            self.pre = "gso_"
            self.creal = C.c_float
        elif self.dtype == np.float64:
            self.lib = C.CDLL(os.path.join(_BUILD, "libgs_oracle64.so"))
            self.pre = "gso64_"
            self.creal = C.c_double
        else:
            raise ValueError("dtype must be float32 or float64")

    # -- helpers ---------------------------------------------------------
    def _f(self, name):
        return getattr(self.lib, self.pre + name)

    def _r(self, a):
        return np.ascontiguousarray(a, dtype=self.dtype)

    @staticmethod
    def _u(a):
        return np.ascontiguousarray(a, dtype=np.uint32)

    # -- a1 --------------------------------------------------------------
    def camera_build(self, c2w, focalX, focalY, W, H, znear=0.1, zfar=100.0):
        c2w = np.ascontiguousarray(c2w, dtype=np.float64).reshape(16)
        view = np.zeros(16, np.float32)
        proj = np.zeros(16, np.float32)
        fovX, fovY = C.c_float(), C.c_float()
        cam = np.zeros(3, np.float32)
        fn = self.lib.gso_camera_build
        fn.restype = C.c_int
        rc = fn(_ptr(c2w), C.c_float(focalX), C.c_float(focalY), C.c_int(W), C.c_int(H), C.c_double(znear),
                C.c_double(zfar), _ptr(view), _ptr(proj), C.byref(fovX), C.byref(fovY), _ptr(cam))
        if rc != 0:
            raise ValueError("singular c2w")
        return view.reshape(4, 4), proj.reshape(4, 4), fovX.value, fovY.value, cam

    # -- a2 --------------------------------------------------------------
    def activations_forward(self, opacity_raw, scales_raw, rot_raw):
        o, s, q = self._r(opacity_raw).reshape(-1), self._r(scales_raw), self._r(rot_raw)
        N = o.shape[0]
        op, sc, rt = np.empty_like(o), np.empty_like(s), np.empty_like(q)
        self._f("activations_forward")(C.c_int(N), _ptr(o), _ptr(s), _ptr(q), _ptr(op), _ptr(sc), _ptr(rt))
        return op, sc, rt

    def activations_backward(self, opacity_raw, scales_raw, rot_raw, gOp, gSc, gRot):
        o, s, q = self._r(opacity_raw).reshape(-1), self._r(scales_raw), self._r(rot_raw)
        gOp, gSc, gRot = self._r(gOp).reshape(-1), self._r(gSc), self._r(gRot)
        N = o.shape[0]
        do, ds, dq = np.empty_like(o), np.empty_like(s), np.empty_like(q)
        self._f("activations_backward")(C.c_int(N), _ptr(o), _ptr(s), _ptr(q), _ptr(gOp), _ptr(gSc), _ptr(gRot),
                                        _ptr(do), _ptr(ds), _ptr(dq))
        return do, ds, dq

    # -- a3 / a4 ---------------------------------------------------------
    def projection_forward(self, scales, rot, means3d, shs, camCenter, view, proj, fovX, fovY, focalX, focalY,
                           W, H, degree):
        scales, rot, means3d, shs = self._r(scales), self._r(rot), self._r(means3d), self._r(shs)
        N, K = means3d.shape[0], shs.shape[1]
        cam, V, P = self._r(camCenter).reshape(3), self._r(view).reshape(16), self._r(proj).reshape(16)
        r = self.creal
        out = dict(means2d=np.zeros((N, 2), self.dtype), depths=np.zeros(N, self.dtype),
                   color=np.zeros((N, 3), self.dtype), cov2d=np.zeros((N, 2, 2), self.dtype),
                   conic=np.zeros((N, 2, 2), self.dtype), radii=np.zeros(N, self.dtype),
                   rectMin=np.zeros((N, 2), self.dtype), rectMax=np.zeros((N, 2), self.dtype))
        self._f("projection_forward")(C.c_int(N), C.c_int(K), C.c_int(degree), _ptr(scales), _ptr(rot),
                                      _ptr(means3d), _ptr(shs), _ptr(cam), _ptr(V), _ptr(P), r(fovX), r(fovY),
                                      r(focalX), r(focalY), r(W), r(H), *[_ptr(out[k]) for k in
                                      ("means2d", "depths", "color", "cov2d", "conic", "radii", "rectMin", "rectMax")])
        return out

    def projection_backward(self, scales, rot, means3d, shs, camCenter, view, proj, fovX, fovY, focalX, focalY,
                            W, H, degree, cotDepths, cotMeans2d, cotCov2d, cotColor, cotConic):
        scales, rot, means3d, shs = self._r(scales), self._r(rot), self._r(means3d), self._r(shs)
        N, K = means3d.shape[0], shs.shape[1]
        cam, V, P = self._r(camCenter).reshape(3), self._r(view).reshape(16), self._r(proj).reshape(16)
        cd, cm, cc, ccol, ccon = (self._r(cotDepths), self._r(cotMeans2d), self._r(cotCov2d), self._r(cotColor),
                                  self._r(cotConic))
        r = self.creal
        out = dict(gradScales=np.zeros((N, 3), self.dtype), gradRot=np.zeros((N, 4), self.dtype),
                   gradMeans3d=np.zeros((N, 3), self.dtype), gradShs=np.zeros((N, K, 3), self.dtype),
                   gradCamCenterPoint=np.zeros((N, 3), self.dtype))
        self._f("projection_backward")(C.c_int(N), C.c_int(K), C.c_int(degree), _ptr(scales), _ptr(rot),
                                       _ptr(means3d), _ptr(shs), _ptr(cam), _ptr(V), _ptr(P), r(fovX), r(fovY),
                                       r(focalX), r(focalY), r(W), r(H), _ptr(cd), _ptr(cm), _ptr(cc), _ptr(ccol),
                                       _ptr(ccon), *[_ptr(out[k]) for k in
                                       ("gradScales", "gradRot", "gradMeans3d", "gradShs", "gradCamCenterPoint")])
        return out

    # -- a5 --------------------------------------------------------------
    def pack_gaussians(self, means2d, conic, color, opacity, depths):
        means2d, conic, color = self._r(means2d), self._r(conic).reshape(-1, 4), self._r(color)
        opacity, depths = self._r(opacity).reshape(-1), self._r(depths).reshape(-1)
        N = means2d.shape[0]
        packed = np.zeros((N, 11), self.dtype)
        self._f("pack_gaussians")(C.c_int(N), _ptr(means2d), _ptr(conic), _ptr(color), _ptr(opacity), _ptr(depths),
                                  _ptr(packed))
        return packed

    # -- a6 --------------------------------------------------------------
    def tile_bin(self, rectMin, rectMax, radii, depths, W, H, tileW, tileH) -> Binning:
        rectMin, rectMax, radii, depths = self._r(rectMin), self._r(rectMax), self._r(radii), self._r(depths)
        N = radii.shape[0]
        gridW, gridH = (W + tileW - 1) // tileW, (H + tileH - 1) // tileH
        T = gridW * gridH
        touched = np.zeros(N, np.uint32)
        self._f("count_tiles")(C.c_int(N), C.c_int(tileW), C.c_int(tileH), C.c_int(W), C.c_int(H), _ptr(rectMin),
                               _ptr(rectMax), _ptr(radii), _ptr(touched))
        offsets = np.zeros(N, np.uint32)
        scan = self.lib.gso_exclusive_scan
        scan.restype = C.c_uint32
        M = int(scan(C.c_int(N), _ptr(touched), _ptr(offsets)))
        kh, kl, gi = (np.zeros(M, np.uint32) for _ in range(3))
        self._f("generate_keys")(C.c_int(N), C.c_int(tileW), C.c_int(tileH), C.c_int(W), C.c_int(H), _ptr(depths),
                                 _ptr(rectMin), _ptr(rectMax), _ptr(radii), _ptr(offsets), _ptr(kh), _ptr(kl),
                                 _ptr(gi))
        sh, sl, sv = (np.zeros(M, np.uint32) for _ in range(3))
        self.lib.gso_sort_pairs(C.c_uint32(M), _ptr(kh), _ptr(kl), _ptr(gi), _ptr(sh), _ptr(sl), _ptr(sv))
        ranges = np.zeros((T, 2), np.uint32)
        self.lib.gso_tile_ranges(C.c_uint32(M), C.c_uint32(T), _ptr(sh), _ptr(ranges))
        counts = np.zeros(T, np.uint32)
        tc = self.lib.gso_tile_counts
        tc.restype = C.c_uint32
        B = int(tc(C.c_uint32(T), _ptr(ranges), _ptr(counts)))
        return Binning(M, B, touched, offsets, kh, kl, gi, sh, sl, sv, ranges, counts)

    def build_packed_tile_indices(self, sortedIdx, tileRanges, B):
        T = tileRanges.shape[0]
        out = np.zeros((T, max(B, 0)), np.int32)
        if B > 0:
            self.lib.gso_build_packed_tile_indices(C.c_uint32(T), C.c_uint32(B), _ptr(self._u(sortedIdx)),
                                                   _ptr(self._u(tileRanges)), _ptr(out))
        return out

    # -- a7 / a8 ---------------------------------------------------------
    def blend_forward(self, packed, sortedIdx, tileRanges, W, H, tileW, tileH, whiteBg):
        packed, sortedIdx, tileRanges = self._r(packed), self._u(sortedIdx), self._u(tileRanges)
        Pn = W * H
        color, depth, alpha = np.zeros((Pn, 3), self.dtype), np.zeros(Pn, self.dtype), np.zeros(Pn, self.dtype)
        last = np.zeros(Pn, np.uint32)
        self._f("blend_forward")(C.c_int(W), C.c_int(H), C.c_int(tileW), C.c_int(tileH), C.c_int(int(whiteBg)),
                                 _ptr(packed), _ptr(sortedIdx), _ptr(tileRanges), _ptr(color), _ptr(depth),
                                 _ptr(alpha), _ptr(last))
        return color, depth, alpha, last

    def blend_backward(self, packed, sortedIdx, tileRanges, W, H, tileW, tileH, whiteBg, cotColor, cotDepth,
                       cotAlpha, outColor, outDepth, outAlpha, lastContrib):
        packed, sortedIdx, tileRanges = self._r(packed), self._u(sortedIdx), self._u(tileRanges)
        N = packed.shape[0]
        grad = np.zeros((N, 11), self.dtype)
        self._f("blend_backward")(C.c_int(N), C.c_int(W), C.c_int(H), C.c_int(tileW), C.c_int(tileH),
                                  C.c_int(int(whiteBg)), _ptr(packed), _ptr(sortedIdx), _ptr(tileRanges),
                                  _ptr(self._r(cotColor)), _ptr(self._r(cotDepth)), _ptr(self._r(cotAlpha)),
                                  _ptr(self._r(outColor)), _ptr(self._r(outDepth)), _ptr(self._r(outAlpha)),
                                  _ptr(self._u(lastContrib)), _ptr(grad))
        return grad

    # -- a10 / a11 -------------------------------------------------------
    def ssim_window(self, K=11, sigma=1.5):
        w = np.zeros(K * K, np.float32)
        self.lib.gso_ssim_window(C.c_int(K), C.c_float(sigma), _ptr(w))
        return w

    def ssim_forward(self, img1, img2, window=None, K=11):
        img1, img2 = self._r(img1), self._r(img2)
        H, W, Cc = img1.shape
        window = self._r(self.ssim_window(K) if window is None else window)
        outs = [np.zeros((H, W, Cc), self.dtype) for _ in range(6)]
        self._f("ssim_forward")(C.c_int(H), C.c_int(W), C.c_int(Cc), C.c_int(K), _ptr(img1), _ptr(img2),
                                _ptr(window), *[_ptr(o) for o in outs])
        return outs  # ssim, mu1, mu2, sigma1, sigma2, sigma12

    def ssim_backward(self, gradOut, img1, img2, saved, window=None, K=11):
        img1, img2, gradOut = self._r(img1), self._r(img2), self._r(gradOut)
        H, W, Cc = img1.shape
        window = self._r(self.ssim_window(K) if window is None else window)
        g1, g2 = np.zeros_like(img1), np.zeros_like(img2)
        mu1, mu2, s1, s2, s12 = [self._r(a) for a in saved]
        self._f("ssim_backward")(C.c_int(H), C.c_int(W), C.c_int(Cc), C.c_int(K), _ptr(gradOut), _ptr(img1),
                                 _ptr(img2), _ptr(window), _ptr(mu1), _ptr(mu2), _ptr(s1), _ptr(s2), _ptr(s12),
                                 _ptr(g1), _ptr(g2))
        return g1, g2

    def loss_forward_backward(self, render, target, lambdaDssim=0.2, renderDepth=None, targetDepth=None,
                              depthMask=None, lambdaDepth=0.0):
        render, target = self._r(render), self._r(target)
        H, W, _ = render.shape
        cotColor = np.zeros_like(render)
        cotDepth = np.zeros((H, W), self.dtype)
        l1, ss = C.c_double(), C.c_double()
        fn = self._f("loss_forward_backward")
        fn.restype = C.c_double
        rd = None if renderDepth is None else self._r(renderDepth)
        td = None if targetDepth is None else self._r(targetDepth)
        dm = None if depthMask is None else np.ascontiguousarray(depthMask, dtype=np.uint8)
        loss = fn(C.c_int(H), C.c_int(W), _ptr(render), _ptr(target), _ptr(rd), _ptr(td), _ptr(dm),
                  self.creal(lambdaDssim), self.creal(lambdaDepth), _ptr(cotColor), _ptr(cotDepth), C.byref(l1),
                  C.byref(ss))
        return float(loss), cotColor, cotDepth, float(l1.value), float(ss.value)

    # -- next row f2: densify / prune (GaussianTrainer.swift:317-427, 724-908) -----
    def accum_grad_norm(self, xyzGrad, accumIn=None):
        g = self._r(xyzGrad).reshape(-1, 3)
        a = None if accumIn is None else self._r(accumIn)
        out = np.empty(g.shape[0], self.dtype)
        self._f("accum_grad_norm")(C.c_int(g.shape[0]), _ptr(g), _ptr(a), _ptr(out))
        return out

    def classify_gaussians(self, gradAccum, denom, scales, opacity, gradThreshold=0.0002, maxScale=0.01,
                           minOpacity=0.005, allowDensify=True):
        ga, sc, op = self._r(gradAccum), self._r(scales), self._r(opacity).reshape(-1)
        N = ga.shape[0]
        actions, counts = np.empty(N, np.int32), np.empty(N, np.int32)
        r = self.creal
        self._f("classify_gaussians")(C.c_int(N), _ptr(ga), r(denom), _ptr(sc), C.c_int(sc.shape[1]), _ptr(op),
                                      r(gradThreshold), r(maxScale), r(minOpacity), C.c_int(int(allowDensify)),
                                      _ptr(actions), _ptr(counts))
        return actions, counts

    def densify_offsets(self, actions, counts):
        a, c = np.ascontiguousarray(actions, np.int32), np.ascontiguousarray(counts, np.int32)
        off, stats = np.empty(a.shape[0], np.int32), np.zeros(5, np.int64)
        self.lib.gso_densify_offsets(C.c_int(a.shape[0]), _ptr(a), _ptr(c), _ptr(off), _ptr(stats))
        return off, dict(zip(("total", "keep", "split", "clone", "prune"), (int(x) for x in stats)))

    def build_densify_output_map(self, actions, offsets, total):
        a, o = np.ascontiguousarray(actions, np.int32), np.ascontiguousarray(offsets, np.int32)
        g, m = np.zeros(total, np.int32), np.zeros(total, np.int32)
        self.lib.gso_build_densify_output_map(C.c_int(a.shape[0]), _ptr(a), _ptr(o), _ptr(g), _ptr(m))
        return g, m

    def densify_gather(self, params, gather, noiseMode, baseNoise):
        """params: the six raw tensors; returns the six gathered / modified tensors (phases 4-5)."""
        p = {k: self._r(v) for k, v in params.items()}
        total, K = int(len(gather)), p["features_rest"].shape[1] + 1
        g, m = np.ascontiguousarray(gather, np.int32), np.ascontiguousarray(noiseMode, np.int32)
        nz = None if baseNoise is None else self._r(baseNoise)
        out = dict(xyz=np.empty((total, 3), self.dtype), features_dc=np.empty((total, 1, 3), self.dtype),
                   features_rest=np.empty((total, K - 1, 3), self.dtype), scales=np.empty((total, 3), self.dtype),
                   rotation=np.empty((total, 4), self.dtype),
                   opacity=np.empty((total,) + p["opacity"].shape[1:], self.dtype))
        self._f("densify_gather")(C.c_int(total), C.c_int(K), _ptr(p["xyz"]), _ptr(p["features_dc"]),
                                  _ptr(p["features_rest"]), _ptr(p["scales"]), _ptr(p["rotation"]), _ptr(p["opacity"]),
                                  _ptr(g), _ptr(m), _ptr(nz), _ptr(out["xyz"]), _ptr(out["features_dc"]),
                                  _ptr(out["features_rest"]), _ptr(out["scales"]), _ptr(out["rotation"]),
                                  _ptr(out["opacity"]))
        return out

    def split_and_prune(self, params, gradAccum, denom, baseNoiseFn, allowDensify=True, gradThreshold=0.0002,
                        maxScale=0.01, minOpacity=0.005):
        """GaussianTrainer.swift:766-907 end to end.  baseNoiseFn(total) -> [total,3] standard normal.  Returns
        (new params or None when nothing changes, stats)."""
        actions, counts = self.classify_gaussians(gradAccum, denom, params["scales"], params["opacity"], gradThreshold,
                                                  maxScale, minOpacity, allowDensify)
        offsets, st = self.densify_offsets(actions, counts)
        if st["total"] == 0 or (st["split"] == 0 and st["clone"] == 0 and st["prune"] == 0):
            return None, st
        g, m = self.build_densify_output_map(actions, offsets, st["total"])
        noise = baseNoiseFn(st["total"]) if (st["split"] > 0 or st["clone"] > 0) else None
        return self.densify_gather(params, g, m, noise), st

    # -- composed reference pipeline (GaussianTrainer.swift:652-716) -----
    def render_forward(self, params, cam, W, H, tileW, tileH, degree, whiteBg=False):
        """params: dict(xyz, features_dc, features_rest, scales, rotation, opacity) raw; cam: dict(view, proj,
        fovX, fovY, focalX, focalY, camCenter).  Returns every intermediate."""
        op, sc, rt = self.activations_forward(params["opacity"], params["scales"], params["rotation"])
        shs = np.concatenate([self._r(params["features_dc"]), self._r(params["features_rest"])], axis=1)
        pr = self.projection_forward(sc, rt, params["xyz"], shs, cam["camCenter"], cam["view"], cam["proj"],
                                     cam["fovX"], cam["fovY"], cam["focalX"], cam["focalY"], W, H, degree)
        packed = self.pack_gaussians(pr["means2d"], pr["conic"], pr["color"], op, pr["depths"])
        bn = self.tile_bin(pr["rectMin"], pr["rectMax"], pr["radii"], pr["depths"], W, H, tileW, tileH)
        color, depth, alpha, last = self.blend_forward(packed, bn.sortedIdx, bn.tileRanges, W, H, tileW, tileH, whiteBg)
        return dict(opacity=op, scales=sc, rot=rt, shs=shs, proj=pr, packed=packed, bin=bn, color=color,
                    depth=depth, alpha=alpha, last=last)

    def render_backward(self, params, cam, W, H, tileW, tileH, degree, fwd, cotColor, cotDepth, cotAlpha,
                        whiteBg=False):
        bn = fwd["bin"]
        gp = self.blend_backward(fwd["packed"], bn.sortedIdx, bn.tileRanges, W, H, tileW, tileH, whiteBg,
                                 cotColor, cotDepth, cotAlpha, fwd["color"], fwd["depth"], fwd["alpha"], fwd["last"])
        N = gp.shape[0]
        shs = fwd["shs"]
        pb = self.projection_backward(fwd["scales"], fwd["rot"], params["xyz"], shs, cam["camCenter"], cam["view"],
                                      cam["proj"], cam["fovX"], cam["fovY"], cam["focalX"], cam["focalY"], W, H,
                                      degree, gp[:, 10], gp[:, 0:2], np.zeros((N, 4), self.dtype), gp[:, 6:9],
                                      gp[:, 2:6])
        do, ds, dq = self.activations_backward(params["opacity"], params["scales"], params["rotation"],
                                               gp[:, 9], pb["gradScales"], pb["gradRot"])
        return dict(xyz=pb["gradMeans3d"], features_dc=pb["gradShs"][:, :1, :].copy(),
                    features_rest=pb["gradShs"][:, 1:, :].copy(), scales=ds, rotation=dq,
                    opacity=do.reshape(np.shape(params["opacity"])), gradPacked=gp)

    # -- test hooks ------------------------------------------------------
    def cov3d(self, s, q):
        s, q = self._r(s).reshape(3), self._r(q).reshape(4)
        out, rot = np.zeros(9, self.dtype), np.zeros(9, self.dtype)
        self._f("cov3d")(_ptr(s), _ptr(q), _ptr(out), _ptr(rot))
        return out.reshape(3, 3), rot.reshape(3, 3)

    def sh_basis(self, degree, x, y, z):
        b = np.zeros(25, self.dtype)
        r = self.creal
        self._f("sh_basis")(C.c_int(degree), r(x), r(y), r(z), _ptr(b))
        return b
