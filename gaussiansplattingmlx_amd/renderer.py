"""Host-side mirror of the reference's `GaussianRenderer` (Trainer/GaussianRenderer.swift) over the C ABI.

Same constructor, method names, argument meaning and result tuple as the reference class; every device
operation goes through libgsplat_hip.so (hand-written HIP).  torch is used only to own device memory and
the stream.  There is no CPU path: constructing a renderer without the library or a GPU raises, as the
reference's init preconditions do (GaussianRenderer.swift:721-733).
"""
from __future__ import annotations

import ctypes as C
import os
from collections import OrderedDict, namedtuple

import numpy as np
import torch

from . import _lib
from ._lib import GsplatError

TILE_SIZE_H_W = namedtuple("TILE_SIZE_H_W", ["w", "h"])
RenderResult = namedtuple("RenderResult", ["render", "depth", "alpha", "visiility_filter", "radii"])


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t, device):
    if isinstance(t, torch.Tensor):
        return t.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(t, dtype=np.float32), device=device)


class CutPolicy:
    """When does a training view bin under depth cuts (gs_set_view_hints)?  The cuts cost a fixed ~40 us per forward
    (second expansion pass, the host's wait for the forward, an occasional repeated forward) and save ~13 us per
    million pairs they leave out (MI355X, 300 k Gaussians at 800x800: 4.4 M left out = break-even; round 5: 7.3 M left out on
    the same scene at 200 x 200 tiles = +1.5 %, hence the default of 6 M).  So: never on
    a forward while the view's cuts are empty -- they are written by the backward's preparation (loss or backward of
    a forward of the view), so a view that has only been rendered has none -- and a view whose cuts left out fewer than
    min_dropped pairs sits out the next probe_interval forwards, then is tried again."""

    def __init__(self, min_dropped: int = 6_000_000, probe_interval: int = 64):
        self.min_dropped, self.probe_interval = min_dropped, probe_interval
        self.sit_out = 0            # forwards still to run without cuts
        self.last_dropped = 0       # pairs the last cut forward left out
        self.since_empty = 0        # backward preparations since the view's cuts were last empty (0: they are empty now)

    def begin(self, allowed: bool) -> bool:
        """Called once per forward of the view; True = run this one under cuts."""
        use = allowed and self.sit_out == 0 and self.since_empty >= 1
        if allowed and self.sit_out > 0:
            self.sit_out -= 1
        return use

    def renewed(self):
        """The backward's preparation of a forward of this view has been queued: the view's cuts exist from now on."""
        self.since_empty += 1

    def report(self, missed: bool, kept: int, full: int):
        """After a forward under cuts: kept / full pairs (gs_cut_stats)."""
        self.last_dropped = max(int(full) - int(kept), 0)
        if not missed and full > 0 and self.last_dropped < self.min_dropped:
            self.sit_out = self.probe_interval

    def cuts_cleared(self):
        self.since_empty = 0


class GaussianRenderer:
    def __init__(self, active_sh_degree: int, W: int, H: int, TILE_SIZE=TILE_SIZE_H_W(16, 16),
                 whiteBackground: bool = False, useScreenSpaceCustomOp: bool = True, device: int = 0):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("GaussianRenderer needs a GPU: the HIP path has no CPU fallback")
        self.active_sh_degree, self.W, self.H = int(active_sh_degree), int(W), int(H)
        self.TILE_SIZE = TILE_SIZE_H_W(*TILE_SIZE)
        self.whiteBackground = bool(whiteBackground)
        self.useScreenSpaceCustomOp = useScreenSpaceCustomOp
        self.device = torch.device("cuda", device)
        self.profiler = None          # an IntervalProfiler the trainer sets per profiled iteration (GaussianRenderer.swift:66-68)
        torch.cuda.set_device(self.device)
        ctx = C.c_void_p()
        rc = self.lib.gs_ctx_create(device, self.W, self.H, self.TILE_SIZE.w, self.TILE_SIZE.h,
                                    self.active_sh_degree, int(self.whiteBackground), C.byref(ctx))
        if rc != _lib.GS_OK:
            raise GsplatError(rc, "gs_ctx_create failed")
        self.ctx = ctx
        self._check(self.lib.gs_ctx_set_stream(self.ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        win = np.zeros(121, np.float32)
        self.lib.gs_ssim_window(11, C.c_float(1.5), win.ctypes.data_as(C.c_void_p))
        self.ssimWindow = torch.as_tensor(win, device=self.device)
        self._saved = {}
        self.reserved = None
        self._grad_norm_accum = None
        self._work_hints = {}
        self.depthCuts = True          # False: view hints order the forward's work but never cut the binning
        self._cut_policy = {}          # viewKey -> CutPolicy
        self._cut_view = None
        self._hinted_view = None
        self.cutMinDropped = 6_000_000      # (c3 at 16 x 16 tiles leaves out 4 - 5 M: break-even, off; at the app's 200 x 200 tiles 7.3 M: +1.5 %; c5 80 M)
        self.cutProbeInterval = 64
        # 16 x 16 tiles, or a tile size that is not a multiple of 16 (the library then works on block lists: gs_ctx.h)
        self._hints_ok = (self.TILE_SIZE.w, self.TILE_SIZE.h) == (16, 16) or \
            ((self.TILE_SIZE.w % 16 != 0 or self.TILE_SIZE.h % 16 != 0) and os.environ.get("GSPLAT_BLOCK_LISTS", "1") != "0")
        self.targetStatsCache = True   # lossForwardBackward(targetKey=...) keeps the target's SSIM statistics per key
        self._target_cache = OrderedDict()
        self.targetStatsCacheBytes = 8 << 30      # cap of the per-view caches together (6 H W floats each); LRU beyond it

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.gs_set_block_work_buffer(self.ctx, None)
            self.lib.gs_set_loss_target_cache(self.ctx, None, 0)
            self.lib.gs_set_grad_norm_accum(self.ctx, None)
            self.lib.gs_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != _lib.GS_OK:
            raise GsplatError(rc, self.lib.gs_last_error(self.ctx).decode())

    def _t(self, t):
        return _f32(t, self.device)

    def _measure(self, name, body):
        """Runs body inside the profiler's section `name` when the trainer has set one (GaussianRenderer.swift:157-172,
        579-600), plainly otherwise."""
        return self.profiler.measure(name, body) if self.profiler is not None else body()

    def _empty(self, *shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    def reserve(self, max_gaussians: int, max_pairs: int):
        self._check(self.lib.gs_ctx_reserve(self.ctx, int(max_gaussians), int(max_pairs)))
        self.reserved = (int(max_gaussians), int(max_pairs))

    def sync(self):
        self._check(self.lib.gs_sync(self.ctx))

    _TUNING = dict(fwd_waves_per_simd=_lib.TUNE_FWD_WAVES_PER_SIMD, bwd_waves_per_cu=_lib.TUNE_BWD_WAVES_PER_CU,
                   fwd_quadrants=_lib.TUNE_FWD_QUADRANTS, op_fwd_ppl=_lib.TUNE_OP_FWD_PPL,
                   op_bwd_ppl=_lib.TUNE_OP_BWD_PPL, fwd_trace_buffer=_lib.TUNE_FWD_TRACE_BUFFER,
                   depth_gradient=_lib.TUNE_DEPTH_GRADIENT, wide_tile_sort=_lib.TUNE_WIDE_TILE_SORT,
                   host_overflow_errors=_lib.TUNE_HOST_OVERFLOW_ERRORS, splitter_depth_sort=_lib.TUNE_SPLITTER_DEPTH_SORT,
                   colour_riders=_lib.TUNE_COLOUR_RIDERS, fwd_queues=_lib.TUNE_FWD_QUEUES, fwd_four_waves=_lib.TUNE_FWD_FOUR_WAVES,
                   fwd_fold_test_scale=_lib.TUNE_FWD_FOLD_TEST_SCALE, poison_checkpoints=_lib.TUNE_POISON_CHECKPOINTS,
                   render_only=_lib.TUNE_RENDER_ONLY, fwd_pair=_lib.TUNE_FWD_PAIR, fwd_slow_slot=_lib.TUNE_FWD_SLOW_SLOT,
                   trim_rects=_lib.TUNE_TRIM_RECTS)
    _TUNING_DEFAULTS = dict(fwd_waves_per_simd=4, bwd_waves_per_cu=16, fwd_quadrants=1, op_fwd_ppl=1, op_bwd_ppl=1,
                            fwd_trace_buffer=0, depth_gradient=1, wide_tile_sort=1, host_overflow_errors=1, splitter_depth_sort=1,
                            colour_riders=1, fwd_queues=8, fwd_four_waves=-1, fwd_fold_test_scale=1000, poison_checkpoints=0, render_only=0, fwd_pair=-1, fwd_slow_slot=3, trim_rects=2)

    def setTuning(self, **knobs):
        """Launch tuning of THIS renderer's context (gs_ctx_set_tuning); results never depend on it (fwd_four_waves: within
        the parity bars, not to the bit -- sums composed across list chunks instead of accumulated)."""
        for k, v in knobs.items():
            self._check(self.lib.gs_ctx_set_tuning(self.ctx, self._TUNING[k], int(v)))
            self._tuning_now = {**getattr(self, "_tuning_now", {}), k: int(v)}

    def getTuning(self, knob: str) -> int:
        """The value this renderer last set for a knob (the library's default if it never did)."""
        return getattr(self, "_tuning_now", {}).get(knob, self._TUNING_DEFAULTS[knob])

    def colourRidersActive(self, N: int, K: int = 25) -> bool:
        """Do the fused forwards of N Gaussians (from this context's second one on) compute their SH colours as rider workgroups of
        the binning kernels (csrc/projection.hip: K = 25, GS_TUNE_COLOUR_RIDERS = 1 and a depth sort that takes the splitter
        buckets -- 16385 .. 655 360 records by default, binning.hip ss_fits)?  bench.py: the projection stage's time then holds
        the geometry half only."""
        if K != 25 or self.getTuning("colour_riders") != 1 or ((self.TILE_SIZE.w % 16 or self.TILE_SIZE.h % 16) and not self.blockLists):
            return False
        split = self.getTuning("splitter_depth_sort")
        return bool(split) and N > 16384 and (N <= 160 * 4096 or (split >= 2 and N <= 1024 * 4096))

    def stats(self):
        s = (C.c_uint32 * 8)()
        self._check(self.lib.gs_last_stats(self.ctx, s))
        return dict(N_visible=s[0], M=s[1], max_tile_list=s[2], overflow=s[5], capN=s[6], capM=s[7])

    @staticmethod
    def _camera(viewMatrix, projMatrix, cameraCenter, fovX, fovY, focalX, focalY):
        g = lambda a: a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a
        return _lib.make_camera(g(viewMatrix), g(projMatrix), g(cameraCenter), float(g(fovX)), float(g(fovY)),
                                float(g(focalX)), float(g(focalY)))

    # -- activations (GaussianRenderer.swift:936-963) ------------------------------------------------
    def get_scales_from(self, scales):
        return torch.exp(scales)

    def get_rotation_from(self, rotation):
        return rotation / (torch.sqrt(torch.sum(rotation * rotation, dim=1, keepdim=True)) + 1e-8)

    def get_xyz_from(self, xyz):
        return xyz

    def get_features_from(self, features_dc, features_rest):
        return torch.cat([features_dc, features_rest], dim=1)

    def get_opacity_from(self, opacity):
        return torch.sigmoid(opacity)

    # -- projection custom function (GaussianRenderer.swift:494-603) -----------------------------------
    def projectionScreenFused(self, scales, rotations, means3d, shs, cam):
        scales, rotations, means3d, shs = map(self._t, (scales, rotations, means3d, shs))
        N, K = means3d.shape[0], shs.shape[1]
        out = dict(means2d=self._empty(N, 2), depths=self._empty(N), color=self._empty(N, 3),
                   cov2d=self._empty(N, 2, 2), conic=self._empty(N, 2, 2), radii=self._empty(N),
                   rectMin=self._empty(N, 2), rectMax=self._empty(N, 2))
        self._check(self.lib.gs_projection_forward(
            self.ctx, N, K, _p(scales), _p(rotations), _p(means3d), _p(shs), C.byref(cam), _p(out["means2d"]),
            _p(out["depths"]), _p(out["color"]), _p(out["cov2d"]), _p(out["conic"]), _p(out["radii"]),
            _p(out["rectMin"]), _p(out["rectMax"])))
        return out

    def projectionScreenFusedVJP(self, scales, rotations, means3d, shs, cam, cotMeans2d, cotDepths, cotColor,
                                 cotCov2d, cotConic):
        scales, rotations, means3d, shs = map(self._t, (scales, rotations, means3d, shs))
        cotMeans2d, cotDepths, cotColor, cotCov2d, cotConic = map(self._t, (cotMeans2d, cotDepths, cotColor,
                                                                            cotCov2d, cotConic))
        N, K = means3d.shape[0], shs.shape[1]
        out = dict(gradScales=self._empty(N, 3), gradRot=self._empty(N, 4), gradMeans3d=self._empty(N, 3),
                   gradShs=self._empty(N, K, 3), gradCamCenterPoint=self._empty(N, 3))
        self._measure("bwd.projectionScreenFused", lambda: self._check(self.lib.gs_projection_backward(
            self.ctx, N, K, _p(scales), _p(rotations), _p(means3d), _p(shs), C.byref(cam), _p(cotDepths),
            _p(cotMeans2d), _p(cotCov2d), _p(cotColor), _p(cotConic), _p(out["gradScales"]), _p(out["gradRot"]),
            _p(out["gradMeans3d"]), _p(out["gradShs"]), _p(out["gradCamCenterPoint"]))))
        out["gradCameraCenter"] = out["gradCamCenterPoint"].sum(dim=0, keepdim=True)   # :683-684
        return out

    # -- tile binning (GaussianRenderer.swift:333-490) -----------------------------------------------------
    def buildGlobalTileSliceInfo(self, rect, radii, depths, want_dense: bool = False, tileCuts=None):
        """tileCuts: optional u32 [T] per-tile depth cuts (gs_tile_bin_cut: 0 = none, else 0xFFFFFFFF - the largest depth
        key still binned in that tile); the lists are then prefixes of the reference's."""
        rmin, rmax, radii, depths = self._t(rect[0]), self._t(rect[1]), self._t(radii), self._t(depths)
        N = radii.shape[0]
        if tileCuts is None:
            self._check(self.lib.gs_tile_bin(self.ctx, N, _p(rmin), _p(rmax), _p(radii), _p(depths)))
        else:
            cuts = torch.as_tensor(np.ascontiguousarray(tileCuts, dtype=np.uint32).view(np.int32), device=self.device)
            self._check(self.lib.gs_tile_bin_cut(self.ctx, N, _p(rmin), _p(rmax), _p(radii), _p(depths), _p(cuts)))
        M, B = C.c_uint32(), C.c_uint32()
        self._check(self.lib.gs_tile_bin_info(self.ctx, C.byref(M), C.byref(B)))
        T = ((self.W + self.TILE_SIZE.w - 1) // self.TILE_SIZE.w) * ((self.H + self.TILE_SIZE.h - 1) // self.TILE_SIZE.h)
        info = dict(M=M.value, maxTilePairs=B.value, numTiles=T)
        info["sortedGaussIdx"] = self._empty(M.value, dtype=torch.int32)
        info["tileRanges"] = self._empty(T, 2, dtype=torch.int32)
        info["tileCounts"] = self._empty(T, dtype=torch.int32)
        self._check(self.lib.gs_tile_bin_export(self.ctx, _p(info["sortedGaussIdx"]), _p(info["tileRanges"]),
                                                _p(info["tileCounts"])))
        if want_dense:
            dense = torch.zeros((T, B.value), dtype=torch.int32, device=self.device)
            self._check(self.lib.gs_build_packed_tile_indices(self.ctx, B.value, _p(dense)))
            info["packedTileIndices"] = dense
        return info

    def buildPackedGaussians(self, means2d, conic, color, opacity, depths):
        means2d, conic, color, opacity, depths = map(self._t, (means2d, conic, color, opacity, depths))
        N = means2d.shape[0]
        packed = self._empty(N, 11)
        self._check(self.lib.gs_pack_gaussians(self.ctx, N, _p(means2d), _p(conic), _p(color), _p(opacity),
                                               _p(depths), _p(packed)))
        return packed

    # -- tile composite custom function (GaussianRenderer.swift:101-244) -----------------------------------
    def globalTileComposite(self, packedGaussians):
        packed = self._t(packedGaussians)
        P = self.W * self.H
        color, depth, alpha = self._empty(P, 3), self._empty(P), self._empty(P)
        last = self._empty(P, dtype=torch.int32)
        self._check(self.lib.gs_blend_forward(self.ctx, packed.shape[0], _p(packed), _p(color), _p(depth), _p(alpha),
                                              _p(last)))
        self._saved = dict(packed=packed, color=color, depth=depth, alpha=alpha, last=last)
        return color, depth, alpha

    def globalTileCompositeVJP(self, cotColor, cotDepth=None, cotAlpha=None, saved=None):
        s = saved or self._saved
        packed = s["packed"]
        cotColor = self._t(cotColor)
        cotDepth = None if cotDepth is None else self._t(cotDepth)
        cotAlpha = None if cotAlpha is None else self._t(cotAlpha)
        grad = self._empty(packed.shape[0], 11)
        self._measure("bwd.globalTileComposite", lambda: self._check(self.lib.gs_blend_backward(
            self.ctx, packed.shape[0], _p(packed), _p(cotColor), _p(cotDepth), _p(cotAlpha), _p(s["color"]),
            _p(s["depth"]), _p(s["alpha"]), _p(s["last"]), _p(grad))))
        return grad

    # -- render / forward (GaussianRenderer.swift:736-934) ---------------------------------------------------
    def render(self, imageWidth, imageHeight, means2d, cov2d, color, opacity, depths, radii, conic, rect,
               inputIsDepthSorted: bool = False):
        if imageWidth != self.W or imageHeight != self.H:
            raise GsplatError(2, f"Renderer image size mismatch: expected ({self.W}, {self.H}), got "
                                 f"({imageWidth}, {imageHeight})")
        packed = self.buildPackedGaussians(means2d, conic, color, opacity, depths)
        radii_t, depths_t = self._t(radii), self._t(depths)
        N = radii_t.shape[0]
        self._check(self.lib.gs_tile_bin(self.ctx, N, _p(self._t(rect[0])), _p(self._t(rect[1])), _p(radii_t),
                                         _p(depths_t)))
        c, d, a = self.globalTileComposite(packed)
        return RenderResult(c.view(self.H, self.W, 3), d.view(self.H, self.W, 1), a.view(self.H, self.W, 1),
                            radii_t > 0, radii_t)

    def forwardWithCameraParams(self, viewMatrix, projMatrix, cameraCenter, fovX, fovY, focalX, focalY, imageWidth,
                                imageHeight, means3d, shs, opacity, scales, rotations):
        cam = self._camera(viewMatrix, projMatrix, cameraCenter, fovX, fovY, focalX, focalY)
        scales, rotations, means3d, shs = map(self._t, (scales, rotations, means3d, shs))
        o = self.projectionScreenFused(scales, rotations, means3d, shs, cam)
        res = self.render(imageWidth, imageHeight, o["means2d"], o["cov2d"], o["color"], opacity, o["depths"],
                          o["radii"], o["conic"], (o["rectMin"], o["rectMax"]))
        # what the VJP chain needs (the reference keeps it in the closures of its two custom functions)
        self._chain = dict(cam=cam, scales=scales, rotations=rotations, means3d=means3d, shs=shs, blend=self._saved)
        return res

    def forwardWithCameraParamsVJP(self, cotRender, cotDepth=None, cotAlpha=None):
        """The VJP MLX composes for forwardWithCameraParams (GaussianTrainer.swift:719-722 over
        GaussianRenderer.swift:149-185, 85-99, 605-701): blend VJP -> split of gradPacked along buildPackedGaussians'
        columns -> projection VJP.  Returns the gradients w.r.t. the ACTIVATED inputs means3d, shs, opacity, scales,
        rotations (the activation VJPs are the host framework's, as in the reference) plus gradPacked itself."""
        ch = self._chain
        gp = self.globalTileCompositeVJP(cotRender, cotDepth, cotAlpha, saved=ch["blend"])
        N = gp.shape[0]
        # packed columns (GaussianRenderer.swift:45-51): means2d 0:2, conic 2:6, colour 6:9, opacity 9, depth 10
        pb = self.projectionScreenFusedVJP(ch["scales"], ch["rotations"], ch["means3d"], ch["shs"], ch["cam"],
                                           gp[:, 0:2], gp[:, 10], gp[:, 6:9], torch.zeros(N, 4, device=self.device),
                                           gp[:, 2:6])
        return dict(means3d=pb["gradMeans3d"], shs=pb["gradShs"], opacity=gp[:, 9].contiguous(), scales=pb["gradScales"],
                    rotations=pb["gradRot"], gradPacked=gp, gradCameraCenter=pb["gradCameraCenter"])

    def forward(self, camera, means3d, shs, opacity, scales, rotations):
        return self.forwardWithCameraParams(camera.worldViewTransform, camera.projectionMatrix, camera.cameraCenter,
                                            camera.FoVx, camera.FoVy, camera.focalX, camera.focalY,
                                            camera.imageWidth, camera.imageHeight, means3d, shs, opacity, scales,
                                            rotations)

    # -- fused raw-parameter path (the trainer's lossFn, GaussianTrainer.swift:652-686) ------------------------
    def renderForward(self, params: dict, camera, want_radii: bool = False, viewKey=None, depthCuts: bool = True,
                      wantDepth: bool = True):
        """params: raw tensors xyz, features_dc, features_rest, scales, rotation, opacity (device f32).
        wantDepth=False: no depth image (RenderResult.depth is None; a train step without a depth term reads none, and
        the blend then carries no depth sum); the backward of such a forward takes no depth cotangent.
        viewKey: any hashable naming the camera (e.g. the training view index).  When given, the per-block sweep
        lengths this forward measures live in a buffer kept under that key, and the next forward of the same view
        reads them as a scheduling hint (deepest blocks first) and -- depthCuts -- bins every tile only as deep as
        that visit needed it plus a margin (gs_set_view_hints).  A forward under cuts may MISS (a tile needed more
        than it was given): outputs are final only once forwardMissed() said False; renderChecked() does both."""
        cam = camera if isinstance(camera, _lib.gs_camera) else self._camera(
            camera.worldViewTransform, camera.projectionMatrix, camera.cameraCenter, camera.FoVx, camera.FoVy,
            camera.focalX, camera.focalY)
        p = {k: (v if isinstance(v, torch.Tensor) and v.is_cuda and v.dtype == torch.float32 and v.is_contiguous()
                 else self._t(v)) for k, v in params.items()}
        N = p["xyz"].shape[0]
        K = 1 + p["features_rest"].shape[1]
        P = self.W * self.H
        if getattr(self, "_fbuf", None) is None:
            self._fbuf = (self._empty(P, 3), self._empty(P), self._empty(P))
        color, depth, alpha = self._fbuf
        radii = self._empty(N) if want_radii else None
        buf = None
        if viewKey is not None and self._hints_ok:
            buf = self._work_hints.get(viewKey)
            if buf is None:
                n = C.c_int()
                self._check(self.lib.gs_view_hint_words(self.ctx, C.byref(n)))
                buf = self._work_hints[viewKey] = torch.zeros(n.value, dtype=torch.int32, device=self.device)
        self._check(self.lib.gs_set_view_hints(self.ctx, _p(buf), 0 if buf is None else int(buf.numel())))   # hint in, measurement out
        # per-view policy: cuts only where they pay (CutPolicy)
        use = False
        if buf is not None:
            pol = self._cut_policy.get(viewKey)
            if pol is None:
                pol = self._cut_policy[viewKey] = CutPolicy(self.cutMinDropped, self.cutProbeInterval)
            pol.min_dropped, pol.probe_interval = self.cutMinDropped, self.cutProbeInterval
            use = pol.begin(depthCuts and self.depthCuts)
        self._cut_view = viewKey if use else None
        self._hinted_view = viewKey if buf is not None else None     # whose cuts the backward of this forward renews
        self._check(self.lib.gs_set_depth_cuts(self.ctx, 1 if use else 0))
        self._check(self.lib.gs_render_forward(self.ctx, N, K, _p(p["xyz"]), _p(p["features_dc"]),
                                               _p(p["features_rest"]), _p(p["scales"]), _p(p["rotation"]),
                                               _p(p["opacity"]), C.byref(cam), _p(color),
                                               _p(depth if wantDepth else None), _p(alpha), _p(radii)))
        self._fused = dict(params=p, color=color, depth=depth if wantDepth else None, alpha=alpha)
        return RenderResult(color.view(self.H, self.W, 3), depth.view(self.H, self.W, 1) if wantDepth else None,
                            alpha.view(self.H, self.W, 1), None if radii is None else radii > 0, radii)

    def dropDepthCuts(self):
        """Forget every view's depth cuts (after the model was rebuilt): the next forward of each view bins in full
        and the cuts are derived afresh.  The scheduling hints stay.  Host-side only: a view whose policy says "no cuts yet"
        never bins under the words its buffer still holds (CutPolicy.begin), and the backward preparation of its next forward
        rewrites every one of them -- clearing the buffers on the device as well (round 3) was one memset launch per view, a
        hundred per densify event."""
        for pol in self._cut_policy.values():
            pol.cuts_cleared()

    def _cuts_renewed(self):
        """Called where a backward preparation of the last fused forward is queued (its loss or its backward)."""
        if self._hinted_view is not None:
            pol = self._cut_policy.get(self._hinted_view)
            if pol is not None:
                pol.renewed()

    def forwardMissed(self) -> bool:
        """True if the last renderForward ran under depth cuts and has to be repeated with depthCuts=False.  Waits
        for that forward only (work queued behind it keeps the GPU busy meanwhile)."""
        if os.environ.get("GSPLAT_DEBUG_NO_MISS_CHECK"):      # timing experiments only: results are not guaranteed
            return False
        m = C.c_int()
        self._check(self.lib.gs_forward_missed(self.ctx, C.byref(m)))
        if self._cut_view is not None:
            st = (C.c_uint32 * 2)()
            self._check(self.lib.gs_cut_stats(self.ctx, st))
            self._cut_policy[self._cut_view].report(bool(m.value), int(st[0]), int(st[1]))
            self._cut_view = None
        return bool(m.value)

    def renderChecked(self, params: dict, camera, want_radii: bool = False, viewKey=None, wantDepth: bool = True):
        """renderForward, repeated without depth cuts if it missed: outputs are final on return."""
        res = self.renderForward(params, camera, want_radii, viewKey, wantDepth=wantDepth)
        if self.forwardMissed():
            res = self.renderForward(params, camera, want_radii, viewKey, depthCuts=False, wantDepth=wantDepth)
        return res

    @property
    def blockLists(self) -> bool:
        """True when the fused path of this renderer works on block lists (a tile size that is not a multiple of 16: the 16 x 16
        blocks are enumerated per tile and binned, sorted and blended one by one; include/gsplat.h)."""
        return (self.TILE_SIZE.w % 16 != 0 or self.TILE_SIZE.h % 16 != 0) and os.environ.get("GSPLAT_BLOCK_LISTS", "1") != "0"

    def blockWork(self):
        """Sweep length of every pixel block in the last fused forward (list entries up to the block's last contributing one):
        int32 [gs_block_count].  Fused path with 16 x 16 tiles or block lists only."""
        n = C.c_int()
        self._check(self.lib.gs_block_count(self.ctx, C.byref(n)))
        out = self._empty(n.value, dtype=torch.int32)
        self._check(self.lib.gs_copy_block_work(self.ctx, _p(out)))
        return out

    def lastContrib(self):
        out = self._empty(self.H, self.W, dtype=torch.int32)
        self._check(self.lib.gs_copy_last_contrib(self.ctx, _p(out)))
        return out

    STAGES = ("proj_fwd", "bin", "blend_fwd", "loss", "blend_bwd", "proj_bwd", "adam")

    def profile(self, stages=True):
        """stages: True = all, False/None = off, or an iterable of stage names."""
        if stages is True:
            mask = (1 << len(self.STAGES)) - 1
        elif not stages:
            mask = 0
        else:
            mask = sum(1 << self.STAGES.index(s) for s in stages)
        self._check(self.lib.gs_profile_enable(self.ctx, mask))

    def profileRead(self):
        ms, calls = (C.c_float * 8)(), (C.c_int * 8)()
        self._check(self.lib.gs_profile_read(self.ctx, ms, calls))
        return {n: (ms[i], calls[i]) for i, n in enumerate(self.STAGES)}

    def renderBackward(self, cotColor, cotDepth=None, cotAlpha=None, out: dict | None = None):
        p = self._fused["params"]
        g = out or {k: torch.empty_like(v) for k, v in p.items()}
        cotColor = self._t(cotColor)
        cotDepth = None if cotDepth is None else self._t(cotDepth)
        cotAlpha = None if cotAlpha is None else self._t(cotAlpha)
        self._cuts_renewed()
        self._check(self.lib.gs_render_backward(self.ctx, _p(cotColor), _p(cotDepth), _p(cotAlpha), _p(g["xyz"]),
                                                _p(g["features_dc"]), _p(g["features_rest"]), _p(g["scales"]),
                                                _p(g["rotation"]), _p(g["opacity"])))
        return g

    def renderBackwardAdam(self, cotColor, arena, m, v, lrs, beta1=0.9, beta2=0.999, eps=1e-15, grad_scale=1.0,
                           cotDepth=None, cotAlpha=None):
        """Backward with the Adam step fused into the projection backward (single-device steps): the parameters the
        preceding renderForward saw must be views into `arena`; m / v are the moment arenas of the same layout; lrs
        in the reference's parameter order (xyz, f_dc, f_rest, scales, rotation, opacity).  No gradient arena is
        written."""
        cotColor = self._t(cotColor)
        cotDepth = None if cotDepth is None else self._t(cotDepth)
        cotAlpha = None if cotAlpha is None else self._t(cotAlpha)
        lr = (C.c_float * 6)(*[float(x) for x in lrs])
        self._cuts_renewed()
        self._check(self.lib.gs_render_backward_adam(self.ctx, _p(cotColor), _p(cotDepth), _p(cotAlpha), _p(arena), _p(m),
                                                     _p(v), int(arena.numel()), lr, C.c_float(beta1), C.c_float(beta2),
                                                     C.c_float(eps), C.c_float(grad_scale)))

    def renderBackwardDP(self, cotColor, cotDepth=None, cotAlpha=None, out: dict | None = None, colorCot=None):
        """Data-parallel backward: as renderBackward, but returns colorCot[N,3] (colour cotangent after the max(.,0)
        gate) instead of the two SH gradient tensors; see shGradFromViews."""
        p = self._fused["params"]
        g = out or {k: torch.empty_like(p[k]) for k in ("xyz", "scales", "rotation", "opacity")}
        N = p["xyz"].shape[0]
        colorCot = self._empty(N, 3) if colorCot is None else colorCot
        cotColor = self._t(cotColor)
        cotDepth = None if cotDepth is None else self._t(cotDepth)
        cotAlpha = None if cotAlpha is None else self._t(cotAlpha)
        self._cuts_renewed()
        self._check(self.lib.gs_render_backward_dp(self.ctx, _p(cotColor), _p(cotDepth), _p(cotAlpha), _p(g["xyz"]),
                                                   _p(g["scales"]), _p(g["rotation"]), _p(g["opacity"]), _p(colorCot)))
        return g, colorCot

    def shGradFromViewsAdam(self, params: dict, colorCotAll, camCenters, arena, m, v, lr_dc, lr_rest, grad_scale,
                            beta1=0.9, beta2=0.999, eps=1e-15):
        """shGradFromViews + the Adam step of features_dc / features_rest in one pass (they are views into arena).
        Uses params["xyz"] as it is: run it before the geometry slice's Adam step."""
        R, N = int(colorCotAll.shape[0]), int(params["xyz"].shape[0])
        K = int(params["features_rest"].shape[1]) + 1
        cc = np.ascontiguousarray(camCenters, np.float32).reshape(R, 3)
        self._check(self.lib.gs_sh_grad_from_views_adam(
            self.ctx, N, K, R, _p(params["xyz"]), _p(colorCotAll), cc.ctypes.data_as(C.c_void_p), _p(params["features_dc"]),
            _p(params["features_rest"]), _p(arena), _p(m), _p(v), int(arena.numel()), C.c_float(lr_dc), C.c_float(lr_rest),
            C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(grad_scale)))

    def renderBackwardDPBegin(self, cotColor, cotDepth=None, cotAlpha=None, colorCot=None):
        """First half of renderBackwardDP: blend backward + colorCot, so the caller can start exchanging colorCot
        while renderBackwardDPFinish (projection backward, the four geometry gradients) runs."""
        N = self._fused["params"]["xyz"].shape[0]
        colorCot = self._empty(N, 3) if colorCot is None else colorCot
        cotColor = self._t(cotColor)
        cotDepth = None if cotDepth is None else self._t(cotDepth)
        cotAlpha = None if cotAlpha is None else self._t(cotAlpha)
        self._cuts_renewed()
        self._check(self.lib.gs_render_backward_dp_begin(self.ctx, _p(cotColor), _p(cotDepth), _p(cotAlpha), _p(colorCot)))
        return colorCot

    def renderBackwardDPFinish(self, out: dict | None = None):
        p = self._fused["params"]
        g = out or {k: torch.empty_like(p[k]) for k in ("xyz", "scales", "rotation", "opacity")}
        self._check(self.lib.gs_render_backward_dp_finish(self.ctx, _p(g["xyz"]), _p(g["scales"]), _p(g["rotation"]),
                                                          _p(g["opacity"])))
        return g

    def renderBackwardDPFinishGeom(self, out: dict, xyzOwn):
        """Round 6 (ABI 6): renderBackwardDPFinish without the SH rows -- out["xyz"] lacks the view-direction term of the xyz
        gradient (shGradFromViewsAdamDir rebuilds it for all views), xyzOwn [N,3] receives a copy of it for the densify statistic."""
        self._check(self.lib.gs_render_backward_dp_finish_geom(self.ctx, _p(out["xyz"]), _p(out["scales"]), _p(out["rotation"]),
                                                               _p(out["opacity"]), _p(xyzOwn)))
        return out

    def renderBackwardDPGeom(self, cotColor, colorCot, out: dict, xyzOwn, cotDepth=None, cotAlpha=None):
        """renderBackwardDPBegin + renderBackwardDPFinishGeom with one kernel behind the blend backward (gs_render_backward_dp_geom)."""
        cotColor = self._t(cotColor)
        cotDepth = None if cotDepth is None else self._t(cotDepth)
        cotAlpha = None if cotAlpha is None else self._t(cotAlpha)
        self._cuts_renewed()
        self._check(self.lib.gs_render_backward_dp_geom(self.ctx, _p(cotColor), _p(cotDepth), _p(cotAlpha), _p(colorCot), _p(out["xyz"]),
                                                        _p(out["scales"]), _p(out["rotation"]), _p(out["opacity"]), _p(xyzOwn)))
        return out

    def shGradFromViewsAdamDir(self, params: dict, colorCotAll, camCenters, ownXyz, arena, m, v, lr_dc, lr_rest, grad_scale, xyzAdd,
                               beta1=0.9, beta2=0.999, eps=1e-15):
        """shGradFromViewsAdam + xyzAdd = the sum over the R views of the xyz gradient's view-direction term + the densify
        statistic of the views this rank rendered (ownXyz: R entries, the xyzOwn tensor of renderBackwardDPFinishGeom for this
        rank's views, None for the others)."""
        R, N = int(colorCotAll.shape[0]), int(params["xyz"].shape[0])
        K = int(params["features_rest"].shape[1]) + 1
        cc = np.ascontiguousarray(camCenters, np.float32).reshape(R, 3)
        own = (C.c_void_p * R)(*[None if t is None else t.data_ptr() for t in ownXyz])
        self._check(self.lib.gs_sh_grad_from_views_adam_dir(
            self.ctx, N, K, R, _p(params["xyz"]), _p(colorCotAll), cc.ctypes.data_as(C.c_void_p), own, _p(params["features_dc"]),
            _p(params["features_rest"]), _p(arena), _p(m), _p(v), int(arena.numel()), C.c_float(lr_dc), C.c_float(lr_rest),
            C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_float(grad_scale), _p(xyzAdd)))

    def shGradFromViews(self, xyz, colorCotAll, camCenters, K: int, out: dict | None = None):
        """grad features_dc / features_rest summed over the R views whose colorCot[R,N,3] and camera centres
        (host [R,3]) are given: sum_r basis_k(xyz - centre_r) * colorCot_r."""
        xyz, colorCotAll = self._t(xyz), self._t(colorCotAll)
        R, N = int(colorCotAll.shape[0]), int(xyz.shape[0])
        cc = np.ascontiguousarray(camCenters, np.float32).reshape(R, 3)
        g = out or dict(features_dc=self._empty(N, 1, 3), features_rest=self._empty(N, K - 1, 3))
        self._check(self.lib.gs_sh_grad_from_views(self.ctx, N, K, R, _p(xyz), _p(colorCotAll),
                                                   cc.ctypes.data_as(C.c_void_p), _p(g["features_dc"]),
                                                   _p(g["features_rest"])))
        return g

    # -- densify / prune kernels (GaussianTrainer.swift:317-427) and the gather of :858-893 -----------------------
    def setGradNormAccum(self, accum):
        """Fuse the densification statistic into the backward: the following renderBackward* calls add |grad xyz| to
        accum [N] (None = off).  The tensor must stay alive while set."""
        self._grad_norm_accum = accum
        self._check(self.lib.gs_set_grad_norm_accum(self.ctx, _p(accum)))

    def accumGradNorm(self, xyzGrad, accumIn=None, out=None):
        xyzGrad = self._t(xyzGrad)
        N = int(xyzGrad.shape[0])
        out = self._empty(N) if out is None else out
        self._check(self.lib.gs_accum_grad_norm(self.ctx, N, _p(xyzGrad), _p(None if accumIn is None else self._t(accumIn)),
                                                _p(out)))
        return out

    def classifyGaussians(self, gradAccum, denom: float, scales, opacity, gradThreshold=0.0002, maxScale=0.01,
                          minOpacity=0.005, allowDensify=True):
        gradAccum, scales, opacity = self._t(gradAccum), self._t(scales), self._t(opacity)
        N = int(gradAccum.shape[0])
        actions, counts = self._empty(N, dtype=torch.int32), self._empty(N, dtype=torch.int32)
        self._check(self.lib.gs_classify_gaussians(self.ctx, N, _p(gradAccum), C.c_float(denom), _p(scales),
                                                   int(scales.shape[1]) if N else 3, _p(opacity), C.c_float(gradThreshold),
                                                   C.c_float(maxScale), C.c_float(minOpacity), int(bool(allowDensify)),
                                                   _p(actions), _p(counts)))
        return actions, counts

    def densifyOffsets(self, actions, counts):
        """Exclusive scan of the output counts + action counts; synchronises (the reference's .item())."""
        N = int(actions.shape[0])
        offsets = self._empty(N, dtype=torch.int32)
        st = (C.c_longlong * 5)()
        self._check(self.lib.gs_densify_offsets(self.ctx, N, _p(actions), _p(counts), _p(offsets), st))
        return offsets, dict(zip(("total", "keep", "split", "clone", "prune"), (int(x) for x in st)))

    def buildDensifyOutputMap(self, actions, offsets, total: int):
        gather, mode = self._empty(total, dtype=torch.int32), self._empty(total, dtype=torch.int32)
        self._check(self.lib.gs_build_densify_output_map(self.ctx, int(actions.shape[0]), _p(actions), _p(offsets),
                                                         int(total), _p(gather), _p(mode)))
        return gather, mode

    # -- the planned event (gs_densify_plan ...): the count stays on the device, the host waits for the plan alone ----------
    def densifyPlan(self, actions, counts):
        """The scan of densifyOffsets without its wait; the plan stays in the context (densifyPlanRead)."""
        N = int(actions.shape[0])
        offsets = self._empty(N, dtype=torch.int32)
        self._check(self.lib.gs_densify_plan(self.ctx, N, _p(actions), _p(counts), _p(offsets)))
        return offsets

    def densifyPlanRead(self, wait: bool = True):
        """dict(N_new, applies, total, keep, split, clone, prune, N) once the plan kernel has run, else None (wait=False)."""
        plan, ready = (C.c_longlong * 8)(), C.c_int()
        self._check(self.lib.gs_densify_plan_read(self.ctx, int(bool(wait)), plan, C.byref(ready)))
        if not ready.value:
            return None
        return dict(zip(("N_new", "applies", "total", "keep", "split", "clone", "prune", "N"), (int(x) for x in plan)))

    def buildDensifyOutputMapPlanned(self, actions, offsets, capacity: int):
        gather, mode = self._empty(capacity, dtype=torch.int32), self._empty(capacity, dtype=torch.int32)
        self._check(self.lib.gs_build_densify_output_map_planned(self.ctx, int(actions.shape[0]), _p(actions), _p(offsets),
                                                                 int(capacity), _p(gather), _p(mode)))
        return gather, mode

    def densifyGatherPlanned(self, params: dict, gather, noiseMode, noiseSeed: int, out: dict, capacity: int):
        """out: views whose POINTERS are used (rows [0, new count) are written: the storage behind them must hold
        min(new count, capacity) rows)."""
        p = {k: self._t(v) for k, v in params.items()}
        K = int(p["features_rest"].shape[1]) + 1
        self._check(self.lib.gs_densify_gather_planned(self.ctx, int(capacity), K, _p(p["xyz"]), _p(p["features_dc"]),
                                                       _p(p["features_rest"]), _p(p["scales"]), _p(p["rotation"]),
                                                       _p(p["opacity"]), _p(gather), _p(noiseMode),
                                                       C.c_ulonglong(int(noiseSeed) & 0xFFFFFFFFFFFFFFFF), _p(out["xyz"]),
                                                       _p(out["features_dc"]), _p(out["features_rest"]), _p(out["scales"]),
                                                       _p(out["rotation"]), _p(out["opacity"])))

    # tensor ids of gs_densify_gather_planned_packed's arena_order (the reference's parameter order, GaussianModel.swift:46-55)
    _PACKED_IDS = dict(xyz=0, features_dc=1, features_rest=2, scales=3, rotation=4, opacity=5)

    def densifyGatherPlannedPacked(self, params: dict, gather, noiseMode, noiseSeed: int, outBase, capacity: int, arenaOrder):
        """The planned gather into a PACKED arena at outBase (a flat float tensor with room for `capacity` Gaussians): every
        tensor's segment is sized by the plan's new count, padded to four floats, in arenaOrder (tensor names in the order
        they lie in the arena); the tensor starts are computed on the device behind the plan (ABI 6)."""
        p = {k: self._t(v) for k, v in params.items()}
        K = int(p["features_rest"].shape[1]) + 1
        order = (C.c_int * 6)(*[self._PACKED_IDS[k] for k in arenaOrder])
        self._check(self.lib.gs_densify_gather_planned_packed(self.ctx, int(capacity), K, _p(p["xyz"]), _p(p["features_dc"]),
                                                              _p(p["features_rest"]), _p(p["scales"]), _p(p["rotation"]),
                                                              _p(p["opacity"]), _p(gather), _p(noiseMode),
                                                              C.c_ulonglong(int(noiseSeed) & 0xFFFFFFFFFFFFFFFF), _p(outBase), order))

    def densifyNoise(self, seed: int, rows: int):
        """[rows, 3] standard normal, row j a function of (seed, j) alone: what the planned gather adds, as a tensor."""
        out = self._empty(int(rows), 3)
        self._check(self.lib.gs_densify_noise(self.ctx, C.c_ulonglong(int(seed) & 0xFFFFFFFFFFFFFFFF), int(rows), _p(out)))
        return out

    def densifyGather(self, params: dict, gather, noiseMode, baseNoise=None, out: dict | None = None):
        p = {k: self._t(v) for k, v in params.items()}
        total, K = int(gather.shape[0]), int(p["features_rest"].shape[1]) + 1
        o = out or {k: self._empty(total, *v.shape[1:]) for k, v in p.items()}
        nz = None if baseNoise is None else self._t(baseNoise)
        self._check(self.lib.gs_densify_gather(self.ctx, total, K, _p(p["xyz"]), _p(p["features_dc"]),
                                               _p(p["features_rest"]), _p(p["scales"]), _p(p["rotation"]),
                                               _p(p["opacity"]), _p(gather), _p(noiseMode), _p(nz), _p(o["xyz"]),
                                               _p(o["features_dc"]), _p(o["features_rest"]), _p(o["scales"]),
                                               _p(o["rotation"]), _p(o["opacity"])))
        return o

    # -- SSIM custom function + loss (GaussianTrainer.swift:555-723) --------------------------------------------
    def ssim(self, img1, img2, window=None, K: int = 11):
        img1, img2 = self._t(img1), self._t(img2)
        H, W, Cc = img1.shape
        window = self.ssimWindow if window is None else self._t(window)
        outs = [self._empty(H, W, Cc) for _ in range(6)]
        self._check(self.lib.gs_ssim_forward(self.ctx, H, W, Cc, K, _p(img1), _p(img2), _p(window),
                                             *[_p(o) for o in outs]))
        self._ssim_saved = (img1, img2, window, K, outs[1:])
        return outs

    def ssimVJP(self, gradOut, saved=None):
        img1, img2, window, K, maps = saved or self._ssim_saved
        H, W, Cc = img1.shape
        gradOut = self._t(gradOut)
        g1, g2 = torch.empty_like(img1), torch.empty_like(img2)
        self._check(self.lib.gs_ssim_backward(self.ctx, H, W, Cc, K, _p(gradOut), _p(img1), _p(img2), _p(window),
                                              *[_p(m) for m in maps], _p(g1), _p(g2)))
        return g1, g2

    def lossForwardBackward(self, render, target, lambda_dssim: float = 0.2, renderDepth=None, targetDepth=None,
                            depthMask=None, lambda_depth: float = 0.0, out=None, targetKey=None):
        """targetKey: any hashable naming the target image (e.g. the training view index).  When given, the target's
        windowed statistics are kept in a per-key device buffer at the first call (15 MB at 800x800) and read back at the
        later ones (gs_set_loss_target_cache); the results are bit-identical either way.  A key whose target tensor has
        changed (another storage, or written in place since) is refilled; the caches together are capped at
        targetStatsCacheBytes, least recently used keys first."""
        render, target = self._t(render), self._t(target)
        ent = None
        if targetKey is not None and self.targetStatsCache:
            # an entry is valid for ONE image: same storage and no in-place write since it was filled (a target rewritten in
            # place, or another image in a reused allocator block, refills it); invalidateTarget(key) drops it explicitly
            ident = (target.data_ptr(), target._version, tuple(target.shape))
            ent = self._target_cache.get(targetKey)
            if ent is None or ent[1] != ident:
                n = C.c_longlong()
                self._check(self.lib.gs_loss_target_cache_floats(self.ctx, C.byref(n)))
                ent = [self._empty(n.value) if ent is None else ent[0], ident, 0]
                self._target_cache[targetKey] = ent
                # the key being served is the most recently used one BEFORE anything is evicted: a refilled key kept its old
                # position, and with the cache over the cap (targetStatsCacheBytes lowered since) the loop below could pop that
                # very key and the step died in a KeyError (round 4's advisor)
                self._target_cache.move_to_end(targetKey)
                # byte cap: least recently used keys go first (tens of GB otherwise on large images with many views)
                while len(self._target_cache) > 1 and 4 * n.value * len(self._target_cache) > self.targetStatsCacheBytes:
                    self._target_cache.popitem(last=False)
            self._target_cache.move_to_end(targetKey)
            self._check(self.lib.gs_set_loss_target_cache(self.ctx, _p(ent[0]), ent[2]))
        else:
            self._check(self.lib.gs_set_loss_target_cache(self.ctx, None, 0))
        lossOut = out["loss"] if out else self._empty(4)
        cotColor = out["cotColor"] if out else torch.empty_like(render)
        cotDepth = None
        rd = td = dm = None
        if lambda_depth != 0.0:
            rd, td = self._t(renderDepth), self._t(targetDepth)
            dm = depthMask.to(device=self.device, dtype=torch.uint8).contiguous()
            cotDepth = self._empty(self.H, self.W)
        self._cuts_renewed()
        self._check(self.lib.gs_loss_forward_backward(self.ctx, _p(render), _p(target), _p(rd), _p(td), _p(dm),
                                                      C.c_float(lambda_dssim), C.c_float(lambda_depth), _p(lossOut),
                                                      _p(cotColor), _p(cotDepth)))
        if ent is not None:
            ent[2] = 1          # filled -- only now that the kernel which fills it has been queued without an error
        return lossOut, cotColor, cotDepth

    def invalidateTarget(self, targetKey=None):
        """Forget the cached SSIM statistics of one target key (None: of all): its next loss recomputes them."""
        if targetKey is None:
            self._target_cache.clear()
        else:
            self._target_cache.pop(targetKey, None)
