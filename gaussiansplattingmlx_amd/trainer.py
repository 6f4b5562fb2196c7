"""Host-side mirror of the reference trainer's per-iteration path (Trainer/GaussianTrainer.swift:958-1086):
lossFn (render + L1/DSSIM loss) -> valueAndGrad -> per-tensor-LR Adam, over the C ABI.

Adds what the reference lacks: a data-parallel step.  Each rank renders its own view; parameter gradients are
summed over ranks (RCCL over xGMI via torch.distributed) and Adam runs identically on every rank with
grad_scale = 1/world_size (loss = mean over the views of the step).  Two exchanges:

  "allreduce"      one all-reduce over the whole flat gradient arena (N * (11 + 3K) floats);
  "sh_compressed"  the SH gradient of a view is rank-1 per Gaussian (basis_k(xyz - cam) x colorCot[3]), so ranks
                   all-gather colorCot (3 floats / Gaussian / view) plus all-reduce the 11 geometry floats, and each
                   rank rebuilds the summed SH gradient locally: at K = 25 and 8 ranks, 35 floats per Gaussian on
                   the wire instead of 86, and the all-reduce (2x traffic) shrinks to 11.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .renderer import GaussianRenderer, _p

PARAM_ORDER = ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity")   # GaussianModel.swift:46-55
# arena layout: the 11 geometry floats per Gaussian first (one contiguous all-reduce), the SH tensors behind them
ARENA_ORDER = ("xyz", "scales", "rotation", "opacity", "features_dc", "features_rest")


def getLearningRates(current: int, total: int):
    """GaussianModel.swift:56-65."""
    return [0.00016 * max(1.0 - float(current) / float(total), 0.01), 0.0025, 0.0025 / 20, 0.005, 0.001, 0.025]


def arenaLearningRates(current: int, total: int):
    """getLearningRates permuted from PARAM_ORDER into ARENA_ORDER."""
    lr = dict(zip(PARAM_ORDER, getLearningRates(current, total)))
    return [lr[k] for k in ARENA_ORDER]


def view_for(step: int, rank: int, world: int, n_views: int) -> int:
    """View sharding: at step s the job consumes views [s*world, (s+1)*world) of a shared permutation; rank r takes
    the r-th of them.  No data-path collective is needed for the forward/backward; only gradients are exchanged."""
    return (step * world + rank) % n_views


def exchange_sh_compressed(grad_geom: torch.Tensor, cc_local: torch.Tensor, cc_all: torch.Tensor, process_group=None):
    """The two collectives of the sh_compressed exchange: all-gather the colour cotangents [N,3] -> [R,N,3] and sum the
    geometry slice of the gradient arena.  The caller then rebuilds the SH gradients (renderer.shGradFromViews)."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(cc_all.view(-1), cc_local.view(-1), group=process_group)   # flat: every backend takes it
    dist.all_reduce(grad_geom, op=dist.ReduceOp.SUM, group=process_group)


def allreduce_gradients(grad_arena: torch.Tensor, process_group=None) -> float:
    """Sum the flat gradient arena over ranks (one collective for all six tensors).  Returns the scale Adam must
    apply (1/world) so that the step uses the mean over the step's views."""
    import torch.distributed as dist
    world = dist.get_world_size(process_group)
    if world > 1:
        dist.all_reduce(grad_arena, op=dist.ReduceOp.SUM, group=process_group)
    return 1.0 / world


class GaussModel:
    """Six raw parameter tensors as views into one flat f32 arena (so one all-reduce / one Adam launch covers them)."""

    def __init__(self, params: dict, device):
        self.device = device
        shapes = {k: tuple(np.shape(params[k])) for k in ARENA_ORDER}
        sizes = [int(np.prod(shapes[k])) for k in ARENA_ORDER]
        self.numel = int(sum(sizes))
        self.seg_end = np.cumsum(sizes).astype(np.int64)
        self.arena = torch.empty(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.arena)
        self.m = torch.zeros_like(self.arena)
        self.v = torch.zeros_like(self.arena)
        self._views, self._gviews = {}, {}
        off = 0
        for k, n in zip(ARENA_ORDER, sizes):
            self._views[k] = self.arena[off:off + n].view(shapes[k])
            self._gviews[k] = self.grad[off:off + n].view(shapes[k])
            self._views[k].copy_(torch.as_tensor(np.ascontiguousarray(params[k], np.float32)))
            off += n
        self.N = shapes["xyz"][0]
        self.K = shapes["features_rest"][1] + 1
        self.geom_numel = int(self.seg_end[3])     # xyz + scales + rotation + opacity

    def getParams(self):
        return self._views

    def getGrads(self):
        return self._gviews


class GaussianTrainer:
    def __init__(self, model: GaussModel, gaussRender: GaussianRenderer, iterationCount: int = 30000,
                 lambda_dssim: float = 0.2, process_group=None, dp_exchange: str = "sh_compressed",
                 exchange_when_single: bool = False):
        if dp_exchange not in ("sh_compressed", "allreduce"):
            raise ValueError(f"unknown dp_exchange {dp_exchange!r}")
        self.model, self.gaussRender = model, gaussRender
        self.dp_exchange = dp_exchange
        self.iterationCount = iterationCount
        self.lambda_dssim = lambda_dssim
        self.pg = process_group
        self.world = 1
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
        r = gaussRender
        self._loss = r._empty(4)
        self._cot = r._empty(r.H, r.W, 3)
        self._seg_end = (C.c_longlong * 6)(*[int(x) for x in model.seg_end])
        self.iteration = 0
        # exchange_when_single: run the collectives even in a 1-rank group (exercises the RCCL path on one GPU)
        self._exchange = process_group is not None and (self.world > 1 or exchange_when_single)
        if self._exchange and dp_exchange == "sh_compressed":
            if self.world > 16:
                raise ValueError("sh_compressed exchange supports at most 16 ranks per group")
            self._cc_local = r._empty(model.N, 3)
            self._cc_all = r._empty(self.world, model.N, 3)

    def trainStep(self, camera, targetRGB, stepCameras=None):
        """One iteration: forward, loss, backward, (gradient exchange), Adam.  Asynchronous; returns the device
        loss[4].  stepCameras: the cameras of ALL ranks for this step in rank order (every rank derives them from the
        shared view permutation, see view_for), or just their centres [R,3]; required by the sh_compressed exchange."""
        r, m = self.gaussRender, self.model
        res = r.renderForward(m.getParams(), camera)
        r.lossForwardBackward(res.render, targetRGB, self.lambda_dssim, out=dict(loss=self._loss, cotColor=self._cot))
        if not self._exchange:
            r.renderBackward(self._cot, out=m.getGrads())
        elif self.dp_exchange == "allreduce":
            r.renderBackward(self._cot, out=m.getGrads())
            allreduce_gradients(m.grad, self.pg)
        else:
            if stepCameras is None or len(stepCameras) != self.world:
                raise ValueError("sh_compressed exchange needs stepCameras (one camera per rank, rank order)")
            g = m.getGrads()
            r.renderBackwardDP(self._cot, out=g, colorCot=self._cc_local)
            exchange_sh_compressed(m.grad[:m.geom_numel], self._cc_local, self._cc_all, self.pg)
            centres = np.stack([np.asarray(getattr(c, "cameraCenter", c), np.float32).reshape(3) for c in stepCameras])
            r.shGradFromViews(m.getParams()["xyz"], self._cc_all, centres, m.K, out=g)
        lrs = (C.c_float * 6)(*arenaLearningRates(self.iteration, self.iterationCount))
        r._check(r.lib.gs_adam_step(r.ctx, m.numel, _p(m.arena), _p(m.grad), _p(m.m), _p(m.v), 6, self._seg_end, lrs,
                                    C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), C.c_float(1.0 / self.world)))
        self.iteration += 1
        return self._loss
