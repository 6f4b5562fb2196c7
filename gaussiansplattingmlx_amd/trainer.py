"""Host-side mirror of the reference trainer's per-iteration path (Trainer/GaussianTrainer.swift:958-1086):
lossFn (render + L1/DSSIM loss) -> valueAndGrad -> per-tensor-LR Adam, over the C ABI.

Adds what the reference lacks: a data-parallel step.  Each rank renders its own view; parameter gradients are
summed with ONE all-reduce over a flat arena (RCCL over xGMI via torch.distributed) and Adam runs identically
on every rank with grad_scale = 1/world_size (loss = mean over the views of the step).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .renderer import GaussianRenderer, _p

PARAM_ORDER = ("xyz", "features_dc", "features_rest", "scales", "rotation", "opacity")   # GaussianModel.swift:46-55


def getLearningRates(current: int, total: int):
    """GaussianModel.swift:56-65."""
    return [0.00016 * max(1.0 - float(current) / float(total), 0.01), 0.0025, 0.0025 / 20, 0.005, 0.001, 0.025]


def view_for(step: int, rank: int, world: int, n_views: int) -> int:
    """View sharding: at step s the job consumes views [s*world, (s+1)*world) of a shared permutation; rank r takes
    the r-th of them.  No data-path collective is needed for the forward/backward; only gradients are exchanged."""
    return (step * world + rank) % n_views


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
        shapes = {k: tuple(np.shape(params[k])) for k in PARAM_ORDER}
        sizes = [int(np.prod(shapes[k])) for k in PARAM_ORDER]
        self.numel = int(sum(sizes))
        self.seg_end = np.cumsum(sizes).astype(np.int64)
        self.arena = torch.empty(self.numel, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.arena)
        self.m = torch.zeros_like(self.arena)
        self.v = torch.zeros_like(self.arena)
        self._views, self._gviews = {}, {}
        off = 0
        for k, n in zip(PARAM_ORDER, sizes):
            self._views[k] = self.arena[off:off + n].view(shapes[k])
            self._gviews[k] = self.grad[off:off + n].view(shapes[k])
            self._views[k].copy_(torch.as_tensor(np.ascontiguousarray(params[k], np.float32)))
            off += n
        self.N = shapes["xyz"][0]

    def getParams(self):
        return self._views

    def getGrads(self):
        return self._gviews


class GaussianTrainer:
    def __init__(self, model: GaussModel, gaussRender: GaussianRenderer, iterationCount: int = 30000,
                 lambda_dssim: float = 0.2, process_group=None):
        self.model, self.gaussRender = model, gaussRender
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

    def trainStep(self, camera, targetRGB):
        """One iteration: forward, loss, backward, (all-reduce), Adam.  Asynchronous; returns the device loss[4]."""
        r, m = self.gaussRender, self.model
        res = r.renderForward(m.getParams(), camera)
        r.lossForwardBackward(res.render, targetRGB, self.lambda_dssim, out=dict(loss=self._loss, cotColor=self._cot))
        r.renderBackward(self._cot, out=m.getGrads())
        if self.world > 1:
            allreduce_gradients(m.grad, self.pg)
        lrs = (C.c_float * 6)(*getLearningRates(self.iteration, self.iterationCount))
        r._check(r.lib.gs_adam_step(r.ctx, m.numel, _p(m.arena), _p(m.grad), _p(m.m), _p(m.v), 6, self._seg_end, lrs,
                                    C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), C.c_float(1.0 / self.world)))
        self.iteration += 1
        return self._loss
