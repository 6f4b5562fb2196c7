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

from ._lib import GS_ERR_REPLICA_MISMATCH, GsplatError
from .renderer import GaussianRenderer, _p

GS_ERR_WORKSPACE_OVERFLOW = 3

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


def balanced_view_order(costs):
    """A visiting order of the views in which any `world` consecutive ones cost about the same.  A data-parallel step takes
    as long as its SLOWEST rank's view; with the views dealt in index order the eight views of a step are eight independent
    draws of the cost distribution (on the bench scene +-5 % around the mean: the maximum of eight is ~6 % above it, paid
    every step).  Sorted by cost and laid out as a zigzag -- ranks 0, 2, 4 ... of the sorted list going up, then ... 5, 3, 1
    coming down -- every window of consecutive views, cyclically, holds neighbours of the sorted list.  Every view is still
    visited once per pass; only the order changes (the reference draws its view at random every iteration,
    GaussianTrainer.swift:486-498, so no order is prescribed).  costs: one number per view (e.g. the view's traversed
    block-entries); returns a permutation of range(len(costs))."""
    order = sorted(range(len(costs)), key=lambda v: (costs[v], v))
    return order[0::2] + order[1::2][::-1]


def cc_block_floats(N: int) -> int:
    """Floats of one rank's block of the colour-cotangent all-gather (gs_dp_cc_floats): its [N,3] cotangents, then ONE word --
    the rank's overflow flag of the step as 0.0 / 1.0 --, padded to a multiple of four floats."""
    return (3 * int(N) + 1 + 3) & ~3


def exchange_sh_compressed(grad_geom: torch.Tensor, cc_local: torch.Tensor, cc_all: torch.Tensor, process_group=None):
    """The two collectives of the sh_compressed exchange: all-gather every rank's block (cc_block_floats: its colour
    cotangents [N,3] and, behind them, its word of the step's gate) -> [R, block] and sum the geometry slice of the gradient
    arena.  The caller then rebuilds the SH gradients (renderer.shGradFromViews), whose kernel ORs the gathered words into the
    step's gate (gathered_gate is the same thing in torch).  Round 5: the gate has no collective of its own any more."""
    import torch.distributed as dist
    dist.all_gather_into_tensor(cc_all.view(-1), cc_local.view(-1), group=process_group)   # flat: every backend takes it
    dist.all_reduce(grad_geom, op=dist.ReduceOp.SUM, group=process_group)


def gathered_gate(cc_all: torch.Tensor, N: int) -> bool:
    """The step's gate from the gathered blocks [R, cc_block_floats(N)]: was any rank's word raised?  What
    sh_grad_from_views_kernel computes on the device (csrc/projection.hip, adam_gate_word); here for hosts / tests in torch."""
    return bool((cc_all.view(cc_all.shape[0], -1)[:, 3 * int(N)] != 0).any().item())


def allreduce_gradients(grad_arena: torch.Tensor, process_group=None) -> float:
    """Sum the flat gradient arena over ranks (one collective for all six tensors; with the step's gate word riding as one
    more float behind the arena -- 0.0 / 1.0 per rank, so the sum is non-zero on every rank iff some rank's forward
    overflowed -- when the caller passes arena + word).  Returns the scale Adam must apply (1/world) so that the step uses
    the mean over the step's views."""
    import torch.distributed as dist
    world = dist.get_world_size(process_group)
    if world > 1:
        dist.all_reduce(grad_arena, op=dist.ReduceOp.SUM, group=process_group)
    return 1.0 / world


class ReplicaMismatch(GsplatError):
    """check_replicas / gs_dp_check_replicas: the ranks of a data-parallel job do not hold the same model."""

    def __init__(self, msg):
        super().__init__(GS_ERR_REPLICA_MISMATCH, msg)


def check_replicas(N: int, arena: torch.Tensor, process_group=None, rank: int | None = None):
    """"verify with an all-reduce'd checksum every densify step".  Densification is replicated without
    communication (same classify inputs, same noise seed: GaussianTrainer.swift:766-908); a rank whose model diverged would
    hang the job in the next size-dependent collective.  One fixed-size collective: (N, sum of the arena in f64, sum of its
    magnitudes) and their negatives, max-reduced -- maxima and minima in one call.  Raises ReplicaMismatch on EVERY rank (the
    verdict is built from reduced values) unless all three agree.  The torch form of gs_dp_check_replicas."""
    import torch.distributed as dist
    w = torch.stack([torch.tensor(float(N), dtype=torch.float64, device=arena.device), arena.sum(dtype=torch.float64),
                     arena.abs().sum(dtype=torch.float64)])
    mine = [float(x) for x in w.cpu()]
    both = torch.cat([w, -w])
    dist.all_reduce(both, op=dist.ReduceOp.MAX, group=process_group)
    hi, lo = [float(x) for x in both[:3].cpu()], [-float(x) for x in both[3:].cpu()]
    if hi == lo:
        return
    _raise_replica_mismatch(hi, lo, mine, dist.get_rank(process_group) if rank is None else rank)


PLAN_WORDS = ("N_new", "applies", "total", "keep", "split", "clone", "prune", "N")      # gs_densify_plan_read's order


def check_plans(words, process_group=None, device=None, side_stream=None, rank: int | None = None):
    """Round 6: the ranks' densify PLANS (gs_densify_plan_read's eight words) compared in one fixed-size collective -- (w, -w)
    max-reduced: maxima and minima in one call.  On a GPU it runs on side_stream, which does not wait for the current stream's
    queue: the event's gather keeps the device busy while the ranks agree.  Raises ReplicaMismatch on EVERY rank (the verdict is
    built from reduced values) unless all words agree: a rank that planned another N would otherwise hang the job in the next
    size-dependent collective.  The torch form of gs_dp_check_plan."""
    import torch.distributed as dist
    w = torch.tensor([float(x) for x in words], dtype=torch.float64)
    both = torch.cat([w, -w])
    if device is not None and torch.device(device).type == "cuda":
        ctx = torch.cuda.stream(side_stream) if side_stream is not None else torch.cuda.stream(torch.cuda.current_stream(device))
        with ctx:
            dev = both.to(device, non_blocking=False)
            dist.all_reduce(dev, op=dist.ReduceOp.MAX, group=process_group)
            both = dev.cpu()
    else:
        dist.all_reduce(both, op=dist.ReduceOp.MAX, group=process_group)
    n = len(words)
    hi, lo = [float(x) for x in both[:n]], [-float(x) for x in both[n:]]
    if hi == lo:
        return
    diff = " ".join(f"{name} in [{l:.0f}, {h:.0f}]" for name, h, l in zip(PLAN_WORDS, hi, lo) if h != l)
    raise ReplicaMismatch(f"the ranks planned different densify events ({diff}); rank "
                          f"{dist.get_rank(process_group) if rank is None else rank} planned new N = {int(words[0])} from N = {int(words[-1])}")


def check_replicas_begin(N: int, arena: torch.Tensor, process_group=None):
    """check_replicas queued: the checksum and its max all-reduce are enqueued, nothing is read back (check_replicas_end
    does that).  Returns the pending state."""
    import torch.distributed as dist
    w = torch.stack([torch.tensor(float(N), dtype=torch.float64, device=arena.device), arena.sum(dtype=torch.float64),
                     arena.abs().sum(dtype=torch.float64)])
    both = torch.cat([w, -w])
    work = dist.all_reduce(both, op=dist.ReduceOp.MAX, group=process_group, async_op=True)
    return (w, both, work)


def check_replicas_end(pending, process_group=None, rank: int | None = None):
    import torch.distributed as dist
    w, both, work = pending
    work.wait()
    mine = [float(x) for x in w.cpu()]
    hi, lo = [float(x) for x in both[:3].cpu()], [-float(x) for x in both[3:].cpu()]
    if hi == lo:
        return
    _raise_replica_mismatch(hi, lo, mine, dist.get_rank(process_group) if rank is None else rank)


def _raise_replica_mismatch(hi, lo, mine, rank):
    names = ("N", "sum", "magnitudes")
    diff = " ".join(n for n, h, l in zip(names, hi, lo) if h != l)
    raise ReplicaMismatch(f"the ranks hold different models ({diff}): over the ranks N in [{lo[0]:.0f}, {hi[0]:.0f}], sum in "
                          f"[{lo[1]!r}, {hi[1]!r}], sum of magnitudes in [{lo[2]!r}, {hi[2]!r}]; rank {rank} has N = {mine[0]:.0f}, "
                          f"sum = {mine[1]!r}, sum of magnitudes = {mine[2]!r}")


def exchange_summary(dp_impl, dp_exchange, world, N, geom_numel, numel, steps, sums, counts, rccl_version, source, views_per_rank=1):
    """The `exchange` block of a data-parallel bench line from what a timed run accumulated.  sums: milliseconds summed over
    the timed steps -- "gather" / "reduce" = duration of the colour-cotangent all-gather and the gradient all-reduce on the
    communication stream (they include the wait for the slowest peer; "gate": rounds 3-4's 4-byte all-reduce, gone in round 5 --
    its count is 0 and gate_ms null); "exposed_gather" /
    "exposed_reduce" = time the render stream stood waiting for them, i.e. wire time NOT hidden under compute.  counts: how
    many steps contributed a duration per collective (0: the backend keeps none -> null).  Bytes are per rank and step:
    what the rank hands to the collective (out) and what it holds afterwards (in)."""
    n = max(int(steps), 1)
    per = lambda k: round(sums[k] / counts[k], 4) if counts.get(k) else None
    xg, xr = sums["exposed_gather"] / n, sums["exposed_reduce"] / n
    out = dict(dp_impl=dp_impl, dp_exchange=dp_exchange, world=int(world), steps_measured=int(steps), rccl_version=rccl_version,
               timing_source=source, gate_ms=per("gate"), gather_ms=per("gather"), reduce_ms=per("reduce"),
               exposed_gather_ms=round(xg, 4), exposed_reduce_ms=round(xr, 4), exposed_ms=round(xg + xr, 4))
    # round 5: the gate word rides in the step's first payload (behind the colour cotangents / behind the gradient arena)
    if dp_exchange == "sh_compressed":
        blk = 4 * cc_block_floats(N) * int(views_per_rank)        # (a rank's views' blocks one behind the other)
        out.update(gather_bytes_out=blk, gather_bytes_in=blk * int(world), reduce_bytes=4 * int(geom_numel),
                   collectives_per_step=2, gate_rides_in="gather", views_per_rank=int(views_per_rank))
    else:
        out.update(gather_bytes_out=0, gather_bytes_in=0, reduce_bytes=4 * (int(numel) + 1), collectives_per_step=1,
                   gate_rides_in="reduce")
    out["gate_bytes"] = 4
    return out


class GaussModel:
    """Six raw parameter tensors as views into one flat f32 arena (so one all-reduce / one Adam launch covers them).

    Every tensor starts on a 16-byte boundary of the arena (its segment is padded to a multiple of four floats; the pad
    stays zero in all four arenas): the fused kernels address the SH rows as float4 and keep them in registers between
    the staging load and the Adam update only then, and an odd Gaussian count after a densify event used to put them
    on the slower unaligned path.

    The arena is a prefix of a buffer with room for `capacity` Gaussians, and the parameters are double-buffered, so
    a densify / prune event gathers straight into the other buffer and flips -- no allocation, no copy -- as long as
    the new count fits (it regrows by 1.5x otherwise).

    `stride` (round 5): the rows every tensor's segment has room for.  stride = N is the packed layout above (what a
    data-parallel all-reduce of the leading geometry slice wants).  The trainer's planned densify event lays the new
    model out at stride = capacity instead: the six tensors' POINTERS then do not depend on the new count, which is still
    on the device when the gather is queued (split_and_prune); the rows between N and the stride are never read (every
    kernel of the path indexes by Gaussian), numel / seg_end describe the strided arena."""

    def __init__(self, params: dict, device, capacity: int | None = None):
        self.device = device
        self._row = {k: tuple(params[k].shape[1:]) for k in ARENA_ORDER}
        self._per = {k: int(np.prod(self._row[k])) if len(self._row[k]) else 1 for k in ARENA_ORDER}
        self.floats_per_gaussian = int(sum(self._per.values()))
        self.K = self._row["features_rest"][0] + 1
        N = int(params["xyz"].shape[0])
        self._pbuf = [None, None]
        self._cur = 0
        self._gbuf = self._mbuf = self._vbuf = None
        self._staged = None
        self.stride = N
        self._layout(N, max(int(capacity or N), N))
        for k in ARENA_ORDER:
            src = params[k]
            if not isinstance(src, torch.Tensor):
                src = torch.as_tensor(np.ascontiguousarray(src, np.float32))
            self._views[k].copy_(src.reshape(self._views[k].shape))

    def _buf(self, old, floats, zero=False):
        if old is not None and old.numel() >= floats + 4:
            return old
        # (+ 4: one spare word behind every arena -- a data-parallel all-reduce carries the step's gate there)
        return (torch.zeros if zero else torch.empty)(max(floats, 4) + 4, dtype=torch.float32, device=self.device)

    def _offsets(self, N: int):
        """(start of every tensor's segment, total floats) for N Gaussians: segments padded to multiples of 4 floats."""
        starts, off = [], 0
        for k in ARENA_ORDER:
            starts.append(off)
            off += (N * self._per[k] + 3) & ~3
        return starts, off

    def _zero_pads(self, buf, N):
        """The up to three pad floats behind every tensor (a reused buffer may hold an older layout's values there)."""
        starts, total = self._offsets(N)
        for k, off, nxt in zip(ARENA_ORDER, starts, starts[1:] + [total]):
            if off + N * self._per[k] < nxt:
                buf[off + N * self._per[k]:nxt].zero_()

    def _carve(self, buf, N, stride=None):
        views = {}
        starts, _ = self._offsets(N if stride is None else stride)
        for k, off in zip(ARENA_ORDER, starts):
            n = N * self._per[k]
            views[k] = buf[off:off + n].view((N,) + self._row[k])
        return views

    def _layout(self, N: int, capacity: int, pads=("arena", "grad", "m", "v"), stride=None):
        stride = N if stride is None else int(stride)
        floats = self._offsets(max(capacity, stride))[1]
        self.capacity = capacity
        self.N = N
        self.stride = stride
        starts, self.numel = self._offsets(stride)
        self.seg_start = np.asarray(starts, np.int64)
        self.seg_end = np.asarray(starts[1:] + [self.numel], np.int64)      # a segment's pad takes its learning rate (and stays 0)
        self.geom_numel = int(self.seg_end[3])     # xyz + scales + rotation + opacity
        self._pbuf[self._cur] = self._buf(self._pbuf[self._cur], floats, zero=True)
        self._gbuf, self._mbuf, self._vbuf = (self._buf(b, floats, zero=True) for b in (self._gbuf, self._mbuf, self._vbuf))
        self.arena = self._pbuf[self._cur][:self.numel]
        self.grad, self.m, self.v = self._gbuf[:self.numel], self._mbuf[:self.numel], self._vbuf[:self.numel]
        if stride == N:
            for name in pads:  # (each pad is a tiny launch of its own: a caller that zeroes a whole arena anyway leaves it out)
                self._zero_pads(getattr(self, name), N)
        self._views, self._gviews = self._carve(self.arena, N, stride), self._carve(self.grad, N, stride)

    def getParams(self):
        return self._views

    def getGrads(self):
        return self._gviews

    def stagingViews(self, N_new: int, stride=None) -> dict:
        """Views for N_new Gaussians in the OTHER parameter buffer (the densify gather writes them).  stride: rows per
        tensor segment of the staged layout (default: packed, N_new)."""
        cap = self.capacity if N_new <= self.capacity else int(N_new * 1.5)
        stride = N_new if stride is None else int(stride)
        other = 1 - self._cur
        # (zeroed once, when it is allocated: a capacity-strided layout spans rows between N and the stride that no kernel
        # writes, and a consumer that reads `arena` whole -- a checksum, isfinite, a clone -- would meet uninitialised floats)
        self._pbuf[other] = self._buf(self._pbuf[other], self._offsets(max(cap, stride))[1], zero=True)
        self._staged = (N_new, cap, stride)
        return self._carve(self._pbuf[other][:self._offsets(stride)[1]], N_new, stride)

    def stagingBase(self, capacity: int):
        """The OTHER parameter buffer as one flat tensor with room for `capacity` Gaussians in the packed layout (the packed
        planned gather lays the tensors out itself, on the device, once the new count is known there); commitStaged(N_new)
        then flips to it in the packed layout for N_new."""
        cap = max(int(capacity), self.capacity)
        other = 1 - self._cur
        self._pbuf[other] = self._buf(self._pbuf[other], self._offsets(cap)[1], zero=True)
        self._staged = (None, cap, None)
        return self._pbuf[other]

    def commitStaged(self, N_new=None, zero=True):
        """Flip to the staged buffer (split_and_prune phase 6, GaussianTrainer.swift:900-905); gradients and Adam
        moments are zeroed (the reference re-creates the optimizer state, :1104-1109).  N_new: the row count when it was
        not known at staging time (the planned event stages `capacity` rows' worth of pointers); zero=False: the caller
        has zeroed the buffers whole already."""
        n_staged, cap, stride = self._staged
        N_new = n_staged if N_new is None else int(N_new)
        self._staged = None
        self._cur = 1 - self._cur
        self._layout(N_new, cap, pads=("arena",), stride=None if stride is None or (stride == n_staged and N_new == n_staged) else stride)
        if zero:
            self.grad.zero_()                          # gradients and moments are zeroed whole, pads included
            self.resetOptimizerState()

    def zeroOptimizerBuffers(self, grads: bool = False):
        """The moment buffers (and the gradient buffer) zeroed WHOLE -- whatever layout comes next."""
        self._mbuf.zero_()
        self._vbuf.zero_()
        if grads:
            self._gbuf.zero_()

    def commit(self, params: dict):
        """Replace the six tensors by copies of `params` (any source)."""
        out = self.stagingViews(int(params["xyz"].shape[0]))
        for k in ARENA_ORDER:
            src = params[k]
            if not isinstance(src, torch.Tensor):
                src = torch.as_tensor(np.ascontiguousarray(src, np.float32))
            out[k].copy_(src.reshape(out[k].shape))
        self.commitStaged()

    def resetOptimizerState(self):
        self.m.zero_()
        self.v.zero_()


class GaussianTrainer:
    def __init__(self, model: GaussModel, gaussRender: GaussianRenderer, iterationCount: int = 30000,
                 lambda_dssim: float = 0.2, process_group=None, dp_exchange: str = "sh_compressed",
                 exchange_when_single: bool = False, densify: bool = True, fuse_adam: bool = True,
                 exchange_impl: str = "torch", dp_bootstrap=None, views_per_rank: int = 1):
        """exchange_impl: who issues the collectives of a data-parallel step.  "torch": torch.distributed on
        process_group (RCCL when its backend is nccl; gloo for CPU rehearsals).  "native": the library itself
        (gs_dp_step: RCCL on its own side stream, the same event ordering) -- process_group is then only used to hand
        rank 0's RCCL id to the others, or not at all when dp_bootstrap = (id_bytes, rank, world) is given.

        views_per_rank (round 6): V > 1 makes a step take V views on THIS rank -- one update from the mean loss over the
        world x V views of the step ("8 views/step is a semantic extension: loss = mean over views").  Each view
        runs forward + loss + the data-parallel backward exactly as a rank of a V-times larger job would (its colour-cotangent
        block with its gate word behind it, its geometry gradients, its |grad xyz| into the densify statistic); the blocks of
        the rank's views lie one behind the other in the buffer the all-gather hands over (without a process group that buffer
        IS the gathered one: the collective replaced by addressing), the geometry gradients are summed over the rank's views
        before the all-reduce, the SH rebuild runs over all world x V blocks and Adam at grad_scale 1 / (world x V).  With
        world = 1, V = 8 this is BASELINE config 4's arithmetic -- eight views, one update -- on one card.  sh_compressed
        exchange, torch issuer (or no group at all); world x V <= 16."""
        if dp_exchange not in ("sh_compressed", "allreduce"):
            raise ValueError(f"unknown dp_exchange {dp_exchange!r}")
        if exchange_impl not in ("torch", "native"):
            raise ValueError(f"unknown exchange_impl {exchange_impl!r}")
        if dp_bootstrap is not None and exchange_impl != "native":
            raise ValueError("dp_bootstrap = (id, rank, world) is the native exchange's bootstrap: pass exchange_impl='native' "
                             "(the torch exchange needs a process_group)")
        self.viewsPerRank = int(views_per_rank)
        if self.viewsPerRank < 1:
            raise ValueError("views_per_rank must be >= 1")
        if self.viewsPerRank > 1 and (dp_exchange != "sh_compressed" or exchange_impl != "torch"):
            raise ValueError("views_per_rank > 1 takes the sh_compressed exchange with the torch issuer (or no process group)")
        self.exchange_impl = exchange_impl
        self.model, self.gaussRender = model, gaussRender
        self.dp_exchange = dp_exchange
        self.iterationCount = iterationCount
        self.lambda_dssim = lambda_dssim
        self.pg = process_group
        self.world = 1
        self.rank = 0
        if process_group is not None:
            import torch.distributed as dist
            self.world = dist.get_world_size(process_group)
            self.rank = dist.get_rank(process_group)
        if dp_bootstrap is not None:
            _, self.rank, self.world = dp_bootstrap
        r = gaussRender
        # this trainer's loss has no depth term (lambda_depth = 0, the reference's default: GaussianTrainer.swift:280,
        # 949), so no backward of ITS forwards ever brings a depth cotangent and they need not checkpoint the depth sums:
        # the knob is turned off around the trainer's own steps only (_trainStep) -- the renderer is the caller's, and
        # its other users keep whatever they had set
        self._loss = r._empty(4)
        self._cot = r._empty(r.H, r.W, 3)
        self._seg_end = (C.c_longlong * 6)(*[int(x) for x in model.seg_end])
        self.iteration = 0
        # densification (GaussianTrainer.swift:293-300, 304)
        self.gradientThreshold, self.minOpacity, self.maxScale = 0.0002, 0.005, 0.01
        self.densifyFromIter, self.densifyUntilIter, self.maxGaussians = 500, 15000, 1_000_000
        self.split_and_prune_per_iteration = 100
        self.densify = densify
        self.fuse_adam = fuse_adam                     # single-device steps: Adam inside the projection backward
        self.outputDirectory = None                    # set to a path to write iteration_<it>.ply snapshots
        self.save_snapshot_per_iteration = 100
        self.noise_seed = 20260313
        # Densify events WITHOUT a drain of the queue (round 5; single-device trainers -- a data-parallel one learns the count
        # behind its replica check anyway, and wants the packed layout): the count stays on the device for the kernels that
        # need it, the gather into a capacity-strided layout and the optimizer reset are queued BEFORE the host waits, and the
        # host waits for the event's plan alone (split_and_prune).  False: the reference's sequence to the letter -- read the
        # count, then size and queue everything behind it (one drain, ~0.3 ms of idle device per event).
        import os
        self.plannedDensify = os.environ.get("GSPLAT_PLANNED_DENSIFY", "1") != "0"      # (the switch: A/B runs of bench.py)
        # the split / clone noise: "torch" = torch.randn(total, 3) from a generator seeded by (noise_seed, iteration), what
        # rounds 1-4 drew; "library" = gs_densify_noise (row j a function of (seed, j) alone), what a planned event draws
        # inside its gather.  None: "library" where the event is planned, "torch" elsewhere.
        self.noiseSource = None
        self.xyzGradAccumulation = r._empty(model.N).zero_()
        self.denomGradAccumulation = 0
        self.lastDensifyStats = None
        self.forwardMisses = 0                         # forwards repeated without depth cuts (renderer.renderForward)
        # interval profiling (GaussianTrainer.swift:962-966, 1115-1127): every profilingLogInterval-th iteration runs
        # under an IntervalProfiler with the library's stage events on; lastProfileReport keeps its report
        self.enableIntervalProfiling = False
        self.profilingLogInterval = 100
        self.profilingTopKSections = 12
        self.lastProfileReport = None
        self.log = None                                # callable(str) for the reports, e.g. print
        self._committed = False
        # The reference re-reads `params` from the model every split_and_prune_per_iteration iterations whether or not
        # split_and_prune committed anything (GaussianTrainer.swift:1098-1110), and the model's tensors only change at a
        # commit (:900-905): outside the densify window, and at every cadence without a change, its training falls back to
        # the last committed state.  False (default): not mirrored -- training keeps what it has learnt.  True: the
        # reference's trajectory (a copy of the parameters is kept at every commit and restored at those points).
        self.referenceParamReload = False
        self._committed_params = None
        self.overflowRecoveries = 0                    # times the reserved pair capacity had to be regrown (see trainStep)
        self._checked_views = set()                    # views whose first forward has been checked for overflow
        # exchange_when_single: run the collectives even in a 1-rank group (exercises the RCCL path on one GPU)
        self._exchange = (process_group is not None or dp_bootstrap is not None) and (self.world > 1 or exchange_when_single)
        self._native = self._exchange and exchange_impl == "native"
        # the step takes the data-parallel FORM (split backward, gathered colour cotangents, SH rebuild, gate word in the
        # first payload) when it exchanges with other ranks or when this rank brings several views to it
        self._dp = self._exchange or self.viewsPerRank > 1
        # torch issuer: the step's all-gather as a synchronous op on the render stream where the backend is RCCL
        # (_gatherColourCotangents); GSPLAT_DP_INLINE_GATHER=0 keeps it on ProcessGroupNCCL's own stream (A/B)
        self.inlineGather = False
        if self._exchange and not self._native and process_group is not None:
            import torch.distributed as dist
            self.inlineGather = (dist.get_backend(process_group) == "nccl" and
                                 os.environ.get("GSPLAT_DP_INLINE_GATHER", "1") != "0")
        if self._native:
            self._dp_connect(dp_bootstrap)
        if self._dp and dp_exchange == "sh_compressed":
            if self.world * self.viewsPerRank > 16:
                raise ValueError("sh_compressed exchange supports at most 16 views per step (ranks x views_per_rank)")
        # data-parallel: a rank whose forward overflowed its reserved pairs must not be the only one to skip the Adam
        # step, or the replicas drift apart -- every optimizer kernel of a step tests the OR over the ranks of the
        # forwards' overflow words.  Round 5: the word rides in the step's first payload (behind the rank's colour
        # cotangents in the all-gather, whose words the SH rebuild ORs; behind the gradient arena in the all-reduce,
        # summed) -- rounds 3-4 gave it a 4-byte max all-reduce of its own, a third collective per step.
        #
        # And no rank may leave a step on its own: the host-side overflow error is turned off for the trainer's steps
        # (GS_TUNE_HOST_OVERFLOW_ERRORS), an optimizer kernel that finds the gate raised sets a `seen` word, and every
        # overflowCheckInterval-th step ALL ranks read it (one wait) -- it is built from the common gate, so they read the
        # same --, agree on the largest pair count any of them needed (a max all-reduce) and regrow their reserves together
        # (_collectiveOverflowCheck).  Steps in between were skipped by every replica's gate; nothing is applied from a
        # blank render.
        self._gate = None
        self.overflowCheckInterval = 16
        self._xt = None                                # exchange timing (exchangeTimingBegin / exchangeTimingRead)
        if self._dp and not self._native:               # (native: gs_dp_step keeps the gate, gs_dp_check_overflow the look)
            self._gate = torch.zeros(1, dtype=torch.int32, device=r.device)
            self._seen = torch.zeros(1, dtype=torch.int32, device=r.device)
            self._need = torch.zeros(1, dtype=torch.int64, device=r.device)
        self._alloc_exchange_buffers()
        if self._exchange:
            # the replicas must START identical too -- and the check's first call pays for the collective's set-up (a first
            # float64 max-reduce cost the torch exchange ~35 ms at the first densify event of a run) here, not there
            self.checkReplicas()
            if self._plans_events():      # ... likewise the plan check's side stream and its 16-word collective
                self.checkPlans(dict(zip(PLAN_WORDS, (model.N, 0, model.N, model.N, 0, 0, 0, model.N))))

    def _dp_connect(self, bootstrap):
        """gs_dp_init: rank 0 draws the RCCL id, every rank gets it (through process_group, whatever its backend), and the
        library creates its communicator on the renderer's device."""
        from . import _lib
        r = self.gaussRender
        if bootstrap is not None:
            uid = bytes(bootstrap[0])
        else:
            import torch.distributed as dist
            box = [None]
            if self.rank == 0:
                buf = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
                rc = r.lib.gs_dp_unique_id(buf)
                if rc != _lib.GS_OK:
                    raise GsplatError(rc, "gs_dp_unique_id failed (RCCL not loadable?)")
                box[0] = buf.raw
            dist.broadcast_object_list(box, src=dist.get_global_rank(self.pg, 0) if self.pg is not dist.group.WORLD else 0,
                                       group=self.pg)
            uid = box[0]
        r._check(r.lib.gs_dp_init(r.ctx, C.c_char_p(uid), int(self.rank), int(self.world)))

    def closeExchange(self):
        """Drops the library's communicator (native exchange); the renderer's close() does it too."""
        if self._exchange:
            self._resolveReplicaCheck()
        if self._native:
            self.gaussRender.lib.gs_dp_shutdown(self.gaussRender.ctx)
            self._native = self._exchange = False

    def _alloc_exchange_buffers(self):
        """The exchange's buffers for the model's current N, and -- torch exchange -- where the step's gate rides
        (gs_set_overflow_rider / gs_set_gathered_gate / gs_set_update_gate / gs_set_gate_seen; the native exchange does the
        same inside gs_dp_step).  Called again after every committed densify event: N and the arenas have moved."""
        if not self._dp:
            return
        r, m = self.gaussRender, self.model
        N = m.N
        V = self.viewsPerRank
        if self.dp_exchange == "sh_compressed":
            ccf = cc_block_floats(N)
            # this rank's V blocks one behind the other: what the all-gather hands over; without other ranks that buffer IS
            # the gathered one
            self._cc_local = r._empty(V * ccf)
            self._cc_all = r._empty(self.world * V, ccf) if self._exchange else self._cc_local.view(V, ccf)
            self._cc_local.view(V, ccf)[:, 3 * N:].zero_()
        if V > 1:
            # the geometry gradients of the rank's views, each laid out as the gradient arena's leading slice; their sum over
            # the views goes to that slice (zeroed once: a strided layout's rows between N and the stride are never written)
            self._geom_v = torch.zeros(V, m.geom_numel, dtype=torch.float32, device=r.device)
            self._geom_views = []
            for j in range(V):
                views = {}
                for k, off in zip(ARENA_ORDER[:4], m.seg_start[:4]):
                    views[k] = self._geom_v[j, int(off):int(off) + N * m._per[k]].view((N,) + m._row[k])
                self._geom_views.append(views)
            self._loss_v = r._empty(V, 4)
        if self._native:
            return
        if self.fuse_adam and self.dp_exchange == "sh_compressed":
            # the split form of round 6 (include/gsplat.h, gs_render_backward_dp_finish_geom): every view's xyz gradient without its
            # view-direction term, and that term summed over the step's views (the pad behind 3 N stays zero: Adam's last float4)
            x4 = (3 * N + 3) & ~3
            self._xyz_own = torch.zeros(V, x4, dtype=torch.float32, device=r.device)
            self._xyz_add = torch.zeros(x4, dtype=torch.float32, device=r.device)
        if self.dp_exchange == "sh_compressed":
            r._check(r.lib.gs_set_overflow_rider(r.ctx, C.c_void_p(self._cc_local.data_ptr() + 12 * N)))
            r._check(r.lib.gs_set_gathered_gate(r.ctx, cc_block_floats(N), int(self.world * V), _p(self._gate)))
            r._check(r.lib.gs_set_update_gate(r.ctx, _p(self._gate)))
        else:
            # the word behind the gradient arena (GaussModel keeps a spare one): stored by the backward, summed by the
            # all-reduce, tested -- as it lies -- by the Adam kernel (+0.0f is the all-zero word)
            word = C.c_void_p(m._gbuf.data_ptr() + 4 * m.numel)
            r._check(r.lib.gs_set_overflow_rider(r.ctx, word))
            r._check(r.lib.gs_set_gathered_gate(r.ctx, 0, 0, None))
            r._check(r.lib.gs_set_update_gate(r.ctx, word))
        r._check(r.lib.gs_set_gate_seen(r.ctx, _p(self._seen)))

    # -- densification bookkeeping (GaussianTrainer.swift:724-748) ------------------------------------------------
    def addGradientAccumulation(self, xyzGrad=None):
        """accum += |xyz_grad| of THIS rank's view (the reference accumulates per view, :1000); the per-rank
        accumulators are summed over ranks once, when split_and_prune needs them.  Inside trainStep the addition is
        fused into the projection backward (renderer.setGradNormAccum) and this only advances the denominator;
        called with a gradient it runs the stand-alone kernel.  A step the device gate skipped (reserved-capacity
        overflow) still counts in the denominator although its blank render added nothing: at most a few steps per
        regrown reserve, against a threshold that is a mean over ~100."""
        if xyzGrad is not None:
            if self.xyzGradAccumulation.shape[0] != int(xyzGrad.shape[0]):
                self.resetGradientAccumulation()
            self.gaussRender.accumGradNorm(xyzGrad, self.xyzGradAccumulation, out=self.xyzGradAccumulation)
        self.denomGradAccumulation += self.world

    def resetGradientAccumulation(self):
        self.xyzGradAccumulation = self.gaussRender._empty(self.model.N).zero_()
        self.denomGradAccumulation = 0

    def save_snapshot(self, iteration: int):
        """GaussianTrainer.swift:909-930: iteration_<it>.ply in the output directory, raw parameters."""
        import os
        from .ply import PlyWriter
        p = self.model.getParams()
        PlyWriter(self.gaussRender).writeGaussianBinary(p["xyz"], p["features_dc"], p["features_rest"], p["opacity"],
                                                        p["scales"], p["rotation"],
                                                        to=os.path.join(os.fspath(self.outputDirectory),
                                                                        f"iteration_{iteration}.ply"))

    def prewarmDensify(self):
        """Dry run of the whole densify sequence on the current model (nothing is committed): loads the kernels,
        sizes the scan scratch and allocates the second parameter buffer, so the first real event costs what every
        later one does."""
        r, m = self.gaussRender, self.model
        if m.N <= 0:
            return
        p = m.getParams()
        acc = r._empty(m.N).zero_()
        actions, counts = r.classifyGaussians(acc, 1.0, p["scales"], p["opacity"].reshape(-1), self.gradientThreshold,
                                              self.maxScale, self.minOpacity, True)
        offsets, st = r.densifyOffsets(actions, counts)
        if st["total"] <= 0:
            return
        gather, mode = r.buildDensifyOutputMap(actions, offsets, st["total"])
        gen = torch.Generator(device=r.device)
        gen.manual_seed(self.noise_seed)
        noise = torch.randn(st["total"], 3, generator=gen, device=r.device, dtype=torch.float32)
        r.densifyGather(p, gather, mode, noise, out=m.stagingViews(st["total"]))
        m._staged = None
        if self._plans_events():      # ... and the planned form's kernels
            offsets = r.densifyPlan(actions, counts)
            gather, mode = r.buildDensifyOutputMapPlanned(actions, offsets, m.capacity)
            if self._dp:
                r.densifyGatherPlannedPacked(p, gather, mode, self.noise_seed, m.stagingBase(m.capacity), m.capacity, ARENA_ORDER)
            else:
                r.densifyGatherPlanned(p, gather, mode, self.noise_seed, m.stagingViews(m.capacity, stride=m.capacity), m.capacity)
            m._staged = None
            r.densifyPlanRead(wait=True)
            r.densifyNoise(self.noise_seed, 16)

    def _plans_events(self) -> bool:
        return bool(self.plannedDensify) and not self.referenceParamReload

    def _split_and_prune_planned(self, iteration: int, allowDensify: bool):
        """The event with the count left on the device (include/gsplat.h, gs_densify_plan; densify.hip).  Queued before the
        host waits for anything: classify, the scan and the plan, the output map and the gather for `capacity` slots into the
        other parameter buffer at CAPACITY strides (its pointers do not depend on the count), the optimizer reset.  Then the
        host waits for the PLAN -- the device still has the gather and the resets in front of it while the host lays the new
        model out and queues the next step.  The reference's early-outs (:819-847) are the plan's `applies` word: an event
        that changes nothing gathers the identity.  A new count beyond the capacity -- known before anything else is queued,
        the source buffer untouched -- repeats the map and the gather into a larger buffer."""
        r, m = self.gaussRender, self.model
        p = m.getParams()
        # Data-parallel form (round 6): the statistic's all-reduce is queued like any kernel (RCCL on its stream behind this
        # one's work, joined back: no host wait), the gather writes the PACKED layout -- tensor starts computed on the device
        # from the plan, since the step all-reduces the arena's leading geometry slice --, and the ranks compare their PLANS in
        # one fixed-size collective on a side stream as soon as the host has them (a diverged N is caught before the next
        # size-dependent collective without draining the queue); the arena checksum is queued behind the gather and its
        # verdict taken where the host waits anyway (_resolveReplicaCheck).
        packed = self._dp
        if self._exchange:
            self._resolveReplicaCheck()
            if self._native:
                r._check(r.lib.gs_dp_allreduce_sum(r.ctx, _p(self.xyzGradAccumulation), int(m.N)))
            else:
                import torch.distributed as dist
                dist.all_reduce(self.xyzGradAccumulation, op=dist.ReduceOp.SUM, group=self.pg)
        actions, counts = r.classifyGaussians(self.xyzGradAccumulation, float(self.denomGradAccumulation), p["scales"],
                                              p["opacity"].reshape(-1), self.gradientThreshold, self.maxScale,
                                              self.minOpacity, allowDensify)
        offsets = r.densifyPlan(actions, counts)
        seed = self.noise_seed + int(iteration)
        cap = m.capacity

        def gatherInto(cap):
            gather, mode = r.buildDensifyOutputMapPlanned(actions, offsets, cap)
            if packed:
                r.densifyGatherPlannedPacked(p, gather, mode, seed, m.stagingBase(cap), cap, ARENA_ORDER)
            else:
                r.densifyGatherPlanned(p, gather, mode, seed, m.stagingViews(cap, stride=cap), cap)

        gatherInto(cap)
        m.zeroOptimizerBuffers(grads=packed or not self.fuse_adam)      # the reference re-creates the optimizer state at every cadence
        plan = r.densifyPlanRead(wait=True)                   # the plan alone: the gather and the resets are still queued
        st = {k: plan[k] for k in ("total", "keep", "split", "clone", "prune")}
        self.lastDensifyStats = st
        if self._exchange:
            self.checkPlans(plan)                              # before anything is sized by this rank's own N_new
        N_new = plan["N_new"]
        if N_new > cap:
            cap = int(N_new * 1.5)
            gatherInto(cap)
            if not packed:
                m._staged = (cap, cap, cap)
        m.commitStaged(N_new=N_new, zero=False)
        self._committed = True
        if plan["applies"] and (st["split"] > 0 or st["clone"] > 0):
            r.dropDepthCuts()         # (new Gaussians lengthen the sweeps: see split_and_prune)
        if r.reserved is not None and N_new > r.reserved[0]:
            r.reserve(N_new, int(r.reserved[1] * (N_new / max(r.reserved[0], 1)) * 1.1))
        self._seg_end = (C.c_longlong * 6)(*[int(x) for x in m.seg_end])
        self._alloc_exchange_buffers()
        self.resetGradientAccumulation()
        if self._exchange:
            self.checkReplicas(deferred=True)      # queued behind the gather; the verdict where the host next waits
        return st

    def split_and_prune(self, iteration: int):
        """GaussianTrainer.swift:766-907.  Every rank runs it on identical inputs (parameters are replicated, the
        accumulators are all-reduced, the noise comes from a generator seeded by (noise_seed, iteration)), so the new
        model is identical on every rank with no further communication.  Returns the action counts or None."""
        if not (self.densifyFromIter <= iteration <= self.densifyUntilIter):
            return None
        r, m = self.gaussRender, self.model
        N = m.N
        if N <= 0:
            self.resetGradientAccumulation()
            return None
        allowDensify = N < self.maxGaussians
        if self.xyzGradAccumulation.shape[0] != N:
            self.resetGradientAccumulation()
        if self._plans_events() and (self.noiseSource or "library") == "library":
            return self._split_and_prune_planned(iteration, allowDensify)
        self._resolveReplicaCheck()
        if self._native:
            r._check(r.lib.gs_dp_allreduce_sum(r.ctx, _p(self.xyzGradAccumulation), int(N)))
        elif self._exchange:
            import torch.distributed as dist
            dist.all_reduce(self.xyzGradAccumulation, op=dist.ReduceOp.SUM, group=self.pg)
        p = m.getParams()
        actions, counts = r.classifyGaussians(self.xyzGradAccumulation, float(self.denomGradAccumulation), p["scales"],
                                              p["opacity"].reshape(-1), self.gradientThreshold, self.maxScale,
                                              self.minOpacity, allowDensify)
        offsets, st = r.densifyOffsets(actions, counts)
        self.lastDensifyStats = st
        total = st["total"]
        if total <= 0 or (st["split"] == 0 and st["clone"] == 0 and st["prune"] == 0):
            self.resetGradientAccumulation()          # all pruned (:828-832) or nothing to do (:819-826, :843-847)
            return st
        gather, mode = r.buildDensifyOutputMap(actions, offsets, total)
        noise = None
        if (st["split"] > 0 or st["clone"] > 0) and (self.noiseSource or "torch") == "library":
            noise = r.densifyNoise(self.noise_seed + int(iteration), total)
        elif st["split"] > 0 or st["clone"] > 0:
            gen = torch.Generator(device=r.device)
            gen.manual_seed(self.noise_seed + int(iteration))
            noise = torch.randn(total, 3, generator=gen, device=r.device, dtype=torch.float32)
        r.densifyGather(p, gather, mode, noise, out=m.stagingViews(total))
        m.commitStaged()
        self._committed = True
        if self.referenceParamReload:
            self._committed_params = m.arena.clone()
        # The model changed.  New Gaussians (splits, clones) lengthen the sweeps: a stale cut then costs a whole repeated
        # forward, a fresh one 60 us of binning -- measured: every view missed once after such an event --, so the cuts go.
        # An event that only PRUNED (every event of a scene at the maxGaussians cap, GaussianTrainer.swift:300, 808) removes
        # splats of opacity < 0.005: no sweep gets longer by more than the cuts' margin (2 x the sweep + 128 entries), and
        # the cuts are depth keys, not indices, so the compaction of the arrays does not touch them.  They stay -- with 100
        # views and an event every 100 iterations every visit would otherwise be an uncut one (the 2 M garden scene: 1.85
        # instead of 0.47 ms of binning per step).  A cut that does miss is caught as ever (renderer.forwardMissed).
        if st["split"] > 0 or st["clone"] > 0:
            r.dropDepthCuts()
        if r.reserved is not None and total > r.reserved[0]:
            r.reserve(total, int(r.reserved[1] * (total / max(r.reserved[0], 1)) * 1.1))
        self._seg_end = (C.c_longlong * 6)(*[int(x) for x in m.seg_end])
        self._alloc_exchange_buffers()
        self.resetGradientAccumulation()
        if self._exchange:
            self.checkReplicas()       # before the next size-dependent collective
        return st

    def checkReplicas(self, deferred: bool = False):
        """Every rank of a data-parallel job calls this at the same point (after every committed densify event): one
        fixed-size collective over (N, checksum, checksum of magnitudes) of the parameter arena; ReplicaMismatch on EVERY
        rank if the replicas differ -- a diverged N would otherwise hang the job in the next all-gather.
        deferred (round 6, the planned event): the checksum and its collective are QUEUED, the verdict is taken by
        _resolveReplicaCheck where the host next waits for the device anyway (the overflow look every overflowCheckInterval
        steps, the next event, closeExchange) -- the event does not drain the queue; what a hang would come from, a diverged
        count, checkPlans has caught at once."""
        r, m = self.gaussRender, self.model
        self._resolveReplicaCheck()
        if self._native:
            if deferred:
                r._check(r.lib.gs_dp_check_replicas_begin(r.ctx, int(m.N), _p(m.arena), int(m.numel)))
                self._replica_pending = True
                return
            rc = r.lib.gs_dp_check_replicas(r.ctx, int(m.N), _p(m.arena), int(m.numel))
            if rc == GS_ERR_REPLICA_MISMATCH:
                raise ReplicaMismatch(r.lib.gs_last_error(r.ctx).decode())
            r._check(rc)
        elif deferred:
            self._replica_pending = check_replicas_begin(m.N, m.arena, self.pg)
        else:
            check_replicas(m.N, m.arena, self.pg)

    def _resolveReplicaCheck(self):
        """The verdict of a deferred checkReplicas, if one is outstanding (every rank reaches this at the same points)."""
        pending, self._replica_pending = getattr(self, "_replica_pending", None), None
        if not pending:
            return
        if self._native:
            r = self.gaussRender
            rc = r.lib.gs_dp_check_replicas_end(r.ctx)
            if rc == GS_ERR_REPLICA_MISMATCH:
                raise ReplicaMismatch(r.lib.gs_last_error(r.ctx).decode())
            r._check(rc)
        else:
            check_replicas_end(pending, self.pg, self.rank)

    def checkPlans(self, plan: dict):
        """The ranks' densify plans compared (gs_dp_check_plan / check_plans): one fixed-size collective on a side stream, which
        does not wait for this stream's queue; ReplicaMismatch on EVERY rank unless all planned the same event."""
        words = [int(plan[k]) for k in PLAN_WORDS]
        if self._native:
            r = self.gaussRender
            rc = r.lib.gs_dp_check_plan(r.ctx, (C.c_longlong * 8)(*words), 8)
            if rc == GS_ERR_REPLICA_MISMATCH:
                raise ReplicaMismatch(r.lib.gs_last_error(r.ctx).decode())
            r._check(rc)
        else:
            if getattr(self, "_side", None) is None and self.gaussRender.device.type == "cuda":
                self._side = torch.cuda.Stream(device=self.gaussRender.device)
            check_plans(words, self.pg, self.gaussRender.device, getattr(self, "_side", None), self.rank)

    def _recover_overflow(self):
        """A forward needed more (Gaussian, tile) pairs than were reserved (include/gsplat.h, "Overflow"): the device
        gate has kept parameters and moments untouched for every step taken from such a forward, so nothing is
        corrupted -- regrow the reserve to 1.5x what was needed and carry on."""
        r = self.gaussRender
        r.lib.gs_sync(r.ctx)                              # waits; reports (and clears) the deferred error
        st = r.stats()
        need = int(st["M"]) if st["overflow"] else int(st["capM"])
        capN = max(int(st["capN"]), self.model.capacity)
        r.reserve(capN, max(int(need * 1.5) + 65536, int(st["capM"] * 1.5)))
        self.overflowRecoveries += 1

    def _overflow_reported(self) -> bool:
        """Has the device raised an overflow report that nobody has taken delivery of?  No wait: the report lives in host
        memory the device writes (gs_overflow_pending).  Behind a densify event -- whose count read has just waited for every
        forward queued before it -- that answer is complete, so the event needs no second wait to learn that nothing
        happened (round 3 called gs_sync there: a second drain of the queue per event)."""
        r = self.gaussRender
        rep = (C.c_uint32 * 2)()
        r._check(r.lib.gs_overflow_pending(r.ctx, rep))
        return rep[0] != 0

    def checkOverflow(self):
        """Waits for the device and regrows the pair reserve if any forward since the last check overflowed it.
        Returns True if it had to."""
        try:
            self.gaussRender.sync()
        except GsplatError as e:
            if e.code != GS_ERR_WORKSPACE_OVERFLOW:
                raise
            self._recover_overflow()
            return True
        return False

    def _collectiveOverflowCheck(self, force: bool = False):
        """Data-parallel: every rank calls this at the same iterations.  Reads the `seen` word the optimizer kernels set when
        the step's gate (the OR of every rank's overflow word, carried by the step's first collective) made them skip -- the
        same on every rank, one wait; if any step since the last check was gated, the ranks agree on the largest pair count
        needed and every one regrows its reserve to 1.5x that.  Returns True if it did."""
        r = self.gaussRender
        self._resolveReplicaCheck()           # (this look waits for the device anyway)
        if self._native:
            regrown, need = C.c_int(), C.c_longlong()
            r._check(r.lib.gs_dp_check_overflow(r.ctx, C.byref(regrown), C.byref(need)))
            if regrown.value:
                st = r.stats()
                r.reserved = (int(st["capN"]), int(st["capM"]))
                self.overflowRecoveries += 1
            return bool(regrown.value)
        import torch.distributed as dist
        if not force and not bool(self._seen.item()):
            return False
        # What to regrow to comes from the report of the forward that TRIPPED (the library keeps it until it is delivered),
        # not from the last forward's counters: the overflowing step need not be the last of the window, and with views
        # visited round-robin it never is for some views -- every rank would then compute "nothing needed", clear the word
        # and lose that view's steps again and again.
        # (the report is written by kernels on the CTX's stream -- the one the renderer captured when it was built, not
        # necessarily torch's current one: wait for THAT stream, or the report may not have landed, gs_sync below would then
        # deliver and clear it unseen, and the regrow would be sized from the last forward's counters after all)
        r._check(r.lib.gs_wait(r.ctx))
        rep = (C.c_uint32 * 2)()
        r._check(r.lib.gs_overflow_pending(r.ctx, rep))
        kind = int(rep[0])
        need = int(rep[1]) if kind == 1 else 0         # (the pair count belongs to a kind-1 report; a kind-2 one never wrote it)
        try:
            r.sync()                                   # takes delivery of (and clears) this rank's deferred report
        except GsplatError as e:
            if e.code != GS_ERR_WORKSPACE_OVERFLOW:
                raise
        st = r.stats()
        if kind == 0 and st["overflow"]:               # (the last forward itself, not yet reported)
            kind, need = 1, int(st["M"])
        if kind == 2:                                  # checkpoint arena: the pairs fitted; gs_ctx_reserve regrows the arena
            need = max(need, int(st["capM"]))
        elif kind == 0:
            need = 0
        self._need.fill_(need)
        dist.all_reduce(self._need, op=dist.ReduceOp.MAX, group=self.pg)
        need = int(self._need.item())
        self._seen.zero_()
        if need <= 0:
            return False
        capN = max(int(st["capN"]), self.model.capacity)
        # a rank whose own forwards fitted regrows too (need is the maximum over the ranks): replicas keep equal reserves
        r.reserve(capN, int(st["capM"]) if need <= int(st["capM"]) else max(int(need * 1.5) + 65536, int(st["capM"])))
        self.overflowRecoveries += 1
        return True

    # -- exchange timing (measurement only; bench.py's `exchange` block) ---------------------------------------------
    def _gatherColourCotangents(self, xt, inline=None):
        """The step's all-gather of the colour-cotangent blocks (+ gate words) through torch.distributed.  Where nothing is queued
        between it and the kernel that needs its result (the round-6 step: the SH kernel is next) and the backend is RCCL, it is
        issued as a SYNCHRONOUS op: ProcessGroupNCCL then launches it on the current -- the render -- stream itself, with no event
        hop to its own stream and back (1-rank group: 46 us of exposed wait -> the collective's own ~9; the library's issuer
        does the same through a split communicator, dp.hip).  Otherwise (gloo; a projection backward to hide it under) it is
        queued asynchronously and waited for where its result is needed.  Returns the Work to wait for, or None."""
        import torch.distributed as dist
        if inline is None:
            inline = self.inlineGather
        if inline:
            self._xt_mark(xt, "wg0")
            dist.all_gather_into_tensor(self._cc_all.view(-1), self._cc_local.view(-1), group=self.pg, async_op=False)
            self._xt_mark(xt, "wg1")
            return None
        gather = dist.all_gather_into_tensor(self._cc_all.view(-1), self._cc_local.view(-1), group=self.pg, async_op=True)
        if xt is not None:
            xt["work"].update(gather=gather)
        return gather

    def _awaitGather(self, gather, xt):
        if gather is not None:
            self._xt_mark(xt, "wg0")
            gather.wait()
            self._xt_mark(xt, "wg1")

    def exchangeTimingBegin(self):
        """From now on every data-parallel step is timed: how long each collective took and how long the render stream
        stood waiting for it (wire time NOT hidden under compute).  Native exchange: the library's own events around its
        RCCL calls (gs_dp_exchange_timing).  Torch exchange: events on the render stream around every Work.wait(), and the
        collectives' own durations from Work._get_duration() where the backend keeps them (ProcessGroupNCCL with
        TORCH_NCCL_ENABLE_TIMING=1; gloo does not)."""
        if not self._exchange:
            return
        r = self.gaussRender
        if self._native:
            r._check(r.lib.gs_dp_exchange_timing(r.ctx, 1))
        self._xt = []

    def _xt_step(self):
        if self._xt is None or self._native or len(self._xt) >= 512:
            return None
        st = dict(ev={}, work={})
        self._xt.append(st)
        return st

    @staticmethod
    def _xt_mark(st, name):
        if st is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            st["ev"][name] = e

    def exchangeTimingRead(self):
        """Per-step averages in ms since exchangeTimingBegin (waits for the device), or None if nothing was timed."""
        if self._xt is None:
            return None
        r, m = self.gaussRender, self.model
        if self._native:
            ms, steps, ver = (C.c_float * 8)(), C.c_int(), C.c_int()
            r._check(r.lib.gs_dp_exchange_read(r.ctx, ms, C.byref(steps), C.byref(ver)))
            r._check(r.lib.gs_dp_exchange_timing(r.ctx, 0))
            sums = dict(gate=ms[0], gather=ms[1], reduce=ms[2], exposed_gather=ms[3], exposed_reduce=ms[4])
            counts = dict(gate=steps.value, gather=steps.value, reduce=steps.value)
            n, version = steps.value, ver.value
            source = ("HIP events on the library's side stream around each RCCL call, and on the ctx stream around its "
                      "hipStreamWaitEvent on them (gs_dp_exchange_read)")
        else:
            torch.cuda.synchronize(r.device)
            n = len(self._xt)
            sums = dict(gate=0.0, gather=0.0, reduce=0.0, exposed_gather=0.0, exposed_reduce=0.0)
            counts = dict(gate=0, gather=0, reduce=0)
            for st in self._xt:
                ev = st["ev"]
                for a, b, k in (("wg0", "wg1", "exposed_gather"), ("wr0", "wr1", "exposed_reduce")):
                    if a in ev and b in ev:
                        sums[k] += ev[a].elapsed_time(ev[b])
                for k, w in st["work"].items():
                    try:
                        sums[k] += float(w._get_duration())
                        counts[k] += 1
                    except Exception:
                        pass
            try:
                import torch.cuda.nccl as _nccl
                v = _nccl.version()
                version = v[0] * 10000 + v[1] * 100 + v[2] if isinstance(v, tuple) else int(v)
            except Exception:
                version = None
            source = ("torch.cuda events on the render stream around every Work.wait(); collective durations from "
                      "Work._get_duration() (null where the backend keeps none)")
        self._xt = None
        return exchange_summary(self.exchange_impl, self.dp_exchange, self.world, m.N, int(m.geom_numel), int(m.numel), n, sums,
                                counts, version, source, self.viewsPerRank)

    def trainStep(self, camera, targetRGB, stepCameras=None, viewKey=None):
        """One iteration: forward, loss, backward, (gradient exchange), Adam.  Asynchronous; returns the device
        loss[4].  stepCameras: the cameras of ALL ranks for this step in rank order (every rank derives them from the
        shared view permutation, see view_for), or just their centres [R,3]; required by the sh_compressed exchange.
        viewKey: identifies the training view (renderer.renderForward): enables the forward's deepest-first order.

        Reserved-capacity overflow: the step's calls raise GS_ERR_WORKSPACE_OVERFLOW as soon as the host sees the
        device's flag (at the latest at the checks below: the first visit of every view, every densify cadence); the
        reserve is regrown and the step repeated once.  Steps queued in between were skipped on the device (no
        optimizer update from a blank render), never applied."""
        profiled = self.enableIntervalProfiling and (self.iteration % self.profilingLogInterval == 0
                                                     or self.iteration == self.iterationCount - 1)
        step = self._profiledStep if profiled else (self._trainStepMulti if self.viewsPerRank > 1 else self._trainStep)
        r = self.gaussRender
        # knobs of the caller's renderer that this step changes, put back whatever happens
        restore = dict(depth_gradient=r.getTuning("depth_gradient"), host_overflow_errors=r.getTuning("host_overflow_errors"))
        try:
            r.setTuning(depth_gradient=0)
            if self._exchange:
                r.setTuning(host_overflow_errors=0)
                if self.iteration % self.overflowCheckInterval == 0 and self.iteration > 0:
                    self._collectiveOverflowCheck()
            for attempt in range(4):
                try:
                    return step(camera, targetRGB, stepCameras, viewKey)
                except GsplatError as e:
                    if e.code != GS_ERR_WORKSPACE_OVERFLOW or self._exchange or attempt == 3:
                        raise      # (data-parallel steps never raise it: host_overflow_errors is off, see __init__)
                    self._recover_overflow()      # (every regrow is by half at least: a few rounds reach any need)
        finally:
            r.setTuning(**restore)

    def _profiledStep(self, camera, targetRGB, stepCameras, viewKey):
        """One iteration under the reference's IntervalProfiler: host sections by wall clock, device stages by the
        library's HIP events (this iteration waits for the device at its end; the others never do)."""
        import time
        from .profiler import IntervalProfiler
        r = self.gaussRender
        prof = IntervalProfiler(True)
        r.profiler = prof
        it = self.iteration
        t0 = time.perf_counter_ns()
        r.profile(True)
        try:
            body = self._trainStepMulti if self.viewsPerRank > 1 else self._trainStep
            out = prof.measure("train.valueAndGrad.execute", lambda: body(camera, targetRGB, stepCameras, viewKey))
            prof.setDeviceStages(r.profileRead())
        finally:
            r.profile(False)
            r.profiler = None
        self.lastProfileReport = prof.makeReport(it, time.perf_counter_ns() - t0, self.profilingTopKSections, 0.01)
        self.lastProfiler = prof
        if self.log:
            self.log(self.lastProfileReport)
        return out

    def _nativeStep(self, stepCameras):
        """Backward + exchange + Adam of this step through gs_dp_step (the library issues the RCCL calls)."""
        from . import _lib
        r, m = self.gaussRender, self.model
        a = _lib.gs_dp_step_args()
        a.cot_color, a.cot_depth, a.cot_alpha = self._cot.data_ptr(), None, None
        a.params_base, a.grads_base, a.m_base, a.v_base = (t.data_ptr() for t in (m.arena, m.grad, m.m, m.v))
        a.n_arena, a.geom_numel, a.nseg = int(m.numel), int(m.geom_numel), 6
        for i, (e, lr) in enumerate(zip(m.seg_end, arenaLearningRates(self.iteration, self.iterationCount))):
            a.seg_end[i], a.seg_lr[i] = int(e), float(lr)
        a.beta1, a.beta2, a.eps = 0.9, 0.999, 1e-15
        mode = _lib.GS_DP_ALLREDUCE
        if self.dp_exchange == "sh_compressed":
            if stepCameras is None or len(stepCameras) != self.world:
                raise ValueError("sh_compressed exchange needs stepCameras (one camera per rank, rank order)")
            centres = np.ascontiguousarray(np.stack([np.asarray(getattr(c, "cameraCenter", c), np.float32).reshape(3)
                                                     for c in stepCameras]), np.float32)
            a.cam_centers = centres.ctypes.data
            a.color_cot_local, a.color_cot_all = self._cc_local.data_ptr(), self._cc_all.data_ptr()
            mode = _lib.GS_DP_SH_COMPRESSED
        r._cuts_renewed()
        r._check(r.lib.gs_dp_step(r.ctx, mode, C.byref(a)))

    def _beginIteration(self):
        r, m = self.gaussRender, self.model
        if self.referenceParamReload and self._committed_params is None:
            self._committed_params = m.arena.clone()      # what the reference's model holds: the tensors before any step
        if self.densify:
            if self.xyzGradAccumulation.shape[0] != m.N:
                self.resetGradientAccumulation()
            if getattr(r, "_grad_norm_accum", None) is not self.xyzGradAccumulation:
                r.setGradNormAccum(self.xyzGradAccumulation)      # the backward below adds this view's |grad xyz|
        elif getattr(r, "_grad_norm_accum", None) is not None:
            r.setGradNormAccum(None)

    def _forwardAndLoss(self, camera, targetRGB, viewKey, lossOut):
        """lossFn of one view (GaussianTrainer.swift:627-723): forward, L1 + DSSIM loss -> lossOut[4] and the colour cotangent
        in self._cot; the forward is repeated once if it overflowed on the view's first visit, and once without depth cuts if
        it missed under them."""
        r, m = self.gaussRender, self.model
        res = r._measure("train.forward", lambda: r.renderForward(m.getParams(), camera, viewKey=viewKey, wantDepth=False))
        if viewKey is not None and viewKey not in self._checked_views:
            # first visit of a view: its pair count is unknown -- wait for the forward once and make sure it fitted
            # (rank-local also in a data-parallel job: a forward is no collective, and gs_sync reports whatever the knob says)
            self._checked_views.add(viewKey)
            if self.checkOverflow():
                res = r.renderForward(m.getParams(), camera, viewKey=viewKey, wantDepth=False)
        r._measure("train.loss.total", lambda: r.lossForwardBackward(res.render, targetRGB, self.lambda_dssim,
                                                                      out=dict(loss=lossOut, cotColor=self._cot),
                                                                      targetKey=viewKey))
        # depth cuts (renderer.renderForward): nothing that changes state has been queued yet; the loss kernel above
        # keeps the GPU busy while the host learns whether the forward has to be repeated in full
        if viewKey is not None and r.forwardMissed():
            self.forwardMisses += 1
            res = r.renderForward(m.getParams(), camera, viewKey=viewKey, depthCuts=False, wantDepth=False)
            r.lossForwardBackward(res.render, targetRGB, self.lambda_dssim, out=dict(loss=lossOut, cotColor=self._cot),
                                  targetKey=viewKey)

    def _geometryAdam(self, lr: dict, scale: float):
        """Adam on the geometry slice after the all-reduce; the xyz segment's gradient = reduced + the view-direction terms the
        SH kernel rebuilt meanwhile (gs_adam_step_add)."""
        r, m = self.gaussRender, self.model
        glr = (C.c_float * 4)(lr["xyz"], lr["scales"], lr["rotation"], lr["opacity"])
        r._check(r.lib.gs_adam_step_add(r.ctx, m.geom_numel, _p(m.arena), _p(m.grad), _p(m.m), _p(m.v), 4,
                                        (C.c_longlong * 4)(*[int(x) for x in m.seg_end[:4]]), glr, C.c_float(0.9), C.c_float(0.999),
                                        C.c_float(1e-15), C.c_float(scale), _p(self._xyz_add), 3 * m.N))

    def _trainStepMulti(self, cameras, targets, stepCameras=None, viewKeys=None):
        """One iteration over viewsPerRank views of this rank (see __init__): cameras / targets / viewKeys are sequences of
        that length, stepCameras the cameras (or centres) of ALL world x V views of the step, rank-major.  Per view: lossFn,
        then the data-parallel backward -- colour cotangents and the view's gate word into the view's block, the four geometry
        gradients into the view's slice, |grad xyz| into the densify statistic -- exactly what a rank of a one-view-per-rank
        job does; then ONE update: geometry slices summed (+ all-reduce), blocks gathered (or simply there), SH gradients
        rebuilt over all views with their Adam step, geometry Adam, both at grad_scale 1 / (world x V)."""
        r, m = self.gaussRender, self.model
        V, N = self.viewsPerRank, m.N
        if len(cameras) != V or len(targets) != V or (viewKeys is not None and len(viewKeys) != V):
            raise ValueError(f"views_per_rank = {V}: trainStep takes {V} cameras, targets and view keys")
        total = self.world * V
        if stepCameras is None and not self._exchange:
            stepCameras = cameras
        if stepCameras is None or len(stepCameras) != total:
            raise ValueError(f"sh_compressed exchange needs stepCameras: {total} cameras (ranks x views_per_rank, rank-major)")
        self._beginIteration()
        ccf = cc_block_floats(N)
        blocks = self._cc_local.view(V, ccf)
        for j in range(V):
            key = None if viewKeys is None else viewKeys[j]
            self._forwardAndLoss(cameras[j], targets[j], key, self._loss_v[j])
            # the view's gate word goes behind ITS block (gs_set_overflow_rider: the word of the last forward)
            r._check(r.lib.gs_set_overflow_rider(r.ctx, C.c_void_p(blocks[j].data_ptr() + 12 * N)))
            if self.fuse_adam:
                r.renderBackwardDPGeom(self._cot, blocks[j], self._geom_views[j], self._xyz_own[j])
            else:
                r.renderBackwardDPBegin(self._cot, colorCot=blocks[j])
                r.renderBackwardDPFinish(out=self._geom_views[j])
            if self.densify:
                self.addGradientAccumulation()
        g_geom = m.grad[:m.geom_numel]
        torch.sum(self._geom_v, dim=0, out=g_geom)               # the rank's share of the all-reduce, summed by addressing
        reduce = None
        xt = self._xt_step() if self._exchange else None
        if self._exchange:
            import torch.distributed as dist
            gather = self._gatherColourCotangents(xt)
            reduce = dist.all_reduce(g_geom, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            if xt is not None:
                xt["work"].update(reduce=reduce)
            self._awaitGather(gather, xt)
        centres = np.stack([np.asarray(getattr(c, "cameraCenter", c), np.float32).reshape(3) for c in stepCameras])
        scale = 1.0 / total
        if self.fuse_adam:
            lr = dict(zip(PARAM_ORDER, getLearningRates(self.iteration, self.iterationCount)))
            own = [None] * total
            for j in range(V):
                own[self.rank * V + j] = self._xyz_own[j]
            r.shGradFromViewsAdamDir(m.getParams(), self._cc_all, centres, own, m.arena, m.m, m.v, lr["features_dc"],
                                     lr["features_rest"], scale, self._xyz_add)
            if reduce is not None:
                self._xt_mark(xt, "wr0")
                reduce.wait()
                self._xt_mark(xt, "wr1")
            self._geometryAdam(lr, scale)
        else:
            r.shGradFromViews(m.getParams()["xyz"], self._cc_all, centres, m.K, out=m.getGrads())
            if reduce is not None:
                reduce.wait()
            lrs = (C.c_float * 6)(*arenaLearningRates(self.iteration, self.iterationCount))
            r._check(r.lib.gs_adam_step(r.ctx, m.numel, _p(m.arena), _p(m.grad), _p(m.m), _p(m.v), 6, self._seg_end, lrs,
                                        C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), C.c_float(scale)))
        torch.mean(self._loss_v, dim=0, out=self._loss)          # this rank's views; the step's loss is the mean over all of them
        return self._finishIteration()

    def _trainStep(self, camera, targetRGB, stepCameras=None, viewKey=None):
        r, m = self.gaussRender, self.model
        self._beginIteration()
        self._forwardAndLoss(camera, targetRGB, viewKey, self._loss)
        xt = None
        if self._gate is not None:
            # (this step's gate: the backward's first kernel stores the forward's overflow word behind what the step's first
            # collective carries -- the word of the LAST forward, i.e. of a forward repeated without depth cuts if there was
            # one; a rank that repeats adds no collective)
            xt = self._xt_step()
        fused = False
        if self._native:
            self._nativeStep(stepCameras)
            if self.densify:
                self.addGradientAccumulation()
            fused = True
        elif not self._exchange and self.fuse_adam:
            r._measure("bwd.fused+train.optimizer.applySingle", lambda: r.renderBackwardAdam(
                self._cot, m.arena, m.m, m.v, getLearningRates(self.iteration, self.iterationCount)))
            if self.densify:
                self.addGradientAccumulation()
            fused = True
        elif not self._exchange:
            r.renderBackward(self._cot, out=m.getGrads())
            if self.densify:
                self.addGradientAccumulation()
        elif self.dp_exchange == "allreduce":
            r.renderBackward(self._cot, out=m.getGrads())          # adds this view's |grad xyz| before the sum below
            if self.densify:
                self.addGradientAccumulation()
            self._xt_mark(xt, "wr0")                     # (a blocking collective: all of it is exposed)
            allreduce_gradients(m._gbuf[:m.numel + 1], self.pg)      # the arena and, behind it, the step's gate word
            self._xt_mark(xt, "wr1")
        else:
            if stepCameras is None or len(stepCameras) != self.world:
                raise ValueError("sh_compressed exchange needs stepCameras (one camera per rank, rank order)")
            import torch.distributed as dist
            g = m.getGrads()
            # the colour cotangents are ready after the blend backward: their all-gather runs under the projection
            # backward, and the rebuild of the SH gradients under the all-reduce of the geometry slice
            if self.fuse_adam:
                # round 6: no SH rows in the geometry backward (the SH kernel below rebuilds the view-direction term of the xyz
                # gradient and adds the densify statistic), and the colour cotangents + gate word ride in that kernel
                r.renderBackwardDPGeom(self._cot, self._cc_local, g, self._xyz_own[0])
                gather = self._gatherColourCotangents(xt)
            else:
                r.renderBackwardDPBegin(self._cot, colorCot=self._cc_local)      # + this rank's word of the gate at [3 N]
                gather = self._gatherColourCotangents(xt, inline=False)          # (runs under the projection backward below)
                r.renderBackwardDPFinish(out=g)
            if self.densify:
                self.addGradientAccumulation()
            reduce = dist.all_reduce(m.grad[:m.geom_numel], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            centres = np.stack([np.asarray(getattr(c, "cameraCenter", c), np.float32).reshape(3) for c in stepCameras])
            if xt is not None:
                xt["work"].update(reduce=reduce)
            self._awaitGather(gather, xt)
            if self.fuse_adam:
                # SH tensors: gradient rebuild + Adam in one pass (old xyz: the geometry step comes after); then the
                # geometry slice alone goes through gs_adam_step
                lr = dict(zip(PARAM_ORDER, getLearningRates(self.iteration, self.iterationCount)))
                own = [self._xyz_own[0] if q == self.rank else None for q in range(self.world)]
                r.shGradFromViewsAdamDir(m.getParams(), self._cc_all, centres, own, m.arena, m.m, m.v, lr["features_dc"],
                                         lr["features_rest"], 1.0 / self.world, self._xyz_add)
                self._xt_mark(xt, "wr0")
                reduce.wait()
                self._xt_mark(xt, "wr1")
                self._geometryAdam(lr, 1.0 / self.world)
                fused = True
            else:
                r.shGradFromViews(m.getParams()["xyz"], self._cc_all, centres, m.K, out=g)
                self._xt_mark(xt, "wr0")
                reduce.wait()
                self._xt_mark(xt, "wr1")
        if not fused:
            lrs = (C.c_float * 6)(*arenaLearningRates(self.iteration, self.iterationCount))
            r._check(r.lib.gs_adam_step(r.ctx, m.numel, _p(m.arena), _p(m.grad), _p(m.m), _p(m.v), 6, self._seg_end, lrs,
                                        C.c_float(0.9), C.c_float(0.999), C.c_float(1e-15), C.c_float(1.0 / self.world)))
        return self._finishIteration()

    def _finishIteration(self):
        r, m = self.gaussRender, self.model
        it = self.iteration
        self.iteration += 1
        if self.outputDirectory is not None and it % self.save_snapshot_per_iteration == 0:
            self.save_snapshot(it)
        if self.densify and it % self.split_and_prune_per_iteration == 0:
            self._committed = False
            self.split_and_prune(it)
            # the reference re-creates the optimizer state after every call, changed or not (:1098-1110); a committed
            # event has just done so (GaussModel.commitStaged)
            if not self._committed:
                m.resetOptimizerState()
                if self.referenceParamReload and self._committed_params is not None \
                        and self._committed_params.numel() == m.arena.numel():
                    m.arena.copy_(self._committed_params)      # `params = model.getParams()` (:1100): the last COMMITTED tensors
            # the event has just waited for the device (its .item()): the overflow flag is cheap to look at now, and in a
            # data-parallel job every rank is here at the same iteration
            if self._exchange and not self._plans_events():
                self._collectiveOverflowCheck(force=self._committed)
            elif self._exchange:
                pass        # a planned event has not drained the queue: the ranks look together at the next cadence (every overflowCheckInterval steps)
            elif self._overflow_reported():
                self.checkOverflow()
        return self._loss
