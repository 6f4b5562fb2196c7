"""world_size-2 gloo test of the data-parallel exchange (CPU): each rank differentiates its own view with the
oracle, the flat gradient arena is all-reduced once, and the result equals the single-process sum."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    rng = np.random.default_rng(77)
    N, K = 300, 25
    p = dict(xyz=rng.uniform(-0.7, 0.7, (N, 3)), features_dc=rng.normal(0, 1, (N, 1, 3)),
             features_rest=rng.normal(0, 0.05, (N, K - 1, 3)), scales=rng.normal(np.log(0.06), 0.4, (N, 3)),
             rotation=rng.normal(0, 1, (N, 4)), opacity=rng.normal(0.2, 1.0, N))
    p = {k: v.astype(np.float32) for k, v in p.items()}
    cams = [Camera(64, 48, 60.0, 60.0, look_at_c2w(e)) for e in ([2.0, -2.5, 1.5], [-2.2, 2.0, 1.8], [0.5, 3.0, 1.2])]
    return p, cams


def _view_grads(p, cam, W=64, H=48):
    from oracle.oracle import Oracle
    o = Oracle(np.float32)
    c = cam.as_dict()
    fw = o.render_forward(p, c, W, H, 16, 16, 4)
    tgt = np.full((H, W, 3), 0.3, np.float32)
    _, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), tgt, 0.2)
    z = np.zeros(W * H, np.float32)
    return o.render_backward(p, c, W, H, 16, 16, 4, fw, cot.reshape(-1, 3), z, z)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussiansplattingmlx_amd.trainer import ARENA_ORDER, GaussModel, allreduce_gradients, view_for
        p, cams = _scene()
        model = GaussModel(p, torch.device("cpu"))
        v = view_for(0, rank, world, len(cams))
        g = _view_grads(p, cams[v])
        for k in ARENA_ORDER:
            model.getGrads()[k].copy_(torch.as_tensor(g[k].reshape(model.getGrads()[k].shape)))
        # second exchange on a copy of the local gradients: colour cotangents gathered, geometry slice reduced
        from gaussiansplattingmlx_amd.trainer import (ReplicaMismatch, cc_block_floats, check_replicas, exchange_sh_compressed,
                                                      gathered_gate)
        N = model.N
        ccf = cc_block_floats(N)
        g2 = model.grad.clone()
        # a rank's gather block: its colour cotangents, then its word of the step's gate (the backward's first kernel stores it
        # on the device, gs_set_overflow_rider); here rank 1's forward "overflowed"
        cc_local = torch.zeros(ccf)
        cc_local[:3 * N] = torch.as_tensor((g["features_dc"].reshape(-1) / np.float32(0.28209479177387814)).astype(np.float32))
        cc_local[3 * N] = 1.0 if rank == 1 else 0.0
        cc_all = torch.empty(world, ccf)
        exchange_sh_compressed(g2[:model.geom_numel], cc_local, cc_all, dist.group.WORLD)
        gate_sh = gathered_gate(cc_all, N)
        cc_local[3 * N] = 0.0                               # ... and a step in which nobody's did
        cc_quiet = torch.empty(world, ccf)
        exchange_sh_compressed(g2.clone()[:model.geom_numel], cc_local, cc_quiet, dist.group.WORLD)
        gate_quiet = gathered_gate(cc_quiet, N)
        # all-reduce exchange: the word rides behind the arena (GaussModel keeps a spare float there) and is summed with it
        model._gbuf[model.numel] = 1.0 if rank == 1 else 0.0
        scale = allreduce_gradients(model._gbuf[:model.numel + 1], dist.group.WORLD)
        gate_ar = float(model._gbuf[model.numel])
        # replica check: identical replicas pass; a perturbed arena or a different N raises on EVERY rank
        verdicts = []
        for case in ("same", "arena", "N"):
            arena = model.arena.clone()
            n = N
            if rank == 1 and case == "arena":
                arena[1234] += 1e-3
            if rank == 1 and case == "N":
                n = N + 1
            try:
                check_replicas(n, arena, dist.group.WORLD)
                verdicts.append("ok")
            except ReplicaMismatch as e:
                verdicts.append(str(e))
        # round 6, the planned densify event of a data-parallel step: the ranks compare their PLANS (gs_densify_plan_read's eight
        # words) in one fixed-size collective before anything is sized by a rank's own count, and the arena checksum is queued
        # (begin) and judged later (end).  Identical plans pass; a rank whose accumulator was perturbed plans another event --
        # more splits, another N_new -- and BOTH ranks raise
        from gaussiansplattingmlx_amd.trainer import check_plans, check_replicas_begin, check_replicas_end
        plan_verdicts = []
        for case in ("same", "perturbed"):
            words = [N + 40, 1, N + 40, N - 50, 30, 15, 5, N]          # N_new, applies, total, keep, split, clone, prune, N
            if rank == 1 and case == "perturbed":
                words[0] += 2; words[2] += 2; words[3] -= 1; words[4] += 1          # one more Gaussian crossed the threshold on rank 1
            try:
                check_plans(words, dist.group.WORLD, torch.device("cpu"))
                plan_verdicts.append("ok")
            except ReplicaMismatch as e:
                plan_verdicts.append(str(e))
        for case in ("same", "arena"):
            arena = model.arena.clone()
            if rank == 1 and case == "arena":
                arena[77] -= 1e-4
            pending = check_replicas_begin(N, arena, dist.group.WORLD)          # queued at the event ...
            try:
                check_replicas_end(pending, dist.group.WORLD)                   # ... judged where the host next waits
                plan_verdicts.append("ok")
            except ReplicaMismatch as e:
                plan_verdicts.append(str(e))
        q.put((rank, v, scale, model.grad.numpy().copy(), [int(x) for x in model.seg_end], g2.numpy().copy(),
               cc_all[:, :3 * N].reshape(world, N, 3).numpy().copy(), model.geom_numel, gate_sh, gate_quiet, gate_ar, verdicts,
               plan_verdicts))
    finally:
        dist.destroy_process_group()


def test_gradient_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=180) for _ in range(world)], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    assert [r[1] for r in res] == [0, 1] and all(abs(r[2] - 0.5) < 1e-12 for r in res)
    np.testing.assert_array_equal(res[0][3], res[1][3])          # every rank holds the same reduced arena
    from gaussiansplattingmlx_amd.trainer import ARENA_ORDER
    p, cams = _scene()
    seg_end = res[0][4]
    want = np.zeros(seg_end[-1], np.float32)          # every tensor's segment is padded to a multiple of four floats
    for k, off in zip(ARENA_ORDER, [0] + seg_end[:-1]):
        gk = sum(_view_grads(p, cams[v])[k].reshape(-1).astype(np.float32) for v in (0, 1))
        want[off:off + gk.size] = gk
    np.testing.assert_allclose(res[0][3], want, rtol=1e-6, atol=1e-9)
    # sh_compressed exchange: same reduced geometry slice; the SH slice rebuilt from the gathered cotangents
    from oracle.oracle import Oracle
    o = Oracle(np.float32)
    geom = res[0][7]
    assert geom == 300 * 11
    np.testing.assert_array_equal(res[0][5][:geom], res[0][3][:geom])
    np.testing.assert_array_equal(res[0][6], res[1][6])
    rebuilt = 0
    for rnk in (0, 1):
        d = p["xyz"].astype(np.float32) - cams[rnk].cameraCenter.astype(np.float32)[None, :]
        basis = np.stack([o.sh_basis(4, *row) for row in d]).astype(np.float64)
        rebuilt = rebuilt + basis[:, :, None] * res[0][6][rnk].astype(np.float64)[:, None, :]       # [N,25,3]
    sh_want = np.concatenate([res[0][3][geom:geom + 900].reshape(300, 1, 3), res[0][3][geom + 900:].reshape(300, 24, 3)], 1)
    np.testing.assert_allclose(rebuilt, sh_want, rtol=1e-4, atol=1e-6 * np.abs(sh_want).max())
    N = 300
    # 86 floats = 344 B per Gaussian; the 11 geometry floats lead so the compressed exchange reduces one slice
    assert res[0][4] == list(np.cumsum([N * 3, N * 3, N * 4, N, N * 3, N * 72]))
    # round 5, the folded gate: rank 1 alone raised its word -- BOTH ranks gate, through the all-gather (OR of the gathered
    # words) and through the all-reduce (the summed word, non-zero on every rank); a quiet step gates nobody
    for r in res:
        assert r[8] is True and r[9] is False and r[10] == 1.0, r[8:11]
    # ... and the replica check: identical replicas pass on both ranks; a perturbed arena or another N raises on BOTH, with
    # what differs named (the verdict is built from reduced values, so no rank goes on alone)
    for r in res:
        same, arena, n = r[11]
        assert same == "ok"
        assert "GS_ERR_REPLICA_MISMATCH" in arena and "(sum magnitudes)" in arena and "N in [300, 300]" in arena
        assert "GS_ERR_REPLICA_MISMATCH" in n and "(N)" in n and "N in [300, 301]" in n
    assert "rank 0 has N = 300" in res[0][11][2] and "rank 1 has N = 301" in res[1][11][2]
    # round 6: the planned event's plan check and the deferred checksum -- same on both ranks passes, a diverged plan (rank 1
    # planned two more outputs) or a diverged arena raises on BOTH ranks, naming what differs
    for r in res:
        same_plan, bad_plan, same_sum, bad_sum = r[12]
        assert same_plan == "ok" and same_sum == "ok"
        assert "GS_ERR_REPLICA_MISMATCH" in bad_plan and "N_new in [340, 342]" in bad_plan and "split in [30, 31]" in bad_plan
        assert "clone" not in bad_plan.split(";")[0]                               # (what agrees is not listed)
        assert "GS_ERR_REPLICA_MISMATCH" in bad_sum and "(sum magnitudes)" in bad_sum
    assert "rank 0 planned new N = 340" in res[0][12][1] and "rank 1 planned new N = 342" in res[1][12][1]


def test_sh_gradient_is_rank_one_in_colour_cotangent():
    """The identity the sh_compressed exchange rests on, checked on the oracle: a view's SH gradient equals
    basis_k(xyz - cam) x cc with cc = grad_features_dc / basis_0, so summing it over views from the gathered cc
    reproduces the all-reduced SH gradient."""
    from oracle.oracle import Oracle
    p, cams = _scene()
    o = Oracle(np.float32)
    total_dc, total_rest, rebuilt_dc, rebuilt_rest = 0, 0, 0, 0
    for v in (0, 1):
        g = _view_grads(p, cams[v])
        gdc, grest = g["features_dc"].reshape(-1, 3).astype(np.float64), g["features_rest"].reshape(-1, 24, 3)
        d = (p["xyz"].astype(np.float32) - cams[v].cameraCenter.astype(np.float32)[None, :])
        basis = np.stack([o.sh_basis(4, *row) for row in d]).astype(np.float64)        # [N,25]
        cc = gdc / basis[:, :1]
        total_dc, total_rest = total_dc + gdc, total_rest + grest
        rebuilt_dc = rebuilt_dc + basis[:, :1] * cc
        rebuilt_rest = rebuilt_rest + basis[:, 1:, None] * cc[:, None, :]
    np.testing.assert_allclose(rebuilt_dc, total_dc, rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(rebuilt_rest, total_rest, rtol=2e-5, atol=1e-6 * np.abs(total_rest).max())


def test_arena_learning_rates_follow_arena_order():
    from gaussiansplattingmlx_amd.trainer import ARENA_ORDER, PARAM_ORDER, arenaLearningRates, getLearningRates
    ref = dict(zip(PARAM_ORDER, getLearningRates(10, 1000)))
    assert arenaLearningRates(10, 1000) == [ref[k] for k in ARENA_ORDER]
    assert ARENA_ORDER[:4] == ("xyz", "scales", "rotation", "opacity")


def test_view_sharding_covers_every_view_once_per_epoch():
    from gaussiansplattingmlx_amd.trainer import getLearningRates, view_for
    V, world = 8, 4
    seen = [view_for(s, r, world, V) for s in range(V // world) for r in range(world)]
    assert sorted(seen) == list(range(V))
    lrs = getLearningRates(0, 30000)
    assert abs(lrs[0] - 0.00016) < 1e-12 and lrs[2] == 0.0025 / 20 and len(lrs) == 6
    assert abs(getLearningRates(30000, 30000)[0] - 0.0000016) < 1e-12


def test_exchange_block_of_the_bench_line():
    """The N > 1 bench line's `exchange` block (bench.py, trainer.exchange_summary): per-step averages of what was summed,
    null where the backend kept no duration, bytes per rank and step for both exchanges."""
    from gaussiansplattingmlx_amd.trainer import exchange_summary
    N, world = 300_000, 8
    geom, numel = N * 11, N * 86
    sums = dict(gate=0.2, gather=0.9, reduce=1.5, exposed_gather=0.3, exposed_reduce=0.1)
    x = exchange_summary("native", "sh_compressed", world, N, geom, numel, 10, sums, dict(gate=10, gather=10, reduce=10), 22606, "events")
    assert (x["gate_ms"], x["gather_ms"], x["reduce_ms"]) == (0.02, 0.09, 0.15)
    assert (x["exposed_gather_ms"], x["exposed_reduce_ms"], x["exposed_ms"]) == (0.03, 0.01, 0.04)
    # (round 5: the 4-byte gate rides behind the cotangents, a block is padded to four floats: 12 N + 16 bytes per rank)
    assert x["gather_bytes_out"] == 12 * N + 16 and x["gather_bytes_in"] == (12 * N + 16) * world and x["reduce_bytes"] == 4 * geom
    assert x["collectives_per_step"] == 2 and x["gate_rides_in"] == "gather"
    assert x["gate_bytes"] == 4 and x["world"] == 8 and x["steps_measured"] == 10 and x["rccl_version"] == 22606
    assert x["dp_impl"] == "native" and x["dp_exchange"] == "sh_compressed" and x["timing_source"] == "events"
    # gloo keeps no durations: nulls, but the exposed waits are still measured
    y = exchange_summary("torch", "allreduce", 2, N, geom, numel, 4, dict(sums, gate=0.0, gather=0.0, reduce=0.0),
                         dict(gate=0, gather=0, reduce=0), None, "events")
    assert y["gate_ms"] is None and y["gather_ms"] is None and y["reduce_ms"] is None and y["exposed_ms"] == 0.1
    assert y["gather_bytes_out"] == 0 and y["reduce_bytes"] == 4 * (numel + 1) and y["collectives_per_step"] == 1
    # a run that timed nothing does not divide by zero
    z = exchange_summary("torch", "sh_compressed", 2, 0, 0, 0, 0, dict(gate=0, gather=0, reduce=0, exposed_gather=0, exposed_reduce=0), {}, None, "")
    assert z["exposed_ms"] == 0.0 and z["gate_ms"] is None


def test_balanced_view_order_keeps_a_steps_views_alike():
    """Every window of `world` consecutive views of the balanced order -- cyclically -- holds views of similar cost, every view
    once per pass (trainer.balanced_view_order; bench.py --gpus N deals the views to the steps in this order)."""
    from gaussiansplattingmlx_amd.trainer import balanced_view_order
    rng = np.random.default_rng(3)
    costs = list(rng.normal(1.0, 0.05, 100))
    order = balanced_view_order(costs)
    assert sorted(order) == list(range(100))
    world = 8
    def step_costs(o):
        return [max(costs[o[(s * world + q) % 100]] for q in range(world)) for s in range(25)]      # two passes (100 views, 8 ranks)
    naive, balanced = np.mean(step_costs(list(range(100)))), np.mean(step_costs(order))
    mean = np.mean(costs)
    assert balanced < naive and balanced - mean < 0.35 * (naive - mean)          # most of the straggler tax is gone
    spread = [max(costs[order[(i + k) % 100]] for k in range(world)) - min(costs[order[(i + k) % 100]] for k in range(world)) for i in range(100)]
    assert max(spread) < 0.5 * (max(costs) - min(costs))                        # no window mixes the cheapest with the dearest
    assert balanced_view_order([]) == [] and balanced_view_order([5.0]) == [0]
    assert balanced_view_order([1, 1, 1]) in ([0, 2, 1],)                         # ties broken by index: every rank builds the same order
