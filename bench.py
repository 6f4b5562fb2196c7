#!/usr/bin/env python3
"""Benchmark of the render/backward hot path (see DESIGN.md "Measurement").

python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one full training iteration on one view per rank: fused forward (projection, binning, blend), L1 + DSSIM
loss, fused backward, gradient all-reduce (N > 1) and Adam.  Workload: BASELINE.json configs[2]: synthetic Lego
800x800, 300 k Gaussians, SH degree 4 (K = 25), 16x16 tiles; all inputs resident in HBM before timing starts.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md); measured copy ceiling 6290
VALU_PEAK_TFLOPS = 157.3


def algorithmic_bytes(N, K, M, P, T):
    """SURVEY.md 8(d) / BASELINE.md 4, per launch."""
    return dict(
        proj_fwd=N * (44 + 12 * K) + N * 64,
        proj_bwd=N * (44 + 12 * K) + N * 44 + N * (40 + 12 * K),
        bin=N * 24 + M * 12 * (2 * -(-(32 + max(1, (T - 1).bit_length())) // 8) + 1) + T * 8,
        blend_fwd=M * 48 + P * 24,
        blend_bwd=M * 48 + P * 44 + M * 88 + N * 44,
        loss=3 * P * 32 + 3 * P * 40,
        adam=N * (11 + 3 * K) * 28,
    )


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c3_300k_800")
    ap.add_argument("--views", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "several ranks on one card together with GSPLAT_BENCH_DEVICE)")
    ap.add_argument("--no-depth-cuts", action="store_true", help="bin every tile list in full (A/B of the depth cuts)")
    ap.add_argument("--no-view-hints", action="store_true",
                    help="do not reuse a view's previous per-block sweep lengths to order the forward's items")
    ap.add_argument("--dp-exchange", default="sh_compressed", choices=["sh_compressed", "allreduce"],
                    help="gradient exchange for --gpus > 1 (trainer.py)")
    ap.add_argument("--ppl", default="", help="fwd,bwd pixels per lane of the op-level kernels (tuning)")
    ap.add_argument("--residency", default="", help="fwd waves/SIMD, bwd waves/CU of the persistent kernels (tuning)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if os.environ.get("GSPLAT_BENCH_DEVICE"):          # rehearsal only: several ranks on one card
        local_rank = int(os.environ["GSPLAT_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, **(dict(device_id=dev) if args.backend == "nccl" else {}))
        pg = dist.group.WORLD

    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel, view_for

    idx, N, W, H, kind = CONFIGS[args.config]
    params, cams, _ = make_config(args.config, n_views=args.views)
    K = 25
    r = GaussianRenderer(4, W, H, (16, 16), False, device=local_rank)
    r.depthCuts = not args.no_depth_cuts
    if args.ppl:
        f, b = (int(x) for x in args.ppl.split(","))
        r.lib.gs_debug_set_ppl(f, b)
    if args.residency:
        f, b = (int(x) for x in args.residency.split(","))
        r.lib.gs_debug_set_residency(f, b)
    # workspace and parameter arenas carry 1.5x headroom so the densify event in the timed region does not reallocate
    r.reserve(int(N * 1.5), int(os.environ.get("GSPLAT_BENCH_PAIR_CAP", 24 * 1024 * 1024 if N <= 400_000 else 96 * 1024 * 1024)))

    # targets: renders of a perturbed copy of the scene (non-trivial gradients), produced before timing
    tgt_params = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
    targets = []
    for cam in cams:
        targets.append(r.renderForward(tgt_params, cam).render.clone())
    del tgt_params
    model = GaussModel(params, dev, capacity=int(N * 1.5))
    trainer = GaussianTrainer(model, r, iterationCount=30000, process_group=pg, dp_exchange=args.dp_exchange)
    # densify / prune runs at the reference cadence (every 100 iterations inside [500, 15000]); the iteration counter
    # starts so that iteration 600 falls in the middle of the timed region
    trainer.iteration = max(600 - args.warmup - args.steps // 2, 0)
    it0 = trainer.iteration
    gcams = [r._camera(c.worldViewTransform, c.projectionMatrix, c.cameraCenter, c.FoVx, c.FoVy, c.focalX, c.focalY)
             for c in cams]
    V = len(cams)

    def step(i):
        v = view_for(i, rank, world, V)
        trainer.trainStep(gcams[v], targets[v], viewKey=None if args.no_view_hints else v,
                          stepCameras=[cams[view_for(i, q, world, V)] for q in range(world)] if world > 1 else None)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # warm-up; its last steps also find the stage with the largest device time
    r.profile(True)
    trainer.prewarmDensify()
    for i in range(args.warmup):
        step(i)
    r.sync()
    wprof = r.profileRead()
    dom = max(wprof, key=lambda k: wprof[k][0] / max(wprof[k][1], 1)) if args.warmup > 0 else "blend_bwd"
    barrier()
    # timed region: exactly K steps; only the dominant stage carries HIP events (each recorded stage costs two
    # event packets on the stream per step)
    r.profile([dom])
    misses0 = trainer.forwardMisses
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    it_lo, it_hi = it0 + args.warmup, it0 + args.warmup + args.steps          # timed iterations [it_lo, it_hi)
    densify_events = [i for i in range(it_lo, it_hi) if i % trainer.split_and_prune_per_iteration == 0
                      and trainer.densifyFromIter <= i <= trainer.densifyUntilIter]
    cut_info = {"enabled": not args.no_view_hints and not args.no_depth_cuts, "forwards_repeated_in_timed_region": trainer.forwardMisses - misses0}
    densify_info = {"events_in_timed_region": len(densify_events), "at_iterations": densify_events,
                    "last_stats": trainer.lastDensifyStats, "N_after": model.N}
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dom_ms_live = r.profileRead()[dom]
    # per-stage breakdown, outside the timed region
    r.profile(True)
    for i in range(min(args.steps, 10)):
        step(args.warmup + args.steps + i)
    prof = r.profileRead()
    r.profile(False)
    r.sync()
    # spread of single steps (rank 0's device time between per-step events), outside the timed region as well
    nq = 30
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(nq + 1)]
    marks[0].record()
    for i in range(nq):
        step(args.warmup + args.steps + 10 + i)
        marks[i + 1].record()
    torch.cuda.synchronize()
    each = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(nq))
    step_spread = {"p10": round(each[nq // 10], 4), "p50": round(each[nq // 2], 4), "p90": round(each[(nq * 9) // 10], 4)}
    loss = [float(x) for x in trainer._loss.cpu()]
    # replicas must hold bit-identical parameters (same summed gradients, same Adam, same densify decisions)
    replicas_identical = None
    if world > 1:
        import torch.distributed as dist
        chk = torch.stack([model.arena.double().sum(), model.arena.double().abs().sum(),
                           torch.tensor(float(model.N), dtype=torch.float64, device=dev)])
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(torch.equal(lo, hi))

    # workload statistics of the last view + forward-only rate (outside the timed region)
    st = r.stats()
    last = r.lastContrib().to(torch.int64)
    P, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    Hp, Wp = -(-H // 16) * 16, -(-W // 16) * 16
    pad = torch.zeros(Hp, Wp, dtype=torch.int64, device=dev)
    pad[:H, :W] = last
    tile_max = pad.view(Hp // 16, 16, Wp // 16, 16).amax(dim=(1, 3))
    M_eff = int(tile_max.sum().item())
    mean_contrib = float(last.double().mean().item())
    nf = 20
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    for i in range(nf):
        r.renderForward(model.getParams(), gcams[i % V])
    torch.cuda.synchronize()
    fwd_ms = (time.perf_counter() - tf0) / nf * 1e3

    if rank != 0:
        return
    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.steps / elapsed
    M = st["M"]
    stage_ms = {k: (v[0] / max(v[1], 1)) for k, v in prof.items()}
    alg = algorithmic_bytes(model.N, K, M, P, T)          # model.N: after the timed region's densify event, if any
    alg_eff = algorithmic_bytes(model.N, K, M_eff, P, T)
    dom_ms = dom_ms_live[0] / max(dom_ms_live[1], 1)     # measured live in the timed region
    # the blend kernels stop at the tile's last contributing splat, so the bytes one launch must move are those of
    # the M_eff pairs actually traversed (sum over tiles of max nContrib), not of all M binned pairs
    dom_bytes = alg_eff[dom] if dom.startswith("blend") else alg[dom]
    achieved = dom_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    flop_per_pair = {"blend_fwd": 24.0, "blend_bwd": 70.0}
    roof = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None, "algorithmic_bytes": int(dom_bytes),
            "avg_launch_ms": round(dom_ms, 4)}
    roof["traffic"] = pmc_traffic_bytes(dom)
    if dom in flop_per_pair:
        tf = flop_per_pair[dom] * 256.0 * M_eff / (dom_ms * 1e-3) / 1e12
        roof["valu_tflops"] = round(tf, 2)
        roof["valu_frac"] = round(tf / VALU_PEAK_TFLOPS, 4)
    if stage_ms.get("adam", 1.0) == 0.0:       # fused: the projection backward also moves the optimizer's bytes
        # ... and neither writes nor re-reads a gradient arena: params + d(packed) in, six accesses (read and write of
        # parameter and both moments) per element
        Nn = model.N
        alg = dict(alg, proj_bwd=Nn * (44 + 12 * K) + Nn * 64 + Nn * (11 + 3 * K) * 24)
    # a stage that did not run on its own (Adam fused into the projection backward) has no rate
    stages = {k: {"ms": round(stage_ms[k], 4),
                  "GBps": round((alg_eff[k] if k.startswith("blend") else alg[k]) / stage_ms[k] / 1e6, 1)
                  if stage_ms[k] > 0 else None}
              for k in stage_ms}

    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(params, cams[0], W, H, targets[0].cpu().numpy())

    out = {
        "metric": "train views/sec (full step: fwd + L1/DSSIM loss + bwd + Adam + densify/prune at the reference "
                  "cadence), Lego 800x800 300k Gaussians",
        "value": round(value, 3), "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.config}: synthetic Lego cameras {W}x{H}, N={N} Gaussians, SH degree 4 (K=25), "
                               f"16x16 tiles, {V} views, 1 view per rank per step, full train step",
                   "parallelism": f"dp{world}", "dp_exchange": args.dp_exchange if world > 1 else None, "N": N, "W": W, "H": H, "tile": 16},
        "fwd_mpix_per_s": round(P / (fwd_ms * 1e-3) / 1e6, 2), "fwd_ms": round(fwd_ms, 4),
        "roofline": roof, "cpu_baseline": cpu, "stages": stages,
        "workload_stats": {"N_visible": st["N_visible"], "M_pairs": M, "M_eff_pairs_traversed": M_eff,
                           "max_tile_list": st["max_tile_list"], "mean_tile_list": round(M / T, 1),
                           "mean_nContrib": round(mean_contrib, 1)},
        "step_ms_spread": step_spread, "densify": densify_info, "depth_cuts": cut_info, "replicas_identical": replicas_identical, "loss": loss,
    }
    print(json.dumps(out))


def pmc_traffic_bytes(stage):
    """HBM-side bytes per launch of the stage's dominant kernel from the committed rocprofv3 PMC passes
    (profiles/*hbm_traffic_pmc.json: FETCH_SIZE and WRITE_SIZE in separate runs of this same bench; FETCH_SIZE is
    doubled as the MI355X guide prescribes for gfx950).  None when no summary covers the kernel."""
    import glob
    key = {"blend_bwd": "blend_bwd_v2_kernel", "blend_fwd": "blend_fwd_v2", "proj_fwd": "proj_fwd_fused_kernel",
           "proj_bwd": "proj_bwd_fused_kernel", "adam": "adam_kernel", "loss": "loss_fused_kernel"}.get(stage)
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*hbm_traffic_pmc.json")))
    if not key or not files:
        return None
    try:
        kernels = json.load(open(files[-1]))["kernels"]
        for name, v in kernels.items():
            if key in name:
                return int((2.0 * v["FETCH_SIZE_KB_per_launch"] + v["WRITE_SIZE_KB_per_launch"]) * 1024)
    except Exception:
        return None
    return None


def cpu_baseline(params, cam, W, H, target):
    """The CPU oracle (a port of the reference arithmetic; the reference itself cannot run off Apple hardware) timed
    on the host cores for ONE view of the same workload: forward + loss + backward (no optimizer)."""
    import numpy as np
    from oracle.oracle import Oracle
    cores = len(os.sched_getaffinity(0))
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    o = Oracle(np.float32)
    c = cam.as_dict()
    t0 = time.perf_counter()
    fw = o.render_forward(params, c, W, H, 16, 16, 4)
    t1 = time.perf_counter()
    loss, cot, _, _, _ = o.loss_forward_backward(fw["color"].reshape(H, W, 3), target, 0.2)
    o.render_backward(params, c, W, H, 16, 16, 4, fw, cot.reshape(-1, 3), np.zeros(W * H, np.float32),
                      np.zeros(W * H, np.float32))
    t2 = time.perf_counter()
    return {"value": round(1.0 / (t2 - t0), 4), "unit": "views/s", "cores": cores, "kind": "port",
            "sample": f"1 view of the same workload (forward {t1 - t0:.2f} s, loss+backward {t2 - t1:.2f} s), "
                      "no optimizer step", "fwd_mpix_per_s": round(W * H / (t1 - t0) / 1e6, 3)}


if __name__ == "__main__":
    main()
