"""Two data-parallel ranks on ONE card (gloo for the collectives, as bench.py --backend gloo does): the pair reserve of both
is too small for the views' pairs, no view is ever "first-visit" checked, and rank 1's views need more pairs than rank 0's.
What must happen (trainer.py, _collectiveOverflowCheck): nobody raises inside a step (a rank that left a step alone would
leave the other in a collective), every step is gated on BOTH ranks until the 16-step check, at which both agree on the
larger need and regrow, training then proceeds, and the replicas end bit-identical.
Second scenario (`one_view`; the round-3 advisor's finding): only ONE of the four views overflows the reserve, and it is
never the last forward of a 16-step window.  Round 3 sized the regrow from the LAST forward's counters: every rank computed
"nothing needed", cleared the ring and lost that view's steps again and again.  Now the need comes from the library's
sticky report (gs_overflow_pending): the check after the first window regrows, and from then on every step moves the
parameters.
usage: python tools/dp_overflow_rehearsal.py [one_view]    (parent: starts the two ranks as fresh child processes)"""
import json, os, socket, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np, torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gaussiansplattingmlx_amd.camera import Camera, look_at_c2w
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    from gaussiansplattingmlx_amd.scenes import make_gaussians, perturb
    from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel, view_for
    W, H, N = 320, 240, 20000
    params = make_gaussians(N, "trained_like", 11)
    params["scales"] += 0.8
    cams = [Camera(W, H, f, f, look_at_c2w(e)) for f, e in ((260.0, [3.0, -2.5, 2.0]), (200.0, [-2.0, 2.6, 1.6]),
                                                           (240.0, [0.5, 3.0, 1.5]), (180.0, [2.8, 2.2, -1.0]))]
    r0 = GaussianRenderer(4, W, H)
    tp = {k: torch.as_tensor(v, device=r0.device) for k, v in params.items()}
    needs, targets = [], []
    tgt = {k: torch.as_tensor(v, device=r0.device) for k, v in perturb(params, 3).items()}
    for c in cams:
        r0.renderForward(tp, c); needs.append(r0.stats()["M"])
        targets.append(r0.renderForward(tgt, c).render.clone())
    r0.close()
    one_view = os.environ.get("REHEARSAL_MODE") == "one_view"
    if one_view:
        # the hungriest view first: rank 0 renders it at the EVEN steps, so the last step of every 16-step window (i = 15, 31)
        # is another view, whose pairs fit
        order = sorted(range(len(cams)), key=lambda v: -needs[v])
        cams, targets, needs = [cams[v] for v in order], [targets[v] for v in order], [needs[v] for v in order]
    reserve = (needs[0] + needs[1]) // 2 if one_view else min(needs) // 2
    r = GaussianRenderer(4, W, H)
    r.reserve(N, reserve)
    model = GaussModel(params, r.device)
    tr = GaussianTrainer(model, r, iterationCount=1000, process_group=dist.group.WORLD, dp_exchange="sh_compressed", densify=False)
    start = model.arena.clone()
    moved_at = None
    still = []                      # steps that left the parameters alone (gated)
    for i in range(40):
        v = view_for(i, rank, world, len(cams))
        before = model.arena.clone() if one_view else None
        tr.trainStep(cams[v], targets[v], stepCameras=[cams[view_for(i, q, world, len(cams))] for q in range(world)])   # no viewKey: no first-visit check
        if moved_at is None and not torch.equal(model.arena, start):
            moved_at = i
        if one_view and torch.equal(model.arena, before):
            still.append(i)
        if os.environ.get("REHEARSAL_DEBUG") and i < 6:
            print(f"DEBUG rank {rank} step {i} view {v} local word {float(tr._cc_local[3 * N])} gathered words {tr._cc_all[:, 3 * N].tolist()} "
                  f"gate {int(tr._gate)} seen {int(tr._seen)} stats {r.stats()}", flush=True)
    torch.cuda.synchronize()
    chk = torch.stack([model.arena.double().sum().cpu(), model.arena.double().abs().sum().cpu()])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    out = dict(rank=rank, mode="one_view" if one_view else "all_views", needs=needs, reserve_start=reserve, capM_end=r.stats()["capM"],
               recoveries=tr.overflowRecoveries, gated_steps=still,
               first_step_that_moved_parameters=moved_at, replicas_identical=bool(torch.equal(lo, hi)),
               finite=bool(torch.isfinite(model.arena).all()))
    print("REHEARSAL " + json.dumps(out), flush=True)
    dist.destroy_process_group()
    if one_view:
        # the even steps of the first window are gated on BOTH ranks (rank 0's view 0 overflowed), none after the check at 16
        ok = out["replicas_identical"] and out["finite"] and out["recoveries"] == 1 and out["capM_end"] >= max(needs) \
            and still == list(range(0, 16, 2))
    else:
        ok = out["replicas_identical"] and out["finite"] and out["recoveries"] >= 1 and moved_at is not None and moved_at >= 16 \
            and out["capM_end"] >= max(needs)
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    if "RANK" in os.environ:
        child()
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for rk in range(2):
            env = dict(os.environ, RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       REHEARSAL_MODE=sys.argv[1] if len(sys.argv) > 1 else "all_views")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
        rcs = [p.wait(timeout=600) for p in procs]
        print("exit codes", rcs)
        sys.exit(max(rcs))
