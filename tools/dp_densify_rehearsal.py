"""Two data-parallel ranks on ONE card (gloo for the collectives) through DENSIFY EVENTS that outgrow the model's capacity: the
planned event of the data-parallel step (packed gather with device-computed tensor starts, the ranks' plans compared in a
fixed-size collective, the arena checksum queued and judged later) with models that have no capacity headroom, so that several
events regrow.  What must happen: both ranks commit the same N at every event, the replicas stay bit-identical (checked at every
event by the trainer and once more at the end), everything stays finite.
usage: python tools/dp_densify_rehearsal.py [steps]    (parent: starts the two ranks as fresh child processes)"""
import json, os, socket, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np, torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    steps = int(os.environ.get("REHEARSAL_STEPS", "380"))
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
    r = GaussianRenderer(4, W, H)
    tgt = {k: torch.as_tensor(v, device=r.device) for k, v in perturb(params, 3).items()}
    targets = [r.renderForward(tgt, c).render.clone() for c in cams]
    model = GaussModel(params, r.device)                       # capacity = N: the first event that grows it regrows
    tr = GaussianTrainer(model, r, iterationCount=30000, process_group=dist.group.WORLD, dp_exchange="sh_compressed")
    tr.iteration = 470
    tr.gradientThreshold = 5e-6                                # (a small scene: make the events split and clone)
    assert tr._plans_events()
    sizes, caps = [N], [model.capacity]
    for i in range(steps):
        v = view_for(i, rank, world, len(cams))
        tr.trainStep(cams[v], targets[v], viewKey=v, stepCameras=[cams[view_for(i, q, world, len(cams))] for q in range(world)])
        if model.N != sizes[-1]:
            sizes.append(model.N); caps.append(model.capacity)
    tr.checkReplicas()
    torch.cuda.synchronize()
    chk = torch.stack([model.arena.double().sum().cpu(), model.arena.double().abs().sum().cpu(), torch.tensor(float(model.N), dtype=torch.float64)])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    out = dict(rank=rank, steps=steps, sizes=sizes, capacities=caps, regrows=sum(1 for a, b in zip(caps, caps[1:]) if b > a),
               last_event=tr.lastDensifyStats, replicas_identical=bool(torch.equal(lo, hi)), finite=bool(torch.isfinite(model.arena).all()),
               overflow_recoveries=tr.overflowRecoveries)
    print("REHEARSAL " + json.dumps(out), flush=True)
    dist.destroy_process_group()
    ok = out["replicas_identical"] and out["finite"] and len(sizes) >= 3 and out["regrows"] >= 2
    sys.exit(0 if ok else 3)


if __name__ == "__main__":
    if "RANK" in os.environ:
        child()
    else:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        procs = []
        for rk in range(2):
            env = dict(os.environ, RANK=str(rk), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       REHEARSAL_STEPS=sys.argv[1] if len(sys.argv) > 1 else "380")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)], env=env))
        rcs = [p.wait(timeout=900) for p in procs]
        print("exit codes", rcs)
        sys.exit(max(rcs))
