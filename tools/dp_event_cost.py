"""What a densify event costs in each form of the step (diagnostic, round 6): device time of [4 plain steps] against [event step + 4],
and the host time of the event's step call.  usage: python tools/dp_event_cost.py [single|dp1_native|dp1_torch|local8] [planned 0|1]"""
import ctypes as C, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd import _lib
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

form = sys.argv[1] if len(sys.argv) > 1 else "single"
planned = (sys.argv[2] if len(sys.argv) > 2 else "1") != "0"
idx, N, W, H, kind = CONFIGS["c3_300k_800"]
params, cams, _ = make_config("c3_300k_800", n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.6), 24 << 20)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev, capacity=int(N * 1.6))
kw, V = {}, 1
if form == "dp1_native":
    uid = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
    assert r.lib.gs_dp_unique_id(uid) == 0
    kw = dict(exchange_impl="native", dp_bootstrap=(uid.raw, 0, 1), exchange_when_single=True)
elif form == "dp1_torch":
    import torch.distributed as dist
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    kw = dict(process_group=dist.group.WORLD, exchange_when_single=True)
elif form == "local8":
    V = 8
    kw = dict(views_per_rank=8)
tr = GaussianTrainer(model, r, iterationCount=30000, **kw)
tr.plannedDensify = planned
tr.prewarmDensify()

def step(i):
    if V > 1:
        tr.trainStep(cams, targets, viewKey=list(range(8)), stepCameras=cams)
    else:
        tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8, stepCameras=[cams[i % 8]])

for rep in range(3):
    tr.iteration = 557 + 100 * rep      # (39 + 4 steps later the counter stands at 600: the event's step)
    for i in range(39):
        step(i)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for i in range(4):
        step(i)
    e1.record()
    t0 = time.perf_counter()
    step(4)                                            # iteration 600 + 100 rep: the event
    t1 = time.perf_counter()
    for i in range(5, 9):
        step(i)
    e2.record()
    torch.cuda.synchronize()
    print(f"{form} planned={planned} rep {rep} N={model.N}: 4 plain steps {e0.elapsed_time(e1):.3f} ms, event step + 4 {e1.elapsed_time(e2):.3f} ms "
          f"-> event {e1.elapsed_time(e2) - e0.elapsed_time(e1) * 5 / 4:.3f} ms over plain steps (incl. the larger scene's 4 steps); "
          f"host: the event's step call {1e3 * (t1 - t0):.3f} ms", flush=True)
tr.closeExchange()
