"""Soak of the DATA-PARALLEL step on a 1-rank RCCL group (the library's own issuer): tools/soak.py's run through gs_dp_step.
usage: python tools/soak_dp.py [steps]"""
import ctypes, sys, time
import torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd import _lib as gslib
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
model = GaussModel(params, dev)
uid = ctypes.create_string_buffer(gslib.GS_DP_UNIQUE_ID_BYTES)
assert r.lib.gs_dp_unique_id(uid) == 0
import os
FORM = os.environ.get("SOAK_FORM", "native")          # native: a 1-rank RCCL group inside the library; torch: a 1-rank nccl process group; local2: two views per step, no group
if FORM == "native":
    tr = GaussianTrainer(model, r, iterationCount=30000, process_group=None, dp_exchange="sh_compressed", exchange_impl="native",
                         exchange_when_single=True, dp_bootstrap=(uid.raw, 0, 1))
elif FORM == "torch":
    import socket
    import torch.distributed as dist
    s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); port = s_.getsockname()[1]; s_.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    tr = GaussianTrainer(model, r, iterationCount=30000, process_group=dist.group.WORLD, dp_exchange="sh_compressed",
                         exchange_impl="torch", exchange_when_single=True)
else:
    VIEWS = int(os.environ.get("SOAK_VIEWS", "2"))
    tr = GaussianTrainer(model, r, iterationCount=30000, views_per_rank=VIEWS)
tr.iteration = 450
t0 = time.perf_counter()
for i in range(steps):
    v = i % 8
    if FORM in ("native", "torch"):
        loss = tr.trainStep(cams[v], targets[v], viewKey=v, stepCameras=[cams[v]])
    else:
        vs = [(v + j) % 8 for j in range(VIEWS)]
        loss = tr.trainStep([cams[j] for j in vs], [targets[j] for j in vs], viewKey=vs, stepCameras=[cams[j] for j in vs])
    if (i + 1) % 100 == 0:
        l = [float(x) for x in loss.cpu()]
        t1 = time.perf_counter()
        st = r.stats()
        print(f"it {tr.iteration} N {model.N} loss {l[0]:.4f} l1 {l[1]:.4f} ssim {l[2]:.4f} views/s {100 / (t1 - t0):.0f} "
              f"M {st['M']} capN {st['capN']} capM {st['capM']} finite {bool(all(bool(torch.isfinite(v).all()) for v in model.getParams().values()))} densify {tr.lastDensifyStats}", flush=True)
        t0 = time.perf_counter()
