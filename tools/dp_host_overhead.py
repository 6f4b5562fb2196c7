"""Host time to enqueue a step vs. the time the step takes, single-device against the data-parallel step on a 1-rank group
(native and torch exchange): is the 0.1 ms the DP step costs on one rank the host's, or gaps on the device?
usage: python tools/dp_host_overhead.py"""
import ctypes as C, os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd import _lib
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel

name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
gc = [r._camera(c.worldViewTransform, c.projectionMatrix, c.cameraCenter, c.FoVx, c.FoVy, c.focalX, c.focalY) for c in cams]
modes = sys.argv[1:] or ["single", "native_sh", "native_allreduce", "torch_sh"]
for mode in modes:
    model = GaussModel(params, dev, capacity=int(N * 1.5))
    kw = dict(iterationCount=30000, densify=False)
    if mode.startswith("native"):
        uid = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
        assert r.lib.gs_dp_unique_id(uid) == 0
        kw.update(exchange_impl="native", dp_bootstrap=(uid.raw, 0, 1), exchange_when_single=True,
                  dp_exchange="sh_compressed" if mode.endswith("sh") else "allreduce")
    elif mode.startswith("torch"):
        import torch.distributed as dist
        if not dist.is_initialized():
            s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        kw.update(process_group=dist.group.WORLD, exchange_when_single=True, dp_exchange="sh_compressed")
    tr = GaussianTrainer(model, r, **kw)
    sc = [[cams[i % 8]] for i in range(8)]
    for i in range(30):
        tr.trainStep(gc[i % 8], targets[i % 8], viewKey=i % 8, stepCameras=sc[i % 8])
    torch.cuda.synchronize()
    n = 200
    t0 = time.perf_counter()
    for i in range(n):
        tr.trainStep(gc[i % 8], targets[i % 8], viewKey=i % 8, stepCameras=sc[i % 8])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    r.profile(True)
    for i in range(20):
        tr.trainStep(gc[i % 8], targets[i % 8], viewKey=i % 8, stepCameras=sc[i % 8])
    pr = r.profileRead(); r.profile(False)
    print(f"{mode:18s} host enqueue {(t1 - t0) / n * 1e3:.3f} ms/step   total {(t2 - t0) / n * 1e3:.3f} ms/step   stage sum {sum(v[0] for v in pr.values()) / 20:.3f}   regrows {tr.overflowRecoveries}", flush=True)
    tr.closeExchange()
