"""Per-step device time of the data-parallel step (torch exchange, 1-rank nccl group): events between steps."""
import os, sys, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb
from gaussiansplattingmlx_amd.trainer import GaussianTrainer, GaussModel
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=8)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 << 20)
tp = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tp, c).render.clone() for c in cams]
gc = [r._camera(c.worldViewTransform, c.projectionMatrix, c.cameraCenter, c.FoVx, c.FoVy, c.focalX, c.focalY) for c in cams]
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
model = GaussModel(params, dev, capacity=int(N * 1.5))
tr = GaussianTrainer(model, r, iterationCount=30000, densify=False, process_group=dist.group.WORLD, exchange_when_single=True)
sc = [[cams[i % 8]] for i in range(8)]
for i in range(40):
    tr.trainStep(gc[i % 8], targets[i % 8], viewKey=i % 8, stepCameras=sc[i % 8])
torch.cuda.synchronize()
n = 64
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    tr.trainStep(gc[i % 8], targets[i % 8], viewKey=i % 8, stepCameras=sc[i % 8])
    ev[i + 1].record()
torch.cuda.synchronize()
t = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
print("iteration of first timed step", tr.iteration - n)
print("per-step ms:", " ".join(f"{x:.2f}" for x in t))
print("mean", np.mean(t), "median", np.median(t))
dist.destroy_process_group()
