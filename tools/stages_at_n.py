"""Stage times of the train step once the soak scene has grown (densify cadence from iteration 450): the bench line is quoted at
N = 300 k, the trainer's scenes grow to ~1 M.  Prints the library's stage times every 300 iterations.
usage: python tools/stages_at_n.py [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
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
tr = GaussianTrainer(model, r, iterationCount=30000)
tr.iteration = 450
i = 0
while i < steps:
    for _ in range(260):
        tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8); i += 1
    torch.cuda.synchronize()
    r.profile(True)
    t0 = time.perf_counter()
    for _ in range(40):
        tr.trainStep(cams[i % 8], targets[i % 8], viewKey=i % 8); i += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 40 * 1e3
    pr = r.profileRead(); r.profile(False)
    print(json.dumps({"it": tr.iteration, "N": model.N, "M": r.stats()["M"], "ms_per_step": round(dt, 3),
                      "stages": {k: round(v[0] / 40, 4) for k, v in pr.items()}}), flush=True)
