"""Block lists (tile 200) on the c3 scene: what do depth cuts do to a view's pair count and its step time?
usage: python tools/block_lists_cuts.py [tile]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config, perturb

ts = int(sys.argv[1]) if len(sys.argv) > 1 else 200
idx, N, W, H, kind = CONFIGS["c3_300k_800"]
params, cams, _ = make_config("c3_300k_800", n_views=4)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (ts, ts), False)
r.reserve(int(N * 1.5), 24 << 20)
tp = {k: torch.as_tensor(v, device=dev) for k, v in params.items()}
tq = {k: torch.as_tensor(v, device=dev) for k, v in perturb(params, 12345).items()}
targets = [r.renderForward(tq, c).render.clone() for c in cams]
for cuts in (False, True):
    r.cutMinDropped = 0
    r.depthCuts = cuts
    r.dropDepthCuts()
    out = []
    for rep in range(12):
        if rep == 4:
            torch.cuda.synchronize(); t0 = time.perf_counter()
        for v in range(4):
            res = r.renderChecked(tp, cams[v], viewKey=v, wantDepth=False)
            lo, gc, _ = r.lossForwardBackward(res.render, targets[v], 0.2)
            r.renderBackward(gc)
        if rep in (0, 11):
            out.append(r.stats()["M"])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / (8 * 4) * 1e3
    pol = r._cut_policy.get(0)
    print(f"tile {ts} cuts={cuts}: M first visit {out[0]}, last visit {out[1]}, {ms:.4f} ms per forward+loss+backward, "
          f"last_dropped {pol.last_dropped if pol else None}", flush=True)
