"""Histogram of the live-pixel count of the blend backward's (block, 64-entry segment) work items, weighted by item
length (list entries swept), from per-pixel nContrib.  A pixel is live in segment s of its block while nContrib > 64 s.
Source of nContrib: the library's forward (GPU) or, with --oracle, the CPU oracle (same numbers up to the nContrib bar).
usage: python tools/lane_hist.py [config] [--oracle] [--views V]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gaussiansplattingmlx_amd.scenes import make_config
args = [a for a in sys.argv[1:] if not a.startswith("--")]
cfg = args[0] if args else "c3_300k_800"
use_oracle = "--oracle" in sys.argv
V = int(sys.argv[sys.argv.index("--views") + 1]) if "--views" in sys.argv else 2
params, cams, (W, H) = make_config(cfg, n_views=V)
SEG = 64
if use_oracle:
    from oracle.oracle import Oracle
    o = Oracle(np.float32)
    contribs = [o.render_forward(params, c.as_dict(), W, H, 16, 16, 4)["last"].reshape(H, W).astype(np.int64) for c in cams]
else:
    import torch
    from gaussiansplattingmlx_amd.renderer import GaussianRenderer
    r = GaussianRenderer(4, W, H)
    tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
    contribs = []
    for c in cams:
        r.renderForward(tp, c)
        contribs.append(r.lastContrib().cpu().numpy().astype(np.int64))
edges = [0, 16, 32, 64, 96, 128, 192, 256]
rows = []
for nc in contribs:
    Hp, Wp = -(-H // 16) * 16, -(-W // 16) * 16
    pad = np.zeros((Hp, Wp), np.int64); pad[:H, :W] = nc
    blk = pad.reshape(Hp // 16, 16, Wp // 16, 16).transpose(0, 2, 1, 3).reshape(-1, 256)      # [blocks, 256]
    w = blk.max(1)
    nseg = (w + SEG - 1) // SEG
    tot_len = 0
    hist = np.zeros(len(edges) - 1, np.float64)       # item length (entries) by live-pixel bucket
    hist_half = np.zeros(len(edges) - 1, np.float64)
    live_ps = exec_ps = 0
    for s in range(int(nseg.max())):
        sel = nseg > s
        b = blk[sel]
        ln = np.minimum(w[sel] - s * SEG, SEG)                      # entries of this item
        live = (b > s * SEG).sum(1)                                 # pixels live at the item's first entry
        tot_len += ln.sum()
        idx = np.clip(np.searchsorted(edges, live, side="left") - 1, 0, len(edges) - 2)
        np.add.at(hist, idx, ln)
        # pixel-splats: live ones inside the item, and what a 256-wide sweep executes
        live_ps += np.clip(b - s * SEG, 0, SEG).sum()
        exec_ps += (256 * ln).sum()
    rows.append(dict(items=int(nseg.sum()), block_splats=int(tot_len), live_share=float(live_ps / exec_ps),
                     share_of_block_splats_by_live_pixels={f"{edges[i] + 1}-{edges[i + 1]}": round(float(hist[i] / tot_len), 4)
                                                           for i in range(len(edges) - 1)}))
print(json.dumps({"config": cfg, "source": "oracle" if use_oracle else "gpu", "views": rows}, indent=1))
