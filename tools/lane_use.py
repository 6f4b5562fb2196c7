"""Useful-lane share of the blend kernels, computed from the forward's own per-pixel nContrib (no counter sees it: a
finished pixel keeps its EXEC bit and blends on with alpha = 0).  Per 16x16 block b with sweep length w_b = max nContrib:
  block-splats traversed      w_b            (what M_eff sums)
  pixel-splats executed       256 w_b        backward (4 px/lane, whole block per wave)
                              64 sum over the block's four 8x8 quadrants of their own sweep length   forward
  pixel-splats that are live  sum over pixels of nContrib
usage: python tools/lane_use.py [config]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3_300k_800"
params, cams, (W, H) = make_config(cfg, n_views=4)
r = GaussianRenderer(4, W, H)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
rows = []
for cam in cams:
    r.renderForward(tp, cam)
    nc = r.lastContrib().to(torch.int64)
    Hp, Wp = -(-H // 16) * 16, -(-W // 16) * 16
    pad = torch.zeros(Hp, Wp, dtype=torch.int64, device=nc.device); pad[:H, :W] = nc
    blk = pad.view(Hp // 16, 16, Wp // 16, 16)
    w_blk = blk.amax(dim=(1, 3))
    half = pad.view(Hp // 16, 2, 8, Wp // 16, 16).amax(dim=(2, 4))                  # 16x8 halves
    quad = pad.view(Hp // 16, 2, 8, Wp // 16, 2, 8).amax(dim=(2, 5))                 # 8x8 quadrants
    seg = 64
    live = int(pad.sum())
    # backward: items are (block, 64-entry segment); inside a segment a 16x8 half past its own sweep is skipped
    bwd_exec_block = int((256 * w_blk).sum())
    bwd_exec_half = int((128 * half).sum())
    # ... at quadrant granularity (8x8 items for the backward)
    bwd_exec_quad = int((64 * quad).sum())
    rows.append(dict(live_pixel_splats=live, block_splats=int(w_blk.sum()),
                     useful_share_block_items=live / bwd_exec_block, useful_share_half_skip=live / bwd_exec_half,
                     useful_share_quadrant_items=live / bwd_exec_quad))
print(json.dumps({"config": cfg, "views": rows,
                  "mean": {k: float(np.mean([x[k] for x in rows])) for k in rows[0]}}, indent=1))
