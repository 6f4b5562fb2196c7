"""How much of its tile's list a 16x16 block scans when tiles are larger than a block (blend_*_cull_kernel): positions swept per
block = max nContrib over its pixels (a pixel that never terminates sweeps the whole list).  usage: python tools/tile_scan.py [tile]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 200
params, cams, (W, H) = make_config("c3_300k_800", n_views=1)
r = GaussianRenderer(4, W, H, (tile, tile))
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
res = r.renderForward(tp, cams[0])
nc = r.lastContrib().view(H, W).cpu().numpy().astype(np.int64)
alpha = res.alpha.view(H, W).cpu().numpy()
tot = 0; blocks = 0; full = 0; lens = []
for ty in range(0, H, tile):
    for tx in range(0, W, tile):
        sub = nc[ty:ty + tile, tx:tx + tile]
        L = int(sub.max())           # a never-terminating pixel reports the list length
        for by in range(0, sub.shape[0], 16):
            for bx in range(0, sub.shape[1], 16):
                m = int(sub[by:by + 16, bx:bx + 16].max())
                tot += m; blocks += 1; full += int(m == L); lens.append(m)
print(json.dumps({"tile": tile, "blocks": blocks, "positions_scanned_total": tot, "mean_per_block": tot / blocks,
                  "blocks_scanning_their_whole_list": full, "pixels_terminated_share": float((alpha > 1 - 1e-4).mean()),
                  "M": r.stats()["M"], "p50_p90_max": [int(np.percentile(lens, q)) for q in (50, 90, 100)]}))
