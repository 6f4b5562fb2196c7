"""Diagnostic: how many (block, segment) items of the backward have an entirely dead half / quadrant?"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config
params, cams, (W, H) = make_config("c3_300k_800", n_views=2)
r = GaussianRenderer(4, W, H)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
r.renderForward(tp, cams[0])
last = r.lastContrib().cpu().numpy().reshape(H, W).astype(np.int64)
B = last.reshape(H // 16, 16, W // 16, 16).transpose(0, 2, 1, 3)          # [by, bx, y, x]
work = B.max(axis=(2, 3))
segs = (work + 127) // 128
items = int(segs.sum())
half = B.reshape(*B.shape[:2], 2, 8, 16).max(axis=(3, 4))                   # [by,bx,2]
quad = B.reshape(*B.shape[:2], 2, 8, 2, 8).max(axis=(3, 5))                 # [by,bx,2,2]
live_half = live_quad = 0
px_live = 0
for s in range(int(segs.max())):
    sel = segs > s
    live_half += int((half[sel] > 128 * s).sum())
    live_quad += int((quad[sel] > 128 * s).sum())
    px_live += int((B[sel] > 128 * s).sum())
print("items", items, "live halves %.3f" % (live_half / (2 * items)), "live quadrants %.3f" % (live_quad / (4 * items)),
      "live pixels %.3f" % (px_live / (256 * items)))
