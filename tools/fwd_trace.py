"""Diagnostic: per-item timeline of the persistent blend-forward kernel (start/end shader clocks, iterations)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
from gaussiansplattingmlx_amd.scenes import make_config

params, cams, (W, H) = make_config(os.environ.get("FWD_TRACE_CONFIG", "c3_300k_800"), n_views=2)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(params["xyz"].shape[0], 16 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
for _ in range(3):
    r.renderForward(tp, cams[0], viewKey=0)
nItems = ((W + 15) // 16) * ((H + 15) // 16) * 4     # quadrant items (2 per block with the packed kernel)
if len(sys.argv) > 1:
    r.setTuning(fwd_waves_per_simd=int(sys.argv[1]) % 100, fwd_quadrants=int(int(sys.argv[1]) >= 100))
buf = torch.zeros(nItems * 4, dtype=torch.int64, device=r.device)
r.setTuning(fwd_trace_buffer=buf.data_ptr())
r.renderForward(tp, cams[0], viewKey=0)
torch.cuda.synchronize()
r.setTuning(fwd_trace_buffer=0)
t = buf.cpu().numpy().reshape(-1, 4)
t0, t1, it, chunks = t[:, 0], t[:, 1], t[:, 2] & 0xFFFFFFFF, t[:, 2] >> 32
ok = (t1 > 0) & (t0 > 0)
# every XCD has its own clock: rebase each XCD's items on that XCD's first start
xcc = (t[:, 3] >> 32) & 0xF
t0 = t0.copy(); t1 = t1.copy()
for x in np.unique(xcc[ok]):
    sel = ok & (xcc == x)
    b0 = t0[sel].min()
    t0[sel] -= b0; t1[sel] -= b0
    print("xcd", int(x), "items", int(sel.sum()), "span", int(t1[sel].max()), "busy wave-cycles", int((t1 - t0)[sel].sum()))
t0[ok] += 1; t1[ok] += 1
base = 1
dur = (t1 - t0)[ok]
print("items", ok.sum(), "kernel span cycles", t1[ok].max() - base)
bx = t[:, 3] & 0xFFFFFFFF
print("share of items whose XCC_ID == blockIdx.x % 8:", float((xcc[ok] == (bx[ok] % 8)).mean()))
hwid = (t[:, 3] >> 36) & 0xFFFF
cu = (hwid >> 8) & 0xF; se = (hwid >> 13) & 0x7; sh = (hwid >> 12) & 1; simd = (hwid >> 4) & 3
# where do the four quadrant items of a block run?
it4 = np.arange(len(t)) // 4
same_xcd = same_cu = same_simd = n4 = 0
for b in range(0, len(t) // 4):
    sl = slice(4 * b, 4 * b + 4)
    if not ok[sl].all(): continue
    n4 += 1
    same_xcd += len(set(xcc[sl])) == 1
    same_cu += len(set(zip(xcc[sl], se[sl], sh[sl], cu[sl]))) == 1
    same_simd += len(set(zip(xcc[sl], se[sl], sh[sl], cu[sl], simd[sl]))) == 1
print("blocks whose four quadrant items ran on one XCD %.3f, one CU %.3f, one SIMD %.3f" % (same_xcd / n4, same_cu / n4, same_simd / n4))
hw = t[:, 3][ok]
print("distinct hw ids", len(np.unique(hw)))
# items per SIMD, and how long each SIMD's items kept it (under-subscribed launches: 10 k scene, 2500 items on 1024 SIMDs)
from collections import Counter, defaultdict
per = Counter(); last = defaultdict(int)
for i in np.nonzero(ok)[0]:
    k = (int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i]), int(simd[i]))
    per[k] += 1; last[k] = max(last[k], int(t1[i]))
print("SIMDs with items", len(per), "items per SIMD -> SIMDs", sorted(Counter(per.values()).items()))
byc = defaultdict(list)
for k, v in per.items(): byc[v].append(last[k])
print("last end (cycles since the XCD's first start) by items per SIMD:", {v: (int(np.mean(e)), int(np.max(e))) for v, e in sorted(byc.items())})
ev = torch.cuda.Event(enable_timing=True); ev2 = torch.cuda.Event(enable_timing=True)
ev.record(); r.renderForward(tp, cams[0]); ev2.record(); torch.cuda.synchronize(); print("whole forward ms", ev.elapsed_time(ev2))
print("sum iterations", it[ok].sum(), "max", it[ok].max())
cyc_per_it = dur / np.maximum(it[ok], 1)
big = it[ok] > 200
print("cycles/iteration (items > 200 its): mean %.1f  p10 %.1f  p50 %.1f  p90 %.1f" % (
    cyc_per_it[big].mean(), *np.percentile(cyc_per_it[big], [10, 50, 90])))
order = np.argsort(-(t1[ok] - base))[:10]
print("last finishing items: (end, start, iters, chunks, cyc/it, cycles/chunk)")
for o in order:
    print(int(t1[ok][o] - base), int(t0[ok][o] - base), int(it[ok][o]), int(chunks[ok][o]), round(float(cyc_per_it[o]), 1), round(float(dur[o] / max(chunks[ok][o], 1)), 1))
# round 6: cycles of an item = a x blended entries + b x chunks (least squares), over all items and over the items that start in the
# first tenth of the span (four waves on every SIMD) / end in the last fifth (the tail, SIMDs emptying)
A = np.stack([it[ok].astype(np.float64), chunks[ok].astype(np.float64)], 1)
span = float(t1[ok].max() - base)
for name, sel in (("all items", np.ones(len(dur), bool)), ("started in the first 10 %", (t0[ok] - base) < 0.1 * span), ("ended in the last 20 %", (t1[ok] - base) > 0.8 * span)):
    if sel.sum() > 10:
        coef, *_ = np.linalg.lstsq(A[sel], dur[sel].astype(np.float64), rcond=None)
        print(f"{name}: {int(sel.sum())} items, cycles ~ {coef[0]:.1f} x entries + {coef[1]:.1f} x chunks; entries per chunk mean {A[sel, 0].sum() / max(A[sel, 1].sum(), 1):.1f}")
# round 6: is an item's speed a property of the WAVE SLOT it runs in?  (HW_ID[3:0] = wave id within the SIMD)
wave_id = hwid & 0xF
wv = wave_id[ok]
print("cycles per entry by hardware wave slot (items with > 100 entries): slot -> (items, mean, p50, p90)")
for w_ in sorted(set(int(x) for x in wv)):
    sel = (wv == w_) & (it[ok] > 100)
    if sel.sum():
        cpe = dur[sel] / it[ok][sel]
        print(f"  slot {w_}: {int(sel.sum())} items, mean {cpe.mean():.0f}, p50 {np.percentile(cpe, 50):.0f}, p90 {np.percentile(cpe, 90):.0f}; entries of these items: mean {it[ok][sel].mean():.0f}")
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(out_dir):
    np.savez_compressed(os.path.join(out_dir, "fwd_trace_raw.npz"), t=t)
print("the 10 longest items: (cycles, entries, chunks, start)")
for o in np.argsort(-dur)[:10]:
    print(int(dur[o]), int(it[ok][o]), int(chunks[ok][o]), int(t0[ok][o] - base))
# concurrency over time
ends = np.sort(t1[ok] - base)
span = ends[-1]
for frac in (0.05, 0.25, 0.5, 0.6, 0.7, 0.8, 0.9, 0.95):
    tt = span * frac
    running = ((t0[ok] - base) <= tt) & ((t1[ok] - base) > tt)
    print("t=%.0f%% running waves %d" % (frac * 100, running.sum()))
# per persistent wave (blockIdx.x): its own clock is consistent
wid = (t[:, 3] & 0xFFFFFFFF)[ok]
T0, T1, IT = t[:, 0][ok], t[:, 1][ok], it[ok]
spans, busys, nit, sumit = [], [], [], []
for w in np.unique(wid):
    s_ = wid == w
    spans.append(T1[s_].max() - T0[s_].min()); busys.append((T1[s_] - T0[s_]).sum()); nit.append(s_.sum()); sumit.append(IT[s_].sum())
spans, busys, nit, sumit = map(np.array, (spans, busys, nit, sumit))
print("waves", len(spans), "items/wave mean %.2f" % nit.mean())
print("per-wave span cycles: mean %.0f p50 %.0f p90 %.0f max %.0f" % (spans.mean(), *np.percentile(spans, [50, 90]), spans.max()))
print("per-wave busy/span: mean %.3f ; sum iterations per wave mean %.0f max %.0f" % ((busys / spans).mean(), sumit.mean(), sumit.max()))
o_ = np.argsort(-spans)[:8]
print("slowest waves: span, busy, items, iterations:", [(int(spans[i]), int(busys[i]), int(nit[i]), int(sumit[i])) for i in o_])
first = np.array([IT[wid == w][np.argmin(T0[wid == w])] for w in np.unique(wid)])
print("iterations of each wave's FIRST item: mean %.0f max %.0f ; rate of first items cyc/it mean %.0f" % (first.mean(), first.max(), np.mean([(T1[wid == w] - T0[wid == w])[np.argmin(T0[wid == w])] / max(IT[wid == w][np.argmin(T0[wid == w])], 1) for w in np.unique(wid)])))
print("balanced span if iterations were spread evenly at the observed rate: %.0f" % (sumit.mean() * (busys.sum() / sumit.sum())))
h, e = np.histogram(it[ok], bins=[0, 1, 64, 128, 256, 512, 1024, 4096])
print("iterations histogram", list(zip(e[:-1].tolist(), h.tolist())))
print("busy wave-cycles / (span * resident waves): %.3f" % (dur.sum() / (span * max(len(np.unique(hw)), 1))))
print("fixed overhead per item (items with 0 iterations): mean cycles", dur[it[ok] == 0].mean() if (it[ok] == 0).any() else None)

# What a better ORDER could buy (round 5): longest-processing-time-first over the measured item durations on as many workers as
# waves ran, against the span the waves actually took.  (First-order only: an item's duration depends on what shared its SIMD.)
import heapq
durs = sorted((int(d) for d in dur), reverse=True)
nw = len(np.unique(hw))
heap = [0] * nw
for d_ in durs:
    heapq.heappush(heap, heapq.heappop(heap) + d_)
print(f"LPT over the measured item durations on {nw} workers: makespan {max(heap)} cycles, mean load {sum(durs) // nw}, longest item {durs[0]}; "
      f"per-wave span as it ran: see 'per-wave span cycles' above")
dd = np.array(durs)
print("item durations (cycles): " + "  ".join(f"p{q}={int(np.percentile(dd, q))}" for q in (50, 90, 99, 99.9)) + f"  max={dd.max()}  items above 0.8 max: {int((dd > 0.8 * dd.max()).sum())}, above 0.7 max: {int((dd > 0.7 * dd.max()).sum())}")
first = np.array(sorted(((int(t0[ok][i]), int(dur[i])) for i in range(len(dur)))))
