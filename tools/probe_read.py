"""Phase stamps of binning kernels from a probe build (python -m gaussiansplattingmlx_amd.build --variant probe -DGS_PROBE;
GSPLAT_LIB=gaussiansplattingmlx_amd/libgsplat_hip_probe.so python tools/probe_read.py [config]): thread 0 of a workgroup
stamps the 100-MHz clock at the kernel's phases (binning.hip, GS_PROBE_MARK)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gaussiansplattingmlx_amd.scenes import make_config
from gaussiansplattingmlx_amd.renderer import GaussianRenderer
cfg = sys.argv[1] if len(sys.argv) > 1 else "c1_10k_400"
params, cams, (W, H) = make_config(cfg, n_views=4)
r = GaussianRenderer(4, W, H)
r.depthCuts = False
r.reserve(params["xyz"].shape[0], 24 << 20)
t = {k: torch.as_tensor(v, device=r.device) for k, v in params.items()}
for i in range(12):
    r.renderForward(t, cams[i % 4], viewKey=i % 4, wantDepth=True)
torch.cuda.synchronize()
f = r.lib.gs_debug_probe_read; f.restype = C.c_int; f.argtypes = [C.c_void_p, C.c_int]
buf = np.zeros(1 << 16, np.uint64)
assert f(buf.ctypes.data, buf.size) == 0
p = buf.reshape(-1, 16).astype(np.int64)
M = r.stats()["M"]; print("M", M, "sort tiles", (M + 4095) // 4096)
tiny = p[4095]
if tiny[0]:
    k = [int(x) for x in tiny if x]; print("tiny sort, us since start:", [round((x - k[0]) / 100, 2) for x in k])
rows = [q for i, q in enumerate(p[:4000]) if q[0] and q[7] and i >= 8]
if rows:
    a = np.array(rows); t0 = a[:, 0].min()
    names = ["start", "tile-total scan", "pass 1 (low 8 bits)", "LDS exchange", "pass 2 (high 4 bits)", "LDS write", "run gathers", "stores"]
    print(f"wide_scatter: {len(a)} sort blocks; first start -> last end {(a[:, 7].max() - t0) / 100:.2f} us; start spread {(a[:, 0].max() - t0) / 100:.2f} us")
    for j in range(1, 8):
        d = (a[:, j] - a[:, j - 1]) / 100
        print(f"  {names[j]:24s} mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
    print(f"  block total               mean {((a[:, 7] - a[:, 0]) / 100).mean():6.2f}  max {((a[:, 7] - a[:, 0]) / 100).max():6.2f} us")
b0, b1 = p[0], p[1]
if b0[0]: print("block 0 (tile ranges):", round((b0[1] - b0[0]) / 100, 2), "us")
if b1[0] and b1[8]: print("block 1 (seg_base_body):", round((b1[8] - b1[0]) / 100, 2), "us; starts", round((b1[0] - min(q[0] for q in p[:4000] if q[0])) / 100, 2), "us after the first block")
# depth sort (splitter path): ss_hist slots 3072.., ss_scatter 3328.., bucket_sort 3584..
def phases(lo, n, names, last):
    rows = np.array([q for q in p[lo:lo + n] if q[0] and q[last]])
    if not len(rows): return
    t0 = rows[:, 0].min()
    print(f"{names[0]}: {len(rows)} blocks; first start -> last end {(rows[:, last].max() - t0) / 100:.2f} us")
    for j in range(1, last + 1):
        d = (rows[:, j] - rows[:, j - 1]) / 100
        print(f"  {names[j]:28s} mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
    return rows
phases(3072, 256, ["ss_hist", "search + LDS histogram", "row store"], 2)
phases(3328, 256, ["ss_scatter", "load + ranking", "bases (column sums)", "LDS exchange + stores"], 3)
rows = phases(3584, 512, ["bucket_sort", "load + key bits", "passes", "store", "splitters"], 4)
if rows is not None:
    n = rows[:, 8]; tot = (rows[:, 4] - rows[:, 0]) / 100
    print(f"  bucket sizes: mean {n.mean():.0f}  p90 {np.percentile(n, 90):.0f}  max {n.max()}; passes (bytes that vary): {np.bincount([bin(int(v)).count('1') and sum(1 for s in range(0, 32, 8) if (int(v) >> s) & 255) for v in rows[:, 9]])}")
    i = tot.argmax(); print(f"  slowest block: {tot[i]:.2f} us with {n[i]} records; corr(size, time) {np.corrcoef(n, tot)[0, 1]:.2f}")

# expansion (uncut): marks 10..13 of slot blockIdx.x (slice 0 only)
rows = np.array([q for q in p[:3000] if q[10] and q[13]])
if len(rows):
    t0 = rows[:, 10].min()
    print(f"expand: {len(rows)} blocks; first start -> last end {(rows[:, 13].max() - t0) / 100:.2f} us; start spread {(rows[:, 10].max() - t0) / 100:.2f} us")
    for j, nm in ((11, "block sums of all blocks"), (12, "scan + LDS set-up"), (13, "positions")):
        d = (rows[:, j] - rows[:, j - 1]) / 100
        print(f"  {nm:28s} mean {d.mean():6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} us")
