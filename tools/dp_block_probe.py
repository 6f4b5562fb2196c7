"""Which call of the data-parallel step blocks the host?  Queues ~0.4 ms of device work, then times one call on the host."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gaussiansplattingmlx_amd import _lib
from gaussiansplattingmlx_amd.renderer import GaussianRenderer, _p
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config

name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=1)
dev = torch.device("cuda", 0)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=dev) for k, v in params.items()}
uid = C.create_string_buffer(_lib.GS_DP_UNIQUE_ID_BYTES)
assert r.lib.gs_dp_unique_id(uid) == 0
r._check(r.lib.gs_dp_init(r.ctx, C.c_char_p(uid.raw), 0, 1))
word = torch.zeros(4, dtype=torch.int32, device=dev)
buf = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
ev = torch.cuda.Event()
side = torch.cuda.Stream()
def probe(label, fn):
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        r.renderForward(tp, cams[0])
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    print(f"{label:40s} host us: " + " ".join(f"{t:7.1f}" for t in ts), flush=True)
probe("nothing", lambda: None)
probe("gs_copy_overflow_flag", lambda: r.lib.gs_copy_overflow_flag(r.ctx, _p(word)))
probe("gs_dp_allreduce_sum 4 MB", lambda: r.lib.gs_dp_allreduce_sum(r.ctx, _p(buf), 1 << 20))
probe("torch event record + side wait", lambda: (ev.record(), side.wait_event(ev)))
probe("torch copy_ 4 B d2d", lambda: word[1:2].copy_(word[0:1]))
probe("gs_set_update_gate", lambda: r.lib.gs_set_update_gate(r.ctx, _p(word)))
