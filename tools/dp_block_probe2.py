"""Host time of torch.distributed calls on a 1-rank nccl group while the device is busy."""
import os, sys, time, socket
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from gaussiansplattingmlx_amd.renderer import GaussianRenderer, _p
from gaussiansplattingmlx_amd.scenes import CONFIGS, make_config
name = "c3_300k_800"
idx, N, W, H, kind = CONFIGS[name]
params, cams, _ = make_config(name, n_views=1)
dev = torch.device("cuda", 0)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
r = GaussianRenderer(4, W, H, (16, 16), False)
r.reserve(int(N * 1.5), 24 * 1024 * 1024)
tp = {k: torch.as_tensor(v, device=dev) for k, v in params.items()}
ring = torch.zeros(16, dtype=torch.int32, device=dev)
word = ring[3:4]
big = torch.zeros(1 << 20, device=dev)
big2 = torch.zeros(1 << 20, device=dev)
def probe(label, fn, pre=None):
    ts = []
    for _ in range(6):
        torch.cuda.synchronize()
        r.renderForward(tp, cams[0])
        if pre: pre()
        t0 = time.perf_counter(); out = fn(); ts.append((time.perf_counter() - t0) * 1e6)
        torch.cuda.synchronize()
    print(f"{label:52s} host us: " + " ".join(f"{t:7.1f}" for t in ts), flush=True)
probe("all_reduce(word, async)", lambda: dist.all_reduce(word, op=dist.ReduceOp.MAX, async_op=True))
probe("all_reduce(word, async) after forwardMissed()", lambda: dist.all_reduce(word, op=dist.ReduceOp.MAX, async_op=True), pre=lambda: r.forwardMissed())
probe("copy flag + all_reduce(word, async)", lambda: (r.lib.gs_copy_overflow_flag(r.ctx, _p(word)), dist.all_reduce(word, op=dist.ReduceOp.MAX, async_op=True)))
probe("all_reduce(word) then all_gather(big) async both", lambda: (dist.all_reduce(word, op=dist.ReduceOp.MAX, async_op=True), dist.all_gather_into_tensor(big2, big, async_op=True)))
probe("all_gather(big) async alone", lambda: dist.all_gather_into_tensor(big2, big, async_op=True))
def two_then_wait():
    a = dist.all_reduce(word, op=dist.ReduceOp.MAX, async_op=True)
    b = dist.all_gather_into_tensor(big2, big, async_op=True)
    a.wait(); b.wait()
probe("reduce + gather async, wait both", two_then_wait)
def one_wait():
    a = dist.all_reduce(word, op=dist.ReduceOp.MAX, async_op=True); a.wait()
probe("all_reduce async + wait()", one_wait)
dist.destroy_process_group()
