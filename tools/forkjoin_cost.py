import torch, time
dev = torch.device("cuda", 0)
a = torch.zeros(1 << 20, device=dev)      # 4 MB: ~3 us kernel
big = torch.zeros(64 << 20, device=dev)   # 256 MB fill: ~60 us HBM-bound
s2 = torch.cuda.Stream()
main = torch.cuda.current_stream()
def chain(n):
    for _ in range(n): a.add_(1.0)
def run(fork, iters=200):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        a.add_(1.0)                       # "kernel A"
        if fork:
            ev = torch.cuda.Event(); ev.record(main)
            s2.wait_event(ev)
            with torch.cuda.stream(s2):
                big.add_(1.0)             # "kernel B" (HBM-bound, independent)
            ev2 = torch.cuda.Event(); ev2.record(s2)
        else:
            big.add_(1.0)
        chain(14)                          # the "sort" chain: 14 small dependent kernels
        if fork:
            main.wait_event(ev2)
        a.add_(1.0)                       # "blend"
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for _ in range(2):
    print("serial  us/iter %.1f" % run(False), " forked us/iter %.1f" % run(True))
# pure overhead: B tiny
big = torch.zeros(1 << 10, device=dev)
for _ in range(2):
    print("tiny B: serial %.1f forked %.1f" % (run(False), run(True)))
