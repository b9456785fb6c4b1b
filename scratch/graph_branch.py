"""Do two branches of a captured HIP graph run concurrently?  80 tiny kernels on one stream vs 40 + 40 on two."""
import torch, time
d = torch.device("cuda:0")
a = torch.zeros(1024, device=d); b = torch.zeros(1024, device=d)
big = torch.randn(64, 96, 64, 64, device=d)
def chain(t, n):
    for _ in range(n): t.add_(1.0)
def run(mode):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(); s2 = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        if mode == "serial":
            chain(a, 40); chain(b, 40)
        elif mode == "fork":
            s2.wait_stream(s)
            with torch.cuda.stream(s2):
                chain(b, 40)
            chain(a, 40)
            s.wait_stream(s2)
        elif mode == "fork_events":      # b-kernels on the branch, every a-kernel waits for "its" b-kernel
            s2.wait_stream(s)
            evs = []
            with torch.cuda.stream(s2):
                for _ in range(40):
                    b.add_(1.0); e = torch.cuda.Event(); e.record(s2); evs.append(e)
            for e in evs:
                s.wait_event(e); a.add_(1.0)
            s.wait_stream(s2)
        elif mode == "big_serial":
            chain(b, 40)
            for _ in range(10): big.mul_(1.0001)
        elif mode == "big_fork":
            s2.wait_stream(s)
            with torch.cuda.stream(s2):
                chain(b, 40)
            for _ in range(10): big.mul_(1.0001)
            s.wait_stream(s2)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    print("%-12s %8.1f us per replay" % (mode, (time.perf_counter() - t0) / 50 * 1e6), flush=True)
for m in ("serial", "fork", "fork_events", "big_serial", "big_fork"):
    run(m)
