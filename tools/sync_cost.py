"""Diagnostic: what does the host pay for 'everything is done' on an idle device with 5 streams? (bench.py's barrier)"""
import time, torch
dev = torch.device("cuda", 0)
streams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(4)]
x = torch.zeros(1 << 20, device=dev)
def work():
    for s in streams:
        with torch.cuda.stream(s):
            x.add_(1.0)
def t(f, n=200):
    best = []
    for _ in range(n):
        work()
        evs = [torch.cuda.Event() for _ in streams]
        for e, s in zip(evs, streams): e.record(s)
        while not all(e.query() for e in evs): pass
        t0 = time.perf_counter(); f(evs); best.append(time.perf_counter() - t0)
    best.sort()
    return 1e6 * best[len(best) // 2]
print("device idle, median us:")
print("  torch.cuda.synchronize        %.1f" % t(lambda evs: torch.cuda.synchronize(dev)))
print("  stream.synchronize x5         %.1f" % t(lambda evs: [s.synchronize() for s in streams]))
print("  event.synchronize x5          %.1f" % t(lambda evs: [e.synchronize() for e in evs]))
print("  event.query x5                %.1f" % t(lambda evs: [e.query() for e in evs]))
