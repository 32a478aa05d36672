"""Does launching a long HIP graph on a side stream stall the NEXT launch on another stream?  (bench timeline: ~2.4 ms)
side work = 80 elementwise kernels of ~45 us each, as a graph replay or as eager launches; 0.3 ms later a tiny kernel is
launched on the main stream; report when it actually ran relative to the side work's start."""
import time, torch
dev = 'cuda'
x = torch.ones(48 * 1024 * 1024, device=dev)
y = torch.ones(1024, device=dev)
side = torch.cuda.Stream()
def work():
    for _ in range(80):
        x.mul_(1.0000001)
for _ in range(2):
    with torch.cuda.stream(side): work()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        work()
torch.cuda.synchronize()
def trial(mode, delay_s):
    torch.cuda.synchronize()
    e0, e1, e2, e3 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
    main = torch.cuda.current_stream()
    e0.record(main)
    with torch.cuda.stream(side):
        side.wait_event(e0)
        t0 = time.perf_counter()
        if mode == 'graph': g.replay()
        else: work()
        host = time.perf_counter() - t0
        e3.record(side)
    time.sleep(delay_s)
    e1.record(main)
    y.add_(1.0)
    e2.record(main)
    torch.cuda.synchronize()
    return host * 1e3, e0.elapsed_time(e1), e0.elapsed_time(e2), e0.elapsed_time(e3)
for mode in ('eager', 'graph', 'eager', 'graph'):
    for delay in (0.0003, 0.001):
        r = [trial(mode, delay) for _ in range(5)][2:]
        for host, t1, t2, t3 in r[:2]:
            print(f'{mode:6s} delay {delay*1e3:.1f} ms: host launch {host:.3f} ms; main event before tiny kernel at {t1:.3f} ms, tiny kernel done at {t2:.3f} ms; side work done at {t3:.3f} ms')
