"""Does Event.synchronize() return when the work enqueued BEFORE the record has finished, or does it also wait for work enqueued
behind it?  A (5 ms) ; record e ; B (5 ms) ; time e.synchronize()."""
import time, torch
dev = torch.device('cuda', 0)
x = torch.randn(8192, 8192, device=dev)
def busy(n):
    y = x
    for _ in range(n):
        y = y @ x
    return y
busy(2); torch.cuda.synchronize()
t0 = time.perf_counter(); busy(10); torch.cuda.synchronize(); per = (time.perf_counter() - t0) / 10
n = max(1, int(5e-3 / per))
for mode in ('plain', 'timing', 'blocking'):
    for trial in range(3):
        torch.cuda.synchronize()
        e = torch.cuda.Event(enable_timing=(mode == 'timing'), blocking=(mode == 'blocking'))
        t0 = time.perf_counter()
        busy(n)
        e.record()
        t1 = time.perf_counter()
        busy(n)
        t2 = time.perf_counter()
        e.synchronize()
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print(f'{mode}: A+B enqueued in {1e3 * (t2 - t0):.2f} ms; e.synchronize() returned {1e3 * (t3 - t0):.2f} ms after start; everything done at {1e3 * (t4 - t0):.2f} ms (A alone ~ {1e3 * n * per:.2f} ms)')
# with a pinned D2H copy in front of the record, as ClipRunner.launch has
pin = torch.empty(16, dtype=torch.int32).pin_memory()
src = torch.zeros(16, dtype=torch.int32, device=dev)
for trial in range(3):
    torch.cuda.synchronize()
    e = torch.cuda.Event()
    t0 = time.perf_counter()
    busy(n); pin.copy_(src, non_blocking=True); e.record(); busy(n)
    e.synchronize(); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    print(f'with D2H before the record: e.synchronize() returned {1e3 * (t3 - t0):.2f} ms after start; everything done at {1e3 * (t4 - t0):.2f} ms')
