"""Raw host->device copy rate from pinned memory and pageable->pinned staging rate (context for tools/bench_feed.py)."""
import time, torch
from concurrent.futures import ThreadPoolExecutor
n = 64 * 80 * 4096
src = torch.randn(n)
pin = torch.empty(n).pin_memory()
dev = torch.empty(n, device="cuda:0")
torch.cuda.synchronize()
for _ in range(2):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    dev.copy_(pin, non_blocking=True)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / 10
print("H2D pinned 84 MB: %.2f ms = %.1f GB/s" % (t * 1e3, n * 4 / t / 1e9))
t0 = time.perf_counter()
for _ in range(5):
    pin.copy_(src)
t = (time.perf_counter() - t0) / 5
print("pageable->pinned, 1 call: %.2f ms = %.1f GB/s" % (t * 1e3, n * 4 / t / 1e9))
pool = ThreadPoolExecutor(4)
t0 = time.perf_counter()
for _ in range(5):
    js = [pool.submit(pin[i * n // 4:(i + 1) * n // 4].copy_, src[i * n // 4:(i + 1) * n // 4]) for i in range(4)]
    [j.result() for j in js]
t = (time.perf_counter() - t0) / 5
print("pageable->pinned, 4 threads: %.2f ms = %.1f GB/s" % (t * 1e3, n * 4 / t / 1e9))
