"""GPU box: what one device-to-host transfer of a step's results costs (2.56 MB at 4096 x 50), by destination memory."""
import time
import torch
n = 2_560_000
d = torch.zeros(n, dtype=torch.uint8, device="cuda")
pin = torch.empty(n, dtype=torch.uint8).pin_memory()
pag = torch.empty(n, dtype=torch.uint8)
for name, dst, nb in (("pinned, non_blocking + stream sync", pin, True), ("pageable", pag, False)):
    for _ in range(20):
        dst.copy_(d, non_blocking=nb); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        dst.copy_(d, non_blocking=nb); torch.cuda.current_stream().synchronize()
    dt = (time.perf_counter() - t0) / 200
    print(f"{name:40s} {dt * 1e6:8.1f} us per 2.56 MB = {n / dt / 1e9:6.1f} GB/s")
import numpy as np
a = pin.numpy()
t0 = time.perf_counter()
for _ in range(200):
    b = a.copy()
dt = (time.perf_counter() - t0) / 200
print(f"{'numpy copy out of the pinned buffer':40s} {dt * 1e6:8.1f} us")
small = torch.zeros(16, dtype=torch.uint8, device="cuda"); sp = torch.empty(16, dtype=torch.uint8).pin_memory()
t0 = time.perf_counter()
for _ in range(500):
    sp.copy_(small, non_blocking=True); torch.cuda.current_stream().synchronize()
print(f"{'16-byte transfer (latency floor)':40s} {(time.perf_counter() - t0) / 500 * 1e6:8.1f} us")
