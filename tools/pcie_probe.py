"""PCIe rates of the box for the sizes of the C4 batch (pinned host memory, torch): the floor of the host-pointer path."""
import time
import torch
n = 134217728
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for name, fn in (("H2D", lambda: d.copy_(h, non_blocking=True)), ("D2H", lambda: h.copy_(d, non_blocking=True))):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s %d MB pinned: %.3f ms = %.1f GB/s" % (name, n >> 20, dt * 1e3, n / dt / 1e9))
p = torch.empty(n, dtype=torch.uint8)
torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(p); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("H2D pageable: %.3f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
import numpy as np
a = np.empty(n, np.uint8); b = np.empty(n, np.uint8)
t0 = time.perf_counter(); b[:] = a; dt = time.perf_counter() - t0
print("host memcpy 128 MB: %.3f ms = %.1f GB/s" % (dt * 1e3, n / dt / 1e9))
