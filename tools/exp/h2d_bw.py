"""Pinned / pageable host -> device bandwidth of the box (the floor of the per-entry upload): python tools/exp/h2d_bw.py"""
import time, torch
n = 64 << 20
for pinned in (True, False):
    h = torch.empty(n, dtype=torch.uint8, pin_memory=pinned)
    h.fill_(1)
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%s 64 MiB H2D: %.2f ms = %.1f GB/s" % ("pinned" if pinned else "pageable", 1e3 * dt, n / dt / 1e9))
