"""What the PCIe link sustains with both directions busy (the host-memory query stream moves 29 MiB up and 14 MiB down per C3
query): H2D alone, D2H alone, both at once on two streams.  GPU box: python tools/pcie_duplex_probe.py"""
import time
import torch

dev = torch.device("cuda:0")
up_h = torch.zeros(29 * 2**20 // 8, dtype=torch.int64).pin_memory()
up_d = torch.empty_like(up_h, device=dev)
dn_d = torch.zeros(14 * 2**20 // 8, dtype=torch.int64, device=dev)
dn_h = torch.empty(dn_d.shape, dtype=torch.int64).pin_memory()
s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def timed(f, n=30):
    f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def up():
    with torch.cuda.stream(s1):
        up_d.copy_(up_h, non_blocking=True)


def down():
    with torch.cuda.stream(s2):
        dn_h.copy_(dn_d, non_blocking=True)


def both():
    up()
    down()


tu, td, tb = timed(up), timed(down), timed(both)
print("H2D 29 MiB alone: %.3f ms (%.1f GB/s)" % (tu, 29 * 2**20 / tu / 1e6))
print("D2H 14 MiB alone: %.3f ms (%.1f GB/s)" % (td, 14 * 2**20 / td / 1e6))
print("both directions at once: %.3f ms per pair (%.1f GB/s up + down)" % (tb, 43 * 2**20 / tb / 1e6))

