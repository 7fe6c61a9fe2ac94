"""What the reference caller's `torch.cat(outputs, dim=-3)` of the 18 CPU `pixel_val` chunks (test.py:207, 67 MB per
256x256x64 image) and the later release of that tensor cost on this host, with and without glibc's mmap path."""
import ctypes
import sys
import time

import torch

pin = torch.cuda.is_available()
chunks = [torch.randn(2, 3641, 64, 2).pin_memory() if pin else torch.randn(2, 3641, 64, 2) for _ in range(18)]


def probe(tag):
    res = []
    for _ in range(6):
        t0 = time.perf_counter()
        full = torch.cat(chunks, dim=-3)
        t1 = time.perf_counter()
        del full
        t2 = time.perf_counter()
        res.append((round((t1 - t0) * 1e3, 2), round((t2 - t1) * 1e3, 2)))
    print(tag, "(cat ms, free ms):", res)


print("threads", torch.get_num_threads())
probe("default malloc")
libc = ctypes.CDLL("libc.so.6")
M_TRIM_THRESHOLD, M_MMAP_THRESHOLD, M_MMAP_MAX = -1, -3, -4
libc.mallopt(M_MMAP_MAX, 0)
libc.mallopt(M_TRIM_THRESHOLD, 1 << 30)
probe("M_MMAP_MAX=0, M_TRIM_THRESHOLD=1GiB")
torch.set_num_threads(1)
probe("same, 1 thread")
