"""Backward time of the strided Conv4d layers (Encoder4D((1, 8), k, s, p): k3 s2 on 32^4, k5 s4 on 64^4) at B pairs —
what a HIP VJP in place of the library graph (max-pool routing + conv2d backward) could save per training step
(7 such layers per step).   python tools/strided_conv4d_bwd_time.py [B]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import getz                                   # noqa: E402
from coponerf_amd.ufc_ops import HipOps                          # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
ops = HipOps()
for n, (k, s, p) in ((32, (3, 2, 1)), (64, (5, 4, 2))):
    enc = getz.Encoder4D((1, 8), k, s, p).to(dev)
    x = torch.randn(B, 1, n, n, n, n, device=dev, requires_grad=True)
    res = torch.randn(B, 8, 16, 16, 16, 16, device=dev, requires_grad=True)
    def fb():
        y = enc(x, ops, residual=res)
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
        g = torch.ones_like(y)
        t1.record()
        y.backward(g)
        t2.record()
        return t1, t2
    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        x.grad = None
        t1, t2 = fb()
        torch.cuda.synchronize()
        ts.append(t1.elapsed_time(t2))
    with torch.no_grad():
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            enc(x, ops, residual=res)
        torch.cuda.synchronize()
        fwd = (time.perf_counter() - t0) / 5 * 1e3
    print(f"n={n} k{k}s{s}: backward {sorted(ts)[2]:.3f} ms, inference forward {fwd:.3f} ms (B={B})", flush=True)
