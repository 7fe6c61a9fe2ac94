import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from coponerf_amd import getz
dev = torch.device("cuda:0")
torch.manual_seed(0)
enc = getz.SpatialEncoder().to(dev).eval()
x = torch.randn(2, 3, 256, 256, device=dev)
for det in (False, True):
    for bench in (False, True):
        with torch.backends.cudnn.flags(enabled=True, benchmark=bench, deterministic=det), torch.no_grad():
            for _ in range(3):
                enc(x)
            o = [enc(x) for _ in range(4)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                enc(x)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 10 * 1e3
        d = max(float((a - b).abs().max()) for r in o[1:] for a, b in zip(o[0], r))
        print(f"deterministic={det} benchmark={bench}: {ms:.3f} ms, max run-to-run diff {d:.2e}")
