import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from coponerf_amd.ufc_ops import HipOps
dev = torch.device("cuda:0")
hip = HipOps()
for B in (1, 2, 4):
    a, b = torch.randn(B, 4096, 256, device=dev), torch.randn(B, 4096, 256, device=dev)
    with torch.no_grad():
        for _ in range(5): hip.correlation_tokens(a, b, 64)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): hip.correlation_tokens(a, b, 64)
        e1.record(); torch.cuda.synchronize()
    print(B, "us per call (l2norm x1-2 + gemm)", e0.elapsed_time(e1) / 30 * 1e3)
