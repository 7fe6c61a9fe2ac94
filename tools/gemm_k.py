"""K sweep of cpn_gemm_f16 at fixed M, N = 832: time = a + b*K separates the main-loop rate from the K-independent
C-store phase (argument: M)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd._hip import call
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
M, N, ld = int(sys.argv[1]) if len(sys.argv) > 1 else 524288, 832, 896
A = (torch.randn(M, ld, device=dev) * 0.5).half(); W = (torch.randn(N, ld, device=dev) * 0.05).half()
b = torch.randn(N, device=dev); C = torch.empty(M, N, device=dev, dtype=torch.float16)
for K in (64, 128, 256, 448, 864):
    f = lambda: call("cpn_gemm_f16", A.data_ptr(), ld, W.data_ptr(), ld, b.data_ptr(), C.data_ptr(), N, M, N, K, 1, 0, s)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(f"K={K:4d} {e0.elapsed_time(e1)/20:.3f} ms")
