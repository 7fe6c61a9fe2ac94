import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd._hip import call
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
M, N, K = 16384 * 256, 832, 864
for rnd in range(2):
    for ld in (864, 896):
        A = (torch.randn(M, ld, device=dev) * 0.5).half(); W = (torch.randn(N, ld, device=dev) * 0.05).half()
        b = torch.randn(N, device=dev); C = torch.empty(M, N, device=dev, dtype=torch.float16)
        f = lambda: call("cpn_gemm_f16", A.data_ptr(), ld, W.data_ptr(), ld, b.data_ptr(), C.data_ptr(), N, M, N, K, 1, 0, s)
        for _ in range(2): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"ld={ld} M={M} {ms:.3f} ms {2.0*M*N*835/ms/1e9:.1f} TF")
        del A, W, C
