import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd._hip import call
dev = torch.device("cuda:0"); s = torch.cuda.current_stream().cuda_stream
M, N, K, ld = int(sys.argv[1]) if len(sys.argv) > 1 else 524288, 832, 864, 896
A = (torch.randn(M, ld, device=dev) * 0.5).half(); W = (torch.randn(N, ld, device=dev) * 0.05).half()
b = torch.randn(N, device=dev); C = torch.empty(M, N, device=dev, dtype=torch.float16)
for _ in range(3):
    call("cpn_gemm_f16", A.data_ptr(), ld, W.data_ptr(), ld, b.data_ptr(), C.data_ptr(), N, M, N, K, 1, 0, s)
torch.cuda.synchronize()
