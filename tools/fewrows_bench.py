"""The per-ray value projection (M rays x 416 x K = 1664, fp32 out) at the callers' call size: the tiled cpn_gemm_f16 against
cpn_gemm_f16_fewrows, back to back and behind 1 GB of foreign traffic (us per launch)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd._hip import call          # noqa: E402

dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
flush = torch.zeros(128 << 20, dtype=torch.float32, device=dev)
res = {}
N, K = 416, 1664
W = (torch.randn(N, K, device=dev) * 0.05).half()
Wp = torch.empty(N * K, dtype=torch.float16, device=dev)
call("cpn_pack_gemm_frags", W.data_ptr(), K, N, K, Wp.data_ptr(), s)
b = torch.randn(N, device=dev)
for M in (1024, 3641, 8192, 16384):
    A = (torch.randn(M, K, device=dev) * 0.5).half()
    C1, C2 = torch.empty(M, N, device=dev), torch.empty(M, N, device=dev)
    forms = {"tiled": lambda: call("cpn_gemm_f16", A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), C1.data_ptr(), N, M, N, K, 0, 1, s),
             "few rows": lambda: call("cpn_gemm_f16_fewrows", A.data_ptr(), K, Wp.data_ptr(), b.data_ptr(), C2.data_ptr(), N, M, N, K, 0, s)}
    for name, run in forms.items():
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        hot = e0.elapsed_time(e1) / 20 * 1e3
        tot = 0.0
        for _ in range(5):
            flush.add_(1)
            e0.record()
            run()
            e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1) * 1e3 / 5
        res[f"M={M} {name}"] = {"back to back us": round(hot, 1), "flushed us": round(tot, 1)}
    res[f"M={M} equal"] = bool(torch.equal(C1, C2))
print(json.dumps(res, indent=1))
