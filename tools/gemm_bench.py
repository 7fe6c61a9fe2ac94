"""GPU microbenchmark of cpn_gemm_f16 on the dominant shapes of the render path (argument: rays per launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd._hip import call
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
torch.manual_seed(0)
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
M2 = rays * 256
shapes = [("enc1 835->832 relu", M2, 832, 864, 896, 835, 1, 0), ("enc2 832->416", M2, 416, 832, 832, 832, 0, 0),
          ("value 832->416 f32", M2 // 2, 416, 832, 832, 832, 0, 1), ("key 832->128 relu", M2 // 2, 128, 832, 832, 832, 1, 0),
          ("key2 128->128", M2 // 2, 128, 128, 128, 128, 0, 0)]
for name, M, N, K, ld, kalg, relu, f32 in shapes:
    A = (torch.randn(M, ld, device=dev) * 0.5).half()
    W = (torch.randn(N, ld, device=dev) * 0.05).half()
    A[:, K:] = 0; W[:, K:] = 0
    b = torch.randn(N, device=dev)
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.float16)
    f = lambda: call("cpn_gemm_f16", A.data_ptr(), ld, W.data_ptr(), ld, b.data_ptr(), C.data_ptr(), N, M, N, K, relu, f32, s)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    ref = A[:4096, :K].float() @ W[:, :K].float().t() + b
    if relu: ref = ref.clamp_min(0)
    err = (C[:4096].float() - ref).abs().max().item()
    print(f"{name:22s} M={M} {ms:8.3f} ms  {2.0*M*N*kalg/ms/1e9:8.1f} TFLOP/s (alg)  maxerr {err:.2e}")
