"""cpn_wgrad_f32 against the library's TN GEMM + column sum on the Linear-layer shapes of one training step
(profiles/r04_aten_train.txt lists them).   python tools/wgrad_f32_bench.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd.ufc_ops import wgrad_f32      # noqa: E402

dev = torch.device("cuda:0")
SHAPES = [(32768, 256, 1024), (32768, 1024, 256), (32768, 512, 256), (32768, 256, 256), (8192, 256, 1024), (8192, 1024, 256),
          (8192, 256, 256), (2048, 256, 1024), (2048, 1024, 256), (16384, 128, 128), (16384, 128, 416), (16384, 416, 128),
          (512, 256, 256), (32768, 512, 2304)]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


rows = []
for R, O, I in SHAPES:
    dY = torch.randn(R, O, device=dev)
    X = torch.randn(R, I, device=dev)
    t_lib = timed(lambda: (dY.t() @ X, dY.sum(0)))
    t_hip = timed(lambda: wgrad_f32(dY, X, True))
    rows.append({"R": R, "O": O, "I": I, "library_us": round(t_lib, 1), "cpn_wgrad_f32_us": round(t_hip, 1),
                 "tflops": round(2.0 * R * O * I / t_hip / 1e6, 1)})
    print(json.dumps(rows[-1]), flush=True)
