"""cpn_conv_wgrad_planes on the training shapes of the 4-D convolutions (B = 4, 16^4 volumes): microseconds per call per
channel pair, against the time the operands take to stream once.  `--build` (where hipcc is) compiles csrc/ufc.hip with other
workgroup counts into tools/_build/.  Usage: python tools/conv_wgrad_bench.py [--build]"""
import ctypes
import os
import subprocess
import sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
BLOCKS = (256, 512, 1024)
if "--build" in sys.argv:
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    os.makedirs(BUILD, exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                           os.path.join(src, "error.cpp"), "-o", os.path.join(BUILD, "error.o")])
    for nb in BLOCKS:
        obj, out = os.path.join(BUILD, f"ufc_wg{nb}.o"), os.path.join(BUILD, f"libufc_wg{nb}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCPN_WG_BLOCKS={nb}", "-x", "hip",
                               "-c", os.path.join(src, "ufc.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"), "-o", out])
    sys.exit(0)

dev = torch.device("cuda:0")
B, G, H, W = 4, 256, 16, 16
st = torch.cuda.current_stream().cuda_stream
P = ctypes.c_void_p
for nb in BLOCKS:
    path = os.path.join(BUILD, f"libufc_wg{nb}.so")
    if not os.path.exists(path):
        continue
    lib = ctypes.CDLL(path)
    fn = lib.cpn_conv_wgrad_planes
    fn.argtypes = [P, P] + [ctypes.c_int] * 6 + [P, P, P, P]
    lib.cpn_conv_wgrad_scratch.restype = ctypes.c_longlong
    for cin, cout in ((1, 8), (8, 8), (8, 32), (32, 8)):
        x = torch.randn(B, cin, G, H, W, device=dev)
        dy = torch.randn(B, cout, G, H, W, device=dev)
        part = torch.empty(lib.cpn_conv_wgrad_scratch(cin, cout), device=dev)
        dw, db = torch.empty(cout, cin, 3, 3, device=dev), torch.empty(cout, device=dev)
        run = lambda: fn(x.data_ptr(), dy.data_ptr(), B, cin, cout, G, H, W, part.data_ptr(), dw.data_ptr(), db.data_ptr(), st)
        assert run() == 0
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(20):
            run()
        t1.record()
        torch.cuda.synchronize()
        us = t0.elapsed_time(t1) / 20 * 1e3
        stream_us = (x.numel() + dy.numel()) * 4 / 5e12 * 1e6
        print(f"{nb:5d} workgroups  {cin:2d} -> {cout:2d}: {us:7.1f} us per call (operands stream in {stream_us:5.1f} us)")
