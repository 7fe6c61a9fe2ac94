"""Does a second pass over a range of `hid` that fits the Infinity Cache (256 MB) run faster than the first?
Decides whether fusing the first hidden sum into the key-path kernel (re-reading the workgroup's own tile) can pay.
cpn_attend_hidden over one 16 384-ray chunk (7 GB) in sub-ranges of n rays (n x 426 KB): each sub-range once, and each
sub-range twice back to back; with non-temporal loads (the product) and with plain loads.
    python tools/mall_probe.py --build   (where hipcc is)      python tools/mall_probe.py   (on the GPU)"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tools", "_build")
if "--build" in sys.argv:
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    err_o = os.path.join(BUILD, "error.o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", os.path.join(src, "error.cpp"), "-o", err_o])
    for nt in (0, 1):
        obj, out = os.path.join(BUILD, f"attend_nt{nt}.o"), os.path.join(BUILD, f"libattend_nt{nt}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCPN_ATTEND_NT={nt}", "-x", "hip", "-c",
                               os.path.join(src, "attend.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, err_o, "-o", out])
    sys.exit(0)

import torch                                                   # noqa: E402

dev = torch.device("cuda:0")
C, V, S = 16384, 2, 64
T = V * S
hid = torch.empty(C * T, 1664, dtype=torch.float16, device=dev).uniform_(0, 1)
lg = torch.randn(C * T, device=dev)
hbar = torch.empty(C, 1664, dtype=torch.float16, device=dev)
P, I = ctypes.c_void_p, ctypes.c_int
res = {}
for nt in (1, 0):
    lib = ctypes.CDLL(os.path.join(BUILD, f"libattend_nt{nt}.so"))
    fn = lib.cpn_attend_hidden
    fn.argtypes = [P, P, P, P, I, I, I, I, I, I, P, P, P]
    st = torch.cuda.current_stream().cuda_stream

    def sweep(n, passes):
        for r0 in range(0, C, n):
            for _ in range(passes):
                rc = fn(None, None, lg.data_ptr() + r0 * T * 4, hid.data_ptr() + r0 * T * 1664 * 2, 1, V, n, S, 0, n,
                        hbar.data_ptr() + r0 * 1664 * 2, None, st)
                assert rc == 0, rc

    for n in (64, 128, 256, 512, 1024, 4096, 16384):
        out = []
        for passes in (1, 2):
            sweep(n, passes)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                sweep(n, passes)
            e1.record()
            torch.cuda.synchronize()
            out.append(e0.elapsed_time(e1) / 3)
        res[f"nt{nt} n={n} ({n * T * 3328 / 2**20:.0f} MB)"] = {"one_pass_ms": round(out[0], 3), "two_passes_ms": round(out[1], 3),
                                                                "second_pass_ms": round(out[1] - out[0], 3)}
        print(list(res.items())[-1], flush=True)
print(json.dumps(res))
