"""cpn_wgrad_tall_f16 on the training shape (832 x 896 over 4.2 M rows) against torch.mm (hipBLASLt) on the same operands.
`--build` (where hipcc is) compiles the timing-only phase ablations of csrc/wgrad_tall.hip into tools/_build/; they are timed
when present.  Usage: python tools/wgrad_bench.py [--build] [--product-only] [launches ...]"""
import ctypes
import os
import subprocess
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import _hip                      # noqa: E402
from coponerf_amd._hip import call                 # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "tools", "_build")
VARIANTS = {1: "no global loads", 2: "no LDS stage writes", 3: "no loads, no stage", 4: "no fragment reads", 8: "no MFMA",
            12: "no fragment reads, no MFMA", 15: "loop skeleton"}
if "--build" in sys.argv:
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    os.makedirs(BUILD, exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                           os.path.join(src, "error.cpp"), "-o", os.path.join(BUILD, "error.o")])
    for k in VARIANTS:
        obj, out = os.path.join(BUILD, f"wgrad_tall_abl{k}.o"), os.path.join(BUILD, f"libwgrad_tall_abl{k}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCPN_WT_ABLATE={k}", "-x", "hip",
                               "-c", os.path.join(src, "wgrad_tall.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"), "-o", out])
    sys.exit(0)

dev = torch.device("cuda:0")
M, N, K = 4 * 4096 * 2 * 64 * 2, 832, 896
g = torch.Generator(device=dev).manual_seed(1)
dY = (torch.randn(M, N, device=dev, generator=g) * 0.1).half()
X = torch.randn(M, K, device=dev, generator=g).half()
part = torch.empty(_hip.lib().cpn_wgrad_tall_scratch(N, K), device=dev)
dW = torch.empty(N, K, device=dev)
s = torch.cuda.current_stream().cuda_stream


def own():
    call("cpn_wgrad_tall_f16", dY.data_ptr(), N, X.data_ptr(), K, M, N, K, 0, part.data_ptr(), dW.data_ptr(), s)


def lib():
    return torch.mm(dY.t(), X, out_dtype=torch.float32)


iters = [int(a) for a in sys.argv[1:] if a.isdigit()] or [5]
for name, fn, n in [(nm, f, n) for n in iters for nm, f in (("cpn_wgrad_tall_f16", own), ("torch.mm", lib))]:
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(n):
        fn()
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / n
    print(f"[{n} launches]", end=" ")
    print(f"{name:22s} {ms:7.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s  operands at {(M * (N + K) * 2) / ms / 1e9:6.2f} TB/s")
ref = lib()
print("max rel diff vs library:", float((dW - ref).abs().max() / ref.abs().max()))

for k, what in list(VARIANTS.items()):
    path = os.path.join(BUILD, f"libwgrad_tall_{k}.so" if isinstance(k, str) else f"libwgrad_tall_abl{k}.so")
    if not os.path.exists(path):
        continue
    fn = ctypes.CDLL(path).cpn_wgrad_tall_f16
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    run = lambda: fn(dY.data_ptr(), N, X.data_ptr(), K, M, N, K, None, part.data_ptr(), dW.data_ptr(), s)
    run()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(5):
        run()
    t1.record()
    torch.cuda.synchronize()
    print(f"variant {k!s:>3} ({what:28s}) {t0.elapsed_time(t1) / 5:7.3f} ms")
