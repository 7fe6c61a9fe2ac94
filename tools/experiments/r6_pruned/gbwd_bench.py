"""Microbenchmark of cpn_gather_rows_bwd on training-shaped inputs (B pairs x R random rays x S samples); with
`--build` (where hipcc is) a variant without the accumulation phase is built into tools/_build/ and timed beside it:
the difference is what the LDS read-modify-writes cost, the rest is the chunk / row scan."""
import ctypes
import os
import subprocess
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libgbwd_nodrain.so")
if "--build" in sys.argv:
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    for tag, extra in (("nodrain", ["-DCPN_GBWD_NO_DRAIN"]), ("tpy8", ["-DCPN_GBWD_TPY=8"]), ("tpy2", ["-DCPN_GBWD_TPY=2"])):
        objs = []
        for f, ex in (("error.cpp", []), ("backward.hip", extra)):
            o = os.path.join(ROOT, "tools", "_build", f"gbwd_{tag}_" + f.split(".")[0] + ".o")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                                   os.path.join(src, f), "-o", o] + ex)
            objs.append(o)
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs +
                              ["-o", SO.replace("nodrain", tag)])
    sys.exit(0)
sys.path.insert(0, ROOT)
import torch
from coponerf_amd import synthetic as syn
from coponerf_amd.render import RenderEngine
from coponerf_amd._hip import call

B, R, S, H = 4, 4096, 64, 256
dev = torch.device("cuda:0")
inp = syn.make_inputs(B, H, H, R, seed=61)
eng = RenderEngine()
c, q = inp["context"], inp["query"]
g = eng._geometry(c["cam2world"].to(dev), c["intrinsics"].to(dev), q["cam2world"].to(dev), q["intrinsics"].to(dev),
                  q["uv"].to(dev), None, False, S, H, H)
rows = B * R * 2 * S * 2
dx = (torch.randn(rows, 896, device=dev) * 0.1).half()
shapes = [(2 * B, 16, 16, 256), (2 * B, 32, 32, 256), (2 * B, 64, 64, 256), (2 * B, 256, 256, 64)]
st = torch.cuda.current_stream().cuda_stream
from coponerf_amd import _hip
boxes = torch.empty(B * 2 * _hip.lib().cpn_gather_bwd_chunks(R, S) * 16, dtype=torch.int32, device=dev)
P, I = ctypes.c_void_p, ctypes.c_int
variants = [("product", None)]
for tag, label in (("nodrain", "no accumulation (scan only)"), ("tpy8", "8 x 8 pixel tiles (round 1)"), ("tpy2", "8 x 2 pixel tiles")):
    so = SO.replace("nodrain", tag)
    if os.path.exists(so) and "--product-only" not in sys.argv:
        fn = ctypes.CDLL(so).cpn_gather_rows_bwd
        fn.argtypes = [P, I, I, I, P, P, I, I, I, I, I, I, P, P, P, P, P, P]
        variants.append((label, fn))
for cfg, fn in variants:
    ts = []
    for it in range(4):
        dm = [torch.zeros(s, device=dev) for s in shapes]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        args = (dx.data_ptr(), 896, H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), B, 2, R, S,
                0, B * R, dm[0].data_ptr(), dm[1].data_ptr(), dm[2].data_ptr(), dm[3].data_ptr(), boxes.data_ptr(), st)
        if fn is None:
            call("cpn_gather_rows_bwd", *args)
        else:
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"cfg={cfg or 'default':24s} ms={min(ts):8.2f}  sums={[float(d.double().sum()) for d in dm]}")
