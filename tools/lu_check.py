"""cpn_local_units (csrc/local_units.hip): ms per 65 536-ray launch of its three modes and of timing-only ablations.

    python tools/lu_check.py --build     (where hipcc is: variants into tools/_build/)
    python tools/lu_check.py [--rays 65536] [--iters 5]
Flushed regime (2 GB of foreign traffic between launches), as tools/ef_check.py."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
ABLATIONS = {0: "full", 1: "other operand not loaded", 2: "no add rows", 4: "per-row inputs fetched once", 8: "no stores",
             16: "no 128->128 layers", 7: "no loads at all", 15: "no loads, no stores", 31: "first layers + VALU only"}


def build():
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-x", "hip", "-c"]
    for name in ("error.cpp", "streams.cpp"):
        subprocess.check_call(base + [os.path.join(src, name), "-o", os.path.join(BUILD, name.split(".")[0] + ".o")])
    procs = []
    for a in ABLATIONS:
        obj, out = os.path.join(BUILD, f"lu_a{a}.o"), os.path.join(BUILD, f"liblu_a{a}.so")
        procs.append((subprocess.Popen(base + [f"-DCPN_LU_ABLATE={a}", os.path.join(src, "local_units.hip"), "-o", obj]), obj, out))
    for p, obj, out in procs:
        assert p.wait() == 0, obj
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                               os.path.join(BUILD, "streams.o"), "-o", out])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rays", type=int, default=65536)
    a = ap.parse_args()
    if a.build:
        build()
        return
    import torch
    from coponerf_amd import _hip
    dev = torch.device("cuda:0")
    B, V, R, S, n = 1, 2, a.rays, 64, a.rays
    rows = n * V * S
    units = int(_hip.lib().cpn_encode_units(B, R, S, 0, n))
    g = torch.Generator(device="cpu").manual_seed(1)
    rnd = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    loc8, coords9 = rnd(B * V * R * S, 8), rnd(B * V * R, 9)
    w1, b1, w1b, b1b = rnd(128, 16) * 0.2, rnd(128) * 0.1, rnd(128, 16) * 0.2, rnd(128) * 0.1
    w2, wk2 = (rnd(128, 128) * 0.1).half(), (rnd(128, 128) * 0.1).half()
    b2, bk2, add = rnd(128) * 0.1, rnd(128) * 0.1, rnd(n, 128) * 0.1
    khu = (rnd(units * 16, 128) * 0.5).half()
    ce_u = torch.empty(units * 16, 128, dtype=torch.float16, device=dev)
    lg = torch.empty(rows, device=dev)
    lvu = rnd(units * 64, 4)            # the unit-order inputs (cpn_sample_geometry writes them in the product)
    s = torch.cuda.current_stream().cuda_stream
    P, I = ctypes.c_void_p, ctypes.c_int
    dp = lambda t: t.data_ptr()
    flush = torch.zeros(256 << 20, dtype=torch.float32, device=dev)
    res = {}
    for abl, what in ABLATIONS.items():
        path = os.path.join(BUILD, f"liblu_a{abl}.so")
        if not os.path.exists(path):
            continue
        fn = ctypes.CDLL(path).cpn_local_units
        fn.argtypes = [I, P, P, P, I, P, P, P, I, P, P, I, P, P, I, P, P, I, I, I, I, I, I, P, P, P, P]
        calls = {
            "mode0": lambda: fn(0, dp(loc8), dp(coords9), dp(w1), 16, dp(b1), 0, dp(w2), 128, dp(b2), dp(wk2), 128, dp(bk2), 0, 0, 0,
                                dp(khu), B, V, R, S, 0, n, 0, dp(lvu), dp(lg), s),
            "mode0 (scattered inputs)": lambda: fn(0, dp(loc8), dp(coords9), dp(w1), 16, dp(b1), 0, dp(w2), 128, dp(b2), dp(wk2), 128,
                                                   dp(bk2), 0, 0, 0, dp(khu), B, V, R, S, 0, n, 0, 0, dp(lg), s),
            "mode0+store": lambda: fn(0, dp(loc8), dp(coords9), dp(w1), 16, dp(b1), 0, dp(w2), 128, dp(b2), dp(wk2), 128, dp(bk2), 0, 0, 0,
                                      dp(khu), B, V, R, S, 0, n, dp(ce_u), 0, dp(lg), s),
            "mode1": lambda: fn(1, dp(loc8), dp(coords9), dp(w1), 16, dp(b1), dp(add), dp(w2), 128, dp(b2), 0, 0, 0, 0, 0, 0, 0,
                                B, V, R, S, 0, n, dp(ce_u), 0, dp(lg), s),
            "mode2": lambda: fn(2, dp(loc8), dp(coords9), dp(w1), 16, dp(b1), dp(add), dp(w2), 128, dp(b2), dp(wk2), 128, dp(bk2),
                                dp(w1b), 16, dp(b1b), 0, B, V, R, S, 0, n, 0, dp(lvu), dp(lg), s),
            "mode2 (scattered inputs)": lambda: fn(2, dp(loc8), dp(coords9), dp(w1), 16, dp(b1), dp(add), dp(w2), 128, dp(b2), dp(wk2), 128,
                                                   dp(bk2), dp(w1b), 16, dp(b1b), 0, B, V, R, S, 0, n, 0, 0, dp(lg), s),
        }
        row = {}
        for name, run in calls.items():
            for _ in range(2):
                assert run() == 0
            torch.cuda.synchronize()
            tot = 0.0
            for _ in range(a.iters):
                flush.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            row[name] = round(tot / a.iters, 3)
        res[f"a{abl}: {what}"] = row
        print(f"a{abl:<3d} {what:34s}", " ".join(f"{k}={v:.3f}" for k, v in row.items()), flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
