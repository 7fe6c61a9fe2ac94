"""Phase ablation of cpn_encode_hidden (timing only — ablated results are wrong): builds variants of csrc/encode.hip with
-DCPN_ENCODE_ABLATE=k into tools/_build/ (run `python tools/encode_ablate.py --build` where hipcc is, e.g. in the
build container: the .so files travel to the GPU box) and times each on one 16 384-ray chunk of configs[1]."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
# kernel configurations (CPN_ENCODE_MT images per wave tile, CPN_ENCODE_WAVES per workgroup)
CONFIGS = ((1, 16), (2, 8))
# CPN_ENCODE_STORE: 0 = write-back, 1 = non-temporal inside an asm (rounds 2-3), 7 = unconditional nt buffer store the
# compiler counts in vmcnt (round 4, the product)
STORES = (0, 1, 7)
FULL_ABLATION = ((1, 16),)          # the other configurations are only timed in full
VARIANTS = {0: "full", 1: "no table taps", 2: "no hid stores", 3: "no taps, no stores (MFMA + setup)", 4: "no MFMA",
            16: "all taps -> node 0 (L1-hot)", 18: "node-0 taps, no stores",
            32: "stores wrap into a 1.7 MB window (L2-resident)", 33: "no taps, stores into the window"}


def build():
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    err_o = os.path.join(BUILD, "error.o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                           os.path.join(src, "error.cpp"), "-o", err_o])
    # cpn_stream_cus() (persistent grids ask how many CUs their stream may use) lives in streams.cpp
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c",
                           os.path.join(src, "streams.cpp"), "-o", os.path.join(BUILD, "streams.o")])
    for cfg in CONFIGS:
        mt, waves = cfg
        for k in VARIANTS:
            if k and cfg not in FULL_ABLATION:
                continue
            tag = f"{k}_mt{mt}w{waves}"
            obj, out = os.path.join(BUILD, f"encode_abl{tag}.o"), os.path.join(BUILD, f"libencode_abl{tag}.so")
            cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCPN_ENCODE_ABLATE={k}",
                   f"-DCPN_ENCODE_MT={mt}", f"-DCPN_ENCODE_WAVES={waves}", "-x", "hip", "-c",
                   os.path.join(src, "encode.hip"), "-o", obj]
            print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, err_o, os.path.join(BUILD, "streams.o"), "-o", out])


PREFETCH = ((1, 12), (1, 8))          # (CPN_ENCODE_PREFETCH, waves per workgroup)


def build_prefetch_variants():
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    for pf, waves in PREFETCH + ((0, 12),):
        obj, out = os.path.join(BUILD, f"encode_pf{pf}w{waves}.o"), os.path.join(BUILD, f"libencode_pf{pf}w{waves}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCPN_ENCODE_PREFETCH={pf}",
                               f"-DCPN_ENCODE_WAVES={waves}", "-Rpass-analysis=kernel-resource-usage", "-x", "hip", "-c",
                               os.path.join(src, "encode.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                               os.path.join(BUILD, "streams.o"), "-o", out])


def build_store_variants():
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    for st in STORES:
        obj, out = os.path.join(BUILD, f"encode_store{st}.o"), os.path.join(BUILD, f"libencode_store{st}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCPN_ENCODE_STORE={st}",
                               "-x", "hip", "-c", os.path.join(src, "encode.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                               os.path.join(BUILD, "streams.o"), "-o", out])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--build-only", default="")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--ray0", type=int, default=16384, help="first ray of the timed chunk")
    ap.add_argument("--rays", type=int, default=16384, help="rays of the timed chunk (65536 with --ray0 0 = the one-chunk launch)")
    ap.add_argument("--flush", action="store_true", help="stream 8 GB through the caches between launches (as the "
                    "key / attend kernels of the chunk loop do) and time each launch on its own")
    ap.add_argument("--flush-kind", default="rw", choices=("rw", "read", "write"), help="foreign traffic of the flush: "
                    "add_ (read + write), sum (read only) or zero_ (write only)")
    ap.add_argument("--flush-mb", type=int, default=4096, help="size of the flush stream buffer (fp32 add_: 2x traffic)")
    ap.add_argument("--warm", default="", choices=("", "tables", "geometry", "all", "pages", "plain"), help="with --flush: read the node "
                    "tables + level-3 map / the per-sample geometry arrays once after the flush, before the timed launch")
    ap.add_argument("--alt-hid", action="store_true", help="hot loop writing two hid buffers alternately")
    ap.add_argument("--only", default="", help="substring filter on the variant label")
    a = ap.parse_args()
    if a.build:
        if a.build_only != "prefetch":
            build()
            build_store_variants()
        build_prefetch_variants()
        return
    import torch
    from coponerf_amd import CoPoNeRF, synthetic as syn
    dev = torch.device("cuda:0")
    H, S, B, V, n = 256, 64, 1, 2, a.rays
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(syn.make_render_weights(), strict=False)
    model = model.to(dev).eval()
    eng = model._engine
    inp = syn.make_inputs(B, H, H, 0, seed=100, full_image=True)
    z, rel, flow = syn.make_latents(B, H, H, seed=200)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (
        o.to(dev) if torch.is_tensor(o) else type(o)(mv(v) for v in o))
    inp, z, rel = mv(inp), mv(z), rel.to(dev)
    w = eng._weights(model._render_params())
    maps, tabs = eng._feature_maps(z, w)
    ctx, qry = inp["context"], inp["query"]
    g = eng._geometry(ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], qry["uv"], rel, True, S, H, H)
    R = qry["uv"].shape[2]
    hid = torch.empty(n * V * S * 2, 832, dtype=torch.float16, device=dev)
    hids = [hid, torch.empty_like(hid)] if a.alt_hid else [hid]
    it_no = [0]
    s = torch.cuda.current_stream().cuda_stream
    P, I = ctypes.c_void_p, ctypes.c_int
    res = {}
    runs = [(f"mt{c[0]} w{c[1]} {k}: {what}", f"libencode_abl{k}_mt{c[0]}w{c[1]}.so")
            for c in CONFIGS for k, what in VARIANTS.items()]
    runs += [(f"store policy {st}", f"libencode_store{st}.so") for st in STORES]
    runs += [(f"prefetch {pf} w{wv}", f"libencode_pf{pf}w{wv}.so") for pf, wv in PREFETCH + ((0, 12),)]
    wlib = wsink = None
    if a.warm == "plain":
        wlib = ctypes.CDLL(os.path.join(BUILD, "libwrite_bw.so"))
        wlib.wb_read.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        wsink = torch.zeros(1, dtype=torch.int32, device=dev)
    flush_buf = torch.zeros(a.flush_mb << 18, dtype=torch.float32, device=dev) if a.flush else None
    for label, so in runs:
        if a.only and a.only not in label:
            continue
        path = os.path.join(BUILD, so)
        if not os.path.exists(path):
            continue
        fn = ctypes.CDLL(path).cpn_encode_hidden
        fn.argtypes = [P, P, I, I, P, P, P, P, P, I, I, I, I, I, I, P, P]
        fn.restype = I

        def run():
            rc = fn(tabs[0].data_ptr(), maps[3].data_ptr(), H, H,
                    g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), g["pe6"].data_ptr(), w["enc.frag"].data_ptr(),
                    w["query_encode_latent.b"].data_ptr(), B, V, R, S, a.ray0, n, hids[it_no[0] % len(hids)].data_ptr(), s)
            it_no[0] += 1
            assert rc == 0, rc
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        if a.flush:
            tot = 0.0
            for _ in range(a.iters):
                if a.flush_kind == "rw":
                    flush_buf.add_(1)
                elif a.flush_kind == "read":
                    flush_sink = flush_buf.sum()
                else:
                    flush_buf.zero_()
                if a.warm in ("tables", "all"):
                    warm_sink = tabs[0].view(torch.int32).sum() + maps[3].view(torch.int32).sum()
                if a.warm == "plain":          # tables, level-3 map and geometry arrays through plain (cached) loads
                    for t in (tabs[0], maps[3], g["pixel_val"], g["sec_grid"], g["pe6"]):
                        wlib.wb_read(t.data_ptr(), t.numel() * t.element_size() // 16, 1024, wsink.data_ptr(), s)
                if a.warm == "pages":          # one element per 4 KiB page of the output, tables and geometry arrays
                    warm_sink = sum(t.view(-1).view(torch.int16 if t.element_size() == 2 else torch.int32)
                                    [::4096 // t.element_size()].sum()
                                    for t in (hids[it_no[0] % len(hids)], tabs[0], g["pixel_val"], g["sec_grid"], g["pe6"]))
                if a.warm in ("geometry", "all"):
                    warm_sink = g["pixel_val"].sum() + g["sec_grid"].sum() + g["pe6"].sum()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            res[label] = round(tot / a.iters, 3)
            continue
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[label] = round(e0.elapsed_time(e1) / a.iters, 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
