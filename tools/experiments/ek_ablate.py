"""Phase ablation of cpn_encode_key (timing only - ablated results are wrong): variants of csrc/encode_key.hip with
-DCPN_EK_ABLATE=k / -DCPN_EK_WAVES=w in tools/_build/ (`--build` where hipcc is), timed on one launch of configs[1]
(default: the one-chunk shape, 65 536 rays) with 2 GB of foreign traffic between launches (the chunk-loop regime)."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
VARIANTS = {0: "full", 8: "no key MFMA / ring reads", 64: "no ring DMA, no barrier", 72: "no ring, no key MFMA (= encode + 8 bpermute)",
            16: "key MFMAs on ONE LDS fragment (16 -> 1 ds_read_b128 per slice)", 32: "K = 80 MFMAs on one LDS fragment (12 -> 2 reads)",
            48: "both: 28 -> 3 LDS reads per slice", 2: "no hid stores", 10: "no stores, no key MFMA", 1: "no table taps", 66: "no ring/barrier, no stores", 128: "no kh stores"}
WAVES = (12,)
PHASES = (1, 2, 3)          # CPN_EK_PHASES: barrier site per wave (1 = all late, 2 = waves 4-7 early, 3 = all early)


def build():
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c"]
    for name in ("error.cpp", "streams.cpp"):
        subprocess.check_call(base + [os.path.join(src, name), "-o", os.path.join(BUILD, name.split(".")[0] + ".o")])
    for w in WAVES:
        for k, ph in [(k, 2) for k in VARIANTS] + [(0, 1), (0, 3), (0, 4), (8, 1), (2, 1)]:
            tag = f"{k}_w{w}" + (f"p{ph}" if ph != 2 else "")
            obj, out = os.path.join(BUILD, f"ek_{tag}.o"), os.path.join(BUILD, f"libek_{tag}.so")
            subprocess.check_call(base + [f"-DCPN_EK_ABLATE={k}", f"-DCPN_EK_WAVES={w}", f"-DCPN_EK_PHASES={ph}",
                                          os.path.join(src, "encode_key.hip"), "-o", obj])
            subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                                   os.path.join(BUILD, "streams.o"), "-o", out])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--ray0", type=int, default=0)
    ap.add_argument("--hot", action="store_true", help="back-to-back launches instead of the flushed regime")
    ap.add_argument("--only", default="")
    ap.add_argument("--groups", type=lambda t: [int(x) for x in t.split(",")], default=[4, 0, 3], help="cpn_encode_key group values to time")
    a = ap.parse_args()
    if a.build:
        build()
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
    kh = torch.empty(n * V * S, 128, dtype=torch.float16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    P, I = ctypes.c_void_p, ctypes.c_int
    flush = torch.zeros(256 << 20, dtype=torch.float32, device=dev)          # 1 GB buffer: add_ = 2 GB of traffic
    res = {}
    combos = [(w_, k, v, 2, grp) for w_ in WAVES for grp in a.groups for k, v in VARIANTS.items()]
    combos += [(12, 0, "full", 1, 0), (12, 0, "full", 3, 0), (12, 0, "full", 4, 0)]
    for w_, k, what, ph, grp in combos:
        if True:
            label = f"w{w_} group {grp} phases {ph} {k}: {what}"
            path = os.path.join(BUILD, f"libek_{k}_w{w_}" + (f"p{ph}" if ph != 2 else "") + ".so")
            if (a.only and a.only not in label) or not os.path.exists(path):
                continue
            fn = ctypes.CDLL(path).cpn_encode_key
            fn.argtypes = [P, P, I, I, P, P, P, P, P, P, I, P, P, I, I, I, I, I, I, P, P, P]
            fn.restype = I

            def run():
                rc = fn(tabs[0].data_ptr(), maps[3].data_ptr(), H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(),
                        g["pe6"].data_ptr(), w["enc.frag"].data_ptr(), w["query_encode_latent.b"].data_ptr(),
                        w["enc.k80blk"].data_ptr(), grp, w["key_fold.wpk" if grp == 4 else "key_fold.w16"].data_ptr(), w["key_fold.b"].data_ptr(), B, V, R, S, a.ray0,
                        n, hid.data_ptr(), kh.data_ptr(), s)
                assert rc == 0, rc
            for _ in range(2):
                run()
            torch.cuda.synchronize()
            tot = 0.0
            for _ in range(a.iters):
                if not a.hot:
                    flush.add_(1)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                run()
                e1.record()
                torch.cuda.synchronize()
                tot += e0.elapsed_time(e1)
            res[label] = round(tot / a.iters, 3)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
