"""cpn_encode_key / cpn_encode_project (csrc/encode_fused.hip): shapes and ablations against the separate kernels, and their timing.

    python tools/ef_check.py --build          (where hipcc is: variants of encode_fused.hip into tools/_build/)
    python tools/ef_check.py [--rays 65536] [--iters 5] [--hot] [--only TAG]

Correctness (full configs[1] launch): hid / kh of the key form must be BIT-IDENTICAL to cpn_encode_hidden + cpn_gemm_f16 (same MFMA
shapes and k order); kh of the project form likewise, and its val (rows, 416) fp16 must agree with hid . value_fold^T (fp32 product of the
same fp16 operands) to fp16 rounding.  Timing: ms per launch with 2 GB of foreign traffic between launches (the chunk-loop
regime), `--hot` back to back."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
# tag -> -D flags
VARIANTS = {
    "k_w12u1s2": dict(KEY_WAVES=12, KEY_UNITS=1, KEY_SITE=2),               # the product form
    "k_w12u1s2_stage": dict(KEY_WAVES=12, KEY_UNITS=1, KEY_SITE=2, STAGE=1),
    "k_w12u1s1": dict(KEY_WAVES=12, KEY_UNITS=1, KEY_SITE=1),
    "k_w12u1s2_p1": dict(KEY_WAVES=12, KEY_UNITS=1, KEY_SITE=2, PRIO=1),
    "k_w8u2s2": dict(KEY_WAVES=8, KEY_UNITS=2, KEY_SITE=2),
    "k_w8u2s1": dict(KEY_WAVES=8, KEY_UNITS=2, KEY_SITE=1),
    "p_w8u1": dict(PROJ_WAVES=8, PROJ_UNITS=1),                              # the product form of cpn_encode_project
    "p_w8u1_dma": dict(PROJ_WAVES=8, PROJ_UNITS=1, STAGE_PROJECT=0),
    "p_w4u2": dict(PROJ_WAVES=4, PROJ_UNITS=2),
}
ABLATIONS = {1: "no table taps", 2: "no hid / val stores", 4: "no K=80 MFMA", 8: "no key/value MFMA + ring reads", 64: "no ring DMA / barrier",
             72: "no ring, no key/value MFMA", 256: "no DMA, barrier kept", 512: "no barrier, DMA kept", 1024: "16 of the step's pieces fetched",
             2048: "no bpermute of the K=80 accumulators", 4096: "no bpermute of hid to the B layout", 6144: "no bpermutes at all"}
ABLATE_TAGS = ("k_w12u1s2", "p_w8u1")


def build(only=""):
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-x", "hip", "-c"]
    for name in ("error.cpp", "streams.cpp"):
        subprocess.check_call(base + [os.path.join(src, name), "-o", os.path.join(BUILD, name.split(".")[0] + ".o")])
    r4src = os.path.join(ROOT, "tools", "experiments", "encode_key_r4.hip")
    if not only or "r4" in only:
        # the round-4 kernel (one unit per wave, taps issued at the top of their slice), kept as the timing reference
        obj = os.path.join(BUILD, "ek_r4.o")
        subprocess.check_call(base + ["-I", src, r4src, "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                               os.path.join(BUILD, "streams.o"), "-o", os.path.join(BUILD, "libek_r4.so")])
    jobs = [(tag, flags, 0) for tag, flags in VARIANTS.items()]
    jobs += [(tag, VARIANTS[tag], a) for tag in ABLATE_TAGS for a in ABLATIONS]
    procs = []
    for tag, flags, abl in jobs:
        full = tag + (f"_a{abl}" if abl else "")
        if only and only not in full:
            continue
        obj, out = os.path.join(BUILD, f"ef_{full}.o"), os.path.join(BUILD, f"libef_{full}.so")
        d = [f"-DCPN_EF_{k}={v}" for k, v in flags.items()] + [f"-DCPN_EF_ABLATE={abl}"]
        procs.append((subprocess.Popen(base + d + [os.path.join(src, "encode_fused.hip"), "-o", obj]), obj, out))
        if len(procs) >= 8:
            _finish(procs, hipcc)
    _finish(procs, hipcc)


def _finish(procs, hipcc):
    for p, obj, out in procs:
        assert p.wait() == 0, obj
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                               os.path.join(BUILD, "streams.o"), "-o", out])
    procs.clear()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--rays", type=int, default=65536)
    ap.add_argument("--hot", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    if a.build:
        build(a.only)
        return
    import torch
    from coponerf_amd import CoPoNeRF, _hip, synthetic as syn
    from coponerf_amd.render import pack_k80_blocks, pack_project_ring
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
    rows = n * V * S
    hid = torch.empty(rows * 2, 832, dtype=torch.float16, device=dev)
    kh = torch.empty(rows, 128, dtype=torch.float16, device=dev)
    val = torch.empty(rows, 416, dtype=torch.float16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    P, I = ctypes.c_void_p, ctypes.c_int
    k80 = pack_k80_blocks(w["enc.frag"], w["query_encode_latent.b"])
    wring_p = pack_project_ring(w["key_fold.w16"], w["value_fold.w16"], k80)
    res = {"rays": n, "regime": "hot" if a.hot else "flushed"}

    # ---- reference: the separate kernels of the product library (cpn_encode_hidden, then the folded key layer as a GEMM on hid)
    hid_ref = torch.empty_like(hid)
    kh_ref = torch.empty_like(kh)
    geo = (tabs[0].data_ptr(), maps[3].data_ptr(), H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), g["pe6"].data_ptr())

    def run_ref():
        _hip.call("cpn_encode_hidden", *geo, w["enc.frag"].data_ptr(), w["query_encode_latent.b"].data_ptr(), B, V, R, S, 0, n,
                  hid_ref.data_ptr(), s)

    def run_ref_key():
        _hip.call("cpn_gemm_f16", hid_ref.data_ptr(), 1664, w["key_fold.w16"].data_ptr(), 1664, w["key_fold.b"].data_ptr(),
                  kh_ref.data_ptr(), 128, rows, 128, 1664, 1, 0, s)

    def run_r4():
        rc = r4(*geo, w["enc.frag"].data_ptr(), w["query_encode_latent.b"].data_ptr(), k80.data_ptr(), 4, w["key_fold.wpk"].data_ptr(),
                w["key_fold.b"].data_ptr(), B, V, R, S, 0, n, hid.data_ptr(), kh.data_ptr(), s)
        assert rc == 0, rc

    flush = torch.zeros(256 << 20, dtype=torch.float32, device=dev)          # 1 GB buffer: add_ = 2 GB of traffic

    def timed(run):
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
        return round(tot / a.iters, 3)

    res["cpn_encode_hidden (16 waves, no key layer)"] = timed(run_ref)
    run_ref_key()
    r4path = os.path.join(BUILD, "libek_r4.so")
    if os.path.exists(r4path) and (not a.only or "r4" in a.only):
        r4 = ctypes.CDLL(r4path).cpn_encode_key_r4
        r4.argtypes = [P, P, I, I, P, P, P, P, P, P, I, P, P, I, I, I, I, I, I, P, P, P]
        r4.restype = I
        res["round-4 cpn_encode_key (12 waves x 1 unit, tools/experiments/encode_key_r4.hip)"] = timed(run_r4)
    val_ref = None
    if not a.no_check:
        # val reference on a strided subset of sample rows (the full product is 11.6 TFLOP in fp32)
        idx = torch.arange(0, rows, 997, device=dev)
        val_ref = (hid_ref.view(rows, 1664)[idx].float() @ w["value_fold.w16"].float().t())

    # "k_product": cpn_encode_key of the product library as it was last built (coponerf_amd/libcoponerf_hip.so) - with the source
    # edited and not yet rebuilt, the previous kernel beside the new variants on the same box
    tags = ["k_product"] + [t for t in VARIANTS] + [f"{t}_a{k}" for t in ABLATE_TAGS for k in ABLATIONS]
    for tag in tags:
        path = _hip.LIB_PATH if tag == "k_product" else os.path.join(BUILD, f"libef_{tag}.so")
        if (a.only and a.only not in tag and tag != "k_product") or not os.path.exists(path):
            continue
        lib = ctypes.CDLL(path)
        project = tag.startswith("p_")
        if project:
            fn = lib.cpn_encode_project
            fn.argtypes = [P, P, I, I, P, P, P, P, P, I, I, I, I, I, I, P, P, I, P]
            fn.restype = I

            def run():
                rc = fn(tabs[0].data_ptr(), maps[3].data_ptr(), H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(),
                        g["pe6"].data_ptr(), wring_p.data_ptr(), w["key_fold.b"].data_ptr(), B, V, R, S, 0, n, kh.data_ptr(),
                        val.data_ptr(), 0, s)
                assert rc == 0, rc
        else:
            fn = lib.cpn_encode_key
            fn.argtypes = [P, P, I, I, P, P, P, P, P, P, P, I, I, I, I, I, I, P, P, I, P]
            fn.restype = I

            def run():
                rc = fn(tabs[0].data_ptr(), maps[3].data_ptr(), H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(),
                        g["pe6"].data_ptr(), w["enc.frag"].data_ptr(), w["query_encode_latent.b"].data_ptr(),
                        w["key_fold.wpk"].data_ptr(), w["key_fold.b"].data_ptr(), B, V, R, S, 0, n, hid.data_ptr(), kh.data_ptr(), 0, s)
                assert rc == 0, rc
        entry = {}
        if not a.no_check and "_a" not in tag:
            hid.zero_(); kh.zero_(); val.zero_()
            run()
            torch.cuda.synchronize()
            entry["kh_bit_identical"] = bool(torch.equal(kh, kh_ref))
            if not entry["kh_bit_identical"]:
                entry["kh_mismatches"] = int((kh != kh_ref).sum())
                entry["kh_max_abs_diff"] = float((kh.float() - kh_ref.float()).abs().max())
            if project:
                d = (val[idx].float() - val_ref).abs()
                entry["val_max_abs_err"] = float(d.max())
                entry["val_ref_max_abs"] = float(val_ref.abs().max())
                entry["val_rel_l2"] = float(d.norm() / val_ref.norm())
            else:
                entry["hid_bit_identical"] = bool(torch.equal(hid, hid_ref))
                if not entry["hid_bit_identical"]:
                    entry["hid_mismatches"] = int((hid != hid_ref).sum())
        entry["ms"] = timed(run)
        res[tag] = entry
        print(tag, json.dumps(entry), flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
