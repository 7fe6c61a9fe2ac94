"""Does cpn_attend_hidden (reads hid of chunk c, 6.5 TB/s alone) run UNDER cpn_encode_key of chunk c + 1 (writes at 2.5 TB/s,
bound by its L1 / LDS / lock step) when the encoder leaves room on its CUs?  The product encoder takes 12 waves x 160
VGPRs (3 per SIMD: 480 of 512 registers) - nothing else fits beside it, two streams alternate.  With 8 waves (2 per SIMD)
a third of the register file and 4.5 KiB of LDS stay free: room for 2 workgroups of attend_hidden (66 VGPRs, 544 B).

    python tools/coresident_probe.py --build      (where hipcc is)
    python tools/coresident_probe.py              (on the GPU)
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
WAVES = (12, 8)
UNROLLS = (4, 16)


def build():
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c"]
    for name in ("error.cpp", "streams.cpp"):
        subprocess.check_call(base + [os.path.join(src, name), "-o", os.path.join(BUILD, name.split(".")[0] + ".o")])
    for w in WAVES:
        obj, out = os.path.join(BUILD, f"ekco_w{w}.o"), os.path.join(BUILD, f"libekco_w{w}.so")
        subprocess.check_call(base + [f"-DCPN_EK_WAVES={w}", "-DCPN_EK_REGS_FOR=3", os.path.join(src, "encode_key.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                               os.path.join(BUILD, "streams.o"), "-o", out])
    for u in UNROLLS:
        obj, out = os.path.join(BUILD, f"att_u{u}.o"), os.path.join(BUILD, f"libatt_u{u}.so")
        subprocess.check_call(base + [f"-DCPN_ATTEND_UNROLL={u}", os.path.join(src, "attend.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, os.path.join(BUILD, "error.o"),
                               os.path.join(BUILD, "streams.o"), "-o", out])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--iters", type=int, default=4)
    ap.add_argument("--rays", type=int, default=16384)
    a = ap.parse_args()
    if a.build:
        build()
        return
    import torch
    from coponerf_amd import CoPoNeRF, _hip, synthetic as syn
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
    T = V * S
    hid_w = torch.empty(n * T * 2, 832, dtype=torch.float16, device=dev)          # the encoder's target (chunk c + 1)
    kh = torch.empty(n * T, 128, dtype=torch.float16, device=dev)
    hid_r = torch.randn(n * T * 2, 832, device=dev).to(torch.float16)             # the readers' source (chunk c)
    lg = torch.randn(n * T, device=dev)
    hbar = torch.empty(n, 1664, dtype=torch.float16, device=dev)
    hbar2 = torch.empty(n, 1664, dtype=torch.float16, device=dev)
    flush = torch.zeros(256 << 20, dtype=torch.float32, device=dev)
    P, I = ctypes.c_void_p, ctypes.c_int
    lib = _hip.lib()
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    sa, sb, sc = (torch.cuda.Stream(device=dev) for _ in range(3))
    res = {"rays": n}
    att = {}
    for u in UNROLLS:
        path = os.path.join(BUILD, f"libatt_u{u}.so")
        if os.path.exists(path):
            f = ctypes.CDLL(path).cpn_attend_hidden
            f.argtypes = [P, P, P, P, I, I, I, I, I, I, P, P, P]
            f.restype = I
            att[u] = f
    cur = [att[4]]

    def attend(st, out):
        rc = cur[0](0, 0, lg.data_ptr(), hid_r.data_ptr(), B, V, R, S, 0, n, out.data_ptr(), 0, st.cuda_stream)
        assert rc == 0

    for wv in WAVES:
        path = os.path.join(BUILD, f"libekco_w{wv}.so")
        if not os.path.exists(path):
            continue
        fn = ctypes.CDLL(path).cpn_encode_key
        fn.argtypes = [P, P, I, I, P, P, P, P, P, P, I, P, P, I, I, I, I, I, I, P, P, P]
        fn.restype = I

        def encode(st):
            rc = fn(tabs[0].data_ptr(), maps[3].data_ptr(), H, H, g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(),
                    g["pe6"].data_ptr(), w["enc.frag"].data_ptr(), w["query_encode_latent.b"].data_ptr(),
                    w["enc.k80blk"].data_ptr(), 0, w["key_fold.w16"].data_ptr(), w["key_fold.b"].data_ptr(), B, V, R, S, 0,
                    n, hid_w.data_ptr(), kh.data_ptr(), st.cuda_stream)
            assert rc == 0, rc

        def timed(body, sts):
            """[wall, end of each stream's work] in ms after the common start, mean over the iterations"""
            tot = [0.0] * (1 + len(sts))
            for it in range(a.iters + 1):
                flush.add_(1)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                main = torch.cuda.current_stream()
                e0.record(main)
                for st in sts:
                    st.wait_event(e0)
                body()
                ends = []
                for st in sts:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record(st)
                    main.wait_event(ev)
                    ends.append(ev)
                e1.record(main)
                torch.cuda.synchronize()
                if it:
                    tot[0] += e0.elapsed_time(e1)
                    for i, ev in enumerate(ends):
                        tot[1 + i] += e0.elapsed_time(ev)
            out = [round(t / a.iters, 3) for t in tot]
            return out[0] if len(sts) == 1 else out

        for u in att:
            if wv == 12 and u != 4:
                continue
            cur[0] = att[u]
            sname = f"attend unroll {u}"
            r = {}
            r["encode alone"] = timed(lambda: encode(sa), (sa,))
            r["2 x attend alone (one stream)"] = timed(lambda: (attend(sb, hbar), attend(sb, hbar2)), (sb,))
            r["2 x attend alone (two streams)"] = timed(lambda: (attend(sb, hbar), attend(sc, hbar2)), (sb, sc))
            r["encode first, then 2 x attend on a second stream"] = timed(
                lambda: (encode(sa), attend(sb, hbar), attend(sb, hbar2)), (sa, sb))
            r["encode first, then attends on two more streams"] = timed(
                lambda: (encode(sa), attend(sb, hbar), attend(sc, hbar2)), (sa, sb, sc))
            r["encode + 1 attend"] = timed(lambda: (encode(sa), attend(sb, hbar)), (sa, sb))
            res[f"{wv} waves, {sname}"] = r
            print(wv, sname, json.dumps(r), flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
