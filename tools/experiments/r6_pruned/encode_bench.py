"""Time the first encoder layer of one 16 384-ray chunk of the configs[1] workload in both forms:
cpn_encode_hidden (projected tables + K=96 MFMA) and cpn_gather_rows + cpn_gemm_f16 (835 -> 832).
Usage: python tools/encode_bench.py [--rays 16384] [--iters 20] [--only tables|fused|gather]   (prints one JSON line)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, _hip, synthetic as syn       # noqa: E402
from coponerf_amd._hip import call                               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=16384)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="both")
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--rig", default="narrow")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    H = S = None
    H, S, B, V = a.height, a.samples, 1, 2
    model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
    model.load_state_dict(syn.make_render_weights(), strict=False)
    model = model.to(dev).eval()
    eng = model._engine
    inp = syn.make_inputs(B, H, H, 0, seed=100, full_image=True, rig=a.rig)
    z, rel, flow = syn.make_latents(B, H, H, seed=200)
    mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (
        o.to(dev) if torch.is_tensor(o) else type(o)(mv(v) for v in o))
    inp, z, rel = mv(inp), mv(z), rel.to(dev)
    w = eng._weights(model._render_params())
    maps, tabs = eng._feature_maps(z, w)
    ctx, qry = inp["context"], inp["query"]
    g = eng._geometry(ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], qry["uv"], rel, True, S, H, H)
    R = qry["uv"].shape[2]
    n = a.rays
    rows2 = n * V * S * 2
    hid = torch.empty(rows2, 832, dtype=torch.float16, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    res = {"rays": n, "rows": rows2, "height": H, "samples": S, "rig": a.rig}

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.iters

    if a.only in ("both", "tables"):
        def enc():
            call("cpn_encode_hidden", tabs[0].data_ptr(), maps[3].data_ptr(), H, H,
                 g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), g["pe6"].data_ptr(), w["enc.frag"].data_ptr(),
                 w["query_encode_latent.b"].data_ptr(), B, V, R, S, min(16384, R - n), n, hid.data_ptr(), s)
        ms = timeit(enc)
        res["encode_hidden_ms"] = ms
        res["encode_hidden_alg_tflops"] = 2.0 * rows2 * 832 * 835 / ms / 1e9
        res["encode_hidden_hid_GBs"] = rows2 * 1664 / ms / 1e6
    if a.only in ("both", "fused"):
        kh = torch.empty(rows2 // 2 + 65536, 128, dtype=torch.float16, device=dev)      # unit order: + the dead rows of partial units

        def enck():
            call("cpn_encode_key", tabs[0].data_ptr(), maps[3].data_ptr(), H, H,
                 g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), g["pe6"].data_ptr(), w["enc.frag"].data_ptr(),
                 w["query_encode_latent.b"].data_ptr(), w["key_fold.wpk"].data_ptr(),
                 w["key_fold.b"].data_ptr(), B, V, R, S, min(16384, R - n), n, hid.data_ptr(), kh.data_ptr(), 1, s)
        ms = timeit(enck)
        res["encode_key_ms"] = ms
        res["encode_key_hid_GBs"] = rows2 * 1664 / ms / 1e6
    if a.only in ("both", "gather"):
        xin = torch.empty(rows2, _hip.XIN_STRIDE, dtype=torch.float16, device=dev)

        def gat():
            call("cpn_gather_rows", maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(), maps[3].data_ptr(), H, H,
                 g["pixel_val"].data_ptr(), g["sec_grid"].data_ptr(), g["pe6"].data_ptr(), B, V, R, S, 16384, n,
                 xin.data_ptr(), s)

        def gem():
            call("cpn_gemm_f16", xin.data_ptr(), _hip.XIN_STRIDE, w["query_encode_latent.w16"].data_ptr(), _hip.XIN_STRIDE,
                 w["query_encode_latent.b"].data_ptr(), hid.data_ptr(), 832, rows2, 832, _hip.XIN_K, 1, 0, s)
        res["gather_rows_ms"] = timeit(gat)
        res["gemm_ms"] = timeit(gem)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
