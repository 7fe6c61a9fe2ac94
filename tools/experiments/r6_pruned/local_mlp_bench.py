"""cpn_local_mlp on one 16 384-ray chunk of configs[1], both call forms of the render loop (coords_embed written; second
query dotted against it), behind 512 MB of foreign traffic as in the chunk loop, with timing-only ablations built into
tools/_build/ (-DCPN_LMLP_ABLATE=k).   python tools/local_mlp_bench.py --build   |   python tools/local_mlp_bench.py"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BUILD = os.path.join(ROOT, "tools", "_build")
VARIANTS = {0: "full", 1: "no layer-1 MFMA", 2: "no layer-2 MFMA", 3: "no MFMA", 4: "no add loads", 8: "no dot_with loads",
            12: "no add / dot_with loads", 16: "no stores", 31: "loop + input loads only",
            100: "full, dot_with rows in the load layout + 16 ds_bpermute (CPN_LMLP_DOT_LOAD_LAYOUT=1)"}
if "--build" in sys.argv:
    os.makedirs(BUILD, exist_ok=True)
    src = os.path.join(ROOT, "coponerf_amd", "csrc")
    hipcc = "/opt/rocm/bin/hipcc"
    err_o = os.path.join(BUILD, "error.o")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", os.path.join(src, "error.cpp"), "-o", err_o])
    for k in VARIANTS:
        obj, out = os.path.join(BUILD, f"gather_lmlp{k}.o"), os.path.join(BUILD, f"libgather_lmlp{k}.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-DCPN_LMLP_ABLATE={k % 100}", f"-DCPN_LMLP_DOT_LOAD_LAYOUT={1 if k >= 100 else 0}", "-x", "hip", "-c",
                               os.path.join(src, "gather.hip"), "-o", obj])
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", obj, err_o, "-o", out])
    sys.exit(0)

import torch                                                    # noqa: E402
from coponerf_amd import CoPoNeRF, synthetic as syn             # noqa: E402

dev = torch.device("cuda:0")
H, S, B, V, n = 256, 64, 1, 2, 16384
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
eng = model._engine
inp = syn.make_inputs(B, H, H, 0, seed=100, full_image=True)
z, rel, flow = syn.make_latents(B, H, H, seed=200)
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else type(o)(mv(v) for v in o))
inp, z, rel = mv(inp), mv(z), rel.to(dev)
w = eng._weights(model._render_params())
ctx, qry = inp["context"], inp["query"]
g = eng._geometry(ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], qry["uv"], rel, True, S, H, H)
R = qry["uv"].shape[2]
rows = n * V * S
ce = torch.empty(rows, 128, dtype=torch.float16, device=dev)
lg = torch.empty(rows, dtype=torch.float32, device=dev)
addq = torch.randn(n, 128, device=dev)
junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
P, I = ctypes.c_void_p, ctypes.c_int
st = torch.cuda.current_stream().cuda_stream
res = {}
FRAG = int(os.environ.get("LMLP_FRAG", "1"))           # coords_embed in fragment order (the product) or row-major
for k, name in VARIANTS.items():
    lib = ctypes.CDLL(os.path.join(BUILD, f"libgather_lmlp{k}.so"))
    fn = lib.cpn_local_mlp
    fn.argtypes = [P, P, P, I, P, P, P, I, P, I, I, I, I, I, I, P, P, P, I, P]
    dp = lambda t: t.data_ptr()
    def first():
        return fn(dp(g["loc8"]), dp(g["coords9"]), dp(w["query_embed.w"]), 16, dp(w["query_embed.b"]), None, dp(w["query_embed_2.w16"]), 128,
                  dp(w["query_embed_2.b"]), B, V, R, S, 0, n, dp(ce), None, None, FRAG, st)
    def second():
        return fn(dp(g["loc8"]), dp(g["coords9"]), dp(w["query_repeat_embed.w_l"]), 16, dp(w["query_repeat_embed.b"]), dp(addq),
                  dp(w["query_repeat_embed_2.w16"]), 128, dp(w["query_repeat_embed_2.b"]), B, V, R, S, 0, n, None, dp(ce), dp(lg), FRAG, st)
    out = []
    for call_ in (first, second):
        ts = []
        for it in range(6):
            junk.add_(1)                                      # 1 GB of foreign traffic: the chunk loop's cache state
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            assert call_() == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        out.append(round(sorted(ts[1:])[2], 4))
    res[f"{k}: {name}"] = {"writes coords_embed": out[0], "second query -> logits": out[1]}
    print(k, name, out, flush=True)
print(json.dumps(res))
