"""A/B of the fused key layer (cpn_encode_key + cpn_gemm_f16_rowdot) against the round-3 kernels (cpn_encode_hidden +
cpn_gemm_f16_chain_rowdot) in ONE process on ONE box: full-image calls on one stream, the two modes alternating."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn      # noqa: E402

dev = torch.device("cuda:0")
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=64)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else type(o)(mv(v) for v in o))
inp = mv(syn.make_inputs(1, 256, 256, 0, seed=100, full_image=True))
z, rel, flow = syn.make_latents(1, 256, 256, seed=200)
z, rel, flow = mv(z), rel.to(dev), mv(flow)
eng = model._engine
eng.call_lanes = 1
res = {True: [], False: []}
with torch.no_grad():
    for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
        for mode in (True, False):
            eng.fuse_key = mode
            for _ in range(2):
                model(inp, z=z, rel_pose=rel, val=True, flow=flow)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(6):
                model(inp, z=z, rel_pose=rel, val=True, flow=flow)
            torch.cuda.synchronize()
            res[mode].append((time.perf_counter() - t0) / 6 * 1e3)
out = {("fused" if k else "separate"): {"ms_per_image_rounds": [round(x, 2) for x in v], "mean": round(sum(v) / len(v), 3), "min": round(min(v), 3)}
       for k, v in res.items()}
out["fused_minus_separate_ms"] = round(out["fused"]["mean"] - out["separate"]["mean"], 3)
print(json.dumps(out))
