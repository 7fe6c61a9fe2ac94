"""The slot schedule of RenderEngine._render_body against the serial chunk loop: ms per 65 536-ray image and whether the
outputs agree bit for bit.   python tools/slot_probe.py [slot_rays ...]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import CoPoNeRF, synthetic as syn              # noqa: E402

dev = torch.device("cuda:0")
B = int(os.environ.get("PROBE_B", "1"))
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=64)
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
mv = lambda o: {k: mv(v) for k, v in o.items()} if isinstance(o, dict) else (o.to(dev) if torch.is_tensor(o) else o)
inp = mv(syn.make_inputs(B, 256, 256, 0, seed=101, full_image=True))
z, rel, flow = syn.make_latents(B, 256, 256, seed=201)
z, rel, flow = [t.to(dev) for t in z], rel.to(dev), [t.to(dev) for t in flow]
eng = model._engine
eng.call_lanes = 1


def run(steps=6):
    with torch.no_grad():
        for _ in range(2):
            out = model(inp, z=z, rel_pose=rel, val=True, flow=flow)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = model(inp, z=z, rel_pose=rel, val=True, flow=flow)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, {k: out[k].clone() for k in ("rgb", "at_wt", "z")
                                                      if k in out and torch.is_tensor(out[k])}


res = {}
eng.slot_rays = 0
t, ref = run()
res["serial (one chunk)"] = round(t, 2)
print("serial", round(t, 2), list(ref), flush=True)
for sr in [int(a) for a in sys.argv[1:]] or [16384, 8192, 4096, 2048]:
    eng.slot_rays = sr
    t, got = run()
    same = {k: bool(torch.equal(got[k], ref[k])) for k in ref}
    res[f"slots of {sr} rays"] = [round(t, 2), same]
    print(sr, round(t, 2), same, flush=True)
if os.environ.get("PROBE_TRACE"):
    eng.slot_rays = int(os.environ["PROBE_TRACE"])
    eng.slot_trace = []
    with torch.no_grad():
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        out = model(inp, z=z, rel_pose=rel, val=True, flow=flow)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
    torch.cuda.synchronize()
    rows = [(k, round(e0.elapsed_time(go), 3), round(go.elapsed_time(ee), 3), round(go.elapsed_time(se), 3))
            for k, go, ee, se in eng.slot_trace]
    print("slot, start (ms into the call), encoder ms, sums end ms after start")
    for r in rows:
        print(r)
    print("call", round(e0.elapsed_time(e1), 3))
    eng.slot_trace = None
eng.slot_rays = 0
t, _ = run()
res["serial again"] = round(t, 2)
print(json.dumps(res, indent=1))
