"""Engine-level: workspace buffers after a fused-key render vs after a separate-key render of the same case."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from coponerf_amd import CoPoNeRF, synthetic as syn
from tests.helpers import load_case, case_inputs, to_device

name = sys.argv[1] if len(sys.argv) > 1 else "c1_val"
cfg, _ = load_case(name)
dev = torch.device("cuda:0")
inp, z, rel, flow = case_inputs(cfg)
model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=cfg["S"])
model.load_state_dict(syn.make_render_weights(), strict=False)
model = model.to(dev).eval()
eng = model._engine
eng.call_lanes = 1
d = lambda o: to_device(o, dev)
inp, z, rel, flow = d(inp), d(z), rel.to(dev), d(flow)
snaps = {}
for rep in range(3):
    for mode in (True, False):
        eng.fuse_key = mode
        with torch.no_grad():
            out = model(inp, z=z, rel_pose=rel, val=cfg["val"], flow=flow, debug=True)
        torch.cuda.synchronize()
        snap = {k: v.clone() for k, v in eng._ws.items()}
        snap["rgb"] = out["rgb"].clone()
        snap["at_wt"] = out["at_wt"].clone()
        snaps[mode] = snap
    a, b = snaps[True], snaps[False]
    print("rep", rep)
    for k in sorted(set(a) & set(b)):
        if a[k].shape != b[k].shape:
            print("  ", k, "shape", a[k].shape, b[k].shape)
            continue
        df = (a[k].float() - b[k].float()).abs()
        print("   %-14s max diff %.3e  mismatching %d / %d" % (k, float(df.max()), int((df > 0).sum()), df.numel()))
V, S = 2, cfg["S"]
a, b = snaps[True]["c0.hid.0" if "c0.hid.0" in snaps[True] else "hid.0"], snaps[False]["hid.0"]
n2 = cfg["B"] * cfg["R"] * V * S * 2
a, b = a[: n2 * 832].view(n2, 832), b[: n2 * 832].view(n2, 832)
bad = (a != b)
rows = bad.any(dim=1).nonzero().flatten().tolist()
print("bad rows", len(rows))
for rr in rows[:40]:
    cols = bad[rr].nonzero().flatten()
    print("row", rr, "(ray,v,s,j)=", (rr // (V * S * 2), (rr // (S * 2)) % V, (rr // 2) % S, rr % 2), "cols", int(cols[0]), "..", int(cols[-1]), "n", cols.numel(),
          "fused", [round(x, 2) for x in a[rr, cols[:4]].tolist()], "sep", [round(x, 2) for x in b[rr, cols[:4]].tolist()])
