"""Diagnostic (GPU box): stage-by-stage differences between the HIP path and the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from coponerf_amd import CoPoNeRF, synthetic as syn
from oracle import render_ref as orc
from tests.helpers import load_case, case_inputs, to_device

dev = torch.device("cuda:0")
w = syn.make_render_weights()
model = CoPoNeRF.CoPoNeRF(n_view=2); model.load_state_dict(w, strict=False); model = model.to(dev).eval()

def ulp(a, b):
    ai = a.contiguous().view(torch.int32).long(); bi = b.contiguous().view(torch.int32).long()
    return (ai - bi).abs()

for name in sys.argv[1:] or ["c1_val", "train_b2", "wide_val", "hd_val"]:
    cfg, gold = load_case(name)
    inp, z, rel, flow = case_inputs(cfg)
    B, R, S = cfg["B"], cfg["R"], cfg["S"]
    with torch.no_grad():
        ref = orc.forward(inp, z, rel, flow, cfg["val"], w, npoints=S, keep=True)
        model.npoints = S
        out = model(to_device(inp, dev), z=to_device(z, dev), rel_pose=rel.to(dev), val=cfg["val"], flow=to_device(flow, dev), debug=True)
    core = out["_core"]
    print("==", name, cfg)
    def rep(tag, a, b):
        a = a.float().cpu(); b = b.float().cpu()
        d = (a - b).abs()
        u = ulp(a, b)
        print(f"  {tag:12s} maxabs {d.max().item():.3e}  mismatched {float((u>0).float().mean()):.4%}  max ulp {u.max().item()}")
    rep("Tq", core["Tq"], ref["Tq"])
    rep("coords", core["coords"], ref["coords"])
    rep("pixel_val", core["pixel_val"], ref["pixel_val"])
    rep("pt", core["pt"], ref["pt"])
    sec = core["sec_grid"].cpu().view(B, 2, R, S, 2); rsec = ref["sec_grid"].view(B, 2, R, S, 2)
    rep("sec_grid", torch.stack([sec[:, 1], sec[:, 0]], 1), rsec)
    rep("at_wt", core["at_wt"], ref["at_wt"])
    rep("z_local", core["z_local"], ref["z_local"].reshape(-1, 416))
    rep("rgb_raw", core["rgb_raw"], ref["rgb_raw"].reshape(-1, 3))
    rep("rgb", out["rgb"], ref["rgb"])
    rep("rgb~gold", out["rgb"], torch.from_numpy(gold["rgb"]))
    rep("depth_ray", out["depth_ray"], ref["depth_ray"])
