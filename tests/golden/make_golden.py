#!/usr/bin/env python
"""Generate tests/golden/*.npz from the upstream reference, imported read-only from /root/reference.

Runs ONLY in the build container (the GPU box has no /root/reference).  What is
stored is data: outputs of the reference on inputs that both sides regenerate
deterministically from coponerf_amd.synthetic (seeded counter hash), plus tiny
hand-built edge-case inputs.  No reference source text is stored.

    python tests/golden/make_golden.py            # rewrites every fixture

Cases (names match tests/test_oracle_golden.py and tests/test_gpu_parity.py):
  c1_val      B=1 H=64  R=512 S=32 val=True  narrow rig   (BASELINE config 1)
  train_b2    B=2 H=64  R=256 S=32 val=False narrow rig
  wide_val    B=1 H=64  R=256 S=32 val=True  wide rig     (ACID-like, config 4 geometry)
  hd_val      B=1 H=256 R=256 S=64 val=True  narrow rig   (config 2 geometry/sample count)
  peaked_val  B=1 H=64  R=256 S=32 val=True  narrow rig, attention sharpened (key_map_2 / query_embed_2 / query_repeat_embed_2
              x 64: the largest softmax weight of a ray > 0.5 on most rays) and latents at get_z's output statistics
              (synthetic.peaked_weights, latents_at_getz_statistics; `--only peaked_val` writes this fixture alone)
  inter       B=1 H=64  R=6   S=32 val=True  wide rig, with intermediates of every stage
  edges       project_rays on hand-built degenerate rays
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

import ref_shim  # noqa: E402
from coponerf_amd import synthetic as syn  # noqa: E402

OUT_KEYS = ["rgb", "pixel_val", "at_wt", "valid_mask", "depth_ray", "coords", "T_to_C1_pts", "T_to_C2_pts",
            "C2_pts_to_C1", "mask_c2", "matchability_cycle_mask", "at_wt_max", "rel_pose_flip", "gt_rel_pose",
            "gt_rel_pose_flip"]

CASES = {
    "c1_val": dict(B=1, H=64, R=512, S=32, val=True, rig="narrow", seed=0),
    "train_b2": dict(B=2, H=64, R=256, S=32, val=False, rig="narrow", seed=3),
    "wide_val": dict(B=1, H=64, R=256, S=32, val=True, rig="wide", seed=5),
    "hd_val": dict(B=1, H=256, R=256, S=64, val=True, rig="narrow", seed=9),
    "peaked_val": dict(B=1, H=64, R=256, S=32, val=True, rig="narrow", seed=21, peak=64.0, zstats=True),
}


def run_case(cfg, weights, capture=False):
    if cfg.get("peak"):
        weights = syn.peaked_weights(weights, cfg["peak"])
    model = ref_shim.build_reference_model(weights, npoints=cfg["S"], H=cfg["H"])
    inp = syn.make_inputs(cfg["B"], cfg["H"], cfg["H"], cfg["R"], seed=cfg["seed"], rig=cfg["rig"])
    z, rel, flow = syn.make_latents(cfg["B"], cfg["H"], cfg["H"], seed=cfg["seed"] + 1)
    if cfg.get("zstats"):
        z = syn.latents_at_getz_statistics(z)
    rec = {}
    hooks = []
    if capture:
        import torch.nn.functional as F
        from utils_training import geometry
        gs_calls, orig_gs = [], F.grid_sample
        orig_pt = geometry.get_3d_point_epipolar

        def gs(*a, **k):
            o = orig_gs(*a, **k)
            gs_calls.append((k.get("padding_mode", "zeros"), o))
            return o

        def pt_fn(*a, **k):
            o = orig_pt(*a, **k)
            rec["pt"] = o[0].clone()
            return o

        F.grid_sample, geometry.get_3d_point_epipolar = gs, pt_fn
        for name in ["query_encode_latent", "query_encode_latent_2", "latent_value", "key_map_2", "query_embed_2",
                     "query_repeat_embed_2", "encode_latent", "phi"]:
            mod = dict(model.named_modules())[name]
            hooks.append(mod.register_forward_hook(
                lambda m, i, o, name=name: rec.setdefault(name, []).append(o.detach().clone())))
    with torch.no_grad():
        out = model(inp, z=z, rel_pose=rel, val=cfg["val"], flow=flow)
    if capture:
        F.grid_sample, geometry.get_3d_point_epipolar = orig_gs, orig_pt
        for h in hooks:
            h.remove()
        # grid_sample call order inside forward(): 2 (warp) | 4 border (primary) | 1 + 4 (dead flow path) | 4 zeros (secondary)
        border = [o for (p, o) in gs_calls if p == "border"]
        zeros4 = [o for (p, o) in gs_calls if p == "zeros"][-4:]
        rec["prim"] = torch.cat(border[:4], 1).permute(0, 2, 3, 1)          # (N,R,S,832)
        rec["sec"] = torch.cat(zeros4, 1).permute(0, 2, 3, 1)
    return out, rec


def main():
    weights = syn.make_render_weights()
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for name, cfg in CASES.items():
        if only and name != only:
            continue
        out, _ = run_case(cfg, weights)
        blob = {k: out[k].detach().cpu().numpy() for k in OUT_KEYS}
        blob["cfg"] = np.array(repr(cfg))
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **blob)
        print(name, {k: v.shape for k, v in blob.items() if k in ("rgb", "pixel_val")})
    if only:
        return

    cfg = dict(B=1, H=64, R=6, S=32, val=True, rig="wide", seed=11)
    out, rec = run_case(cfg, weights, capture=True)
    blob = {k: out[k].detach().cpu().numpy() for k in OUT_KEYS}
    blob["cfg"] = np.array(repr(cfg))
    N, R, S = 2, cfg["R"], cfg["S"]
    blob["pt"] = rec["pt"].numpy()
    blob["prim"] = rec["prim"].numpy().astype(np.float16)       # fp16 storage: 5e-4 rel, test tolerance says so
    blob["sec"] = rec["sec"].numpy().astype(np.float16)
    # conv outputs are (B,C,R,S); the encoder pair is called 4x: (v0 own, v0 other, v1 own, v1 other)
    enc = torch.stack(rec["query_encode_latent_2"], 0)           # (4,B,416,R,S)
    blob["enc"] = enc[:, 0].permute(0, 2, 3, 1).numpy()          # (4,R,S,416)
    blob["value"] = rec["latent_value"][0].permute(0, 2, 3, 1).numpy()        # (N,R,S,416)
    blob["key"] = rec["key_map_2"][0].permute(0, 2, 3, 1).numpy()             # (N,R,S,128)
    blob["ce"] = rec["query_embed_2"][0].permute(0, 2, 3, 1).numpy()
    blob["q2"] = rec["query_repeat_embed_2"][0].permute(0, 2, 3, 1).numpy()
    blob["ze"] = rec["encode_latent"][0].permute(0, 2, 1).numpy()             # (N,R,128)
    blob["rgb_raw"] = rec["phi"][0].numpy()
    np.savez_compressed(os.path.join(HERE, "inter.npz"), **blob)
    print("inter", {k: v.shape for k, v in blob.items() if k not in OUT_KEYS})

    # ---- edge cases of the segment clipper (epipolar.py:175-253), inputs stored with outputs
    ref_shim.install()
    from models.epipolar import project_rays
    K = torch.tensor([[0.8, 0.0, 0.5], [0.0, 0.8, 0.5], [0.0, 0.0, 1.0]])
    o = torch.tensor([
        [0.0, 0.0, 0.0],      # ray starts at the camera centre
        [0.1, 0.0, -0.5],     # origin behind the image plane, pointing forward
        [0.3, 0.2, 1.0],      # ray parallel to the image plane
        [5.0, 5.0, 1.0],      # ray that never meets the frame
        [0.05, -0.02, 0.5],   # both ends inside the frame
        [0.2, 0.0, 0.0],      # origin on the z=0 plane but not at the camera
        [0.0, 0.0, 2.0],      # pointing straight back at the camera
        [-0.7, 0.1, 0.3],     # enters through the x=0 edge
        [0.0, 0.0, 1.0],      # along the optical axis
        [1e-7, 0.0, 0.0],     # within epsilon of the camera centre
    ])
    d = torch.nn.functional.normalize(torch.tensor([
        [0.1, 0.05, 1.0], [0.0, 0.1, 1.0], [1.0, 0.0, 0.0], [1.0, 1.0, 0.0], [0.01, 0.02, 1.0],
        [0.0, 0.0, 1.0], [0.0, 0.0, -1.0], [1.0, 0.0, 0.2], [0.0, 0.0, 1.0], [0.3, -0.2, 1.0]]), dim=-1)
    extr = torch.eye(4)[None]
    res = project_rays(o[None].clone(), d[None].clone(), extr, K[None].clone())
    np.savez_compressed(os.path.join(HERE, "edges.npz"), o=o.numpy(), d=d.numpy(), K=K.numpy(),
                        **{k: v[0].numpy() for k, v in res.items()})
    print("edges", {k: v[0].tolist() for k, v in res.items() if k in ("overlaps_image",)})


if __name__ == "__main__":
    main()
