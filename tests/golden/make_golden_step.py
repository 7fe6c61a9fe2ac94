#!/usr/bin/env python
"""Gradient fixture of the WHOLE training step (get_z + render + loss + backward, /root/reference wrapper.py:104-138), from the
upstream reference imported read-only from /root/reference.  Runs ONLY in the build container.

    python tests/golden/make_golden_step.py        # writes tests/golden/step.npz  (about a minute on 8 cores)
    python tests/golden/make_golden_step.py --rays 4096 --tags img,aux --out step_r4096.npz
                                                   # the same step at the training ray count of BASELINE configs[2] per pair
                                                   # (one pair: four would not fit this container's 62 GB); ~10 minutes

Case: one 256x256 synthetic pair, 256 query rays, 64 samples, val=False, the module in the mode the reference trains it in
(never switched to eval: batch statistics in the trunk's BatchNorm), deterministic values for all 744 state_dict
entries (the ones of getz.npz).  Three losses, each differentiated from its own forward pass:

  img    image_loss alone                              = |gt - rgb|.mean()            (loss_function.py:63-69)
  full   image_loss + cycle_loss + pose_loss, summed as wrapper.py:109-123 does (each `.mean()`ed, weight 1): the cycle
         term differentiates through the auxiliary outputs T_to_C1_pts / C2_pts_to_C1 (CoPoNeRF.py:493-541), the pose
         term through rel_pose (get_z's pose head).  The terms are computed by the reference's own LFLoss.__call__
         (the object is created without running __init__, whose SSIM window needs a CUDA tensor type).  On synthetic
         weights the flows are not cycle-consistent, so the reference's validity masks switch the cycle term OFF
         (cycle_loss == 0, recorded): this loss pins the pose path, not the auxiliary outputs.
  aux    image_loss + 0.01 * huber_loss(T_to_C1_pts, C2_pts_to_C1).mean() + 0.1 * depth_ray.mean(): the cycle term's
         own integrand (loss_function.py:112-120, utils.py:604-605) WITHOUT the validity masks, composed in this script
         from the reference's outputs, so that gradients do run through the attention-weighted expected point, its two
         reprojections, the clamped depth and the flow look-up (CoPoNeRF.py:493-541, utils.py:52-69) into `at_wt`, the
         render weights, `z` and the flow head of get_z.

Stored per loss and per parameter that received a gradient: L2 norm, max |g| and a strided sample (<= 331 entries) of
the gradient; the list of parameters WITHOUT a gradient; the loss terms; rgb and rel_pose of the forward pass.  Data only.
"""
import contextlib
import io
import os
import sys
import time
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
warnings.filterwarnings("ignore")

import ref_shim  # noqa: E402
from coponerf_amd import synthetic as syn  # noqa: E402
from coponerf_amd import CoPoNeRF as prod  # noqa: E402

CFG = dict(B=1, H=256, R=256, S=64, iseed=71, nsample=331)


def sample_stride(numel: int) -> int:
    return max(1, numel // CFG["nsample"])


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=CFG["R"])
    ap.add_argument("--tags", default="img,full,aux")
    ap.add_argument("--out", default="step.npz")
    args = ap.parse_args()
    CFG["R"] = args.rays
    ref_shim.install()
    for name in ("lietorch", "lpips"):                     # imported at module level by loss_function.py, unused by these terms
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["lietorch"].SE3 = None
    from models import CoPoNeRF as ref_mod
    from models import loss_function as lf

    shapes = {k: tuple(v.shape) for k, v in prod.CoPoNeRF(n_view=2).state_dict().items()}
    weights = syn.make_full_weights(shapes)
    c = CFG
    rec = {"nsample": np.int64(c["nsample"]), "rays": np.int64(c["R"])}
    for tag in args.tags.split(","):
        with contextlib.redirect_stdout(io.StringIO()):
            model = ref_mod.CoPoNeRF(n_view=2, npoints=c["S"])
        model.load_state_dict(weights, strict=True)
        assert model.training
        inp = syn.make_inputs(c["B"], c["H"], c["H"], c["R"], seed=c["iseed"])
        gt = {"rgb": inp["query"]["rgb"].clone()}
        t0 = time.time()
        out = model(inp, val=False)
        loss_fn = object.__new__(lf.LFLoss)
        loss_fn.depth, loss_fn.ssim = False, False
        loss_fn.cycle = loss_fn.pose = (tag == "full")
        loss_fn.w1, loss_fn.w2, loss_fn.w3 = 0.01, 1.0, 1.0
        losses, _ = loss_fn(inp, out, gt, ITER=0, model=model)
        total = 0.0
        if tag == "aux":
            from utils_training.utils import huber_loss
            losses["cycle_unmasked"] = 0.01 * huber_loss(out["T_to_C1_pts"], out["C2_pts_to_C1"]).mean()
            losses["depth_mean"] = 0.1 * out["depth_ray"].mean()
        for lname, l in losses.items():
            rec[f"{tag}|loss|{lname}"] = np.float64(l.mean().item())
            total = total + l.mean()
        t1 = time.time()
        total.backward()
        t2 = time.time()
        print(tag, {k: float(v.mean()) for k, v in losses.items()}, "forward %.1f s, backward %.1f s" % (t1 - t0, t2 - t1))
        rec[f"{tag}|rgb"] = out["rgb"].detach().numpy().astype(np.float32)
        rec[f"{tag}|rel_pose"] = out["rel_pose"].detach().numpy().astype(np.float32)
        none = []
        for name, p in model.named_parameters():
            if p.grad is None:
                none.append(name)
                continue
            flat = p.grad.detach().reshape(-1)
            rec[f"{tag}|{name}|norm"] = np.float64(flat.double().norm().item())
            rec[f"{tag}|{name}|max"] = np.float32(flat.abs().max().item())
            rec[f"{tag}|{name}|sample"] = flat[:: sample_stride(flat.numel())].numpy().astype(np.float32)
        rec[f"{tag}|none"] = np.array(none)
        print(tag, "parameters without gradient:", len(none))
    path = os.path.join(HERE, args.out)
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
