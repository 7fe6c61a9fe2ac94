#!/usr/bin/env python
"""Gradient fixture of the training path, from the upstream reference imported read-only from /root/reference
(SURVEY.md §8(c) G9).  Runs ONLY in the build container.

    python tests/golden/make_golden_grads.py        # writes tests/golden/grads.npz

Case: B=2, H=64, R=80, S=32, val=False, narrow rig (the case of tests/test_gpu_train.py).  Loss =
sum(rgb * coef) + sum(at_wt * cw) with counter-hash coefficients.  Stored: for each render parameter and each feature
map the gradient's L2 norm, max |g| and a strided sample (every 61st element) - data only.
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

CFG = dict(B=2, H=64, R=80, S=32, wseed=17, iseed=51, zseed=52, cseed=53, wtseed=54, stride=61)


def main():
    c = CFG
    weights = syn.make_render_weights(seed=c["wseed"])
    model = ref_shim.build_reference_model(weights, npoints=c["S"], H=c["H"])
    model.train()
    inp = syn.make_inputs(c["B"], c["H"], c["H"], c["R"], seed=c["iseed"])
    z, rel, flow = syn.make_latents(c["B"], c["H"], c["H"], seed=c["zseed"])
    coef = syn.normal((c["B"], 1, c["R"], 3), seed=c["cseed"])
    cw = syn.normal((2 * c["B"], c["R"], c["S"]), seed=c["wtseed"]) * 0.3
    z = [t.clone().requires_grad_(True) for t in z]
    out = model(inp, z=z, rel_pose=rel, val=False, flow=flow)
    loss = (out["rgb"] * coef).sum() + (out["at_wt"] * cw).sum()
    loss.backward()
    rec = {"loss": np.float64(loss.item()), "stride": np.int64(c["stride"])}
    params = dict(model.named_parameters())
    for name in weights:
        g = params[name].grad
        assert g is not None, name
        flat = g.detach().reshape(-1)
        rec[f"{name}|norm"] = np.float64(flat.double().norm().item())
        rec[f"{name}|max"] = np.float32(flat.abs().max().item())
        rec[f"{name}|sample"] = flat[:: c["stride"]].numpy().astype(np.float32)
    for i, t in enumerate(z):
        flat = t.grad.detach().reshape(-1)
        rec[f"z{i}|norm"] = np.float64(flat.double().norm().item())
        rec[f"z{i}|max"] = np.float32(flat.abs().max().item())
        rec[f"z{i}|sample"] = flat[:: c["stride"]].numpy().astype(np.float32)
    path = os.path.join(HERE, "grads.npz")
    np.savez_compressed(path, **rec)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB, loss", loss.item())


if __name__ == "__main__":
    main()
