"""Shared helpers for the parity tests (CPU and GPU)."""
import ast
import os

import numpy as np
import torch

from coponerf_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    blob = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
    cfg = ast.literal_eval(str(blob.pop("cfg")))
    return cfg, blob


def case_inputs(cfg):
    """Regenerate exactly the inputs tests/golden/make_golden.py fed to the reference."""
    inp = syn.make_inputs(cfg["B"], cfg["H"], cfg["H"], cfg["R"], seed=cfg["seed"], rig=cfg["rig"])
    z, rel, flow = syn.make_latents(cfg["B"], cfg["H"], cfg["H"], seed=cfg["seed"] + 1)
    if cfg.get("zstats"):
        z = syn.latents_at_getz_statistics(z)
    return inp, z, rel, flow


def case_weights(cfg, weights):
    """The weights make_golden.py gave the reference for this case (cfg["peak"]: attention sharpened, synthetic.peaked_weights)."""
    return syn.peaked_weights(weights, cfg["peak"]) if cfg.get("peak") else weights


def to_device(obj, dev):
    if torch.is_tensor(obj):
        return obj.to(dev)
    if isinstance(obj, dict):
        return {k: to_device(v, dev) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(to_device(v, dev) for v in obj)
    return obj


def tap_indices(pixel_val, H):
    """int64 floor tap indices of the 4 feature levels for border-mode sampling."""
    from oracle.render_ref import bilinear_taps
    out = []
    for s in (H // 16, H // 8, H // 4, H):
        ix, iy, _, _ = bilinear_taps(pixel_val, s, s, border=True)
        out.append(torch.stack((ix, iy), -1))
    return out
