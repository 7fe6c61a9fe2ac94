#!/usr/bin/env python
"""Generate tests/golden/getz.npz and ufc_ops.npz from the upstream model (build container only; see make_golden.py).

  getz.npz     CoPoNeRF.get_z on one 256x256 synthetic pair with deterministic weights for all 744 state_dict entries:
               rel_pose, the 4 flows, strided samples + statistics of the 4 latent maps.
  ufc_ops.npz  Encoder4D (Conv4d + GroupNorm + ReLU) for the three (kernel, stride, padding) variants UFC uses,
               aggregation.correlation, aggregation.soft_argmax (both directions), LinearAttention — on small inputs;
               plus the upstream gradients of the three Encoder4D cases (input and every parameter).
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
from coponerf_amd import CoPoNeRF as prod  # noqa: E402


def main():
    ref_shim.install()
    import io, contextlib
    from models import CoPoNeRF as ref_mod
    from models.conv4d import Encoder4D
    from models import aggregation as agg

    # ------------------------------------------------------------------ full get_z
    shapes = {k: tuple(v.shape) for k, v in prod.CoPoNeRF(n_view=2).state_dict().items()}
    weights = syn.make_full_weights(shapes)
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_mod.CoPoNeRF(n_view=2)
    missing, unexpected = model.load_state_dict(weights, strict=True)
    model.eval()
    inp = syn.make_inputs(1, 256, 256, 64, seed=41)
    with torch.no_grad():
        z, rel_pose, flows = model.get_z(inp)
    blob = {"rel_pose": rel_pose.numpy()}
    for i, f in enumerate(flows):
        blob[f"flow{i}"] = f.numpy()
    strides = [(4, 2), (8, 4), (16, 8), (8, 16)]            # (channel stride, spatial stride) per level
    for i, t in enumerate(z):
        cs, ss = strides[i]
        blob[f"z{i}_sample"] = t[:, ::cs, ::ss, ::ss].numpy()
        blob[f"z{i}_stats"] = np.array([float(t.mean()), float(t.std()), float(t.abs().max())], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "getz.npz"), **blob)
    print("getz", {k: v.shape for k, v in blob.items()}, "rel_pose", rel_pose[0, :3].tolist())

    # ------------------------------------------------------------------ operator fixtures
    ops = {}
    for tag, (cin, mid, k, s, p, n) in {"k3s1": (3, 5, 3, 1, 1, 6), "k3s2": (1, 8, 3, 2, 1, 10), "k5s4": (1, 8, 5, 4, 2, 16)}.items():
        enc = Encoder4D(corr_levels=(cin, mid), kernel_size=((k,) * 4,), stride=((s,) * 4,), padding=((p,) * 4,), group=(1,))
        shp = {kk: tuple(v.shape) for kk, v in enc.state_dict().items()}
        enc.load_state_dict(syn.make_full_weights(shp, seed=70 + s))
        x = syn.normal((2, cin, n, n, n, n), seed=80 + s)
        with torch.no_grad():
            y = enc(x)
        ops[f"enc4d_{tag}"] = y.numpy()
        # the upstream module's OWN gradients (autograd through Conv4d / MaxPool4d / GroupNorm / ReLU) for the same case
        xg = x.clone().requires_grad_(True)
        yg = enc(xg)
        (yg * syn.normal(tuple(yg.shape), seed=85 + s)).sum().backward()
        ops[f"enc4d_{tag}_dx"] = xg.grad.numpy()
        for name, prm in enc.named_parameters():
            ops[f"enc4d_{tag}_d.{name}"] = prm.grad.numpy()
    a = syn.normal((2, 36, 24), seed=90)
    b = syn.normal((2, 36, 24), seed=91)
    to_map = lambda t: t.transpose(1, 2).reshape(2, 24, 6, 6)
    ops["correlation"] = agg.correlation(to_map(a), to_map(b)).numpy()
    c = syn.normal((2, 1, 6, 6, 6, 6), seed=92) * 0.2
    gx, gy = agg.soft_argmax(c.permute(0, 1, 4, 5, 2, 3).flatten(1, 3))
    ops["t_to_s"] = torch.cat((gx, gy), 1).numpy()
    gx, gy = agg.soft_argmax(c.flatten(1, 3))
    ops["s_to_t"] = torch.cat((gx, gy), 1).numpy()
    q, k_, v = syn.normal((2, 30, 4, 8), 93), syn.normal((2, 30, 4, 8), 94), syn.normal((2, 30, 4, 12), 95)
    ops["linear_attention"] = agg.LinearAttention()(q, k_, v).numpy()
    np.savez_compressed(os.path.join(HERE, "ufc_ops.npz"), **ops)
    print("ufc_ops", {k: v.shape for k, v in ops.items()})


if __name__ == "__main__":
    main()
