#!/usr/bin/env python
"""Loss curve of the WHOLE model under training, from the upstream reference imported read-only from /root/reference: the
reference's own step (/root/reference wrapper.py:104-151 for one rank: forward val=False with get_z inside, image loss,
backward, clip_grad_norm_(max_norm=1), Adam) repeated on one fixed 256x256 pair with EVERY parameter trainable - encoder, UFC,
pose head and render layers.  Runs ONLY in the build container (about a minute per step).

    python tests/golden/make_golden_converge.py [--steps 10] [--lr 5e-4]      # writes tests/golden/converge.npz

Stored: the loss before each update (steps + 1 values: the last one is the loss after the final update), rgb of the first and
of the last forward pass, and per parameter the L2 norm of its total displacement plus a strided sample of the end values for a
few tensors on either side of `z`.  Data only.  tests/test_gpu_converge.py trains the HIP path from the same start and holds
its curve beside this one (nothing showed the encoder / UFC converge under the fp16 render gradients).
"""
import argparse
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

CFG = dict(B=1, H=256, R=256, S=64, iseed=71)
WATCH = ("encoder.model.layer1.0.conv1.weight", "encoder.model.layer4.2.bn2.weight", "conv_map.weight",
         "feature_cost_aggregation.layers.0.0.q_proj.weight", "feature_cost_aggregation.layers.2.0.mlp_corr.conv4d.0.0.query_conv.weight",
         "feature_cost_aggregation.proj_feat.0.0.weight", "query_encode_latent.weight", "key_map.weight", "phi.lin_out.weight")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--lr", type=float, default=5e-4)
    ap.add_argument("--out", default="converge.npz")
    a = ap.parse_args()
    ref_shim.install()
    for name in ("lietorch", "lpips"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["lietorch"].SE3 = None
    from models import CoPoNeRF as ref_mod

    shapes = {k: tuple(v.shape) for k, v in prod.CoPoNeRF(n_view=2).state_dict().items()}
    weights = syn.make_full_weights(shapes)
    c = CFG
    with contextlib.redirect_stdout(io.StringIO()):
        model = ref_mod.CoPoNeRF(n_view=2, npoints=c["S"])
    model.load_state_dict(weights, strict=True)
    assert model.training
    start = {k: p.detach().clone() for k, p in model.named_parameters()}
    inp = syn.make_inputs(c["B"], c["H"], c["H"], c["R"], seed=c["iseed"])
    gt = inp["query"]["rgb"].clone()
    opt = torch.optim.Adam(model.parameters(), lr=a.lr)
    losses, rec = [], {"steps": np.int64(a.steps), "lr": np.float64(a.lr), "rays": np.int64(c["R"])}
    for it in range(a.steps + 1):
        t0 = time.time()
        out = model(inp, val=False)
        loss = (gt - out["rgb"]).abs().mean()                        # models/loss_function.py:63-69 (no NaN in this case)
        losses.append(float(loss))
        if it == 0:
            rec["rgb_first"] = out["rgb"].detach().numpy().astype(np.float32)
        if it == a.steps:
            rec["rgb_last"] = out["rgb"].detach().numpy().astype(np.float32)
            break
        opt.zero_grad(set_to_none=True)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=1.0)   # wrapper.py:142-146
        opt.step()
        print(f"step {it}: loss {losses[-1]:.6f}  ({time.time() - t0:.0f} s)", flush=True)
    rec["loss"] = np.array(losses, dtype=np.float64)
    moved = {k: float((p.detach() - start[k]).double().norm()) for k, p in model.named_parameters()}
    rec["moved_names"] = np.array(list(moved))
    rec["moved_norm"] = np.array([moved[k] for k in moved], dtype=np.float64)
    for k in WATCH:
        p = dict(model.named_parameters())[k].detach().reshape(-1)
        st = max(1, p.numel() // 331)
        rec[f"end|{k}"] = p[::st].numpy().astype(np.float32)
        rec[f"start|{k}"] = start[k].reshape(-1)[::st].numpy().astype(np.float32)
    path = os.path.join(HERE, a.out)
    np.savez_compressed(path, **rec)
    print("losses", losses)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
