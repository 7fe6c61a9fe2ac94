"""Where the ~1e-2 gradient error upstream of `z` comes from - a CPU experiment on the fp32 oracle alone.

Autograd through oracle/render_ref.py twice from the same inputs: once as is, once with some INPUTS of the first encoder layer
rounded to fp16 and everything else - forward and backward - left in fp32.  If the gradients with respect to the latent maps
`z` move by as much as the HIP path's differ from the reference's (1e-2 relative L2), the error is a property of where the
gradient is evaluated (a fraction of the first layer's 832 ReLU masks per row flips when its pre-activation moves by a few
1e-4), not of the backward pass's arithmetic: no fp32 island in the backward can remove it.

    python tools/grad_floor_probe.py [--rays 256 1024]

Measured in the build container (B = 2, 64 x 64 maps, S = 32, relative L2 of dL/dz per pyramid level):
    rays   rounded                 z0       z1       z2       z3
     256   z                    see profiles/r06_grad_floor.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coponerf_amd import synthetic as syn      # noqa: E402
from oracle import render_ref as orc           # noqa: E402  (test infrastructure: this probe is a measurement, not the product)


def run(inp, z, rel, flow, weights, coef, S, round_z, round_w):
    w = {k: v.clone() for k, v in weights.items()}
    zz = [t.clone() for t in z]
    if round_w:
        for k in w:
            if k.startswith("query_encode_latent."):
                w[k] = w[k].half().float()
    if round_z:
        zz = [t.half().float() for t in zz]
    w = {k: v.requires_grad_(True) for k, v in w.items()}
    zz = [t.requires_grad_(True) for t in zz]
    out = orc.forward(inp, zz, rel, flow, False, w, npoints=S)
    (out["rgb"] * coef).sum().backward()
    return out["rgb"].detach(), [t.grad for t in zz], {k: v.grad for k, v in w.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, nargs="+", default=[256, 1024])
    ap.add_argument("--hip", action="store_true",
                    help="on an MI355X: also the HIP training path's gradients against the exact oracle AND against the oracle "
                         "evaluated at the fp16-rounded inputs (what is left then is the backward pass's own arithmetic)")
    a = ap.parse_args()
    B, H, S = 2, 64, 32
    weights = syn.make_render_weights(seed=17)
    z, rel, flow = syn.make_latents(B, H, H, seed=52)
    print("relative L2 change of the fp32 oracle's own gradients when inputs of the first layer are rounded to fp16")
    print(f"{'rays':>6s} {'rounded':>22s} {'rgb max':>9s} " + " ".join(f"{'dz%d' % i:>9s}" for i in range(4)) +
          f" {'dW first':>9s} {'dW key':>9s} {'dW value':>9s}")
    for R in a.rays:
        inp = syn.make_inputs(B, H, H, R, seed=51)
        coef = syn.normal((B, 1, R, 3), seed=53)
        base = run(inp, z, rel, flow, weights, coef, S, False, False)
        for label, rz, rw in (("z", True, False), ("first-layer weights", False, True), ("both", True, True)):
            r = run(inp, z, rel, flow, weights, coef, S, rz, rw)
            rel_l2 = lambda a_, b_: float((a_ - b_).norm() / b_.norm())
            print(f"{R:6d} {label:>22s} {float((r[0] - base[0]).abs().max()):9.2e} " +
                  " ".join(f"{rel_l2(r[1][i], base[1][i]):9.2e}" for i in range(4)) +
                  "".join(f" {rel_l2(r[2][k], base[2][k]):9.2e}" for k in ("query_encode_latent.weight", "key_map.weight",
                                                                            "latent_value.weight")))
            if label == "both":
                rounded = r
        if a.hip:
            from coponerf_amd import CoPoNeRF
            dev = torch.device("cuda:0")
            from tests.helpers import to_device
            mv = lambda o: to_device(o, dev)
            model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
            model.load_state_dict(weights, strict=False)
            model = model.to(dev).train()
            zh = [t.to(dev).requires_grad_(True) for t in z]
            out = model(mv(inp), z=zh, rel_pose=rel.to(dev), val=False, flow=mv(flow))
            (out["rgb"] * coef.to(dev)).sum().backward()
            P = dict(model.named_parameters())
            for label, ref in (("HIP vs exact oracle", base), ("HIP vs rounded oracle", rounded)):
                print(f"{R:6d} {label:>22s} {float((out['rgb'].detach().cpu() - ref[0]).abs().max()):9.2e} " +
                      " ".join(f"{rel_l2(zh[i].grad.cpu(), ref[1][i]):9.2e}" for i in range(4)) +
                      "".join(f" {rel_l2(P[k].grad.cpu(), ref[2][k]):9.2e}" for k in ("query_encode_latent.weight", "key_map.weight",
                                                                                         "latent_value.weight")))


if __name__ == "__main__":
    main()
