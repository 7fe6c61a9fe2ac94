"""Dynamic range of the fp16 parts of the render path (node features and node tables are rounded to fp16, csrc/encode.hip;
hid, the key path's hidden layer and the attention-weighted hidden sums are fp16): the HIP render against the fp32 oracle
with the latent maps scaled by {1, 4, 16, 64} and the first layer's weights by {1, 4}.  The envelope inside which
`north_star`'s 1e-3 bar on rgb holds is asserted; every point of the sweep is printed (DESIGN.md §2 records it)."""
import pytest
import torch

from coponerf_amd import synthetic as syn
from tests.helpers import to_device

pytestmark = pytest.mark.gpu

Z_SCALES = (1.0, 4.0, 16.0, 64.0)
W_SCALES = (1.0, 4.0)
# Measured on MI355X (round 4): the error is RELATIVE - 3.2e-4 ... 4.8e-4 of max(1, |rgb|max) at every point of the sweep,
# z_local to 2.8e-4 ... 3.4e-4 of its scale, at_wt <= 1.6e-4, nothing saturates up to z x 64 and W1 x 4 (|z_local| = 262).
# `north_star`'s absolute 1e-3 therefore holds while |rgb| <= ~2.5, i.e. at the three points of ABS_ENVELOPE (|rgb|max
# 1.0 / 1.15 / 1.14; at z x 16 the output itself is 4x larger and the absolute error 1.3e-3); the relative bar everywhere.
ABS_ENVELOPE = {(1.0, 1.0), (4.0, 1.0), (1.0, 4.0)}


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def test_rgb_error_over_latent_and_weight_scale(dev):
    from coponerf_amd import CoPoNeRF
    from oracle import render_ref as orc
    B, H, R, S = 1, 64, 192, 32
    inp = syn.make_inputs(B, H, H, R, seed=23)
    z0, rel, flow = syn.make_latents(B, H, H, seed=24)
    rows, bad = [], []
    for ws in W_SCALES:
        weights = syn.make_render_weights(seed=7)
        weights["query_encode_latent.weight"] = weights["query_encode_latent.weight"] * ws
        model = CoPoNeRF.CoPoNeRF(n_view=2, npoints=S)
        model.load_state_dict(weights, strict=False)
        model = model.to(dev).eval()
        for zs in Z_SCALES:
            z = [t * zs for t in z0]
            with torch.no_grad():
                ref = orc.forward(inp, z, rel, flow, True, weights, npoints=S, keep=True)
                out = model(to_device(inp, dev), z=to_device(z, dev), rel_pose=rel.to(dev), val=True, flow=to_device(flow, dev),
                            debug=True)
            rgb, want = out["rgb"].cpu(), ref["rgb"]
            finite = bool(torch.isfinite(rgb).all())
            scale = max(1.0, float(want.abs().max()))
            err = float((rgb - want).abs().max()) if finite else float("inf")
            e_wt = float((out["at_wt"].cpu() - ref["at_wt"]).abs().max())
            e_zl = float((out["_core"]["z_local"].cpu() - ref["z_local"].reshape(-1, 416)).abs().max())
            zl_scale = float(ref["z_local"].abs().max())
            rows.append((zs, ws, err, err / scale, scale, e_wt, e_zl / max(zl_scale, 1e-30), zl_scale, finite))
            if not (finite and err <= 1e-3 * scale) or ((zs, ws) in ABS_ENVELOPE and err > 1e-3) or e_wt > 2e-3:
                bad.append(rows[-1])
    print("z scale  W1 scale  rgb max-abs   / max(1,|rgb|)  |rgb|max    at_wt err   z_local rel   |z_local|max  finite")
    for r in rows:
        print("%7.0f  %8.0f  %11.3e  %13.3e  %9.3e  %10.3e  %11.3e  %11.3e  %s" % r)
    assert not bad, bad
