"""get_z: the oracle's UFC operators and the product's (stock-op) glue against fixtures of the upstream model (CPU)."""
import os

import numpy as np
import pytest
import torch

from coponerf_amd import CoPoNeRF, synthetic as syn
from coponerf_amd.getz import Encoder4D, positional_encodings
from oracle.ufc_ref import TorchOps
from tests.helpers import GOLDEN


@pytest.fixture(scope="module")
def ops_gold():
    return dict(np.load(os.path.join(GOLDEN, "ufc_ops.npz")))


def test_oracle_ufc_operators_match_reference(ops_gold):
    for tag, (cin, mid, k, s, p, n) in {"k3s1": (3, 5, 3, 1, 1, 6), "k3s2": (1, 8, 3, 2, 1, 10), "k5s4": (1, 8, 5, 4, 2, 16)}.items():
        enc = Encoder4D((cin, mid), k, s, p)
        shp = {kk: tuple(v.shape) for kk, v in enc.state_dict().items()}
        enc.load_state_dict(syn.make_full_weights(shp, seed=70 + s))
        x = syn.normal((2, cin, n, n, n, n), seed=80 + s)
        with torch.no_grad():
            y = enc(x, TorchOps)
        assert (y - torch.from_numpy(ops_gold[f"enc4d_{tag}"])).abs().max() <= 2e-5, tag
    a, b = syn.normal((2, 36, 24), seed=90), syn.normal((2, 36, 24), seed=91)
    corr = TorchOps.correlation_tokens(a, b, 6)
    assert (corr[:, 0] - torch.from_numpy(ops_gold["correlation"])).abs().max() <= 1e-6
    c = syn.normal((2, 1, 6, 6, 6, 6), seed=92) * 0.2
    t2s, s2t = TorchOps.soft_argmax_pair(c)
    assert (t2s - torch.from_numpy(ops_gold["t_to_s"])).abs().max() <= 1e-5
    assert (s2t - torch.from_numpy(ops_gold["s_to_t"])).abs().max() <= 1e-5
    q, k_, v = syn.normal((2, 30, 4, 8), 93), syn.normal((2, 30, 4, 8), 94), syn.normal((2, 30, 4, 12), 95)
    want = torch.from_numpy(ops_gold["linear_attention"])                        # upstream LinearAttention.forward
    assert (TorchOps.linear_attention(q, k_, v) - want).abs().max() <= 1e-5
    cm = TorchOps.linear_attention(q, k_, v.permute(0, 2, 3, 1).contiguous(), channel_major=True)   # (B,H,Dv,L) layout
    assert (cm.permute(0, 3, 1, 2) - want).abs().max() <= 1e-5


def test_positional_encodings_closed_form():
    """The vectorised table equals the reference's 4096-iteration loop formula (backbone.py:209-278) on a few entries."""
    fx, fy, cx, cy = (torch.tensor([[v]]) for v in (0.8, 0.8, 0.5, 0.5))
    pos = positional_encodings(fx, fy, cx, cy, n=64)
    assert pos.shape == (1, 4096, 6)
    lin = torch.linspace(-1, 1, 64)
    K = torch.tensor([[[1.6, 0.0, 0.0], [0.0, 1.6, 0.0], [0.0, 0.0, 1.0]]])
    for (j, k) in [(0, 0), (5, 17), (63, 2), (31, 63)]:
        w = torch.inverse(K) @ torch.tensor([lin[k], lin[j], 1.0])
        p3, p4 = (w[:, 1] / w[:, 2]).item(), (w[:, 0] / w[:, 2]).item()
        got = pos[0, k * 64 + j]
        assert torch.allclose(got, torch.tensor([p3 * p3, p4 * p4, p3 * p4, p3, p4, 1.0]), atol=1e-6)


def test_get_z_glue_matches_reference():
    """Product get_z with the oracle's CPU operators plugged in == upstream get_z (same 744-entry state_dict)."""
    gold = dict(np.load(os.path.join(GOLDEN, "getz.npz")))
    model = CoPoNeRF.CoPoNeRF(n_view=2)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(syn.make_full_weights(shapes), strict=True)
    model.eval()
    inp = syn.make_inputs(1, 256, 256, 64, seed=41)
    with torch.no_grad():
        z, rel_pose, flows = model.get_z(inp, ops=TorchOps)
    assert model.H == 256 and model.W == 256
    assert [tuple(t.shape) for t in z] == [(2, 256, 16, 16), (2, 256, 32, 32), (2, 256, 64, 64), (2, 64, 256, 256)]
    strides = [(4, 2), (8, 4), (16, 8), (8, 16)]
    for i, t in enumerate(z):
        cs, ss = strides[i]
        g = torch.from_numpy(gold[f"z{i}_sample"])
        assert (t[:, ::cs, ::ss, ::ss] - g).abs().max() <= 2e-3 * max(1.0, float(g.abs().max())), i
        assert abs(float(t.mean()) - gold[f"z{i}_stats"][0]) <= 1e-3
    for i, f in enumerate(flows):
        g = torch.from_numpy(gold[f"flow{i}"])
        assert f.shape == g.shape == (1, 2, 64, 64)
        assert (f - g).abs().max() <= (5e-2 if i < 2 else 2e-3), i          # pixel units / normalised units
    assert (rel_pose - torch.from_numpy(gold["rel_pose"])).abs().max() <= 2e-3


def _enc4d_grads(tag, cin, mid, k, s, p, n, ops, dev="cpu"):
    """Input / parameter gradients of coponerf_amd.getz.Encoder4D on the fixture case `tag` with operator set `ops`."""
    enc = Encoder4D((cin, mid), k, s, p)
    shp = {kk: tuple(v.shape) for kk, v in enc.state_dict().items()}
    enc.load_state_dict(syn.make_full_weights(shp, seed=70 + s))
    enc = enc.to(dev)
    x = syn.normal((2, cin, n, n, n, n), seed=80 + s).to(dev).requires_grad_(True)
    y = enc(x, ops)
    (y * syn.normal(tuple(y.shape), seed=85 + s).to(dev)).sum().backward()
    out = {"dx": x.grad.cpu()}
    out.update({"d." + name: prm.grad.cpu() for name, prm in enc.named_parameters()})
    return out


ENC4D_CASES = {"k3s1": (3, 5, 3, 1, 1, 6), "k3s2": (1, 8, 3, 2, 1, 10), "k5s4": (1, 8, 5, 4, 2, 16)}


def test_oracle_encoder4d_gradients_match_upstream(ops_gold):
    """Autograd through the oracle's Conv4d / pooling / GroupNorm / ReLU restatement == the upstream module's own
    gradients (tests/golden/ufc_ops.npz: input and every parameter, all three (kernel, stride, padding) variants)."""
    for tag, cfg in ENC4D_CASES.items():
        got = _enc4d_grads(tag, *cfg, ops=TorchOps)
        for name, g in got.items():
            want = torch.from_numpy(ops_gold[f"enc4d_{tag}_{name}"])
            assert (g - want).abs().max() <= 2e-5 * max(1.0, float(want.abs().max())), (tag, name)


def test_cost_volume_attention_restructuring_is_exact():
    """ufc_ops.cost_volume_attention_torch — (D diag(Z) phi(q)) . ((U^T phi(k))^T v_low), what the HIP path evaluates —
    against the reference's order of operations (oracle TorchOps: interpolate up, LinearAttention, interpolate down),
    values and gradients, on the CPU."""
    import torch.nn.functional as F
    from coponerf_amd.ufc_ops import cost_volume_attention_torch
    from oracle.ufc_ref import TorchOps

    class Shim:
        @staticmethod
        def resize_bilinear(x, size):
            return F.interpolate(x, size=(size, size), mode="bilinear", align_corners=True)

        @staticmethod
        def resize_bilinear_adjoint(x, size):
            N, C = x.shape[:2]
            return torch.ops.aten.upsample_bilinear2d_backward(x.contiguous(), list(x.shape[-2:]), [N, C, size, size], True, None, None)

    for fs, hs in ((8, 4), (4, 4), (12, 5)):
        B, H, D, ht = 2, 2, 32, 3
        g = torch.Generator().manual_seed(fs * 10 + hs)
        q = (torch.randn(B, fs * fs, H, D, generator=g) * 0.7).double().requires_grad_(True)
        k = (torch.randn(B, fs * fs, H, D, generator=g) * 0.7).double().requires_grad_(True)
        v = torch.randn(B, H, hs, hs, ht, ht, generator=g).double().requires_grad_(True)
        r = torch.randn(B, H, hs, hs, ht, ht, generator=g).double()
        want = TorchOps.cost_volume_attention(q, k, v, fs, residual=r)
        got = cost_volume_attention_torch(Shim, q, k, v, fs, residual=r)
        assert got.shape == want.shape
        assert (got - want).abs().max() <= 1e-12 * max(1.0, float(want.abs().max())), (fs, hs)
        w = torch.randn(want.shape, generator=g).double()
        gw = torch.autograd.grad((want * w).sum(), (q, k, v))
        gg = torch.autograd.grad((got * w).sum(), (q, k, v))
        for a, b_ in zip(gg, gw):
            assert (a - b_).abs().max() <= 1e-11 * max(1.0, float(b_.abs().max())), (fs, hs)
