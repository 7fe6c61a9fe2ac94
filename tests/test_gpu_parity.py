"""Parity of the HIP render path with the CPU oracle — runs on a real MI355X (`pytest -m gpu`).

Every compute call goes through the C ABI of coponerf_amd/libcoponerf_hip.so; the oracle
(oracle/render_ref.py) is only the checker.  Bars (BASELINE.json north_star):
  * sample coordinates / tap indices: bit-exact
  * rendered RGB: within 1e-3 abs
"""
import numpy as np
import pytest
import torch

from coponerf_amd import synthetic as syn
from tests.helpers import load_case, case_inputs, to_device, tap_indices

pytestmark = pytest.mark.gpu

RGB_TOL = 1e-3          # north_star: "within 1e-3 abs on rendered RGB"


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def weights():
    return syn.make_render_weights()


@pytest.fixture(scope="module")
def model(dev, weights):
    from coponerf_amd import CoPoNeRF
    m = CoPoNeRF.CoPoNeRF(n_view=2)
    missing, unexpected = m.load_state_dict(weights, strict=False)
    assert not unexpected
    return m.to(dev).eval()


def run_pair(model, dev, weights, cfg):
    from oracle import render_ref as orc
    inp, z, rel, flow = case_inputs(cfg)
    with torch.no_grad():
        ref = orc.forward(inp, z, rel, flow, cfg["val"], weights, npoints=cfg["S"], keep=True)
        model.npoints = cfg["S"]
        out = model(to_device(inp, dev), z=to_device(z, dev), rel_pose=rel.to(dev), val=cfg["val"],
                    flow=to_device(flow, dev), debug=True)
    return ref, out


@pytest.mark.parametrize("name", ["c1_val", "train_b2", "wide_val", "hd_val"])
def test_forward_parity(name, model, dev, weights):
    cfg, gold = load_case(name)
    ref, out = run_pair(model, dev, weights, cfg)
    core = out["_core"]
    B, R, S, H = cfg["B"], cfg["R"], cfg["S"], cfg["H"]
    # ---- indices: bit-exact against the oracle
    assert out["pixel_val"].device.type == "cpu"
    assert torch.equal(out["pixel_val"], ref["pixel_val"]), "pixel_val not bit-identical"
    assert torch.equal(core["coords"].cpu(), ref["coords"]), "Pluecker coords not bit-identical"
    sec = core["sec_grid"].cpu().view(B, 2, R, S, 2)                 # [b,v] = view-v samples seen from image 1-v
    assert torch.equal(sec[:, 1], ref["sec_grid"].view(B, 2, R, S, 2)[:, 0])
    assert torch.equal(sec[:, 0], ref["sec_grid"].view(B, 2, R, S, 2)[:, 1])
    assert torch.equal(core["pt"].cpu(), ref["pt"]), "closest points (float64 island) not bit-identical"
    assert torch.equal(out["valid_mask"].cpu(), ref["valid_mask"])
    for a, b in zip(tap_indices(out["pixel_val"], H), tap_indices(torch.from_numpy(gold["pixel_val"]), H)):
        assert (a != b).float().mean() <= 1e-4                        # vs the upstream reference itself
    # ---- values
    assert (out["rgb"].cpu() - ref["rgb"]).abs().max() <= RGB_TOL
    assert (out["rgb"].cpu() - torch.from_numpy(gold["rgb"])).abs().max() <= RGB_TOL
    assert (out["at_wt"].cpu() - ref["at_wt"]).abs().max() <= 2e-3
    assert (core["z_local"].cpu() - ref["z_local"].reshape(-1, 416)).abs().max() <= 5e-3
    assert abs(float(out["at_wt"].view(B, 2, R, S).sum(dim=(1, 3)).mean()) - 1.0) < 1e-5
    assert (out["depth_ray"].cpu() - ref["depth_ray"]).abs().max() <= 2e-2
    assert out["at_wt_max"].dtype == torch.int64 and out["at_wt_max"].shape == (2 * B, R, 1)
    for k in ("rel_pose_flip", "gt_rel_pose", "gt_rel_pose_flip"):
        assert (out[k].cpu() - ref[k]).abs().max() <= 1e-5
    for k in ("mask_c2", "matchability_cycle_mask"):
        assert (out[k].cpu() != ref[k]).float().mean() <= 2e-2
    assert set(ref_keys()) <= set(out.keys())


def ref_keys():
    return ["flow", "coords", "uv", "pixel_val", "at_wts", "at_wt", "at_wt_max", "matchability_cycle_mask", "mask_c2",
            "T_to_C1_pts", "T_to_C2_pts", "C2_pts_to_C1", "depth_ray", "valid_mask", "rgb", "z", "rel_pose",
            "rel_pose_flip", "gt_rel_pose", "gt_rel_pose_flip"]


def test_layer_by_layer_mode_matches_folded_mode(model, dev, weights):
    """RenderEngine(fold_value=False) evaluates query_encode_latent_2 / latent_value / key_map per sample exactly as
    the reference orders them; the default folded evaluation must agree with it and with the oracle."""
    cfg, gold = load_case("c1_val")
    ref, out_fold = run_pair(model, dev, weights, cfg)
    model._engine.fold_value = False
    try:
        _, out_plain = run_pair(model, dev, weights, cfg)
    finally:
        model._engine.fold_value = True
    assert torch.equal(out_plain["pixel_val"], out_fold["pixel_val"])
    assert (out_plain["rgb"] - out_fold["rgb"]).abs().max() <= 5e-4
    assert (out_plain["rgb"].cpu() - ref["rgb"]).abs().max() <= RGB_TOL
    assert (out_plain["at_wt"] - out_fold["at_wt"]).abs().max() <= 1e-3


def test_stage_intermediates(model, dev, weights):
    """inter fixture: gathers / encoder / attention stages against the oracle AND the upstream intermediates."""
    cfg, gold = load_case("inter")
    ref, out = run_pair(model, dev, weights, cfg)
    core = out["_core"]
    assert torch.equal(out["pixel_val"], ref["pixel_val"])
    assert (core["rgb_raw"].cpu() - torch.from_numpy(gold["rgb_raw"]).reshape(-1, 3)).abs().max() <= RGB_TOL
    gpt = torch.from_numpy(gold["pt"])      # near-parallel lines: |pt| up to ~1e3, error grows with magnitude
    assert ((core["pt"].cpu() - gpt).abs() / (1 + gpt.abs())).max() <= 2e-4


def test_gemm_f16_against_torch(dev):
    """cpn_gemm_f16 vs an fp32 torch matmul on the same fp16-rounded operands; ragged M, K tail of 32, both tiles."""
    from coponerf_amd._hip import call
    torch.manual_seed(0)
    s = torch.cuda.current_stream().cuda_stream
    for (M, N, K, ld) in [(1000, 832, 864, 896), (513, 416, 832, 832), (256, 128, 128, 128), (77, 128, 832, 832),
                          (4096, 832, 864, 896)]:
        A = torch.zeros(M, ld, device=dev, dtype=torch.float16)
        A[:, :K] = torch.randn(M, K, device=dev) * 0.5
        Wt = torch.zeros(N, ld, device=dev, dtype=torch.float16)
        Wt[:, :K] = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        ref = A[:, :K].float() @ Wt[:, :K].float().t() + bias
        for relu in (0, 1):
            for f32 in (0, 1):
                C = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32 if f32 else torch.float16)
                call("cpn_gemm_f16", A.data_ptr(), ld, Wt.data_ptr(), ld, bias.data_ptr(), C.data_ptr(), N, M, N, K,
                     relu, f32, s)
                want = ref.clamp_min(0) if relu else ref
                err = (C.float() - want).abs().max().item()
                assert err <= (2e-4 if f32 else 4e-3) * max(1.0, want.abs().max().item()), (M, N, K, relu, f32, err)


def test_linear_f32_against_torch(dev):
    from coponerf_amd._hip import call
    torch.manual_seed(1)
    s = torch.cuda.current_stream().cuda_stream
    for (M, N, K) in [(100, 128, 416), (64, 3, 128), (1000, 128, 32), (17, 128, 128)]:
        X = torch.randn(M, K, device=dev)
        Wt = torch.randn(N, K, device=dev) * 0.1
        b = torch.randn(N, device=dev)
        res = torch.randn(M, N, device=dev)
        Y = torch.empty(M, N, device=dev)
        call("cpn_linear_f32", X.data_ptr(), K, Wt.data_ptr(), K, b.data_ptr(), res.data_ptr(), N, Y.data_ptr(), N,
             M, N, K, 1, 1, s)
        want = torch.relu(torch.relu(X).double() @ Wt.double().t() + b.double() + res.double()).float()
        assert (Y - want).abs().max() <= 1e-4


def test_gather_against_grid_sample(dev):
    """cpn_gather_rows vs ATen grid_sample on fp16-rounded maps: border + zeros padding, huge coordinates."""
    from coponerf_amd._hip import call
    from oracle.render_ref import gather_levels
    torch.manual_seed(2)
    B, V, R, S, H = 1, 2, 8, 16, 64
    N = B * V
    z = [torch.randn(N, 256, H // 16, H // 16), torch.randn(N, 256, H // 8, H // 8),
         torch.randn(N, 256, H // 4, H // 4), torch.randn(N, 64, H, H)]
    z = [t.half().float() for t in z]
    pv = torch.rand(N, R, S, 2) * 2.4 - 1.2
    sg = torch.rand(N, R, S, 2) * 3 - 1.5
    sg[0, 0, 0] = torch.tensor([1e10, -1e10])
    sg[1, 0, 1] = torch.tensor([-1.0, 1.0])
    pe = torch.rand(N, R, S, 6)
    s = torch.cuda.current_stream().cuda_stream
    maps = []
    for t in z:
        n, c, h, w = t.shape
        d = torch.empty(n, h, w, c, dtype=torch.float16, device=dev)
        src = t.to(dev).contiguous()
        call("cpn_nchw_to_nhwc_f16", src.data_ptr(), d.data_ptr(), n, c, h, w, s)
        assert torch.equal(d.float().cpu(), t.permute(0, 2, 3, 1))
        maps.append(d)
    xin = torch.zeros(B * R * V * S * 2, 896, dtype=torch.float16, device=dev)
    pvd, sgd, ped = pv.to(dev), sg.to(dev), pe.to(dev)
    call("cpn_gather_rows", maps[0].data_ptr(), maps[1].data_ptr(), maps[2].data_ptr(), maps[3].data_ptr(), H, H,
         pvd.data_ptr(), sgd.data_ptr(), ped.data_ptr(), B, V, R, S, 0, B * R, xin.data_ptr(), s)
    got = xin.float().cpu().view(B, R, V, S, 2, 896)
    prim = gather_levels(z, pv, "border").view(B, V, R, S, 832)
    # secondary of view v: OTHER image sampled at view v's reprojected coordinates
    z_swapped = [t.view(B, V, *t.shape[1:]).flip(1).reshape(t.shape) for t in z]
    sec = gather_levels(z_swapped, sg, "zeros").view(B, V, R, S, 832)
    for v in range(V):
        assert (got[:, :, v, :, 0, :832] - prim[:, v]).abs().max() <= 4e-3
        assert (got[:, :, v, :, 1, :832] - sec[:, v]).abs().max() <= 4e-3
        pe5 = pe.view(B, V, R, S, 6)
        assert (got[:, :, v, :, 0, 832:835] - pe5[:, v, :, :, 0:3]).abs().max() <= 1e-3
        assert (got[:, :, v, :, 1, 832:835] - pe5[:, v, :, :, 3:6]).abs().max() <= 1e-3
    assert got[..., 835:864].abs().max() == 0


def test_full_size_properties(model, dev, weights):
    """BASELINE config-2 sizes (256x256, S=64): size-independent properties instead of an oracle run."""
    H, S, R = 256, 64, 8192
    inp = to_device(syn.make_inputs(1, H, H, R, seed=21), dev)
    z, rel, flow = syn.make_latents(1, H, H, seed=22)
    z, rel, flow = to_device(z, dev), rel.to(dev), to_device(flow, dev)
    model.npoints = S
    with torch.no_grad():
        full = model(inp, z=z, rel_pose=rel, val=True, flow=flow)
        # rays are independent work units: rendering two halves reproduces the whole bit for bit
        halves = []
        for sl in (slice(0, R // 2), slice(R // 2, R)):
            part = {"context": inp["context"], "query": dict(inp["query"], uv=inp["query"]["uv"][:, :, sl].contiguous(),
                                                             rgb=inp["query"]["rgb"][:, :, sl])}
            halves.append(model(part, z=z, rel_pose=rel, val=True, flow=flow))
    assert torch.equal(torch.cat([h["rgb"] for h in halves], dim=2), full["rgb"])
    assert torch.equal(torch.cat([h["pixel_val"] for h in halves], dim=1), full["pixel_val"])
    w = full["at_wt"].view(1, 2, R, S)
    assert (w.sum(dim=(1, 3)) - 1).abs().max() < 1e-5 and float(w.min()) >= 0
    assert torch.isfinite(full["rgb"]).all()
    invalid = full["valid_mask"][..., 0] == 0
    assert (full["rgb"][:, 0][invalid] == 1).all()


@pytest.mark.parametrize("H,S,R,rig", [(256, 64, 192, "wide"), (512, 128, 96, "narrow")])
def test_other_baseline_configs_against_oracle(H, S, R, rig, model, dev, weights):
    """BASELINE configs 4 (ACID-like wide baseline, 256x256x64) and 5 (512x512, 128 samples per ray), render path:
    same bars as the fixture cases — indices bit-identical to the oracle, rgb within 1e-3."""
    from oracle import render_ref as orc
    inp = syn.make_inputs(1, H, H, R, seed=33, rig=rig)
    z, rel, flow = syn.make_latents(1, H, H, seed=34)
    with torch.no_grad():
        ref = orc.forward(inp, z, rel, flow, True, weights, npoints=S, keep=True)
        model.npoints = S
        out = model(to_device(inp, dev), z=to_device(z, dev), rel_pose=rel.to(dev), val=True, flow=to_device(flow, dev),
                    debug=True)
    assert torch.equal(out["pixel_val"], ref["pixel_val"])
    assert torch.equal(out["_core"]["pt"].cpu(), ref["pt"])
    assert (out["rgb"].cpu() - ref["rgb"]).abs().max() <= RGB_TOL
    assert (out["at_wt"].cpu() - ref["at_wt"]).abs().max() <= 2e-3
    assert torch.equal(out["valid_mask"].cpu(), ref["valid_mask"])


@pytest.mark.parametrize("B,H,S,R,val", [(3, 96, 24, 77, True), (2, 48, 40, 33, False), (1, 64, 7, 129, True)])
def test_ragged_shapes_against_oracle(B, H, S, R, val, model, dev, weights):
    """Sizes that are multiples of nothing convenient: odd ray counts, sample counts that are not powers of two (7, 24,
    40), batch 3, ray chunks that end in the middle of a batch element (chunk_rays = 50)."""
    from oracle import render_ref as orc
    inp = syn.make_inputs(B, H, H, R, seed=91)
    z, rel, flow = syn.make_latents(B, H, H, seed=92)
    old_chunk, old_s = model._engine.chunk_rays, model.npoints
    try:
        with torch.no_grad():
            ref = orc.forward(inp, z, rel, flow, val, weights, npoints=S, keep=True)
            model.npoints, model._engine.chunk_rays = S, 50
            out = model(to_device(inp, dev), z=to_device(z, dev), rel_pose=rel.to(dev), val=val,
                        flow=to_device(flow, dev), debug=True)
    finally:
        model._engine.chunk_rays, model.npoints = old_chunk, old_s
    assert torch.equal(out["pixel_val"], ref["pixel_val"])
    assert torch.equal(out["_core"]["pt"].cpu(), ref["pt"])
    assert (out["rgb"].cpu() - ref["rgb"]).abs().max() <= RGB_TOL
    assert (out["at_wt"].cpu() - ref["at_wt"]).abs().max() <= 2e-3
    assert torch.equal(out["valid_mask"].cpu(), ref["valid_mask"])


def test_missing_library_is_loud(monkeypatch):
    from coponerf_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", "/nonexistent/libcoponerf_hip.so")
    with pytest.raises(_hip.HipLibraryError):
        _hip.lib()
