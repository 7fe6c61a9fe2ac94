"""Host-side logic that needs no GPU: the camera block against the oracle's pose algebra, synthetic generators."""
import torch

from coponerf_amd import _hip, synthetic as syn
from coponerf_amd.render import build_camera_block
from oracle import render_ref as orc


def test_camera_block_matches_oracle_pose_algebra():
    for val in (False, True):
        inp = syn.make_inputs(3, 64, 64, 16, seed=4)
        _, rel, _ = syn.make_latents(3, 64, 64)
        ctx, qry = inp["context"], inp["query"]
        cam, Tq = build_camera_block(ctx["cam2world"], ctx["intrinsics"], qry["cam2world"], qry["intrinsics"], rel, val, 64)
        oTq, oM, oA1, oA2 = orc.pose_algebra(ctx["cam2world"], qry["cam2world"], rel, val)
        cam = cam.view(3, 2, _hip.CAM_STRIDE)
        assert torch.equal(Tq, oTq)
        assert torch.equal(cam[:, :, _hip.CAM_TQ:_hip.CAM_TQ + 16].reshape(3, 2, 4, 4), oTq)
        assert torch.equal(cam[:, :, _hip.CAM_M:_hip.CAM_M + 16].reshape(3, 2, 4, 4), oM)
        assert torch.equal(cam[:, 0, _hip.CAM_AOWN:_hip.CAM_AOWN + 16].reshape(3, 4, 4), oA1[:, 0])
        assert torch.equal(cam[:, 1, _hip.CAM_AOWN:_hip.CAM_AOWN + 16].reshape(3, 4, 4), oA2[:, 1])
        assert torch.equal(cam[:, 0, _hip.CAM_AOTH:_hip.CAM_AOTH + 16].reshape(3, 4, 4), oA2[:, 0])
        assert torch.equal(cam[:, 1, _hip.CAM_AOTH:_hip.CAM_AOTH + 16].reshape(3, 4, 4), oA1[:, 1])
        Kn = ctx["intrinsics"][:, :, :3, :3].clone()
        Kn[:, :, :2] = Kn[:, :, :2] / 64
        assert torch.equal(cam[:, :, _hip.CAM_KN:_hip.CAM_KN + 9].reshape(3, 2, 3, 3), Kn)
        assert torch.equal(cam[:, 0, _hip.CAM_KO:_hip.CAM_KO + 4], cam[:, 1, _hip.CAM_KC:_hip.CAM_KC + 4])


def test_synthetic_generators_are_deterministic():
    a = syn.make_inputs(2, 64, 64, 32, seed=5)
    b = syn.make_inputs(2, 64, 64, 32, seed=5)
    assert torch.equal(a["query"]["uv"], b["query"]["uv"]) and torch.equal(a["context"]["rgb"], b["context"]["rgb"])
    assert not torch.equal(a["query"]["uv"][0], a["query"]["uv"][1])
    n = syn.normal((100000,), 3)
    assert abs(float(n.mean())) < 0.02 and abs(float(n.std()) - 1) < 0.02
    u = syn.uniform((100000,), 3, -1, 1)
    assert float(u.min()) >= -1 and float(u.max()) < 1 and abs(float(u.mean())) < 0.02
    full = syn.make_inputs(1, 16, 16, 0, full_image=True)["query"]["uv"]
    assert full.shape == (1, 1, 256, 2) and full[0, 0, 17].tolist() == [1.0, 1.0]


def test_forward_requires_hip_device():
    from coponerf_amd import CoPoNeRF
    import pytest
    m = CoPoNeRF.CoPoNeRF(n_view=2).eval()
    inp = syn.make_inputs(1, 64, 64, 8)
    z, rel, flow = syn.make_latents(1, 64, 64)
    with torch.no_grad(), pytest.raises(RuntimeError, match="HIP device"):
        m(inp, z=z, rel_pose=rel, val=True, flow=flow)
