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


def test_pending_host_tensor_waits_at_first_use_of_values():
    """render.PendingHostTensor: `.cpu()` / metadata do not wait for the copy event, the first value access does, once."""
    from coponerf_amd.render import PendingHostTensor

    class FakeEvent:
        def __init__(self):
            self.waits = 0

        def synchronize(self):
            self.waits += 1

    ev = FakeEvent()
    host = torch.arange(24, dtype=torch.float32).view(2, 3, 2, 2)
    t = PendingHostTensor.wrap(host, ev)
    assert isinstance(t, torch.Tensor) and t.device.type == "cpu"
    assert t.shape == (2, 3, 2, 2) and t.dtype == torch.float32 and t.size(1) == 3 and t.dim() == 4
    assert t.cpu() is t                                   # the caller's `.cpu()` (test.py:194) neither copies nor waits
    assert ev.waits == 0
    other = PendingHostTensor.wrap(host.clone() + 100, FakeEvent())
    joined = torch.cat([t, other], dim=-3)                # the caller's join (test.py:207)
    assert ev.waits == 1 and type(joined) is torch.Tensor and joined.shape == (2, 6, 2, 2)
    assert torch.equal(joined, torch.cat([host, host + 100], dim=1))
    mixed = torch.cat([PendingHostTensor.wrap(host.clone(), FakeEvent()), host + 1], dim=0)     # not all pending: plain cat
    assert type(mixed) is torch.Tensor and torch.equal(mixed, torch.cat([host, host + 1], dim=0))
    assert torch.equal(joined[:, :3], host) and torch.equal(joined[:, 3:], host + 100)
    assert float(t.sum()) == float(host.sum()) and ev.waits == 1          # waited once, then a plain pinned tensor
    assert torch.equal(torch.from_numpy(t.numpy()), host)
    t2 = PendingHostTensor.wrap(host.clone(), FakeEvent())
    assert torch.equal(t2[0], host[0]) and t2._cpn_ready is None


def test_uv_rows_reads_ray_chunks_in_place():
    from coponerf_amd.render import _uv_rows
    full = torch.arange(2 * 1 * 10 * 2, dtype=torch.float32).view(2, 1, 10, 2)
    chunk = torch.chunk(full, 3, dim=2)[1]                # rays 4..7 of both pairs: batch stride 20 floats
    u, stride = _uv_rows(chunk, 2, 4)
    assert stride == 20 and u.data_ptr() == chunk.data_ptr()
    assert torch.equal(u, chunk[:, 0])
    u1, s1 = _uv_rows(full[:1].double(), 1, 10)           # other dtypes are converted
    assert s1 == 20 and u1.dtype == torch.float32 and u1.is_contiguous()
    weird = full.permute(0, 1, 3, 2).contiguous().permute(0, 1, 3, 2)   # (x, y) not adjacent in memory
    u2, s2 = _uv_rows(weird, 2, 10)
    assert u2.is_contiguous() and s2 == 20 and torch.equal(u2, full[:, 0])


def test_guard_on_device_matches_the_host_guard():
    """dist.guard_on_device (flag and clip coefficient as tensors, no host read) against dist.guard_and_clip: same flag, the
    coefficient it would have applied; a NaN makes the flag 0 and leaves the coefficient finite; no gradients at all -> flag 1."""
    import torch
    from coponerf_amd import dist as cd
    torch.manual_seed(0)
    ps = [torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(5)), torch.nn.Parameter(torch.randn(2))]
    for p in ps[:2]:
        p.grad = torch.randn_like(p) * 3
    ok, total, coef = cd.guard_on_device(ps, 0.5)
    ref = [p.grad.clone() for p in ps[:2]]
    fin, tot = cd.guard_and_clip(ps, 0.5)
    assert fin and float(ok) == 1.0 and abs(float(total) - float(tot)) < 1e-12
    for p, g in zip(ps[:2], ref):
        assert torch.allclose(p.grad, g * coef)
    ok2, _, none = cd.guard_on_device(ps, 0.0)
    assert float(ok2) == 1.0 and none is None
    ps[1].grad[2] = float("nan")
    ok3, _, coef3 = cd.guard_on_device(ps, 0.5)
    assert float(ok3) == 0.0 and bool(torch.isfinite(coef3))
    for p in ps:
        p.grad = None
    ok4, tot4, c4 = cd.guard_on_device(ps, 0.5)
    assert float(ok4) == 1.0 and float(tot4) == 0.0 and c4 is None


def test_flow_product_cache_is_lru_weak_and_version_checked():
    """aux_outputs.flow_products: the products of a flow pair are computed once per pair (a full-image render is 18 forward()
    calls on the same flows); the cache holds the three most recently used pairs, matches on tensor IDENTITY and version, and
    keeps the flow tensors only weakly (a captured get_z graph's output outliving its graph crashed the interpreter at exit)."""
    import gc
    import weakref

    import torch

    from coponerf_amd import aux_outputs as ao

    ao._FLOW_CACHE.clear()
    mk = lambda seed: [torch.randn(1, 2, 64, 64, generator=torch.Generator().manual_seed(seed + i)) for i in range(4)]
    pairs = [mk(10 * k) for k in range(4)]
    (m0, u0), hit = ao.flow_products(pairs[0], 256)
    assert not hit and m0.shape == (1, 256, 256) and u0.shape == (1, 2, 256, 256)
    (m0b, u0b), hit = ao.flow_products(pairs[0], 256)
    assert hit and m0b is m0 and u0b is u0
    # equal values in other tensors are another pair
    clone = [t.clone() for t in pairs[0]]
    (_, _), hit = ao.flow_products(clone, 256)
    assert not hit
    # an in-place change bumps the version: recomputed
    pairs[0][1].mul_(0.5)
    (m0c, _), hit = ao.flow_products(pairs[0], 256)
    assert not hit and m0c is not m0
    # three entries, least recently used out
    ao._FLOW_CACHE.clear()
    for p in pairs[:3]:
        assert not ao.flow_products(p, 256)[1]
    assert ao.flow_products(pairs[0], 256)[1]                      # 0 is now the most recent, 1 the oldest
    assert not ao.flow_products(pairs[3], 256)[1]                  # evicts 1
    assert ao.flow_products(pairs[0], 256)[1] and ao.flow_products(pairs[2], 256)[1] and ao.flow_products(pairs[3], 256)[1]
    assert not ao.flow_products(pairs[1], 256)[1]
    # the cache does not keep a pair's flow tensors alive
    ref = weakref.ref(pairs[1][0])
    ao.flow_products(pairs[1], 256)
    pairs[1] = None
    gc.collect()
    assert ref() is None
    assert not ao.flow_products(mk(99), 256)[1]                    # a dead entry in the list is skipped, not dereferenced into a match
    ao._FLOW_CACHE.clear()
