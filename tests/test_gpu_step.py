"""The whole training step on the MI355X — get_z + render + loss + backward through the HIP path, exactly what
coponerf_amd.train_step.TrainStep differentiates — against the upstream reference's own gradients of the same step
(tests/golden/step.npz, made by tests/golden/make_golden_step.py from /root/reference wrapper.py:104-138): every
parameter with a gradient (encoder, UFC, conv_map, pose head, render layers), for the image loss, the reference's
image + cycle + pose loss, and the unmasked cycle / depth composition that differentiates through the auxiliary outputs."""
import pytest
import torch

from coponerf_amd import synthetic as syn
from tests import step_case as sc
from tests.helpers import to_device

pytestmark = pytest.mark.gpu

# fp16 activations in the per-sample MLPs (fp32 upstream): with 256 rays a handful of ReLU masks of the per-ray layers
# differ between the two forwards, and everything upstream of `z` inherits that.  Measured on MI355X (the test prints the
# table): worst tensor 1.3e-2 relative L2 (trunk BatchNorm parameters), worst single entry 7.3e-2 of its tensor's max.
# The per-ray decoder `phi` sees 256 rows only: one flipped ReLU mask there is 1/256 of a layer's gradient (3.1e-2 on
# phi.blocks.2.fc_0.weight in the `aux` case; tests/test_gpu_train.py allows 0.12 for the same reason at ~100 rays).
REL_L2 = lambda name: 8e-2 if name.startswith("phi.") else 3e-2
REL_MAX = 0.15


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev):
    from coponerf_amd import CoPoNeRF
    m = CoPoNeRF.CoPoNeRF(n_view=2, npoints=sc.CFG["S"])
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(syn.make_full_weights(shapes), strict=True)
    return m.to(dev)


@pytest.mark.parametrize("tag", sc.TAGS)
def test_step_gradients_match_reference(tag, model, dev):
    fx = sc.fixture()
    assert model.training
    inp, gt = sc.inputs()
    inp, gt = to_device(inp, dev), gt.to(dev)
    model.zero_grad(set_to_none=True)
    out = model(inp, val=False)                       # get_z inside, as wrapper.py:107 calls it
    assert (out["rgb"].detach().cpu() - torch.from_numpy(fx[f"{tag}|rgb"])).abs().max() <= 1e-3
    assert (out["rel_pose"].detach().cpu() - torch.from_numpy(fx[f"{tag}|rel_pose"])).abs().max() <= 2e-5
    terms = sc.loss_terms(tag, out, gt)
    for name, t in terms.items():
        want = float(fx[f"{tag}|loss|{name}"])
        assert abs(float(t.detach()) - want) <= 1e-3 * max(1.0, abs(want)), (name, float(t.detach()), want)
    sum(terms.values()).backward()
    rows, bad = sc.compare(tag, {n: p.grad for n, p in model.named_parameters()}, fx, rel_l2=REL_L2, rel_max=REL_MAX)
    print(f"[{tag}] worst tensors vs the upstream gradients:\n" + sc.report(rows, 16))
    assert not bad, sc.report(bad, 40)


RENDER_LAYERS = ("query_encode_latent", "query_encode_latent_2", "latent_value", "key_map", "key_map_2", "query_embed", "query_embed_2",
                 "query_repeat_embed", "query_repeat_embed_2", "encode_latent")
# measured over four runs (the backward has atomics: the last digits move): per-ray decoder phi <= 2.8e-3 relative L2 (3.1e-2 at
# 256 rays: that was ReLU-mask flips of single rays), the per-sample render layers <= 1.4e-3, everything upstream of z (UFC,
# conv_map, the trunk with its batch-statistics BatchNorm) <= 1.9e-2 with the trunk's BatchNorm parameters on top, as at 256 rays
REL_L2_4096 = lambda name: 6e-3 if (name.startswith("phi.") or name.split(".")[0] in RENDER_LAYERS) else 3e-2
REL_MAX_4096 = 0.10


@pytest.mark.parametrize("tag", ("img", "aux"))
def test_step_gradients_match_reference_at_4096_rays(tag, model, dev):
    """The same step at the ray count BASELINE configs[2] trains with per pair (4 096 query rays; tests/golden/step_r4096.npz:
    the upstream reference's gradients, made by make_golden_step.py --rays 4096): with 16 x the rays of the case above a
    flipped ReLU mask of a per-ray layer is 1/4096 of a gradient, and what remains is what the fp16 activation-gradient path
    itself costs: the render layers and the decoder agree with the fp32 reference to a few 1e-3 (bar 6e-3), the tensors
    upstream of z stay at the 1e-2 of the small case (bar 3e-2).  One pair: four (configs[2]'s batch) need more host memory for
    the reference's own backward than the build container has."""
    fx = sc.fixture("step_r4096.npz")
    assert int(fx["rays"]) == 4096
    inp, gt = sc.inputs(4096)
    inp, gt = to_device(inp, dev), gt.to(dev)
    model.zero_grad(set_to_none=True)
    out = model(inp, val=False)
    assert (out["rgb"].detach().cpu() - torch.from_numpy(fx[f"{tag}|rgb"])).abs().max() <= 1e-3
    assert (out["rel_pose"].detach().cpu() - torch.from_numpy(fx[f"{tag}|rel_pose"])).abs().max() <= 2e-5
    terms = sc.loss_terms(tag, out, gt)
    for name, t in terms.items():
        want = float(fx[f"{tag}|loss|{name}"])
        assert abs(float(t.detach()) - want) <= 1e-3 * max(1.0, abs(want)), (name, float(t.detach()), want)
    sum(terms.values()).backward()
    rows, bad = sc.compare(tag, {n: p.grad for n, p in model.named_parameters()}, fx, rel_l2=REL_L2_4096, rel_max=REL_MAX_4096)
    print(f"[{tag}, 4096 rays] worst tensors vs the upstream gradients:\n" + sc.report(rows, 16))
    print(f"[{tag}, 4096 rays] worst per-ray decoder / render-layer tensors:\n" +
          sc.report([r for r in rows if r[5].startswith("phi.")][:3] + [r for r in rows if r[5].split(".")[0] in RENDER_LAYERS][:3], 6))
    assert not bad, sc.report(bad, 40)


def test_trunk_fp16_backward_against_fp32_backward(model, dev):
    """The trunk's convolution backward on fp16 operands (getz._ConvF16BwdFn) against the library's fp32 backward of the same
    step, with the run-to-run spread of the fp32 backward itself beside it (the render backward accumulates with atomics, so
    two fp32 passes already differ upstream of z), and the fp16 results dx / dw two decades below fp16's largest number."""
    from coponerf_amd import getz
    inp, gt = sc.inputs(4096)
    inp, gt = to_device(inp, dev), gt.to(dev)
    getz._TRUNK_BWD_TARGET[0] = 4.0            # (process-wide; tests that force skipped steps back it off)

    def trunk_grads(f16: bool):
        old = getz.F16_TRUNK_BACKWARD
        getz.F16_TRUNK_BACKWARD = f16
        try:
            model.zero_grad(set_to_none=True)
            out = model(inp, val=False)
            sum(sc.loss_terms("img", out, gt).values()).backward()
            # the convolution weights: what the fp16 kernels produce directly.  (The BatchNorm parameters' gradients are sums
            # with heavy cancellation - layer2.3.bn1.bias moves by 1e-2 ... 3e-2 between two runs of EITHER backward's
            # neighbours - and are held to the upstream gradients by the tests above.)
            return {n: p.grad.detach().clone() for n, p in model.named_parameters()
                    if n.startswith("encoder.") and p.grad is not None and p.dim() == 4}
        finally:
            getz.F16_TRUNK_BACKWARD = old

    def apart(a, b):
        rows = sorted((((a[n] - g).norm() / g.norm().clamp_min(1e-30)).item(), n) for n, g in b.items())
        return rows[-1]

    ref = trunk_grads(False)
    again = trunk_grads(False)
    getz.F16_BWD_TRACE = []
    try:
        half = trunk_grads(True)
        trace = getz.F16_BWD_TRACE
    finally:
        getz.F16_BWD_TRACE = None
    assert len(trace) >= 30                                   # the 3x3 / 1x1 layers of the trunk took the fp16 path
    top = max(float(t) for _, dx, dw in trace for t in (dx, dw) if t is not None)
    spread, worst = apart(again, ref), apart(half, ref)
    print(f"largest fp16 entry of the trunk backward: {top:.1f}")
    print(f"trunk gradients: fp32 backward run to run {spread[0]:.2e} ({spread[1]}), fp16-operand backward vs fp32 {worst[0]:.2e} ({worst[1]})")
    assert top < 65504 / 64
    assert worst[0] < max(5e-3, 3 * spread[0])
