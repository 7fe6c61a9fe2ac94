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
