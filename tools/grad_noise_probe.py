import torch, sys
sys.path.insert(0, '/root/repo')
from coponerf_amd import synthetic as syn, getz
from tests import step_case as sc
from tests.helpers import to_device
from coponerf_amd import CoPoNeRF
dev = torch.device("cuda:0")
m = CoPoNeRF.CoPoNeRF(n_view=2, npoints=sc.CFG["S"])
shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
m.load_state_dict(syn.make_full_weights(shapes), strict=True)
m = m.to(dev)
fx = sc.fixture("step_r4096.npz")
inp, gt = sc.inputs(4096)
inp, gt = to_device(inp, dev), gt.to(dev)
for mode in (True, False):
    getz.F16_TRUNK_BACKWARD = mode
    for it in range(5):
        m.zero_grad(set_to_none=True)
        out = m(inp, val=False)
        sum(sc.loss_terms("img", out, gt).values()).backward()
        rows, bad = sc.compare("img", {n: p.grad for n, p in m.named_parameters()}, fx, rel_l2=lambda n: 3e-2, rel_max=0.10)
        enc = [r for r in rows if r[5].startswith("encoder.")]
        print("f16" if mode else "f32", it, "worst", sc.report(rows, 1).strip()[:110], "| bad", len(bad))
