import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from coponerf_amd import synthetic as syn
from coponerf_amd.ufc_ops import HipOps
from oracle.ufc_ref import TorchOps
dev = torch.device("cuda:0")
for (B, cin, cout, n) in ((1, 8, 32, 16), (1, 32, 8, 16), (1, 4, 8, 8), (1, 4, 8, 4)):
    x = syn.normal((B, cin, n, n, n, n), seed=31 + n)
    wq, ws = syn.normal((cout, cin, 3, 3), seed=32) * 0.2, syn.normal((cout, cin, 3, 3), seed=33) * 0.2
    bq, bs = syn.normal((cout,), seed=34) * 0.1, syn.normal((cout,), seed=35) * 0.1
    gw, gb = torch.ones(cout), torch.zeros(cout)
    with torch.no_grad():
        want = TorchOps().conv4d_gn_relu(x, wq, bq, ws, bs, 3, 1, 1, gw, gb, 1e-5)
        got = HipOps().conv4d_gn_relu(*[t.to(dev) for t in (x, wq, bq, ws, bs)], 3, 1, 1, gw.to(dev), gb.to(dev), 1e-5).cpu()
    d = (got - want).abs()
    idx = torch.nonzero(d > 1e-4)
    print((B, cin, cout, n), "max", float(d.max()), "bad", idx.shape[0], "of", d.numel())
    if idx.shape[0]:
        import collections
        for dim in range(1, 6):
            print(" dim", dim, sorted(collections.Counter(idx[:, dim].tolist()).items())[:20])
